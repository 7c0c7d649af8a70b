"""Registers / LDS / scratch per kernel of one source of the library, from the gfx950 code object's metadata:
    python scripts/kernel_regs.py convd.hip [extra hipcc flags]
(the numbers that decide how many workgroups share a CU; hipcc cross-compiles without a GPU)."""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
LLVM = '/opt/rocm/lib/llvm/bin'


def main():
    src = ROOT / 'fastmot_amd' / 'csrc' / sys.argv[1]
    sys.path.insert(0, str(ROOT))
    from fastmot_amd.build import FLAGS, FILE_FLAGS
    with tempfile.TemporaryDirectory() as td:
        obj, co = f'{td}/d.o', f'{td}/d.co'
        flags = [f for f in FLAGS if f not in ('-shared', '-fPIC')] + FILE_FLAGS.get(src.name, []) + sys.argv[2:]
        subprocess.run(['/opt/rocm/bin/hipcc'] + flags + ['--cuda-device-only', '-c', str(src), '-o', obj], check=True)
        subprocess.run([f'{LLVM}/clang-offload-bundler', '--unbundle', '--type=o', f'--input={obj}',
                        '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', f'--output={co}'], check=True)
        notes = subprocess.run([f'{LLVM}/llvm-readelf', '--notes', co], capture_output=True, text=True).stdout
    rows = []
    for blk in notes.split('- .agpr_count:')[1:]:
        get = lambda k: (re.search(rf'\.{k}:\s+(\S+)', blk) or [None, '?'])[1]
        name = get('name')
        dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
        dem = re.sub(r'\(anonymous namespace\)::', '', dem)
        dem = dem[:dem.find('(')] if '(' in dem else dem
        rows.append((dem.replace('void ', ''), blk.split()[0], get('vgpr_count'), get('sgpr_count'), get('vgpr_spill_count'),
                     get('group_segment_fixed_size'), get('private_segment_fixed_size'), get('max_flat_workgroup_size')))
    print(f'{"kernel":<60} {"agpr":>5} {"vgpr":>5} {"sgpr":>5} {"spill":>5} {"lds":>7} {"scratch":>7} {"wg":>5}')
    for r in sorted(rows):
        print(f'{r[0][:60]:<60} {r[1]:>5} {r[2]:>5} {r[3]:>5} {r[4]:>5} {r[5]:>7} {r[6]:>7} {r[7]:>5}')


if __name__ == '__main__':
    main()
