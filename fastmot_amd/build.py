"""Builds libfastmot_hip.so in-tree with hipcc for gfx950 (MI355X).

    python -m fastmot_amd.build [--force]

hipcc cross-compiles without a GPU; the .so is git-ignored but travels with gpurun snapshots.
"""
import subprocess
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / 'csrc'
OUT = PKG / 'libfastmot_hip.so'
HIPCC = '/opt/rocm/bin/hipcc'
# -ffp-contract=off: association/Kalman decisions compare doubles produced with separate IEEE
# mul/add like NumPy (see assoc.hip header); conv kernels use explicit MFMA/fma intrinsics.
import os
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-ffp-contract=off',
         '-Wno-unused-result'] + os.environ.get('FASTMOT_EXTRA_HIPCC_FLAGS', '').split()   # profiling builds (-DFM_*_TIMING)


# Per-file flags.  -fno-slp-vectorize: no packed-fp32 VALU arithmetic (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 pairs
# formed by the SLP vectoriser) in the KLT kernels -- round 3 traced the LK results that differed under load to such a
# chain mis-executing in lanes 48..63 while VALU-heavy wavefronts of another kernel share the CU (DESIGN 5b,
# csrc/diag.hip is the stand-alone reproducer and is built the same way so that only its hand-written chain is packed).
FILE_FLAGS = {'flow.hip': ['-fno-slp-vectorize'], 'diag.hip': ['-fno-slp-vectorize']}


def sources():
    return sorted(CSRC.glob('*.hip'))


FLAGS_STAMP = PKG / 'build' / 'flags.txt'      # the flags the in-tree library was built with (profiling builds differ)


def needs_build():
    if not OUT.exists():
        return True
    if not FLAGS_STAMP.exists() or FLAGS_STAMP.read_text() != ' '.join(FLAGS):
        return True
    t = OUT.stat().st_mtime
    deps = list(CSRC.glob('*')) + [PKG.parent / 'include' / 'fastmot_hip.h', Path(__file__)]
    return any(p.stat().st_mtime > t for p in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return OUT
    if FLAGS_STAMP.exists() and FLAGS_STAMP.read_text() != ' '.join(FLAGS):
        force = True                                  # every object file must be rebuilt with the current flags
    objs = []
    procs = []
    bdir = PKG / 'build'
    bdir.mkdir(exist_ok=True)
    for src in sources():
        obj = bdir / (src.stem + '.o')
        objs.append(obj)
        if not force and obj.exists() and obj.stat().st_mtime > max(
                src.stat().st_mtime, *(h.stat().st_mtime for h in CSRC.glob('*.h')),
                (PKG.parent / 'include' / 'fastmot_hip.h').stat().st_mtime):
            continue
        cmd = [HIPCC] + [f for f in FLAGS if f != '-shared'] + FILE_FLAGS.get(src.name, []) + ['-c', str(src), '-o', str(obj)]
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f'hipcc failed on {src}:\n{out.decode()}')
        if verbose and out.strip():
            print(out.decode())
    cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', str(OUT)] + [str(o) for o in objs]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if res.returncode != 0:
        raise RuntimeError(f'link failed:\n{res.stdout.decode()}')
    FLAGS_STAMP.write_text(' '.join(FLAGS))
    if verbose:
        print(f'built {OUT}')
    return OUT


if __name__ == '__main__':
    build(force='--force' in sys.argv)
