#!/bin/bash
# HBM-side traffic of the detector's conv kernels (LDS-tiled, streamed, fused residual unit): two separate rocprofv3 PMC passes (FETCH_SIZE and
# WRITE_SIZE do not fit one pass on gfx950) over 6 graph replays of YOLOv4@608, summed per kernel.
# Run on the GPU box from the repo root:  bash scripts/collect_pmc.sh  -> gpurun_out/pmc_*.txt + gpurun_out/pmc_conv.json
set -e
export TMPDIR=/tmp
mkdir -p gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_$c
    rocprofv3 --pmc $c -d /tmp/pmc_$c -o f -- python scripts/trace_net.py 0 > /dev/null 2>&1
    python scripts/rocpd_pmc.py "$(find /tmp/pmc_$c -name '*.db' | head -1)" $c > gpurun_out/pmc_$c.txt
done
python - <<'PY'
import json
def load(c):
    for line in open(f'gpurun_out/pmc_{c}.txt'):
        if line.startswith('JSON '):
            return json.loads(line[5:])
f, w = load('FETCH_SIZE'), load('WRITE_SIZE')
conv = [k for k in f if k.startswith('conv_igemm_kernel') or 'resblock_kernel' in k or k.startswith('convs_kernel') or k.startswith('convs_halo_kernel') or k.startswith('convd_kernel') or 'stem2_kernel' in k or 'pair11_kernel' in k]
calls = sum(f[k]['calls'] for k in conv)
fetch_kb = sum(f[k]['total'] for k in conv)
write_kb = sum(w[k]['total'] for k in conv if k in w)
out = dict(kernel='conv_igemm_kernel + convd_kernel + convs_kernel + convs_halo_kernel + resblock_kernel + stem2_kernel + pair11_kernel (all template instances)', launches=calls, replays=6,
           fetch_size_kb_raw=fetch_kb, write_size_kb_raw=write_kb,
           correction='FETCH_SIZE x2 (gfx950: 128 B requests tallied at 64 B for 16 B/lane loads); WRITE_SIZE raw (uncalibrated)',
           traffic_bytes_per_launch=round((2 * fetch_kb + write_kb) * 1024 / calls),
           fetch_bytes_per_frame=round(2 * fetch_kb * 1024 / 6), write_bytes_per_frame=round(write_kb * 1024 / 6),
           splitk_reduce_fetch_kb_raw=f.get('splitk_reduce_kernel', {}).get('total'),
           splitk_reduce_write_kb_raw=w.get('splitk_reduce_kernel', {}).get('total'))
json.dump(out, open('gpurun_out/pmc_conv.json', 'w'), indent=1)
print(json.dumps(out))
PY
# MFMA utilisation / wait states of the same replays (two more PMC passes, four SQ counters each)
set +e
rm -rf /tmp/pmc_sq1 /tmp/pmc_sq2
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES -d /tmp/pmc_sq1 -o f -- python scripts/trace_net.py 0 > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS -d /tmp/pmc_sq2 -o f -- python scripts/trace_net.py 0 > /dev/null 2>&1
python scripts/rocpd_pmc_multi.py "$(find /tmp/pmc_sq1 -name '*.db' | head -1)" > gpurun_out/pmc_sq_yolo.txt 2>&1
python scripts/rocpd_pmc_multi.py "$(find /tmp/pmc_sq2 -name '*.db' | head -1)" >> gpurun_out/pmc_sq_yolo.txt 2>&1
# per layer: MFMA utilisation (busy cycles / duration x 2.4 GHz x 1024 SIMDs) and read amplification (2 x FETCH_SIZE / algorithmic
# reads); the JSON line is merged into pmc_conv.json (bench.py reads roofline.mfma_util / read_amplification from there)
python scripts/layer_pmc.py YOLOv4_608 /tmp/pmc_sq1 /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE > gpurun_out/pmc_layers.txt 2>&1
python - <<'PY'
import json
d = json.load(open('gpurun_out/pmc_conv.json'))
for line in open('gpurun_out/pmc_layers.txt'):
    if line.startswith('JSON '):
        L = json.loads(line[5:])
        d.update(mfma_util=L['mfma_util'], read_amplification=L['read_amplification'], per_kernel_class=L['per_kernel_class'],
                 mfma_busy_cycles=L['mfma_busy_cycles'], mfma_busy_expected=L['mfma_busy_expected'],
                 algorithmic_read_bytes=L['algorithmic_read_bytes'], clock_ghz_assumed=L['clock_ghz_assumed'])
json.dump(d, open('gpurun_out/pmc_conv.json', 'w'), indent=1)
print({k: d.get(k) for k in ('mfma_util', 'read_amplification')})
PY
