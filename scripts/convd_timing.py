"""Profiling build only (scripts/build_timing_lib.sh convd.hip -DFM_CONVD_TIMING; run with
FASTMOT_LIB_PATH=fastmot_amd/libfastmot_hip_timing.so): where a workgroup of convd.hip spends its cycles.  For a few
(layer, configuration) pairs: cycle stamps of wave 0 of workgroup 0 and of a later workgroup -- per K step the wait
for the step's DMA (vmcnt), the barrier, the issue of the next step's DMA, the fragment reads + MFMAs -- and the
same under ablations (no DMA inside the loop / no MFMA phase / no output phase).

    python scripts/convd_timing.py > gpurun_out/convd_timing.txt"""
import os
os.environ.setdefault('FASTMOT_RANDOM_WEIGHTS', '1')
os.environ['FASTMOT_GRAPHS'] = '0'
import ctypes as C
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fastmot_amd.runtime import get_context
from fastmot_amd.engine import HipNet, NET_DETECTOR
from fastmot_amd.models.graph import Graph, RandomWeights

CASES = [
    # (cin, cout, k, stride, h, w), (bm, bn, kg, ns, loader waves)
    ((128, 256, 3, 1, 76, 76), (128, 128, 1, 4, 1)),
    ((128, 256, 3, 1, 76, 76), (128, 64, 2, 3, 1)),
    ((128, 256, 3, 1, 76, 76), (64, 64, 1, 4, 0)),
    ((256, 512, 3, 1, 38, 38), (64, 64, 2, 4, 1)),
    ((256, 256, 1, 1, 76, 76), (64, 64, 1, 4, 0)),
    ((128, 128, 3, 1, 160, 160), (128, 128, 1, 4, 1)),
]


def stamps(ctx, abl):
    st = (C.c_longlong * 528)()
    ctx.lib.fm_debug_convd_stamps(st, C.c_int(abl))
    return np.array(list(st)).reshape(2, 264)


def main():
    ctx = get_context()
    for shape, cfg in CASES:
        cin, cout, k, stride, h, w = shape
        g = Graph(RandomWeights(seed=1), (h, w), cin)
        g.convs_max_pixels = 0
        y = g.conv('c', g.input, cout, k, stride, 'leaky')
        ctx.set_option('convd_cfg', cfg[0] | cfg[1] << 8 | cfg[2] << 16 | cfg[3] << 20 | cfg[4] << 24)
        net = HipNet(ctx, NET_DETECTOR, g, 1)
        net.write(g.input, np.random.default_rng(0).normal(0, 1, (1, h, w, cin)).astype(np.float16))
        print(f'## k{k}s{stride} {h}x{w}x{cin} -> {cout}, forced {cfg}')
        for abl, tag in ((0, 'full'), (1, 'no DMA in the loop'), (2, 'no reads / MFMAs'), (4, 'no output phase'), (3, 'neither DMA nor MFMA')):
            stamps(ctx, abl)                                   # sets the ablation for the following launches
            for _ in range(3):
                net.run(1)
            ctx.synchronize()
            ev = float(np.mean(net.profile_layers(1, 20))) * 1e3
            st = stamps(ctx, 0)
            print(f'   {tag:<22} HIP-event time of the launch {ev:.2f} us')
            for slot, name in ((0, 'wg 0'), (1, 'wg 43')):
                s = st[slot]
                if s[5] <= s[0]:
                    continue
                per = int(s[6])
                it = s[8:8 + 4 * min(per, 64)].reshape(-1, 4)
                wait = it[1:, 0] - it[:-1, 3]                  # end of compute of the previous step -> vmcnt satisfied
                bar = it[:, 1] - it[:, 0]
                iss = it[:, 2] - it[:, 1]
                comp = it[:, 3] - it[:, 2]
                step = np.diff(it[:, 0])
                print(f'   {tag:<22} {name:<6} cfg {int(s[7])} steps {per}: prologue {s[1] - s[0]} + first issues {s[2] - s[1]}'
                      f' | loop {s[3] - s[2]} ({(s[3] - s[2]) / max(per, 1):.0f} / step) | reduce+transpose {s[4] - s[3]} | output {s[5] - s[4]}'
                      f' | total {s[5] - s[0]} cycles in {(s[263] - s[262]) * 10} ns = {(s[5] - s[0]) / max((s[263] - s[262]) * 10, 1):.2f} GHz')
                if per > 1:
                    print(f'      per step (median): wait {np.median(wait):.0f}  barrier {np.median(bar):.0f}  issue {np.median(iss):.0f}'
                          f'  reads+mfma {np.median(comp):.0f}  period {np.median(step):.0f};  first step: wait-for-first-data {it[0, 0] - s[2]}')
                    print('      periods:', ' '.join(str(int(v)) for v in step[:24]))
        net.close()
    ctx.set_option('convd_cfg', 0)


if __name__ == '__main__':
    main()
