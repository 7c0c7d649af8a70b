"""Profiling build only (FASTMOT_EXTRA_HIPCC_FLAGS=-DFM_GFTT_TIMING): phase cycles of gftt_select_kernel's workgroups
on a frame of the benchmark clip (the stamps of the LAST prepare call are read).  usage: gftt_timing.py [config, default 1]"""
import os as _os
_os.environ.setdefault('FASTMOT_RANDOM_WEIGHTS', '1')
import sys, ctypes as C
sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import bench
from fastmot_amd import Track
from fastmot_amd.detector import DeviceFrame
from fastmot_amd.runtime import get_context
from synthetic import SyntheticVideo

CFG = bench.CONFIGS[int(sys.argv[1]) if len(sys.argv) > 1 else 1]
video = SyntheticVideo(CFG['size'], n_ids=CFG['n_dets'], n_frames=bench.RING, seed=100)
ctx = get_context()
ctx.frame_configure(CFG['size'][0], CFG['size'][1], bench.RING)
for i, fr in enumerate(video.frames):
    ctx.frame_ring_store(i, fr)
mot = bench.build_mot(CFG, video)
Track._count = 0
mot.reset(1 / 30.)
names = ['tile maxima', 'gather', 'sort', 'greedy', 'ellipse+out']
for s in range(12):
    mot.detector._frame_idx = s
    mot.step(DeviceFrame(s))
    if s < 8:
        continue
    st = (C.c_longlong * 512)()
    ctx.lib.fm_debug_gftt_stamps(st)
    rows = []
    for b in range(64):
        v = [st[b * 8 + i] for i in range(8)]
        if v[5] > v[0] > 0:
            rows.append((v[5] - v[0], b, [v[i + 1] - v[i] for i in range(5)], v[6], v[7] >> 20, v[7] & 0xfffff))
    rows.sort(reverse=True)
    print(f'frame {s}: {len(rows)} workgroups with stamps; slowest three (cycles @2.4 GHz):')
    import numpy as np
    print('  phase means:', '  '.join(f'{nm} {int(np.mean([r[2][i] for r in rows]))}' for i, nm in enumerate(names)), ' mean pixels', int(np.mean([r[4] for r in rows])), ' mean candidates', int(np.mean([r[3] for r in rows])))
    for tot, b, ph, n, npx, nacc in rows[:3]:
        print(f'  wg {b:2d}: total {tot:7d}  ' + '  '.join(f'{nm} {p}' for nm, p in zip(names, ph)) + f'   candidates {n} pixels {npx} accepted {nacc}')
