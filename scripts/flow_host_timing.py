"""Where the KLT + Kalman job of a step (the native worker, fm_track_predict_async) spends its HOST time inside bench.py's
config[1] pipeline: the library's own lap counters of fm_flow_predict (begin / prepare / LK incl. its waits / estimate) and
their sub-stages (FASTMOT_FLOW_TIMING_VERBOSE).

    python scripts/flow_host_timing.py [--steps 300]"""
import argparse
import ctypes as C
import os
import sys
import time

os.environ.setdefault('FASTMOT_FLOW_TIMING_VERBOSE', '1')
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=300)
    args = ap.parse_args()
    from fastmot_amd import Track, models
    models.allow_random_weights()
    from fastmot_amd.runtime import get_context
    from synthetic import SyntheticVideo
    ctx = get_context()
    RING = bench.RING
    cfg = bench.CONFIGS[1]
    size = cfg['size']
    ctx.frame_configure(size[0], size[1], RING)
    host_frames = ctx.pinned_frames(RING)
    video = SyntheticVideo(size, n_ids=cfg['n_dets'], n_frames=RING, seed=100)
    for i, fr in enumerate(video.frames):
        host_frames[i] = fr
    frames = [host_frames[i] for i in range(RING)]
    mot = bench.build_mot(cfg, video)
    Track._count = 0
    mot.reset(1 / 30.)

    def run(n, start):
        for s in range(start, start + n):
            i = bench.ping_pong(s, RING)
            mot.detector._frame_idx = i
            nxt = frames[bench.ping_pong(s + 1, RING)] if s + 1 < start + n else None
            mot.step(frames[i], next_frame=nxt)
    run(250, 0)
    ctx.synchronize()
    out = (C.c_double * 5)()
    ctx.lib.fm_flow_timing(out, C.c_int(1))
    t0 = time.perf_counter()
    run(args.steps, 250)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    ctx.lib.fm_flow_timing(out, C.c_int(0))
    n = max(out[4], 1.0)
    print(f'{args.steps / dt:.1f} frames/s; fm_flow_predict per call (ms): begin {out[0] / n:.3f}  prepare {out[1] / n:.3f}  '
          f'lk {out[2] / n:.3f}  estimate {out[3] / n:.3f}  (calls {int(out[4])})')


if __name__ == '__main__':
    main()
