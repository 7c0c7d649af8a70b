"""Where a pipelined MOT.step spends its time, on the GPU's clock and the host's, WITH hipGraph replays (rocprofv3's
kernel trace cannot follow hipGraphLaunch in this pipeline, and launching the layers one by one changes the timing):
the library records one timed HIP event per stage boundary on the stage's own stream (fm_trace_start / fm_trace_read,
include/fastmot_hip.h) and this script adds the main thread's marks on the same time axis.

    python scripts/trace_pipeline.py [--config 1] [--steps 40] [--show 4]

prints the frame rate of the traced window, per-stage statistics (time since the start of the step the event fell into,
and durations) and the raw event list of the last `--show` steps.  Same workload as bench.py (it builds the pipeline with
bench.py's own functions)."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench  # noqa: E402

NAMES = {15: 'det: first layer (input pixels computed in it) done', 10: 'det: stream reaches the pass', 11: 'det: inputs ready, preprocessed', 12: 'det: network done',
         13: 'det: decode done', 20: 'post: begins', 21: 'post: ends', 30: 'copy(next): begins', 31: 'copy(next): ends',
         32: 'reid: begins', 33: 'reid: ends', 40: 'lk: begins', 41: 'lk: ends',
         14: 'det: preprocess begins', 22: 'post: sort done', 23: 'post: bit matrix done', 34: 'reid: crops done',
         35: 'reid: network done', 42: 'klt: pyramid begins', 43: 'klt: pyramid done', 44: 'klt: keypoint kernels begin',
         45: 'klt: keypoint kernels done', 46: 'klt: background FAST begins', 47: 'klt: background FAST done',
         50: 'kalman step begins', 51: 'kalman step done', 52: 'kalman update begins', 53: 'kalman update done',
         54: 'assoc: pairwise begins', 55: 'assoc: pairwise done', 56: 'assoc: stage cost begins', 57: 'assoc: stage cost done'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', type=int, default=1)
    ap.add_argument('--steps', type=int, default=40)
    ap.add_argument('--settle', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=40)
    ap.add_argument('--show', type=int, default=4)
    ap.add_argument('--nms-candidates', type=int, default=1500)
    args = ap.parse_args()
    cfg = bench.CONFIGS[args.config]
    from fastmot_amd import Track, models
    models.allow_random_weights()
    from fastmot_amd.runtime import get_context
    from synthetic import SyntheticVideo
    ctx = get_context()
    RING = bench.RING
    size = cfg['size']
    video = SyntheticVideo(size, n_ids=cfg['n_dets'], n_frames=RING, seed=100)
    ctx.frame_configure(size[0], size[1], RING)
    host_frames = ctx.pinned_frames(RING)
    for i, fr in enumerate(video.frames):
        host_frames[i] = fr
    frames = [host_frames[i] for i in range(RING)]
    mot = bench.build_mot(cfg, video, nms_candidates=args.nms_candidates)
    Track._count = 0
    mot.reset(1 / 30.)

    marks = []                                  # (step, name, host ns)

    def wrap(obj, attr, name):
        fn = getattr(obj, attr)

        def inner(*a, **k):
            r = fn(*a, **k)
            marks.append((cur[0], name, time.perf_counter_ns()))
            return r
        setattr(obj, attr, inner)
    cur = [0]
    wrap(mot.detector, 'postprocess', 'host: detections collected')
    wrap(mot.extractors[0], 'postprocess', 'host: embeddings collected')
    wrap(mot.extractors[0], 'extract_async', 'host: reid enqueued')
    wrap(mot.tracker, 'prepare_detections', 'host: prepare_detections done')
    wrap(mot.tracker, 'update_begin', 'host: update_begin done')
    import fastmot_amd.mot as mot_mod
    wrap(mot_mod._NativeFlowJob, 'result', 'host: klt+kalman job joined')
    wrap(mot.detector, 'prefetch', 'host: prefetch enqueued')

    def run(n, start, prefetch):
        for s in range(start, start + n):
            i = bench.ping_pong(s, RING)
            mot.detector._frame_idx = i
            nxt = frames[bench.ping_pong(s + 1, RING)] if prefetch and s + 1 < start + n else None
            cur[0] = s
            marks.append((s, 'host: step begins', time.perf_counter_ns()))
            mot.step(frames[i], next_frame=nxt)
            marks.append((s, 'host: step ends', time.perf_counter_ns()))

    run(args.settle, 0, False)
    pos = args.settle
    run(args.warmup, pos, True)
    pos += args.warmup
    # (the last warm-up step had no next frame: the first traced step runs its own detector pass)
    marks.clear()
    ctx.synchronize()
    zero_ns = ctx.trace_start(64 * (args.steps + 4))
    t0 = time.perf_counter()
    run(args.steps, pos, True)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    tags, ms = ctx.trace_read()
    print(f'# config {args.config}: {args.steps} traced steps, {args.steps / dt:.1f} frames/s ({dt / args.steps * 1e3:.3f} ms per step)')

    begins = np.array([(t - zero_ns) * 1e-6 for s, nm, t in marks if nm == 'host: step begins'])
    ev = [((t - zero_ns) * 1e-6, nm) for s, nm, t in marks] + [(float(m), NAMES.get(int(t), str(t))) for t, m in zip(tags, ms)]
    ev.sort()
    # statistics over the steady part (skip the first 5 steps): offset of every event from the start of the step it fell into
    stats = {}
    for t, nm in ev:
        k = int(np.searchsorted(begins, t, side='right')) - 1
        if k < 5 or k >= len(begins) - 1:
            continue
        stats.setdefault(nm, []).append(t - begins[k])
    print('# event                               n   offset from the step start it fell into: median  (p10 .. p90)  ms')
    for nm in sorted(stats, key=lambda x: np.median(stats[x])):
        v = np.array(stats[nm])
        print(f'{nm:34s} {len(v):4d}   {np.median(v):7.3f}  ({np.percentile(v, 10):7.3f} .. {np.percentile(v, 90):7.3f})')
    # durations between paired tags (in order of occurrence)
    def pairs(a, b):
        ta = [float(m) for t, m in zip(tags, ms) if t == a]
        tb = [float(m) for t, m in zip(tags, ms) if t == b]
        n = min(len(ta), len(tb))
        return np.array(tb[:n]) - np.array(ta[:n])
    print('# durations (ms): median  (p10 .. p90)')
    for a, b, nm in ((10, 11, 'det: waiting for inputs + preprocess'), (11, 12, 'det: network'), (11, 15, 'det: first layer incl. resize'), (12, 13, 'det: decode'),
                     (20, 21, 'post: sort + NMS + D2H'), (30, 31, 'copy of the next frame'), (32, 33, 'reid: crop + network'),
                     (40, 41, 'lk kernel')):
        d = pairs(a, b)[5:]
        if len(d):
            print(f'{nm:40s} {np.median(d):7.3f}  ({np.percentile(d, 10):7.3f} .. {np.percentile(d, 90):7.3f})')
    t10 = np.array([float(m) for t, m in zip(tags, ms) if t == 10])
    t13 = np.array([float(m) for t, m in zip(tags, ms) if t == 13])
    t11 = np.array([float(m) for t, m in zip(tags, ms) if t == 11])
    n = min(len(t11), len(t13))
    if n > 6:
        idle = t11[1:n] - t13[:n - 1]
        print(f'detector stream: pass-to-pass period {np.median(np.diff(t11[5:n])):7.3f} ms; decode(k) done -> network(k+1) begins '
              f'{np.median(idle[5:]):7.3f} ms  ({np.percentile(idle[5:], 10):7.3f} .. {np.percentile(idle[5:], 90):7.3f})')
    if args.show:
        lo = begins[-args.show - 1] if len(begins) > args.show else begins[0]
        print(f'# raw events of the last {args.show} steps (ms since the first of them)')
        for t, nm in ev:
            if t >= lo:
                print(f'{t - lo:9.3f}  {nm}')


if __name__ == '__main__':
    main()
