"""What the detector pass costs inside the pipeline as a function of what shares the GPU with it: bench.py's config[1]
pipeline with fewer objects and / or without the ReID network (embeddings replaced by constants on the host), and the
detector alone back to back.  Prints frames/s and the HIP-event time of the detector's layer sequence per variant.

    python scripts/interference.py [--steps 300]"""
import argparse
import os
import sys
import time

import numpy as np

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
    base = bench.CONFIGS[1]
    size = base['size']
    ctx.frame_configure(size[0], size[1], RING)
    host_frames = ctx.pinned_frames(RING)
    for n_dets, reid in ((50, True), (50, False), (8, True), (8, False)):
        cfg = dict(base, n_dets=n_dets)
        video = SyntheticVideo(size, n_ids=n_dets, n_frames=RING, seed=100)
        for i, fr in enumerate(video.frames):
            host_frames[i] = fr
        frames = [host_frames[i] for i in range(RING)]
        mot = bench.build_mot(cfg, video)
        if not reid:
            ext = mot.extractors[0]
            ext.extract_async = lambda frame, tlbrs, _e=ext: setattr(_e, '_n', len(tlbrs))
            ext.postprocess = lambda _e=ext: _e.null_embeddings(range(_e._n)).astype(np.float32)
            ext.last_num_features = 0
        Track._count = 0
        mot.reset(1 / 30.)

        def run(n, start):
            mot.detector.net_ms.clear()
            for s in range(start, start + n):
                i = bench.ping_pong(s, RING)
                mot.detector._frame_idx = i
                nxt = frames[bench.ping_pong(s + 1, RING)] if s + 1 < start + n else None
                mot.step(frames[i], next_frame=nxt)
            return list(mot.detector.net_ms)
        run(250, 0)
        ctx.synchronize()
        t0 = time.perf_counter()
        net = run(args.steps, 250)
        ctx.synchronize()
        dt = time.perf_counter() - t0
        print(f'{n_dets:3d} objects, ReID network {"on " if reid else "off"}: {args.steps / dt:7.1f} frames/s, detector network '
              f'{np.mean(net):.4f} ms in the pipeline', flush=True)
        if n_dets == 50 and reid:
            det = mot.detector
            for _ in range(50):
                det.detect_async(frames[0]); det.postprocess()
            det.net_ms.clear()
            t0 = time.perf_counter()
            for k in range(200):
                det.detect_async(frames[k % RING]); det.postprocess()
            dt = time.perf_counter() - t0
            print(f'    detector alone, back to back (upload + preprocess + network + decode + NMS): {200 / dt:7.1f} passes/s, network '
                  f'{np.mean(det.net_ms):.4f} ms', flush=True)
        del mot


if __name__ == '__main__':
    main()
