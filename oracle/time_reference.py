"""TEST INFRASTRUCTURE -- times the UNMODIFIED reference MultiTracker (fastmot/tracker.py, Kalman + association +
lifecycle; Numba decorators replaced by the identity through oracle/ref_shim.py, i.e. interpreted CPython, NOT the
Numba-compiled code the reference runs in production) on the scripted 50-track / 1080p scene of tests/scenes.py, with
the scripted stand-in for Flow.predict (OpenCV is not available).  Build container only (/root/reference):

    python oracle/time_reference.py > profiles/r02_reference_cpython_timing.txt
    /opt/conda/bin/python3.9 oracle/time_reference.py --real-numba > profiles/r03_reference_numba_timing.txt
        (the reference's @njit functions compiled by the image's Numba 0.54.1, oracle/real_numba.py: what the
         reference's tracker stage costs as it really runs, compile time excluded by a warm-up pass)
"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT / 'oracle'), str(ROOT / 'tests')]

import ref_shim  # noqa: E402
import scenes  # noqa: E402


def main():
    import os
    real = '--real-numba' in sys.argv
    ns = ref_shim.load_reference(real_numba=real)
    if real:
        import numba
        print(f'# reference MultiTracker with its @njit functions compiled by Numba {numba.__version__} (the reference pins 0.48), '
              f'host: {os.cpu_count()} logical CPUs, 1 thread used, compile time excluded (one untimed pass first)')
    else:
        print(f'# reference MultiTracker under the no-op numba shim (interpreted CPython), host: {os.cpu_count()} logical CPUs,'
              ' 1 thread used')
    for name in ('s50_skip1_cosine', 's50_skip2_euclid', 's300_4k_multiclass'):
        scene = scenes.Scene(name)
        best = None
        for rep in range(4 if real else 3):
            ns.track.Track._count = 0
            tracker = ns.tracker.MultiTracker(scene.size, scene.metric, **scenes.tracker_kwargs(name))
            t0 = time.perf_counter()
            scenes.run_scene(tracker, scene, record_states=False)
            dt = time.perf_counter() - t0
            if real and rep == 0:
                continue                      # JIT compilation
            best = dt if best is None else min(best, dt)
        print(f'{name}: {scene.n_frames} frames, {scene.n_ids} identities, detector_frame_skip={scene.skip}: '
              f'{best / scene.n_frames * 1e3:.2f} ms/frame = {scene.n_frames / best:.1f} frames/s '
              '(Kalman + association + lifecycle only; KLT scripted, detector / ReID injected)')


if __name__ == '__main__':
    main()
