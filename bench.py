#!/usr/bin/env python3
"""Benchmark of the FastMOT per-frame hot path on MI355X (BASELINE.json metric:
"end-to-end tracker FPS @1080p/50 dets").

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config {1,2,4}]

A step = one MOT.step() on one synthetic BGR frame that lies in page-locked HOST memory: the H2D copy of the
frame (6.2 MB at 1080p) is INSIDE the timed region ("includes H2D of each frame"), then
detector network (seeded random weights) -> decode -> DIoU-NMS -> [injected detections] -> KLT (pyramids, GFTT,
FAST, LK, RANSAC) -> OSNet on the crops -> batched Kalman -> association (cost kernels + LAP).

`value` is measured with the next frame handed to MOT.step (a capture queue that already holds it: the
detector of frame t+1 overlaps the ReID / association stages of frame t; results are bit-identical).
`variants` reports, on shorter runs of the same build in the same process, the strictly sequential rate
(no next-frame prefetch) and the rate with the frames already resident in HBM (round-1 definition).

For N > 1 every rank tracks its own stream on its own GPU (weak scaling); the only collective is the opt-out
ReID-gallery all-gather (RCCL) on a side stream.  value = total frames / max-over-ranks time.
Rank 0 prints ONE JSON line.
"""
import argparse
import gc
import json
import os
import sys
import time
from pathlib import Path
from types import SimpleNamespace

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(1, str(ROOT / 'tests'))        # the synthetic workload (tests/synthetic.py) is test / bench infrastructure

MFMA_PEAK_TFLOPS = 2500.0     # dense fp16/bf16 MFMA peak of MI355X (MI355X_MICROARCH.md)
RING = 32
SETTLE_MIN, SETTLE_MAX, SETTLE_CHECK = 150, 450, 10

# BASELINE.json configs that fit one GPU.  [0] is the detector-disabled CPU plumbing case and [3] is config[1]
# on 8 GPUs (= --gpus 8 --config 1); both are covered by the parity tests / the scaling run, not bench lines.
CONFIGS = {
    1: dict(name='BASELINE config[1]', size=(1920, 1080), n_dets=50, yolo='YOLOv4_608', reid='OSNet025', skip=1,
            labels=(1,), desc='single 1080p stream, YOLOv4 608x608 (80 cls) + OSNet-x0.25, detector_frame_skip=1'),
    2: dict(name='BASELINE config[2]', size=(1920, 1080), n_dets=50, yolo='YOLOv4CSP_640', reid='OSNet10', skip=5,
            labels=(1,), desc='single 1080p stream, YOLOv4-CSP 640x640 (80 cls) + OSNet-x1.0, '
                              'detector_frame_skip=5 (KLT-heavy)'),
    4: dict(name='BASELINE config[4]', size=(3840, 2160), n_dets=300, yolo='YOLOv4P6_1280', reid='OSNet025', skip=1,
            labels=(0, 1, 2), desc='MOT20-style dense 4K stream, YOLOv4-P6 1280x1280, 300 detections/frame, '
                                   '3 classes (one OSNet-x0.25 per class), detector_frame_skip=1'),
}


SIZE, N_DETS = CONFIGS[1]['size'], CONFIGS[1]['n_dets']      # config[1] shorthands for scripts/


def tracker_cfg():
    return SimpleNamespace(
        max_age=6, age_penalty=2, motion_weight=0.2, max_assoc_cost=0.8, max_reid_cost=0.6, iou_thresh=0.4,
        duplicate_thresh=0.8, occlusion_thresh=0.7, conf_thresh=0.5, confirm_hits=1, history_size=50,
        kalman_filter_cfg=SimpleNamespace(std_factor_acc=2.25, std_offset_acc=78.5, std_factor_det=(0.08, 0.08),
                                          std_factor_klt=(0.14, 0.14), min_std_det=(4.0, 4.0),
                                          min_std_klt=(5.0, 5.0), init_pos_weight=5, init_vel_weight=12,
                                          vel_coupling=0.6, vel_half_life=2),
        flow_cfg=SimpleNamespace(bg_feat_scale_factor=(0.1, 0.1), opt_flow_scale_factor=(0.5, 0.5),
                                 feat_density=0.005, feat_dist_factor=0.06, ransac_max_iter=500, ransac_conf=0.99,
                                 max_error=100, inlier_thresh=4, bg_feat_thresh=10,
                                 obj_feat_params=SimpleNamespace(maxCorners=1000, qualityLevel=0.06, blockSize=3),
                                 opt_flow_params=SimpleNamespace(winSize=(5, 5), maxLevel=5, criteria=(3, 10, 0.03))))


def build_mot(cfg, video, gallery_sync=None, nms_candidates=1500):
    """nms_candidates: the YOLO heads' biases are scripted (synthetic.scripted_head_weights: same seeded random
    network otherwise) so that about this many candidate boxes per frame pass `conf_thresh` and the candidate sort /
    DIoU-NMS kernels run on real work inside the timed region (detector.py:322-365); 0 = purely random heads (nothing
    passes).  The tracker is fed the scripted detections either way."""
    import fastmot_amd.mot as mot_mod
    from fastmot_amd.detector import YOLODetector
    from synthetic import InjectedYOLODetector, scripted_head_weights
    mot_mod.YOLODetector = InjectedYOLODetector
    tcfg = tracker_cfg()
    if gallery_sync is not None:
        tcfg.gallery_sync = gallery_sync
    weights = None
    if nms_candidates:
        weights = scripted_head_weights(cfg['size'], cfg['yolo'], cfg['labels'] if len(cfg['labels']) > 1 else cfg['labels'][0],
                                         video.frames[0], nms_candidates)
    try:
        mot = mot_mod.MOT(cfg['size'], detector_type='YOLO', detector_frame_skip=cfg['skip'], class_ids=cfg['labels'],
                          yolo_detector_cfg=SimpleNamespace(model=cfg['yolo'], conf_thresh=0.25, nms_thresh=0.5,
                                                            max_area=800000, min_aspect_ratio=1.2,
                                                            max_candidates=8192, weights=weights),
                          feature_extractor_cfgs=tuple(SimpleNamespace(model=cfg['reid'], batch_size=int(os.environ.get('FASTMOT_BENCH_REID_BATCH', '64')))
                                                       for _ in cfg['labels']),
                          tracker_cfg=tcfg)
    finally:
        mot_mod.YOLODetector = YOLODetector
    mot.detector.bind_video(video, labels=cfg['labels'])
    return mot


def ping_pong(s, n):
    """Frame index of step s over a clip of n frames played forwards and backwards (positions stay
    continuous; a plain wrap-around would teleport every object once per clip)."""
    p = 2 * (n - 1)
    s %= p
    return s if s < n else p - s


def cpu_leg(cfg, video, mot, budget_s=20.0):
    """cpu_baseline + parity in one pass.  kind=port: the numpy restatement (oracle/cpu_tracker.py +
    cv_oracle.py) of the reference's CPU tracker path -- KLT + Kalman + association, detector / ReID networks
    excluded exactly as in the reference's TensorRT-disabled configuration (BASELINE config[0]) -- timed on one
    host core over a bounded number of frames of the same synthetic clip, fed with the embeddings the HIP OSNet
    produced so that its per-frame output can be compared with the HIP pipeline's (the `parity` object)."""
    sys.path.insert(0, str(ROOT / 'oracle'))
    import e2e_check
    n = min(video.n_frames, RING)
    hip, emb = e2e_check.hip_pass(mot, video, n, cfg['skip'], prefetch=False)
    tkw = vars(tracker_cfg())
    ora, dt, done = e2e_check.oracle_pass(cfg['size'], mot.extractors[0].metric.lower(), tkw, video, n, cfg['skip'],
                                          emb, budget_s=budget_s, labels=cfg['labels'])
    parity = e2e_check.compare(hip[:done], ora)
    parity['oracle'] = 'oracle/cpu_tracker.py + cv_oracle.py on the same frames, detections and HIP embeddings'
    # the detector's OWN output on a frame of the timed clip (the tracker is fed scripted detections): preprocess,
    # decode of the engine's head tensors, candidate sort, DIoU-NMS and box filters against the oracle
    import detector_check
    chain, _ = detector_check.check(mot.detector, video.frames[0])
    parity['detector_chain'] = chain
    parity['detector_chain_identical'] = chain['detector_chain_identical']
    base = {'value': round(done / dt, 3), 'unit': 'frames/s', 'cores': 1, 'kind': 'port',
            'implementation': 'numpy-port: oracle/cpu_tracker.py + cv_oracle.py (also the checker of `parity`)',
            'sample': f'{done} frames of the same {cfg["size"][0]}x{cfg["size"][1]}/{video.n_ids}-detection '
                      f'synthetic clip (detector_frame_skip={cfg["skip"]}); numpy port of KLT+Kalman+association '
                      '(oracle/), detector+ReID networks excluded (injected), NOT the Numba-compiled reference'}
    return base, parity


def _compiled_worker(args):
    """One host process = one independent video stream through the compiled CPU port (multi-core leg)."""
    cfg, seed, budget_s = args
    if _AFFINITY_AT_START is not None:            # forked from a process get_context() has bound to the GPU's NUMA node
        try:
            os.sched_setaffinity(0, _AFFINITY_AT_START)
        except OSError:
            pass
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(1)
    except ImportError:
        pass
    sys.path.insert(0, str(ROOT / 'oracle'))
    import c_baseline
    from synthetic import SyntheticVideo
    video = SyntheticVideo(cfg['size'], n_ids=cfg['n_dets'], n_frames=8, seed=seed)
    r = c_baseline.time_clip(cfg, video, tracker_cfg(), budget_s)
    return r['value']


def compiled_baseline(cfg, video, budget_s=10.0):
    """cpu_baseline, kind=port (compiled): the reference's TensorRT-disabled CPU path with the parts the reference
    runs as compiled code (OpenCV KLT / RANSAC) in plain C (oracle/c_baseline.c, gcc -O3, no fast-math) under the
    reference's Python orchestration -- the 'Numba-class proxy' of SURVEY.md section 8d.  Timed on ONE thread on the
    bench clip, and on ALL usable cores as that many independent streams (the path shards by stream, one process
    each, as it does across GPUs); None when gcc / the library is unavailable."""
    sys.path.insert(0, str(ROOT / 'oracle'))
    try:
        import c_baseline
        c_baseline.lib()
    except (ImportError, OSError, RuntimeError):
        return None
    one = c_baseline.time_clip(cfg, video, tracker_cfg(), budget_s)
    one['kind'] = 'port'
    one['implementation'] = 'compiled-port: oracle/c_baseline.c (gcc -O3) under the reference\'s Python orchestration'
    cores = usable_cpus()
    if cores > 1:
        import multiprocessing as mp
        try:
            with mp.get_context('fork').Pool(cores) as pool:
                t0 = time.perf_counter()
                rates = pool.map(_compiled_worker, [(cfg, 200 + i, budget_s) for i in range(cores)])
                wall = time.perf_counter() - t0
            one['all_cores'] = {'value': round(float(sum(rates)), 2), 'unit': 'frames/s', 'cores': cores,
                                'how': f'{cores} independent streams, one single-threaded process each (sum of their '
                                       f'rates; {wall:.0f} s wall incl. building each process\'s clip)'}
        except (OSError, RuntimeError) as err:
            one['all_cores'] = {'error': str(err)}
    return one


def reference_numba_constant():
    """The reference's OWN tracker stage (Kalman + association + life cycle; KLT scripted, networks injected) with its
    @njit functions compiled by a real Numba, timed in the BUILD container (no Numba runs on the GPU box): a labelled
    constant with its provenance file, not a measurement of this run."""
    path = ROOT / 'profiles' / 'r03_reference_numba_timing.txt'
    try:
        line = next(l for l in path.read_text().splitlines() if l.startswith('s50_skip1'))
        fps = float(line.split('=')[-1].split('frames/s')[0])
    except (OSError, StopIteration, ValueError):
        return None
    return {'value': fps, 'unit': 'frames/s', 'cores': 1, 'kind': 'reference (partial: Kalman + association + life cycle '
            'only, no KLT / detector / ReID)', 'measured': 'build container, not this run',
            'source': 'profiles/r03_reference_numba_timing.txt (oracle/time_reference.py --real-numba, Numba 0.54.1)'}


HBM_PEAK_GBS, PCIE_PEAK_GBS, FP64_PEAK_TFLOPS = 8000.0, 63.0, 78.6      # MI355X_MICROARCH.md chip table (spec)

# stage boundaries the library stamps with HIP events on the stage's own stream (fm_trace_mark, csrc/*.hip)
STAGE_TAGS = (('detector first launch: resize + BGR->RGB + normalise computed inside the fused stem (frame -> first stored tensor)', 11, 15),
              ('detector network (conv engine, first launch included)', 11, 12),
              ('head decode + threshold + compaction', 12, 13),
              ('candidate sort + greedy DIoU-NMS + box filters + write-back (whole post-processing)', 20, 21),
              ('candidate sort', 20, 22), ('DIoU-NMS bit matrix', 22, 23),
              ('NMS scan + box filters + write-back', 23, 21), ('next frame H2D copy', 30, 31),
              ('ReID first launch: crop + resize + normalise computed inside the 7x7 stem (boxes read from pinned memory)', 32, 34), ('ReID network (OSNet) + head', 34, 35),
              ('embedding export to pinned memory', 35, 33), ('KLT gray + pyramid + Scharr', 42, 43),
              ('KLT keypoint bookkeeping + GFTT', 44, 45), ('KLT background FAST', 46, 47), ('KLT pyramidal LK', 40, 41),
              ('Kalman warp + predict + KLT update', 50, 51), ('Kalman detection update', 52, 53),
              ('pairwise cost terms (cdist + Mahalanobis + IoU)', 54, 55), ('stage cost gather + gate', 56, 57))


def stage_durations(t_start, t_end):
    """Durations of a stage from the time stamps of its start and end marks.  An occurrence is the FIRST start mark since
    the previous end mark, up to that end mark (both marks sit on one stream, in order).  Marks may repeat inside an
    occurrence -- a batch larger than the ReID network's maximum runs in chunks, each with its own crop / network marks --
    so pairing the i-th start with the i-th end is wrong as soon as the counts differ (round 4's first config[4] table
    had a negative crop stage from exactly that)."""
    ta = np.sort(np.asarray(t_start, np.float64))
    tb = np.sort(np.asarray(t_end, np.float64))
    if not len(ta) or not len(tb):
        return np.zeros(0)
    first = np.searchsorted(ta, np.concatenate(([-np.inf], tb[:-1])), side='right')    # first start after the previous end
    ok = (first < len(ta)) & (ta[np.minimum(first, len(ta) - 1)] <= tb)
    return tb[ok] - ta[first[ok]]


def stage_rooflines(ctx, cfg, mot, run_steps, n_steps=48):
    """per-stage achieved GB/s or TFLOP/s and roofline fraction INSIDE the pipelined step.
    Durations: HIP events on each stage's own stream (fm_trace_*), median over the occurrences in `n_steps` traced
    steps; work: the algorithmic bytes / FLOPs of DESIGN.md section 3 for this configuration.  Latency-bound stages
    (serial scans, tiny launches) are reported in microseconds with their share of a step."""
    ctx.synchronize()
    ctx.trace_start(128 * (n_steps + 4))
    run_steps(n_steps)
    ctx.synchronize()
    tags, ms = ctx.trace_read()
    W, H = cfg['size']
    D = T = cfg['n_dets']
    m = mot.detector.model
    _, in_h, in_w = m.INPUT_SHAPE
    det_flops, _ = mot.detector.backend.cost(1)
    first = mot.detector.graph.layers[0]             # stem, stem pair or stem pair + pointwise conv (FM_OP_STEMCONV / FM_OP_STEM2)
    first_out = first['out']
    ext = mot.extractors[0]
    # The ReID marks bracket the step's extractor work: the crop stage ends at the first chunk's network mark (a batch
    # larger than the network's maximum runs in chunks), the network stage at the end mark -- with one extractor per
    # class (config[4]) the calls follow each other on the stream and the row spans all of them: the work is the step's.
    d_call = D
    d_chunk = min(-(-D // len(mot.extractors)), ext.backend.max_batch)
    ext_flops, ext_bytes = ext.backend.cost(d_call)
    head_bytes = sum((5 + m.NUM_CLASSES) * (len(a) // 2) * (in_h // f) * (in_w // f) * 4 for a, f in zip(m.ANCHORS, m.LAYER_FACTORS))
    _, eh, ew = ext.model.INPUT_SHAPE
    K = max(mot.detector.last_candidates, 1)
    flow = cfg.get('flow_scale', 0.5)
    px0 = int(W * flow) * int(H * flow)
    pyr_px = sum(px0 / 4 ** l for l in range(6))
    work = {
        (11, 15): ('hbm', W * H * 3 + first_out.h * first_out.w * first['cout'] * 2,
                   f'frame u8 in + the first launch\'s fp16 output ({first_out.h} x {first_out.w} x {first["cout"]}; op {first["op"]})'),
        11: ('mfma', det_flops, 'conv FLOPs (2 MAC)'),
        12: ('hbm', head_bytes, 'fp32 head tensors in'),
        20: ('latency', None, f'K = {K} candidates over conf_thresh: K^2 key comparisons from LDS, then one round per NMS survivor'),
        22: ('latency', None, f'K^2/2 = {K * K // 2} pair tests (fp32 IoU pre-test, exact fp64 DIoU near the threshold)'),
        23: ('latency', None, f'greedy scan over {-(-K // 64)} chunks of 64, one workgroup'),
        30: ('pcie', W * H * 3, 'frame u8'),
        32: ('hbm', d_chunk * (eh // 2) * (ew // 2) * 16 * 2, f'first chunk, {d_chunk} crops: the stem\'s fp16 output (+ <= crop pixels in)'),
        34: ('hbm', ext_bytes, f'{d_call} crops (all classes): activations + weights fp16; {ext_flops / 1e9:.1f} GFLOP'),
        35: ('latency', None, f'{D} x 512 fp32'),
        42: ('hbm', W * H * 3 + W * H + px0 + pyr_px * 5.25, 'frame in, gray + half + 6-level pyramid + int16 Scharr pairs out'),
        44: ('latency', None, f'{T} track crops: rect / ellipse / mask filters, min-eigenvalue map, corner selection'),
        46: ('latency', None, 'FAST-9/16 + NMS on the 0.1-scale background image'),
        40: ('latency', None, 'VALU issue: <= 6 levels x <= 10 iterations x 25 taps per point, one wavefront per 2 points'),
        50: ('latency', T * 576 * 2, 'fp64 state read + write, one wavefront per track'),
        52: ('latency', T * 576 * 2, 'fp64 state read + write'),
        54: ('fp64', 6.0 * T * D * 512, 'fp64 FLOPs'),
        56: ('latency', None, 'cascade stages of one update'),
    }
    out = []
    for name, a, b in STAGE_TAGS:
        d = stage_durations(ms[tags == a], ms[tags == b])
        n = len(d)
        if n < 4:
            continue
        d = d[n // 8:]                                      # (the first steps of the window still fill the pipeline)
        us = float(np.median(d)) * 1e3
        per_step = n / n_steps
        bound, amount, what = work.get((a, b)) or work[a]
        row = {'stage': name, 'us': round(us, 2), 'per_step': round(per_step, 2), 'bound': bound, 'work': what}
        if amount and us > 0:
            if bound in ('hbm', 'pcie', 'latency'):
                row['GB/s'] = round(amount / us / 1e3, 1)
                peak = PCIE_PEAK_GBS if bound == 'pcie' else HBM_PEAK_GBS
                row['frac'] = round(amount / us / 1e3 / peak, 4)
                row['peak'] = f'{peak:g} GB/s'
            else:
                row['TFLOP/s'] = round(amount / us / 1e6, 3)
                peak = MFMA_PEAK_TFLOPS if bound == 'mfma' else FP64_PEAK_TFLOPS
                row['frac'] = round(amount / us / 1e6 / peak, 5)
                row['peak'] = f'{peak:g} TFLOP/s'
        out.append(row)
    return out


_AFFINITY_AT_START = sorted(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else None


def usable_cpus():
    """Host cores this process may use: the affinity mask it STARTED with (get_context() narrows the calling thread to the
    GPU's NUMA node afterwards: the CPU baseline must not inherit that -- ADVICE r4), capped by a cgroup-v2 CPU quota
    when there is one."""
    n = len(_AFFINITY_AT_START) if _AFFINITY_AT_START is not None else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def flow_threads_for(local_world):
    """FASTMOT_FLOW_THREADS for one of `local_world` ranks on this node: main thread + prediction worker + pool must
    fit the rank's share of the cores; between 1 (no pool) and the library's own maximum of 7."""
    share = usable_cpus() // max(local_world, 1)
    return max(1, min(7, share - 2))


def pmc_traffic(n_conv_launches=None):
    """(bytes per conv launch, source, stale) from the newest committed rocprofv3 PMC passes (profiles/*_pmc_conv.json,
    produced by scripts/collect_pmc.sh: separate --pmc FETCH_SIZE / WRITE_SIZE runs of the detector network; FETCH_SIZE
    doubled per the gfx950 note of MI355X_MICROARCH.md).  PMC counters cannot be collected from inside this process,
    hence the file; the 4th item carries the MFMA utilisation of the conv launches and their read amplification from
    the same file when it has them; `stale` is True when the file counted another number of conv launches per frame than the network
    of this run has (the kernels changed since the passes were taken); None when no file is there."""
    for name in ('r06_pmc_conv.json', 'r05_pmc_conv.json', 'r04_pmc_conv.json', 'r03_pmc_conv.json', 'r02_pmc_conv.json', 'r01_pmc_conv.json'):
        try:
            with open(ROOT / 'profiles' / name) as f:
                d = json.load(f)
            per_frame = d['launches'] / d['replays']
            stale = n_conv_launches is not None and abs(per_frame - n_conv_launches) > 0.5
            # (round 5: the same passes normalised per layer by scripts/layer_pmc.py -- MFMA busy cycles over duration x
            # 2.4 GHz x 1024 SIMDs, 2 x FETCH_SIZE over the algorithmic read bytes)
            extra = {k: d.get(k) for k in ('mfma_util', 'read_amplification')}
            return d['traffic_bytes_per_launch'], f'profiles/{name} (separate rocprofv3 --pmc passes)', bool(stale), extra
        except (OSError, KeyError, ValueError, ZeroDivisionError):
            continue
    return None


class TorchCtl:
    """torch.distributed as the harness communicator (FASTMOT_BENCH_TORCH=1 with backend nccl; gloo in the CPU test of
    the harness): the same two calls as gallery.RcclComm -- barrier(), allgather_small(values) -> array [world, n]."""

    def __init__(self, dist, torch, device=None):
        self.dist, self.torch, self.device = dist, torch, device
        self.world = dist.get_world_size()

    def allgather_small(self, values):
        v = self.torch.tensor(np.atleast_1d(np.asarray(values, np.float64)), device=self.device)
        out = [self.torch.zeros_like(v) for _ in range(self.world)]
        self.dist.all_gather(out, v)
        return np.stack([t.cpu().numpy() for t in out])

    def barrier(self):
        if self.device is not None:
            self.torch.cuda.synchronize()
        self.dist.barrier()
        if self.device is not None:
            self.torch.cuda.synchronize()

    def close(self):
        pass


class Harness:
    """Control flow of the benchmark around the workload: untimed settle phase, fences, the timed region and the max over
    ranks -- over a communicator `comm` (None for one process) with barrier() and allgather_small().  Every decision that
    changes how many steps a rank runs is taken by ALL ranks in one collective, so that every rank issues the same
    number of collectives (harness and gallery alike) whatever its own clock says (VERDICT r4 item 7;
    tests/test_bench_harness.py runs it with two gloo processes whose step times differ)."""

    def __init__(self, run, device_sync, comm=None, settle_min=SETTLE_MIN, settle_max=SETTLE_MAX,
                 settle_check=SETTLE_CHECK, clock=time.perf_counter):
        self.run, self.device_sync, self.comm, self.clock = run, device_sync, comm, clock
        self.settle_min, self.settle_max, self.settle_check = settle_min, settle_max, settle_check

    def fence(self):
        self.device_sync()                 # hipDeviceSynchronize on this rank's GPU (all streams of the pipeline)
        if self.comm is not None:
            self.comm.barrier()

    def all_ranks(self, flag):
        """True iff `flag` holds on every rank (one small collective on the harness channel; N = 1: the flag)."""
        if self.comm is None:
            return bool(flag)
        return bool(np.all(self.comm.allgather_small([1.0 if flag else 0.0]) > 0.5))

    def max_over_ranks(self, x):
        return float(x) if self.comm is None else float(self.comm.allgather_small([x]).max())

    def timed(self, n, start, frames, prefetch):
        # the interpreter's cyclic collector is parked for the timed region (and run right before it): a generation-2
        # pass over the heap torch's import leaves behind takes milliseconds, and the driver's window is 20 steps = 20 ms
        gc_on = gc.isenabled()
        gc.collect()
        gc.disable()
        try:
            self.fence()
            t0 = self.clock()
            net_ms = self.run(n, start, frames, prefetch)
            self.fence()
            return self.clock() - t0, net_ms
        finally:
            if gc_on:
                gc.enable()

    def settle(self, start, frames):
        """Untimed settle phase before the warm-up the command line asks for: a fresh process runs its first few
        hundred steps slower (GPU clocks ramping, hipGraphs instantiated, the prediction worker and the RANSAC pool
        asleep), and a short `--steps 20 --warmup 5` run would time exactly that.  At least settle_min steps, then
        until the last 20 step times lie within 3 % of their median, at most settle_max.  The decision to stop is
        taken every settle_check steps and, with N > 1, by ALL ranks together: every rank runs the same number of
        steps, hence the same number of gallery collectives (a rank that stopped on its own clock would leave the
        others waiting in an exchange it never issues)."""
        times = []
        s = start
        while s - start < self.settle_max:
            t0 = self.clock()
            self.run(1, s, frames, False)
            times.append(self.clock() - t0)
            s += 1
            if s - start >= self.settle_min and (s - start - self.settle_min) % self.settle_check == 0:
                last = np.array(times[-20:])
                if self.all_ranks(np.all(np.abs(last - np.median(last)) <= 0.03 * np.median(last))):
                    break
        return s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=300)
    ap.add_argument('--warmup', type=int, default=30)
    ap.add_argument('--config', type=int, default=1, choices=sorted(CONFIGS))
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-variants', action='store_true', help='skip the sequential / resident-frames side runs')
    ap.add_argument('--no-prefetch', dest='prefetch', action='store_false',
                    help='strictly sequential steps: do not start the detector on frame t+1 during frame t')
    ap.add_argument('--resident', action='store_true', help='frames resident in HBM (no H2D in the timed region)')
    ap.add_argument('--nms-candidates', type=int, default=1500,
                    help='candidate boxes per frame the scripted YOLO heads let through conf_thresh (0: purely random heads, none)')
    ap.add_argument('--no-gallery-sync', dest='gallery', action='store_false',
                    help='N > 1: disable the cross-stream ReID-gallery all-gather (RCCL)')
    args = ap.parse_args()
    cfg = CONFIGS[args.config]

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} needs torch.distributed.run with nproc-per-node {args.gpus}')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if world > 1 and 'FASTMOT_FLOW_THREADS' not in os.environ:
        # N processes share the node's host cores: size every rank's RANSAC worker pool to its share (the library's
        # default assumes the whole machine; its workers and the prediction worker spin, bounded, while they wait)
        os.environ['FASTMOT_FLOW_THREADS'] = str(flow_threads_for(int(os.environ.get('LOCAL_WORLD_SIZE', world))))

    from fastmot_amd import Track, models
    from fastmot_amd.detector import DeviceFrame
    models.allow_random_weights()          # no weight files offline: seeded random parameters (stated in `data`)
    from fastmot_amd.runtime import get_context
    from synthetic import SyntheticVideo
    ctx = get_context()                    # device = LOCAL_RANK (torchrun convention)

    # Collectives of the harness itself (barrier around the timed region, max over ranks).  Default: the library's own
    # RCCL binding (fm_gallery_*, control channel) -- the process never imports torch, and the gallery exchange inside
    # the timed region runs on the same binding.  FASTMOT_BENCH_TORCH=1: torch.distributed "nccl" for both instead.
    ctl = dist = torch = None
    if world > 1:
        if os.environ.get('FASTMOT_BENCH_TORCH', '0') == '1':
            import torch
            import torch.distributed as dist
            torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')) % max(torch.cuda.device_count(), 1))
            dist.init_process_group('nccl', rank=rank, world_size=world)
            ctl = TorchCtl(dist, torch, device='cuda')
        else:
            from fastmot_amd.gallery import RcclComm
            ctl = RcclComm(ctx, 64, channel=1)

    size = cfg['size']
    video = SyntheticVideo(size, n_ids=cfg['n_dets'], n_frames=RING, seed=100 + rank)
    ctx.frame_configure(size[0], size[1], RING)
    host_frames = ctx.pinned_frames(RING)                 # page-locked host memory = the capture queue
    for i, fr in enumerate(video.frames):
        host_frames[i] = fr
        ctx.frame_ring_store(i, fr)                       # resident copies for the `resident` variant only
    pinned = [host_frames[i] for i in range(RING)]
    resident = [DeviceFrame(i) for i in range(RING)]

    sync = None
    if world > 1 and args.gallery:
        from fastmot_amd.gallery import GallerySync
        sync = GallerySync(history_size=tracker_cfg().history_size, feat_dim=512)
    mot = build_mot(cfg, video, gallery_sync=sync, nms_candidates=args.nms_candidates)
    Track._count = 0
    mot.reset(1 / 30.)

    def run(n, start, frames, prefetch):
        mot.detector.net_ms.clear()
        for s in range(start, start + n):
            i = ping_pong(s, RING)
            mot.detector._frame_idx = i
            nxt = frames[ping_pong(s + 1, RING)] if prefetch and s + 1 < start + n else None
            mot.step(frames[i], next_frame=nxt)
        return list(mot.detector.net_ms)

    frames = resident if args.resident else pinned

    hs = Harness(run, ctx.synchronize, comm=ctl)
    pos = hs.settle(0, frames)
    settle_steps = pos
    hs.fence()                             # every rank enters warm-up and timed region at the same exchange index
    run(args.warmup, pos, frames, args.prefetch)
    pos += args.warmup
    elapsed, net_ms = hs.timed(args.steps, pos, frames, args.prefetch)
    elapsed = hs.max_over_ranks(elapsed)
    pos += args.steps

    variants = None
    if world == 1 and not args.no_variants:
        nv = 100
        variants = {'steps_each': nv}
        for key, fr, pf in (('h2d_sequential_fps', pinned, False), ('resident_prefetch_fps', resident, True),
                            ('resident_sequential_fps', resident, False)):
            run(4, pos, fr, pf)
            dt, _ = hs.timed(nv, pos + 4, fr, pf)
            pos += nv + 4
            variants[key] = round(nv / dt, 2)

    stages_roof = None
    if world == 1 and not args.no_variants:
        def traced(n):
            nonlocal pos
            run(n, pos, pinned, args.prefetch)
            pos += n
        run(4, pos, pinned, args.prefetch)
        pos += 4
        stages_roof = stage_rooflines(ctx, cfg, mot, traced)

    if rank == 0:
        flops, _ = mot.detector.backend.cost(1)
        n_launch = len(mot.detector.graph.layers)
        net_avg_ms = float(np.mean(net_ms)) if len(net_ms) else float('nan')
        achieved = flops / (net_avg_ms * 1e-3) / 1e12
        from fastmot_amd.utils import Profiler
        stages = {k: round(Profiler.get_avg_millis(k), 3) for k in ('preproc', 'detect', 'track', 'extract', 'assoc')}
        from fastmot_amd.models import graph as _G
        n_conv = sum(1 for d in mot.detector.graph.layers if d['op'] in _G.CONV_OPS + (_G.OP_RESBLOCK, _G.OP_STEM2, _G.OP_PAIR11))
        traffic = pmc_traffic(n_conv) if args.config == 1 else None      # the PMC passes are of YOLOv4@608
        metric = ('end-to-end tracker FPS @1080p/50 dets' if args.config == 1 else
                  f'end-to-end tracker FPS @{size[0]}x{size[1]}/{cfg["n_dets"]} dets, detector_frame_skip={cfg["skip"]} '
                  f'({cfg["name"]})')
        out = {
            'metric': metric,
            'value': round(world * args.steps / elapsed, 2),
            'unit': 'frames/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 4),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f16', 'data': 'synthetic',
            'config': {'workload': f'{cfg["name"]}: {cfg["desc"]}, seeded random weights, {cfg["n_dets"]} injected '
                                   'detections/frame, ' +
                                   ('frames resident in HBM' if args.resident else
                                    'frames in pinned host memory, H2D per frame included'),
                       'next_frame_prefetch': bool(args.prefetch), 'streams_per_gpu': 1,
                       'parallelism': f'1 stream/GPU x {world}',
                       'harness_collectives': None if world == 1 else ('torch.distributed nccl' if dist is not None else
                                                                         'fm_gallery_* control channel (RCCL via the C ABI, no torch in the process)'),
                       'host_threads_per_rank': {'ransac_pool': os.environ.get('FASTMOT_FLOW_THREADS', 'library default'),
                                                 'usable_cpus': usable_cpus()},
                       'numa': dict(getattr(ctx, 'numa', {}), pinned_frames_node=__import__('fastmot_amd.runtime', fromlist=['x']).numa_node_of(host_frames)),
                       'gallery_allgather': None if sync is None else sync.stats(),
                       'visible_tracks': len(list(mot.visible_tracks())),
                       'yolo_candidates_nms_in': mot.detector.last_candidates,
                       'yolo_candidates_nms_out': mot.detector.last_real_count,
                       'stage_ms': stages},
            'roofline': {'bound': 'mfma', 'kernel': f'conv kernels of the detector ({cfg["yolo"]}: {n_launch} launches '
                                                    'per frame incl. fused residual units / SPP; the first one also resizes '
                                                    'and normalises the frame), measured with HIP events on the detector '
                                                    'stream inside the pipeline, on every 4th pass of the timed region (the '
                                                    'event pair itself costs ~1 % of the frame rate)',
                         'achieved': round(achieved, 3), 'peak': MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': round(achieved / MFMA_PEAK_TFLOPS, 5),
                         # HBM-side bytes per launch: NOT measured in this run (PMC counters cannot be collected from
                         # inside the process) but read from the committed rocprofv3 --pmc passes of the same network
                         'traffic': traffic[0] if traffic else None, 'traffic_source': traffic[1] if traffic else None,
                         'traffic_stale': traffic[2] if traffic else None,
                         # matrix-pipe busy cycles / (duration x 2.4 GHz x 1024 SIMDs) and HBM-side reads / algorithmic
                         # reads of the conv launches, from the same PMC passes (stand-alone replays of the network)
                         'mfma_util': traffic[3].get('mfma_util') if traffic else None,
                         'read_amplification': traffic[3].get('read_amplification') if traffic else None,
                         'flop_per_frame': flops, 'net_ms_per_frame': round(net_avg_ms, 4), 'net_ms_samples': len(net_ms),
                         'avg_launch_us': round(net_avg_ms * 1e3 / n_launch, 3)},
        }
        out['config']['settle_steps_untimed'] = settle_steps
        if stages_roof is not None:
            out['stage_roofline'] = stages_roof
        if variants is not None:
            out['variants'] = variants
            # what the UNMODIFIED reference app.py gets (it calls mot.step(frame), no next_frame): first-class number
            out['sequential_fps'] = variants['h2d_sequential_fps']
        if world == 1 and not args.no_cpu_baseline:
            port, out['parity'] = cpu_leg(cfg, video, mot)
            comp = compiled_baseline(cfg, video)
            # the mandatory baseline is the compiled port (1 thread + all usable cores); the numpy port -- the parity
            # checker, ~12x slower -- rides along; the reference's own Numba-compiled tracker stage is a labelled constant
            out['cpu_baseline'] = comp if comp is not None else port
            out['cpu_baseline_numpy_port'] = port
            ref = reference_numba_constant()
            if ref is not None and args.config == 1:
                out['reference_numba_tracker_stage'] = ref
        print('stage ms (Profiler, incl. warmup):', stages, file=sys.stderr)
        print(json.dumps(out), flush=True)
    if sync is not None:
        sync.close()
    if ctl is not None:
        ctl.barrier()
        ctl.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
