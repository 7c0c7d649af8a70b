"""TEST INFRASTRUCTURE (oracle side): the detector's own output, checked where it is timed.

    check(det, frame) -> dict

runs `YOLODetector(frame)` (preprocess -> network -> decode -> candidate sort -> DIoU-NMS -> box filters, all on the
GPU) and compares every stage behind the network with the CPU restatement of the reference on the engine's OWN head
tensors (the network itself is compared with PyTorch in tests/test_fullsize_gpu.py):

  * preprocess   np_oracle.yolo_preprocess (detector.py:289-320)               -> uint8 pixels identical
  * decode       np_oracle.yolo_decode (plugins/yolo_layer.cu:127-230)          -> every candidate row, rel 5e-6
                 + np_oracle.filter_scale (detector.py:329-341)                    (device: fast-exp intrinsics); the
                                                                                    candidate SET may differ only for
                                                                                    scores within 2e-5 of conf_thresh
  * sort         (class asc, box_conf desc, candidate index asc)                 -> the device's sorted rows obey it
  * NMS + final  np_oracle.nms_finalize (detector.py:343-364, rect.py:199-244)   -> run on the DEVICE's candidate rows:
                 filters                                                            detections bit-identical
  * whole chain  np_oracle.filter_dets on the oracle's own decode               -> same detections (boxes +-1 px: the
                                                                                    fast-exp decode, conf rel 5e-6)
  * tie order    the reference sorts with an unstable quicksort; the NMS is re-run with the ties in reversed and in
                 random order: `tie_order_invariant` says whether the result can depend on what the reference leaves
                 undefined (known deviations)

Used by tests/test_detector_chain_gpu.py and by bench.py's `parity` object; never by the product."""
import numpy as np

import np_oracle as o


def oracle_rows(det):
    """All candidate rows [sum A*H*W, 7] the reference's plugin would emit for the engine's current head tensors
    (heads in LAYER_FACTORS order, (anchor, cell) order inside a head: detector.py:283)."""
    m = det.model
    rows = []
    for i, head in enumerate(det.heads):
        t = det.backend.read(head, 1)[0]                       # [gh, gw, >= (5+C)*A] fp32
        na = len(m.ANCHORS[i]) // 2
        t = t[..., :(5 + m.NUM_CLASSES) * na]
        rows.append(o.yolo_decode(np.ascontiguousarray(t.transpose(2, 0, 1)), m.ANCHORS[i], m.NUM_CLASSES,
                                  (m.INPUT_SHAPE[2], m.INPUT_SHAPE[1]), m.SCALES[i], m.NEW_COORDS))
    return np.concatenate(rows)


def check(det, frame, check_preprocess=True, n_tie_orders=3):
    from fastmot_amd.detector import YOLODetector
    ctx, m = det.ctx, det.model
    YOLODetector.detect_async(det, frame)
    dets = YOLODetector.postprocess(det).copy()
    cand = ctx.detect_raw_candidates().copy()                  # the device's SORTED candidate rows [k, 8]
    orig = cand[:, 7].copy().view(np.int32)
    res = {'candidates': int(len(cand)), 'detections': int(len(dets))}

    # ---- decode + threshold + scale
    rows = oracle_rows(det)                                    # (heads are read before anything re-runs the engine)
    exp, idx = o.filter_scale(rows, det.upscaled_sz, det.bbox_offset, det.label_mask, det.conf_thresh)
    score = rows[:, 4] * rows[:, 6]
    only_dev = np.setdiff1d(orig, idx)
    only_ora = np.setdiff1d(idx, orig)
    edge = np.concatenate([only_dev, only_ora]).astype(int)
    res['candidate_set_diff'] = int(len(edge))
    res['candidate_set_ok'] = bool(np.all(np.abs(score[edge] - np.float32(det.conf_thresh)) <=
                                          2e-5 * det.conf_thresh)) if len(edge) else True
    pos = {int(v): k for k, v in enumerate(idx)}
    common = np.array([k for k, v in enumerate(orig) if int(v) in pos], int)
    e = exp[[pos[int(orig[k])] for k in common]]
    c = cand[common, :7]
    scale = np.array([det.upscaled_sz[0], det.upscaled_sz[1], det.upscaled_sz[0], det.upscaled_sz[1], 1, 1, 1], float)
    rel = np.abs(c.astype(float) - e) / (np.abs(e) + 1e-3 * scale)
    res['decode_max_rel'] = float(rel.max()) if len(common) else 0.0
    res['decode_ok'] = bool(res['decode_max_rel'] <= 5e-6) and bool(np.array_equal(c[:, 5], e[:, 5]))

    # ---- sort order of the device's rows
    key = list(zip(cand[:, 5].astype(int).tolist(), (-cand[:, 4].astype(float)).tolist(), orig.tolist()))
    res['sorted_ok'] = key == sorted(key)

    # ---- NMS + final filters on the device's own candidates (candidate order = ascending original index)
    by_orig = np.argsort(orig, kind='stable')
    d = cand[by_orig, :7]
    tl, lb, cf = o.nms_finalize(d, det.nms_thresh, det.max_area, det.min_aspect_ratio)
    res['nms_identical'] = bool(len(dets) == len(tl) and np.array_equal(dets.tlbr, tl) and
                                np.array_equal(dets.label, lb) and np.array_equal(dets.conf, cf))
    res['ties_box_conf'] = int(len(d) - len(np.unique(np.stack([d[:, 5], d[:, 4]], 1), axis=0)))
    inv = True
    rng = np.random.default_rng(0)
    for t in range(n_tie_orders):
        rank = -np.arange(len(d)) if t == 0 else rng.permutation(len(d))
        tl2, lb2, cf2 = o.nms_finalize(d, det.nms_thresh, det.max_area, det.min_aspect_ratio, tie_rank=rank)
        # (the ORDER of equal-confidence survivors may follow the tie order; the set of detections may not)
        a = sorted(map(tuple, np.column_stack([lb, tl, cf]).tolist()))
        b = sorted(map(tuple, np.column_stack([lb2, tl2, cf2]).tolist()))
        inv &= a == b
    res['tie_order_invariant'] = bool(inv)

    # ---- the whole post-chain on the oracle's own decode
    tl3, lb3, cf3 = o.nms_finalize(exp, det.nms_thresh, det.max_area, det.min_aspect_ratio)
    ok = len(tl3) == len(dets)
    if ok and len(dets):
        order, used = [], np.zeros(len(dets), bool)
        for j in range(len(tl3)):                              # rows whose confidences agree to 1e-5 may swap places
            dist = np.abs(dets.tlbr - tl3[j]).max(1) + 1e6 * ((dets.label != lb3[j]) | used)
            order.append(int(np.argmin(dist)))
            used[order[-1]] = True
        order = np.array(order)
        ok = bool(np.array_equal(dets.label[order], lb3) and np.abs(dets.tlbr[order] - tl3).max() <= 1 and
                  np.allclose(dets.conf[order], cf3, rtol=5e-6, atol=0))
        res['chain_box_max_px'] = float(np.abs(dets.tlbr[order] - tl3).max())
    res['chain_vs_oracle_decode_ok'] = bool(ok)
    res['oracle_detections'] = int(len(tl3))

    # ---- preprocess (last: it re-runs the first stage into the engine's input tensor, which the arena may share)
    if check_preprocess:
        if not isinstance(frame, np.ndarray):
            frame = ctx.frame_read()
        ctx.detect_preprocess_only()
        inp = det.backend.read(det.graph.input, 1)[0]            # [h, w, 3] RGB
        e_in = o.yolo_preprocess(frame, m.INPUT_SHAPE[1:], det.roi if m.LETTERBOX else None)
        res['preprocess_identical'] = bool(np.array_equal(np.rint(inp.transpose(2, 0, 1) * 255),
                                                          np.rint(e_in * 255)))
    res['detector_chain_identical'] = bool(res['candidate_set_ok'] and res['decode_ok'] and res['sorted_ok'] and
                                           res['nms_identical'] and res['chain_vs_oracle_decode_ok'] and
                                           res.get('preprocess_identical', True))
    return res, dets
