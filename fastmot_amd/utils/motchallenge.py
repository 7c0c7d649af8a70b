"""MOTChallenge I/O and a minimal CLEAR-MOT / IDF1 scorer.

* `format_rows` / `write_rows`: the result rows app.py:91-97 writes
  (`frame,id,x,y,w,h,-1,-1,-1`, boxes rescaled from the processing size to the stream resolution);
* `read_txt`: gt.txt / result files -> {frame: [(id, tlwh), ...]};
* `evaluate`: MOTA / MOTP / FP / FN / IDSW (Bernardin & Stiefelhagen 2008: per-frame assignment that keeps
  the previous frame's correspondences when they are still valid, IoU threshold 0.5) and IDF1 / IDP / IDR
  (Ristani et al. 2016: one global bipartite matching of ground-truth and predicted identities).
With data mounted this turns "MOTA / IDF1 within 0.5 pt of the reference" into a measurable statement
(eval/results/MOT20-01.txt of the reference is a result file in exactly this format).
"""
from collections import defaultdict

import numpy as np
from scipy.optimize import linear_sum_assignment


def format_rows(frame_count, tracks, resize_to, resolution):
    """tracks: iterable with .trk_id and .tlbr (processing coordinates); -> list of text rows."""
    resize_to = np.asarray(resize_to, float)
    resolution = np.asarray(resolution, float)
    rows = []
    for track in tracks:
        tl = track.tlbr[:2] / resize_to * resolution
        br = track.tlbr[2:] / resize_to * resolution
        w, h = br - tl + 1
        rows.append(f'{frame_count},{track.trk_id},{tl[0]:.6f},{tl[1]:.6f},{w:.6f},{h:.6f},-1,-1,-1\n')
    return rows


def write_rows(txt, frame_count, tracks, resize_to, resolution):
    for row in format_rows(frame_count, tracks, resize_to, resolution):
        txt.write(row)


def read_txt(path, min_conf=None, classes=None):
    """MOTChallenge text file -> {frame: [(id, np.array([x, y, w, h])), ...]}.  For gt.txt pass
    min_conf=1 (column 7 is the "consider" flag) and classes={1} (pedestrian)."""
    out = defaultdict(list)
    data = np.loadtxt(path, delimiter=',', ndmin=2)
    for row in data:
        if min_conf is not None and len(row) > 6 and row[6] < min_conf:
            continue
        if classes is not None and len(row) > 7 and int(row[7]) not in classes:
            continue
        out[int(row[0])].append((int(row[1]), row[2:6].astype(float)))
    return dict(out)


def _iou_matrix(a, b):
    """a: [n,4] tlwh, b: [m,4] tlwh -> IoU [n,m] (continuous coordinates, as the MOTChallenge devkit)."""
    if len(a) == 0 or len(b) == 0:
        return np.zeros((len(a), len(b)))
    ax2, ay2 = a[:, 0] + a[:, 2], a[:, 1] + a[:, 3]
    bx2, by2 = b[:, 0] + b[:, 2], b[:, 1] + b[:, 3]
    iw = np.clip(np.minimum(ax2[:, None], bx2[None]) - np.maximum(a[:, None, 0], b[None, :, 0]), 0, None)
    ih = np.clip(np.minimum(ay2[:, None], by2[None]) - np.maximum(a[:, None, 1], b[None, :, 1]), 0, None)
    inter = iw * ih
    union = (a[:, 2] * a[:, 3])[:, None] + (b[:, 2] * b[:, 3])[None] - inter
    return np.where(union > 0, inter / np.maximum(union, 1e-12), 0.)


def evaluate(gt, res, iou_thresh=0.5):
    """gt, res: {frame: [(id, tlwh)]} -> dict(mota, motp, idf1, idp, idr, fp, fn, idsw, n_gt, n_res, tp)."""
    frames = sorted(set(gt) | set(res))
    fp = fn = idsw = tp = 0
    iou_sum = 0.
    prev = {}                               # gt id -> res id of the previous frame's correspondence
    last = {}                               # gt id -> last res id it was matched to (for switches)
    pair_tp = defaultdict(int)              # (gt id, res id) -> frames matched at IoU >= thresh
    gt_count, res_count = defaultdict(int), defaultdict(int)
    for f in frames:
        g = gt.get(f, [])
        r = res.get(f, [])
        gids, gbox = [i for i, _ in g], np.array([b for _, b in g], float).reshape(-1, 4)
        rids, rbox = [i for i, _ in r], np.array([b for _, b in r], float).reshape(-1, 4)
        for i in gids:
            gt_count[i] += 1
        for i in rids:
            res_count[i] += 1
        iou = _iou_matrix(gbox, rbox)
        # IDF1 bookkeeping: any overlap >= thresh counts for the identity pair
        for a in range(len(gids)):
            for b in range(len(rids)):
                if iou[a, b] >= iou_thresh:
                    pair_tp[gids[a], rids[b]] += 1
        match = {}
        used_r = set()
        # 1) keep still-valid correspondences of the previous frame
        for a, gi in enumerate(gids):
            ri = prev.get(gi)
            if ri is not None and ri in rids:
                b = rids.index(ri)
                if iou[a, b] >= iou_thresh and b not in used_r:
                    match[a] = b
                    used_r.add(b)
        # 2) Hungarian on the rest
        ra = [a for a in range(len(gids)) if a not in match]
        rb = [b for b in range(len(rids)) if b not in used_r]
        if ra and rb:
            cost = 1. - iou[np.ix_(ra, rb)]
            cost[cost > 1. - iou_thresh] = 1e6
            rr, cc = linear_sum_assignment(cost)
            for x, y in zip(rr, cc):
                if cost[x, y] < 1e6:
                    match[ra[x]] = rb[y]
        new_prev = {}
        for a, b in match.items():
            gi, ri = gids[a], rids[b]
            tp += 1
            iou_sum += iou[a, b]
            if gi in last and last[gi] != ri:
                idsw += 1
            last[gi] = ri
            new_prev[gi] = ri
        prev = new_prev
        fn += len(gids) - len(match)
        fp += len(rids) - len(match)
    n_gt, n_res = sum(gt_count.values()), sum(res_count.values())
    # IDF1: global identity matching minimising IDFP + IDFN
    G, R = sorted(gt_count), sorted(res_count)
    idtp = 0
    if G and R:
        n = len(G) + len(R)
        cost = np.zeros((n, n))
        big = 1e9
        cost[:len(G), :len(R)] = [[gt_count[a] + res_count[b] - 2 * pair_tp.get((a, b), 0) for b in R] for a in G]
        cost[:len(G), len(R):] = big
        cost[len(G):, :len(R)] = big
        for i, a in enumerate(G):
            cost[i, len(R) + i] = gt_count[a]           # gt identity left unmatched: all its boxes are IDFN
        for j, b in enumerate(R):
            cost[len(G) + j, j] = res_count[b]          # predicted identity left unmatched: all IDFP
        rr, cc = linear_sum_assignment(cost)
        for x, y in zip(rr, cc):
            if x < len(G) and y < len(R):
                idtp += pair_tp.get((G[x], R[y]), 0)
    idfn, idfp = n_gt - idtp, n_res - idtp
    return dict(mota=1. - (fn + fp + idsw) / n_gt if n_gt else float('nan'),
                motp=iou_sum / tp if tp else float('nan'),
                idf1=2 * idtp / (2 * idtp + idfp + idfn) if (idtp + idfp + idfn) else float('nan'),
                idp=idtp / (idtp + idfp) if (idtp + idfp) else float('nan'),
                idr=idtp / (idtp + idfn) if (idtp + idfn) else float('nan'),
                fp=fp, fn=fn, idsw=idsw, n_gt=n_gt, n_res=n_res, tp=tp)
