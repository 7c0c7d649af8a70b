"""Two-branch schedule of the detector's layer table (models/graph.py plan_branches / happens_before / plan_arena):
CPU-side checks of the plan itself -- every memory dependency is ordered, tensors that share arena bytes are never
live at the same time under the plan, the heads really run beside the PAN path.  (That the results do not change is
a GPU matter: tests/test_fullsize_gpu.py compares every tensor of YOLOv4@608 with PyTorch under the plan, and
test_branches_change_nothing_gpu compares the plan with the single chain bit for bit.)"""
import itertools

import numpy as np
import pytest

from fastmot_amd.models import YOLO
from fastmot_amd.models.graph import Graph


def footprints(g, i):
    d = g.layers[i]
    reads = [(v.tid, v.coff, v.coff + v.cpad) for v in d['ins']]
    if d['res'] is not None:
        reads.append((d['res'].tid, d['res'].coff, d['res'].coff + d['res'].cpad))
    o = d['out']
    return reads, [(o.tid, o.coff, o.coff + o.cpad)]


@pytest.mark.parametrize('name', ['YOLOv4_608', 'YOLOv4CSP_640', 'YOLOv4P6_1280', 'YOLOv4'])
def test_plan_orders_every_dependency_and_every_shared_byte(name):
    g, heads = YOLO.get_model(name).build_graph()
    ts, ls, _ = g.tables(1, True, branches=True)
    plan = g.branch_plan
    assert plan is not None and sum(b for b, _, _ in plan) >= 4            # something runs on the second branch
    before = Graph.happens_before(plan)
    n = len(plan)
    # (1) structural sanity of the wait / signal fields
    for i, (br, wait, sig) in enumerate(plan):
        if wait >= 0:
            assert wait < i and plan[wait][0] != br and plan[wait][2] == 1
    # (2) same-tensor hazards are ordered
    for b in range(n):
        rb, wb = footprints(g, b)
        for a in range(b):
            ra, wa = footprints(g, a)
            hit = any(x[0] == y[0] and x[1] < y[2] and y[1] < x[2] for x in wb for y in ra + wa) or \
                  any(x[0] == y[0] and x[1] < y[2] and y[1] < x[2] for x in rb for y in wa)
            if hit:
                assert a in before[b], (name, a, b)
    # (3) tensors sharing arena bytes: all accesses of one precede all accesses of the other
    acc = {}
    for i in range(n):
        r, w = footprints(g, i)
        for t, _, _ in r + w:
            acc.setdefault(t, []).append(i)
    size = [h * w * c * (4 if f32 else 2) for (h, w, c, f32) in g.tensors]
    off = [ts[t].offset for t in range(len(g.tensors))]
    shared = 0
    for t, u in itertools.combinations(sorted(acc), 2):
        if off[t] < off[u] + size[u] and off[u] < off[t] + size[t]:
            shared += 1
            assert all(a in before[b] for a in acc[t] for b in acc[u]) or \
                   all(b in before[a] for a in acc[t] for b in acc[u]), (name, t, u)
    assert shared > 10                                                    # (the arena does reuse memory)
    # (4) every head's conv runs beside layers of the other branch: neither precedes the other
    head_layers = [i for i in range(n) if g.layers[i]['out'].tid in {h.tid for h in heads}]
    assert len(head_layers) == len(heads)
    parallel = 0
    for h in head_layers[:-1]:
        others = [j for j in range(n) if plan[j][0] != plan[h][0] and j not in before[h] and h not in before[j]]
        parallel += bool(others)
    assert parallel == len(heads) - 1


def test_single_chain_when_nothing_is_independent():
    from fastmot_amd.models.graph import RandomWeights
    g = Graph(RandomWeights(seed=1), (32, 32), 8)
    x = g.conv('a', g.input, 16, 3, 1, 'leaky')
    x = g.conv('b', x, 16, 1, 1, 'leaky')
    g.outputs = [g.conv('c', x, 16, 3, 1, 'leaky')]
    g.tables(1, True, branches=True)
    assert g.branch_plan is None


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['YOLOv4_608', 'YOLOv4CSP_640'])
def test_branches_change_nothing_gpu(ctx, name, monkeypatch):
    """The same layer table as one chain and as two branches (captured into the hipGraph as parallel paths, and launched
    eagerly): head tensors bit-identical, repeatedly."""
    from fastmot_amd.engine import HipNet, NET_DETECTOR
    from fastmot_amd.models.graph import RandomWeights
    model = YOLO.get_model(name)
    _, H, W = model.INPUT_SHAPE
    x = np.random.default_rng(1).uniform(0, 1, (1, H, W, 3)).astype(np.float16)
    outs = {}
    for mode in ('0', '1'):
        monkeypatch.setenv('FASTMOT_BRANCHES', mode)
        g, heads = model.build_graph(RandomWeights(seed=3))
        net = HipNet(ctx, NET_DETECTOR, g, 1, reuse_buffers=True)
        assert (g.branch_plan is not None) == (mode == '1')
        runs = []
        for graphs in (1, 1, 0, 1):                    # capture, replay, eager, replay
            ctx.set_option('use_graphs', graphs)
            net.write(g.input, x)
            net.run(1)
            runs.append([net.read(h, 1).copy() for h in heads])
        ctx.set_option('use_graphs', 1)
        net.close()
        for r in runs[1:]:
            for a, b in zip(runs[0], r):
                np.testing.assert_array_equal(a, b)
        outs[mode] = runs[0]
    for a, b in zip(outs['0'], outs['1']):
        assert np.isfinite(a).all() and np.abs(a).max() > 0
        np.testing.assert_array_equal(a, b)
