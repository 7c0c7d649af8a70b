"""TEST INFRASTRUCTURE: plain PyTorch fp32 (CPU, NCHW) interpreter of a models/graph.py layer table
-- the reference the conv-engine kernels are compared against (the reference's own conv arithmetic
is TensorRT's, which is neither available nor pinned).
`emulate_fp16_storage=True` rounds every layer output to fp16, as the engine stores activations."""
import numpy as np
import torch
import torch.nn.functional as F

from fastmot_amd.models import graph as G


def act_fn(x, act):
    if act == G.ACT['leaky']:
        return F.leaky_relu(x, 0.1)
    if act == G.ACT['mish']:
        return x * torch.tanh(F.softplus(x))
    if act == G.ACT['relu']:
        return F.relu(x)
    if act == G.ACT['logistic']:
        return torch.sigmoid(x)
    if act == G.ACT['swish']:
        return x * torch.sigmoid(x)
    return x


def run_graph(graph, x_nchw, emulate_fp16_storage=True):
    """x_nchw: float tensor [N, C, H, W].  Returns dict tid -> tensor [N, cpad, h, w] and the
    embedding matrix if the graph has a head."""
    n = x_nchw.shape[0]
    bufs = {}
    for tid, (h, w, c, f32) in enumerate(graph.tensors):
        bufs[tid] = torch.zeros(n, c, h, w)
    x = x_nchw.float()
    if emulate_fp16_storage:
        x = x.half().float()
    bufs[graph.input.tid][:, :x.shape[1]] = x
    params = {idx: (w, b) for idx, w, b in graph.conv_params}
    gates = {}
    emb = None

    def rd(v):
        return bufs[v.tid][:, v.coff:v.coff + v.c]

    def wr(v, val, f32=False):
        if emulate_fp16_storage and not f32:
            val = val.half().float()
        bufs[v.tid][:, v.coff:v.coff + v.c] = val

    def step(d, wb):
        nonlocal emb
        op = d['op']
        xin = rd(d['ins'][0])
        if op == G.OP_OSTAIL:            # the layers the fused launch replaces, with their own references
            for sd, swb in d['sub']:
                step(sd, swb)
        elif op in G.CONV_OPS + (G.OP_STEMCONV,):
            w, b = wb
            y = F.conv2d(xin, torch.from_numpy(w), torch.from_numpy(b), stride=d['stride'], padding=d['pad'])
            if d['res_mode'] == G.RES_BEFORE_ACT:
                y = y + rd(d['res'])
            y = act_fn(y, d['act'])
            if d['res_mode'] == G.RES_AFTER_ACT:
                y = y + rd(d['res'])
            if d['up'] == 2:
                y = F.interpolate(y, scale_factor=2, mode='nearest')
            wr(d['out'], y, f32=bool(graph.tensors[d['out'].tid][3]))
        elif op == G.OP_RESBLOCK:
            w1, b1, w2, b2 = (torch.from_numpy(np.asarray(a, np.float32)) for a in d['res_ref'])
            y = act_fn(F.conv2d(xin, w1, b1), d['act'])
            if emulate_fp16_storage:
                y = y.half().float()
            wr(d['out'], act_fn(F.conv2d(y, w2, b2, padding=1), d['act']) + xin)
        elif op == G.OP_PAIR11:
            w1, b1, act1, w2, b2 = d['pair_ref']
            t = act_fn(F.conv2d(xin, torch.from_numpy(np.asarray(w1, np.float32)), torch.from_numpy(np.asarray(b1, np.float32))), act1)
            if emulate_fp16_storage:
                t = t.half().float()
            cat = torch.cat([t, rd(d['ins'][1])], dim=1)
            wr(d['out'], act_fn(F.conv2d(cat, torch.from_numpy(np.asarray(w2, np.float32)), torch.from_numpy(np.asarray(b2, np.float32))),
                                d['act']))
        elif op == G.OP_STEM2:
            w1, b1, act1, w2, b2 = d['stem2_ref']
            three = 'stem3_ref' in d
            y = act_fn(F.conv2d(xin, torch.from_numpy(np.asarray(w1, np.float32)), torch.from_numpy(np.asarray(b1, np.float32)),
                                padding=1), act1)
            if emulate_fp16_storage:
                y = y.half().float()
            y = act_fn(F.conv2d(y, torch.from_numpy(np.asarray(w2, np.float32)), torch.from_numpy(np.asarray(b2, np.float32)),
                                stride=2, padding=1), d['stem3_ref'][0] if three else d['act'])
            if three:
                if emulate_fp16_storage:
                    y = y.half().float()
                _, w3, b3 = d['stem3_ref']
                y = act_fn(F.conv2d(y, torch.from_numpy(np.asarray(w3, np.float32)), torch.from_numpy(np.asarray(b3, np.float32))), d['act'])
            wr(d['out'], y)
        elif op == G.OP_DWCONV3:
            w, b = wb
            y = F.conv2d(xin, torch.from_numpy(w), torch.from_numpy(b), padding=1, groups=xin.shape[1])
            wr(d['out'], act_fn(y, d['act']))
        elif op == G.OP_LITECONV:
            c = d['cout']
            for gi, (v, ref) in enumerate(zip(d['ins'], d['lite_ref'])):
                pw, wd, bd = (torch.from_numpy(np.asarray(a, np.float32)) for a in ref)
                y = F.conv2d(rd(v), pw)
                if emulate_fp16_storage:
                    y = y.half().float()
                y = F.conv2d(y, wd, bd, padding=1, groups=y.shape[1])
                wr(d['out'].slice(gi * c, c), act_fn(y, d['act']))
        elif op == G.OP_LITECHAIN:
            c = d['cout']
            refs = iter(d['lite_ref'])
            for t in range(4):
                y = xin
                for _ in range(t + 1):
                    pw, wd, bd = (torch.from_numpy(np.asarray(a, np.float32)) for a in next(refs))
                    y = F.conv2d(y, pw)
                    if emulate_fp16_storage:
                        y = y.half().float()
                    y = act_fn(F.conv2d(y, wd, bd, padding=1, groups=y.shape[1]), d['act'])
                    if emulate_fp16_storage:
                        y = y.half().float()
                wr(d['out'].slice(t * c, c), y)
        elif op == G.OP_GATED_SUM:
            w1, b1, w2, b2 = (torch.from_numpy(np.asarray(a, np.float32)) for a in d['gate_ref'])
            y = 0
            for v in d['ins']:
                xv = rd(v)
                hid = F.relu(xv.mean(dim=(2, 3)) @ w1.T + b1)
                y = y + xv * torch.sigmoid(hid @ w2.T + b2)[:, :, None, None]
            wr(d['out'], y)
        elif op == G.OP_SPP:
            c = d['cout']
            for i, k in enumerate((13, 9, 5)):
                wr(d['out'].slice(i * c, c), F.max_pool2d(xin, k, 1, k // 2))
        elif op == G.OP_MAXPOOL:
            pe = d.get('pad_end', d['pad'])
            xp = F.pad(xin, (d['pad'], pe, d['pad'], pe), value=float('-inf'))
            wr(d['out'], F.max_pool2d(xp, d['k'], d['stride'], 0))
        elif op == G.OP_AVGPOOL:
            wr(d['out'], F.avg_pool2d(xin, d['k'], d['stride'], d['pad']))
        elif op == G.OP_UPSAMPLE2:
            wr(d['out'], F.interpolate(xin, scale_factor=2, mode='nearest'))
        elif op == G.OP_ADD:
            wr(d['out'], xin + rd(d['ins'][1]))
        elif op == G.OP_COPY:
            wr(d['out'], xin)
        elif op == G.OP_GATE:
            w1, b1, w2, b2 = (torch.from_numpy(np.asarray(a, np.float32)) for a in d['gate_ref'])
            gap = xin.mean(dim=(2, 3))
            hid = F.relu(gap @ w1.T + b1)
            gates[d['gates'][0]] = torch.sigmoid(hid @ w2.T + b2)
        elif op == G.OP_GATE_SUM:
            y = 0
            for v, gid in zip(d['ins'], d['gates']):
                y = y + rd(v) * gates[gid][:, :, None, None]
            wr(d['out'], y)
        elif op == G.OP_HEAD:
            w, b = (torch.from_numpy(np.asarray(a, np.float32)) for a in d['head_ref'])
            feat = F.relu(xin.mean(dim=(2, 3)) @ w.T + b)
            emb = feat / feat.norm(dim=1, keepdim=True)
        else:
            raise ValueError(op)

    for idx, d in enumerate(graph.layers):
        step(d, params.get(idx))
    return bufs, emb
