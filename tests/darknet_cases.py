"""TEST INFRASTRUCTURE: small Darknet cfgs exercising every lowering of fastmot_amd/models/darknet.py,
a writer of random `.weights` files in Darknet order and an INDEPENDENT PyTorch interpretation of a cfg
(yolo2onnx.py:558-863 semantics: SAME_LOWER convs == darknet pad, SAME_UPPER maxpool, Add shortcut,
Split/Concat routes, nearest upsample) that shares no code with the graph builder."""
import io

import numpy as np
import torch
import torch.nn.functional as F

MINI_V4 = """
[net]
width=96
height=64
channels=3

[convolutional]
batch_normalize=1
filters=16
size=3
stride=1
pad=1
activation=mish

# CSP stage
[convolutional]
batch_normalize=1
filters=32
size=3
stride=2
pad=1
activation=mish

[convolutional]
batch_normalize=1
filters=16
size=1
stride=1
pad=1
activation=mish

[route]
layers = -2

[convolutional]
batch_normalize=1
filters=16
size=1
stride=1
pad=1
activation=mish

[convolutional]
batch_normalize=1
filters=16
size=1
stride=1
pad=1
activation=mish

[convolutional]
batch_normalize=1
filters=16
size=3
stride=1
pad=1
activation=mish

[shortcut]
from=-3
activation=linear

[convolutional]
batch_normalize=1
filters=16
size=1
stride=1
pad=1
activation=mish

[route]
layers = -1,-7

[convolutional]
batch_normalize=1
filters=32
size=1
stride=1
pad=1
activation=mish

[convolutional]
batch_normalize=1
filters=64
size=3
stride=2
pad=1
activation=leaky

[convolutional]
batch_normalize=1
filters=32
size=1
stride=1
pad=1
activation=leaky

### SPP ###
[maxpool]
stride=1
size=5

[route]
layers=-2

[maxpool]
stride=1
size=9

[route]
layers=-4

[maxpool]
stride=1
size=13

[route]
layers=-1,-3,-5,-6
### End SPP ###

[convolutional]
batch_normalize=1
filters=32
size=1
stride=1
pad=1
activation=leaky

[convolutional]
batch_normalize=1
filters=16
size=1
stride=1
pad=1
activation=leaky

[upsample]
stride=2

[route]
layers = 10

[convolutional]
batch_normalize=1
filters=16
size=1
stride=1
pad=1
activation=leaky

[route]
layers = -1, -3

[convolutional]
batch_normalize=1
filters=32
size=3
stride=1
pad=1
activation=leaky

[convolutional]
size=1
stride=1
pad=1
filters=21
activation=linear

[yolo]
mask = 0,1,2
anchors = 12, 16, 19, 36, 40, 28, 36, 75, 76, 55, 72, 146
classes=2
num=6
scale_x_y = 1.2

[route]
layers = -3

[convolutional]
batch_normalize=1
size=3
stride=2
pad=1
filters=32
activation=leaky

[route]
layers = -1, 19

[convolutional]
batch_normalize=1
filters=64
size=3
stride=1
pad=1
activation=leaky

[convolutional]
size=1
stride=1
pad=1
filters=21
activation=linear

[yolo]
mask = 3,4,5
anchors = 12, 16, 19, 36, 40, 28, 36, 75, 76, 55, 72, 146
classes=2
num=6
scale_x_y = 1.1
"""

MINI_TINY = """
[net]
width=64
height=64
channels=3

[convolutional]
batch_normalize=1
filters=16
size=3
stride=2
pad=1
activation=leaky

[convolutional]
batch_normalize=1
filters=32
size=3
stride=1
pad=1
activation=leaky

[route]
layers=-1
groups=2
group_id=1

[convolutional]
batch_normalize=1
filters=16
size=3
stride=1
pad=1
activation=leaky

[convolutional]
batch_normalize=1
filters=16
size=3
stride=1
pad=1
activation=leaky

[route]
layers = -1,-2

[convolutional]
batch_normalize=1
filters=32
size=1
stride=1
pad=1
activation=leaky

[route]
layers = -6,-1

[maxpool]
size=2
stride=2

[convolutional]
batch_normalize=1
filters=32
size=3
stride=1
pad=1
activation=swish

[maxpool]
size=2
stride=1

# shortcut that cannot fold: the conv is also read by the route below
[convolutional]
batch_normalize=1
filters=32
size=3
stride=1
pad=1
activation=leaky

[shortcut]
from=-2
activation=linear

[route]
layers = -2, -1

# upsample that cannot fold: its source is read again
[convolutional]
batch_normalize=1
filters=16
size=1
stride=1
pad=1
activation=leaky

[upsample]
stride=2

[route]
layers = -2

[convolutional]
size=1
stride=1
pad=1
filters=14
activation=logistic

[yolo]
mask = 0,1
anchors = 10,14, 23,27, 37,58
classes=2
num=3
new_coords=1

[route]
layers = -5

[convolutional]
size=1
stride=1
pad=1
filters=14
activation=logistic

[yolo]
mask = 1,2
anchors = 10,14, 23,27, 37,58
classes=2
num=3
new_coords=1
"""


def _conv(filters, size, stride=1, act='mish', bn=1):
    return (f'[convolutional]\n' + ('batch_normalize=1\n' if bn else '') +
            f'filters={filters}\nsize={size}\nstride={stride}\npad=1\nactivation={act}\n\n')


# CSP stage whose residual units are wide enough for the fused kernel (64 channels; mid 32 and 64); the
# last unit's output is a concat operand (written in place), the second shortcut uses the list form
MINI_RES = ('[net]\nwidth=48\nheight=40\nchannels=3\n\n' + _conv(64, 3, 2) + _conv(64, 1) + '[route]\nlayers = -2\n\n' +
            _conv(64, 1) + _conv(32, 1) + _conv(64, 3) + '[shortcut]\nfrom=-3\nactivation=linear\n\n' +
            _conv(64, 1) + _conv(64, 3) + '[shortcut]\nfrom=-3\nactivation=linear\n\n' +
            '[route]\nlayers = -1,-9\n\n' + _conv(21, 1, act='linear', bn=0) +
            '[yolo]\nmask = 0,1,2\nanchors = 10,14, 23,27, 37,58\nclasses=2\nnum=3\n')


def yolov4_cfg(width=64, height=64, classes=3):
    """yolov4.cfg (AlexeyAB/darknet: CSPDarknet53 + SPP + PAN, 110 [convolutional] sections) generated
    section by section; relative route/shortcut indices are computed from the recorded section indices."""
    out, idx = [f'[net]\nwidth={width}\nheight={height}\nchannels=3\n\n'], [-1]

    def add(text):
        out.append(text)
        idx[0] += 1
        return idx[0]

    def conv(f, k, s=1, act='mish', bn=1):
        return add(_conv(f, k, s, act, bn))

    def route(*targets):          # absolute section indices -> relative
        cur = idx[0] + 1
        return add('[route]\nlayers = ' + ','.join(str(t - cur) for t in targets) + '\n\n')

    conv(32, 3)
    stage_out = []
    for down, h, m, n in ((64, 64, 32, 1), (128, 64, 64, 2), (256, 128, 128, 8), (512, 256, 256, 8), (1024, 512, 512, 4)):
        d = conv(down, 3, 2)
        a = conv(h, 1)
        route(d)
        conv(h, 1)
        for _ in range(n):
            conv(m, 1)
            conv(h, 3)
            add('[shortcut]\nfrom=-3\nactivation=linear\n\n')
        post = conv(h, 1)
        route(post, a)
        stage_out.append(conv(down, 1))
    lk = dict(act='leaky')
    conv(512, 1, **lk); conv(1024, 3, **lk); x = conv(512, 1, **lk)
    p5 = add('[maxpool]\nstride=1\nsize=5\n\n'); route(x)
    p9 = add('[maxpool]\nstride=1\nsize=9\n\n'); route(x)
    p13 = add('[maxpool]\nstride=1\nsize=13\n\n'); route(p13, p9, p5, x)
    conv(512, 1, **lk); conv(1024, 3, **lk); n19 = conv(512, 1, **lk)
    tops = [n19]
    for f, lateral in ((256, stage_out[3]), (128, stage_out[2])):
        conv(f, 1, **lk)
        up = add('[upsample]\nstride=2\n\n')
        route(lateral)
        lat = conv(f, 1, **lk)
        route(lat, up)
        for _ in range(2):
            conv(f, 1, **lk); conv(2 * f, 3, **lk)
        tops.append(conv(f, 1, **lk))
    anchors = 'anchors = 12,16, 19,36, 40,28, 36,75, 76,55, 72,146, 142,110, 192,243, 459,401'

    def head(f, mask, scale):
        conv(2 * f, 3, **lk)
        conv((classes + 5) * 3, 1, act='linear', bn=0)
        add(f'[yolo]\nmask = {mask}\n{anchors}\nclasses={classes}\nnum=9\nscale_x_y = {scale}\n\n')

    head(128, '0,1,2', 1.2)
    prev = tops[2]
    for f, lateral, mask, scale in ((256, tops[1], '3,4,5', 1.1), (512, tops[0], '6,7,8', 1.05)):
        route(prev)
        dn = conv(f, 3, 2, **lk)
        route(dn, lateral)
        for _ in range(2):
            conv(f, 1, **lk); conv(2 * f, 3, **lk)
        prev = conv(f, 1, **lk)
        head(f, mask, scale)
    return ''.join(out)


def conv_sections(cfg_layers):
    """(filters, cin, k, bn) of every [convolutional] in cfg order, by an independent shape walk."""
    net, layers = cfg_layers[0], cfg_layers[1:]
    ch = []
    out = []
    for i, L in enumerate(layers):
        t = L['type']
        prev = ch[i - 1] if i else int(net.get('channels', 3))
        if t == 'convolutional':
            out.append((int(L['filters']), prev, int(L.get('size', 1)), int(L.get('batch_normalize', 0)) == 1))
            ch.append(int(L['filters']))
        elif t == 'route':
            srcs = [i + int(r) if int(r) < 0 else int(r) for r in L['layers']]
            c = sum(ch[j] for j in srcs)
            if 'groups' in L:
                c //= int(L['groups'])
            ch.append(c)
        else:
            ch.append(prev)
    return out


def random_weights_file(cfg_layers, seed=0):
    rng = np.random.default_rng(seed)
    buf = io.BytesIO()
    buf.write(np.array([0, 2, 5, 0, 0], np.int32).tobytes())
    for cout, cin, k, bn in conv_sections(cfg_layers):
        if bn:
            buf.write(rng.normal(0, 0.1, cout).astype(np.float32).tobytes())       # beta (bias)
            buf.write(rng.uniform(0.8, 1.2, cout).astype(np.float32).tobytes())    # gamma (scale)
            buf.write(rng.normal(0, 0.1, cout).astype(np.float32).tobytes())       # mean
            buf.write(rng.uniform(0.8, 1.2, cout).astype(np.float32).tobytes())    # var
        else:
            buf.write(rng.normal(0, 0.1, cout).astype(np.float32).tobytes())
        buf.write(rng.normal(0, np.sqrt(1.0 / (cin * k * k)), (cout, cin, k, k)).astype(np.float32).tobytes())
    return buf.getvalue()


def torch_darknet(cfg_layers, weights_bytes, x):
    """x: [N, C, H, W] float32 -> list of head tensors (inputs of the [yolo] sections)."""
    net, layers = cfg_layers[0], cfg_layers[1:]
    data = np.frombuffer(weights_bytes, np.float32, offset=20)
    pos = 0

    def take(n):
        nonlocal pos
        v = torch.from_numpy(data[pos:pos + n].copy())
        pos += n
        return v

    outs, heads = [], []
    for i, L in enumerate(layers):
        t = L['type']
        prev = outs[i - 1] if i else x
        if t == 'convolutional':
            cout, k, s = int(L['filters']), int(L.get('size', 1)), int(L.get('stride', 1))
            cin = prev.shape[1]
            if int(L.get('batch_normalize', 0)):
                beta, gamma, mean, var = take(cout), take(cout), take(cout), take(cout)
                w = take(cout * cin * k * k).reshape(cout, cin, k, k)
                y = F.conv2d(prev, w, None, s, k // 2 if int(L.get('pad', 0)) else 0)
                y = gamma[None, :, None, None] * (y - mean[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + 1e-5) \
                    + beta[None, :, None, None]
            else:
                b = take(cout)
                w = take(cout * cin * k * k).reshape(cout, cin, k, k)
                y = F.conv2d(prev, w, b, s, k // 2 if int(L.get('pad', 0)) else 0)
            a = L.get('activation', 'linear')
            if a == 'leaky':
                y = F.leaky_relu(y, 0.1)
            elif a == 'mish':
                y = y * torch.tanh(F.softplus(y))
            elif a == 'swish':
                y = y * torch.sigmoid(y)
            elif a == 'logistic':
                y = torch.sigmoid(y)
            else:
                assert a == 'linear'
            outs.append(y)
        elif t == 'maxpool':
            k, s = int(L['size']), int(L.get('stride', 1))
            h, w_ = prev.shape[2:]
            th = max((-(-h // s) - 1) * s + k - h, 0)
            tw = max((-(-w_ // s) - 1) * s + k - w_, 0)
            xp = F.pad(prev, (tw // 2, tw - tw // 2, th // 2, th - th // 2), value=float('-inf'))
            outs.append(F.max_pool2d(xp, k, s))
        elif t == 'upsample':
            outs.append(F.interpolate(prev, scale_factor=int(L.get('stride', 2)), mode='nearest'))
        elif t == 'shortcut':
            j = int(L['from'])
            outs.append(prev + outs[i + j if j < 0 else j])
        elif t == 'route':
            srcs = [i + int(r) if int(r) < 0 else int(r) for r in L['layers']]
            if len(srcs) == 1:
                v = outs[srcs[0]]
                if 'groups' in L:
                    c = v.shape[1] // int(L['groups'])
                    v = v[:, int(L['group_id']) * c:(int(L['group_id']) + 1) * c]
                outs.append(v)
            else:
                outs.append(torch.cat([outs[j] for j in srcs], 1))
        elif t == 'yolo':
            heads.append(prev)
            outs.append(prev)
    assert pos == len(data)
    return heads
