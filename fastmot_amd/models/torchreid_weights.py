"""torchreid OSNet checkpoint -> parameter source of the OSNet layer table (models/reid.py).

The reference converts the torchreid model to ONNX offline and builds a TensorRT engine from it
(fastmot/models/reid.py:48-92, README model zoo: osnet_x0_25_msmt17 / osnet_x1_0_msdc); here the
checkpoint's state_dict is read directly.  The mapping below follows torchreid/models/osnet.py:

    conv1                      ConvLayer        conv1.conv / conv1.bn
    conv{2,3,4}.{0,1}          OSBlock          .conv1 (Conv1x1: conv, bn)
                                                .conv2a | .conv2b.{0,1} | .conv2c.{0,1,2} | .conv2d.{0..3}
                                                    LightConv3x3: conv1 (1x1), conv2 (depthwise 3x3), bn
                                                .gate.fc1 / .gate.fc2 (1x1 convs with bias)
                                                .conv3 (Conv1x1Linear: conv, bn), .downsample (same)
    conv{2,3}.2.0              Conv1x1          transition before the 2x2 average pool
    conv5                      Conv1x1
    fc.0 / fc.1                Linear + BatchNorm1d

PyTorch is used for unpickling only (`torch.load`); a plain dict of arrays works as well -- that is how the
reference's ONNX export of the same model arrives (models/onnx_reader.torchreid_state_dict_from_onnx).
"""
import numpy as np

_STREAM = {1: 'conv2a', 2: 'conv2b', 3: 'conv2c', 4: 'conv2d'}


def _key(name):
    """Layer-table parameter name (models/reid.py) -> (state_dict prefix of the conv / linear, prefix of the
    batch norm or None)."""
    parts = name.split('.')
    if name == 'conv1' or name == 'conv5':
        return f'{name}.conv', f'{name}.bn'
    if name == 'fc':
        return 'fc.0', 'fc.1'
    stage = parts[0]                                   # conv2 / conv3 / conv4
    if parts[1] == 't':
        return f'{stage}.2.0.conv', f'{stage}.2.0.bn'
    block = f'{stage}.{parts[1]}'
    tail = parts[2:]
    if tail == ['conv1']:
        return f'{block}.conv1.conv', f'{block}.conv1.bn'
    if tail == ['conv3']:
        return f'{block}.conv3.conv', f'{block}.conv3.bn'
    if tail == ['down']:
        return f'{block}.downsample.conv', f'{block}.downsample.bn'
    if tail[0] == 'gate':
        return f'{block}.gate.{tail[1]}', None
    if tail[0].startswith('s') and tail[-1] in ('pw', 'dw'):
        t, i = int(tail[0][1:]), int(tail[1])
        lc = f'{block}.{_STREAM[t]}' if t == 1 else f'{block}.{_STREAM[t]}.{i}'
        return (f'{lc}.conv1', None) if tail[-1] == 'pw' else (f'{lc}.conv2', f'{lc}.bn')
    raise KeyError(f'no torchreid parameter for layer {name!r}')


class TorchreidWeights:
    def __init__(self, source):
        if isinstance(source, dict):
            sd = source
        else:
            import torch
            sd = torch.load(source, map_location='cpu')
        if 'state_dict' in sd and not any(k.endswith('.weight') for k in sd):
            sd = sd['state_dict']
        self.sd = {(k[7:] if k.startswith('module.') else k): np.asarray(getattr(v, 'numpy', lambda: v)(), np.float32)
                   for k, v in sd.items() if not k.endswith('num_batches_tracked')}
        self.used = set()

    def _get(self, key, shape=None):
        if key not in self.sd:
            raise KeyError(f'checkpoint has no parameter {key!r}')
        self.used.add(key)
        a = self.sd[key]
        if shape is not None and tuple(a.shape) != tuple(shape):
            raise ValueError(f'{key}: checkpoint shape {tuple(a.shape)} != expected {tuple(shape)}')
        return a

    def _bn(self, prefix, c):
        return dict(gamma=self._get(prefix + '.weight', (c,)), beta=self._get(prefix + '.bias', (c,)),
                    mean=self._get(prefix + '.running_mean', (c,)), var=self._get(prefix + '.running_var', (c,)))

    def conv(self, name, cout, cin, k, bn=True, gain=1.0, groups=1):
        ck, bk = _key(name)
        p = dict(w=self._get(ck + '.weight', (cout, cin // groups, k, k)))
        if ck + '.bias' in self.sd:
            p['bias'] = self._get(ck + '.bias', (cout,))
        elif not bn:
            p['bias'] = np.zeros(cout, np.float32)
        if bn:
            if bk is None:
                raise KeyError(f'layer {name!r} has no batch norm in torchreid')
            if bk + '.weight' in self.sd:
                p.update(self._bn(bk, cout))
            elif ck + '.bias' not in self.sd:
                # (a source whose exporter folded BatchNorm into the conv -- ONNX, models/onnx_reader.py -- carries
                # the folded bias instead of the four BatchNorm vectors)
                raise KeyError(f'checkpoint has neither {bk}.* nor a folded {ck}.bias')
        return p

    def linear(self, name, cout, cin, bn=False):
        ck, bk = _key(name)
        p = dict(w=self._get(ck + '.weight', (cout, cin)))
        if ck + '.bias' in self.sd:
            p['bias'] = self._get(ck + '.bias', (cout,))
        if bn and (bk + '.weight' in self.sd or ck + '.bias' not in self.sd):
            p.update(self._bn(bk, cout))
        return p

    def unused(self):
        """Checkpoint entries the layer table did not consume (the classifier is expected)."""
        return sorted(k for k in self.sd if k not in self.used and not k.startswith('classifier'))
