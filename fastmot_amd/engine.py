"""HipNet: a layer table (models/graph.py) instantiated on the device (fm_net_* of the C ABI).
Counterpart of TRTInference (fastmot/utils/inference.py:39-125)."""
import ctypes as C

import numpy as np

from . import _lib

NET_DETECTOR, NET_EXTRACTOR, NET_EXTRACTOR_B = 0, 1, 2


class HipNet:
    def __init__(self, ctx, which, graph, max_batch, reuse_buffers=False):
        """reuse_buffers: share activation memory between tensors with disjoint live ranges (the
        production setting; intermediate tensors can then not be read back after a run)."""
        self.ctx, self.which, self.graph, self.max_batch = ctx, which, graph, max_batch
        ts, ls, blob = graph.tables(max_batch, reuse_buffers)
        self._keep = (ts, ls, blob)
        _lib.check(ctx.lib.fm_net_create(ctx.handle, C.c_int(which), C.c_int(max_batch), C.c_int(len(ts)), ts,
                                         C.c_int(len(ls)), ls, blob, C.c_size_t(len(blob)),
                                         C.c_int(graph.n_gates), C.c_int(graph.gate_c),
                                         C.c_size_t(graph.arena_bytes)))

    def run(self, batch):
        _lib.check(self.ctx.lib.fm_net_run(self.ctx.handle, C.c_int(self.which), C.c_int(batch)))

    def write(self, view, array):
        """array: [batch, h, w, c_logical] float -> stored NHWC fp16 with zero channel padding
        (only whole tensors: view.coff must be 0)."""
        h, w, cpad, f32 = self.graph.tensors[view.tid]
        assert view.coff == 0 and not f32
        a = np.asarray(array)
        full = np.zeros((a.shape[0], h, w, cpad), np.float16)
        full[..., :a.shape[-1]] = a
        _lib.check(self.ctx.lib.fm_net_tensor_write(self.ctx.handle, C.c_int(self.which), C.c_int(view.tid),
                                                    full.ctypes.data_as(C.c_void_p), C.c_size_t(full.nbytes)))

    def read(self, view, batch):
        h, w, cpad, f32 = self.graph.tensors[view.tid]
        full = np.empty((batch, h, w, cpad), np.float32 if f32 else np.float16)
        _lib.check(self.ctx.lib.fm_net_tensor_read(self.ctx.handle, C.c_int(self.which), C.c_int(view.tid),
                                                   full.ctypes.data_as(C.c_void_p), C.c_size_t(full.nbytes)))
        return full[..., view.coff:view.coff + view.c].astype(np.float32)

    def read_embeddings(self, n):
        out = np.empty((n, self.ctx.feat_dim), np.float32)
        _lib.check(self.ctx.lib.fm_net_read_embeddings(self.ctx.handle, C.c_int(n), out.ctypes.data_as(C.c_void_p)))
        return out

    def cost(self, batch):
        f, b = C.c_double(0), C.c_double(0)
        _lib.check(self.ctx.lib.fm_net_cost(self.ctx.handle, C.c_int(self.which), C.c_int(batch), C.byref(f), C.byref(b)))
        return f.value, b.value

    def profile(self, batch, iters=3):
        conv_ms, other_ms = C.c_double(0), C.c_double(0)
        n_conv, n_other = C.c_int(0), C.c_int(0)
        _lib.check(self.ctx.lib.fm_net_profile(self.ctx.handle, C.c_int(self.which), C.c_int(batch), C.c_int(iters),
                                               C.byref(conv_ms), C.byref(other_ms), C.byref(n_conv), C.byref(n_other)))
        return dict(conv_ms=conv_ms.value, other_ms=other_ms.value, n_conv=n_conv.value, n_other=n_other.value)

    def profile_layers(self, batch, iters=5):
        out = np.zeros(len(self.graph.layers))
        _lib.check(self.ctx.lib.fm_net_profile_layers(self.ctx.handle, C.c_int(self.which), C.c_int(batch), C.c_int(iters),
                                                      out.ctypes.data_as(C.c_void_p)))
        return out

    def close(self):
        _lib.check(self.ctx.lib.fm_net_destroy(self.ctx.handle, C.c_int(self.which)))
