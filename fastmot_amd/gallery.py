"""Cross-stream ReID gallery exchange (multi-GPU, one process per GPU / video stream).

NOT in the reference (which is single process, single stream).  Streams shard naturally one per GPU
with no data-path collective; the only optional exchange is an all-gather of every rank's lost-track
gallery (MultiTracker.hist_tracks: <= history_size entries of {track id, label, feature count,
512-d average feature}, ~104 KB per rank) so that an identity that left camera A can be re-identified
on camera B.  It runs through the library's C ABI (fm_gallery_*: RCCL's ncclAllGather over xGMI, bound in
csrc/gallery.hip, no torch at run time); the multi-process CPU tests inject a torch.distributed "gloo" communicator.  The payload is latency bound (<= 832 KB for 8 ranks): ONE fixed-size all_gather
per exchange, strictly opt-in -- with it disabled every stream's results are bit-identical to a
single-GPU run.

Off the critical path: `exchange()` ENQUEUES this detector frame's all-gather on a side stream (RCCL) and
returns the entries gathered by the PREVIOUS exchange, whose completion event has long fired; the tracker
therefore never blocks on the collective (one detector frame of staleness for foreign identities, which
left the other camera many frames ago anyway).

Wire format (one uint8 row block per rank, fixed size): header int64[4] = {n_entries, done, 0, 0};
meta int64[history_size][3] = {track id, label, feature count} -- IDs travel as int64, never through
float32; feats float32[history_size][dim].

End of stream: ranks must issue the same number of collectives.  A rank whose stream has ended calls
`close()`, which keeps contributing empty galleries with done = 1 until the gathered headers show every
rank done; ranks that are still tracking simply see an empty gallery from the finished ones.  (A rank that
dies without close() surfaces as the process group's timeout, `init_process_group(timeout=...)`.)

Foreign entries are appended AFTER the local history rows of the ReID cost matrix, so local
tie-breaks (greedy first-minimum order, tracker.py:229-241) are unchanged; an entry that has been
matched is consumed (`consume`) and never offered again.
"""
import os
import socket
import sys
import time

import numpy as np


class RcclComm:
    """The collective through the library's C ABI (fm_gallery_*: csrc/gallery.hip binds librccl.so, ncclAllGather on a
    side stream with pinned staging) -- no torch in the process.  Bootstrap: rank 0 creates the 128-byte communicator
    id and serves it to the other ranks over one TCP connection each (MASTER_ADDR : MASTER_PORT + 29 + channel, the
    rendezvous variables torchrun / any launcher already exports); world 1 needs no channel.
    channel 0 = the gallery, 1 = control messages (`barrier`, `allgather_small`)."""
    name = 'rccl'
    on_gpu = True

    def __init__(self, ctx, row_bytes, rank=None, world=None, addr=None, port=None, unique_id=None, timeout=120.,
                 channel=0):
        if 'torch' in sys.modules:
            raise RuntimeError('RcclComm needs a process without torch: RCCL binds the HSA runtime by bare library '
                               'name and would pick the copy torch ships (use TorchComm there)')
        self.ctx, self.channel = ctx, channel
        self.rank = int(os.environ.get('RANK', 0)) if rank is None else rank
        self.world = int(os.environ.get('WORLD_SIZE', 1)) if world is None else world
        self.row_bytes = row_bytes
        if unique_id is None:
            addr = addr or os.environ.get('MASTER_ADDR', '127.0.0.1')
            port = int(os.environ.get('MASTER_PORT', 29500)) + 29 + channel if port is None else port
            unique_id = self._bootstrap(addr, port, timeout)
        ctx.gallery_init(channel, self.world, self.rank, unique_id, row_bytes)

    def _bootstrap(self, addr, port, timeout):
        if self.rank == 0:
            uid = self.ctx.gallery_unique_id()
            if self.world > 1:
                with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as srv:
                    srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                    srv.bind((addr, port))
                    srv.listen(self.world)
                    srv.settimeout(timeout)
                    for _ in range(self.world - 1):
                        conn, _ = srv.accept()
                        with conn:
                            conn.sendall(uid)
            return uid
        deadline = time.time() + timeout
        while True:
            try:
                with socket.create_connection((addr, port), timeout=5) as c:
                    buf = b''
                    while len(buf) < 128:
                        chunk = c.recv(128 - len(buf))
                        if not chunk:
                            raise ConnectionError('rank 0 closed the bootstrap connection')
                        buf += chunk
                    return buf
            except (ConnectionError, OSError):
                if time.time() > deadline:
                    raise
                time.sleep(0.05)

    def issue(self, row):
        self.ctx.gallery_allgather_async(self.channel, row)

    def complete(self):
        """-> (uint8[world * row_bytes], milliseconds the exchange occupied its stream)"""
        return self.ctx.gallery_allgather_wait(self.channel, self.world, self.row_bytes)

    def allgather_small(self, values):
        """float64 values (row_bytes / 8 of them at most) of every rank -> array [world, n]; also a barrier."""
        row = np.zeros(self.row_bytes // 8, np.float64)
        v = np.atleast_1d(np.asarray(values, np.float64))
        row[:len(v)] = v
        self.issue(row.view(np.uint8))
        rows, _ = self.complete()
        return rows.view(np.float64).reshape(self.world, -1)[:, :len(v)]

    def barrier(self):
        self.allgather_small([0.0])

    def close(self):
        self.ctx.gallery_destroy(self.channel)


class TorchComm:
    """The same exchange through torch.distributed, for processes that run torch anyway: backend gloo (the
    multi-process CPU tests) or nccl (= RCCL of the ROCm copy torch ships: side stream, pinned staging)."""

    def __init__(self, row_bytes, group=None):
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError('torch.distributed must be initialised for TorchComm')
        self.torch, self.dist, self.group = torch, dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        backend = dist.get_backend(group)
        self.on_gpu = backend == 'nccl'
        self.name = 'rccl (torch.distributed)' if self.on_gpu else backend
        self._send = torch.zeros(row_bytes, dtype=torch.uint8, pin_memory=self.on_gpu)
        self._recv = torch.zeros(self.world * row_bytes, dtype=torch.uint8, pin_memory=self.on_gpu)
        if self.on_gpu:
            dev = torch.device('cuda', torch.cuda.current_device())
            self._send_dev = torch.zeros(row_bytes, dtype=torch.uint8, device=dev)
            self._recv_dev = torch.zeros(self.world * row_bytes, dtype=torch.uint8, device=dev)
            self._stream = torch.cuda.Stream(device=dev)
            self._ev0 = torch.cuda.Event(enable_timing=True)
            self._ev1 = torch.cuda.Event(enable_timing=True)
        self._work = None

    def issue(self, row):
        torch = self.torch
        self._send.numpy()[:] = row
        if self.on_gpu:
            with torch.cuda.stream(self._stream):
                self._ev0.record(self._stream)
                self._send_dev.copy_(self._send, non_blocking=True)
                self.dist.all_gather_into_tensor(self._recv_dev, self._send_dev, group=self.group)
                self._recv.copy_(self._recv_dev, non_blocking=True)
                self._ev1.record(self._stream)
            self._work = self._ev1
        else:
            self._work = self.dist.all_gather_into_tensor(self._recv, self._send, group=self.group, async_op=True)

    def complete(self):
        ms = None
        if self.on_gpu:
            self._work.synchronize()
            ms = self._ev0.elapsed_time(self._ev1)
        else:
            self._work.wait()
        self._work = None
        return self._recv.numpy(), ms

    def close(self):
        pass


class GallerySync:
    def __init__(self, history_size=50, feat_dim=512, period=1, group=None, asynchronous=True, comm=None):
        """comm: the communicator (RcclComm / TorchComm / anything with rank, world, issue(row), complete()).  Default:
        RcclComm on this process's context with the launcher's RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT; TorchComm
        when the process has initialised torch.distributed (gloo: the CPU tests)."""
        self.history_size, self.feat_dim, self.period = history_size, feat_dim, max(int(period), 1)
        H, D = history_size, feat_dim
        self._hdr = 32
        self._meta_bytes = H * 3 * 8
        self.row_bytes = self._hdr + self._meta_bytes + H * D * 4
        if comm is None:
            comm = self._default_comm(group)
        self.comm = comm
        self.rank, self.world = comm.rank, comm.world
        self.backend = comm.name
        self.on_gpu = comm.on_gpu
        self.asynchronous = asynchronous
        self._calls = 0
        self._consumed = set()     # (rank, trk_id) of foreign entries that were re-identified here
        self._done_seen = False
        self.foreign = []          # list of dicts: rank, trk_id, label, count, feat
        self.n_collectives = 0
        self._wait_s = 0.
        self._enqueue_s = 0.
        self._gpu_ms = []
        self._pending = False      # an exchange is in flight

    def _default_comm(self, group):
        """No torch in the process: the C-ABI communicator.  A process that already runs torch.distributed (it has
        loaded torch's own ROCm copy, see RcclComm) keeps its collectives there."""
        torch = sys.modules.get('torch')
        if torch is not None and torch.distributed.is_available() and torch.distributed.is_initialized():
            return TorchComm(self.row_bytes, group)
        from .runtime import get_context
        return RcclComm(get_context(), self.row_bytes)

    # ------------------------------------------------------------------ wire format
    def pack(self, entries, done=False):
        """entries: iterable of (trk_id, label, count, avg_feat[feat_dim]) -> uint8[row_bytes] (the LAST
        history_size entries when there are more)."""
        H, D = self.history_size, self.feat_dim
        buf = np.zeros(self.row_bytes, np.uint8)
        entries = list(entries)[-H:]
        hdr = buf[:self._hdr].view(np.int64)
        hdr[0], hdr[1] = len(entries), int(done)
        meta = buf[self._hdr:self._hdr + self._meta_bytes].view(np.int64).reshape(H, 3)
        feats = buf[self._hdr + self._meta_bytes:].view(np.float32).reshape(H, D)
        for i, (trk_id, label, count, feat) in enumerate(entries):
            meta[i] = (trk_id, label, count)
            feats[i] = feat
        return buf

    def unpack(self, gathered):
        """gathered: uint8 [world * row_bytes] -> (entries of all OTHER ranks in rank-major order, all_done)."""
        H, D = self.history_size, self.feat_dim
        out = []
        all_done = True
        rows = np.asarray(gathered).reshape(self.world, self.row_bytes)
        for r in range(self.world):
            row = rows[r]
            hdr = row[:self._hdr].view(np.int64)
            all_done = all_done and bool(hdr[1])
            if r == self.rank:
                continue
            meta = row[self._hdr:self._hdr + self._meta_bytes].view(np.int64).reshape(H, 3)
            feats = row[self._hdr + self._meta_bytes:].view(np.float32).reshape(H, D)
            for i in range(int(hdr[0])):
                key = (r, int(meta[i, 0]))
                if meta[i, 2] > 0 and key not in self._consumed:
                    out.append(dict(rank=r, trk_id=int(meta[i, 0]), label=int(meta[i, 1]), count=int(meta[i, 2]),
                                    feat=feats[i].copy()))
        return out, all_done

    # ------------------------------------------------------------------ collective
    def _issue(self, entries, done=False):
        t0 = time.perf_counter()
        self.comm.issue(self.pack(entries, done))
        self._pending = True
        self.n_collectives += 1
        self._enqueue_s += time.perf_counter() - t0

    def _complete(self):
        """Waits for the in-flight exchange (normally long finished) and publishes its entries."""
        if not self._pending:
            return
        t0 = time.perf_counter()
        rows, stream_ms = self.comm.complete()
        self._pending = False
        if stream_ms is not None:
            self._gpu_ms.append(stream_ms)
        self._wait_s += time.perf_counter() - t0
        self.foreign, self._done_seen = self.unpack(rows)

    def exchange(self, entries, force=False):
        """Called once per detector frame with the local gallery.  Every `period` calls: publishes the result of
        the previous all-gather and enqueues the next one (asynchronous=True), or runs one all-gather to
        completion (asynchronous=False / force).  Returns the current foreign entries."""
        self._calls += 1
        if not force and (self._calls - 1) % self.period:
            return self.foreign
        if self.asynchronous and not force:
            self._complete()
            self._issue(entries)
        else:
            self._complete()
            self._issue(entries)
            self._complete()
        return self.foreign

    def consume(self, rank, trk_id):
        """A foreign identity was re-identified on this stream: it is not offered again."""
        self._consumed.add((rank, trk_id))
        self.foreign = [e for e in self.foreign if (e['rank'], e['trk_id']) != (rank, trk_id)]

    def close(self, max_rounds=100000):
        """End-of-stream protocol (see module docstring): collective on every rank."""
        self._complete()
        rounds = 0
        while not self._done_seen and rounds < max_rounds:
            self._issue([], done=True)
            self._complete()
            rounds += 1
        self.comm.close()
        return rounds

    def stats(self):
        """For bench.py: what the exchange cost this rank."""
        n = max(self.n_collectives, 1)
        out = dict(backend='rccl' if self.on_gpu else self.backend, world=self.world, collectives=self.n_collectives,
                   bytes_per_rank=self.row_bytes, asynchronous=self.asynchronous,
                   host_enqueue_us=round(self._enqueue_s / n * 1e6, 1), host_wait_us=round(self._wait_s / n * 1e6, 1))
        if self._gpu_ms:
            out['allgather_stream_us'] = round(float(np.mean(self._gpu_ms)) * 1e3, 1)
        return out


def stream_shard(n_streams, rank, world):
    """Streams owned by `rank` when n_streams are dealt round-robin over `world` ranks (1 stream per
    GPU is the nominal configuration; more streams than GPUs run back to back on one GPU)."""
    return [s for s in range(n_streams) if s % world == rank]
