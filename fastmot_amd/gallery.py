"""Cross-stream ReID gallery exchange (multi-GPU, one process per GPU / video stream).

NOT in the reference (which is single process, single stream).  Streams shard naturally one per GPU
with no data-path collective; the only optional exchange is an all-gather of every rank's lost-track
gallery (MultiTracker.hist_tracks: <= history_size entries of {track id, label, feature count,
512-d average feature}, ~104 KB per rank) so that an identity that left camera A can be re-identified
on camera B.  It runs through torch.distributed: backend "nccl" is RCCL on ROCm (xGMI), "gloo" is
used by the CPU tests.  The payload is latency bound (<= 832 KB for 8 ranks): ONE fixed-size all_gather
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
import time

import numpy as np


class GallerySync:
    def __init__(self, history_size=50, feat_dim=512, period=1, group=None, asynchronous=True):
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError('torch.distributed must be initialised (backend nccl on GPUs, gloo on CPU)')
        self.torch, self.dist, self.group = torch, dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.history_size, self.feat_dim, self.period = history_size, feat_dim, max(int(period), 1)
        self.backend = dist.get_backend(group)
        self.on_gpu = self.backend == 'nccl'
        self.device = torch.device('cuda', torch.cuda.current_device()) if self.on_gpu else torch.device('cpu')
        self.asynchronous = asynchronous
        self._calls = 0
        self._consumed = set()     # (rank, trk_id) of foreign entries that were re-identified here
        self._done_seen = False
        self.foreign = []          # list of dicts: rank, trk_id, label, count, feat
        self.n_collectives = 0
        self._wait_s = 0.
        self._enqueue_s = 0.
        self._gpu_ms = []

        H, D = history_size, feat_dim
        self._hdr = 32
        self._meta_bytes = H * 3 * 8
        self.row_bytes = self._hdr + self._meta_bytes + H * D * 4
        # staging: page-locked host mirrors on the GPU path so that both copies are asynchronous
        pin = self.on_gpu
        self._send_host = torch.zeros(self.row_bytes, dtype=torch.uint8, pin_memory=pin)
        self._recv_host = torch.zeros(self.world * self.row_bytes, dtype=torch.uint8, pin_memory=pin)
        if self.on_gpu:
            self._send_dev = torch.zeros(self.row_bytes, dtype=torch.uint8, device=self.device)
            self._recv_dev = torch.zeros(self.world * self.row_bytes, dtype=torch.uint8, device=self.device)
            self._stream = torch.cuda.Stream(device=self.device)
            self._ev0 = torch.cuda.Event(enable_timing=True)
            self._ev1 = torch.cuda.Event(enable_timing=True)
        self._pending = None       # in-flight exchange: torch Work (cpu) or CUDA event (gpu)

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
        torch, dist = self.torch, self.dist
        self._send_host.numpy()[:] = self.pack(entries, done)
        if self.on_gpu:
            # side stream: H2D of the local row, RCCL all-gather, D2H of all rows, completion event
            with torch.cuda.stream(self._stream):
                self._ev0.record(self._stream)
                self._send_dev.copy_(self._send_host, non_blocking=True)
                dist.all_gather_into_tensor(self._recv_dev, self._send_dev, group=self.group)
                self._recv_host.copy_(self._recv_dev, non_blocking=True)
                self._ev1.record(self._stream)
            self._pending = self._ev1
        else:
            self._pending = dist.all_gather_into_tensor(self._recv_host, self._send_host, group=self.group,
                                                        async_op=True)
        self.n_collectives += 1
        self._enqueue_s += time.perf_counter() - t0

    def _complete(self):
        """Waits for the in-flight exchange (normally long finished) and publishes its entries."""
        if self._pending is None:
            return
        t0 = time.perf_counter()
        if self.on_gpu:
            self._pending.synchronize()
            self._gpu_ms.append(self._ev0.elapsed_time(self._ev1))
        else:
            self._pending.wait()
        self._pending = None
        self._wait_s += time.perf_counter() - t0
        self.foreign, self._done_seen = self.unpack(self._recv_host.numpy())

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
