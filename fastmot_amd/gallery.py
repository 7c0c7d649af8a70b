"""Cross-stream ReID gallery exchange (multi-GPU, one process per GPU / video stream).

NOT in the reference (which is single process, single stream).  Streams shard naturally one per GPU
with no data-path collective; the only optional exchange is an all-gather of every rank's lost-track
gallery (MultiTracker.hist_tracks: <= history_size entries of {track id, label, feature count,
512-d average feature}, ~104 KB per rank) so that an identity that left camera A can be re-identified
on camera B.  It runs through torch.distributed: backend "nccl" is RCCL on ROCm (xGMI), "gloo" is
used by the CPU tests.  The payload is latency bound (<= 832 KB for 8 ranks), so it is one fixed-size
all_gather per exchange, off the per-frame critical path, and strictly opt-in: with it disabled
every stream's results are bit-identical to a single-GPU run.

Foreign entries are appended AFTER the local history rows of the ReID cost matrix, so local
tie-breaks (greedy first-minimum order, tracker.py:229-241) are unchanged.
"""
import numpy as np


class GallerySync:
    def __init__(self, history_size=50, feat_dim=512, period=1, group=None):
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError('torch.distributed must be initialised (backend nccl on GPUs, gloo on CPU)')
        self.torch, self.dist, self.group = torch, dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.history_size, self.feat_dim, self.period = history_size, feat_dim, max(int(period), 1)
        self.device = torch.device('cuda', torch.cuda.current_device()) if dist.get_backend(group) == 'nccl' \
            else torch.device('cpu')
        self._calls = 0
        self.foreign = []      # list of dicts: rank, trk_id, label, count, feat

    def pack(self, entries):
        """entries: iterable of (trk_id, label, count, avg_feat[feat_dim]) -> float32 [history_size, 3 + dim];
        unused rows have count 0."""
        buf = np.zeros((self.history_size, 3 + self.feat_dim), np.float32)
        for i, (trk_id, label, count, feat) in enumerate(list(entries)[-self.history_size:]):
            buf[i, 0], buf[i, 1], buf[i, 2] = trk_id, label, count
            buf[i, 3:] = feat
        return buf

    def unpack(self, gathered):
        """gathered: [world, history_size, 3 + dim] -> entries of all OTHER ranks, rank-major order."""
        out = []
        for r in range(self.world):
            if r == self.rank:
                continue
            for row in gathered[r]:
                if row[2] > 0:
                    out.append(dict(rank=r, trk_id=int(row[0]), label=int(row[1]), count=int(row[2]),
                                    feat=row[3:].astype(np.float32)))
        return out

    def exchange(self, entries, force=False):
        """All-gathers the local gallery every `period` calls; returns the current foreign entries."""
        self._calls += 1
        if not force and (self._calls - 1) % self.period:
            return self.foreign
        local = self.torch.from_numpy(self.pack(entries)).to(self.device)
        gathered = [self.torch.empty_like(local) for _ in range(self.world)]
        self.dist.all_gather(gathered, local, group=self.group)
        self.foreign = self.unpack(np.stack([g.cpu().numpy() for g in gathered]))
        return self.foreign


def stream_shard(n_streams, rank, world):
    """Streams owned by `rank` when n_streams are dealt round-robin over `world` ranks (1 stream per
    GPU is the nominal configuration; more streams than GPUs run back to back on one GPU)."""
    return [s for s in range(n_streams) if s % world == rank]
