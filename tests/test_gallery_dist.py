"""CPU, world_size 2 over gloo: the multi-GPU path (stream sharding + the asynchronous ReID-gallery
all-gather with its end-of-stream protocol)."""
import os
import socket
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
BIG = 2 ** 40            # track ids beyond float32's 2^24 integer range must survive the wire format


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, str(ROOT))
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from fastmot_amd.gallery import GallerySync, stream_shard
    rng = np.random.default_rng(rank)
    entries = [(BIG + 10 * rank + i, 1, 2 + i, rng.normal(0, 1, 16).astype(np.float32)) for i in range(3 + rank)]

    # (1) synchronous exchanges with a period: call 1 and 3 run a collective, call 2 returns the cached result
    sync = GallerySync(history_size=5, feat_dim=16, period=2, asynchronous=False)
    foreign = sync.exchange(entries)
    again = sync.exchange([])
    third = sync.exchange(entries[:1])
    assert sync.n_collectives == 2
    sync.close()

    # (2) asynchronous exchanges; the two streams have a different number of detector frames (3 vs 5):
    # exchange k returns what exchange k-1 gathered, close() keeps the finished rank in the collective
    a = GallerySync(history_size=5, feat_dim=16)
    seen = []
    n_frames = 3 if rank == 0 else 5
    for f in range(n_frames):
        got = a.exchange(entries[:1 + f % 3])
        seen.append([e['trk_id'] - BIG for e in got])
        if f == 1 and got:
            a.consume(got[0]['rank'], got[0]['trk_id'])      # re-identified here: never offered again
    rounds = a.close()
    np.savez(Path(out_dir) / f'rank{rank}.npz',
             ids=np.array([e['trk_id'] for e in foreign]), ranks=np.array([e['rank'] for e in foreign]),
             feats=np.array([e['feat'] for e in foreign]), cached=len(again), third=len(third),
             shard=np.array(stream_shard(5, rank, world)), mine=np.array([e[3] for e in entries]),
             seen=np.array([len(s) for s in seen]), first_seen=np.array([s[0] if s else -1 for s in seen]),
             rounds=rounds, collectives=a.n_collectives)
    dist.barrier()
    dist.destroy_process_group()


def test_gallery_allgather_gloo(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / 'rank0.npz'), np.load(tmp_path / 'rank1.npz')
    # rank 0 sees rank 1's 4 entries and vice versa (3 entries), int64 ids and features bit-identical
    assert (r0['ids'] - BIG).tolist() == [10, 11, 12, 13] and set(r0['ranks'].tolist()) == {1}
    assert (r1['ids'] - BIG).tolist() == [0, 1, 2] and set(r1['ranks'].tolist()) == {0}
    np.testing.assert_array_equal(r0['feats'], r1['mine'])
    np.testing.assert_array_equal(r1['feats'], r0['mine'])
    assert int(r0['cached']) == 4 and int(r0['third']) == 1
    assert r0['shard'].tolist() == [0, 2, 4] and r1['shard'].tolist() == [1, 3]
    # asynchronous: nothing at the first exchange, then the other rank's gallery of the PREVIOUS frame
    # (1, 2, 3 entries per frame); the entry consumed at frame 1 is gone from frame 2 on
    assert r0['seen'].tolist() == [0, 1, 1]          # frame 2: two entries gathered, one of them consumed
    assert r1['seen'].tolist() == [0, 1, 1, 2, 0]    # frames 3/4: rank 0 is in close() with an empty gallery ...
    assert r0['first_seen'].tolist()[1] == 10 and r1['first_seen'].tolist()[1] == 0
    # both ranks issued the same number of collectives and left close() together
    assert int(r0['collectives']) == int(r1['collectives'])
    assert int(r0['rounds']) >= 2 and int(r1['rounds']) >= 1


def test_rccl_comm_bootstrap_distributes_the_unique_id():
    """RcclComm's TCP bootstrap (rank 0 serves the 128-byte communicator id to the other ranks) with the device
    calls replaced by a recorder: every rank must join with rank 0's id, its own rank and the common row size."""
    import socket
    import threading
    from fastmot_amd.gallery import RcclComm

    class FakeCtx:
        def __init__(self):
            self.joined = None

        def gallery_unique_id(self):
            return bytes(range(128))

        def gallery_init(self, channel, world, rank, uid, row_bytes):
            self.joined = (channel, world, rank, uid, row_bytes)

    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    ctxs = [FakeCtx() for _ in range(3)]
    errs = []

    def join(r):
        try:
            RcclComm(ctxs[r], 64, rank=r, world=3, addr='127.0.0.1', port=port, channel=1, timeout=30)
        except Exception as e:      # noqa: BLE001
            errs.append(e)
    import sys
    saved = sys.modules.pop('torch', None)          # (RcclComm refuses processes that have loaded torch's ROCm copy)
    try:
        th = [threading.Thread(target=join, args=(r,)) for r in (2, 1, 0)]       # clients first: they retry
        for t in th:
            t.start()
        for t in th:
            t.join(60)
    finally:
        if saved is not None:
            sys.modules['torch'] = saved
    assert not errs, errs
    assert [c.joined for c in ctxs] == [(1, 3, r, bytes(range(128)), 64) for r in range(3)]
