"""CPU, world_size 2 over gloo: the multi-GPU path (stream sharding + ReID-gallery all-gather)."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]


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
    sync = GallerySync(history_size=5, feat_dim=16, period=2)
    rng = np.random.default_rng(rank)
    entries = [(10 * rank + i, 1, 2 + i, rng.normal(0, 1, 16).astype(np.float32)) for i in range(3 + rank)]
    foreign = sync.exchange(entries)
    again = sync.exchange([])               # period 2: no new collective, cached result
    third = sync.exchange(entries[:1])      # third call exchanges again
    np.savez(Path(out_dir) / f'rank{rank}.npz',
             ids=np.array([e['trk_id'] for e in foreign]), ranks=np.array([e['rank'] for e in foreign]),
             feats=np.array([e['feat'] for e in foreign]), cached=len(again), third=len(third),
             shard=np.array(stream_shard(5, rank, world)),
             mine=np.array([e[3] for e in entries]))
    dist.barrier()
    dist.destroy_process_group()


def test_gallery_allgather_gloo(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / 'rank0.npz'), np.load(tmp_path / 'rank1.npz')
    # rank 0 sees rank 1's 4 entries and vice versa (3 entries), features bit-identical
    assert r0['ids'].tolist() == [10, 11, 12, 13] and set(r0['ranks'].tolist()) == {1}
    assert r1['ids'].tolist() == [0, 1, 2] and set(r1['ranks'].tolist()) == {0}
    np.testing.assert_array_equal(r0['feats'], r1['mine'])
    np.testing.assert_array_equal(r1['feats'], r0['mine'])
    assert int(r0['cached']) == 4 and int(r0['third']) == 1
    assert r0['shard'].tolist() == [0, 2, 4] and r1['shard'].tolist() == [1, 3]
