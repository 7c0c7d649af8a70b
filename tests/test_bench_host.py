"""Host-side helpers of bench.py (no GPU): clip order, per-rank sizing of the RANSAC worker pool."""
import bench


def test_ping_pong_is_continuous():
    seq = [bench.ping_pong(s, 5) for s in range(20)]
    assert seq[:9] == [0, 1, 2, 3, 4, 3, 2, 1, 0]
    assert all(abs(a - b) <= 1 for a, b in zip(seq, seq[1:]))


def test_flow_threads_per_rank(monkeypatch):
    assert bench.usable_cpus() >= 1
    monkeypatch.setattr(bench, 'usable_cpus', lambda: 16)
    assert [bench.flow_threads_for(n) for n in (1, 2, 4, 8)] == [7, 6, 2, 1]
    monkeypatch.setattr(bench, 'usable_cpus', lambda: 256)
    assert [bench.flow_threads_for(n) for n in (1, 8)] == [7, 7]
