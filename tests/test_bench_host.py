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


def test_stage_durations_pairs_repeated_marks():
    """fm_trace marks of a stage: one occurrence = the first start since the previous end, up to that end."""
    import numpy as np
    # plain alternation
    np.testing.assert_allclose(bench.stage_durations([0., 10, 20], [4., 13, 29]), [4, 3, 9])
    # start mark repeated inside an occurrence (chunks of a batch): the first one counts
    np.testing.assert_allclose(bench.stage_durations([0., 1, 2, 30, 31, 32], [5., 40]), [5, 10])
    # end mark repeated (the crop stage of a chunked batch ends at every chunk's network mark): the first end counts
    np.testing.assert_allclose(bench.stage_durations([0., 30], [5., 6, 7, 35, 36, 37]), [5, 5])
    # an end without a start since the previous end is dropped; unsorted input is accepted
    np.testing.assert_allclose(bench.stage_durations([30., 0.], [5., 20, 35]), [5, 5])
    assert len(bench.stage_durations([], [1.])) == 0 and len(bench.stage_durations([1.], [])) == 0
