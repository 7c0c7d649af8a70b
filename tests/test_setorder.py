"""fastmot_amd.utils.setorder.unmatched_order (closed form used by the tracker) against the full restatement of
Numba's integer set in oracle/numba_set.py, and what distinguishes both from CPython's set order."""
import numpy as np

import numba_set
from fastmot_amd.utils.setorder import IntSet, unmatched_order


def test_closed_form_equals_restated_container_exhaustive_small():
    for n in range(0, 13):
        for bits in range(1 << n):
            matched = [k for k in range(n) if bits >> k & 1]
            assert unmatched_order(n, matched) == numba_set.difference_order(n, matched), (n, matched)


def test_closed_form_equals_restated_container_random():
    rng = np.random.default_rng(0)
    differs_from_cpython = 0
    for _ in range(4000):
        n = int(rng.integers(1, 400))
        keep = int(rng.integers(0, min(n, 40) + 1)) if rng.random() < 0.7 else int(rng.integers(0, n + 1))
        rest = rng.choice(n, keep, replace=False)
        matched = sorted(set(range(n)) - set(rest.tolist()))
        rng.shuffle(matched)
        a = unmatched_order(n, matched)
        assert a == numba_set.difference_order(n, matched), (n, matched)
        assert sorted(a) == sorted(rest.tolist())
        differs_from_cpython += a != list(set(range(n)) - set(matched))
    assert differs_from_cpython > 0          # (the point of the exercise)


def test_known_orders():
    # 50 detections, two unmatched: table 128 -> 16 slots; 35 lands in slot 3, 20 in slot 4
    assert unmatched_order(50, [k for k in range(50) if k not in (20, 35)]) == [35, 20]
    # CPython keeps these ascending in ITS table of 8 only by accident of the values: 7 -> slot 7, 8 -> slot 0
    assert list(set(range(50)) - set(k for k in range(50) if k not in (7, 8))) == [8, 7]
    assert unmatched_order(50, [k for k in range(50) if k not in (7, 8)]) == [7, 8]
    # three survivors that collide in the shrunk table: linear probes
    assert unmatched_order(50, [k for k in range(50) if k not in (7, 23, 39)]) == [7, 23, 39]
    # nothing shrinks when most survive
    assert unmatched_order(20, [3, 5]) == [k for k in range(20) if k not in (3, 5)]


def test_container_basics():
    s = numba_set.NumbaIntSet(range(100))
    assert len(s) == 100 and s.mask + 1 == 256 and list(s) == list(range(100))
    t = numba_set.NumbaIntSet()
    for k in range(9):
        t.add(k)
    assert t.mask + 1 == 64 and 8 in t and 9 not in t        # 8th insertion: 2 * 8 >= 16 -> x4


def test_discard_sequences_equal_restated_container():
    rng = np.random.default_rng(1)
    for _ in range(1500):
        n = int(rng.integers(1, 300))
        a, b = IntSet(n), numba_set.NumbaIntSet(range(n))
        for k in rng.choice(n, int(rng.integers(0, n + 1)), replace=False).tolist() + [n + 3]:
            a.discard(k)
            b.discard(k)
        assert list(a) == list(b)


def _real_numba_cases():
    """tests/golden/numba_set_order.npz: answers of a REAL Numba (0.54.1, oracle/pin_with_numba.py) for
    list(set(range(n)) - set(removed)) and for set(range(n)) with the same elements discard()ed one by one."""
    from pathlib import Path
    z = np.load(Path(__file__).parent / 'golden' / 'numba_set_order.npz')
    for i, n in enumerate(z['n'].tolist()):
        removed = z['removed'][z['removed_off'][i]:z['removed_off'][i + 1]].tolist()
        lo, hi = z['out_off'][i], z['out_off'][i + 1]
        yield n, removed, z['difference'][lo:hi].tolist(), z['discarded'][lo:hi].tolist()


def test_restatement_equals_real_numba():
    """the oracle's container against the jit-compiled thing itself (matching.py:59-60, detector.py:196-211)"""
    cases = 0
    for n, removed, difference, discarded in _real_numba_cases():
        assert numba_set.difference_order(n, removed) == difference, (n, removed)
        s = numba_set.NumbaIntSet(range(n))
        for k in removed:
            s.discard(k)
        assert list(s) == discarded, (n, removed)
        cases += 1
    assert cases > 900


def test_product_order_equals_real_numba():
    """what the tracker / the SSD merge of the product use (fastmot_amd/utils/setorder.py), against the same answers"""
    for n, removed, difference, discarded in _real_numba_cases():
        assert unmatched_order(n, removed) == difference, (n, removed)
        s = IntSet(n)
        for k in removed:
            s.discard(k)
        assert list(s) == discarded, (n, removed)
