"""Iteration order of `list(set(range(n)) - set(matched))` INSIDE the reference's `@nb.njit` code
(fastmot/utils/matching.py:58-60): Numba's integer set, not CPython's.

The order of the unmatched rows / columns is observable -- it is the order in which tracks are marked missed and new
track IDs are handed out (tracker.py:250-293) -- and the two hash tables iterate differently as soon as the difference
is small against the range (Numba shrinks the table after the difference and re-inserts, CPython builds the result in
a fresh 8-slot table).  What the reference really executes is Numba 0.48 (`requirements.txt`), so that is the order
reproduced here; oracle/numba_set.py is the full restatement of the container this closed form is tested against
(tests/test_setorder.py).  Pinned by the real thing since round 3: an Anaconda Numba 0.54.1 found in the image
(oracle/real_numba.py, oracle/pin_with_numba.py) answers 996 cases around every table-growth boundary up to n = 700
identically -- the fixture tests/golden/numba_set_order.npz, against which this module is tested on every machine.
Open: the reference pins Numba 0.48, the pin ran 0.54.1 (the set implementation's published source is the same in
both).

Closed form: all keys k < n are smaller than the table (size >= 2 n, hash(k) = k), so `set(range(n))` holds key k in
slot k and the survivors of the difference are met in ascending order.  If the table is at least four times the
minimum for the survivors (max(2 * survivors, 16)) it is halved down to the smallest power of two that still holds
them, and the survivors are re-inserted in ascending order at `k & mask` with Numba's probe sequence (three linear
probes, then index = 5 * index + 1 + (perturb >>= 5)); the result is read in slot order."""
MINSIZE = 16


def unmatched_order(n, matched):
    gone = set(matched)
    rest = [k for k in range(n) if k not in gone]
    size = MINSIZE
    while size < 2 * n:
        size <<= 1
    min_entries = max(2 * len(rest), MINSIZE)
    if 4 * min_entries > size or size <= MINSIZE:
        return rest
    while (size >> 1) >= min_entries:
        size >>= 1
    mask = size - 1
    if not rest or rest[-1] <= mask:
        return rest                      # nothing wraps: slot k again
    table = [-1] * size
    for k in rest:
        index, perturb = k & mask, k
        for _ in range(3):
            if table[index] < 0:
                break
            index = (index + 1) & mask
        else:
            while table[index] >= 0:
                perturb >>= 5
                index = (index * 5 + 1 + perturb) & mask
        table[index] = k
    return [k for k in table if k >= 0]


class IntSet:
    """The same container in general form for the one other place the reference iterates a set inside jitted code:
    SSDDetector._merge (detector.py:196-211: `keep = set(range(n))`, `keep.discard(k)` per merged detection -- the
    table may shrink after every discard -- then `dets[np.array(list(keep))]`).  Keys are non-negative ints and
    hash(k) = k, so one list of keys with two markers is the whole table."""
    _EMPTY, _DELETED = -1, -2

    def __init__(self, n):
        size = MINSIZE
        while size < 2 * n:
            size <<= 1
        self.table = list(range(n)) + [self._EMPTY] * (size - n)
        self.used = n

    def _slot(self, k):
        mask = len(self.table) - 1
        index, perturb, probes = k & mask, k, 0
        while True:
            v = self.table[index]
            if v == k or v == self._EMPTY:
                return index
            probes += 1
            if probes < 3:
                index = (index + 1) & mask
            elif probes == 3:
                index = (index + 1) & mask            # (the probe loop ends on the slot after its third check)
            else:
                perturb >>= 5
                index = (index * 5 + 1 + perturb) & mask

    def discard(self, k):
        i = self._slot(k)
        if self.table[i] != k:
            return
        self.table[i] = self._DELETED
        self.used -= 1
        size = len(self.table)
        min_entries = max(2 * self.used, MINSIZE)
        if 4 * min_entries <= size and size > MINSIZE:
            while (size >> 1) >= min_entries:
                size >>= 1
            live = [v for v in self.table if v >= 0]
            self.table = [self._EMPTY] * size
            for v in live:
                self.table[self._slot(v)] = v

    def __iter__(self):
        return (v for v in self.table if v >= 0)
