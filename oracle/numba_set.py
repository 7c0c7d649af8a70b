"""TEST INFRASTRUCTURE ONLY -- restatement of Numba's reflected/typed `set` of integers (numba/targets/setobj.py in the
reference's pinned Numba 0.48, `requirements.txt:3`; the same algorithm lives in numba/cpython/setobj.py of later
releases), as far as fastmot/utils/matching.py:59-60 uses it INSIDE `@nb.njit`:

    unmatched_rows = list(set(range(cost.shape[0])) - set(m_rows))

The iteration order of that difference decides the order of the unmatched track / detection ids and with it the order in
which new track IDs are handed out.  CPython's set and Numba's set are different hash tables (minimum
size 8 vs 16, growth x4 at 3/5 load vs to >= 2 x used, no shrinking vs `downsize` after a difference), so the de-jitted
reference (oracle/ref_shim.py: decorators stubbed) can order these lists differently from the reference as it really
runs.  ref_shim therefore hands `_get_assignment_matches` THIS set type in place of the builtin.

PINNED against a real Numba (round 3): the image's /opt/conda Python 3.9 carries Numba 0.54.1, which imports once its
NumPy-version gate is satisfied in-process (oracle/real_numba.py).  oracle/pin_with_numba.py compares this container
with `list(set(range(n)) - set(removed))` and with one-by-one `discard()` inside @njit functions over 996 (n, removed)
cases (n = 0..69 and sizes around every table-growth boundary up to 700): 0 mismatches; the cases and the real answers
are committed as tests/golden/numba_set_order.npz and checked on every machine (tests/test_setorder.py), for this
container and for the product's fastmot_amd/utils/setorder.py.  The reference pins Numba 0.48 (`requirements.txt:3`),
which is not available here; 0.54.1's numba/cpython/setobj.py is the same algorithm as far as it is restated below.
The restatement follows the published source:
  * open addressing; entry = (hash, key); EMPTY = -1, DELETED = -2; MINSIZE = 16; LINEAR_PROBES = 3
  * hash(int64 i) = i for 0 <= i < 2**61 - 1 (numba/targets/hashing.py; -1 -> -2, not reachable here)
  * probe sequence of `_lookup`: index = h & mask; three linear probes (index, index + 1, index + 2, wrapping), then
    repeatedly  perturb >>= 5;  index = (index * 5 + 1 + perturb) & mask
  * `set(iterable)` with a known length n: table = MINSIZE doubled until >= 2 * n (`choose_alloc_size`), then add();
    add() -> `upsize`: when 2 * used >= size the table is multiplied by 4 until > 2 * used and rebuilt
  * `a - b` = `a.copy()` (raw copy when there are no deleted entries) followed by `difference_update(b)`: every key of b
    is removed (entry -> DELETED, no resize), then ONE `downsize(used)`: min_entries = max(2 * used, MINSIZE); if
    size >= 4 * min_entries and size > MINSIZE the table is halved while the half is still >= min_entries and rebuilt
  * a rebuild re-inserts the live entries in the OLD table's slot order; iteration = slot order.
"""
EMPTY, DELETED = -1, -2
MINSIZE = 16
LINEAR_PROBES = 3


class NumbaIntSet:
    def __init__(self, iterable=()):
        items = list(iterable)
        size = MINSIZE
        while size < 2 * len(items):
            size <<= 1
        self._alloc(size)
        for v in items:
            self.add(v)

    # ---- table
    def _alloc(self, size):
        self.hashes = [EMPTY] * size
        self.keys = [0] * size
        self.mask = size - 1
        self.used = 0
        self.fill = 0

    @staticmethod
    def _hash(v):
        v = int(v)
        assert 0 <= v < (1 << 61) - 1, 'only the non-negative small integers of matching.py are restated'
        return v

    def _lookup(self, key, h, for_insert):
        """-> (found, index); for_insert: index = first DELETED slot of the chain, else the EMPTY slot that ended it."""
        mask = self.mask
        perturb = h
        index = h & mask
        free = -1

        def check(i):
            nonlocal free
            eh = self.hashes[i]
            if eh == h and self.keys[i] == key:
                return 'found'
            if eh == EMPTY:
                return 'empty'
            if for_insert and eh == DELETED and free == -1:
                free = i
            return None
        for _ in range(LINEAR_PROBES):
            r = check(index)
            if r == 'found':
                return True, index
            if r == 'empty':
                return False, (free if free != -1 else index)
            index = (index + 1) & mask
        while True:
            r = check(index)
            if r == 'found':
                return True, index
            if r == 'empty':
                return False, (free if free != -1 else index)
            perturb >>= 5
            index = (index * 5 + 1 + perturb) & mask

    def _add_key(self, key, h, do_resize):
        found, i = self._lookup(key, h, True)
        if found:
            return
        if self.hashes[i] == EMPTY:
            self.fill += 1
        self.hashes[i] = h
        self.keys[i] = key
        self.used += 1
        if do_resize:
            self._upsize(self.used)

    def _rebuild(self, new_size):
        old = [(self.hashes[i], self.keys[i]) for i in range(self.mask + 1) if self.hashes[i] >= 0]
        self._alloc(new_size)
        for h, k in old:
            self._add_key(k, h, False)

    def _upsize(self, nitems):
        min_entries = nitems << 1
        size = self.mask + 1
        if min_entries >= size:
            new_size = size
            while True:
                new_size <<= 2
                if not min_entries >= new_size:
                    break
            self._rebuild(new_size)

    def _downsize(self, nitems):
        min_entries = max(nitems << 1, MINSIZE)
        size = self.mask + 1
        if (min_entries << 2) <= size and MINSIZE < size:
            new_size = size
            while True:
                half = new_size >> 1
                if min_entries > half:
                    break
                new_size = half
            self._rebuild(new_size)

    # ---- the operations matching.py uses
    def add(self, v):
        self._add_key(int(v), self._hash(v), True)

    def discard(self, v):
        """set.discard: `_remove_key(..., do_resize=True)` -- the table may shrink after EVERY removal."""
        found, i = self._lookup(int(v), self._hash(v), False)
        if found:
            self.hashes[i] = DELETED
            self.used -= 1
            self._downsize(self.used)

    def copy(self):
        other = NumbaIntSet()
        if self.used == self.fill:                      # no deleted entries: raw copy of the payload
            other.hashes, other.keys = list(self.hashes), list(self.keys)
            other.mask, other.used, other.fill = self.mask, self.used, self.fill
        else:
            size = MINSIZE
            while size < 2 * self.used:
                size <<= 1
            other._alloc(size)
            for h, k in ((self.hashes[i], self.keys[i]) for i in range(self.mask + 1) if self.hashes[i] >= 0):
                other._add_key(k, h, False)
        return other

    def difference_update(self, other):
        for v in other:
            found, i = self._lookup(v, self._hash(v), False)
            if found:
                self.hashes[i] = DELETED
                self.used -= 1
        self._downsize(self.used)

    def __sub__(self, other):
        s = self.copy()
        s.difference_update(other)
        return s

    def __iter__(self):
        return (self.keys[i] for i in range(self.mask + 1) if self.hashes[i] >= 0)

    def __len__(self):
        return self.used

    def __contains__(self, v):
        return self._lookup(int(v), self._hash(v), False)[0]


def difference_order(n, removed):
    """list(set(range(n)) - set(removed)) as Numba orders it."""
    return list(NumbaIntSet(range(n)) - NumbaIntSet(removed))
