"""Shared helpers for the tests: the portable RNG of SURVEY.md §8(d), digests of App. B,
and the structural invariants of the reference's test/utils.jl:68-113."""
import numpy as np

M64 = (1 << 64) - 1


class SplitMix64:
    def __init__(self, seed):
        self.x = seed & M64

    def next(self):
        self.x = (self.x + 0x9E3779B97F4A7C15) & M64
        z = self.x
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
        return z ^ (z >> 31)

    def randint(self, n):          # uniform in [1, n]
        return 1 + self.next() % n

    def unit12(self):              # uniform double in [1, 2)
        return 1.0 + (self.next() >> 11) * 2.0 ** -53


def splitmix_array(seed, n):
    """n raw 64-bit outputs, vectorised (numpy uint64 wraps mod 2^64)."""
    with np.errstate(over="ignore"):
        idx = np.arange(1, n + 1, dtype=np.uint64)
        x = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def unit12_array(seed, n):
    return 1.0 + (splitmix_array(seed, n) >> np.uint64(11)).astype(np.float64) * 2.0 ** -53


H0 = 1469598103934665603
FNV = 1099511628211


def layout_digest(keys, occ):
    h = H0
    for pos in np.nonzero(occ)[0]:
        h = ((h ^ (int(pos) + 1)) * FNV) & M64
        h = ((h ^ (int(keys[pos]) & M64)) * FNV) & M64
    return h


def table_digest(t):
    h = H0
    for v in t:
        h = ((h ^ (int(v) & M64)) * FNV) & M64
    return h


def check_semaphores(keys, vals, occ, semaphores):
    """test/utils.jl:68-92 — table <-> slot consistency; returns the number of live semaphores."""
    live = 0
    for pid, pos in enumerate(semaphores, start=1):
        if pos != 0:
            assert occ[pos - 1] and keys[pos - 1] == 0 and vals[pos - 1] == float(pid), (pid, pos)
            live += 1
    in_array = 0
    for pos in np.nonzero(occ)[0]:
        if keys[pos] == 0:
            assert semaphores[int(vals[pos]) - 1] == pos + 1
            in_array += 1
    assert live == in_array
    return live


def check_key_order(keys, occ):
    """strictly increasing keys inside every partition (intent of test/utils.jl:94-113)."""
    pred = None
    for pos in np.nonzero(occ)[0]:
        k = int(keys[pos])
        if k == 0:
            pred = None
        else:
            if pred is not None:
                assert pred < k, (pos, pred, k)
            pred = k


def layouts_equal(a, b):
    """slot-for-slot equality of two exported layouts (tuples keys, vals, occ[, ...])."""
    ka, va, oa = a[0], a[1], a[2]
    kb, vb, ob = b[0], b[1], b[2]
    if len(oa) != len(ob) or not np.array_equal(oa, ob):
        return False
    m = oa.astype(bool)
    return bool(np.array_equal(ka[m], kb[m]) and np.array_equal(va[m].view(np.uint64), vb[m].view(np.uint64)))
