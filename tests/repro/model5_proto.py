"""Prototype of the typed per-epoch append replay ("model v5": runs WITH semaphore cells on 8-slot segments, BASELINE config 5), checked
bit for bit against the CPU oracle.  Development scratch: validates the state abstraction (suffix counts per level, leaf state (s, g),
cross-leaf shifts), the surviving-event reconstruction of the bitmap and the hand-back of the trailing partial epoch before the HIP
implementation in csrc/appendmodel.hip (k_append_model5).

    python tests/repro/model5_proto.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dsa_loader  # noqa: E402
import oracle_binding  # noqa: E402
from model3_proto import Geo, pattern, suffix_cells  # noqa: E402


def trailing_gaps(W, c):
    bits = pattern(W, c)
    nz = np.nonzero(bits)[0]
    return W - (int(nz[-1]) + 1)


def slow_op(geo, occ, is_sem, stats):
    """ONE append on the bitmap, the reference's way: _insert! behind the tail (an element) / after the last slot (a semaphore,
    src/pcsr.jl:99-112), _look_for_rebalance! + _even_rebalance! (src/pma.jl:94-141).  Returns False when the op needs _extend!."""
    C = geo.C
    nz = np.nonzero(occ)[0]
    tail = int(nz[-1]) + 1                       # 1-based
    if not is_sem and tail < C:
        occ[tail] = 1; ip = tail + 1
    else:
        z = np.nonzero(occ[:C - 1] == 0)[0]
        pe = int(z[-1]) + 1
        if occ[C - 1]:
            occ[pe - 1] = 1
        elif pe == C - 1:
            occ[C - 1] = 1
        else:
            occ[pe - 1] = 1; occ[C - 2] = 0; occ[C - 1] = 1
        ip = C
    for k in range(geo.H + 1):
        W = geo.W[k]
        ws = ((ip - 1) // W) * W
        c = int(occ[ws:ws + W].sum())
        if geo.acc(k, c):
            if k > 0:
                occ[ws:ws + W] = pattern(W, c)
                stats[0] += 1; stats[1] += W
            return True
    return False


def model5_run(geo, occ, types):
    """Returns (bitmap after the consumed cells, consumed, rebalances, slots).  Consumes whole leaf epochs only: the trailing partial
    epoch (and an epoch that ends in _extend!) is left to the per-op replay."""
    C, H = geo.C, geo.H
    assert geo.seg == 8 and geo.hi[0] == 7 and geo.lo[0] == 1
    cnt = [int(occ[C - geo.W[k]:].sum()) for k in range(H + 1)]
    leaf = occ[C - 8:]
    s = cnt[0]
    if s < 1:
        return occ.copy(), 0, 0, 0
    nz = np.nonzero(leaf)[0]
    g = 8 - (int(nz[-1]) + 1)
    sems = np.nonzero(types)[0].tolist() + [1 << 60]
    sp = 0
    t = 0
    reb = slots = 0
    end = len(types)
    ev = {}
    while True:
        while sems[sp] < t:
            sp += 1
        ns = sems[sp]
        j = 7 - s
        cross = (s + g == 8) and ns == t + j
        ln = 8 - s + (1 if cross else 0)
        if t + ln > end:
            break
        kstar = 1
        if cross:
            kstar = None
            for k in range(1, H + 1):
                if cnt[k] + j < geo.W[k] - 1:
                    kstar = k
                    break
            if kstar is None:
                break
        new = [cnt[k] + (ln if k >= kstar else ln - 1) for k in range(H + 1)]
        kacc = None
        for k in range(1, H + 1):
            if geo.acc(k, new[k]):
                kacc = k
                break
        if kacc is None:
            break                                    # _extend!: the epoch is left to the per-op replay, which stops in front of its last op
        cnt = new
        t += ln
        reb += 1; slots += geo.W[kacc]
        c = cnt[kacc]
        for i in range(1, kacc):
            cnt[i] = suffix_cells(geo.W[kacc], c, geo.W[i])
            ev.pop(i, None)
        ev[kacc] = c
        s = suffix_cells(geo.W[kacc], c, 8)
        g = trailing_gaps(geo.W[kacc], c)
        assert 1 <= s <= 7 and 0 <= g <= 8 - s, (kacc, c, s, g)
    out = occ.copy()
    for k in sorted(ev.keys(), reverse=True):
        out[C - geo.W[k]:] = pattern(geo.W[k], ev[k])
    return out, t, reb, slots


def check_matrix(dsa, oracle, ncols0, batches, per=16, m=5000, seed=3, verbose=True):
    rng = np.random.default_rng(seed)
    B = dsa.dynamicsparse(fill_mode=False, binding=oracle)
    col = 0

    def make(ncols):
        nonlocal col
        I, J = [], []
        for _ in range(ncols):
            col += 1
            n = per if per > 0 else int(rng.integers(1, -per + 1))
            rows = sorted(rng.choice(m, size=n, replace=False) + 1)
            I += [int(r) for r in rows]; J += [col] * n
        return np.array(I, dtype=np.int64), np.array(J, dtype=np.int64)
    I, J = make(ncols0)
    B.set_batch(I, J, np.ones(len(I)))
    for nb in batches:
        L0 = B.export_layout(0)
        info = L0["info"]
        geo = Geo(info["capacity"], info["segment_capacity"], info["height"])
        I, J = make(nb)
        types = []
        last = None
        for jj in J:
            if jj != last:
                types.append(1); last = jj
            types.append(0)
        types = np.array(types, dtype=np.uint8)
        out, used, reb, slots = model5_run(geo, L0["occ"], types)
        # the rest of the run, op by op
        stats = [reb, slots]
        done = used
        ok = True
        while done < len(types) and ok:
            ok = slow_op(geo, out, bool(types[done]), stats)
            if ok:
                done += 1
        B.set_batch(I, J, np.ones(len(I)))
        L1 = B.export_layout(0)
        if L1["info"]["capacity"] != info["capacity"]:
            assert not ok, "the oracle extended, the replay did not notice"
            if verbose:
                print("  batch of %d columns: _extend! inside at cell %d (model consumed %d) — layouts not compared" % (nb, done, used))
            continue
        assert ok and done == len(types)
        assert np.array_equal(out, L1["occ"]), ("bitmap", np.nonzero(out != L1["occ"])[0][:10], used, len(types))
        dreb = L1["info"]["stat_rebalances"] - info["stat_rebalances"]
        dslots = L1["info"]["stat_window_slots"] - info["stat_window_slots"]
        assert (dreb, dslots) == (stats[0], stats[1]), ("stats", dreb, stats[0], dslots, stats[1])
        if verbose:
            print("  v5: batch of %d columns (%d cells) on cap %d: model consumed %d, bitmap + statistics ok (%d rebalances)" % (nb, len(types), geo.C, used, dreb))


if __name__ == "__main__":
    dsa = dsa_loader.load()
    oracle = oracle_binding.load(dsa)
    check_matrix(dsa, oracle, 3000, [500, 1000, 300, 1000, 5, 17, 1], per=16)
    check_matrix(dsa, oracle, 4000, [700, 1000, 1000, 3], per=-20)          # column lengths 1..20
    check_matrix(dsa, oracle, 2000, [300, 300, 300, 300, 300, 300], per=-3, seed=5)   # short columns: semaphores every 2..4 cells
    check_matrix(dsa, oracle, 1500, [2000], per=-1, seed=6)                  # one element per column
    for sd in range(10, 30):
        check_matrix(dsa, oracle, 200 + 37 * sd, [50 + sd, 400, 90], per=-(1 + sd % 9), seed=sd, verbose=False)
    print("model v5 prototype: all checks passed")
