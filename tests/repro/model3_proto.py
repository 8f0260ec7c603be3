"""Prototype of the count-only append replay ("model v3"), checked against the CPU oracle.  Development scratch: validates the
theory (tables V / X, descent, final reconstruction) before the HIP implementation in csrc/sequencer.hip.

    python tests/repro/model3_proto.py
"""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import dsa_loader  # noqa: E402
import oracle_binding  # noqa: E402


class Geo:
    def __init__(self, cap, seg, height):
        self.C, self.seg, self.H = cap, seg, height
        t_h, t_0, p_h, p_0 = 0.7, 0.92, 0.3, 0.08
        t_d = (t_h - t_0) / height
        p_d = (p_h - p_0) / height
        self.W = [seg << h for h in range(height + 1)]
        self.lo = [math.ceil((p_0 + p_d * h) * self.W[h]) for h in range(height + 1)]
        self.hi = [math.floor((t_0 + t_d * h) * self.W[h]) for h in range(height + 1)]

    def acc(self, k, c):
        return self.lo[k] <= c <= self.hi[k]


def gap_D(W, E, k):
    return math.floor(k * (W / E))


def gaps_le(W, E, q):
    if E <= 0:
        return 0
    k = int((q + 1) * (E / W))
    k = max(0, min(E, k))
    while k < E and gap_D(W, E, k + 1) <= q:
        k += 1
    while k > 0 and gap_D(W, E, k) > q:
        k -= 1
    return k


def suffix_cells(W, c, w):
    """cells in the last w offsets of spread!(c cells over W slots)"""
    E = W - c
    return w - (E - gaps_le(W, E, W - w))


def pattern(W, c):
    E = W - c
    bits = np.ones(W, dtype=np.uint8)
    for k in range(1, E + 1):
        bits[gap_D(W, E, k) - 1] = 0
    return bits


class Model3:
    def __init__(self, geo, Lp):
        self.g = geo
        self.Lp = Lp
        self.Vm = {}
        self.Xm = {}
        self.table_lookups = 0
        self.top_events = 0

    def S_all(self, j, c):
        g = self.g
        return [suffix_cells(g.W[j], c, g.W[i]) for i in range(j)]

    def leaf_reject_time(self, c0):
        g = self.g
        if c0 + 1 < g.lo[0] or c0 + 1 > g.hi[0]:
            return 1
        return g.hi[0] - c0 + 1

    def V(self, k, c):
        key = (k, c)
        if key not in self.Vm:
            cnt = self.S_all(k, c)
            assert 1 <= cnt[0] <= self.g.seg - 1, ("leaf precondition", k, c, cnt[0])
            taus, st = self.taus(cnt, k)
            self.Vm[key] = (taus[k], st[k][0], st[k][1])
        return self.Vm[key]

    def X(self, k, c):
        g = self.g
        if not g.acc(k, c):
            return (0, 0, 0)
        key = (k, c)
        if key not in self.Xm:
            dt, reb, slots = 0, 0, 0
            cc = c
            while g.acc(k, cc):
                v = self.V(k, cc)
                dt += v[0]; reb += 1 + v[1]; slots += g.W[k] + v[2]
                cc += v[0]
            self.Xm[key] = (dt, reb, slots)
        return self.Xm[key]

    def taus(self, cnt, upto):
        """tau[i] = time of the first visit to level i (i = 1..upto) from counts cnt[0..upto-1]; st[i] = (rebalances, slots) of the
        complete chains below level i"""
        tau = [0] * (upto + 1)
        st = [(0, 0)] * (upto + 1)
        t = self.leaf_reject_time(cnt[0])
        if upto >= 1:
            tau[1] = t
        reb = slots = 0
        for i in range(1, upto):
            x = self.X(i, cnt[i] + t)
            self.table_lookups += 1
            t += x[0]; reb += x[1]; slots += x[2]
            tau[i + 1] = t
            st[i + 1] = (reb, slots)
        return tau, st

    def descend(self, kmax, cnt, b, ev, stats):
        """apply b appends from counts cnt[0..kmax-1]; level kmax is not visited within them.  Returns the number of trailing
        leaf-only ops."""
        g = self.g
        while True:
            if kmax == 0:
                return b
            tau, st = self.taus(cnt, kmax)
            if tau[1] > b:
                return b
            m = 1
            while m + 1 <= kmax and tau[m + 1] <= b:
                m += 1
            assert m < kmax, "level kmax visited inside the budget"
            stats[0] += st[m][0]; stats[1] += st[m][1]
            t = tau[m]
            c = cnt[m] + t
            assert g.acc(m, c)
            stats[0] += 1; stats[1] += g.W[m]
            while True:
                v = self.V(m, c)
                if t + v[0] > b:
                    break
                t += v[0]; c += v[0]
                assert g.acc(m, c)
                stats[0] += 1 + v[1]; stats[1] += g.W[m] + v[2]
            for i in range(m):
                ev.pop(i, None)
            ev[m] = c
            cnt = self.S_all(m, c)
            b -= t
            kmax = m

    def run(self, occ, R, types=None):
        """occ: uint8[C]; returns (new occ, consumed ops, rebalances, slots)"""
        g = self.g
        C, H, Lp = g.C, g.H, min(self.Lp, g.H)
        cnt = [int(occ[C - g.W[k]:].sum()) for k in range(H + 1)]
        assert 1 <= cnt[0] <= g.seg - 1
        ev = {}
        stats = [0, 0]
        t = 0
        leaf_ops = 0
        while True:
            tau, st = self.taus(cnt[:Lp + 1], Lp + 1)
            T = tau[Lp + 1]
            if t + T > R:
                leaf_ops = self.descend(Lp + 1, cnt[:Lp + 1], R - t, ev, stats)
                t = R
                break
            # visit to level Lp + 1 at append t + T
            kacc = None
            for k in range(Lp + 1, H + 1):
                if g.acc(k, cnt[k] + T):
                    kacc = k
                    break
            if kacc is None:
                leaf_ops = self.descend(Lp + 1, cnt[:Lp + 1], T - 1, ev, stats)
                t = t + T - 1
                break
            self.top_events += 1
            t += T
            stats[0] += st[Lp + 1][0] + 1; stats[1] += st[Lp + 1][1] + g.W[kacc]
            for k in range(Lp + 1, H + 1):
                cnt[k] += T
            c = cnt[kacc]
            for i in range(kacc):
                ev.pop(i, None)
            ev[kacc] = c
            low = self.S_all(kacc, c)
            assert 1 <= low[0] <= g.seg - 1
            cnt[:kacc] = low
        out = occ.copy()
        for k in sorted(ev.keys(), reverse=True):
            out[C - g.W[k]:] = pattern(g.W[k], ev[k])
        # the trailing leaf-only ops, bit-exact (src/writes.jl:26-43, src/pcsr.jl:99-112)
        for j in range(t - leaf_ops, t):
            is_sem = bool(types[j]) if types is not None else False
            leaf = out[C - g.seg:]
            nz = np.nonzero(leaf)[0]
            L = C - g.seg + int(nz[-1]) + 1          # 1-based tail
            if L < C and not is_sem:
                out[L] = 1
            else:
                z = np.nonzero(out[:C - 1] == 0)[0]
                pe = int(z[-1]) + 1                  # nearest empty slot left of the last slot (1-based)
                assert pe > C - g.seg
                if out[C - 1]:
                    out[pe - 1] = 1
                elif pe == C - 1:
                    out[C - 1] = 1
                else:
                    out[pe - 1] = 1; out[C - 2] = 0; out[C - 1] = 1
        return out, t, stats[0], stats[1]


def check_vector(dsa, oracle, n0, R, Lp, seed=1, grow_from=None):
    keys0 = np.arange(1, n0 + 1, dtype=np.int64) * 2
    vals0 = np.ones(n0)
    if grow_from is not None:          # a vector grown from a few keys keeps its small segments
        b = dsa.dynamicsparsevec(keys0[:grow_from], vals0[:grow_from], binding=oracle)
        b.set_batch(keys0[grow_from:], vals0[grow_from:])
    else:
        b = dsa.dynamicsparsevec(keys0, vals0, binding=oracle)
    nxt = 2 * n0 + 1
    done = 0
    m3 = None
    while done < R:
        info = b.info()
        geo = Geo(info["capacity"], info["segment_capacity"], info["height"])
        _, _, occ = b.export_layout()
        m3 = Model3(geo, Lp)
        out, used, reb, slots = m3.run(occ, R - done)
        ks = np.arange(nxt, nxt + used, dtype=np.int64)
        if used:
            b.set_batch(ks, np.ones(used))
        info2 = b.info()
        _, _, occ2 = b.export_layout()
        assert info2["capacity"] == info["capacity"], "the oracle extended inside the predicted run"
        assert np.array_equal(out, occ2), ("bitmap", n0, done, used, np.nonzero(out != occ2)[0][:10])
        assert info2["stat_rebalances"] - info["stat_rebalances"] == reb, ("reb", info2["stat_rebalances"] - info["stat_rebalances"], reb)
        assert info2["stat_window_slots"] - info["stat_window_slots"] == slots, ("slots",)
        print(f"  n0={n0} cap={geo.C} seg={geo.seg} H={geo.H} Lp={Lp}: {used} appends ok, reb={reb} slots={slots} "
              f"top events={m3.top_events} lookups={m3.table_lookups} V entries={len(m3.Vm)} X entries={len(m3.Xm)}")
        done += used
        nxt += used
        if done < R:
            # the next op extends (or needs the general path): one op through the oracle
            b.set_batch(np.array([nxt], dtype=np.int64), np.ones(1))
            assert b.info()["capacity"] != info["capacity"], "run ended without an extend"
            nxt += 1
            done += 1
    return m3


if __name__ == "__main__" and "v4" not in sys.argv:
    dsa = dsa_loader.load()
    oracle = oracle_binding.load(dsa)
    for n0, R, Lp in [(50000, 60000, 4), (90000, 40000, 7), (200000, 100000, 9)]:
        check_vector(dsa, oracle, n0, R, Lp, grow_from=3)
    for n0, R, Lp in [(40000, 30000, 4), (40000, 30000, 99), (40000, 30000, 1), (300000, 200000, 8), (50000, 777, 6), (46000, 5, 3),
                      (46000, 14, 3), (46000, 15, 3), (46000, 16, 3), (46000, 17, 3), (700000, 100000, 8), (30000, 100000, 5)]:
        check_vector(dsa, oracle, n0, R, Lp)


# ---------------------------------------------------------------------------------------------------------------------------------
# model v4: typed runs (semaphore cells) on 8-slot segments — leaf state (s, g), cross-leaf shifts, counts above
def trailing_gaps(W, c):
    bits = pattern(W, c)
    nz = np.nonzero(bits)[0]
    return W - (int(nz[-1]) + 1)


def model4_run(geo, occ, types):
    """types[j] = 1 for a semaphore cell; returns (suffix counts of levels >= 1 at the end, rebalances, slots, stop reason)"""
    C, H = geo.C, geo.H
    assert geo.seg == 8 and geo.hi[0] == 7
    cnt = [int(occ[C - geo.W[k]:].sum()) for k in range(H + 1)]
    leaf = occ[C - 8:]
    s = cnt[0]
    nz = np.nonzero(leaf)[0]
    g = 8 - (int(nz[-1]) + 1)
    sems = np.nonzero(types)[0].tolist() + [1 << 60]
    sp = 0
    t = 0
    reb = slots = 0
    end = len(types)
    while True:
        while sems[sp] < t:
            sp += 1
        ns = sems[sp]
        j = 7 - s
        cross = (s + g == 8) and ns == t + j
        ln = 8 - s + (1 if cross else 0)
        if t + ln > end:
            rest = end - t
            # the remaining ops stay inside the leaf epoch (a cross-leaf op among them would not count below its level: ignored here,
            # the caller compares levels >= 4 only)
            for k in range(1, H + 1):
                cnt[k] += rest
            return cnt, reb, slots, "end"
        kstar = 1
        if cross:
            kstar = None
            for k in range(1, H + 1):
                if cnt[k] + j < geo.W[k] - 1:
                    kstar = k
                    break
            assert kstar is not None
        for k in range(1, H + 1):
            cnt[k] += ln if k >= kstar else ln - 1
        t += ln
        kacc = None
        for k in range(1, H + 1):
            if geo.acc(k, cnt[k]):
                kacc = k
                break
        if kacc is None:
            return cnt, reb, slots, "extend"
        reb += 1; slots += geo.W[kacc]
        c = cnt[kacc]
        for i in range(1, kacc):
            cnt[i] = suffix_cells(geo.W[kacc], c, geo.W[i])
        s = suffix_cells(geo.W[kacc], c, 8)
        g = trailing_gaps(geo.W[kacc], c)
        assert 1 <= s <= 7 and g >= 0, (kacc, c, s, g)       # g == 0: fl(E * fl(W / E)) < W puts the last gap on W - 1 and a cell on the last slot


def check_matrix_v4(dsa, oracle, ncols0, batches, per=16, m=5000, seed=3):
    rng = np.random.default_rng(seed)
    B = dsa.dynamicsparse(fill_mode=False, binding=oracle)
    col = 0

    def make(ncols):
        nonlocal col
        I, J = [], []
        for _ in range(ncols):
            col += 1
            n = per if per > 0 else int(rng.integers(1, 20))
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
        for j in J:
            if j != last:
                types.append(1); last = j
            types.append(0)
        types = np.array(types, dtype=np.uint8)
        cnt, reb, slots, why = model4_run(geo, L0["occ"], types)
        B.set_batch(I, J, np.ones(len(I)))
        L1 = B.export_layout(0)
        if L1["info"]["capacity"] != info["capacity"]:
            print("  batch of %d columns: extend inside (model said %s) — skipped" % (nb, why))
            continue
        occ1 = L1["occ"]
        real = [int(occ1[geo.C - geo.W[k]:].sum()) for k in range(geo.H + 1)]
        assert real[4:] == cnt[4:], ("counts", real, cnt)
        dreb = L1["info"]["stat_rebalances"] - info["stat_rebalances"]
        dslots = L1["info"]["stat_window_slots"] - info["stat_window_slots"]
        assert (dreb, dslots) == (reb, slots), ("stats", dreb, reb, dslots, slots)
        print("  v4: batch of %d columns (%d cells) on cap %d ok: %d rebalances, %d slots" % (nb, len(types), geo.C, reb, slots))


if __name__ == "__main__" and "v4" in sys.argv:
    dsa = dsa_loader.load()
    oracle = oracle_binding.load(dsa)
    check_matrix_v4(dsa, oracle, 6000, [500, 1000, 300, 1000], per=16)
    check_matrix_v4(dsa, oracle, 9000, [700, 1000, 1000], per=0)
