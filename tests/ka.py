"""Procedural known-answer scenarios KA-1..KA-9 of SURVEY.md App. B, runnable on any backend.
Each returns a dict of observations compared with tests/golden/survey_known_answers.json."""
import numpy as np

from util import SplitMix64, layout_digest, table_digest


def _occ_pos(o):
    return (np.nonzero(o)[0] + 1).tolist()


def ka4(dsa, b):
    v = dsa.dynamicsparsevec([], [], binding=b)
    ext = []
    cap = v.info()["capacity"]
    for k in range(1, 1001):
        v[k] = float(k)
        c = v.info()["capacity"]
        if c != cap:
            ext.append([k, c])
            cap = c
    k_, v_, o = v.export_layout()
    inf = v.info()
    p = _occ_pos(o)
    return dict(extends=ext, capacity=inf["capacity"], seg=inf["segment_capacity"], height=inf["height"],
                first=p[:8], last=p[-4:], digest=hex(layout_digest(k_, o)))


def ka5(dsa, b):
    v = dsa.dynamicsparsevec([], [], binding=b)
    ext = []
    cap = v.info()["capacity"]
    for k in range(1000, 0, -1):
        v[k] = float(k)
        c = v.info()["capacity"]
        if c != cap:
            ext.append([k, c])
            cap = c
    k_, v_, o = v.export_layout()
    p = _occ_pos(o)
    return dict(extends=ext, first=p[:8], last=p[-4:], digest=hex(layout_digest(k_, o)))


def ka6(dsa, b, batch=False):
    g = SplitMix64(42)
    ks = [1 + g.next() % 10 ** 6 for _ in range(5000)]
    v = dsa.dynamicsparsevec([], [], binding=b)
    vals = [1.0 + i % 7 for i in range(5000)]
    if batch:
        v.set_batch(ks, vals)
    else:
        for i, k in enumerate(ks):
            v[k] = vals[i]
    out = {}
    k_, v_, o = v.export_layout()
    inf = v.info()
    out["a"] = dict(capacity=inf["capacity"], n=inf["nb_elements"], seg=inf["segment_capacity"],
                    height=inf["height"], digest=hex(layout_digest(k_, o)))
    if batch:
        v.set_batch(ks[0::3], [0.0] * len(ks[0::3]))
    else:
        for i in range(0, 5000, 3):
            v[ks[i]] = 0
    k_, v_, o = v.export_layout()
    inf = v.info()
    out["b"] = dict(capacity=inf["capacity"], n=inf["nb_elements"], digest=hex(layout_digest(k_, o)))
    shr = []
    cap = inf["capacity"]
    if batch:
        idx = [i for i in range(5000) if i % 3 != 0]
        v.set_batch([ks[i] for i in idx], [0.0] * len(idx))
    else:
        for i in range(5000):
            if i % 3 != 0:
                v[ks[i]] = 0
                c = v.info()["capacity"]
                if c != cap:
                    shr.append([i, c])
                    cap = c
    k_, v_, o = v.export_layout()
    inf = v.info()
    out["c"] = dict(capacity=inf["capacity"], n=inf["nb_elements"], seg=inf["segment_capacity"],
                    nb_segs=inf["nb_segments"], height=inf["height"], digest=hex(layout_digest(k_, o)))
    if not batch:
        out["c"]["shrinks"] = shr
    return out


def _mat_obs(a, o):
    L = a.export_layout(o)
    inf = L["info"]
    return dict(capacity=inf["capacity"], n=inf["nb_elements"], seg=inf["segment_capacity"], height=inf["height"],
                partitions=inf["nb_partitions"], nnz=inf["nb_elements"] - inf["nb_partitions"],
                digest=hex(layout_digest(L["keys"], L["occ"])), sem_digest=hex(table_digest(L["semaphores"])),
                sems6=L["semaphores"][:6].tolist())


def ka7_ops():
    g = SplitMix64(7)
    ops = []
    for _ in range(3000):
        r = 1 + g.next() % 50
        c = 1 + g.next() % 80
        z = g.next() % 4
        v = 0.0 if z == 0 else float(1 + g.next() % 9)
        ops.append((r, c, v))
    return ops


def ka7_9(dsa, b, batch=False):
    a = dsa.dynamicsparse([1], [1], [1.0], binding=b)
    ops = ka7_ops()
    if batch:
        a.set_batch([o[0] for o in ops], [o[1] for o in ops], [o[2] for o in ops])
    else:
        for r, c, v in ops:
            a[r, c] = v
    out = {"ka7": dict(col=_mat_obs(a, 0), row=_mat_obs(a, 1))}
    x = np.array([1.0 / j for j in range(1, 81)])
    y = a.mul(x)
    yi, yv = a.mul((list(range(1, 81)), x.tolist()))
    out["ka7"]["y4"] = [float(t).hex() for t in y[:4]]
    out["ka7"]["y4_sparse"] = [float(t).hex() for t in yv[:4]]
    out["ka7"]["touched"] = int(len(yi))
    for c in range(10, 21):
        a.deletecolumn(c)
    for r in range(5, 10):
        a.deleterow(r)
    out["ka8"] = dict(col=_mat_obs(a, 0), row=_mat_obs(a, 1))
    a[2, 15] = 3.0
    a[3, 81] = 4.0
    out["ka9"] = dict(col=_mat_obs(a, 0), row=_mat_obs(a, 1))
    return out
