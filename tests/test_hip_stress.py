"""GPU stress test of the host side of the shim (csrc/dsa_host.hip: caching allocator, pinned landing-area leases, stream pool,
helper threads of matrix batches, pinned polling): two host threads drive many handles with random API calls at the same time
(ctypes releases the GIL inside a call, so the library really runs concurrently), under faulthandler — a native crash prints every
thread's Python stack.  Values are checked against plain Python dictionaries, structures with the device invariant checker."""
import faulthandler
import threading

import numpy as np
import pytest

from util import SplitMix64

pytestmark = pytest.mark.gpu


def _worker(dsa, hip, seed, seconds_of_work, errors):
    try:
        g = SplitMix64(seed)
        vecs, mats = [], []                       # (handle, model dict)
        for step in range(seconds_of_work):
            r = g.next() % 100
            if r < 12 or not vecs:
                n = 1 + g.next() % 40
                keys = sorted({1 + g.next() % 5000 for _ in range(n)})
                vals = [float(1 + g.next() % 9) for _ in keys]
                v = dsa.dynamicsparsevec(keys, vals, binding=hip)
                if g.next() % 3 == 0:
                    v.set_wait_policy(1)
                vecs.append((v, dict(zip(keys, vals))))
            elif r < 20 or not mats:
                A = dsa.dynamicsparse(fill_mode=False, binding=hip)
                if g.next() % 3 == 0:
                    A.set_wait_policy(1)
                mats.append((A, {}))
            elif r < 45:
                v, model = vecs[g.next() % len(vecs)]
                n = 1 + g.next() % 300
                ks = [1 + g.next() % 5000 for _ in range(n)]
                vs = [float(g.next() % 5) for _ in ks]
                v.set_batch(ks, vs)
                for k, x in zip(ks, vs):
                    if x == 0.0:
                        model.pop(k, None)
                    else:
                        model[k] = x
            elif r < 60:
                v, model = vecs[g.next() % len(vecs)]
                k = 1 + g.next() % 5000
                assert v[k] == model.get(k, 0.0), ("vec get", k)
                if g.next() % 4 == 0:
                    kk, vv = v.nonzeros()
                    assert kk.tolist() == sorted(model) and vv.tolist() == [model[q] for q in sorted(model)]
            elif r < 80:
                A, model = mats[g.next() % len(mats)]
                n = 1 + g.next() % 400
                I = [1 + g.next() % 300 for _ in range(n)]
                J = [1 + g.next() % 200 for _ in range(n)]
                V = [float(g.next() % 6) for _ in range(n)]
                A.set_batch(I, J, V)
                for i, j, x in zip(I, J, V):
                    if x == 0.0:
                        model.pop((i, j), None)
                    else:
                        model[(i, j)] = x
            elif r < 90:
                A, model = mats[g.next() % len(mats)]
                if model:
                    j = 1 + g.next() % 200
                    col = sorted((i, x) for (i, jj), x in model.items() if jj == j)
                    got = A.col_view(j) if col else None
                    if col:
                        assert got == [(i, x) for i, x in col], ("col view", j)
                    i = 1 + g.next() % 300
                    assert A[i, j] == model.get((i, j), 0.0)
                    if g.next() % 3 == 0:
                        x = np.ones(200)
                        y = A.mul(x, dense_out=300)
                        ref = np.zeros(300)
                        for (ii, jj), xx in model.items():
                            ref[ii - 1] += xx
                        np.testing.assert_allclose(y, ref, rtol=1e-12, atol=0)
            elif r < 95 and len(vecs) > 2:
                v, _ = vecs.pop(g.next() % len(vecs))
                assert not v.check()[2:7].any()
                v.close()
            elif len(mats) > 2:
                A, _ = mats.pop(g.next() % len(mats))
                for o in (dsa.COLMAJOR, dsa.ROWMAJOR):
                    assert not A.check(o)[2:7].any()
                A.close()
        for v, model in vecs:
            assert v.nnz() == len(model) and not v.check()[2:7].any()
        for A, model in mats:
            assert A.nnz() == len(model)
    except BaseException as e:          # noqa: BLE001 — reported by the main thread
        errors.append((seed, repr(e)))
        raise


def test_two_host_threads_many_handles_random_calls(dsa, hip):
    faulthandler.enable()
    errors = []
    ts = [threading.Thread(target=_worker, args=(dsa, hip, 1000 + t, 1500, errors)) for t in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    dsa.pool_trim(0, binding=hip)
