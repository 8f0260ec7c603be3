import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import dsa_loader
dsa = dsa_loader.load(); hip = dsa.product()
rng = np.random.default_rng(5)
keys = np.sort(rng.choice(10 ** 6, size=20000, replace=False)) + 1
vals = rng.random(20000) + 1.0
src = dsa.dynamicsparsevec(keys, vals, binding=hip)
more = rng.choice(10 ** 6, size=5000) + 1
src.set_batch(more, np.where(rng.random(5000) < 0.2, 0.0, 2.0))
k, v, o = src.export_layout()
seg = src.info()["segment_capacity"]
t = dsa.import_vector_layout(k, v, o, seg, n=len(src), binding=hip)
print("len", len(t), len(src), "info", t.info(), src.info())
ka, va = t.nonzeros(); kb, vb = src.nonzeros()
print("nonzeros equal", np.array_equal(ka, kb), np.array_equal(va, vb), len(ka), len(kb))
print("eq t==src", t == src, "src==t", src == t, "t==t", t == t)
t2 = dsa.import_vector_layout(k, v, o, seg, n=len(src), binding=hip)
print("eq t==t2", t == t2)
s2 = dsa.dynamicsparsevec(ka, va, binding=hip)
print("eq src==s2 (rebuilt, other layout)", src == s2, len(s2), len(src))
