"""Host-side mirror of the reference's public surface (src/DynamicSparseArrays.jl:5-16)
over the C ABI of include/dsa.h.

The reference is a Julia package and there is no Julia toolchain in the build image,
so this Python module plays the role of the Julia wrapper for testing: same names
(`dynamicsparsevec`, `dynamicsparse`, `deletecolumn`, `deleterow`, `addrow`,
`closefillmode`, `shrink_size`, `nbpartitions`, `nnz`, indexing with ``[]``), same
argument meaning, same error behaviour (ArgumentError -> ValueError subclass,
BoundsError -> IndexError subclass, ErrorException -> RuntimeError subclass).
The Julia wrapper a maintainer would ship is in INTEGRATION.md / julia/.

Every object takes the `Binding` it runs on; by default that is the HIP product
library (`binding.product()`), which raises if it is not built.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import binding as B
from .binding import Binding, INFO, INFO_COUNT, P_F64, P_I64, P_U8, VP, _f64, _i64

COMBINE = {"+": 0, "add": 0, "*": 1, "mul": 1, "last": 2}
COLMAJOR, ROWMAJOR = 0, 1


def _bind(b):
    return b if b is not None else B.product()


class _Handle:
    _destroy = None

    def __init__(self, b: Binding, h):
        self.b = b
        self.h = h

    def close(self):
        if self.h is not None:
            self.b.call(self._destroy, self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DynamicSparseVector(_Handle):
    """DynamicSparseVector{Int64,Float64}  (src/vector.jl:1-4)."""
    _destroy = "vec_destroy"

    def __getitem__(self, key):                       # src/vector.jl:73
        out = C.c_double()
        self.b.call("vec_get", self.h, int(key), C.byref(out))
        return out.value

    def __setitem__(self, key, value):                # src/vector.jl:76-81
        self.b.call("vec_set", self.h, int(key), float(value))

    def get_batch(self, keys):
        k, kp = _i64(keys)
        out = np.empty(len(k), dtype=np.float64)
        self.b.call("vec_get_batch", self.h, kp, len(k), out.ctypes.data_as(P_F64))
        return out

    def set_batch(self, keys, vals):
        k, kp = _i64(keys)
        v, vp = _f64(vals)
        assert len(k) == len(v)
        self.b.call("vec_set_batch", self.h, kp, vp, len(k))

    def __len__(self):                                # length(v)  src/vector.jl:69
        out = C.c_int64()
        self.b.call("vec_len", self.h, C.byref(out))
        return out.value

    def nnz(self):                                    # src/vector.jl:88
        out = C.c_int64()
        self.b.call("vec_nnz", self.h, C.byref(out))
        return out.value

    def shrink_size(self):                            # shrink_size!  src/vector.jl:64
        self.b.call("vec_shrink_size", self.h)

    def info(self):
        a = np.zeros(INFO_COUNT, dtype=np.int64)
        self.b.call("vec_info", self.h, a.ctypes.data_as(P_I64))
        return {k: int(a[i]) for k, i in INFO.items()}

    def nonzeros(self):
        """(keys, values) of the stored entries in iteration (slot) order  src/vector.jl:71,93-109."""
        n = self.nnz()
        k = np.empty(max(n, 1), dtype=np.int64)
        v = np.empty(max(n, 1), dtype=np.float64)
        out = C.c_int64()
        self.b.call("vec_nonzeros", self.h, k.ctypes.data_as(P_I64), v.ctypes.data_as(P_F64), len(k), C.byref(out))
        return k[:out.value], v[:out.value]

    def __iter__(self):
        k, v = self.nonzeros()
        return iter(zip(k.tolist(), v.tolist()))

    def export_layout(self):
        cap = self.info()["capacity"]
        k = np.empty(cap, dtype=np.int64)
        v = np.empty(cap, dtype=np.float64)
        o = np.empty(cap, dtype=np.uint8)
        self.b.call("vec_export_layout", self.h, k.ctypes.data_as(P_I64), v.ctypes.data_as(P_F64),
                    o.ctypes.data_as(P_U8), cap)
        return k, v, o

    def rebalance_root(self):
        self.b.call("vec_rebalance_root", self.h)

    def check(self):
        """device-side invariant checker (HIP library only): report[2..6] must be 0."""
        r = np.zeros(8, dtype=np.int64)
        self.b.call("vec_check", self.h, r.ctypes.data_as(P_I64))
        return r

    def set_wait_policy(self, policy):
        """0 (DSA_WAIT_SPIN): blocking calls poll pinned memory; 1 (DSA_WAIT_BLOCK): they park in hipStreamSynchronize first."""
        self.b.call("vec_set_wait_policy", self.h, int(policy))

    def __eq__(self, other):                          # src/vector.jl:85-87, src/pma.jl:236-266
        if not isinstance(other, DynamicSparseVector):
            return NotImplemented
        if other.b is not self.b:
            raise B.DsaArgumentError(B.EARG, "vectors of two different libraries")
        out = C.c_int32()
        self.b.call("vec_equal", self.h, other.h, C.byref(out))
        return bool(out.value)

    def axpby(self, alpha, other, beta):
        """alpha*self + beta*other as ascending (keys, values): the SparseVector of v1 + v2 / v1 - v2 / -v
        (AbstractSparseVector fallbacks over src/vector.jl:93-109; test/functional/math.jl:53-94)."""
        if other.b is not self.b:
            raise B.DsaArgumentError(B.EARG, "vectors of two different libraries")
        cap = max(self.nnz() + other.nnz(), 1)
        k = np.empty(cap, dtype=np.int64)
        v = np.empty(cap, dtype=np.float64)
        out = C.c_int64()
        self.b.call("vec_axpby", self.h, float(alpha), other.h, float(beta), k.ctypes.data_as(P_I64), v.ctypes.data_as(P_F64),
                    cap, C.byref(out))
        return k[:out.value], v[:out.value]

    def __add__(self, other):
        return self.axpby(1.0, other, 1.0)

    def __sub__(self, other):
        return self.axpby(1.0, other, -1.0)

    def __neg__(self):
        k, v = self.nonzeros()
        return k, -v

    def filter(self, f):
        """filter(f, v)  src/vector.jl:83 -> src/pma.jl:224-234: a NEW dynamic vector of the stored (key, value) pairs e with f(e)."""
        k, v = self.nonzeros()
        sel = np.fromiter((bool(f((int(a), float(b)))) for a, b in zip(k, v)), dtype=bool, count=len(k))
        return dynamicsparsevec(k[sel], v[sel], binding=self.b)

    __hash__ = None


def dynamicsparsevec(I, V, combine="+", n=None, binding: Binding | None = None) -> DynamicSparseVector:
    """dynamicsparsevec(I, V, [combine, n])  src/vector.jl:44-62."""
    b = _bind(binding)
    if len(I) != len(V):
        raise B.DsaArgumentError(B.EARG, "keys & nonzeros vectors must have same length.")
    k, kp = _i64(I)
    v, vp = _f64(V)
    h = VP()
    b.call("vec_create", kp, vp, len(k), COMBINE[combine], -1 if n is None else int(n), C.byref(h))
    return DynamicSparseVector(b, h)


def import_vector_layout(keys, vals, occ, segment_capacity, n=None, binding: Binding | None = None) -> DynamicSparseVector:
    """A vector restored from an exported layout (`DynamicSparseVector.export_layout` + its segment capacity): snapshot / restore,
    and the way a test puts a structure into an arbitrary state.  No reference counterpart (include/dsa.h: dsa_vec_import_layout)."""
    b = _bind(binding)
    k, kp = _i64(keys)
    v, vp = _f64(vals)
    o = np.ascontiguousarray(occ, dtype=np.uint8)
    assert len(k) == len(v) == len(o)
    if n is None:
        n = int(k[o.astype(bool)].max()) if o.any() else 0
    h = VP()
    b.call("vec_import_layout", kp, vp, o.ctypes.data_as(P_U8), len(o), int(segment_capacity), int(n), C.byref(h))
    return DynamicSparseVector(b, h)


class PackedCSC(_Handle):
    """PackedCSC{Int64,Float64}  (src/pcsr.jl:4-9)."""
    _destroy = "pcsc_destroy"

    def __getitem__(self, idx):                       # src/pcsr.jl:228-232
        key, partition = idx
        out = C.c_double()
        self.b.call("pcsc_get", self.h, int(key), int(partition), C.byref(out))
        return out.value

    def __setitem__(self, idx, value):                # src/pcsr.jl:294-310
        key, partition = idx
        self.b.call("pcsc_set", self.h, float(value), int(key), int(partition))

    def deletepartition(self, partition):             # src/pcsr.jl:188-204
        self.b.call("pcsc_deletepartition", self.h, int(partition))

    def nnz(self):
        out = C.c_int64()
        self.b.call("pcsc_nnz", self.h, C.byref(out))
        return out.value

    def nbpartitions(self):
        out = C.c_int64()
        self.b.call("pcsc_nbpartitions", self.h, C.byref(out))
        return out.value

    def info(self):
        a = np.zeros(INFO_COUNT, dtype=np.int64)
        self.b.call("pcsc_info", self.h, a.ctypes.data_as(P_I64))
        return {k: int(a[i]) for k, i in INFO.items()}

    def export_layout(self):
        inf = self.info()
        cap, tl = inf["capacity"], max(inf["table_len"], 1)
        k = np.empty(cap, dtype=np.int64)
        v = np.empty(cap, dtype=np.float64)
        o = np.empty(cap, dtype=np.uint8)
        s = np.zeros(tl, dtype=np.int64)
        self.b.call("pcsc_export_layout", self.h, k.ctypes.data_as(P_I64), v.ctypes.data_as(P_F64),
                    o.ctypes.data_as(P_U8), cap, s.ctypes.data_as(P_I64), tl)
        return k, v, o, s[:inf["table_len"]]


def packedcsc(row_keys, values, combine="+", binding: Binding | None = None) -> PackedCSC:
    """PackedCSC(row_keys::Vector{Vector}, values::Vector{Vector}, combine)  src/pcsr.jl:26-63."""
    b = _bind(binding)
    assert len(row_keys) == len(values)
    colptr = np.zeros(len(row_keys) + 1, dtype=np.int64)
    for p, r in enumerate(row_keys):
        assert len(r) == len(values[p])
        colptr[p + 1] = colptr[p] + len(r)
    rk = np.array([x for r in row_keys for x in r], dtype=np.int64)
    vv = np.array([x for r in values for x in r], dtype=np.float64)
    if len(rk) == 0:
        rk = np.zeros(1, dtype=np.int64)
        vv = np.zeros(1, dtype=np.float64)
    h = VP()
    b.call("pcsc_create", colptr.ctypes.data_as(P_I64), len(row_keys), rk.ctypes.data_as(P_I64),
           vv.ctypes.data_as(P_F64), COMBINE[combine], C.byref(h))
    return PackedCSC(b, h)


def import_packedcsc_layout(keys, vals, occ, segment_capacity, semaphores, binding: Binding | None = None) -> PackedCSC:
    """A PackedCSC restored from an exported layout (include/dsa.h: dsa_pcsc_import_layout)."""
    b = _bind(binding)
    k, kp = _i64(keys)
    v, vp = _f64(vals)
    o = np.ascontiguousarray(occ, dtype=np.uint8)
    s, sp = _i64(semaphores if len(semaphores) else [0])
    h = VP()
    b.call("pcsc_import_layout", kp, vp, o.ctypes.data_as(P_U8), len(o), int(segment_capacity), sp, len(semaphores), C.byref(h))
    return PackedCSC(b, h)


def packedcsc_empty(binding: Binding | None = None) -> PackedCSC:
    b = _bind(binding)
    h = VP()
    b.call("pcsc_create_empty", C.byref(h))
    return PackedCSC(b, h)


class Transposed:
    """transpose(mat)  src/operations.jl:1-9."""

    def __init__(self, mat):
        self.array = mat

    def __getitem__(self, idx):
        r, c = idx
        return self.array[c, r]

    def __setitem__(self, idx, val):
        r, c = idx
        self.array[c, r] = val

    def size(self):
        m, n = self.array.size()
        return (n, m)

    def mul(self, x, **kw):
        return self.array.mul(x, transpose=True, **kw)


class DynamicSparseMatrix(_Handle):
    """DynamicSparseMatrix{Int64,Int64,Float64}  (src/matrix.jl:1-8)."""
    _destroy = "mat_destroy"

    def __setitem__(self, idx, val):                  # src/matrix.jl:43-62
        row, col = idx
        self.b.call("mat_set", self.h, float(val), int(row), int(col))

    def __getitem__(self, idx):                       # src/matrix.jl:64-68
        row, col = idx
        out = C.c_double()
        self.b.call("mat_get", self.h, int(row), int(col), C.byref(out))
        return out.value

    def set_batch(self, I, J, V):
        i, ip = _i64(I)
        j, jp = _i64(J)
        v, vp = _f64(V)
        assert len(i) == len(j) == len(v)
        self.b.call("mat_set_batch", self.h, ip, jp, vp, len(i))

    def get_batch(self, I, J):
        i, ip = _i64(I)
        j, jp = _i64(J)
        out = np.empty(len(i), dtype=np.float64)
        self.b.call("mat_get_batch", self.h, ip, jp, len(i), out.ctypes.data_as(P_F64))
        return out

    def addrow(self, row, colids, vals):              # addrow!  src/matrix.jl:113-124
        c, cp = _i64(colids)
        v, vp = _f64(vals)
        assert len(c) == len(v)
        self.b.call("mat_addrow", self.h, int(row), cp, vp, len(c))

    def closefillmode(self):                          # closefillmode!  src/matrix.jl:126-134
        self.b.call("mat_closefillmode", self.h)

    def deletecolumn(self, col):                      # deletecolumn!  src/matrix.jl:95-102
        self.b.call("mat_deletecolumn", self.h, int(col))

    def deleterow(self, row):                         # deleterow!  src/matrix.jl:104-111
        self.b.call("mat_deleterow", self.h, int(row))

    def _view(self, name, key):
        cap = 64
        while True:
            k = np.empty(cap, dtype=np.int64)
            v = np.empty(cap, dtype=np.float64)
            n = C.c_int64()
            try:
                self.b.call(name, self.h, int(key), k.ctypes.data_as(P_I64), v.ctypes.data_as(P_F64), cap, C.byref(n))
            except B.DsaError as e:
                if e.code == B.ECAP:
                    cap *= 8
                    continue
                raise
            return list(zip(k[:n.value].tolist(), v[:n.value].tolist()))

    def col_view(self, col):                          # @view m[:, col]  src/matrix.jl:83-88
        return self._view("mat_col_view", col)

    def row_view(self, row):                          # @view m[row, :]  src/matrix.jl:70-81
        return self._view("mat_row_view", row)

    def _view_dev(self, name, key, d_keys, d_vals, cap):
        """the view delivered into HBM: d_keys / d_vals are device addresses (int64 / float64 arrays of cap entries, e.g.
        tensor.data_ptr()); returns the number of cells; the copy is enqueued on the orientation's stream (sync())"""
        n = C.c_int64()
        self.b.call(name, self.h, int(key), C.c_void_p(int(d_keys)), C.c_void_p(int(d_vals)), int(cap), C.byref(n))
        return n.value

    def sync(self):                                   # dsa_mat_sync: everything enqueued on the handle's streams has finished
        self.b.call("mat_sync", self.h)

    def col_view_dev(self, col, d_rows, d_vals, cap):  # @view m[:, col] into device memory
        return self._view_dev("mat_col_view_dev", col, d_rows, d_vals, cap)

    def row_view_dev(self, row, d_cols, d_vals, cap):  # @view m[row, :] into device memory
        return self._view_dev("mat_row_view_dev", row, d_cols, d_vals, cap)

    def col_slice(self, col):                         # m[:, col]  src/pcsr.jl:285-291
        h = VP()
        self.b.call("mat_col_slice", self.h, int(col), C.byref(h))
        return DynamicSparseVector(self.b, h)

    def row_slice(self, row):                         # m[row, :]  src/pcsr.jl:269-283
        h = VP()
        self.b.call("mat_row_slice", self.h, int(row), C.byref(h))
        return DynamicSparseVector(self.b, h)

    def nnz(self):
        out = C.c_int64()
        self.b.call("mat_nnz", self.h, C.byref(out))
        return out.value

    def size(self):
        m, n = C.c_int64(), C.c_int64()
        self.b.call("mat_size", self.h, C.byref(m), C.byref(n))
        return (m.value, n.value)

    def nbpartitions(self, orientation):
        out = C.c_int64()
        self.b.call("mat_nbpartitions", self.h, orientation, C.byref(out))
        return out.value

    def info(self, orientation):
        a = np.zeros(INFO_COUNT, dtype=np.int64)
        self.b.call("mat_info", self.h, orientation, a.ctypes.data_as(P_I64))
        return {k: int(a[i]) for k, i in INFO.items()}

    def export_layout(self, orientation):
        inf = self.info(orientation)
        cap, tl = inf["capacity"], max(inf["table_len"], 1)
        k = np.empty(cap, dtype=np.int64)
        v = np.empty(cap, dtype=np.float64)
        o = np.empty(cap, dtype=np.uint8)
        s = np.zeros(tl, dtype=np.int64)
        ck = np.zeros(tl, dtype=np.int64)
        cl = np.zeros(tl, dtype=np.uint8)
        self.b.call("mat_export_layout", self.h, orientation, k.ctypes.data_as(P_I64), v.ctypes.data_as(P_F64),
                    o.ctypes.data_as(P_U8), cap, s.ctypes.data_as(P_I64), ck.ctypes.data_as(P_I64),
                    cl.ctypes.data_as(P_U8), tl)
        n = inf["table_len"]
        return dict(keys=k, vals=v, occ=o, semaphores=s[:n], col_keys=ck[:n], col_live=cl[:n], info=inf)

    def rebalance_root(self, orientation):
        self.b.call("mat_rebalance_root", self.h, orientation)

    def check(self, orientation):
        """device-side invariant checker (HIP library only): report[2..6] must be 0."""
        r = np.zeros(8, dtype=np.int64)
        self.b.call("mat_check", self.h, orientation, r.ctypes.data_as(P_I64))
        return r

    def set_wait_policy(self, policy):
        """0 (DSA_WAIT_SPIN): blocking calls poll pinned memory; 1 (DSA_WAIT_BLOCK): they park in hipStreamSynchronize first."""
        self.b.call("mat_set_wait_policy", self.h, int(policy))

    def transpose(self):
        return Transposed(self)

    @property
    def T(self):
        return Transposed(self)

    def mul(self, x, transpose=False, dense_out=None, out=None):
        """mat * v / transpose(mat) * v  (src/operations.jl:14-36).

        `x` is a DynamicSparseVector, a (indices, values) pair of the stored entries
        (ascending indices), or a dense numpy array.  Sparse inputs return
        (indices, values) of the touched rows, ascending — the `_mul_output` shape;
        a dense array returns a dense array of length size(mat, 1 | 2).
        `out` = (int64 array, float64 array) the caller keeps across products (each at least as long as the result): the
        result is fetched into them and views are returned — a long result does not pay for fresh pages every time.
        """
        if isinstance(x, np.ndarray):
            m, n = self.size()
            ny = (n if transpose else m) if dense_out is None else dense_out
            xx, xp = _f64(x)
            y = np.empty(max(ny, 1), dtype=np.float64)
            self.b.call("mat_spmv_dense", self.h, 1 if transpose else 0, xp, len(xx), y.ctypes.data_as(P_F64), ny)
            return y[:ny]
        if isinstance(x, DynamicSparseVector):
            xi, xv = x.nonzeros()
        else:
            xi, xv = x
        xi, xip = _i64(xi)
        xv, xvp = _f64(xv)
        tr = 1 if transpose else 0
        n_out = C.c_int64()
        if "mat_spmv_sparse_begin" in self.b.SIGNATURES:
            # compute, learn the number of touched rows, allocate exactly that, fetch (one product, whatever the result size)
            self.b.call("mat_spmv_sparse_begin", self.h, tr, xip, xvp, len(xi), C.byref(n_out))
            cnt = n_out.value
            if out is not None and len(out[0]) >= cnt and len(out[1]) >= cnt:
                yi, yv = out[0][:cnt], out[1][:cnt]
            else:
                yi = np.empty(cnt, dtype=np.int64)
                yv = np.empty(cnt, dtype=np.float64)
            if cnt:
                self.b.call("mat_spmv_sparse_fetch", self.h, yi.ctypes.data_as(P_I64), yv.ctypes.data_as(P_F64), cnt, C.byref(n_out))
            return yi, yv
        m, n = self.size()
        cap = max(n if transpose else m, 1)                # the touched rows are at most all rows
        yi = np.empty(cap, dtype=np.int64)
        yv = np.empty(cap, dtype=np.float64)
        self.b.call("mat_spmv_sparse", self.h, tr, xip, xvp, len(xi), yi.ctypes.data_as(P_I64), yv.ctypes.data_as(P_F64), cap, C.byref(n_out))
        return yi[:n_out.value].copy(), yv[:n_out.value].copy()

    def mul_dev(self, d_xi, d_xv, nx, d_yi, d_yv, cap, d_count, transpose=False):
        """the sparse product with every operand in HBM (device addresses, e.g. tensor.data_ptr()); stream-ordered, no host wait"""
        self.b.call("mat_spmv_sparse_dev", self.h, 1 if transpose else 0, C.c_void_p(int(d_xi)), C.c_void_p(int(d_xv)), int(nx),
                    C.c_void_p(int(d_yi)), C.c_void_p(int(d_yv)), int(cap), C.c_void_p(int(d_count)))


def dynamicsparse(I=None, J=None, V=None, m=None, n=None, fill_mode=True,
                  binding: Binding | None = None) -> DynamicSparseMatrix:
    """dynamicsparse(I, J, V, [m, n])  src/matrix.jl:15-19  /  dynamicsparse(Ti,Tj,Tv; fill_mode)  :31-41."""
    b = _bind(binding)
    h = VP()
    if I is None:
        b.call("mat_create_empty", 1 if fill_mode else 0, C.byref(h))
        return DynamicSparseMatrix(b, h)
    if not (len(I) == len(J) == len(V)):
        raise B.DsaArgumentError(B.EARG, "rows, columns, and nonzeros do not have same length.")
    i, ip = _i64(I)
    j, jp = _i64(J)
    v, vp = _f64(V)
    b.call("mat_create_from_coo", ip, jp, vp, len(i), -1 if m is None else int(m), -1 if n is None else int(n),
           C.byref(h))
    return DynamicSparseMatrix(b, h)


# free-function spellings of the exported names (src/DynamicSparseArrays.jl:5-16)
def deletecolumn(mat, col):
    mat.deletecolumn(col)
    return True


def deleterow(mat, row):
    mat.deleterow(row)
    return True


def addrow(mat, row, colids, vals):
    mat.addrow(row, colids, vals)
    return True


def closefillmode(mat):
    mat.closefillmode()
    return True


def shrink_size(vec):
    vec.shrink_size()


def nbpartitions(obj, orientation=None):
    return obj.nbpartitions() if orientation is None else obj.nbpartitions(orientation)


def deletepartition(pcsc, partition):
    pcsc.deletepartition(partition)


def nnz(obj):
    return obj.nnz()


def pool_idle_bytes(binding: Binding | None = None) -> int:
    """idle HBM the library's caching allocator holds for reuse (dsa_pool_idle_bytes)"""
    out = C.c_int64()
    _bind(binding).call("pool_idle_bytes", C.byref(out))
    return out.value


def pool_trim(keep_bytes=0, binding: Binding | None = None):
    """release idle HBM blocks until at most keep_bytes remain (dsa_pool_trim)"""
    _bind(binding).call("pool_trim", int(keep_bytes))


def dev_switches(binding: Binding | None = None):
    """(names, enabled): the library's table of development switches and whether this process honours them
    (only with DSA_DEV=1 in the environment: a release process ignores them — dsa_dev_switches)"""
    buf = C.create_string_buffer(2048)
    on = C.c_int32()
    _bind(binding).call("dev_switches", buf, 2048, C.byref(on))
    return buf.value.decode().split(), bool(on.value)
