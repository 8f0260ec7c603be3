# DynamicSparseArraysAMD.jl — Julia host module over libdsa_hip.so (include/dsa.h).
#
# Drop-in for the PMA / PCSR hot path of DynamicSparseArrays.jl under Coluna: the same exported names
# (reference src/DynamicSparseArrays.jl:5-16), every method a thin `ccall` into the C-ABI HIP shim.
# The shim fixes K = L = Int64, T = Float64 (SURVEY.md §8b); this file carries the part of the contract
# that cannot cross a C ABI:
#   * the KEY-MAPPING LAYER: containers are parametric in their key types like the reference's
#     (`DynamicSparseVector{K}`, `DynamicSparseMatrix{K,L}`); every key crosses the ABI through
#     `keyint(k)::Int64` (order-preserving: the library keeps partitions and cells in Int64 order, the
#     reference in `isless` order of K) and comes back through `keyfrom(K, i)`.  Methods exist for the
#     Integer types and for `Char` (test/functional/sparsematrix.jl:302-336); an id struct (Coluna's
#     VarId / ConstrId) needs two one-line methods, or only `keyint` — then the container remembers
#     the keys it has been given (`KeyMap.seen`) to translate results back;
#   * arbitrary `combine` functions: `+`, `*` and "last" run in the library, anything else is folded
#     here, left to right in input order, before the call (src/vector.jl:10-36, src/pcsr.jl:365-398).
# NOTE: there is no Julia toolchain in the build image, so this file has not been executed; the ABI
# itself is exercised through the Python mirror (dynamicsparsearrays.jl_amd/api.py) by tests/, and
# tests/test_library_symbols.py checks every `ccall` below against include/dsa.h.
module DynamicSparseArraysAMD

using SparseArrays

export DynamicSparseVector, DynamicSparseMatrix, DynamicMatrixColView, PackedCSC, dynamicsparsevec, dynamicsparse, nbpartitions,
       deletecolumn!, deleterow!, deletepartition!, addrow!, closefillmode!, shrink_size!, set_device!, shard_range, dynamicsparse_shard, comm_unique_id, ShardComm, shard_allreduce!,
       shard_spmv_allreduce!, set_wait_policy!, WAIT_SPIN, WAIT_BLOCK, pool_idle_bytes, pool_trim!,
       keyint, keyfrom, col_view_dev!, row_view_dev!, spmv_sparse_dev!

const libdsa = get(ENV, "DSA_HIP_LIB", joinpath(@__DIR__, "..", "csrc", "libdsa_hip.so"))

# status codes of include/dsa.h -> the exception type the reference throws at the same site
const DSA_OK = Int32(0)
function _check(rc::Int32)
    rc == DSA_OK && return
    msg = unsafe_string(ccall((:dsa_last_error_message, libdsa), Cstring, ()))
    rc == 1 && throw(ArgumentError(msg))          # DSA_EARG
    rc == 9 && throw(ArgumentError(msg))          # DSA_EKEY (0 is the semaphore key)
    rc == 2 && throw(BoundsError(msg))            # DSA_EBOUNDS
    error(msg)                                    # ErrorException: EDELETED, EFULL, EMODE, EASSERT, EHIP, ECAP
end

const COMBINE = IdDict{Function,Int32}(+ => Int32(0), * => Int32(1))
const COMBINE_LAST = Int32(2)

"one process per GPU: select the device before creating handles (dsa_set_device)"
set_device!(dev::Integer) = _check(ccall((:dsa_set_device, libdsa), Int32, (Int32,), dev))
"idle HBM the library's caching allocator holds for reuse — dsa_pool_idle_bytes"
function pool_idle_bytes()
    out = Ref{Int64}(0)
    _check(ccall((:dsa_pool_idle_bytes, libdsa), Int32, (Ptr{Int64},), out))
    return out[]
end
"release idle HBM blocks until at most `keep_bytes` remain — dsa_pool_trim"
pool_trim!(keep_bytes::Integer = 0) = _check(ccall((:dsa_pool_trim, libdsa), Int32, (Int64,), keep_bytes))
"(names, enabled): the library's development switches and whether this process honours them (only with ENV[\"DSA_DEV\"] = \"1\") — dsa_dev_switches"
function dev_switches()
    buf = Vector{UInt8}(undef, 2048); on = Ref{Int32}(0)
    _check(ccall((:dsa_dev_switches, libdsa), Int32, (Ptr{UInt8}, Int64, Ref{Int32}), buf, 2048, on))
    return split(unsafe_string(pointer(buf))), on[] != 0
end
const WAIT_SPIN = Int32(0)      # blocking calls poll a pinned word (lowest latency, one core busy)
const WAIT_BLOCK = Int32(1)     # blocking calls park in hipStreamSynchronize first (a Julia process that runs many tasks)
function device_count()
    n = Ref{Int32}(0)
    _check(ccall((:dsa_device_count, libdsa), Int32, (Ref{Int32},), n))
    return n[]
end

"column keys (col0, col0 + ncols] owned by shard `shard` (0-based) of `nshards` — dsa_shard_range"
function shard_range(n::Integer, nshards::Integer, shard::Integer)
    c0 = Ref{Int64}(0); nc = Ref{Int64}(0)
    _check(ccall((:dsa_shard_range, libdsa), Int32, (Int64, Int32, Int32, Ref{Int64}, Ref{Int64}), n, nshards, shard, c0, nc))
    return c0[], nc[]
end

# ------------------------------------------------------------------ key mapping  (SURVEY.md §8b)
"`keyint(k)::Int64`: the key as it crosses the C ABI.  Must preserve the order of `isless` on the key type; 0 is the semaphore key of matrices (src/pcsr.jl:23)."
keyint(k::Integer) = Int64(k)
keyint(k::Char) = Int64(codepoint(k))
"`keyfrom(K, i)`: inverse of `keyint`.  Optional for user types: without it a container translates through the keys it has seen."
keyfrom(::Type{K}, i::Int64) where {K<:Integer} = K(i)
keyfrom(::Type{Char}, i::Int64) = Char(i)

struct KeyMap{K}
    seen::Union{Nothing,Dict{Int64,K}}            # nothing: keyfrom(K, i) exists
end
KeyMap{K}() where {K} = KeyMap{K}(hasmethod(keyfrom, Tuple{Type{K},Int64}) ? nothing : Dict{Int64,K}())
function _in(km::KeyMap{K}, k) where {K}
    kk = convert(K, k)
    i = keyint(kk)::Int64
    km.seen === nothing || (km.seen[i] = kk)
    return i
end
_in(km::KeyMap{K}, ks::AbstractVector) where {K} = Int64[_in(km, k) for k in ks]
_out(km::KeyMap{K}, i::Int64) where {K} = km.seen === nothing ? keyfrom(K, i) : km.seen[i]
_out(km::KeyMap{K}, is::Vector{Int64}) where {K} = K[_out(km, i) for i in is]

# duplicates of a key (vector) or of a (partition, key) pair folded left to right in input order with an arbitrary `combine`;
# returns the order-preserving selection of first occurrences with the folded values (src/vector.jl:10-36)
function _prefold(keys::Vector{NTuple{N,Int64}}, V::Vector{Float64}, combine::Function) where {N}
    first_at = Dict{NTuple{N,Int64},Int}()
    keep = Int[]; vals = Float64[]
    for k in eachindex(keys)
        j = get(first_at, keys[k], 0)
        if j == 0
            push!(keep, k); push!(vals, V[k]); first_at[keys[k]] = length(keep)
        else
            vals[j] = combine(vals[j], V[k])
        end
    end
    return keep, vals
end

# ------------------------------------------------------------------ vector  (reference src/vector.jl)
mutable struct DynamicSparseVector{K} <: AbstractSparseVector{Float64,K}
    h::Ptr{Cvoid}
    keys::KeyMap{K}
    function DynamicSparseVector{K}(h::Ptr{Cvoid}, km::KeyMap{K} = KeyMap{K}()) where {K}
        v = new{K}(h, km)
        finalizer(x -> ccall((:dsa_vec_destroy, libdsa), Int32, (Ptr{Cvoid},), x.h), v)
        return v
    end
end

"how the blocking calls of `v` wait for the device: WAIT_SPIN (default) or WAIT_BLOCK — dsa_vec_set_wait_policy"
set_wait_policy!(v::DynamicSparseVector, policy::Integer) =
    _check(ccall((:dsa_vec_set_wait_policy, libdsa), Int32, (Ptr{Cvoid}, Int32), v.h, policy))

function dynamicsparsevec(I::AbstractVector{K}, V::AbstractVector, combine::Function = +, n::Integer = -1) where {K}
    length(I) == length(V) || throw(ArgumentError("keys & nonzeros vectors must have same length."))
    km = KeyMap{K}()
    Ii = _in(km, I); Vf = Vector{Float64}(V)
    op = get(COMBINE, combine, COMBINE_LAST)
    if !haskey(COMBINE, combine)       # arbitrary combine: fold duplicates here; the library then sees distinct keys
        keep, Vf = _prefold([(i,) for i in Ii], Vf, combine)
        Ii = Ii[keep]
    end
    out = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve Ii Vf _check(ccall((:dsa_vec_create, libdsa), Int32,
        (Ptr{Int64}, Ptr{Float64}, Int64, Int32, Int64, Ref{Ptr{Cvoid}}), Ii, Vf, length(Ii), op, Int64(n), out))
    return DynamicSparseVector{K}(out[], km)
end
dynamicsparsevec(I::AbstractVector, V::AbstractVector, n::Integer) = dynamicsparsevec(I, V, +, n)
function dynamicsparsevec(::Type{K}, ::Type{Float64}) where {K}
    out = Ref{Ptr{Cvoid}}(C_NULL)
    _check(ccall((:dsa_vec_create_empty, libdsa), Int32, (Ref{Ptr{Cvoid}},), out))
    return DynamicSparseVector{K}(out[])
end

function Base.getindex(v::DynamicSparseVector, key)
    out = Ref{Float64}(0.0)
    _check(ccall((:dsa_vec_get, libdsa), Int32, (Ptr{Cvoid}, Int64, Ref{Float64}), v.h, _in(v.keys, key), out))
    return out[]
end
function Base.setindex!(v::DynamicSparseVector, value, key)
    _check(ccall((:dsa_vec_set, libdsa), Int32, (Ptr{Cvoid}, Int64, Float64), v.h, _in(v.keys, key), Float64(value)))
    return v
end
"n getindex calls in one ccall"
function getindex_batch(v::DynamicSparseVector, keys::AbstractVector)
    ki = _in(v.keys, keys); out = Vector{Float64}(undef, length(ki))
    GC.@preserve ki out _check(ccall((:dsa_vec_get_batch, libdsa), Int32,
        (Ptr{Cvoid}, Ptr{Int64}, Int64, Ptr{Float64}), v.h, ki, length(ki), out))
    return out
end
"n sequential setindex! calls in one ccall (sequential-equivalent batch)"
function setindex_batch!(v::DynamicSparseVector, keys::AbstractVector, vals::AbstractVector)
    ki = _in(v.keys, keys); vf = Vector{Float64}(vals)
    GC.@preserve ki vf _check(ccall((:dsa_vec_set_batch, libdsa), Int32,
        (Ptr{Cvoid}, Ptr{Int64}, Ptr{Float64}, Int64), v.h, ki, vf, length(ki)))
    return v
end
function Base.length(v::DynamicSparseVector)
    out = Ref{Int64}(0); _check(ccall((:dsa_vec_len, libdsa), Int32, (Ptr{Cvoid}, Ref{Int64}), v.h, out)); out[]
end
Base.size(v::DynamicSparseVector) = (length(v),)
function SparseArrays.nnz(v::DynamicSparseVector)
    out = Ref{Int64}(0); _check(ccall((:dsa_vec_nnz, libdsa), Int32, (Ptr{Cvoid}, Ref{Int64}), v.h, out)); out[]
end
shrink_size!(v::DynamicSparseVector) = _check(ccall((:dsa_vec_shrink_size, libdsa), Int32, (Ptr{Cvoid},), v.h))
function _stored_int(v::DynamicSparseVector)
    n = nnz(v); ks = Vector{Int64}(undef, max(n, 1)); vs = Vector{Float64}(undef, max(n, 1)); m = Ref{Int64}(0)
    GC.@preserve ks vs _check(ccall((:dsa_vec_nonzeros, libdsa), Int32,
        (Ptr{Cvoid}, Ptr{Int64}, Ptr{Float64}, Int64, Ref{Int64}), v.h, ks, vs, length(ks), m))
    return resize!(ks, m[]), resize!(vs, m[])
end
_stored(v::DynamicSparseVector) = ((ks, vs) = _stored_int(v); (_out(v.keys, ks), vs))
SparseArrays.nonzeroinds(v::DynamicSparseVector) = _stored(v)[1]
SparseArrays.nonzeros(v::DynamicSparseVector) = _stored(v)[2]
Base.iterate(v::DynamicSparseVector, st = (zip(_stored(v)...), nothing)) =
    (r = st[2] === nothing ? iterate(st[1]) : iterate(st[1], st[2]); r === nothing ? nothing : (r[1], (st[1], r[2])))

# v1 == v2  (src/vector.jl:85-87): compared on the device, only the verdict comes back
function Base.:(==)(a::DynamicSparseVector{K}, b::DynamicSparseVector{K}) where {K}
    out = Ref{Int32}(0)
    _check(ccall((:dsa_vec_equal, libdsa), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{Int32}), a.h, b.h, out))
    return out[] == 1
end
# v1 + v2, v1 - v2: the SparseVector the AbstractSparseVector fallbacks of the reference produce (test/functional/math.jl:53-94),
# merged on the device.  Mixed operands (SparseVector with DynamicSparseVector) and -v keep using the stdlib fallbacks over
# nonzeroinds / nonzeros above.
function _axpby(a::DynamicSparseVector{K}, alpha::Float64, b::DynamicSparseVector{K}, beta::Float64) where {K<:Integer}
    length(a) == length(b) || throw(DimensionMismatch("dimensions must match"))
    cap = max(nnz(a) + nnz(b), 1); ks = Vector{Int64}(undef, cap); vs = Vector{Float64}(undef, cap); m = Ref{Int64}(0)
    GC.@preserve ks vs _check(ccall((:dsa_vec_axpby, libdsa), Int32,
        (Ptr{Cvoid}, Float64, Ptr{Cvoid}, Float64, Ptr{Int64}, Ptr{Float64}, Int64, Ref{Int64}), a.h, alpha, b.h, beta, ks, vs, cap, m))
    return SparseVector(length(a), K.(resize!(ks, m[])), resize!(vs, m[]))
end
Base.:(+)(a::DynamicSparseVector{K}, b::DynamicSparseVector{K}) where {K<:Integer} = _axpby(a, 1.0, b, 1.0)
Base.:(-)(a::DynamicSparseVector{K}, b::DynamicSparseVector{K}) where {K<:Integer} = _axpby(a, 1.0, b, -1.0)
# filter(f, v)  (src/vector.jl:83 -> src/pma.jl:224-234): the predicate runs in Julia on the packed entries, the result is a new vector
function Base.filter(f, v::DynamicSparseVector{K}) where {K}
    ks, vs = _stored(v)
    keep = [f((ks[i], vs[i])) for i in eachindex(ks)]
    return dynamicsparsevec(ks[keep], vs[keep])
end

# ------------------------------------------------------------------ matrix  (reference src/matrix.jl)
mutable struct DynamicSparseMatrix{K,L}
    h::Ptr{Cvoid}
    rows::KeyMap{K}
    cols::KeyMap{L}
    function DynamicSparseMatrix{K,L}(h::Ptr{Cvoid}, rk::KeyMap{K} = KeyMap{K}(), ck::KeyMap{L} = KeyMap{L}()) where {K,L}
        m = new{K,L}(h, rk, ck)
        finalizer(x -> ccall((:dsa_mat_destroy, libdsa), Int32, (Ptr{Cvoid},), x.h), m)
        return m
    end
end

"how the blocking calls of `m` wait for the device: WAIT_SPIN (default) or WAIT_BLOCK — dsa_mat_set_wait_policy"
set_wait_policy!(m::DynamicSparseMatrix, policy::Integer) =
    _check(ccall((:dsa_mat_set_wait_policy, libdsa), Int32, (Ptr{Cvoid}, Int32), m.h, policy))

# dynamicsparse(I, J, V[, m, n][, combine])  src/matrix.jl:15-19 -> dynamicsparsecolmajor(J, I, V, combine) src/pcsr.jl:433-445:
# the library folds duplicates of (i, j) with +; any other combine is folded here first
function dynamicsparse(I::AbstractVector{K}, J::AbstractVector{L}, V::AbstractVector, m = -1, n = -1, combine::Function = +) where {K,L}
    length(I) == length(J) == length(V) ||
        throw(ArgumentError("rows, columns, and nonzeros do not have same length."))
    rk = KeyMap{K}(); ck = KeyMap{L}()
    Ii = _in(rk, I); Ji = _in(ck, J); Vf = Vector{Float64}(V)
    if combine !== +
        keep, Vf = _prefold([(Ji[k], Ii[k]) for k in eachindex(Ii)], Vf, combine)
        Ii = Ii[keep]; Ji = Ji[keep]
    end
    mi = m isa Integer && m < 0 ? Int64(-1) : _in(rk, m); ni = n isa Integer && n < 0 ? Int64(-1) : _in(ck, n)
    out = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve Ii Ji Vf _check(ccall((:dsa_mat_create_from_coo, libdsa), Int32,
        (Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Int64, Int64, Int64, Ref{Ptr{Cvoid}}), Ii, Ji, Vf, length(Ii), mi, ni, out))
    return DynamicSparseMatrix{K,L}(out[], rk, ck)
end
dynamicsparse(I::AbstractVector, J::AbstractVector, V::AbstractVector, combine::Function) = dynamicsparse(I, J, V, -1, -1, combine)
function dynamicsparse(::Type{K}, ::Type{L}, ::Type{Float64}; fill_mode = true) where {K,L}
    out = Ref{Ptr{Cvoid}}(C_NULL)
    _check(ccall((:dsa_mat_create_empty, libdsa), Int32, (Int32, Ref{Ptr{Cvoid}}), fill_mode ? 1 : 0, out))
    return DynamicSparseMatrix{K,L}(out[])
end

function Base.setindex!(a::DynamicSparseMatrix, val, row, col)
    _check(ccall((:dsa_mat_set, libdsa), Int32, (Ptr{Cvoid}, Float64, Int64, Int64), a.h, Float64(val), _in(a.rows, row), _in(a.cols, col)))
    return a
end
function setindex_batch!(a::DynamicSparseMatrix, I::AbstractVector, J::AbstractVector, V::AbstractVector)
    Ii = _in(a.rows, I); Ji = _in(a.cols, J); Vf = Vector{Float64}(V)
    GC.@preserve Ii Ji Vf _check(ccall((:dsa_mat_set_batch, libdsa), Int32,
        (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Int64), a.h, Ii, Ji, Vf, length(Ii)))
    return a
end
function Base.getindex(a::DynamicSparseMatrix, row, col)
    out = Ref{Float64}(0.0)
    _check(ccall((:dsa_mat_get, libdsa), Int32, (Ptr{Cvoid}, Int64, Int64, Ref{Float64}), a.h, _in(a.rows, row), _in(a.cols, col), out))
    return out[]
end
function addrow!(a::DynamicSparseMatrix, row, colids::AbstractVector, vals::AbstractVector)
    ci = _in(a.cols, colids); vf = Vector{Float64}(vals)
    GC.@preserve ci vf _check(ccall((:dsa_mat_addrow, libdsa), Int32,
        (Ptr{Cvoid}, Int64, Ptr{Int64}, Ptr{Float64}, Int64), a.h, _in(a.rows, row), ci, vf, length(ci)))
    return true
end
closefillmode!(a::DynamicSparseMatrix) = (_check(ccall((:dsa_mat_closefillmode, libdsa), Int32, (Ptr{Cvoid},), a.h)); true)
deletecolumn!(a::DynamicSparseMatrix, col) = (_check(ccall((:dsa_mat_deletecolumn, libdsa), Int32, (Ptr{Cvoid}, Int64), a.h, _in(a.cols, col))); true)
deleterow!(a::DynamicSparseMatrix, row) = (_check(ccall((:dsa_mat_deleterow, libdsa), Int32, (Ptr{Cvoid}, Int64), a.h, _in(a.rows, row))); true)
function SparseArrays.nnz(a::DynamicSparseMatrix)
    out = Ref{Int64}(0); _check(ccall((:dsa_mat_nnz, libdsa), Int32, (Ptr{Cvoid}, Ref{Int64}), a.h, out)); out[]
end
# size(A) is the running maximum of the keys written with a non-zero value, as keys (src/matrix.jl:44-47; `size == (5, 'e')`
# at test/functional/sparsematrix.jl:302-336)
function Base.size(a::DynamicSparseMatrix)
    m = Ref{Int64}(0); n = Ref{Int64}(0)
    _check(ccall((:dsa_mat_size, libdsa), Int32, (Ptr{Cvoid}, Ref{Int64}, Ref{Int64}), a.h, m, n))
    return (_out(a.rows, m[]), _out(a.cols, n[]))
end
Base.size(a::DynamicSparseMatrix, i) = size(a)[i]
function _size_int(a::DynamicSparseMatrix)
    m = Ref{Int64}(0); n = Ref{Int64}(0)
    _check(ccall((:dsa_mat_size, libdsa), Int32, (Ptr{Cvoid}, Ref{Int64}, Ref{Int64}), a.h, m, n)); (m[], n[])
end
function nbpartitions(a::DynamicSparseMatrix, orientation::Integer)     # 0 = colmajor, 1 = rowmajor
    out = Ref{Int64}(0); _check(ccall((:dsa_mat_nbpartitions, libdsa), Int32, (Ptr{Cvoid}, Int32, Ref{Int64}), a.h, orientation, out)); out[]
end

# per-column iteration: DynamicMatrixColView (src/views.jl:3-36).  The reference's view is a lazy cursor over the slot array;
# here the occupied cells of the column's slot range are packed on the device in slot order (K-pack, one launch) and the
# view iterates over that snapshot: `for (row, val) in @view A[:, j]`.
struct DynamicMatrixColView{K,L}
    col_key::L
    rows::Vector{K}
    vals::Vector{Float64}
end
Base.iterate(dv::DynamicMatrixColView, i::Int = 1) = i > length(dv.rows) ? nothing : ((dv.rows[i], dv.vals[i]), i + 1)
Base.length(dv::DynamicMatrixColView) = length(dv.rows)
Base.eltype(::Type{DynamicMatrixColView{K,L}}) where {K,L} = Tuple{K,Float64}

# ccall needs a literal (symbol, library) pair: the four entry points are stamped out with @eval
for (fname, sym) in ((:_col_view, :dsa_mat_col_view), (:_row_view, :dsa_mat_row_view))
    @eval function $fname(a::DynamicSparseMatrix, key::Int64)
        cap = 64
        while true
            ks = Vector{Int64}(undef, cap); vs = Vector{Float64}(undef, cap); n = Ref{Int64}(0)
            rc = GC.@preserve ks vs ccall(($(QuoteNode(sym)), libdsa), Int32,
                (Ptr{Cvoid}, Int64, Ptr{Int64}, Ptr{Float64}, Int64, Ref{Int64}), a.h, key, ks, vs, cap, n)
            rc == 8 && (cap *= 8; continue)          # DSA_ECAP
            _check(rc)
            return resize!(ks, n[]), resize!(vs, n[])
        end
    end
end
function Base.view(a::DynamicSparseMatrix{K,L}, ::Colon, col) where {K,L}                     # src/matrix.jl:83-88
    ks, vs = _col_view(a, _in(a.cols, col))
    return DynamicMatrixColView{K,L}(convert(L, col), _out(a.rows, ks), vs)
end
function Base.view(a::DynamicSparseMatrix{K,L}, row, ::Colon) where {K,L}                     # src/matrix.jl:70-81 (fill mode: buffer row)
    ks, vs = _row_view(a, _in(a.rows, row))
    return collect(zip(_out(a.cols, ks), vs))
end

for (fname, sym) in ((:_col_slice, :dsa_mat_col_slice), (:_row_slice, :dsa_mat_row_slice))
    @eval function $fname(a::DynamicSparseMatrix, key::Int64)
        out = Ref{Ptr{Cvoid}}(C_NULL)
        _check(ccall(($(QuoteNode(sym)), libdsa), Int32, (Ptr{Cvoid}, Int64, Ref{Ptr{Cvoid}}), a.h, key, out))
        return out[]
    end
end
"`@view A[:, col]` delivered into HBM: `d_rows` / `d_vals` are device pointers to `cap` Int64 / Float64 entries (mapped key integers, see
keyint); returns the number of cells; the copy is enqueued on the colmajor stream — dsa_mat_col_view_dev (the row form: dsa_mat_row_view_dev)"
function col_view_dev!(a::DynamicSparseMatrix, col, d_rows::Ptr{Cvoid}, d_vals::Ptr{Cvoid}, cap::Integer)
    n = Ref{Int64}(0)
    _check(ccall((:dsa_mat_col_view_dev, libdsa), Int32, (Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ref{Int64}), a.h, _in(a.cols, col), d_rows, d_vals, cap, n))
    return n[]
end
function row_view_dev!(a::DynamicSparseMatrix, row, d_cols::Ptr{Cvoid}, d_vals::Ptr{Cvoid}, cap::Integer)
    n = Ref{Int64}(0)
    _check(ccall((:dsa_mat_row_view_dev, libdsa), Int32, (Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ref{Int64}), a.h, _in(a.rows, row), d_cols, d_vals, cap, n))
    return n[]
end
# A[:, col] / A[row, :]: the new vector is built device to device (view kernel -> one spread launch); nothing but its handle comes back
Base.getindex(a::DynamicSparseMatrix{K,L}, ::Colon, col) where {K,L} = DynamicSparseVector{K}(_col_slice(a, _in(a.cols, col)), a.rows)
Base.getindex(a::DynamicSparseMatrix{K,L}, row, ::Colon) where {K,L} = DynamicSparseVector{L}(_row_slice(a, _in(a.rows, row)), a.cols)
"n getindex calls in one ccall"
function getindex_batch(a::DynamicSparseMatrix, I::AbstractVector, J::AbstractVector)
    Ii = _in(a.rows, I); Ji = _in(a.cols, J); out = Vector{Float64}(undef, length(Ii))
    GC.@preserve Ii Ji out _check(ccall((:dsa_mat_get_batch, libdsa), Int32,
        (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}, Int64, Ptr{Float64}), a.h, Ii, Ji, length(Ii), out))
    return out
end

"the column-range shard `shard` of `nshards` of the matrix given by its triples (local column keys 1..ncols) — one process per GPU"
function dynamicsparse_shard(I::Vector{Int64}, J::Vector{Int64}, V::Vector{Float64}, m::Int64, n::Int64, nshards::Integer, shard::Integer)
    out = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve I J V _check(ccall((:dsa_shard_create_from_coo, libdsa), Int32,
        (Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Int64, Int64, Int64, Int32, Int32, Ref{Ptr{Cvoid}}), I, J, V, length(I), m, n, nshards, shard, out))
    return DynamicSparseMatrix{Int64,Int64}(out[])
end

# ---- the collective of the column-range sharded product: RCCL behind the C ABI (include/dsa.h: dsa_comm_*), one process per GPU.
# Rank 0 calls comm_unique_id() and hands the 128 bytes to the other ranks (MPI.Bcast!, a file, ...); every rank then calls
# ShardComm(rank, nranks, id) — a collective — and shard_spmv_allreduce!(shard, comm, x, y) leaves y = A x on every rank.
"128 opaque bytes identifying a new communicator (ncclUniqueId) — dsa_comm_unique_id"
function comm_unique_id()
    id = Vector{UInt8}(undef, 128)
    GC.@preserve id _check(ccall((:dsa_comm_unique_id, libdsa), Int32, (Ptr{UInt8},), id))
    return id
end
mutable struct ShardComm
    h::Ptr{Cvoid}
    function ShardComm(rank::Integer, nranks::Integer, id::Union{Nothing,Vector{UInt8}})
        out = Ref{Ptr{Cvoid}}(C_NULL)
        idp = id === nothing ? Ptr{UInt8}(C_NULL) : pointer(id)
        GC.@preserve id _check(ccall((:dsa_comm_init, libdsa), Int32, (Int32, Int32, Ptr{UInt8}, Ref{Ptr{Cvoid}}), rank, nranks, idp, out))
        c = new(out[])
        finalizer(x -> ccall((:dsa_comm_destroy, libdsa), Int32, (Ptr{Cvoid},), x.h), c)
        return c
    end
end
"y (a device pointer to m Float64 in HBM) <- sum over the ranks, in place, asynchronous on `stream` — dsa_shard_allreduce_dev.
`stream` has no default: the shard's product runs on the matrix's own non-blocking stream, which is NOT ordered against the legacy
null stream, so the caller names the stream y was produced on (or uses shard_spmv_allreduce!, which takes the matrix's)."
shard_allreduce!(c::ShardComm, d_y::Ptr{Cvoid}, m::Integer, stream::Ptr{Cvoid}) =
    _check(ccall((:dsa_shard_allreduce_dev, libdsa), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}), c.h, d_y, m, stream))
"y = A x of the whole sharded matrix on every rank: local product + all-reduce, x / y device pointers — dsa_shard_spmv_allreduce_dev"
shard_spmv_allreduce!(a::DynamicSparseMatrix, c::ShardComm, d_x::Ptr{Cvoid}, nx::Integer, d_y::Ptr{Cvoid}, ny::Integer) =
    _check(ccall((:dsa_shard_spmv_allreduce_dev, libdsa), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Int64), a.h, c.h, d_x, nx, d_y, ny))

# ------------------------------------------------------------------ PackedCSC  (reference src/pcsr.jl:4-339)
mutable struct PackedCSC{K}
    h::Ptr{Cvoid}
    keys::KeyMap{K}
    function PackedCSC{K}(h::Ptr{Cvoid}, km::KeyMap{K} = KeyMap{K}()) where {K}
        p = new{K}(h, km)
        finalizer(x -> ccall((:dsa_pcsc_destroy, libdsa), Int32, (Ptr{Cvoid},), x.h), p)
        return p
    end
end
"PackedCSC(row_keys, values, combine): one vector of keys / values per partition (src/pcsr.jl:26-63)"
function PackedCSC(row_keys::Vector{Vector{K}}, values::Vector{<:AbstractVector}, combine::Function = +) where {K}
    length(row_keys) == length(values) || throw(ArgumentError("Must have same number of partitions."))
    km = KeyMap{K}()
    colptr = Int64[0]; keys = Int64[]; vals = Float64[]          # CSC-style offsets (0-based, nparts + 1 entries)
    for (p, (ks, vs)) in enumerate(zip(row_keys, values))
        length(ks) == length(vs) || throw(ArgumentError("Partition $p: keys & values must have same length."))
        ki = _in(km, ks); vf = Vector{Float64}(vs)
        if !haskey(COMBINE, combine)      # arbitrary combine: folded per partition here, the library then sees distinct keys
            keep, vf = _prefold([(i,) for i in ki], vf, combine)
            ki = ki[keep]
        end
        append!(keys, ki); append!(vals, vf); push!(colptr, length(keys))
    end
    out = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve colptr keys vals _check(ccall((:dsa_pcsc_create, libdsa), Int32,
        (Ptr{Int64}, Int64, Ptr{Int64}, Ptr{Float64}, Int32, Ref{Ptr{Cvoid}}),
        colptr, length(row_keys), keys, vals, get(COMBINE, combine, COMBINE_LAST), out))
    return PackedCSC{K}(out[], km)
end
function PackedCSC(::Type{K}, ::Type{Float64}) where {K}                    # src/pcsr.jl:65-68
    out = Ref{Ptr{Cvoid}}(C_NULL)
    _check(ccall((:dsa_pcsc_create_empty, libdsa), Int32, (Ref{Ptr{Cvoid}},), out))
    return PackedCSC{K}(out[])
end
function Base.getindex(p::PackedCSC, key, partition::Integer)              # src/pcsr.jl:228-232
    out = Ref{Float64}(0.0)
    _check(ccall((:dsa_pcsc_get, libdsa), Int32, (Ptr{Cvoid}, Int64, Int64, Ref{Float64}), p.h, _in(p.keys, key), Int64(partition), out))
    return out[]
end
function Base.setindex!(p::PackedCSC, value, key, partition::Integer)      # src/pcsr.jl:294-310
    _check(ccall((:dsa_pcsc_set, libdsa), Int32, (Ptr{Cvoid}, Float64, Int64, Int64), p.h, Float64(value), _in(p.keys, key), Int64(partition)))
    return p
end
deletepartition!(p::PackedCSC, partition::Integer) =                          # src/pcsr.jl:188-204
    (_check(ccall((:dsa_pcsc_deletepartition, libdsa), Int32, (Ptr{Cvoid}, Int64), p.h, Int64(partition))); nothing)
function SparseArrays.nnz(p::PackedCSC)
    out = Ref{Int64}(0); _check(ccall((:dsa_pcsc_nnz, libdsa), Int32, (Ptr{Cvoid}, Ref{Int64}), p.h, out)); out[]
end
function nbpartitions(p::PackedCSC)
    out = Ref{Int64}(0); _check(ccall((:dsa_pcsc_nbpartitions, libdsa), Int32, (Ptr{Cvoid}, Ref{Int64}), p.h, out)); out[]
end

# ------------------------------------------------------------------ SpMV  (reference src/operations.jl)
struct Transposed{T}; array::T; end
Base.transpose(a::DynamicSparseMatrix) = Transposed(a)
Base.size(t::Transposed) = reverse(size(t.array))

# y = A x for a sparse x given by its stored entries: the touched rows in ascending order (_mul_output, src/operations.jl:11-12);
# Integer output keys -> sparsevec, any other key type -> Dict (the reference returns its accumulator Dict there)
function _spmv_sparse(a::DynamicSparseMatrix, tr::Bool, xi::Vector{Int64}, xv::Vector{Float64}, out_keys::KeyMap{KO}, n::Int64) where {KO}
    # two ccalls: the product (result left with the handle), then the copy-out into vectors of exactly the result's size
    k = Ref{Int64}(0)
    _check(GC.@preserve xi xv ccall((:dsa_mat_spmv_sparse_begin, libdsa), Int32,
        (Ptr{Cvoid}, Int32, Ptr{Int64}, Ptr{Float64}, Int64, Ref{Int64}), a.h, tr ? 1 : 0, xi, xv, length(xi), k))
    yi = Vector{Int64}(undef, k[]); yv = Vector{Float64}(undef, k[])
    if k[] > 0
        _check(GC.@preserve yi yv ccall((:dsa_mat_spmv_sparse_fetch, libdsa), Int32,
            (Ptr{Cvoid}, Ptr{Int64}, Ptr{Float64}, Int64, Ref{Int64}), a.h, yi, yv, k[], k))
    end
    return KO <: Integer ? sparsevec(KO.(yi), yv, n) : Dict{KO,Float64}(_out(out_keys, yi[j]) => yv[j] for j in eachindex(yi))
end
"the sparse product with every operand in HBM (device pointers; mapped key integers): xi / xv in, the touched rows yi / yv and their count
out, stream-ordered, no host wait — dsa_mat_spmv_sparse_dev"
function spmv_sparse_dev!(a::DynamicSparseMatrix, tr::Bool, d_xi::Ptr{Cvoid}, d_xv::Ptr{Cvoid}, nx::Integer, d_yi::Ptr{Cvoid}, d_yv::Ptr{Cvoid},
                          cap::Integer, d_count::Ptr{Cvoid})
    _check(ccall((:dsa_mat_spmv_sparse_dev, libdsa), Int32, (Ptr{Cvoid}, Int32, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}),
                 a.h, tr ? 1 : 0, d_xi, d_xv, nx, d_yi, d_yv, cap, d_count))
    return a
end
_entries(km::KeyMap, v::DynamicSparseVector) = _stored_int(v)
_entries(km::KeyMap, v::SparseVector) = (_in(km, rowvals(v)), Vector{Float64}(nonzeros(v)))
const SpVec = Union{DynamicSparseVector,SparseVector}
Base.:(*)(a::DynamicSparseMatrix, v::SpVec) = _spmv_sparse(a, false, _entries(a.cols, v)..., a.rows, _size_int(a)[1])          # src/operations.jl:14-24
Base.:(*)(t::Transposed{<:DynamicSparseMatrix}, v::SpVec) = _spmv_sparse(t.array, true, _entries(t.array.rows, v)..., t.array.cols, _size_int(t.array)[2])   # :26-36
Base.:(*)(v::SpVec, t::Transposed{<:DynamicSparseMatrix}) = t.array * v                                                            # :38-48
Base.:(*)(v::SpVec, a::DynamicSparseMatrix) = transpose(a) * v                                                                     # :50-60
"dense x, dense y — the column-generation pricing product on device-resident data (Integer keys 1..n)"
function Base.:(*)(a::DynamicSparseMatrix, x::Vector{Float64})
    y = Vector{Float64}(undef, _size_int(a)[1])
    GC.@preserve x y _check(ccall((:dsa_mat_spmv_dense, libdsa), Int32,
        (Ptr{Cvoid}, Int32, Ptr{Float64}, Int64, Ptr{Float64}, Int64), a.h, 0, x, length(x), y, length(y)))
    return y
end

end # module
