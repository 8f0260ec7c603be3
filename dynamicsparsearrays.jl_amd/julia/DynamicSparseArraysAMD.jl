# DynamicSparseArraysAMD.jl — Julia host module over libdsa_hip.so (include/dsa.h).
#
# Drop-in for the PMA / PCSR hot path of DynamicSparseArrays.jl under Coluna: the same exported names
# (reference src/DynamicSparseArrays.jl:5-16) with K = L = Int64, T = Float64, every method a thin
# `ccall` into the C-ABI HIP shim.  NOTE: there is no Julia toolchain in the build image, so this file
# has not been executed; the ABI itself is exercised through the Python mirror
# (dynamicsparsearrays.jl_amd/api.py) by tests/.
module DynamicSparseArraysAMD

using SparseArrays

export DynamicSparseVector, DynamicSparseMatrix, PackedCSC, dynamicsparsevec, dynamicsparse, nbpartitions,
       deletecolumn!, deleterow!, deletepartition!, addrow!, closefillmode!, shrink_size!, set_device!, shard_range, dynamicsparse_shard

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

"one process per GPU: select the device before creating handles (dsa_set_device)"
set_device!(dev::Integer) = _check(ccall((:dsa_set_device, libdsa), Int32, (Int32,), dev))
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

# ------------------------------------------------------------------ vector  (reference src/vector.jl)
mutable struct DynamicSparseVector <: AbstractSparseVector{Float64,Int64}
    h::Ptr{Cvoid}
    function DynamicSparseVector(h::Ptr{Cvoid})
        v = new(h)
        finalizer(x -> ccall((:dsa_vec_destroy, libdsa), Int32, (Ptr{Cvoid},), x.h), v)
        return v
    end
end

function dynamicsparsevec(I::Vector{Int64}, V::Vector{Float64}, combine::Function = +, n::Int64 = -1)
    length(I) == length(V) || throw(ArgumentError("keys & nonzeros vectors must have same length."))
    if !haskey(COMBINE, combine)       # arbitrary combine: fold duplicates on the Julia side (src/vector.jl:10-36)
        p = sortperm(I); I = I[p]; V = V[p]
        keep = Int[]; 
        for k in eachindex(I)
            if !isempty(keep) && I[keep[end]] == I[k]
                V[keep[end]] = combine(V[keep[end]], V[k])
            else
                push!(keep, k)
            end
        end
        I = I[keep]; V = V[keep]; op = Int32(2)
    else
        op = COMBINE[combine]
    end
    out = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve I V _check(ccall((:dsa_vec_create, libdsa), Int32,
        (Ptr{Int64}, Ptr{Float64}, Int64, Int32, Int64, Ref{Ptr{Cvoid}}), I, V, length(I), op, n, out))
    return DynamicSparseVector(out[])
end
dynamicsparsevec(I::Vector{Int64}, V::Vector{Float64}, n::Int64) = dynamicsparsevec(I, V, +, n)

function Base.getindex(v::DynamicSparseVector, key::Integer)
    out = Ref{Float64}(0.0)
    _check(ccall((:dsa_vec_get, libdsa), Int32, (Ptr{Cvoid}, Int64, Ref{Float64}), v.h, key, out))
    return out[]
end
function Base.setindex!(v::DynamicSparseVector, value, key::Integer)
    _check(ccall((:dsa_vec_set, libdsa), Int32, (Ptr{Cvoid}, Int64, Float64), v.h, key, Float64(value)))
    return v
end
dynamicsparsevec(::Type{Int64}, ::Type{Float64}) = (out = Ref{Ptr{Cvoid}}(C_NULL);
    _check(ccall((:dsa_vec_create_empty, libdsa), Int32, (Ref{Ptr{Cvoid}},), out)); DynamicSparseVector(out[]))
"n getindex calls in one ccall"
function getindex_batch(v::DynamicSparseVector, keys::Vector{Int64})
    out = Vector{Float64}(undef, length(keys))
    GC.@preserve keys out _check(ccall((:dsa_vec_get_batch, libdsa), Int32,
        (Ptr{Cvoid}, Ptr{Int64}, Int64, Ptr{Float64}), v.h, keys, length(keys), out))
    return out
end
"n sequential setindex! calls in one ccall (sequential-equivalent batch)"
function setindex_batch!(v::DynamicSparseVector, keys::Vector{Int64}, vals::Vector{Float64})
    GC.@preserve keys vals _check(ccall((:dsa_vec_set_batch, libdsa), Int32,
        (Ptr{Cvoid}, Ptr{Int64}, Ptr{Float64}, Int64), v.h, keys, vals, length(keys)))
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
function _stored(v::DynamicSparseVector)
    n = nnz(v); ks = Vector{Int64}(undef, max(n, 1)); vs = Vector{Float64}(undef, max(n, 1)); m = Ref{Int64}(0)
    GC.@preserve ks vs _check(ccall((:dsa_vec_nonzeros, libdsa), Int32,
        (Ptr{Cvoid}, Ptr{Int64}, Ptr{Float64}, Int64, Ref{Int64}), v.h, ks, vs, length(ks), m))
    return resize!(ks, m[]), resize!(vs, m[])
end
SparseArrays.nonzeroinds(v::DynamicSparseVector) = _stored(v)[1]
SparseArrays.nonzeros(v::DynamicSparseVector) = _stored(v)[2]
Base.iterate(v::DynamicSparseVector, st = (zip(_stored(v)...), nothing)) =
    (r = st[2] === nothing ? iterate(st[1]) : iterate(st[1], st[2]); r === nothing ? nothing : (r[1], (st[1], r[2])))

# v1 == v2  (src/vector.jl:85-87): compared on the device, only the verdict comes back
function Base.:(==)(a::DynamicSparseVector, b::DynamicSparseVector)
    out = Ref{Int32}(0)
    _check(ccall((:dsa_vec_equal, libdsa), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{Int32}), a.h, b.h, out))
    return out[] == 1
end
# v1 + v2, v1 - v2: the SparseVector the AbstractSparseVector fallbacks of the reference produce (test/functional/math.jl:53-94),
# merged on the device.  Mixed operands (SparseVector with DynamicSparseVector) and -v keep using the stdlib fallbacks over
# nonzeroinds / nonzeros above.
function _axpby(a::DynamicSparseVector, alpha::Float64, b::DynamicSparseVector, beta::Float64)
    length(a) == length(b) || throw(DimensionMismatch("dimensions must match"))
    cap = max(nnz(a) + nnz(b), 1); ks = Vector{Int64}(undef, cap); vs = Vector{Float64}(undef, cap); m = Ref{Int64}(0)
    GC.@preserve ks vs _check(ccall((:dsa_vec_axpby, libdsa), Int32,
        (Ptr{Cvoid}, Float64, Ptr{Cvoid}, Float64, Ptr{Int64}, Ptr{Float64}, Int64, Ref{Int64}), a.h, alpha, b.h, beta, ks, vs, cap, m))
    return SparseVector(length(a), resize!(ks, m[]), resize!(vs, m[]))
end
Base.:(+)(a::DynamicSparseVector, b::DynamicSparseVector) = _axpby(a, 1.0, b, 1.0)
Base.:(-)(a::DynamicSparseVector, b::DynamicSparseVector) = _axpby(a, 1.0, b, -1.0)
# filter(f, v)  (src/vector.jl:83 -> src/pma.jl:224-234): the predicate runs in Julia on the packed entries, the result is a new vector
function Base.filter(f, v::DynamicSparseVector)
    ks, vs = _stored(v)
    keep = [f((ks[i], vs[i])) for i in eachindex(ks)]
    return dynamicsparsevec(ks[keep], vs[keep])
end

# ------------------------------------------------------------------ matrix  (reference src/matrix.jl)
mutable struct DynamicSparseMatrix
    h::Ptr{Cvoid}
    function DynamicSparseMatrix(h::Ptr{Cvoid})
        m = new(h)
        finalizer(x -> ccall((:dsa_mat_destroy, libdsa), Int32, (Ptr{Cvoid},), x.h), m)
        return m
    end
end

function dynamicsparse(I::Vector{Int64}, J::Vector{Int64}, V::Vector{Float64}, m::Int64 = -1, n::Int64 = -1)
    length(I) == length(J) == length(V) ||
        throw(ArgumentError("rows, columns, and nonzeros do not have same length."))
    out = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve I J V _check(ccall((:dsa_mat_create_from_coo, libdsa), Int32,
        (Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Int64, Int64, Int64, Ref{Ptr{Cvoid}}), I, J, V, length(I), m, n, out))
    return DynamicSparseMatrix(out[])
end
function dynamicsparse(::Type{Int64}, ::Type{Int64}, ::Type{Float64}; fill_mode = true)
    out = Ref{Ptr{Cvoid}}(C_NULL)
    _check(ccall((:dsa_mat_create_empty, libdsa), Int32, (Int32, Ref{Ptr{Cvoid}}), fill_mode ? 1 : 0, out))
    return DynamicSparseMatrix(out[])
end

function Base.setindex!(a::DynamicSparseMatrix, val, row::Int64, col::Int64)
    _check(ccall((:dsa_mat_set, libdsa), Int32, (Ptr{Cvoid}, Float64, Int64, Int64), a.h, Float64(val), row, col))
    return a
end
function setindex_batch!(a::DynamicSparseMatrix, I::Vector{Int64}, J::Vector{Int64}, V::Vector{Float64})
    GC.@preserve I J V _check(ccall((:dsa_mat_set_batch, libdsa), Int32,
        (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Int64), a.h, I, J, V, length(I)))
    return a
end
function Base.getindex(a::DynamicSparseMatrix, row::Int64, col::Int64)
    out = Ref{Float64}(0.0)
    _check(ccall((:dsa_mat_get, libdsa), Int32, (Ptr{Cvoid}, Int64, Int64, Ref{Float64}), a.h, row, col, out))
    return out[]
end
function addrow!(a::DynamicSparseMatrix, row::Int64, colids::Vector{Int64}, vals::Vector{Float64})
    GC.@preserve colids vals _check(ccall((:dsa_mat_addrow, libdsa), Int32,
        (Ptr{Cvoid}, Int64, Ptr{Int64}, Ptr{Float64}, Int64), a.h, row, colids, vals, length(colids)))
    return true
end
closefillmode!(a::DynamicSparseMatrix) = (_check(ccall((:dsa_mat_closefillmode, libdsa), Int32, (Ptr{Cvoid},), a.h)); true)
deletecolumn!(a::DynamicSparseMatrix, col::Int64) = (_check(ccall((:dsa_mat_deletecolumn, libdsa), Int32, (Ptr{Cvoid}, Int64), a.h, col)); true)
deleterow!(a::DynamicSparseMatrix, row::Int64) = (_check(ccall((:dsa_mat_deleterow, libdsa), Int32, (Ptr{Cvoid}, Int64), a.h, row)); true)
function SparseArrays.nnz(a::DynamicSparseMatrix)
    out = Ref{Int64}(0); _check(ccall((:dsa_mat_nnz, libdsa), Int32, (Ptr{Cvoid}, Ref{Int64}), a.h, out)); out[]
end
function Base.size(a::DynamicSparseMatrix)
    m = Ref{Int64}(0); n = Ref{Int64}(0)
    _check(ccall((:dsa_mat_size, libdsa), Int32, (Ptr{Cvoid}, Ref{Int64}, Ref{Int64}), a.h, m, n)); (m[], n[])
end
Base.size(a::DynamicSparseMatrix, i) = size(a)[i]
function nbpartitions(a::DynamicSparseMatrix, orientation::Integer)     # 0 = colmajor, 1 = rowmajor
    out = Ref{Int64}(0); _check(ccall((:dsa_mat_nbpartitions, libdsa), Int32, (Ptr{Cvoid}, Int32, Ref{Int64}), a.h, orientation, out)); out[]
end

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
            return collect(zip(resize!(ks, n[]), resize!(vs, n[])))
        end
    end
end
Base.view(a::DynamicSparseMatrix, ::Colon, col::Int64) = _col_view(a, col)   # src/matrix.jl:83-88
Base.view(a::DynamicSparseMatrix, row::Int64, ::Colon) = _row_view(a, row)   # src/matrix.jl:70-81

for (fname, sym) in ((:_col_slice, :dsa_mat_col_slice), (:_row_slice, :dsa_mat_row_slice))
    @eval function $fname(a::DynamicSparseMatrix, key::Int64)
        out = Ref{Ptr{Cvoid}}(C_NULL)
        _check(ccall(($(QuoteNode(sym)), libdsa), Int32, (Ptr{Cvoid}, Int64, Ref{Ptr{Cvoid}}), a.h, key, out))
        return DynamicSparseVector(out[])
    end
end
Base.getindex(a::DynamicSparseMatrix, ::Colon, col::Int64) = _col_slice(a, col)
Base.getindex(a::DynamicSparseMatrix, row::Int64, ::Colon) = _row_slice(a, row)
"n getindex calls in one ccall"
function getindex_batch(a::DynamicSparseMatrix, I::Vector{Int64}, J::Vector{Int64})
    out = Vector{Float64}(undef, length(I))
    GC.@preserve I J out _check(ccall((:dsa_mat_get_batch, libdsa), Int32,
        (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}, Int64, Ptr{Float64}), a.h, I, J, length(I), out))
    return out
end

"the column-range shard `shard` of `nshards` of the matrix given by its triples (local column keys 1..ncols) — one process per GPU"
function dynamicsparse_shard(I::Vector{Int64}, J::Vector{Int64}, V::Vector{Float64}, m::Int64, n::Int64, nshards::Integer, shard::Integer)
    out = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve I J V _check(ccall((:dsa_shard_create_from_coo, libdsa), Int32,
        (Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Int64, Int64, Int64, Int32, Int32, Ref{Ptr{Cvoid}}), I, J, V, length(I), m, n, nshards, shard, out))
    return DynamicSparseMatrix(out[])
end

# ------------------------------------------------------------------ PackedCSC  (reference src/pcsr.jl:4-339)
mutable struct PackedCSC
    h::Ptr{Cvoid}
    function PackedCSC(h::Ptr{Cvoid})
        p = new(h)
        finalizer(x -> ccall((:dsa_pcsc_destroy, libdsa), Int32, (Ptr{Cvoid},), x.h), p)
        return p
    end
end
"PackedCSC(row_keys, values, combine): one vector of keys / values per partition (src/pcsr.jl:26-63)"
function PackedCSC(row_keys::Vector{Vector{Int64}}, values::Vector{Vector{Float64}}, combine::Function = +)
    length(row_keys) == length(values) || throw(ArgumentError("Must have same number of partitions."))
    colptr = Int64[0]; keys = Int64[]; vals = Float64[]          # CSC-style offsets (0-based, nparts + 1 entries)
    for (p, (ks, vs)) in enumerate(zip(row_keys, values))
        length(ks) == length(vs) || throw(ArgumentError("Partition $p: keys & values must have same length."))
        append!(keys, ks); append!(vals, vs); push!(colptr, length(keys))
    end
    out = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve colptr keys vals _check(ccall((:dsa_pcsc_create, libdsa), Int32,
        (Ptr{Int64}, Int64, Ptr{Int64}, Ptr{Float64}, Int32, Ref{Ptr{Cvoid}}),
        colptr, length(row_keys), keys, vals, get(COMBINE, combine, Int32(0)), out))
    return PackedCSC(out[])
end
PackedCSC() = (out = Ref{Ptr{Cvoid}}(C_NULL); _check(ccall((:dsa_pcsc_create_empty, libdsa), Int32, (Ref{Ptr{Cvoid}},), out)); PackedCSC(out[]))
function Base.getindex(p::PackedCSC, key::Int64, partition::Int64)          # src/pcsr.jl:228-232
    out = Ref{Float64}(0.0)
    _check(ccall((:dsa_pcsc_get, libdsa), Int32, (Ptr{Cvoid}, Int64, Int64, Ref{Float64}), p.h, key, partition, out))
    return out[]
end
function Base.setindex!(p::PackedCSC, value, key::Int64, partition::Int64)  # src/pcsr.jl:294-310
    _check(ccall((:dsa_pcsc_set, libdsa), Int32, (Ptr{Cvoid}, Float64, Int64, Int64), p.h, Float64(value), key, partition))
    return p
end
deletepartition!(p::PackedCSC, partition::Int64) =                           # src/pcsr.jl:188-204
    (_check(ccall((:dsa_pcsc_deletepartition, libdsa), Int32, (Ptr{Cvoid}, Int64), p.h, partition)); nothing)
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

function _spmv_sparse(a::DynamicSparseMatrix, tr::Bool, xi::Vector{Int64}, xv::Vector{Float64}, n::Int64)
    cap = 1024
    while true
        yi = Vector{Int64}(undef, cap); yv = Vector{Float64}(undef, cap); k = Ref{Int64}(0)
        rc = GC.@preserve xi xv yi yv ccall((:dsa_mat_spmv_sparse, libdsa), Int32,
            (Ptr{Cvoid}, Int32, Ptr{Int64}, Ptr{Float64}, Int64, Ptr{Int64}, Ptr{Float64}, Int64, Ref{Int64}),
            a.h, tr ? 1 : 0, xi, xv, length(xi), yi, yv, cap, k)
        rc == 8 && (cap *= 16; continue)
        _check(rc)
        return sparsevec(resize!(yi, k[]), resize!(yv, k[]), n)       # _mul_output  src/operations.jl:11-12
    end
end
Base.:(*)(a::DynamicSparseMatrix, v::DynamicSparseVector) = _spmv_sparse(a, false, _stored(v)..., size(a, 1))
Base.:(*)(a::DynamicSparseMatrix, v::SparseVector{Float64,Int64}) = _spmv_sparse(a, false, rowvals(v), nonzeros(v), size(a, 1))
Base.:(*)(t::Transposed{DynamicSparseMatrix}, v::DynamicSparseVector) = _spmv_sparse(t.array, true, _stored(v)..., size(t.array, 2))
Base.:(*)(t::Transposed{DynamicSparseMatrix}, v::SparseVector{Float64,Int64}) = _spmv_sparse(t.array, true, rowvals(v), nonzeros(v), size(t.array, 2))
Base.:(*)(v::DynamicSparseVector, t::Transposed{DynamicSparseMatrix}) = t.array * v
Base.:(*)(v::SparseVector{Float64,Int64}, t::Transposed{DynamicSparseMatrix}) = t.array * v
Base.:(*)(v::DynamicSparseVector, a::DynamicSparseMatrix) = transpose(a) * v
Base.:(*)(v::SparseVector{Float64,Int64}, a::DynamicSparseMatrix) = transpose(a) * v
"dense x, dense y — the column-generation pricing product on device-resident data"
function Base.:(*)(a::DynamicSparseMatrix, x::Vector{Float64})
    y = Vector{Float64}(undef, size(a, 1))
    GC.@preserve x y _check(ccall((:dsa_mat_spmv_dense, libdsa), Int32,
        (Ptr{Cvoid}, Int32, Ptr{Float64}, Int64, Ptr{Float64}, Int64), a.h, 0, x, length(x), y, length(y)))
    return y
end

end # module
