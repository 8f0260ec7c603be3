# DynamicSparseArrays.jl — the module under the REFERENCE's name, so that `using DynamicSparseArrays` in Coluna (or in the
# reference's own test-suite) resolves to the MI355X-native implementation without touching the caller: put this
# directory on LOAD_PATH in front of the registered package.  It re-exports exactly the twelve names the reference exports
# (src/DynamicSparseArrays.jl:5-16) from DynamicSparseArraysAMD.jl, plus — unexported, as in the reference — the names its
# tests and Coluna reach through the module (`DynamicSparseArrays.PackedCSC`, `DynamicSparseArrays.nbpartitions`, ...).
# Unexecuted here (no Julia toolchain in the build image); tests/test_library_symbols.py checks the export lists.
module DynamicSparseArrays

include("DynamicSparseArraysAMD.jl")
using .DynamicSparseArraysAMD

export DynamicSparseVector,
       DynamicSparseMatrix,
       DynamicMatrixColView,
       dynamicsparsevec,
       dynamicsparse,
       nbpartitions,
       deletepartition!,
       deletecolumn!,
       deleterow!,
       addrow!,
       closefillmode!,
       shrink_size!

const PackedCSC = DynamicSparseArraysAMD.PackedCSC
const Transposed = DynamicSparseArraysAMD.Transposed
const keyint = DynamicSparseArraysAMD.keyint            # extend these two for id structs used as keys (key-mapping layer)
const keyfrom = DynamicSparseArraysAMD.keyfrom
const set_device! = DynamicSparseArraysAMD.set_device!
const getindex_batch = DynamicSparseArraysAMD.getindex_batch
const setindex_batch! = DynamicSparseArraysAMD.setindex_batch!

end # module
