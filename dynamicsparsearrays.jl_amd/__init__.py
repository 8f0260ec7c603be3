"""MI355X-native packed-memory-array / packed-CSR engine (gfx950).

The directory name carries a dot, so it cannot be imported with a plain `import`
statement; load it with :func:`load` from `dsa_loader.py` at the repo root (tests,
bench.py and __graft_entry__.py do), which registers it as module ``dsa_amd``.

Layout: ``csrc/`` HIP kernels + the C ABI (``include/dsa.h``) built into
``csrc/libdsa_hip.so``; ``binding.py`` ctypes binding; ``api.py`` host-side mirror
of the reference's public surface; ``julia/`` the `ccall` wrapper a Julia host uses.
"""
from . import binding  # noqa: F401
from .api import (  # noqa: F401
    COLMAJOR, ROWMAJOR, DynamicSparseMatrix, DynamicSparseVector, PackedCSC, Transposed,
    addrow, closefillmode, deletecolumn, deletepartition, deleterow, dynamicsparse,
    dynamicsparsevec, import_packedcsc_layout, import_vector_layout, nbpartitions, nnz, packedcsc, packedcsc_empty, pool_idle_bytes,
    pool_trim, shrink_size, dev_switches,
)
from .binding import Binding, DsaArgumentError, DsaBoundsError, DsaError, DsaErrorException, product  # noqa: F401
