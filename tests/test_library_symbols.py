"""CPU test: libdsa_hip.so builds for gfx950, loads, and exports every symbol include/dsa.h declares
(no compute call is made — there is no GPU in the build container)."""
import ctypes
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_in_header():
    txt = open(os.path.join(ROOT, "include", "dsa.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(dsa_[a-z0-9_]+)\s*\(", txt)))


def test_library_builds_and_exports_header_symbols(dsa):
    csrc = os.path.join(ROOT, "dynamicsparsearrays.jl_amd", "csrc")
    subprocess.check_call(["make", "-C", csrc, "-j4", "libdsa_hip.so"])
    lib = ctypes.CDLL(os.path.join(csrc, "libdsa_hip.so"))
    names = declared_in_header()
    assert len(names) >= 45
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    # the ctypes binding declares the same set
    bound = {"dsa_" + n for n in dsa.Binding.declared_symbols(device_api=True)}
    assert bound == set(names), sorted(bound ^ set(names))


def test_product_code_object_is_gfx950_only():
    csrc = os.path.join(ROOT, "dynamicsparsearrays.jl_amd", "csrc")
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--list", "--type=o",
                          "--input=" + os.path.join(csrc, "libdsa_hip.so")], capture_output=True, text=True)
    if out.returncode == 0 and out.stdout.strip():
        targets = [t for t in out.stdout.split() if "amdgcn" in t]
        assert targets and all("gfx950" in t for t in targets), targets


def test_product_does_not_reference_the_oracle():
    """The product path must not import, link or execute anything under oracle/."""
    pkg = os.path.join(ROOT, "dynamicsparsearrays.jl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".jl")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert "liboracle" not in txt and "oracle.hpp" not in txt and "ora_" not in txt.replace("ora_`", ""), (dirpath, f)



def test_shard_range_matches_the_host_layer(dsa):
    """dsa_shard_range (pure host arithmetic, callable without a GPU) and sharding.column_range split n columns the same way;
    the ranges tile 1..n without gaps or overlaps."""
    import ctypes as C
    from dsa_amd import sharding
    b = dsa.product()
    for n, G in [(10, 1), (10, 3), (1_000_000, 8), (7, 8), (0, 2), (10_000_000, 8)]:
        end = 0
        for g in range(G):
            c0, nc = C.c_int64(), C.c_int64()
            b.call("shard_range", n, G, g, C.byref(c0), C.byref(nc))
            assert (c0.value, nc.value) == sharding.column_range(g, G, n)
            assert c0.value == end
            end += nc.value
        assert end == n
    c0, nc = C.c_int64(), C.c_int64()
    import pytest
    with pytest.raises(dsa.DsaArgumentError):
        b.call("shard_range", 10, 2, 2, C.byref(c0), C.byref(nc))
