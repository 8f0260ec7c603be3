#!/usr/bin/env python3
"""Writes tests/golden/reference_cases.json.

These are the deterministic known-answer vectors held by the reference's OWN tests
(atoptima/DynamicSparseArrays.jl v0.7.2, test/ and README.md), re-expressed as data:
inputs + expected outputs only, each with the reference file:line it comes from.
Nothing here is computed — the reference is Julia and cannot run in the build image;
the expected values are the literals the reference's tests assert.  Char keys used by
a few reference tests ('a'..'e') are mapped to 1..5.

Slots are written as [key, value] or null (= `nothing`).  Scenario steps are
interpreted by tests/scenario.py against any backend (CPU oracle / HIP library).
"""
import json
import os

N = None
cases = {}

# ---------------------------------------------------------------- find
A1 = [N, [3, 10], [4, 10], N, [8, 10], N, [9, 10]]
A3 = [N, [3, 10], N, N, [9, 10], N, [10, 10], N, [3, 10], N, N, [2, 1], N]
A4 = [N, [3, 10], N, [9, 10], N, [10, 10], [3, 10], [3, 10], N, [2, 1], N]
cases["find"] = [
    # test/unit/finds.jl:4-23 — whole array
    dict(ref="test/unit/finds.jl:4-23", array=A1, frm=1, to=7,
         queries=[[1, 0, N], [2, 0, N], [3, 2, [3, 10]], [4, 3, [4, 10]], [5, 3, [4, 10]], [7, 3, [4, 10]],
                  [8, 5, [8, 10]], [9, 7, [9, 10]], [100, 7, [9, 10]]]),
    # test/unit/finds.jl:26-48 — sub-array 4..6: predecessor may lie in the left outside
    dict(ref="test/unit/finds.jl:26-48", array=A1, frm=4, to=6,
         queries=[[1, 3, [4, 10]], [2, 3, [4, 10]], [3, 3, [4, 10]], [4, 3, [4, 10]], [5, 3, [4, 10]],
                  [7, 3, [4, 10]], [8, 5, [8, 10]], [9, 5, [8, 10]], [100, 5, [8, 10]]]),
    # test/unit/finds.jl:62-75 — semaphore inside the searched range (the bug narrative)
    dict(ref="test/unit/finds.jl:62-75", array=A3, frm=2, to=8, queries=[[4, 2, [3, 10]], [2, 0, N]]),
    # test/unit/finds.jl:86-91 — semaphore excluded from the range (the fix)
    dict(ref="test/unit/finds.jl:86-91", array=A3, frm=3, to=8, queries=[[4, 2, [3, 10]], [2, 2, [3, 10]]]),
    # test/unit/finds.jl:97-107 — empty column
    dict(ref="test/unit/finds.jl:101-103", array=A4, frm=3, to=6, queries=[[2, 2, [3, 10]]]),
    dict(ref="test/unit/finds.jl:104-105", array=A4, frm=8, to=7, queries=[[2, 7, [3, 10]]]),
    dict(ref="test/unit/finds.jl:106-107", array=A4, frm=9, to=11, queries=[[1, 8, [3, 10]]]),
]

# ---------------------------------------------------------------- insert! / delete! / purge!
cases["insert"] = [
    # test/unit/writes.jl:5-26 — sequence on the whole array; the last step must fail (array full)
    dict(ref="test/unit/writes.jl:5-26", array=[[2, 10], N, [3, 10], [5, 10], [6, 10], N, [7, 10]],
         steps=[dict(key=4, val=10, frm=1, to=7,
                     expect=[[2, 10], N, [3, 10], [4, 10], [5, 10], [6, 10], [7, 10]]),
                dict(key=1, val=10, frm=1, to=7,
                     expect=[[1, 10], [2, 10], [3, 10], [4, 10], [5, 10], [6, 10], [7, 10]]),
                dict(key=2, val=11, frm=1, to=7,
                     expect=[[1, 10], [2, 11], [3, 10], [4, 10], [5, 10], [6, 10], [7, 10]]),
                dict(key=8, val=10, frm=1, to=7, error="EFULL")]),
    # test/unit/writes.jl:28-46 — inside sub-array 3..6 (key found outside the range is NOT overwritten)
    dict(ref="test/unit/writes.jl:28-46",
         array=[[2, 10], N, N, [3, 10], [5, 10], [6, 10], N, [7, 10]],
         steps=[dict(key=1, val=10, frm=3, to=6,
                     expect=[[2, 10], [1, 10], N, [3, 10], [5, 10], [6, 10], N, [7, 10]]),
                dict(key=1, val=11, frm=3, to=6,
                     expect=[[2, 10], [1, 10], [1, 11], [3, 10], [5, 10], [6, 10], N, [7, 10]]),
                dict(key=4, val=10, frm=3, to=6,
                     expect=[[2, 10], [1, 10], [1, 11], [3, 10], [4, 10], [5, 10], [6, 10], [7, 10]])]),
]
cases["delete"] = dict(
    ref="test/unit/writes.jl:52-70", array=[[2, 10], [3, 10], N, [8, 10], [9, 10], N, [10, 10]],
    steps=[dict(op="delete", key=2, out=[1, True], expect=[N, [3, 10], N, [8, 10], [9, 10], N, [10, 10]]),
           dict(op="delete", key=2, out=[0, False], expect=[N, [3, 10], N, [8, 10], [9, 10], N, [10, 10]]),
           dict(op="purge", frm=3, to=5, out=[4, 2], expect=[N, [3, 10], N, N, N, N, [10, 10]])])

# ---------------------------------------------------------------- _arrays_equal (src/pma.jl:236-260)
cases["arrays_equal"] = [
    dict(ref="test/unit/comparison.jl:2-5", a=[N, [1, 1], N, N, [2, 1], N, [3, 2]], b=[N, [1, 1], [2, 1], [3, 2], N], expect=True),
    dict(ref="test/unit/comparison.jl:7-10", a=[N, [1, 1], N, N, [2, 1], N, [3, 2]], b=[N, [1, 1], [2, 1], [3, 2], [4, 2]], expect=False),
]

# ---------------------------------------------------------------- scenarios
S = []

# README.md:19-40
S.append(dict(name="readme_vector", ref="README.md:19-27", kind="vector",
              create=dict(I=[1, 10, 3, 5, 3], V=[1.0, 2.4, 7.1, 1.1, 1.0]),
              steps=[["get", 3, 8.1], ["set", 78, 1.5], ["get", 2, 0.0], ["set", 2, 0], ["get", 78, 1.5]]))
S.append(dict(name="readme_matrix", ref="README.md:30-40", kind="matrix",
              create=dict(I=[1, 2, 3, 2, 6, 7, 1, 6, 8], J=[1, 1, 1, 2, 2, 2, 3, 3, 3], V=[2, 3, 4, 2, 4, 5, 3, 5, 7]),
              steps=[["set", 4, 1, 1], ["set", 2, 2, 0], ["deletecolumn", 2], ["get", 2, 6, 0.0], ["get", 4, 1, 1.0],
                     ["check_invariants"]]))

# test/functional/sparsevector.jl:2-58
S.append(dict(name="vec_empty", ref="test/functional/sparsevector.jl:2-4", kind="vector",
              create=dict(I=[], V=[]), steps=[["len", 0], ["nnz", 0], ["capacity", 256]]))
VI = [1, 2, 5, 5, 3, 10, 1, 8, 1, 5]
VV = [1.0, 3.5, 2.1, 8.5, 2.1, 1.1, 5.0, 7.8, 1.1, 2.0]
S.append(dict(name="vec_simple_use_add", ref="test/functional/sparsevector.jl:7-58", kind="vector",
              create=dict(I=VI, V=VV),
              steps=[["capacity", 16], ["nnz", 6],                                         # :12 "16-element ... 6 stored"
                     ["get", 1, 1.0 + 1.1 + 5.0], ["get", 2, 3.5], ["get", 3, 2.1], ["get", 4, 0.0],
                     ["get", 5, 2.1 + 8.5 + 2.0], ["get", 8, 7.8], ["get", 10, 1.1],
                     ["len", 10],
                     ["set", 1, 0], ["set", 2, 0], ["set", 3, 0], ["set", 22, 0], ["set", 1001, 1.8],
                     ["set", 987, 4.7], ["set", 2, 15 / 3], ["set", 4, 42],
                     ["get", 1, 0.0], ["get", 2, 15 / 3], ["get", 3, 0.0], ["get", 4, 42.0],
                     ["get", 1001, 1.8], ["get", 987, 4.7],
                     ["iter", [[2, 5.0], [4, 42.0], [5, 12.6], [8, 7.8], [10, 1.1], [987, 4.7], [1001, 1.8]]],
                     ["len", 1001]]))
S.append(dict(name="vec_simple_use_mul", ref="test/functional/sparsevector.jl:22-29", kind="vector",
              create=dict(I=VI, V=VV, combine="*"),
              steps=[["get", 1, 1.0 * 1.1 * 5.0], ["get", 2, 3.5], ["get", 3, 2.1], ["get", 5, 2.1 * 8.5 * 2.0],
                     ["get", 6, 0.0], ["get", 8, 7.8], ["get", 10, 1.1]]))
S.append(dict(name="vec_equality_after_shrink", ref="test/functional/sparsevector.jl:64-77", kind="vector_pair",
              a=dict(I=[1, 2, 3, 5, 6, 8, 9], V=[1.0, 1.0, 1.0, 2.0, 1.0, 1.0, 3.0]),
              b=dict(I=[1, 2, 3, 5, 6, 8, 9, 10, 11], V=[1.0, 1.0, 1.0, 2.0, 1.0, 1.0, 3.0, 2.0, 3.0]),
              steps=[["expect_equal", False], ["b_set", 10, 0], ["b_set", 11, 0], ["shrink_both"],
                     ["expect_equal", True]]))

# filter(f, vec) with f = "key is even"  test/functional/sparsevector.jl:81-86
S.append(dict(name="vec_filter_even_keys", ref="test/functional/sparsevector.jl:81-86", kind="vector",
              create=dict(I=[1, 2, 3], V=[2.0, 3.0, 4.0]),
              steps=[["filter_even_keys_equals", [2], [3.0]]]))

# test/functional/sparsematrix.jl:7-119  PackedCSC
DENSE1 = [[2, 0, 3], [3, 2, 0], [4, 0, 0], [0, 0, 0], [0, 0, 0], [0, 4, 5], [0, 5, 0], [0, 0, 7]]
DENSE1B = [[4, 3, 3], [3, 2, 0], [0, 0, 0], [0, 1, 0], [0, 0, 0], [0, 4, 5], [0, 5, 0], [0, 0, 7]]
S.append(dict(name="pcsc_simple_use", ref="test/functional/sparsematrix.jl:7-101", kind="pcsc",
              create=dict(row_keys=[[1, 2, 3], [2, 6, 7], [1, 6, 8]], values=[[2, 3, 4], [2, 4, 5], [3, 5, 7]]),
              steps=[["nbpartitions", 3], ["check_invariants"], ["nnz", 9],
                     ["dense", DENSE1],
                     ["set", 1, 1, 4], ["add", 1, 2, 3], ["set", 3, 1, 0], ["set", 4, 2, 1],
                     ["nnz", 10], ["nbpartitions", 3], ["dense", DENSE1B],
                     ["set", 10, 5, 9], ["nnz", 11], ["nbpartitions", 5],          # :82-84 auto-added partitions 4, 5
                     ["set", 1, 4, 2], ["nnz", 12], ["check_invariants"],
                     ["deletepartition", 2], ["nbpartitions", 4], ["nnz", 7],    # 12 - 5 entries of partition 2
                     ["check_invariants", 4],
                     ["expect_error", "EDELETED", ["set", 1, 2, 1]]]))
S.append(dict(name="pcsc_duplicates_and_empty_column", ref="test/functional/sparsematrix.jl:104-119", kind="pcsc",
              create=dict(row_keys=[[1, 2, 3, 1, 2], [], [2, 6, 7, 7, 5], [1, 6, 8, 2, 1]],
                          values=[[2, 3, 4, 1, 1], [], [2, 4, 5, 1, 1], [3, 5, 7, 1, 1]]),
              steps=[["nbpartitions", 4], ["check_invariants"], ["nnz", 11],
                     ["dense", [[3, 0, 0, 4], [4, 0, 2, 1], [4, 0, 0, 0], [0, 0, 0, 0], [0, 0, 1, 0], [0, 0, 4, 5],
                                [0, 0, 6, 0], [0, 0, 0, 7]]]]))

# test/functional/sparsematrix.jl:174-299  DynamicSparseMatrix
S.append(dict(name="matrix_simple_use_A", ref="test/functional/sparsematrix.jl:174-285", kind="matrix",
              create=dict(I=[1, 2, 3, 2, 6, 7, 1, 6, 8], J=[1, 1, 1, 2, 2, 2, 3, 3, 3], V=[2, 3, 4, 2, 4, 5, 3, 5, 7]),
              steps=[["check_invariants"], ["nnz", 9], ["size", 8, 3], ["dense", DENSE1],
                     ["set", 1, 1, 4], ["add", 1, 2, 3], ["set", 3, 1, 0], ["set", 4, 2, 1],
                     ["nnz", 10], ["size", 8, 3], ["dense", DENSE1B],
                     ["row_view", 2, [[1, 3.0], [2, 2.0]]],                       # :217-223 row 2 has 2 entries
                     ["col_view", 2, [[1, 3.0], [2, 2.0], [4, 1.0], [6, 4.0], [7, 5.0]]],   # :226-232 nnz(column)==5
                     ["set", 10, 5, 9], ["nnz", 11], ["nbpartitions", 0, 4],
                     ["set", 1, -1, 1], ["set", 1, 4, 2], ["set", 3, 4, 5],
                     ["get", 1, 4, 2.0], ["get", 3, 4, 5.0], ["get", 1, -1, 1.0],
                     ["nbpartitions", 0, 6], ["check_invariants"],
                     ["deletecolumn", 2], ["nbpartitions", 1, 8], ["nbpartitions", 0, 5],
                     ["check_invariants", 5, 8],
                     ["set", 1, 2, 1], ["get", 1, 2, 1.0], ["check_invariants", 6, 8]]))
S.append(dict(name="matrix_duplicate_combine_B", ref="test/functional/sparsematrix.jl:288-299", kind="matrix",
              create=dict(I=[1, 1, 2, 4, 3, 5, 1, 3, 1, 5, 1, 5, 4], J=[4, 3, 3, 7, 18, 9, 3, 18, 4, 2, 3, 1, 7],
                          V=[1, 8, 10, 2, -5, 3, 2, 1, 1, 1, 5, 3, 2]),
              steps=[["get", 1, 4, 2.0], ["get", 1, 3, 15.0], ["get", 4, 7, 4.0], ["get", 3, 18, -4.0],
                     ["get", 5, 9, 3.0], ["get", 5, 2, 1.0], ["get", 5, 1, 3.0], ["get", 2, 3, 10.0],
                     ["check_invariants"],
                     # dynsparsematrix_deletions (defined at :385-410, never called by the reference's runner)
                     ["deletecolumn", 3], ["check_invariants"],
                     ["get", 1, 3, 0.0], ["get", 2, 3, 0.0], ["get", 3, 3, 0.0], ["get", 4, 3, 0.0], ["get", 5, 3, 0.0]]))
# Char columns 'a'..'e' -> 1..5
S.append(dict(name="matrix_char_columns_C", ref="test/functional/sparsematrix.jl:302-336", kind="matrix",
              create=dict(I=[1, 1, 2, 4, 1, 2, 4, 5, 5, 2], J=[1, 3, 3, 1, 4, 1, 5, 5, 3, 4],
                          V=[1, 2, 3, 4, 5, 6, 7, 8, 9, 10]),
              steps=[["get", 1, 1, 1.0], ["get", 1, 3, 2.0], ["get", 2, 3, 3.0], ["get", 4, 1, 4.0], ["get", 1, 4, 5.0],
                     ["get", 2, 1, 6.0], ["get", 4, 5, 7.0], ["get", 5, 5, 8.0], ["get", 5, 3, 9.0], ["get", 2, 4, 10.0],
                     ["size", 5, 5],
                     ["set", 2, 2, 11], ["get", 2, 2, 11.0],
                     ["nbpartitions", 1, 4], ["nbpartitions", 0, 5],
                     ["deletecolumn", 1], ["get", 1, 1, 0.0], ["get", 2, 1, 0.0],
                     ["deleterow", 5], ["get", 5, 3, 0.0], ["get", 5, 5, 0.0],
                     ["nbpartitions", 1, 3], ["nbpartitions", 0, 4], ["check_invariants"]]))
S.append(dict(name="matrix_insertions_and_gets", ref="test/functional/sparsematrix.jl:341-361", kind="matrix",
              create=dict(I=[1, 4, 3, 5], J=[4, 7, 18, 9], V=[1, 2, -5, 3]),
              steps=[["set", 2, 7, 8], ["get", 2, 7, 8.0], ["set", 1, 2, 21], ["get", 1, 2, 21.0],
                     ["set", 10, 33, 21], ["get", 10, 33, 21.0], ["set", 55, 54, 53], ["get", 55, 54, 53.0],
                     ["get", 1, 4, 1.0], ["get", 4, 7, 2.0], ["get", 3, 18, -5.0], ["get", 5, 9, 3.0],
                     ["check_invariants"]]))

# test/unit/views.jl:1-42
S.append(dict(name="views", ref="test/unit/views.jl:1-42", kind="matrix",
              create=dict(I=[1, 1, 2, 4, 3, 5, 1, 4, 1, 5, 1, 5, 4, 4, 3, 9, 1],
                          J=[4, 3, 3, 7, 18, 9, 3, 18, 4, 2, 3, 1, 7, 3, 3, 3, 18],
                          V=[1, 8, 10, 2, -5, 3, 2, 1, 1, 1, 5, 3, 2, 1, 7, 8, 1]),
              steps=[["row_view", 5, [[1, 3.0], [2, 1.0], [9, 3.0]]],
                     ["col_view", 3, [[1, 15.0], [2, 10.0], [3, 7.0], [4, 1.0], [9, 8.0]]],
                     ["col_view", 18, [[1, 1.0], [3, -5.0], [4, 1.0]]]]))

# test/unit/spmv.jl:5-128   ('a'..'e' -> 1..5)
SPI = [1, 1, 1, 2, 2, 3, 4, 4, 4]
SPJ = [1, 3, 5, 2, 4, 4, 1, 4, 5]
SPV = [1, 2, 1, 2, 1, 3, 3, 2, 2]
S.append(dict(name="spmv_1", ref="test/unit/spmv.jl:5-26", kind="matrix", create=dict(I=SPI, J=SPJ, V=SPV),
              steps=[["mul", 0, [1, 3, 5], [1, 1, 1], {"1": 4.0, "4": 5.0}, [2, 3, 5]]]))
S.append(dict(name="spmv_2_transposed", ref="test/unit/spmv.jl:29-59", kind="matrix", create=dict(I=SPI, J=SPJ, V=SPV),
              steps=[["mul", 1, [1, 3, 5], [1, 1, 1], {"1": 1.0, "3": 2.0, "4": 3.0, "5": 1.0}, [2]],
                     ["set", 5, 2, 5], ["get", 5, 2, 5.0]]))
S.append(dict(name="spmv_3_empty_rows_cols", ref="test/unit/spmv.jl:61-82", kind="matrix",
              create=dict(I=[1, 1, 3, 3, 4, 4, 4, 6, 6, 6], J=[2, 4, 1, 3, 1, 3, 6, 1, 3, 6],
                          V=[1, 2, 1, 1, 1, 2, 1, 1, 1, 1]),
              steps=[["mul", 0, [2, 5, 6], [1, 1, 1], {"1": 1.0, "4": 1.0, "6": 1.0}, [2, 3, 5]]]))
S.append(dict(name="spmv_4_after_deletes", ref="test/unit/spmv.jl:84-128", kind="matrix",
              create=dict(I=[1, 1, 3, 3, 4, 4, 4, 6, 6, 6], J=[2, 4, 1, 3, 1, 3, 6, 1, 3, 6],
                          V=[1, 2, 1, 1, 1, 1, 1, 1, 1, 1]),
              steps=[["mul", 0, [2, 3, 5, 6], [1, 1, 1, 1], {"1": 1.0, "3": 1.0, "4": 2.0, "6": 2.0}, [2, 5]],
                     ["deletecolumn", 3],
                     ["mul", 0, [2, 3, 5, 6], [1, 1, 1, 1], {"1": 1.0, "4": 1.0, "6": 1.0}, [2, 3, 5]],
                     ["deleterow", 4],
                     ["mul", 0, [2, 3, 5, 6], [1, 1, 1, 1], {"1": 1.0, "6": 1.0}, [2, 3, 4, 5]]]))

# test/functional/sparsematrix.jl:412-467, 509-518  fill mode
FV = [[1, 0, 0, 2, 0, 7, 0, 0, 0, 9, 1, 2],
      [0, 3, 0, 0, 1, 1, 0, 0, 0, 1, 0, 2],
      [0, 0, 0, 1, 1, 2, 0, 0, 1, 2, 0, 0],
      [0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 1],
      [1, 2, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0]]
steps = []
for i, row in enumerate(FV, start=1):
    cols = [j for j, v in enumerate(row, start=1) if v != 0]
    steps.append(["addrow", i, cols, [row[j - 1] for j in cols]])
steps += [["set", 1, 2, 2], ["set", 1, 1, 1]]          # in fill mode writes accumulate at flush (:433-437)
FV2 = [list(r) for r in FV]
FV2[0][1] = 2
FV2[0][0] += 1
steps += [["expect_error", "EMODE", ["col_view", 1, []]],
          ["closefillmode"], ["dense", FV2], ["dense_rowmajor", FV2], ["check_invariants"],
          ["addrow", 7, [1, 3, 4, 5], [2, 3, 6, 7]],
          ["get", 7, 1, 2.0], ["get", 7, 3, 3.0], ["get", 7, 4, 6.0], ["get", 7, 5, 7.0],
          ["expect_error", "EMODE", ["closefillmode"]]]
S.append(dict(name="fill_mode", ref="test/functional/sparsematrix.jl:412-467", kind="matrix",
              create=dict(fill_mode=True), steps=steps))
S.append(dict(name="fill_mode_close_empty", ref="test/functional/sparsematrix.jl:509-518", kind="matrix",
              create=dict(fill_mode=True), steps=[["closefillmode"], ["nnz", 0], ["get", 1, 1, 0.0]]))
S.append(dict(name="fill_mode_off_close_errors", ref="test/functional/sparsematrix.jl:515-517", kind="matrix",
              create=dict(fill_mode=False), steps=[["expect_error", "EMODE", ["closefillmode"]]]))
# test/unit/views.jl:44-67 buffer is host-side plumbing (BufferView); its flush result is what matters:
S.append(dict(name="fill_mode_setindex_flush", ref="test/unit/views.jl:44-67", kind="matrix",
              create=dict(fill_mode=True),
              steps=[["set", 1, 2, 1], ["set", 2, 1, 2], ["set", 2, 2, 3], ["set", 3, 1, 4], ["set", 3, 2, 5],
                     ["set", 1, 7, 3], ["closefillmode"],
                     ["row_view", 1, [[2, 1.0], [7, 3.0]]], ["nnz", 6], ["size", 3, 7]]))

cases["scenarios"] = S

out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_cases.json")
with open(out, "w") as f:
    json.dump(cases, f, indent=1)
print("wrote", out, len(S), "scenarios")
