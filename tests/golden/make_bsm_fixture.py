"""Writes tests/golden/bsm_fixtures.json: the golden BlockSparseMatrix layouts that the reference's
own unit test holds (test/BlockSparseMatrix.jl:5-19 and :46-59, expected values :25,:64,:87).
These are data (inputs + expected outputs) transcribed from that test, not reference code."""
import json
import os

fix1 = dict(
    pattern=[[0, 0, 1], [1, 0, 0], [0, 1, 1], [0, 0, 0]],   # block rows x block cols (test line 15)
    rowsizes=[3, 3, 1, 1], colsizes=[2, 3, 1],
    blocks=[dict(i=2, j=1, rows=3, cols=2, values=[1, 2, 3, 4, 5, 6]),          # reshape(1:6,(3,2)) col-major
            dict(i=3, j=2, rows=1, cols=3, values=[7, 8, 9]),
            dict(i=1, j=3, rows=3, cols=1, values=[10, 11, 12]),
            dict(i=3, j=3, rows=1, cols=1, values=[13])],
    dense=[[0, 0, 0, 0, 0, 10], [0, 0, 0, 0, 0, 11], [0, 0, 0, 0, 0, 12], [1, 4, 0, 0, 0, 0],
           [2, 5, 0, 0, 0, 0], [3, 6, 0, 0, 0, 0], [0, 0, 7, 8, 9, 13], [0, 0, 0, 0, 0, 0]],
    size=[8, 6], nnz=13)
fix2 = dict(
    pattern=[[0, 0, 0], [1, 1, 0], [0, 1, 1]],
    rowsizes=[2, 3, 1], colsizes=[2, 3, 1],
    blocks=[dict(i=2, j=1, rows=3, cols=2, values=[1, 2, 3, 4, 5, 6]),
            dict(i=3, j=2, rows=1, cols=3, values=[13, 14, 15]),
            dict(i=2, j=2, rows=3, cols=3, values=[7, 8, 9, 8, 10, 11, 9, 11, 12]),
            dict(i=3, j=3, rows=1, cols=1, values=[16])],
    dense=[[0, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 0], [1, 4, 7, 8, 9, 0], [2, 5, 8, 10, 11, 0],
           [3, 6, 9, 11, 12, 0], [0, 0, 13, 14, 15, 16]],
    size=[6, 6], nnz=19, nnz_symmetric=28)
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bsm_fixtures.json")
json.dump(dict(fixture1=fix1, fixture2=fix2), open(out, "w"), indent=1)
print("wrote", out)
