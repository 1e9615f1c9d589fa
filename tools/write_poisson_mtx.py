#!/usr/bin/env python3
"""write poisson<n*n>.mtx in the format of the reference's generator (src/common/poisson.cpp:9-37)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle_py as orc
n, path = int(sys.argv[1]), sys.argv[2]
A = orc.poisson2d(n); rp, c, v = A.rowptr, A.col, A.val
with open(path, "w") as f:
    f.write("%MatrixMarket matrix coordinate real general\n")
    f.write("%d %d %d\n" % (A.shape[0], A.shape[1], A.nnz))
    for i in range(A.shape[0]):
        for k in range(rp[i], rp[i + 1]):
            f.write("%d %d %d\n" % (i + 1, c[k] + 1, int(v[k])))
