#!/usr/bin/env python3
"""the reference's published size end to end through the drop-in driver: poisson<n·n>.mtx (src/common/poisson.cpp:9-37 format, n = 1000 →
1e6 rows, the case of src/GPU_CUDAC++/results.txt:84-95: 3.03 s for the reference's solve on its Xeon) written with mgs_mtx_write, then
`mgs_bicg poisson<n·n> device` (file read + upload + device setup + BiCGSTABiml to 1e-6) timed as a whole process.
usage: e2e_poisson.py [n=1000]"""
import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import scipy.sparse as sp
import multigridsolver_amd as mg
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
I = sp.identity(n); T = sp.diags([-1, -1], [-1, 1], shape=(n, n))
A = (sp.kron(I, sp.diags([4], [0], shape=(n, n)) + T) + sp.kron(T, I)).tocsr(); A.sort_indices()
d = tempfile.mkdtemp(prefix="mgs_e2e_")
name = f"poisson{n * n}"
t0 = time.perf_counter()
mg.write_mtx(os.path.join(d, name + ".mtx"), A.shape[0], A.shape[1], A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float64))
print(f"wrote {name}.mtx: {A.shape[0]} rows, {A.nnz} entries, {os.path.getsize(os.path.join(d, name + '.mtx')) / 1e6:.0f} MB in {time.perf_counter() - t0:.2f} s")
exe = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "multigridsolver_amd", "cpp", "mgs_bicg")
env = dict(os.environ, MGS_MATRIX_DIR=d)
for rep in range(2):
    t0 = time.perf_counter()
    r = subprocess.run([exe, name, "device"], env=env, capture_output=True, text=True)
    dt = time.perf_counter() - t0
    print(f"run {rep}: mgs_bicg {name} device: {dt:.2f} s wall for the whole process (rc {r.returncode})")
    print("   " + " | ".join(l.strip() for l in (r.stdout + r.stderr).splitlines() if l.strip())[:600])
