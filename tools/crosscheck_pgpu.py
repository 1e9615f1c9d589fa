#!/usr/bin/env python3
"""Cross-check in the reference's own terms (src/GPU_CUDAC++/results.txt:48-51): feed the P this build's
GPU setup driver wrote (multigridsolver_amd/cpp/mgs_agmg → gpurun_out/pgpu/<name>promatrix_gpu.mtx, produced on
the GPU box by tools/make_pgpu.sh) to the REAL reference solve (oracle/_ref/ref_dump = src/common/bicg.cpp:
ILUT + SparseLU two-grid BiCGSTAB, tol 1e-6) and compare the iteration count with the reference's own CPU
setup.  Runs in the build container only (needs /root/reference); writes tests/golden/pgpu_crosscheck.json."""
import gzip, hashlib, json, os, shutil, subprocess, sys, tempfile
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(REPO, "oracle", "_ref", "ref_dump")
INP = os.path.join(REPO, "tests", "golden", "inputs")
PG = os.path.join(REPO, "gpurun_out", "pgpu")
tmp = tempfile.mkdtemp()
out = {"what": "reference bicg (ILUT two-grid BiCGSTAB, tol 1e-6, src/common/bicg.cpp) iterations with P from this build's "
               "GPU setup (mgs_agmg 10 2 8) vs P from the reference CPU setup (src/CPU_C++/main 10 2 8)", "cases": {}}
for m in ["CSky3d30", "CSky3d10", "CSky2d20", "poisson10000"]:
    a = os.path.join(tmp, m + ".mtx")
    if m == "CSky3d30":
        with gzip.open(os.path.join(INP, m + ".mtx.gz"), "rb") as f, open(a, "wb") as g:
            shutil.copyfileobj(f, g)
    elif m == "poisson10000":
        subprocess.run([sys.executable, os.path.join(REPO, "tools", "write_poisson_mtx.py"), "100", a], check=True)
    else:
        shutil.copy(os.path.join(INP, m + ".mtx"), a)
    pg = os.path.join(PG, m + "promatrix_gpu.mtx"); pc = os.path.join(tmp, m + "promatrix_cpu.mtx")
    dg, dc = os.path.join(tmp, "g_" + m), os.path.join(tmp, "c_" + m); os.makedirs(dg); os.makedirs(dc)
    subprocess.run([REF, "agmg", a, "10", "2", "8", pc, dc], check=True, capture_output=True)
    subprocess.run([REF, "case", a, pg, dg], check=True, capture_output=True)
    subprocess.run([REF, "case", a, pc, dc], check=True, capture_output=True)
    rd = lambda d, f: open(os.path.join(d, f)).read().split()
    g, c = rd(dg, "bicg_ilut.txt"), rd(dc, "bicg_ilut.txt")
    out["cases"][m] = {"P_gpu_sha256": hashlib.sha256(open(pg, "rb").read()).hexdigest(), "P_gpu_shape": open(pg).read().split("\n")[1],
                       "P_cpu_shape": open(pc).read().split("\n")[1],
                       "ref_bicg_ilut_with_P_gpu": {"status": int(g[0]), "iterations": int(g[1]), "tol": float(g[2])},
                       "ref_bicg_ilut_with_P_cpu": {"status": int(c[0]), "iterations": int(c[1]), "tol": float(c[2])},
                       "ref_bicg_jacobi2grid_w05_iterations": {"P_gpu": int(rd(dg, "bicg_jac_w05.txt")[1]), "P_cpu": int(rd(dc, "bicg_jac_w05.txt")[1])}}
json.dump(out, open(os.path.join(REPO, "tests", "golden", "pgpu_crosscheck.json"), "w"), indent=1)
print(json.dumps(out["cases"], indent=1))
shutil.rmtree(tmp)
