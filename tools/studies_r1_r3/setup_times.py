#!/usr/bin/env python3
"""Device AGMG setup time (one multiple-pairwise aggregation, ktg 10 npass 2 tou 8 — what the reference's
setup programs do) on the sizes src/GPU_CUDAC++/results.txt:28-42 lists.  Not the headline path; context
for SURVEY §8 row f-1."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import multigridsolver_amd as mg
ctx = mg.Context(0)
ref = {1000: (1.62, 0.36), 1500: (4.33, 0.62), 2000: (8.26, 0.99), 2500: (13.51, 1.63), 3000: (19.77, 2.96), 3500: (27.54, 4.39)}
out = []
for n in (1000, 1500, 2000, 2500, 3000, 3500):
    A = ctx.poisson2d(n); ctx.sync()
    best = 1e9
    for rep in range(3):
        ctx.sync(); t0 = time.perf_counter()
        h = mg.Hierarchy(A, 0.6, 1, 1).coarsen(10.0, 2, 8.0, 0, 2)
        ctx.sync(); best = min(best, time.perf_counter() - t0)
        nc = h.level_shape(1)[0]; del h
    out.append({"problem": f"Poisson{n} ({n*n} rows)", "seconds_mi355x_f64": best, "coarse_rows": nc,
                "reference_cpu_seconds": ref[n][0], "reference_L4_gpu_f32_seconds": ref[n][1]})
    print(out[-1], flush=True)
    del A
json.dump({"source_of_reference_numbers": "src/GPU_CUDAC++/results.txt:28-35 (Xeon 2.2 GHz / NVIDIA L4, float32)", "rows": out},
          open(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "gpurun_out", "setup_times.json"), "w"), indent=1)
