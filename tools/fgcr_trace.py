#!/usr/bin/env python3
"""Workload for a kernel trace of ONE FGCR(10) + K(4, energy) solve at 512^3 (run under rocprofv3 --kernel-trace by tools/run_trace_fgcr.sh);
with argument `summarize <kernel_trace.csv>`: time per FGCR iteration by kernel class (K-cycle levels >= 1 and the fine-level passes told apart by grid size)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 2 and sys.argv[1] == "summarize":
    import csv
    rows = list(csv.DictReader(open(sys.argv[2])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    def short(n): return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    # the solve = everything after the last fill_kernel of fine size (x.fill(0)) — marker: the workload prints nothing; take the last 40 % by time instead
    N3 = 512 ** 3
    fine = lambda r: int(r.get("Grid_Size_X", r.get("Grid_Size", 0)))
    # find the marker dispatches: two consecutive rand_kernel launches mark the start of the traced solve
    names = [short(r["Kernel_Name"]) for r in rows]
    marks = [i for i, nm in enumerate(names) if nm.startswith("rand_kernel")]
    s = marks[-1] + 1
    t0 = int(rows[s]["Start_Timestamp"]); t1 = int(rows[-1]["End_Timestamp"])
    by = {}
    spmv_fine = 0
    for i in range(s, len(rows)):
        d = int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])
        nm = names[i]
        big = fine(rows[i]) >= 100_000_000 or (("mdot" in nm or "maxpy" in nm or "axpb" in nm or "dot" in nm or "nrm" in nm) and d > 150_000)
        key = (nm[:60], "fine" if big else "coarse")
        e = by.setdefault(key, [0, 0]); e[0] += 1; e[1] += d
        if nm.startswith("csr_rowblock_coded_kernel<0") and big: spmv_fine += 1
    its = max(spmv_fine - 1, 1)
    gaps = [int(rows[i]["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"]) for i in range(s + 1, len(rows))]
    big = sorted(((g, i) for i, g in enumerate(gaps, start=s + 1) if g > 20000), reverse=True)
    print(f"gaps between consecutive kernels: total {sum(g for g in gaps if g > 0) / 1e6:.2f} ms; {len(big)} gaps above 20 us sum {sum(g for g, _ in big) / 1e6:.2f} ms; largest: "
          + ", ".join(f"{g / 1e3:.0f} us before `{names[i][:40]}`" for g, i in big[:8]) + "\n")
    print(f"traced solve: {(t1 - t0) / 1e6:.1f} ms wall, {sum(v[1] for v in by.values()) / 1e6:.1f} ms in kernels, fine-level SpMV launches {spmv_fine} (iterations ~ {its})\n")
    print("| kernel | level | launches | total ms | ms per iteration |\n|---|---|---|---|---|")
    for (nm, lv), (c, d) in sorted(by.items(), key=lambda kv: -kv[1][1])[:28]:
        print(f"| `{nm}` | {lv} | {c} | {d / 1e6:.2f} | {d / 1e6 / its:.3f} |")
    sys.exit(0)
os.environ.setdefault("MGS_ARENA_GB", "110")
GRAPH = os.environ.get("FGCR_TRACE_GRAPH", "0") == "1"      # replayed cycles as in the product: needs torch loaded first (see below)
if GRAPH:
    import torch
    torch.cuda.init()
import multigridsolver_amd as mg
N = 512; n = N ** 3
ctx = mg.Context(0)
A = ctx.poisson3d(N)
h = mg.Hierarchy(A, 0.6, 1, 1).coarsen(10.0, 2, 8.0, 2500, 32).finalize()
b = ctx.vec(n).rand(seed=0); x = ctx.vec(n)
A.optimize()
if not GRAPH:
    ctx.set_option("graph", 0)                # eager launches: every kernel its own dispatch (and rocprofv3 7.2 crashes in the capture of this solve when torch is not loaded first)
for _ in range(2): h.vcycle(b, x)
SOLVER = os.environ.get("FGCR_TRACE_SOLVER", "fgcr")   # "bicgstab": BiCGSTAB + plain V-cycle instead
if SOLVER == "fgcr":
    ctx.set_option("kcycle_energy", 1); h.set_kcycle(4)
    for _ in range(2): h.vcycle(b, x)
solve = (lambda: mg.fgcr(A, x, b, h, 10, 300, 1e-10)) if SOLVER == "fgcr" else (lambda: mg.bicgstab(A, x, b, h, 300, 1e-10))
x.fill(0.0); solve()                           # warm (operands, graph capture)
b.rand(seed=0); x.fill(0.0); b.rand(seed=0)    # marker: rand_kernel launches
st, it, tol = solve()
ctx.sync()
print("status", st, "iterations", it, "tol", tol, flush=True)
ctx.close()
