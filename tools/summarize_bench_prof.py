#!/usr/bin/env python3
"""profiles/<tag>_bench_prof.md: rocprofv3 --kernel-trace --stats of `python bench.py --no-cpu` itself — the fine-level
SpMV kernel's average duration in the trace next to the HIP-event figure bench.py printed in the same run."""
import csv, json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(REPO, "gpurun_out", "prof_bench"); tag = sys.argv[1]
bench = json.loads(open(os.path.join(src, "bench.json")).read().strip().split("\n")[-1])
rows = list(csv.DictReader(open(os.path.join(src, "fine_level_trace.csv"))))
byk = {}
for r in rows:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    byk.setdefault(name, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
with open(os.path.join(REPO, "profiles", f"{tag}_bench_prof.md"), "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu  ({tag})\n\n")
    f.write(f"bench.py line of this run: value {bench['value']:.2f} V-cycles/s, ms_per_step {bench['ms_per_step']:.3f}, "
            f"roofline.achieved {bench['roofline']['achieved']:.0f} GB/s, ms_per_launch (HIP events) {bench['roofline']['ms_per_launch']:.4f}\n\n")
    f.write("Fine-level (512^3) dispatches of the row-block kernels in the trace:\n\n| kernel | launches | avg us | min us | max us |\n|---|---|---|---|---|\n")
    for k, v in sorted(byk.items()):
        f.write(f"| {k} | {len(v)} | {sum(v)/len(v)/1e3:.1f} | {min(v)/1e3:.1f} | {max(v)/1e3:.1f} |\n")
    # the launches bench.py times for the roofline: the first 3 warm-up + kernel_reps launches of the coded SpMV kernel in the trace (start
    # order); the later ones belong to the Krylov solves (BiCGSTAB's carry the fused dot epilogue and read other vectors)
    spmv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in rows
                  if r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").startswith("csr_rowblock_coded_kernel<0, 7, false, false>"))
    reps = 20
    leg = [d for _, d in spmv[3:3 + reps]]
    if leg:
        f.write(f"\nRoofline leg of the coded SpMV kernel (launches 4..{3 + len(leg)} of it in start order = the {len(leg)} launches between bench.py's HIP events): "
                f"avg {sum(leg)/len(leg)/1e3:.1f} us (min {min(leg)/1e3:.1f}, max {max(leg)/1e3:.1f}) against {bench['roofline']['ms_per_launch'] * 1e3:.1f} us by HIP events; "
                f"the remaining {len(spmv) - 3 - len(leg)} launches are the solvers' products.\n")
    f.write("\n`csr_rowblock_coded_kernel<0, ...>` = SpMV with the pattern-coded index (the roofline kernel; also launched twice per BiCGSTAB iteration), "
            "`<1>` residual (= fused pre pass on the scaled values; in the cycle itself the fine level runs `csr_group_pre_kernel`: pre pass + restriction in one kernel), `<2>` Jacobi, `<5>` fused post pass (template tail `<…, HALO, VAL>`: `VAL = true` rows belong to the opt-in value-pattern leg, not to the headline); `csr_rowblock_slice_kernel<0>` = plain CSR SpMV "
            "(timed once for comparison, `rowcode` = 0), `csr_rowblock_fused_kernel<3>/<4>` = gather forms of the fused passes (unfused/A-B legs only).\n\nTop of the --stats table (all sizes mixed):\n\n| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
    for r in list(csv.DictReader(open(os.path.join(src, "kernel_stats.csv"))))[:12]:
        nm = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        f.write(f"| {nm} | {r['Calls']} | {int(r['TotalDurationNs'])/1e6:.2f} | {float(r['AverageNs'])/1e3:.1f} | {r['Percentage']} |\n")
print(open(os.path.join(REPO, "profiles", f"{tag}_bench_prof.md")).read())
