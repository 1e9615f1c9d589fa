#!/usr/bin/env python3
"""Condense gpurun_out/prof (rocprofv3 --kernel-trace --stats pass + separate --pmc FETCH_SIZE /
WRITE_SIZE passes of tools/prof_workload.py) into a small tracked summary under profiles/.
usage: summarize_prof.py <tag> [grid]"""
import collections
import csv
import glob
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(REPO, "gpurun_out", "prof")
tag = sys.argv[1]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 512
n = N ** 3
nnz = 7 * n - 6 * N * N
# level-1 rows of the device-built hierarchy (aggregates of 4, a few odd shapes at the domain boundary): n/4 to within 1e-5 —
# enters only the 8·n_c term of the post pass and the restriction terms of the grouped pre pass
nc1 = n // 4
# grouped_pre = pre pass + restriction (SURVEY §8d-d3 formulas) in one kernel
ALG = {"fused_pre": 12 * nnz + 28 * n + 4, "fused_post": 12 * nnz + 40 * n + 8 * nc1 + 4,
       "grouped_pre": (12 * nnz + 28 * n + 4) + (4 * (nc1 + 1) + 12 * n + 8 * nc1),
       "spmv": 12 * nnz + 20 * n + 4, "residual": 12 * nnz + 28 * n + 4, "jacobi": 12 * nnz + 36 * n + 4,
       "axpby(calibration)": 32 * n}


def newest(pattern):
    return max(glob.glob(pattern), key=os.path.getmtime)


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0]


def classify(name):
    s = short(name)
    for pre in ("csr_rowblock_kernel<", "csr_rowblock_slice_kernel<", "csr_rowblock_coded_kernel<"):
        if s.startswith(pre):
            return {"0": "spmv", "1": "residual", "2": "jacobi", "5": "fused_post"}.get(s[len(pre)], None)
    if s.startswith("csr_group_pre_kernel<"): return "grouped_pre"
    if s.startswith("csr_rowblock_fused_kernel<3"): return "fused_pre"
    if s.startswith("csr_rowblock_fused_kernel<4") or s.startswith("csr_rowblock_fused_kernel<5"): return "fused_post"
    if s.startswith("axpbypcz_kernel"): return "axpby(calibration)"
    return None


stats = list(csv.DictReader(open(newest(os.path.join(src, "trace", "*", "*_kernel_stats.csv")))))
trace = list(csv.DictReader(open(newest(os.path.join(src, "trace", "*", "*_kernel_trace.csv")))))
fine = collections.defaultdict(list)   # fine-level dispatches only (grid covers n rows)
for r in trace:
    k = classify(r["Kernel_Name"])
    if k and (int(r["Grid_Size_X"]) >= (n // 8 if k == "grouped_pre" else n) or k.startswith("axpby")):
        fine[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))


def pmc(sub, counter):
    out = collections.defaultdict(list)
    for r in csv.DictReader(open(newest(os.path.join(src, sub, "*", "*_counter_collection.csv")))):
        k = classify(r["Kernel_Name"])
        if k and r["Counter_Name"] == counter and (int(r["Grid_Size"]) >= (n // 8 if k == "grouped_pre" else n) or k.startswith("axpby")):
            out[k].append(float(r["Counter_Value"]))
    return {k: sorted(v)[len(v) // 2] for k, v in out.items()}


fetch, write = pmc("pmc_fetch", "FETCH_SIZE"), pmc("pmc_write", "WRITE_SIZE")
# calibration (MI355X_MICROARCH.md §HBM): z = 2x+3b+0.5z reads 24n bytes, writes 8n bytes
cal_read = 24 * n / (fetch["axpby(calibration)"] * 1024)
cal_write = 8 * n / (write["axpby(calibration)"] * 1024)
rows = []
for k in [k for k in ["spmv", "residual", "jacobi", "fused_pre", "grouped_pre", "fused_post", "axpby(calibration)"] if k in fine and k in fetch and k in write]:
    d = sorted(fine[k]); med = d[len(d) // 2]; avg = sum(d) / len(d)
    fb = fetch[k] * 1024 * cal_read; wb = write[k] * 1024 * cal_write
    rows.append({"kernel": k, "launches": len(d), "avg_us": avg / 1e3, "median_us": med / 1e3, "algorithmic_bytes": ALG[k],
                 "algorithmic_GBps_at_avg": ALG[k] / avg, "FETCH_SIZE_KB": fetch[k], "WRITE_SIZE_KB": write[k],
                 "hbm_read_bytes_corrected": fb, "hbm_write_bytes_corrected": wb, "traffic_bytes": fb + wb,
                 "traffic_over_algorithmic": (fb + wb) / ALG[k], "fabric_GBps_at_avg": (fb + wb) / avg})


def last_cycle(rows, name_key, grid_key):
    """rows of the final V-cycle of the workload: from the last fine-level dispatch of the cycle's first kernel (grouped pre pass,
    else the fused pre pass / residual on the fine level) to the end of the run"""
    first = None
    for i, r in enumerate(rows):
        k = classify(r[name_key])
        if k in ("grouped_pre", "fused_pre") and int(r[grid_key]) >= n // 8:
            first = i
    return rows[first:] if first is not None else []


def cycle_counter(sub, counter):
    rows = [r for r in csv.DictReader(open(newest(os.path.join(src, sub, "*", "*_counter_collection.csv")))) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    cyc = last_cycle(rows, "Kernel_Name", "Grid_Size")
    return sum(float(r["Counter_Value"]) for r in cyc), len(cyc)


trace.sort(key=lambda r: int(r["Dispatch_Id"]))
cyc_t = last_cycle(trace, "Kernel_Name", "Grid_Size_X")
cyc_fetch, nd_f = cycle_counter("pmc_fetch", "FETCH_SIZE")
cyc_write, nd_w = cycle_counter("pmc_write", "WRITE_SIZE")
vcycle = None
if cyc_t and nd_f == len(cyc_t) and nd_w == len(cyc_t):
    rd, wr = cyc_fetch * 1024 * cal_read, cyc_write * 1024 * cal_write
    vcycle = {"dispatches": len(cyc_t), "kernel_ms": sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in cyc_t) / 1e6,
              "wall_ms": (int(cyc_t[-1]["End_Timestamp"]) - int(cyc_t[0]["Start_Timestamp"])) / 1e6,
              "hbm_read_bytes_corrected": rd, "hbm_write_bytes_corrected": wr, "traffic_bytes": rd + wr,
              "note": "sum over every dispatch of the last V-cycle of the workload (eager launches), FETCH_SIZE/WRITE_SIZE passes corrected like the per-kernel rows"}
import subprocess
try:      # the commit whose build was profiled (the summary is written in the build container, right after the GPU call that ran this snapshot)
    code_commit = subprocess.run(["git", "-C", REPO, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
    dirty = bool(subprocess.run(["git", "-C", REPO, "status", "--porcelain", "--", "multigridsolver_amd", "include", "bench.py", "tools/prof_workload.py"], capture_output=True, text=True).stdout.strip())
except Exception:  # noqa: BLE001
    code_commit, dirty = None, None
import hashlib
# identity of the row-block kernels' source: bench.py applies this summary's bytes only to a build of the same text (the file as it is in the
# working tree NOW — summarise right after the profiled run, before editing kernels)
ksha = hashlib.sha256(open(os.path.join(REPO, "multigridsolver_amd", "csrc", "kernels_spmv.hip"), "rb").read()).hexdigest()[:16]
summary = {"tag": tag, "grid": N, "rows": n, "nnz": nnz, "code_commit": code_commit, "code_dirty": dirty, "kernel_source_sha": ksha, "vcycle": vcycle, "tool": "rocprofv3 (ROCm 7.2), tools/run_rocprof.sh, tools/prof_workload.py",
           "fetch_size_correction": cal_read, "write_size_correction": cal_write,
           "note": "FETCH_SIZE on gfx950 reports half the bytes of this access pattern (8-byte lanes): calibrated on axpby's known 16n read bytes; "
                   "counters come from L2's fabric-side requests, so Infinity-Cache hits are included (traffic >= HBM bytes). "
                   "Kernel times under the profiler are ~5-15% longer than the un-profiled HIP-event times bench.py reports.",
           "fine_level_kernels": rows,
           "top_kernels_by_total_time": [{"name": short(r["Name"]), "calls": int(r["Calls"]), "total_ms": int(r["TotalDurationNs"]) / 1e6,
                                          "avg_us": float(r["AverageNs"]) / 1e3, "pct": float(r["Percentage"])} for r in stats[:16]]}
os.makedirs(os.path.join(REPO, "profiles"), exist_ok=True)
json.dump(summary, open(os.path.join(REPO, "profiles", f"{tag}_summary.json"), "w"), indent=1)
with open(os.path.join(REPO, "profiles", f"{tag}_summary.md"), "w") as f:
    f.write(f"# rocprofv3 summary `{tag}` — {N}^3 7-pt Poisson ({n} rows, {nnz} nnz), 1x MI355X\n\n")
    f.write(summary["note"] + "\n\n")
    f.write(f"FETCH_SIZE correction x{cal_read:.3f}, WRITE_SIZE correction x{cal_write:.3f} (calibrated on axpby).\n\n")
    f.write("| fine-level kernel | launches | avg us | algorithmic GB | alg. GB/s | traffic GB (PMC, corrected) | traffic/alg | fabric GB/s |\n|---|---|---|---|---|---|---|---|\n")
    for r in rows:
        f.write(f"| {r['kernel']} | {r['launches']} | {r['avg_us']:.1f} | {r['algorithmic_bytes'] / 1e9:.3f} | {r['algorithmic_GBps_at_avg']:.0f} | "
                f"{r['traffic_bytes'] / 1e9:.3f} | {r['traffic_over_algorithmic']:.3f} | {r['fabric_GBps_at_avg']:.0f} |\n")
    if vcycle:
        f.write(f"\nOne V-cycle (last cycle of the workload, {vcycle['dispatches']} dispatches): kernel time {vcycle['kernel_ms']:.3f} ms, wall {vcycle['wall_ms']:.3f} ms, "
                f"PMC traffic {vcycle['traffic_bytes'] / 1e9:.3f} GB (read {vcycle['hbm_read_bytes_corrected'] / 1e9:.3f} + write {vcycle['hbm_write_bytes_corrected'] / 1e9:.3f}) "
                f"= {vcycle['traffic_bytes'] / vcycle['kernel_ms'] / 1e6:.0f} GB/s over its kernel time.\n")
    f.write("\n| kernel (whole workload incl. setup) | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
    for r in summary["top_kernels_by_total_time"]:
        f.write(f"| {r['name']} | {r['calls']} | {r['total_ms']:.2f} | {r['avg_us']:.1f} | {r['pct']:.2f} |\n")
# keep the raw stats CSV too (small)
import shutil
shutil.copy(newest(os.path.join(src, "trace", "*", "*_kernel_stats.csv")), os.path.join(REPO, "profiles", f"{tag}_kernel_stats.csv"))
print(open(os.path.join(REPO, "profiles", f"{tag}_summary.md")).read())
