#!/usr/bin/env python3
"""Condense gpurun_out/pmc_sq (tools/run_pmc_sq.sh) into profiles/<tag>_counters.md: per fine-level kernel the median of every
counter over its fine-level dispatches, plus the derived figures the kernels are judged by (mean waves per SIMD, share of wave
cycles parked in s_waitcnt/barrier, LDS bank-conflict share, L2 hit rate, L1->L2 read latency).
usage: summarize_pmc_sq.py <tag> [grid]"""
import collections
import csv
import glob
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(REPO, "gpurun_out", "pmc_sq")
tag = sys.argv[1]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 512
n = N ** 3


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def classify(name):
    s = short(name)
    if s.startswith("csr_rowblock_coded_kernel<"):
        return {"0": "spmv (coded)", "1": "residual (coded)", "2": "jacobi (coded)", "5": "post pass (coded, on A·P)"}.get(s[len("csr_rowblock_coded_kernel<")])
    if s.startswith("csr_rowblock_slice_kernel<0"): return "spmv (plain CSR)"
    if s.startswith("csr_group_pre_kernel<"): return "grouped pre pass"
    if s.startswith("axpbypcz_kernel"): return "axpbypcz (calibration)"
    return None


vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(src, "*", "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = classify(r["Kernel_Name"])
        if not k:
            continue
        fine = int(r["Grid_Size"]) >= (n // 8 if k == "grouped pre pass" else n) or k.startswith("axpby")
        if fine:
            vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
med = {k: {c: sorted(v)[len(v) // 2] for c, v in d.items()} for k, d in vals.items()}
order = [k for k in ("spmv (coded)", "spmv (plain CSR)", "residual (coded)", "jacobi (coded)", "grouped pre pass", "post pass (coded, on A·P)", "axpbypcz (calibration)") if k in med]
counters = sorted({c for d in med.values() for c in d})


def g(k, c):
    return med[k].get(c)


def ratio(a, b):
    return None if a is None or not b else a / b


derived = {}
for k in order:
    d = {}
    # SQ_* cycle counters tick in quad-cycles per wave resp. per SE; ratios of same-unit counters are what is read here
    d["wave cycles parked (WAIT_ANY / WAVE_CYCLES)"] = ratio(g(k, "SQ_WAIT_ANY"), g(k, "SQ_WAVE_CYCLES"))
    d["issue stalls (WAIT_INST_ANY / WAVE_CYCLES)"] = ratio(g(k, "SQ_WAIT_INST_ANY"), g(k, "SQ_WAVE_CYCLES"))
    d["issuing (ACTIVE_INST_ANY / WAVE_CYCLES)"] = ratio(g(k, "SQ_ACTIVE_INST_ANY"), g(k, "SQ_WAVE_CYCLES"))
    # SQ_LEVEL_WAVES reads 0 on this stack; resident waves follow from the cycle sums: WAVE_CYCLES (summed over waves) / BUSY_CU_CYCLES
    # (summed over SIMDs: a kernel that keeps every slot of a CU full reads 8.0 here, i.e. 8 waves per SIMD = 32 per CU)
    d["mean resident waves per SIMD, of 8 (WAVE_CYCLES / BUSY_CU_CYCLES)"] = ratio(g(k, "SQ_WAVE_CYCLES"), g(k, "SQ_BUSY_CU_CYCLES"))
    d["mean wave lifetime, quad-cycles (WAVE_CYCLES / WAVES)"] = ratio(g(k, "SQ_WAVE_CYCLES"), g(k, "SQ_WAVES"))
    d["LDS bank-conflict share (BANK_CONFLICT / IDX_ACTIVE)"] = ratio(g(k, "SQ_LDS_BANK_CONFLICT"), g(k, "SQ_LDS_IDX_ACTIVE"))
    d["L2 hit rate (HIT / (HIT + MISS))"] = ratio(g(k, "TCC_HIT_sum"), (g(k, "TCC_HIT_sum") or 0) + (g(k, "TCC_MISS_sum") or 0))
    d["L2 tag stall per request (TAG_STALL / REQ)"] = ratio(g(k, "TCC_TAG_STALL_sum"), g(k, "TCC_REQ_sum"))
    d["L1 hit share (1 - TCC_READ_REQ / TOTAL_CACHE_ACCESSES)"] = None if not g(k, "TCP_TOTAL_CACHE_ACCESSES_sum") else 1 - g(k, "TCP_TCC_READ_REQ_sum") / g(k, "TCP_TOTAL_CACHE_ACCESSES_sum")
    d["L1->L2 read latency, cycles (READ_REQ_LATENCY / READ_REQ)"] = ratio(g(k, "TCP_TCC_READ_REQ_LATENCY_sum"), g(k, "TCP_TCC_READ_REQ_sum"))
    d["vector-memory instructions in flight per CU (INST_LEVEL_VMEM / BUSY_CU_CYCLES)"] = ratio(g(k, "SQ_INST_LEVEL_VMEM"), g(k, "SQ_BUSY_CU_CYCLES"))
    d["VMEM read instr per wave"] = ratio(g(k, "SQ_INSTS_VMEM_RD"), g(k, "SQ_WAVES"))
    d["VMEM write instr per wave"] = ratio(g(k, "SQ_INSTS_VMEM_WR"), g(k, "SQ_WAVES"))
    d["LDS instr per wave"] = ratio(g(k, "SQ_INSTS_LDS"), g(k, "SQ_WAVES"))
    d["VALU instr per wave"] = ratio(g(k, "SQ_INSTS_VALU"), g(k, "SQ_WAVES"))
    d["SALU instr per wave"] = ratio(g(k, "SQ_INSTS_SALU"), g(k, "SQ_WAVES"))
    derived[k] = d
os.makedirs(os.path.join(REPO, "profiles"), exist_ok=True)
json.dump({"tag": tag, "grid": N, "median_counters": med, "derived": derived, "dispatches": {k: {c: len(v) for c, v in d.items()} for k, d in vals.items()}},
          open(os.path.join(REPO, "profiles", f"{tag}_counters.json"), "w"), indent=1)
with open(os.path.join(REPO, "profiles", f"{tag}_counters.md"), "w") as f:
    f.write(f"# rocprofv3 --pmc counters `{tag}` — fine-level kernels of the {N}^3 7-pt Poisson operator, 1x MI355X\n\n"
            "One `--pmc` pass per counter group (tools/run_pmc_sq.sh), median over the kernel's fine-level dispatches. SQ cycle counters are in "
            "quad-cycles; only ratios of like counters are interpreted.\n\n## derived\n\n")
    f.write("| figure | " + " | ".join(order) + " |\n|---|" + "---|" * len(order) + "\n")
    for name in next(iter(derived.values())).keys():
        f.write(f"| {name} | " + " | ".join("-" if derived[k][name] is None else f"{derived[k][name]:.4g}" for k in order) + " |\n")
    f.write("\n## raw medians\n\n| counter | " + " | ".join(order) + " |\n|---|" + "---|" * len(order) + "\n")
    for c in counters:
        f.write(f"| {c} | " + " | ".join("-" if g(k, c) is None else f"{g(k, c):.6g}" for k in order) + " |\n")
print(open(os.path.join(REPO, "profiles", f"{tag}_counters.md")).read())
