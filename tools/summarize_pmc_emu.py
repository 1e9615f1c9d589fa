#!/usr/bin/env python3
"""profiles/<tag>_rank8_traffic.md from tools/run_pmc_emu.sh: FETCH_SIZE / WRITE_SIZE of every dispatch of the emulated rank's last cycle
(corrections of profiles/<tag>_summary.json), beside the one-GPU cycle's figure divided by the rank count.  usage: summarize_pmc_emu.py tag [ranks=8]"""
import csv, json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]; ranks = int(sys.argv[2]) if len(sys.argv) > 2 else 8
src = os.path.join(REPO, "gpurun_out", "pmc_emu")
S = json.load(open(os.path.join(REPO, "profiles", f"{tag}_summary.json")))
cf, cw = S["fetch_size_correction"], S["write_size_correction"]
def short(n): return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
def cycle(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    post = [i for i, r in enumerate(rows) if "kernel<5" in short(r["Kernel_Name"])]
    big = max(int(rows[i]["Grid_Size"]) for i in post)
    ends = [i for i in post if int(rows[i]["Grid_Size"]) == big]
    return rows[ends[-2] + 1: ends[-1] + 1]
f = cycle(os.path.join(src, "fetch_tail.csv"), "FETCH_SIZE"); w = cycle(os.path.join(src, "write_tail.csv"), "WRITE_SIZE")
assert len(f) == len(w) and all(short(a["Kernel_Name"]) == short(b["Kernel_Name"]) for a, b in zip(f, w)), (len(f), len(w))
rd = [float(r["Counter_Value"]) * 1024 * cf for r in f]; wr = [float(r["Counter_Value"]) * 1024 * cw for r in w]
tot = sum(rd) + sum(wr); one = S["vcycle"]["traffic_bytes"]
with open(os.path.join(REPO, "profiles", f"{tag}_rank8_traffic.md"), "w") as o:
    o.write(f"# HBM traffic of one cycle of the emulated middle rank of {ranks} (`tools/run_pmc_emu.sh`, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)\n\n")
    o.write(f"Corrections as in `{tag}_summary.json` (FETCH_SIZE x{cf:.3f}, WRITE_SIZE x{cw:.3f}).  {len(f)} dispatches: read {sum(rd) / 1e9:.3f} GB + write {sum(wr) / 1e9:.3f} GB = "
            f"**{tot / 1e9:.3f} GB** per cycle and rank; the one-GPU cycle's {one / 1e9:.2f} GB / {ranks} = {one / ranks / 1e9:.3f} GB "
            f"(ratio {tot / (one / ranks):.3f}: halo payloads, the replicated tail and RCCL's own buffers on top of the rank's share).\n\n")
    o.write("| # | kernel | grid | read MB | write MB |\n|---|---|---|---|---|\n")
    for i, (a, r_, w_) in enumerate(zip(f, rd, wr)):
        o.write(f"| {i} | `{short(a['Kernel_Name'])[:60]}` | {a['Grid_Size']} | {r_ / 1e6:.2f} | {w_ / 1e6:.2f} |\n")
print(open(os.path.join(REPO, "profiles", f"{tag}_rank8_traffic.md")).read()[:1500])
