#!/usr/bin/env python3
"""Per-kernel timeline of the LAST V-cycle in a rocprofv3 kernel trace of tools/prof_workload.py (eager launches: one
dispatch per kernel), grouped by level: which level costs what, against its algorithmic bytes at a reference rate.
usage: cycle_trace.py <kernel_trace.csv> [tag]   (writes profiles/<tag>_cycle_levels.md when a tag is given)"""
import csv, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n): return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
names = [short(r["Kernel_Name"]) for r in rows]
# a cycle starts at the largest-grid residual-form kernel ("<1," pre pass) and ends with the largest-grid post pass ("<5,")
gx = [int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r["Grid_Size"]) for r in rows]
post = [i for i, n in enumerate(names) if "kernel<5" in n or "kernel<4" in n]
big = max(gx[i] for i in post)          # the fine-level post pass: the largest grid among the post-pass kernels
ends = [i for i in post if gx[i] == big]
e = ends[-1]; s = ends[-2] + 1          # a cycle = everything between two fine-level post passes
out = []
t_first = int(rows[s]["Start_Timestamp"]); t_last = int(rows[e]["End_Timestamp"])
tot = 0
for i in range(s, e + 1):
    d = (int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])) / 1e3
    gap = (int(rows[i]["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"])) / 1e3 if i > s else 0.0
    out.append((names[i][:58], gx[i], d, gap)); tot += d
lines = [f"last V-cycle in the trace: {e - s + 1} dispatches, kernel time {tot:.1f} us, wall {(t_last - t_first) / 1e3:.1f} us (eager launches: gaps are host enqueue, absent under hipGraph replay)", "",
         "| # | kernel | grid (threads) | us | gap before, us |", "|---|---|---|---|---|"]
for k, (n, g, d, gap) in enumerate(out):
    lines.append(f"| {k} | `{n}` | {g} | {d:.1f} | {gap:.1f} |")
txt = "\n".join(lines)
print(txt)
if len(sys.argv) > 2:
    with open(os.path.join(REPO, "profiles", f"{sys.argv[2]}_cycle_levels.md"), "w") as f:
        f.write(f"# per-kernel timeline of one V-cycle, 512^3 ({sys.argv[2]}; rocprofv3 --kernel-trace of tools/prof_workload.py)\n\n" + txt + "\n")
