#!/bin/bash
# kernel trace of the emulated middle rank (tools/emulate_rank.py, native transport, cycle replayed from its hipGraph): the last ~120 dispatches
# usage: run_trace_emu.sh <outname> [ranks=8] [grid=512]
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/trace_emu_$1
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
export EMU_ONE=1 EMU_REPS=10
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 $ROOT/tools/emulate_rank.py ${3:-512} ${2:-8} > $OUT/trace.log 2>&1
F=$(ls $OUT/t/*/*_kernel_trace.csv | head -1)
python3 - "$F" > $OUT/cycle.md <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n): return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
names = [short(r["Kernel_Name"]) for r in rows]
# a cycle ends with the fine-level post pass: the largest-grid "<5," kernel
gx = [int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r["Grid_Size"]) for r in rows]
post = [i for i, n in enumerate(names) if "kernel<5" in n]
big = max(gx[i] for i in post)
ends = [i for i in post if gx[i] == big]
e, s = ends[-1], ends[-2] + 1
t0 = int(rows[s]["Start_Timestamp"]); tot = 0.0
print(f"last cycle: {e - s + 1} dispatches, wall {(int(rows[e]['End_Timestamp']) - t0) / 1e3:.1f} us")
print("| # | kernel | grid | start us | us | gap us |\n|---|---|---|---|---|---|")
for i in range(s, e + 1):
    d = (int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])) / 1e3
    gap = (int(rows[i]["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"])) / 1e3 if i > s else 0.0
    print(f"| {i - s} | `{names[i][:64]}` | {gx[i]} | {(int(rows[i]['Start_Timestamp']) - t0) / 1e3:.1f} | {d:.1f} | {gap:.1f} |")
PY
rm -rf $OUT/t
cat $OUT/cycle.md
