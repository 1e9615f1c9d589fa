#!/bin/bash
# kernel-trace pass only (no PMC) of tools/prof_workload.py; usage: run_trace_only.sh <outname> [grid]   (MGS_OPTIONS passes through)
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/trace_$1
GRID=${2:-512}
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 $ROOT/tools/prof_workload.py $GRID 2 > $OUT/trace.log 2>&1
F=$(ls $OUT/t/*/*_kernel_trace.csv | head -1); head -1 $F > $OUT/cycle_tail.csv; tail -150 $F >> $OUT/cycle_tail.csv
rm -rf $OUT/t
python3 $ROOT/tools/cycle_trace.py $OUT/cycle_tail.csv > $OUT/cycle.md
