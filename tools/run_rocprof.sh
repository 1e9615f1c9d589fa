#!/bin/bash
# Runs on the GPU box: kernel-trace stats + separate PMC passes (FETCH_SIZE / WRITE_SIZE cannot
# share a pass on gfx950: TCC has 4 slots, FETCH_SIZE costs 3 and WRITE_SIZE 2).
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof
GRID=${1:-512}
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
export MGS_ARENA_GB=${MGS_ARENA_GB:-100}     # the arena bench.py reserves by default: same placement policy in the profiled workload
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/tools/prof_workload.py $GRID 5 > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/tools/prof_workload.py $GRID 2 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/tools/prof_workload.py $GRID 2 > $OUT/pmc_write.log 2>&1
find $OUT -name "*.csv" | head -20
# keep only compact files in gpurun_out (the trace CSV of 512^3 is small: few hundred dispatches)
du -sh $OUT
# the last ~150 dispatches (the final V-cycle) for tools/cycle_trace.py
F=$(ls $OUT/trace/*/*_kernel_trace.csv | head -1); head -1 $F > $OUT/cycle_tail.csv; tail -150 $F >> $OUT/cycle_tail.csv
