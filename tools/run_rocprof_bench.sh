#!/bin/bash
# rocprofv3 --kernel-trace --stats of bench.py itself (GPU box); the summary is made by tools/summarize_bench_prof.py
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_bench
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --no-cpu > $OUT/bench.json 2> $OUT/bench.log
# the full trace is large (BiCGSTAB etc.): keep the stats table and the fine-level rows of the trace only
F=$(ls $OUT/trace/*/*_kernel_trace.csv | head -1)
head -1 $F > $OUT/fine_level_trace.csv
awk -F, '($0 ~ /csr_rowblock/ && $0 ~ /134217728|134218752|134479872/) || ($0 ~ /csr_group_pre_kernel/ && $0 ~ /,3[0-9][0-9][0-9][0-9][0-9][0-9][0-9],/)' $F >> $OUT/fine_level_trace.csv || true
cp $(ls $OUT/trace/*/*_kernel_stats.csv | head -1) $OUT/kernel_stats.csv
rm -rf $OUT/trace
ls -la $OUT
