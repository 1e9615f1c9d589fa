#!/bin/bash
# kernel trace of one FGCR(10) + K(4, energy) solve at 512^3 → gpurun_out/trace_fgcr/summary.md   (FGCR_TRACE_GRAPH=1: cycles replayed from their hipGraph, as in the product)
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/trace_fgcr
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 $ROOT/tools/fgcr_trace.py > $OUT/trace.log 2>&1
F=$(ls $OUT/t/*/*_kernel_trace.csv | head -1)
python3 $ROOT/tools/fgcr_trace.py summarize $F > $OUT/summary.md
rm -rf $OUT/t
cat $OUT/summary.md
