#!/bin/bash
# kernel trace of the emulated middle rank (tools/emulate_rank.py), GPU box
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_emu
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
export EMU_ONE=1 EMU_REPS=10 MASTER_ADDR=127.0.0.1 MASTER_PORT=29933 RANK=0 WORLD_SIZE=1
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/trace -- python3 $ROOT/tools/emulate_rank.py 512 8 100000 > $OUT/emu.log 2>&1
F=$(ls $OUT/trace/*/*_kernel_trace.csv | head -1)
# keep the last 400 dispatches only (a few cycles)
head -1 $F > $OUT/kernel_trace_tail.csv; tail -400 $F >> $OUT/kernel_trace_tail.csv
M=$(ls $OUT/trace/*/*_memory_copy_trace.csv 2>/dev/null | head -1); if [ -n "$M" ]; then head -1 $M > $OUT/memcpy_tail.csv; tail -100 $M >> $OUT/memcpy_tail.csv; fi
rm -rf $OUT/trace
cat $OUT/emu.log | tail -3
