#!/bin/bash
# HBM-traffic counters of one cycle of the emulated middle rank (tools/emulate_rank.py; native transport, graph replay): one --pmc pass per counter,
# no trace domain in a counter pass.  usage: run_pmc_emu.sh [ranks=8] [grid=512]; summary: tools/summarize_pmc_emu.py
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_emu
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
export EMU_ONE=1 EMU_REPS=6
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $ROOT/tools/emulate_rank.py ${2:-512} ${1:-8} > $OUT/fetch.log 2>&1
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $ROOT/tools/emulate_rank.py ${2:-512} ${1:-8} > $OUT/write.log 2>&1
echo "write pass done"
for p in fetch write; do F=$(ls $OUT/$p/*/*_counter_collection.csv | head -1); head -1 $F > $OUT/${p}_tail.csv; tail -400 $F >> $OUT/${p}_tail.csv; done
rm -rf $OUT/fetch $OUT/write
du -sh $OUT
