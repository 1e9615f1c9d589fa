#!/bin/bash
# Runs on the GPU box: wave-occupancy / wait-state / LDS / L2 counters of the fine-level kernels (prof_workload.py), one rocprofv3
# --pmc pass per counter group (SQ has 8 slots per pass, TCC 4; no trace domain in a counter pass).  usage: run_pmc_sq.sh [grid] [extra MGS_OPTIONS]
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_sq
GRID=${1:-512}
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
export MGS_ARENA_GB=${MGS_ARENA_GB:-100}     # the arena bench.py reserves by default: same placement policy in the profiled workload
[ -n "$2" ] && export MGS_OPTIONS="$2"
pass() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $ROOT/tools/prof_workload.py $GRID 2 > $OUT/$name.log 2>&1; echo "pass $name done"; }
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
pass sq3 SQ_LEVEL_WAVES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum
pass tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum
du -sh $OUT
