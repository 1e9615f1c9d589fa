#!/bin/bash
# kernel trace of tools/ab_group2.py (several grouping variants in ONE process): per-variant duration of the fine-level grouped kernel
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/trace_ab
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 $ROOT/tools/studies_r1_r3/ab_group2.py "$@" > $OUT/log.txt 2>&1
F=$(ls $OUT/t/*/*_kernel_trace.csv | head -1)
python3 - "$F" <<'PY'
import csv, sys, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    g = int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r["Grid_Size"])
    if g < 30_000_000: continue
    for key in ("csr_group_pre_kernel", "coded_kernel<1", "coded_kernel<5", "restrict_agg_kernel"):
        if key in n: d[(key, g)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items()):
    v = sorted(v); print(f"{k[0]:26s} grid {k[1]:>10d}: n={len(v):4d} median {v[len(v)//2]:8.1f} us  min {v[0]:8.1f}")
PY
rm -rf $OUT/t
tail -5 $OUT/log.txt
