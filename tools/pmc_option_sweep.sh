#!/bin/bash
# FETCH_SIZE of the fine-level kernels for several values of one option (separate rocprofv3 --pmc passes of tools/prof_workload.py).
# usage: pmc_option_sweep.sh <option> <value> [value ...]     (grid 512)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OPT=$1; shift
cd /tmp; export TMPDIR=/tmp
for V in "$@"; do
  OUT=$ROOT/gpurun_out/pmc_sweep/${OPT}_$V
  rm -rf $OUT; mkdir -p $OUT
  export MGS_OPTIONS="$OPT=$V"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/p -- python3 $ROOT/tools/prof_workload.py 512 1 > $OUT/log 2>&1 || exit 1
  python3 - "$OUT" "$OPT=$V" <<'PY'
import csv, glob, sys, collections
out, tag = sys.argv[1], sys.argv[2]
n = 512 ** 3
acc = collections.defaultdict(list)
for f in glob.glob(out + "/p/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE" and int(r["Grid_Size"]) >= n // 8:
            acc[r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    v.sort()
    print(f"{tag}: {k[:60]:60s} launches {len(v):2d}  read bytes/row {v[len(v)//2] * 1024 * 2 / n:.2f}")
PY
  rm -rf $OUT/p
done
