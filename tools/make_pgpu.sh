#!/bin/bash
# GPU box: run the setup driver on the bundled operators and bring the P files back (gpurun_out/pgpu/)
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
T=$(mktemp -d); mkdir -p $T/matrices $T/src/common $ROOT/gpurun_out/pgpu
cp $ROOT/tests/golden/inputs/CSky3d10.mtx $ROOT/tests/golden/inputs/CSky2d20.mtx $T/matrices/
gunzip -c $ROOT/tests/golden/inputs/CSky3d30.mtx.gz > $T/matrices/CSky3d30.mtx
python3 $ROOT/tools/write_poisson_mtx.py 100 $T/matrices/poisson10000.mtx
cd $T/src/common
for m in CSky3d30 CSky3d10 CSky2d20 poisson10000; do
  $ROOT/multigridsolver_amd/cpp/mgs_agmg $m 10 2 8 > $ROOT/gpurun_out/pgpu/$m.log 2>&1
  cp $T/matrices/${m}promatrix_gpu.mtx $ROOT/gpurun_out/pgpu/
  $ROOT/multigridsolver_amd/cpp/mgs_bicg $m gpu >> $ROOT/gpurun_out/pgpu/$m.log 2>&1
done
ls -la $ROOT/gpurun_out/pgpu/
