#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out
mkdir -p $O
step() { local t=$1 out=$2; shift 2; timeout -k 10 $t "$@" > $out 2> $out.err; local rc=$?; echo "rc=$rc  ($*)"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIME LIMIT: stopping the call"; exit 1; fi; return 0; }
echo "== lorentz tests"; step 600 $O/pytest_ab.log python -m pytest tests/test_lorentz_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x -k "not precision"; tail -3 $O/pytest_ab.log
for i in 1 2; do
echo "== new ($i)"; step 300 $O/ops_new_$i.log python tools/time_ops.py 4096; grep march $O/ops_new_$i.log
echo "== prev ($i)"; SWMHD_LIBRARY=$R/tools/libswmhd_prev.so step 300 $O/ops_prev_$i.log python tools/time_ops.py 4096; grep march $O/ops_prev_$i.log
done
echo "== new 1024"; step 300 $O/ops_new_1k.log python tools/time_ops.py 1024; grep march $O/ops_new_1k.log
SWMHD_LIBRARY=$R/tools/libswmhd_prev.so step 300 $O/ops_prev_1k.log python tools/time_ops.py 1024; grep march $O/ops_prev_1k.log
exit 0
