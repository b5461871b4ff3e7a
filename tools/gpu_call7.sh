#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out
mkdir -p $O
step() { local t=$1 out=$2; shift 2; timeout -k 10 $t "$@" > $out 2> $out.err; local rc=$?; echo "rc=$rc  ($*)"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIME LIMIT: stopping the call"; exit 1; fi; return 0; }
echo "== model tests"; step 600 $O/pytest_gpu7.log python -m pytest tests/test_model_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x; tail -3 $O/pytest_gpu7.log
for i in 1 2 3; do
echo "== stage times new ($i)"; step 300 $O/stage7_new_$i.log python tools/stage_times.py; tail -1 $O/stage7_new_$i.log
echo "== stage times prev ($i)"; SWMHD_LIBRARY=$R/tools/libswmhd_prev.so step 300 $O/stage7_prev_$i.log python tools/stage_times.py; tail -1 $O/stage7_prev_$i.log
done
exit 0
