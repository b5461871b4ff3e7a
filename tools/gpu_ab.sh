#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out
mkdir -p $O
step() { local t=$1 out=$2; shift 2; timeout -k 10 $t "$@" > $out 2> $out.err; local rc=$?; echo "rc=$rc  ($*)"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIME LIMIT: stopping the call"; exit 1; fi; return 0; }
echo "== model tests"; step 600 $O/pytest_ab.log python -m pytest tests/test_model_gpu.py -m gpu -q -x; tail -3 $O/pytest_ab.log
for i in 1 2 3; do
echo "== stage times default ($i)"; step 300 $O/st_a_$i.log python tools/stage_times.py; tail -1 $O/st_a_$i.log
echo "== stage times VI stage 3 on the stage-2 variant ($i)"; SWMHD_T_VI_STAGE3_AS_2=1 step 300 $O/st_b_$i.log python tools/stage_times.py; tail -1 $O/st_b_$i.log
done
exit 0
