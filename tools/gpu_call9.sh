#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out
mkdir -p $O
step() { local t=$1 out=$2; shift 2; timeout -k 10 $t "$@" > $out 2> $out.err; local rc=$?; echo "rc=$rc  ($*)"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIME LIMIT: stopping the call"; exit 1; fi; return 0; }
echo "== packed fp32 tests"; step 600 $O/pytest_gpu9.log python -m pytest tests/test_model_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x -k "packed or fp32 or f32 or precision"; tail -15 $O/pytest_gpu9.log
echo "== configs f32 packed"; step 300 $O/configs9_pk.log python tools/run_configs.py --only config5 --out $O/configs9_pk.json; cut -c1-330 $O/configs9_pk.log
echo "== configs f32 unpacked"; SWMHD_T_NOPK=1 step 300 $O/configs9_nopk.log python tools/run_configs.py --only f32 --out $O/configs9_nopk.json; cut -c1-330 $O/configs9_nopk.log
exit 0
