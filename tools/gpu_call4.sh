#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out
mkdir -p $O
step() { local t=$1 out=$2; shift 2; timeout -k 10 $t "$@" > $out 2> $out.err; local rc=$?; echo "rc=$rc  ($*)"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIME LIMIT: stopping the call"; exit 1; fi; return 0; }
echo "== pytest bounded"; step 600 $O/pytest_gpu4.log python -m pytest tests/test_bounded_gpu.py -m gpu -q -x; tail -30 $O/pytest_gpu4.log
echo "== pytest all"; step 900 $O/pytest_gpu4b.log python -m pytest tests -m gpu -q; tail -6 $O/pytest_gpu4b.log
exit 0
