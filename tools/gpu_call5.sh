#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out
mkdir -p $O
step() { local t=$1 out=$2; shift 2; timeout -k 10 $t "$@" > $out 2> $out.err; local rc=$?; echo "rc=$rc  ($*)"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIME LIMIT: stopping the call"; exit 1; fi; return 0; }
echo "== lorentz + diagnostics tests, LDS kernel"; step 600 $O/pytest_gpu5a.log python -m pytest tests/test_lorentz_gpu.py tests/test_diagnostics_gpu.py -m gpu -q; tail -3 $O/pytest_gpu5a.log
echo "== lorentz tests, DPP kernel"; SWMHD_OP_DPP=1 step 600 $O/pytest_gpu5b.log python -m pytest tests/test_lorentz_gpu.py tests/test_fullsize_gpu.py -m gpu -q -k "not precision"; tail -3 $O/pytest_gpu5b.log
for i in 1 2; do
echo "== time ops LDS ($i)"; step 200 $O/ops_lds_$i.log python tools/time_ops.py 4096; grep jacobian $O/ops_lds_$i.log
echo "== time ops DPP ($i)"; SWMHD_OP_DPP=1 step 200 $O/ops_dpp_$i.log python tools/time_ops.py 4096; grep jacobian $O/ops_dpp_$i.log
done
echo "== time ops 16384x2048-ish: N=8192"; step 200 $O/ops_lds_8k.log python tools/time_ops.py 8192; grep "jacobian   march" $O/ops_lds_8k.log
SWMHD_OP_DPP=1 step 200 $O/ops_dpp_8k.log python tools/time_ops.py 8192; grep "jacobian   march" $O/ops_dpp_8k.log
exit 0
