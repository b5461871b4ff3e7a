#!/bin/bash
# round-2 GPU call 2: parity of the reworked conservative kernel; A/B 3 vs 2 workgroups per CU; slabs with the prologue
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out
mkdir -p $O
step() { local t=$1 out=$2; shift 2; timeout -k 10 $t "$@" > $out 2> $out.err; local rc=$?; echo "rc=$rc  ($*)"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIME LIMIT: stopping the call"; exit 1; fi; return 0; }
echo "== pytest model + fullsize"; step 900 $O/pytest_gpu2.log python -m pytest tests/test_model_gpu.py tests/test_fullsize_gpu.py tests/test_lorentz_gpu.py -m gpu -x -q; tail -4 $O/pytest_gpu2.log
echo "== configs (cons 3 wg/cu)"; step 600 $O/configs_c3.log python tools/run_configs.py --out $O/configs_c3.json; cut -c1-330 $O/configs_c3.log
echo "== configs (cons 2 wg/cu)"; SWMHD_LIBRARY=$R/tools/libswmhd_cons2.so step 300 $O/configs_c2.log python tools/run_configs.py --only config2,config4 --out $O/configs_c2.json; cut -c1-330 $O/configs_c2.log
echo "== bench conservative (3 wg/cu)"; step 300 $O/bench_cons3.json python bench.py --formulation Conservative --cpu-seconds 0; cut -c1-260 $O/bench_cons3.json
echo "== bench conservative (2 wg/cu)"; SWMHD_LIBRARY=$R/tools/libswmhd_cons2.so step 300 $O/bench_cons2.json python bench.py --formulation Conservative --cpu-seconds 0; cut -c1-260 $O/bench_cons2.json
exit 0
