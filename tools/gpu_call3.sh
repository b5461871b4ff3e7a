#!/bin/bash
# round-2 GPU call 3: whole parity suite with gather-on-read + lazy halos; configs; bench
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out
mkdir -p $O
step() { local t=$1 out=$2; shift 2; timeout -k 10 $t "$@" > $out 2> $out.err; local rc=$?; echo "rc=$rc  ($*)"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIME LIMIT: stopping the call"; exit 1; fi; return 0; }
echo "== pytest -m gpu"; step 900 $O/pytest_gpu3.log python -m pytest tests -m gpu -q; tail -8 $O/pytest_gpu3.log
echo "== configs"; step 600 $O/configs3.log python tools/run_configs.py --out $O/configs3.json; cut -c1-330 $O/configs3.log
echo "== bench"; step 300 $O/bench3.json python bench.py; cut -c1-300 $O/bench3.json
echo "== bench force-ring"; step 300 $O/bench3_ring.json python bench.py --force-ring --cpu-seconds 0; cut -c1-300 $O/bench3_ring.json
exit 0
