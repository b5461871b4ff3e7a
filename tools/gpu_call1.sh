#!/bin/bash
# round-2 GPU call 1: parity suite, headline bench, A/B of the prologue / strip width, all configs.
# A step that is killed at its time limit ends the whole call (no further GPU step after a hang).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out
mkdir -p $O
step() {   # step <seconds> <stdout-file> <cmd...>
    local t=$1 out=$2; shift 2
    timeout -k 10 $t "$@" > $out 2> $out.err
    local rc=$?
    echo "rc=$rc  ($*)"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIME LIMIT: stopping the call"; exit 1; fi
    return 0
}
echo "== pytest -m gpu"; step 900 $O/pytest_gpu.log python -m pytest tests -m gpu -x -q; tail -5 $O/pytest_gpu.log
echo "== bench default"; step 300 $O/bench.json python bench.py; cut -c1-600 $O/bench.json
echo "== bench old warm-up rows (NT 256)"; SWMHD_T_WARMUP_ROWS=1 SWMHD_T_NT=256 step 300 $O/bench_warmrows.json python bench.py --cpu-seconds 0; cut -c1-300 $O/bench_warmrows.json
echo "== bench prologue NT 256"; SWMHD_T_NT=256 step 300 $O/bench_nt256.json python bench.py --cpu-seconds 0; cut -c1-300 $O/bench_nt256.json
echo "== configs"; step 600 $O/configs.log python tools/run_configs.py --out $O/configs.json; cut -c1-420 $O/configs.log
echo "== configs, old warm-up"; SWMHD_T_WARMUP_ROWS=1 SWMHD_T_NT=256 step 300 $O/configs_warmrows.log python tools/run_configs.py --only config3 --out $O/configs_warmrows.json; cut -c1-420 $O/configs_warmrows.log
exit 0
