#!/bin/bash
# Everything under profiles/r03 that is not a rocprofv3 pass (those: tools/profile_r03.sh): the bench lines of the workloads DESIGN.md
# quotes, every single-GPU BASELINE configuration, and the ring-of-one rehearsals.  Run on the GPU box; files land in gpurun_out/r03/collect.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03/collect
mkdir -p $O
cd $R
step() { local t=$1 out=$2; shift 2; timeout -k 10 $t "$@" > $out 2> $out.err; local rc=$?; echo "rc=$rc  ($*)" | cut -c1-160; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIME LIMIT: stopping"; exit 1; fi; return 0; }
step 300 $O/bench_default.json python3 bench.py
step 300 $O/bench_cons.json python3 bench.py --formulation Conservative --cpu-seconds 0
step 300 $O/bench_ring.json python3 bench.py --force-ring --cpu-seconds 0
step 300 $O/bench_c4.json python3 bench.py --config 4 --cpu-seconds 0 --steps 30
step 300 $O/bench_c5f64.json python3 bench.py --config 5 --cpu-seconds 0 --steps 50
step 300 $O/bench_c5f32.json python3 bench.py --config 5 --dtype f32 --cpu-seconds 0 --steps 50
step 600 $O/configs.log python3 tools/run_configs.py --out $O/configs.json
step 600 $O/ring_rehearsal.log python3 tools/ring_rehearsal.py --out $O/ring_rehearsal.json
step 600 $O/ring_rehearsal_c4.log python3 tools/ring_rehearsal.py --config 4 --out $O/ring_rehearsal_c4.json
ls -la $O
exit 0
