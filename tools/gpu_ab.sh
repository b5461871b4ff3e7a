#!/bin/bash
# scratch A/B driver: cons kernel occupancy masks
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out
mkdir -p $O
step() { local t=$1 out=$2; shift 2; timeout -k 10 $t "$@" > $out 2> $out.err; local rc=$?; echo "rc=$rc  ($*)"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIME LIMIT: stopping the call"; exit 1; fi; return 0; }
echo "== model tests"; step 600 $O/pytest_ab.log python -m pytest tests/test_model_gpu.py tests/test_fullsize_gpu.py tests/test_ring_gpu.py -m gpu -q -x; tail -3 $O/pytest_ab.log
for v in default w3_0 w3_160 w3_170; do
  if [ $v = default ]; then unset SWMHD_LIBRARY; else export SWMHD_LIBRARY=$R/tools/libswmhd_$v.so; fi
  echo "== $v"; step 300 $O/ab_bench_$v.json python bench.py --formulation Conservative --cpu-seconds 0; python -c "import json;d=json.load(open('$O/ab_bench_$v.json'));print('4096^2 cons', round(d['value']), d['ms_per_step'])"
  step 300 $O/ab_cfg_$v.log python tools/run_configs.py --only config2,config4; python - <<PY
import json,re
for l in open('$O/ab_cfg_$v.log'):
    m=re.match(r'(\S+) (\{.*\})', l)
    if m:
        d=json.loads(m.group(2)); print('  ', m.group(1), round(d['Mcell_steps_per_s']), 'Mcell-steps/s', round(d['rk3_step_ms'],4), 'ms', 'tend', round(d['tendency_kernel_us'],1))
PY
done
exit 0
