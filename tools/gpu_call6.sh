#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out
mkdir -p $O
step() { local t=$1 out=$2; shift 2; timeout -k 10 $t "$@" > $out 2> $out.err; local rc=$?; echo "rc=$rc  ($*)"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIME LIMIT: stopping the call"; exit 1; fi; return 0; }
echo "== model tests (rotation build)"; step 600 $O/pytest_gpu6.log python -m pytest tests/test_model_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x; tail -3 $O/pytest_gpu6.log
for i in 1 2; do
echo "== bench rotation ($i)"; step 300 $O/bench6_rot_$i.json python bench.py --cpu-seconds 0; python -c "import json;d=json.load(open('$O/bench6_rot_$i.json'));print(d['value'],d['ms_per_step'],d['roofline']['avg_launch_ms'])"
echo "== bench no rotation ($i)"; SWMHD_LIBRARY=$R/tools/libswmhd_norot.so step 300 $O/bench6_norot_$i.json python bench.py --cpu-seconds 0; python -c "import json;d=json.load(open('$O/bench6_norot_$i.json'));print(d['value'],d['ms_per_step'],d['roofline']['avg_launch_ms'])"
done
echo "== stage times"; step 300 $O/stage6_rot.log python tools/stage_times.py; cat $O/stage6_rot.log | tail -5
SWMHD_LIBRARY=$R/tools/libswmhd_norot.so step 300 $O/stage6_norot.log python tools/stage_times.py; cat $O/stage6_norot.log | tail -5
exit 0
