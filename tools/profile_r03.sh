#!/bin/bash
# rocprofv3 passes behind the numbers bench.py prints (run on the GPU box; raw CSVs land in gpurun_out/prof_r03/, summarise them
# afterwards with tools/profile_summary.py, which writes profiles/r03/*.json together with the kernel-source hash).
# Kernel trace and every --pmc set are SEPARATE runs (counters never share a run with a trace domain other than the kernel trace).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r03
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
step() { local t=$1 log=$2; shift 2; timeout -k 10 $t "$@" > $log 2> $log.err; local rc=$?; echo "rc=$rc  ($*)" | cut -c1-200; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIME LIMIT: stopping"; exit 1; fi; return 0; }
BENCH="python3 $R/bench.py --cpu-seconds 0"
step 400 $O/stats_bench.json rocprofv3 --kernel-trace --stats -d $O/stats -o step --output-format csv -- $BENCH --steps 100 --warmup 20
step 400 $O/valu_bench.json rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS -d $O/pmc_valu -o valu --output-format csv -- $BENCH --steps 3 --warmup 1
step 400 $O/wait_bench.json rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_wait -o wait --output-format csv -- $BENCH --steps 3 --warmup 1
step 400 $O/fetch_bench.json rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o fetch --output-format csv -- $BENCH --steps 3 --warmup 1
step 400 $O/write_bench.json rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o write --output-format csv -- $BENCH --steps 3 --warmup 1
# operator kernels (the reference's own hot path): kernel trace of tools/time_ops.py
step 400 $O/ops_time.log rocprofv3 --kernel-trace --stats -d $O/ops -o ops --output-format csv -- python3 $R/tools/time_ops.py 4096
find $O -name "*.csv" | head -30
# keep the merge-back small: the per-dispatch trace of the 100-step run is the only big file
find $O -name "*kernel_trace.csv" -size +20M -delete
exit 0
