#!/bin/bash
# one GPU call: DP test on a shared GPU, 2-rank bench rehearsal (gloo), other workloads, fresh kernel profile
set -o pipefail
mkdir -p gpurun_out/sa
timeout -k 10 300 python -u -m pytest tests/test_dp_gpu.py -x -q --timeout 240 --timeout-method thread > gpurun_out/sa/dp_test.log 2>&1 || { tail -30 gpurun_out/sa/dp_test.log; exit 1; }
tail -2 gpurun_out/sa/dp_test.log
true
true
timeout -k 10 500 python tools/bench_configs.py > gpurun_out/sa/configs.jsonl 2> gpurun_out/sa/configs.err || { tail -30 gpurun_out/sa/configs.err; exit 1; }
cat gpurun_out/sa/configs.jsonl
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/sa/prof -o r1c -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/sa/prof_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/sa/prof_err.log || { tail -20 $GRAFT_REPO_ROOT/gpurun_out/sa/prof_err.log; exit 1; }
cat $GRAFT_REPO_ROOT/gpurun_out/sa/prof_bench.json
ls $GRAFT_REPO_ROOT/gpurun_out/sa/prof | head
