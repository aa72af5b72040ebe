#!/bin/bash
# rocprofv3 kernel trace of bench.py (ViT-B/16 224) and of the 384-px workload; CSV stats land under gpurun_out/<tag>/
# Per-kernel durations are only meaningful when launches do not share the chip, so the weight-gradient side stream is
# switched off here (bench.py does the same in its event-bracketed steps); tools/gpu_trace.sh looks at the overlapped run.
set -o pipefail
export VITK_WGRAD_STREAM=0
tag=${1:-prof}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/b224 -o b224 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/b224_bench.json 2> $out/b224_err.log || { tail -20 $out/b224_err.log; exit 1; }
cat $out/b224_bench.json
if [ "$2" = "384" ]; then
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/b384 -o b384 -- python3 $GRAFT_REPO_ROOT/tools/bench_configs.py --only vit_b_384 --steps 5 --warmup 2 > $out/b384_bench.json 2> $out/b384_err.log || { tail -20 $out/b384_err.log; exit 1; }
cat $out/b384_bench.json
fi
find $out -name "*kernel_stats.csv"
