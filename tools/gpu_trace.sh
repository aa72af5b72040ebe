#!/bin/bash
# rocprofv3 kernel trace (start / end of every launch) of a few un-bracketed bench.py steps, to look at how the launches of
# the two backward streams share the chip; CSV lands under gpurun_out/<tag>/.  usage: gpu_trace.sh <tag> [steps]
set -o pipefail
tag=${1:-trace}
steps=${2:-6}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/t -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps $steps --warmup 3 --prof-every 0 --no-cpu-baseline > $out/bench.json 2> $out/err.log || { tail -20 $out/err.log; exit 1; }
cat $out/bench.json | cut -c1-160
find $out -name "*kernel_trace.csv" -size -60M
