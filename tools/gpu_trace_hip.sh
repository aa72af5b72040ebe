#!/bin/bash
# kernel trace + HIP API trace of un-bracketed bench.py steps: which host call coincides with a stalled launch?
set -o pipefail
tag=${1:-trace_hip}
steps=${2:-20}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --hip-trace --output-format csv -d $out/t -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps $steps --warmup 3 --prof-every 0 --no-cpu-baseline > $out/bench.json 2> $out/err.log || { tail -20 $out/err.log; exit 1; }
cut -c1-160 $out/bench.json
ls -la $out/t
