#!/bin/bash
# kernel tables of the other workloads
set -o pipefail
export VITK_WGRAD_STREAM=0
out=$GRAFT_REPO_ROOT/gpurun_out/s5
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for w in vit_l_224 vit_b_384; do
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/p_$w -o $w -- python3 $GRAFT_REPO_ROOT/tools/bench_configs.py --only $w --steps 5 --warmup 2 > $out/${w}_bench.json 2> $out/${w}_err.log || { tail -20 $out/${w}_err.log; exit 1; }
cat $out/${w}_bench.json
done
