#!/bin/bash
# Kernel tables of all headline workloads: rocprofv3 kernel trace of bench.py (ViT-B/16 224) and of the ViT-L/16, 384-px and distillation workloads,
# then every workload once without the profiler.  Per-kernel durations are only meaningful when launches do not share the chip,
# so the weight-gradient side stream is off in the profiled runs (bench.py does the same in its event-bracketed steps).
set -o pipefail
tag=${1:-r3prof}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export VITK_WGRAD_STREAM=0 VITK_KD_OVERLAP=0
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/p_vit_b_224 -o vit_b_224 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/vit_b_224_bench.json 2> $out/vit_b_224_err.log || { tail -20 $out/vit_b_224_err.log; exit 1; }
cat $out/vit_b_224_bench.json
for w in vit_l_224 vit_b_384 deit_b_kd; do
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/p_$w -o $w -- python3 $GRAFT_REPO_ROOT/tools/bench_configs.py --only $w --steps 5 --warmup 2 > $out/${w}_bench.json 2> $out/${w}_err.log || { tail -20 $out/${w}_err.log; exit 1; }
cat $out/${w}_bench.json
done
unset VITK_WGRAD_STREAM VITK_KD_OVERLAP
timeout -k 10 600 python3 $GRAFT_REPO_ROOT/tools/bench_configs.py --steps 8 --warmup 3 > $out/all_workloads.jsonl 2> $out/all_workloads.err || { tail -20 $out/all_workloads.err; exit 1; }
cat $out/all_workloads.jsonl
find $out -name "*kernel_stats.csv"
