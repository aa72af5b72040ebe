#!/bin/bash
# HBM-side traffic of every kernel of bench.py's step: two counter passes (their own runs, counters only), then the summary
# that bench.py's roofline.traffic reads.  usage (on the GPU box): bash tools/pmc_hbm.sh [tag]
#   -> gpurun_out/<tag>/{f,w}/..._counter_collection.csv ; afterwards, here:
#   python tools/summarize_pmc_hbm.py gpurun_out/<tag> profiles/r3_pmc_hbm.md tools/pmc_hbm.json
tag=${1:-pmc_hbm}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
export VITK_WGRAD_STREAM=0
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 170 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/f -o f -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $out/f.log 2>&1 || { tail -5 $out/f.log; exit 1; }
timeout -s KILL 170 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/w -o w -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $out/w.log 2>&1 || { tail -5 $out/w.log; exit 1; }
find $out -name "*counter_collection.csv"
