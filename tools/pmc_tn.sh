#!/bin/bash
# HBM-side fetch bytes of the weight-gradient GEMM per shape, with and without the bias gradient (counters only).
# usage (GPU box): bash tools/pmc_tn.sh [tag]
tag=${1:-pmc_tn}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/db -o db -- python3 $GRAFT_REPO_ROOT/tools/bench_gemm.py tn 3 > $out/db.log 2>&1 || { tail -5 $out/db.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/nodb -o nodb -- python3 $GRAFT_REPO_ROOT/tools/bench_gemm.py tn 3 nodb > $out/nodb.log 2>&1 || { tail -5 $out/nodb.log; exit 1; }
find $out -name "*counter_collection.csv"
