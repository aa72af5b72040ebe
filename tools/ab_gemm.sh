#!/bin/bash
# A/B micro-benchmark of library builds inside ONE gpurun call (devices differ by several percent):
# usage: tools/ab_gemm.sh "nt|tn|all" reps base stg2 stg4 ...   (base = the in-tree build, NAME = ab/libvitk_NAME.so); two rounds
which=$1; reps=$2; shift 2
for round in 1 2; do
for v in "$@"; do
  if [ "$v" = base ]; then unset VITK_LIB_PATH; else export VITK_LIB_PATH=$GRAFT_REPO_ROOT/ab/libvitk_$v.so; fi
  echo "== $v (round $round)"
  timeout -k 10 120 python3 tools/bench_gemm.py $which $reps || exit 1
done
done
