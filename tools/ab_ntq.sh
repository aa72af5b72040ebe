#!/bin/bash
# A/B of NT-kernel builds inside one gpurun call.  usage: tools/ab_ntq.sh reps SPEC...   SPEC = p:NAME (8 waves) | q:NAME (4 waves,
# VITK_GEMM_NTQ=1); NAME = base (the in-tree build) or ab/libvitk_NAME.so; two rounds
reps=$1; shift 1
for round in 1 2; do
for spec in "$@"; do
  kind=${spec%%:*}; name=${spec#*:}
  if [ "$kind" = q ]; then export VITK_GEMM_NTQ=1; else unset VITK_GEMM_NTQ; fi
  if [ "$name" = base ]; then unset VITK_LIB_PATH; else export VITK_LIB_PATH=$GRAFT_REPO_ROOT/ab/libvitk_$name.so; fi
  echo "== $spec (round $round)"
  timeout -k 10 120 python3 tools/bench_gemm.py nt $reps || exit 1
done
done
