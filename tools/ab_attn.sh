#!/bin/bash
# A/B micro-benchmark of library builds on the attention kernels inside ONE gpurun call (devices differ by several percent):
# usage: tools/ab_attn.sh reps base NAME ...   (base = the in-tree build, NAME = ab/libvitk_NAME.so); two rounds
reps=$1; shift 1
for round in 1 2; do
for v in "$@"; do
  if [ "$v" = base ]; then unset VITK_LIB_PATH; else export VITK_LIB_PATH=$GRAFT_REPO_ROOT/ab/libvitk_$v.so; fi
  echo "== $v (round $round)"
  timeout -k 10 120 python3 tools/bench_attn.py $reps || exit 1
done
done
