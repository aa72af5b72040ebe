#!/bin/bash
# A/B of library builds on the attention micro-benchmark inside one gpurun call: tools/ab_attn.sh base NAME ... (two rounds)
for round in 1 2; do
for v in "$@"; do
  if [ "$v" = base ]; then unset VITK_LIB_PATH; else export VITK_LIB_PATH=$GRAFT_REPO_ROOT/ab/libvitk_$v.so; fi
  echo "== $v (round $round)"
  timeout -k 10 120 python3 tools/bench_attn.py 20 2>&1 | grep attention || exit 1
done
done
