#!/bin/bash
# A/B of library builds on the whole training step inside one gpurun call: usage: tools/ab_bench.sh steps base NAME ...; two rounds
steps=$1; shift 1
for round in 1 2; do
for v in "$@"; do
  if [ "$v" = base ]; then unset VITK_LIB_PATH; else export VITK_LIB_PATH=$GRAFT_REPO_ROOT/ab/libvitk_$v.so; fi
  echo "== $v (round $round)"
  timeout -k 10 200 python3 bench.py --steps $steps --warmup 4 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'img/s', d['ms_per_step'], 'ms')" || exit 1
done
done
