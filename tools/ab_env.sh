#!/bin/bash
# A/B of environment settings inside one gpurun call: usage: tools/ab_env.sh "CMD" "VAR=val ..." "VAR=val ..." ...  ("-" = nothing set); two rounds
cmd=$1; shift 1
for round in 1 2; do
for spec in "$@"; do
  echo "== $spec (round $round)"
  if [ "$spec" = "-" ]; then timeout -k 10 200 $cmd || exit 1; else env $spec timeout -k 10 200 $cmd || exit 1; fi
done
done
