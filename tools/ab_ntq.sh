#!/bin/bash
# tools/ab_ntq.py over several library builds in ONE gpurun call: usage tools/ab_ntq.sh rounds reps base f80 ...  (NAME = ab/libvitk_NAME.so)
rounds=$1; reps=$2; shift 2
for v in "$@"; do
  if [ "$v" = base ]; then unset VITK_LIB_PATH; else export VITK_LIB_PATH=$GRAFT_REPO_ROOT/ab/libvitk_$v.so; fi
  echo "== $v"
  timeout -k 10 200 python3 tools/ab_ntq.py $rounds $reps || exit 1
done
