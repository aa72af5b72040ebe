#!/bin/bash
# PMC pass over the attention micro-benchmark: where do the waves of the backward kernel spend their cycles
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_attn
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $out/p1 -o p1 -- python3 $GRAFT_REPO_ROOT/tools/bench_attn.py 3 > $out/p1.log 2>&1 || { tail -5 $out/p1.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_MFMA --output-format csv -d $out/p2 -o p2 -- python3 $GRAFT_REPO_ROOT/tools/bench_attn.py 3 > $out/p2.log 2>&1 || { tail -5 $out/p2.log; exit 1; }
ls $out/p1 $out/p2
