#!/bin/bash
# Matrix-pipe / wait counters of the GEMM kernels, per epilogue instantiation, over tools/bench_gemm.py (each shape of the
# ViT-B step a few times).  Two counter passes in runs of their own.  usage (GPU box): bash tools/pmc_gemm.sh [tag]
#   afterwards: python tools/summarize_pmc_gemm.py gpurun_out/<tag> profiles/r3_pmc_gemm.md
tag=${1:-pmc_gemm}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $out/p1 -o p1 -- python3 $GRAFT_REPO_ROOT/tools/bench_gemm.py all 2 > $out/p1.log 2>&1 || { tail -5 $out/p1.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $out/p2 -o p2 -- python3 $GRAFT_REPO_ROOT/tools/bench_gemm.py all 2 > $out/p2.log 2>&1 || { tail -5 $out/p2.log; exit 1; }
find $out -name "*counter_collection.csv"
