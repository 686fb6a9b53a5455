#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/profile.sh <tag> [prof_target args]
# Writes kernel-trace stats (ITERS launches in one warmed process, default 25; cold ones reported apart) and PMC passes
# under gpurun_out/<tag>/ ; BENCH_JSON=<file>: the bench line of the same build, compared in the summary.  Copy what matters to profiles/.
set -u
tag=$1; shift
out=$PWD/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 tools/prof_target.py --iters ${ITERS:-25} "$@" > $out/trace.log 2>&1 || exit 1
pass() {
  name=$1; shift
  timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -- python3 tools/prof_target.py --iters 1 ${EXTRA:-} > $out/$name.log 2>&1
}
EXTRA="$*"
pass pmc1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY || exit 1
pass pmc2 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU || exit 1
pass pmc3 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_FLAT GRBM_GUI_ACTIVE || exit 1
pass pmc4 FETCH_SIZE || exit 1
pass pmc5 WRITE_SIZE || exit 1
python3 tools/prof_summary.py $out ${BENCH_JSON:-} > $out/summary.txt 2>&1
cat $out/summary.txt
