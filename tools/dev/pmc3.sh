#!/bin/bash
# usage (GPU box, repo root): tools/dev/pmc3.sh <outdir> <cmd...>  - the three SQ passes of tools/pmc_run.sh only
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_UNALIGNED_STALL"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $out/p$i -- "$@" > $out/p$i.log 2>&1
done
