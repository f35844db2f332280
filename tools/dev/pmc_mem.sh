#!/bin/bash
# usage (GPU box, repo root): tools/dev/pmc_mem.sh <outdir> <cmd...>  - L2 (TCC) request / hit / stall counters and the L1's view of them
# (every pass under its own timeout: rocprofv3 aborts on an unknown counter name and then sits in its signal handler)
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out
i=0
for set in "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum TCC_WRITE_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
           "TCC_TAG_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_SRC_FIFO_FULL_sum TCC_LATENCY_FIFO_FULL_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCC_BUSY_sum TCC_CYCLE_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 5 100 rocprofv3 --pmc $set --output-format csv -d $out/p$i -- "$@" > $out/p$i.log 2>&1 || echo "pass $i failed"
done
