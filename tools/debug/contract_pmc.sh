#!/bin/bash
# PMC passes over tools/debug/contract_only.py (a few counters per pass); per-kernel averages printed by pmc_avg.py
cd /tmp; export TMPDIR=/tmp
i=0
for grp in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum" \
           "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "GRBM_GUI_ACTIVE TCC_BUSY_avr TCP_TCR_TCP_STALL_CYCLES_sum" \
           "TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_STALL_MULTI_MISS_sum" \
           "TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/cp$i -- python $GRAFT_REPO_ROOT/tools/debug/contract_only.py > /tmp/cp$i.log 2>&1
  python $GRAFT_REPO_ROOT/tools/debug/pmc_avg.py /tmp/cp$i
done
