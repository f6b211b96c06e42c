#!/bin/bash
# PMC passes (issue / wait state, texture-address + L1 stalls, L1 / L2 traffic and latency, LDS + instruction mix) over one convolution shape.
# Usage (GPU box): tools/pmc_conv.sh <tag> <one_kernel.py conv args...>
# The TA_* / TCP_* stall counters go out in sets of at most two TA and three TCP counters: all eight in ONE pass is more than the blocks' counter
# slots on gfx950 -- rocprofv3 answered "error code 38: Request exceeds the capabilities of the hardware to collect", aborted and hung in its
# finaliser (gpurun_out/r03/call22.txt, gpurun_out/pmc_w64_2.log).  Each pass has its own timeout (tools/pmc_pass.sh) and the chain stops at
# the first failure.
tag=$1; shift
d=$(dirname "$0")
$d/pmc_pass.sh ${tag}_1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS" "$@" &&
$d/pmc_pass.sh ${tag}_2a "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "$@" &&
$d/pmc_pass.sh ${tag}_2b "TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum" "$@" &&
$d/pmc_pass.sh ${tag}_3a "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_READ_sum" "$@" &&
$d/pmc_pass.sh ${tag}_3b "TCP_TCP_LATENCY_sum TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum" "$@" &&
$d/pmc_pass.sh ${tag}_4 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_CYCLES_VMEM_WR" "$@"
