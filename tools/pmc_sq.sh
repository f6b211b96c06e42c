#!/bin/bash
# One SQ-counter pass (wave cycles, wait / issue split, MFMA busy, instruction counts) over one kernel.  Usage (GPU box): tools/pmc_sq.sh <tag> <one_kernel.py args...>
tag=$1; shift
$(dirname "$0")/pmc_pass.sh ${tag} "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "$@"
