#!/bin/bash
# One pass with GRBM_GUI_ACTIVE (summed over the 8 XCDs: effective clock = value / 8 / kernel wall time), SQ_BUSY_CYCLES and the matrix-pipe busy
# cycles over one kernel: MFMA busy as a fraction of KERNEL WALL TIME = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8).
# Usage (GPU box): tools/pmc_busy.sh <tag> <one_kernel.py args...>
tag=$1; shift
$(dirname "$0")/pmc_pass.sh ${tag} "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU" "$@"
