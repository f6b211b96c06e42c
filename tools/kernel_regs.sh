#!/bin/bash
# Per-kernel VGPR / scratch / occupancy of one csrc/*.hip file (compiler remarks; no GPU needed).
# Usage: tools/kernel_regs.sh conv_f16s.hip
cd "$(dirname "$0")/../cardiac-segmentation-optical-flow_amd/csrc" || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-gpu-rdc -c "$1" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 |
  grep -E "Function Name|VGPRs:|AGPRs|ScratchSize|Occupancy|LDS Size" | sed -e 's/^.*remark: [^ ]* *//' -e 's/ \[-Rpass.*$//' |
  awk '/Name:/{if(l)print l; sub(/.*Name: /,""); l=$0; next}{l=l" | "$0}END{print l}' | c++filt
