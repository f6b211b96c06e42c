#!/bin/bash
# round-4 measurement session: in-kernel clocks (clock build) + GRBM / MFMA-busy passes on the dominant convolution shapes
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r04
CINEFLOW_LIB=$PWD/cardiac-segmentation-optical-flow_amd/cineflow/libcineflow_hip_clock.so timeout -k 10 200 python tools/power_probe.py > gpurun_out/r04/power_probe.log 2>&1 || exit 1
tail -12 gpurun_out/r04/power_probe.log
for spec in "wino128 wino 128 0 128 128 128 1" "wino256 wino 256 256 64 256 64 1" "stream64 conv 64 0 256 64 3 1 128" "direct128 conv 128 0 128 128 3 1 128"; do
  set -- $spec; tag=$1; shift
  tools/pmc_busy.sh $tag "$@" > gpurun_out/r04/pmc_busy_$tag.log 2>&1 || { tail -5 gpurun_out/r04/pmc_busy_$tag.log; exit 1; }
  cat gpurun_out/r04/pmc_busy_$tag.log
done
