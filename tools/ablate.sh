#!/bin/bash
# timing-only ablations of conv_f16s (results are WRONG with dbg != 0)
for d in 0 1 2 4 8 9 15; do
  echo "== CF_F16S_DBG=$d"
  CF_F16S_DBG=$d python tools/microbench.py --only conv 2>/dev/null | grep -E "C 128\+0   128x128 ->  128 k3 s1|C  64\+0   256x256 ->   64 k3 s1|256\+0    32x32  -> 2048 k1" | cut -c1-75
done
