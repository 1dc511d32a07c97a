#!/bin/bash
# A/B helper (GPU box): rebuild the library with each flag set in turn and time the Winograd layers (tools/vol_bench.py).
# usage: tools/ab_layers.sh "<flags A>" "<flags B>" ...      (an empty string = the default build)
for flags in "$@"; do
  MVSN_HIPCC_FLAGS="$flags" python multi_view_stereonet_amd/build.py --force > /dev/null 2>&1 || { echo "build failed: $flags"; continue; }
  echo "== [$flags]"
  MVSN_HIPCC_FLAGS="$flags" timeout 300 python tools/vol_bench.py 2>&1 | grep -E "median"
done
