#!/bin/bash
# A/B helper (GPU box): rebuild the library with each flag set in turn and run a timing tool.
# usage: tools/ab_tool.sh tools/<tool>.py "<flags A>" "<flags B>" ...      (an empty string = the default build)
tool=$1; shift
for flags in "$@"; do
  MVSN_HIPCC_FLAGS="$flags" python multi_view_stereonet_amd/build.py --force > /dev/null 2>&1 || { echo "build failed: $flags"; continue; }
  echo "== [$flags]"
  MVSN_HIPCC_FLAGS="$flags" timeout 300 python $tool 2>&1 | grep -E "median|rror|launch|checksum"
done
