for flags in "$@"; do
  MVSN_HIPCC_FLAGS="$flags" python multi_view_stereonet_amd/build.py --force > /dev/null 2>&1 || { echo "build failed: $flags"; continue; }
  echo "== [$flags]"
  MVSN_HIPCC_FLAGS="$flags" timeout 300 python tools/head_bench.py 2>&1 | grep -E "median|Error|error"
done
