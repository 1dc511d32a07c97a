#!/bin/bash
# A/B helper (GPU box): rebuild the library with each flag set in turn and print bench throughput.
# usage: tools/ab_build.sh "<flags A>" "<flags B>" ...
for flags in "$@"; do
  MVSN_HIPCC_FLAGS="$flags" python multi_view_stereonet_amd/build.py --force > /dev/null 2>&1 || { echo "build failed: $flags"; continue; }
  echo "== $flags"
  timeout 300 python -m pytest tests -m gpu -x -q -k "conv2d or resid or golden" 2>&1 | tail -1
  for i in 1 2; do timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1; python -c "import json; d=json.load(open('bench_detail.json')); print(d['value'], d['ms_per_step'], d.get('bf16x3_split_tier',{}).get('value'))"; done
done
