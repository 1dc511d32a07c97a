#!/bin/bash
# A/B helper (GPU box): rebuild the library with each flag set in turn, run the bench twice, print the headline and the
# main kernel groups.   usage: tools/ab_bench.sh "<flags A>" "<flags B>" ...   (an empty string = the default build)
for flags in "$@"; do
  MVSN_HIPCC_FLAGS="$flags" python multi_view_stereonet_amd/build.py --force > /dev/null 2>&1 || { echo "build failed: $flags"; continue; }
  echo "== [$flags]"
  for i in 1 2; do
    MVSN_HIPCC_FLAGS="$flags" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-tiers > /dev/null 2>&1; python -c "
import json; d=json.load(open('bench_detail.json')); k=d['kernel_ms_per_step']
print(round(d['value'],1), round(d['ms_per_step'],2), 'frac', round(d['roofline']['frac'],4), 'conv3d', round(k['mvsn_conv_forward[conv3d k3 32->32 wino]'],2), 'L0', round(sum(v for n,v in k.items() if ' L0' in n),2), 'L1', round(sum(v for n,v in k.items() if ' L1' in n),2), 'rel', d['l1_vs_ref']['mean_rel'])"
  done
done
