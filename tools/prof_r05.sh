#!/bin/bash
# Round-5 evidence in one gpurun call: tools/prof_final.sh (suite, smoke, bench, sustain, soak, trace + PMC passes, batch-1
# trace), the PMC traffic file bench.py reads, the refiner towers level by level (tools/prof_levels.sh) and kernel traces of
# BASELINE configs 4 / 5 at batch (the slab chain and the wide-tile regulariser).  TAG names the output directories.
set -u
cd "$GRAFT_REPO_ROOT"
TAG=${TAG:-r05_final}
OUT=gpurun_out/$TAG
TAG=$TAG SOAK=${SOAK:-600} SOAKG=${SOAKG:-200} bash tools/prof_final.sh
python tools/pmc_traffic_json.py $OUT 512 > $OUT/pmc_traffic.json
TAG=${TAG}_levels bash tools/prof_levels.sh > $OUT/prof_levels.log 2>&1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for cfg in "config4 128" "config5 32"; do
  set -- $cfg
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o $1 -- python bench.py --config $1 --batch $2 --steps 3 --warmup 1 \
    --no-cpu-baseline --no-tiers > $OUT/bench_$1_b$2_under_trace.json 2> $OUT/$1_trace.log
  rm -f $OUT/$1_kernel_trace.csv
done
MVSN_GRID=30,40,96 python tools/chain_bench.py 128 256 > $OUT/chain_bench_30x40.txt 2>&1
MVSN_GRID=32,64,128 python tools/chain_bench.py 128 256 > $OUT/chain_bench_32x64.txt 2>&1
python tools/chain_bench.py 256 512 > $OUT/chain_bench_16x32.txt 2>&1
rm -f $OUT/*agent_info.csv $OUT/*.log
TAG=r05_bf16s bash tools/prof_feature_tier.sh > /dev/null 2>&1
ls $OUT gpurun_out/${TAG}_levels
