#!/bin/bash
# Round-6 evidence in ONE gpurun call (GPU box): the GPU suite, smoke, the default bench line as the driver runs it (timed),
# a sustained run, soak, rocprofv3 kernel-trace stats + FETCH / WRITE / SQ passes of the bench command, the refiner towers
# level by level, kernel traces of BASELINE configs 4 / 5 at batch, the chain and 30x40-regulariser benches, the bf16
# feature tier's passes -- then the three PMC JSON files bench.py quotes, generated ON the box from those passes (same
# library, same digest) and the bench lines that carry them.  Everything lands under gpurun_out/<TAG>*; copy what is kept
# to profiles/.  (bench.py: last stdout line = the compact line; bench_detail.json = the full record.)
set -u
cd "$GRAFT_REPO_ROOT"
R=${R:-r06}
TAG=${TAG:-${R}_final}
OUT=gpurun_out/$TAG
mkdir -p $OUT/json
TAG=$TAG SOAK=${SOAK:-600} SOAKG=${SOAKG:-200} bash tools/prof_final.sh > $OUT/prof_final.log 2>&1
python tools/pmc_traffic_json.py $OUT 512 > $OUT/json/${R}_pmc_traffic.json
TAG=${TAG}_levels bash tools/prof_levels.sh > $OUT/prof_levels.log 2>&1
python tools/level_profile.py json gpurun_out/${TAG}_levels 256 > $OUT/json/${R}_level_pmc.json
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for cfg in "config4 128" "config5 32"; do
  set -- $cfg
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o $1 -- python bench.py --config $1 --batch $2 --steps 3 --warmup 1 \
    --no-cpu-baseline --no-tiers 2> $OUT/$1_trace.log | tail -1 > $OUT/bench_$1_b$2_under_trace.json
  cp bench_detail.json $OUT/bench_$1_b$2_under_trace_detail.json
  rm -f $OUT/$1_kernel_trace.csv
done
MVSN_GRID=30,40,96 python tools/chain_bench.py 128 256 > $OUT/chain_bench_30x40.txt 2>&1
MVSN_GRID=32,64,128 python tools/chain_bench.py 128 256 > $OUT/chain_bench_32x64.txt 2>&1
python tools/chain_bench.py 256 512 > $OUT/chain_bench_16x32.txt 2>&1
python tools/vol30x40_bench.py 2>&1 | grep -v amdgpu.ids > $OUT/vol30x40_bench.txt
TAG=${R}_bf16s bash tools/prof_feature_tier.sh > /dev/null 2>&1
cp gpurun_out/${R}_bf16s/feature_tier_pmc.json $OUT/json/${R}_bf16_feature_tier_pmc.json
cp $OUT/json/*.json profiles/
# the lines that carry the counters (same library: the digest in the JSON files is this build's)
( time python bench.py 2> /dev/null | tail -1 > $OUT/bench_default.json ) 2> $OUT/bench_default_time.txt
cp bench_detail.json $OUT/bench_default_detail.json
python bench.py --steps 20 --warmup 5 2> /dev/null | tail -1 > $OUT/bench_with_counters.json
cp bench_detail.json $OUT/bench_with_counters_detail.json
python bench.py --config config4 --batch 128 --steps 5 --warmup 2 --no-cpu-baseline --no-tiers 2> /dev/null | tail -1 > $OUT/bench_config4_b128.json
cp bench_detail.json $OUT/bench_config4_b128_detail.json
MVSN_BENCH_BACKEND=nccl python bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-tiers 2> $OUT/bench_rccl_world1.err | tail -1 > $OUT/bench_rccl_world1.json
python tools/slab_soak.py 2>&1 | grep -v amdgpu.ids > $OUT/slab_soak.txt
rm -f $OUT/*agent_info.csv $OUT/*_trace.log
wc -c $OUT/bench_default.json $OUT/bench_with_counters.json
ls $OUT
