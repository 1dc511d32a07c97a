#!/bin/bash
# Round-6 evidence in ONE gpurun call (GPU box).  Order matters: every counter pass first, the JSON files bench.py quotes
# generated ON the box from those passes (same library, same digest), THEN every bench line that is kept -- so each kept line
# carries `roofline.traffic` (round 5 kept four lines taken before their counters existed).
#   1. tools/prof_final.sh: GPU suite, smoke, soak, rocprofv3 kernel-trace stats + FETCH / WRITE / SQ passes of the headline
#      bench command, batch-1 trace (SKIP_FINAL=1: reuse what gpurun_out/<TAG> already holds of this library)
#   2. FETCH / WRITE passes of `bench.py --config config4 --batch 128` / `--config config5 --batch 32`, the refiner towers level
#      by level, the bf16 feature tier's passes
#   3. profiles/<R>_pmc_traffic.json (headline + configs 4 / 5), <R>_level_pmc.json, <R>_bf16_feature_tier_pmc.json
#   4. the kept lines: `python bench.py` as the driver runs it (timed), --steps 20, sustained, under the tracer, configs 4 / 5
#      under the tracer, config 4 at 128 images, a world-size-1 RCCL line; chain / regulariser benches; slab soak
# (bench.py: last stdout line = the compact line; bench_detail.json = the full record.)
set -u
cd "$GRAFT_REPO_ROOT"
R=${R:-r06}
TAG=${TAG:-${R}_final}
OUT=gpurun_out/$TAG
mkdir -p $OUT/json
if [ "${SKIP_FINAL:-0}" = "1" ]; then   # (gpurun_out does not travel: seed it from what profiles/ keeps of THIS library)
  cp -r profiles/$TAG/. $OUT/
  mkdir -p gpurun_out/${TAG}_levels gpurun_out/${R}_bf16s
  cp -r profiles/${R}_levels/. gpurun_out/${TAG}_levels/
  cp -r profiles/${R}_bf16s/. gpurun_out/${R}_bf16s/
else
  TAG=$TAG SOAK=${SOAK:-600} SOAKG=${SOAKG:-200} bash tools/prof_final.sh > $OUT/prof_final.log 2>&1
  TAG=${TAG}_levels bash tools/prof_levels.sh > $OUT/prof_levels.log 2>&1
  TAG=${R}_bf16s bash tools/prof_feature_tier.sh > /dev/null 2>&1
fi
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for cfg in "config4 128" "config5 32"; do
  set -- $cfg
  mkdir -p $OUT/$1
  PMC="python bench.py --config $1 --batch $2 --steps 1 --warmup 1 --no-cpu-baseline --no-tiers"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/$1 -o pmc_fetch -- $PMC > /dev/null 2> $OUT/$1/pmc_fetch.log
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/$1 -o pmc_write -- $PMC > /dev/null 2> $OUT/$1/pmc_write.log
  python tools/pmc_summary.py $OUT/$1/pmc_fetch_counter_collection.csv > $OUT/$1/pmc_fetch_summary.csv
  python tools/pmc_summary.py $OUT/$1/pmc_write_counter_collection.csv > $OUT/$1/pmc_write_summary.csv
  rm -f $OUT/$1/*_counter_collection.csv $OUT/$1/*agent_info.csv $OUT/$1/*.log
done
python tools/pmc_traffic_json.py $OUT 512 config4=$OUT/config4:128 config5=$OUT/config5:128 > $OUT/json/${R}_pmc_traffic.json
python tools/level_profile.py json gpurun_out/${TAG}_levels 256 > $OUT/json/${R}_level_pmc.json
cp gpurun_out/${R}_bf16s/feature_tier_pmc.json $OUT/json/${R}_bf16_feature_tier_pmc.json
cp $OUT/json/*.json profiles/
# ---- the kept lines (this library's counters are in profiles/ now)
( time python bench.py 2> /dev/null | tail -1 > $OUT/bench_default.json ) 2> $OUT/bench_default_time.txt
cp bench_detail.json $OUT/bench_default_detail.json
python bench.py --steps 20 --warmup 5 2> /dev/null | tail -1 > $OUT/bench_with_counters.json
cp bench_detail.json $OUT/bench_with_counters_detail.json
python bench.py --steps 300 --warmup 5 --no-cpu-baseline --no-tiers --sustain 24 2> /dev/null | tail -1 > $OUT/bench_sustain.json
cp bench_detail.json $OUT/bench_sustain_detail.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-tiers \
  2> $OUT/trace.log | tail -1 > $OUT/bench_under_trace.json
cp bench_detail.json $OUT/bench_under_trace_detail.json
rm -f $OUT/trace_kernel_trace.csv
for cfg in "config4 128" "config5 32"; do
  set -- $cfg
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o $1 -- python bench.py --config $1 --batch $2 --steps 3 --warmup 1 \
    --no-cpu-baseline --no-tiers 2> $OUT/$1_trace.log | tail -1 > $OUT/bench_$1_b$2_under_trace.json
  cp bench_detail.json $OUT/bench_$1_b$2_under_trace_detail.json
  rm -f $OUT/$1_kernel_trace.csv
done
python bench.py --config config4 --batch 128 --steps 5 --warmup 2 --no-cpu-baseline --no-tiers 2> /dev/null | tail -1 > $OUT/bench_config4_b128.json
cp bench_detail.json $OUT/bench_config4_b128_detail.json
MVSN_BENCH_BACKEND=nccl python bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-tiers 2> $OUT/bench_rccl_world1.err | tail -1 > $OUT/bench_rccl_world1.json
MVSN_GRID=30,40,96 python tools/chain_bench.py 128 256 > $OUT/chain_bench_30x40.txt 2>&1
MVSN_GRID=32,64,128 python tools/chain_bench.py 128 256 > $OUT/chain_bench_32x64.txt 2>&1
python tools/chain_bench.py 256 512 > $OUT/chain_bench_16x32.txt 2>&1
python tools/vol30x40_bench.py 2>&1 | grep -v amdgpu.ids > $OUT/vol30x40_bench.txt
python tools/slab_soak.py 2>&1 | grep -v amdgpu.ids > $OUT/slab_soak.txt
rm -f $OUT/*agent_info.csv $OUT/*_trace.log
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    if f.endswith("_detail.json"):
        continue
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], len(open(f).read()), round(d["value"], 1), round(d["roofline"]["frac"], 4), d["roofline"]["traffic"])
PY
