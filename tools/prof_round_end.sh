#!/bin/bash
# The round's whole evidence set in ONE gpurun call (GPU box): tools/prof_r05.sh (suite, smoke, bench, sustain, soak, traces,
# PMC passes, level passes, config-4/5 traces, chain benches, the bf16 feature tier's passes), then the three PMC JSON files
# bench.py quotes generated ON the box from those passes (same library, same digest), then the bench lines that carry them
# (`python bench.py` as the driver runs it, timed; `--steps 20 --warmup 5`) and the slab soak.  Everything lands under
# gpurun_out/<TAG>*; the JSON files also under gpurun_out/<TAG>/json/ -- copy them to profiles/.
set -u
cd "$GRAFT_REPO_ROOT"
TAG=${TAG:-r05_final}
OUT=gpurun_out/$TAG
TAG=$TAG bash tools/prof_r05.sh > /dev/null 2>&1
mkdir -p $OUT/json
python tools/pmc_traffic_json.py $OUT 512 > $OUT/json/r05_pmc_traffic.json
python tools/level_profile.py json gpurun_out/${TAG}_levels 256 > $OUT/json/r05_level_pmc.json
cp gpurun_out/r05_bf16s/feature_tier_pmc.json $OUT/json/r05_bf16_feature_tier_pmc.json
cp $OUT/json/*.json profiles/
export TMPDIR=/tmp
( time python bench.py ) > $OUT/bench_default.json 2> $OUT/bench_default_time.txt
python bench.py --steps 20 --warmup 5 > $OUT/bench_with_counters.json 2> /dev/null
python tools/slab_soak.py 2>&1 | grep -v amdgpu.ids > $OUT/slab_soak.txt
ls $OUT
