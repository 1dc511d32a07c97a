#!/bin/bash
# bf16 feature tier evidence (GPU box, via gpurun): the regulariser alone per precision tier (tools/vol_tiers.py) at
# config 5's and the headline's shapes, a kernel trace and FETCH_SIZE / WRITE_SIZE passes of the config-5 run, and the
# JSON bench.py quotes (copy it to profiles/r05_bf16_feature_tier_pmc.json).  TAG names the output directory.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${TAG:-r05_bf16s}
mkdir -p $OUT
python tools/vol_tiers.py 128 32 64 128 2>&1 | grep -v amdgpu.ids > $OUT/vol_tiers_config5.txt
python tools/vol_tiers.py 512 16 32 64 2>&1 | grep -v amdgpu.ids > $OUT/vol_tiers_headline.txt
CMD="python tools/vol_tiers.py 128 32 64 128"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o vt -- $CMD > /dev/null 2> $OUT/vt_trace.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT -o vt_fetch -- $CMD > /dev/null 2> $OUT/vt_fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT -o vt_write -- $CMD > /dev/null 2> $OUT/vt_write.log
python tools/pmc_summary.py $OUT/vt_fetch_counter_collection.csv > $OUT/vt_fetch_summary.csv
python tools/pmc_summary.py $OUT/vt_write_counter_collection.csv > $OUT/vt_write_summary.csv
python tools/vol_tiers.py json $OUT 128 > $OUT/feature_tier_pmc.json
rm -f $OUT/*_counter_collection.csv $OUT/*agent_info.csv $OUT/vt_kernel_trace.csv $OUT/vt_domain_stats.csv $OUT/*.log
ls $OUT
