#!/bin/bash
# bf16 feature tier evidence (GPU box, via gpurun): the regulariser alone per precision tier (tools/vol_tiers.py) at
# config 5's and the headline's shapes, a kernel trace and FETCH_SIZE / WRITE_SIZE passes of the config-5 run, and the
# JSON bench.py quotes (copy it to profiles/r05_bf16_feature_tier_pmc.json).  TAG names the output directory.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${TAG:-r05_bf16s}
mkdir -p $OUT
rm -f $OUT/chain_bench_cost_storage.txt
python tools/vol_tiers.py 128 32 64 128 2>&1 | grep -v amdgpu.ids > $OUT/vol_tiers_config5.txt
python tools/vol_tiers.py 512 16 32 64 2>&1 | grep -v amdgpu.ids > $OUT/vol_tiers_headline.txt
CMD="python tools/vol_tiers.py 128 32 64 128"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o vt -- $CMD > /dev/null 2> $OUT/vt_trace.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT -o vt_fetch -- $CMD > /dev/null 2> $OUT/vt_fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT -o vt_write -- $CMD > /dev/null 2> $OUT/vt_write.log
# Kernel A at config 5's shapes (128 chains of 32x64, D = 128: the slab plan), fp32 and bf16 cost volume: bytes written
for c16 in 0 1; do
  MVSN_GRID=32,64,128 MVSN_FORMS=banded-auto MVSN_COST_BF16=$c16 python tools/chain_bench.py 128 2>&1 | grep "N=" >> $OUT/chain_bench_cost_storage.txt
  MVSN_GRID=32,64,128 MVSN_FORMS=banded-auto MVSN_COST_BF16=$c16 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT -o chain_write$c16 \
    -- python tools/chain_bench.py 128 > /dev/null 2> $OUT/chain_write$c16.log
  python tools/pmc_summary.py $OUT/chain_write${c16}_counter_collection.csv > $OUT/chain_write${c16}_summary.csv
  # ... and at the headline's (512 chains of 16x32, D = 64: the plane-resident kernel, no hand-off traffic)
  MVSN_FORMS=winograd MVSN_COST_BF16=$c16 python tools/chain_bench.py 512 2>&1 | grep "N=" >> $OUT/chain_bench_cost_storage.txt
  MVSN_FORMS=winograd MVSN_COST_BF16=$c16 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT -o chainw_write$c16 \
    -- python tools/chain_bench.py 512 > /dev/null 2> $OUT/chainw_write$c16.log
  python tools/pmc_summary.py $OUT/chainw_write${c16}_counter_collection.csv > $OUT/chainw_write${c16}_summary.csv
done
python tools/pmc_summary.py $OUT/vt_fetch_counter_collection.csv > $OUT/vt_fetch_summary.csv
python tools/pmc_summary.py $OUT/vt_write_counter_collection.csv > $OUT/vt_write_summary.csv
python tools/vol_tiers.py json $OUT 128 > $OUT/feature_tier_pmc.json
rm -f $OUT/*_counter_collection.csv $OUT/*agent_info.csv $OUT/vt_kernel_trace.csv $OUT/vt_domain_stats.csv $OUT/*.log
ls $OUT
