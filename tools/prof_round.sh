#!/bin/bash
# Round profile (run on the GPU box via gpurun): kernel-trace stats of the default bench command, then
# separate PMC passes for HBM traffic (FETCH_SIZE / WRITE_SIZE cannot share a pass) and MFMA busy.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
TAG=${TAG:-r02}
OUT=gpurun_out/$TAG
mkdir -p $OUT
BENCH="python bench.py --steps ${STEPS:-5} --warmup 2 --no-cpu-baseline --no-tiers"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- $BENCH 2> $OUT/trace.log | tail -1 > $OUT/bench_under_trace.json
cp bench_detail.json $OUT/bench_under_trace_detail.json
PMC="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-tiers"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT -o pmc_fetch -- $PMC > /dev/null 2> $OUT/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT -o pmc_write -- $PMC > /dev/null 2> $OUT/pmc_write.log
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT -o pmc_sq -- $PMC > /dev/null 2> $OUT/pmc_sq.log
ls -la $OUT | head -30
# keep only the small summaries (the per-dispatch CSVs are tens of MB)
python tools/pmc_summary.py $OUT/pmc_fetch_counter_collection.csv > $OUT/pmc_fetch_summary.csv
python tools/pmc_summary.py $OUT/pmc_write_counter_collection.csv > $OUT/pmc_write_summary.csv
python tools/pmc_summary.py $OUT/pmc_sq_counter_collection.csv > $OUT/pmc_sq_summary.csv
rm -f $OUT/*_counter_collection.csv $OUT/trace_kernel_trace.csv
head -12 $OUT/trace_kernel_stats.csv
