#!/bin/bash
# SQ / LDS counter passes for the bench command (run on the GPU box through gpurun); extra bench flags in $BENCH_ARGS.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc
mkdir -p $OUT
CMD="python bench.py --batch ${BATCH:-32} --steps 1 --warmup 0 --no-cpu-baseline --no-tiers ${BENCH_ARGS:-}"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS \
  --output-format csv -d $OUT -o pass1 -- $CMD > $OUT/pass1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE \
  --output-format csv -d $OUT -o pass2 -- $CMD > $OUT/pass2.log 2>&1
python tools/pmc_summary.py $OUT/pass1_counter_collection.csv > $OUT/pass1_summary.csv
python tools/pmc_summary.py $OUT/pass2_counter_collection.csv > $OUT/pass2_summary.csv
rm -f $OUT/*_counter_collection.csv
grep -E "kernel,|chain_wino|conv_wino_kernel<0, 2, 3, 1, true" $OUT/pass1_summary.csv $OUT/pass2_summary.csv | cut -c1-260
