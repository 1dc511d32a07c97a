#!/bin/bash
# Counter passes of the refiner towers, one level per process (GPU box, via gpurun): FETCH_SIZE, WRITE_SIZE and the SQ
# set cannot share a pass.  Summaries + the per-level timelines land in gpurun_out/$TAG; tools/level_profile.py json
# turns them into profiles/<round>_level_pmc.json.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
TAG=${TAG:-r04_levels}
B=${BATCH:-256}
OUT=gpurun_out/$TAG
mkdir -p $OUT
for L in ${LEVELS:-0 1 2 3}; do
  CMD="python tools/level_profile.py run $L $B"
  $CMD > $OUT/L${L}_timeline.json 2> $OUT/L${L}_timeline.err
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT -o L${L}_fetch -- $CMD > /dev/null 2> $OUT/L${L}_fetch.log
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT -o L${L}_write -- $CMD > /dev/null 2> $OUT/L${L}_write.log
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE \
    --output-format csv -d $OUT -o L${L}_sq -- $CMD > /dev/null 2> $OUT/L${L}_sq.log
  for K in fetch write sq; do
    python tools/pmc_summary.py $OUT/L${L}_${K}_counter_collection.csv > $OUT/L${L}_${K}_summary.csv
  done
  rm -f $OUT/L${L}_*_counter_collection.csv $OUT/*agent_info.csv
done
python tools/level_profile.py json $OUT $B > $OUT/level_pmc.json
head -c 1500 $OUT/level_pmc.json
