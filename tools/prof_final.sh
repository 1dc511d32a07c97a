#!/bin/bash
# Round-end evidence (GPU box, via gpurun): GPU suite, default bench line, sustained run, soak, rocprofv3 trace + PMC
# passes of the bench command (tools/prof_round.sh), batch-1 kernel trace.  Everything lands in gpurun_out/$TAG.
set -u
cd "$GRAFT_REPO_ROOT"
TAG=${TAG:-r03_final}
OUT=gpurun_out/$TAG
mkdir -p $OUT
(timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -3) > $OUT/gpu_suite.txt; cat $OUT/gpu_suite.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee $OUT/smoke.txt
# (bench.py: last stdout line = the compact line the driver parses; bench_detail.json = the full record)
timeout 600 python bench.py --steps 20 --warmup 5 2> $OUT/bench_default.err | tail -1 > $OUT/bench_default.json
cp bench_detail.json $OUT/bench_default_detail.json
wc -c $OUT/bench_default.json
timeout 300 python bench.py --steps 300 --warmup 5 --no-cpu-baseline --no-tiers --sustain 24 2> /dev/null | tail -1 > $OUT/bench_sustain.json
cp bench_detail.json $OUT/bench_sustain_detail.json
timeout 1200 python tools/soak.py ${SOAK:-3000} > $OUT/soak.json 2> $OUT/soak.err
timeout 600 python tools/soak.py ${SOAKG:-1000} graphed > $OUT/soak_graphed.json 2>> $OUT/soak.err
TAG=$TAG timeout 900 bash tools/prof_round.sh > $OUT/prof_round.log 2>&1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o b1 -- python tools/b1_trace.py 1 > $OUT/b1_trace.log 2>&1
rm -f $OUT/b1_kernel_trace.csv $OUT/*agent_info.csv
ls $OUT
python - <<PY
import json
for f in ("bench_default", "bench_sustain", "soak", "soak_graphed"):
    try:
        d = json.loads(open("$OUT/%s.json" % f).read().strip().splitlines()[-1])
        print(f, d.get("value"), d.get("ms_per_step"), d.get("all_bit_identical"))
    except Exception as e:
        print(f, "unreadable", e)
PY
