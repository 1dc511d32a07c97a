set -u
cd "$GRAFT_REPO_ROOT"
TAG=${TAG:-r04_final}
OUT=gpurun_out/$TAG
mkdir -p $OUT
(timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -3) > $OUT/gpu_suite.txt; cat $OUT/gpu_suite.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee $OUT/smoke.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err
TAG=$TAG timeout 900 bash tools/prof_round.sh > $OUT/prof_round.log 2>&1
python tools/pmc_traffic_json.py $OUT 512 > $OUT/pmc_traffic.json
python -c "
import json
d=json.loads(open('$OUT/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])
k=d['kernel_ms_per_step']
for n,v in sorted(k.items(), key=lambda kv:-kv[1])[:16]: print(round(v,3), n)
"
