mkdir -p gpurun_out/r04b
(timeout 900 python -m pytest tests -m gpu -x -q -k "chain or forward or banded or plan or graph" 2>&1 | tail -5) > gpurun_out/r04b/tests.txt
python tools/chain_bench.py 256 512 > gpurun_out/r04b/chain_new.txt 2>&1
MVSN_HIPCC_FLAGS="-DMVSN_CW_NO_XCD_PAIRS" python multi_view_stereonet_amd/build.py --force > /dev/null 2>&1
python tools/chain_bench.py 256 512 > gpurun_out/r04b/chain_noxcd.txt 2>&1
python multi_view_stereonet_amd/build.py --force > /dev/null 2>&1
python tools/chain_bench.py 256 512 > gpurun_out/r04b/chain_new2.txt 2>&1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r04b -o fetch -- python tools/chain_bench.py 512 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r04b -o write -- python tools/chain_bench.py 512 > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/r04b/fetch_counter_collection.csv > gpurun_out/r04b/fetch_summary.csv
python tools/pmc_summary.py gpurun_out/r04b/write_counter_collection.csv > gpurun_out/r04b/write_summary.csv
rm -f gpurun_out/r04b/*_counter_collection.csv
timeout 300 python bench.py --steps 10 --warmup 3 --no-tiers --no-cpu-baseline > gpurun_out/r04b/bench.json 2> gpurun_out/r04b/bench.err
cat gpurun_out/r04b/tests.txt gpurun_out/r04b/chain_*.txt; grep chain_wino gpurun_out/r04b/*_summary.csv
