mkdir -p gpurun_out/r04c
(timeout 1200 python -m pytest tests -m gpu -x -q -k "chain or ragged or direct or forward or banded" 2>&1 | tail -5) > gpurun_out/r04c/tests.txt
python tools/chain_bench.py 256 > gpurun_out/r04c/chain_16x32.txt 2>&1
MVSN_GRID=30,40,96 python tools/chain_bench.py 32 256 > gpurun_out/r04c/chain_30x40.txt 2>&1
cat gpurun_out/r04c/*.txt
