set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_a
timeout 1500 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "launch_geometry or multi_rank_line" -s 2>&1 | tail -15 > gpurun_out/r05_a/new_tests.txt
# chain phases at 256 chains (stamps build)
MVSN_HIPCC_FLAGS="-DMVSN_CHAIN_STAMPS" python multi_view_stereonet_amd/build.py --force > /dev/null 2>&1
MVSN_HIPCC_FLAGS="-DMVSN_CHAIN_STAMPS" timeout 300 python tools/chain_phases.py 128 > gpurun_out/r05_a/chain_phases_b128.txt 2>&1
# s2 counted wait A/B
bash tools/ab_tool.sh tools/s2_bench.py "" "-DMVSN_S2_CNTWAIT=0" > gpurun_out/r05_a/s2_cntwait.txt 2>&1
python multi_view_stereonet_amd/build.py --force > /dev/null 2>&1
timeout 600 python tools/chain_bench.py 128 256 > gpurun_out/r05_a/chain_bench_16x32.txt 2>&1
MVSN_GRID=30,40,96 timeout 600 python tools/chain_bench.py 128 256 > gpurun_out/r05_a/chain_bench_30x40.txt 2>&1
MVSN_GRID=32,64,128 timeout 600 python tools/chain_bench.py 128 256 > gpurun_out/r05_a/chain_bench_32x64.txt 2>&1
