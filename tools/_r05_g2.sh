set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_b
timeout 1500 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "slab" --timeout 300 -s 2>&1 | tail -40 > gpurun_out/r05_b/slab_tests.txt
MVSN_GRID=30,40,96 timeout 600 python tools/chain_bench.py 128 256 > gpurun_out/r05_b/chain_bench_30x40.txt 2>&1
MVSN_GRID=32,64,128 timeout 600 python tools/chain_bench.py 128 256 > gpurun_out/r05_b/chain_bench_32x64.txt 2>&1
MVSN_HIPCC_FLAGS="-DMVSN_CHAIN_STAMPS" python multi_view_stereonet_amd/build.py --force > /dev/null 2>&1
MVSN_HIPCC_FLAGS="-DMVSN_CHAIN_STAMPS" timeout 300 python tools/slab_phases.py 64 32 64 > gpurun_out/r05_b/slab_phases_32x64.txt 2>&1
MVSN_HIPCC_FLAGS="-DMVSN_CHAIN_STAMPS" timeout 300 python tools/slab_phases.py 85 30 40 > gpurun_out/r05_b/slab_phases_30x40.txt 2>&1
