cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_x2conv.py tests/test_gpu_parity.py -x -q -m gpu -k "conv or hifigan or resblock or transpose or config3 or x2" 2>&1 | tail -3
SET_AMD_LIB=build/exp/libset_amd_prevepi.so HSTAGES=1 timeout 300 python tools/hifigan_bench.py 2>&1 | grep "ms/forward\| up " 
HSTAGES=1 timeout 300 python tools/hifigan_bench.py 2>&1 | grep "ms/forward\| up "
