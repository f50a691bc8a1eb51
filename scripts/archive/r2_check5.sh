set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_rts96.py -m gpu -x -q 2>&1 | tail -8
timeout 120 python scripts/small_batch.py 2>&1 | tail -12
