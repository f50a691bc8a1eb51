#!/bin/bash
# GPU-in-the-loop refinement of the shipped RTS-96 order (6 minutes), then of the RTS-24 order (4 minutes)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r3c36; mkdir -p $O; cd $R
timeout 500 python scripts/order_tune_gpu.py rts96 scripts/orders/96_c1.txt 360 1 2>&1 | tee $O/gpu_tune96.log
timeout 400 python scripts/order_tune_gpu.py rts24 scripts/orders/24_s11.txt 240 1 2>&1 | tee $O/gpu_tune24.log
