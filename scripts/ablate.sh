#!/bin/bash
# Phase ablation of relmc_eval_kernel at a fixed iteration count (profiling only; results are garbage by design).
cd $(dirname $0)/../powersystemsreliabilityassessment_amd/csrc
python3 gen_elim.py elim_nb24.inc
B="/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -shared"
mkdir -p ablate
$B -DRELMC_ABLATE_FIXIT=12 -o ablate/full.so relmc_abi.hip &
$B -DRELMC_ABLATE_FIXIT=12 -DRELMC_ABLATE_NO_ELIM -o ablate/noelim.so relmc_abi.hip &
$B -DRELMC_ABLATE_FIXIT=12 -DRELMC_ABLATE_NO_ELIM -DRELMC_ABLATE_NO_ASSEMBLE -o ablate/noelim_noasm.so relmc_abi.hip &
$B -DRELMC_ABLATE_FIXIT=1 -o ablate/one.so relmc_abi.hip &
wait
ls -la ablate
