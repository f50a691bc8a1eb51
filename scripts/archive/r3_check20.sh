#!/bin/bash
# where a wavefront's cycles go in the current kernel: main phases (RTS-24, RTS-96) and the per-group setup
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r3c20; mkdir -p $O; cd $R
RELMC_LIB_PATH=$R/powersystemsreliabilityassessment_amd/csrc/ablate/librelmc_pt.so python scripts/phase_timing.py 2>&1 | tee $O/phase24.log
RELMC_LIB_PATH=$R/powersystemsreliabilityassessment_amd/csrc/ablate/librelmc_pt.so python scripts/phase_timing96.py 2>&1 | tee $O/phase96.log
RELMC_LIB_PATH=$R/powersystemsreliabilityassessment_amd/csrc/ablate/librelmc_pti.so python scripts/phase_timing_init.py 2>&1 | tee $O/init24.log
