#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r3c3; mkdir -p $O; cd $R
python scripts/variant_check.py v_pair2 base v_pair2 base 2>&1 | tee $O/v24.log
python scripts/variant_check96.py v_pair2 base 2>&1 | tee $O/v96.log
RELMC_LIB_PATH=$R/powersystemsreliabilityassessment_amd/csrc/ablate/librelmc_pt.so python scripts/phase_timing.py 2>&1 | tee $O/phase24.log
python scripts/phase_timing96.py 2>&1 | tee $O/phase96.log
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $O/pytest.log
