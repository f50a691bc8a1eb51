#!/bin/bash
# nform: 1/D and Np/D of an injection from one reciprocal (of N = mu+ z- + mu- z+) instead of three; pup: partner-line loads of the assembly in one batch
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r3c12; mkdir -p $O; cd $R
python scripts/variant_check.py base nform pup pupn base nform pup pupn 2>&1 | tee $O/v24.log
python scripts/variant_check96.py base pup base pup 2>&1 | tee $O/v96.log
