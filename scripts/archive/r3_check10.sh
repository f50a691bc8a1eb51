#!/bin/bash
# wide tile at three wavefronts per SIMD: 10 waves per workgroup under a 168-VGPR cap (w10), the same with the line / injection table values re-read instead of held (w10l)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r3c10; mkdir -p $O; cd $R
python scripts/variant_check96.py base w10 w10l base w10 2>&1 | tee $O/v96.log
