#!/bin/bash
# after `gpurun -- bash scripts/r5_final.sh`: gpurun_out/r5_final + gpurun_out/prof_r5f_* -> profiles/r5_final (summaries, full bench lines, logs)
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
O=gpurun_out/r5_final; D=profiles/r5_final
for w in nsq24 rts96 seq; do python scripts/summarize_profile.py r5f r5_final $w | tail -1 | cut -c1-60; done
cp $O/pytest_gpu.log $O/golden_pin.log $O/converged.log $O/pcie.log $O/pmc_mix.log $O/wave_tail.log $D/
for f in bench_default bench_driver_shape bench_rts96 bench_seq bench_2rank_shared bench_8rank_shared bench_8rank_strong_1e8 bench_8rank_seq_1000y bench_8rank_rts96_1e7; do
  grep '^{' $O/$f.json | tail -1 | python -m json.tool > $D/${f}_full.json
done
echo r5_final > profiles/current.txt
