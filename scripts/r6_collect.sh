#!/bin/bash
# after `gpurun -- bash scripts/r6_final.sh`: gpurun_out/r6_final + gpurun_out/prof_r6f_* + gpurun_out/screen_r6f -> profiles/r6_final
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
O=gpurun_out/r6_final; D=profiles/r6_final; mkdir -p $D
for w in nsq24 rts96 seq; do python scripts/summarize_profile.py r6f r6_final $w | tail -1 | cut -c1-60; done
cp $O/pytest_gpu.log $O/golden_pin.log $O/converged.log $O/wave_tail.log $D/
for f in bench_default bench_driver_shape bench_rts96 bench_seq bench_2rank_shared bench_8rank_shared bench_8rank_strong_1e8 bench_8rank_seq_1000y bench_8rank_rts96_1e7; do
  grep '^{' $O/$f.json | tail -1 | python -m json.tool > $D/${f}_full.json
done
for w in rts24 rts96 seq; do f=$(find gpurun_out/screen_r6f/$w -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $D/kernel_stats_screened_$w.csv; done
cp $O/numfail96_device.json $D/ 2>/dev/null
echo r6_final > profiles/current.txt
