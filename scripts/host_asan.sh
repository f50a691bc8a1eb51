#!/bin/bash
# The library's host code under ASan + UBSan on the CPU box (csrc/Makefile: host-asan): the CPU tests that drive the scheduler, the placement
# search, the order tuner, the symbolic models (relmc_debug_symbolic), the estimators, option / error paths and the loaders, run against
# ablate/librelmc_hostasan.so with the sanitizer runtime preloaded into python.  Leak checking is off (python itself never frees at exit).
#   bash scripts/host_asan.sh [log]
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
LOG=${1:-profiles/r6_final/host_asan.log}
make -C powersystemsreliabilityassessment_amd/csrc host-asan > /tmp/host_asan_build.log 2>&1 || { tail -20 /tmp/host_asan_build.log; exit 1; }
RT=$(/opt/rocm/bin/hipcc --offload-arch=gfx950 -print-file-name=libclang_rt.asan-x86_64.so)
export RELMC_LIB_PATH=$R/powersystemsreliabilityassessment_amd/csrc/ablate/librelmc_hostasan.so
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1:detect_stack_use_after_return=1:strict_string_checks=1
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
{
  echo "# host code of librelmc under -fsanitize=address,undefined (device code not instrumented), $(date -u +%FT%TZ)"
  echo "# library $RELMC_LIB_PATH, runtime $RT"
  LD_PRELOAD=$RT python -m pytest tests/test_schedule.py tests/test_host.py tests/test_matpower.py tests/test_c_abi.py tests/test_screen.py -q -m "not gpu" -p no:cacheprovider 2>&1
  echo "pytest rc $?"
  # the order tuner on both shipped cases (thousands of schedules built and costed) and the symbolic models of the why-not page
  echo "## order tuner, RTS-24 (3000 evaluations) and RTS-96 (600)"
  LD_PRELOAD=$RT timeout 1800 python scripts/order_tune.py rts24 11 3000 2>&1 | tail -3; echo "rc ${PIPESTATUS[0]}"
  LD_PRELOAD=$RT timeout 1800 python scripts/order_tune.py rts96 11 600 2>&1 | tail -3; echo "rc ${PIPESTATUS[0]}"
  echo "## relmc_debug_symbolic model paths (typed passes are the dev build's; the leaf-free model is an argument of the shipped entry point)"
  LD_PRELOAD=$RT timeout 900 python - <<'PY' 2>&1 | tail -8
import sys; sys.path.insert(0, ".")
from powersystemsreliabilityassessment_amd import case24, case96
from tests import schedule_interp as si
for name, case in (("rts24", case24.rts24()), ("rts96", case96.rts96())):
    for variant, leaf in ((0, -1), (1, -1), (2, -1), (0, 0), (0, 1)):
        s = si.symbolic(case, variant, case.elim_order if variant == 0 else None, model_leaf_free=leaf)
        print(name, "variant", variant, "leaf-free model", leaf, "passes", s.npass, "tasks", int((s.tasks[..., 0] != 0xffff).sum()))
PY
  echo "rc ${PIPESTATUS[0]}"
} > $LOG 2>&1
grep -n "ERROR: AddressSanitizer\|runtime error\|pytest rc\|^rc \|passed\|failed" $LOG
