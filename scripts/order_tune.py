"""Tunes the primary elimination order of a case against the library's own scheduler (relmc_tune_order: host only, no GPU):
    python scripts/order_tune.py rts96 <seed> <evaluations> [start order "a,b,c,..."]
prints the LDS instructions per Newton step and the dependent passes before / after and the order (external bus numbers, 0-based, the
reference bus last) -- what `case24.RTS24_ELIM_ORDER` / `case96.RTS96_ELIM_ORDER` hold.  Developer tool (round 3)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from powersystemsreliabilityassessment_amd import api, case24, case96

if __name__ == "__main__":
    name = sys.argv[1]; seed = int(sys.argv[2]); evals = int(sys.argv[3])
    case = case24.rts24() if name == "rts24" else case96.rts96()
    start = [int(v) for v in sys.argv[4].split(",")] if len(sys.argv) > 4 else None
    t = time.time()
    order, st = api.tune_order(case, evals, seed, start)
    print("%s seed %d, %d evaluations in %.1f s: LDS instructions per Newton step %d -> %d, dependent passes %d -> %d" % (
        name, seed, evals, time.time() - t, st["lds_before"], st["lds_after"], st["passes_before"], st["passes_after"]))
    print("order=" + ",".join(str(int(v)) for v in order), flush=True)
