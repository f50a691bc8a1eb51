"""IEEE RTS-96 (three-area, 73-bus) study case — BASELINE config 5, SURVEY.md §8f rank 3 / Appendix F.

The reference repository has no RTS-96 data; SURVEY Appendix F gives the construction (restated from Grigg et al.,
"The IEEE Reliability Test System-1996", IEEE T-PWRS 14(3), 1999) and this module follows it:

* three copies of RTS-24 (areas 1xx / 2xx / 3xx) plus bus 325  -> 73 buses; bus ``100*a + i`` is index ``24*(a-1) + i - 1``,
  bus 325 is index 72;
* branches: the 38 RTS-24 branches of area 1, of area 2, of area 3, then the five ties and the 323-325 transformer
  (x p.u., rating MW): 107-203 (0.161, 175), 113-215 (0.075, 500), 123-217 (0.074, 500), 325-121 (0.097, 500),
  318-223 (0.104, 500), 323-325 (0.009, 722)  -> 120 branches;
* generators: the 33 RTS-24 rows per area (32 units + the synchronous condenser; always-up rows 15, 48, 81) -> 99 rows;
* loads: the RTS-24 bus loads in every area -> 51 load buses, 8550 MW; one reference bus for the whole system (113);
* reliability data: the RTS-24 vectors of ``case24_failrate.m`` for every area, *including* its ``brdur`` ordering
  quirk (so that an area of RTS-96 behaves like the reference's RTS-24).  Tie-line outage rates follow the RTS-79
  mileage law visible in the RTS-24 table (138 kV: 0.22 + 0.0052/mi, 230 kV: 0.283 + 0.0035/mi) with the RTS-96 tie
  lengths 42, 52, 51, 67, 72 mi: 0.44, 0.47, 0.46, 0.52, 0.54 per year, durations 10 h (138 kV) / 11 h (230 kV);
  the 323-325 transformer uses the RTS transformer values 0.02 per year, 768 h.  [RECALLED/derived, documented
  here because the reference offers nothing to pin them; they only set component unavailabilities.]
"""
from __future__ import annotations

import numpy as np

from . import case24
from .case24 import Case

N_AREAS = 3
REF_BUS_INDEX = 12                      # bus 113
# (from bus number, to bus number, x p.u., rating MW, lambda 1/yr, duration h)
TIES = (
    (107, 203, 0.161, 175.0, 0.44, 10.0),
    (113, 215, 0.075, 500.0, 0.47, 11.0),
    (123, 217, 0.074, 500.0, 0.46, 11.0),
    (325, 121, 0.097, 500.0, 0.52, 11.0),
    (318, 223, 0.104, 500.0, 0.54, 11.0),
    (323, 325, 0.009, 722.0, 0.02, 768.0),
)


def bus_index(number: int) -> int:
    """RTS-96 bus number (101..124, 201..224, 301..325) -> 0-based index."""
    area, i = divmod(number, 100)
    return 72 if number == 325 else 24 * (area - 1) + i - 1


def failrate96() -> dict:
    d = case24.case24_failrate()
    tie_l = np.array([t[4] for t in TIES]); tie_r = np.array([t[5] for t in TIES])
    return dict(genmttf=np.tile(d["genmttf"], N_AREAS), genmttr=np.tile(d["genmttr"], N_AREAS),
                brlambda=np.concatenate([np.tile(d["brlambda"], N_AREAS), tie_l]),
                brdur=np.concatenate([np.tile(d["brdur"], N_AREAS), tie_r]))


def failprob96() -> np.ndarray:
    """failprob.m:23,31-35,39 on the RTS-96 vectors: [99 generator rows, 120 branches]."""
    d = failrate96()
    pg = d["genmttr"] / (d["genmttf"] + d["genmttr"])
    pb = d["brlambda"] / (d["brlambda"] + 8760.0 / d["brdur"])
    return np.concatenate([pg, pb])


def seqmeantime96() -> np.ndarray:
    """seqmeantime.m:21-36 on the RTS-96 vectors."""
    d = failrate96()
    return np.column_stack([np.concatenate([d["genmttf"], 8760.0 / d["brlambda"]]), np.concatenate([d["genmttr"], d["brdur"]])])


# Primary elimination order of the device solver's static schedule (0-based bus numbers, reference bus last), tuned offline against the
# library's own scheduler: scripts/order_search.py (critical-path model, seed 1: 19 -> 17 update passes) refined by
# `python scripts/order_tune.py rts96 1 60000 <that order>` = relmc_tune_order(evaluations=60000, seed=1, start=...).
# 215 -> 200 LDS instructions per Newton step, 32 -> 28 dependent passes (update 19 -> 16, back substitution 11 -> 10); kernel 73.7 -> 71.6 ms
# per 1e6 scenarios; the primary order fails about as often as the rule's (107-118 against 110 units per 2e8 samples, all converge further on).
RTS96_ELIM_ORDER = np.array([69, 41, 27, 3, 61, 2, 67, 42, 4, 17, 47, 60, 11, 53, 66, 18, 36, 54, 43, 5, 52, 71, 24, 51, 59, 64, 30, 44, 21, 6, 62, 29, 37, 14, 58, 48, 35, 55, 7, 49, 28, 68, 1, 34, 13, 25, 19, 31, 45, 40, 23, 56, 16, 38, 72, 65, 0, 57, 10, 33, 32, 39, 63, 15, 50, 9, 8, 70, 26, 22, 20, 46, 12], dtype=np.int32)


def rts96() -> Case:
    nb = 24 * N_AREAS + 1
    bus_pd = np.concatenate([np.tile(case24.BUS_PD, N_AREAS), [0.0]])
    gen_bus = np.concatenate([case24.GEN_BUS - 1 + 24 * a for a in range(N_AREAS)]).astype(np.int32)
    gen_pmax = np.tile(case24.GEN_PMAX, N_AREAS); gen_pmin = np.tile(case24.GEN_PMIN, N_AREAS)
    tap = np.where(case24.BR_TAP == 0, 1.0, case24.BR_TAP)
    br_from = np.concatenate([case24.BR_FROM - 1 + 24 * a for a in range(N_AREAS)] + [[bus_index(t[0]) for t in TIES]]).astype(np.int32)
    br_to = np.concatenate([case24.BR_TO - 1 + 24 * a for a in range(N_AREAS)] + [[bus_index(t[1]) for t in TIES]]).astype(np.int32)
    br_b = np.concatenate([np.tile(1.0 / (case24.BR_X * tap), N_AREAS), [1.0 / t[2] for t in TIES]])
    br_rate = np.concatenate([np.tile(case24.BR_RATE, N_AREAS), [t[3] for t in TIES]])
    load_buses, vpmin, vpmax = case24.dispatchable_load_model(bus_pd)
    ng, nl = gen_bus.size, br_b.size
    always_up = np.zeros(ng + nl, dtype=np.uint8)
    always_up[[case24.SYNC_COMP_INDEX - 1 + 33 * a for a in range(N_AREAS)]] = 1
    return Case(
        base_mva=case24.BASE_MVA, nb=nb, ng=int(ng), nl=int(nl), nd=int(load_buses.size), ref_bus=REF_BUS_INDEX,
        bus_pd=bus_pd, inj_bus=np.concatenate([gen_bus, load_buses]).astype(np.int32),
        inj_pmin=np.concatenate([gen_pmin, vpmin]), inj_pmax=np.concatenate([gen_pmax, vpmax]),
        inj_cost=np.concatenate([np.zeros(ng), np.ones(load_buses.size)]),
        br_from=br_from, br_to=br_to, br_b=br_b, br_rate=br_rate,
        unavail=failprob96(), always_up=always_up, total_load=float(bus_pd.sum()), elim_order=RTS96_ELIM_ORDER.copy())
