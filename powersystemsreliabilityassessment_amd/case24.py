"""IEEE RTS-24 study case for the HL2 non-sequential Monte Carlo path.

Host-side data model:

* network data  = MATPOWER ``case24_ieee_rts`` as loaded at
  ``Montecarlo_nsq_single/nsqMain.m:42`` (the file itself is NOT in the reference
  repo; restated from the public IEEE RTS-79 data);
* reliability data = ``Montecarlo_nsq_single/case24_failrate.m:23-78`` verbatim,
  including the ``brdur`` ordering quirk (positions 6 and 9 hold 768 and 35);
* ``failprob()``  = ``Montecarlo_nsq_single/failprob.m:1-41``;
* ``dispatchable_load_model()`` = the load -> virtual-generator conversion of
  ``nsqMain.m:121-153``.

Everything here is plain numpy; the arrays are handed to the HIP library
through ``relmc_case_load`` (include/relmc.h).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

BASE_MVA = 100.0
REF_BUS = 13  # 1-based; MATPOWER bus type 3 in case24_ieee_rts
SYNC_COMP_INDEX = 15  # 1-based generator row forced up, mc_sampling.m:40-41

# bus Pd (MW), buses 1..24 (same numbers as Montecarlo_seq/case24_loadprofile.m:77-95)
BUS_PD = np.array(
    [108, 97, 180, 74, 71, 136, 125, 171, 175, 195, 0, 0,
     265, 194, 317, 100, 0, 333, 181, 128, 0, 0, 0, 0], dtype=np.float64)

# generators: (bus, Pmax, Pmin), 33 rows in case24_ieee_rts order
_GEN_ROWS = (
    [(1, 20, 16)] * 2 + [(1, 76, 15.2)] * 2 +
    [(2, 20, 16)] * 2 + [(2, 76, 15.2)] * 2 +
    [(7, 100, 25)] * 3 +
    [(13, 197, 69)] * 3 +
    [(14, 0, 0)] +                      # synchronous condenser (row 15)
    [(15, 12, 2.4)] * 5 + [(15, 155, 54.3)] +
    [(16, 155, 54.3)] +
    [(18, 400, 100)] +
    [(21, 400, 100)] +
    [(22, 50, 10)] * 6 +
    [(23, 155, 54.3)] * 2 + [(23, 350, 140)]
)
GEN_BUS = np.array([r[0] for r in _GEN_ROWS], dtype=np.int32)
GEN_PMAX = np.array([r[1] for r in _GEN_ROWS], dtype=np.float64)
GEN_PMIN = np.array([r[2] for r in _GEN_ROWS], dtype=np.float64)

# branches: (from, to, x p.u., rateA MW, tap ratio (0 = line)), 38 rows
_BRANCH_ROWS = [
    (1, 2, .0139, 175, 0), (1, 3, .2112, 175, 0), (1, 5, .0845, 175, 0),
    (2, 4, .1267, 175, 0), (2, 6, .1920, 175, 0), (3, 9, .1190, 175, 0),
    (3, 24, .0839, 400, 1.03), (4, 9, .1037, 175, 0), (5, 10, .0883, 175, 0),
    (6, 10, .0605, 175, 0), (7, 8, .0614, 175, 0), (8, 9, .1651, 175, 0),
    (8, 10, .1651, 175, 0), (9, 11, .0839, 400, 1.03), (9, 12, .0839, 400, 1.03),
    (10, 11, .0839, 400, 1.02), (10, 12, .0839, 400, 1.02), (11, 13, .0476, 500, 0),
    (11, 14, .0418, 500, 0), (12, 13, .0476, 500, 0), (12, 23, .0966, 500, 0),
    (13, 23, .0865, 500, 0), (14, 16, .0389, 500, 0), (15, 16, .0173, 500, 0),
    (15, 21, .0490, 500, 0), (15, 21, .0490, 500, 0), (15, 24, .0519, 500, 0),
    (16, 17, .0259, 500, 0), (16, 19, .0231, 500, 0), (17, 18, .0144, 500, 0),
    (17, 22, .1053, 500, 0), (18, 21, .0259, 500, 0), (18, 21, .0259, 500, 0),
    (19, 20, .0396, 500, 0), (19, 20, .0396, 500, 0), (20, 23, .0216, 500, 0),
    (20, 23, .0216, 500, 0), (21, 22, .0678, 500, 0),
]
BR_FROM = np.array([r[0] for r in _BRANCH_ROWS], dtype=np.int32)
BR_TO = np.array([r[1] for r in _BRANCH_ROWS], dtype=np.int32)
BR_X = np.array([r[2] for r in _BRANCH_ROWS], dtype=np.float64)
BR_RATE = np.array([r[3] for r in _BRANCH_ROWS], dtype=np.float64)
BR_TAP = np.array([r[4] for r in _BRANCH_ROWS], dtype=np.float64)


def case24_failrate() -> dict:
    """Reliability parameters, verbatim from case24_failrate.m:23-78."""
    genmttf = np.array([
        450, 450, 1960, 1960, 450,
        450, 1960, 1960, 1200, 1200,
        1200, 950, 950, 950, 10000,
        2940, 2940, 2940, 2940, 2940,
        960, 960, 1100, 1100, 1980,
        1980, 1980, 1980, 1980, 1980,
        960, 960, 1150], dtype=np.float64)
    genmttr = np.array([
        50, 50, 40, 40, 50,
        50, 40, 40, 50, 50,
        50, 50, 50, 50, 0.1,
        60, 60, 60, 60, 60,
        40, 40, 150, 150, 20,
        20, 20, 20, 20, 20,
        40, 40, 100], dtype=np.float64)
    genweeks = np.array([
        2, 2, 3, 3, 2,
        2, 3, 3, 3, 3,
        3, 4, 4, 4, 0.1,
        2, 2, 2, 2, 2,
        4, 4, 6, 6, 2,
        2, 2, 2, 2, 2,
        4, 4, 5], dtype=np.float64)
    brlambda = np.array([
        0.24, 0.51, 0.33, 0.39, 0.48, 0.38,
        0.02, 0.36, 0.34, 0.33, 0.30, 0.44,
        0.44, 0.02, 0.02, 0.02, 0.02, 0.40,
        0.39, 0.40, 0.52, 0.49, 0.38, 0.33,
        0.41, 0.41, 0.41, 0.35, 0.34, 0.32,
        0.54, 0.35, 0.35, 0.38, 0.38, 0.34,
        0.34, 0.45], dtype=np.float64)
    # NOTE: order kept exactly as in the reference
    brdur = np.array([
        16, 10, 10, 10, 10, 768, 10, 10, 35, 10, 10, 10,
        10, 768, 768, 768, 768, 11, 11, 11, 11, 11, 11, 11,
        11, 11, 11, 11, 11, 11, 11, 11, 11, 11, 11, 11,
        11, 11], dtype=np.float64)
    return dict(genmttf=genmttf, genmttr=genmttr, genweeks=genweeks,
                brlambda=brlambda, brdur=brdur)


def failprob() -> np.ndarray:
    """Unavailability vector [gens (33), branches (38)] — failprob.m:23,31-35,39."""
    d = case24_failrate()
    prob_gen = d["genmttr"] / (d["genmttf"] + d["genmttr"])
    branch_mu = 8760.0 / d["brdur"]
    prob_branch = d["brlambda"] / (d["brlambda"] + branch_mu)
    return np.concatenate([prob_gen, prob_branch])


@dataclass
class Case:
    """Flat description of one study case, the argument of relmc_case_load.

    All indices are 0-based here (the reference's MATLAB files are 1-based).
    ``inj_*`` lists the LP's injection variables in MATPOWER generator order:
    the ``ng`` real generators followed by the ``nd`` virtual generators that
    model the loads (nsqMain.m:136-150).
    """
    base_mva: float
    nb: int
    ng: int
    nl: int
    nd: int
    ref_bus: int
    bus_pd: np.ndarray        # [nb] MW (original loads; zeroed in the OPF model, nsqMain.m:153)
    inj_bus: np.ndarray       # [ng+nd] int32
    inj_pmin: np.ndarray      # [ng+nd] MW
    inj_pmax: np.ndarray      # [ng+nd] MW
    inj_cost: np.ndarray      # [ng+nd] $/MWh (0 real, 1 virtual; nsqMain.m:128,132-133)
    br_from: np.ndarray       # [nl] int32
    br_to: np.ndarray         # [nl] int32
    br_b: np.ndarray          # [nl] susceptance 1/(x*tap) p.u. (makeBdc)
    br_rate: np.ndarray       # [nl] MW (0 = unconstrained)
    unavail: np.ndarray       # [ng+nl] failure probabilities (failprob.m)
    always_up: np.ndarray     # [ng+nl] uint8, 1 = never sampled as failed (mc_sampling.m:40-41)
    total_load: float = field(default=0.0)  # TestSystem.load, nsqMain.m:125
    elim_order: np.ndarray | None = None    # optional schedule tunable: primary elimination order of the solver (0-based buses, reference bus last; relmc_case_order_hint)

    @property
    def ncomp(self) -> int:
        return self.ng + self.nl

    @property
    def ninj(self) -> int:
        return self.ng + self.nd


def dispatchable_load_model(bus_pd: np.ndarray):
    """nsqMain.m:121-150: one virtual generator per load bus, Pg=-Pd, Pmax=0, Pmin=-Pd."""
    load_buses = np.flatnonzero(bus_pd != 0)          # nsqMain.m:121
    pmin = -bus_pd[load_buses]
    pmax = np.zeros(load_buses.size)
    return load_buses.astype(np.int32), pmin, pmax


# Primary elimination order of the device solver's static schedule for this network (0-based bus numbers, reference bus last), tuned offline
# against the library's own scheduler: `python scripts/order_tune.py rts24 11 60000` = relmc_tune_order(evaluations=60000, seed=11) from the
# built-in rule's order: 174 -> 168 LDS instructions per Newton step, 21 dependent passes as before, kernel 17.75 -> 17.50 ms per 1e6 scenarios.
# An order changes only the rounding of the factorisation, but this LP's optimal face is degenerate, so it can move a state's nodal split:
# of 18 orders tuned this way (seeds 1-20; profiles/r3_order/select24.log) 4 move the 0.3 percent-mass fixture state "G24 + G33 out" by one
# iteration and 8 move bus 8's nodal sum of 3e5 sampled states by 0.2-0.4 percent against the C oracle; 6 keep both pins (every fixture
# state's iteration count, every bus' nodal sum within 6e-4 -- the rule's order: 3.3e-4), and the fastest of those is this one (3.8e-4).
RTS24_ELIM_ORDER = np.array([4, 21, 18, 5, 19, 17, 3, 6, 22, 23, 1, 20, 2, 11, 13, 16, 10, 0, 14, 15, 7, 8, 9, 12], dtype=np.int32)


def rts24() -> Case:
    """The case the reference's nsqMain builds before its Monte Carlo loop."""
    load_buses, vpmin, vpmax = dispatchable_load_model(BUS_PD)
    tap = np.where(BR_TAP == 0, 1.0, BR_TAP)
    always_up = np.zeros(GEN_BUS.size + BR_X.size, dtype=np.uint8)
    always_up[SYNC_COMP_INDEX - 1] = 1
    return Case(
        base_mva=BASE_MVA, nb=24, ng=int(GEN_BUS.size), nl=int(BR_X.size),
        nd=int(load_buses.size), ref_bus=REF_BUS - 1,
        bus_pd=BUS_PD.copy(),
        inj_bus=np.concatenate([GEN_BUS - 1, load_buses]).astype(np.int32),
        inj_pmin=np.concatenate([GEN_PMIN, vpmin]),
        inj_pmax=np.concatenate([GEN_PMAX, vpmax]),
        inj_cost=np.concatenate([np.zeros(GEN_BUS.size), np.ones(load_buses.size)]),
        br_from=(BR_FROM - 1).astype(np.int32), br_to=(BR_TO - 1).astype(np.int32),
        br_b=1.0 / (BR_X * tap), br_rate=BR_RATE.copy(),
        unavail=failprob(), always_up=always_up,
        total_load=float(BUS_PD.sum()),
        elim_order=RTS24_ELIM_ORDER.copy(),
    )


def thresholds_u32(case: Case) -> np.ndarray:
    """Integer Bernoulli thresholds: component k fails iff draw_u32 < floor(U_k * 2^32).

    Strict '<' as in mc_sampling.m:35; always-up components get threshold 0
    (mc_sampling.m:40-41).
    """
    t = np.floor(case.unavail * 4294967296.0)
    t = np.clip(t, 0, 4294967295.0).astype(np.uint64)
    t[case.always_up != 0] = 0
    return t.astype(np.uint32)
