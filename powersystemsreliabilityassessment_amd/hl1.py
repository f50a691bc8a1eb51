"""HL1 (copper-sheet) generating-adequacy track — SURVEY.md §8f rank 1, BASELINE config 1.

Mirror of the reference's Julia module GeneratingAdequacy/PowerSystemAdequacy.jl:
  Generator / LoadModel / ReliabilityResult   (:20-52)
  run_analytical(gens, load; step_size)       (:113-163)  exact COPT convolution, host arithmetic (numpy)
  run_non_sequential_mc(gens, load, iterations)(:169-208)  Monte Carlo, evaluated by the HIP library
                                                            (relmc_hl1_load / relmc_hl1_nsq)
`rts24_generators()` / `rts24_load()` give the IEEE RTS-79 fleet and the 8736-hour reference load
curve (Montecarlo_seq/anloducurve.m) whose exact answers are the published LOLE 9.3941 h/yr and
EUE 1176.29 MWh/yr.
"""
from __future__ import annotations

import ctypes as C
import time
from dataclasses import dataclass, field

import numpy as np

from . import _abi, case24, loadcurve


@dataclass
class Generator:                      # PowerSystemAdequacy.jl:20-37
    id: int
    capacity: float
    mttf: float
    mttr: float

    @property
    def for_rate(self) -> float:
        lam, mu = 1.0 / self.mttf, 1.0 / self.mttr
        return lam / (lam + mu)


@dataclass
class LoadModel:                      # :39-45
    hourly_load: np.ndarray

    @property
    def peak_load(self) -> float:
        return float(np.max(self.hourly_load))


@dataclass
class ReliabilityResult:              # :47-53
    method: str
    lole_hours_yr: float
    eue_mwh_yr: float
    computation_time: float
    convergence_history: np.ndarray = field(default_factory=lambda: np.zeros(0))


def rts24_generators() -> list:
    d = case24.case24_failrate()
    return [Generator(i + 1, float(case24.GEN_PMAX[i]), float(d["genmttf"][i]), float(d["genmttr"][i]))
            for i in range(case24.GEN_PMAX.size) if case24.GEN_PMAX[i] > 0]     # 32 units (sync condenser has no capacity)


def rts24_load(hours: int = 8736) -> LoadModel:
    _, _, lf = loadcurve.anloducurve(hours)
    return LoadModel(2850.0 * lf)


def add_unit_convolution(probs: np.ndarray, unit: Generator, step_size: float) -> np.ndarray:
    """COPT recursion with capacity rounding split between neighbouring steps (:67-111).
    `probs[k]` = P(outage = k*step_size)."""
    Cc, q = unit.capacity, unit.for_rate
    p = 1.0 - q
    max_old = (probs.size - 1) * step_size if probs.size else 0.0
    n_new = int(np.ceil((max_old + Cc) / step_size)) + 1
    new = np.zeros(n_new)

    def shifted(k):            # get_prob(X - k*step) for all X
        out = np.zeros(n_new)
        if k < n_new:
            m = min(probs.size, n_new - k)
            out[k:k + m] = probs[:m]
        return out

    lower = int(np.floor(Cc / step_size))
    if abs(Cc - lower * step_size) < 1e-5:
        new = shifted(0) * p + shifted(lower) * q
    else:
        alpha = (Cc - lower * step_size) / step_size
        new = shifted(0) * p + shifted(lower) * (q * (1.0 - alpha)) + shifted(lower + 1) * (q * alpha)
    return new


def run_analytical(gens, load: LoadModel, step_size: float = 10.0) -> ReliabilityResult:
    """Exact (up to the capacity step) LOLE / EUE by convolution (:113-163)."""
    t0 = time.time()
    probs = np.array([1.0])
    for g in gens:
        probs = add_unit_convolution(probs, g, step_size)
    outage = np.arange(probs.size) * step_size
    installed = sum(g.capacity for g in gens)
    cum = np.cumsum(probs[::-1])[::-1]                         # P(outage >= X)
    tail_w = np.cumsum((outage * probs)[::-1])[::-1]           # sum_{k>=i} outage_k p_k
    lole = eue = 0.0
    for load_mw in np.asarray(load.hourly_load, dtype=float):
        reserve = installed - load_mw
        idx = int(np.floor(reserve / step_size)) + 1           # 0-based index of the first state with outage > reserve
        if 0 <= idx < probs.size:
            lole += cum[idx]
            eue += tail_w[idx] - reserve * cum[idx]
        elif idx < 0:
            lole += 1.0
            eue += (load_mw - installed) + float(outage @ probs)
    return ReliabilityResult("Analytical", lole, eue, time.time() - t0)


def run_non_sequential_mc(gens, load: LoadModel, iterations: int, *, seed: int = 1, engine=None) -> ReliabilityResult:
    """PowerSystemAdequacy.jl:169-208 on the GPU: one fleet state per iteration swept over the whole
    hourly load curve; convergence history = running LOLE every 100 iterations (:202-204)."""
    from . import api
    eng = engine or api.default_engine()
    L = eng.L
    L.relmc_hl1_load.argtypes = [C.c_void_p, C.c_int32, _abi.c_double_p, _abi.c_double_p, C.c_int32, _abi.c_double_p]
    L.relmc_hl1_nsq.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int64, C.POINTER(Hl1Acc), _abi.c_double_p, _abi.c_double_p]
    t0 = time.time()
    cap = np.ascontiguousarray([g.capacity for g in gens], dtype=np.float64)
    forr = np.ascontiguousarray([g.for_rate for g in gens], dtype=np.float64)
    hl = np.ascontiguousarray(load.hourly_load, dtype=np.float64)
    key = (cap.tobytes(), forr.tobytes(), hl.tobytes())
    if getattr(eng, "_hl1_loaded", None) != key:           # the fleet and the sorted load curve stay on the device between calls on the same model
        eng._check(L.relmc_hl1_load(eng._h, cap.size, cap.ctypes.data_as(_abi.c_double_p), forr.ctypes.data_as(_abi.c_double_p),
                                    hl.size, hl.ctypes.data_as(_abi.c_double_p)), "relmc_hl1_load")
        eng._hl1_loaded = key
    acc = Hl1Acc()
    it_lole = np.zeros(iterations)
    eng._check(L.relmc_hl1_nsq(eng._h, int(seed), 0, int(iterations), C.byref(acc), it_lole.ctypes.data_as(_abi.c_double_p), None),
               "relmc_hl1_nsq")
    k = np.arange(100, iterations + 1, 100)
    history = np.cumsum(it_lole)[k - 1] / k if k.size else np.zeros(0)
    return ReliabilityResult("Non-Sequential MC", acc.sum_lole / iterations, acc.sum_eue / iterations, time.time() - t0, history)


class Hl1Acc(C.Structure):
    _fields_ = [("n", C.c_int64), ("sum_lole", C.c_double), ("sum_eue", C.c_double),
                ("sum_lole2", C.c_double), ("sum_eue2", C.c_double)]
