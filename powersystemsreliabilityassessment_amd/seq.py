"""Sequential (chronological) HL2 track — SURVEY.md §8f rank 2, BASELINE config 4.

Host-side mirror of /root/reference/Montecarlo_seq/:
  seqmeantime()                                   seqmeantime.m:21-36   [MTTF MTTR] matrix
  seq_mcsampling(rel, Ng, Nl, num_years, hours)   seq_mcsampling.m:2    chronological up/down histories
  seq_mcsimulation(status, load_scale, ...)       seq_mcsimulation.m:1  hourly DC-OPF with scaled loads (batched)
  calnlc(series)                                  calnlc.m:22-32        number of loss events
  seqMain(...)                                    seqMain.m:85-262      yearly loop, CoV stop, post-processing
All evaluation runs in the HIP library (relmc_seq_* of include/relmc.h).  Deliberate differences:
counter-based RNG keyed by (seed, global year, component, event); every year is sampled independently
starting all-up, which is what seqMain.m:91 does (it calls seq_mcsampling with num_years = 1).
"""
from __future__ import annotations

import ctypes as C
import time
from dataclasses import dataclass, field

import numpy as np

from . import _abi, api, case24, loadcurve

HOURS_PER_YEAR = 8736        # seqMain.m:38
COV_THRESHOLD = 0.05         # seqMain.m:40
CURTAIL_THRESHOLD = 0.01     # seqMain.m:41
MAX_SIM_YEARS = 4000         # seqMain.m:39


SeqYear = _abi.SeqYear


def seqmeantime() -> np.ndarray:
    """[(Ng+Nl) x 2] = [MTTF, MTTR]; branches: MTTF = 8760/lambda, MTTR = r  (seqmeantime.m:27-28)."""
    d = case24.case24_failrate()
    mttf = np.concatenate([d["genmttf"], 8760.0 / d["brlambda"]])
    mttr = np.concatenate([d["genmttr"], d["brdur"]])
    return np.column_stack([mttf, mttr])


def calnlc(series) -> int:
    """Count 0->1 starts of the loss flag, the first hour counts (calnlc.m:22-32)."""
    s = np.asarray(series).astype(int)
    return int((np.diff(s) == 1).sum() + (1 if s.size and s[0] == 1 else 0))


def _bind(L):
    dp, u8p, i32p = _abi.c_double_p, _abi.c_uint8_p, _abi.c_int32_p
    L.relmc_seq_load.argtypes = [C.c_void_p, dp, dp, C.c_int32, dp]
    L.relmc_seq_mcsampling.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int32, u8p]
    L.relmc_seq_mcsimulation.argtypes = [C.c_void_p, u8p, dp, C.c_int64, C.POINTER(_abi.SolverOpts), dp, dp, i32p, i32p]
    L.relmc_seq_years.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int32, C.POINTER(_abi.SolverOpts), C.c_double,
                                  C.POINTER(SeqYear), C.POINTER(_abi.Acc)]
    L.relmc_seq_opts_default.argtypes = [C.POINTER(_abi.SeqOpts)]
    L.relmc_seq_opts_default.restype = None
    L.relmc_seq_run.argtypes = [C.c_void_p, C.POINTER(_abi.SeqOpts), C.POINTER(_abi.SeqResult)]
    L.relmc_seq_run.restype = C.c_int32


class SeqEngine:
    """Sequential-track front end of one api.Engine (one GPU)."""

    def __init__(self, engine: api.Engine | None = None, reliability_data=None, hours_per_year: int = HOURS_PER_YEAR,
                 load_scale_factors=None):
        self.eng = engine or api.default_engine()
        self.L = self.eng.L
        _bind(self.L)
        self.rel = np.ascontiguousarray(seqmeantime() if reliability_data is None else reliability_data, dtype=np.float64)
        self.hours = int(hours_per_year)
        if load_scale_factors is None:
            _, _, load_scale_factors = loadcurve.anloducurve(self.hours)        # seqMain.m:67
        self.load_factors = np.ascontiguousarray(load_scale_factors, dtype=np.float64)
        mttf = np.ascontiguousarray(self.rel[:, 0]); mttr = np.ascontiguousarray(self.rel[:, 1])
        self.eng._check(self.L.relmc_seq_load(self.eng._h, mttf.ctypes.data_as(_abi.c_double_p), mttr.ctypes.data_as(_abi.c_double_p),
                                              self.hours, self.load_factors.ctypes.data_as(_abi.c_double_p)), "relmc_seq_load")

    # seq_mcsampling.m:2 — returns [(Ng+Nl) x (num_years*hours)] like the reference (1 = down)
    def seq_mcsampling(self, reliability_data=None, numGenerators=None, numLines=None, num_years: int = 1,
                       hours_per_year: int | None = None, *, seed: int = 1, first_year: int = 0) -> np.ndarray:
        if hours_per_year is not None and int(hours_per_year) != self.hours:
            raise ValueError("hours_per_year differs from the loaded chronology")
        if reliability_data is not None and not np.array_equal(np.asarray(reliability_data, dtype=float), self.rel):
            raise ValueError("reliability_data differs from the loaded [MTTF MTTR] matrix (build a new SeqEngine)")
        nc = self.eng.case.ncomp
        out = np.zeros((int(num_years) * self.hours, nc), dtype=np.uint8)
        self.eng._check(self.L.relmc_seq_mcsampling(self.eng._h, int(seed), int(first_year), int(num_years),
                                                    out.ctypes.data_as(_abi.c_uint8_p)), "relmc_seq_mcsampling")
        return out.T

    # seq_mcsimulation.m:1, batched over hours
    def seq_mcsimulation(self, component_status, load_scale_factor, TestSystem=None, mpopt=None, load_bus_indices=None,
                         numGenerators=None, numLines=None, *, return_info: bool = False):
        nc = self.eng.case.ncomp
        st = np.asarray(component_status)
        single = st.ndim == 1
        st = np.ascontiguousarray(st.reshape(-1, nc) != 0, dtype=np.uint8)
        n = st.shape[0]
        sc = np.ascontiguousarray(np.broadcast_to(np.asarray(load_scale_factor, dtype=np.float64), (n,)))
        o = mpopt if mpopt is not None else api.mpoption()
        dns = np.zeros(n); nodal = np.zeros((n, self.eng.case.nb)); status = np.zeros(n, dtype=np.int32); iters = np.zeros(n, dtype=np.int32)
        self.eng._check(self.L.relmc_seq_mcsimulation(self.eng._h, st.ctypes.data_as(_abi.c_uint8_p), sc.ctypes.data_as(_abi.c_double_p), n,
                                                      C.byref(o), dns.ctypes.data_as(_abi.c_double_p), nodal.ctypes.data_as(_abi.c_double_p),
                                                      status.ctypes.data_as(_abi.c_int32_p), iters.ctypes.data_as(_abi.c_int32_p)),
                        "relmc_seq_mcsimulation")
        res = (float(dns[0]), nodal[0]) if single else (dns, nodal)
        return res + (dict(status=status, iters=iters),) if return_info else res

    def seq_years(self, seed: int, first_year: int, n_years: int, mpopt=None, curtail_threshold: float = CURTAIL_THRESHOLD):
        """Fused evaluation of whole simulated years: (ens[n], dlc[n], nlc[n], n_contingency[n], Acc)."""
        o = mpopt if mpopt is not None else api.mpoption()
        yrs = (SeqYear * int(n_years))()
        acc = _abi.Acc()
        self.eng._check(self.L.relmc_seq_years(self.eng._h, int(seed), int(first_year), int(n_years), C.byref(o), float(curtail_threshold),
                                               yrs, C.byref(acc)), "relmc_seq_years")
        raw = np.frombuffer(yrs, dtype=np.dtype([("ens", "<f8"), ("dlc", "<f8"), ("nlc", "<f8"), ("n_contingency", "<i8")]), count=int(n_years))
        return raw["ens"].copy(), raw["dlc"].copy(), raw["nlc"].copy(), raw["n_contingency"].copy(), acc

    # seqMain.m:85-262
    def seqMain(self, max_sim_years: int = MAX_SIM_YEARS, cov_threshold: float = COV_THRESHOLD,
                curtail_threshold: float = CURTAIL_THRESHOLD, *, seed: int = 1, mpopt=None, batch_years: int = 0) -> "SeqResult":
        """The yearly loop, its CoV stop and the post-processing run below the C ABI (relmc_seq_run); with a communicator in the engine's
        context (dist.NativeComm / dist.HostComm) the same call on every rank IS the multi-rank run: the years of every batch shard over the
        ranks and every rank returns the same result.  batch_years: years per launch round over all ranks (0 = 64 per rank); the result
        does not depend on it."""
        o = _abi.SeqOpts()
        self.L.relmc_seq_opts_default(C.byref(o))
        o.cov_threshold, o.max_years, o.batch_years = float(cov_threshold), int(max_sim_years), int(batch_years)
        o.seed, o.curtail_threshold = int(seed), float(curtail_threshold)
        if mpopt is not None:
            o.solver = mpopt
        n = int(max_sim_years)
        yrs = (SeqYear * n)(); cum_eens = np.zeros(n); cum_cov = np.zeros(n)
        o.years_cap = n
        o.results_year = yrs; o.cum_eens = cum_eens.ctypes.data_as(_abi.c_double_p); o.cum_cov = cum_cov.ctypes.data_as(_abi.c_double_p)
        res = _abi.SeqResult()
        self.eng._check(self.L.relmc_seq_run(self.eng._h, C.byref(o), C.byref(res)), "relmc_seq_run")
        y = int(res.final_year)
        raw = np.frombuffer(yrs, dtype=np.dtype([("ens", "<f8"), ("dlc", "<f8"), ("nlc", "<f8"), ("n_contingency", "<i8")]), count=n)[:y]
        a = np.column_stack([raw["ens"], raw["dlc"], raw["nlc"]]).reshape(-1, 3)
        nb, nc = self.eng.case.nb, self.eng.case.ncomp
        return SeqResult(
            final_year=y, eens=res.eens, cov=res.cov, lole=res.lole, lolf=res.lolf,
            results_year=dict(plc=a[:, 1] / self.hours, nlc=a[:, 2].copy(), dlc=a[:, 1].copy(), dns=a[:, 0] / self.hours, ens=a[:, 0].copy()),
            results_cum=dict(eens=cum_eens[:y].copy(), cov=cum_cov[:y].copy()),
            nodal_eens_avg=np.array(res.nodal_eens_avg[:nb]), comp_importance=np.array(res.comp_importance[:nc]),
            total_loss_hours=int(res.acc.n_fail), years_evaluated=y, n_lp=int(res.acc.n), n_singular=int(res.acc.n_singular),
            n_infeasible=int(res.acc.n_infeasible), n_nonconverged=int(res.acc.n_nonconverged),
            elapsed_time=res.wall_seconds, kernel_seconds=res.kernel_seconds, converged=bool(res.converged), acc=res.acc)

    def seqMain_distributed(self, max_sim_years: int = MAX_SIM_YEARS, cov_threshold: float = COV_THRESHOLD, *, seed: int = 1,
                            mpopt=None, batch_years: int = 0, device=None) -> "SeqResult":
        """seqMain over the ranks of the engine's communicator (one process per GPU): the same relmc_seq_run call on every rank."""
        return self.seqMain(max_sim_years, cov_threshold, seed=seed, mpopt=mpopt, batch_years=batch_years)


@dataclass
class SeqResult:
    final_year: int
    eens: float
    cov: float
    lole: float
    lolf: float
    results_year: dict
    results_cum: dict
    nodal_eens_avg: np.ndarray
    comp_importance: np.ndarray
    total_loss_hours: int
    years_evaluated: int
    n_lp: int
    n_singular: int
    n_infeasible: int
    n_nonconverged: int
    elapsed_time: float
    kernel_seconds: float
    converged: bool = True
    acc: _abi.Acc = field(repr=False, default=None)

    # -- what seqMain prints (seqMain.m:187-197, 206-249) ---------------------------------------------------------
    def progress_lines(self, every: int = 10) -> list[str]:
        """The loop's progress print (seqMain.m:187-190: every 10th year from the second on) and the convergence line (:195)."""
        out = []
        for y in range(2, self.final_year + 1):
            if y % every == 0:
                out.append("Year %4d | EENS: %.4f MWh/yr | CoV: %.4f" % (y, self.results_cum["eens"][y - 1], self.results_cum["cov"][y - 1]))
        if self.converged:
            out.append("Convergence Reached at Year %d!" % self.final_year)
        return out

    def top_buses(self, k: int = 5):
        """[(bus number 1-based, EENS MWh/yr)] of the k worst buses, seqMain.m:221-228 (zero entries are not listed)."""
        order = np.argsort(-self.nodal_eens_avg, kind="stable")[:k]
        return [(int(i) + 1, float(self.nodal_eens_avg[i])) for i in order if self.nodal_eens_avg[i] > 0]

    def top_components(self, k: int = 5, numGenerators: int = 33):
        """[(type, id 1-based, P(down | loss hour))] of the k most critical components, seqMain.m:235-245."""
        order = np.argsort(-self.comp_importance, kind="stable")[:k]
        return [(("Gen", int(c) + 1) if c < numGenerators else ("Line", int(c) - numGenerators + 1)) + (float(self.comp_importance[c]),) for c in order]

    def report(self, progress_every: int = 10, numGenerators: int = 33) -> str:
        """Text of seqMain.m's console output from the yearly loop on (:187-197 progress, :206-249 results, nodal indices, weak points),
        same wording and number formats."""
        L = list(self.progress_lines(progress_every))
        L += ["", "--- SIMULATION COMPLETE ---",
              "EENS (Expected Energy Not Supplied): %.4f MWh/yr" % self.eens,
              "LOLE (Loss of Load Expectation):     %.4f hr/yr" % self.lole,
              "LOLF (Loss of Load Frequency):       %.4f occ/yr" % self.lolf, "", "--- NODAL RELIABILITY INDICES ---",
              "Top 5 Buses by EENS (MWh/yr):"]
        L += ["  Bus %2d: %.4f MWh/yr" % bv for bv in self.top_buses(5)]
        L += ["", "--- WEAK POINT DETECTION ---"]
        if self.total_loss_hours > 0:
            L.append("Top 5 Critical Components (Prob. Down given System Failure):")
            L += ["  %s %2d: %.2f%%" % (t, i, v * 100.0) for t, i, v in self.top_components(5, numGenerators)]
        else:
            L.append("No failure events recorded to analyze weak points.")
        return "\n".join(L)

    def write_nodal_csv(self, path: str) -> None:
        """seq_nodal_results.csv as seqMain.m:255-257."""
        with open(path, "w") as f:
            f.write("BusID,EENS_MWh_yr\n")
            for i, v in enumerate(self.nodal_eens_avg):
                f.write(f"{i + 1},{float(v):.15g}\n")

    def save_mat(self, path: str) -> None:
        """seq_reliability_results.mat with the variables of seqMain.m:261-262."""
        from scipy.io import savemat
        savemat(path, dict(results_year=self.results_year, results_cum=self.results_cum,
                           nodal_eens_avg=self.nodal_eens_avg[None, :], comp_importance=self.comp_importance[:, None]))
