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


class SeqYear(C.Structure):
    _fields_ = [("ens", C.c_double), ("dlc", C.c_double), ("nlc", C.c_double), ("n_contingency", C.c_int64)]


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
        a = np.array([(y.ens, y.dlc, y.nlc, y.n_contingency) for y in yrs], dtype=np.float64).reshape(-1, 4)
        return a[:, 0], a[:, 1], a[:, 2], a[:, 3].astype(np.int64), acc

    def seqMain_distributed(self, max_sim_years: int = MAX_SIM_YEARS, cov_threshold: float = COV_THRESHOLD, *, seed: int = 1,
                            mpopt=None, batch_years: int = 512, device=None) -> dict:
        """seqMain over the ranks of the default torch.distributed group (one process per GPU): years shard
        contiguously per super-batch, see dist.seq_run_distributed."""
        from . import dist as rdist

        def fn(sd, first, n):
            e, d, n_, _, acc = self.seq_years(sd, first, n, mpopt)
            return e, d, n_, acc
        return rdist.seq_run_distributed(fn, seed=seed, cov_threshold=cov_threshold, max_sim_years=max_sim_years,
                                         batch_years=batch_years, device=device)

    # seqMain.m:85-262
    def seqMain(self, max_sim_years: int = MAX_SIM_YEARS, cov_threshold: float = COV_THRESHOLD,
                curtail_threshold: float = CURTAIL_THRESHOLD, *, seed: int = 1, mpopt=None, batch_years: int = 64) -> "SeqResult":
        t0 = time.time()
        ens, dlc, nlc = [], [], []
        total = _abi.Acc()
        eens_hist, cov_hist = [], []
        from . import dist as rdist
        done, final_year = 0, 0
        kernel_ms = 0.0
        stop = False
        while done < max_sim_years and not stop:
            m = min(batch_years, max_sim_years - done)
            e, d, n_, _, acc = self.seq_years(seed, done, m, mpopt, curtail_threshold)
            kernel_ms += self.eng.last_kernel_ms()
            used = m
            for k in range(m):
                ens.append(e[k]); dlc.append(d[k]); nlc.append(n_[k])
                y = len(ens)
                mean = float(np.mean(ens))                                          # seqMain.m:180
                eens_hist.append(mean)
                cov = float(np.std(ens, ddof=1) / (mean * np.sqrt(y))) if y > 1 and mean > 0 else 0.0   # :183-185
                cov_hist.append(cov)
                if y > 1 and 0 < cov < cov_threshold:                               # :194
                    stop, used = True, k + 1
                    break
            if used < m:
                # the reference stops inside this batch: its post-processing accumulators (seqMain.m:146-159)
                # cover the years up to the stopping year only, so evaluate exactly those again
                _, _, _, _, acc = self.seq_years(seed, done, used, mpopt, curtail_threshold)
                kernel_ms += self.eng.last_kernel_ms()
            total = rdist.merge(total, acc)
            done += used
        final_year = len(ens)
        ens_a, dlc_a, nlc_a = np.array(ens), np.array(dlc), np.array(nlc)
        nb, nc = self.eng.case.nb, self.eng.case.ncomp
        loss_hours = int(total.n_fail)
        years_eval = done
        return SeqResult(
            final_year=final_year, eens=eens_hist[-1], cov=cov_hist[-1], lole=float(dlc_a.mean()), lolf=float(nlc_a.mean()),
            results_year=dict(plc=dlc_a / self.hours, nlc=nlc_a, dlc=dlc_a, dns=ens_a / self.hours, ens=ens_a),
            results_cum=dict(eens=np.array(eens_hist), cov=np.array(cov_hist)),
            nodal_eens_avg=np.array(total.sum_nodal[:nb]) / years_eval,                                  # :218
            comp_importance=(np.array(total.comp_fail[:nc], dtype=np.float64) / loss_hours if loss_hours else np.zeros(nc)),   # :233
            total_loss_hours=loss_hours, years_evaluated=years_eval, n_lp=int(total.n), n_singular=int(total.n_singular),
            n_infeasible=int(total.n_infeasible), n_nonconverged=int(total.n_nonconverged),
            elapsed_time=time.time() - t0, kernel_seconds=kernel_ms * 1e-3)


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
