"""TEST INFRASTRUCTURE ONLY — ctypes binding of oracle/librelmc_oracle.so (relmc_oracle.c).

Importable only from tests/, tests/golden/make_*.py, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from powersystemsreliabilityassessment_amd import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("RELMC_ORACLE_LIB_PATH") or os.path.join(_HERE, "librelmc_oracle.so")     # override: the sanitizer build (make -C oracle asan)


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "relmc_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "librelmc_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        u8p, dp, i32p = _abi.c_uint8_p, _abi.c_double_p, _abi.c_int32_p
        L.orc_mc_sampling.argtypes = [C.POINTER(_abi.CaseDesc), C.c_uint64, C.c_uint64, C.c_int64, u8p]
        L.orc_mc_sampling.restype = C.c_int32
        L.orc_mc_simulation.argtypes = [C.POINTER(_abi.CaseDesc), u8p, C.c_int64,
                                        C.POINTER(_abi.SolverOpts), dp, dp, i32p, i32p, i32p, C.c_int32]
        L.orc_mc_simulation.restype = C.c_int32
        L.orc_nsq_accumulate.argtypes = [C.POINTER(_abi.CaseDesc), C.c_uint64, C.c_uint64, C.c_int64,
                                         C.POINTER(_abi.SolverOpts), C.c_int32, C.c_int32,
                                         C.POINTER(_abi.Acc)]
        L.orc_nsq_accumulate.restype = C.c_int32
        i64p = _abi.c_int64_p
        L.orc_nsq_database.argtypes = [C.POINTER(_abi.CaseDesc), C.c_uint64, C.c_double, C.c_int64, C.c_int64, C.POINTER(_abi.SolverOpts),
                                       C.c_int32, C.c_int64, u8p, i64p, dp, i32p, dp, i32p, i32p, i32p, i64p,
                                       C.c_int64, dp, dp, dp, dp, i64p, i64p, C.POINTER(_abi.Acc)]
        L.orc_nsq_database.restype = C.c_int32
        L.orc_nsq_indices.argtypes = [C.POINTER(_abi.Acc), C.c_int32, C.c_int32, C.c_double,
                                      C.POINTER(_abi.Indices)]
        L.orc_nsq_indices.restype = None
        L.orc_thresholds.argtypes = [C.POINTER(_abi.CaseDesc), _abi.c_uint32_p]
        L.orc_philox4x32_10.argtypes = [_abi.c_uint32_p, _abi.c_uint32_p, _abi.c_uint32_p]
        L.orc_max_threads.restype = C.c_int32
        L.orc_seq_mcsimulation.argtypes = [C.POINTER(_abi.CaseDesc), u8p, dp, C.c_int64, C.POINTER(_abi.SolverOpts), dp, dp, i32p, i32p, i32p, C.c_int32]
        L.orc_seq_mcsimulation.restype = C.c_int32
        L.orc_seq_mcsampling.argtypes = [C.c_int32, dp, dp, C.c_int32, C.c_uint64, C.c_uint64, C.c_int32, u8p]
        L.orc_seq_mcsampling.restype = C.c_int32
        L.orc_seq_years.argtypes = [C.POINTER(_abi.CaseDesc), dp, dp, C.c_int32, dp, C.c_uint64, C.c_uint64, C.c_int32,
                                    C.POINTER(_abi.SolverOpts), C.c_double, C.c_int32, dp, C.POINTER(_abi.Acc)]
        L.orc_seq_years.restype = C.c_int32
        L.orc_hl1_nsq.argtypes = [C.c_int32, dp, dp, C.c_int32, dp, C.c_uint64, C.c_uint64, C.c_int64, dp, dp]
        L.orc_hl1_nsq.restype = C.c_int32
        _lib = L
    return _lib


class Oracle:
    def __init__(self, case):
        self.case = case
        self.h = _abi.CaseHolder(case)
        self.L = lib()

    def thresholds(self):
        out = np.zeros(self.case.ncomp, dtype=np.uint32)
        self.L.orc_thresholds(C.byref(self.h.desc), out.ctypes.data_as(_abi.c_uint32_p))
        return out

    def mc_sampling(self, seed, first_index, n):
        out = np.zeros((n, self.case.ncomp), dtype=np.uint8)
        rc = self.L.orc_mc_sampling(C.byref(self.h.desc), seed, first_index, n,
                                    out.ctypes.data_as(_abi.c_uint8_p))
        assert rc == 0
        return out

    def mc_simulation(self, states, policy=_abi.RELMC_REFERENCE_EMULATE, opts=None, nthreads=1):
        states = np.ascontiguousarray(states, dtype=np.uint8).reshape(-1, self.case.ncomp)
        n = states.shape[0]
        o = opts or _abi.default_solver_opts(policy)
        dns = np.zeros(n)
        nodal = np.zeros((n, self.case.nb))
        status = np.zeros(n, dtype=np.int32)
        iters = np.zeros(n, dtype=np.int32)
        relaxed = np.zeros(n, dtype=np.int32)
        rc = self.L.orc_mc_simulation(C.byref(self.h.desc), states.ctypes.data_as(_abi.c_uint8_p), n,
                                      C.byref(o), dns.ctypes.data_as(_abi.c_double_p),
                                      nodal.ctypes.data_as(_abi.c_double_p),
                                      status.ctypes.data_as(_abi.c_int32_p),
                                      iters.ctypes.data_as(_abi.c_int32_p),
                                      relaxed.ctypes.data_as(_abi.c_int32_p), nthreads)
        assert rc == 0
        return dict(dns=dns, nodal=nodal, status=status, iters=iters, relaxed=relaxed)

    def nsq_accumulate(self, seed, first_index, n, policy=_abi.RELMC_REFERENCE_EMULATE, opts=None,
                       nthreads=None, memo=True):
        o = opts or _abi.default_solver_opts(policy)
        acc = _abi.Acc()
        nt = nthreads or self.L.orc_max_threads()
        rc = self.L.orc_nsq_accumulate(C.byref(self.h.desc), seed, first_index, n, C.byref(o), nt,
                                       1 if memo else 0, C.byref(acc))
        assert rc == 0
        return acc

    def nsq_database(self, seed, beta_limit=0.0017, max_iterations=100000, samples_per_batch=100,
                     policy=_abi.RELMC_REFERENCE_EMULATE, opts=None, nthreads=None, max_rows=None):
        """nsqMain.m:208-308 in the reference's own database form.  Returns dict(states, count, dns, flag, nodal, status,
        iters, relaxed, beta_history, edns_history, lole_history, plc_history, iterations, acc)."""
        o = opts or _abi.default_solver_opts(policy)
        cap = int(max_rows or max_iterations)
        nc, nb = self.case.ncomp, self.case.nb
        ncp = (int(max_iterations) + int(samples_per_batch) - 1) // int(samples_per_batch)
        d = dict(states=np.zeros((cap, nc), dtype=np.uint8), count=np.zeros(cap, dtype=np.int64), dns=np.zeros(cap),
                 flag=np.zeros(cap, dtype=np.int32), nodal=np.zeros((cap, nb)), status=np.zeros(cap, dtype=np.int32),
                 iters=np.zeros(cap, dtype=np.int32), relaxed=np.zeros(cap, dtype=np.int32))
        h = [np.zeros(ncp) for _ in range(4)]
        rows, cp, its = C.c_int64(), C.c_int64(), C.c_int64()
        acc = _abi.Acc()
        p = lambda a, t: a.ctypes.data_as(t)
        rc = self.L.orc_nsq_database(C.byref(self.h.desc), seed, beta_limit, max_iterations, samples_per_batch, C.byref(o),
                                     nthreads or self.L.orc_max_threads(), cap, p(d["states"], _abi.c_uint8_p), p(d["count"], _abi.c_int64_p),
                                     p(d["dns"], _abi.c_double_p), p(d["flag"], _abi.c_int32_p), p(d["nodal"], _abi.c_double_p),
                                     p(d["status"], _abi.c_int32_p), p(d["iters"], _abi.c_int32_p), p(d["relaxed"], _abi.c_int32_p),
                                     C.byref(rows), ncp, p(h[0], _abi.c_double_p), p(h[1], _abi.c_double_p), p(h[2], _abi.c_double_p),
                                     p(h[3], _abi.c_double_p), C.byref(cp), C.byref(its), C.byref(acc))
        assert rc == 0, rc
        n, k = int(rows.value), int(cp.value)
        out = {key: v[:n] for key, v in d.items()}
        out.update(beta_history=h[0][:k], edns_history=h[1][:k], lole_history=h[2][:k], plc_history=h[3][:k],
                   iterations=int(its.value), acc=acc)
        return out

    def indices(self, acc, hours=8760.0):
        out = _abi.Indices()
        self.L.orc_nsq_indices(C.byref(acc), self.case.nb, self.case.ncomp, hours, C.byref(out))
        return out

    def max_threads(self):
        return int(self.L.orc_max_threads())

    # ---- sequential track (Montecarlo_seq/) ------------------------------------------------------
    def seq_mcsampling(self, rel, hours, seed, first_year, num_years):
        """states[num_years*hours, ncomp] (1 = down), seq_mcsampling.m:35-76 with independent years."""
        mttf = np.ascontiguousarray(rel[:, 0], dtype=np.float64); mttr = np.ascontiguousarray(rel[:, 1], dtype=np.float64)
        out = np.zeros((num_years * hours, self.case.ncomp), dtype=np.uint8)
        rc = self.L.orc_seq_mcsampling(self.case.ncomp, mttf.ctypes.data_as(_abi.c_double_p), mttr.ctypes.data_as(_abi.c_double_p),
                                       hours, seed, first_year, num_years, out.ctypes.data_as(_abi.c_uint8_p))
        assert rc == 0
        return out

    def seq_mcsimulation(self, states, load_scale, policy=_abi.RELMC_REFERENCE_EMULATE, nthreads=1):
        states = np.ascontiguousarray(states, dtype=np.uint8).reshape(-1, self.case.ncomp)
        n = states.shape[0]
        sc = np.ascontiguousarray(np.broadcast_to(np.asarray(load_scale, dtype=np.float64), (n,)))
        o = _abi.default_solver_opts(policy)
        dns = np.zeros(n); nodal = np.zeros((n, self.case.nb))
        status = np.zeros(n, dtype=np.int32); iters = np.zeros(n, dtype=np.int32); relaxed = np.zeros(n, dtype=np.int32)
        p = lambda a, t: a.ctypes.data_as(t)
        rc = self.L.orc_seq_mcsimulation(C.byref(self.h.desc), p(states, _abi.c_uint8_p), p(sc, _abi.c_double_p), n, C.byref(o),
                                         p(dns, _abi.c_double_p), p(nodal, _abi.c_double_p), p(status, _abi.c_int32_p),
                                         p(iters, _abi.c_int32_p), p(relaxed, _abi.c_int32_p), nthreads)
        assert rc == 0
        return dict(dns=dns, nodal=nodal, status=status, iters=iters, relaxed=relaxed)

    def seq_years(self, rel, hours, load_factors, seed, first_year, n_years, policy=_abi.RELMC_REFERENCE_EMULATE,
                  threshold=0.01, nthreads=None, opts=None):
        mttf = np.ascontiguousarray(rel[:, 0], dtype=np.float64); mttr = np.ascontiguousarray(rel[:, 1], dtype=np.float64)
        lf = np.ascontiguousarray(load_factors, dtype=np.float64)
        o = opts if opts is not None else _abi.default_solver_opts(policy)
        yrs = np.zeros((n_years, 4)); acc = _abi.Acc()
        rc = self.L.orc_seq_years(C.byref(self.h.desc), mttf.ctypes.data_as(_abi.c_double_p), mttr.ctypes.data_as(_abi.c_double_p), hours,
                                  lf.ctypes.data_as(_abi.c_double_p), seed, first_year, n_years, C.byref(o), threshold,
                                  nthreads or self.L.orc_max_threads(), yrs.ctypes.data_as(_abi.c_double_p), C.byref(acc))
        assert rc == 0
        return yrs, acc


def hl1_nsq(capacity, for_rate, hourly_load, seed, first_index, n):
    """Oracle of PowerSystemAdequacy.jl:169-208: per-iteration (loss hours, unserved energy)."""
    L = lib()
    cap = np.ascontiguousarray(capacity, dtype=np.float64)
    forr = np.ascontiguousarray(for_rate, dtype=np.float64)
    hl = np.ascontiguousarray(hourly_load, dtype=np.float64)
    lole, eue = np.zeros(n), np.zeros(n)
    p = lambda a: a.ctypes.data_as(_abi.c_double_p)
    rc = L.orc_hl1_nsq(cap.size, p(cap), p(forr), hl.size, p(hl), seed, first_index, n, p(lole), p(eue))
    assert rc == 0
    return lole, eue
