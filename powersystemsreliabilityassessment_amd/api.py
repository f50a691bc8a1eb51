"""Host-side mirror of the reference's operator interface for the HL2 non-sequential path.

The reference exposes three MATLAB entry points; the methods below keep their
names, argument meaning and outputs and route every evaluation to the HIP library through
the C ABI of include/relmc.h:

  mc_sampling(failure_probabilities, num_samples, numGenerators, numLines)
      -> eqstatus [num_samples x (Ng+Nl)], 1 = failed        (mc_sampling.m:2)
  mc_simulation(component_states, TestSystem, mpopt, numGenerators, numLines)
      -> (dns, nodal_dns[1 x Nb])                             (mc_simulation.m:1)
  nsqMain(...)   the `while beta > beta_limit` loop + post-processing (nsqMain.m:208-406)

Differences that are deliberate and documented in DESIGN.md: the sampler is a counter-based
RNG keyed by (seed, global scenario index) instead of MATLAB's unseeded global stream, and
mc_simulation accepts a whole matrix of states (the reference's parfor, nsqMain.m:257-263).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import _abi, _lib, case24

REFERENCE_EMULATE = _abi.RELMC_REFERENCE_EMULATE
PHYSICAL = _abi.RELMC_PHYSICAL


class RelmcError(RuntimeError):
    pass


def mpoption(singular_policy: int = REFERENCE_EMULATE, **overrides) -> _abi.SolverOpts:
    """Solver options = what nsqMain.m:185-186 asks MATPOWER for (DC model, MIPS, flow limits on),
    i.e. MIPS defaults: feastol 5e-6, gradtol/comptol/costtol 1e-6, max_it 150.  screen=1 turns the zero-curtailment pre-screen on
    (include/relmc.h: states with a proven LP optimum of 0 are counted, not solved; every output but the iteration statistics unchanged)."""
    o = _abi.default_solver_opts(singular_policy)
    for k, v in overrides.items():
        if not hasattr(o, k):
            raise TypeError(f"unknown solver option {k!r}")
        setattr(o, k, v)
    return o


@dataclass
class NsqResult:
    """Workspace variables nsqMain leaves behind (nsqMain.m:282-308, 348-349, 366-376, 404-405)."""
    accumulated_edns: float
    accumulated_lole: float
    plc: float
    current_beta: float
    current_iteration: int
    nodal_eens: np.ndarray
    comp_importance: np.ndarray
    beta_history: np.ndarray
    edns_history: np.ndarray
    lole_history: np.ndarray
    plc_history: np.ndarray
    converged: bool
    mean_iters: float
    n_singular: int
    n_infeasible: int
    n_nonconverged: int
    elapsed_time: float
    kernel_seconds: float
    acc: _abi.Acc = field(repr=False, default=None)
    samples_per_batch: int = 100
    beta_limit: float = 0.0017
    database_row_count: int | None = None      # unique states evaluated (only the database path keeps them)
    hours_per_year: float = 8760.0
    n_screened: int = 0             # samples the zero-curtailment pre-screen counted without solving (mpoption(screen=1); 0 otherwise)

    # -- what nsqMain prints (nsqMain.m:314-317, 325-393) ---------------------------------------------------------
    def progress_lines(self, every: int = 1000) -> list[str]:
        """The loop's progress print (nsqMain.m:314-317): one line whenever the sample count is a multiple of `every`."""
        out = []
        for k in range(len(self.beta_history)):
            it = (k + 1) * self.samples_per_batch
            if it > self.current_iteration:
                it = self.current_iteration
            if it % every == 0:
                out.append("Iteration %6d: Beta = %.6f, EDNS = %.4f MW, LOLE = %.4f hr/yr"
                           % (it, self.beta_history[k], self.edns_history[k], self.lole_history[k]))
        return out

    def top_buses(self, k: int = 5):
        """[(bus number 1-based, EENS MWh/yr)] of the k worst buses, nsqMain.m:351-358 (zero entries are not listed)."""
        order = np.argsort(-self.nodal_eens, kind="stable")[:k]
        return [(int(i) + 1, float(self.nodal_eens[i]) * self.hours_per_year) for i in order if self.nodal_eens[i] > 0]

    def top_components(self, k: int = 5, numGenerators: int | None = None):
        """[(type, id 1-based, P(down | system failure))] of the k most critical components, nsqMain.m:378-389."""
        ng = numGenerators if numGenerators is not None else self._ng
        order = np.argsort(-self.comp_importance, kind="stable")[:k]
        return [(("Gen", int(c) + 1) if c < ng else ("Line", int(c) - ng + 1)) + (float(self.comp_importance[c]),) for c in order]

    def report(self, progress_every: int = 1000) -> str:
        """Text of nsqMain.m's console output from the Monte Carlo loop on (:314-317 progress, :325-393 results,
        nodal indices and weak-point detection), same wording and number formats."""
        L = list(self.progress_lines(progress_every))
        L += ["", "========================================", "SECTION 8: SIMULATION RESULTS", "========================================",
              "Total simulation time: %.2f seconds" % self.elapsed_time, "Total iterations: %d" % self.current_iteration]
        if self.database_row_count is not None:
            L.append("Unique states evaluated: %d" % self.database_row_count)
        L += ["Convergence achieved: %s" % ("YES" if self.current_beta <= self.beta_limit else "NO"), "", "--- RELIABILITY INDICES ---",
              "EDNS (Expected Demand Not Supplied): %.4f MW" % self.accumulated_edns,
              "LOLE (Loss of Load Expectation): %.4f hours/year" % self.accumulated_lole,
              "PLC (Probability of Load Curtailment): %.6f" % self.plc,
              "Beta (Coefficient of Variation): %.6f" % self.current_beta, "", "--- NODAL RELIABILITY INDICES ---",
              "Top 5 Buses by EENS (MWh/yr):"]
        L += ["  Bus %2d: %.4f MWh/yr" % bv for bv in self.top_buses(5)]
        L += ["", "--- WEAK POINT DETECTION ---"]
        if self.acc is not None and self.acc.n_fail > 0:
            L.append("Top 5 Critical Components (Prob. Down given System Failure):")
            L += ["  %s %2d: %.2f%%" % (t, i, v * 100.0) for t, i, v in self.top_components(5)]
        else:
            L.append("No failure events recorded to analyze weak points.")
        return "\n".join(L)

    _ng: int = field(repr=False, default=33)

    def write_nodal_csv(self, path: str, hours_per_year: float = 8760.0) -> None:
        """nodal_results.csv exactly as nsqMain.m:398-400 writes it (BusID, EENS_MWh_yr)."""
        with open(path, "w") as f:
            f.write("BusID,EENS_MWh_yr\n")
            for i, v in enumerate(self.nodal_eens):
                f.write(f"{i + 1},{float(v) * hours_per_year:.15g}\n")

    def save_mat(self, path: str) -> None:
        """reliability_results.mat with the variables of nsqMain.m:404-405."""
        from scipy.io import savemat
        savemat(path, dict(accumulated_edns=self.accumulated_edns, accumulated_lole=self.accumulated_lole,
                           nodal_eens=self.nodal_eens[None, :], comp_importance=self.comp_importance[:, None],
                           beta_history=self.beta_history[None, :], edns_history=self.edns_history[None, :]))


def tune_order(case: case24.Case, evaluations: int = 20000, seed: int = 1, start=None):
    """relmc_tune_order (host only, no GPU): searches the primary elimination order of `case` against the library's own scheduler.
    Returns (order [nb] int32, dict(lds_before, passes_before, lds_after, passes_after)): LDS instructions per Newton step and
    dependent passes of the start order (the rule's when start is None) and of the result."""
    L = _lib.load()
    holder = _abi.CaseHolder(case)
    out = np.zeros(case.nb, dtype=np.int32); st = np.zeros(4, dtype=np.int32)
    s0 = None if start is None else np.ascontiguousarray(start, dtype=np.int32)
    if s0 is not None and s0.size != case.nb:
        raise ValueError("start must list every bus once")
    rc = L.relmc_tune_order(C.addressof(holder.desc), int(evaluations), int(seed), None if s0 is None else s0.ctypes.data_as(_abi.c_int32_p),
                            out.ctypes.data_as(_abi.c_int32_p), st.ctypes.data_as(_abi.c_int32_p))
    if rc != 0:
        raise RelmcError(f"relmc_tune_order failed ({rc})")
    return out, dict(lds_before=int(st[0]), passes_before=int(st[1]), lds_after=int(st[2]), passes_after=int(st[3]))


class Engine:
    """One context = one GPU (one process per GPU).  Owns the device-resident case tables."""

    def __init__(self, case: case24.Case | None = None, device: int = 0, elim_order="case", debug_switches=()):
        self.L = _lib.load()
        h = C.c_void_p()
        rc = self.L.relmc_ctx_create(int(device), C.byref(h))
        if rc != 0:
            raise RelmcError(f"relmc_ctx_create(device={device}) failed with {rc}: no usable HIP device "
                             "(this package has no CPU fallback)")
        self._h = h
        self.device = device
        self.case = None
        self._holder = None
        for key in debug_switches:              # diagnosis switches that must be in place before the case is loaded (tests: "no_retry")
            self.debug_set(key)
        self.load_case(case or case24.rts24(), elim_order)

    # -- lifetime ---------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self.L.relmc_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            msg = self.L.relmc_last_error(self._h)
            raise RelmcError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")

    # -- setup ------------------------------------------------------------------------------
    def load_case(self, case: case24.Case, elim_order="case"):
        """relmc_case_load.  elim_order: the primary elimination order of the solver schedule (external bus numbers, reference bus last:
        relmc_case_order_hint) -- "case" takes `case.elim_order` when the case carries one (the RTS-24 / RTS-96 orders of this package were
        tuned offline with `tune_order`), None the library's rule, "tune" = run `tune_order` (20000 evaluations, seed 1) right now."""
        holder = _abi.CaseHolder(case)
        if isinstance(elim_order, str) and elim_order == "tune":          # tune now (host only; ~1-3 ms per evaluation), for a network without a stored order
            elim_order, _ = tune_order(case, 20000, 1, getattr(case, "elim_order", None))
        order = getattr(case, "elim_order", None) if isinstance(elim_order, str) and elim_order == "case" else elim_order
        if order is not None:
            o = np.ascontiguousarray(order, dtype=np.int32)
            self._check(self.L.relmc_case_order_hint(self._h, o.ctypes.data_as(_abi.c_int32_p), int(o.size)), "relmc_case_order_hint")
        else:
            self._check(self.L.relmc_case_order_hint(self._h, None, 0), "relmc_case_order_hint")
        self._check(self.L.relmc_case_load(self._h, C.byref(holder.desc)), "relmc_case_load")
        self.case, self._holder = case, holder

    def thresholds(self) -> np.ndarray:
        out = np.zeros(self.case.ncomp, dtype=np.uint32)
        self._check(self.L.relmc_case_thresholds(self._h, out.ctypes.data_as(_abi.c_uint32_p)),
                    "relmc_case_thresholds")
        return out

    # -- mc_sampling.m:2 ----------------------------------------------------------------------
    def mc_sampling(self, failure_probabilities=None, num_samples: int = 100, numGenerators=None,
                    numLines=None, *, seed: int = 1, first_index: int = 0) -> np.ndarray:
        ng = self.case.ng if numGenerators is None else int(numGenerators)
        nl = self.case.nl if numLines is None else int(numLines)
        if (ng, nl) != (self.case.ng, self.case.nl):
            raise ValueError("numGenerators/numLines do not match the loaded case")
        if failure_probabilities is not None:
            fp = np.asarray(failure_probabilities, dtype=np.float64).ravel()
            if fp.size != ng + nl:
                raise ValueError("failure_probabilities must have numGenerators+numLines entries")
            if not np.array_equal(fp, self.case.unavail):
                # the thresholds live in the device case: swapping them here would silently change every later
                # nsq_accumulate / nsqMain call and drop a loaded sequential model
                raise ValueError("failure_probabilities differ from the loaded case: call "
                                 "Engine.load_case(dataclasses.replace(case, unavail=...)) explicitly")
        n = int(num_samples)
        out = np.zeros((n, ng + nl), dtype=np.uint8)
        self._check(self.L.relmc_mc_sampling(self._h, int(seed), int(first_index), n,
                                             out.ctypes.data_as(_abi.c_uint8_p)), "relmc_mc_sampling")
        return out

    # -- mc_simulation.m:1 (batched) ------------------------------------------------------------
    def mc_simulation(self, component_states, TestSystem=None, mpopt=None, numGenerators=None,
                      numLines=None, *, return_info: bool = False):
        """Returns (dns, nodal_dns).  For a single state: scalar dns and a [Nb] vector, as the
        reference; for a matrix of states: dns[n] and nodal_dns[n, Nb]."""
        if TestSystem is not None and TestSystem is not self.case:
            self.load_case(TestSystem)
        st = np.asarray(component_states)
        single = st.ndim == 1
        st = np.ascontiguousarray(st.reshape(-1, self.case.ncomp) != 0, dtype=np.uint8)
        n = st.shape[0]
        o = mpopt if mpopt is not None else mpoption()
        dns = np.zeros(n)
        nodal = np.zeros((n, self.case.nb))
        status = np.zeros(n, dtype=np.int32)
        iters = np.zeros(n, dtype=np.int32)
        self._check(self.L.relmc_mc_simulation(self._h, st.ctypes.data_as(_abi.c_uint8_p), n, C.byref(o),
                                               dns.ctypes.data_as(_abi.c_double_p),
                                               nodal.ctypes.data_as(_abi.c_double_p),
                                               status.ctypes.data_as(_abi.c_int32_p),
                                               iters.ctypes.data_as(_abi.c_int32_p)), "relmc_mc_simulation")
        if single:
            res = (float(dns[0]), nodal[0])
        else:
            res = (dns, nodal)
        if return_info:
            return res + (dict(status=status, iters=iters),)
        return res

    def mc_simulation_dense(self, component_states, mpopt=None):
        """Test hook: mc_simulation with every Newton step solved by the dense, partially pivoted last resort of the retry path
        (relmc_debug_mc_simulation_dense).  Returns (dns[n], nodal[n, Nb], dict(status, iters))."""
        st = np.ascontiguousarray(np.asarray(component_states).reshape(-1, self.case.ncomp) != 0, dtype=np.uint8)
        n = st.shape[0]
        o = mpopt if mpopt is not None else mpoption()
        dns, nodal = np.zeros(n), np.zeros((n, self.case.nb))
        status, iters = np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int32)
        self._check(self.L.relmc_debug_mc_simulation_dense(self._h, st.ctypes.data_as(_abi.c_uint8_p), n, C.byref(o),
                                                           dns.ctypes.data_as(_abi.c_double_p), nodal.ctypes.data_as(_abi.c_double_p),
                                                           status.ctypes.data_as(_abi.c_int32_p), iters.ctypes.data_as(_abi.c_int32_p)),
                    "relmc_debug_mc_simulation_dense")
        return dns, nodal, dict(status=status, iters=iters)

    def screen_states(self, component_states, load_scale=None) -> np.ndarray:
        """Test hook: the zero-curtailment certificate of the pre-screen (mpoption(screen=1), relmc_screen.hip) for given states, optionally at
        per-state load scale factors (the sequential track's hours).  Returns bool[n]: True = the pre-screen would count this state without solving it."""
        st = np.ascontiguousarray(np.asarray(component_states).reshape(-1, self.case.ncomp) != 0, dtype=np.uint8)
        n = st.shape[0]
        out = np.zeros(n, dtype=np.uint8)
        sc = None if load_scale is None else np.ascontiguousarray(np.broadcast_to(np.asarray(load_scale, dtype=np.float64), (n,)))
        self._check(self.L.relmc_debug_screen_states(self._h, st.ctypes.data_as(_abi.c_uint8_p), None if sc is None else sc.ctypes.data_as(_abi.c_double_p), n,
                                                     out.ctypes.data_as(_abi.c_uint8_p)), "relmc_debug_screen_states")
        return out.astype(bool)

    def mc_simulation_dev(self, states_ptr: int, n: int, dns_ptr: int, nodal_ptr: int = 0,
                          status_ptr: int = 0, iters_ptr: int = 0, mpopt=None):
        """Same with every buffer already in this GPU's HBM (raw device addresses, e.g. tensor.data_ptr())."""
        o = mpopt if mpopt is not None else mpoption()
        self._check(self.L.relmc_mc_simulation_dev(self._h, states_ptr, int(n), C.byref(o), dns_ptr,
                                                   nodal_ptr or None, status_ptr or None, iters_ptr or None),
                    "relmc_mc_simulation_dev")

    # -- fused loop body ---------------------------------------------------------------------
    def nsq_accumulate(self, seed: int, first_index: int, n: int, mpopt=None) -> _abi.Acc:
        o = mpopt if mpopt is not None else mpoption()
        acc = _abi.Acc()
        self._check(self.L.relmc_nsq_accumulate(self._h, int(seed), int(first_index), int(n), C.byref(o),
                                                C.byref(acc)), "relmc_nsq_accumulate")
        return acc

    def nsq_accumulate_distinct(self, seed: int, first_index: int, n: int, mpopt=None):
        """The same accumulators through the reference's dedupe (nsqMain.m:220-245): every distinct state of the range is
        solved once and counted with its multiplicity.  Returns (Acc, number of distinct states)."""
        o = mpopt if mpopt is not None else mpoption()
        acc = _abi.Acc()
        nd = C.c_int64()
        self._check(self.L.relmc_nsq_accumulate_distinct(self._h, int(seed), int(first_index), int(n), C.byref(o), C.byref(acc),
                                                         C.byref(nd)), "relmc_nsq_accumulate_distinct")
        return acc, int(nd.value)

    # -- the reference's persistent unique-state database (nsqMain.m:91-99, 220-278) -----------------
    def db_reset(self):
        self._check(self.L.relmc_db_reset(self._h), "relmc_db_reset")

    def nsq_db_batch(self, seed: int, first_index: int, n: int, mpopt=None):
        """One pass of the reference's loop body over samples [first_index, first_index+n): dedupe, count bumps for known
        states, evaluation of the new ones, indices from the whole database.  Returns (Acc of the whole database, DbStats)."""
        o = mpopt if mpopt is not None else mpoption()
        acc, st = _abi.Acc(), _abi.DbStats()
        self._check(self.L.relmc_nsq_db_batch(self._h, int(seed), int(first_index), int(n), C.byref(o), C.byref(acc),
                                              C.byref(st)), "relmc_nsq_db_batch")
        return acc, st

    def db_accumulate(self) -> _abi.Acc:
        """The accumulators of the whole database again (nsqMain.m:282-301 over every row; no sampling)."""
        acc = _abi.Acc()
        self._check(self.L.relmc_db_accumulate(self._h, C.byref(acc)), "relmc_db_accumulate")
        return acc

    def db_size(self):
        rows, samples = C.c_int64(), C.c_int64()
        self._check(self.L.relmc_db_size(self._h, C.byref(rows), C.byref(samples)), "relmc_db_size")
        return int(rows.value), int(samples.value)

    def retry_stats(self):
        """(units evaluated a second time under the alternate elimination order, how many of them then converged) since the
        case was loaded (relmc_retry_stats)."""
        u, c = C.c_int64(), C.c_int64()
        self._check(self.L.relmc_retry_stats(self._h, C.byref(u), C.byref(c)), "relmc_retry_stats")
        return int(u.value), int(c.value)

    def retry_overflow(self) -> int:
        """Units that did not fit the kernel's list of non-converged units and kept their first-attempt results (relmc_retry_overflow)."""
        u = C.c_int64()
        self._check(self.L.relmc_retry_overflow(self._h, C.byref(u)), "relmc_retry_overflow")
        return int(u.value)

    def retry_dense_stats(self):
        """(units that went to the dense, partially pivoted last resort since the case was loaded, how many of them it converged on)."""
        u, c = C.c_int64(), C.c_int64()
        self._check(self.L.relmc_retry_dense_stats(self._h, C.byref(u), C.byref(c)), "relmc_retry_dense_stats")
        return int(u.value), int(c.value)

    def case_order(self):
        """(primary static elimination order 0/1/2, failures of each probed order among the 8192 calibration states; -1 = not
        probed) — relmc_case_order."""
        p = C.c_int32(); f = (C.c_int32 * 3)()
        self._check(self.L.relmc_case_order(self._h, C.byref(p), f), "relmc_case_order")
        return int(p.value), [int(x) for x in f]

    def db_export(self, first_row: int = 0, n_rows: int | None = None) -> dict:
        """Rows of the database in the reference's column layout (nsqMain.m:91-99): states, count, dns, flag, nodal
        (+ solver status and iteration count)."""
        rows, _ = self.db_size()
        n = rows - first_row if n_rows is None else int(n_rows)
        nc, nb = self.case.ncomp, self.case.nb
        out = dict(states=np.zeros((n, nc), dtype=np.uint8), count=np.zeros(n, dtype=np.int64), dns=np.zeros(n),
                   flag=np.zeros(n, dtype=np.int32), nodal=np.zeros((n, nb)), status=np.zeros(n, dtype=np.int32),
                   iters=np.zeros(n, dtype=np.int32), relaxed=np.zeros(n, dtype=np.uint8))
        self._check(self.L.relmc_db_export(self._h, int(first_row), n, out["states"].ctypes.data_as(_abi.c_uint8_p),
                                           out["count"].ctypes.data_as(_abi.c_int64_p), out["dns"].ctypes.data_as(_abi.c_double_p),
                                           out["flag"].ctypes.data_as(_abi.c_int32_p), out["nodal"].ctypes.data_as(_abi.c_double_p),
                                           out["status"].ctypes.data_as(_abi.c_int32_p), out["iters"].ctypes.data_as(_abi.c_int32_p),
                                           out["relaxed"].ctypes.data_as(_abi.c_uint8_p)),
                    "relmc_db_export")
        return out

    def db_import(self, rows: dict, mpopt=None):
        """Resume: the rows of an earlier `db_export()` (same case) back into the EMPTY database, in the same order; `mpopt` = the solver
        options they were computed under.  The next `nsq_db_batch` / `nsqMain(distinct_states=2)` continues the run."""
        o = mpopt if mpopt is not None else mpoption()
        f = lambda a, dt: np.ascontiguousarray(a, dtype=dt)
        st, cnt, dns, nod = f(rows["states"], np.uint8), f(rows["count"], np.int64), f(rows["dns"], np.float64), f(rows["nodal"], np.float64)
        n = st.shape[0]
        if st.shape != (n, self.case.ncomp) or nod.shape != (n, self.case.nb) or cnt.shape != (n,) or dns.shape != (n,):
            raise ValueError("db_import: rows do not have the loaded case's shapes")
        opt = lambda k, dt, pt: (f(rows[k], dt), pt) if k in rows and rows[k] is not None else (None, None)
        stat, _ = opt("status", np.int32, None); its, _ = opt("iters", np.int32, None); rel, _ = opt("relaxed", np.uint8, None)
        self._check(self.L.relmc_db_import(self._h, C.byref(o), n, st.ctypes.data_as(_abi.c_uint8_p), cnt.ctypes.data_as(_abi.c_int64_p),
                                           dns.ctypes.data_as(_abi.c_double_p), nod.ctypes.data_as(_abi.c_double_p),
                                           stat.ctypes.data_as(_abi.c_int32_p) if stat is not None else None,
                                           its.ctypes.data_as(_abi.c_int32_p) if its is not None else None,
                                           rel.ctypes.data_as(_abi.c_uint8_p) if rel is not None else None), "relmc_db_import")

    def comm_set_timeout(self, seconds: float):
        """relmc_comm_set_timeout: wall-clock guard of communicator init and of every collective through this context (<= 0: off)."""
        self._check(self.L.relmc_comm_set_timeout(self._h, float(seconds)), "relmc_comm_set_timeout")

    def pci_bus_id(self) -> str:
        """PCI bus id of the GPU this context drives (relmc_device_pci_bus_id)."""
        buf = C.create_string_buffer(64)
        self._check(self.L.relmc_device_pci_bus_id(self._h, buf, 64), "relmc_device_pci_bus_id")
        return buf.value.decode()

    def debug_set(self, key: str, value: bool = True):
        """Diagnosis switch of the context (tests): no_retry, retry_dense_first, nsq_no_stretch, db_no_probe (relmc_debug_set)."""
        self._check(self.L.relmc_debug_set(self._h, key.encode(), int(bool(value))), "relmc_debug_set")

    def last_kernel_ms(self) -> float:
        ms = C.c_double()
        self._check(self.L.relmc_last_kernel_ms(self._h, C.byref(ms)), "relmc_last_kernel_ms")
        return ms.value

    def indices(self, acc: _abi.Acc, hours_per_year: float = 8760.0) -> _abi.Indices:
        out = _abi.Indices()
        self.L.relmc_nsq_indices(C.byref(acc), self.case.nb, self.case.ncomp, hours_per_year, C.byref(out))
        return out

    # -- nsqMain.m:208-406 ---------------------------------------------------------------------
    def nsqMain(self, beta_limit: float = 0.0017, max_iterations: int = 100000,
                samples_per_batch: int = 100, *, seed: int = 1, mpopt=None,
                hours_per_year: float = 8760.0, distinct_states: bool | int | str = False, verbose: bool = False) -> NsqResult:
        """Defaults are the reference's (nsqMain.m:60-62).  Small batches are evaluated many checkpoints per launch by the
        library, so the reference's batch of 100 costs about the same as one large batch.
        distinct_states: False = every sample solved; True / 1 = distinct states of each batch solved once;
        "database" / 2 = the reference's persistent unique-state database across batches (nsqMain.m:220-278).
        verbose: print what the reference prints (progress every 1000 samples, results, top-5 buses / components)."""
        o = _abi.NsqOpts()
        self.L.relmc_nsq_opts_default(C.byref(o))
        o.beta_limit, o.max_samples, o.batch = float(beta_limit), int(max_iterations), int(samples_per_batch)
        o.seed, o.hours_per_year = int(seed), float(hours_per_year)
        mode = 2 if distinct_states in ("database", 2) else (1 if distinct_states else 0)
        o.distinct_states = mode
        if mpopt is not None:
            o.solver = mpopt
        ncp = (int(max_iterations) + int(samples_per_batch) - 1) // int(samples_per_batch)
        hist = [np.zeros(ncp) for _ in range(4)]
        o.history_cap = ncp
        o.beta_history, o.edns_history, o.lole_history, o.plc_history = (
            h.ctypes.data_as(_abi.c_double_p) for h in hist)
        res = _abi.NsqResult()
        self._check(self.L.relmc_nsq_run(self._h, C.byref(o), C.byref(res)), "relmc_nsq_run")
        k = int(res.checkpoints)
        nb, nc = self.case.nb, self.case.ncomp
        out = NsqResult(
            accumulated_edns=res.idx.edns, accumulated_lole=res.idx.lole, plc=res.idx.plc,
            current_beta=res.idx.beta, current_iteration=int(res.idx.n),
            nodal_eens=np.array(res.idx.nodal_eens[:nb]), comp_importance=np.array(res.idx.comp_importance[:nc]),
            beta_history=hist[0][:k], edns_history=hist[1][:k], lole_history=hist[2][:k], plc_history=hist[3][:k],
            converged=bool(res.converged), mean_iters=res.idx.mean_iters, n_singular=int(res.acc.n_singular),
            n_infeasible=int(res.acc.n_infeasible), n_nonconverged=int(res.acc.n_nonconverged), n_screened=int(res.acc.n_screened),
            elapsed_time=res.wall_seconds, kernel_seconds=res.kernel_seconds, acc=res.acc,
            samples_per_batch=int(samples_per_batch), beta_limit=float(beta_limit),
            database_row_count=self.db_size()[0] if mode == 2 else None, hours_per_year=float(hours_per_year), _ng=self.case.ng)
        if verbose:
            print(out.report())
        return out


# ---- module-level functions with the reference's names (default engine on RTS-24) ---------------
_default_engine: Engine | None = None


def default_engine() -> Engine:
    global _default_engine
    if _default_engine is None:
        _default_engine = Engine()
    return _default_engine


def mc_sampling(failure_probabilities, num_samples, numGenerators, numLines, **kw):
    return default_engine().mc_sampling(failure_probabilities, num_samples, numGenerators, numLines, **kw)


def mc_simulation(component_states, TestSystem=None, mpopt=None, numGenerators=None, numLines=None, **kw):
    return default_engine().mc_simulation(component_states, TestSystem, mpopt, numGenerators, numLines, **kw)


def nsqMain(**kw):
    return default_engine().nsqMain(**kw)
