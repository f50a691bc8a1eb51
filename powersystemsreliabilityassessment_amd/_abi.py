"""ctypes mirror of the plain-C structs in include/relmc.h (data layout only, no code path).

Both the product loader (``_lib.py``) and the test-only oracle binding
(``oracle/coracle.py``) describe their arguments with these classes.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

RELMC_MAX_BUS = 128
RELMC_MAX_COMP = 256

RELMC_REFERENCE_EMULATE = 0
RELMC_PHYSICAL = 1

ST_CONVERGED, ST_MAXIT, ST_NUMFAIL, ST_SINGULAR = 0, 1, 2, 3

c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)
c_uint8_p = C.POINTER(C.c_uint8)
c_uint32_p = C.POINTER(C.c_uint32)
c_int64_p = C.POINTER(C.c_int64)


class CaseDesc(C.Structure):
    _fields_ = [
        ("base_mva", C.c_double),
        ("nb", C.c_int32), ("ng", C.c_int32), ("nl", C.c_int32), ("nd", C.c_int32),
        ("ref_bus", C.c_int32),
        ("bus_pd", c_double_p),
        ("inj_bus", c_int32_p),
        ("inj_pmin", c_double_p), ("inj_pmax", c_double_p), ("inj_cost", c_double_p),
        ("br_from", c_int32_p), ("br_to", c_int32_p),
        ("br_b", c_double_p), ("br_rate", c_double_p),
        ("unavail", c_double_p),
        ("always_up", c_uint8_p),
        ("total_load", C.c_double),
    ]


class SolverOpts(C.Structure):
    _fields_ = [
        ("singular_policy", C.c_int32), ("max_it", C.c_int32),
        ("feastol", C.c_double), ("gradtol", C.c_double), ("comptol", C.c_double),
        ("costtol", C.c_double), ("xi", C.c_double), ("sigma", C.c_double), ("z0", C.c_double),
        ("alpha_min", C.c_double), ("max_stepsize", C.c_double),
        ("screen", C.c_int32), ("reserved", C.c_int32),
    ]


class Acc(C.Structure):
    _fields_ = [
        ("n", C.c_int64), ("n_fail", C.c_int64), ("n_singular", C.c_int64),
        ("n_infeasible", C.c_int64), ("n_nonconverged", C.c_int64), ("sum_iters", C.c_int64),
        ("comp_fail", C.c_int64 * RELMC_MAX_COMP),
        ("n_screened", C.c_int64),
        ("sum_dns", C.c_double), ("sum_dns2", C.c_double),
        ("sum_nodal", C.c_double * RELMC_MAX_BUS),
    ]

    N_INT = 6 + RELMC_MAX_COMP + 1
    N_DBL = 2 + RELMC_MAX_BUS

    def to_arrays(self):
        """(int64[N_INT], float64[N_DBL]) views copied out — the all-reduce payload."""
        raw = np.frombuffer(bytes(self), dtype=np.uint8)
        ints = raw[: self.N_INT * 8].view(np.int64).copy()
        dbls = raw[self.N_INT * 8:].view(np.float64).copy()
        return ints, dbls

    @classmethod
    def from_arrays(cls, ints, dbls):
        ints = np.ascontiguousarray(ints, dtype=np.int64)
        dbls = np.ascontiguousarray(dbls, dtype=np.float64)
        assert ints.size == cls.N_INT and dbls.size == cls.N_DBL
        return cls.from_buffer_copy(ints.tobytes() + dbls.tobytes())


# relmc_allreduce_fn: int32 fn(void* user, relmc_acc* acc_inout)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.POINTER(Acc))
# relmc_allreduce_f64_fn: int32 fn(void* user, double* buf_inout, int64 count)
ALLREDUCE_F64_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, c_double_p, C.c_int64)


class Indices(C.Structure):
    _fields_ = [
        ("n", C.c_int64),
        ("edns", C.c_double), ("lole", C.c_double), ("plc", C.c_double), ("beta", C.c_double),
        ("eens", C.c_double), ("mean_iters", C.c_double),
        ("nodal_eens", C.c_double * RELMC_MAX_BUS),
        ("comp_importance", C.c_double * RELMC_MAX_COMP),
    ]


class NsqOpts(C.Structure):
    _fields_ = [
        ("beta_limit", C.c_double), ("max_samples", C.c_int64), ("batch", C.c_int64),
        ("seed", C.c_uint64), ("hours_per_year", C.c_double),
        ("solver", SolverOpts),
        ("history_cap", C.c_int64),
        ("beta_history", c_double_p), ("edns_history", c_double_p),
        ("lole_history", c_double_p), ("plc_history", c_double_p),
        ("distinct_states", C.c_int32),
    ]


class NsqResult(C.Structure):
    _fields_ = [
        ("acc", Acc), ("idx", Indices),
        ("checkpoints", C.c_int64), ("converged", C.c_int32),
        ("wall_seconds", C.c_double), ("kernel_seconds", C.c_double),
        ("batches", C.c_int64),
    ]


class SeqYear(C.Structure):
    _fields_ = [("ens", C.c_double), ("dlc", C.c_double), ("nlc", C.c_double), ("n_contingency", C.c_int64)]


class SeqOpts(C.Structure):
    _fields_ = [
        ("cov_threshold", C.c_double), ("max_years", C.c_int32), ("batch_years", C.c_int32), ("seed", C.c_uint64),
        ("curtail_threshold", C.c_double), ("solver", SolverOpts),
        ("years_cap", C.c_int64), ("results_year", C.POINTER(SeqYear)), ("cum_eens", c_double_p), ("cum_cov", c_double_p),
    ]


class SeqResult(C.Structure):
    _fields_ = [
        ("final_year", C.c_int32), ("converged", C.c_int32),
        ("eens", C.c_double), ("cov", C.c_double), ("lole", C.c_double), ("lolf", C.c_double), ("plc", C.c_double),
        ("n_contingency", C.c_int64), ("acc", Acc),
        ("nodal_eens_avg", C.c_double * RELMC_MAX_BUS), ("comp_importance", C.c_double * RELMC_MAX_COMP),
        ("wall_seconds", C.c_double), ("kernel_seconds", C.c_double),
    ]


class DbStats(C.Structure):
    _fields_ = [("rows", C.c_int64), ("samples", C.c_int64), ("new_rows", C.c_int64), ("batch_distinct", C.c_int64)]


assert C.sizeof(Acc) == (Acc.N_INT + Acc.N_DBL) * 8


def default_solver_opts(policy: int = RELMC_REFERENCE_EMULATE) -> SolverOpts:
    """MIPS defaults MATPOWER uses for OPF_ALG_DC=200 (nsqMain.m:185-186)."""
    return SolverOpts(singular_policy=policy, max_it=150, feastol=5e-6, gradtol=1e-6,
                      comptol=1e-6, costtol=1e-6, xi=0.99995, sigma=0.1, z0=1.0,
                      alpha_min=1e-8, max_stepsize=1e10, screen=0, reserved=0)


class CaseHolder:
    """Keeps the numpy arrays a CaseDesc points into alive."""

    def __init__(self, case):
        f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
        self.case = case
        self._keep = dict(
            bus_pd=f64(case.bus_pd), inj_bus=i32(case.inj_bus), inj_pmin=f64(case.inj_pmin),
            inj_pmax=f64(case.inj_pmax), inj_cost=f64(case.inj_cost), br_from=i32(case.br_from),
            br_to=i32(case.br_to), br_b=f64(case.br_b), br_rate=f64(case.br_rate),
            unavail=f64(case.unavail),
            always_up=np.ascontiguousarray(case.always_up, dtype=np.uint8))
        k = self._keep
        self.desc = CaseDesc(
            base_mva=case.base_mva, nb=case.nb, ng=case.ng, nl=case.nl, nd=case.nd,
            ref_bus=case.ref_bus,
            bus_pd=k["bus_pd"].ctypes.data_as(c_double_p),
            inj_bus=k["inj_bus"].ctypes.data_as(c_int32_p),
            inj_pmin=k["inj_pmin"].ctypes.data_as(c_double_p),
            inj_pmax=k["inj_pmax"].ctypes.data_as(c_double_p),
            inj_cost=k["inj_cost"].ctypes.data_as(c_double_p),
            br_from=k["br_from"].ctypes.data_as(c_int32_p),
            br_to=k["br_to"].ctypes.data_as(c_int32_p),
            br_b=k["br_b"].ctypes.data_as(c_double_p),
            br_rate=k["br_rate"].ctypes.data_as(c_double_p),
            unavail=k["unavail"].ctypes.data_as(c_double_p),
            always_up=k["always_up"].ctypes.data_as(c_uint8_p),
            total_load=case.total_load)
