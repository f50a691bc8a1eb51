"""CPU interpreter of the static solver schedule (test infrastructure, no GPU).

`relmc_debug_symbolic` (csrc/relmc_schedule.hip) runs the host-side symbolic analysis of `relmc_case_load` without a device and hands
back the pass program the evaluation kernel interprets.  `solve()` executes that program with numpy on one bus-pair system
[[M, B'], [B, -E]] in exactly the kernel's data layout (2x2 blocks in a flat workspace W, the right-hand side as a pseudo bus)
and returns the solution, which the tests compare with numpy.linalg.solve on the dense matrix.  Every pass form is covered:
full / half / quarter block updates, pivot inversion, full / half back substitution.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from powersystemsreliabilityassessment_amd import _abi, _lib

LF_EXISTS, LF_OWNER = 1, 2


@dataclass
class Schedule:
    tile: int
    rw: int
    nb: int
    noff: int
    nws: int
    off_rhs: int
    npass: int
    npass_upd: int
    npass_inv: int
    npass_updh: int
    npass_updq: int
    nzero: int
    scen_doubles: int
    lds_bytes: int
    conflict_before: int
    conflict_after: int
    nl: int
    flags: int
    bwd_half: int              # bit k: back-substitution pass k is in half form
    tasks: np.ndarray          # [npass, rw, 4] uint16
    pass_ntask: np.ndarray
    b_int: np.ndarray          # external -> internal bus
    l_blk: np.ndarray          # [nl] W offset of the owner line's block (0xffff: not an owner)
    l_info: np.ndarray
    zero_off: np.ndarray
    maxdeg: tuple = (0, 0)     # longest line list among the buses of bus slot 0 / 1 (the gathers' unrolled steps stop there)
    maxinj: tuple = (0, 0)     # longest injection list per bus slot

    @property
    def npass_bwd(self):
        return self.npass - self.npass_upd - self.npass_inv


def symbolic(case, order_variant: int = 0, order=None, model_leaf_free: int = -1) -> Schedule:
    L = _lib.load()
    f = L.relmc_debug_symbolic
    f.restype = C.c_int32
    f.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                  C.c_void_p, C.c_char_p, C.c_int32, C.c_int32]
    hint = None if order is None else np.ascontiguousarray(order, dtype=np.int32)
    holder = _abi.CaseHolder(case)
    hdr = np.zeros(24, np.int32)
    tasks = np.zeros(96 * 64 * 4, np.uint16)
    pnt = np.zeros(96, np.uint8)
    b_int = np.zeros(256, np.uint8)
    l_blk = np.zeros(256, np.uint16)
    l_info = np.zeros(256, np.uint32)
    zero_off = np.zeros(512, np.uint16)
    err = C.create_string_buffer(512)
    rc = f(C.byref(holder.desc), order_variant, None if hint is None else hint.ctypes.data, 0 if hint is None else int(hint.size), hdr.ctypes.data, tasks.ctypes.data, tasks.size, pnt.ctypes.data, b_int.ctypes.data,
           l_blk.ctypes.data, l_info.ctypes.data, zero_off.ctypes.data, err, 512, int(model_leaf_free))
    if rc != 0:
        raise RuntimeError(f"relmc_debug_symbolic failed ({rc}): {err.value.decode()}")
    h = [int(x) for x in hdr]
    npass, rw = h[6], h[1]
    return Schedule(tile=h[0], rw=rw, nb=h[2], noff=h[3], nws=h[4], off_rhs=h[5], npass=npass, npass_upd=h[7], npass_inv=h[8],
                    npass_updh=h[9], npass_updq=h[10], nzero=h[11], scen_doubles=h[12], lds_bytes=h[13], conflict_before=h[14],
                    conflict_after=h[15], nl=h[17], flags=h[18], bwd_half=(h[19] & 0xffffffff) | ((h[20] & 0xffffffff) << 32),
                    tasks=tasks[: npass * rw * 4].reshape(npass, rw, 4).copy(), pass_ntask=pnt[:npass].copy(), b_int=b_int[: h[2]].copy(),
                    l_blk=l_blk[: h[17]].copy(), l_info=l_info[: h[17]].copy(), zero_off=zero_off[: h[11]].copy(),
                    maxdeg=(h[21] & 0xff, (h[21] >> 8) & 0xff), maxinj=((h[21] >> 16) & 0xff, (h[21] >> 24) & 0xff))


def random_system(s: Schedule, rng, line_on=None):
    """Block system with the sparsity the kernel assembles: per bus D_i = [[m, b], [b, -e]], per in-service bus pair
    K = [[-g, -c], [-c, 0]] in the owner line's block.  Returns (W, dense matrix K [2nb x 2nb], rhs [2nb]) with unknowns ordered
    (theta_0, lambda_0, theta_1, ...) by INTERNAL bus number."""
    nb = s.nb
    W = np.zeros(s.nws + 8)
    A = np.zeros((2 * nb, 2 * nb))
    g_pair = {}
    for l in range(s.nl):
        inf = int(s.l_info[l])
        if not ((inf >> 24) & LF_EXISTS):
            continue
        if line_on is not None and not line_on[l]:
            continue
        f, t = inf & 0xff, (inf >> 8) & 0xff
        key = (max(f, t), min(f, t))
        g, c = rng.uniform(0.1, 2.0), rng.uniform(1.0, 30.0)
        gg, cc = g_pair.get(key, (0.0, 0.0))
        g_pair[key] = (gg + g, cc + c)
    md = np.zeros(nb); bd = np.zeros(nb)
    for (hi, lo), (g, c) in g_pair.items():
        md[hi] += g; md[lo] += g; bd[hi] += c; bd[lo] += c
        A[2 * hi, 2 * lo] = A[2 * lo, 2 * hi] = -g
        A[2 * hi, 2 * lo + 1] = A[2 * lo + 1, 2 * hi] = -c        # K[th_hi][lam_lo]
        A[2 * hi + 1, 2 * lo] = A[2 * lo, 2 * hi + 1] = -c        # K[lam_hi][th_lo]
    e = rng.uniform(0.0, 3.0, nb) * (rng.uniform(size=nb) < 0.7)     # structural zeros on buses without injections
    md += rng.uniform(0.01, 0.2, nb)                                 # keeps the test system regular whatever the outage pattern
    for i in range(nb):
        A[2 * i, 2 * i] = md[i]; A[2 * i, 2 * i + 1] = A[2 * i + 1, 2 * i] = bd[i]; A[2 * i + 1, 2 * i + 1] = -e[i]
        W[4 * i: 4 * i + 4] = [md[i], bd[i], bd[i], -e[i]]
    # owner-line blocks (hi, lo): [[-g, -c], [-c, 0]]
    for l in range(s.nl):
        off = int(s.l_blk[l])
        if off == 0xffff:
            continue
        inf = int(s.l_info[l]); f, t = inf & 0xff, (inf >> 8) & 0xff
        key = (max(f, t), min(f, t))
        g, c = g_pair.get(key, (0.0, 0.0))
        W[off: off + 4] = [-g, -c, -c, 0.0]
    for z in s.zero_off:
        W[int(z): int(z) + 4] = 0.0
    rhs = rng.normal(size=2 * nb)
    for i in range(nb):
        W[s.off_rhs + 2 * i: s.off_rhs + 2 * i + 2] = rhs[2 * i: 2 * i + 2]
    return W, A, rhs


def _inv(D):
    pm, pb, pe = D[0], D[1], -D[3]
    q = 1.0 / (pm * pe + pb * pb)
    return pe * q, pb * q, -pm * q


def solve(s: Schedule, W: np.ndarray) -> np.ndarray:
    """Runs the pass program on W (in place) the way the kernel's lanes do: every lane of a pass loads its operands before any lane
    stores (write-after-read inside a pass is legal, read-after-write is not)."""
    W = W.copy()
    npu, npi = s.npass_upd, s.npass_inv
    npuh = npu - s.npass_updq
    npuf = npuh - s.npass_updh
    for p in range(s.npass):
        stores = []
        for r in range(s.rw):
            d = [int(x) for x in s.tasks[p, r]]
            if d[0] == 0xffff:
                continue
            if p < npuf:                                       # full block update (or the rhs row)
                vec = bool(d[0] & 0x8000)
                T, Wa, Wb, D = d[0] & 0x7fff, d[1], d[2], d[3]
                P00, P01, P11 = _inv(W[D: D + 4])
                b0, b1 = W[Wb: Wb + 2], W[Wb + 2: Wb + 4]
                for row in range(1 if vec else 2):
                    a = W[Wa + 2 * row: Wa + 2 * row + 2]
                    g0, g1 = a[0] * P00 + a[1] * P01, a[0] * P01 + a[1] * P11
                    t = W[T + 2 * row: T + 2 * row + 2].copy()
                    t[0] -= g0 * b0[0] + g1 * b0[1]; t[1] -= g0 * b1[0] + g1 * b1[1]
                    stores.append((T + 2 * row, t))
            elif p < npuh:                                     # half form: one row of T
                T, Wa, Wb, D = d
                P00, P01, P11 = _inv(W[D: D + 4])
                a = W[Wa: Wa + 2]; b0, b1 = W[Wb: Wb + 2], W[Wb + 2: Wb + 4]
                g0, g1 = a[0] * P00 + a[1] * P01, a[0] * P01 + a[1] * P11
                t = W[T: T + 2].copy()
                t[0] -= g0 * b0[0] + g1 * b0[1]; t[1] -= g0 * b1[0] + g1 * b1[1]
                stores.append((T, t))
            elif p < npu:                                      # quarter form: one element of T
                T, Wa, Wb, D = d
                P00, P01, P11 = _inv(W[D: D + 4])
                a = W[Wa: Wa + 2]; b0 = W[Wb: Wb + 2]
                g0, g1 = a[0] * P00 + a[1] * P01, a[0] * P01 + a[1] * P11
                stores.append((T, np.array([W[T] - (g0 * b0[0] + g1 * b0[1])])))
            elif p < npu + npi:                                # D <- inv(D), y <- P y
                D, Y = d[0], d[1]
                P00, P01, P11 = _inv(W[D: D + 4])
                y = W[Y: Y + 2]
                stores.append((D, np.array([P00, P01, P01, P11])))
                stores.append((Y, np.array([P00 * y[0] + P01 * y[1], P01 * y[0] + P11 * y[1]])))
            elif (s.bwd_half >> (p - npu - npi)) & 1:          # half form: one component of y_i per lane
                Yi, Wk, Pr, Ya = d
                w0, w1 = W[Wk: Wk + 2], W[Wk + 2: Wk + 4]
                pr = W[Pr: Pr + 2]; x = W[Ya: Ya + 2]
                u0, u1 = w0[0] * x[0] + w1[0] * x[1], w0[1] * x[0] + w1[1] * x[1]
                stores.append((Yi, np.array([W[Yi] - (pr[0] * u0 + pr[1] * u1)])))
            else:                                              # y_i -= P_i W' x_a
                Yi, Wk, P, Ya = d
                w0, w1 = W[Wk: Wk + 2], W[Wk + 2: Wk + 4]
                p0, p1 = W[P: P + 2], W[P + 2: P + 4]
                x = W[Ya: Ya + 2]
                u0, u1 = w0[0] * x[0] + w1[0] * x[1], w0[1] * x[0] + w1[1] * x[1]
                y = W[Yi: Yi + 2].copy()
                y[0] -= p0[0] * u0 + p0[1] * u1; y[1] -= p1[0] * u0 + p1[1] * u1
                stores.append((Yi, y))
        for off, v in stores:
            W[off: off + len(v)] = v
    return W[s.off_rhs: s.off_rhs + 2 * s.nb].copy()
