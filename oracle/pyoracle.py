"""TEST INFRASTRUCTURE ONLY — Python oracles for the HL2 state-evaluation path.

Not part of the product: only tests/, tests/golden/make_*.py, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this module.

PARITY UNPINNED (per state): the arithmetic of the path lives in MATPOWER
(runopf -> dcopf_solver -> qps_mips -> mips), an un-vendored, un-pinned dependency
of the reference (Montecarlo_nsq_single/mc_simulation.m:41; README.md:46-48), and the
reference holds no tests or per-state golden vectors.  What pins these oracles:
  * ``lp_highs``  : the LP of SURVEY.md Appendix C solved by scipy/HiGHS pins the
                    per-state total curtailment (unique LP optimum value);
  * ``mips_full`` : the published MIPS algorithm (Wang/Zimmerman, MATPOWER mips.m,
                    restated from SURVEY.md Appendix B) on the *unreduced* MATPOWER
                    formulation pins the nodal split and iteration counts;
  * the reference's golden artifacts (reliability_results.mat / nodal_results.csv,
    converted to tests/golden/nsq_golden.json) pin the converged indices statistically.

Conventions: a *state* is a uint8 vector [ng+nl], 1 = failed (mc_sampling.m:18-19).
"""
from __future__ import annotations

import numpy as np

# singular-state policies
REFERENCE_EMULATE = 0
PHYSICAL = 1

# status codes (shared with include/relmc.h)
ST_CONVERGED = 0
ST_MAXIT = 1          # max_it reached (MIPS eflag 0)
ST_NUMFAIL = 2        # MIPS "numerically failed" inside the loop (eflag -1)
ST_SINGULAR = 3       # isolated bus, REFERENCE_EMULATE: start point returned (fact 11)

MIPS_DEFAULTS = dict(xi=0.99995, sigma=0.1, z0=1.0, alpha_min=1e-8, max_it=150,
                     feastol=5e-6, gradtol=1e-6, comptol=1e-6, costtol=1e-6,
                     max_stepsize=1e10)


# ----------------------------------------------------------------------------
# topology helpers
# ----------------------------------------------------------------------------
def islands(case, br_on):
    """Connected components of the in-service network.  Returns label[nb] where
    labels are the lowest bus index of each component."""
    nb = case.nb
    parent = list(range(nb))

    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a

    for l in range(case.nl):
        if br_on[l]:
            a, b = find(int(case.br_from[l])), find(int(case.br_to[l]))
            if a != b:
                if a < b:
                    parent[b] = a
                else:
                    parent[a] = b
    return np.array([find(i) for i in range(nb)], dtype=np.int64)


def preprocess(case, state, policy):
    """Decide pins / relaxations for one state.

    Returns dict(singular, pins, drop_bal, pmin, n_relaxed) where
      singular : REFERENCE_EMULATE and some bus has no in-service branch
      pins     : bus indices whose angle is fixed to 0 (ref + one per extra island)
      drop_bal : buses whose (redundant) balance row is dropped (islands with no injection)
      pmin     : per-injection lower bound after over-generation relaxation (MW)
    """
    ng, nl, nb = case.ng, case.nl, case.nb
    state = np.asarray(state).astype(bool)
    gen_on = ~state[:ng]
    br_on = ~state[ng:ng + nl]
    inj_on = np.concatenate([gen_on, np.ones(case.nd, dtype=bool)])
    deg = np.zeros(nb, dtype=int)
    for l in range(nl):
        if br_on[l]:
            deg[case.br_from[l]] += 1
            deg[case.br_to[l]] += 1
    singular = (policy == REFERENCE_EMULATE) and bool(np.any(deg == 0))
    lab = islands(case, br_on)
    pins, drop_bal = [case.ref_bus], []
    pmin = case.inj_pmin.copy()
    n_relaxed = 0
    ref_lab = lab[case.ref_bus]
    pmax = case.inj_pmax.copy()
    # an emulated-singular state is never solved: MATPOWER's own start point is returned, so
    # the island rules below must not touch its bounds
    for root in ([] if singular else np.unique(lab)):
        members = np.flatnonzero(lab == root)
        inj_here = [j for j in range(case.ninj) if inj_on[j] and lab[case.inj_bus[j]] == root]
        pin = case.ref_bus if root == ref_lab else int(members.max())   # rule 1
        if root != ref_lab:
            pins.append(pin)
        has_load = any(j >= ng for j in inj_here)
        has_gen = any(j < ng and case.inj_pmax[j] > 0 for j in inj_here)
        if inj_here and not has_load:
            # rule 2: island without load: its generators cannot deliver anything -> decommit
            for j in inj_here:
                inj_on[j] = False
            inj_here = []
            n_relaxed += 1
        elif has_load and not has_gen:
            # rule 3: island without generation: all of its load is shed (p fixed at 0; the LP
            # has no interior there)
            for j in inj_here:
                if j >= ng:
                    pmin[j] = 0.0
        elif sum(case.inj_pmin[j] for j in inj_here) > 1e-9:
            # rule 4: over-generation (sum of lower bounds > 0): relax Pmin of the island's units
            for j in inj_here:
                if j < ng:
                    pmin[j] = 0.0
            n_relaxed += 1
        # rule 5: no free (boxed) injection left -> the island's balance rows are dependent
        if not any(pmax[j] - pmin[j] > 0 for j in inj_here):
            drop_bal.append(pin)
    return dict(singular=singular, pins=sorted(pins), drop_bal=sorted(drop_bal), pmin=pmin,
                n_relaxed=n_relaxed, gen_on=gen_on, br_on=br_on, inj_on=inj_on)


# ----------------------------------------------------------------------------
# LP assembly in MATPOWER's formulation
# ----------------------------------------------------------------------------
def build_lp(case, state, policy=REFERENCE_EMULATE):
    """x = [theta (nb); p_inj (in-service injections, MATPOWER gen order)] in p.u.

    Returns dict with c, A (rows: nb balance + constrained in-service flows), l, u,
    xmin, xmax, x0, inj_idx (map LP injection column -> case injection index), pre.
    """
    pre = preprocess(case, state, policy)
    nb, base = case.nb, case.base_mva
    inj_idx = np.flatnonzero(pre["inj_on"])
    npj = inj_idx.size
    nx = nb + npj
    br_idx = np.flatnonzero(pre["br_on"])
    # makeBdc: Bf (nl_on x nb), Bbus = Cft' * Bf
    Bf = np.zeros((br_idx.size, nb))
    for r, l in enumerate(br_idx):
        Bf[r, case.br_from[l]] = case.br_b[l]
        Bf[r, case.br_to[l]] = -case.br_b[l]
    Cft = np.zeros((br_idx.size, nb))
    for r, l in enumerate(br_idx):
        Cft[r, case.br_from[l]] = 1.0
        Cft[r, case.br_to[l]] = -1.0
    Bbus = Cft.T @ Bf
    # Pmis: Bbus*Va - Cg*Pg = -(Pd+Gs)/base  (Pd zeroed: nsqMain.m:153)
    Amis = np.zeros((nb, nx))
    Amis[:, :nb] = Bbus
    for cidx, j in enumerate(inj_idx):
        Amis[case.inj_bus[j], nb + cidx] = -1.0
    bal_rows = [i for i in range(nb) if i not in pre["drop_bal"]]
    # Pf: -rate <= Bf*Va <= rate for branches with rateA != 0 (OPF_FLOW_LIM, nsqMain.m:186)
    il = [r for r, l in enumerate(br_idx) if case.br_rate[l] != 0]
    Apf = np.zeros((len(il), nx))
    Apf[:, :nb] = Bf[il, :]
    upf = np.array([case.br_rate[br_idx[r]] / base for r in il])
    A = np.vstack([Amis[bal_rows, :], Apf])
    l = np.concatenate([np.zeros(len(bal_rows)), -upf])
    u = np.concatenate([np.zeros(len(bal_rows)), upf])
    xmin = np.full(nx, -np.inf)
    xmax = np.full(nx, np.inf)
    for b in pre["pins"]:
        xmin[b] = xmax[b] = 0.0
    xmin[nb:] = pre["pmin"][inj_idx] / base
    xmax[nb:] = case.inj_pmax[inj_idx] / base
    c = np.zeros(nx)
    c[nb:] = case.inj_cost[inj_idx] * base
    # dcopf_solver interior start: midpoint of bounds, all angles = ref angle (0)
    lb = np.where(np.isinf(xmin), -1e10, xmin)
    ub = np.where(np.isinf(xmax), 1e10, xmax)
    x0 = (lb + ub) / 2.0
    x0[:nb] = 0.0
    return dict(c=c, A=A, l=l, u=u, xmin=xmin, xmax=xmax, x0=x0, inj_idx=inj_idx,
                pre=pre, nb=nb)


def _finish(case, lp, x, f):
    """mc_simulation.m:54-59 (dns) and :62-99 (nodal)."""
    base = case.base_mva
    dns = f + case.total_load
    if dns < 0.1:
        dns = 0.0
    nodal = np.zeros(case.nb)
    if dns > 0:
        for cidx, j in enumerate(lp["inj_idx"]):
            if j >= case.ng:
                shed = x[lp["nb"] + cidx] * base - case.inj_pmin[j]
                if shed > 1e-3:
                    nodal[case.inj_bus[j]] = shed
    return dns, nodal


def lp_highs(case, state, policy=PHYSICAL):
    """Vertex LP solution by scipy/HiGHS: pins the per-state total dns only."""
    from scipy.optimize import linprog
    lp = build_lp(case, state, policy)
    if lp["pre"]["singular"]:
        x0 = lp["x0"]
        dns, nodal = _finish(case, lp, x0, float(lp["c"] @ x0))
        return dict(dns=dns, nodal=nodal, status=ST_SINGULAR, feasible=True)
    A, l, u = lp["A"], lp["l"], lp["u"]
    eq = np.abs(u - l) <= 1e-12
    A_ub = np.vstack([A[~eq], -A[~eq]])
    b_ub = np.concatenate([u[~eq], -l[~eq]])
    bounds = [(None if np.isinf(a) else a, None if np.isinf(b) else b)
              for a, b in zip(lp["xmin"], lp["xmax"])]
    res = linprog(lp["c"], A_ub=A_ub, b_ub=b_ub, A_eq=A[eq], b_eq=u[eq], bounds=bounds,
                  method="highs")
    if res.status != 0:
        return dict(dns=float("nan"), nodal=np.full(case.nb, np.nan), status=-1, feasible=False)
    dns, nodal = _finish(case, lp, res.x, float(res.fun))
    return dict(dns=dns, nodal=nodal, status=ST_CONVERGED, feasible=True, raw_dns=res.fun + case.total_load)


# ----------------------------------------------------------------------------
# MIPS on the unreduced formulation
# ----------------------------------------------------------------------------
def mips_lp(c, A, l, u, xmin, xmax, x0, opt=None):
    """Primal-dual interior point for  min c'x  s.t. l<=Ax<=u, xmin<=x<=xmax.

    Follows MATPOWER's mips.m step by step for a linear cost and linear constraints
    (step_control off).  Returns (x, f, eflag, iterations) with eflag 1 converged,
    0 max_it, -1 numerically failed.
    """
    o = dict(MIPS_DEFAULTS)
    if opt:
        o.update(opt)
    nx = x0.size
    eps = np.finfo(float).eps
    # add var limits to linear constraints
    AA = np.vstack([np.eye(nx), A])
    ll = np.concatenate([xmin, l])
    uu = np.concatenate([xmax, u])
    ieq = np.flatnonzero(np.abs(uu - ll) <= eps)
    igt = np.flatnonzero((uu >= 1e10) & (ll > -1e10))
    ilt = np.flatnonzero((ll <= -1e10) & (uu < 1e10))
    ibx = np.flatnonzero((np.abs(uu - ll) > eps) & (uu < 1e10) & (ll > -1e10))
    Ae, be = AA[ieq], uu[ieq]
    Ai = np.vstack([AA[ilt], -AA[igt], AA[ibx], -AA[ibx]])
    bi = np.concatenate([uu[ilt], -ll[igt], uu[ibx], -ll[ibx]])
    neq, niq = Ae.shape[0], Ai.shape[0]

    x = x0.copy()
    f = float(c @ x)
    h = Ai @ x - bi
    g = Ae @ x - be
    gamma = 1.0
    lam = np.zeros(neq)
    z = o["z0"] * np.ones(niq)
    mu = z.copy()
    k = h < -o["z0"]
    z[k] = -h[k]
    k = gamma / z > o["z0"]
    mu[k] = gamma / z[k]
    f0 = f
    Lx = c + Ae.T @ lam + Ai.T @ mu

    def conds(x, z, lam, mu, g, h, Lx, f, f0):
        maxh = h.max() if h.size else -np.inf
        feas = max(np.abs(g).max() if g.size else 0.0, maxh) / (1 + max(np.abs(x).max(), np.abs(z).max()))
        grad = np.abs(Lx).max() / (1 + max(np.abs(lam).max() if lam.size else 0.0, np.abs(mu).max()))
        comp = (z @ mu) / (1 + np.abs(x).max())
        cost = abs(f - f0) / (1 + abs(f0))
        return feas, grad, comp, cost

    feas, grad, comp, cost = conds(x, z, lam, mu, g, h, Lx, f, f0)
    converged = (feas < o["feastol"] and grad < o["gradtol"] and comp < o["comptol"]
                 and cost < o["costtol"])
    eflag, i = 0, 0
    while not converged and i < o["max_it"]:
        i += 1
        zinv = 1.0 / z
        M = Ai.T @ ((mu * zinv)[:, None] * Ai)
        N = Lx + Ai.T @ ((mu * h + gamma) * zinv)
        K = np.block([[M, Ae.T], [Ae, np.zeros((neq, neq))]])
        rhs = np.concatenate([-N, -g])
        try:
            with np.errstate(all="ignore"):
                dxdlam = np.linalg.solve(K, rhs)
        except np.linalg.LinAlgError:
            dxdlam = np.full(nx + neq, np.nan)
        if np.any(np.isnan(dxdlam)) or np.linalg.norm(dxdlam) > o["max_stepsize"]:
            eflag = -1
            break
        dx, dlam = dxdlam[:nx], dxdlam[nx:]
        dz = -h - z - Ai @ dx
        dmu = -mu + zinv * (gamma - mu * dz)
        k = dz < 0
        alphap = min(o["xi"] * np.min(z[k] / -dz[k]), 1.0) if k.any() else 1.0
        k = dmu < 0
        alphad = min(o["xi"] * np.min(mu[k] / -dmu[k]), 1.0) if k.any() else 1.0
        x = x + alphap * dx
        z = z + alphap * dz
        lam = lam + alphad * dlam
        mu = mu + alphad * dmu
        if niq > 0:
            gamma = o["sigma"] * (z @ mu) / niq
        f = float(c @ x)
        h = Ai @ x - bi
        g = Ae @ x - be
        Lx = c + Ae.T @ lam + Ai.T @ mu
        feas, grad, comp, cost = conds(x, z, lam, mu, g, h, Lx, f, f0)
        if (feas < o["feastol"] and grad < o["gradtol"] and comp < o["comptol"]
                and cost < o["costtol"]):
            converged = True
        else:
            if (np.any(np.isnan(x)) or alphap < o["alpha_min"] or alphad < o["alpha_min"]
                    or gamma < eps or gamma > 1 / eps):
                eflag = -1
                break
            f0 = f
    if converged:
        eflag = 1
    return x, f, eflag, i


def mips_full(case, state, policy=REFERENCE_EMULATE, opt=None):
    """mc_simulation(state) with MATPOWER's DC-OPF + MIPS restated (unreduced KKT)."""
    lp = build_lp(case, state, policy)
    if lp["pre"]["singular"]:
        # isolated bus: KKT has an exactly zero column, MATLAB '\' returns Inf/NaN, mips breaks
        # out of iteration 1 and returns x0 (mc_simulation.m:41,54)
        x, f = lp["x0"], float(lp["c"] @ lp["x0"])
        dns, nodal = _finish(case, lp, x, f)
        return dict(dns=dns, nodal=nodal, status=ST_SINGULAR, iters=0, f=f,
                    n_relaxed=lp["pre"]["n_relaxed"])
    x, f, eflag, it = mips_lp(lp["c"], lp["A"], lp["l"], lp["u"], lp["xmin"], lp["xmax"],
                              lp["x0"], opt)
    dns, nodal = _finish(case, lp, x, f)
    status = {1: ST_CONVERGED, 0: ST_MAXIT, -1: ST_NUMFAIL}[eflag]
    return dict(dns=dns, nodal=nodal, status=status, iters=it, f=f,
                n_relaxed=lp["pre"]["n_relaxed"], x=x, inj_idx=lp["inj_idx"])


# ----------------------------------------------------------------------------
# counter-based sampling (Philox4x32-10), numpy restatement of mc_sampling.m
# ----------------------------------------------------------------------------
_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)


def philox4x32_10(ctr, key):
    """Vectorised Philox4x32-10 (Salmon et al., SC'11).  ctr: [...,4] uint32, key: [...,2]."""
    c = [np.asarray(ctr[..., i], dtype=np.uint32).copy() for i in range(4)]
    k0 = np.asarray(key[..., 0], dtype=np.uint32).copy()
    k1 = np.asarray(key[..., 1], dtype=np.uint32).copy()
    mask = np.uint64(0xFFFFFFFF)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = _M0 * c[0].astype(np.uint64)
            p1 = _M1 * c[2].astype(np.uint64)
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & mask).astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & mask).astype(np.uint32)
            c = [hi1 ^ c[1] ^ k0, lo1, hi0 ^ c[3] ^ k1, lo0]
            k0 = (k0 + _W0).astype(np.uint32)
            k1 = (k1 + _W1).astype(np.uint32)
    return np.stack(c, axis=-1)


def mc_sampling(thresholds, seed, first_index, n):
    """eqstatus[n, ncomp] uint8 (1 = failed): component k of global scenario i fails iff
    philox(ctr=(i_lo, i_hi, k>>2, 0), key=(seed_lo, seed_hi))[k&3] < thresholds[k]
    (strict '<' as mc_sampling.m:35; threshold 0 = always up, :40-41)."""
    thresholds = np.asarray(thresholds, dtype=np.uint32)
    ncomp = thresholds.size
    nblk = (ncomp + 3) // 4
    idx = np.uint64(first_index) + np.arange(n, dtype=np.uint64)
    ctr = np.zeros((n, nblk, 4), dtype=np.uint32)
    ctr[..., 0] = (idx & np.uint64(0xFFFFFFFF)).astype(np.uint32)[:, None]
    ctr[..., 1] = (idx >> np.uint64(32)).astype(np.uint32)[:, None]
    ctr[..., 2] = np.arange(nblk, dtype=np.uint32)[None, :]
    key = np.zeros((n, nblk, 2), dtype=np.uint32)
    key[..., 0] = np.uint32(seed & 0xFFFFFFFF)
    key[..., 1] = np.uint32((seed >> 32) & 0xFFFFFFFF)
    draws = philox4x32_10(ctr, key).reshape(n, nblk * 4)[:, :ncomp]
    return (draws < thresholds[None, :]).astype(np.uint8)
