"""Where the device's rare extra iterations come from: MIPS on one state with the Newton step solved three ways --
pivoted LU of the reduced (theta, lambda) system, a 2x2-block LDL' of the same system in a static order without pivoting (what
the kernel does, here in numpy), and the latter with the soft injection of every bus taken from its balance row -- printing the
residual of the UNREDUCED KKT system per block of rows.  CPU only; developer tool (uses the numpy oracle)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po

def block_ldl_solve(Kr, rr, nth, pairs, extra, refine=0):
    """2x2 block pivots (theta_i, lambda_row) in the order of `pairs`; `extra` = unknowns without a partner (eliminated last, dense LU)."""
    n = Kr.shape[0]
    perm = [k for pr in pairs for k in pr] + list(extra)
    P = np.array(perm)
    A = Kr[np.ix_(P, P)].copy(); b = rr[P].copy()
    A0 = A.copy(); b0 = b.copy()
    nb2 = 2 * len(pairs)
    def factor_solve(A, b):
        A = A.copy(); b = b.copy()
        for k in range(0, nb2, 2):
            D = A[k:k+2, k:k+2]
            m, bb, e = D[0,0], D[0,1], -D[1,1]
            det = m*e + bb*bb
            Pinv = np.array([[e, bb],[bb, -m]]) / det
            L = A[k+2:, k:k+2] @ Pinv
            A[k+2:, k+2:] -= L @ A[k:k+2, k+2:]
            b[k+2:] -= L @ b[k:k+2]
            A[k+2:, k:k+2] = L
            A[k:k+2, k:k+2] = Pinv
        x = np.zeros(n)
        if n > nb2:
            x[nb2:] = np.linalg.solve(A[nb2:, nb2:], b[nb2:])
        for k in range(nb2-2, -1, -2):
            x[k:k+2] = A[k:k+2,k:k+2] @ (b[k:k+2] - A[k:k+2, k+2:] @ x[k+2:])
        return x
    x = factor_solve(A, b)
    for _ in range(refine):
        r = b0 - A0 @ x
        x = x + factor_solve(A, r)
    out = np.zeros(n); out[P] = x
    return out


import importlib.util
spec=importlib.util.spec_from_file_location("trc",os.path.join(ROOT,"tests/test_random_cases.py"))
m=importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
from oracle import coracle

def run(lp, mode, refine=0, trace=False, order=None, balance=False):
    c,A,l,u,xmin,xmax,x0,nth = lp["c"],lp["A"],lp["l"],lp["u"],lp["xmin"],lp["xmax"],lp["x0"],lp["nb"]
    o = dict(po.MIPS_DEFAULTS); nx = x0.size; eps = np.finfo(float).eps
    AA = np.vstack([np.eye(nx), A]); ll = np.concatenate([xmin, l]); uu = np.concatenate([xmax, u])
    ieq = np.flatnonzero(np.abs(uu - ll) <= eps)
    igt = np.flatnonzero((uu >= 1e10) & (ll > -1e10)); ilt = np.flatnonzero((ll <= -1e10) & (uu < 1e10))
    ibx = np.flatnonzero((np.abs(uu - ll) > eps) & (uu < 1e10) & (ll > -1e10))
    Ae, be = AA[ieq], uu[ieq]
    Ai = np.vstack([AA[ilt], -AA[igt], AA[ibx], -AA[ibx]]); bi = np.concatenate([uu[ilt], -ll[igt], uu[ibx], -ll[ibx]])
    neq, niq = Ae.shape[0], Ai.shape[0]
    x = x0.copy(); f = float(c @ x); h = Ai @ x - bi; g = Ae @ x - be; gamma = 1.0
    lam = np.zeros(neq); z = o["z0"] * np.ones(niq); mu = z.copy()
    k = h < -o["z0"]; z[k] = -h[k]; k = gamma / z > o["z0"]; mu[k] = gamma / z[k]
    f0 = f; Lx = c + Ae.T @ lam + Ai.T @ mu
    Ath, Ap = Ae[:, :nth], Ae[:, nth:]
    # pairing: balance row r of bus i pairs with theta_i; pinned theta rows are "extra"
    pairs = []; extra = []
    pin_rows = [r for r in range(neq) if not Ap[r].any() and np.count_nonzero(Ath[r]) == 1]
    pinned = {int(np.flatnonzero(Ath[r])[0]): r for r in pin_rows}
    bal_rows = [r for r in range(neq) if r not in pin_rows]
    # balance row -> its bus: row of Amis for bus i has diagonal Bbus[i,i] > 0 largest
    row_bus = {}
    for r in bal_rows:
        row_bus[r] = int(np.argmax(Ath[r]))
    buses = [row_bus[r] for r in bal_rows]
    assert len(set(buses)) == len(buses)
    for r in bal_rows:
        i = row_bus[r]
        if i in pinned: continue
        pairs.append((i, nth + r))
    if order is not None: pairs = [pairs[k] for k in order(len(pairs))]
    for i, r in pinned.items():
        extra += [i, nth + r]
        # the balance row of the pinned bus
        rb = [q for q in bal_rows if row_bus[q] == i]
        extra += [nth + q for q in rb]
    covered = set(k for pr in pairs for k in pr) | set(extra)
    extra += [k for k in range(nth + neq) if k not in covered]
    conv = False; i = 0; eflag = 0
    def conds():
        feas = max(np.abs(g).max(), h.max()) / (1 + max(np.abs(x).max(), np.abs(z).max()))
        grad = np.abs(Lx).max() / (1 + max(np.abs(lam).max(), np.abs(mu).max()))
        comp = (z @ mu) / (1 + np.abs(x).max()); cost = abs(f - f0) / (1 + abs(f0))
        return feas, grad, comp, cost
    while not conv and i < o["max_it"]:
        i += 1
        zinv = 1.0 / z
        M = Ai.T @ ((mu * zinv)[:, None] * Ai); N = Lx + Ai.T @ ((mu * h + gamma) * zinv)
        K = np.block([[M, Ae.T], [Ae, np.zeros((neq, neq))]]); rhs = np.concatenate([-N, -g])
        ref = np.linalg.solve(K, rhs)
        Mth = M[:nth, :nth]; D = np.diag(M)[nth:].copy(); Nth, Np = N[:nth], N[nth:]
        E = Ap @ (Ap / D).T
        Kr = np.block([[Mth, Ath.T], [Ath, -E]]); rr = np.concatenate([-Nth, -g + Ap @ (Np / D)])
        if mode == "lu": s = np.linalg.solve(Kr, rr)
        else: s = block_ldl_solve(Kr, rr, nth, pairs, extra, refine)
        dth, dl = s[:nth], s[nth:]
        dp = (-Np - Ap.T @ dl) / D
        if balance:
            for r in range(neq):
                js = np.flatnonzero(Ap[r] != 0)
                if js.size == 0: continue
                j = js[np.argmin(D[js])]
                others = Ap[r, js] @ dp[js] - Ap[r, j] * dp[j]
                dp[j] = (-g[r] - Ath[r] @ dth - others) / Ap[r, j]
        sol = np.concatenate([dth, dp, dl])
        res = K @ sol - rhs
        if trace: print("   res: th %.1e p %.1e eq %.1e |" % (np.abs(res[:nth]).max(), np.abs(res[nth:nx]).max(), np.abs(res[nx:]).max()), end="")
        if trace: print(i, "relerr step %.2e" % (np.linalg.norm(sol - ref) / np.linalg.norm(ref)), "gamma %.1e" % gamma, "minD %.1e maxE %.1e minMdiag %.1e" % (D.min(), np.diag(E).max(), np.diag(Mth).min()), end="")
        dx, dlam = sol[:nx], sol[nx:]
        dz = -h - z - Ai @ dx; dmu = -mu + zinv * (gamma - mu * dz)
        k = dz < 0; alphap = min(o["xi"] * np.min(z[k] / -dz[k]), 1.0) if k.any() else 1.0
        k = dmu < 0; alphad = min(o["xi"] * np.min(mu[k] / -dmu[k]), 1.0) if k.any() else 1.0
        x = x + alphap * dx; z = z + alphap * dz; lam = lam + alphad * dlam; mu = mu + alphad * dmu
        gamma = o["sigma"] * (z @ mu) / niq
        f = float(c @ x); h = Ai @ x - bi; g = Ae @ x - be; Lx = c + Ae.T @ lam + Ai.T @ mu
        feas, grad, comp, cost = conds()
        if trace: print("    alpha %.3f %.3f feas %.1e grad %.1e comp %.1e cost %.1e" % (alphap, alphad, feas, grad, comp, cost))
        if feas < o["feastol"] and grad < o["gradtol"] and comp < o["comptol"] and cost < o["costtol"]: conv = True
        else:
            if np.any(np.isnan(x)) or alphap < o["alpha_min"] or alphad < o["alpha_min"] or gamma < eps or gamma > 1 / eps:
                eflag = -1; break
            f0 = f
    return i, (1 if conv else eflag), f

if __name__ == "__main__":
    s=[c for c in m.CASES if c[1]==16][0]
    seed, nb, chords, ng, lbs, tight, par, pminf = s
    case=m.random_case(np.random.default_rng(1000+seed), nb, chords, ng, lbs, tight, par, pminf)
    orc=coracle.Oracle(case)
    st=orc.mc_sampling(seed,0,4000)
    lp=po.build_lp(case, st[1498], 0)
    print(run(lp,"lu"))
    print(run(lp,"ldl",trace=True))
    print(run(lp,"ldl",balance=True,trace=True))
