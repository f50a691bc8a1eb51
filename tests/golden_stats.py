"""Statistics that pin a device run to the reference's golden artifacts (TEST INFRASTRUCTURE; numpy / scipy only, no GPU, no oracle).

The reference's goldens are single unseeded Monte Carlo runs: NSQ N = 1e5 i.i.d. samples
(Montecarlo_nsq_single/reliability_results.mat), SEQ 1 245 i.i.d. simulated years (Montecarlo_seq/seq_reliability_results.mat).  They cannot
be reproduced sample by sample (MATLAB's global stream), so the pin is statistical -- and should then use everything the goldens hold:

  NSQ   joint chi-square (Mahalanobis distance) of the golden nodal-EENS vector, of the golden importance vector and of (EDNS, PLC) from
        the converged device means, with the EXACT per-sample covariances taken from the device's unique-state database rows, scaled to the
        golden run's sample count (nsqMain.m:348-349, 366-376, 286-296).
  SEQ   two-sample Kolmogorov-Smirnov of the 1 245 golden annual (ens, dlc, nlc) against the device's years (seqMain.m:162-176); nodal
        EENS per bus and component importance against the spread of device replicas of the golden run's length (seqMain.m:218, 233).

  NSQ   per-checkpoint batch fingerprint (round 6): edns_history and beta_history are exact enough to reconstruct, for each of the golden run's
        1 000 batches of 100 samples, the batch sum S_k of dns and the batch sum of squares Q_k (nsqMain.m:286-287, 299-308).  Every S_k is an
        integer number of MW plus a 0.2-5 micro-MW fractional part: the sum of MIPS' termination residuals f + 2850 (mc_simulation.m:54) over
        the batch's shed samples -- a reference-held pin on WHERE the interior point stops (comptol / sigma / xi / z0 all move it).  34 of
        the Q_k carry one sample of 1 425 MW = half the system load: the isolated-bus state of REFERENCE_EMULATE, in the reference's own data.

Every function returns plain numbers; the callers (tests/test_gpu_parity.py, tests/test_seq.py, tests/tools/golden_pin.py) decide what to assert.
"""
from __future__ import annotations

import numpy as np
from scipy import stats


def weighted_mean_cov(x: np.ndarray, w: np.ndarray):
    """Mean vector and covariance matrix of the rows of x [n, d] under the weights w [n] (occurrence counts): exact moments of the
    per-sample distribution the rows stand for."""
    w = np.asarray(w, dtype=np.float64)
    x = np.asarray(x, dtype=np.float64)
    W = w.sum()
    mu = (w[:, None] * x).sum(0) / W
    xc = x - mu
    cov = (xc * w[:, None]).T @ xc / W
    return mu, cov


def mahalanobis_chi2(golden: np.ndarray, mu: np.ndarray, cov_per_sample: np.ndarray, n_golden: float, n_device: float | None = None,
                     keep: np.ndarray | None = None, rcond: float = 1e-10):
    """T = (g - mu)' [cov (1/n_golden + 1/n_device)]^+ (g - mu)  ->  (T, dof, p) with p = P(chi2(dof) > T).

    golden = the reference's vector (a mean over n_golden i.i.d. samples), mu / cov_per_sample = the device's mean and per-sample covariance
    (n_device samples; None = exact).  keep selects the coordinates to test; directions without variance are dropped by the pseudo-inverse
    (dof = rank) and must then agree exactly (checked by the caller)."""
    g = np.asarray(golden, dtype=np.float64); mu = np.asarray(mu, dtype=np.float64)
    if keep is None:
        keep = np.ones(g.size, dtype=bool)
    d = (g - mu)[keep]
    S = cov_per_sample[np.ix_(keep, keep)] * (1.0 / n_golden + (1.0 / n_device if n_device else 0.0))
    ev, U = np.linalg.eigh(S)
    ok = ev > rcond * ev.max()
    z = (U[:, ok].T @ d) / np.sqrt(ev[ok])
    T = float((z * z).sum())
    dof = int(ok.sum())
    return T, dof, float(stats.chi2.sf(T, dof))


def nsq_joint_pin(db: dict, golden: dict, load_bus: np.ndarray, always_up: np.ndarray, fail_threshold: float = 1e-4, min_expected: float = 5.0):
    """The three joint statistics of the NSQ golden run against a converged device database (rows of Engine.db_export()):
    'nodal' (17 load buses, chi2(17)), 'importance' (components expected down in at least `min_expected` of the golden run's failed samples),
    'edns_plc' (chi2(2)).  Each value = dict(T, dof, p)."""
    c = db["count"].astype(np.float64)
    N = c.sum()
    n_g = float(golden["n_samples"])
    out = {}
    # nodal EENS vector, nsqMain.m:348-349 (the reference's nodal_eens is the per-sample mean in MW)
    mu, cov = weighted_mean_cov(db["nodal"], c)
    T, dof, p = mahalanobis_chi2(np.array(golden["nodal_eens"]), mu, cov, n_g, N, keep=np.asarray(load_bus, dtype=bool))
    out["nodal"] = dict(T=T, dof=dof, p=p, mean=mu)
    # (EDNS, PLC), nsqMain.m:286-296
    flag = (db["dns"] > fail_threshold).astype(np.float64)
    mu2, cov2 = weighted_mean_cov(np.column_stack([db["dns"], flag]), c)
    g2 = np.array([golden["accumulated_edns"], golden["accumulated_lole"] / 8760.0])
    T, dof, p = mahalanobis_chi2(g2, mu2, cov2, n_g, N)
    out["edns_plc"] = dict(T=T, dof=dof, p=p, mean=mu2)
    # importance = P(component down | failure), nsqMain.m:366-376: given its number of failed samples the golden vector is a mean over
    # n_fail i.i.d. draws from the failed-state distribution
    f = flag > 0
    n_fail_g = g2[1] * n_g
    q, covq = weighted_mean_cov(db["states"][f].astype(np.float64), c[f])
    keep = (q * n_fail_g >= min_expected) & ~np.asarray(always_up, dtype=bool)
    T, dof, p = mahalanobis_chi2(np.array(golden["comp_importance"]), q, covq, n_fail_g, c[f].sum(), keep=keep)
    out["importance"] = dict(T=T, dof=dof, p=p, mean=q, tested=int(keep.sum()), n_fail_golden=float(n_fail_g))
    return out


def ks_two_sample(golden: np.ndarray, device: np.ndarray):
    """Two-sample Kolmogorov-Smirnov (asymptotic p; ties -- years without loss -- make it conservative).  Returns (D, p)."""
    r = stats.ks_2samp(np.asarray(golden, dtype=np.float64), np.asarray(device, dtype=np.float64), method="asymp")
    return float(r.statistic), float(r.pvalue)


def replica_z(golden: np.ndarray, replicas: np.ndarray):
    """z-scores of the golden vector against R device replicas [R, d] of the golden run's length: (g - mean) / sd over replicas
    (the replicas' spread IS the standard error of a run of that length).  Coordinates without spread get z = 0 when they agree, inf otherwise."""
    g = np.asarray(golden, dtype=np.float64)
    m = replicas.mean(0); s = replicas.std(0, ddof=1)
    z = np.zeros_like(g)
    nz = s > 0
    z[nz] = (g[nz] - m[nz]) / s[nz]
    z[~nz & (np.abs(g - m) > 1e-12)] = np.inf
    return z, m, s


def replica_chi2_rank(golden: np.ndarray, replicas: np.ndarray, keep: np.ndarray | None = None):
    """Sum of squared z-scores of the golden vector, calibrated by the replicas themselves: each replica's own statistic against the
    others (leave-one-out mean and sd) gives the null distribution, so no independence or normality is assumed (loss hours inside an outage
    event are strongly dependent).  Returns (T_golden, p_empirical, T_replicas): p = (1 + #{T_r >= T_g}) / (R + 1)."""
    R, d = replicas.shape
    if keep is None:
        keep = replicas.std(0, ddof=1) > 0
    X = replicas[:, keep]; g = np.asarray(golden, dtype=np.float64)[keep]
    tot = X.sum(0); tot2 = (X * X).sum(0)
    Tr = np.zeros(R)
    for r in range(R):
        m = (tot - X[r]) / (R - 1)
        v = (tot2 - X[r] ** 2) / (R - 1) - m * m
        v = np.maximum(v * (R - 1) / (R - 2), 1e-300)
        Tr[r] = (((X[r] - m) ** 2) / v).sum()
    m = X.mean(0); v = np.maximum(X.var(0, ddof=1), 1e-300)
    Tg = float((((g - m) ** 2) / v).sum())
    p = (1.0 + float((Tr >= Tg).sum())) / (R + 1.0)
    return Tg, p, Tr


# ---- per-checkpoint batch fingerprint of an NSQ run (nsqMain.m:286-287, 299-308: edns_history, beta_history) -------------------------------
def batch_sums(edns_history, samples_per_batch: int) -> np.ndarray:
    """S_k = sum of dns over the k-th batch, from edns_history[k] = (S_1 + ... + S_k) / (k * samples_per_batch) (nsqMain.m:286-287, 306).
    Exact to ~2 ulp of the running total (3e-10 MW at 1e5 samples), i.e. three orders below the termination residuals it resolves."""
    e = np.asarray(edns_history, dtype=np.float64)
    N = np.arange(1, e.size + 1, dtype=np.float64) * float(samples_per_batch)
    return np.diff(e * N, prepend=0.0)


def batch_sumsq(beta_history, edns_history, samples_per_batch: int) -> np.ndarray:
    """Q_k = sum of dns^2 over the k-th batch: nsqMain.m:299-301 has beta = sqrt(sum c (d - e)^2) / N / e, so the running
    sum of squares is (beta N e)^2 + N e^2.  Good to ~1e-7 relative of the running total: integer-MW^2 resolution, not the micro-MW one."""
    e = np.asarray(edns_history, dtype=np.float64); b = np.asarray(beta_history, dtype=np.float64)
    N = np.arange(1, e.size + 1, dtype=np.float64) * float(samples_per_batch)
    return np.diff((b * N * e) ** 2 + N * e * e, prepend=0.0)


def batch_fingerprint(beta_history, edns_history, samples_per_batch: int) -> dict:
    """dict(S, Q, whole, frac): whole = round(S) (MW shed by the batch when every shed state's LP value is an integer, as on RTS-24 with its
    integer unit sizes and loads), frac = S - whole = the batch's sum of termination residuals."""
    S = batch_sums(edns_history, samples_per_batch)
    Q = batch_sumsq(beta_history, edns_history, samples_per_batch)
    whole = np.round(S)
    return dict(S=S, Q=Q, whole=whole, frac=S - whole)


def event_batches(Q: np.ndarray, dns_event: float, margin: float = 0.985):
    """Batches holding at least one sample of dns >= margin * dns_event (Q_k >= (margin * dns_event)^2): boolean mask."""
    return np.asarray(Q) >= (margin * dns_event) ** 2


def event_magnitude(Q: np.ndarray, mask: np.ndarray):
    """Estimate of the event's dns from the batch sums of squares alone: sqrt(mean Q over event batches - mean Q over the others), with its
    standard error (the others' spread carries over to the event batches' remainder).  Returns (D_hat, se)."""
    Q = np.asarray(Q, dtype=np.float64); m = np.asarray(mask, dtype=bool)
    k = int(m.sum())
    if k == 0:
        return float("nan"), float("nan")
    d2 = Q[m].mean() - Q[~m].mean()
    se2 = Q[~m].std(ddof=1) * np.sqrt(1.0 / k + 1.0 / (~m).sum())
    D = float(np.sqrt(max(d2, 0.0)))
    return D, float(se2 / (2.0 * D)) if D > 0 else float("inf")
