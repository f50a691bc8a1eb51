"""CPU checks of tests/golden_stats.py (the statistics behind the golden pins of test_gpu_parity.py / test_seq.py): known answers, calibration
under the null on synthetic draws, and power against the kind of shift that separates the two singular-bus policies."""
import numpy as np
from scipy import stats

import golden_stats as gs


def test_weighted_moments_equal_the_expanded_sample():
    rng = np.random.default_rng(0)
    x = rng.normal(size=(50, 4)); w = rng.integers(1, 9, 50)
    mu, cov = gs.weighted_mean_cov(x, w)
    full = np.repeat(x, w, axis=0)
    np.testing.assert_allclose(mu, full.mean(0), rtol=1e-12)
    np.testing.assert_allclose(cov, np.cov(full.T, bias=True), rtol=1e-11, atol=1e-14)


def test_mahalanobis_is_chi2_under_the_null_and_rejects_a_shift():
    rng = np.random.default_rng(1)
    d, n_g = 6, 400
    A = rng.normal(size=(d, d)); cov = A @ A.T
    L = np.linalg.cholesky(cov)
    ps = []
    for _ in range(400):
        g = (L @ rng.normal(size=(d, n_g))).mean(1)                   # a "golden" mean over n_g i.i.d. samples
        T, dof, p = gs.mahalanobis_chi2(g, np.zeros(d), cov, n_g)
        assert dof == d
        ps.append(p)
    assert stats.kstest(ps, "uniform").pvalue > 1e-3                  # p-values uniform under the null
    shift = 5.0 * np.sqrt(np.diag(cov) / n_g)
    assert gs.mahalanobis_chi2(shift, np.zeros(d), cov, n_g)[2] < 1e-3
    # a coordinate without variance is dropped from the degrees of freedom
    cov0 = cov.copy(); cov0[0, :] = 0; cov0[:, 0] = 0
    assert gs.mahalanobis_chi2(np.zeros(d), np.zeros(d), cov0, n_g)[1] == d - 1


def test_replica_statistics_calibrate_themselves():
    rng = np.random.default_rng(2)
    R, d = 120, 10
    # dependent, skewed coordinates (like loss hours inside outage events): the null distribution must come from the replicas
    base = rng.gamma(0.5, 1.0, size=(R + 1, d)); X = base + 0.7 * base[:, :1]
    Tg, p, Tr = gs.replica_chi2_rank(X[0], X[1:])
    assert Tr.shape == (R,) and 1.0 / (R + 1) <= p <= 1.0
    far = X[1:].mean(0) + 12 * X[1:].std(0, ddof=1)
    assert gs.replica_chi2_rank(far, X[1:])[1] == 1.0 / (R + 1)
    z, m, s = gs.replica_z(far, X[1:])
    np.testing.assert_allclose(z, 12.0, rtol=1e-9)
    const = np.ones((R, 2)); zz, _, _ = gs.replica_z(np.array([1.0, 2.0]), const)
    assert zz[0] == 0 and np.isinf(zz[1])


def test_ks_sees_a_zero_inflated_mixture_change():
    rng = np.random.default_rng(3)
    def years(n, p_event):                                            # most years lose little, event years lose thousands of MWh
        return np.where(rng.random(n) < p_event, rng.gamma(2.0, 4000.0, n), rng.gamma(0.3, 300.0, n))
    assert gs.ks_two_sample(years(1245, 0.27), years(100_000, 0.27))[1] > 0.001
    assert gs.ks_two_sample(years(1245, 0.27), years(100_000, 0.02))[1] < 1e-9
