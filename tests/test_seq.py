"""Sequential HL2 track (BASELINE config 4): chronology, scaled-load DC-OPF, annual indices.

CPU tests pin the oracle (oracle/relmc_oracle.c orc_seq_*) against a pure-Python restatement of
seq_mcsampling.m, the HiGHS / numpy-MIPS fixture of seq_mcsimulation.m's scaled model and the reference's own
golden seq_reliability_results.mat.  GPU tests compare the HIP path (relmc_seq_* through the C ABI) with the oracle.
"""
import json
import math
import os

import numpy as np
import pytest

from powersystemsreliabilityassessment_amd import _abi, case24, loadcurve, seq

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
HPY = 8736


@pytest.fixture(scope="module")
def seq_golden():
    with open(os.path.join(GOLDEN, "seq_golden.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def hours_fixture(case):
    with open(os.path.join(GOLDEN, "seq_hours_fixture.json")) as f:
        d = json.load(f)
    st = np.zeros((len(d["hours"]), case.ncomp), dtype=np.uint8)
    for i, x in enumerate(d["hours"]):
        st[i, x["failed"]] = 1
    d["matrix"] = st
    d["scale"] = np.array([x["load_scale"] for x in d["hours"]])
    return d


@pytest.fixture(scope="module")
def rel():
    return seq.seqmeantime()


@pytest.fixture(scope="module")
def load_factors():
    return loadcurve.anloducurve(HPY)[2]


def _chronology_py(rel, hpy, seed, year, comps):
    """seq_mcsampling.m:35-76 restated line by line (one year, all-up start) on the counter-based stream."""
    from oracle import pyoracle as po
    out = np.zeros((hpy, rel.shape[0]), dtype=np.uint8)
    for k in comps:
        current, up, ev = 0, True, 0
        while current < hpy:
            r = po.philox4x32_10(np.array([year & 0xffffffff, year >> 32, k | 0x80000000, ev >> 2], dtype=np.uint32),
                                 np.array([seed & 0xffffffff, seed >> 32], dtype=np.uint32))
            u = (float(r[ev & 3]) + 0.5) * 2.0 ** -32
            if up:
                current += int(math.floor(-rel[k, 0] * math.log(u) + 0.5))
            else:
                dur = int(math.ceil(-rel[k, 1] * math.log(u)))
                out[current:min(current + dur, hpy), k] = 1
                current += dur
            up = not up
            ev += 1
    return out


# ---------------------------------------------------------------------------------------------- CPU
def test_seqmeantime_values(case, rel):
    """seqmeantime.m:21-36: generators keep their MTTF/MTTR, branches get 8760/lambda and r (with the brdur quirk)."""
    d = case24.case24_failrate()
    assert rel.shape == (case.ncomp, 2)
    np.testing.assert_array_equal(rel[:33, 0], d["genmttf"]); np.testing.assert_array_equal(rel[:33, 1], d["genmttr"])
    np.testing.assert_allclose(rel[33:, 0], 8760.0 / d["brlambda"])
    assert rel[33 + 10, 0] == pytest.approx(29200.0) and rel[33 + 10, 1] == 10.0          # L11: 0.3 /yr, 10 h
    assert list(np.flatnonzero(rel[33:, 1] == 768.0) + 1) == [6, 14, 15, 16, 17]           # transformer repair times
    assert rel[14, 0] == 1e4 and rel[14, 1] == 0.1                                         # synchronous condenser row


def test_calnlc():
    """calnlc.m:22-32."""
    assert seq.calnlc([0, 0, 0]) == 0
    assert seq.calnlc([1, 1, 0, 1, 0, 0, 1]) == 3
    assert seq.calnlc([0, 1, 1, 1]) == 1
    assert seq.calnlc([]) == 0


def test_oracle_chronology_matches_python_restatement(oracle, rel):
    comps = [0, 5, 14, 22, 32, 33, 38, 43, 70]
    for year in (0, 3):
        ref = _chronology_py(rel, HPY, 11, year, comps)
        got = oracle.seq_mcsampling(rel, HPY, 11, year, 1)
        np.testing.assert_array_equal(got[:, comps], ref[:, comps])
    # years are independent streams: sampling [0,3) contains year 2 sampled on its own
    a = oracle.seq_mcsampling(rel, HPY, 5, 0, 3)
    b = oracle.seq_mcsampling(rel, HPY, 5, 2, 1)
    np.testing.assert_array_equal(a[2 * HPY:], b)
    assert a[0].sum() == 0 and a[HPY].sum() == 0 and a[2 * HPY].sum() == 0                # every year starts all-up


def test_oracle_chronology_statistics(oracle, rel):
    """Down-time fraction ~ E[ceil TTR] / (MTTF + E[ceil TTR]) minus the all-up start transient."""
    ny = 40
    s = oracle.seq_mcsampling(rel, HPY, 3, 0, ny).reshape(ny, HPY, -1)
    frac = s.mean(axis=(0, 1))
    for k in (0, 8, 22, 32):                      # U12, U76, U400, U350
        mttr_eff = rel[k, 1] + 0.5                # ceil() adds half an hour on average
        expect = mttr_eff / (rel[k, 0] + mttr_eff)
        assert frac[k] == pytest.approx(expect, rel=0.25), k
    assert frac[14] < 2e-4                        # synchronous condenser: MTTF 1e4 h, MTTR 0.1 h -> 1 h outages, rarely
    assert 0.6 < (s.sum(axis=2) > 0).mean() < 0.9  # most hours have something down (seqMain's contingency hours)


def test_oracle_scaled_lp_vs_highs_and_mips(oracle, hours_fixture):
    """seq_mcsimulation.m's scaled model: oracle dns = HiGHS optimum (1e-5 MW) and = numpy MIPS in iterations."""
    for name, pol in (("emulate", _abi.RELMC_REFERENCE_EMULATE), ("physical", _abi.RELMC_PHYSICAL)):
        r = oracle.seq_mcsimulation(hours_fixture["matrix"], hours_fixture["scale"], pol, nthreads=8)
        for i, x in enumerate(hours_fixture["hours"]):
            e = x[name]
            assert r["status"][i] == e["status"], (name, i)
            assert r["iters"][i] == e["iters"], (name, i)
            assert r["dns"][i] == pytest.approx(e["dns"], abs=1e-6), (name, i)
            if e["highs_dns"] is not None and e["status"] == 0:
                hd = e["highs_dns"] if e["highs_dns"] >= 0.1 else 0.0
                assert r["dns"][i] == pytest.approx(hd, abs=2e-5), (name, i)
    # scale 1 is the non-sequential model
    a = oracle.seq_mcsimulation(hours_fixture["matrix"][:40], 1.0)
    b = oracle.mc_simulation(hours_fixture["matrix"][:40])
    np.testing.assert_array_equal(a["dns"], b["dns"])


def test_oracle_year_indices_from_parts(oracle, rel, load_factors):
    """orc_seq_years = seqMain.m:91-176 assembled from its parts (sampling, contingency hours, OPF, calnlc)."""
    yrs, acc = oracle.seq_years(rel, HPY, load_factors, 1, 1, 1)
    st = oracle.seq_mcsampling(rel, HPY, 1, 1, 1)
    hours = np.flatnonzero(st.sum(1) > 0)                                    # seqMain.m:97
    r = oracle.seq_mcsimulation(st[hours], load_factors[hours], nthreads=8)
    prof = np.zeros(HPY); prof[hours] = r["dns"]
    flag = prof > 0.01                                                       # seqMain.m:141
    assert yrs[0, 3] == hours.size and acc.n == hours.size
    assert yrs[0, 0] == pytest.approx(prof.sum(), rel=1e-12)
    assert yrs[0, 1] == flag.sum() and yrs[0, 2] == seq.calnlc(flag)
    assert acc.n_fail == flag.sum()
    np.testing.assert_array_equal(np.array(acc.comp_fail[:71]), st[flag].sum(0))
    np.testing.assert_allclose(np.array(acc.sum_nodal[:24]), r["nodal"][flag[hours]].sum(0), rtol=1e-12, atol=1e-9)


def test_oracle_vs_golden_statistics(oracle, rel, load_factors, seq_golden):
    """A short oracle run agrees with the reference's golden 1245-year run within sampling error."""
    g_ens, g_dlc, g_nlc = (np.array(seq_golden[k], dtype=float) for k in ("ens", "dlc", "nlc"))
    assert seq_golden["final_year"] == 1245
    assert g_ens.mean() == pytest.approx(seq_golden["cum_eens"][-1], rel=1e-12)
    assert g_ens.std(ddof=1) / (g_ens.mean() * math.sqrt(1245)) == pytest.approx(seq_golden["cum_cov"][-1], rel=1e-9)
    ny = 24
    yrs, acc = oracle.seq_years(rel, HPY, load_factors, 2, 0, ny)
    for col, g in ((0, g_ens), (1, g_dlc), (2, g_nlc)):
        se = g.std(ddof=1) * math.sqrt(1.0 / ny + 1.0 / g.size)
        assert abs(yrs[:, col].mean() - g.mean()) < 4 * se, (col, yrs[:, col].mean(), g.mean())
    assert acc.n_nonconverged == 0


# ---------------------------------------------------------------------------------------------- GPU
@pytest.fixture(scope="module")
def seqeng(engine):
    return seq.SeqEngine(engine)


@pytest.mark.gpu
def test_gpu_chronology_bit_exact(seqeng, oracle, rel):
    got = seqeng.seq_mcsampling(rel, 33, 38, 3, HPY, seed=9, first_year=5)
    ref = oracle.seq_mcsampling(rel, HPY, 9, 5, 3)
    assert got.shape == (71, 3 * HPY)
    np.testing.assert_array_equal(got.T, ref)


@pytest.mark.gpu
@pytest.mark.parametrize("hpy,years", [(24, 2), (257, 5), (1000, 40), (8736, 1), (8736, 200), (20000, 3), (65535, 2)])
def test_gpu_chronology_segments(engine, oracle, rel, hpy, years):
    """The chronology kernel cuts a year into hour segments by the year's length and the number of years in the call (one workgroup per
    segment, its masks in LDS): other lengths and counts than the reference's 8736 x 1, and failure-prone components (MTTF of a few hours to a
    few days: hundreds of intervals per year, many of them across segment borders), against the oracle's chronology."""
    fast = rel.copy()
    fast[::3, 0] = np.linspace(3.0, 90.0, len(fast[::3]))          # every third component fails every few hours to days
    fast[1::3, 1] = np.linspace(1.0, 400.0, len(fast[1::3]))       # ... and repairs of up to weeks
    for data in (rel, fast):
        se = seq.SeqEngine(engine, reliability_data=data, hours_per_year=hpy, load_scale_factors=np.ones(hpy))
        got = se.seq_mcsampling(num_years=years, seed=4, first_year=2**33 + 7)
        ref = oracle.seq_mcsampling(data, hpy, 4, 2**33 + 7, years)
        np.testing.assert_array_equal(got.T, ref)
    seq.SeqEngine(engine)                                          # the module's engine goes on with the reference's chronology


@pytest.mark.gpu
def test_gpu_scaled_hours_parity(seqeng, oracle, hours_fixture):
    from powersystemsreliabilityassessment_amd import api
    for name, pol in (("emulate", _abi.RELMC_REFERENCE_EMULATE), ("physical", _abi.RELMC_PHYSICAL)):
        dns, nodal, info = seqeng.seq_mcsimulation(hours_fixture["matrix"], hours_fixture["scale"], mpopt=api.mpoption(pol), return_info=True)
        r = oracle.seq_mcsimulation(hours_fixture["matrix"], hours_fixture["scale"], pol, nthreads=8)
        np.testing.assert_array_equal(info["status"], r["status"])
        np.testing.assert_array_equal(info["iters"], r["iters"])
        np.testing.assert_allclose(dns, r["dns"], rtol=0, atol=1e-6)
        np.testing.assert_allclose(nodal.sum(1), r["nodal"].sum(1), rtol=0, atol=2e-3)     # per-bus split is degenerate
        for i, x in enumerate(hours_fixture["hours"]):
            assert dns[i] == pytest.approx(x[name]["dns"], abs=1e-6)
    d1, n1 = seqeng.seq_mcsimulation(hours_fixture["matrix"][3], hours_fixture["scale"][3])
    assert isinstance(d1, float) and n1.shape == (24,)


@pytest.mark.gpu
def test_gpu_years_match_oracle(seqeng, oracle, rel, load_factors):
    ny = 3
    ens, dlc, nlc, ncont, acc = seqeng.seq_years(4, 10, ny)
    yrs, oacc = oracle.seq_years(rel, HPY, load_factors, 4, 10, ny)
    np.testing.assert_array_equal(ncont, yrs[:, 3].astype(np.int64))
    np.testing.assert_array_equal(dlc, yrs[:, 1]); np.testing.assert_array_equal(nlc, yrs[:, 2])
    np.testing.assert_allclose(ens, yrs[:, 0], rtol=1e-9, atol=1e-6)
    assert (acc.n, acc.n_fail, acc.n_singular, acc.n_infeasible, acc.n_nonconverged) == \
           (oacc.n, oacc.n_fail, oacc.n_singular, oacc.n_infeasible, oacc.n_nonconverged)
    assert acc.sum_iters == oacc.sum_iters
    np.testing.assert_array_equal(np.array(acc.comp_fail[:71]), np.array(oacc.comp_fail[:71]))
    assert sum(acc.sum_nodal[:24]) == pytest.approx(sum(oacc.sum_nodal[:24]), rel=1e-6)


@pytest.mark.gpu
def test_gpu_years_second_attempts_bookkeeping(seqeng, oracle, rel, load_factors):
    """The sequential track lists the hours it ends non-converged and re-evaluates them under the further elimination orders
    like every other path.  No hour of 2.6e7 does so naturally (scripts/seq_retry_scan.py), so the mechanics are
    exercised with an iteration limit of 7, which ends nearly every hour at MAXIT: the listed hours (4096 + n / 256 at most) come back
    through the second attempt, the others are accumulated by the kernel, and counts, loss hours and energies must be those of
    the oracle under the same limit."""
    from powersystemsreliabilityassessment_amd import api
    o = api.mpoption(api.REFERENCE_EMULATE); o.max_it = 7
    u0, v0 = seqeng.eng.retry_stats()[0], seqeng.eng.retry_overflow()
    ens, dlc, nlc, ncont, acc = seqeng.seq_years(4, 10, 2, mpopt=o)
    u1, v1 = seqeng.eng.retry_stats()[0], seqeng.eng.retry_overflow()
    yrs, oacc = oracle.seq_years(rel, HPY, load_factors, 4, 10, 2, opts=o)
    assert acc.n == oacc.n == int(ncont.sum()) and acc.n_nonconverged == oacc.n_nonconverged > 0.5 * acc.n
    cap = 4096 + acc.n // 256                     # the list holds at least this for a call of acc.n units (more if an earlier call grew it)
    assert (u1 - u0) + (v1 - v0) == acc.n_nonconverged and u1 - u0 >= min(acc.n_nonconverged, cap)
    assert (acc.n_fail, acc.n_singular, acc.n_infeasible, acc.sum_iters) == (oacc.n_fail, oacc.n_singular, oacc.n_infeasible, oacc.sum_iters)
    np.testing.assert_array_equal(dlc, yrs[:, 1]); np.testing.assert_array_equal(nlc, yrs[:, 2])
    np.testing.assert_allclose(ens, yrs[:, 0], rtol=1e-7)
    np.testing.assert_array_equal(np.array(acc.comp_fail[:71]), np.array(oacc.comp_fail[:71]))
    assert acc.sum_dns == pytest.approx(oacc.sum_dns, rel=1e-7)


@pytest.mark.gpu
def test_gpu_seqmain_vs_golden(seqeng, seq_golden, tmp_path):
    """BASELINE config 4: run seqMain to CoV < 5 % and compare with the reference's golden run (1245 years,
    EENS 4266.87 MWh/yr, LOLE 14.33 h/yr, LOLF 2.465 occ/yr) within sampling error of both runs."""
    r = seqeng.seqMain(seed=1)
    g_ens, g_dlc, g_nlc = (np.array(seq_golden[k], dtype=float) for k in ("ens", "dlc", "nlc"))
    assert 0 < r.cov < 0.05 and 300 < r.final_year < 4000
    assert r.results_cum["cov"][-2] >= 0.05 or r.final_year == 2
    n = r.final_year
    for mine, g in ((r.results_year["ens"], g_ens), (r.results_year["dlc"], g_dlc), (r.results_year["nlc"], g_nlc)):
        se = math.sqrt(np.var(mine, ddof=1) / n + g.var(ddof=1) / g.size)
        assert abs(np.mean(mine) - g.mean()) < 4 * se, (np.mean(mine), g.mean(), se)
    assert r.eens == pytest.approx(np.mean(r.results_year["ens"])) and r.lole == pytest.approx(np.mean(r.results_year["dlc"]))
    assert r.years_evaluated == r.final_year and r.total_loss_hours == int(r.results_year["dlc"].sum())
    # weak points: the same three units lead (the per-bus nodal EENS and the whole importance vector are pinned against the golden run's
    # own standard errors by test_gpu_seq_golden_distribution_pin below)
    g_imp = np.array(seq_golden["comp_importance"])
    assert set(np.argsort(-r.comp_importance)[:3]) == set(np.argsort(-g_imp)[:3]) == {22, 23, 32}
    assert r.nodal_eens_avg.sum() == pytest.approx(r.eens, rel=0.02)
    r.write_nodal_csv(str(tmp_path / "seq_nodal_results.csv")); r.save_mat(str(tmp_path / "seq_reliability_results.mat"))
    csv = np.loadtxt(str(tmp_path / "seq_nodal_results.csv"), delimiter=",", skiprows=1)
    np.testing.assert_allclose(csv[:, 1], r.nodal_eens_avg, rtol=1e-12)


@pytest.mark.gpu
def test_gpu_seq_run_below_the_abi_equals_the_python_loop(seqeng):
    """relmc_seq_run (the seqMain loop, its CoV stop and post-processing below the C ABI) against the Python restatement of the same loop
    around relmc_seq_years (dist.seq_run_distributed, the loop the gloo test checks the sharding arithmetic of): same stopping year, annual
    indices bit for bit, CoV curve to rounding, accumulators cut at the stopping year, whatever the batch size."""
    from powersystemsreliabilityassessment_amd import dist as rdist

    def fn(sd, first, n):
        e, d, n_, _, acc = seqeng.seq_years(sd, first, n)
        return e, d, n_, acc
    ref = rdist.seq_run_distributed(fn, seed=5, cov_threshold=0.08, max_sim_years=2000, batch_years=100, rank=0, world=1)
    for by in (0, 33, 700):
        r = seqeng.seqMain(2000, 0.08, seed=5, batch_years=by)
        assert r.converged and r.final_year == ref["final_year"] and 50 < r.final_year < 2000
        np.testing.assert_array_equal(np.column_stack([r.results_year["ens"], r.results_year["dlc"], r.results_year["nlc"]]), ref["years"])
        np.testing.assert_allclose(r.results_cum["cov"][1:], ref["cum_cov"][1:], rtol=1e-12)
        np.testing.assert_allclose(r.results_cum["eens"], ref["cum_eens"], rtol=1e-13)
        assert r.results_cum["cov"][0] == 0.0 and r.cov == r.results_cum["cov"][-1] < 0.08 <= r.results_cum["cov"][-2]
        ai, ad = r.acc.to_arrays(); bi, bd = ref["acc"].to_arrays()
        np.testing.assert_array_equal(ai, bi)
        np.testing.assert_allclose(ad, bd, rtol=1e-12, atol=1e-9)
        assert r.total_loss_hours == int(r.results_year["dlc"].sum()) and r.lole == pytest.approx(ref["lole"]) and r.lolf == pytest.approx(ref["lolf"])
        np.testing.assert_allclose(r.nodal_eens_avg, np.array(ref["acc"].sum_nodal[:24]) / r.final_year, rtol=1e-12)
    # no convergence within the horizon: every year kept, converged = False; a horizon of one year; options the library refuses
    r = seqeng.seqMain(40, 1e-6, seed=5)
    assert not r.converged and r.final_year == 40 and r.years_evaluated == 40
    r1 = seqeng.seqMain(1, 0.05, seed=5)
    assert r1.final_year == 1 and not r1.converged and r1.cov == 0.0 and r1.results_year["ens"][0] == r.results_year["ens"][0]
    import ctypes as C
    o = _abi.SeqOpts(); seqeng.L.relmc_seq_opts_default(C.byref(o)); res = _abi.SeqResult()
    buf = np.zeros(10)
    o.max_years = 100; o.years_cap = 10; o.cum_eens = buf.ctypes.data_as(_abi.c_double_p)          # history buffer shorter than the horizon
    assert seqeng.L.relmc_seq_run(seqeng.eng._h, C.byref(o), C.byref(res)) == -1 and b"years_cap" in seqeng.L.relmc_last_error(seqeng.eng._h)
    o.cum_eens = None; o.max_years = 0
    assert seqeng.L.relmc_seq_run(seqeng.eng._h, C.byref(o), C.byref(res)) == -1


@pytest.mark.gpu
def test_gpu_seq_run_zero_ens_leading_years_and_batch_independent_bookkeeping(seqeng):
    """relmc_seq_run: (a) while no simulated year has had curtailment the running CoV is 0 / 0 = NaN as seqMain.m:184 -- cum_cov says so,
    the stop test (:194) rejects it, and a run that ends there returns cov = NaN (documented in relmc.h); (b) the stopping year, kernel_seconds' order
    of magnitude and the second-attempt counters do not depend on batch_years (a batch that is cut at the stopping year is taken again over its
    used part and the discarded pass is erased from the bookkeeping)."""
    seed = next(s for s in range(1, 200) if seqeng.seq_years(s, 0, 2)[0].sum() == 0.0)          # a seed whose first two years lose nothing
    lead = seqeng.seqMain(max_sim_years=2, cov_threshold=0.05, seed=seed)
    assert lead.final_year == 2 and not lead.converged and lead.eens == 0.0 and np.isnan(lead.cov)
    assert lead.results_cum["cov"][0] == 0.0 and np.isnan(lead.results_cum["cov"][1])
    runs = {}
    for by in (0, 33, 700):
        u0 = seqeng.eng.retry_stats()
        r = seqeng.seqMain(max_sim_years=700, cov_threshold=0.12, seed=seed, batch_years=by)
        u1 = seqeng.eng.retry_stats()
        runs[by] = (r, (u1[0] - u0[0], u1[1] - u0[1]))
    r0 = runs[0][0]
    assert r0.converged and 2 < r0.final_year < 700 and np.isnan(r0.results_cum["cov"][1]) and r0.cov < 0.12
    first_loss = int(np.flatnonzero(r0.results_year["ens"] > 0)[0])
    assert first_loss >= 2 and np.all(np.isnan(r0.results_cum["cov"][1:first_loss])) and np.all(np.isfinite(r0.results_cum["cov"][first_loss:]))
    for by in (33, 700):
        r, units = runs[by]
        assert r.final_year == r0.final_year and r.cov == r0.cov and np.array_equal(r.results_year["ens"], r0.results_year["ens"])
        assert units == runs[0][1]                                            # second attempts: the same units whatever the batch
        assert r.n_lp == r0.n_lp and np.array_equal(r.acc.to_arrays()[0], r0.acc.to_arrays()[0])
        assert 0.5 < r.kernel_seconds / r0.kernel_seconds < 2.0                # the discarded pass of a cut batch is not in kernel_seconds


@pytest.mark.gpu
def test_gpu_long_run_tightens_on_golden(seqeng, seq_golden):
    """20 000 simulated years (about 1.4e8 hourly OPFs): the golden means sit within the golden run's own error."""
    ens, dlc, nlc = [], [], []
    for b in range(20):
        e, d, n_, _, acc = seqeng.seq_years(77, b * 1000, 1000)
        ens.append(e); dlc.append(d); nlc.append(n_)
        assert acc.n_nonconverged == 0
    for mine, key in ((np.concatenate(ens), "ens"), (np.concatenate(dlc), "dlc"), (np.concatenate(nlc), "nlc")):
        g = np.array(seq_golden[key], dtype=float)
        se = math.sqrt(mine.var(ddof=1) / mine.size + g.var(ddof=1) / g.size)
        assert abs(mine.mean() - g.mean()) < 4 * se, (key, mine.mean(), g.mean(), se)


@pytest.mark.gpu
def test_gpu_seq_golden_distribution_pin(seqeng, seq_golden, capsys):
    """The reference's golden sequential run holds a DISTRIBUTION: 1 245 annual (ens, dlc, nlc) triples (seqMain.m:162-176), 24 nodal EENS
    (:218) and 71 importances (:233).  Against 100 device replicas of 1 245 years (124 500 years) per policy:
      * two-sample Kolmogorov-Smirnov of the golden annual ens / dlc / nlc against all device years: REFERENCE_EMULATE passes at p > 0.01,
        PHYSICAL is REJECTED (p < 1e-6 on ens and dlc) -- the test sees the 1/2-load artifact of the isolated-bus hours that carries about
        two thirds of the sequential EENS;
      * nodal EENS of every one of the 17 load buses within 4 standard errors of a 1 245-year run (= the spread of the replicas), and the
        17-vector's sum of squared z-scores inside the replicas' own distribution of that statistic;
      * the importance vector the same way (loss hours inside one outage event are dependent, so the null distribution comes from the
        replicas, not from a binomial), L11's share within 4 standard errors -- 0.223 golden against 0.232 +- 0.015 here, 0.0013 under PHYSICAL.
    Measured (profiles/r4_final/golden_pin.log, 200 replicas): emulate KS p = 0.39 / 0.77 / 0.76, nodal max |z| 1.3, importance p = 0.93;
    physical KS p = 2e-47 / 9e-17 / 3e-10, bus 7 z = 79, L11 z = 235."""
    import golden_stats as gs
    from powersystemsreliabilityassessment_amd import api
    case = seqeng.eng.case
    Y, R = seq_golden["final_year"], 100
    load_bus = case.bus_pd > 0
    g_nodal = np.array(seq_golden["nodal_eens_avg"]); g_imp = np.array(seq_golden["comp_importance"])
    assert Y == 1245 and int(load_bus.sum()) == 17
    out = {}
    for name, pol in (("emulate", api.REFERENCE_EMULATE), ("physical", api.PHYSICAL)):
        yrs = {k: [] for k in ("ens", "dlc", "nlc")}
        nod = np.zeros((R, case.nb)); imp = np.zeros((R, case.ncomp))
        for r in range(R):
            e, d, n_, _, acc = seqeng.seq_years(1, r * Y, Y, mpopt=api.mpoption(pol))
            yrs["ens"].append(e); yrs["dlc"].append(d); yrs["nlc"].append(n_)
            assert acc.n_nonconverged == 0 and acc.n_fail == int(d.sum())
            nod[r] = np.array(acc.sum_nodal[:case.nb]) / Y                                    # seqMain.m:218
            imp[r] = np.array(acc.comp_fail[:case.ncomp], dtype=float) / max(1, acc.n_fail)   # seqMain.m:233
        o = {k: gs.ks_two_sample(np.array(seq_golden[k], dtype=float), np.concatenate(yrs[k])) for k in yrs}
        o["z_nodal"], _, _ = gs.replica_z(g_nodal, nod)
        o["nodal_T"], o["nodal_p"], _ = gs.replica_chi2_rank(g_nodal, nod, keep=load_bus)
        keep = (imp.mean(0) > 2e-3) & ~case.always_up.astype(bool)
        o["imp_T"], o["imp_p"], _ = gs.replica_chi2_rank(g_imp, imp, keep=keep)
        o["z_imp"], o["imp_mean"], _ = gs.replica_z(g_imp, imp)
        out[name] = o
    with capsys.disabled():
        for name, o in out.items():
            print("\n   golden SEQ run vs %s: KS p ens %.3g dlc %.3g nlc %.3g; nodal max |z| %.2f, vector p %.3f; importance p %.3f, L11 z %.2f"
                  % (name, o["ens"][1], o["dlc"][1], o["nlc"][1], np.abs(o["z_nodal"][load_bus]).max(), o["nodal_p"], o["imp_p"], o["z_imp"][43]), end="")
    e, ph = out["emulate"], out["physical"]
    assert min(e["ens"][1], e["dlc"][1], e["nlc"][1]) > 0.01
    assert np.abs(e["z_nodal"][load_bus]).max() < 4.0 and e["nodal_p"] > 0.01
    assert np.all(g_nodal[~load_bus] == 0) and np.all(e["z_nodal"][~load_bus] == 0)
    assert e["imp_p"] > 0.01 and abs(e["z_imp"][43]) < 4.0
    # ... and the policy the reference does not follow is rejected on every count
    assert ph["ens"][1] < 1e-6 and ph["dlc"][1] < 1e-6 and ph["nlc"][1] < 1e-3
    assert abs(ph["z_nodal"][6]) > 10 and ph["nodal_p"] < 0.02 and ph["imp_p"] < 0.02 and abs(ph["z_imp"][43]) > 10


# ---------------------------------------------------------------------------------------------- multi-rank (CPU, gloo)
_SEQ_WORKER = """
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch.distributed as dist
from powersystemsreliabilityassessment_amd import case24, dist as rdist, loadcurve, seq
from oracle import coracle
rank, world = int(sys.argv[1]), int(sys.argv[2])
if world > 1:
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[3], RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
orc = coracle.Oracle(case24.rts24()); rel = seq.seqmeantime(); lf = loadcurve.anloducurve(8736)[2]
def fn(seed, first, n):                                  # stand-in evaluator (oracle) for SeqEngine.seq_years
    y, acc = orc.seq_years(rel, 8736, lf, seed, first, n, nthreads=4)
    return y[:, 0], y[:, 1], y[:, 2], acc
r = rdist.seq_run_distributed(fn, seed=6, cov_threshold=0.62, max_sim_years=9, batch_years=3, rank=rank, world=world)
if rank == 0:
    ti, td = r["acc"].to_arrays()
    np.savez(sys.argv[4], ti=ti, td=td, years=r["years"], cov=r["cum_cov"], final_year=r["final_year"])
if world > 1:
    dist.destroy_process_group()
"""


def test_seq_distributed_driver_gloo_world2(tmp_path):
    """Year sharding over 2 ranks = the single-process loop: same stopping year, same annual indices, same accumulators."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "seq_worker.py"
    script.write_text(_SEQ_WORKER.format(root=root))
    port = str(29950 + os.getpid() % 40)
    outs = [tmp_path / "w2.npz", tmp_path / "w1.npz"]
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2", port, str(outs[0])]) for r in range(2)]
    procs.append(subprocess.Popen([sys.executable, str(script), "0", "1", port, str(outs[1])]))
    for p in procs:
        assert p.wait(timeout=900) == 0
    a, b = np.load(outs[0]), np.load(outs[1])
    assert int(a["final_year"]) == int(b["final_year"]) >= 2
    np.testing.assert_array_equal(a["years"], b["years"])
    np.testing.assert_array_equal(a["cov"], b["cov"])
    np.testing.assert_array_equal(a["ti"], b["ti"])
    np.testing.assert_allclose(a["td"], b["td"], rtol=1e-12, atol=1e-9)
    assert int(a["ti"][0]) > 0
