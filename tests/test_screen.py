"""The zero-curtailment pre-screen (relmc_solver_opts.screen = 1, csrc/relmc_screen.hip).

Soundness rests on mc_simulation.m:57-59 (dns < 0.1 -> 0) and :65 (nodal shed only when dns > 0): a state with a proven LP optimum of zero
curtailment has the outputs (0, zeros) whatever the interior point does.  What is tested:
  CPU  the host model of the certificate (tests/tools/screen_model.py) issues no false certificate against the C oracle -- fixtures, sampled
       states, sequential hours at their load factors, the random networks of test_random_cases.py;
  GPU  the device's certificate equals the model's state by state; with screen = 1 every accumulator but the iteration sum (and n_screened)
       equals screen = 0 -- integers exactly, fp64 sums to summation order -- on 2e6 RTS-24 / 3e5 RTS-96 samples under both policies, on the
       per-sample dns behind the checkpoint histories, on the state database's rows, on 125 sequential years and on the random networks.
"""
import importlib.util
import os

import numpy as np
import pytest

from powersystemsreliabilityassessment_amd import _abi, api, case96, loadcurve, seq

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, rel))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    return m


@pytest.fixture(scope="module")
def model():
    return _load("screen_model", "tests/tools/screen_model.py")


@pytest.fixture(scope="module")
def case96_():
    return case96.rts96()


@pytest.fixture(scope="module")
def oracle96(case96_):
    from oracle import coracle
    return coracle.Oracle(case96_)


# ---------------------------------------------------------------------------------------------- CPU: the model against the oracle
def test_model_certificate_is_sound_on_rts24(case, oracle, model, states_fixture):
    ptdf, lodf = model.tables(case)
    # flows of the intact system at peak load, all units on, proportional dispatch: inside every rating (the base case is certified)
    assert model.certify(case, ptdf, lodf, np.zeros((1, case.ncomp), dtype=np.uint8))[0]
    # line 11 (bus 7 - bus 8) is RTS-24's only bridge; its outage is never certified, alone or with a second line
    assert np.flatnonzero(np.isnan(lodf[0])).tolist() == [10]
    st = states_fixture["matrix"]
    for pol in (_abi.RELMC_REFERENCE_EMULATE, _abi.RELMC_PHYSICAL):
        ref = oracle.mc_simulation(st, pol, nthreads=8)
        cert = model.certify(case, ptdf, lodf, st)
        assert not np.any(cert & (ref["dns"] != 0)) and not np.any(cert & (ref["status"] != 0)) and not np.any(cert & (ref["relaxed"] != 0))
    d = oracle.nsq_database(1, beta_limit=0.0, max_iterations=400_000, samples_per_batch=100_000, nthreads=8)
    cert = model.certify(case, ptdf, lodf, d["states"])
    c = d["count"].astype(float)
    assert not np.any(cert & (d["dns"] != 0)) and not np.any(cert & (d["status"] != 0)) and not np.any(cert & (d["relaxed"] != 0))
    share, of_zero = c[cert].sum() / c.sum(), c[cert].sum() / c[d["dns"] == 0].sum()
    print(f"\nRTS-24 certificate: {share:.4f} of 4e5 samples, {of_zero:.4f} of the zero-curtailment ones")
    assert 0.905 < share < 0.92 and of_zero > 0.9995
    # base topology only (what VERDICT r5 probed at 85 %; 86.5 % with flows allowed ON their rating); one line out: 91.4 %; two: 91.5 % = all of them
    base = model.certify(case, ptdf, lodf, d["states"], max_lines_out=0)
    assert 0.84 < c[base].sum() / c.sum() < 0.87 and not np.any(base & ~cert)


def test_model_certificate_is_sound_on_sequential_hours(case, oracle, model):
    """Contingency hours of three simulated years at their own load factors (seqMain.m:97-133): 99 % certified, none falsely."""
    ptdf, lodf = model.tables(case)
    rel = seq.seqmeantime(); lf = loadcurve.anloducurve(8736)[2]
    n_cert = n_cont = 0
    for y in range(3):
        st = oracle.seq_mcsampling(rel, 8736, 1, y, 1)
        hrs = np.flatnonzero(st.any(1))
        r = oracle.seq_mcsimulation(st[hrs], lf[hrs], nthreads=8)
        cert = model.certify(case, ptdf, lodf, st[hrs], load_scale=lf[hrs])
        assert not np.any(cert & (r["dns"] != 0)) and not np.any(cert & (r["relaxed"] != 0))
        n_cert += int(cert.sum()); n_cont += hrs.size
    assert n_cert / n_cont > 0.96          # 0.979 on these three years, 0.992 over twenty (tests/tools/screen_model.py seq 20)


def test_model_certificate_is_sound_on_random_networks(model):
    """The random networks of test_random_cases.py (2 ... 100 buses; double circuits, unlimited branches, Pmin > 0, forced-up components):
    no certified state sheds load in the oracle."""
    from oracle import coracle
    trc = _load("trc", "tests/test_random_cases.py")
    tot = cert_tot = 0
    for spec in trc.CASES:
        seed, nb, chords, ng, lbs, tight, par, pminf = spec
        case = trc.random_case(np.random.default_rng(1000 + seed), nb, chords, ng, lbs, tight, par, pminf)
        orc = coracle.Oracle(case)
        n = 1500 if nb <= 32 else 400
        st = orc.mc_sampling(seed, 0, n)
        ref = orc.mc_simulation(st, _abi.RELMC_PHYSICAL, nthreads=8)
        ptdf, lodf = model.tables(case)
        cert = model.certify(case, ptdf, lodf, st)
        assert not np.any(cert & (ref["dns"] != 0)), spec
        tot += n; cert_tot += int(cert.sum())
    assert cert_tot > 0.2 * tot


def _library_tables(case):
    """relmc_debug_screen_tables (host only): dict(pmin, rng, f_min, f_rng, f_load, lim, gpair [nl, ng, 2], hmat [nl, nl], bridge, sums) or None."""
    import ctypes as C
    from powersystemsreliabilityassessment_amd import _lib
    L = _lib.load()
    f = L.relmc_debug_screen_tables
    f.restype = C.c_int32
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    ng, nl = case.ng, case.nl
    n = 2 * ng + 4 * nl + 2 * nl * ng + nl * nl
    out = np.zeros(n); br = np.zeros(nl, dtype=np.uint8); sums = np.zeros(2)
    holder = _abi.CaseHolder(case)
    rc = f(C.byref(holder.desc), out.ctypes.data, n, br.ctypes.data, sums.ctypes.data)
    assert rc in (0, 1), rc
    if rc == 0:
        return None
    o = 0
    def take(k, shape=None):
        nonlocal o
        a = out[o:o + k]; o += k
        return a if shape is None else a.reshape(shape)
    return dict(pmin=take(ng), rng=take(ng), f_min=take(nl), f_rng=take(nl), f_load=take(nl), lim=take(nl), gpair=take(2 * nl * ng, (nl, ng, 2)),
                hmat=take(nl * nl, (nl, nl)), bridge=br.astype(bool), sums=sums)


def test_library_tables_equal_numpy_ptdf_and_lodf(case, case96_, model):
    """The certificate's tables as relmc_case_load builds them (host arithmetic, relmc_debug_screen_tables: no device) against numpy's PTDF / outage matrix
    of tests/tools/screen_model.py, on RTS-24, RTS-96 and the random networks; a case without a certificate says so."""
    import dataclasses
    trc = _load("trc", "tests/test_random_cases.py")
    cases = [case, case96_] + [trc.random_case(np.random.default_rng(1000 + sp[0]), *sp[1:]) for sp in trc.CASES]
    for c in cases:
        t = _library_tables(c)
        assert t is not None
        ptdf, lodf = model.tables(c)
        ng = c.ng
        pg = ptdf[:, c.inj_bus[:ng]]                                     # [nl, ng]
        np.testing.assert_allclose(t["gpair"][:, :, 0], pg * c.inj_pmin[:ng], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(t["gpair"][:, :, 1], pg * (c.inj_pmax[:ng] - c.inj_pmin[:ng]), rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(t["f_min"], pg @ c.inj_pmin[:ng], rtol=1e-9, atol=1e-8)
        np.testing.assert_allclose(t["f_load"], ptdf @ c.bus_pd, rtol=1e-9, atol=1e-8)
        assert np.array_equal(t["bridge"], np.isnan(lodf[0]))
        H = model.ptdf_h(c)[1]
        np.testing.assert_allclose(t["hmat"], H.T, rtol=1e-8, atol=1e-10)                   # library: [line out m][line l] = H[l, m]
        lim = np.where(c.br_rate > 0, c.br_rate + 1e-9, np.inf)
        assert np.array_equal(t["lim"], lim) and t["sums"][0] == pytest.approx(c.inj_pmin[:ng].sum()) and t["sums"][1] == pytest.approx((c.inj_pmax[:ng] - c.inj_pmin[:ng]).sum())
    keep = np.arange(case.nl) != 10                                      # RTS-24 without the branch that holds bus 7: not one island, no PTDF
    cut = dataclasses.replace(case, nl=case.nl - 1, br_from=case.br_from[keep], br_to=case.br_to[keep], br_b=case.br_b[keep], br_rate=case.br_rate[keep],
                              unavail=np.concatenate([case.unavail[:case.ng], case.unavail[case.ng:][keep]]),
                              always_up=np.concatenate([case.always_up[:case.ng], case.always_up[case.ng:][keep]]), elim_order=None)
    assert _library_tables(cut) is None


# ---------------------------------------------------------------------------------------------- GPU
def _split(acc):
    """(integers that must be identical, iteration sum, n_screened, doubles)"""
    ai, ad = acc.to_arrays()
    return np.concatenate([ai[:5], ai[6:-1]]), int(ai[5]), int(ai[-1]), ad


@pytest.mark.gpu
def test_device_certificate_equals_the_model(engine, oracle, model, states_fixture, case):
    ptdf, lodf = model.tables(case)
    st = np.vstack([states_fixture["matrix"], oracle.mc_sampling(9, 0, 60_000)])
    dev = engine.screen_states(st)
    mod = model.certify(case, ptdf, lodf, st)
    assert np.array_equal(dev, mod) and 0.85 < dev[len(states_fixture["matrix"]):].mean() < 0.93
    ref = oracle.mc_simulation(states_fixture["matrix"], _abi.RELMC_REFERENCE_EMULATE, nthreads=8)
    assert not np.any(dev[:len(ref["dns"])] & (ref["dns"] != 0))
    # at load scale factors (the sequential track's hours)
    rng = np.random.default_rng(4)
    sc = rng.uniform(0.3, 1.0, st.shape[0])
    assert np.array_equal(engine.screen_states(st, sc), model.certify(case, ptdf, lodf, st, load_scale=sc))
    assert engine.screen_states(st[:0]).shape == (0,)


@pytest.mark.gpu
@pytest.mark.parametrize("policy", [api.REFERENCE_EMULATE, api.PHYSICAL])
def test_screened_accumulate_equals_unscreened_rts24_2e6(engine, policy, capsys):
    n = 2_000_000
    a = engine.nsq_accumulate(3, 10**9, n, api.mpoption(policy)); ta = engine.last_kernel_ms()
    b = engine.nsq_accumulate(3, 10**9, n, api.mpoption(policy, screen=1)); tb = engine.last_kernel_ms()
    ia, ita, sa, da = _split(a); ib, itb, sb, db = _split(b)
    assert np.array_equal(ia, ib) and sa == 0 and 0.905 * n < sb < 0.92 * n and itb < 0.13 * ita
    np.testing.assert_allclose(db, da, rtol=1e-13, atol=0)
    assert a.n == b.n == n and a.n_nonconverged == b.n_nonconverged == 0
    with capsys.disabled():
        print(f"\n   pre-screen RTS-24 policy {policy}: {sb / n:.4f} of 2e6 samples certified, {ta:.2f} -> {tb:.2f} ms", end="")
    # a range that is all certificates but a handful, a range of one sample, an empty range
    c0, c1 = engine.nsq_accumulate(3, 5, 1, api.mpoption(policy)), engine.nsq_accumulate(3, 5, 1, api.mpoption(policy, screen=1))
    assert np.array_equal(_split(c0)[0], _split(c1)[0]) and c1.n == 1
    assert engine.nsq_accumulate(3, 5, 0, api.mpoption(policy, screen=1)).n == 0


@pytest.mark.gpu
def test_screened_per_batch_dedupe(engine, case96_, capsys):
    """relmc_nsq_accumulate_distinct behind the pre-screen: the certificate first, then sort + run-length encoding of the uncovered samples only and one
    solve per distinct state among them.  Integers = the fused pass' (screened and not), sums to their order, n_screened = the fused screened pass'."""
    n = 1_000_000
    for eng, m in ((engine, n), (api.Engine(case96_), 300_000)):
        for policy in (api.REFERENCE_EMULATE, api.PHYSICAL):
            plain = eng.nsq_accumulate(5, 77, m, api.mpoption(policy))
            fused = eng.nsq_accumulate(5, 77, m, api.mpoption(policy, screen=1))
            d0, nd0 = eng.nsq_accumulate_distinct(5, 77, m, api.mpoption(policy)); t0 = eng.last_kernel_ms()
            d1, nd1 = eng.nsq_accumulate_distinct(5, 77, m, api.mpoption(policy, screen=1)); t1 = eng.last_kernel_ms()
            for x in (fused, d0, d1):
                assert np.array_equal(_split(x)[0], _split(plain)[0])
                np.testing.assert_allclose(_split(x)[3], _split(plain)[3], rtol=1e-12, atol=0)
            assert d1.n_screened == fused.n_screened > 0.9 * m and d0.n_screened == 0 and d1.n == m
            assert 0 < nd1 < nd0 and nd1 < m - d1.n_screened + 1          # distinct states among the uncovered samples only
        with capsys.disabled():
            print(f"\n   per-batch dedupe behind the pre-screen, {eng.case.nb} buses, {m} samples: {nd0} -> {nd1} distinct states solved, {t0:.2f} -> {t1:.2f} ms", end="")
        # one sample, an empty range, a range the certificate covers entirely
        assert eng.nsq_accumulate_distinct(5, 3, 1, api.mpoption(screen=1))[0].n == 1 and eng.nsq_accumulate_distinct(5, 3, 0, api.mpoption(screen=1))[0].n == 0
        if eng is not engine: eng.close()
    # the sampling loop in this mode: the same stopping batch and indices as without the pre-screen
    a = engine.nsqMain(beta_limit=0.01, max_iterations=5_000_000, samples_per_batch=100_000, seed=2, distinct_states="batch")
    b = engine.nsqMain(beta_limit=0.01, max_iterations=5_000_000, samples_per_batch=100_000, seed=2, distinct_states="batch", mpopt=api.mpoption(screen=1))
    assert a.current_iteration == b.current_iteration and a.plc == b.plc and b.n_screened > 0.9 * b.current_iteration and a.n_screened == 0
    np.testing.assert_allclose([b.accumulated_edns, b.current_beta], [a.accumulated_edns, a.current_beta], rtol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("policy", [api.REFERENCE_EMULATE, api.PHYSICAL])
def test_screened_accumulate_equals_unscreened_rts96_3e5(case96_, policy, capsys):
    eng = api.Engine(case96_)
    try:
        n = 300_000
        a = eng.nsq_accumulate(1, 0, n, api.mpoption(policy)); ta = eng.last_kernel_ms()
        b = eng.nsq_accumulate(1, 0, n, api.mpoption(policy, screen=1)); tb = eng.last_kernel_ms()
        ia, ita, sa, da = _split(a); ib, itb, sb, db = _split(b)
        assert np.array_equal(ia, ib) and sa == 0 and 0.975 * n < sb < 0.99 * n
        np.testing.assert_allclose(db, da, rtol=1e-13, atol=0)
        with capsys.disabled():
            print(f"\n   pre-screen RTS-96 policy {policy}: {sb / n:.4f} of 3e5 samples certified, {ta:.2f} -> {tb:.2f} ms", end="")
    finally:
        eng.close()


@pytest.mark.gpu
def test_device_certificate_is_sound_on_rts96(case96_, oracle96, model):
    eng = api.Engine(case96_)
    try:
        st = oracle96.mc_sampling(2, 0, 20_000)
        dev = eng.screen_states(st)
        ptdf, lodf = model.tables(case96_)
        assert np.array_equal(dev, model.certify(case96_, ptdf, lodf, st))
        ref = oracle96.mc_simulation(st, _abi.RELMC_REFERENCE_EMULATE, nthreads=16)
        assert not np.any(dev & (ref["dns"] != 0)) and not np.any(dev & (ref["status"] != 0)) and dev.mean() > 0.95
    finally:
        eng.close()


@pytest.mark.gpu
def test_screened_checkpoint_histories_and_stopping_point(engine):
    """relmc_nsq_run at the reference's checkpoint spacing of 100 (nsqMain.m:60): the per-sample dns behind the histories is the unscreened
    run's, so the loop stops at the same checkpoint with the same histories."""
    for limit, cap in ((0.01, 400_000), (0.0, 150_000)):
        a = engine.nsqMain(beta_limit=limit, max_iterations=cap, samples_per_batch=100, seed=1)
        b = engine.nsqMain(beta_limit=limit, max_iterations=cap, samples_per_batch=100, seed=1, mpopt=api.mpoption(screen=1))
        assert a.current_iteration == b.current_iteration and a.converged == b.converged and len(a.beta_history) == len(b.beta_history)
        assert np.array_equal(a.plc_history, b.plc_history)
        np.testing.assert_allclose(b.edns_history, a.edns_history, rtol=1e-13)
        np.testing.assert_allclose(b.beta_history, a.beta_history, rtol=1e-10)
        # the first stretch (24 576 samples) is the per-sample values themselves summed in sampling order on the host: bit-identical; later
        # checkpoints start from the accumulators of the stretches before, whose fp64 sums differ in their summation order (1e-16)
        assert np.array_equal(a.edns_history[:240], b.edns_history[:240]) and np.array_equal(a.beta_history[:240], b.beta_history[:240])
        assert np.array_equal(_split(a.acc)[0], _split(b.acc)[0]) and b.n_screened > 0.85 * b.current_iteration and a.n_screened == 0
    assert a.current_iteration == 150_000


@pytest.mark.gpu
def test_screened_database_rows(engine):
    """The state database with the pre-screen: rows, counts, dns, flags and nodal split identical; certified rows carry 0 iterations and the
    screened bit, n_screened is their count-weighted number."""
    kw = dict(beta_limit=0.0, max_iterations=1_000_000, samples_per_batch=250_000, seed=2, distinct_states="database")
    a = engine.nsqMain(**kw); ra = engine.db_export(); engine.db_reset()
    b = engine.nsqMain(mpopt=api.mpoption(screen=1), **kw); rb = engine.db_export(); engine.db_reset()
    assert a.database_row_count == b.database_row_count
    for k in ("states", "count", "dns", "flag", "nodal", "status"):
        assert np.array_equal(ra[k], rb[k]), k
    assert np.array_equal(ra["relaxed"], rb["relaxed"] & 1) and not (ra["relaxed"] & 2).any()          # bit 1 of the exported column: certified, never solved
    assert np.array_equal(_split(a.acc)[0], _split(b.acc)[0])
    np.testing.assert_allclose(_split(b.acc)[3], _split(a.acc)[3], rtol=1e-13)
    skipped = (rb["relaxed"] & 2) != 0
    assert np.array_equal(skipped, (rb["iters"] == 0) & (rb["status"] == 0)) and np.all(rb["dns"][skipped] == 0) and b.n_screened == int(rb["count"][skipped].sum())
    # export -> import -> export keeps the bit, and with it n_screened of the resumed database
    engine.nsqMain(mpopt=api.mpoption(screen=1), **kw); rows = engine.db_export(); engine.db_reset()
    engine.db_import(rows, mpopt=api.mpoption(screen=1))
    assert engine.db_accumulate().n_screened == b.n_screened and np.array_equal(engine.db_export()["relaxed"], rows["relaxed"])
    engine.db_reset()
    assert 0.5 < skipped.mean() < 0.8 and np.array_equal(ra["iters"][~skipped], rb["iters"][~skipped])
    assert b.n_screened > 0.88 * b.current_iteration and a.n_screened == 0
    # a database filled without the pre-screen refuses a batch with it (results of other solver options)
    engine.nsq_db_batch(2, 0, 1000)
    with pytest.raises(api.RelmcError, match="other solver options"):
        engine.nsq_db_batch(2, 1000, 1000, api.mpoption(screen=1))
    engine.db_reset()


@pytest.mark.gpu
def test_screened_sequential_years(engine, capsys):
    """125 simulated years (BASELINE configs[3] per GPU): annual (ens, dlc, nlc, contingency hours) identical, accumulators identical but
    for the iteration sum; ~99 % of the contingency hours are certified at their own load factor."""
    se = seq.SeqEngine(engine)
    n = 125
    e0, d0, l0, c0, a0 = se.seq_years(1, 0, n); t0 = engine.last_kernel_ms()
    e1, d1, l1, c1, a1 = se.seq_years(1, 0, n, mpopt=api.mpoption(screen=1)); t1 = engine.last_kernel_ms()
    assert np.array_equal(d0, d1) and np.array_equal(l0, l1) and np.array_equal(c0, c1)
    np.testing.assert_allclose(e1, e0, rtol=1e-13, atol=0)
    assert np.array_equal(e0, e1)                               # hourly curtailments summed per year in the same order
    i0, it0, s0, dd0 = _split(a0); i1, it1, s1, dd1 = _split(a1)
    assert np.array_equal(i0, i1) and s0 == 0 and s1 > 0.98 * a1.n and a0.n == a1.n == int(c0.sum())
    np.testing.assert_allclose(dd1, dd0, rtol=1e-13, atol=0)
    with capsys.disabled():
        print(f"\n   pre-screen sequential: {s1 / a1.n:.4f} of {a1.n} contingency hours certified, {t0:.2f} -> {t1:.2f} ms per 125 years", end="")
    r0 = se.seqMain(max_sim_years=300, cov_threshold=0.0)
    r1 = se.seqMain(max_sim_years=300, cov_threshold=0.0, mpopt=api.mpoption(screen=1))
    assert r0.final_year == r1.final_year == 300 and r0.eens == r1.eens and r0.lole == r1.lole and r0.lolf == r1.lolf
    np.testing.assert_allclose(r1.nodal_eens_avg, r0.nodal_eens_avg, rtol=1e-12)
    assert np.array_equal(r0.comp_importance, r1.comp_importance) and np.array_equal(r0.results_cum["cov"], r1.results_cum["cov"])


@pytest.mark.gpu
def test_screened_random_networks(model):
    """The random networks: device certificate = model, screened accumulators = unscreened, a case whose base topology has a bridge or
    unlimited branches included."""
    from oracle import coracle
    trc = _load("trc", "tests/test_random_cases.py")
    for spec in trc.CASES:
        seed, nb, chords, ng, lbs, tight, par, pminf = spec
        case = trc.random_case(np.random.default_rng(1000 + seed), nb, chords, ng, lbs, tight, par, pminf)
        eng = api.Engine(case)
        try:
            st = eng.mc_sampling(None, 3000, seed=seed)
            ptdf, lodf = model.tables(case)
            assert np.array_equal(eng.screen_states(st), model.certify(case, ptdf, lodf, st)), spec
            for pol in (api.REFERENCE_EMULATE, api.PHYSICAL):
                a, b = eng.nsq_accumulate(seed, 0, 20_000, api.mpoption(pol)), eng.nsq_accumulate(seed, 0, 20_000, api.mpoption(pol, screen=1))
                ia, _, _, da = _split(a); ib, _, sb, db = _split(b)
                assert np.array_equal(ia, ib), spec
                np.testing.assert_allclose(db, da, rtol=1e-12, atol=1e-9)
        finally:
            eng.close()


@pytest.mark.gpu
def test_screen_with_second_attempts_and_on_a_case_without_a_certificate(engine, case):
    """(a) The survivors' unit numbers through the retry list (MODE 7: a listed unit is named by its sample offset): with an iteration limit of 7 every
    solved unit goes to the further orders and comes back non-converged -- the per-sample dns behind the histories is mc_simulation's for the uncovered
    samples and 0 for the certified ones.  (b) A case whose base topology is not one island (RTS-24 without the branch that holds bus 7) has no PTDF: the pre-screen then
    certifies nothing and screen = 1 is screen = 0."""
    import dataclasses
    import golden_stats as gs
    # an iteration limit of 7 ends every solved unit non-converged (twice: the further orders do no better).  The identity with screen = 0 is a statement
    # about CONVERGED solves -- a certified unit is counted with its proven optimum 0, not with the last iterate of a solve that was cut short -- so the
    # reference here is mc_simulation of the same samples under the same limit, with the certified ones set to 0
    o0, o1 = api.mpoption(max_it=7), api.mpoption(max_it=7, screen=1)
    n = 20_000
    st = engine.mc_sampling(None, n, seed=5)
    cert = engine.screen_states(st)
    dns, _, info = engine.mc_simulation(st, mpopt=o0, return_info=True)
    assert np.all(info["status"][~cert] != 0) and 0.85 < cert.mean() < 0.95
    want = np.where(cert, 0.0, dns)
    u0 = engine.retry_stats()
    rb = engine.nsqMain(beta_limit=0.0, max_iterations=n, samples_per_batch=100, seed=5, mpopt=o1)
    u1 = engine.retry_stats()
    solved = ~cert & (info["status"] != 3)                       # isolated-bus states are not iterated under REFERENCE_EMULATE
    assert u1[0] - u0[0] == int(solved.sum()) and rb.n_screened == int(cert.sum()) and rb.n_nonconverged == int(solved.sum())
    np.testing.assert_allclose(gs.batch_sums(rb.edns_history, 100), want.reshape(-1, 100).sum(1), rtol=1e-7, atol=1e-4)       # every listed unit came back to ITS sample (a misplaced one would move a batch by ~100 MW; the last iterate of a cut-short solve is not good to 1e-9)
    assert rb.acc.n_fail == int((want > 1e-4).sum())
    keep = np.arange(case.nl) != 10
    cut = dataclasses.replace(case, nl=case.nl - 1, br_from=case.br_from[keep], br_to=case.br_to[keep], br_b=case.br_b[keep], br_rate=case.br_rate[keep],
                              unavail=np.concatenate([case.unavail[:case.ng], case.unavail[case.ng:][keep]]),
                              always_up=np.concatenate([case.always_up[:case.ng], case.always_up[case.ng:][keep]]), elim_order=None)
    eng = api.Engine(cut)
    try:
        st = eng.mc_sampling(None, 2000, seed=3)
        assert not eng.screen_states(st).any()
        for pol in (api.REFERENCE_EMULATE, api.PHYSICAL):
            x, y = eng.nsq_accumulate(3, 0, 20_000, api.mpoption(pol)), eng.nsq_accumulate(3, 0, 20_000, api.mpoption(pol, screen=1))
            assert np.array_equal(x.to_arrays()[0], y.to_arrays()[0]) and y.n_screened == 0
            assert np.array_equal(x.to_arrays()[1], y.to_arrays()[1])               # the very same launches: bit for bit
    finally:
        eng.close()
