"""IEEE RTS-96 (BASELINE config 5): case builder, oracle pinning, GPU parity
on the one-scenario-per-wavefront tile."""
import json
import os

import numpy as np
import pytest

from powersystemsreliabilityassessment_amd import _abi, case24, case96

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def case96_():
    return case96.rts96()


@pytest.fixture(scope="module")
def oracle96(case96_):
    from oracle import coracle
    return coracle.Oracle(case96_)


@pytest.fixture(scope="module")
def fixture96(case96_):
    with open(os.path.join(GOLDEN, "rts96_states_fixture.json")) as f:
        d = json.load(f)
    st = np.zeros((len(d["states"]), case96_.ncomp), dtype=np.uint8)
    for i, x in enumerate(d["states"]):
        st[i, x["failed"]] = 1
    d["matrix"] = st
    return d


def _three_area_states(case96_, area_states):
    """RTS-96 states made of three RTS-24 states (one per area) with the five tie lines out: three electrically separate RTS-24 systems
    (bus 325 hangs on 323 through its transformer, which stays in service: a dead end without load or generation)."""
    k = area_states.shape[0] // 3
    st = np.zeros((k, case96_.ncomp), dtype=np.uint8)
    for a in range(3):
        part = area_states[a * k:(a + 1) * k]
        st[:, 33 * a:33 * (a + 1)] = part[:, :33]
        st[:, 99 + 38 * a:99 + 38 * (a + 1)] = part[:, 33:]
    st[:, 99 + 114:99 + 119] = 1
    return st


# ---------------------------------------------------------------------------------------------- CPU
def test_case96_construction(case96_):
    c = case96_
    assert (c.nb, c.ng, c.nl, c.nd, c.ncomp) == (73, 99, 120, 51, 219)
    assert c.total_load == 8550.0 and c.inj_pmax[:99].sum() == 3 * 3405.0
    assert list(np.flatnonzero(c.always_up)) == [14, 47, 80]                       # the three synchronous condensers
    assert c.ref_bus == case96.bus_index(113) == 12
    assert case96.bus_index(325) == 72 and case96.bus_index(201) == 24 and case96.bus_index(324) == 71
    ties = list(zip(c.br_from[114:], c.br_to[114:]))
    assert ties == [(6, 26), (12, 38), (22, 40), (72, 20), (65, 46), (70, 72)]
    np.testing.assert_allclose(c.br_b[114:], 1.0 / np.array([0.161, 0.075, 0.074, 0.097, 0.104, 0.009]))
    # every area carries the RTS-24 data
    r24 = case24.rts24()
    for a in range(3):
        np.testing.assert_array_equal(c.br_b[38 * a:38 * (a + 1)], r24.br_b)
        np.testing.assert_array_equal(c.unavail[33 * a:33 * (a + 1)], r24.unavail[:33])
        np.testing.assert_array_equal(c.unavail[99 + 38 * a:99 + 38 * (a + 1)], r24.unavail[33:])
        np.testing.assert_array_equal(c.bus_pd[24 * a:24 * (a + 1)], r24.bus_pd)
    u = c.unavail[99 + 114:]
    np.testing.assert_allclose(u, [0.44 / (0.44 + 876), 0.47 / (0.47 + 8760 / 11), 0.46 / (0.46 + 8760 / 11), 0.52 / (0.52 + 8760 / 11),
                                   0.54 / (0.54 + 8760 / 11), 0.02 / (0.02 + 8760 / 768)], rtol=1e-12)
    assert case96.seqmeantime96().shape == (219, 2)


def test_oracle96_with_the_ties_out_is_three_rts24_systems(oracle96, oracle, case96_):
    """A pin of the RTS-96 case (recalled, not validated by the reference) that does not depend on the recalled part: with
    the five tie lines out the 73-bus LP decouples into three RTS-24 LPs, so its curtailment must be the sum of what the RTS-24 case -- the one
    the reference's goldens pin -- gives for the three area states (the LP optimum is unique), area by area in the nodal sums too.  Checks the
    replication of the area data, the bus / branch / generator numbering and the island rules (areas 2 and 3 have no reference bus)."""
    rng = np.random.default_rng(96)
    k = 16
    areas = (rng.random((3 * k, 71)) < 0.09).astype(np.uint8)       # heavy outages: most areas shed load
    areas[:, 14] = 0
    st = _three_area_states(case96_, areas)
    for pol in (_abi.RELMC_PHYSICAL,):
        r96 = oracle96.mc_simulation(st, pol, nthreads=8)
        r24 = oracle.mc_simulation(areas, pol, nthreads=8)
        want = r24["dns"][:k] + r24["dns"][k:2 * k] + r24["dns"][2 * k:]
        ok = np.all(np.isin(r96["status"], (0,))) and np.all(r24["status"] == 0)
        assert ok and (want > 1.0).sum() >= k // 2
        np.testing.assert_allclose(r96["dns"], want, rtol=0, atol=2e-5)
        for a in range(3):
            np.testing.assert_allclose(r96["nodal"][:, 24 * a:24 * (a + 1)].sum(1), r24["nodal"][a * k:(a + 1) * k].sum(1), rtol=0, atol=2e-2)
        assert np.all(r96["nodal"][:, 72] == 0)


def test_oracle96_vs_fixture(oracle96, fixture96):
    """C oracle on RTS-96 = numpy MIPS restatement in iterations/status, = HiGHS optimum in value."""
    for name, pol in (("emulate", _abi.RELMC_REFERENCE_EMULATE), ("physical", _abi.RELMC_PHYSICAL)):
        r = oracle96.mc_simulation(fixture96["matrix"], pol, nthreads=8)
        off_by_one = 0
        for i, x in enumerate(fixture96["states"]):
            e = x[name]
            assert r["status"][i] == e["status"], (name, i)
            # two fp64 implementations of the same iteration may stop one iteration apart when a termination
            # test lands within rounding of its tolerance (1 of 317 states here); never more
            assert abs(r["iters"][i] - e["iters"]) <= 1, (name, i)
            off_by_one += int(r["iters"][i] != e["iters"])
            assert r["dns"][i] == pytest.approx(e["dns"], abs=1e-6), (name, i)
            if e["highs_dns"] is not None and e["status"] == 0:
                hd = e["highs_dns"] if e["highs_dns"] >= 0.1 else 0.0
                assert r["dns"][i] == pytest.approx(hd, abs=5e-5), (name, i)
        assert off_by_one <= 3
    assert sum(1 for x in fixture96["states"] if x["emulate"]["status"] == 3) >= 5     # isolated-bus states are covered


# ---------------------------------------------------------------------------------------------- GPU
@pytest.fixture(scope="module")
def engine96(case96_):
    from powersystemsreliabilityassessment_amd import api
    eng = api.Engine(case96_, device=0)
    yield eng
    eng.close()


@pytest.mark.gpu
def test_gpu96_sampling_bit_exact(engine96, oracle96, case96_):
    got = engine96.mc_sampling(case96_.unavail, 4096, 99, 120, seed=5, first_index=10**12)
    ref = oracle96.mc_sampling(5, 10**12, 4096)
    np.testing.assert_array_equal(np.asarray(got.todense() if hasattr(got, "todense") else got, dtype=np.uint8), ref)
    np.testing.assert_array_equal(engine96.thresholds(), oracle96.thresholds())
    assert ref[:, [14, 47, 80]].sum() == 0


@pytest.mark.gpu
def test_gpu96_states_parity(engine96, oracle96, fixture96):
    from powersystemsreliabilityassessment_amd import api
    for name, pol in (("emulate", _abi.RELMC_REFERENCE_EMULATE), ("physical", _abi.RELMC_PHYSICAL)):
        dns, nodal, info = engine96.mc_simulation(fixture96["matrix"], mpopt=api.mpoption(pol), return_info=True)
        r = oracle96.mc_simulation(fixture96["matrix"], pol, nthreads=16)
        bad = np.flatnonzero((info["status"] != r["status"]) | (np.abs(info["iters"] - r["iters"]) > 1))
        assert bad.size == 0, (name, bad[:10], info["status"][bad[:10]], r["status"][bad[:10]], info["iters"][bad[:10]], r["iters"][bad[:10]])
        assert int((info["iters"] != r["iters"]).sum()) <= 3          # termination test within rounding of its tolerance
        np.testing.assert_allclose(dns, r["dns"], rtol=0, atol=1e-5)
        np.testing.assert_allclose(nodal.sum(1), r["nodal"].sum(1), rtol=0, atol=5e-3)
        assert (nodal >= 0).all() and (nodal <= engine96.case.bus_pd[None, :] + 1e-6).all()
        np.testing.assert_allclose(nodal.sum(1)[dns > 0], dns[dns > 0], rtol=0, atol=5e-3)
        for i, x in enumerate(fixture96["states"]):
            assert dns[i] == pytest.approx(x[name]["dns"], abs=1e-5)


@pytest.mark.gpu
def test_gpu96_accumulate_matches_oracle(engine96, oracle96):
    n = 20000                                                     # ~ 290 shedding samples; the oracle takes ~ 25 s on the GPU box's cores
    acc = engine96.nsq_accumulate(7, 123456, n)
    ref = oracle96.nsq_accumulate(7, 123456, n, _abi.RELMC_REFERENCE_EMULATE)
    ai, ad = acc.to_arrays(); ri, rd = ref.to_arrays()
    np.testing.assert_array_equal(ai[:5], ri[:5])                 # n, n_fail, n_singular, n_infeasible, n_nonconverged
    assert abs(int(ai[5]) - int(ri[5])) <= 20                     # sum of iterations (see test_gpu96_states_parity)
    np.testing.assert_array_equal(ai[6:], ri[6:])                 # component-down counts during loss
    np.testing.assert_allclose(ad[:2], rd[:2], rtol=1e-8)         # sum dns, sum dns^2
    # Per-bus sums of the sampled run: 2 % (measured round 3: 0.84 %, 23 MW on the worst bus).  The split of a state's curtailment over
    # the buses is a point of a degenerate optimal face; on the heavy-outage states of the fixtures (317 + 67 states when measured, 112 GW shed in
    # total) device and C oracle end tens of MW apart on single buses with totals equal to 5e-4 MW (tests/tools/nodal96_agg.py: per-bus
    # sums over those states differ by 4 % in the median, 23 % at worst; the C oracle and the numpy restatement do the same to each
    # other), which is why that set pins totals, not buses.
    assert ad[2:].sum() == pytest.approx(rd[2:].sum(), rel=1e-6)
    np.testing.assert_allclose(ad[2:], rd[2:], rtol=2e-2, atol=1.0)
    assert acc.n == n and acc.n_nonconverged == 0


@pytest.mark.gpu
def test_gpu96_with_the_ties_out_is_three_rts24_systems(engine96, engine, case96_):
    """The same decoupling on the device, on sampled area states at two outage levels: the 64-lane tile's curtailment of a three-island RTS-96
    state equals the sum of the 16-lane tile's on the three RTS-24 area states (1e-5 MW), per area in the nodal sums, under both policies
    wherever no bus is isolated (multi-bus islands without the reference bus are solved island-aware in both)."""
    from powersystemsreliabilityassessment_amd import api
    rng = np.random.default_rng(97)
    k = 400
    sampled = engine.mc_sampling(None, 3 * k, seed=96, first_index=0)
    heavy = (rng.random((3 * k, 71)) < 0.08).astype(np.uint8); heavy[:, 14] = 0
    # bus 323 keeps the transformer to 325 when all its RTS-24 lines are out (a two-bus island, not an isolated bus): keep one of them in service
    l23 = 33 + int(np.flatnonzero((case24.BR_FROM == 23) | (case24.BR_TO == 23))[0])
    sampled[2 * k:, l23] = 0; heavy[2 * k:, l23] = 0
    for areas in (sampled, heavy):
        st = _three_area_states(case96_, areas)
        for pol in (api.PHYSICAL, api.REFERENCE_EMULATE):
            d96, n96, i96 = engine96.mc_simulation(st, mpopt=api.mpoption(pol), return_info=True)
            d24, n24, i24 = engine.mc_simulation(areas, mpopt=api.mpoption(pol), return_info=True)
            sing24 = (i24["status"] == 3).reshape(3, k).any(0)                 # an isolated bus in some area: the emulated singular case
            assert np.array_equal(i96["status"] == 3, sing24), (np.flatnonzero((i96["status"] == 3) != sing24)[:5], np.unique(i96["status"]), np.unique(i24["status"]))
            conv = (i96["status"] == 0) & (i24["status"] == 0).reshape(3, k).all(0)      # heavy-outage states may end non-converged on either tile (6.7e-7 of the sampled ones)
            assert conv.sum() >= (~sing24).sum() - 2
            want = d24[:k] + d24[k:2 * k] + d24[2 * k:]
            part = d24.reshape(3, k)
            clean = conv & ~sing24 & ~(((part > 0) & (part < 0.2)).any(0)) & ~((want > 0) & (want < 0.2))      # away from the 0.1 MW noise filter
            assert clean.sum() > 0.9 * k or pol == api.REFERENCE_EMULATE
            np.testing.assert_allclose(d96[clean], want[clean], rtol=0, atol=1e-5)
            for a in range(3):
                np.testing.assert_allclose(n96[clean][:, 24 * a:24 * (a + 1)].sum(1), n24[a * k:(a + 1) * k][clean].sum(1), rtol=0, atol=2e-2)
            assert np.all(n96[:, 72] == 0)
        assert (want > 1.0).sum() > (20 if areas is sampled else k // 2)


@pytest.mark.gpu
@pytest.mark.parametrize("policy", [_abi.RELMC_REFERENCE_EMULATE, _abi.RELMC_PHYSICAL], ids=["emulate", "physical"])
def test_gpu96_sampled_state_contract_5e4(engine96, oracle96, policy, capsys):
    """The numerical contract on SAMPLED RTS-96 states (the first 5e4 samples of seed 1, device against the C oracle, state by state), under
    BOTH policies (round 4 ran REFERENCE_EMULATE only): status identical, |dns difference| <= 1e-6 MW, iteration counts equal but for +-1 on
    fewer than 0.1 % of the states, per-bus nodal sums to 1 % (emulate) / 2 % (physical) with the worst bus printed (round 3 kept this as a builder-run log over 2e5 samples: 0 / 0 / 0.0135 %,
    profiles/r3_final/sampled_vs_oracle_rts96.log)."""
    from powersystemsreliabilityassessment_amd import api
    n = 50_000
    st = engine96.mc_sampling(None, n, seed=1, first_index=0)
    dns, nodal, info = engine96.mc_simulation(st, mpopt=api.mpoption(policy), return_info=True)
    ref = oracle96.mc_simulation(st, policy, nthreads=16)
    np.testing.assert_array_equal(info["status"], ref["status"])
    assert np.abs(dns - ref["dns"]).max() <= 1e-6
    di = np.abs(info["iters"] - ref["iters"])
    assert int((di > 1).sum()) == 0 and int((di == 1).sum()) < n // 1000
    dev_n, orc_n = nodal.sum(0), ref["nodal"].sum(0)
    m = orc_n > 0
    rel = np.zeros_like(orc_n); rel[m] = np.abs(dev_n[m] - orc_n[m]) / orc_n[m]
    worst = int(np.argmax(rel))
    with capsys.disabled():
        print(f"\n   RTS-96 {'emulate' if policy == 0 else 'physical'}: per-bus nodal sums over 5e4 sampled states, worst bus {worst + 1}: {rel[worst]:.2e} "
              f"(dns sum rel {abs(dns.sum() - ref['dns'].sum()) / ref['dns'].sum():.1e}; iterations +-1 on {int((di == 1).sum())} states)", end="")
    # the split of a shed among equally priced loads is a point of a degenerate optimal face: measured over 3e5 states 0.6 % (emulate) /
    # 1.6 % (physical, bus 13) (profiles/r5_final/sampled_vs_oracle_rts96.log); bounds = that + margin, per policy (round 5 allowed 2 % for both)
    np.testing.assert_allclose(dev_n[m], orc_n[m], rtol=1e-2 if policy == _abi.RELMC_REFERENCE_EMULATE else 2e-2, atol=1.0)
    assert np.all(dev_n[~m] == 0)


@pytest.mark.gpu
def test_gpu96_run_to_convergence(engine96, case96_):
    """BASELINE config 5 shape: RTS-96 NSQ to beta < 2 %; the copper-sheet COPT of the 96-unit fleet bounds PLC from below."""
    from powersystemsreliabilityassessment_amd import hl1
    r = engine96.nsqMain(beta_limit=0.02, max_iterations=4_000_000, samples_per_batch=200_000, seed=1)
    # a few scenarios per million pass the optimum but miss MIPS' gradient test by rounding noise (KKT conditioning
    # ~ 1/gamma) and end "numerically failed" with the optimal curtailment: never more than 1e-5
    assert r.converged and r.n_nonconverged <= max(1, r.current_iteration // 100000)
    d = case96.failrate96()
    gens = [hl1.Generator(k, float(case96_.inj_pmax[k]), float(d["genmttf"][k]), float(d["genmttr"][k])) for k in range(99) if case96_.inj_pmax[k] > 0]
    exact = hl1.run_analytical(gens, hl1.LoadModel(np.array([8550.0])), step_size=1.0)
    plc_lb = exact.lole_hours_yr                     # one "hour" of constant peak load -> probability
    se = np.sqrt(r.plc * (1 - r.plc) / r.current_iteration)
    assert r.plc > plc_lb - 4 * se
    assert r.accumulated_edns > exact.eue_mwh_yr - 4 * r.current_beta * r.accumulated_edns
    assert np.all(np.asarray(r.nodal_eens)[[10, 11, 16, 20, 21, 22, 23, 72]] == 0)      # buses without load never shed


@pytest.mark.gpu
def test_gpu96_distinct_state_path(engine96):
    n = 200000
    plain = engine96.nsq_accumulate(9, 0, n)
    acc, nd = engine96.nsq_accumulate_distinct(9, 0, n)
    pi, pd = plain.to_arrays(); ai, ad = acc.to_arrays()
    np.testing.assert_array_equal(ai, pi)
    np.testing.assert_allclose(ad, pd, rtol=1e-10, atol=1e-7)
    assert 0 < nd <= n


@pytest.mark.gpu
def test_gpu96_state_database_matches_oracle(engine96, oracle96):
    """The persistent unique-state database (nsqMain.m:220-278) on the wide tile (219-bit keys) against the oracle's
    database-form restatement of the loop: same rows in the same order, counts, flags, dns; and against the per-sample path."""
    from powersystemsreliabilityassessment_amd import api
    n, batch, seed = 2400, 800, 6
    r = engine96.nsqMain(beta_limit=0.0, max_iterations=n, samples_per_batch=batch, seed=seed, distinct_states="database")
    db = engine96.db_export()
    ref = oracle96.nsq_database(seed, 0.0, n, batch, nthreads=16, max_rows=n)
    assert r.database_row_count == len(ref["count"]) == len(db["count"]) and db["count"].sum() == n
    np.testing.assert_array_equal(db["states"], ref["states"])
    np.testing.assert_array_equal(db["count"], ref["count"])
    np.testing.assert_array_equal(db["flag"], ref["flag"])
    np.testing.assert_array_equal(db["status"], ref["status"])
    np.testing.assert_allclose(db["dns"], ref["dns"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(r.edns_history, ref["edns_history"], rtol=1e-8)
    np.testing.assert_allclose(r.beta_history, ref["beta_history"], rtol=1e-6)
    ai, ad = r.acc.to_arrays(); ri, rd = ref["acc"].to_arrays()
    np.testing.assert_array_equal(ai[:5], ri[:5]); np.testing.assert_array_equal(ai[6:], ri[6:])
    plain = engine96.nsq_accumulate(seed, 0, n)
    pi, pd = plain.to_arrays()
    np.testing.assert_array_equal(ai, pi)
    np.testing.assert_allclose(ad, pd, rtol=1e-10, atol=1e-7)
    # a larger run on the device only: database == per-sample path, integers exact
    n2 = 300_000
    b = engine96.nsqMain(beta_limit=0.0, max_iterations=n2, samples_per_batch=100_000, seed=2, distinct_states="database")
    a = engine96.nsq_accumulate(2, 0, n2)
    np.testing.assert_array_equal(b.acc.to_arrays()[0], a.to_arrays()[0])
    np.testing.assert_allclose(b.acc.to_arrays()[1], a.to_arrays()[1], rtol=1e-10, atol=1e-6)
    assert b.database_row_count > 65_536                   # grew past the initial capacity


# ---- the states on which the device solver ends "numerically failed" (6.7e-7 of the RTS-96 scenarios) -----------------
@pytest.fixture(scope="module")
def numfail96(case96_):
    with open(os.path.join(GOLDEN, "rts96_numfail_fixture.json")) as f:
        d = json.load(f)
    st = np.zeros((len(d["states"]), case96_.ncomp), dtype=np.uint8)
    for i, x in enumerate(d["states"]):
        st[i, x["failed"]] = 1
    d["matrix"] = st
    return d


def test_oracle96_on_device_numfail_states(oracle96, numfail96):
    """All 51 states of a 1e8-sample run (seed 1) on which the DEVICE's primary order ended NUMFAIL, through the C oracle: it reproduces what
    it returned on the GPU box, its curtailment equals numpy MIPS' and HiGHS' to 1e-6 MW on every state, and the two LU-based
    oracles themselves disagree about the termination status on some of them (heavy-outage states whose Newton systems span
    16 decades: whether the four MIPS tests hold in the same iteration is decided by rounding)."""
    for name, pol in (("emulate", _abi.RELMC_REFERENCE_EMULATE), ("physical", _abi.RELMC_PHYSICAL)):
        r = oracle96.mc_simulation(numfail96["matrix"], pol, nthreads=8)
        differ = 0
        for i, x in enumerate(numfail96["states"]):
            e = x[name]
            assert r["status"][i] == e["c_oracle"]["status"] and r["iters"][i] == e["c_oracle"]["iters"], (name, i)
            assert r["dns"][i] == pytest.approx(e["numpy_mips"]["dns"], abs=1e-6), (name, i)
            if e["highs_dns"] is not None:
                assert r["dns"][i] == pytest.approx(e["highs_dns"], abs=5e-5), (name, i)
            assert e["c_oracle"]["status"] in (0, 2) and e["numpy_mips"]["status"] in (0, 2)
            differ += int(e["c_oracle"]["status"] != e["numpy_mips"]["status"])
        assert 1 <= differ <= 12                      # recorded: 7 of 51


@pytest.mark.gpu
def test_gpu96_numfail_states_vs_oracle(engine96, oracle96, numfail96):
    """The 51 states (of 1e8 samples, seed 1) on which the device's PRIMARY elimination order -- the tuned order the package ships for
    RTS-96, scanned in round 3 with the retries off -- ends "numerically failed".  The curtailment is the optimum on every one of
    them (<= 1e-5 MW from the oracle, numpy MIPS and HiGHS).  Every entry point evaluates such a unit again under a second and, if
    need be, a third static order, and then with the dense pivoted solve."""
    from powersystemsreliabilityassessment_amd import api
    for name, pol in (("emulate", _abi.RELMC_REFERENCE_EMULATE), ("physical", _abi.RELMC_PHYSICAL)):
        before = engine96.retry_stats()
        dns, nodal, info = engine96.mc_simulation(numfail96["matrix"], mpopt=api.mpoption(pol), return_info=True)
        after = engine96.retry_stats()
        N = len(numfail96["states"])
        assert after[0] - before[0] == N and after[1] - before[1] >= N - 1          # all of them went to the further orders
        r = oracle96.mc_simulation(numfail96["matrix"], pol, nthreads=16)
        np.testing.assert_allclose(dns, r["dns"], rtol=0, atol=1e-5)
        assert set(np.unique(info["status"])) <= {0, 2}
        dev_ok, orc_ok = info["status"] == 0, r["status"] == 0
        assert dev_ok.sum() >= N - 1 and (dev_ok & orc_ok).sum() >= orc_ok.sum() - 1
        both = dev_ok & orc_ok
        # heavy-outage states solved under the further orders: the fixture records, per state, the status and iteration count the device's
        # production entry point returned when it was generated (tests/tools/numfail96_device.py) and its distance to the C oracle's count
        # (make_golden.py --numfail96-device): exactly those, state by state -- a static schedule is deterministic
        # ... for the code object the record was taken on.  Another binary (compiler, FMA contraction, scheduling) may move a retried state by an
        # iteration: then the recorded values are a reference, the bound below is the test, and the fixture wants regenerating
        # (tests/tools/numfail96_device.py on the GPU box, then make_golden.py --numfail96-device).
        from powersystemsreliabilityassessment_amd import _lib
        same_binary = numfail96.get("device_retried_from", {}).get("code_object_sha256") == _lib.code_object_sha256()
        gaps, moved = [], 0
        for i, x in enumerate(numfail96["states"]):
            rec = x[name]["device_retried"]
            exact = (int(info["status"][i]), int(info["iters"][i])) == (rec["status"], rec["iters"])
            assert exact or not same_binary, (name, i, info["status"][i], info["iters"][i], rec)
            moved += int(not exact)
            if both[i]:
                gaps.append(int(info["iters"][i]) - int(r["iters"][i]))
                if exact:
                    assert rec["iters_minus_c_oracle"] is not None and gaps[-1] == rec["iters_minus_c_oracle"], (name, i)
        if moved:
            print(f"numfail96 {name}: {moved} of {N} retried states differ from the record of another code object "
                  f"({numfail96.get('device_retried_from', {}).get('code_object_sha256', '?')[:12]}): regenerate the fixture's device record")
            assert moved <= N // 5
        assert max(abs(g) for g in gaps) <= 8 and np.mean([g == 0 for g in gaps]) > 0.75             # what the recorded gaps amount to
        print(f"numfail96 {name}: iteration gaps device - oracle over {len(gaps)} states: {dict((g, gaps.count(g)) for g in sorted(set(gaps)))}")
        for i, x in enumerate(numfail96["states"]):
            e = x[name]
            assert dns[i] == pytest.approx(e["numpy_mips"]["dns"], abs=1e-5)
            if e["highs_dns"] is not None:
                assert dns[i] == pytest.approx(e["highs_dns"] if e["highs_dns"] >= 0.1 else 0.0, abs=5e-5)
        np.testing.assert_allclose(nodal.sum(1)[dns > 0], dns[dns > 0], rtol=0, atol=5e-3)


@pytest.mark.gpu
def test_gpu_second_larger_case_on_one_context_resizes_the_retry_scratch(case96_, oracle96, numfail96):
    """One context, two cases: RTS-24 first, with retries forced (an iteration limit of 7 lists nearly every unit), then RTS-96
    on the SAME engine and the numfail fixture through mc_simulation, whose 51 states all go to the further orders.  The scratch rows of the
    re-evaluation are sized with the loaded case's bus count (24, then 73): the second case's retries must not write past the first's."""
    from powersystemsreliabilityassessment_amd import api
    eng = api.Engine(case24.rts24())
    o = api.mpoption(); o.max_it = 7
    acc = eng.nsq_accumulate(1, 0, 20000, o)
    assert eng.retry_stats()[0] > 4000 and acc.n == 20000
    eng.load_case(case96_)
    assert eng.retry_stats() == (0, 0)
    N = len(numfail96["states"])
    R = 80                                                          # 4 080 listed units: the list of a per-state call holds 4096 + n / 256
    big = np.tile(numfail96["matrix"], (R, 1))
    dns, nodal, info = eng.mc_simulation(big, return_info=True)
    assert eng.retry_stats()[0] == R * N and eng.retry_overflow() == 0
    r = oracle96.mc_simulation(numfail96["matrix"], _abi.RELMC_REFERENCE_EMULATE, nthreads=16)
    np.testing.assert_allclose(dns.reshape(R, N), np.tile(r["dns"], (R, 1)), rtol=0, atol=1e-5)
    np.testing.assert_array_equal(nodal.reshape(R, N, -1), np.tile(nodal[:N], (R, 1, 1)))
    np.testing.assert_allclose(nodal[:N].sum(1)[dns[:N] > 0], dns[:N][dns[:N] > 0], rtol=0, atol=5e-3)
    assert (info["status"][:N] == 0).sum() >= N - 1
    # and back to the small case
    eng.load_case(case24.rts24())
    assert eng.nsq_accumulate(1, 0, 20000).n_nonconverged == 0
    eng.close()


@pytest.mark.gpu
def test_gpu_diagnosis_switches_of_a_context(case96_, numfail96):
    """relmc_debug_set (the test hooks that replaced round 3's environment switches): `no_retry` leaves the states of the numfail fixture with
    their first-attempt status and no unit re-evaluated; `retry_dense_first` sends the listed units straight to the dense pivoted solve;
    `nsq_no_stretch` runs nsqMain one launch per batch with the same checkpoints; an unknown switch is refused."""
    from powersystemsreliabilityassessment_amd import api
    st = numfail96["matrix"]; N = st.shape[0]
    plain = api.Engine(case96_)
    d0, _, i0 = plain.mc_simulation(st, return_info=True)
    assert plain.retry_stats()[0] == N and (i0["status"] == 0).sum() >= N - 1
    off = api.Engine(case96_, debug_switches=("no_retry",))       # set before the case is loaded: the order calibration reads it
    d1, _, i1 = off.mc_simulation(st, return_info=True)
    assert off.retry_stats() == (0, 0) and (i1["status"] == 2).sum() >= N - 5           # the fixture IS the set of states the primary order fails on
    np.testing.assert_allclose(d1, d0, rtol=0, atol=1e-5)                               # the curtailment is the optimum either way
    with pytest.raises(api.RelmcError, match="unknown switch"):
        off.debug_set("no_such_switch")
    off.close()
    plain.debug_set("retry_dense_first")
    u0 = plain.retry_dense_stats()[0]
    d2, _, i2 = plain.mc_simulation(st, return_info=True)
    assert plain.retry_dense_stats()[0] - u0 == N and (i2["status"] == 0).sum() >= N - 3
    np.testing.assert_allclose(d2, d0, rtol=0, atol=1e-5)
    plain.debug_set("retry_dense_first", False)
    a = plain.nsqMain(beta_limit=0.0, max_iterations=6000, samples_per_batch=500, seed=2)
    plain.debug_set("nsq_no_stretch")
    b = plain.nsqMain(beta_limit=0.0, max_iterations=6000, samples_per_batch=500, seed=2)
    assert len(a.beta_history) == len(b.beta_history) == 12 and np.array_equal(a.acc.to_arrays()[0], b.acc.to_arrays()[0])
    np.testing.assert_allclose(a.beta_history, b.beta_history, rtol=1e-9)
    plain.close()


@pytest.mark.gpu
def test_gpu96_dense_last_resort_on_the_numfail_states(engine96, oracle96, numfail96):
    """The 51 RTS-96 states the primary static order ends 'numerically failed' on, through the dense partially pivoted solve alone (the
    third retry level, reached by 2 units in 1e9 samples once the further static orders have had their turn): it converges on at least as
    many as the C oracle's pivoted LU does on the same states, agrees with the oracle's status on nearly all, and the
    curtailment equals the oracle's to 1e-5 MW on every state whatever the status."""
    from powersystemsreliabilityassessment_amd import api
    st = numfail96["matrix"]
    for pol in (_abi.RELMC_REFERENCE_EMULATE, _abi.RELMC_PHYSICAL):
        dns, nodal, info = engine96.mc_simulation_dense(st, api.mpoption(pol))
        r = oracle96.mc_simulation(st, pol, nthreads=16)
        conv_dev, conv_orc = int((info["status"] == 0).sum()), int((r["status"] == 0).sum())
        assert conv_dev >= conv_orc - 2, (conv_dev, conv_orc)
        assert int((info["status"] != r["status"]).sum()) <= 6
        np.testing.assert_allclose(dns, r["dns"], rtol=0, atol=1e-5)
        both = (info["status"] == 0) & (r["status"] == 0)
        assert (np.abs(info["iters"][both] - r["iters"][both]) <= 1).mean() > 0.9


@pytest.mark.gpu
def test_gpu96_retry_in_every_path(engine96, numfail96):
    """The sample of the fixture's first state (index recorded by the scan) through the fused, the distinct-state, the database
    and the per-state paths: each re-evaluates it, and the accumulators agree with one another."""
    from powersystemsreliabilityassessment_amd import api
    idx = int(numfail96["states"][0]["emulate"]["index"])
    lo, n = idx - 1000, 2048
    u0 = engine96.retry_stats()[0]
    a = engine96.nsq_accumulate(1, lo, n)
    u1 = engine96.retry_stats()[0]
    d, _ = engine96.nsq_accumulate_distinct(1, lo, n)
    u2 = engine96.retry_stats()[0]
    engine96.db_reset()
    b, _ = engine96.nsq_db_batch(1, lo, n)
    u3 = engine96.retry_stats()[0]
    st = engine96.mc_sampling(None, n, seed=1, first_index=lo)
    dns, nodal, info = engine96.mc_simulation(st, return_info=True)
    u4 = engine96.retry_stats()[0]
    assert (u1 - u0, u2 - u1, u3 - u2, u4 - u3) == (1, 1, 1, 1)
    assert a.n_nonconverged == d.n_nonconverged == b.n_nonconverged == int((info["status"] == 1).sum() + (info["status"] == 2).sum()) == 0
    ai, ad = a.to_arrays()
    for other in (d, b):
        oi, od = other.to_arrays()
        assert np.array_equal(ai, oi)
        np.testing.assert_allclose(od, ad, rtol=1e-9, atol=1e-6)
    assert ai[5] == info["iters"].sum() and abs(ad[0] - dns.sum()) < 1e-6
    engine96.db_reset()


@pytest.mark.gpu
def test_gpu96_retry_through_the_scaled_load_entry_point(seqeng96, engine96, numfail96):
    """seq_mcsimulation (host buffers, per-state load scale) on the fixture's states: scale 1 must give mc_simulation's
    results (all of them go through the further orders with their scale), and a scale of 0.97 still converges everywhere."""
    from powersystemsreliabilityassessment_amd import api
    st = numfail96["matrix"]
    u0 = engine96.retry_stats()[0]
    d1, n1, i1 = seqeng96.seq_mcsimulation(st, 1.0, return_info=True)
    u1 = engine96.retry_stats()[0]
    d0, n0, i0 = engine96.mc_simulation(st, return_info=True)
    assert u1 - u0 == len(st)
    np.testing.assert_array_equal(d1, d0); np.testing.assert_array_equal(n1, n0)
    np.testing.assert_array_equal(i1["status"], i0["status"]); np.testing.assert_array_equal(i1["iters"], i0["iters"])
    d2, n2, i2 = seqeng96.seq_mcsimulation(st, np.full(len(st), 0.97), return_info=True)
    assert np.all(i2["status"] == 0) and np.all(d2 <= d1 + 1e-9)


@pytest.mark.gpu
def test_gpu96_nonconverged_rate(engine96):
    """2e7 scenarios: the primary order ends 5e-7 of them non-converged (10 expected here), the further orders none
    (0 of the 51 in the first 1e8 samples, scripts/order_soak.py)."""
    u0 = engine96.retry_stats()[0]
    acc = engine96.nsq_accumulate(1, 0, 20_000_000)
    assert acc.n_nonconverged <= 1 and 3 <= engine96.retry_stats()[0] - u0 <= 40


# ---- the sequential track on the wide tile (219 components: 8 mask words per hour) --------------------------------------
@pytest.fixture(scope="module")
def seqeng96(engine96):
    from powersystemsreliabilityassessment_amd import seq
    return seq.SeqEngine(engine96, reliability_data=case96.seqmeantime96())


@pytest.mark.gpu
def test_gpu96_sequential_track(seqeng96, oracle96, engine96):
    """seq_mcsampling / seq_mcsimulation / the fused year loop on RTS-96 (seq_mcsampling.m:35-76, seq_mcsimulation.m:38-72,
    seqMain.m:85-176): chronology bit-exact against the oracle, sampled contingency hours against the oracle's scaled LP, and
    the fused years against the same hours evaluated one by one through the batched entry point."""
    from powersystemsreliabilityassessment_amd import api
    rel, hpy = case96.seqmeantime96(), seqeng96.hours
    st = seqeng96.seq_mcsampling(rel, 99, 120, 2, hpy, seed=4, first_year=7)              # [ncomp x 2*hours]
    ref = oracle96.seq_mcsampling(rel, hpy, 4, 7, 2)                                     # [2*hours x ncomp]
    np.testing.assert_array_equal(st.T, ref)
    assert 0.2 < ref.any(axis=1).mean() < 1.0 and not ref[:, [14, 47, 80]].all(axis=0).any()
    # sampled contingency hours of year 7 against the oracle (scaled loads)
    year = ref[:hpy]
    hours = np.flatnonzero(year.any(axis=1))
    pick = hours[np.linspace(0, hours.size - 1, 120).astype(int)]
    lf = seqeng96.load_factors[pick]
    dns, nodal, info = seqeng96.seq_mcsimulation(year[pick], lf, return_info=True)
    r = oracle96.seq_mcsimulation(year[pick], lf, nthreads=16)
    np.testing.assert_array_equal(info["status"], r["status"])
    np.testing.assert_allclose(dns, r["dns"], rtol=0, atol=1e-5)
    assert np.abs(info["iters"] - r["iters"]).max() <= 1
    # the fused year loop == the same contingency hours through the batched entry point
    ens, dlc, nlc, ncont, acc = seqeng96.seq_years(4, 7, 2)
    for y in range(2):
        yr = ref[y * hpy:(y + 1) * hpy]
        hrs = np.flatnonzero(yr.any(axis=1))
        assert ncont[y] == hrs.size
        d, _ = seqeng96.seq_mcsimulation(yr[hrs], seqeng96.load_factors[hrs])
        prof = np.zeros(hpy); prof[hrs] = d
        assert ens[y] == pytest.approx(prof.sum(), rel=1e-12, abs=1e-9)
        loss = prof > 0.01
        assert dlc[y] == loss.sum() and nlc[y] == int((np.diff(loss.astype(int)) == 1).sum() + int(loss[0]))
    assert acc.n == int(ncont.sum()) and acc.n_fail == int(dlc.sum())
