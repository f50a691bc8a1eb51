#!/usr/bin/env python3
"""Generates the committed golden fixtures (run in the build container only).

  python tests/golden/make_golden.py [--reference /root/reference]

Outputs (all small JSON, data only):
  nsq_golden.json       the reference's own golden NSQ artifacts
                        (Montecarlo_nsq_single/reliability_results.mat + nodal_results.csv,
                        written by nsqMain.m:398-405), converted with scipy.io.loadmat.
  states_fixture.json   a fixed list of component states with, per state and policy,
                        the scipy/HiGHS LP value (unique optimum -> pins total dns) and the
                        numpy MIPS restatement's dns / nodal / iterations / status
                        (oracle/pyoracle.py).
  nsq_seed1_1e5.json    the estimator outputs (EDNS, LOLE, PLC, beta, nodal EENS, component
                        importance; nsqMain.m:282-301,348-349,366-376) for seed 1, N = 1e5,
                        computed in Python exactly as the reference does it: unique-state
                        database with occurrence counts (nsqMain.m:220-245,269-301).

  seq_golden.json       the reference's golden sequential artifacts
                        (Montecarlo_seq/seq_reliability_results.mat + seq_nodal_results.csv,
                        written by seqMain.m:255-262): annual ens/dlc/nlc of the 1245 simulated
                        years, the cumulative EENS / CoV curves, nodal EENS and importance.
  seq_hours_fixture.json  scaled-load hours (state, load factor) with the scipy/HiGHS LP value
                        of seq_mcsimulation.m's scaled model and the numpy MIPS restatement.

  rts96_numfail_fixture.json RTS-96 states on which the device solver ended NUMFAIL in a 1e8-sample run, with all oracles' results
  rts96_states_fixture.json  RTS-96 (case96.py) states with the HiGHS LP value and the numpy
                        MIPS restatement's dns / iterations / status, both policies.

Nothing here is read at test time from /root/reference; the JSON files are.
"""
from __future__ import annotations

import argparse
import json
import multiprocessing as mp
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from powersystemsreliabilityassessment_amd import case24  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

CASE = case24.rts24()


def _eval(args):
    idx_list, policy = args
    st = np.zeros(CASE.ncomp, dtype=np.uint8)
    st[list(idx_list)] = 1
    m = po.mips_full(CASE, st, policy)
    h = po.lp_highs(CASE, st, policy)
    return dict(dns=float(m["dns"]), nodal=[float(v) for v in m["nodal"]], iters=int(m["iters"]),
                status=int(m["status"]), relaxed=int(m["n_relaxed"]),
                highs_dns=(None if not h["feasible"] else float(h["dns"])))


def golden_from_reference(ref):
    import scipy.io as sio
    m = sio.loadmat(os.path.join(ref, "Montecarlo_nsq_single", "reliability_results.mat"))
    csv = np.loadtxt(os.path.join(ref, "Montecarlo_nsq_single", "nodal_results.csv"),
                     delimiter=",", skiprows=1)
    out = dict(
        source="Montecarlo_nsq_single/reliability_results.mat + nodal_results.csv (nsqMain.m:398-405)",
        n_samples=100000, samples_per_batch=100,
        accumulated_edns=float(m["accumulated_edns"].ravel()[0]),
        accumulated_lole=float(m["accumulated_lole"].ravel()[0]),
        nodal_eens=[float(v) for v in m["nodal_eens"].ravel()],
        comp_importance=[float(v) for v in m["comp_importance"].ravel()],
        beta_history=[float(v) for v in m["beta_history"].ravel()],
        edns_history=[float(v) for v in m["edns_history"].ravel()],
        nodal_results_csv_eens_mwh_yr=[float(v) for v in csv[:, 1]],
    )
    with open(os.path.join(HERE, "nsq_golden.json"), "w") as f:
        json.dump(out, f)
    print("nsq_golden.json: EDNS", out["accumulated_edns"], "LOLE", out["accumulated_lole"])


def export_layout_from_reference(ref):
    """File layouts of the reference's exports (nsqMain.m:398-405, seqMain.m:255-262): CSV header and row count, .mat variable names
    and shapes -- what tests/test_host.py::test_nsq_exports_have_the_references_layout compares the writers of api.py against."""
    import scipy.io as sio
    out = {}
    for key, d, mat, csv in (("nsq", "Montecarlo_nsq_single", "reliability_results.mat", "nodal_results.csv"),
                             ("seq", "Montecarlo_seq", "seq_reliability_results.mat", "seq_nodal_results.csv")):
        with open(os.path.join(ref, d, csv)) as f:
            lines = f.read().splitlines()
        m = sio.loadmat(os.path.join(ref, d, mat))
        out[key] = dict(csv=csv, mat=mat, csv_header=lines[0], csv_rows=len(lines) - 1, csv_first_column=[ln.split(",")[0] for ln in lines[1:]],
                        mat_variables={k: list(v.shape) for k, v in m.items() if not k.startswith("__")})
    with open(os.path.join(HERE, "export_layout.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("export_layout.json:", {k: (v["csv_header"], sorted(v["mat_variables"])) for k, v in out.items()})


def golden_seq_from_reference(ref):
    import scipy.io as sio
    m = sio.loadmat(os.path.join(ref, "Montecarlo_seq", "seq_reliability_results.mat"), squeeze_me=True, struct_as_record=False)
    csv = np.loadtxt(os.path.join(ref, "Montecarlo_seq", "seq_nodal_results.csv"), delimiter=",", skiprows=1)
    cum, yr = m["results_cum"], m["results_year"]
    ny = int(np.asarray(cum.eens).size)
    out = dict(
        source="Montecarlo_seq/seq_reliability_results.mat + seq_nodal_results.csv (seqMain.m:255-262)",
        final_year=ny, hours_per_year=8736, cov_threshold=0.05, curtail_threshold=0.01,
        ens=[float(v) for v in yr.ens[:ny]], dlc=[int(v) for v in yr.dlc[:ny]], nlc=[int(v) for v in yr.nlc[:ny]],
        cum_eens=[float(v) for v in cum.eens], cum_cov=[float(v) for v in cum.cov],
        nodal_eens_avg=[float(v) for v in np.asarray(m["nodal_eens_avg"]).ravel()],
        comp_importance=[float(v) for v in np.asarray(m["comp_importance"]).ravel()],
        nodal_results_csv_eens_mwh_yr=[float(v) for v in csv[:, 1]],
    )
    with open(os.path.join(HERE, "seq_golden.json"), "w") as f:
        json.dump(out, f)
    print("seq_golden.json: years", ny, "EENS", out["cum_eens"][-1], "LOLE", float(np.mean(out["dlc"])), "LOLF", float(np.mean(out["nlc"])))


def _scaled_case(scale):
    """seq_mcsimulation.m:38-42: virtual-generator Pmin (= -load) and Pload times the hourly factor."""
    import dataclasses
    c = dataclasses.replace(CASE)
    for name in ("inj_pmin", "inj_lo"):
        if hasattr(c, name):
            a = np.array(getattr(c, name), dtype=float).copy()
            a[CASE.ng:] *= scale
            setattr(c, name, a)
    c.total_load = CASE.total_load * scale
    return c


def _eval_scaled(args):
    idx_list, scale, policy = args
    c = _scaled_case(scale)
    st = np.zeros(CASE.ncomp, dtype=np.uint8)
    st[list(idx_list)] = 1
    m = po.mips_full(c, st, policy)
    h = po.lp_highs(c, st, policy)
    return dict(dns=float(m["dns"]), nodal=[float(v) for v in m["nodal"]], iters=int(m["iters"]), status=int(m["status"]),
                highs_dns=(None if not h["feasible"] else float(h["dns"])))


def seq_hours_fixture(n_hours):
    """Random (state, load factor) hours: states from the NSQ sampler thinned to the interesting ones
    plus the SPECIAL list, factors spanning the annual curve's range [0.33, 1]."""
    th = case24.thresholds_u32(CASE)
    s = po.mc_sampling(th, 7, 0, 200 * n_hours)
    rng = np.random.default_rng(7)
    lists = [list(map(int, np.flatnonzero(u))) for u in s if u.sum() >= 5][:n_hours] + [sorted(x) for x in SPECIAL]
    scales = [float(x) for x in rng.uniform(0.33, 1.0, len(lists))]
    for i in range(0, len(lists), 9):
        scales[i] = 1.0                                            # some at the annual peak
    with mp.Pool(8) as pool:
        res0 = pool.map(_eval_scaled, [(l, sc, po.REFERENCE_EMULATE) for l, sc in zip(lists, scales)], chunksize=8)
        res1 = pool.map(_eval_scaled, [(l, sc, po.PHYSICAL) for l, sc in zip(lists, scales)], chunksize=8)
    out = dict(description="hour -> oracle results of the scaled-load DC-OPF (seq_mcsimulation.m)", ncomp=CASE.ncomp,
               hours=[dict(failed=l, load_scale=sc, emulate=a, physical=b) for l, sc, a, b in zip(lists, scales, res0, res1)])
    with open(os.path.join(HERE, "seq_hours_fixture.json"), "w") as f:
        json.dump(out, f)
    bad = [x for x in out["hours"] if x["physical"]["highs_dns"] is not None
           and abs(x["physical"]["highs_dns"] - x["physical"]["dns"]) > 1e-5]
    print("seq_hours_fixture.json:", len(lists), "hours; MIPS-vs-HiGHS mismatches:", len(bad),
          "; loss hours", sum(1 for x in out["hours"] if x["physical"]["dns"] > 0.01))


SPECIAL = [
    [],                          # all up
    [22, 32], [23, 32], [22, 23], [21, 22, 23],      # big-unit outages
    [33 + 10],                   # L11 (7-8) out: bus 7 isolated (fact 11)
    [33 + 10, 23], [33 + 10, 8, 9, 10],
    [33 + 30, 33 + 37],          # L31 + L38: bus 22 isolated with 6 x U50, no load
    [33 + 6, 33 + 26],           # L7 + L27: bus 24 isolated, no injection at all
    [33 + 0, 33 + 1, 33 + 2],    # bus 1 isolated (L1,L2,L3)
    [33 + 1, 33 + 5, 33 + 6],    # buses {3, 24} islanded via L2, L6, L7 out?  (3-9 is L6, 3-24 L7, 1-3 L2)
    [33 + 27, 33 + 29, 33 + 30], # bus 17 isolated (L28, L30, L31)
    [33 + 5], [33 + 8], [33 + 13, 33 + 14], [33 + 15, 33 + 16],
    [12, 13, 22, 23, 32],        # heavy generation loss
    list(range(0, 33)),          # every generator out (sync cond included in the mask)
]


def states_fixture(n_sample):
    th = case24.thresholds_u32(CASE)
    s = po.mc_sampling(th, 1, 0, n_sample)
    uniq = np.unique(s, axis=0)
    lists = [list(map(int, np.flatnonzero(u))) for u in uniq]
    seen = {tuple(l) for l in lists}
    for sp in SPECIAL:
        if tuple(sorted(sp)) not in seen:
            lists.append(sorted(sp))
            seen.add(tuple(sorted(sp)))
    with mp.Pool(8) as pool:
        res0 = pool.map(_eval, [(l, po.REFERENCE_EMULATE) for l in lists], chunksize=8)
        res1 = pool.map(_eval, [(l, po.PHYSICAL) for l in lists], chunksize=8)
    out = dict(description="state -> oracle results; failed = 0-based component indices "
                           "(0..32 generators, 33..70 branches)",
               seed=1, n_sample=n_sample, ncomp=CASE.ncomp,
               states=[dict(failed=l, emulate=a, physical=b) for l, a, b in zip(lists, res0, res1)])
    with open(os.path.join(HERE, "states_fixture.json"), "w") as f:
        json.dump(out, f)
    bad = [x for x in out["states"] if x["physical"]["highs_dns"] is not None
           and abs(x["physical"]["highs_dns"] - x["physical"]["dns"]) > 1e-5]
    print("states_fixture.json:", len(lists), "states; MIPS-vs-HiGHS mismatches:", len(bad))


def nsq_fixture(n):
    """nsqMain.m:208-318 in its own database form, batch = 100."""
    th = case24.thresholds_u32(CASE)
    s = po.mc_sampling(th, 1, 0, n)
    uniq, inv, counts = np.unique(s, axis=0, return_inverse=True, return_counts=True)
    lists = [list(map(int, np.flatnonzero(u))) for u in uniq]
    out = dict(seed=1, n=n, hours_per_year=8760.0)
    for name, pol in (("emulate", po.REFERENCE_EMULATE), ("physical", po.PHYSICAL)):
        with mp.Pool(8) as pool:
            res = pool.map(_eval, [(l, pol) for l in lists], chunksize=16)
        dns = np.array([r["dns"] for r in res])
        nodal = np.array([r["nodal"] for r in res])
        flag = (dns > 1e-4).astype(float)                                   # nsqMain.m:270
        edns = float((counts * dns).sum() / n)                              # :286-287
        plc = float((counts * flag).sum() / n)                              # :295-296
        lole = plc * 8760.0                                                 # :290-292
        beta = float(np.sqrt((counts * (dns - edns) ** 2).sum()) / n / edns)  # :299-301
        nodal_eens = (counts[:, None] * nodal).sum(0) / n                    # :348-349
        fw = counts * flag
        imp = (uniq.astype(float).T @ fw) / fw.sum()                        # :366-376
        out[name] = dict(edns=edns, plc=plc, lole=lole, beta=beta,
                         nodal_eens=[float(v) for v in nodal_eens],
                         comp_importance=[float(v) for v in imp],
                         n_fail=int((counts * flag).sum()),
                         n_singular=int(sum(c for c, r in zip(counts, res) if r["status"] == po.ST_SINGULAR)),
                         sum_iters=int(sum(c * r["iters"] for c, r in zip(counts, res))),
                         n_distinct=len(lists))
        print(name, "EDNS", edns, "PLC", plc, "beta", beta)
    with open(os.path.join(HERE, "nsq_seed1_1e5.json"), "w") as f:
        json.dump(out, f)


def _eval96(args):
    from powersystemsreliabilityassessment_amd import case96
    idx_list, policy = args
    c = case96.rts96()
    st = np.zeros(c.ncomp, dtype=np.uint8)
    st[list(idx_list)] = 1
    m = po.mips_full(c, st, policy)
    h = po.lp_highs(c, st, policy)
    return dict(dns=float(m["dns"]), nodal=[float(v) for v in m["nodal"]], iters=int(m["iters"]), status=int(m["status"]),
                relaxed=int(m["n_relaxed"]), highs_dns=(None if not h["feasible"] else float(h["dns"])))


def rts96_fixture(n_sample):
    from powersystemsreliabilityassessment_amd import case96
    c = case96.rts96()
    B = 99                                                     # first branch component
    special = [
        [], [22, 32], [22 + 33, 23 + 33, 32 + 33], [22, 23, 32, 55, 56, 65, 88, 89, 98],      # big units in one / all areas
        [B + 10], [B + 38 + 10], [B + 76 + 10], [B + 10, B + 48],                                # L11 of an area: bus x07 isolated
        [B + 114], [B + 115, B + 116], [B + 114, B + 115, B + 116, B + 117, B + 118],           # ties out: areas separate
        [B + 117, B + 119],                                      # 325-121 and 323-325 out: bus 325 isolated (no load, no unit)
        [B + 114, B + 115, B + 116, B + 117, B + 118, 22, 23, 32],   # area 1 islanded and short of generation
        [B + 30, B + 37], [B + 76 + 30, B + 76 + 37],            # bus x22 isolated with 6 x U50, no load
        [B + 6, B + 26], [B + 0, B + 1, B + 2], [B + 27, B + 29, B + 30],
        list(range(0, 33)),                                      # every unit of area 1 out
        list(range(0, 99)),                                      # every unit out
    ]
    th = case24.thresholds_u32(c)
    s = po.mc_sampling(th, 1, 0, n_sample)
    lists = [list(map(int, np.flatnonzero(u))) for u in np.unique(s, axis=0)]
    heavy = po.mc_sampling(th, 3, 0, 40 * n_sample)
    lists += [list(map(int, np.flatnonzero(u))) for u in heavy if u.sum() >= 10][:60]
    seen = {tuple(l) for l in lists}
    for sp in special:
        if tuple(sorted(sp)) not in seen:
            lists.append(sorted(sp)); seen.add(tuple(sorted(sp)))
    with mp.Pool(8) as pool:
        res0 = pool.map(_eval96, [(l, po.REFERENCE_EMULATE) for l in lists], chunksize=4)
        res1 = pool.map(_eval96, [(l, po.PHYSICAL) for l in lists], chunksize=4)
    out = dict(description="RTS-96 state -> oracle results; failed = 0-based component indices (0..98 generator rows, 99..218 branches)",
               ncomp=c.ncomp, states=[dict(failed=l, emulate=a, physical=b) for l, a, b in zip(lists, res0, res1)])
    with open(os.path.join(HERE, "rts96_states_fixture.json"), "w") as f:
        json.dump(out, f)
    bad = [x for x in out["states"] if x["physical"]["highs_dns"] is not None and abs(x["physical"]["highs_dns"] - x["physical"]["dns"]) > 1e-5
           and x["physical"]["status"] == 0]
    print("rts96_states_fixture.json:", len(lists), "states; MIPS-vs-HiGHS mismatches:", len(bad),
          "; loss states", sum(1 for x in out["states"] if x["physical"]["dns"] > 0),
          "; status histogram", np.bincount([x["emulate"]["status"] for x in out["states"]], minlength=4).tolist())


def rts96_numfail_fixture(scan_json):
    """States of a 1e8-sample RTS-96 run (seed 1) on which the DEVICE solver ended "numerically failed"
    (tests/tools/numfail96.py on the GPU box; its JSON carries the device's and the C oracle's results): here the numpy MIPS
    restatement and the HiGHS LP value are added, so the fixture pins what the oracles say about exactly these states."""
    with open(scan_json) as f:
        scan = json.load(f)
    lists = sorted({tuple(x["failed"]) for x in scan["states"]})
    with mp.Pool(8) as pool:
        res0 = pool.map(_eval96, [(l, po.REFERENCE_EMULATE) for l in lists], chunksize=2)
        res1 = pool.map(_eval96, [(l, po.PHYSICAL) for l in lists], chunksize=2)
    by = {(x["policy"], tuple(x["failed"])): x for x in scan["states"]}
    out = dict(description="RTS-96 states (seed 1, first 1e8 samples) where the device solver's status was NUMFAIL; per policy: numpy MIPS "
                           "(status/iters/dns), HiGHS LP value, C oracle and device results as scanned on the GPU box by tests/tools/numfail96.py",
               n_scanned=scan["n_scanned"], seed=scan["seed"], states=[])
    for l, a, b in zip(lists, res0, res1):
        ent = dict(failed=list(l))
        for name, r in (("emulate", a), ("physical", b)):
            sc = by.get((name, l))
            ent[name] = dict(numpy_mips=dict(status=r["status"], iters=r["iters"], dns=r["dns"]), highs_dns=r["highs_dns"],
                             c_oracle=sc["c_oracle"] if sc else None, device=sc["gpu"] if sc else None, index=sc["index"] if sc else None)
        out["states"].append(ent)
    with open(os.path.join(HERE, "rts96_numfail_fixture.json"), "w") as f:
        json.dump(out, f)
    agree = sum(1 for e in out["states"] for n in ("emulate", "physical") if e[n]["c_oracle"] and e[n]["c_oracle"]["status"] == e[n]["numpy_mips"]["status"])
    print("rts96_numfail_fixture.json:", len(lists), "states; C oracle == numpy MIPS status on", agree, "of", 2 * len(lists))


def rts96_numfail_device(device_json):
    """Adds what the device's PRODUCTION entry point returns for the numfail states (tests/tools/numfail96_device.py on the GPU box: status and
    iteration count after the further elimination orders / the dense solve) to rts96_numfail_fixture.json as `device_retried`, together with the
    distance of that count to the C oracle's: the GPU parity test asserts both per state."""
    path = os.path.join(HERE, "rts96_numfail_fixture.json")
    with open(path) as f:
        fx = json.load(f)
    with open(device_json) as f:
        dev = json.load(f)
    assert dev["n_states"] == len(fx["states"])
    fx["device_retried_from"] = dict(code_object_sha256=dev["code_object_sha256"], version=dev["version"],
                                     how="tests/tools/numfail96_device.py on an MI355X, merged by make_golden.py --numfail96-device")
    for name in ("emulate", "physical"):
        for i, e in enumerate(fx["states"]):
            it, stt = dev[name]["iters"][i], dev[name]["status"][i]
            co = e[name]["c_oracle"]
            e[name]["device_retried"] = dict(status=stt, iters=it, dns=dev[name]["dns"][i],
                                             iters_minus_c_oracle=(it - co["iters"]) if (co and co["status"] == 0 and stt == 0) else None)
    with open(path, "w") as f:
        json.dump(fx, f)
    gaps = [e[n]["device_retried"]["iters_minus_c_oracle"] for e in fx["states"] for n in ("emulate", "physical")]
    gaps = [g for g in gaps if g is not None]
    print("rts96_numfail_fixture.json: device_retried recorded for", len(fx["states"]), "states x 2 policies; iteration gaps to the C oracle:",
          {g: gaps.count(g) for g in sorted(set(gaps))})


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--n-states-sample", type=int, default=3000)
    ap.add_argument("--n-nsq", type=int, default=100000)
    ap.add_argument("--only-seq", action="store_true")
    ap.add_argument("--only-rts96", action="store_true")
    ap.add_argument("--only-layout", action="store_true", help="export_layout.json alone (file layouts of the reference's exports)")
    ap.add_argument("--numfail96", default="", help="scan JSON of tests/tools/numfail96.py -> rts96_numfail_fixture.json")
    ap.add_argument("--numfail96-device", default="", help="JSON of tests/tools/numfail96_device.py -> `device_retried` entries of rts96_numfail_fixture.json")
    a = ap.parse_args()
    if a.numfail96_device:
        rts96_numfail_device(a.numfail96_device)
        sys.exit(0)
    if a.numfail96:
        rts96_numfail_fixture(a.numfail96)
        sys.exit(0)
    if a.only_rts96:
        rts96_fixture(240)
        sys.exit(0)
    export_layout_from_reference(a.reference)
    if a.only_layout:
        sys.exit(0)
    golden_seq_from_reference(a.reference)
    seq_hours_fixture(160)
    if a.only_seq:
        sys.exit(0)
    golden_from_reference(a.reference)
    states_fixture(a.n_states_sample)
    nsq_fixture(a.n_nsq)
    rts96_fixture(240)
