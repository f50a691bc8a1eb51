"""CPU tests of the host logic and of the C-ABI library (load + exports; no compute without a GPU)."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from powersystemsreliabilityassessment_amd import _abi, _lib, dist as rdist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "relmc.h")).read()
    declared = set(re.findall(r"\b(relmc_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.EXPORTS)
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    L = _lib.load()
    for s in declared:
        assert hasattr(L, s), s
    assert b"gfx950" in L.relmc_version()


def test_struct_layouts_and_defaults():
    L = _lib.load()
    o = _abi.SolverOpts()
    L.relmc_solver_opts_default(C.byref(o))
    d = _abi.default_solver_opts()
    for f, _ in _abi.SolverOpts._fields_:
        assert getattr(o, f) == getattr(d, f), f
    assert (o.max_it, o.feastol, o.xi, o.sigma) == (150, 5e-6, 0.99995, 0.1)
    n = _abi.NsqOpts()
    L.relmc_nsq_opts_default(C.byref(n))
    assert (n.beta_limit, n.max_samples, n.batch, n.hours_per_year) == (0.0017, 100000, 100, 8760.0)   # nsqMain.m:60-62


def test_no_device_fails_loudly():
    """Without a GPU every entry point must fail (no CPU fallback); with one, bad arguments must."""
    L = _lib.load()
    h = C.c_void_p()
    rc = L.relmc_ctx_create(0, C.byref(h))
    if rc != 0:
        assert rc == -2 and not h.value            # RELMC_ERR_NO_DEVICE
        from powersystemsreliabilityassessment_amd import api
        with pytest.raises(api.RelmcError):
            api.Engine()
    else:
        acc = _abi.Acc()
        assert L.relmc_nsq_accumulate(h, 1, 0, 10, None, C.byref(acc)) == -5      # RELMC_ERR_NO_CASE
        L.relmc_ctx_destroy(h)
    assert L.relmc_ctx_create(0, None) == -1


def test_estimators_host_arithmetic(oracle):
    """relmc_nsq_indices / relmc_acc_merge (product, host only) == the oracle's restatement."""
    L = _lib.load()
    a = oracle.nsq_accumulate(3, 0, 20000, 0)
    b = oracle.nsq_accumulate(3, 20000, 15000, 0)
    m = _abi.Acc()
    L.relmc_acc_zero(C.byref(m)); L.relmc_acc_merge(C.byref(m), C.byref(a)); L.relmc_acc_merge(C.byref(m), C.byref(b))
    whole = oracle.nsq_accumulate(3, 0, 35000, 0)
    mi, md = m.to_arrays(); wi, wd = whole.to_arrays()
    assert np.array_equal(mi, wi) and np.allclose(md, wd, rtol=1e-12)
    out = _abi.Indices()
    L.relmc_nsq_indices(C.byref(m), 24, 71, 8760.0, C.byref(out))
    ref = oracle.indices(m)
    for f in ("n", "edns", "lole", "plc", "beta", "eens", "mean_iters"):
        assert getattr(out, f) == getattr(ref, f), f
    assert list(out.nodal_eens) == list(ref.nodal_eens) and list(out.comp_importance) == list(ref.comp_importance)
    d = rdist.indices_from_acc(m, 24, 71)
    assert d["edns"] == pytest.approx(out.edns) and d["beta"] == pytest.approx(out.beta) and d["lole"] == pytest.approx(out.lole)
    # empty accumulator: no NaNs (nsqMain.m:299-301 would divide by zero; SURVEY Appendix E guard)
    z = _abi.Acc(); L.relmc_nsq_indices(C.byref(z), 24, 71, 8760.0, C.byref(out))
    assert out.n == 0 and out.edns == 0


def test_shard_range_partitions_exactly():
    for start, count, world in ((0, 10, 3), (5, 1_000_003, 8), (7, 3, 8), (0, 0, 2)):
        cover = []
        for r in range(world):
            lo, cnt = rdist.shard_range(start, count, r, world)
            cover.extend(range(lo, lo + cnt)) if count < 100 else cover.append((lo, cnt))
        if count < 100:
            assert cover == list(range(start, start + count))
        else:
            assert cover[0][0] == start and sum(c for _, c in cover) == count
            assert all(cover[i][0] + cover[i][1] == cover[i + 1][0] for i in range(world - 1))


def test_acc_array_roundtrip():
    a = _abi.Acc(n=5, n_fail=2, sum_dns=3.5, sum_dns2=9.25)
    a.comp_fail[70] = 4; a.sum_nodal[23] = 1.25
    i, d = a.to_arrays()
    b = _abi.Acc.from_arrays(i, d)
    assert (b.n, b.n_fail, b.sum_dns, b.sum_dns2, b.comp_fail[70], b.sum_nodal[23]) == (5, 2, 3.5, 9.25, 4, 1.25)


_WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch.distributed as dist
from powersystemsreliabilityassessment_amd import case24, dist as rdist
from oracle import coracle
rank, world = int(sys.argv[1]), int(sys.argv[2])
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[3], RANK=str(rank), WORLD_SIZE=str(world))
dist.init_process_group("gloo", rank=rank, world_size=world)
case = case24.rts24(); orc = coracle.Oracle(case)
fn = lambda seed, first, n: orc.nsq_accumulate(seed, first, n, 0, nthreads=2)     # stand-in evaluator (oracle)
idx, total, hist = rdist.nsq_run_distributed(fn, case.nb, case.ncomp, seed=9, beta_limit=0.02, max_samples=60000, batch=7000)
# the same loop with a per-rank running total (what a rank's persistent state database returns): cumulative mode
state = dict(acc=None)
def fn_cum(seed, first, n):
    part = orc.nsq_accumulate(seed, first, n, 0, nthreads=2) if n > 0 else None
    if part is not None:
        state["acc"] = part if state["acc"] is None else rdist.merge(state["acc"], part)
    from powersystemsreliabilityassessment_amd import _abi
    return state["acc"] if state["acc"] is not None else _abi.Acc()
idx2, total2, hist2 = rdist.nsq_run_distributed(fn_cum, case.nb, case.ncomp, seed=9, beta_limit=0.02, max_samples=60000, batch=7000, cumulative=True)
if rank == 0:
    ti, td = total.to_arrays()
    ci, cd = total2.to_arrays()
    assert np.array_equal(ti, ci) and np.allclose(td, cd, rtol=1e-12) and len(hist) == len(hist2)
    np.savez(sys.argv[4], ti=ti, td=td, edns=idx["edns"], beta=idx["beta"], n=idx["n"], ncheck=len(hist))
dist.destroy_process_group()
"""


def test_distributed_driver_gloo_world2(tmp_path, oracle):
    """N > 1 path on CPU: 2 ranks over gloo, oracle as the per-rank evaluator; the merged accumulators
    must equal the single-process run over the same global index range (integers exactly)."""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT))
    out = tmp_path / "rank0.npz"
    port = str(29600 + os.getpid() % 300)
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2", port, str(out)]) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    got = np.load(out)
    n = int(got["n"])
    assert n % 7000 == 0 and 0 < n <= 63000 and int(got["ncheck"]) == n // 7000
    single = oracle.nsq_accumulate(9, 0, n, 0)
    si, sd = single.to_arrays()
    assert np.array_equal(got["ti"], si)
    np.testing.assert_allclose(got["td"], sd, rtol=1e-11, atol=1e-9)
    assert float(got["beta"]) <= 0.02 or n >= 60000


_STRETCH_WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch.distributed as dist
from powersystemsreliabilityassessment_amd import case24, dist as rdist
from oracle import coracle
rank, world = int(sys.argv[1]), int(sys.argv[2])
if world > 1:
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[3], RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
case = case24.rts24(); orc = coracle.Oracle(case)
acc_fn = lambda seed, first, n: orc.nsq_accumulate(seed, first, n, 0, nthreads=4)
dns_fn = lambda seed, first, n: orc.mc_simulation(orc.mc_sampling(seed, first, n), 0, nthreads=4)["dns"]
idx, total, hist, ncoll = rdist.nsq_run_stretches(dns_fn, acc_fn, case.nb, case.ncomp, seed=1, beta_limit=0.03, max_samples=200000, batch=100,
                                                  rank=rank, world=world)
if rank == 0:
    ti, td = total.to_arrays()
    np.savez(sys.argv[4], ti=ti, td=td, hist=np.array(hist), ncoll=ncoll)
if world > 1:
    dist.destroy_process_group()
"""


def test_checkpoint_stretches_gloo_world2(tmp_path, oracle):
    """relmc_nsq_run's checkpoint stretches over N ranks, restated in Python (dist.nsq_run_stretches) with the oracle as the evaluator: two
    gloo ranks at the reference's spacing of 100 samples stop at the ONE-rank run's checkpoint -- the checkpoint the oracle's own batch-by-batch
    database loop stops at --, with its history and accumulators, in one collective per stretch (+ one for the cut)."""
    script = tmp_path / "stretch.py"
    script.write_text(_STRETCH_WORKER.format(root=ROOT))
    o2, o1 = tmp_path / "w2.npz", tmp_path / "w1.npz"
    port = str(29300 + os.getpid() % 90)
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2", port, str(o2)]) for r in range(2)]
    one = subprocess.Popen([sys.executable, str(script), "0", "1", "0", str(o1)])
    for p in procs + [one]:
        assert p.wait(timeout=900) == 0
    g2, g1 = np.load(o2), np.load(o1)
    h2, h1 = g2["hist"], g1["hist"]
    assert h2.shape == h1.shape and np.array_equal(h2[:, 0], h1[:, 0]) and np.array_equal(h2[:, 4], h1[:, 4])      # same checkpoints, same loss counts
    np.testing.assert_allclose(h2[:, 1:4], h1[:, 1:4], rtol=1e-11)
    assert np.array_equal(g2["ti"], g1["ti"])
    np.testing.assert_allclose(g2["td"], g1["td"], rtol=1e-11, atol=1e-9)
    n = int(h1[-1, 0])
    assert h1[-1, 1] <= 0.03 < h1[-2, 1] and n % 100 == 0 and int(g2["ncoll"]) in (2, 3, 4) and int(g1["ncoll"]) == 0
    ref = oracle.nsq_database(1, beta_limit=0.03, max_iterations=200000, samples_per_batch=100, nthreads=8)          # nsqMain.m:208-308 batch by batch
    assert ref["iterations"] == n and len(ref["beta_history"]) == h1.shape[0]
    np.testing.assert_allclose(h1[:, 1], ref["beta_history"], rtol=1e-9)
    np.testing.assert_allclose(h1[:, 2], ref["edns_history"], rtol=1e-11)


_AR_WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch.distributed as dist
from powersystemsreliabilityassessment_amd import _abi, dist as rdist
rank, world = int(sys.argv[1]), int(sys.argv[2])
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[3], RANK=str(rank), WORLD_SIZE=str(world))
dist.init_process_group("gloo", rank=rank, world_size=world)
ints = np.zeros(_abi.Acc.N_INT, dtype=np.int64); dbls = np.zeros(_abi.Acc.N_DBL)
ints[0] = (1 << 61) + 12345 + rank            # far above 2^53: must still come out exact
ints[1] = 7 * (rank + 1); ints[6 + 70] = (1 << 40) + rank
dbls[0] = 0.1 * (rank + 1); dbls[2 + 23] = 1e-3 + rank
out = rdist.allreduce_acc(_abi.Acc.from_arrays(ints, dbls))
if rank == 0:
    oi, od = out.to_arrays()
    np.savez(sys.argv[4], oi=oi, od=od)
dist.destroy_process_group()
"""


def test_allreduce_acc_exact_large_counters(tmp_path):
    """The accumulator all-reduce carries int64 counters exactly (two fp64 words each), whatever their size, and every
    rank enters the one collective unconditionally (no rank-local check in front of it)."""
    script = tmp_path / "ar.py"
    script.write_text(_AR_WORKER.format(root=ROOT))
    out = tmp_path / "ar.npz"
    port = str(29900 + os.getpid() % 90)
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2", port, str(out)]) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    got = np.load(out)
    assert int(got["oi"][0]) == 2 * ((1 << 61) + 12345) + 1 and int(got["oi"][1]) == 21 and int(got["oi"][6 + 70]) == 2 * (1 << 40) + 1
    assert got["od"][0] == pytest.approx(0.3, rel=1e-15) and got["od"][2 + 23] == pytest.approx(1.002, rel=1e-15)


_STUCK_WORKER = r"""
import os, sys, time
sys.path.insert(0, {root!r})
import torch, torch.distributed as dist
from powersystemsreliabilityassessment_amd import dist as rdist
rank, world = int(sys.argv[1]), 2
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[2], RANK=str(rank), WORLD_SIZE=str(world))
dist.init_process_group("gloo", rank=rank, world_size=world)
if rank == 1:
    time.sleep(600)                     # the stuck peer: never enters the collective
with rdist.Watchdog(3.0, "torch.distributed.all_reduce of relmc_acc", rank=rank, world=world, device="cpu"):
    dist.all_reduce(torch.zeros(4, dtype=torch.float64))
print("collective returned", flush=True)
"""


def test_native_comm_init_with_a_deadline_reports_errors_and_stalls():
    """NativeComm.try_init (what bench.py --comm auto uses): the blocking communicator init on a helper thread -- an error comes back as text,
    a call that does not return within the deadline is abandoned and reported as stalled, success hands the communicator over."""
    import time

    class Refused(rdist.NativeComm):
        def __init__(self, engine, rank, world):
            raise RuntimeError("ncclCommInitRank: invalid usage")

    class Stuck(rdist.NativeComm):
        def __init__(self, engine, rank, world):
            time.sleep(30)

    class Fine(rdist.NativeComm):
        def __init__(self, engine, rank, world):
            self.rank, self.world = rank, world
    c, why = Refused.try_init(None, 0, 2, 5.0)
    assert c is None and "invalid usage" in why and "stalled" not in why
    n0 = rdist.abandoned_threads()
    t0 = time.time()
    c, why = Stuck.try_init(None, 0, 2, 0.3)
    assert c is None and "stalled" in why and time.time() - t0 < 5 and rdist.abandoned_threads() == n0 + 1
    c, why = Fine.try_init(None, 1, 2, 5.0)
    assert why == "" and (c.rank, c.world) == (1, 2)


def test_a_communicator_whose_peer_stalled_is_never_finalised():
    """ADVICE r4 (bench.py --comm auto): when a PEER's RCCL bootstrap stalls, this rank's own communicator did come up; dropping the last
    reference to its Engine would run relmc_ctx_destroy -> ncclCommDestroy against a peer stuck inside RCCL, unguarded.  The fallback pins the
    (communicator, engine) pair for the life of the process (dist.keep_forever) and the process leaves through os._exit."""
    import gc
    import weakref

    class Obj:
        pass
    comm, eng = Obj(), Obj()
    wc, we = weakref.ref(comm), weakref.ref(eng)
    n0 = rdist.kept_forever()
    rdist.keep_forever(comm, eng)
    del comm, eng
    gc.collect()
    assert wc() is not None and we() is not None and rdist.kept_forever() == n0 + 1
    src = open(os.path.join(ROOT, "bench.py")).read()
    i = src.index("rdist.keep_forever(comm, eng)")
    assert i < src.index("eng = api.Engine(case, device=local_rank)", i) < src.index("comm = rdist.HostComm(eng, rank, world, device)", i)     # pinned BEFORE the names are rebound
    assert "rdist.abandoned_threads() or rdist.kept_forever()" in src and "os._exit(0)" in src


def test_watchdog_turns_a_stuck_peer_into_a_diagnosis(tmp_path):
    """First contact with N > 1 ranks must not be able to hang: a peer that never enters the collective (here: a gloo rank that sleeps) makes
    the waiting rank print who it is and what it was waiting for and leave with exit code 86 after the guard's limit, instead of sitting in
    the collective until the launcher's own timeout.  A block that finishes in time is left alone."""
    import time
    with rdist.Watchdog(5.0, "nothing"):
        pass
    fired = []
    with rdist.Watchdog(0.2, "a slow block", on_expiry=lambda: fired.append(1)):
        time.sleep(0.6)
    assert fired == [1]
    script = tmp_path / "stuck.py"
    script.write_text(_STUCK_WORKER.format(root=ROOT))
    port = str(29300 + os.getpid() % 90)
    p1 = subprocess.Popen([sys.executable, str(script), "1", port])
    try:
        t0 = time.time()
        p0 = subprocess.run([sys.executable, str(script), "0", port], capture_output=True, text=True, timeout=240)
        assert p0.returncode == rdist.Watchdog.EXIT_CODE, (p0.returncode, p0.stderr[-800:])
        assert "rank 0 of 2" in p0.stderr and "has waited 3 s in torch.distributed.all_reduce of relmc_acc" in p0.stderr and "collective returned" not in p0.stdout
        assert time.time() - t0 < 200
    finally:
        p1.kill(); p1.wait()


_FAKE_RANK = """
import os, sys, time
r, w = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert os.environ["LOCAL_RANK"] == str(r) and os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0
assert "torch" not in sys.modules
print(f"rank {r} of {w} args {sys.argv[1:]}", flush=True)
mode = sys.argv[1]
if mode == "ok":
    sys.exit(0)
if mode == "one_fails":          # rank 1 leaves with the watchdog's code at once; the others would sit for a minute
    if r == 1:
        sys.exit(86)
    time.sleep(60)
"""


def test_bench_gpus_flag_starts_the_ranks_itself(tmp_path):
    """`python bench.py --gpus N` with no launcher around it IS the launcher (the flag was parsed and never read, so the plain
    command measured one GPU and said n_gpus 1).  The parent must not import torch, must hand every rank RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT and the same arguments, relay only rank 0's stdout, and end with the first failing rank's code after
    stopping the rest by pid; a launcher whose WORLD_SIZE disagrees with --gpus is refused before anything is imported."""
    import importlib.util
    import time
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    fake = tmp_path / "fake_rank.py"
    fake.write_text(_FAKE_RANK)
    drv = tmp_path / "drv.py"
    drv.write_text(f"import sys, importlib.util\nspec = importlib.util.spec_from_file_location('bench_mod', {os.path.join(ROOT, 'bench.py')!r})\n"
                   f"m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)\n"
                   f"rc = m.launch_ranks(int(sys.argv[1]), sys.argv[2:], script={str(fake)!r})\nassert 'torch' not in sys.modules\nsys.exit(rc)\n")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    ok = subprocess.run([sys.executable, str(drv), "8", "ok", "--steps", "2"], capture_output=True, text=True, timeout=120, env=env)
    assert ok.returncode == 0, ok.stderr[-800:]
    assert ok.stdout.strip() == "rank 0 of 8 args ['ok', '--steps', '2']"                    # rank 0's stdout only
    assert all(f"rank {r} of 8" in ok.stderr for r in range(1, 8))                           # the other ranks' stdout lands on stderr
    t0 = time.time()
    bad = subprocess.run([sys.executable, str(drv), "4", "one_fails"], capture_output=True, text=True, timeout=120, env=env)
    assert bad.returncode == 86 and time.time() - t0 < 30, (bad.returncode, bad.stderr[-800:])
    assert "rank 1 of 4" in bad.stderr and "left with code 86: stopping the other ranks" in bad.stderr
    # a launcher that disagrees with --gpus: refused with both numbers, before torch is imported (fast)
    t0 = time.time()
    mis = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], capture_output=True, text=True, timeout=60,
                         env=dict(env, RANK="0", WORLD_SIZE="2", LOCAL_RANK="0"))
    assert mis.returncode == 7 and "WORLD_SIZE=2" in mis.stderr and "--gpus says 8" in mis.stderr and mis.stdout == ""
    mis1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"], capture_output=True, text=True, timeout=60,
                          env=dict(env, RANK="0", WORLD_SIZE="2", LOCAL_RANK="0"))
    assert mis1.returncode == 7
    # no GPU here: the plain command starts its two ranks, both refuse loudly, the parent reports the code (never a silent CPU path)
    import torch
    if not torch.cuda.is_available():
        nog = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                             timeout=300, env=env)
        assert nog.returncode != 0 and "needs a GPU" in nog.stderr and "stopping the other ranks" in nog.stderr and '"metric"' not in nog.stdout


def _plain_bench(tmp_path, tag, extra, expect_ok=True, timeout=900):
    """bench.py called the way the driver calls it: no external launcher."""
    import json
    f = tmp_path / f"acc_{tag}.json"
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--dump-acc", str(f)] + extra,
                         capture_output=True, text=True, timeout=timeout, env=env)
    if not expect_ok:
        return out
    assert out.returncode == 0, out.stderr[-3000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(line) == 1, out.stdout[-2000:]
    return json.loads(line[0]), json.load(open(f))


def _same_acc(a, b):
    assert a["ints"] == b["ints"]
    np.testing.assert_allclose([float.fromhex(x) for x in a["dbls"]], [float.fromhex(x) for x in b["dbls"]], rtol=1e-11, atol=1e-9)


@pytest.mark.gpu
def test_plain_bench_command_runs_two_ranks(tmp_path):
    """`python bench.py --gpus 2 ...` with NO external launcher: two ranks (here on the one GPU of the box, host collective), n_gpus 2, the
    communicator itself reports 2 ranks, and the merged accumulators equal the 1-rank run over the same global scenario range.  Asking for
    more GPUs than the node has, without --share-device, is refused with both numbers."""
    j2, a2 = _plain_bench(tmp_path, "p2", ["--gpus", "2", "--share-device", "--comm", "host", "--steps", "2", "--warmup", "1", "--batch", "50000"])
    j1, a1 = _plain_bench(tmp_path, "p1", ["--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "100000", "--no-time-to-cov", "--no-secondary"])
    assert j2["n_gpus"] == 2 and j2["comm"]["nranks_seen"] == 2 and j2["comm"]["backend"] == "host-collective" and len(j2["comm"]["devices"]) == 2
    assert j2["launcher"].startswith("bench.py --gpus N") and j1["launcher"].startswith("none") and j1["n_gpus"] == 1
    assert j2["indices"]["n"] == j1["indices"]["n"] == 200000
    _same_acc(a2, a1)
    assert set(j2["omitted"]["keys"]) >= {"cpu_baseline", "secondary", "sustained"} and "omitted" not in j1
    ttc = j2["time_to_cov_1pct"]
    assert "relmc_nsq_run" in ttc["loop"] and ttc["beta"] < 0.01 and ttc["batch"] == 65536 and ttc["samples"] % 65536 == 0
    db = j2["distinct_state_path"]
    assert db["per_rank_database"] is True and len(db["rows_per_rank"]) == 2 and db["beta"] < 0.0017
    assert 1.0 <= db["redundant_solves_x"] <= 2.0 and max(db["rows_per_rank"]) <= db["rows_single_database"] <= sum(db["rows_per_rank"])
    import torch
    if torch.cuda.device_count() < 3:
        bad = _plain_bench(tmp_path, "p3", ["--gpus", "3", "--steps", "1", "--warmup", "0"], expect_ok=False)
        assert bad.returncode == 6 and "3 ranks on this node (--gpus 3) but it has 1 GPU" in bad.stderr and '"metric"' not in bad.stdout


@pytest.mark.gpu
def test_plain_bench_command_eight_rank_rehearsal(tmp_path):
    """BASELINE configs[2] / [3] / [4] are 8-GPU shapes and no 8-GPU node has ever run them: rehearse the plain command with EIGHT ranks
    sharing the one GPU (host collective).  Weak scaling with the time-to-CoV loop at 32 768 x 8 samples per check, strong scaling over a
    fixed 8e6-sample step, and the sequential workload over 8 x 2 years: eight PCI ids gathered, eight ranks in the communicator's own
    count, accumulators equal to ONE rank over the same global range (integers exactly)."""
    w8, aw8 = _plain_bench(tmp_path, "w8", ["--gpus", "8", "--share-device", "--comm", "host", "--steps", "2", "--warmup", "1", "--batch", "25000"])
    w1, aw1 = _plain_bench(tmp_path, "w1", ["--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "200000", "--no-time-to-cov", "--no-secondary"])
    assert w8["n_gpus"] == 8 and w8["comm"]["nranks_seen"] == 8 and len(w8["comm"]["devices"]) == 8 and len(set(w8["comm"]["devices"])) == 1
    assert len(w8["kernel_ms_per_rank"]) == 8 and min(w8["kernel_ms_per_rank"]) > 0
    assert w8["indices"]["n"] == w1["indices"]["n"] == 400000
    _same_acc(aw8, aw1)
    ttc = w8["time_to_cov_1pct"]
    assert ttc["batch"] == 32768 * 8 and ttc["samples"] % (32768 * 8) == 0 and ttc["beta"] < 0.01 and "relmc_nsq_run" in ttc["loop"]
    # ... and at the reference's own checkpoint spacing of 100 samples (nsqMain.m:60) over the eight ranks: the library's multi-rank loop walks stretches of
    # checkpoints and stops at the ONE-rank run's checkpoint with its history
    from powersystemsreliabilityassessment_amd import api
    e1 = api.Engine()
    r1 = e1.nsqMain(beta_limit=0.01, max_iterations=5_000_000, samples_per_batch=100, seed=1)
    e1.close()
    c100 = ttc["checkpoints_of_100"]
    assert c100["samples"] == r1.current_iteration == 211_200 and c100["converged"] and c100["checkpoints"] == len(r1.beta_history) and c100["n_fail"] == r1.acc.n_fail
    np.testing.assert_allclose(c100["beta_history_head"], r1.beta_history[:3], rtol=1e-12)
    np.testing.assert_allclose(c100["beta_history_tail"], r1.beta_history[-3:], rtol=1e-10)
    assert c100["beta"] == pytest.approx(r1.current_beta, rel=1e-10) and c100["edns_mw"] == pytest.approx(r1.accumulated_edns, rel=1e-12)
    db = w8["distinct_state_path"]
    assert len(db["rows_per_rank"]) == 8 and 1.0 <= db["redundant_solves_x"] <= 8.0
    s8, as8 = _plain_bench(tmp_path, "s8", ["--gpus", "8", "--share-device", "--comm", "host", "--steps", "1", "--warmup", "0", "--scaling", "strong", "--total", "8000000",
                                            "--no-time-to-cov"])
    s1, as1 = _plain_bench(tmp_path, "s1", ["--gpus", "1", "--steps", "1", "--warmup", "0", "--scaling", "strong", "--total", "8000000", "--no-time-to-cov", "--no-secondary"])
    assert s8["scaling"] == "strong" and s8["n_gpus"] == 8 and s8["indices"]["n"] == s1["indices"]["n"] == 8000000
    _same_acc(as8, as1)
    q8, aq8 = _plain_bench(tmp_path, "q8", ["--gpus", "8", "--share-device", "--comm", "host", "--workload", "seq", "--years", "2", "--steps", "1", "--warmup", "0"])
    q1, aq1 = _plain_bench(tmp_path, "q1", ["--gpus", "1", "--workload", "seq", "--years", "16", "--steps", "1", "--warmup", "0"])
    assert q8["n_gpus"] == 8 and q8["comm"]["nranks_seen"] == 8 and q8["indices"]["n"] == q1["indices"]["n"] > 0
    _same_acc(aq8, aq1)
    print(f"8-rank rehearsal on one GPU: weak {w8['value']:.3g}/s, strong {s8['value']:.3g}/s, seq {q8['value']:.3g} hourly OPFs/s; "
          f"database redundancy x{db['redundant_solves_x']:.2f}")


@pytest.mark.gpu
def test_bench_two_ranks_share_device(tmp_path):
    """bench.py's N > 1 path (torch.distributed launcher, one all-reduce per step) with two ranks on the one GPU of the box
    (gloo, --share-device): the merged accumulators of the timed steps equal the single-rank run over the same global
    scenario range, in weak and in strong scaling."""
    import json
    def run(nproc, extra, tag, ttc=False):
        f = tmp_path / f"acc_{tag}.json"
        port = str(29700 + (os.getpid() + nproc + len(tag)) % 200)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
               "--master-port", port, os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), "--steps", "2", "--warmup", "1",
               "--backend", "gloo", "--comm", "torch", "--share-device", "--no-cpu-baseline", "--no-secondary", "--dump-acc", str(f)] + ([] if ttc else ["--no-time-to-cov"]) + extra
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        line = [ln for ln in out.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln][-1]
        return json.loads(line), json.load(open(f))
    j2, a2 = run(2, ["--batch", "50000"], "w2", ttc=True)        # with the multi-rank wall time to CoV < 1 %
    assert j2["time_to_cov_1pct"]["beta"] < 0.01 and j2["time_to_cov_1pct"]["samples"] % 65536 == 0
    j1, a1 = run(1, ["--batch", "100000"], "w1")
    assert j2["n_gpus"] == 2 and j2["scaling"] == "weak" and j2["indices"]["n"] == j1["indices"]["n"] == 200000
    assert a2["ints"] == a1["ints"]
    np.testing.assert_allclose([float.fromhex(x) for x in a2["dbls"]], [float.fromhex(x) for x in a1["dbls"]], rtol=1e-11, atol=1e-9)
    for key in ("frac", "frac_executed", "frac_dense_equiv", "lds_pipe_busy", "lds_conflict_frac", "valu_busy", "waves_per_simd", "traffic", "counters_source"):
        assert key in j1["roofline"], key
    s2, b2 = run(2, ["--scaling", "strong", "--total", "120000"], "s2")
    s1, b1 = run(1, ["--scaling", "strong", "--total", "120000"], "s1")
    assert s2["scaling"] == "strong" and s2["indices"]["n"] == s1["indices"]["n"] == 240000 and b2["ints"] == b1["ints"]


@pytest.mark.gpu
def test_bench_comm_paths_are_self_evidencing(tmp_path):
    """bench.py's JSON line says who carried the collective and how many ranks THE COMMUNICATOR reports (RCCL had never run
    with more than one rank and the line could not show it).  On the one GPU of the box: (1) N = 1 carries the fields (`nranks_seen: 1`);
    (2) two ranks with --comm host: torch's gloo collective registered as the library's transport, the multi-rank nsqMain loop runs below
    the C ABI (relmc_nsq_run) and gives the single-rank accumulators; (3) two ranks with --comm native on ONE device: the 128-byte id
    travels over gloo (no torch nccl group in the process), RCCL refuses the duplicate GPU, and every rank leaves with code 3 and RCCL's
    text on stderr -- no hang, no fallback."""
    import json
    def run(nproc, extra, tag, expect_ok=True):
        f = tmp_path / f"acc_{tag}.json"
        port = str(29400 + (os.getpid() + 7 * nproc + len(tag)) % 200)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
               "--master-port", port, os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), "--steps", "2", "--warmup", "1",
               "--backend", "gloo", "--share-device", "--no-cpu-baseline", "--dump-acc", str(f)] + extra
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        if not expect_ok:
            return out
        assert out.returncode == 0, out.stderr[-2000:]
        line = [ln for ln in out.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln][-1]
        return json.loads(line), json.load(open(f))
    j1, a1 = run(1, ["--batch", "100000", "--no-time-to-cov", "--no-sustained"], "n1")
    assert j1["comm"]["backend"] == "none" and j1["comm"]["nranks_seen"] == 1 and j1["comm"]["allreduce_bytes"] == C.sizeof(_abi.Acc)
    assert len(j1["comm"]["devices"]) == 1 and re.fullmatch(r"[0-9a-fA-F]{4}:[0-9a-fA-F]{2}:[0-9a-fA-F]{2}\.[0-9a-fA-F]", j1["comm"]["devices"][0]), j1["comm"]["devices"]
    # the other BASELINE configs ride along outside the timed region: RTS-96, sequential, HL1 copper sheet
    sec = j1["secondary"]
    assert set(sec) == {"rts96", "seq", "hl1"} and all(sec[k]["value"] > 0 and sec[k]["ms_per_step"] > 0 and sec[k]["kernel_ms_avg"] > 0 for k in sec)
    assert sec["rts96"]["n_nonconverged"] == 0 and 12.0 < sec["rts96"]["mean_ipm_iterations"] < 13.5 and 0.05 < sec["rts96"]["frac_executed"] < 0.3
    # ... and the same workloads behind the zero-curtailment pre-screen (relmc_solver_opts.screen), beside the headline, never as `value`
    scr = j1["screened"]
    assert {"nsq24", "rts96", "seq", "time_to_cov_1pct", "distinct_state_path"} <= set(scr)
    assert 0.88 < scr["nsq24"]["n_screened_frac"] < 0.93 and 0.96 < scr["rts96"]["n_screened_frac"] < 0.995 and scr["seq"]["n_screened_frac"] > 0.98
    assert scr["nsq24"]["value"] > 2.0 * j1["value"] and scr["rts96"]["value"] > 2.0 * sec["rts96"]["value"] and scr["seq"]["value"] > 2.0 * sec["seq"]["value"]
    assert scr["time_to_cov_1pct"]["samples"] == 211_200 and all(scr[k]["n_nonconverged"] == 0 for k in ("nsq24", "rts96", "seq"))
    assert sec["seq"]["n_nonconverged"] == 0 and sec["seq"]["years_per_s"] > 100 and sec["hl1"]["lole_h_per_yr"] == pytest.approx(9.39, rel=0.05)
    assert len(j1["kernel_ms_per_rank"]) == 1 and j1["kernel_ms_per_rank"][0] > 0
    j2, a2 = run(2, ["--batch", "50000", "--comm", "host"], "h2")
    assert len(j2["comm"]["devices"]) == 2 and j2["comm"]["devices"][0] == j2["comm"]["devices"][1] == j1["comm"]["devices"][0]      # --share-device says so
    assert j2["comm"]["backend"] == "host-collective" and j2["comm"]["nranks_seen"] == 2 and j2["comm"]["allreduce_calls"] >= 3
    assert j2["comm"]["allreduce_us_avg"] > 0 and len(j2["kernel_ms_per_rank"]) == 2 and min(j2["kernel_ms_per_rank"]) > 0
    assert a2["ints"] == a1["ints"]
    np.testing.assert_allclose([float.fromhex(x) for x in a2["dbls"]], [float.fromhex(x) for x in a1["dbls"]], rtol=1e-11, atol=1e-9)
    ttc = j2["time_to_cov_1pct"]
    assert "relmc_nsq_run" in ttc["loop"] and ttc["beta"] < 0.01 and ttc["samples"] % 65536 == 0
    # the sequential workload over two ranks: the annual indices travel through the library's communicator (relmc_comm_allreduce_f64 over the
    # registered host collective), the accumulators of 2 x 8 years per step equal one rank's 16
    q2, c2 = run(2, ["--workload", "seq", "--years", "8", "--comm", "host", "--no-time-to-cov"], "q2")
    q1, c1 = run(1, ["--workload", "seq", "--years", "16", "--no-time-to-cov"], "q1")
    assert q2["unit"] == "hourly DC-OPFs/s" and q2["comm"]["nranks_seen"] == 2 and q2["indices"]["n"] == q1["indices"]["n"] > 0 and c2["ints"] == c1["ints"]
    np.testing.assert_allclose([float.fromhex(x) for x in c2["dbls"]], [float.fromhex(x) for x in c1["dbls"]], rtol=1e-11, atol=1e-9)
    bad = run(2, ["--batch", "20000", "--comm", "native", "--comm-timeout", "60", "--no-time-to-cov"], "x2", expect_ok=False)
    if "has waited" in bad.stderr:          # RCCL's bootstrap stalls once in a while on this pool: the library's guard ended the run (exit code 86); once more
        bad = run(2, ["--batch", "20000", "--comm", "native", "--comm-timeout", "60", "--no-time-to-cov"], "x3", expect_ok=False)
    assert bad.returncode != 0
    assert bad.stderr.count("communicator init failed") >= 1 and "ncclCommInitRank" in bad.stderr, bad.stderr[-1500:]
    assert '"metric"' not in bad.stdout
    # the default for N > 1 (--comm auto) tries the same, and when RCCL refuses every rank agrees to fall back to the host collective -- and says so
    f2, fa = run(2, ["--batch", "50000", "--comm-timeout", "60", "--no-time-to-cov"], "f2")
    assert f2["comm"]["backend"] == "host-collective" and "ncclCommInitRank" in f2["comm"]["fallback"] and f2["comm"]["nranks_seen"] == 2
    assert fa["ints"] == a1["ints"]


def test_nsqmain_report_wording():
    """The console report of nsqMain.m:314-317, 325-393 from a result object (host formatting only, no device)."""
    from powersystemsreliabilityassessment_amd import api
    nodal = np.zeros(24); nodal[[5, 14, 2]] = [2.9, 1.13, 1.0]
    imp = np.zeros(71); imp[[22, 23, 32, 11, 33 + 5]] = [0.53, 0.54, 0.30, 0.16, 0.03]
    acc = _abi.Acc(n=3000, n_fail=250)
    r = api.NsqResult(15.1969, 744.33, 0.084969, 0.0145, 3000, nodal, imp, np.array([0.08, 0.05, 0.0145]), np.array([14.0, 15.5, 15.1969]),
                      np.array([700.0, 750.0, 744.33]), np.array([0.08, 0.085, 0.084969]), False, 12.19, 1, 0, 0, 2.5, 0.1, acc=acc,
                      samples_per_batch=1000, beta_limit=0.0017, database_row_count=412, _ng=33)
    rep = r.report().splitlines()
    assert rep[0] == "Iteration   1000: Beta = 0.080000, EDNS = 14.0000 MW, LOLE = 700.0000 hr/yr"       # nsqMain.m:315-316
    assert "Total iterations: 3000" in rep and "Unique states evaluated: 412" in rep and "Convergence achieved: NO" in rep
    assert "EDNS (Expected Demand Not Supplied): 15.1969 MW" in rep and "PLC (Probability of Load Curtailment): 0.084969" in rep
    assert rep[rep.index("Top 5 Buses by EENS (MWh/yr):") + 1] == "  Bus  6: 25404.0000 MWh/yr"          # zero buses are not listed (:355)
    assert sum(1 for ln in rep if ln.startswith("  Bus")) == 3
    k = rep.index("Top 5 Critical Components (Prob. Down given System Failure):")
    assert rep[k + 1:k + 6] == ["  Gen 24: 54.00%", "  Gen 23: 53.00%", "  Gen 33: 30.00%", "  Gen 12: 16.00%", "  Line  6: 3.00%"]
    assert r.top_buses(2) == [(6, pytest.approx(2.9 * 8760)), (15, pytest.approx(1.13 * 8760))]
    quiet = api.NsqResult(0.0, 0.0, 0.0, float("inf"), 100, np.zeros(24), np.zeros(71), np.array([np.inf]), np.zeros(1), np.zeros(1), np.zeros(1),
                          False, 12.0, 0, 0, 0, 0.1, 0.0, acc=_abi.Acc(n=100))
    assert "No failure events recorded to analyze weak points." in quiet.report()


def test_seqmain_report_wording():
    """The console output of seqMain.m:187-197, 206-249 from a result object (host formatting only, no device)."""
    from powersystemsreliabilityassessment_amd import seq
    n = 23
    nodal = np.zeros(24); nodal[[17, 14, 5]] = [499.7, 433.75, 325.15]
    imp = np.zeros(71); imp[[22, 23, 32, 43, 11]] = [0.45, 0.46, 0.30, 0.22, 0.16]
    ens = np.linspace(1000.0, 8000.0, n); cum = np.cumsum(ens) / np.arange(1, n + 1)
    cov = np.linspace(0.5, 0.049, n); cov[0] = 0.0
    r = seq.SeqResult(final_year=n, eens=float(cum[-1]), cov=0.049, lole=14.3309, lolf=2.4651, results_year=dict(ens=ens, dlc=np.ones(n), nlc=np.ones(n)),
                      results_cum=dict(eens=cum, cov=cov), nodal_eens_avg=nodal, comp_importance=imp, total_loss_hours=330, years_evaluated=n, n_lp=1000,
                      n_singular=3, n_infeasible=0, n_nonconverged=0, elapsed_time=0.1, kernel_seconds=0.05, converged=True)
    rep = r.report().splitlines()
    assert rep[0] == "Year   10 | EENS: %.4f MWh/yr | CoV: %.4f" % (cum[9], cov[9]) and rep[1].startswith("Year   20 |")       # seqMain.m:187-190
    assert rep[2] == "Convergence Reached at Year 23!"                                                                        # :195
    assert "EENS (Expected Energy Not Supplied): %.4f MWh/yr" % cum[-1] in rep and "LOLE (Loss of Load Expectation):     14.3309 hr/yr" in rep
    assert "LOLF (Loss of Load Frequency):       2.4651 occ/yr" in rep
    assert rep[rep.index("Top 5 Buses by EENS (MWh/yr):") + 1] == "  Bus 18: 499.7000 MWh/yr" and sum(1 for ln in rep if ln.startswith("  Bus")) == 3
    k = rep.index("Top 5 Critical Components (Prob. Down given System Failure):")
    assert rep[k + 1:k + 6] == ["  Gen 24: 46.00%", "  Gen 23: 45.00%", "  Gen 33: 30.00%", "  Line 11: 22.00%", "  Gen 12: 16.00%"]
    import dataclasses
    quiet = dataclasses.replace(r, total_loss_hours=0, converged=False)
    assert "No failure events recorded to analyze weak points." in quiet.report() and "Convergence Reached" not in quiet.report()


def test_bench_counter_fields_come_from_the_committed_profile():
    """bench.py's pipe utilisations / traffic are read from profiles/<current>/ and say so; every workload has them."""
    import bench
    for wl in ("nsq24", "rts96", "seq"):
        c = bench.counters_from_profile(wl, 1_000_000, 256)
        assert c["counters_source"].startswith("from_profile: profiles/"), wl
        assert 0.3 < c["lds_pipe_busy"] < 0.9 and 0.3 < c["valu_busy"] < 0.9 and 0.1 < c["lds_conflict_frac"] < 0.6
        assert c["waves_per_simd"] == 2.0 and c["mfma_fp64_ops"] == 0.0 and c["traffic"] > 0
    assert bench.dense_flop_per_iter(24) == pytest.approx(40525.67, rel=1e-6)          # order 47


def test_bench_says_when_the_quoted_counters_are_not_about_this_binary(tmp_path):
    """The roofline's pipe counters are copied from a committed profile (a program cannot read its own PMCs): the line must tell when that
    profile was taken on ANOTHER binary or when the kernel no longer takes the profiled time.  The summary records the code-object hash of the
    profiled library (sha256 of its .hip_fatbin) and the minimum traced launch; one flipped hex digit, a drifted kernel time, or a summary
    without a hash each set roofline.counters_stale."""
    import json
    import shutil
    import bench
    h = _lib.code_object_sha256()
    assert re.fullmatch(r"[0-9a-f]{64}", h) and h == _lib.code_object_sha256(_lib.LIB_PATH)
    other = tmp_path / "other.so"                          # the hash covers the device code: one changed byte inside .hip_fatbin changes it
    blob = bytearray(open(_lib.LIB_PATH, "rb").read())
    at = blob.index(b"__CLANG_OFFLOAD_BUNDLE__") if b"__CLANG_OFFLOAD_BUNDLE__" in blob else blob.index(b"CCOB")
    blob[at + 64] ^= 1
    other.write_bytes(bytes(blob))
    assert _lib.code_object_sha256(str(other)) != h
    from oracle import coracle
    with pytest.raises(_lib.RelmcLibraryError):
        _lib.code_object_sha256(coracle.build())                                          # a CPU library carries no device code
    cur = open(os.path.join(ROOT, "profiles", "current.txt")).read().strip()
    pd = tmp_path / "profiles"
    (pd / cur).mkdir(parents=True)
    (pd / "current.txt").write_text(cur)
    summ = json.load(open(os.path.join(ROOT, "profiles", cur, "pmc_summary.json")))
    def counters(ms, **edit):
        d = dict(summ, code_object_sha256=h, kernel_ms_min=17.40, units_per_traced_launch=1_000_000)
        d.update(edit)
        d = {k: v for k, v in d.items() if v is not None}
        json.dump(d, open(pd / cur / "pmc_summary.json", "w"))
        return bench.counters_from_profile("nsq24", 1_000_000, 256, kernel_ms_avg=ms, code_hash=h, profiles_dir=str(pd))
    ok = counters(17.45)
    assert ok["counters_stale"] is False and "counters_stale_why" not in ok and ok["profile_code_object_sha256"] == h and ok["profile_kernel_ms_min"] == 17.40
    flipped = h[:-1] + ("0" if h[-1] != "0" else "1")
    bad = counters(17.45, code_object_sha256=flipped)
    assert bad["counters_stale"] is True and "!= this library's" in bad["counters_stale_why"]
    slow = counters(18.2)
    assert slow["counters_stale"] is True and "more than 3 %" in slow["counters_stale_why"]
    assert counters(17.45, units_per_traced_launch=2_000_000)["counters_stale"] is True      # the minimum is scaled to the launch size
    assert counters(8.70, units_per_traced_launch=2_000_000)["counters_stale"] is False
    nohash = counters(17.45, code_object_sha256=None)
    assert nohash["counters_stale"] is True and "no code-object hash" in nohash["counters_stale_why"]
    assert bench.counters_from_profile("nsq24", 1_000_000, 256, profiles_dir=str(tmp_path / "nothing"))["counters_stale"] is True
    # the committed summaries of the current profile carry the hash of a library (this one, as long as the device code is unchanged)
    for fn in ("pmc_summary.json", "pmc_summary_rts96.json", "pmc_summary_seq.json"):
        d = json.load(open(os.path.join(ROOT, "profiles", cur, fn)))
        assert re.fullmatch(r"[0-9a-f]{64}", d.get("code_object_sha256", "")) and d["kernel_ms_min"] > 0, fn


# ---- exports (nsqMain.m:398-405, seqMain.m:255-262): the files the reference leaves behind, same names and layout ------------------
def _layout():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "export_layout.json")) as f:
        return json.load(f)


def test_nsq_exports_have_the_references_layout(tmp_path):
    """api.NsqResult.write_nodal_csv / save_mat against the layout of the reference's own nodal_results.csv and
    reliability_results.mat (tests/golden/export_layout.json, made from those files by make_golden.py): header, row labels,
    variable names and shapes; the values read back are the ones written (CSV = nodal_eens x 8760, nsqMain.m:399)."""
    from scipy.io import loadmat
    from powersystemsreliabilityassessment_amd import api
    lay = _layout()["nsq"]
    rng = np.random.default_rng(3)
    nodal = rng.uniform(0, 3, 24); nodal[[10, 11, 16, 20, 21, 22, 23]] = 0.0
    r = api.NsqResult(accumulated_edns=14.9, accumulated_lole=735.9, plc=0.084, current_beta=0.0145, current_iteration=100000,
                      nodal_eens=nodal, comp_importance=rng.uniform(0, 0.5, 71), beta_history=rng.uniform(0.01, 0.3, 1000),
                      edns_history=rng.uniform(10, 30, 1000), lole_history=rng.uniform(500, 900, 1000), plc_history=rng.uniform(0.05, 0.1, 1000),
                      converged=False, mean_iters=12.2, n_singular=34, n_infeasible=0, n_nonconverged=0, elapsed_time=1.0, kernel_seconds=0.5)
    csv, mat = str(tmp_path / lay["csv"]), str(tmp_path / lay["mat"])
    r.write_nodal_csv(csv); r.save_mat(mat)
    lines = open(csv).read().splitlines()
    assert lines[0] == lay["csv_header"] and len(lines) - 1 == lay["csv_rows"]
    assert [ln.split(",")[0] for ln in lines[1:]] == lay["csv_first_column"]
    np.testing.assert_allclose([float(ln.split(",")[1]) for ln in lines[1:]], nodal * 8760.0, rtol=1e-14)
    m = loadmat(mat)
    got = {k: list(v.shape) for k, v in m.items() if not k.startswith("__")}
    assert got == lay["mat_variables"]
    assert float(m["accumulated_edns"].ravel()[0]) == 14.9 and float(m["accumulated_lole"].ravel()[0]) == 735.9
    np.testing.assert_array_equal(m["nodal_eens"].ravel(), nodal); np.testing.assert_array_equal(m["comp_importance"].ravel(), r.comp_importance)
    np.testing.assert_array_equal(m["beta_history"].ravel(), r.beta_history); np.testing.assert_array_equal(m["edns_history"].ravel(), r.edns_history)
    assert all(v.dtype == np.float64 for k, v in m.items() if not k.startswith("__"))


def test_seq_exports_have_the_references_layout(tmp_path):
    from scipy.io import loadmat
    from powersystemsreliabilityassessment_amd import seq
    lay = _layout()["seq"]
    rng = np.random.default_rng(4)
    ny = 50
    yr = dict(ens=rng.uniform(0, 9e3, ny), dlc=rng.integers(0, 40, ny).astype(float), nlc=rng.integers(0, 9, ny).astype(float), plc=rng.uniform(0, 5e-3, ny))
    cum = dict(eens=np.cumsum(yr["ens"]) / np.arange(1, ny + 1), cov=rng.uniform(0.04, 0.5, ny))
    r = seq.SeqResult(final_year=ny, eens=float(cum["eens"][-1]), cov=0.049, lole=14.3, lolf=2.4, results_year=yr, results_cum=cum,
                      nodal_eens_avg=rng.uniform(0, 500, 24), comp_importance=rng.uniform(0, 0.5, 71), total_loss_hours=700, years_evaluated=ny,
                      n_lp=340000, n_singular=3, n_infeasible=1, n_nonconverged=0, elapsed_time=1.0, kernel_seconds=0.4)
    csv, mat = str(tmp_path / lay["csv"]), str(tmp_path / lay["mat"])
    r.write_nodal_csv(csv); r.save_mat(mat)
    lines = open(csv).read().splitlines()
    assert lines[0] == lay["csv_header"] and len(lines) - 1 == lay["csv_rows"] and [ln.split(",")[0] for ln in lines[1:]] == lay["csv_first_column"]
    np.testing.assert_allclose([float(ln.split(",")[1]) for ln in lines[1:]], r.nodal_eens_avg, rtol=1e-14)       # MWh/yr already (seqMain.m:218)
    m = loadmat(mat)
    assert {k: list(v.shape) for k, v in m.items() if not k.startswith("__")} == lay["mat_variables"]
    m2 = loadmat(mat, squeeze_me=True, struct_as_record=False)
    np.testing.assert_array_equal(np.asarray(m2["results_year"].ens), yr["ens"]); np.testing.assert_array_equal(np.asarray(m2["results_cum"].cov), cum["cov"])


def test_julia_host_ships_the_same_elimination_orders():
    """Every host runs the same pass program: the tuned primary orders in julia/RelMC.jl are the Python package's."""
    from powersystemsreliabilityassessment_amd import case24, case96
    jl = open(os.path.join(ROOT, "julia", "RelMC.jl")).read()
    for name, want in (("RTS24_ELIM_ORDER", case24.RTS24_ELIM_ORDER), ("RTS96_ELIM_ORDER", case96.RTS96_ELIM_ORDER)):
        m = re.search(r"const %s = Int32\[([^\]]*)\]" % name, jl)
        assert m and [int(v) for v in m.group(1).split(",")] == list(want), name
