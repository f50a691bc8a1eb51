"""Multi-GPU driver: scenarios shard by global index, one all-reduce per convergence check.

SURVEY.md §8e: the reference's only parallelism is the `parfor` over sampled states
(nsqMain.m:257-263); scenarios are i.i.d., so every super-batch [done, done+batch) of the global
scenario index range is split contiguously across the ranks, each rank runs the fused
sample -> evaluate -> reduce kernel on its slice, and the additive accumulators (relmc_acc,
~390 words, counters carried as exact fp64) are summed with ONE all-reduce (RCCL over xGMI when the process group is "nccl").
Because the sampler is keyed by (seed, global index), the merged accumulators do not depend on
the number of ranks (integers exactly, fp64 sums up to summation order).
"""
from __future__ import annotations

import math

import numpy as np

from . import _abi


def shard_range(start: int, count: int, rank: int, world: int):
    """Contiguous slice of [start, start+count) owned by `rank`."""
    lo = start + (count * rank) // world
    hi = start + (count * (rank + 1)) // world
    return lo, hi - lo


def allreduce_acc(acc: _abi.Acc, device=None) -> _abi.Acc:
    """Sum relmc_acc over the default process group (no-op without one)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return acc
    ints, dbls = acc.to_arrays()
    on_gpu = dist.get_backend() == "nccl"
    dev = (device if device is not None else torch.device("cuda", torch.cuda.current_device())) if on_gpu else None
    # ONE collective, unconditionally (no rank-local check may keep a rank out of it): every int64 counter rides along as
    # two fp64 words (low / high 32 bits), each exact in fp64 also after the sum over up to 2^20 ranks
    u = ints.astype(np.uint64)
    lo = (u & np.uint64(0xffffffff)).astype(np.float64)
    hi = (u >> np.uint64(32)).astype(np.float64)
    buf = torch.from_numpy(np.concatenate([lo, hi, dbls]))
    if on_gpu:
        buf = buf.to(dev)
    dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    out = buf.cpu().numpy()
    n = ints.size
    tot = [int(round(h)) * (1 << 32) + int(round(l)) for l, h in zip(out[:n], out[n:2 * n])]
    if max(tot, default=0) >= (1 << 63):
        raise OverflowError("relmc_acc counter exceeds int64 after the all-reduce")       # same on every rank: raised together
    return _abi.Acc.from_arrays(np.array(tot, dtype=np.int64), out[2 * n:])


def _broadcast_bytes(payload, rank: int, world: int, key: str = "relmc_comm_id"):
    """128 bytes from rank 0 to everybody through whatever rendezvous the host already has: the default torch.distributed group if
    there is one (any backend; gloo is enough), else a TCPStore on MASTER_ADDR / MASTER_PORT + 1.  No collective on the GPUs."""
    import os
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        box = [payload]
        dist.broadcast_object_list(box, src=0)
        return box[0]
    from datetime import timedelta
    store = dist.TCPStore(os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ.get("MASTER_PORT", "29500")) + 1, world, rank == 0,
                          timeout=timedelta(seconds=120))
    if rank == 0:
        store.set(key, payload)
    return bytes(store.get(key))


_ABANDONED = []          # helper threads left inside a stalled RCCL call (NativeComm.try_init)


def abandoned_threads() -> int:
    return len(_ABANDONED)


_KEPT = []               # (communicator, engine) pairs that must never be finalised: a peer is (or may be) stuck inside RCCL on that communicator


def keep_forever(*objs) -> None:
    """Pin objects for the life of the process: dropping the last reference to an Engine runs relmc_ctx_destroy -> ncclCommDestroy, which
    can hang when a peer of that communicator stalled.  A process that called this leaves through os._exit (no finalisers)."""
    _KEPT.append(objs)


def kept_forever() -> int:
    return len(_KEPT)


class NativeComm:
    """The library's own RCCL communicator (relmc_comm_*, include/relmc.h): what a Julia / C host would use.  Only the 128-byte
    unique id is exchanged through the host's rendezvous (`_broadcast_bytes`); the process needs no torch `nccl` group beside it
    (bench.py --comm native initialises torch.distributed with gloo for exactly that reason: ONE RCCL user per process).
    Raises api.RelmcError with RCCL's message if the communicator cannot be built (e.g. two ranks on one GPU: "invalid usage")."""

    kind = "rccl-native"

    def __init__(self, engine, rank: int, world: int):
        import ctypes as C
        self.eng, self.rank, self.world = engine, rank, world
        uid = (C.c_uint8 * 128)()
        if rank == 0:
            engine._check(engine.L.relmc_comm_unique_id(uid), "relmc_comm_unique_id")
        uid = (C.c_uint8 * 128).from_buffer_copy(_broadcast_bytes(bytes(uid), rank, world))
        engine._check(engine.L.relmc_comm_init(engine._h, world, rank, uid), "relmc_comm_init")

    @classmethod
    def try_init(cls, engine, rank: int, world: int, seconds: float):
        """The same with a deadline, for hosts that have a second transport to fall back to: (communicator, "") or (None, why).  The blocking
        relmc_comm_init runs on a helper thread; RCCL's error comes back as text, and if the call has not returned after `seconds` (RCCL's
        bootstrap has been seen to stall) the thread is ABANDONED -- it cannot be cancelled, the process never calls RCCL again -- and the
        caller is told "stalled" (`abandoned_threads()` > 0 then: leave the process with os._exit when done).  seconds <= 0: no deadline."""
        import threading
        box = {}

        def work():
            try:
                box["comm"] = cls(engine, rank, world)
            except Exception as e:                     # RelmcError with RCCL's message, or whatever the rendezvous raised
                box["err"] = f"{type(e).__name__}: {e}"
        th = threading.Thread(target=work, daemon=True)
        th.start()
        th.join(seconds if seconds and seconds > 0 else None)
        if th.is_alive():
            _ABANDONED.append(th)
            return None, f"relmc_comm_init (ncclCommInitRank) had not returned after {seconds:.0f} s: stalled, thread abandoned"
        return box.get("comm"), box.get("err", "")

    def allreduce_acc(self, acc: _abi.Acc) -> _abi.Acc:
        import ctypes as C
        out = _abi.Acc.from_buffer_copy(bytes(acc))
        self.eng._check(self.eng.L.relmc_comm_allreduce_acc(self.eng._h, C.byref(out)), "relmc_comm_allreduce_acc")
        return out

    def allreduce_f64(self, buf: np.ndarray) -> np.ndarray:
        """relmc_comm_allreduce_f64: sum over the ranks of a vector of doubles (same result on every rank)."""
        out = np.ascontiguousarray(buf, dtype=np.float64).copy()
        self.eng._check(self.eng.L.relmc_comm_allreduce_f64(self.eng._h, out.ctypes.data_as(_abi.c_double_p), int(out.size)), "relmc_comm_allreduce_f64")
        return out

    def allgather_rows(self, mine: np.ndarray, counts) -> np.ndarray:
        """Rows of every rank in rank order (ranks may own different numbers of rows): each rank fills its own rows of a zeroed matrix
        and the matrix is summed over the ranks -- x + 0 + ... + 0 is exact, so this IS an all-gather (what relmc_seq_run does inside)."""
        mine = np.ascontiguousarray(mine, dtype=np.float64)
        cols = mine.shape[1]
        full = np.zeros((int(sum(counts)), cols))
        lo = int(sum(counts[:self.rank]))
        full[lo:lo + mine.shape[0]] = mine
        return self.allreduce_f64(full.ravel()).reshape(-1, cols)

    def info(self) -> dict:
        return comm_info(self.eng)

    def close(self):
        self.eng.L.relmc_comm_destroy(self.eng._h)


class HostComm(NativeComm):
    """The host's own transport registered with the library (relmc_comm_set_host_allreduce): here torch.distributed's default group
    (gloo on CPU tensors, or nccl = RCCL).  The multi-rank loop itself stays below the C ABI (relmc_nsq_run)."""

    kind = "host-collective"

    def __init__(self, engine, rank: int, world: int, device=None, vector: bool = True):
        """vector: also register the vector transport (relmc_comm_set_host_allreduce_f64): the library's vector all-reduces are then one
        callback each instead of one per 130 doubles through the relmc_acc callback."""
        import ctypes as C
        self.eng, self.rank, self.world = engine, rank, world

        def _cb(_user, acc_p):
            try:
                red = allreduce_acc(_abi.Acc.from_buffer_copy(bytes(acc_p.contents)), device)
                C.memmove(acc_p, C.byref(red), C.sizeof(_abi.Acc))
                return 0
            except Exception:      # never let an exception cross the C frame
                import traceback
                traceback.print_exc()
                return 1
        self._cb = _abi.ALLREDUCE_FN(_cb)          # kept alive as long as the communicator
        engine._check(engine.L.relmc_comm_set_host_allreduce(engine._h, world, rank, self._cb, None), "relmc_comm_set_host_allreduce")

        def _cbv(_user, buf_p, count):              # vector transport: one torch all-reduce per relmc_comm_allreduce_f64 call
            try:
                import torch
                import torch.distributed as tdist
                a = np.ctypeslib.as_array(buf_p, shape=(int(count),))
                t = torch.from_numpy(a.copy())
                if tdist.get_backend() == "nccl":
                    t = t.to(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
                tdist.all_reduce(t, op=tdist.ReduceOp.SUM)
                a[:] = t.cpu().numpy()
                return 0
            except Exception:
                import traceback
                traceback.print_exc()
                return 1
        if vector:
            self._cbv = _abi.ALLREDUCE_F64_FN(_cbv)
            engine._check(engine.L.relmc_comm_set_host_allreduce_f64(engine._h, self._cbv, None), "relmc_comm_set_host_allreduce_f64")


class Watchdog:
    """Wall-clock guard of one blocking step of the collective path on the Python side (process-group init, the first collective of a
    transport the library does not own).  `with Watchdog(seconds, what, ...)`: if the block has not finished after `seconds`, the rank says who
    it is and what it was waiting for on stderr and leaves the process with exit code 86 -- a peer that never arrives turns into a diagnosis,
    not into the launcher's half-hour timeout.  Never a re-exec: a fresh start is the launcher's business.  seconds <= 0: no guard.
    (Collectives that go through the library's own communicator are guarded below the C ABI: relmc_comm_set_timeout.)"""

    EXIT_CODE = 86

    def __init__(self, seconds: float, what: str, *, rank: int = 0, world: int = 1, device=None, on_expiry=None):
        self.seconds, self.what, self.rank, self.world, self.device = float(seconds), what, rank, world, device
        self._on_expiry = on_expiry
        self._done = None
        self._thread = None

    def _run(self):
        import os
        import sys
        if self._done.wait(self.seconds):
            return
        print(f"relmc watchdog: rank {self.rank} of {self.world} (pid {os.getpid()}, device {self.device}) has waited {self.seconds:.0f} s in {self.what}: "
              f"a peer never arrived (wrong rank count, a rank that died or took another path, the rendezvous address, or the fabric).  "
              f"Leaving with exit code {self.EXIT_CODE}.", file=sys.stderr, flush=True)
        if self._on_expiry is not None:
            self._on_expiry()
        else:
            os._exit(self.EXIT_CODE)

    def __enter__(self):
        import threading
        if self.seconds > 0:
            self._done = threading.Event()
            self._thread = threading.Thread(target=self._run, daemon=True)
            self._thread.start()
        return self

    def __exit__(self, *exc):
        if self._thread is not None:
            self._done.set()
            self._thread.join()
        return False


def comm_info(engine) -> dict:
    """relmc_comm_info: what the context's communicator itself reports."""
    import ctypes as C
    kind, n, r, calls, sec = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int64(), C.c_double()
    engine._check(engine.L.relmc_comm_info(engine._h, C.byref(kind), C.byref(n), C.byref(r), C.byref(calls), C.byref(sec)), "relmc_comm_info")
    return dict(kind={0: "none", 1: "rccl", 2: "host"}[kind.value], nranks_seen=n.value, rank=r.value, allreduce_calls=calls.value,
                allreduce_seconds=sec.value)


def merge(a: _abi.Acc, b: _abi.Acc) -> _abi.Acc:
    """a + b, field by field (relmc_acc_merge: host arithmetic of the library, no device); the arguments are left alone."""
    import ctypes as C
    from . import _lib
    out = _abi.Acc.from_buffer_copy(a)
    _lib.load().relmc_acc_merge(C.byref(out), C.byref(b))
    return out


def indices_from_acc(acc: _abi.Acc, nb: int, ncomp: int, hours_per_year: float = 8760.0) -> dict:
    """nsqMain.m:286-301,348-349,366-376 from merged sums (host arithmetic only)."""
    n = float(acc.n)
    edns = acc.sum_dns / n
    plc = acc.n_fail / n
    ss = max(acc.sum_dns2 - n * edns * edns, 0.0)
    beta = float(np.sqrt(ss) / n / edns) if edns > 0 else float("inf")
    return dict(n=int(acc.n), edns=edns, plc=plc, lole=plc * hours_per_year, eens=edns * hours_per_year,
                beta=beta, mean_iters=acc.sum_iters / n,
                nodal_eens=np.array(acc.sum_nodal[:nb]) / n,
                comp_importance=(np.array(acc.comp_fail[:ncomp], dtype=np.float64) / acc.n_fail
                                 if acc.n_fail else np.zeros(ncomp)))


def nsq_run_distributed(accumulate_fn, nb: int, ncomp: int, *, seed: int = 1, beta_limit: float = 0.0017,
                        max_samples: int = 100000, batch: int = 100, hours_per_year: float = 8760.0,
                        rank: int | None = None, world: int | None = None, device=None, allreduce=None, cumulative: bool = False,
                        engine=None, mpopt=None, distinct_states=False):
    """The nsqMain loop (nsqMain.m:208-318) over `world` ranks.

    engine = an api.Engine that holds a communicator (NativeComm / HostComm): the call goes to relmc_nsq_run, which IS the multi-rank
    loop (accumulate_fn is not used).  Without an engine the loop below runs in Python around accumulate_fn: the restatement the CPU
    tests check the sharding arithmetic with (gloo, no GPU).

    accumulate_fn(seed, first_index, n) -> _abi.Acc evaluates a scenario range on THIS rank
    (Engine.nsq_accumulate in production).  Returns (indices dict, merged Acc, history list).
    cumulative=True: accumulate_fn returns the accumulators of EVERYTHING this rank has evaluated so far (the rank's own
    persistent state database: lambda s, lo, n: engine.nsq_db_batch(s, lo, n)[0]); the all-reduced result then IS the
    running total instead of an increment.
    """
    import torch.distributed as dist
    if engine is not None:
        # production: the loop is the library's (relmc_nsq_run shards every batch over the ranks of the engine's communicator and
        # all-reduces once per batch); this function only hands the options over
        r = engine.nsqMain(beta_limit=beta_limit, max_iterations=max_samples, samples_per_batch=batch, seed=seed, mpopt=mpopt,
                           distinct_states=distinct_states)
        idx = indices_from_acc(r.acc, nb, ncomp, hours_per_year)
        done = [min((k + 1) * batch, r.current_iteration) for k in range(len(r.beta_history))]
        return idx, r.acc, list(zip(done, r.beta_history, r.edns_history, r.lole_history, r.plc_history))
    if rank is None or world is None:
        if dist.is_available() and dist.is_initialized():
            rank, world = dist.get_rank(), dist.get_world_size()
        else:
            rank, world = 0, 1
    total = _abi.Acc()
    done, beta, hist = 0, float("inf"), []
    idx = None
    while beta > beta_limit and done < max_samples:
        m = min(batch, max_samples - done)
        lo, cnt = shard_range(done, m, rank, world)
        part = accumulate_fn(seed, lo, cnt) if (cnt > 0 or cumulative) else _abi.Acc()
        part = allreduce(part) if allreduce is not None else allreduce_acc(part, device)
        total = part if cumulative else merge(total, part)
        done += m
        idx = indices_from_acc(total, nb, ncomp, hours_per_year)
        beta = idx["beta"]
        hist.append((done, beta, idx["edns"], idx["lole"], idx["plc"]))
    return idx, total, hist


def nsq_run_stretches(sample_dns_fn, accumulate_fn, nb: int, ncomp: int, *, seed: int = 1, beta_limit: float = 0.0017, max_samples: int = 100000,
                      batch: int = 100, hours_per_year: float = 8760.0, rank: int | None = None, world: int | None = None, round_samples: int = 8192):
    """Python restatement of relmc_nsq_run's CHECKPOINT STRETCHES over `world` ranks (csrc/relmc_simulate.hip): what the CPU gloo
    test checks the stretch arithmetic with (which checkpoints two ranks share, the packed all-reduce, the cut-and-retake rule).

    sample_dns_fn(seed, first, n) -> dns of every sample of the range in sampling order; accumulate_fn(seed, first, n) -> _abi.Acc.
    Every rank evaluates its contiguous slice of a stretch, folds it into per-checkpoint partial (sum dns, sum dns^2, losses) triples, ONE all-reduce
    of [3 x checkpoints | accumulators as doubles] per stretch; all ranks walk the checkpoints and stop at the same one; a cut stretch is taken again
    over its used part.  Returns (indices dict, merged Acc, history list, number of collectives)."""
    import torch
    import torch.distributed as dist
    if rank is None or world is None:
        rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)
    per = (1 << 18) // batch * batch
    n_coll = 0

    def shared(lo0, ln, with_trip):
        nonlocal n_coll
        lo, cnt = shard_range(lo0, ln, rank, world)
        ncp = -(-ln // batch) if with_trip else 0
        box = np.zeros(3 * ncp + _abi.Acc.N_INT + _abi.Acc.N_DBL)
        part = accumulate_fn(seed, lo, cnt) if cnt > 0 else _abi.Acc()
        if with_trip and cnt > 0:
            d = np.asarray(sample_dns_fn(seed, lo, cnt), dtype=np.float64)
            k = (lo - lo0 + np.arange(cnt)) // batch
            np.add.at(box, 3 * k, d); np.add.at(box, 3 * k + 1, d * d); np.add.at(box, 3 * k + 2, (d > 1e-4).astype(np.float64))
        ai, ad = part.to_arrays()
        box[3 * ncp:3 * ncp + ai.size] = ai.astype(np.float64); box[3 * ncp + ai.size:] = ad
        if world > 1:
            t = torch.from_numpy(box); dist.all_reduce(t, op=dist.ReduceOp.SUM); n_coll += 1
        ints = np.rint(box[3 * ncp:3 * ncp + ai.size]).astype(np.int64)
        return box[:3 * ncp].reshape(-1, 3), _abi.Acc.from_arrays(ints, box[3 * ncp + ai.size:])

    total, done, beta, hist = _abi.Acc(), 0, float("inf"), []
    first = max(25600 // batch * batch, batch); least = max(1600 // batch * batch, batch)
    while beta > beta_limit and done < max_samples:
        ln = done // batch * batch if done > first else first
        final = False
        if done > 0 and beta_limit > 0 and beta < 1e6 and beta > beta_limit:
            need = done * (beta / beta_limit) ** 2
            target = 0.9 * need if done < 0.85 * need else 1.03 * need
            l = math.ceil((target - done) / batch) * batch
            ln = least if l < least else (per if l > per else int(l))
            final = not (done < 0.85 * need)
        ln = min(ln, per)
        if not final:
            snapped = (ln // round_samples) * round_samples // batch * batch
            if ln >= 2 * round_samples and snapped >= least:
                ln = snapped
        m = min(max_samples - done, ln)
        trip, part = shared(done, m, True)
        run_n, run_f, run_s, run_s2 = int(total.n), int(total.n_fail), float(total.sum_dns), float(total.sum_dns2)
        used = 0
        for k in range(trip.shape[0]):
            b = min(batch, m - used)
            run_n += b; run_f += int(round(trip[k, 2])); run_s += trip[k, 0]; run_s2 += trip[k, 1]; used += b
            e = run_s / run_n; ss = max(run_s2 - run_n * e * e, 0.0)
            beta = math.sqrt(ss) / run_n / e if e > 0 else float("inf")
            hist.append((done + used, beta, e, run_f / run_n * hours_per_year, run_f / run_n))
            if beta <= beta_limit:
                break
        if used < m:
            _, part = shared(done, used, False)
        total = merge(total, part)
        done += used
        idx = indices_from_acc(total, nb, ncomp, hours_per_year)
        beta = idx["beta"]
        hist[-1] = (done, beta, idx["edns"], idx["lole"], idx["plc"])
    return indices_from_acc(total, nb, ncomp, hours_per_year), total, hist, n_coll


def allgather_years(arr: np.ndarray, counts, device=None) -> np.ndarray:
    """Concatenate per-rank [n_r, 3] annual (ens, dlc, nlc) arrays in rank order (ranks may own n or n+1 years)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return arr
    m = max(counts)
    pad = np.zeros((m, 3)); pad[:arr.shape[0]] = arr
    t = torch.from_numpy(pad)
    if dist.get_backend() == "nccl":
        t = t.to(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return np.concatenate([o.cpu().numpy()[:c] for o, c in zip(out, counts)], axis=0)


def seq_run_distributed(years_fn, *, seed: int = 1, cov_threshold: float = 0.05, max_sim_years: int = 4000,
                        batch_years: int = 64, rank: int | None = None, world: int | None = None, device=None):
    """The seqMain loop (seqMain.m:85-199) over `world` ranks: every super-batch of simulated years
    [done, done + batch_years) is split contiguously across the ranks (years are independent streams keyed by
    (seed, global year)), the annual (ens, dlc, nlc) triples are all-gathered in year order so that every rank
    walks the same CoV curve and stops at the same year, and the post-processing accumulators are all-reduced.

    years_fn(seed, first_year, n_years) -> (ens[n], dlc[n], nlc[n], Acc) evaluates years on THIS rank
    (SeqEngine.seq_years in production).  Returns dict(final_year, eens, cov, lole, lolf, years[n,3], cum_eens,
    cum_cov, acc) — acc covers exactly the years up to the stopping year, as seqMain.m:146-159.
    """
    import torch.distributed as dist
    if rank is None or world is None:
        if dist.is_available() and dist.is_initialized():
            rank, world = dist.get_rank(), dist.get_world_size()
        else:
            rank, world = 0, 1
    total = _abi.Acc()
    years = np.zeros((0, 3)); cum_eens, cum_cov = [], []
    done, stop = 0, False
    while done < max_sim_years and not stop:
        m = min(batch_years, max_sim_years - done)
        shards = [shard_range(done, m, r, world) for r in range(world)]
        lo, cnt = shards[rank]
        if cnt > 0:
            e, d, n_, acc = years_fn(seed, lo, cnt)
            mine = np.column_stack([e, d, n_])
        else:
            mine, acc = np.zeros((0, 3)), _abi.Acc()
        batch = allgather_years(mine, [c for _, c in shards], device)
        used = m
        for k in range(m):
            years = np.vstack([years, batch[k:k + 1]])
            y = years.shape[0]
            mean = float(years[:, 0].mean())
            with np.errstate(invalid="ignore", divide="ignore"):                     # no curtailment yet: 0 / 0 = NaN, as seqMain.m:184
                cov = float(np.float64(years[:, 0].std(ddof=1)) / (mean * np.sqrt(y))) if y > 1 else 0.0
            cum_eens.append(mean); cum_cov.append(cov)
            if y > 1 and 0 < cov < cov_threshold:
                stop, used = True, k + 1
                break
        if used < m:        # stopping year inside this batch: accumulate exactly the years up to it
            hi = done + used
            cnt2 = max(0, min(lo + cnt, hi) - lo)
            acc = years_fn(seed, lo, cnt2)[3] if cnt2 > 0 else _abi.Acc()
        total = merge(total, allreduce_acc(acc, device))
        done += used
    return dict(final_year=years.shape[0], eens=cum_eens[-1], cov=cum_cov[-1], lole=float(years[:, 1].mean()),
                lolf=float(years[:, 2].mean()), years=years, cum_eens=np.array(cum_eens), cum_cov=np.array(cum_cov), acc=total)
