#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path: Monte Carlo DC-OPF scenarios/sec, IEEE RTS-24.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by the driver as  python -m torch.distributed.run --nproc-per-node N ... bench.py)

A "step" is one pass of the fused hot path (sample -> DC-OPF interior point -> accumulate) over one
batch of `--batch` (default 1e6 = BASELINE.json configs[1]) synthetic scenarios PER GPU, followed
by the convergence check's all-reduce of the index accumulators when N > 1 (weak scaling: the
per-GPU batch is fixed).  Scenarios are generated in-kernel by the counter-based sampler, so there
is no input to stage: the timed region starts with everything it needs resident in HBM (the case
tables, ~5 KB).  Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6         # MI355X fp64 vector = matrix peak: 256 CU x 4 SIMD x 16 FMA/clk x 2 x 2.4 GHz
FLOP_PER_ITER = 40.5e3          # algorithmic flops of one reduced (order 47) Newton step


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=1_000_000, help="scenarios per GPU per step")
    ap.add_argument("--workload", choices=["nsq24", "rts96", "seq"], default="nsq24",
                    help="nsq24 = BASELINE configs[1] (the headline, default); rts96 = configs[4] shape; seq = configs[3] shape")
    ap.add_argument("--years", type=int, default=125, help="seq workload: simulated years per GPU per step")
    ap.add_argument("--backend", default="nccl", help="process-group backend (nccl = RCCL; gloo only to exercise the N > 1 path on a 1-GPU box)")
    ap.add_argument("--share-device", action="store_true", help="testing only: every rank uses GPU 0")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--policy", choices=["emulate", "physical"], default="emulate")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-time-to-cov", action="store_true", help="skip the nsqMain run (profiling: only identical 1e6 launches)")
    ap.add_argument("--cpu-sample", type=int, default=0, help="scenarios of the CPU baseline sample (0 = auto)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from powersystemsreliabilityassessment_amd import api, case24, dist as rdist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    if args.workload != "nsq24":
        return secondary_workload(args, world, rank, local_rank, device)
    case = case24.rts24()
    eng = api.Engine(case, device=local_rank)
    policy = api.REFERENCE_EMULATE if args.policy == "emulate" else api.PHYSICAL
    opts = api.mpoption(policy)
    B = args.batch

    def step(k):
        # global scenario index space: step k, rank r owns [ (k*world + r)*B, +B )
        acc = eng.nsq_accumulate(args.seed, (k * world + rank) * B, B, opts)
        ms = eng.last_kernel_ms()
        acc = rdist.allreduce_acc(acc, device)       # the convergence check's single all-reduce
        return acc, ms

    def sync():
        if world > 1:
            dist.barrier(device_ids=[local_rank]) if args.backend == "nccl" else dist.barrier()
        torch.cuda.synchronize()

    for k in range(args.warmup):
        step(k)
    sync()
    t0 = time.perf_counter()
    total = None
    kernel_ms = []
    for k in range(args.steps):
        acc, ms = step(args.warmup + k)
        kernel_ms.append(ms)
        total = acc if total is None else rdist.merge(total, acc)
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        n_total = int(total.n)
        idx = rdist.indices_from_acc(total, case.nb, case.ncomp)
        value = n_total / elapsed
        avg_kernel_s = sum(kernel_ms) / len(kernel_ms) * 1e-3
        flop_per_scen = idx["mean_iters"] * FLOP_PER_ITER
        achieved = B * flop_per_scen / avg_kernel_s / 1e12          # per GPU, dominant kernel
        out = {
            "metric": "Monte Carlo DC-OPF scenarios/sec (RTS-24 HL2 non-sequential)",
            "value": value, "unit": "scenarios/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "HL2 non-sequential MCS, IEEE RTS-24 DC-OPF load shedding, "
                                   f"{B:d} samples per GPU per step (BASELINE configs[1])",
                       "scenarios_per_step_per_gpu": B, "policy": args.policy, "seed": args.seed,
                       "parallelism": f"scenario-index sharding x{world}, 1 all-reduce of relmc_acc per step"},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP64_PEAK_TFLOPS, "traffic": hbm_traffic_from_profile(B),
                         "kernel": "relmc_eval_kernel<0>", "kernel_ms_avg": avg_kernel_s * 1e3,
                         "algorithmic_flop_per_scenario": flop_per_scen, "mean_ipm_iterations": idx["mean_iters"],
                         "executed_flop_per_iteration_sparse_schedule": sparse_flop_per_iteration(eng, case)},
            "indices": {"n": n_total, "edns_mw": idx["edns"], "lole_h_per_yr": idx["lole"], "plc": idx["plc"],
                        "beta": idx["beta"], "n_singular": int(total.n_singular),
                        "n_nonconverged": int(total.n_nonconverged)},
        }
        # wall-time to EENS CoV < 1 % (second half of BASELINE.json's metric), single GPU loop of nsqMain
        if world == 1 and not args.no_time_to_cov:
            t1 = time.perf_counter()
            r = eng.nsqMain(beta_limit=0.01, max_iterations=5_000_000, samples_per_batch=100_000,
                            seed=args.seed, mpopt=opts)
            out["time_to_cov_1pct"] = {"seconds": time.perf_counter() - t1, "samples": r.current_iteration,
                                       "beta": r.current_beta, "edns_mw": r.accumulated_edns,
                                       "batch": 100_000, "converged": r.converged}
        if world == 1 and not args.no_time_to_cov:
            # the reference's own speed trick (nsqMain.m:220-245), reported beside the headline, never as `value`:
            # distinct states of a batch solved once and weighted by multiplicity (sampling + device sort + evaluation)
            eng.nsq_accumulate_distinct(args.seed, 0, B, opts)
            t1 = time.perf_counter()
            acc_d, nd = eng.nsq_accumulate_distinct(args.seed, 7 * B, B, opts)
            dt = time.perf_counter() - t1
            t1 = time.perf_counter()
            r = eng.nsqMain(beta_limit=0.01, max_iterations=5_000_000, samples_per_batch=100_000, seed=args.seed, mpopt=opts, distinct_states=True)
            out["distinct_state_path"] = {"samples_per_s": B / dt, "batch": B, "distinct_states": nd, "ms": dt * 1e3,
                                          "time_to_cov_1pct_seconds": time.perf_counter() - t1, "samples": r.current_iteration, "beta": r.current_beta}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(case, policy, args.seed, args.cpu_sample)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def sparse_flop_per_iteration(eng, case):
    """Floating-point operations one Newton iteration of the shipped sparse solver executes per scenario: the static
    schedule's task counts (43 flop per 2x2 block update, 29 per right-hand-side update, 21 per pivot inversion, 14 per
    back-substitution task) plus the per-element work on lines, injections and buses (about 60 / 70 / 30 flop each)."""
    import ctypes as C
    out = (C.c_int32 * 9)()
    eng.L.relmc_debug_schedule(eng._h, out)
    noff, ntask = out[3], out[8]
    nblock = ntask - case.nb - 2 * noff
    return 43.0 * nblock + 29.0 * noff + 21.0 * case.nb + 14.0 * noff + 60.0 * case.nl + 70.0 * case.ninj + 30.0 * case.nb


def secondary_workload(args, world, rank, local_rank, device):
    """The two other GPU configurations of BASELINE.json, same timing contract, one JSON line:
       rts96: non-sequential MCS on the 73-bus RTS-96 (one scenario per wavefront tile), --batch scenarios per GPU per step;
       seq:   sequential MCS on RTS-24, --years simulated years per GPU per step (8736 hourly states each, only the
              contingency hours are evaluated, seqMain.m:97), annual indices all-gathered per step."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from powersystemsreliabilityassessment_amd import api, case24, case96, dist as rdist, seq as rseq

    def sync():
        if world > 1:
            dist.barrier(device_ids=[local_rank]) if args.backend == "nccl" else dist.barrier()
        torch.cuda.synchronize()

    if args.workload == "rts96":
        case = case96.rts96()
        eng = api.Engine(case, device=local_rank)
        B = args.batch
        opts = api.mpoption(api.REFERENCE_EMULATE)

        def step(k):
            acc = eng.nsq_accumulate(args.seed, (k * world + rank) * B, B, opts)
            ms = eng.last_kernel_ms()
            return rdist.allreduce_acc(acc, device), ms, B
        unit, metric = "scenarios/s", "Monte Carlo DC-OPF scenarios/sec (RTS-96 HL2 non-sequential)"
        workload = f"HL2 non-sequential MCS, IEEE RTS-96 (73 buses, 120 branches, 99 generator rows) DC-OPF load shedding, {B} samples per GPU per step (BASELINE configs[4] shape)"
    else:
        case = case24.rts24()
        eng = api.Engine(case, device=local_rank)
        sq = rseq.SeqEngine(eng)
        Y = args.years

        def step(k):
            e, d, n_, ncont, acc = sq.seq_years(args.seed, (k * world + rank) * Y, Y)
            ms = eng.last_kernel_ms()
            rdist.allgather_years(np.column_stack([e, d, n_]), [Y] * world, device)
            return rdist.allreduce_acc(acc, device), ms, int(acc.n)
        unit, metric = "hourly DC-OPFs/s", "Monte Carlo hourly DC-OPF evaluations/sec (RTS-24 HL2 sequential)"
        workload = f"HL2 sequential MCS, RTS-24, {Y} simulated years x 8736 h per GPU per step, contingency hours only (BASELINE configs[3] shape)"

    for k in range(args.warmup):
        step(k)
    sync()
    t0 = time.perf_counter()
    total, kernel_ms, units_rank = None, [], 0
    for k in range(args.steps):
        acc, ms, u = step(args.warmup + k)
        kernel_ms.append(ms); units_rank += u
        total = acc if total is None else rdist.merge(total, acc)
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        n_total = int(total.n)
        mean_iters = total.sum_iters / n_total
        fl = sparse_flop_per_iteration(eng, case)
        avg_kernel_s = sum(kernel_ms) / len(kernel_ms) * 1e-3
        achieved = (units_rank / args.steps) * mean_iters * fl / avg_kernel_s / 1e12
        out = {"metric": metric, "value": n_total / elapsed, "unit": unit, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f64", "data": "synthetic",
               "config": {"workload": workload, "seed": args.seed, "parallelism": f"index / year sharding x{world}, accumulators all-reduced per step"},
               "roofline": {"bound": "mfma", "achieved": achieved, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP64_PEAK_TFLOPS,
                            "traffic": None, "kernel": "relmc_eval_kernel", "kernel_ms_avg": avg_kernel_s * 1e3,
                            "flop_model": "operations of the sparse block LDL' schedule + per-element work (bench.sparse_flop_per_iteration); "
                                          "SURVEY 8d's dense count would exceed the peak on this workload",
                            "flop_per_iteration": fl, "mean_ipm_iterations": mean_iters},
               "indices": {"n": n_total, "edns_mw": total.sum_dns / n_total, "loss_fraction": total.n_fail / n_total,
                           "n_singular": int(total.n_singular), "n_nonconverged": int(total.n_nonconverged)}}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def hbm_traffic_from_profile(batch):
    """HBM bytes per launch of the dominant kernel, from the committed rocprofv3 PMC passes of the profile named in
    profiles/current.txt (pmc_summary.json: FETCH_SIZE and WRITE_SIZE collected in separate runs of this command, gfx950
    correction applied); scaled to this batch.  None if no profile is committed."""
    try:
        with open(os.path.join(ROOT, "profiles", "current.txt")) as fh:
            name = fh.read().strip()
        with open(os.path.join(ROOT, "profiles", name, "pmc_summary.json")) as fh:
            t = json.load(fh).get("hbm_traffic")
        return t["bytes_per_scenario"] * batch if t else None
    except (OSError, ValueError, KeyError):
        return None


def cpu_baseline(case, policy, seed, n_sample):
    """The CPU restatement (oracle/relmc_oracle.c: MATPOWER formulation, dense LU KKT solve) timed on
    the host cores of this box on a bounded sample of the same scenario stream.  Reported baseline
    only — never the thing measured as `value`."""
    from oracle import coracle
    orc = coracle.Oracle(case)
    cores = min(orc.max_threads(), len(os.sched_getaffinity(0)))
    quota = None
    try:                                       # container CPU quota (cgroup v2): the box shows 256 CPUs but grants fewer
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = max(1, int(round(int(q) / int(per))))
            cores = min(cores, quota)
    except (OSError, ValueError):
        pass
    t0 = time.perf_counter()
    orc.nsq_accumulate(seed, 0, 200 * cores, policy, nthreads=cores, memo=False)     # calibration
    rate = 200 * cores / (time.perf_counter() - t0)
    n = n_sample or int(max(2000, min(2_000_000, rate * 15.0)))                      # ~15 s of CPU work
    t0 = time.perf_counter()
    acc = orc.nsq_accumulate(seed, 0, n, policy, nthreads=cores, memo=False)
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    n1 = 1500
    orc.nsq_accumulate(seed, 0, n1, policy, nthreads=1, memo=False)                  # single-core figure as well
    dt1 = time.perf_counter() - t1
    return {"value": n / dt, "unit": "scenarios/s", "cores": cores, "kind": "port", "single_core_value": n1 / dt1, "cpu_quota": quota,
            "sample": f"first {n} scenarios of the same seed, every scenario solved (no state memo), "
                      f"{dt:.1f} s on {cores} OpenMP threads (host CPUs visible {os.cpu_count()}, cgroup quota {quota})",
            "edns_mw": acc.sum_dns / acc.n}


if __name__ == "__main__":
    main()
