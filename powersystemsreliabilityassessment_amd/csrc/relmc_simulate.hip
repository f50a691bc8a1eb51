// relmc_simulate.hip — mc_simulation over host or device buffers (mc_simulation.m:1, batched like the parfor of nsqMain.m:257-263), the fused
// sample -> evaluate -> reduce pass (relmc_nsq_accumulate) and the nsqMain loop itself, single- and multi-rank (relmc_nsq_run, nsqMain.m:208-318).
#include <chrono>
#include <cmath>
#include <cstring>
#include <thread>

#include "relmc_ctx.h"

namespace relmc_host {

// ---- host-buffer evaluation: states in pageable host memory -> dns / nodal / status / iterations in host memory ---------
// What a MATLAB / Julia / Python caller of mc_simulation hands over.  The range is cut into chunks of kPipeChunk states that
// run through a double-buffered pipeline on three streams (H2D, kernel, D2H) with pinned staging buffers; the host copies
// chunk k in and chunk k-2 out while the GPU works on chunk k-1.  Device and staging buffers are allocated once per context.
constexpr int64_t kPipeChunk = 131072;

void pipe_free(relmc_ctx* ctx)
{
    auto& P = ctx->pipe;
    for (int b = 0; b < 2; ++b) {
        for (void* p : {(void*)P.d_st[b], (void*)P.d_sc[b], (void*)P.d_dns[b], (void*)P.d_nod[b], (void*)P.d_stat[b], (void*)P.d_it[b]}) if (p) (void)hipFree(p);
        for (void* p : {(void*)P.h_st[b], (void*)P.h_sc[b], (void*)P.h_dns[b], (void*)P.h_nod[b], (void*)P.h_stat[b], (void*)P.h_it[b]}) if (p) (void)hipHostFree(p);
        for (hipEvent_t e : {P.e_up[b], P.e_ks[b], P.e_ke[b], P.e_down[b]}) if (e) (void)hipEventDestroy(e);
    }
    if (P.up) (void)hipStreamDestroy(P.up);
    if (P.down) (void)hipStreamDestroy(P.down);
    P = relmc_ctx::HostPipe();
}

int pipe_ensure(relmc_ctx* ctx)
{
    auto& P = ctx->pipe;
    if (P.ready && P.ncomp == ctx->ncomp && P.nb == ctx->nb) return RELMC_OK;
    pipe_free(ctx);
    const size_t c = (size_t)kPipeChunk, nc = (size_t)ctx->ncomp, nb = (size_t)ctx->nb;
    bool ok = hipStreamCreateWithFlags(&P.up, hipStreamNonBlocking) == hipSuccess && hipStreamCreateWithFlags(&P.down, hipStreamNonBlocking) == hipSuccess;
    for (int b = 0; b < 2 && ok; ++b) {
        ok = hipEventCreate(&P.e_up[b]) == hipSuccess && hipEventCreate(&P.e_ks[b]) == hipSuccess && hipEventCreate(&P.e_ke[b]) == hipSuccess &&
             hipEventCreate(&P.e_down[b]) == hipSuccess &&
             hipMalloc(&P.d_st[b], c * nc) == hipSuccess && hipMalloc(&P.d_sc[b], c * 8) == hipSuccess && hipMalloc(&P.d_dns[b], c * 8) == hipSuccess &&
             hipMalloc(&P.d_nod[b], c * nb * 8) == hipSuccess && hipMalloc(&P.d_stat[b], c * 4) == hipSuccess && hipMalloc(&P.d_it[b], c * 4) == hipSuccess &&
             hipHostMalloc(&P.h_st[b], c * nc, hipHostMallocDefault) == hipSuccess && hipHostMalloc(&P.h_sc[b], c * 8, hipHostMallocDefault) == hipSuccess &&
             hipHostMalloc(&P.h_dns[b], c * 8, hipHostMallocDefault) == hipSuccess && hipHostMalloc(&P.h_nod[b], c * nb * 8, hipHostMallocDefault) == hipSuccess &&
             hipHostMalloc(&P.h_stat[b], c * 4, hipHostMallocDefault) == hipSuccess && hipHostMalloc(&P.h_it[b], c * 4, hipHostMallocDefault) == hipSuccess;
    }
    if (!ok) { pipe_free(ctx); return fail(ctx, RELMC_ERR_HIP, "host-buffer pipeline: allocation failed"); }
    P.ready = true; P.ncomp = ctx->ncomp; P.nb = ctx->nb;
    return RELMC_OK;
}

// memcpy spread over a few threads: one core moves ~8 GB/s, the nodal output of 1e6 states is 200 MB
void par_memcpy(void* dst, const void* src, size_t bytes)
{
    const size_t kMin = (size_t)4 << 20;
    int nt = bytes / kMin > 4 ? 4 : (int)(bytes / kMin);
    if (nt <= 1) { std::memcpy(dst, src, bytes); return; }
    std::vector<std::thread> th;
    const size_t per = ((bytes / nt) + 63) & ~(size_t)63;
    for (int t = 1; t < nt; ++t) {
        const size_t off = per * t, len = t == nt - 1 ? bytes - off : per;
        th.emplace_back([=]() { std::memcpy((char*)dst + off, (const char*)src + off, len); });
    }
    std::memcpy(dst, src, per);
    for (auto& t : th) t.join();
}

int pipe_run(relmc_ctx* ctx, const uint8_t* states, const double* load_scale, int64_t n, const relmc_solver_opts& o, double fail_threshold,
             double* dns, double* nodal, int32_t* status, int32_t* iters)
{
    int rc = pipe_ensure(ctx);
    if (rc) return rc;
    auto& P = ctx->pipe;
    const size_t nc = (size_t)ctx->ncomp, nb = (size_t)ctx->nb;
    const int64_t nchunk = (n + kPipeChunk - 1) / kPipeChunk;
    double kernel_ms = 0.0;
    auto drain = [&](int64_t k) -> int {                      // chunk k's results: wait for its D2H, copy out of the staging buffers
        const int b = (int)(k & 1);
        const int64_t lo = k * kPipeChunk, m = (n - lo) < kPipeChunk ? (n - lo) : kPipeChunk;
        HIP_TRY(ctx, hipEventSynchronize(P.e_down[b]));
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, P.e_ks[b], P.e_ke[b]) == hipSuccess) kernel_ms += ms;
        std::memcpy(dns + lo, P.h_dns[b], sizeof(double) * (size_t)m);
        if (nodal) par_memcpy(nodal + (size_t)lo * nb, P.h_nod[b], sizeof(double) * (size_t)m * nb);
        if (status) std::memcpy(status + lo, P.h_stat[b], sizeof(int32_t) * (size_t)m);
        if (iters) std::memcpy(iters + lo, P.h_it[b], sizeof(int32_t) * (size_t)m);
        return RELMC_OK;
    };
    for (int64_t k = 0; k < nchunk; ++k) {
        const int b = (int)(k & 1);
        const int64_t lo = k * kPipeChunk, m = (n - lo) < kPipeChunk ? (n - lo) : kPipeChunk;
        if (k >= 2) { rc = drain(k - 2); if (rc) return rc; }     // frees slot b (device buffers and staging)
        par_memcpy(P.h_st[b], states + (size_t)lo * nc, (size_t)m * nc);
        if (load_scale) std::memcpy(P.h_sc[b], load_scale + lo, sizeof(double) * (size_t)m);
        HIP_TRY(ctx, hipMemcpyAsync(P.d_st[b], P.h_st[b], (size_t)m * nc, hipMemcpyHostToDevice, P.up));
        if (load_scale) HIP_TRY(ctx, hipMemcpyAsync(P.d_sc[b], P.h_sc[b], sizeof(double) * (size_t)m, hipMemcpyHostToDevice, P.up));
        HIP_TRY(ctx, hipEventRecord(P.e_up[b], P.up));
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, P.e_up[b], 0));
        EvalArgs a = make_args(o);
        a.fail_threshold = fail_threshold;
        a.n = m; a.states = P.d_st[b]; a.load_scale = load_scale ? P.d_sc[b] : nullptr;
        a.dns = P.d_dns[b]; a.nodal = nodal ? P.d_nod[b] : nullptr; a.status = status ? P.d_stat[b] : nullptr; a.iters = iters ? P.d_it[b] : nullptr;
        int rows = 0;
        rc = fail_arm(ctx, a, lo, k == 0, n);
        if (rc) return rc;
        rc = launch_eval(ctx, 1, a, &rows, P.e_ks[b], P.e_ke[b]);
        if (rc) return rc;
        HIP_TRY(ctx, hipStreamWaitEvent(P.down, P.e_ke[b], 0));
        HIP_TRY(ctx, hipMemcpyAsync(P.h_dns[b], P.d_dns[b], sizeof(double) * (size_t)m, hipMemcpyDeviceToHost, P.down));
        if (nodal) HIP_TRY(ctx, hipMemcpyAsync(P.h_nod[b], P.d_nod[b], sizeof(double) * (size_t)m * nb, hipMemcpyDeviceToHost, P.down));
        if (status) HIP_TRY(ctx, hipMemcpyAsync(P.h_stat[b], P.d_stat[b], sizeof(int32_t) * (size_t)m, hipMemcpyDeviceToHost, P.down));
        if (iters) HIP_TRY(ctx, hipMemcpyAsync(P.h_it[b], P.d_it[b], sizeof(int32_t) * (size_t)m, hipMemcpyDeviceToHost, P.down));
        HIP_TRY(ctx, hipEventRecord(P.e_down[b], P.down));
    }
    for (int64_t k = nchunk >= 2 ? nchunk - 2 : 0; k < nchunk; ++k) { rc = drain(k); if (rc) return rc; }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    RetryOut ro;
    const ScaleFn scale = [&](unsigned long long u) { return load_scale[u]; };
    rc = fail_retry(ctx, o, fail_threshold, load_scale ? &scale : nullptr, ro, &kernel_ms);
    if (rc) return rc;
    for (size_t r = 0; r < ro.rec.size(); ++r) {              // the second attempt's results in the place of the first's
        const size_t u = (size_t)ro.rec[r].unit;
        dns[u] = ro.dns[r];
        if (nodal) std::memcpy(nodal + u * nb, &ro.nodal[r * nb], sizeof(double) * nb);
        if (status) status[u] = ro.meta[r] & 3;
        if (iters) iters[u] = (int32_t)((uint32_t)ro.meta[r] >> 8);
    }
    ctx->last_kernel_ms = kernel_ms;
    return RELMC_OK;
}

// dns_dev (optional, n doubles): dns of every sample of the range in sampling order, beside the accumulators
int nsq_accumulate_impl(relmc_ctx* ctx, uint64_t seed, uint64_t first_index, int64_t n, const relmc_solver_opts* opts, relmc_acc* acc_out, double* dns_dev)
{
    if (!ctx) return RELMC_ERR_INVALID;
    if (!ctx->has_case) return fail(ctx, RELMC_ERR_NO_CASE, "relmc_nsq_accumulate: no case loaded");
    if (n < 0 || !acc_out) return fail(ctx, RELMC_ERR_INVALID, "relmc_nsq_accumulate: bad arguments");
    relmc_acc_zero(acc_out);
    if (n == 0) return RELMC_OK;
    relmc_solver_opts o;
    if (opts) o = *opts; else relmc_solver_opts_default(&o);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // a wavefront row counts its scenarios in 32 bits: split very large ranges
    // screen = 1 (relmc_screen.hip): a pre-pass draws the masks and runs the zero-curtailment certificate, one thread per sample; only the samples it
    // does not cover are solved (MODE 7, from the stored masks, in ascending sample order), the others add 1 to n and to n_screened
    const bool screen = o.screen != 0 && ctx->screen.tab.valid != 0;
    const int64_t kMaxPerLaunch = screen ? kScreenChunk : (int64_t)1 << 31;
    double ms_total = 0.0;
    for (int64_t done = 0; done < n;) {
        const int64_t m = (n - done) < kMaxPerLaunch ? (n - done) : kMaxPerLaunch;
        EvalArgs a = make_args(o);
        a.seed = seed; a.first_index = first_index + (uint64_t)done; a.n = m;
        a.dns = dns_dev ? dns_dev + done : nullptr;
        int64_t certified = 0;
        if (screen) {
            uint32_t ns = 0;
            if (dns_dev) HIP_TRY(ctx, hipMemsetAsync(dns_dev + done, 0, sizeof(double) * (size_t)m, ctx->stream));
            const int rc0 = screen_prepass_nsq(ctx, seed, first_index + (uint64_t)done, m, &ns, &ms_total);
            if (rc0) return rc0;
            certified = m - (int64_t)ns;
            a.n = (int64_t)ns; a.memo_keys = ctx->screen.keys; a.memo_perm = ctx->screen.idx;
        }
        int blocks = 0;
        relmc_acc part;
        relmc_acc_zero(&part);
        uint32_t listed = 0;
        for (int attempt = 0; a.n > 0; ++attempt) {
            int rc = fail_arm(ctx, a, done, true, m);
            if (rc) return rc;
            rc = launch_eval(ctx, screen ? 7 : 0, a, &blocks);
            if (rc) return rc;
            rc = launch_finalize(ctx, blocks);
            if (rc) return rc;
            // accumulators and the count of listed units come back in ONE synchronisation, through the context's pinned staging words (two more
            // blocking 4-byte copies per launch used to follow the kernel)
            HIP_TRY(ctx, hipMemcpyAsync(&ctx->hstage->acc, ctx->dacc, sizeof(relmc_acc), hipMemcpyDeviceToHost, ctx->stream));
            if (a.fail_count) HIP_TRY(ctx, hipMemcpyAsync(&ctx->hstage->fail_cnt, ctx->dfail_count, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
            rc = finish_timing(ctx);
            if (rc) return rc;
            ms_total += ctx->last_kernel_ms;
            part = ctx->hstage->acc;
            // more non-converged units than the list holds (a case the calibration did not foresee): a longer list and the same chunk again --
            // the launch is a function of (seed, range) alone, so the second one lists them all
            listed = a.fail_count ? ctx->hstage->fail_cnt : 0u;
            if (a.fail_list == nullptr || listed <= ctx->fail_cap || ctx->fail_cap >= kFailCapMax || attempt >= 2) break;
            HIP_TRY(ctx, hipMemset(ctx->dfail_count, 0, sizeof(uint32_t)));
            rc = fail_list_ensure(ctx, listed + listed / 8 > kFailCapMax ? kFailCapMax : listed + listed / 8);
            if (rc) return rc;
        }
        part.n += certified; part.n_screened += certified;
        int rc = RELMC_OK;
        RetryOut ro;
        if (a.n > 0) rc = fail_retry(ctx, o, a.fail_threshold, nullptr, ro, &ms_total, a.fail_count ? &listed : nullptr);
        if (rc) return rc;
        for (size_t r = 0; r < ro.rec.size(); ++r) {
            acc_add_unit(&part, ro.rec[r], ro.dns[r], ro.meta[r], &ro.nodal[r * (size_t)ctx->nb], ctx->nb, ctx->ncomp, a.fail_threshold);
            if (dns_dev) HIP_TRY(ctx, hipMemcpy(dns_dev + ro.rec[r].unit, &ro.dns[r], sizeof(double), hipMemcpyHostToDevice));
        }
        relmc_acc_merge(acc_out, &part);
        done += m;
    }
    ctx->last_kernel_ms = ms_total;
    return RELMC_OK;
}

}  // namespace relmc_host

using namespace relmc_host;

extern "C" {

int32_t relmc_mc_simulation_dev(relmc_ctx* ctx, const uint8_t* states_dev, int64_t n, const relmc_solver_opts* opts,
                                double* dns_dev, double* nodal_dev, int32_t* status_dev, int32_t* iters_dev)
{
    if (!ctx) return RELMC_ERR_INVALID;
    if (!ctx->has_case) return fail(ctx, RELMC_ERR_NO_CASE, "relmc_mc_simulation: no case loaded");
    if (n < 0 || (n > 0 && (!states_dev || !dns_dev))) return fail(ctx, RELMC_ERR_INVALID, "relmc_mc_simulation: bad arguments");
    if (n == 0) return RELMC_OK;
    relmc_solver_opts o;
    if (opts) o = *opts; else relmc_solver_opts_default(&o);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    EvalArgs a = make_args(o);
    a.n = n; a.states = states_dev; a.dns = dns_dev; a.nodal = nodal_dev; a.status = status_dev; a.iters = iters_dev;
    int blocks = 0;
    int rc = fail_arm(ctx, a, 0, true, n);
    if (rc) return rc;
    rc = launch_eval(ctx, 1, a, &blocks);
    if (rc) return rc;
    rc = finish_timing(ctx);
    if (rc) return rc;
    RetryOut ro;
    double ms = ctx->last_kernel_ms;
    rc = fail_retry(ctx, o, a.fail_threshold, nullptr, ro, &ms);
    if (rc) return rc;
    ctx->last_kernel_ms = ms;
    for (size_t r = 0; r < ro.rec.size(); ++r) {          // the second attempt's results in the place of the first's
        const size_t u = (size_t)ro.rec[r].unit, nb = (size_t)ctx->nb;
        const int32_t st = ro.meta[r] & 3, it = (int32_t)((uint32_t)ro.meta[r] >> 8);
        HIP_TRY(ctx, hipMemcpy(dns_dev + u, &ro.dns[r], sizeof(double), hipMemcpyHostToDevice));
        if (nodal_dev) HIP_TRY(ctx, hipMemcpy(nodal_dev + u * nb, &ro.nodal[r * nb], sizeof(double) * nb, hipMemcpyHostToDevice));
        if (status_dev) HIP_TRY(ctx, hipMemcpy(status_dev + u, &st, sizeof(int32_t), hipMemcpyHostToDevice));
        if (iters_dev) HIP_TRY(ctx, hipMemcpy(iters_dev + u, &it, sizeof(int32_t), hipMemcpyHostToDevice));
    }
    return RELMC_OK;
}

int32_t relmc_mc_simulation(relmc_ctx* ctx, const uint8_t* states_host, int64_t n, const relmc_solver_opts* opts,
                            double* dns_host, double* nodal_host, int32_t* status_host, int32_t* iters_host)
{
    if (!ctx) return RELMC_ERR_INVALID;
    if (!ctx->has_case) return fail(ctx, RELMC_ERR_NO_CASE, "relmc_mc_simulation: no case loaded");
    if (n < 0 || (n > 0 && (!states_host || !dns_host))) return fail(ctx, RELMC_ERR_INVALID, "relmc_mc_simulation: bad arguments");
    if (n == 0) return RELMC_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    relmc_solver_opts o;
    if (opts) o = *opts; else relmc_solver_opts_default(&o);
    return pipe_run(ctx, states_host, nullptr, n, o, 1e-4 /* nsqMain.m:270 */, dns_host, nodal_host, status_host, iters_host);
}

int32_t relmc_nsq_accumulate(relmc_ctx* ctx, uint64_t seed, uint64_t first_index, int64_t n, const relmc_solver_opts* opts,
                             relmc_acc* acc_out)
{
    return nsq_accumulate_impl(ctx, seed, first_index, n, opts, acc_out, nullptr);
}

// nsqMain.m:208-318: batches until beta <= beta_limit or max_samples, then the post-processing of :345-376
int32_t relmc_nsq_run(relmc_ctx* ctx, const relmc_nsq_opts* o, relmc_nsq_result* res)
{
    if (!ctx) return RELMC_ERR_INVALID;
    if (!ctx->has_case) return fail(ctx, RELMC_ERR_NO_CASE, "relmc_nsq_run: no case loaded");
    if (!o || !res || o->batch <= 0 || o->max_samples <= 0) return fail(ctx, RELMC_ERR_INVALID, "relmc_nsq_run: bad options");
    std::memset(res, 0, sizeof(*res));
    const auto t0 = std::chrono::steady_clock::now();
    const int nb = ctx->nb, ncomp = ctx->ncomp;
    double beta = INFINITY, kernel_ms = 0.0;
    int64_t done = 0, cp = 0;
    if (o->distinct_states == 2) { const int rc0 = relmc_db_reset(ctx); if (rc0) return rc0; }
    auto checkpoint = [&](const relmc_indices& ix) {
        if (cp < o->history_cap) {
            if (o->beta_history) o->beta_history[cp] = ix.beta;
            if (o->edns_history) o->edns_history[cp] = ix.edns;
            if (o->lole_history) o->lole_history[cp] = ix.lole;
            if (o->plc_history) o->plc_history[cp] = ix.plc;
        }
        cp++;
    };
    constexpr int64_t kStretch = 1 << 18, kStretchMaxBatch = 32768;
    // More than one rank (relmc_comm_init / relmc_comm_set_host_allreduce): every batch [done, done + m) of the global sample stream is split
    // contiguously over the ranks, each evaluates its slice, ONE all-reduce of the accumulators per batch (the convergence check), and every
    // rank computes the same indices and stops at the same batch -- the parfor of nsqMain.m:257-263 with the loop around it, so that a C,
    // Julia or MATLAB host calls this one function on every rank.  The sampler is keyed by (seed, global index): the integers of the result
    // do not depend on the number of ranks, the fp64 sums only in their summation order.  The state database (distinct_states = 2) is per
    // rank (each rank's rows are the states of ITS slices); its accumulators are cumulative, so they are all-reduced as they are.
    const int nranks = comm_ranks(ctx);
    // ... with small batches (the reference's own is 100, nsqMain.m:60) in STRETCHES like the single-rank loop below: every rank evaluates its
    // contiguous slice of a stretch of whole batches with the dns of each of its samples, folds it into per-checkpoint (sum dns, sum dns^2, losses)
    // partial triples, ONE all-reduce of 3 x checkpoints + the accumulators (as doubles: the counts are exact below 2^53) per stretch gives every rank
    // every checkpoint of the stretch; all ranks walk the same checkpoints and cut at the same one; a cut stretch is taken again over its used part
    // (one more all-reduce), so the run stops at the batch-by-batch loop's checkpoint with its accumulators.  Stretch lengths follow the
    // single-rank rule, so the checkpoints at which the history restarts from the accumulators are the single-rank run's.
    if (nranks > 1 && o->distinct_states == 0 && o->batch <= kStretchMaxBatch && !ctx->sw.nsq_no_stretch) {
        const int64_t R = nranks, r = ctx->comm_rank;
        const int64_t per = kStretch / o->batch * o->batch;
        HIP_TRY(ctx, hipSetDevice(ctx->device));
        if (ctx->hist_cap < per) {
            if (ctx->dhist) (void)hipFree(ctx->dhist);
            if (ctx->hhist) (void)hipHostFree(ctx->hhist);
            ctx->dhist = ctx->hhist = nullptr; ctx->hist_cap = 0;
            HIP_TRY(ctx, hipMalloc(&ctx->dhist, sizeof(double) * (size_t)per));
            HIP_TRY(ctx, hipHostMalloc(&ctx->hhist, sizeof(double) * (size_t)per, hipHostMallocDefault));
            ctx->hist_cap = per;
        }
        constexpr int64_t NI = (int64_t)(offsetof(relmc_acc, sum_dns) / sizeof(int64_t)), ND = (int64_t)((sizeof(relmc_acc) - offsetof(relmc_acc, sum_dns)) / sizeof(double));
        std::vector<double> box;
        auto pack = [&](const relmc_acc& a, double* q) {
            const int64_t* ai = reinterpret_cast<const int64_t*>(&a); const double* ad = &a.sum_dns;
            for (int64_t k = 0; k < NI; ++k) q[k] = (double)ai[k];
            for (int64_t k = 0; k < ND; ++k) q[NI + k] = ad[k];
        };
        auto unpack = [&](const double* q, relmc_acc& a) {
            int64_t* ai = reinterpret_cast<int64_t*>(&a); double* ad = &a.sum_dns;
            for (int64_t k = 0; k < NI; ++k) ai[k] = (int64_t)std::llround(q[k]);
            for (int64_t k = 0; k < ND; ++k) ad[k] = q[NI + k];
        };
        // this rank's slice of [lo0, lo0 + len): accumulators (and, with trip, the per-checkpoint partial triples), all-reduced over the ranks
        auto shared_eval = [&](int64_t lo0, int64_t len, double* trip, int64_t ncp, relmc_acc* out) -> int {
            const int64_t lo = lo0 + len * r / R, cnt = lo0 + len * (r + 1) / R - lo;
            relmc_acc part;
            relmc_acc_zero(&part);
            int rc = RELMC_OK;
            if (cnt > 0) {
                rc = nsq_accumulate_impl(ctx, o->seed, (uint64_t)lo, cnt, &o->solver, &part, trip ? ctx->dhist : nullptr);
                if (rc == RELMC_OK) kernel_ms += ctx->last_kernel_ms;
                if (rc == RELMC_OK && trip) {
                    if (hipMemcpyAsync(ctx->hhist, ctx->dhist, sizeof(double) * (size_t)cnt, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                        hipStreamSynchronize(ctx->stream) != hipSuccess) rc = fail(ctx, RELMC_ERR_HIP, "relmc_nsq_run: copy of the per-sample dns failed");
                }
            }
            const std::string local_err = ctx->err;
            box.assign((size_t)(3 * ncp + NI + ND), 0.0);
            if (rc == RELMC_OK) {
                if (trip) for (int64_t i = 0; i < cnt; ++i) {
                    const double v = ctx->hhist[(size_t)i];
                    double* t = &box[(size_t)(3 * ((lo - lo0 + i) / o->batch))];
                    t[0] += v; t[1] = std::fma(v, v, t[1]); t[2] += v > 1e-4 /* nsqMain.m:270 */ ? 1.0 : 0.0;
                }
                pack(part, &box[(size_t)(3 * ncp)]);
            } else box[(size_t)(3 * ncp)] = NAN;             // a rank whose slice failed still enters the collective and says so where every rank looks
            const int rc_ar = comm_allreduce_f64(ctx, box.data(), (int64_t)box.size());
            if (rc != RELMC_OK) { ctx->err = local_err; return rc; }
            if (rc_ar) return rc_ar;
            if (box[(size_t)(3 * ncp)] != box[(size_t)(3 * ncp)]) return fail(ctx, RELMC_ERR_HIP, "relmc_nsq_run: another rank failed to evaluate its slice of the stretch (see that rank's relmc_last_error)");
            if (trip) std::memcpy(trip, box.data(), sizeof(double) * (size_t)(3 * ncp));
            unpack(&box[(size_t)(3 * ncp)], *out);
            return RELMC_OK;
        };
        std::vector<double> trip;
        while (beta > o->beta_limit && done < o->max_samples) {
            // stretch length: the single-rank rule (below), on the same (done, beta): ~25 600 samples, then as many as the run holds, then from beta
            const int64_t first = 25600 / o->batch > 0 ? 25600 / o->batch * o->batch : o->batch;
            const int64_t least = 1600 / o->batch > 0 ? 1600 / o->batch * o->batch : o->batch;
            int64_t len = done > first ? done / o->batch * o->batch : first;
            bool final_stretch = false;
            if (done > 0 && o->beta_limit > 0.0 && beta < 1e6 && beta > o->beta_limit) {
                const double need = (double)done * (beta / o->beta_limit) * (beta / o->beta_limit);
                const double target = (double)done < 0.85 * need ? 0.9 * need : 1.03 * need;
                const double l = std::ceil((target - (double)done) / (double)o->batch) * (double)o->batch;
                len = l < (double)least ? least : (l > (double)per ? per : (int64_t)l);
                final_stretch = !((double)done < 0.85 * need);
            }
            if (len > per) len = per;
            if (!final_stretch) {
                const int64_t round = (int64_t)ctx->num_cu * ctx->blocks_per_cu * (ctx->tile == 0 ? Tile24::WPB * Tile24::SPW : Tile96::WPB * Tile96::SPW);
                const int64_t snapped = (len / round) * round / o->batch * o->batch;
                if (len >= 2 * round && snapped >= least) len = snapped;
            }
            const int64_t m = (o->max_samples - done) < len ? (o->max_samples - done) : len;
            const int64_t ncp = (m + o->batch - 1) / o->batch;
            trip.assign((size_t)(3 * ncp), 0.0);
            relmc_acc part;
            const int64_t ru0 = ctx->retry_units, rc0_ = ctx->retry_converged, rd0 = ctx->retry_dense_units, rdc0 = ctx->retry_dense_converged, ro0 = ctx->retry_overflow;
            const double kernel_ms0 = kernel_ms;
            int rc = shared_eval(done, m, trip.data(), ncp, &part);
            if (rc) return rc;
            relmc_acc run = res->acc;
            int64_t used = 0;
            for (int64_t k = 0; k < ncp; ++k) {
                const int64_t b = (m - used) < o->batch ? (m - used) : o->batch;
                run.n += b; run.n_fail += (int64_t)std::llround(trip[(size_t)(3 * k + 2)]); run.sum_dns += trip[(size_t)(3 * k)]; run.sum_dns2 += trip[(size_t)(3 * k + 1)];
                used += b;
                relmc_indices ix;
                relmc_nsq_indices(&run, 0, 0, o->hours_per_year, &ix);
                beta = ix.beta;
                checkpoint(ix);
                if (beta <= o->beta_limit) break;
            }
            if (used < m) {                                    // cut: the stretch again over its used part, on every rank (they all see the same beta)
                ctx->retry_units = ru0; ctx->retry_converged = rc0_; ctx->retry_dense_units = rd0; ctx->retry_dense_converged = rdc0; ctx->retry_overflow = ro0;
                kernel_ms = kernel_ms0;
                rc = shared_eval(done, used, nullptr, 0, &part);
                if (rc) return rc;
            }
            relmc_acc_merge(&res->acc, &part);
            done += used;
            relmc_nsq_indices(&res->acc, nb, ncomp, o->hours_per_year, &res->idx);
            beta = res->idx.beta;
            cp--;
            checkpoint(res->idx);
        }
    }
    else if (nranks > 1) {
        const int64_t R = nranks, r = ctx->comm_rank;
        while (beta > o->beta_limit && done < o->max_samples) {
            const int64_t m = (o->max_samples - done) < o->batch ? (o->max_samples - done) : o->batch;
            const int64_t lo = done + m * r / R, cnt = done + m * (r + 1) / R - lo;
            relmc_acc part;
            relmc_acc_zero(&part);
            int rc = RELMC_OK;
            if (o->distinct_states == 2) rc = relmc_nsq_db_batch(ctx, o->seed, (uint64_t)lo, cnt, &o->solver, &part, nullptr);      // cumulative over this rank's slices
            else if (cnt > 0) rc = o->distinct_states ? relmc_nsq_accumulate_distinct(ctx, o->seed, (uint64_t)lo, cnt, &o->solver, &part, nullptr)
                                                      : relmc_nsq_accumulate(ctx, o->seed, (uint64_t)lo, cnt, &o->solver, &part);
            if (rc == RELMC_OK && (cnt > 0 || o->distinct_states == 2)) kernel_ms += ctx->last_kernel_ms;
            // a rank whose slice failed still enters the collective (the others would wait for it for ever) and says so in a counter no
            // evaluation ever makes negative: every rank then returns an error from the same batch
            const std::string local_err = ctx->err;
            if (rc != RELMC_OK) { relmc_acc_zero(&part); part.n_nonconverged = -((int64_t)1 << 40); }
            const int rc_ar = relmc_comm_allreduce_acc(ctx, &part);
            if (rc != RELMC_OK) { ctx->err = local_err; return rc; }
            if (rc_ar) return rc_ar;
            if (part.n_nonconverged < 0) return fail(ctx, RELMC_ERR_HIP, "relmc_nsq_run: another rank failed to evaluate its slice of the batch (see that rank's relmc_last_error)");
            if (o->distinct_states == 2) res->acc = part; else relmc_acc_merge(&res->acc, &part);
            done += m;
            relmc_nsq_indices(&res->acc, nb, ncomp, o->hours_per_year, &res->idx);
            beta = res->idx.beta;
            checkpoint(res->idx);
        }
    }
    else
    // Small batches (the reference's own is 100 samples, nsqMain.m:60) would make every checkpoint one launch of a nearly
    // empty grid.  They are evaluated many at a time instead: one pass returns the accumulators of the whole stretch and
    // the dns of each of its samples; the four indices of every checkpoint inside it (nsqMain.m:286-301 need only the dns
    // sums and the loss count) follow on the host.  If beta reaches its limit inside the stretch, the stretch is cut at that
    // checkpoint and taken again over the shorter range (the database is first put back to its rows and counts of before
    // the stretch), so that the result is the one of the batch-by-batch loop.  Only for batches whose launch is overhead-bound (a launch
    // costs 0.2-0.4 ms whatever its size, i.e. as much as 1e4 scenarios), and with stretches sized from the run's own beta (below; up to 2^18
    // samples each): what a cut throws away stays a few per cent of the run.
    if ((o->distinct_states == 0 || o->distinct_states == 2) && o->batch <= kStretchMaxBatch &&
        !ctx->sw.nsq_no_stretch /* diagnosis: one launch per batch */) {
        const bool use_db = o->distinct_states == 2;
        const int64_t per = kStretch / o->batch * o->batch;       // buffer size = longest stretch
        HIP_TRY(ctx, hipSetDevice(ctx->device));
        if (ctx->hist_cap < per) {
            if (ctx->dhist) (void)hipFree(ctx->dhist);
            if (ctx->hhist) (void)hipHostFree(ctx->hhist);
            ctx->dhist = ctx->hhist = nullptr; ctx->hist_cap = 0;
            HIP_TRY(ctx, hipMalloc(&ctx->dhist, sizeof(double) * (size_t)per));
            HIP_TRY(ctx, hipHostMalloc(&ctx->hhist, sizeof(double) * (size_t)per, hipHostMallocDefault));
            ctx->hist_cap = per;
        }
        const double* const hd = ctx->hhist;
        while (beta > o->beta_limit && done < o->max_samples) {
            // How long a stretch?  beta falls like 1 / sqrt(n), so the run will need about done * (beta / limit)^2 samples: go to 90 % of that in
            // one stretch, then to 103 % of the (then better) prediction -- a stretch that is cut is taken again over its used part, so the last one
            // should be short (round 3: beta < 1 % at the reference's batch of 100 in 5.9 instead of 9.4 ms; doubling stretches evaluated 416 k
            // samples for a run of 211 k).  Without a prediction (first stretch, no loss yet, limit 0): ~25 600 samples, then as many as the run holds.
            const int64_t first = 25600 / o->batch > 0 ? 25600 / o->batch * o->batch : o->batch;      // ~25 600 samples, whole batches
            const int64_t least = 1600 / o->batch > 0 ? 1600 / o->batch * o->batch : o->batch;
            int64_t len = done > first ? done / o->batch * o->batch : first;
            bool final_stretch = false;
            if (done > 0 && o->beta_limit > 0.0 && beta < 1e6 && beta > o->beta_limit) {
                const double need = (double)done * (beta / o->beta_limit) * (beta / o->beta_limit);
                const double target = (double)done < 0.85 * need ? 0.9 * need : 1.03 * need;
                const double l = std::ceil((target - (double)done) / (double)o->batch) * (double)o->batch;
                len = l < (double)least ? least : (l > (double)per ? per : (int64_t)l);
                final_stretch = !((double)done < 0.85 * need);
            }
            if (len > per) len = per;
            // A launch costs as many rounds as its busiest wavefront walks scenario groups: 24 576 samples are three groups for every wavefront of
            // the 16-lane tile's grid, 25 600 make some walk a fourth (0.45 against 0.60 ms).  Stretches that are not the last one end just below
            // a whole number of rounds (in whole batches); the last one keeps its length -- it has to reach the stopping point.
            if (!use_db && !final_stretch) {
                const int64_t round = (int64_t)ctx->num_cu * ctx->blocks_per_cu * (ctx->tile == 0 ? Tile24::WPB * Tile24::SPW : Tile96::WPB * Tile96::SPW);
                const int64_t snapped = (len / round) * round / o->batch * o->batch;
                if (len >= 2 * round && snapped >= least) len = snapped;
            }
            const int64_t m = (o->max_samples - done) < len ? (o->max_samples - done) : len;
            relmc_acc part;
            int rc;
            const int64_t rows0 = ctx->db_n, samples0 = ctx->db_samples;
            // what a stretch that is cut and taken again must not count twice: its second attempts, its kernel time
            const int64_t ru0 = ctx->retry_units, rc0_ = ctx->retry_converged, rd0 = ctx->retry_dense_units, rdc0 = ctx->retry_dense_converged, ro0 = ctx->retry_overflow;
            const double kernel_ms0 = kernel_ms;
            if (use_db) {
                rc = db_snapshot(ctx);
                if (rc) return rc;
                rc = relmc_nsq_db_batch(ctx, o->seed, (uint64_t)done, m, &o->solver, nullptr, nullptr);
                if (rc) return rc;
                kernel_ms += ctx->last_kernel_ms;
                rc = db_sample_dns(ctx, o->seed, (uint64_t)done, m, ctx->dhist);
                if (rc) return rc;
            } else {
                rc = nsq_accumulate_impl(ctx, o->seed, (uint64_t)done, m, &o->solver, &part, ctx->dhist);
                if (rc) return rc;
                kernel_ms += ctx->last_kernel_ms;
            }
            HIP_TRY(ctx, hipMemcpyAsync(ctx->hhist, ctx->dhist, sizeof(double) * (size_t)m, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            relmc_acc run = res->acc;                       // only n, n_fail, sum_dns, sum_dns2 are advanced per checkpoint
            int64_t used = 0;
            while (used < m) {
                const int64_t b = (m - used) < o->batch ? (m - used) : o->batch;
                double sd = 0.0, sd2 = 0.0; int64_t nf = 0;
                for (int64_t i = used; i < used + b; ++i) { const double v = hd[(size_t)i]; sd += v; sd2 = std::fma(v, v, sd2); nf += v > 1e-4 /* nsqMain.m:270 */; }
                if (sd != sd) return fail(ctx, RELMC_ERR_HIP, "relmc_nsq_run: a sampled state is missing from the database");
                run.n += b; run.n_fail += nf; run.sum_dns += sd; run.sum_dns2 += sd2;
                used += b;
                relmc_indices ix;
                relmc_nsq_indices(&run, 0, 0, o->hours_per_year, &ix);
                beta = ix.beta;
                checkpoint(ix);
                if (beta <= o->beta_limit) break;
            }
            if (used < m) {                                    // the discarded stretch leaves no trace in the bookkeeping
                ctx->retry_units = ru0; ctx->retry_converged = rc0_; ctx->retry_dense_units = rd0; ctx->retry_dense_converged = rdc0; ctx->retry_overflow = ro0;
                kernel_ms = kernel_ms0;
            }
            if (used < m && use_db) {                          // stopped inside the stretch: the database as it was, then the shorter range
                rc = db_rewind(ctx, rows0, samples0);
                if (rc) return rc;
                rc = relmc_nsq_db_batch(ctx, o->seed, (uint64_t)done, used, &o->solver, nullptr, nullptr);
                if (rc) return rc;
                kernel_ms += ctx->last_kernel_ms;
            } else if (used < m) {
                rc = relmc_nsq_accumulate(ctx, o->seed, (uint64_t)done, used, &o->solver, &part);
                if (rc) return rc;
                kernel_ms += ctx->last_kernel_ms;
            }
            if (use_db) {
                const auto t1 = std::chrono::steady_clock::now();
                rc = db_accumulate(ctx, &res->acc);                    // nsqMain.m:282-301 over all rows
                if (rc) return rc;
                kernel_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count();
            } else relmc_acc_merge(&res->acc, &part);
            done += used;
            // the stretch's last checkpoint from the accumulators themselves (what the caller is handed), not from the host sums
            relmc_nsq_indices(&res->acc, nb, ncomp, o->hours_per_year, &res->idx);
            beta = res->idx.beta;
            cp--;
            checkpoint(res->idx);
        }
    }
    else
    while (beta > o->beta_limit && done < o->max_samples) {
        const int64_t m = (o->max_samples - done) < o->batch ? (o->max_samples - done) : o->batch;
        relmc_acc part;
        int rc;
        if (o->distinct_states == 2) {
            // the reference's own loop body: persistent unique-state database, indices recomputed from all of its rows
            rc = relmc_nsq_db_batch(ctx, o->seed, (uint64_t)done, m, &o->solver, &res->acc, nullptr);
        } else {
            rc = o->distinct_states ? relmc_nsq_accumulate_distinct(ctx, o->seed, (uint64_t)done, m, &o->solver, &part, nullptr)
                                    : relmc_nsq_accumulate(ctx, o->seed, (uint64_t)done, m, &o->solver, &part);
            if (rc == RELMC_OK) relmc_acc_merge(&res->acc, &part);
        }
        if (rc) return rc;
        kernel_ms += ctx->last_kernel_ms;
        done += m;
        relmc_nsq_indices(&res->acc, nb, ncomp, o->hours_per_year, &res->idx);
        beta = res->idx.beta;
        checkpoint(res->idx);
    }
    res->checkpoints = cp < o->history_cap ? cp : o->history_cap;     // history entries written
    res->batches = cp;
    res->converged = beta <= o->beta_limit ? 1 : 0;
    res->kernel_seconds = kernel_ms * 1e-3;
    res->wall_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    ctx->last_kernel_ms = kernel_ms;
    return RELMC_OK;
}

}  // extern "C"
