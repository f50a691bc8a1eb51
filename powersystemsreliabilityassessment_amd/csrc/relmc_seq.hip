// relmc_seq.hip — the sequential (chronological) HL2 track of /root/reference/Montecarlo_seq/ (seq_mcsampling.m, seq_mcsimulation.m, calnlc.m,
// seqMain.m:85-249) and the HL1 copper-sheet model of GeneratingAdequacy/PowerSystemAdequacy.jl:169-208.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>

#include "relmc_ctx.h"
#include "relmc_seq_kernels.h"

namespace relmc_host {

void seq_free(relmc_ctx* ctx)
{
    for (void* q : {(void*)ctx->sq_dm, (void*)ctx->sq_hours, (void*)ctx->sq_curt, (void*)ctx->sq_counts, (void*)ctx->sq_off, (void*)ctx->sq_year, (void*)ctx->dseq,
                    (void*)ctx->dlf, (void*)ctx->dhl1, (void*)ctx->dsorted, (void*)ctx->dsuffix, (void*)ctx->h1_lole, (void*)ctx->h1_eue, (void*)ctx->h1_part}) if (q) (void)hipFree(q);
    ctx->h1_lole = ctx->h1_eue = ctx->h1_part = nullptr; ctx->h1_cap = 0; ctx->h1_part_cap = 0;
    ctx->sq_dm = nullptr; ctx->sq_hours = nullptr; ctx->sq_curt = nullptr; ctx->sq_counts = ctx->sq_off = nullptr; ctx->sq_year = nullptr;
    ctx->sq_dm_words = 0; ctx->sq_nh = 0; ctx->sq_years = 0;
    ctx->dseq = nullptr; ctx->dlf = nullptr; ctx->dhl1 = nullptr; ctx->dsorted = ctx->dsuffix = nullptr; ctx->has_seq = false; ctx->has_hl1 = false;
}

namespace {
// chronology of years [first_year, first_year + n_years) into device masks (every word written)
// *dmasks_out == nullptr on entry: a buffer is allocated for the caller (who frees it); otherwise the masks go into the caller's buffer
int seq_sample(relmc_ctx* ctx, uint64_t seed, uint64_t first_year, int n_years, uint32_t** dmasks_out)
{
    const size_t words = (size_t)n_years * ctx->hseq.hpy * ctx->hseq.mw;
    const bool own = *dmasks_out == nullptr;
    uint32_t* dm = *dmasks_out;
    if (own) HIP_TRY(ctx, hipMalloc(&dm, words * sizeof(uint32_t)));
    // segments of at most ~48 KB of masks in LDS, and at least ~2 workgroups per CU over the launch (a segment's chains are walked from hour 0: cheap)
    const int hpy = ctx->hseq.hpy, mw = ctx->hseq.mw, cap = (48 * 1024) / (4 * mw) - 64;
    int nseg = (hpy + cap - 1) / cap;
    while ((int64_t)n_years * nseg < 2 * (int64_t)ctx->num_cu && nseg < 16 && hpy / (nseg + 1) >= 256) ++nseg;
    const int seg_len = (hpy + nseg - 1) / nseg;
    nseg = (hpy + seg_len - 1) / seg_len;
    int sl = seg_len;                                            // LDS row stride = 64 / mw (mod 64): conflict-free word-major rows
    while (sl % 64 != (64 / mw) % 64) ++sl;
    hipLaunchKernelGGL(relmc_seq_sampling_kernel, dim3((unsigned)((int64_t)n_years * nseg)), dim3(256), (size_t)sl * mw * sizeof(uint32_t), ctx->stream,
                       ctx->dseq, seed, first_year, nseg, seg_len, sl, dm);
    if (hipGetLastError() != hipSuccess) { if (own) (void)hipFree(dm); return fail(ctx, RELMC_ERR_HIP, "seq: sampling launch failed"); }
    *dmasks_out = dm;
    return RELMC_OK;
}
}  // namespace

}  // namespace relmc_host

using namespace relmc_host;

extern "C" {

// ---- sequential HL2: Montecarlo_seq/ ---------------------------------------------------------------------
int32_t relmc_seq_load(relmc_ctx* ctx, const double* mttf, const double* mttr, int32_t hpy, const double* load_factors)
{
    if (!ctx) return RELMC_ERR_INVALID;
    if (!ctx->has_case) return fail(ctx, RELMC_ERR_NO_CASE, "relmc_seq_load: no case loaded");
    if (!mttf || !mttr || !load_factors || hpy < 1 || hpy > 65535) return fail(ctx, RELMC_ERR_INVALID, "relmc_seq_load: bad arguments");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    SeqCase& q = ctx->hseq;
    std::memset(&q, 0, sizeof(q));
    if (ctx->ncomp > SEQ_NCOMPMAX) return fail(ctx, RELMC_ERR_UNSUPPORTED, "relmc_seq_load: more than 256 components");
    q.ncomp = ctx->ncomp; q.hpy = hpy; q.mw = ctx->tile == 0 ? Tile24::OW : Tile96::OW;
    for (int k = 0; k < q.ncomp; ++k) {
        if (!(mttf[k] > 0) || !(mttr[k] > 0)) return fail(ctx, RELMC_ERR_INVALID, "relmc_seq_load: MTTF / MTTR must be positive");
        q.mttf[k] = mttf[k]; q.mttr[k] = mttr[k];
    }
    if (!ctx->dseq) HIP_TRY(ctx, hipMalloc(&ctx->dseq, sizeof(SeqCase)));
    if (ctx->dlf) (void)hipFree(ctx->dlf);
    ctx->dlf = nullptr;
    HIP_TRY(ctx, hipMalloc(&ctx->dlf, sizeof(double) * hpy));
    ctx->hlf.assign(load_factors, load_factors + hpy);
    HIP_TRY(ctx, hipMemcpy(ctx->dseq, &q, sizeof(q), hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(ctx->dlf, load_factors, sizeof(double) * hpy, hipMemcpyHostToDevice));
    ctx->has_seq = true;
    return RELMC_OK;
}

int32_t relmc_seq_mcsampling(relmc_ctx* ctx, uint64_t seed, uint64_t first_year, int32_t num_years, uint8_t* state_host)
{
    if (!ctx) return RELMC_ERR_INVALID;
    if (!ctx->has_seq) return fail(ctx, RELMC_ERR_NO_CASE, "relmc_seq_mcsampling: relmc_seq_load has not been called");
    if (num_years < 0 || (num_years > 0 && !state_host)) return fail(ctx, RELMC_ERR_INVALID, "relmc_seq_mcsampling: bad arguments");
    if (num_years == 0) return RELMC_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    uint32_t* dm = nullptr;
    int rc = seq_sample(ctx, seed, first_year, num_years, &dm);
    if (rc) return rc;
    const int64_t nh = (int64_t)num_years * ctx->hseq.hpy;
    const size_t bytes = (size_t)nh * ctx->hseq.ncomp;
    uint8_t* dst = nullptr;
    if (hipMalloc(&dst, bytes) != hipSuccess) { (void)hipFree(dm); return fail(ctx, RELMC_ERR_HIP, "relmc_seq_mcsampling: allocation failed"); }
    hipLaunchKernelGGL(relmc_seq_expand_kernel, dim3(ctx->num_cu * 8), dim3(256), 0, ctx->stream, dm, nh, ctx->hseq.ncomp, ctx->hseq.mw, dst);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess ||
        hipMemcpy(state_host, dst, bytes, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(ctx, RELMC_ERR_HIP, "relmc_seq_mcsampling: kernel / copy failed");
    (void)hipFree(dm); (void)hipFree(dst);
    return rc;
}

int32_t relmc_seq_mcsimulation(relmc_ctx* ctx, const uint8_t* states_host, const double* load_scale_host, int64_t n, const relmc_solver_opts* opts,
                               double* dns_host, double* nodal_host, int32_t* status_host, int32_t* iters_host)
{
    if (!ctx) return RELMC_ERR_INVALID;
    if (!ctx->has_case) return fail(ctx, RELMC_ERR_NO_CASE, "relmc_seq_mcsimulation: no case loaded");
    if (n < 0 || (n > 0 && (!states_host || !dns_host || !load_scale_host))) return fail(ctx, RELMC_ERR_INVALID, "relmc_seq_mcsimulation: bad arguments");
    if (n == 0) return RELMC_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    relmc_solver_opts o;
    if (opts) o = *opts; else relmc_solver_opts_default(&o);
    return pipe_run(ctx, states_host, load_scale_host, n, o, 0.01 /* CURTAIL_THRESHOLD, seqMain.m:41 */, dns_host, nodal_host, status_host, iters_host);
}

int32_t relmc_seq_years(relmc_ctx* ctx, uint64_t seed, uint64_t first_year, int32_t n_years, const relmc_solver_opts* opts,
                        double curtail_threshold, relmc_seq_year* years_out, relmc_acc* acc_out)
{
    if (!ctx) return RELMC_ERR_INVALID;
    if (!ctx->has_seq) return fail(ctx, RELMC_ERR_NO_CASE, "relmc_seq_years: relmc_seq_load has not been called");
    if (n_years < 0 || !acc_out || (n_years > 0 && !years_out)) return fail(ctx, RELMC_ERR_INVALID, "relmc_seq_years: bad arguments");
    relmc_acc_zero(acc_out);
    if (n_years == 0) return RELMC_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    relmc_solver_opts o;
    if (opts) o = *opts; else relmc_solver_opts_default(&o);
    const int hpy = ctx->hseq.hpy;
    // the device buffers of a step live in the context and only ever grow
    const size_t nh = (size_t)n_years * hpy, words = nh * (size_t)ctx->hseq.mw;
    bool alloc_ok = true;
    if (ctx->sq_dm_words < words) { if (ctx->sq_dm) (void)hipFree(ctx->sq_dm); ctx->sq_dm = nullptr; ctx->sq_dm_words = 0;
                                    alloc_ok = hipMalloc(&ctx->sq_dm, words * sizeof(uint32_t)) == hipSuccess; if (alloc_ok) ctx->sq_dm_words = words; }
    if (alloc_ok && ctx->sq_nh < nh) { if (ctx->sq_hours) (void)hipFree(ctx->sq_hours); if (ctx->sq_curt) (void)hipFree(ctx->sq_curt); ctx->sq_hours = nullptr; ctx->sq_curt = nullptr; ctx->sq_nh = 0;
                                       alloc_ok = hipMalloc(&ctx->sq_hours, nh * sizeof(uint16_t)) == hipSuccess && hipMalloc(&ctx->sq_curt, nh * sizeof(double)) == hipSuccess; if (alloc_ok) ctx->sq_nh = nh; }
    if (alloc_ok && ctx->sq_years < n_years) {
        for (void* q : {(void*)ctx->sq_counts, (void*)ctx->sq_off, (void*)ctx->sq_year}) if (q) (void)hipFree(q);
        ctx->sq_counts = ctx->sq_off = nullptr; ctx->sq_year = nullptr; ctx->sq_years = 0;
        // sq_counts: [0, n) listed hours per year, [cap, cap + n) contingency hours per year (pre-screen)
        alloc_ok = hipMalloc(&ctx->sq_counts, sizeof(uint32_t) * 2 * n_years) == hipSuccess && hipMalloc(&ctx->sq_off, sizeof(uint32_t) * (n_years + 1)) == hipSuccess &&
                   hipMalloc(&ctx->sq_year, sizeof(double) * 3 * n_years) == hipSuccess;
        if (alloc_ok) ctx->sq_years = n_years;
    }
    if (!alloc_ok) return fail(ctx, RELMC_ERR_HIP, "relmc_seq_years: device allocation failed");
    uint32_t* dm = ctx->sq_dm; uint16_t* const dhours = ctx->sq_hours; uint32_t* const dcounts = ctx->sq_counts; uint32_t* const doff = ctx->sq_off;
    double* const dcurt = ctx->sq_curt; double* const dyear = ctx->sq_year;
    auto cleanup = [&]() {};
    int rc = seq_sample(ctx, seed, first_year, n_years, &dm);
    if (rc) return rc;
    std::vector<uint32_t> counts(n_years), ncont(n_years), off(n_years + 1, 0);
    bool ok = hipMemsetAsync(dcurt, 0, nh * sizeof(double), ctx->stream) == hipSuccess;
    // screen = 1 (relmc_screen.hip): the contingency hours are counted as before, but only the ones the zero-curtailment certificate does not cover
    // -- at the hour's own load factor -- are listed for the interior point; a covered hour's curtailment stays the 0 it was set to above
    const bool screen = o.screen != 0 && ctx->screen.tab.valid != 0;
    uint32_t* const dncont = dcounts + ctx->sq_years;
    if (screen) ok = ok && screen_seq_compact(ctx, dm, n_years, dhours, dcounts, dncont) == RELMC_OK;
    else hipLaunchKernelGGL(relmc_seq_compact_kernel, dim3(n_years), dim3(256), 0, ctx->stream, dm, hpy, ctx->hseq.mw, dhours, dcounts);
    ok = ok && hipGetLastError() == hipSuccess && hipMemcpyAsync(counts.data(), dcounts, sizeof(uint32_t) * n_years, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess &&
         (!screen || hipMemcpyAsync(ncont.data(), dncont, sizeof(uint32_t) * n_years, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess) &&
         hipStreamSynchronize(ctx->stream) == hipSuccess;
    if (!ok) { cleanup(); return fail(ctx, RELMC_ERR_HIP, "relmc_seq_years: compaction failed"); }
    int64_t certified = 0;
    for (int y = 0; y < n_years; ++y) {
        off[y + 1] = off[y] + counts[y];
        years_out[y].n_contingency = screen ? ncont[y] : counts[y];
        if (screen) certified += (int64_t)ncont[y] - (int64_t)counts[y];
    }
    const int64_t nlp = off[n_years];
    double ms = 0.0;
    if (screen) { float t = 0.f; if (hipEventElapsedTime(&t, ctx->screen.ev0, ctx->screen.ev1) == hipSuccess) ms += t; }      // the certificate's pre-pass (the stream was synchronised above)
    if (nlp > 0) {
        if (hipMemcpyAsync(doff, off.data(), sizeof(uint32_t) * (n_years + 1), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { cleanup(); return fail(ctx, RELMC_ERR_HIP, "relmc_seq_years: H2D failed"); }
        EvalArgs a = make_args(o);
        a.fail_threshold = curtail_threshold;
        a.n = nlp; a.seq_masks = dm; a.seq_offsets = doff; a.seq_hours = dhours; a.load_factors = ctx->dlf; a.curt = dcurt;
        a.seq_nyears = n_years; a.seq_hpy = hpy;
        int blocks = 0;
        rc = fail_arm(ctx, a, 0, true, a.n);
        if (rc) { cleanup(); return rc; }
        rc = launch_eval(ctx, 2, a, &blocks);
        if (rc) { cleanup(); return rc; }
        // accumulators and the count of listed hours in one synchronisation (pinned staging), as in the fused non-sequential pass
        if (launch_finalize(ctx, blocks) != RELMC_OK || hipMemcpyAsync(&ctx->hstage->acc, ctx->dacc, sizeof(relmc_acc), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            (a.fail_count && hipMemcpyAsync(&ctx->hstage->fail_cnt, ctx->dfail_count, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess)) { cleanup(); return fail(ctx, RELMC_ERR_HIP, "relmc_seq_years: finalize failed"); }
        rc = finish_timing(ctx);
        if (rc) { cleanup(); return rc; }
        *acc_out = ctx->hstage->acc;
        const uint32_t listed = a.fail_count ? ctx->hstage->fail_cnt : 0u;
        ms += ctx->last_kernel_ms;
        RetryOut ro;                                                   // hours the primary elimination order did not converge on
        const ScaleFn scale = [&](unsigned long long u) { return ctx->hlf[(size_t)(u % (unsigned long long)hpy)]; };
        rc = fail_retry(ctx, o, curtail_threshold, &scale, ro, &ms, a.fail_count ? &listed : nullptr);
        if (rc) { cleanup(); return rc; }
        for (size_t r = 0; r < ro.rec.size(); ++r) {
            acc_add_unit(acc_out, ro.rec[r], ro.dns[r], ro.meta[r], &ro.nodal[r * (size_t)ctx->nb], ctx->nb, ctx->ncomp, curtail_threshold);
            if (hipMemcpy(dcurt + ro.rec[r].unit, &ro.dns[r], sizeof(double), hipMemcpyHostToDevice) != hipSuccess) { cleanup(); return fail(ctx, RELMC_ERR_HIP, "relmc_seq_years: H2D failed"); }
        }
    }
    acc_out->n += certified; acc_out->n_screened += certified;
    hipLaunchKernelGGL(relmc_seq_annual_kernel, dim3(n_years), dim3(256), 0, ctx->stream, dcurt, hpy, curtail_threshold, dyear);
    std::vector<double> yr((size_t)3 * n_years);
    if (hipGetLastError() != hipSuccess || hipMemcpyAsync(yr.data(), dyear, sizeof(double) * 3 * n_years, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess) { cleanup(); return fail(ctx, RELMC_ERR_HIP, "relmc_seq_years: annual indices failed"); }
    for (int y = 0; y < n_years; ++y) { years_out[y].ens = yr[3 * y]; years_out[y].dlc = yr[3 * y + 1]; years_out[y].nlc = yr[3 * y + 2]; }
    ctx->last_kernel_ms = ms;
    cleanup();
    return RELMC_OK;
}

void relmc_seq_opts_default(relmc_seq_opts* o)
{
    if (!o) return;
    std::memset(o, 0, sizeof(*o));
    o->cov_threshold = 0.05;      /* seqMain.m:40 */
    o->max_years = 4000;          /* seqMain.m:39 */
    o->curtail_threshold = 0.01;  /* seqMain.m:41 */
    o->batch_years = 0;           /* 64 per rank */
    o->seed = 1;
    relmc_solver_opts_default(&o->solver);
}

// seqMain.m:85-199 (the yearly loop with its CoV stop) + :211-249 (LOLE / LOLF, nodal EENS, component importance), single- and multi-rank.
// Years are independent streams keyed by (seed, global year) and every year starts all-up (seqMain.m:91 samples ONE year per call), so the
// years [done, done + m) of a batch are split contiguously over the ranks of the context's communicator; the annual (ens, dlc, nlc) triples are
// all-gathered in year order (a sum all-reduce of a vector in which every rank fills its own slots), every rank walks the same CoV curve
// (:180-185) and stops at the same year (:194); the post-processing accumulators (:146-159) cover exactly the years up to the stopping year --
// the batch it falls into is evaluated again over its used part -- and are all-reduced once per batch.
int32_t relmc_seq_run(relmc_ctx* ctx, const relmc_seq_opts* o, relmc_seq_result* res)
{
    if (!ctx) return RELMC_ERR_INVALID;
    if (!ctx->has_seq) return fail(ctx, RELMC_ERR_NO_CASE, "relmc_seq_run: relmc_seq_load has not been called");
    if (!o || !res || o->max_years < 1 || o->batch_years < 0 || !(o->cov_threshold >= 0.0) ||
        ((o->results_year || o->cum_eens || o->cum_cov) && o->years_cap < o->max_years))
        return fail(ctx, RELMC_ERR_INVALID, "relmc_seq_run: bad options (max_years >= 1, batch_years >= 0; history buffers need years_cap >= max_years entries)");
    std::memset(res, 0, sizeof(*res));
    const auto t0 = std::chrono::steady_clock::now();
    const int R = comm_ranks(ctx), r = R > 1 ? ctx->comm_rank : 0;
    const int64_t batch = o->batch_years > 0 ? o->batch_years : (int64_t)64 * R;
    std::vector<double> ens, dlc, nlc, trip;
    std::vector<relmc_seq_year> mine;
    double kernel_ms = 0.0, mean = 0.0, cov = 0.0;
    int64_t done = 0; bool stop = false;
    int64_t n_cont = 0;
    auto eval = [&](int64_t lo, int64_t cnt, relmc_acc* acc) -> int {      // this rank's years [lo, lo + cnt)
        relmc_acc_zero(acc);
        if (cnt <= 0) return RELMC_OK;
        mine.resize((size_t)cnt);
        const int rc = relmc_seq_years(ctx, o->seed, (uint64_t)lo, (int32_t)cnt, &o->solver, o->curtail_threshold, mine.data(), acc);
        if (rc == RELMC_OK) kernel_ms += ctx->last_kernel_ms;
        return rc;
    };
    while (done < o->max_years && !stop) {
        const int64_t m = (o->max_years - done) < batch ? (o->max_years - done) : batch;
        const int64_t lo = done + m * r / R, cnt = done + m * (r + 1) / R - lo;
        relmc_acc acc;
        // what a batch that is cut at the stopping year and taken again must not count twice (as relmc_nsq_run): second attempts, kernel time
        const int64_t ru0 = ctx->retry_units, rcv0 = ctx->retry_converged, rd0 = ctx->retry_dense_units, rdc0 = ctx->retry_dense_converged, ro0 = ctx->retry_overflow;
        const double kernel_ms0 = kernel_ms;
        int rc = eval(lo, cnt, &acc);
        const std::string local_err = ctx->err;
        // annual triples of the batch in year order on every rank; a rank whose years failed says so with NaNs every rank will see
        trip.assign((size_t)(4 * m), 0.0);
        for (int64_t k = 0; k < cnt; ++k) {
            const size_t q = (size_t)(4 * (lo - done + k));
            if (rc == RELMC_OK) { trip[q] = mine[(size_t)k].ens; trip[q + 1] = mine[(size_t)k].dlc; trip[q + 2] = mine[(size_t)k].nlc; trip[q + 3] = (double)mine[(size_t)k].n_contingency; }
            else trip[q] = trip[q + 1] = trip[q + 2] = trip[q + 3] = NAN;
        }
        if (rc != RELMC_OK && cnt == 0) trip[0] = NAN;
        if (R > 1) {
            const int rc2 = comm_allreduce_f64(ctx, trip.data(), 4 * m);
            if (rc != RELMC_OK) { ctx->err = local_err; return rc; }
            if (rc2) return rc2;
            for (double v : trip) if (v != v) return fail(ctx, RELMC_ERR_HIP, "relmc_seq_run: another rank failed to evaluate its years of the batch (see that rank's relmc_last_error)");
        } else if (rc != RELMC_OK) return rc;
        int64_t used = m;
        for (int64_t k = 0; k < m; ++k) {
            ens.push_back(trip[(size_t)(4 * k)]); dlc.push_back(trip[(size_t)(4 * k + 1)]); nlc.push_back(trip[(size_t)(4 * k + 2)]);
            const size_t y = ens.size();
            double s = 0.0; for (double v : ens) s += v;
            mean = s / (double)y;                                                       // seqMain.m:180
            cov = 0.0;
            if (y > 1) {                                                                // :183-185, std = sample standard deviation
                double ss = 0.0; for (double v : ens) ss += (v - mean) * (v - mean);
                const double sd = std::sqrt(ss / (double)(y - 1));
                cov = sd / (mean * std::sqrt((double)y));                               // all years so far without curtailment: 0 / 0 = NaN, as the reference (:184)
            }
            if (o->results_year) { relmc_seq_year& Y = o->results_year[y - 1]; Y.ens = ens.back(); Y.dlc = dlc.back(); Y.nlc = nlc.back(); Y.n_contingency = (int64_t)trip[(size_t)(4 * k + 3)]; }
            if (o->cum_eens) o->cum_eens[y - 1] = mean;
            if (o->cum_cov) o->cum_cov[y - 1] = cov;
            n_cont += (int64_t)trip[(size_t)(4 * k + 3)];
            if (y > 1 && cov < o->cov_threshold && cov > 0.0) { stop = true; used = k + 1; break; }      // :194
        }
        if (used < m) {
            // the reference stops inside this batch: its accumulators (seqMain.m:146-159) hold the years up to the stopping year only
            const int64_t hi = done + used;
            const int64_t cnt2 = (lo + cnt < hi ? lo + cnt : hi) - lo;
            // the discarded pass leaves no trace in the bookkeeping: kernel_seconds and relmc_retry_stats do not depend on batch_years
            ctx->retry_units = ru0; ctx->retry_converged = rcv0; ctx->retry_dense_units = rd0; ctx->retry_dense_converged = rdc0; ctx->retry_overflow = ro0;
            kernel_ms = kernel_ms0;
            rc = eval(lo, cnt2 > 0 ? cnt2 : 0, &acc);
        }
        if (R > 1) {
            const std::string e2 = ctx->err;
            if (rc != RELMC_OK) { relmc_acc_zero(&acc); acc.n_nonconverged = -((int64_t)1 << 40); }
            const int rc2 = relmc_comm_allreduce_acc(ctx, &acc);
            if (rc != RELMC_OK) { ctx->err = e2; return rc; }
            if (rc2) return rc2;
            if (acc.n_nonconverged < 0) return fail(ctx, RELMC_ERR_HIP, "relmc_seq_run: another rank failed to re-evaluate its years up to the stopping year");
        } else if (rc != RELMC_OK) return rc;
        relmc_acc_merge(&res->acc, &acc);
        done += used;
    }
    const int64_t Y = (int64_t)ens.size();
    res->final_year = (int32_t)Y; res->converged = stop ? 1 : 0;
    res->eens = mean; res->cov = cov;
    double sd = 0.0, sn = 0.0; for (int64_t k = 0; k < Y; ++k) { sd += dlc[(size_t)k]; sn += nlc[(size_t)k]; }
    res->lole = sd / (double)Y; res->lolf = sn / (double)Y;                             // :212-213
    res->plc = sd / ((double)Y * (double)ctx->hseq.hpy);
    res->n_contingency = n_cont;
    for (int i = 0; i < ctx->nb; ++i) res->nodal_eens_avg[i] = res->acc.sum_nodal[i] / (double)Y;                                        // :218
    for (int k = 0; k < ctx->ncomp; ++k) res->comp_importance[k] = res->acc.n_fail ? (double)res->acc.comp_fail[k] / (double)res->acc.n_fail : 0.0;   // :233
    res->kernel_seconds = kernel_ms * 1e-3;
    res->wall_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    ctx->last_kernel_ms = kernel_ms;
    return RELMC_OK;
}

// ---- HL1 copper sheet: PowerSystemAdequacy.jl:169-208 --------------------------------------------------
int32_t relmc_hl1_load(relmc_ctx* ctx, int32_t ngen, const double* capacity_mw, const double* for_rate, int32_t nhours,
                       const double* hourly_load_mw)
{
    if (!ctx) return RELMC_ERR_INVALID;
    if (!capacity_mw || !for_rate || !hourly_load_mw || ngen < 1 || nhours < 1) return fail(ctx, RELMC_ERR_INVALID, "relmc_hl1_load: bad arguments");
    if (ngen > NCOMPMAX) return fail(ctx, RELMC_ERR_UNSUPPORTED, "relmc_hl1_load: more than 128 units");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    Hl1Case h; std::memset(&h, 0, sizeof(h));
    h.ngen = ngen; h.nhours = nhours;
    for (int g = 0; g < ngen; ++g) {
        double t = std::floor(for_rate[g] * 4294967296.0);
        if (!(t > 0)) t = 0;
        if (t > 4294967295.0) t = 4294967295.0;
        h.thr[g] = (uint32_t)t; h.cap[g] = capacity_mw[g];
    }
    std::vector<double> sorted(hourly_load_mw, hourly_load_mw + nhours), suffix(nhours + 1, 0.0);
    std::sort(sorted.begin(), sorted.end());
    for (int k = nhours - 1; k >= 0; --k) suffix[k] = suffix[k + 1] + sorted[k];
    if (ctx->dsorted) (void)hipFree(ctx->dsorted);
    if (ctx->dsuffix) (void)hipFree(ctx->dsuffix);
    ctx->dsorted = ctx->dsuffix = nullptr;
    if (!ctx->dhl1) HIP_TRY(ctx, hipMalloc(&ctx->dhl1, sizeof(Hl1Case)));
    HIP_TRY(ctx, hipMalloc(&ctx->dsorted, sizeof(double) * nhours));
    HIP_TRY(ctx, hipMalloc(&ctx->dsuffix, sizeof(double) * (nhours + 1)));
    HIP_TRY(ctx, hipMemcpy(ctx->dhl1, &h, sizeof(h), hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(ctx->dsorted, sorted.data(), sizeof(double) * nhours, hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(ctx->dsuffix, suffix.data(), sizeof(double) * (nhours + 1), hipMemcpyHostToDevice));
    ctx->hl1_hours = nhours; ctx->has_hl1 = true;
    return RELMC_OK;
}

int32_t relmc_hl1_nsq(relmc_ctx* ctx, uint64_t seed, uint64_t first_index, int64_t n, relmc_hl1_acc* acc, double* iter_lole_host,
                      double* iter_eue_host)
{
    if (!ctx) return RELMC_ERR_INVALID;
    if (!ctx->has_hl1) return fail(ctx, RELMC_ERR_NO_CASE, "relmc_hl1_nsq: relmc_hl1_load has not been called");
    if (n < 0 || !acc) return fail(ctx, RELMC_ERR_INVALID, "relmc_hl1_nsq: bad arguments");
    std::memset(acc, 0, sizeof(*acc));
    if (n == 0) return RELMC_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int64_t blocks = (n + 255) / 256;
    if (blocks > (int64_t)ctx->num_cu * 8) blocks = (int64_t)ctx->num_cu * 8;
    // per-iteration outputs and block partials live in the context between calls (three hipMalloc / hipFree pairs were 0.5 ms of a 0.8 ms call)
    if ((iter_lole_host || iter_eue_host) && n > ctx->h1_cap) {
        if (ctx->h1_lole) (void)hipFree(ctx->h1_lole);
        if (ctx->h1_eue) (void)hipFree(ctx->h1_eue);
        ctx->h1_lole = ctx->h1_eue = nullptr; ctx->h1_cap = 0;
        HIP_TRY(ctx, hipMalloc(&ctx->h1_lole, sizeof(double) * (size_t)n));
        HIP_TRY(ctx, hipMalloc(&ctx->h1_eue, sizeof(double) * (size_t)n));
        ctx->h1_cap = n;
    }
    if (blocks > ctx->h1_part_cap) {
        if (ctx->h1_part) (void)hipFree(ctx->h1_part);
        ctx->h1_part = nullptr; ctx->h1_part_cap = 0;
        HIP_TRY(ctx, hipMalloc(&ctx->h1_part, sizeof(double) * 4 * (size_t)blocks));
        ctx->h1_part_cap = blocks;
    }
    double* const dl = iter_lole_host ? ctx->h1_lole : nullptr; double* const de = iter_eue_host ? ctx->h1_eue : nullptr; double* const dpart = ctx->h1_part;
    int rc = RELMC_OK;
    (void)hipEventRecord(ctx->ev0, ctx->stream);
    hipLaunchKernelGGL(relmc_hl1_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, ctx->dhl1, ctx->dsorted, ctx->dsuffix, seed,
                       first_index, n, dl, de, dpart);
    (void)hipEventRecord(ctx->ev1, ctx->stream);
    std::vector<double> part((size_t)4 * blocks);
    if (hipGetLastError() != hipSuccess || hipMemcpyAsync(part.data(), dpart, sizeof(double) * 4 * blocks, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess)
        rc = fail(ctx, RELMC_ERR_HIP, "relmc_hl1_nsq: launch failed");
    if (rc == RELMC_OK && iter_lole_host && hipMemcpyAsync(iter_lole_host, dl, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = fail(ctx, RELMC_ERR_HIP, "relmc_hl1_nsq: D2H failed");
    if (rc == RELMC_OK && iter_eue_host && hipMemcpyAsync(iter_eue_host, de, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = fail(ctx, RELMC_ERR_HIP, "relmc_hl1_nsq: D2H failed");
    if (rc == RELMC_OK && finish_timing(ctx) != RELMC_OK) rc = fail(ctx, RELMC_ERR_HIP, "relmc_hl1_nsq: synchronisation failed");
    if (rc) return rc;
    acc->n = n;
    for (int64_t b = 0; b < blocks; ++b) { acc->sum_lole += part[4 * b]; acc->sum_eue += part[4 * b + 1]; acc->sum_lole2 += part[4 * b + 2]; acc->sum_eue2 += part[4 * b + 3]; }
    return RELMC_OK;
}

}  // extern "C"
