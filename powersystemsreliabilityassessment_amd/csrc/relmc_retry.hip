// relmc_retry.hip — second chances for the units the primary static elimination order does not converge on: the kernel's list of such
// units, their re-evaluation under further static orders and, last, under a dense partially pivoted solve (what MATLAB's `\` does under
// MIPS, mc_simulation.m:41).  DESIGN.md 6.3.
#include <algorithm>
#include <cstring>

#include "relmc_ctx.h"

namespace relmc_host {

void retry_free(relmc_ctx* ctx)
{
    for (void* p : {(void*)ctx->rkeys, (void*)ctx->rdns, (void*)ctx->rmeta, (void*)ctx->rnodal, (void*)ctx->rscale, (void*)ctx->dfail, (void*)ctx->dfail_count,
                    (void*)ctx->ddense}) if (p) (void)hipFree(p);
    ctx->rkeys = nullptr; ctx->rdns = nullptr; ctx->rmeta = nullptr; ctx->rnodal = nullptr; ctx->rscale = nullptr; ctx->rcap = 0; ctx->rnb = 0;
    ctx->dfail = nullptr; ctx->dfail_count = nullptr; ctx->fail_cap = 0; ctx->fail_dirty = false;
    ctx->ddense = nullptr; ctx->dense_bytes = 0;
}

// ---- second chance for the units the primary elimination order does not converge on ---------------------------------------------
// The block elimination runs in an order fixed per case; on a few states (6.7e-7 of the RTS-96 scenarios, 4e-10 on RTS-24) that order
// meets a stiff line next to a bus with an interior injection and the Newton steps of the last iterations lose their digits (DESIGN.md
// 6.3).  Which states depends on the order: of the 67 such RTS-96 states in 1e8 samples, 66 converge under the same elimination rule with
// the ties broken the other way (same pass counts).  The kernel lists the units it ends non-converged instead of accumulating them; they
// are evaluated again here under that second order and their results take the place of the first attempt's.
// List capacity: 4096 + 1/256 of the units of the call.  relmc_case_load's calibration leaves a primary order in place only if it fails on
// at most 0.1 % of a probe sample, so a list of 0.39 % + 4096 entries does not overflow on a calibrated case; if it does anyway the fused
// path grows the list and evaluates the chunk again (the launch is deterministic), the other paths count the units that kept their
// first-attempt results in relmc_retry_overflow.
uint32_t fail_cap_for(int64_t call_units)
{
    // steady-state size: 4096 + 1/256 of the call's units, at most 2^20 entries (48 MB; a 1e9-sample call used to hold 190 MB for a list that a
    // calibrated case fills to a few hundred entries).  The fused path grows the list up to kFailCapMax and re-runs the chunk if it overflows.
    const int64_t c = (int64_t)kFailCapMin + call_units / 256;
    return c > (int64_t)kFailCapSteady ? kFailCapSteady : (uint32_t)c;
}
int fail_list_ensure(relmc_ctx* ctx, uint32_t cap)
{
    if (!ctx->dfail_count) {
        HIP_TRY(ctx, hipMalloc(&ctx->dfail_count, sizeof(uint32_t)));
        HIP_TRY(ctx, hipMemsetAsync(ctx->dfail_count, 0, sizeof(uint32_t), ctx->stream));
    }
    if (cap <= ctx->fail_cap) return RELMC_OK;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->dfail) (void)hipFree(ctx->dfail);
    ctx->dfail = nullptr; ctx->fail_cap = 0;
    HIP_TRY(ctx, hipMalloc(&ctx->dfail, sizeof(FailRec) * (size_t)cap));
    ctx->fail_cap = cap;
    return RELMC_OK;
}


int alt_ensure(relmc_ctx* ctx, int v)          // v = 0, 1: which further order
{
    if (ctx->alt_state[v]) return ctx->alt_state[v] > 0 ? RELMC_OK : RELMC_ERR_UNSUPPORTED;
    ctx->alt_state[v] = -1;
    if (!ctx->case_copy.valid) return RELMC_ERR_UNSUPPORTED;
    const std::string keep = ctx->err;
    const int rc = case_load_image(ctx, &ctx->case_copy.d, v + 1);
    ctx->err = keep;                               // a case whose further order does not fit simply has one attempt less
    if (rc == RELMC_OK) ctx->alt_state[v] = 1;
    return rc;
}

// arms the kernel's list for the next launch(es); `reset` zeroes the count (first launch of a call) and sizes the list for the
// `call_units` units all launches of the call evaluate together
int fail_arm(relmc_ctx* ctx, EvalArgs& a, int64_t unit_base, bool reset, int64_t call_units)
{
    a.fail_list = nullptr; a.fail_count = nullptr; a.fail_cap = 0; a.unit_base = unit_base;
    if (ctx->sw.no_retry) return RELMC_OK;      // diagnosis switch: the first attempt's results as they are
    // (a case whose further static orders do not fit the tile is armed all the same: the dense pivoted level needs no second image)
    if (reset || !ctx->dfail) {
        const uint32_t want = fail_cap_for(call_units);
        const int rc = fail_list_ensure(ctx, want > ctx->fail_cap ? want : ctx->fail_cap);
        if (rc) return rc;
    }
    // the count is zero whenever a call has collected its list (fail_retry zeroes it after a non-empty one), so the common case costs no
    // memset launch; only a call that was abandoned between arming and collecting leaves it to be cleared here
    if (reset && ctx->fail_dirty) { HIP_TRY(ctx, hipMemsetAsync(ctx->dfail_count, 0, sizeof(uint32_t), ctx->stream)); }
    if (reset) ctx->fail_dirty = true;
    a.fail_list = ctx->dfail; a.fail_count = ctx->dfail_count; a.fail_cap = ctx->fail_cap;
    return RELMC_OK;
}

// units listed by the completed launches of the call (may exceed the capacity: the excess kept its first-attempt results)
int fail_listed(relmc_ctx* ctx, uint32_t* cnt)
{
    *cnt = 0;
    if (!ctx->dfail_count) return RELMC_OK;
    HIP_TRY(ctx, hipMemcpy(cnt, ctx->dfail_count, sizeof(*cnt), hipMemcpyDeviceToHost));
    return RELMC_OK;
}

// After the launches of a call have completed: the listed units (ascending), evaluated under the second order.  `scale` (optional) maps a
// unit to its load scale factor.  out.rec is empty when nothing was listed.  Adds the retry kernel's time to *ms.
int fail_retry(relmc_ctx* ctx, const relmc_solver_opts& o, double fail_threshold, const ScaleFn* scale_fn, RetryOut& out, double* ms, const uint32_t* known_count)
{
    const bool have_scale = scale_fn != nullptr;
    auto scale = [&](unsigned long long u) { return (*scale_fn)(u); };
    out.rec.clear();
    if (!ctx->dfail_count) return RELMC_OK;
    uint32_t cnt = 0;
    if (known_count) cnt = *known_count;            // the caller read the count with its results (one synchronisation)
    else HIP_TRY(ctx, hipMemcpy(&cnt, ctx->dfail_count, sizeof(cnt), hipMemcpyDeviceToHost));
    ctx->fail_dirty = false;
    if (cnt == 0) return RELMC_OK;
    HIP_TRY(ctx, hipMemset(ctx->dfail_count, 0, sizeof(uint32_t)));
    if (cnt > ctx->fail_cap) {                               // the units beyond the list were accumulated by the kernel as they were
        ctx->retry_overflow += (int64_t)(cnt - ctx->fail_cap);
        cnt = ctx->fail_cap;
    }
    out.rec.resize(cnt);
    HIP_TRY(ctx, hipMemcpy(out.rec.data(), ctx->dfail, sizeof(FailRec) * cnt, hipMemcpyDeviceToHost));
    std::sort(out.rec.begin(), out.rec.end(), [](const FailRec& x, const FailRec& y) { return x.unit < y.unit; });
    const int ow = ctx->tile == 0 ? Tile24::OW : Tile96::OW;
    const size_t nb = (size_t)ctx->nb;
    if ((int64_t)cnt > ctx->rcap || ctx->nb > ctx->rnb) {    // scratch rows of the re-evaluation, sized by what was listed (twice: the third order's compact rows)
        for (void* p : {(void*)ctx->rkeys, (void*)ctx->rdns, (void*)ctx->rmeta, (void*)ctx->rnodal, (void*)ctx->rscale}) if (p) (void)hipFree(p);
        ctx->rkeys = nullptr; ctx->rdns = nullptr; ctx->rmeta = nullptr; ctx->rnodal = nullptr; ctx->rscale = nullptr; ctx->rcap = 0; ctx->rnb = 0;
        size_t rc2 = kFailCapMin; while (rc2 < cnt) rc2 *= 2;
        HIP_TRY(ctx, hipMalloc(&ctx->rkeys, sizeof(uint32_t) * rc2 * 2 * 8));
        HIP_TRY(ctx, hipMalloc(&ctx->rdns, sizeof(double) * rc2 * 2));
        HIP_TRY(ctx, hipMalloc(&ctx->rmeta, sizeof(int32_t) * rc2 * 2));
        HIP_TRY(ctx, hipMalloc(&ctx->rnodal, sizeof(double) * rc2 * 2 * nb));
        HIP_TRY(ctx, hipMalloc(&ctx->rscale, sizeof(double) * rc2 * 2));
        ctx->rcap = (int64_t)rc2; ctx->rnb = ctx->nb;
    }
    std::vector<uint32_t> keys((size_t)cnt * ow);
    for (uint32_t r = 0; r < cnt; ++r) for (int q = 0; q < ow; ++q) keys[(size_t)r * ow + q] = out.rec[r].mask[q];
    HIP_TRY(ctx, hipMemcpy(ctx->rkeys, keys.data(), sizeof(uint32_t) * keys.size(), hipMemcpyHostToDevice));
    if (have_scale) {
        std::vector<double> sc(cnt);
        for (uint32_t r = 0; r < cnt; ++r) sc[r] = scale(out.rec[r].unit);
        HIP_TRY(ctx, hipMemcpy(ctx->rscale, sc.data(), sizeof(double) * cnt, hipMemcpyHostToDevice));
    }
    out.dns.resize(cnt); out.meta.resize(cnt); out.nodal.resize((size_t)cnt * nb);
    // first the whole list under the second order, then whatever is still non-converged under the third (the sets of states the three
    // orders fail on were disjoint on the 67 RTS-96 states of the fixture).  A case without further orders (they do not fit the tile)
    // repeats the primary one, so that the callers' bookkeeping is one path.
    const bool dense_first = ctx->sw.retry_dense_first;      // tests: the listed units straight to the dense pivoted solve
    // level 0, 1: the further static orders; level 2: the dense, partially pivoted solve (what MATLAB's `\` does under mips) for whatever
    // no static order converged on
    for (int level = dense_first ? relmc_ctx::kAlt : 0; level <= relmc_ctx::kAlt; ++level) {
        const bool dense = level == relmc_ctx::kAlt;
        const bool have = dense || alt_ensure(ctx, level) == RELMC_OK;
        if (level > 0 && !have) continue;
        EvalArgs a = make_args(o);
        a.fail_threshold = fail_threshold;
        a.load_scale = have_scale ? ctx->rscale : nullptr;
        int rows = 0, rc;
        if (level == 0 || (dense && dense_first)) {
            a.n = (int64_t)cnt; a.memo_keys = ctx->rkeys; a.db_first = 0; a.dns = ctx->rdns; a.status = ctx->rmeta; a.nodal = ctx->rnodal;
            rc = dense ? launch_eval(ctx, 6, a, &rows) : launch_eval(ctx, 4, a, &rows, nullptr, nullptr, have ? 1 : 0);
            if (rc) return rc;
            if (dense) ctx->retry_dense_units += cnt;
            const double before = ctx->last_kernel_ms;
            rc = finish_timing(ctx);
            if (rc) return rc;
            if (ms) *ms += ctx->last_kernel_ms;
            ctx->last_kernel_ms = before;
            HIP_TRY(ctx, hipMemcpy(out.meta.data(), ctx->rmeta, sizeof(int32_t) * cnt, hipMemcpyDeviceToHost));
            if (dense) for (uint32_t r = 0; r < cnt; ++r) if ((out.meta[r] & 3) == 0 || (out.meta[r] & 3) == 3) ctx->retry_dense_converged += 1;
        } else {
            // what the second order left non-converged, compacted behind the list's rows and evaluated under the third order in ONE launch
            // (round 2 launched once per unit); the results are copied over the rows they belong to
            std::vector<uint32_t> idx;
            for (uint32_t r = 0; r < cnt; ++r) if ((out.meta[r] & 3) == 1 || (out.meta[r] & 3) == 2) idx.push_back(r);
            if (idx.empty()) break;
            if (dense) ctx->retry_dense_units += (int64_t)idx.size();
            const size_t m = idx.size(), base = (size_t)ctx->rcap;
            std::vector<uint32_t> k2(m * ow); std::vector<double> s2(m);
            for (size_t q = 0; q < m; ++q) { for (int w = 0; w < ow; ++w) k2[q * ow + w] = out.rec[idx[q]].mask[w]; if (have_scale) s2[q] = scale(out.rec[idx[q]].unit); }
            HIP_TRY(ctx, hipMemcpy(ctx->rkeys + base * ow, k2.data(), sizeof(uint32_t) * k2.size(), hipMemcpyHostToDevice));
            if (have_scale) HIP_TRY(ctx, hipMemcpy(ctx->rscale + base, s2.data(), sizeof(double) * m, hipMemcpyHostToDevice));
            a.n = (int64_t)m; a.memo_keys = ctx->rkeys; a.db_first = (int64_t)base; a.dns = ctx->rdns; a.status = ctx->rmeta; a.nodal = ctx->rnodal;
            a.load_scale = have_scale ? ctx->rscale + base : nullptr;
            rc = dense ? launch_eval(ctx, 6, a, &rows) : launch_eval(ctx, 4, a, &rows, nullptr, nullptr, level + 1);
            if (rc) return rc;
            const double before = ctx->last_kernel_ms;
            rc = finish_timing(ctx);
            if (rc) return rc;
            if (ms) *ms += ctx->last_kernel_ms;
            ctx->last_kernel_ms = before;
            for (size_t q = 0; q < m; ++q) {
                HIP_TRY(ctx, hipMemcpyAsync(ctx->rdns + idx[q], ctx->rdns + base + q, sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
                HIP_TRY(ctx, hipMemcpyAsync(ctx->rmeta + idx[q], ctx->rmeta + base + q, sizeof(int32_t), hipMemcpyDeviceToDevice, ctx->stream));
                HIP_TRY(ctx, hipMemcpyAsync(ctx->rnodal + (size_t)idx[q] * nb, ctx->rnodal + (base + q) * nb, sizeof(double) * nb, hipMemcpyDeviceToDevice, ctx->stream));
            }
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            if (dense) {
                std::vector<int32_t> m2(m);
                HIP_TRY(ctx, hipMemcpy(m2.data(), ctx->rmeta + base, sizeof(int32_t) * m, hipMemcpyDeviceToHost));
                for (size_t q = 0; q < m; ++q) if ((m2[q] & 3) == 0 || (m2[q] & 3) == 3) ctx->retry_dense_converged += 1;
            } else {
                HIP_TRY(ctx, hipMemcpy(out.meta.data(), ctx->rmeta, sizeof(int32_t) * cnt, hipMemcpyDeviceToHost));      // what is still left, for the next level
            }
        }
    }
    HIP_TRY(ctx, hipMemcpy(out.dns.data(), ctx->rdns, sizeof(double) * cnt, hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(out.meta.data(), ctx->rmeta, sizeof(int32_t) * cnt, hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(out.nodal.data(), ctx->rnodal, sizeof(double) * cnt * nb, hipMemcpyDeviceToHost));
    ctx->retry_units += cnt;
    for (uint32_t r = 0; r < cnt; ++r) if ((out.meta[r] & 3) == 0 || (out.meta[r] & 3) == 3) ctx->retry_converged += 1;
    return RELMC_OK;
}

// the accumulators of one unit (what the kernel's output section adds for it), count-weighted
void acc_add_unit(relmc_acc* acc, const FailRec& rec, double dns, int32_t meta, const double* nodal, int nb, int ncomp, double fail_threshold)
{
    const long long w = (long long)rec.weight;
    const int status = meta & 3, it = (int)((uint32_t)meta >> 8);
    const bool fail = dns > fail_threshold;
    acc->n += w;
    if (fail) acc->n_fail += w;
    if (status == 3) acc->n_singular += w;
    if (status == 1 || status == 2) acc->n_nonconverged += w;
    if (meta & 4) acc->n_infeasible += w;
    acc->sum_iters += (long long)it * w;
    if (dns != 0.0) { acc->sum_dns += (double)w * dns; acc->sum_dns2 += (double)w * dns * dns; }
    if (fail) for (int k = 0; k < ncomp; ++k) if ((rec.mask[k >> 5] >> (k & 31)) & 1u) acc->comp_fail[k] += w;
    for (int i = 0; i < nb; ++i) acc->sum_nodal[i] += (double)w * nodal[i];
}

}  // namespace relmc_host

using namespace relmc_host;

extern "C" {

int32_t relmc_retry_stats(const relmc_ctx* ctx, int64_t* units_out, int64_t* converged_out)
{
    if (!ctx) return RELMC_ERR_INVALID;
    if (units_out) *units_out = ctx->retry_units;
    if (converged_out) *converged_out = ctx->retry_converged;
    return RELMC_OK;
}

int32_t relmc_retry_overflow(const relmc_ctx* ctx, int64_t* units_out)
{
    if (!ctx || !units_out) return RELMC_ERR_INVALID;
    *units_out = ctx->retry_overflow;
    return RELMC_OK;
}

// units that went to the dense pivoted last resort since the case was loaded, and how many of them it converged on
int32_t relmc_retry_dense_stats(const relmc_ctx* ctx, int64_t* units_out, int64_t* converged_out)
{
    if (!ctx) return RELMC_ERR_INVALID;
    if (units_out) *units_out = ctx->retry_dense_units;
    if (converged_out) *converged_out = ctx->retry_dense_converged;
    return RELMC_OK;
}

}  // extern "C"
