// relmc_core.hip — context lifetime, relmc_case_load (device tables + order calibration), the launcher of the evaluation kernels (the one
// translation unit that holds their device code), mc_sampling, and the estimator arithmetic of include/relmc.h.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>

#include "relmc_ctx.h"
#ifdef RELMC_DEV_SWITCHES
#include "relmc_dev_switches.h"
#endif
#include "relmc_kernels.hip"

static_assert(sizeof(relmc::DevAcc) == sizeof(relmc_acc), "device accumulator image must match relmc_acc");

namespace relmc_host {

const char* const kNoCtx = "relmc: null context";

int fail(relmc_ctx* ctx, int code, const std::string& msg)
{
    if (ctx) ctx->err = msg;
    return code;
}

EvalArgs make_args(const relmc_solver_opts& o)
{
    EvalArgs a;
    std::memset(&a, 0, sizeof(a));
    a.policy = o.singular_policy; a.max_it = o.max_it;
    a.feastol = o.feastol; a.gradtol = o.gradtol; a.comptol = o.comptol; a.costtol = o.costtol;
    a.xi = o.xi; a.sigma = o.sigma; a.z0 = o.z0; a.alpha_min = o.alpha_min; a.max_stepsize = o.max_stepsize;
    a.fail_threshold = 1e-4;                 // nsqMain.m:270
    return a;
}

namespace {
template <class TL>
int grid_for(relmc_ctx* ctx, int64_t n)
{
    const int64_t groups = ((n + TL::SPW - 1) / TL::SPW + TL::WPB - 1) / TL::WPB;
    int64_t g = (int64_t)ctx->num_cu * ctx->blocks_per_cu;
    if (g > groups) g = groups;
    if (g < 1) g = 1;
    return (int)g;
}

int ensure_partial(relmc_ctx* ctx, size_t bytes)
{
    if (bytes <= ctx->partial_bytes) return RELMC_OK;
    if (ctx->dpartial) (void)hipFree(ctx->dpartial);
    ctx->dpartial = nullptr; ctx->partial_bytes = 0;
    HIP_TRY(ctx, hipMalloc(&ctx->dpartial, bytes));
    ctx->partial_bytes = bytes;
    return RELMC_OK;
}

// launches the evaluation kernel of one tile; *rows_out = scenario rows holding partial accumulators
template <int MODE, class TL>
int launch_eval_t(relmc_ctx* ctx, EvalArgs& a, int* rows_out, hipEvent_t ev_start, hipEvent_t ev_stop, int alt)
{
    int blocks = grid_for<TL>(ctx, a.n);
    if (MODE == 6) {
        // dense last resort: [scenario rows of the grid][2 nb (2 nb + 1)] doubles of scratch; a small grid keeps it small (the units are few)
        if (blocks > 64) blocks = 64;
        const size_t n = 2 * (size_t)ctx->nb, stride = n * (n + 1), need = sizeof(double) * stride * (size_t)blocks * TL::WPB * TL::SPW;
        if (need > ctx->dense_bytes) {
            if (ctx->ddense) (void)hipFree(ctx->ddense);
            ctx->ddense = nullptr; ctx->dense_bytes = 0;
            HIP_TRY(ctx, hipMalloc(&ctx->ddense, need));
            ctx->dense_bytes = need;
        }
        a.dense = ctx->ddense; a.dense_stride = stride;
    }
    int rc = ensure_partial(ctx, sizeof(PartialT<TL>) * 64 * TL::WPB * (size_t)blocks);
    if (rc) return rc;
    a.partial = ctx->dpartial;
    a.scen_doubles = alt ? ctx->alt_scen_doubles[alt - 1] : ctx->scen_doubles;
    // the grid fills the device: first-dispatched and later wavefronts share every SIMD (see the kernel's priority balancing)
    a.prio_mode = blocks != ctx->num_cu * ctx->blocks_per_cu ? 0u : (ctx->blocks_per_cu == 2 ? 1u : (ctx->blocks_per_cu == 1 && TL::WPB >= 8 ? 2u : 0u));
    a.stash_off = alt ? ctx->alt_stash_off[alt - 1] : ctx->stash_off;
    a.case_bytes = (uint32_t)offsetof(DevCaseT<TL>, task);
#if defined(RELMC_PHASE_TIMING) || defined(RELMC_TRACE)
    if (!ctx->dtiming) HIP_TRY(ctx, hipMalloc(&ctx->dtiming, sizeof(unsigned long long) * 8 * 65536));
    a.timing = ctx->dtiming; ctx->timing_waves = blocks * TL::WPB;
#else
    a.timing = nullptr;
#endif
    HIP_TRY(ctx, hipEventRecord(ev_start ? ev_start : ctx->ev0, ctx->stream));
    hipLaunchKernelGGL((relmc_eval_kernel<MODE, TL>), dim3(blocks), dim3(64 * TL::WPB), alt ? ctx->alt_lds_bytes[alt - 1] : ctx->lds_bytes, ctx->stream,
                       reinterpret_cast<const DevCaseT<TL>*>(alt ? ctx->dcase_alt[alt - 1] : ctx->dcase), a);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventRecord(ev_stop ? ev_stop : ctx->ev1, ctx->stream));
    *rows_out = blocks * TL::WPB * TL::SPW;
    return RELMC_OK;
}

template <class TL>
int launch_eval_mode(relmc_ctx* ctx, int mode, EvalArgs& a, int* rows_out, hipEvent_t e0, hipEvent_t e1, int alt)
{
    switch (mode) {
        case 0: return launch_eval_t<0, TL>(ctx, a, rows_out, e0, e1, alt);
        case 1: return launch_eval_t<1, TL>(ctx, a, rows_out, e0, e1, alt);
        case 2: return launch_eval_t<2, TL>(ctx, a, rows_out, e0, e1, alt);
        case 3: return launch_eval_t<3, TL>(ctx, a, rows_out, e0, e1, alt);
        case 4: return launch_eval_t<4, TL>(ctx, a, rows_out, e0, e1, alt);
        case 5: return launch_eval_t<5, TL>(ctx, a, rows_out, e0, e1, alt);
        case 6: return launch_eval_t<6, TL>(ctx, a, rows_out, e0, e1, alt);
        case 7: return launch_eval_t<7, TL>(ctx, a, rows_out, e0, e1, alt);
        default: return fail(ctx, RELMC_ERR_INVALID, "launch_eval: unknown mode");
    }
}

// every instantiation of the tile may use `bytes` of dynamic LDS (the case tables + one workspace per scenario row)
template <class TL>
int eval_set_lds(relmc_ctx* ctx, int bytes)
{
    for (const void* f : {reinterpret_cast<const void*>(&relmc_eval_kernel<0, TL>), reinterpret_cast<const void*>(&relmc_eval_kernel<1, TL>),
                          reinterpret_cast<const void*>(&relmc_eval_kernel<2, TL>), reinterpret_cast<const void*>(&relmc_eval_kernel<3, TL>),
                          reinterpret_cast<const void*>(&relmc_eval_kernel<4, TL>), reinterpret_cast<const void*>(&relmc_eval_kernel<5, TL>),
                          reinterpret_cast<const void*>(&relmc_eval_kernel<6, TL>), reinterpret_cast<const void*>(&relmc_eval_kernel<7, TL>)})
        HIP_TRY(ctx, hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    return RELMC_OK;
}
}  // namespace

int launch_eval(relmc_ctx* ctx, int mode, EvalArgs& a, int* rows_out, hipEvent_t ev_start, hipEvent_t ev_stop, int alt)
{
    if (ctx->tile == 0) return launch_eval_mode<Tile24>(ctx, mode, a, rows_out, ev_start, ev_stop, alt);
    return launch_eval_mode<Tile96>(ctx, mode, a, rows_out, ev_start, ev_stop, alt);
}

// deterministic reduction of the partial records into the device image of relmc_acc
int launch_finalize(relmc_ctx* ctx, int rows)
{
    if (ctx->tile == 0)
        hipLaunchKernelGGL(relmc_finalize_kernel<Tile24>, dim3(FIN_ITEMS), dim3(FIN_THREADS), 0, ctx->stream, reinterpret_cast<const DevCaseT<Tile24>*>(ctx->dcase),
                           reinterpret_cast<const PartialT<Tile24>*>(ctx->dpartial), rows, ctx->dacc);
    else
        hipLaunchKernelGGL(relmc_finalize_kernel<Tile96>, dim3(FIN_ITEMS), dim3(FIN_THREADS), 0, ctx->stream, reinterpret_cast<const DevCaseT<Tile96>*>(ctx->dcase),
                           reinterpret_cast<const PartialT<Tile96>*>(ctx->dpartial), rows, ctx->dacc);
    HIP_TRY(ctx, hipGetLastError());
    return RELMC_OK;
}

int finish_timing(relmc_ctx* ctx)
{
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    float ms = 0.f;
    HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    ctx->last_kernel_ms = ms;
    return RELMC_OK;
}

namespace {
template <class TL>
int case_load_impl(relmc_ctx* ctx, const relmc_case_desc* d, DevCaseT<TL>& C, int order_variant = 0)
{
    const bool alt = order_variant != 0;            // the alternate image: geometry into the alt_* fields, nothing else of the context changes
    constexpr int WPB = TL::WPB;
    SymGeom geom;
    {
        SymOpts so = sym_opts_default();
        so.order_hint = ctx->order_hint.data(); so.n_hint = (int)ctx->order_hint.size();
        std::string err;
        const int rc = case_symbolic<TL>(d, C, order_variant, so, geom, err);
        if (rc) return fail(ctx, rc, err);
    }
    const uint32_t stash_off = geom.stash_off, scen = geom.scen_doubles, lds_bytes = geom.lds_bytes;
    const int nb = d->nb, ng = d->ng, nl = d->nl, ncomp = ng + nl;
    if (!alt) { ctx->conflict_before = geom.conflict_before; ctx->conflict_after = geom.conflict_after; }
    else { ctx->alt_conflict_before[order_variant - 1] = geom.conflict_before; ctx->alt_conflict_after[order_variant - 1] = geom.conflict_after; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (alt) {
        const int v = order_variant - 1;
        if (!ctx->dcase_alt[v]) HIP_TRY(ctx, hipMalloc(&ctx->dcase_alt[v], sizeof(DevCaseT<Tile96>) > sizeof(DevCaseT<Tile24>) ? sizeof(DevCaseT<Tile96>) : sizeof(DevCaseT<Tile24>)));
        ctx->alt_stash_off[v] = stash_off; ctx->alt_scen_doubles[v] = scen; ctx->alt_lds_bytes[v] = lds_bytes;
        uint32_t most = ctx->lds_bytes;
        for (int q = 0; q < relmc_ctx::kAlt; ++q) if (ctx->alt_lds_bytes[q] > most) most = ctx->alt_lds_bytes[q];
        { const int rc = eval_set_lds<TL>(ctx, (int)most); if (rc) return rc; }
        HIP_TRY(ctx, hipMemcpyAsync(ctx->dcase_alt[v], &C, sizeof(C), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        return RELMC_OK;
    }
    ctx->stash_off = stash_off; ctx->scen_doubles = scen; ctx->lds_bytes = lds_bytes;
    { const int rc = eval_set_lds<TL>(ctx, (int)ctx->lds_bytes); if (rc) return rc; }
    int bpc = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, relmc_eval_kernel<0, TL>, 64 * WPB, ctx->lds_bytes) != hipSuccess || bpc < 1) bpc = 1;
    ctx->blocks_per_cu = bpc;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->dcase, &C, sizeof(C), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->nb = nb; ctx->ng = ng; ctx->nl = nl; ctx->ncomp = ncomp;
    ctx->has_seq = false;
    ctx->has_case = true;
    return RELMC_OK;
}
}  // namespace

int case_load_image(relmc_ctx* ctx, const relmc_case_desc* d, int order_variant)
{
    if (ctx->tile == 0) { auto C = std::make_unique<DevCaseT<Tile24>>(); return case_load_impl<Tile24>(ctx, d, *C, order_variant); }
    auto C = std::make_unique<DevCaseT<Tile96>>();
    return case_load_impl<Tile96>(ctx, d, *C, order_variant);
}

namespace {
// Which static order should run first?  relmc_case_load evaluates a fixed sample of states under the primary order and counts the
// non-converged ones.  None (RTS-24, RTS-96: the rates are 4e-10 and 6.7e-7) keeps everything as it is; a case on which the primary order
// fails often (a 7-bus network of the fuzz run: 6 % of its states) gets the two further orders built and probed on the same sample, and
// the one with the fewest failures becomes the primary, the others the retry levels.
constexpr int64_t kProbeSamples = 8192;
int order_probe(relmc_ctx* ctx, int alt, int32_t* failures)
{
    relmc_solver_opts o; relmc_solver_opts_default(&o);
    EvalArgs a = make_args(o);
    a.seed = 0x5eedca5eull; a.first_index = 0; a.n = kProbeSamples;
    {
        const int rc0 = fail_list_ensure(ctx, fail_cap_for(kProbeSamples) > ctx->fail_cap ? fail_cap_for(kProbeSamples) : ctx->fail_cap);
        if (rc0) return rc0;
    }
    HIP_TRY(ctx, hipMemsetAsync(ctx->dfail_count, 0, sizeof(uint32_t), ctx->stream));
    a.fail_list = ctx->dfail; a.fail_count = ctx->dfail_count; a.fail_cap = ctx->fail_cap; a.unit_base = 0;
    int rows = 0;
    int rc = launch_eval(ctx, 5, a, &rows, nullptr, nullptr, alt);      // MODE 5 = MODE 0 under its own kernel name
    if (rc) return rc;
    uint32_t cnt = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&cnt, ctx->dfail_count, sizeof(cnt), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemset(ctx->dfail_count, 0, sizeof(uint32_t)));
    ctx->fail_dirty = false;
    *failures = (int32_t)cnt;
    return RELMC_OK;
}

int order_calibrate(relmc_ctx* ctx)
{
    ctx->order_primary = 0; ctx->order_probe[0] = ctx->order_probe[1] = ctx->order_probe[2] = -1;
    if (ctx->sw.no_retry) return RELMC_OK;
    int rc = order_probe(ctx, 0, &ctx->order_probe[0]);
    if (rc) return rc;
    if ((int64_t)ctx->order_probe[0] * 1000 <= kProbeSamples) return RELMC_OK;          // at most 0.1 %: the retry levels deal with those
    int best = 0;
    for (int v = 0; v < relmc_ctx::kAlt; ++v) {
        if (alt_ensure(ctx, v) != RELMC_OK) continue;
        rc = order_probe(ctx, v + 1, &ctx->order_probe[v + 1]);
        if (rc) return rc;
        if (ctx->order_probe[v + 1] < ctx->order_probe[best]) best = v + 1;
    }
    if (best != 0 && ctx->order_probe[best] * 2 <= ctx->order_probe[0]) {
        const int v = best - 1;                        // that image becomes the primary, the former primary takes its retry level
        std::swap(ctx->dcase, ctx->dcase_alt[v]);
        std::swap(ctx->scen_doubles, ctx->alt_scen_doubles[v]); std::swap(ctx->lds_bytes, ctx->alt_lds_bytes[v]); std::swap(ctx->stash_off, ctx->alt_stash_off[v]);
        std::swap(ctx->conflict_before, ctx->alt_conflict_before[v]); std::swap(ctx->conflict_after, ctx->alt_conflict_after[v]);
        // the host copy follows the image that runs: relmc_debug_schedule (and with it bench.py's operation count) describes the active schedule
        if (ctx->tile == 0) HIP_TRY(ctx, hipMemcpy(&ctx->hcase24, ctx->dcase, sizeof(ctx->hcase24), hipMemcpyDeviceToHost));
        else HIP_TRY(ctx, hipMemcpy(&ctx->hcase96, ctx->dcase, sizeof(ctx->hcase96), hipMemcpyDeviceToHost));
        ctx->order_primary = best;
        int bpc = 0; hipError_t e = hipSuccess;
        const int lds = (int)ctx->lds_bytes;
        const int most = lds > (int)ctx->alt_lds_bytes[v] ? lds : (int)ctx->alt_lds_bytes[v];
        if (ctx->tile == 0) {
            const int rc2 = eval_set_lds<Tile24>(ctx, most); if (rc2) return rc2;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, relmc_eval_kernel<0, Tile24>, 64 * Tile24::WPB, ctx->lds_bytes) == hipSuccess && bpc >= 1) ctx->blocks_per_cu = bpc;
        } else {
            const int rc2 = eval_set_lds<Tile96>(ctx, most); if (rc2) return rc2;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, relmc_eval_kernel<0, Tile96>, 64 * Tile96::WPB, ctx->lds_bytes) == hipSuccess && bpc >= 1) ctx->blocks_per_cu = bpc;
        }
        (void)e;
    }
    return RELMC_OK;
}
}  // namespace

}  // namespace relmc_host

using namespace relmc_host;

extern "C" {

const char* relmc_version(void) { return "relmc 0.9 (gfx950; DPP-row IPM tiles 16x4 and 64x1, sparse 2x2-block LDL' in LDS, static schedules with a tunable elimination order + dense pivoted last resort, device state database, zero-curtailment pre-screen, nsqMain and seqMain loops below the ABI with checkpoint stretches over N ranks, guarded collectives)"; }

const char* relmc_last_error(const relmc_ctx* ctx) { return ctx ? ctx->err.c_str() : kNoCtx; }

int32_t relmc_ctx_create(int32_t device_id, relmc_ctx** out)
{
    if (!out) return RELMC_ERR_INVALID;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return RELMC_ERR_NO_DEVICE;
    if (device_id < 0 || device_id >= ndev) return RELMC_ERR_INVALID;
    relmc_ctx* ctx = new (std::nothrow) relmc_ctx();
    if (!ctx) return RELMC_ERR_INVALID;
    ctx->device = device_id;
    hipDeviceProp_t prop;
    if (hipSetDevice(device_id) != hipSuccess || hipGetDeviceProperties(&prop, device_id) != hipSuccess ||
        hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess ||
        hipMalloc(&ctx->dcase, sizeof(DevCaseT<Tile96>) > sizeof(DevCaseT<Tile24>) ? sizeof(DevCaseT<Tile96>) : sizeof(DevCaseT<Tile24>)) != hipSuccess || hipMalloc(&ctx->dacc, sizeof(DevAcc)) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void**>(&ctx->hstage), sizeof(relmc_ctx::HostStage), hipHostMallocDefault) != hipSuccess) {
        relmc_ctx_destroy(ctx);          // releases whatever was created before the failure
        return RELMC_ERR_NO_DEVICE;
    }
    ctx->num_cu = prop.multiProcessorCount;
    ctx->blocks_per_cu = 1;
#ifdef RELMC_DEV_SWITCHES      // diagnosis builds only (csrc/Makefile: ablate/librelmc_dev.so); the default build has relmc_debug_set alone
    relmc_dev_switches_context(ctx->sw);
#endif
    *out = ctx;
    return RELMC_OK;
}

void relmc_ctx_destroy(relmc_ctx* ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->dpartial) (void)hipFree(ctx->dpartial);
    if (ctx->dcase) (void)hipFree(ctx->dcase);
    if (ctx->dacc) (void)hipFree(ctx->dacc);
    if (ctx->hstage) (void)hipHostFree(ctx->hstage);
    if (ctx->dtiming) (void)hipFree(ctx->dtiming);
    if (ctx->dhist) (void)hipFree(ctx->dhist);
    if (ctx->hhist) (void)hipHostFree(ctx->hhist);
    for (void* p : {ctx->dcase_alt[0], ctx->dcase_alt[1]}) if (p) (void)hipFree(p);
    retry_free(ctx);
    seq_free(ctx);
    screen_free(ctx);
    comm_free(ctx);
    pipe_free(ctx);
    memo_free(ctx);
    db_free(ctx);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

void relmc_solver_opts_default(relmc_solver_opts* o)
{
    if (!o) return;
    o->singular_policy = RELMC_REFERENCE_EMULATE;
    o->max_it = 150;
    o->feastol = 5e-6; o->gradtol = 1e-6; o->comptol = 1e-6; o->costtol = 1e-6;
    o->xi = 0.99995; o->sigma = 0.1; o->z0 = 1.0; o->alpha_min = 1e-8; o->max_stepsize = 1e10;
    o->screen = 0; o->reserved = 0;
}

void relmc_nsq_opts_default(relmc_nsq_opts* o)
{
    if (!o) return;
    std::memset(o, 0, sizeof(*o));
    o->beta_limit = 0.0017;       /* nsqMain.m:60 */
    o->max_samples = 100000;      /* nsqMain.m:61 */
    o->batch = 100;               /* nsqMain.m:62 */
    o->seed = 1;
    o->hours_per_year = 8760.0;   /* nsqMain.m:292 */
    relmc_solver_opts_default(&o->solver);
}

// Build the device tables from the plain case description: internal bus numbering = elimination
// order of the sparse block LDL' (level-then-min-fill, reference bus last), symbolic fill, the
// static task schedule the kernel interprets, incidence lists, thresholds.  Mirrors what
// nsqMain.m:42-167 prepares once before its Monte Carlo loop.
int32_t relmc_case_load(relmc_ctx* ctx, const relmc_case_desc* d)
{
    if (!ctx) return RELMC_ERR_INVALID;
    if (!d || !d->bus_pd || !d->inj_bus || !d->inj_pmin || !d->inj_pmax || !d->inj_cost || !d->br_from ||
        !d->br_to || !d->br_b || !d->br_rate || !d->unavail || !d->always_up)
        return fail(ctx, RELMC_ERR_INVALID, "relmc_case_load: null field in case description");
    const int nb = d->nb, ng = d->ng, nl = d->nl, nd = d->nd;
    if (nb < 1 || ng < 0 || nl < 0 || nd < 0 || d->ref_bus < 0 || d->ref_bus >= nb || !(d->base_mva > 0))
        return fail(ctx, RELMC_ERR_INVALID, "relmc_case_load: inconsistent sizes");
    ctx->has_case = false;
    db_free(ctx);                  // the state database belongs to the case it was filled for
    {   // the description is kept: the second elimination order (retry of non-converged units) is built from it when first needed
        auto& cc = ctx->case_copy;
        const int ninj = ng + nd, ncomp = ng + nl;
        cc.bus_pd.assign(d->bus_pd, d->bus_pd + nb); cc.inj_bus.assign(d->inj_bus, d->inj_bus + ninj);
        cc.inj_pmin.assign(d->inj_pmin, d->inj_pmin + ninj); cc.inj_pmax.assign(d->inj_pmax, d->inj_pmax + ninj); cc.inj_cost.assign(d->inj_cost, d->inj_cost + ninj);
        cc.br_from.assign(d->br_from, d->br_from + nl); cc.br_to.assign(d->br_to, d->br_to + nl); cc.br_b.assign(d->br_b, d->br_b + nl); cc.br_rate.assign(d->br_rate, d->br_rate + nl);
        cc.unavail.assign(d->unavail, d->unavail + ncomp); cc.always_up.assign(d->always_up, d->always_up + ncomp);
        cc.d = *d;
        cc.d.bus_pd = cc.bus_pd.data(); cc.d.inj_bus = cc.inj_bus.data(); cc.d.inj_pmin = cc.inj_pmin.data(); cc.d.inj_pmax = cc.inj_pmax.data(); cc.d.inj_cost = cc.inj_cost.data();
        cc.d.br_from = cc.br_from.data(); cc.d.br_to = cc.br_to.data(); cc.d.br_b = cc.br_b.data(); cc.d.br_rate = cc.br_rate.data();
        cc.d.unavail = cc.unavail.data(); cc.d.always_up = cc.always_up.data();
        cc.valid = true;
        retry_free(ctx);           // the scratch rows of the re-evaluation are sized for the case that was loaded (its bus count)
        ctx->alt_state[0] = ctx->alt_state[1] = 0; ctx->retry_units = 0; ctx->retry_converged = 0; ctx->retry_overflow = 0;
        ctx->retry_dense_units = 0; ctx->retry_dense_converged = 0;
    }
    // smallest tile that holds the case: 16-lane rows (four scenarios per wavefront) or one scenario per wavefront
    if (fits_tile24(d)) {
        ctx->tile = 0;
        int rc = case_load_impl<Tile24>(ctx, d, ctx->hcase24);
        ctx->order_hint.clear();                         // a hint is for one relmc_case_load
        if (rc == RELMC_OK) rc = screen_build(ctx, d);
        return rc ? rc : order_calibrate(ctx);
    }
    ctx->tile = 1;
    int rc = case_load_impl<Tile96>(ctx, d, ctx->hcase96);
    ctx->order_hint.clear();
    if (rc == RELMC_OK) rc = screen_build(ctx, d);
    return rc ? rc : order_calibrate(ctx);
}

int32_t relmc_case_order_hint(relmc_ctx* ctx, const int32_t* order, int32_t n)
{
    if (!ctx || n < 0 || (n > 0 && !order)) return RELMC_ERR_INVALID;
    ctx->order_hint.assign(order, order + n);          // validated against the case by relmc_case_load
    return RELMC_OK;
}

int32_t relmc_case_order(const relmc_ctx* ctx, int32_t* primary_out, int32_t probe_failures_out[3])
{
    if (!ctx) return RELMC_ERR_INVALID;
    if (!ctx->has_case) return RELMC_ERR_NO_CASE;
    if (primary_out) *primary_out = ctx->order_primary;
    if (probe_failures_out) for (int k = 0; k < 3; ++k) probe_failures_out[k] = ctx->order_probe[k];
    return RELMC_OK;
}

int32_t relmc_case_thresholds(const relmc_ctx* ctx, uint32_t* out)
{
    if (!ctx || !out) return RELMC_ERR_INVALID;
    if (!ctx->has_case) return RELMC_ERR_NO_CASE;
    std::memcpy(out, ctx->tile == 0 ? ctx->hcase24.thr : ctx->hcase96.thr, sizeof(uint32_t) * ctx->ncomp);
    return RELMC_OK;
}

int32_t relmc_mc_sampling_dev(relmc_ctx* ctx, uint64_t seed, uint64_t first_index, int64_t n, uint8_t* eqstatus_dev)
{
    if (!ctx) return RELMC_ERR_INVALID;
    if (!ctx->has_case) return fail(ctx, RELMC_ERR_NO_CASE, "relmc_mc_sampling: no case loaded");
    if (n < 0 || (n > 0 && !eqstatus_dev)) return fail(ctx, RELMC_ERR_INVALID, "relmc_mc_sampling: bad arguments");
    if (n == 0) return RELMC_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int64_t total = n * ((ctx->ncomp + 3) / 4);
    int64_t blocks = (total + 255) / 256;
    if (blocks > (int64_t)ctx->num_cu * 16) blocks = (int64_t)ctx->num_cu * 16;
    if (ctx->tile == 0)
        hipLaunchKernelGGL(relmc_sampling_kernel<Tile24>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, reinterpret_cast<const DevCaseT<Tile24>*>(ctx->dcase), seed, first_index, n, eqstatus_dev);
    else
        hipLaunchKernelGGL(relmc_sampling_kernel<Tile96>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, reinterpret_cast<const DevCaseT<Tile96>*>(ctx->dcase), seed, first_index, n, eqstatus_dev);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RELMC_OK;
}

int32_t relmc_mc_sampling(relmc_ctx* ctx, uint64_t seed, uint64_t first_index, int64_t n, uint8_t* eqstatus_host)
{
    if (!ctx) return RELMC_ERR_INVALID;
    if (!ctx->has_case) return fail(ctx, RELMC_ERR_NO_CASE, "relmc_mc_sampling: no case loaded");
    if (n < 0 || (n > 0 && !eqstatus_host)) return fail(ctx, RELMC_ERR_INVALID, "relmc_mc_sampling: bad arguments");
    if (n == 0) return RELMC_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    uint8_t* dbuf = nullptr;
    const size_t bytes = (size_t)n * ctx->ncomp;
    HIP_TRY(ctx, hipMalloc(&dbuf, bytes));
    int rc = relmc_mc_sampling_dev(ctx, seed, first_index, n, dbuf);
    if (rc == RELMC_OK && hipMemcpy(eqstatus_host, dbuf, bytes, hipMemcpyDeviceToHost) != hipSuccess)
        rc = fail(ctx, RELMC_ERR_HIP, "relmc_mc_sampling: device-to-host copy failed");
    (void)hipFree(dbuf);
    return rc;
}

int32_t relmc_last_kernel_ms(const relmc_ctx* ctx, double* ms)
{
    if (!ctx || !ms) return RELMC_ERR_INVALID;
    *ms = ctx->last_kernel_ms;
    return RELMC_OK;
}

void relmc_acc_zero(relmc_acc* acc) { if (acc) std::memset(acc, 0, sizeof(*acc)); }

void relmc_acc_merge(relmc_acc* d, const relmc_acc* s)
{
    if (!d || !s) return;
    d->n += s->n; d->n_fail += s->n_fail; d->n_singular += s->n_singular; d->n_infeasible += s->n_infeasible;
    d->n_nonconverged += s->n_nonconverged; d->sum_iters += s->sum_iters;
    for (int k = 0; k < RELMC_MAX_COMP; ++k) d->comp_fail[k] += s->comp_fail[k];
    d->n_screened += s->n_screened;
    d->sum_dns += s->sum_dns; d->sum_dns2 += s->sum_dns2;
    for (int i = 0; i < RELMC_MAX_BUS; ++i) d->sum_nodal[i] += s->sum_nodal[i];
}

// nsqMain.m:286-301 (EDNS, LOLE, PLC, beta), :348-349 (nodal), :366-376 (component importance),
// written for per-sample sums: the reference's count-weighted database sums are the same numbers.
void relmc_nsq_indices(const relmc_acc* a, int32_t nb, int32_t ncomp, double hours, relmc_indices* out)
{
    if (!a || !out) return;
    std::memset(out, 0, sizeof(*out));
    out->n = a->n;
    if (a->n <= 0) return;
    const double N = (double)a->n;
    out->edns = a->sum_dns / N;
    out->plc = (double)a->n_fail / N;
    out->lole = out->plc * hours;
    out->eens = out->edns * hours;
    double ss = a->sum_dns2 - N * out->edns * out->edns;
    if (ss < 0) ss = 0;
    out->beta = out->edns > 0 ? std::sqrt(ss) / N / out->edns : INFINITY;   // guard of SURVEY.md App. E (beta = NaN)
    out->mean_iters = (double)a->sum_iters / N;
    if (nb > RELMC_MAX_BUS) nb = RELMC_MAX_BUS;
    if (ncomp > RELMC_MAX_COMP) ncomp = RELMC_MAX_COMP;
    for (int i = 0; i < nb; ++i) out->nodal_eens[i] = a->sum_nodal[i] / N;
    for (int k = 0; k < ncomp; ++k) out->comp_importance[k] = a->n_fail ? (double)a->comp_fail[k] / (double)a->n_fail : 0.0;
}

// test hook: DPP semantics probe (tests/test_gpu_parity.py); in[64] -> out[512]
int32_t relmc_dpp_probe(relmc_ctx* ctx, const double* in_host, double* out_host)
{
    if (!ctx || !in_host || !out_host) return RELMC_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    double* din = nullptr; double* dout = nullptr;
    HIP_TRY(ctx, hipMalloc(&din, sizeof(double) * 64));
    HIP_TRY(ctx, hipMalloc(&dout, sizeof(double) * 512));
    HIP_TRY(ctx, hipMemcpy(din, in_host, sizeof(double) * 64, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(relmc_dpp_probe_kernel, dim3(1), dim3(64), 0, ctx->stream, din, dout);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(out_host, dout, sizeof(double) * 512, hipMemcpyDeviceToHost));
    (void)hipFree(din); (void)hipFree(dout);
    return RELMC_OK;
}
}  // extern "C"
