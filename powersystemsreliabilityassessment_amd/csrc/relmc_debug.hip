// relmc_debug.hip — introspection and test hooks that are not part of include/relmc.h (bound by the Python test suite and the profiling
// scripts through ctypes): the active schedule's shape, every Newton step through the dense pivoted solve, per-phase cycle counters and
// per-iteration traces of the profiling builds, the diagnosis switches of a context.
#include <cstdio>
#include <cstring>

#include "relmc_ctx.h"

using namespace relmc_host;

extern "C" {

// introspection: {npass_upd, npass_inv, npass_bwd, noff, nzero, nws, lds_bytes, blocks_per_cu, total tasks}
int32_t relmc_debug_schedule(const relmc_ctx* ctx, int32_t* out9)
{
    if (!ctx || !out9 || !ctx->has_case) return RELMC_ERR_INVALID;
    auto fill = [&](const auto& C) {
        int ntask = 0;
        for (int p = 0; p < C.npass; ++p) ntask += C.pass_ntask[p];
        out9[0] = C.npass_upd; out9[1] = C.npass_inv; out9[2] = C.npass - C.npass_upd - C.npass_inv; out9[3] = C.noff;
        out9[4] = C.nzero; out9[5] = (int)C.nws; out9[6] = (int)ctx->lds_bytes; out9[7] = ctx->blocks_per_cu; out9[8] = ntask;
        if (verbose()) fprintf(stderr, "relmc: modelled LDS conflict cycles per Newton step %ld -> %ld\n", ctx->conflict_before, ctx->conflict_after);
        if (verbose()) fprintf(stderr, "relmc: longest line / injection list per bus slot: %d %d / %d %d\n", (int)C.maxdeg_s[0], (int)C.maxdeg_s[1], (int)C.maxinj_s[0], (int)C.maxinj_s[1]);
        if (verbose()) { fprintf(stderr, "relmc: tasks per pass:"); for (int p = 0; p < C.npass; ++p) fprintf(stderr, " %d", (int)C.pass_ntask[p]); fprintf(stderr, "\n"); }
    };
    if (ctx->tile == 0) fill(ctx->hcase24); else fill(ctx->hcase96);
    return RELMC_OK;
}

// test hook: mc_simulation with EVERY Newton step solved by the dense, partially pivoted last resort (MODE 6) instead of the static
// sparse schedule -- tests/test_gpu_parity.py compares it with the shipped solver and with the C oracle (whose LU pivots as well)
int32_t relmc_debug_mc_simulation_dense(relmc_ctx* ctx, const uint8_t* states_host, int64_t n, const relmc_solver_opts* opts, double* dns_host,
                                        double* nodal_host, int32_t* status_host, int32_t* iters_host)
{
    if (!ctx || !ctx->has_case || !states_host || !dns_host || n < 0) return RELMC_ERR_INVALID;
    if (n == 0) return RELMC_OK;
    relmc_solver_opts o;
    if (opts) o = *opts; else relmc_solver_opts_default(&o);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int ow = mask_words(ctx), ncomp = ctx->ncomp, nb = ctx->nb;
    std::vector<uint32_t> keys((size_t)n * ow, 0u);
    for (int64_t r = 0; r < n; ++r) for (int k = 0; k < ncomp; ++k) if (states_host[(size_t)r * ncomp + k]) keys[(size_t)r * ow + (k >> 5)] |= 1u << (k & 31);
    uint32_t* dk = nullptr; double* dd = nullptr; int32_t* dm = nullptr; double* dn = nullptr;
    auto cleanup = [&]() { (void)hipFree(dk); (void)hipFree(dd); (void)hipFree(dm); (void)hipFree(dn); };
    if (hipMalloc(&dk, sizeof(uint32_t) * keys.size()) != hipSuccess || hipMalloc(&dd, sizeof(double) * (size_t)n) != hipSuccess ||
        hipMalloc(&dm, sizeof(int32_t) * (size_t)n) != hipSuccess || hipMalloc(&dn, sizeof(double) * (size_t)n * nb) != hipSuccess) { cleanup(); return fail(ctx, RELMC_ERR_HIP, "dense simulation: device allocation failed"); }
    int rc = RELMC_OK;
    if (hipMemcpy(dk, keys.data(), sizeof(uint32_t) * keys.size(), hipMemcpyHostToDevice) != hipSuccess) rc = fail(ctx, RELMC_ERR_HIP, "dense simulation: H2D failed");
    EvalArgs a = make_args(o);
    a.n = n; a.memo_keys = dk; a.db_first = 0; a.dns = dd; a.status = dm; a.nodal = dn;
    int rows = 0;
    if (rc == RELMC_OK) rc = launch_eval(ctx, 6, a, &rows);
    if (rc == RELMC_OK) rc = finish_timing(ctx);
    std::vector<int32_t> meta((size_t)n);
    if (rc == RELMC_OK && (hipMemcpy(dns_host, dd, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost) != hipSuccess ||
                           hipMemcpy(meta.data(), dm, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost) != hipSuccess ||
                           (nodal_host && hipMemcpy(nodal_host, dn, sizeof(double) * (size_t)n * nb, hipMemcpyDeviceToHost) != hipSuccess)))
        rc = fail(ctx, RELMC_ERR_HIP, "dense simulation: D2H failed");
    cleanup();
    if (rc) return rc;
    for (int64_t r = 0; r < n; ++r) { if (status_host) status_host[r] = meta[(size_t)r] & 3; if (iters_host) iters_host[r] = (int32_t)((uint32_t)meta[(size_t)r] >> 8); }
    return RELMC_OK;
}

// profiling hook (only meaningful in -DRELMC_PHASE_TIMING builds): per-phase cycle sums of the last launch
int32_t relmc_debug_phase_cycles(relmc_ctx* ctx, unsigned long long* out8)
{
    if (!ctx || !out8) return RELMC_ERR_INVALID;
    for (int k = 0; k < 8; ++k) out8[k] = 0;
    if (!ctx->dtiming || ctx->timing_waves <= 0) return RELMC_OK;
    std::vector<unsigned long long> h((size_t)ctx->timing_waves * 8);
    HIP_TRY(ctx, hipMemcpy(h.data(), ctx->dtiming, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    for (int w = 0; w < ctx->timing_waves; ++w) for (int k = 0; k < 8; ++k) out8[k] += h[(size_t)w * 8 + k];
    return RELMC_OK;
}

// debug hook (only meaningful in -DRELMC_TRACE builds): per-iteration termination quantities of the first scenario
// of the last launch, 8 doubles per iteration {feascond, gradcond, compcond, costcond, alpha_p, alpha_d, gamma, f}
int32_t relmc_debug_trace(relmc_ctx* ctx, double* out, int32_t n_doubles)
{
    if (!ctx || !out || n_doubles < 0) return RELMC_ERR_INVALID;
    for (int k = 0; k < n_doubles; ++k) out[k] = 0.0;
    if (!ctx->dtiming || n_doubles > 8 * 65536) return RELMC_OK;
    HIP_TRY(ctx, hipMemcpy(out, ctx->dtiming, sizeof(double) * n_doubles, hipMemcpyDeviceToHost));
    return RELMC_OK;
}

// diagnosis switches of the context (tests): "no_retry", "retry_dense_first", "nsq_no_stretch", "db_no_probe"; value 0 / 1.
// no_retry must be set before relmc_case_load (the order calibration and the list arming must agree).
int32_t relmc_debug_set(relmc_ctx* ctx, const char* key, int32_t value)
{
    if (!ctx || !key) return RELMC_ERR_INVALID;
    const bool v = value != 0;
    if (!std::strcmp(key, "no_retry")) ctx->sw.no_retry = v;
    else if (!std::strcmp(key, "retry_dense_first")) ctx->sw.retry_dense_first = v;
    else if (!std::strcmp(key, "nsq_no_stretch")) ctx->sw.nsq_no_stretch = v;
    else if (!std::strcmp(key, "db_no_probe")) ctx->sw.db_no_probe = v;
    else return fail(ctx, RELMC_ERR_INVALID, std::string("relmc_debug_set: unknown switch ") + key);
    return RELMC_OK;
}

}  // extern "C"
