// relmc_database.hip — the reference's dedupe (nsqMain.m:220-229) and its persistent unique-state database (nsqMain.m:91-99, 232-278) on
// the device: outage masks sorted and run-length encoded with rocPRIM, rows in HBM behind an open-addressing table of row ids, only new
// states evaluated, the indices as count-weighted sums over all rows (nsqMain.m:282-301, 348-349, 366-376).
#include <chrono>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "relmc_ctx.h"
#include "relmc_db_kernels.h"

namespace relmc_host {

void memo_free(relmc_ctx* ctx)
{
    for (void* p : {(void*)ctx->mk, (void*)ctx->mperm0, (void*)ctx->mperm1, (void*)ctx->mhead, (void*)ctx->muid, (void*)ctx->mstart, (void*)ctx->mnu,
                    (void*)ctx->mch0, (void*)ctx->mch1, ctx->mtmp, (void*)ctx->mmiss, (void*)ctx->mk2}) if (p) (void)hipFree(p);
    ctx->mk = ctx->mperm0 = ctx->mperm1 = ctx->mhead = ctx->muid = ctx->mstart = ctx->mnu = ctx->mmiss = ctx->mk2 = nullptr; ctx->mch0 = ctx->mch1 = nullptr; ctx->mtmp = nullptr;
    ctx->memo_cap = 0; ctx->memo_tmp_bytes = 0;
}

namespace {
// nsqMain.m:220-229 on the device for the samples [first_index, first_index + m): outage masks (ctx->mk), sample indices
// sorted by mask (*perm_out; stable, so every run starts with its earliest sample), run starts (ctx->mstart) and the number
// of distinct states.  Leaves the stream synchronised.
// `keep`: the buffers hold live data (the database's gathered miss keys): growing them would lose it, so that is an error instead
int memo_alloc(relmc_ctx* ctx, int64_t m, bool keep = false)
{
    auto tmp_for = [&](int64_t q) {
        size_t tmp_sort = 0, tmp_scan = 0;
        (void)rocprim::radix_sort_pairs(nullptr, tmp_sort, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr,
                                        (size_t)q, 0u, 64u, ctx->stream);
        (void)rocprim::exclusive_scan(nullptr, tmp_scan, (uint32_t*)nullptr, (uint32_t*)nullptr, 0u, (size_t)q, rocprim::plus<uint32_t>(), ctx->stream);
        return tmp_sort > tmp_scan ? tmp_sort : tmp_scan;
    };
    size_t tmp_need = tmp_for(m);
    if (keep && m <= ctx->memo_cap && tmp_need > ctx->memo_tmp_bytes) {
        // live keys, enough room for them, but rocprim asks for more scratch at this size than at the sizes the scratch was made for: the scratch
        // holds no data, so it alone grows
        if (ctx->mtmp) (void)hipFree(ctx->mtmp);
        ctx->mtmp = nullptr; ctx->memo_tmp_bytes = 0;
        HIP_TRY(ctx, hipMalloc(&ctx->mtmp, tmp_need));
        ctx->memo_tmp_bytes = tmp_need;
        return RELMC_OK;
    }
    if (m > ctx->memo_cap || tmp_need > ctx->memo_tmp_bytes) {
        if (keep) return fail(ctx, RELMC_ERR_INVALID, "state database: the miss list is longer than the buffers it was gathered into (internal)");
        // rocprim switches algorithms with the size and its scratch need is not monotone across the switch points: size the scratch for
        // every smaller power-of-two fraction too, so that a later call on fewer items (the miss list of the same batch) never reallocates
        for (int64_t q = m >> 1; q >= 1; q >>= 1) { const size_t t = tmp_for(q); if (t > tmp_need) tmp_need = t; }
        for (void* p : {(void*)ctx->mk, (void*)ctx->mperm0, (void*)ctx->mperm1, (void*)ctx->mhead, (void*)ctx->muid, (void*)ctx->mstart, (void*)ctx->mnu,
                        (void*)ctx->mch0, (void*)ctx->mch1, ctx->mtmp, (void*)ctx->mmiss, (void*)ctx->mk2}) if (p) (void)hipFree(p);
        ctx->mk = ctx->mperm0 = ctx->mperm1 = ctx->mhead = ctx->muid = ctx->mstart = ctx->mnu = ctx->mmiss = ctx->mk2 = nullptr; ctx->mch0 = ctx->mch1 = nullptr; ctx->mtmp = nullptr;
        ctx->memo_cap = 0; ctx->memo_tmp_bytes = 0;
        HIP_TRY(ctx, hipMalloc(&ctx->mk, sizeof(uint32_t) * (size_t)m * 8));
        HIP_TRY(ctx, hipMalloc(&ctx->mk2, sizeof(uint32_t) * (size_t)m * 8));
        HIP_TRY(ctx, hipMalloc(&ctx->mmiss, sizeof(uint32_t) * (size_t)m));
        HIP_TRY(ctx, hipMalloc(&ctx->mperm0, sizeof(uint32_t) * (size_t)m));
        HIP_TRY(ctx, hipMalloc(&ctx->mperm1, sizeof(uint32_t) * (size_t)m));
        HIP_TRY(ctx, hipMalloc(&ctx->mhead, sizeof(uint32_t) * (size_t)m));
        HIP_TRY(ctx, hipMalloc(&ctx->muid, sizeof(uint32_t) * (size_t)m));
        HIP_TRY(ctx, hipMalloc(&ctx->mstart, sizeof(uint32_t) * ((size_t)m + 1)));
        HIP_TRY(ctx, hipMalloc(&ctx->mnu, sizeof(uint32_t) * 4));
        HIP_TRY(ctx, hipMalloc(&ctx->mch0, sizeof(unsigned long long) * (size_t)m));
        HIP_TRY(ctx, hipMalloc(&ctx->mch1, sizeof(unsigned long long) * (size_t)m));
        HIP_TRY(ctx, hipMalloc(&ctx->mtmp, tmp_need));
        ctx->memo_cap = m; ctx->memo_tmp_bytes = tmp_need;
    }
    return RELMC_OK;
}

// keys_ready: ctx->mk already holds the m masks (the database's miss list); otherwise they are generated from the sampler
int memo_prepare(relmc_ctx* ctx, uint64_t seed, uint64_t first_index, int64_t m, uint32_t* nu_out, uint32_t** perm_out, bool keys_ready = false)
{
    const int ow = ctx->tile == 0 ? Tile24::OW : Tile96::OW;
    const int nchunk = (ctx->ncomp + 63) / 64;
    { const int rc = memo_alloc(ctx, m, keys_ready); if (rc) return rc; }
    int64_t gb = (m + 255) / 256; if (gb > (int64_t)ctx->num_cu * 16) gb = (int64_t)ctx->num_cu * 16;
    const dim3 grid((unsigned)gb), blk(256);
    if (!keys_ready) {
        if (ctx->tile == 0) hipLaunchKernelGGL(relmc_memo_keys_kernel<Tile24>, grid, blk, 0, ctx->stream, reinterpret_cast<const DevCaseT<Tile24>*>(ctx->dcase), seed, first_index, m, ctx->mk);
        else hipLaunchKernelGGL(relmc_memo_keys_kernel<Tile96>, grid, blk, 0, ctx->stream, reinterpret_cast<const DevCaseT<Tile96>*>(ctx->dcase), seed, first_index, m, ctx->mk);
    }
    hipLaunchKernelGGL(relmc_memo_iota_kernel, grid, blk, 0, ctx->stream, m, ctx->mperm0);
    uint32_t* pin = ctx->mperm0; uint32_t* pout = ctx->mperm1;
    for (int c = 0; c < nchunk; ++c) {               // LSD: stable sort by chunk 0, then 1, ...
        hipLaunchKernelGGL(relmc_memo_chunk_kernel, grid, blk, 0, ctx->stream, ctx->mk, pin, ow, c, m, ctx->mch0);
        const int bits = ctx->ncomp - 64 * c < 64 ? ctx->ncomp - 64 * c : 64;
        size_t tb = ctx->memo_tmp_bytes;
        HIP_TRY(ctx, rocprim::radix_sort_pairs(ctx->mtmp, tb, ctx->mch0, ctx->mch1, pin, pout, (size_t)m, 0u, (unsigned)bits, ctx->stream));
        uint32_t* t = pin; pin = pout; pout = t;
    }
    hipLaunchKernelGGL(relmc_memo_heads_kernel, grid, blk, 0, ctx->stream, ctx->mk, pin, ow, m, ctx->mhead);
    { size_t tb = ctx->memo_tmp_bytes;
      HIP_TRY(ctx, rocprim::exclusive_scan(ctx->mtmp, tb, ctx->mhead, ctx->muid, 0u, (size_t)m, rocprim::plus<uint32_t>(), ctx->stream)); }
    hipLaunchKernelGGL(relmc_memo_starts_kernel, grid, blk, 0, ctx->stream, ctx->mhead, ctx->muid, m, ctx->mstart, ctx->mnu);
    HIP_TRY(ctx, hipGetLastError());
    uint32_t nu = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&nu, ctx->mnu, sizeof(nu), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *nu_out = nu; *perm_out = pin;
    return RELMC_OK;
}
}  // namespace

void db_free(relmc_ctx* ctx)
{
    for (void* p : {(void*)ctx->db_keys, (void*)ctx->db_count, (void*)ctx->db_dns, (void*)ctx->db_meta, (void*)ctx->db_nodal, (void*)ctx->db_table,
                    (void*)ctx->db_partial, (void*)ctx->db_snap}) if (p) (void)hipFree(p);
    ctx->db_snap = nullptr; ctx->db_snap_cap = 0;
    ctx->db_keys = nullptr; ctx->db_count = nullptr; ctx->db_dns = nullptr; ctx->db_meta = nullptr; ctx->db_nodal = nullptr; ctx->db_table = nullptr;
    ctx->db_partial = nullptr; ctx->db_partial_cap = 0;
    ctx->db_cap = 0; ctx->db_n = 0; ctx->db_samples = 0; ctx->db_tcap = 0; ctx->db_has_opts = false; ctx->db_invalid = false;
}

// room for `need` rows: the arrays double (contents copied on the device) and the table of row ids is rebuilt
int db_ensure(relmc_ctx* ctx, int64_t need)
{
    if (need <= ctx->db_cap) return RELMC_OK;
    if (need >= (int64_t)0xfffffff0ll) return fail(ctx, RELMC_ERR_UNSUPPORTED, "state database: more than 2^32 rows");
    const int ow = ctx->tile == 0 ? Tile24::OW : Tile96::OW;
    const int nb = ctx->nb;
    int64_t cap = ctx->db_cap ? ctx->db_cap * 2 : (int64_t)1 << 16;
    while (cap < need) cap *= 2;
    uint32_t* keys = nullptr; unsigned long long* count = nullptr; double* dns = nullptr; int32_t* meta = nullptr; double* nodal = nullptr; uint32_t* table = nullptr;
    uint64_t tcap = 1; while (tcap < (uint64_t)cap * 2) tcap <<= 1;
    auto bail = [&]() { (void)hipFree(keys); (void)hipFree(count); (void)hipFree(dns); (void)hipFree(meta); (void)hipFree(nodal); (void)hipFree(table); };
    if (hipMalloc(&keys, sizeof(uint32_t) * (size_t)cap * ow) != hipSuccess || hipMalloc(&count, sizeof(unsigned long long) * (size_t)cap) != hipSuccess ||
        hipMalloc(&dns, sizeof(double) * (size_t)cap) != hipSuccess || hipMalloc(&meta, sizeof(int32_t) * (size_t)cap) != hipSuccess ||
        hipMalloc(&nodal, sizeof(double) * (size_t)cap * nb) != hipSuccess || hipMalloc(&table, sizeof(uint32_t) * tcap) != hipSuccess) {
        bail(); return fail(ctx, RELMC_ERR_HIP, "state database: device allocation failed");
    }
    const size_t n = (size_t)ctx->db_n;
    bool ok = hipMemsetAsync(table, 0xff, sizeof(uint32_t) * tcap, ctx->stream) == hipSuccess;
    if (n) {
        ok = ok && hipMemcpyAsync(keys, ctx->db_keys, sizeof(uint32_t) * n * ow, hipMemcpyDeviceToDevice, ctx->stream) == hipSuccess &&
             hipMemcpyAsync(count, ctx->db_count, sizeof(unsigned long long) * n, hipMemcpyDeviceToDevice, ctx->stream) == hipSuccess &&
             hipMemcpyAsync(dns, ctx->db_dns, sizeof(double) * n, hipMemcpyDeviceToDevice, ctx->stream) == hipSuccess &&
             hipMemcpyAsync(meta, ctx->db_meta, sizeof(int32_t) * n, hipMemcpyDeviceToDevice, ctx->stream) == hipSuccess &&
             hipMemcpyAsync(nodal, ctx->db_nodal, sizeof(double) * n * nb, hipMemcpyDeviceToDevice, ctx->stream) == hipSuccess;
        if (ok) {
            int64_t gb = ((int64_t)n + 255) / 256; if (gb > (int64_t)ctx->num_cu * 16) gb = (int64_t)ctx->num_cu * 16;
            hipLaunchKernelGGL(relmc_db_rehash_kernel, dim3((unsigned)gb), dim3(256), 0, ctx->stream, keys, (uint64_t)n, ow, table, tcap - 1);
            ok = hipGetLastError() == hipSuccess;
        }
    }
    ok = ok && hipStreamSynchronize(ctx->stream) == hipSuccess;
    if (!ok) { bail(); return fail(ctx, RELMC_ERR_HIP, "state database: growing the arrays failed"); }
    for (void* p : {(void*)ctx->db_keys, (void*)ctx->db_count, (void*)ctx->db_dns, (void*)ctx->db_meta, (void*)ctx->db_nodal, (void*)ctx->db_table}) if (p) (void)hipFree(p);
    ctx->db_keys = keys; ctx->db_count = count; ctx->db_dns = dns; ctx->db_meta = meta; ctx->db_nodal = nodal; ctx->db_table = table;
    ctx->db_cap = cap; ctx->db_tcap = tcap;
    return RELMC_OK;
}

bool same_opts(const relmc_solver_opts& a, const relmc_solver_opts& b)
{
    return a.singular_policy == b.singular_policy && a.max_it == b.max_it && a.feastol == b.feastol && a.gradtol == b.gradtol && a.comptol == b.comptol &&
           a.costtol == b.costtol && a.xi == b.xi && a.sigma == b.sigma && a.z0 == b.z0 && a.alpha_min == b.alpha_min && a.max_stepsize == b.max_stepsize && a.screen == b.screen;
}

// nsqMain.m:282-301, 348-349, 366-376: count-weighted sums over every row of the database -> *acc_out
int db_accumulate(relmc_ctx* ctx, relmc_acc* acc_out)
{
    relmc_acc_zero(acc_out);
    if (ctx->db_n == 0) return RELMC_OK;
    const int ow = ctx->tile == 0 ? Tile24::OW : Tile96::OW;
    // the split into blocks depends on the number of rows only, so the fp64 sums do not depend on how the rows arrived
    const uint64_t rows = (uint64_t)ctx->db_n;
    const uint64_t chunk = 256;                          // small chunks: the reduction is latency-bound per block, so use many blocks
    uint64_t nblk = (rows + chunk - 1) / chunk;
    uint64_t per = chunk;
    if (nblk > 4096) { per = (rows + 4095) / 4096; nblk = (rows + per - 1) / per; }
    if ((int)nblk > ctx->db_partial_cap) {
        if (ctx->db_partial) (void)hipFree(ctx->db_partial);
        ctx->db_partial = nullptr; ctx->db_partial_cap = 0;
        HIP_TRY(ctx, hipMalloc(&ctx->db_partial, sizeof(DevAcc) * 4096));
        ctx->db_partial_cap = 4096;
    }
    hipLaunchKernelGGL(relmc_db_reduce_kernel, dim3((unsigned)nblk), dim3(256), 0, ctx->stream, ow, ctx->nb, ctx->ncomp, 1e-4, ctx->db_keys, ctx->db_count,
                       ctx->db_dns, ctx->db_meta, ctx->db_nodal, rows, per, ctx->db_partial);
    hipLaunchKernelGGL(relmc_db_final_kernel, dim3(sizeof(DevAcc) / 8), dim3(64), 0, ctx->stream, ctx->db_partial, (int)nblk, ctx->dacc);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(acc_out, ctx->dacc, sizeof(*acc_out), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RELMC_OK;
}

// the counts of the present rows, saved for db_rewind (relmc_nsq_run takes small batches a stretch at a time and may have to cut one)
int db_snapshot(relmc_ctx* ctx)
{
    const int64_t rows0 = ctx->db_n;
    if (rows0 > ctx->db_snap_cap) {
        if (ctx->db_snap) (void)hipFree(ctx->db_snap);
        ctx->db_snap = nullptr; ctx->db_snap_cap = 0;
        HIP_TRY(ctx, hipMalloc(&ctx->db_snap, sizeof(unsigned long long) * (size_t)ctx->db_cap));
        ctx->db_snap_cap = ctx->db_cap;
    }
    if (rows0) HIP_TRY(ctx, hipMemcpyAsync(ctx->db_snap, ctx->db_count, sizeof(unsigned long long) * (size_t)rows0, hipMemcpyDeviceToDevice, ctx->stream));
    return RELMC_OK;
}

// the database as it was at the snapshot: its first rows0 rows with their counts of then, the table of row ids rebuilt
int db_rewind(relmc_ctx* ctx, int64_t rows0, int64_t samples0)
{
    ctx->db_n = rows0; ctx->db_samples = samples0;
    if (rows0) HIP_TRY(ctx, hipMemcpyAsync(ctx->db_count, ctx->db_snap, sizeof(unsigned long long) * (size_t)rows0, hipMemcpyDeviceToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(ctx->db_table, 0xff, sizeof(uint32_t) * ctx->db_tcap, ctx->stream));
    if (rows0) {
        int64_t gb = (rows0 + 255) / 256; if (gb > (int64_t)ctx->num_cu * 16) gb = (int64_t)ctx->num_cu * 16;
        hipLaunchKernelGGL(relmc_db_rehash_kernel, dim3((unsigned)gb), dim3(256), 0, ctx->stream, ctx->db_keys, (uint64_t)rows0, mask_words(ctx), ctx->db_table, ctx->db_tcap - 1);
        HIP_TRY(ctx, hipGetLastError());
    }
    return RELMC_OK;
}

// dns of every sample of [first_index, first_index + m) looked up in its row (NaN where the state is not in the database)
int db_sample_dns(relmc_ctx* ctx, uint64_t seed, uint64_t first_index, int64_t m, double* dns_dev)
{
    int64_t gs = (m + 255) / 256; if (gs > (int64_t)ctx->num_cu * 16) gs = (int64_t)ctx->num_cu * 16;
    if (ctx->tile == 0) hipLaunchKernelGGL(relmc_db_sample_dns_kernel<Tile24>, dim3((unsigned)gs), dim3(256), 0, ctx->stream, reinterpret_cast<const DevCaseT<Tile24>*>(ctx->dcase),
                                           seed, first_index, m, ctx->db_keys, ctx->db_dns, ctx->db_table, ctx->db_tcap - 1, dns_dev);
    else hipLaunchKernelGGL(relmc_db_sample_dns_kernel<Tile96>, dim3((unsigned)gs), dim3(256), 0, ctx->stream, reinterpret_cast<const DevCaseT<Tile96>*>(ctx->dcase),
                            seed, first_index, m, ctx->db_keys, ctx->db_dns, ctx->db_table, ctx->db_tcap - 1, dns_dev);
    HIP_TRY(ctx, hipGetLastError());
    return RELMC_OK;
}

namespace {
const char* kDbInvalid = "state database: inconsistent after an earlier error (counts advanced without their batch): relmc_db_reset first";
int db_batch_impl(relmc_ctx* ctx, uint64_t seed, uint64_t first_index, int64_t n, const relmc_solver_opts* opts, relmc_acc* acc_out, relmc_db_stats* stats_out);

int db_batch_impl(relmc_ctx* ctx, uint64_t seed, uint64_t first_index, int64_t n, const relmc_solver_opts* opts, relmc_acc* acc_out, relmc_db_stats* stats_out)
{
    relmc_solver_opts o;
    if (opts) o = *opts; else relmc_solver_opts_default(&o);
    if (ctx->db_has_opts && ctx->db_n > 0 && !same_opts(o, ctx->db_opts))
        return fail(ctx, RELMC_ERR_INVALID, "relmc_nsq_db_batch: the database holds results of other solver options (relmc_db_reset first)");
    ctx->db_opts = o; ctx->db_has_opts = true;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int ow = ctx->tile == 0 ? Tile24::OW : Tile96::OW;
    const int64_t kMaxPerLaunch = (int64_t)1 << 27;
    double ms_total = 0.0;
    int64_t new_total = 0, distinct_total = 0;
    for (int64_t done = 0; done < n;) {
        const int64_t m = (n - done) < kMaxPerLaunch ? (n - done) : kMaxPerLaunch;
        const auto t0 = std::chrono::steady_clock::now();
        uint32_t nu = 0; uint32_t* perm = nullptr;
        int rc;
        // A warm database is probed per sample first (most samples are known states: their counts grow right there) and only the
        // misses go through the dedupe; an empty database takes the whole batch through it.
        const bool probe_first = ctx->db_n > 0 && !ctx->sw.db_no_probe;
        if (probe_first) {
            rc = memo_alloc(ctx, m);
            if (rc) return rc;
            uint32_t* dmiss = ctx->mnu + 2;
            HIP_TRY(ctx, hipMemsetAsync(dmiss, 0, sizeof(uint32_t), ctx->stream));
            int64_t gp = (m + 1023) / 1024; if (gp > (int64_t)ctx->num_cu * 8) gp = (int64_t)ctx->num_cu * 8;     // 1024 samples per block and flush
            if (ctx->tile == 0) hipLaunchKernelGGL(relmc_db_probe_kernel<Tile24>, dim3((unsigned)gp), dim3(256), 0, ctx->stream, reinterpret_cast<const DevCaseT<Tile24>*>(ctx->dcase),
                                                   seed, first_index + (uint64_t)done, m, ctx->db_keys, ctx->db_count, ctx->db_table, ctx->db_tcap - 1, ctx->mmiss, ctx->mk2, dmiss);
            else hipLaunchKernelGGL(relmc_db_probe_kernel<Tile96>, dim3((unsigned)gp), dim3(256), 0, ctx->stream, reinterpret_cast<const DevCaseT<Tile96>*>(ctx->dcase),
                                    seed, first_index + (uint64_t)done, m, ctx->db_keys, ctx->db_count, ctx->db_table, ctx->db_tcap - 1, ctx->mmiss, ctx->mk2, dmiss);
            HIP_TRY(ctx, hipGetLastError());
            uint32_t n_miss = 0;
            HIP_TRY(ctx, hipMemcpyAsync(&n_miss, dmiss, sizeof(n_miss), hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            if (n_miss == 0) {
                ctx->db_samples += m;
                ms_total += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                done += m;
                continue;
            }
            // misses in ascending sample order (append order is arbitrary), their masks gathered into the dedupe's key array
            int64_t gm = ((int64_t)n_miss + 255) / 256; if (gm > (int64_t)ctx->num_cu * 16) gm = (int64_t)ctx->num_cu * 16;
            hipLaunchKernelGGL(relmc_memo_iota_kernel, dim3((unsigned)gm), dim3(256), 0, ctx->stream, (int64_t)n_miss, ctx->mperm0);
            uint32_t* idx_sorted = ctx->mhead; uint32_t* pos_sorted = ctx->mperm1;
            { size_t tb = ctx->memo_tmp_bytes, need = 0;
              (void)rocprim::radix_sort_pairs(nullptr, need, ctx->mmiss, idx_sorted, ctx->mperm0, pos_sorted, (size_t)n_miss, 0u, 32u, ctx->stream);
              if (need > tb) return fail(ctx, RELMC_ERR_HIP, "relmc_nsq_db_batch: sort scratch too small");
              HIP_TRY(ctx, rocprim::radix_sort_pairs(ctx->mtmp, tb, ctx->mmiss, idx_sorted, ctx->mperm0, pos_sorted, (size_t)n_miss, 0u, 32u, ctx->stream)); }
            hipLaunchKernelGGL(relmc_db_gather_keys_kernel, dim3((unsigned)gm), dim3(256), 0, ctx->stream, ctx->mk2, pos_sorted, ow, n_miss, ctx->mk);
            HIP_TRY(ctx, hipGetLastError());
            rc = memo_prepare(ctx, seed, 0, (int64_t)n_miss, &nu, &perm, /*keys_ready=*/true);      // :220-229 on the misses
        } else {
            rc = memo_prepare(ctx, seed, first_index + (uint64_t)done, m, &nu, &perm);              // :220-229
        }
        if (rc) return rc;
        rc = db_ensure(ctx, ctx->db_n + (int64_t)nu);
        if (rc) return rc;
        // :232-245: states already in the database collect their counts, the others are flagged.  Scratch arrays of the
        // run-length step are dead by now and reused: first-sample index (sort key) / distinct-state id pairs.
        uint32_t* first_idx = ctx->mhead; uint32_t* uid = ctx->muid;
        uint32_t* first_sorted = reinterpret_cast<uint32_t*>(ctx->mch0); uint32_t* u_sorted = reinterpret_cast<uint32_t*>(ctx->mch1);
        uint32_t* dnew = ctx->mnu + 1;
        HIP_TRY(ctx, hipMemsetAsync(dnew, 0, sizeof(uint32_t), ctx->stream));
        int64_t gb = ((int64_t)nu + 255) / 256; if (gb > (int64_t)ctx->num_cu * 16) gb = (int64_t)ctx->num_cu * 16; if (gb < 1) gb = 1;
        hipLaunchKernelGGL(relmc_db_lookup_kernel, dim3((unsigned)gb), dim3(256), 0, ctx->stream, ctx->mk, perm, ctx->mstart, nu, ow, ctx->db_keys, ctx->db_count,
                           ctx->db_table, ctx->db_tcap - 1, first_idx, uid, dnew);
        HIP_TRY(ctx, hipGetLastError());
        uint32_t n_new = 0;
        HIP_TRY(ctx, hipMemcpyAsync(&n_new, dnew, sizeof(n_new), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        double eval_ms = 0.0;
        if (n_new > 0) {
            // new rows in the order of first appearance (unique(...,'stable'), :220): sort the flagged states by first sample index
            size_t tb = ctx->memo_tmp_bytes, need = 0;
            (void)rocprim::radix_sort_pairs(nullptr, need, first_idx, first_sorted, uid, u_sorted, (size_t)nu, 0u, 32u, ctx->stream);
            if (need > tb) return fail(ctx, RELMC_ERR_HIP, "relmc_nsq_db_batch: sort scratch too small");
            HIP_TRY(ctx, rocprim::radix_sort_pairs(ctx->mtmp, tb, first_idx, first_sorted, uid, u_sorted, (size_t)nu, 0u, 32u, ctx->stream));
            int64_t gi = ((int64_t)n_new + 255) / 256; if (gi > (int64_t)ctx->num_cu * 16) gi = (int64_t)ctx->num_cu * 16;
            hipLaunchKernelGGL(relmc_db_insert_kernel, dim3((unsigned)gi), dim3(256), 0, ctx->stream, ctx->mk, perm, ctx->mstart, u_sorted, n_new, ow, (uint64_t)ctx->db_n,
                               ctx->db_keys, ctx->db_count, ctx->db_table, ctx->db_tcap - 1);
            HIP_TRY(ctx, hipGetLastError());
            // :257-278: evaluate the new states, results into their rows
            EvalArgs a = make_args(o);
            a.n = (int64_t)n_new; a.memo_keys = ctx->db_keys; a.db_first = ctx->db_n;
            a.dns = ctx->db_dns; a.status = ctx->db_meta; a.nodal = ctx->db_nodal;
            // screen = 1 (relmc_screen.hip): the new rows the zero-curtailment certificate covers get their (0, zeros) right away, the interior
            // point runs over the list of the others
            if (o.screen != 0 && ctx->screen.tab.valid != 0) {
                uint32_t n_eval = 0;
                rc = screen_prepass_rows(ctx, ctx->db_n, (int64_t)n_new, &n_eval);
                if (rc) return rc;
                a.n = (int64_t)n_eval; a.memo_perm = ctx->screen.idx;
            }
            int rows = 0;
            RetryOut ro;
            if (a.n > 0) {
                rc = fail_arm(ctx, a, 0, true, a.n);
                if (rc) return rc;
                rc = launch_eval(ctx, 4, a, &rows);
                if (rc) return rc;
                rc = finish_timing(ctx);
                if (rc) return rc;
                eval_ms = ctx->last_kernel_ms;
                rc = fail_retry(ctx, o, a.fail_threshold, nullptr, ro, &eval_ms);
                if (rc) return rc;
            }
            for (size_t r = 0; r < ro.rec.size(); ++r) {          // into the rows the first attempt filled
                const size_t row = (size_t)ctx->db_n + (size_t)ro.rec[r].unit, nbz = (size_t)ctx->nb;
                HIP_TRY(ctx, hipMemcpy(ctx->db_dns + row, &ro.dns[r], sizeof(double), hipMemcpyHostToDevice));
                HIP_TRY(ctx, hipMemcpy(ctx->db_meta + row, &ro.meta[r], sizeof(int32_t), hipMemcpyHostToDevice));
                HIP_TRY(ctx, hipMemcpy(ctx->db_nodal + row * nbz, &ro.nodal[r * nbz], sizeof(double) * nbz, hipMemcpyHostToDevice));
            }
            ctx->db_n += (int64_t)n_new;
        }
        ctx->db_samples += m;
        (void)eval_ms;
        ms_total += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        new_total += n_new; distinct_total += nu;
        done += m;
    }
    if (acc_out) {
        const auto t0 = std::chrono::steady_clock::now();
        int rc = db_accumulate(ctx, acc_out);                                               // :282-301, 348-349, 366-376
        if (rc) return rc;
        ms_total += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    ctx->last_kernel_ms = ms_total;       // sampling + dedupe + lookup + evaluation of the new states + database reduction (host-timed)
    if (stats_out) { stats_out->rows = ctx->db_n; stats_out->samples = ctx->db_samples; stats_out->new_rows = new_total; stats_out->batch_distinct = distinct_total; }
    return RELMC_OK;
}
}  // namespace

}  // namespace relmc_host

using namespace relmc_host;

extern "C" {

// nsqMain.m:220-245 per launch: the sampled range's distinct states are evaluated once each and counted with their
// multiplicities.  Same accumulators as relmc_nsq_accumulate (integers identical, sums up to summation order).
int32_t relmc_nsq_accumulate_distinct(relmc_ctx* ctx, uint64_t seed, uint64_t first_index, int64_t n, const relmc_solver_opts* opts,
                                      relmc_acc* acc_out, int64_t* n_distinct_out)
{
    if (!ctx) return RELMC_ERR_INVALID;
    if (!ctx->has_case) return fail(ctx, RELMC_ERR_NO_CASE, "relmc_nsq_accumulate_distinct: no case loaded");
    if (n < 0 || !acc_out) return fail(ctx, RELMC_ERR_INVALID, "relmc_nsq_accumulate_distinct: bad arguments");
    relmc_acc_zero(acc_out);
    if (n_distinct_out) *n_distinct_out = 0;
    if (n == 0) return RELMC_OK;
    relmc_solver_opts o;
    if (opts) o = *opts; else relmc_solver_opts_default(&o);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // screen = 1 (relmc_screen.hip): the zero-curtailment certificate runs first, one thread per sample; only the masks it does not cover are sorted and
    // run-length encoded, and only their distinct states are solved.  Certified samples add 1 to n and to n_screened and nothing else.
    const bool screen = o.screen != 0 && ctx->screen.tab.valid != 0;
    const int64_t kMaxPerLaunch = screen ? kScreenChunk : (int64_t)1 << 27;      // 32-bit weighted counters per scenario row; the pre-pass' buffers
    double ms_total = 0.0;
    int64_t distinct_total = 0;
    for (int64_t done = 0; done < n;) {
        const int64_t m = (n - done) < kMaxPerLaunch ? (n - done) : kMaxPerLaunch;
        const auto t0 = std::chrono::steady_clock::now();
        uint32_t nu = 0; uint32_t* pin = nullptr;
        int64_t certified = 0;
        int rc = RELMC_OK;
        if (screen) {
            uint32_t ns = 0;
            rc = memo_alloc(ctx, m);                                 // before the masks go into ctx->mk: memo_prepare must not move it
            if (rc == RELMC_OK) rc = screen_prepass_nsq(ctx, seed, first_index + (uint64_t)done, m, &ns, nullptr);
            if (rc == RELMC_OK) rc = screen_gather_keys(ctx, ns, ctx->mk);
            if (rc == RELMC_OK && ns > 0) rc = memo_prepare(ctx, seed, 0, (int64_t)ns, &nu, &pin, /*keys_ready=*/true);
            certified = m - (int64_t)ns;
        } else rc = memo_prepare(ctx, seed, first_index + (uint64_t)done, m, &nu, &pin);
        if (rc) return rc;
        const double prep_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        relmc_acc part;
        relmc_acc_zero(&part);
        if (nu > 0) {
            EvalArgs a = make_args(o);
            a.n = (int64_t)nu; a.memo_keys = ctx->mk; a.memo_perm = pin; a.memo_start = ctx->mstart;
            int rows = 0;
            rc = fail_arm(ctx, a, 0, true, a.n);
            if (rc) return rc;
            rc = launch_eval(ctx, 3, a, &rows);
            if (rc) return rc;
            rc = launch_finalize(ctx, rows);
            if (rc) return rc;
            HIP_TRY(ctx, hipMemcpyAsync(&part, ctx->dacc, sizeof(part), hipMemcpyDeviceToHost, ctx->stream));
            rc = finish_timing(ctx);
            if (rc) return rc;
            ms_total += ctx->last_kernel_ms;
            RetryOut ro;
            rc = fail_retry(ctx, o, a.fail_threshold, nullptr, ro, &ms_total);
            if (rc) return rc;
            for (size_t r = 0; r < ro.rec.size(); ++r)
                acc_add_unit(&part, ro.rec[r], ro.dns[r], ro.meta[r], &ro.nodal[r * (size_t)ctx->nb], ctx->nb, ctx->ncomp, a.fail_threshold);
        }
        ms_total += prep_ms;                                     // (pre-screen +) sampling + sort + run-length encoding, host-timed; the evaluation kernel above
        part.n += certified; part.n_screened += certified;
        relmc_acc_merge(acc_out, &part);
        distinct_total += nu;
        done += m;
    }
    ctx->last_kernel_ms = ms_total;
    if (n_distinct_out) *n_distinct_out = distinct_total;
    return RELMC_OK;
}

int32_t relmc_db_reset(relmc_ctx* ctx)
{
    if (!ctx) return RELMC_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ctx->db_n = 0; ctx->db_samples = 0; ctx->db_has_opts = false; ctx->db_invalid = false;
    if (ctx->db_table) { HIP_TRY(ctx, hipMemsetAsync(ctx->db_table, 0xff, sizeof(uint32_t) * ctx->db_tcap, ctx->stream)); HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); }
    return RELMC_OK;
}

int32_t relmc_db_size(const relmc_ctx* ctx, int64_t* rows_out, int64_t* samples_out)
{
    if (!ctx) return RELMC_ERR_INVALID;
    if (rows_out) *rows_out = ctx->db_n;
    if (samples_out) *samples_out = ctx->db_samples;
    return RELMC_OK;
}

int32_t relmc_nsq_db_batch(relmc_ctx* ctx, uint64_t seed, uint64_t first_index, int64_t n, const relmc_solver_opts* opts,
                           relmc_acc* acc_out, relmc_db_stats* stats_out)
{
    if (!ctx) return RELMC_ERR_INVALID;
    if (!ctx->has_case) return fail(ctx, RELMC_ERR_NO_CASE, "relmc_nsq_db_batch: no case loaded");
    if (n < 0) return fail(ctx, RELMC_ERR_INVALID, "relmc_nsq_db_batch: bad arguments");
    if (ctx->db_invalid) return fail(ctx, RELMC_ERR_INVALID, kDbInvalid);
    // The per-sample probe bumps the counts of known rows before the steps that can still fail (scratch, growth past 2^32 rows, the
    // evaluation launch); an error return after that leaves counts without their samples, so the database is closed until it is reset.
    const int64_t rows0 = ctx->db_n, samples0 = ctx->db_samples;
    const int rc = db_batch_impl(ctx, seed, first_index, n, opts, acc_out, stats_out);
    if (rc != RELMC_OK && n > 0 && (rows0 > 0 || ctx->db_n != rows0 || ctx->db_samples != samples0)) ctx->db_invalid = true;
    return rc;
}

int32_t relmc_db_accumulate(relmc_ctx* ctx, relmc_acc* acc_out)
{
    if (!ctx || !acc_out) return RELMC_ERR_INVALID;
    if (!ctx->has_case) return fail(ctx, RELMC_ERR_NO_CASE, "relmc_db_accumulate: no case loaded");
    if (ctx->db_invalid) return fail(ctx, RELMC_ERR_INVALID, kDbInvalid);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return db_accumulate(ctx, acc_out);
}

int32_t relmc_db_export(relmc_ctx* ctx, int64_t first_row, int64_t n_rows, uint8_t* states_host, int64_t* count_host, double* dns_host,
                        int32_t* flag_host, double* nodal_host, int32_t* status_host, int32_t* iters_host, uint8_t* relaxed_host)
{
    if (!ctx) return RELMC_ERR_INVALID;
    if (!ctx->has_case) return fail(ctx, RELMC_ERR_NO_CASE, "relmc_db_export: no case loaded");
    if (ctx->db_invalid) return fail(ctx, RELMC_ERR_INVALID, kDbInvalid);
    if (first_row < 0 || n_rows < 0 || first_row + n_rows > ctx->db_n) return fail(ctx, RELMC_ERR_INVALID, "relmc_db_export: row range outside the database");
    if (n_rows == 0) return RELMC_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int ow = ctx->tile == 0 ? Tile24::OW : Tile96::OW;
    const int ncomp = ctx->ncomp, nb = ctx->nb;
    const size_t n = (size_t)n_rows, f = (size_t)first_row;
    std::vector<uint32_t> keys; std::vector<unsigned long long> cnt; std::vector<double> dns; std::vector<int32_t> meta;
    if (states_host) { keys.resize(n * ow); HIP_TRY(ctx, hipMemcpy(keys.data(), ctx->db_keys + f * ow, sizeof(uint32_t) * n * ow, hipMemcpyDeviceToHost)); }
    if (count_host) { cnt.resize(n); HIP_TRY(ctx, hipMemcpy(cnt.data(), ctx->db_count + f, sizeof(unsigned long long) * n, hipMemcpyDeviceToHost)); }
    if (dns_host || flag_host) { dns.resize(n); HIP_TRY(ctx, hipMemcpy(dns.data(), ctx->db_dns + f, sizeof(double) * n, hipMemcpyDeviceToHost)); }
    if (status_host || iters_host || relaxed_host) { meta.resize(n); HIP_TRY(ctx, hipMemcpy(meta.data(), ctx->db_meta + f, sizeof(int32_t) * n, hipMemcpyDeviceToHost)); }
    if (nodal_host) HIP_TRY(ctx, hipMemcpy(nodal_host, ctx->db_nodal + f * nb, sizeof(double) * n * nb, hipMemcpyDeviceToHost));
    for (size_t r = 0; r < n; ++r) {
        if (states_host) for (int k = 0; k < ncomp; ++k) states_host[r * ncomp + k] = (uint8_t)((keys[r * ow + (k >> 5)] >> (k & 31)) & 1u);
        if (count_host) count_host[r] = (int64_t)cnt[r];
        if (dns_host) dns_host[r] = dns[r];
        if (flag_host) flag_host[r] = dns[r] > 1e-4 ? 1 : 0;                  // nsqMain.m:270
        if (status_host) status_host[r] = meta[r] & 3;
        if (iters_host) iters_host[r] = (int32_t)((uint32_t)meta[r] >> 8);
        if (relaxed_host) relaxed_host[r] = (uint8_t)((meta[r] >> 2) & 3);        // bit 0: an island needed Pmin relaxation / decommit; bit 1: certified by the pre-screen, never solved
    }
    return RELMC_OK;
}

// Resume (nsqMain.m:91-99 keeps state_database in the workspace; the reference's `save` at :404-405 is where a run could be continued
// from): rows exported by relmc_db_export go back into an EMPTY database in the same order -- keys, counts, results, the table of row
// ids -- so the next relmc_nsq_db_batch continues the run as if it had never stopped.  status / iters / relaxed may be NULL (then 0).
int32_t relmc_db_import(relmc_ctx* ctx, const relmc_solver_opts* opts, int64_t n_rows, const uint8_t* states_host, const int64_t* count_host,
                        const double* dns_host, const double* nodal_host, const int32_t* status_host, const int32_t* iters_host, const uint8_t* relaxed_host)
{
    if (!ctx) return RELMC_ERR_INVALID;
    if (!ctx->has_case) return fail(ctx, RELMC_ERR_NO_CASE, "relmc_db_import: no case loaded");
    if (n_rows < 0 || (n_rows > 0 && (!states_host || !count_host || !dns_host || !nodal_host))) return fail(ctx, RELMC_ERR_INVALID, "relmc_db_import: bad arguments");
    if (ctx->db_invalid || ctx->db_n != 0) return fail(ctx, RELMC_ERR_INVALID, "relmc_db_import: the database is not empty (relmc_db_reset first)");
    if (n_rows == 0) return RELMC_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int ow = ctx->tile == 0 ? Tile24::OW : Tile96::OW;
    const int ncomp = ctx->ncomp, nb = ctx->nb;
    const size_t n = (size_t)n_rows;
    int rc = db_ensure(ctx, n_rows);
    if (rc) return rc;
    std::vector<uint32_t> keys(n * ow, 0u); std::vector<unsigned long long> cnt(n); std::vector<int32_t> meta(n);
    int64_t samples = 0;
    for (size_t r = 0; r < n; ++r) {
        for (int k = 0; k < ncomp; ++k) if (states_host[r * ncomp + k]) keys[r * ow + (k >> 5)] |= 1u << (k & 31);
        if (count_host[r] <= 0) return fail(ctx, RELMC_ERR_INVALID, "relmc_db_import: a row with a count below 1");
        cnt[r] = (unsigned long long)count_host[r]; samples += count_host[r];
        meta[r] = (status_host ? (status_host[r] & 3) : 0) | ((relaxed_host && (relaxed_host[r] & 1)) ? 4 : 0) | ((relaxed_host && (relaxed_host[r] & 2)) ? 8 : 0) | ((iters_host ? iters_host[r] : 0) << 8);
    }
    HIP_TRY(ctx, hipMemcpyAsync(ctx->db_keys, keys.data(), sizeof(uint32_t) * n * ow, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->db_count, cnt.data(), sizeof(unsigned long long) * n, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->db_dns, dns_host, sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->db_meta, meta.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->db_nodal, nodal_host, sizeof(double) * n * nb, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(ctx->db_table, 0xff, sizeof(uint32_t) * ctx->db_tcap, ctx->stream));
    int64_t gb = ((int64_t)n + 255) / 256; if (gb > (int64_t)ctx->num_cu * 16) gb = (int64_t)ctx->num_cu * 16;
    hipLaunchKernelGGL(relmc_db_rehash_kernel, dim3((unsigned)gb), dim3(256), 0, ctx->stream, ctx->db_keys, (uint64_t)n, ow, ctx->db_table, ctx->db_tcap - 1);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->db_n = n_rows; ctx->db_samples = samples;
    if (opts) { ctx->db_opts = *opts; ctx->db_has_opts = true; } else { relmc_solver_opts_default(&ctx->db_opts); ctx->db_has_opts = true; }
    return RELMC_OK;
}

}  // extern "C"
