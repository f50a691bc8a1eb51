// relmc_db_kernels.h — kernels of the reference's dedupe and unique-state database on the device (nsqMain.m:91-99, 220-278): outage masks of a
// sampled range, their run-length encoding after the sort, the open-addressing table of row ids, count-weighted reduction of the rows.
#pragma once
#include "relmc_devfn.h"

namespace relmc {

// ---- distinct-state path (nsqMain.m:220-245): masks of a sampled range, sorted and run-length encoded on the device ----
// one thread per scenario: the same draws as relmc_sampling_kernel / MODE 0, packed as OW mask words
template <class TL>
__global__ void __launch_bounds__(256) relmc_memo_keys_kernel(const DevCaseT<TL>* __restrict__ C, uint64_t seed, uint64_t first_index,
                                                              int64_t n, uint32_t* __restrict__ keys)
{
    constexpr int OW = TL::OW;
    const int ncomp = C->ncomp, nblk = (ncomp + 3) >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t gi = first_index + (uint64_t)i;
        uint32_t w[OW];
#pragma unroll
        for (int q = 0; q < OW; ++q) w[q] = 0;
        for (int blk = 0; blk < nblk; ++blk) {
            uint32_t r[4];
            philox4x32_10((uint32_t)gi, (uint32_t)(gi >> 32), (uint32_t)blk, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
            uint32_t nib = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) { const int k = blk * 4 + e; if (k < ncomp && r[e] < C->thr[k]) nib |= 1u << e; }
#pragma unroll
            for (int q = 0; q < OW; ++q) if (q == (blk >> 3)) w[q] |= nib << ((blk & 7) * 4);
        }
#pragma unroll
        for (int q = 0; q < OW; ++q) keys[(size_t)i * OW + q] = w[q];
    }
}

// 64-bit chunk c of the masks in the current order (LSD radix passes, least significant chunk first)
__global__ void __launch_bounds__(256) relmc_memo_chunk_kernel(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ perm, int ow, int c,
                                                               int64_t n, unsigned long long* __restrict__ out)
{
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t* kp = keys + (size_t)perm[j] * ow + 2 * c;
        out[j] = (unsigned long long)kp[0] | ((unsigned long long)kp[1] << 32);
    }
}

__global__ void __launch_bounds__(256) relmc_memo_iota_kernel(int64_t n, uint32_t* __restrict__ perm)
{
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x) perm[j] = (uint32_t)j;
}

// head[j] = 1 when sorted position j starts a new distinct mask
__global__ void __launch_bounds__(256) relmc_memo_heads_kernel(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ perm, int ow, int64_t n,
                                                               uint32_t* __restrict__ head)
{
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x) {
        uint32_t h = 1;
        if (j > 0) {
            const uint32_t* ka = keys + (size_t)perm[j] * ow; const uint32_t* kb = keys + (size_t)perm[j - 1] * ow;
            h = 0;
            for (int q = 0; q < ow; ++q) h |= (ka[q] != kb[q]) ? 1u : 0u;
        }
        head[j] = h;
    }
}

// start[u] = first sorted position of distinct mask u; start[n_distinct] = n; *n_distinct_out = number of distinct masks
__global__ void __launch_bounds__(256) relmc_memo_starts_kernel(const uint32_t* __restrict__ head, const uint32_t* __restrict__ uid, int64_t n,
                                                                uint32_t* __restrict__ start, uint32_t* __restrict__ n_distinct_out)
{
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x) {
        if (head[j]) start[uid[j]] = (uint32_t)j;
        if (j == n - 1) { const uint32_t nu = uid[j] + head[j]; start[nu] = (uint32_t)n; *n_distinct_out = nu; }
    }
}

// ---- persistent state database (nsqMain.m:91-99, 220-278): one row per distinct state ever sampled -------------------
// Rows live in HBM as parallel arrays keys[cap][ow] (outage mask words), count[cap], dns[cap], meta[cap] (status |
// relaxed << 2 | iterations << 8), nodal[cap][nb]; an open-addressing table of row ids (linear probing, full-key
// compares) finds a state.  Rows are appended in the order of first appearance in the global sample stream, which makes
// the database (and every fp64 sum over it) independent of the batch size.
constexpr uint32_t DB_EMPTY = 0xffffffffu;
DEVFI uint64_t db_hash(const uint32_t* k, int ow)
{
    uint64_t h = 0x9E3779B97F4A7C15ull;
    for (int q = 0; q < ow; ++q) { h = (h ^ k[q]) * 0xff51afd7ed558ccdull; h ^= h >> 29; }
    return h;
}

// nsqMain.m:232-245 for the distinct states of one batch (unique within the batch, so no two threads touch the same row):
// known state -> its count grows by the multiplicity; unknown -> flagged with the index of its first sample (the sort key
// that orders the new rows by first appearance); states[u] = keys[perm[start[u]]], multiplicity start[u+1] - start[u]
__global__ void __launch_bounds__(256) relmc_db_lookup_kernel(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ perm,
                                                              const uint32_t* __restrict__ start, uint32_t nu, int ow,
                                                              const uint32_t* __restrict__ db_keys, unsigned long long* __restrict__ db_count,
                                                              const uint32_t* __restrict__ table, uint64_t tmask,
                                                              uint32_t* __restrict__ first_idx, uint32_t* __restrict__ uid, uint32_t* __restrict__ n_new)
{
    for (uint32_t u = blockIdx.x * blockDim.x + threadIdx.x; u < nu; u += gridDim.x * blockDim.x) {
        const uint32_t s0 = start[u], first = perm[s0];          // stable sort: the run's first entry is the earliest sample
        const uint32_t* k = keys + (size_t)first * ow;
        uint64_t h = db_hash(k, ow) & tmask;
        uint32_t found = DB_EMPTY;
        for (;;) {
            const uint32_t r = table[h];
            if (r == DB_EMPTY) break;
            const uint32_t* dk = db_keys + (size_t)r * ow;
            bool eq = true;
            for (int q = 0; q < ow; ++q) eq = eq && dk[q] == k[q];
            if (eq) { found = r; break; }
            h = (h + 1) & tmask;
        }
        uid[u] = u;
        if (found != DB_EMPTY) { db_count[found] += (unsigned long long)(start[u + 1] - s0); first_idx[u] = DB_EMPTY; }
        else { first_idx[u] = first; atomicAdd(n_new, 1u); }
    }
}

// nsqMain.m:269-278, the state and count columns of the new rows: row db_n + k = k-th new state in order of first appearance
__global__ void __launch_bounds__(256) relmc_db_insert_kernel(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ perm,
                                                              const uint32_t* __restrict__ start, const uint32_t* __restrict__ sorted_u, uint32_t n_new,
                                                              int ow, uint64_t db_n, uint32_t* __restrict__ db_keys,
                                                              unsigned long long* __restrict__ db_count, uint32_t* __restrict__ table, uint64_t tmask)
{
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n_new; k += gridDim.x * blockDim.x) {
        const uint32_t u = sorted_u[k], s0 = start[u];
        const uint32_t* src = keys + (size_t)perm[s0] * ow;
        const uint64_t row = db_n + k;
        uint32_t* dst = db_keys + row * ow;
        for (int q = 0; q < ow; ++q) dst[q] = src[q];
        db_count[row] = (unsigned long long)(start[u + 1] - s0);
        uint64_t h = db_hash(src, ow) & tmask;
        while (atomicCAS(&table[h], DB_EMPTY, (uint32_t)row) != DB_EMPTY) h = (h + 1) & tmask;   // keys are distinct: claim the first free slot
    }
}

// nsqMain.m:232-245 per SAMPLE, before any sorting (round 2b): every sample of the batch computes its mask and probes the table.
// Hit (the large majority once the database is warm: >= 96 % on RTS-24): the row's count grows — pre-aggregated per block in a
// small LDS hash so that the all-up state and the single-outage states do not serialise on one L2 atomic.  Miss: the sample's
// index and mask are appended to a miss list; only that list goes through the dedupe sort.
template <class TL>
__global__ void __launch_bounds__(256) relmc_db_probe_kernel(const DevCaseT<TL>* __restrict__ C, uint64_t seed, uint64_t first_index, int64_t n,
                                                             const uint32_t* __restrict__ db_keys, unsigned long long* __restrict__ db_count,
                                                             const uint32_t* __restrict__ table, uint64_t tmask,
                                                             uint32_t* __restrict__ miss_idx, uint32_t* __restrict__ miss_keys, uint32_t* __restrict__ n_miss)
{
    constexpr int OW = TL::OW;
    constexpr int LH = 512;
    __shared__ uint32_t lrow[LH];
    __shared__ uint32_t lcnt[LH];
    const int ncomp = C->ncomp, nblk = (ncomp + 3) >> 2, tid = threadIdx.x;
    for (int k = tid; k < LH; k += 256) { lrow[k] = DB_EMPTY; lcnt[k] = 0; }
    __syncthreads();
    constexpr int SUB = 4;                                   // samples per thread between two flushes of the block's table
    for (int64_t base = (int64_t)blockIdx.x * 256 * SUB; base < n; base += (int64_t)gridDim.x * 256 * SUB) {
      for (int sub = 0; sub < SUB; ++sub) {
        const int64_t i = base + sub * 256 + tid;
        if (i < n) {
            const uint64_t gi = first_index + (uint64_t)i;
            uint32_t w[OW];
#pragma unroll
            for (int q = 0; q < OW; ++q) w[q] = 0;
            for (int blk = 0; blk < nblk; ++blk) {
                uint32_t r[4];
                philox4x32_10((uint32_t)gi, (uint32_t)(gi >> 32), (uint32_t)blk, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
                uint32_t nib = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) { const int k = blk * 4 + e; if (k < ncomp && r[e] < C->thr[k]) nib |= 1u << e; }
#pragma unroll
                for (int q = 0; q < OW; ++q) if (q == (blk >> 3)) w[q] |= nib << ((blk & 7) * 4);
            }
            uint64_t h = db_hash(w, OW) & tmask;
            uint32_t found = DB_EMPTY;
            for (;;) {
                const uint32_t r = table[h];
                if (r == DB_EMPTY) break;
                const uint32_t* dk = db_keys + (size_t)r * OW;
                bool eq = true;
#pragma unroll
                for (int q = 0; q < OW; ++q) eq = eq && dk[q] == w[q];
                if (eq) { found = r; break; }
                h = (h + 1) & tmask;
            }
            if (found != DB_EMPTY) {
                uint32_t sl = (found * 2654435761u) >> 23;                 // 9 bits
                for (int tries = 0; ; ++tries) {
                    const uint32_t old = atomicCAS(&lrow[sl], DB_EMPTY, found);
                    if (old == DB_EMPTY || old == found) { atomicAdd(&lcnt[sl], 1u); break; }
                    if (tries == LH) { atomicAdd(&db_count[found], 1ull); break; }      // block table full: straight to memory
                    sl = (sl + 1) & (LH - 1);
                }
            } else {
                const uint32_t pos = atomicAdd(n_miss, 1u);
                miss_idx[pos] = (uint32_t)i;
#pragma unroll
                for (int q = 0; q < OW; ++q) miss_keys[(size_t)pos * OW + q] = w[q];
            }
        }
      }
        __syncthreads();
        for (int k = tid; k < LH; k += 256) {
            if (lrow[k] != DB_EMPTY) { atomicAdd(&db_count[lrow[k]], (unsigned long long)lcnt[k]); lrow[k] = DB_EMPTY; lcnt[k] = 0; }
        }
        __syncthreads();
    }
}

// dns of every sample of a range whose states are all in the database already (the range has just been through
// relmc_nsq_db_batch): out[i] = dns of the row holding sample i's state, NaN if there is none.  Feeds the per-checkpoint
// indices of small batches (relmc_nsq_run), which need the order of the samples the count-weighted rows no longer have.
template <class TL>
__global__ void __launch_bounds__(256) relmc_db_sample_dns_kernel(const DevCaseT<TL>* __restrict__ C, uint64_t seed, uint64_t first_index, int64_t n,
                                                                  const uint32_t* __restrict__ db_keys, const double* __restrict__ db_dns,
                                                                  const uint32_t* __restrict__ table, uint64_t tmask, double* __restrict__ out)
{
    constexpr int OW = TL::OW;
    const int ncomp = C->ncomp, nblk = (ncomp + 3) >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const uint64_t gi = first_index + (uint64_t)i;
        uint32_t w[OW];
#pragma unroll
        for (int q = 0; q < OW; ++q) w[q] = 0;
        for (int blk = 0; blk < nblk; ++blk) {
            uint32_t r[4];
            philox4x32_10((uint32_t)gi, (uint32_t)(gi >> 32), (uint32_t)blk, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
            uint32_t nib = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) { const int k = blk * 4 + e; if (k < ncomp && r[e] < C->thr[k]) nib |= 1u << e; }
#pragma unroll
            for (int q = 0; q < OW; ++q) if (q == (blk >> 3)) w[q] |= nib << ((blk & 7) * 4);
        }
        uint64_t h = db_hash(w, OW) & tmask;
        double v = __builtin_nan("");
        for (;;) {
            const uint32_t r = table[h];
            if (r == DB_EMPTY) break;
            const uint32_t* dk = db_keys + (size_t)r * OW;
            bool eq = true;
#pragma unroll
            for (int q = 0; q < OW; ++q) eq = eq && dk[q] == w[q];
            if (eq) { v = db_dns[r]; break; }
            h = (h + 1) & tmask;
        }
        out[i] = v;
    }
}

// the misses in ascending sample order: keys[r] = miss_keys[pos_sorted[r]] (the dedupe's stable sort then keeps, within equal masks,
// the earliest sample first)
__global__ void __launch_bounds__(256) relmc_db_gather_keys_kernel(const uint32_t* __restrict__ miss_keys, const uint32_t* __restrict__ pos_sorted, int ow,
                                                                   uint32_t n, uint32_t* __restrict__ keys)
{
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) {
        const uint32_t* src = miss_keys + (size_t)pos_sorted[r] * ow;
        for (int q = 0; q < ow; ++q) keys[(size_t)r * ow + q] = src[q];
    }
}

// table of row ids rebuilt after the database has grown
__global__ void __launch_bounds__(256) relmc_db_rehash_kernel(const uint32_t* __restrict__ db_keys, uint64_t rows, int ow,
                                                              uint32_t* __restrict__ table, uint64_t tmask)
{
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t h = db_hash(db_keys + r * ow, ow) & tmask;
        while (atomicCAS(&table[h], DB_EMPTY, (uint32_t)r) != DB_EMPTY) h = (h + 1) & tmask;
    }
}

// nsqMain.m:282-301, 348-349, 366-376 over the whole database: the count-weighted sums of the rows (stage 1: one partial
// accumulator image per block of `chunk` consecutive rows; fp64 sums in a fixed order, integer sums by LDS atomics)
__global__ void __launch_bounds__(256) relmc_db_reduce_kernel(int ow, int nb, int ncomp, double fail_threshold, const uint32_t* __restrict__ keys,
                                                              const unsigned long long* __restrict__ count, const double* __restrict__ dns,
                                                              const int32_t* __restrict__ meta, const double* __restrict__ nodal,
                                                              uint64_t rows, uint64_t chunk, DevAcc* __restrict__ partial)
{
    __shared__ unsigned long long si[6 + 256 + 1];
    __shared__ double sd[2][256];
    __shared__ double sn[8][128];
    const int t = threadIdx.x;
    for (int k = t; k < 6 + 256 + 1; k += 256) si[k] = 0ull;
    __syncthreads();
    const uint64_t lo = (uint64_t)blockIdx.x * chunk, hi = lo + chunk < rows ? lo + chunk : rows;
    unsigned long long c_n = 0, c_fail = 0, c_sing = 0, c_inf = 0, c_nc = 0, c_it = 0, c_scr = 0;
    double s1 = 0.0, s2 = 0.0;
    for (uint64_t r = lo + t; r < hi; r += 256) {
        const unsigned long long c = count[r];
        const double d = dns[r];
        const uint32_t m = (uint32_t)meta[r];
        const uint32_t st = m & 3u;
        c_n += c;
        c_it += c * (unsigned long long)(m >> 8);
        if (st == 3u) c_sing += c;
        if (st == 1u || st == 2u) c_nc += c;
        if (m & 4u) c_inf += c;
        if (m & 8u) c_scr += c;                         // row certified by the pre-screen, never solved (relmc_screen.hip)
        if (d != 0.0) { const double cd = (double)c; s1 = __builtin_fma(cd, d, s1); s2 = __builtin_fma(cd * d, d, s2); }
        if (d > fail_threshold) {                        // nsqMain.m:270
            c_fail += c;
            for (int q = 0; q < ow; ++q) {
                uint32_t w = keys[r * ow + q];
                while (w) { const int b = __ffs((int)w) - 1; w &= w - 1; atomicAdd(&si[6 + 32 * q + b], c); }
            }
        }
    }
    atomicAdd(&si[0], c_n); atomicAdd(&si[1], c_fail); atomicAdd(&si[2], c_sing); atomicAdd(&si[3], c_inf); atomicAdd(&si[4], c_nc); atomicAdd(&si[5], c_it); atomicAdd(&si[6 + 256], c_scr);
    sd[0][t] = s1; sd[1][t] = s2;
    // nodal columns: thread (g, bl) sums bus columns bl, bl + 32, ... over the rows lo + g, lo + g + 8, ...
    const int g = t >> 5, bl = t & 31;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    for (uint64_t r = lo + g; r < hi; r += 8) {
        if (dns[r] > 0.0) {                              // mc_simulation.m:65: nodal shed only where load was curtailed
            const double cd = (double)count[r];
            const double* nr = nodal + r * nb;
            if (bl < nb) a0 = __builtin_fma(cd, nr[bl], a0);
            if (bl + 32 < nb) a1 = __builtin_fma(cd, nr[bl + 32], a1);
            if (bl + 64 < nb) a2 = __builtin_fma(cd, nr[bl + 64], a2);
            if (bl + 96 < nb) a3 = __builtin_fma(cd, nr[bl + 96], a3);
        }
    }
    sn[g][bl] = a0; sn[g][bl + 32] = a1; sn[g][bl + 64] = a2; sn[g][bl + 96] = a3;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (t < off) { sd[0][t] += sd[0][t + off]; sd[1][t] += sd[1][t + off]; }
        __syncthreads();
    }
    DevAcc& out = partial[blockIdx.x];
    long long* oi = reinterpret_cast<long long*>(&out);
    for (int k = t; k < 6 + 256 + 1; k += 256) oi[k] = (long long)si[k];
    if (t == 0) { out.sum_dns = sd[0][0]; out.sum_dns2 = sd[1][0]; }
    if (t < 128) out.sum_nodal[t] = ((sn[0][t] + sn[1][t]) + (sn[2][t] + sn[3][t])) + ((sn[4][t] + sn[5][t]) + (sn[6][t] + sn[7][t]));
    (void)ncomp;
}

// stage 2: one wavefront per accumulator word sums the block partials lane-strided and combines them by a fixed butterfly
__global__ void __launch_bounds__(64) relmc_db_final_kernel(const DevAcc* __restrict__ partial, int nblocks, DevAcc* __restrict__ out)
{
    constexpr int NI = 6 + 256 + 1;
    const int item = blockIdx.x, lane = threadIdx.x;
    long long si = 0; double sd = 0.0;
    for (int b = lane; b < nblocks; b += 64) {
        if (item < NI) si += reinterpret_cast<const long long*>(&partial[b])[item];
        else sd += (&partial[b].sum_dns)[item - NI];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { si += __shfl_xor(si, off); sd += __shfl_xor(sd, off); }
    if (lane == 0) {
        if (item < NI) reinterpret_cast<long long*>(out)[item] = si;
        else (&out->sum_dns)[item - NI] = sd;
    }
}
}  // namespace relmc
