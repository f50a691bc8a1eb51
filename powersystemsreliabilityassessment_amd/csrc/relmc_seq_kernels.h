// relmc_seq_kernels.h — kernels of the sequential track (Montecarlo_seq/: chronology sampling, contingency-hour compaction, annual indices)
// and of the HL1 copper-sheet model (GeneratingAdequacy/PowerSystemAdequacy.jl:169-208).
#pragma once
#include "relmc_devfn.h"

namespace relmc {

// seq_mcsampling.m:35-76: alternate TTF = round(-MTTF ln U) and TTR = ceil(-MTTR ln U), every year starts all-up (seqMain.m:91 calls it
// with num_years = 1).  U of event e of component k in global year y = (philox(ctr=(y_lo, y_hi, k | 0x80000000, e >> 2), key=seed)[e & 3] + 0.5) / 2^32.
// One workgroup per (year, segment of seg_len hours).  A lane walks one component's chain from hour 0 (tens of events; components dealt
// round-robin to the four wavefronts, so each has its share of the frequently failing units); a down interval that reaches into the segment
// is filled by the whole wavefront, 64 hours per step, into the segment's masks in LDS (word-major, row stride `sl` = 64 / mw mod 64 so that
// both the fill and the copy-out are conflict-free).  The finished segment leaves as one coalesced copy to masks[year][hour][mw x u32]
// (bit k) -- every word written: no zero-fill, no global atomics.
__global__ void __launch_bounds__(256) relmc_seq_sampling_kernel(const SeqCase* __restrict__ Q, uint64_t seed, uint64_t first_year,
                                                                 int32_t nseg, int32_t seg_len, int32_t sl, uint32_t* __restrict__ masks)
{
    extern __shared__ uint32_t seg[];
    const int ncomp = Q->ncomp, hpy = Q->hpy, mw = Q->mw;
    const int y = (int)(blockIdx.x / (unsigned)nseg), sg = (int)(blockIdx.x - (unsigned)y * (unsigned)nseg);
    const int lo = sg * seg_len, hi = lo + seg_len < hpy ? lo + seg_len : hpy;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int q = threadIdx.x; q < sl * mw; q += 256) seg[q] = 0u;
    __syncthreads();
    const uint64_t gy = first_year + (uint64_t)y;
    for (int k0 = 0; k0 < ncomp; k0 += 256) {
        const int k = k0 + lane * 4 + wv;
        const bool mine = k < ncomp;
        const double mttf = mine ? Q->mttf[k] : 1.0, mttr = mine ? Q->mttr[k] : 1.0;
        long long current = mine ? 0 : (long long)hi;
        bool up = true;
        uint32_t w[4];
        for (int ev = 0; ; ++ev) {
            const bool act = current < hi;                                           // events from `hi` on belong to later segments
            if (!__any(act)) break;
            if ((ev & 3) == 0)
                philox4x32_10((uint32_t)gy, (uint32_t)(gy >> 32), (uint32_t)k | 0x80000000u, (uint32_t)(ev >> 2), (uint32_t)seed, (uint32_t)(seed >> 32), w);
            const double l = log(((double)w[ev & 3] + 0.5) * 2.3283064365386963e-10);   // U in (0, 1)
            int fb = 0, fe = 0;
            if (act) {
                if (up) current += (long long)__builtin_floor(-mttf * l + 0.5);     // round(), seq_mcsampling.m:53
                else {
                    const long long dur = (long long)__builtin_ceil(-mttr * l);      // ceil(), >= 1 h, seq_mcsampling.m:60
                    const long long b = current > lo ? current : lo, e = current + dur < hi ? current + dur : hi;   // e: one past the last down hour
                    if (e > b) { fb = (int)(b - lo); fe = (int)(e - lo); }
                    current += dur;
                }
            }
            up = !up;
            for (uint64_t pend = __ballot(fe > fb); pend; pend &= pend - 1) {
                const int src = __builtin_ctzll(pend);
                const int sb = __builtin_amdgcn_readlane(fb, src), se = __builtin_amdgcn_readlane(fe, src), sk = k0 + src * 4 + wv;
                uint32_t* const row = seg + (sk >> 5) * sl;
                const uint32_t bit = 1u << (sk & 31);
                for (int h = sb + lane; h < se; h += 64) atomicOr(&row[h], bit);     // the other wavefronts write other bits of the same words
            }
        }
    }
    __syncthreads();
    uint32_t* const out = masks + ((size_t)y * hpy + (size_t)lo) * mw;
    const int nw = (hi - lo) * mw;
    for (int q = threadIdx.x; q < nw; q += 256) out[q] = seg[(q % mw) * sl + q / mw];
}

// masks -> uint8 states [years][hours][ncomp] (the materialised seq_mcsampling output)
__global__ void __launch_bounds__(256) relmc_seq_expand_kernel(const uint32_t* __restrict__ masks, int64_t nhours_total, int ncomp, int mw,
                                                               uint8_t* __restrict__ states)
{
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nhours_total * ncomp; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t h = t / ncomp; const int k = (int)(t - h * ncomp);
        states[t] = (masks[h * mw + (k >> 5)] >> (k & 31)) & 1u;
    }
}

// seqMain.m:97-100: hours with at least one component down, kept in ascending order (one workgroup per year)
__global__ void __launch_bounds__(256) relmc_seq_compact_kernel(const uint32_t* __restrict__ masks, int hpy, int mw, uint16_t* __restrict__ hours,
                                                                uint32_t* __restrict__ counts)
{
    __shared__ uint32_t wsum[4];
    __shared__ uint32_t base;
    const int y = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) base = 0;
    __syncthreads();
    for (int h0 = 0; h0 < hpy; h0 += 256) {
        const int h = h0 + tid;
        bool f = false;
        if (h < hpy) { const uint32_t* m = masks + ((size_t)y * hpy + h) * mw; uint32_t o = 0; for (int q = 0; q < mw; ++q) o |= m[q]; f = o != 0; }
        const uint64_t b = __ballot(f);
        const uint32_t before = (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wv] = (uint32_t)__popcll(b);
        __syncthreads();
        uint32_t off = base;
        for (int q = 0; q < wv; ++q) off += wsum[q];
        if (f) hours[(size_t)y * hpy + off + before] = (uint16_t)h;
        __syncthreads();
        if (tid == 0) base += wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
    }
    if (tid == 0) counts[y] = base;
}

// seqMain.m:136-176 + calnlc.m:22-32: annual loss hours (dlc), loss events (nlc = rising edges of the loss flag,
// hour 1 counts) and energy not supplied; one workgroup per year, fixed summation order.
__global__ void __launch_bounds__(256) relmc_seq_annual_kernel(const double* __restrict__ curt, int hpy, double threshold,
                                                               double* __restrict__ year_out /* [years][3] = ens, dlc, nlc */)
{
    __shared__ double red[3][256];
    const int y = blockIdx.x, tid = threadIdx.x;
    const double* c = curt + (size_t)y * hpy;
    double ens = 0.0, dlc = 0.0, nlc = 0.0;
    for (int h = tid; h < hpy; h += 256) {
        const double v = c[h];
        const bool f = v > threshold;
        ens += v;
        if (f) { dlc += 1.0; if (h == 0 || !(c[h - 1] > threshold)) nlc += 1.0; }
    }
    red[0][tid] = ens; red[1][tid] = dlc; red[2][tid] = nlc;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (tid < off) { red[0][tid] += red[0][tid + off]; red[1][tid] += red[1][tid + off]; red[2][tid] += red[2][tid + off]; }
        __syncthreads();
    }
    if (tid < 3) year_out[(size_t)y * 3 + tid] = red[tid][0];
}

// ---- HL1 copper sheet (PowerSystemAdequacy.jl:169-208): one thread per iteration ------------------
// sorted[] = hourly loads ascending, suffix[k] = sum(sorted[k:]); loss hours = #{load > cap}, deficit by suffix sums
__global__ void __launch_bounds__(256) relmc_hl1_kernel(const Hl1Case* __restrict__ H, const double* __restrict__ sorted,
                                                        const double* __restrict__ suffix, uint64_t seed, uint64_t first_index,
                                                        int64_t n, double* __restrict__ iter_lole, double* __restrict__ iter_eue,
                                                        double* __restrict__ partial)
{
    __shared__ double red[4][4];
    const int ngen = H->ngen, nh = H->nhours;
    double s_l = 0.0, s_e = 0.0, s_l2 = 0.0, s_e2 = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t gi = first_index + (uint64_t)i;
        double cap = 0.0;
        for (int blk = 0; blk * 4 < ngen; ++blk) {
            uint32_t w[4];
            philox4x32_10((uint32_t)gi, (uint32_t)(gi >> 32), (uint32_t)blk, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), w);
#pragma unroll
            for (int e = 0; e < 4; ++e) { const int g = blk * 4 + e; if (g < ngen && !(w[e] < H->thr[g])) cap += H->cap[g]; }
        }
        int lo = 0, hi = nh;                          // first index with sorted[idx] > cap  (cap < load, :192)
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (sorted[mid] > cap) hi = mid; else lo = mid + 1; }
        const double hours = (double)(nh - lo);
        const double eue = lo < nh ? suffix[lo] - cap * hours : 0.0;
        if (iter_lole) iter_lole[i] = hours;
        if (iter_eue) iter_eue[i] = eue;
        s_l += hours; s_e += eue; s_l2 = __builtin_fma(hours, hours, s_l2); s_e2 = __builtin_fma(eue, eue, s_e2);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { s_l += __shfl_xor(s_l, off); s_e += __shfl_xor(s_e, off); s_l2 += __shfl_xor(s_l2, off); s_e2 += __shfl_xor(s_e2, off); }
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[wv][0] = s_l; red[wv][1] = s_e; red[wv][2] = s_l2; red[wv][3] = s_e2; }
    __syncthreads();
    if (threadIdx.x < 4) partial[(size_t)blockIdx.x * 4 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}
}  // namespace relmc
