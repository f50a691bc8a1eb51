// relmc_screen.hip — the zero-curtailment pre-screen (relmc_solver_opts.screen = 1; SURVEY 8f rank 4 "copper-sheet pre-screen", the reference's
// counterpart of "do not solve what you already know" is its state database, nsqMain.m:220-245).
//
// What makes it sound: mc_simulation.m:57-59 zeroes dns < 0.1 and :65 reports the nodal split only when dns > 0, so for a state whose LP optimum is
// zero curtailment the reference's outputs are exactly (0, zeros(1, nb)) whatever MIPS' trajectory was.  An explicit dispatch that serves all load
// inside every unit and line limit PROVES that optimum.  The certificate tried here (tests/tools/screen_model.py is its host model, 0 false
// certificates against the oracle): units in service loaded proportionally between Pmin and Pmax, DC flows through the base-topology PTDF --
// states with one or two lines out through the outage system (I - H_MM) x = F_M, states with more lines out (or an outage that splits the network) never --
// inside (or on) every rating.
// It covers 91.4 % of the RTS-24 samples (99.9 % of the zero-curtailment ones), 97.0 % of RTS-96's, 99.2 % of the sequential
// track's contingency hours; everything else goes through the interior point as before.
//
// Data flow of the fused non-sequential pass (relmc_nsq_accumulate with screen = 1): one thread per sample draws the outage mask (the same Philox
// draws as MODE 0) and runs the certificate; the samples it does not cover are listed in ascending order (rocprim::select on a counting iterator: stable,
// so every fp64 sum is reproducible) and evaluated by relmc_eval_kernel<7> from the stored masks.  Certified samples add 1 to n and to n_screened and
// nothing else.  HBM traffic of the pre-pass: (4 OW + 1) bytes per sample written, the survivors' masks read once.
#include <chrono>
#include <cmath>
#include <cstring>
#include <limits>

#include <rocprim/rocprim.hpp>

#include "relmc_ctx.h"
#include "relmc_devfn.h"

namespace relmc {

// The certificate for one state: mask words m[OW] (bit k = component k failed: units first, then lines), load scale factor (1 in the
// non-sequential path).  One thread per state; the tables are read-only and shared by every thread (L1 / L2 hits).
template <int OW>
DEVFI bool screen_certify(const ScreenTab& T, const uint32_t (&m)[OW], double scale)
{
    const int ng = T.ng, nl = T.nl;
    double lo = T.sum_pmin, rg = T.sum_rng;
    int nlo = 0, m1 = 0, m2 = 0, nout = 0;
    // the units out of service, one byte each (at most 16: a state with more goes to the interior point), so that the loop over the lines below
    // walks a short list instead of scanning the mask words again for every line
    unsigned long long list0 = 0ull, list1 = 0ull;
#pragma unroll
    for (int q = 0; q < OW; ++q) {
        uint32_t w = m[q];
        while (w) {
            const int k = 32 * q + (__ffs((int)w) - 1);
            w &= w - 1;
            if (k < ng) {
                lo -= T.pmin[k]; rg -= T.rng[k];
                if (nout < 8) list0 |= (unsigned long long)k << (8 * nout); else if (nout < 16) list1 |= (unsigned long long)k << (8 * (nout - 8));
                nout += 1;
            } else { if (nlo == 0) m1 = k - ng; else m2 = k - ng; nlo += 1; }
        }
    }
    if (nlo > 2 || nout > 16) return false;                     // three or more lines out (or an improbable number of units): the interior point
    const double L = T.total_load * scale;
    if (!(lo <= L) || !(L <= lo + rg) || !(rg > 0.0)) return false;   // capacity short of the load, or over-generation at Pmin
    const double t = (L - lo) / rg;                              // every unit in service at Pmin + t (Pmax - Pmin), 0 <= t <= 1
    auto flow = [&](int l) -> double {
        double a = T.f_min[l], b = T.f_rng[l];
        const double2* g = reinterpret_cast<const double2*>(T.gpair) + (size_t)l * ng;      // {PTDF * Pmin, PTDF * range} of unit k on line l: one 16-byte load
        for (int i = 0; i < nout; ++i) {
            const int k = (int)(((i < 8 ? list0 >> (8 * i) : list1 >> (8 * (i - 8)))) & 0xffull);
            const double2 v = g[k];
            a -= v.x; b -= v.y;
        }
        return __builtin_fma(t, b, a) - scale * T.f_load[l];
    };
    // Lines out: transfers x on their terminals with (I - H_MM) x = F_M cancel what would flow through them; F' = F + H[:, M] x (H[m][l] = flow on l per unit
    // sent from from(m) to to(m) in the base topology).  One line: x = F_m / (1 - H_mm).  Two: the 2 x 2 system.  A singular system = the outage splits the
    // network (a bridge; a pair that is a cut): never certified.
    double x1 = 0.0, x2 = 0.0;
    const double* h1 = T.hmat + (size_t)m1 * nl;
    const double* h2 = T.hmat + (size_t)m2 * nl;
    if (nlo == 1) {
        const double den = 1.0 - h1[m1];
        if (!(__builtin_fabs(den) >= 1e-8)) return false;
        x1 = flow(m1) / den;
    } else if (nlo == 2) {
        const double a11 = 1.0 - h1[m1], a12 = -h2[m1], a21 = -h1[m2], a22 = 1.0 - h2[m2];      // row of line m1: (1 - H[m1][m1]) x1 - H[m1][m2] x2 = F_m1, with H[l][m] = hmat[m][l]
        const double det = a11 * a22 - a12 * a21;
        if (!(__builtin_fabs(det) >= 1e-8)) return false;
        const double F1 = flow(m1), F2 = flow(m2);
        x1 = (a22 * F1 - a12 * F2) / det; x2 = (a11 * F2 - a21 * F1) / det;
    }
    // the lines, eight at a time: a unit's byte is taken out of the list once per eight lines and its eight pairs are in flight together; per line the
    // same subtractions in the same order as flow(l), so the decisions are flow(l)'s
    constexpr int CH = 8;
    const double2* const gp = reinterpret_cast<const double2*>(T.gpair);
    for (int l0 = 0; l0 < nl; l0 += CH) {
        double a[CH], b[CH];
        int lj[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) { lj[j] = l0 + j < nl ? l0 + j : nl - 1; a[j] = T.f_min[lj[j]]; b[j] = T.f_rng[lj[j]]; }
        for (int i = 0; i < nout; ++i) {
            const int k = (int)(((i < 8 ? list0 >> (8 * i) : list1 >> (8 * (i - 8)))) & 0xffull);
#pragma unroll
            for (int j = 0; j < CH; ++j) { const double2 v = gp[(size_t)lj[j] * ng + k]; a[j] -= v.x; b[j] -= v.y; }
        }
        bool over = false;
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int l = l0 + j;
            if (l < nl && !(nlo >= 1 && (l == m1 || (nlo == 2 && l == m2)))) {      // a line out carries nothing
                double f = __builtin_fma(t, b[j], a[j]) - scale * T.f_load[l];
                if (nlo >= 1) f = __builtin_fma(h1[l], x1, f);
                if (nlo == 2) f = __builtin_fma(h2[l], x2, f);
                over = over || !(__builtin_fabs(f) <= T.lim[l]);
            }
        }
        if (over) return false;
    }
    return true;
}

// flags[i]: 0 = certified, 1 = to be solved.  keys[i][OW] = the sample's outage mask (the same draws as relmc_memo_keys_kernel / MODE 0).
template <class TL>
__global__ void __launch_bounds__(256) relmc_screen_sample_kernel(const DevCaseT<TL>* __restrict__ C, const ScreenTab T, uint64_t seed, uint64_t first_index,
                                                                  int64_t n, uint32_t* __restrict__ keys, uint8_t* __restrict__ flags)
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
            for (int e = 0; e < 4; ++e) { const int k = blk * 4 + e; if (k < ncomp && r[e] < C->thr[k]) nib |= 1u << e; }   // strict '<', mc_sampling.m:35
#pragma unroll
            for (int q = 0; q < OW; ++q) if (q == (blk >> 3)) w[q] |= nib << ((blk & 7) * 4);
        }
        const bool cert = T.valid && screen_certify<OW>(T, w, 1.0);
        flags[i] = cert ? 0 : 1;
        if (!cert) {
#pragma unroll
            for (int q = 0; q < OW; ++q) keys[(size_t)i * OW + q] = w[q];
        }
    }
}

// the certificate of given masks: keys[first + i][OW], optional load scale per state -> flags[i] (0 = certified).  Database rows, test hook.
template <int OW>
__global__ void __launch_bounds__(256) relmc_screen_keys_kernel(const ScreenTab T, const uint32_t* __restrict__ keys, const double* __restrict__ load_scale,
                                                                int64_t n, uint8_t* __restrict__ flags)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t w[OW];
#pragma unroll
        for (int q = 0; q < OW; ++q) w[q] = keys[(size_t)i * OW + q];
        flags[i] = (T.valid && screen_certify<OW>(T, w, load_scale ? load_scale[i] : 1.0)) ? 0 : 1;
    }
}

// out[j][ow] = keys[idx[j]][ow]: the uncovered samples' masks packed in ascending sample order (the per-batch dedupe sorts them)
__global__ void __launch_bounds__(256) relmc_screen_gather_keys_kernel(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ idx, int64_t ns, int ow,
                                                                       uint32_t* __restrict__ out)
{
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < ns * ow; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t j = t / ow; const int q = (int)(t - j * ow);
        out[t] = keys[(size_t)idx[j] * ow + q];
    }
}

// uint8 states [n][ncomp] -> mask words [n][OW]
__global__ void __launch_bounds__(256) relmc_screen_pack_kernel(const uint8_t* __restrict__ states, int64_t n, int ncomp, int ow, uint32_t* __restrict__ keys)
{
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n * ow; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = t / ow; const int q = (int)(t - i * ow);
        uint32_t w = 0;
        for (int b = 0; b < 32; ++b) { const int k = 32 * q + b; if (k < ncomp && states[i * ncomp + k]) w |= 1u << b; }
        keys[t] = w;
    }
}

// new database rows [first, first + n): the certified ones get their results right here -- dns 0, status converged | screened (bit 3) | 0 iterations,
// nodal zeros: exactly what the interior point's outputs reduce to (mc_simulation.m:57-59, 65) -- the others are flagged for MODE 4
template <int OW>
__global__ void __launch_bounds__(256) relmc_screen_rows_kernel(const ScreenTab T, const uint32_t* __restrict__ keys, int64_t first, int64_t n, int nb,
                                                                double* __restrict__ dns, int32_t* __restrict__ meta, double* __restrict__ nodal,
                                                                uint8_t* __restrict__ flags)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = first + i;
        uint32_t w[OW];
#pragma unroll
        for (int q = 0; q < OW; ++q) w[q] = keys[(size_t)r * OW + q];
        const bool cert = T.valid && screen_certify<OW>(T, w, 1.0);
        flags[i] = cert ? 0 : 1;
        if (cert) {
            dns[r] = 0.0; meta[r] = 8;
            for (int b = 0; b < nb; ++b) nodal[(size_t)r * nb + b] = 0.0;
        }
    }
}

// seqMain.m:97-100 behind the pre-screen, in two steps.  (1) one thread per hour of the chronology: flags[h] = 0 no component down, 1 a contingency
// hour the certificate covers at the hour's own load factor, 2 a contingency hour for the interior point.  (2) one workgroup per year (as
// relmc_seq_compact_kernel): the count of contingency hours and the ascending list of the flag-2 hours.  (One kernel of one workgroup per year
// that ran the certificate itself kept 125 workgroups busy for 0.72 ms per 125 years: the certificate wants the whole device.)
template <int OW>
__global__ void __launch_bounds__(256) relmc_seq_flag_kernel(const ScreenTab T, const uint32_t* __restrict__ masks, const double* __restrict__ load_factors,
                                                             int hpy, int64_t total_hours, uint8_t* __restrict__ flags)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total_hours; i += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t* mp = masks + (size_t)i * OW;
        uint32_t w[OW]; uint32_t o = 0;
#pragma unroll
        for (int q = 0; q < OW; ++q) { w[q] = mp[q]; o |= w[q]; }
        uint8_t f = 0;
        if (o != 0) f = (T.valid && screen_certify<OW>(T, w, load_factors[(int)(i % hpy)])) ? 1 : 2;
        flags[i] = f;
    }
}

__global__ void __launch_bounds__(256) relmc_seq_compact_flags_kernel(const uint8_t* __restrict__ flags, int hpy, uint16_t* __restrict__ hours,
                                                                      uint32_t* __restrict__ counts, uint32_t* __restrict__ ncont)
{
    __shared__ uint32_t wsum[4], csum[4];
    __shared__ uint32_t base, cbase;
    const int y = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) { base = 0; cbase = 0; }
    __syncthreads();
    for (int h0 = 0; h0 < hpy; h0 += 256) {
        const int h = h0 + tid;
        const uint8_t fl = h < hpy ? flags[(size_t)y * hpy + h] : (uint8_t)0;
        const bool cont = fl != 0, f = fl == 2;
        const uint64_t b = __ballot(f), bc = __ballot(cont);
        const uint32_t before = (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
        if (lane == 0) { wsum[wv] = (uint32_t)__popcll(b); csum[wv] = (uint32_t)__popcll(bc); }
        __syncthreads();
        uint32_t off = base;
        for (int q = 0; q < wv; ++q) off += wsum[q];
        if (f) hours[(size_t)y * hpy + off + before] = (uint16_t)h;
        __syncthreads();
        if (tid == 0) { base += wsum[0] + wsum[1] + wsum[2] + wsum[3]; cbase += csum[0] + csum[1] + csum[2] + csum[3]; }
        __syncthreads();
    }
    if (tid == 0) { counts[y] = base; ncont[y] = cbase; }
}

struct ScreenFlagSet { __device__ bool operator()(uint8_t f) const { return f != 0; } };

}  // namespace relmc

namespace relmc_host {

void screen_free(relmc_ctx* ctx)
{
    auto& S = ctx->screen;
    for (void* p : {(void*)S.dtab, (void*)S.keys, (void*)S.flags, (void*)S.idx, (void*)S.dcount, S.tmp}) if (p) (void)hipFree(p);
    if (S.ev0) (void)hipEventDestroy(S.ev0);
    if (S.ev1) (void)hipEventDestroy(S.ev1);
    S = relmc_ctx::Screen();
}

namespace {
// dense inverse of the reduced susceptance matrix by Gauss-Jordan with partial pivoting; false = singular (the base topology is not one island)
bool invert(std::vector<double>& A, int n)
{
    std::vector<double> I((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i) I[(size_t)i * n + i] = 1.0;
    double amax = 0.0;
    for (double v : A) amax = std::fabs(v) > amax ? std::fabs(v) : amax;
    for (int c = 0; c < n; ++c) {
        int p = c; double best = std::fabs(A[(size_t)c * n + c]);
        for (int r = c + 1; r < n; ++r) if (std::fabs(A[(size_t)r * n + c]) > best) { best = std::fabs(A[(size_t)r * n + c]); p = r; }
        if (!(best > 1e-11 * amax)) return false;
        if (p != c) for (int j = 0; j < n; ++j) { std::swap(A[(size_t)c * n + j], A[(size_t)p * n + j]); std::swap(I[(size_t)c * n + j], I[(size_t)p * n + j]); }
        const double rp = 1.0 / A[(size_t)c * n + c];
        for (int j = 0; j < n; ++j) { A[(size_t)c * n + j] *= rp; I[(size_t)c * n + j] *= rp; }
        for (int r = 0; r < n; ++r) {
            if (r == c) continue;
            const double f = A[(size_t)r * n + c];
            if (f == 0.0) continue;
            for (int j = 0; j < n; ++j) { A[(size_t)r * n + j] -= f * A[(size_t)c * n + j]; I[(size_t)r * n + j] -= f * I[(size_t)c * n + j]; }
        }
    }
    A.swap(I);
    return true;
}
}  // namespace

// The certificate's tables as host arrays (pure arithmetic: no device, no context -- relmc_debug_screen_tables hands them to the CPU test suite).
// Layout of h: pmin[ng], rng[ng], f_min[nl], f_rng[nl], f_load[nl], lim[nl], gpair[nl][ng][2], hmat[nl (line out m)][nl]; bridge[nl] (1 - H[m][m] = 0).  false = a case the certificate
// cannot describe (the base topology is not one island, no unit has a range, a malformed branch).
bool screen_tables(const relmc_case_desc* d, std::vector<double>& h, std::vector<uint8_t>& bridge, double* sum_pmin_out, double* sum_rng_out)
{
    const int nb = d->nb, ng = d->ng, nl = d->nl;
    if (nl < 1 || ng < 1 || nb < 2) return false;
    std::vector<double> X((size_t)nb * nb, 0.0);             // X[i][j], reference row / column zero
    {
        const int n = nb - 1;
        auto red = [&](int i) { return i < d->ref_bus ? i : i - 1; };
        std::vector<double> B((size_t)n * n, 0.0);
        for (int l = 0; l < nl; ++l) {
            const int f = d->br_from[l], t = d->br_to[l]; const double b = d->br_b[l];
            if (f < 0 || f >= nb || t < 0 || t >= nb || f == t) return false;
            if (f != d->ref_bus) B[(size_t)red(f) * n + red(f)] += b;
            if (t != d->ref_bus) B[(size_t)red(t) * n + red(t)] += b;
            if (f != d->ref_bus && t != d->ref_bus) { B[(size_t)red(f) * n + red(t)] -= b; B[(size_t)red(t) * n + red(f)] -= b; }
        }
        if (!invert(B, n)) return false;
        for (int i = 0; i < nb; ++i) for (int j = 0; j < nb; ++j)
            if (i != d->ref_bus && j != d->ref_bus) X[(size_t)i * nb + j] = B[(size_t)red(i) * n + red(j)];
    }
    auto ptdf = [&](int l, int bus) { return d->br_b[l] * (X[(size_t)d->br_from[l] * nb + bus] - X[(size_t)d->br_to[l] * nb + bus]); };
    const size_t n_d = (size_t)2 * ng + (size_t)4 * nl + (size_t)2 * nl * ng + (size_t)nl * nl;       // the pair table starts 16-byte aligned (2 ng + 4 nl doubles before it)
    h.assign(n_d, 0.0);
    double* pmin = h.data(); double* rng = pmin + ng; double* f_min = rng + ng; double* f_rng = f_min + nl; double* f_load = f_rng + nl; double* lim = f_load + nl;
    double* gpair = lim + nl; double* hmat = gpair + (size_t)2 * nl * ng;
    bridge.assign((size_t)nl, 0);
    double sum_pmin = 0.0, sum_rng = 0.0;
    for (int k = 0; k < ng; ++k) {
        pmin[k] = d->inj_pmin[k]; rng[k] = d->inj_pmax[k] - d->inj_pmin[k];
        if (!(rng[k] >= 0.0) || d->inj_bus[k] < 0 || d->inj_bus[k] >= nb) return false;
        sum_pmin += pmin[k]; sum_rng += rng[k];
    }
    if (!(sum_rng > 0.0)) return false;
    // A flow may sit ON its rating (RTS-24: the capacity in service equals the load in 1 % of the samples; every unit then runs at Pmax and the bridge to
    // bus 7 carries exactly its 175 MW): 1e-9 MW of slack for the rounding of the PTDF sums -- four orders inside the 5e-6 p.u. MIPS accepts as feasible.
    constexpr double kSlackMW = 1e-9;
    for (int l = 0; l < nl; ++l) {
        double a = 0.0, b = 0.0, c = 0.0;
        for (int k = 0; k < ng; ++k) {
            const double p = ptdf(l, d->inj_bus[k]);
            gpair[2 * ((size_t)l * ng + k)] = p * pmin[k]; gpair[2 * ((size_t)l * ng + k) + 1] = p * rng[k];
            a += p * pmin[k]; b += p * rng[k];
        }
        for (int i = 0; i < nb; ++i) c += ptdf(l, i) * d->bus_pd[i];
        f_min[l] = a; f_rng[l] = b; f_load[l] = c;
        lim[l] = d->br_rate[l] > 0.0 ? d->br_rate[l] + kSlackMW : std::numeric_limits<double>::infinity();
    }
    for (int m = 0; m < nl; ++m) {                            // H[m][l]: flow on l per unit sent from from(m) to to(m); a bridge has H[m][m] = 1
        const int fm = d->br_from[m], tm = d->br_to[m];
        for (int l = 0; l < nl; ++l) hmat[(size_t)m * nl + l] = ptdf(l, fm) - ptdf(l, tm);
        if (std::fabs(1.0 - hmat[(size_t)m * nl + m]) < 1e-8) bridge[(size_t)m] = 1;
    }
    *sum_pmin_out = sum_pmin; *sum_rng_out = sum_rng;
    return true;
}

// relmc_case_load: PTDF / LODF tables of the certificate into one device buffer.  A case the certificate cannot describe gets valid = 0:
// screen = 1 then certifies nothing.
int screen_build(relmc_ctx* ctx, const relmc_case_desc* d)
{
    screen_free(ctx);                                        // the work buffers are sized for the tile of the case that was loaded
    auto& S = ctx->screen;
    const int ng = d->ng, nl = d->nl;
    S.tab.nl = nl; S.tab.ng = ng; S.tab.valid = 0; S.tab.total_load = d->total_load;
    std::vector<double> h; std::vector<uint8_t> bridge;
    double sum_pmin = 0.0, sum_rng = 0.0;
    if (!screen_tables(d, h, bridge, &sum_pmin, &sum_rng)) return RELMC_OK;
    const size_t n_d = h.size();
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t bytes = n_d * sizeof(double) + (size_t)nl;
    HIP_TRY(ctx, hipMalloc(&S.dtab, bytes));
    HIP_TRY(ctx, hipMemcpy(S.dtab, h.data(), n_d * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(reinterpret_cast<unsigned char*>(S.dtab) + n_d * sizeof(double), bridge.data(), (size_t)nl, hipMemcpyHostToDevice));
    const double* dd = reinterpret_cast<const double*>(S.dtab);
    S.tab.sum_pmin = sum_pmin; S.tab.sum_rng = sum_rng;
    S.tab.pmin = dd; S.tab.rng = dd + ng; S.tab.f_min = dd + 2 * ng; S.tab.f_rng = S.tab.f_min + nl; S.tab.f_load = S.tab.f_rng + nl; S.tab.lim = S.tab.f_load + nl;
    S.tab.gpair = S.tab.lim + nl; S.tab.hmat = S.tab.gpair + (size_t)2 * nl * ng;
    S.tab.bridge = reinterpret_cast<const uint8_t*>(dd + n_d);
    S.tab.valid = 1;
    return RELMC_OK;
}

namespace {
int screen_buffers(relmc_ctx* ctx, int64_t m)
{
    auto& S = ctx->screen;
    if (!S.ev0) { HIP_TRY(ctx, hipEventCreate(&S.ev0)); HIP_TRY(ctx, hipEventCreate(&S.ev1)); }
    if (!S.dcount) HIP_TRY(ctx, hipMalloc(&S.dcount, sizeof(uint32_t) * 2));
    if (m <= S.cap) return RELMC_OK;
    for (void* p : {(void*)S.keys, (void*)S.flags, (void*)S.idx, S.tmp}) if (p) (void)hipFree(p);
    S.keys = nullptr; S.flags = nullptr; S.idx = nullptr; S.tmp = nullptr; S.cap = 0; S.tmp_bytes = 0;
    const int ow = mask_words(ctx);
    size_t tb = 0;
    (void)rocprim::select(nullptr, tb, rocprim::counting_iterator<uint32_t>(0u), (uint8_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)m, ScreenFlagSet(), ctx->stream);
    HIP_TRY(ctx, hipMalloc(&S.keys, sizeof(uint32_t) * (size_t)ow * (size_t)m));
    HIP_TRY(ctx, hipMalloc(&S.flags, (size_t)m));
    HIP_TRY(ctx, hipMalloc(&S.idx, sizeof(uint32_t) * (size_t)m));
    HIP_TRY(ctx, hipMalloc(&S.tmp, tb ? tb : 16));
    S.cap = m; S.tmp_bytes = tb;
    return RELMC_OK;
}

// idx[0 .. *n_out) = ascending positions i < m with flags[i] != 0; leaves the stream synchronised
int screen_select(relmc_ctx* ctx, int64_t m, uint32_t* n_out)
{
    auto& S = ctx->screen;
    size_t tb = S.tmp_bytes;
    HIP_TRY(ctx, rocprim::select(S.tmp, tb, rocprim::counting_iterator<uint32_t>(0u), S.flags, S.idx, S.dcount, (size_t)m, ScreenFlagSet(), ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(&ctx->hstage->fail_cnt, S.dcount, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *n_out = ctx->hstage->fail_cnt;
    return RELMC_OK;
}

int64_t grid256(const relmc_ctx* ctx, int64_t n, int per_cu = 16)
{
    int64_t g = (n + 255) / 256;
    if (g > (int64_t)ctx->num_cu * per_cu) g = (int64_t)ctx->num_cu * per_cu;
    return g < 1 ? 1 : g;
}
}  // namespace

// Pre-pass of the fused non-sequential path over the samples [first_index, first_index + m): masks of the uncovered samples in ctx->screen.keys (at the
// sample's own position), their ascending positions in ctx->screen.idx, *n_surv of them; *ms += the pre-pass' device time.
int screen_prepass_nsq(relmc_ctx* ctx, uint64_t seed, uint64_t first_index, int64_t m, uint32_t* n_surv, double* ms)
{
    int rc = screen_buffers(ctx, m);
    if (rc) return rc;
    auto& S = ctx->screen;
    HIP_TRY(ctx, hipEventRecord(S.ev0, ctx->stream));
    const int64_t g = grid256(ctx, m);
    if (ctx->tile == 0) hipLaunchKernelGGL(relmc_screen_sample_kernel<Tile24>, dim3((unsigned)g), dim3(256), 0, ctx->stream, reinterpret_cast<const DevCaseT<Tile24>*>(ctx->dcase), S.tab, seed, first_index, m, S.keys, S.flags);
    else hipLaunchKernelGGL(relmc_screen_sample_kernel<Tile96>, dim3((unsigned)g), dim3(256), 0, ctx->stream, reinterpret_cast<const DevCaseT<Tile96>*>(ctx->dcase), S.tab, seed, first_index, m, S.keys, S.flags);
    HIP_TRY(ctx, hipGetLastError());
    size_t tb = S.tmp_bytes;
    HIP_TRY(ctx, rocprim::select(S.tmp, tb, rocprim::counting_iterator<uint32_t>(0u), S.flags, S.idx, S.dcount, (size_t)m, ScreenFlagSet(), ctx->stream));
    HIP_TRY(ctx, hipEventRecord(S.ev1, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(&ctx->hstage->fail_cnt, S.dcount, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *n_surv = ctx->hstage->fail_cnt;
    float t = 0.f;
    if (ms && hipEventElapsedTime(&t, S.ev0, S.ev1) == hipSuccess) *ms += t;
    return RELMC_OK;
}

int screen_gather_keys(relmc_ctx* ctx, uint32_t n_surv, uint32_t* keys_out)
{
    if (n_surv == 0) return RELMC_OK;
    auto& S = ctx->screen;
    const int ow = mask_words(ctx);
    hipLaunchKernelGGL(relmc_screen_gather_keys_kernel, dim3((unsigned)grid256(ctx, (int64_t)n_surv * ow)), dim3(256), 0, ctx->stream, S.keys, S.idx, (int64_t)n_surv, ow, keys_out);
    HIP_TRY(ctx, hipGetLastError());
    return RELMC_OK;
}

// New database rows [first, first + n) (keys already in place): certified rows are filled in, the others' offsets from `first` listed ascending in
// ctx->screen.idx; *n_surv of them.
int screen_prepass_rows(relmc_ctx* ctx, int64_t first, int64_t n, uint32_t* n_surv)
{
    int rc = screen_buffers(ctx, n);
    if (rc) return rc;
    auto& S = ctx->screen;
    const int64_t g = grid256(ctx, n);
    if (ctx->tile == 0) hipLaunchKernelGGL(relmc_screen_rows_kernel<Tile24::OW>, dim3((unsigned)g), dim3(256), 0, ctx->stream, S.tab, ctx->db_keys, first, n, ctx->nb, ctx->db_dns, ctx->db_meta, ctx->db_nodal, S.flags);
    else hipLaunchKernelGGL(relmc_screen_rows_kernel<Tile96::OW>, dim3((unsigned)g), dim3(256), 0, ctx->stream, S.tab, ctx->db_keys, first, n, ctx->nb, ctx->db_dns, ctx->db_meta, ctx->db_nodal, S.flags);
    HIP_TRY(ctx, hipGetLastError());
    return screen_select(ctx, n, n_surv);
}

// seqMain.m:97-100 with the certificate: per year the count of contingency hours (ncont) and the listed hours the certificate does not cover (counts)
int screen_seq_compact(relmc_ctx* ctx, const uint32_t* masks, int n_years, uint16_t* hours, uint32_t* counts, uint32_t* ncont)
{
    // (timed between the context's screen events: relmc_seq_years adds it to the device time it reports once the stream has been synchronised)
    const int hpy = ctx->hseq.hpy;
    const int64_t total = (int64_t)n_years * hpy;
    int rc = screen_buffers(ctx, total);
    if (rc) return rc;
    auto& S = ctx->screen;
    const int64_t g = grid256(ctx, total);
    HIP_TRY(ctx, hipEventRecord(S.ev0, ctx->stream));
    if (ctx->tile == 0) hipLaunchKernelGGL(relmc_seq_flag_kernel<Tile24::OW>, dim3((unsigned)g), dim3(256), 0, ctx->stream, S.tab, masks, ctx->dlf, hpy, total, S.flags);
    else hipLaunchKernelGGL(relmc_seq_flag_kernel<Tile96::OW>, dim3((unsigned)g), dim3(256), 0, ctx->stream, S.tab, masks, ctx->dlf, hpy, total, S.flags);
    HIP_TRY(ctx, hipGetLastError());
    hipLaunchKernelGGL(relmc_seq_compact_flags_kernel, dim3(n_years), dim3(256), 0, ctx->stream, S.flags, hpy, hours, counts, ncont);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventRecord(S.ev1, ctx->stream));
    return RELMC_OK;
}

}  // namespace relmc_host

using namespace relmc_host;

extern "C" {

// Host-only introspection (no device, no context; not part of include/relmc.h): the certificate's tables as relmc_case_load builds them, for the CPU test
// suite to hold against numpy's PTDF / LODF.  out_doubles[2 ng + 4 nl + 2 nl ng + nl nl] in the layout of screen_tables, bridge_out[nl], sums_out[2] =
// {sum Pmin, sum range}.  Returns 1 = tables built, 0 = the case has no certificate, < 0 = bad arguments.
int32_t relmc_debug_screen_tables(const relmc_case_desc* d, double* out_doubles, int64_t cap, uint8_t* bridge_out, double* sums_out)
{
    if (!d || !out_doubles || !bridge_out || !sums_out) return RELMC_ERR_INVALID;
    std::vector<double> h; std::vector<uint8_t> bridge;
    if (!screen_tables(d, h, bridge, &sums_out[0], &sums_out[1])) return 0;
    if ((int64_t)h.size() > cap) return RELMC_ERR_INVALID;
    std::memcpy(out_doubles, h.data(), sizeof(double) * h.size());
    std::memcpy(bridge_out, bridge.data(), bridge.size());
    return 1;
}

// Test hook (not part of include/relmc.h): the device's certificate for given states.  states_host [n][ncomp] (1 = failed), load_scale_host optional
// [n] (the sequential track's hourly factor; null = 1) -> certified_host[n] (1 = the pre-screen would skip this state).
int32_t relmc_debug_screen_states(relmc_ctx* ctx, const uint8_t* states_host, const double* load_scale_host, int64_t n, uint8_t* certified_host)
{
    if (!ctx || !ctx->has_case || n < 0 || (n > 0 && (!states_host || !certified_host))) return RELMC_ERR_INVALID;
    if (n == 0) return RELMC_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = screen_buffers(ctx, n);
    if (rc) return rc;
    auto& S = ctx->screen;
    const int ow = mask_words(ctx), ncomp = ctx->ncomp;
    uint8_t* dst = nullptr; double* dsc = nullptr;
    HIP_TRY(ctx, hipMalloc(&dst, (size_t)n * ncomp));
    if (load_scale_host && hipMalloc(&dsc, sizeof(double) * (size_t)n) != hipSuccess) { (void)hipFree(dst); return fail(ctx, RELMC_ERR_HIP, "relmc_debug_screen_states: allocation failed"); }
    bool ok = hipMemcpyAsync(dst, states_host, (size_t)n * ncomp, hipMemcpyHostToDevice, ctx->stream) == hipSuccess &&
              (!dsc || hipMemcpyAsync(dsc, load_scale_host, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, ctx->stream) == hipSuccess);
    if (ok) {
        hipLaunchKernelGGL(relmc_screen_pack_kernel, dim3((unsigned)grid256(ctx, n * ow)), dim3(256), 0, ctx->stream, dst, n, ncomp, ow, S.keys);
        if (ctx->tile == 0) hipLaunchKernelGGL(relmc_screen_keys_kernel<Tile24::OW>, dim3((unsigned)grid256(ctx, n)), dim3(256), 0, ctx->stream, S.tab, S.keys, dsc, n, S.flags);
        else hipLaunchKernelGGL(relmc_screen_keys_kernel<Tile96::OW>, dim3((unsigned)grid256(ctx, n)), dim3(256), 0, ctx->stream, S.tab, S.keys, dsc, n, S.flags);
        ok = hipGetLastError() == hipSuccess && hipMemcpyAsync(certified_host, S.flags, (size_t)n, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess &&
             hipStreamSynchronize(ctx->stream) == hipSuccess;
    }
    (void)hipFree(dst); if (dsc) (void)hipFree(dsc);
    if (!ok) return fail(ctx, RELMC_ERR_HIP, "relmc_debug_screen_states: kernel / copy failed");
    for (int64_t i = 0; i < n; ++i) certified_host[i] = certified_host[i] ? 0 : 1;
    return RELMC_OK;
}

}  // extern "C"
