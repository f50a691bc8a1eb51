// relmc_kernels.hip — gfx950 (MI355X / CDNA4) kernels of the HL2 non-sequential Monte Carlo
// state-evaluation path: mc_sampling + mc_simulation of the reference
// (Montecarlo_nsq_single/mc_sampling.m:24-45, mc_simulation.m:32-99, nsqMain.m:270,282-301),
// i.e. Bernoulli outage sampling, the MATPOWER DC-OPF load-curtailment LP solved by the MIPS
// primal-dual interior-point iteration, and the index accumulators.
//
// Mapping: ONE SCENARIO PER 16-LANE DPP ROW, four scenarios per wavefront, one
// wavefront per workgroup.  The reduced symmetric KKT system of an IPM iteration
//     [ Mth  B' ] [dth ]   [ -Nth            ]      Mth = Bf' diag(mu/z) Bf  (weighted Laplacian)
//     [ B   -E  ] [dlam] = [ -g - C D^-1 Np  ]      E   = C D^-1 C'          (diagonal)
// (order 2*nb = 48: generator/load steps dp are eliminated analytically) lives entirely in VGPRs:
// KKT row rho = 16*slot + lane is held by one lane as 48 + 1 doubles per slot.  Gaussian
// elimination broadcasts the pivot row with DPP `row_newbcast` (v_mov_b64_dpp), so the
// n^3/3 update is pure v_fma_f64 at full 64-lane occupancy with no LDS traffic and no
// cross-lane shuffles through memory.  LDS only carries the sparse per-scenario vectors that
// need a gather/scatter (line <-> bus <-> injection incidence) and the static case tables.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "relmc_dev.h"

namespace relmc {

#define DEVFI __device__ __forceinline__

template <int CTRL>
DEVFI double dppd(double v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, true); }
template <int CTRL>
DEVFI uint32_t dppu(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xf, 0xf, true); }

// broadcast lane L of every 16-lane row to the whole row
#define BCAST(L, x) dppd<0x150 + (L)>(x)
// all-reduce over the 16 lanes of a row (row_ror 8,4,2,1); every lane gets bit-identical results
DEVFI double row_sum(double v) { v += dppd<0x128>(v); v += dppd<0x124>(v); v += dppd<0x122>(v); v += dppd<0x121>(v); return v; }
DEVFI double row_max(double v) { v = __builtin_fmax(v, dppd<0x128>(v)); v = __builtin_fmax(v, dppd<0x124>(v)); v = __builtin_fmax(v, dppd<0x122>(v)); v = __builtin_fmax(v, dppd<0x121>(v)); return v; }
DEVFI double row_min(double v) { v = __builtin_fmin(v, dppd<0x128>(v)); v = __builtin_fmin(v, dppd<0x124>(v)); v = __builtin_fmin(v, dppd<0x122>(v)); v = __builtin_fmin(v, dppd<0x121>(v)); return v; }
DEVFI uint32_t row_or(uint32_t v) { v |= dppu<0x128>(v); v |= dppu<0x124>(v); v |= dppu<0x122>(v); v |= dppu<0x121>(v); return v; }
DEVFI uint32_t row_add(uint32_t v) { v += dppu<0x128>(v); v += dppu<0x124>(v); v += dppu<0x122>(v); v += dppu<0x121>(v); return v; }

// 1/x to ~1 ulp: v_rcp_f64 + two Newton steps (no IEEE division sequence)
DEVFI double frcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(r, e, r);
    return r;
}

// Philox4x32-10 (Salmon et al. SC'11); counter (i_lo, i_hi, block, 0), key = seed
DEVFI void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4])
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
        const uint32_t h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
        c0 = n0; c1 = l1; c2 = n2; c3 = l0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

DEVFI bool outbit(uint32_t o0, uint32_t o1, uint32_t o2, uint32_t o3, int k)
{
    const uint32_t w = k < 64 ? (k < 32 ? o0 : o1) : (k < 96 ? o2 : o3);
    return (w >> (k & 31)) & 1u;
}

// per-scenario LDS scratch (one per DPP row); 917 doubles = odd stride -> rows land on different banks
struct ScenLds {
    double Mcomb[CARR], Bcomb[CARR], Bswap[CARR], Ecomb[CARR];
    double Lg[NLT], Llx[NLT], Lq[NLT], LF[NLT];
    double Ip[NIT], IinvD[NIT], INpD[NIT];
    double sol[2 * NBT];
    double pad;
};

#include "elim_nb24.inc"

#define DINF __builtin_inf()

template <bool FROM_RNG, bool WRITE_OUT>
__global__ void __launch_bounds__(64) relmc_eval_kernel(const DevCase* __restrict__ gcase, const EvalArgs a)
{
    __shared__ DevCase C;
    __shared__ ScenLds SC[4];
    const int lane = threadIdx.x, rlane = lane & 15, row = lane >> 4;
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(gcase);
        uint32_t* dst = reinterpret_cast<uint32_t*>(&C);
        for (int i = lane; i < (int)(sizeof(DevCase) / 4); i += 64) dst[i] = src[i];
    }
    ScenLds& S = SC[row];
    for (int i = rlane; i < CARR; i += ROWL) { S.Mcomb[i] = 0.0; S.Bcomb[i] = 0.0; S.Bswap[i] = 0.0; S.Ecomb[i] = 0.0; }
    __syncthreads();

    const int ng = C.ng, ncomp = C.ncomp, nb = C.nb;
    const double base = C.base_mva;
    const double eps = 2.220446049250313e-16;

    // ---- static per-lane tables -------------------------------------------------------
    uint32_t linfo[LS]; int lpart[LS]; double lb[LS], lr[LS];
#pragma unroll
    for (int s = 0; s < LS; ++s) { const int l = 16 * s + rlane; linfo[s] = C.l_info[l]; lpart[s] = C.l_partner[l]; lb[s] = C.l_b[l]; lr[s] = C.l_rate[l]; }
    uint32_t iinfo[IS];
#pragma unroll
    for (int s = 0; s < IS; ++s) iinfo[s] = C.i_info[16 * s + rlane];
    // KKT rows: slot0 = theta(bus rlane); slot1 = theta(bus 16+rlane) for rlane<8, lambda(bus 8+rlane) else;
    //           slot2 = lambda(bus rlane)
    const bool s1lam = rlane >= 8;
    const int s1bus = 16 + (rlane & 7);
#define TROW(s) ((s) == 1 ? &C.T[s1bus][0] : &C.T[rlane][0])
#define ARR_A(s) ((s) == 0 ? S.Mcomb : ((s) == 2 ? S.Bcomb : (s1lam ? S.Bcomb : S.Mcomb)))
#define ARR_B(s) ((s) == 0 ? S.Bswap : ((s) == 2 ? S.Ecomb : (s1lam ? S.Ecomb : S.Bswap)))
    // symbolic fill masks of the block elimination, wave-uniform (SGPRs): guards of gen_elim.py
#define RELMC_FM(i) const uint32_t fmask_##i = __builtin_amdgcn_readfirstlane(C.fill[i]);
    RELMC_FM(0) RELMC_FM(1) RELMC_FM(2) RELMC_FM(3) RELMC_FM(4) RELMC_FM(5) RELMC_FM(6) RELMC_FM(7)
    RELMC_FM(8) RELMC_FM(9) RELMC_FM(10) RELMC_FM(11) RELMC_FM(12) RELMC_FM(13) RELMC_FM(14) RELMC_FM(15)
    RELMC_FM(16) RELMC_FM(17) RELMC_FM(18) RELMC_FM(19) RELMC_FM(20) RELMC_FM(21) RELMC_FM(22) RELMC_FM(23)
#define FILLMASK(i) fmask_##i

    // ---- accumulators (nsqMain.m:282-301 in per-sample form) ---------------------------
    double acc_dns = 0.0, acc_dns2 = 0.0, acc_shed[IS] = {0.0, 0.0, 0.0, 0.0};
    uint32_t acc_n = 0, acc_nfail = 0, acc_nsing = 0, acc_ninf = 0, acc_nnc = 0, acc_iters = 0;
    uint32_t acc_cfi[IS] = {0, 0, 0, 0}, acc_cfl[LS] = {0, 0, 0};

    const int64_t ngroups = (a.n + 3) >> 2;
    for (int64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const int64_t sidx = grp * 4 + row;
        const bool live = sidx < a.n;

        // per-scenario state ------------------------------------------------------------
        bool l_on[LS], l_act[LS], i_on[IS], i_box[IS];
        double LFv[LS], LGv[LS], lzp[LS], lzm[LS], lmup[LS], lmum[LS];
        double ip[IS], ilo[IS], ilam[IS], izp[IS], izm[IS], imup[IS], imum[IS];
        double BV_0 = 0.0, BV_1 = 0.0, BV_2 = 0.0;
        RELMC_K_DECL
        uint32_t o0 = 0, o1 = 0, o2 = 0, o3 = 0, pinned = 0, dropped = 0;
        double gamma = 1.0, fval = 0.0, f0 = 0.0, alphap = 1.0, alphad = 1.0, zmu = 0.0;
        uint32_t niq = 0;
        int it = 0, status = 0;
        bool infeas = false, singular = false, iterating = false;
#pragma unroll
        for (int s = 0; s < LS; ++s) { l_on[s] = false; l_act[s] = false; LFv[s] = 0; LGv[s] = 0; lzp[s] = 1; lzm[s] = 1; lmup[s] = 1; lmum[s] = 1; }
#pragma unroll
        for (int s = 0; s < IS; ++s) { i_on[s] = false; i_box[s] = false; ip[s] = 0; ilo[s] = 0; ilam[s] = 0; izp[s] = 1; izm[s] = 1; imup[s] = 1; imum[s] = 1; }

        if (live) {
            // ===== mc_sampling.m:24-41: Bernoulli outage state (1 = failed) ==================
            if (FROM_RNG) {
                const uint64_t gi = a.first_index + (uint64_t)sidx;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int blk = rlane + 16 * h;
                    uint32_t nib = 0;
                    if (blk * 4 < ncomp) {
                        uint32_t w[4];
                        philox4x32_10((uint32_t)gi, (uint32_t)(gi >> 32), (uint32_t)blk, 0u, (uint32_t)a.seed, (uint32_t)(a.seed >> 32), w);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int k = blk * 4 + e;
                            if (k < ncomp && w[e] < C.thr[k]) nib |= 1u << e;   // strict '<', mc_sampling.m:35
                        }
                    }
                    const uint32_t sh = nib << ((rlane & 7) * 4);
                    if (h == 0) { if (rlane < 8) o0 |= sh; else o1 |= sh; }
                    else { if (rlane < 8) o2 |= sh; else o3 |= sh; }
                }
            } else {
                const uint8_t* st = a.states + sidx * ncomp;
#pragma unroll
                for (int q = 0; q < NCOMPMAX / 16; ++q) {
                    const int k = rlane + 16 * q;
                    if (k < ncomp && st[k]) {
                        const uint32_t bit = 1u << (k & 31);
                        if (q < 2) o0 |= bit; else if (q < 4) o1 |= bit; else if (q < 6) o2 |= bit; else o3 |= bit;
                    }
                }
            }
            o0 = row_or(o0); o1 = row_or(o1); o2 = row_or(o2); o3 = row_or(o3);

            // ===== mc_simulation.m:32-37: component status -> model ==========================
#pragma unroll
            for (int s = 0; s < LS; ++s) {
                const int l = 16 * s + rlane;
                const uint32_t fl = linfo[s] >> 24;
                l_on[s] = (fl & LF_EXISTS) && !outbit(o0, o1, o2, o3, ng + l);
                l_act[s] = l_on[s] && (fl & LF_LIMITED);
            }
#pragma unroll
            for (int s = 0; s < IS; ++s) {
                const int j = 16 * s + rlane;
                const uint32_t kind = (iinfo[s] >> 8) & 0xff;
                i_on[s] = kind == IK_VIRTUAL || (kind == IK_REAL && !outbit(o0, o1, o2, o3, j));
                ilo[s] = C.i_lo[j];
            }

            // ===== topology: adjacency, isolated buses, islands ===============================
            uint32_t adj0 = 0, adj1 = 0;
            bool iso = false;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int i = 16 * t + rlane;
                uint32_t adj = 0;
                if (i < NBT && ((C.exist_mask >> i) & 1u)) {
                    const int nlb = C.b_nline[i];
                    for (int e = 0; e < nlb; ++e) {
                        const uint32_t ent = C.b_line[i][e];
                        const int l = ent & 0x7f;
                        if (!outbit(o0, o1, o2, o3, ng + l)) {
                            const uint32_t inf = C.l_info[l];
                            adj |= 1u << ((ent & 0x80) ? (inf & 0xff) : ((inf >> 8) & 0xff));
                        }
                    }
                    iso = iso || adj == 0;
                }
                if (t == 0) adj0 = adj; else adj1 = adj;
            }
            const uint64_t isob = __ballot(iso);
            const bool iso_any = ((isob >> (16 * row)) & 0xffffull) != 0;
            // a bus without any in-service branch makes MATPOWER's KKT matrix exactly singular; the
            // reference consumes the start point (mc_simulation.m:41,54)
            singular = (a.policy == 0) && iso_any;
            pinned = ~C.exist_mask & 0xffffffu;
            dropped = pinned;

            if (!singular) {
                uint32_t remaining = C.exist_mask;
                for (int guard = 0; guard < NBT; ++guard) {
                    const bool more = remaining != 0;
                    if (!__any(more)) break;
                    if (more) {
                        uint32_t R = 1u << (__ffs((int)remaining) - 1);
                        for (int sweep = 0; sweep < NBT; ++sweep) {
                            uint32_t c = 0;
                            if ((R >> rlane) & 1u) c |= adj0;
                            if (rlane < 8 && ((R >> (16 + rlane)) & 1u)) c |= adj1;
                            const uint32_t Rn = R | row_or(c);
                            const bool ch = Rn != R;
                            R = Rn;
                            if (!__any(ch)) break;
                        }
                        // rule 1: the island's angle reference is its bus that is eliminated last
                        // (sequence = internal buses 16..23, 0..15 = the host's min-fill order with the
                        // reference bus at the very end, internal 15)
                        const uint32_t Rlo = R & 0xffffu;
                        const int pin = 31 - __clz((int)(Rlo ? Rlo : R));
                        // island rules 2-5 (DESIGN.md "island policy")
                        uint32_t cnt = 0; double losum = 0.0; bool inI[IS];
#pragma unroll
                        for (int s = 0; s < IS; ++s) {
                            const int j = 16 * s + rlane;
                            const uint32_t kind = (iinfo[s] >> 8) & 0xff;
                            inI[s] = i_on[s] && ((R >> (iinfo[s] & 0xff)) & 1u);
                            if (inI[s]) {
                                cnt += 1u;
                                if (kind == IK_VIRTUAL) cnt += 1u << 8; else if (C.i_hi[j] > 0.0) cnt += 1u << 16;
                                losum += C.i_pmin_mw[j];
                            }
                        }
                        cnt = row_add(cnt); losum = row_sum(losum);
                        const uint32_t n_inj = cnt & 0xff, n_load = (cnt >> 8) & 0xff, n_gen = cnt >> 16;
                        if (n_inj && !n_load) {                  // rule 2: no load -> decommit the island's units
#pragma unroll
                            for (int s = 0; s < IS; ++s) if (inI[s]) { i_on[s] = false; inI[s] = false; }
                            infeas = true;
                        } else if (n_load && !n_gen) {           // rule 3: no generation -> all load shed (p fixed 0)
#pragma unroll
                            for (int s = 0; s < IS; ++s) if (inI[s] && ((iinfo[s] >> 8) & 0xff) == IK_VIRTUAL) ilo[s] = 0.0;
                        } else if (losum > 1e-9) {               // rule 4: over-generation -> relax Pmin
#pragma unroll
                            for (int s = 0; s < IS; ++s) if (inI[s] && ((iinfo[s] >> 8) & 0xff) == IK_REAL) ilo[s] = 0.0;
                            infeas = true;
                        }
                        uint32_t nfree = 0;
#pragma unroll
                        for (int s = 0; s < IS; ++s) if (inI[s] && C.i_hi[16 * s + rlane] - ilo[s] > 0.0) nfree += 1u;
                        nfree = row_add(nfree);
                        if (!nfree) dropped |= 1u << pin;        // rule 5: dependent balance rows
                        pinned |= 1u << pin;
                        remaining &= ~R;
                    }
                }
            }

            // ===== constant (per scenario) susceptance blocks: Bbus with pins/drops masked ======
#pragma unroll
            for (int s = 0; s < LS; ++s) {
                const uint32_t inf = linfo[s];
                if ((inf >> 24) & LF_OWNER) {
                    const int f = inf & 0xff, t = (inf >> 8) & 0xff, p = (inf >> 16) & 0xff;
                    double v = l_on[s] ? lb[s] : 0.0;
                    const int pr = lpart[s];
                    if (pr >= 0 && !outbit(o0, o1, o2, o3, ng + pr)) v += C.l_b[pr];
                    v = -v;
                    const int lo_b = f < t ? f : t, hi_b = f < t ? t : f;
                    const double vlo = (((pinned >> hi_b) | (dropped >> lo_b)) & 1u) ? 0.0 : v;   // (row lam_lo, col th_hi)
                    const double vhi = (((pinned >> lo_b) | (dropped >> hi_b)) & 1u) ? 0.0 : v;   // (row lam_hi, col th_lo)
                    S.Bcomb[p] = vlo; S.Bcomb[PMAX + p] = vhi;
                    S.Bswap[p] = vhi; S.Bswap[PMAX + p] = vlo;
                }
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int i = 16 * t + rlane;
                if (i < NBT) {
                    double d = 0.0;
                    const int nlb = C.b_nline[i];
                    for (int e = 0; e < nlb; ++e) {
                        const int l = C.b_line[i][e] & 0x7f;
                        if (!outbit(o0, o1, o2, o3, ng + l)) d += C.l_b[l];
                    }
                    if (((pinned | dropped) >> i) & 1u) d = 0.0;
                    S.Bcomb[DIAG0 + i] = d; S.Bswap[DIAG0 + i] = d;
                }
            }

            // ===== dcopf_solver start point + mips initialisation (SURVEY.md Appendix B 2,4) =====
            double fl = 0.0;
            uint32_t nq = 0;
#pragma unroll
            for (int s = 0; s < LS; ++s) {
                if (l_act[s]) {
                    const double h = -lr[s];                       // x0: all angles 0 -> flow 0
                    double z = a.z0; if (h < -a.z0) z = -h;
                    double mu = a.z0; if (1.0 / z > a.z0) mu = 1.0 / z;
                    lzp[s] = z; lzm[s] = z; lmup[s] = mu; lmum[s] = mu;
                    nq += 2;
                }
            }
#pragma unroll
            for (int s = 0; s < IS; ++s) {
                const int j = 16 * s + rlane;
                if (i_on[s]) {
                    const double hi = C.i_hi[j], lo = ilo[s];
                    i_box[s] = hi - lo > eps;
                    ip[s] = i_box[s] ? 0.5 * (lo + hi) : hi;
                    if (i_box[s]) {
                        const double h = -0.5 * (hi - lo);
                        double z = a.z0; if (h < -a.z0) z = -h;
                        double mu = a.z0; if (1.0 / z > a.z0) mu = 1.0 / z;
                        izp[s] = z; izm[s] = z; imup[s] = mu; imum[s] = mu;
                        nq += 2;
                    }
                    fl += C.i_cost[j] * ip[s];
                }
            }
            niq = row_add(nq);
            fval = row_sum(fl);
            f0 = fval;
            iterating = !singular;
            status = singular ? 3 : 0;
        }

        // ===== mips main loop =======================================
        while (__any(iterating)) {
            if (iterating) {
                // ---- evaluate h, Lx, barrier terms; scatter to LDS; convergence norms -------------
                double mx_gh = -DINF, mx_x = 0.0, mx_z = 0.0, mx_lx = 0.0, mx_lammu = 0.0;
                bool nanx = false;
#pragma unroll
                for (int s = 0; s < LS; ++s) {
                    const int l = 16 * s + rlane;
                    double g = 0.0, lx = 0.0, q = 0.0;
                    if (l_on[s]) {
                        lx = LGv[s];
                        if (l_act[s]) {
                            const double b = lb[s], rr = lr[s];
                            const double hp = LFv[s] - rr, hm = -LFv[s] - rr;
                            const double rzp = frcp(lzp[s]), rzm = frcp(lzm[s]);
                            g = b * b * (lmup[s] * rzp + lmum[s] * rzm);
                            lx = __builtin_fma(b, lmup[s] - lmum[s], lx);
                            q = b * ((lmup[s] * hp + gamma) * rzp - (lmum[s] * hm + gamma) * rzm);
                            mx_gh = __builtin_fmax(mx_gh, __builtin_fmax(hp, hm));
                            mx_z = __builtin_fmax(mx_z, __builtin_fmax(lzp[s], lzm[s]));
                            mx_lammu = __builtin_fmax(mx_lammu, __builtin_fmax(lmup[s], lmum[s]));
                        }
                    }
                    S.Lg[l] = g; S.Llx[l] = lx; S.Lq[l] = lx + q; S.LF[l] = LFv[s];
                }
#pragma unroll
                for (int s = 0; s < IS; ++s) {
                    const int j = 16 * s + rlane;
                    double invD = 0.0, npd = 0.0, pv = 0.0;
                    if (i_on[s]) {
                        pv = ip[s];
                        mx_x = __builtin_fmax(mx_x, __builtin_fabs(pv));
                        nanx = nanx || pv != pv;
                        if (i_box[s]) {
                            const double hp = pv - C.i_hi[j], hm = ilo[s] - pv;
                            const double rzp = frcp(izp[s]), rzm = frcp(izm[s]);
                            const double D = imup[s] * rzp + imum[s] * rzm;
                            const double lxp = C.i_cost[j] - ilam[s] + (imup[s] - imum[s]);
                            const double np = lxp + (imup[s] * hp + gamma) * rzp - (imum[s] * hm + gamma) * rzm;
                            invD = frcp(D); npd = np * invD;
                            mx_lx = __builtin_fmax(mx_lx, __builtin_fabs(lxp));
                            mx_gh = __builtin_fmax(mx_gh, __builtin_fmax(hp, hm));
                            mx_z = __builtin_fmax(mx_z, __builtin_fmax(izp[s], izm[s]));
                            mx_lammu = __builtin_fmax(mx_lammu, __builtin_fmax(imup[s], imum[s]));
                        }
                    }
                    S.Ip[j] = pv; S.IinvD[j] = invD; S.INpD[j] = npd;
                }
                // weighted-Laplacian off-diagonals (pair owners), pinned columns removed
#pragma unroll
                for (int s = 0; s < LS; ++s) {
                    const uint32_t inf = linfo[s];
                    if ((inf >> 24) & LF_OWNER) {
                        const int l = 16 * s + rlane, f = inf & 0xff, t = (inf >> 8) & 0xff, p = (inf >> 16) & 0xff;
                        double gs = S.Lg[l];
                        if (lpart[s] >= 0) gs += S.Lg[lpart[s]];
                        const double v = (((pinned >> f) | (pinned >> t)) & 1u) ? 0.0 : -gs;
                        S.Mcomb[p] = v; S.Mcomb[PMAX + p] = v;
                    }
                }
                // bus gathers -> KKT diagonals and right-hand sides
                auto theta_eval = [&](const int bi, const double bv, double& rhs) {
                    double md = 0.0, lx = 0.0, nq_ = 0.0;
                    const int nlb = C.b_nline[bi];
                    for (int e = 0; e < nlb; ++e) {
                        const uint32_t ent = C.b_line[bi][e];
                        const int l = ent & 0x7f;
                        const double sg = (ent & 0x80) ? -1.0 : 1.0;
                        md += S.Lg[l];
                        lx = __builtin_fma(sg, S.Llx[l], lx);
                        nq_ = __builtin_fma(sg, S.Lq[l], nq_);
                    }
                    if ((pinned >> bi) & 1u) {        // fixed angle: identity row; its multiplier is -lx
                        S.Mcomb[DIAG0 + bi] = 1.0; rhs = 0.0;
                        mx_lammu = __builtin_fmax(mx_lammu, __builtin_fabs(lx));
                    } else {
                        S.Mcomb[DIAG0 + bi] = md; rhs = -nq_;
                        mx_lx = __builtin_fmax(mx_lx, __builtin_fabs(lx));
                    }
                    mx_x = __builtin_fmax(mx_x, __builtin_fabs(bv));
                    nanx = nanx || bv != bv;
                };
                auto lambda_eval = [&](const int bi, const double bv, double& rhs) {
                    double bal = 0.0, E = 0.0, ssum = 0.0;
                    const int nlb = C.b_nline[bi];
                    for (int e = 0; e < nlb; ++e) {
                        const uint32_t ent = C.b_line[bi][e];
                        bal = __builtin_fma((ent & 0x80) ? -1.0 : 1.0, S.LF[ent & 0x7f], bal);
                    }
                    const int nib = C.b_ninj[bi];
                    for (int e = 0; e < nib; ++e) {
                        const int j = C.b_inj[bi][e];
                        bal -= S.Ip[j]; E += S.IinvD[j]; ssum += S.INpD[j];
                    }
                    if ((dropped >> bi) & 1u) {       // dependent / non-existent balance row
                        S.Ecomb[DIAG0 + bi] = -1.0; rhs = 0.0;
                    } else {
                        S.Ecomb[DIAG0 + bi] = -E; rhs = -bal - ssum;
                        mx_gh = __builtin_fmax(mx_gh, __builtin_fabs(bal));
                        mx_lammu = __builtin_fmax(mx_lammu, __builtin_fabs(bv));
                    }
                };
                theta_eval(rlane, BV_0, RHS_0);
                if (!s1lam) theta_eval(s1bus, BV_1, RHS_1); else lambda_eval(s1bus, BV_1, RHS_1);
                lambda_eval(rlane, BV_2, RHS_2);

                // ---- convergence test (mips.m feascond/gradcond/compcond/costcond) --------------
                mx_gh = row_max(mx_gh); mx_x = row_max(mx_x); mx_z = row_max(mx_z);
                mx_lx = row_max(mx_lx); mx_lammu = row_max(mx_lammu);
                const uint64_t nanb = __ballot(nanx);
                const bool xnan = ((nanb >> (16 * row)) & 0xffffull) != 0;
                const double feascond = mx_gh / (1.0 + __builtin_fmax(mx_x, mx_z));
                const double gradcond = mx_lx / (1.0 + mx_lammu);
                const double compcond = zmu / (1.0 + mx_x);
                const double costcond = __builtin_fabs(fval - f0) / (1.0 + __builtin_fabs(f0));
                const bool conv = it > 0 && feascond < a.feastol && gradcond < a.gradtol && compcond < a.comptol && costcond < a.costtol;
#ifdef RELMC_ABLATE_FIXIT
                if (it >= RELMC_ABLATE_FIXIT) { status = 0; iterating = false; }
                else if (false) {}
#else
                if (conv) { status = 0; iterating = false; }
#endif
                else if (it > 0 && (xnan || alphap < a.alpha_min || alphad < a.alpha_min || gamma < eps || gamma > 1.0 / eps)) {
#ifdef RELMC_DEBUG_STATUS
                    status = xnan ? 10 : alphap < a.alpha_min ? 11 : alphad < a.alpha_min ? 12 : gamma < eps ? 13 : 14;
#else
                    status = 2;
#endif
                    iterating = false;
                }
                else if (it >= a.max_it) { status = 1; iterating = false; }
            }
            if (iterating) {
                f0 = fval;
                it += 1;
                // ---- Newton step: assemble, eliminate, back-substitute (all in VGPRs) -----------
#ifndef RELMC_ABLATE_NO_ASSEMBLE
                RELMC_K_ASSEMBLE
#else
                { double seed_ = S.Mcomb[DIAG0 + rlane];
#define RELMC_SEEDK(sv, cv) K_##sv##_##cv = seed_ + cv;
                  RELMC_K_FOREACH(RELMC_SEEDK) }
#endif
#ifndef RELMC_ABLATE_NO_ELIM
                RELMC_K_ELIMINATE
                RELMC_K_BACKSUB
#else
                { double acc_ = 0.0;
#define RELMC_SUMK(sv, cv) acc_ += K_##sv##_##cv;
                  RELMC_K_FOREACH(RELMC_SUMK)
                  SOL_0 = RHS_0 * 1e-3 + acc_ * 1e-30; SOL_1 = RHS_1 * 1e-3; SOL_2 = RHS_2 * 1e-3; }
#endif
                S.sol[rlane] = SOL_0; S.sol[16 + rlane] = SOL_1; S.sol[32 + rlane] = SOL_2;
                double step2 = SOL_0 * SOL_0 + SOL_1 * SOL_1 + SOL_2 * SOL_2;
                double rp = DINF, rd = DINF;
                double dF[LS], dG[LS], ldzp[LS], ldzm[LS], ldmup[LS], ldmum[LS];
#pragma unroll
                for (int s = 0; s < LS; ++s) {
                    dF[s] = 0; dG[s] = 0; ldzp[s] = 0; ldzm[s] = 0; ldmup[s] = 0; ldmum[s] = 0;
                    if (l_on[s]) {
                        const int f = linfo[s] & 0xff, t = (linfo[s] >> 8) & 0xff;
                        const int lf = f < 16 ? 32 + f : f + 8, lt = t < 16 ? 32 + t : t + 8;
                        dF[s] = lb[s] * (S.sol[f] - S.sol[t]);
                        dG[s] = lb[s] * (S.sol[lf] - S.sol[lt]);
                        if (l_act[s]) {
                            const double hp = LFv[s] - lr[s], hm = -LFv[s] - lr[s];
                            const double rzp = frcp(lzp[s]), rzm = frcp(lzm[s]);
                            ldzp[s] = -hp - lzp[s] - dF[s];
                            ldzm[s] = -hm - lzm[s] + dF[s];
                            ldmup[s] = -lmup[s] + (gamma - lmup[s] * ldzp[s]) * rzp;
                            ldmum[s] = -lmum[s] + (gamma - lmum[s] * ldzm[s]) * rzm;
                            if (ldzp[s] < 0.0) rp = __builtin_fmin(rp, lzp[s] * frcp(-ldzp[s]));
                            if (ldzm[s] < 0.0) rp = __builtin_fmin(rp, lzm[s] * frcp(-ldzm[s]));
                            if (ldmup[s] < 0.0) rd = __builtin_fmin(rd, lmup[s] * frcp(-ldmup[s]));
                            if (ldmum[s] < 0.0) rd = __builtin_fmin(rd, lmum[s] * frcp(-ldmum[s]));
                        }
                    }
                }
                double dpv[IS], dlb[IS], idzp[IS], idzm[IS], idmup[IS], idmum[IS];
#pragma unroll
                for (int s = 0; s < IS; ++s) {
                    const int j = 16 * s + rlane;
                    dpv[s] = 0; dlb[s] = 0; idzp[s] = 0; idzm[s] = 0; idmup[s] = 0; idmum[s] = 0;
                    if (i_on[s]) {
                        const int bi = iinfo[s] & 0xff;
                        dlb[s] = S.sol[bi < 16 ? 32 + bi : bi + 8];
                        if (i_box[s]) {
                            dpv[s] = __builtin_fma(dlb[s], S.IinvD[j], -S.INpD[j]);   // dp = (-Np + dlam)/D
                            const double hp = ip[s] - C.i_hi[j], hm = ilo[s] - ip[s];
                            const double rzp = frcp(izp[s]), rzm = frcp(izm[s]);
                            idzp[s] = -hp - izp[s] - dpv[s];
                            idzm[s] = -hm - izm[s] + dpv[s];
                            idmup[s] = -imup[s] + (gamma - imup[s] * idzp[s]) * rzp;
                            idmum[s] = -imum[s] + (gamma - imum[s] * idzm[s]) * rzm;
                            if (idzp[s] < 0.0) rp = __builtin_fmin(rp, izp[s] * frcp(-idzp[s]));
                            if (idzm[s] < 0.0) rp = __builtin_fmin(rp, izm[s] * frcp(-idzm[s]));
                            if (idmup[s] < 0.0) rd = __builtin_fmin(rd, imup[s] * frcp(-idmup[s]));
                            if (idmum[s] < 0.0) rd = __builtin_fmin(rd, imum[s] * frcp(-idmum[s]));
                            step2 = __builtin_fma(dpv[s], dpv[s], step2);
                        }
                    }
                }
                step2 = row_sum(step2);
                if (!(step2 <= a.max_stepsize * a.max_stepsize)) {
                    // NaN or |dxdlam| > max_stepsize: "numerically failed", x is NOT updated
#ifdef RELMC_DEBUG_STATUS
                    status = step2 != step2 ? 20 : 21;
#else
                    status = 2;
#endif
                    iterating = false;
                } else {
                    rp = row_min(rp); rd = row_min(rd);
                    alphap = __builtin_fmin(a.xi * rp, 1.0);
                    alphad = __builtin_fmin(a.xi * rd, 1.0);
                    double zl = 0.0, fl = 0.0;
#pragma unroll
                    for (int s = 0; s < LS; ++s) {
                        if (l_on[s]) {
                            LFv[s] = __builtin_fma(alphap, dF[s], LFv[s]);
                            LGv[s] = __builtin_fma(alphad, dG[s], LGv[s]);
                            if (l_act[s]) {
                                lzp[s] = __builtin_fma(alphap, ldzp[s], lzp[s]); lzm[s] = __builtin_fma(alphap, ldzm[s], lzm[s]);
                                lmup[s] = __builtin_fma(alphad, ldmup[s], lmup[s]); lmum[s] = __builtin_fma(alphad, ldmum[s], lmum[s]);
                                zl = __builtin_fma(lzp[s], lmup[s], zl); zl = __builtin_fma(lzm[s], lmum[s], zl);
                            }
                        }
                    }
#pragma unroll
                    for (int s = 0; s < IS; ++s) {
                        if (i_on[s]) {
                            ilam[s] = __builtin_fma(alphad, dlb[s], ilam[s]);
                            if (i_box[s]) {
                                ip[s] = __builtin_fma(alphap, dpv[s], ip[s]);
                                izp[s] = __builtin_fma(alphap, idzp[s], izp[s]); izm[s] = __builtin_fma(alphap, idzm[s], izm[s]);
                                imup[s] = __builtin_fma(alphad, idmup[s], imup[s]); imum[s] = __builtin_fma(alphad, idmum[s], imum[s]);
                                zl = __builtin_fma(izp[s], imup[s], zl); zl = __builtin_fma(izm[s], imum[s], zl);
                            }
                            fl = __builtin_fma(C.i_cost[16 * s + rlane], ip[s], fl);
                        }
                    }
                    BV_0 = __builtin_fma(alphap, SOL_0, BV_0);
                    BV_1 = __builtin_fma(s1lam ? alphad : alphap, SOL_1, BV_1);
                    BV_2 = __builtin_fma(alphad, SOL_2, BV_2);
                    zmu = row_sum(zl);
                    fval = row_sum(fl);
                    if (niq > 0) gamma = a.sigma * zmu / (double)niq;
                }
            }
        }

        // ===== mc_simulation.m:54-99 (dns, noise filters, nodal shed) + nsqMain.m:270 =============
        if (live) {
            double dns = fval + C.total_load;
            if (dns < 0.1) dns = 0.0;
            const bool fail = dns > 1e-4;
            double shed[IS];
#pragma unroll
            for (int s = 0; s < IS; ++s) {
                const int j = 16 * s + rlane;
                shed[s] = 0.0;
                if (((iinfo[s] >> 8) & 0xff) == IK_VIRTUAL && dns > 0.0) {
                    const double v = ip[s] * base - C.i_pmin_mw[j];     // Pg - Pmin, mc_simulation.m:86
                    if (v > 1e-3) shed[s] = v;                           // mc_simulation.m:90
                }
                acc_shed[s] += shed[s];
                if (fail && ((iinfo[s] >> 8) & 0xff) == IK_REAL && outbit(o0, o1, o2, o3, j)) acc_cfi[s] += 1;
            }
#pragma unroll
            for (int s = 0; s < LS; ++s)
                if (fail && ((linfo[s] >> 24) & LF_EXISTS) && outbit(o0, o1, o2, o3, ng + 16 * s + rlane)) acc_cfl[s] += 1;
            acc_n += 1;
            acc_dns += dns; acc_dns2 = __builtin_fma(dns, dns, acc_dns2);
            acc_nfail += fail ? 1 : 0;
            acc_nsing += status == 3 ? 1 : 0;
            acc_nnc += (status == 1 || status == 2) ? 1 : 0;
            acc_ninf += infeas ? 1 : 0;
            acc_iters += (uint32_t)it;
            if (WRITE_OUT) {
#pragma unroll
                for (int s = 0; s < IS; ++s) S.Ip[16 * s + rlane] = shed[s];
                if (rlane == 0) {
                    a.dns[sidx] = dns;
                    if (a.status) a.status[sidx] = status;
                    if (a.iters) a.iters[sidx] = it;
                }
                if (a.nodal) {
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const int i = 16 * t + rlane;
                        if (i < NBT && ((C.exist_mask >> i) & 1u)) {
                            const int vj = C.b_vinj[i];
                            a.nodal[sidx * nb + C.b_ext[i]] = vj >= 0 ? S.Ip[vj] : 0.0;
                        }
                    }
                }
            }
        }
    }

    // ---- per-lane partials; relmc_finalize_kernel reduces them in a fixed order ----------------
    Partial& P = a.partial[(size_t)blockIdx.x * 64 + lane];
    P.dns = acc_dns; P.dns2 = acc_dns2;
#pragma unroll
    for (int s = 0; s < IS; ++s) { P.shed[s] = acc_shed[s]; P.cf_inj[s] = acc_cfi[s]; }
#pragma unroll
    for (int s = 0; s < LS; ++s) P.cf_line[s] = acc_cfl[s];
    P.n = acc_n; P.nfail = acc_nfail; P.nsing = acc_nsing; P.ninf = acc_ninf; P.nnc = acc_nnc; P.iters = acc_iters;
    P.pad = 0;
}

// device image of relmc_acc (include/relmc.h): 6 + 256 int64, then 2 + 128 doubles
struct DevAcc {
    long long n, n_fail, n_singular, n_infeasible, n_nonconverged, sum_iters;
    long long comp_fail[256];
    double sum_dns, sum_dns2;
    double sum_nodal[128];
};

// one workgroup; every output element is summed over (workgroup, row) in a fixed order, so the
// accumulators are bit-reproducible for a given launch geometry
__global__ void __launch_bounds__(256) relmc_finalize_kernel(const DevCase* __restrict__ C, const Partial* __restrict__ part,
                                                             int nblocks, DevAcc* __restrict__ out)
{
    const int t = threadIdx.x;
    const int nrows = nblocks * 4;
    for (int item = t; item < 8 + 256 + 128; item += blockDim.x) {
        if (item < 6) {
            long long s = 0;
            for (int r = 0; r < nrows; ++r) {
                const Partial& p = part[(size_t)r * 16];
                const uint32_t v = item == 0 ? p.n : item == 1 ? p.nfail : item == 2 ? p.nsing : item == 3 ? p.ninf : item == 4 ? p.nnc : p.iters;
                s += v;
            }
            (&out->n)[item] = s;
        } else if (item < 8) {
            double s = 0.0;
            for (int r = 0; r < nrows; ++r) { const Partial& p = part[(size_t)r * 16]; s += item == 6 ? p.dns : p.dns2; }
            if (item == 6) out->sum_dns = s; else out->sum_dns2 = s;
        } else if (item < 8 + 256) {
            const int k = item - 8;
            long long s = 0;
            if (k < C->ncomp) {
                const bool isgen = k < C->ng;
                const int idx = isgen ? k : k - C->ng;
                const int ln = idx & 15, sl = idx >> 4;
                for (int r = 0; r < nrows; ++r) {
                    const Partial& p = part[(size_t)r * 16 + ln];
                    s += isgen ? p.cf_inj[sl] : p.cf_line[sl];
                }
            }
            out->comp_fail[k] = s;
        } else {
            const int i = item - 8 - 256;
            double s = 0.0;
            // i = external bus number; internal tile position through b_int
            const int ii = i < C->nb ? C->b_int[i] : -1;
            if (ii >= 0 && C->b_vinj[ii] >= 0) {
                const int j = C->b_vinj[ii], ln = j & 15, sl = j >> 4;
                for (int r = 0; r < nrows; ++r) s += part[(size_t)r * 16 + ln].shed[sl];
            }
            out->sum_nodal[i] = s;
        }
    }
}

// mc_sampling.m:2 materialised: eqstatus[n x ncomp] uint8, one thread per (scenario, 4-component block)
__global__ void __launch_bounds__(256) relmc_sampling_kernel(const DevCase* __restrict__ C, uint64_t seed, uint64_t first_index,
                                                             int64_t n, uint8_t* __restrict__ eqstatus)
{
    const int ncomp = C->ncomp, nblk = (ncomp + 3) >> 2;
    const int64_t total = n * nblk;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = t / nblk;
        const int blk = (int)(t - i * nblk);
        const uint64_t gi = first_index + (uint64_t)i;
        uint32_t w[4];
        philox4x32_10((uint32_t)gi, (uint32_t)(gi >> 32), (uint32_t)blk, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), w);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = blk * 4 + e;
            if (k < ncomp) eqstatus[i * ncomp + k] = w[e] < C->thr[k] ? 1 : 0;
        }
    }
}

// probe used by the unit tests: out[lane] = value broadcast from lane L of the lane's own DPP row,
// and the row all-reduces (checks the row_newbcast / row_ror semantics the solver relies on)
__global__ void relmc_dpp_probe_kernel(const double* __restrict__ in, double* __restrict__ out)
{
    const int t = threadIdx.x;
    const double v = in[t];
    out[t] = BCAST(5, v);
    out[64 + t] = row_sum(v);
    out[128 + t] = row_max(v);
    out[192 + t] = row_min(v);
    out[256 + t] = (double)row_or(1u << (t & 15));
    out[320 + t] = frcp(v);
}

}  // namespace relmc
