// relmc_kernels.hip — gfx950 (MI355X / CDNA4) kernels of the HL2 non-sequential Monte Carlo
// state-evaluation path: mc_sampling + mc_simulation of the reference
// (Montecarlo_nsq_single/mc_sampling.m:24-45, mc_simulation.m:32-99, nsqMain.m:270,282-301),
// i.e. Bernoulli outage sampling, the MATPOWER DC-OPF load-curtailment LP solved by the MIPS
// primal-dual interior-point iteration, and the index accumulators.
//
// Mapping: ONE SCENARIO PER 16-LANE DPP ROW, four scenarios per wavefront.
//  * per-scenario vectors (lines, injections, buses) live in registers, one element per
//    (lane, slot); scalars of the iteration come from DPP row_ror all-reduces;
//  * the Newton step solves the reduced symmetric KKT system
//        [ Mth  B' ] [dth ]   [ -Nth            ]     Mth = Bf' diag(mu/z) Bf (weighted Laplacian)
//        [ B   -E  ] [dlam] = [ -g - C D^-1 Np  ]     E   = C D^-1 C'         (diagonal)
//    by a SPARSE 2x2-block LDL' (one block per bus pair (theta_i, lambda_i), pivot
//    det = -(m*e + b^2) free of cancellation) whose symbolic structure, fill and task schedule
//    are computed once per case on the host: the kernel only interprets a static list of
//    passes (16 independent block tasks per pass and scenario) on a ~4 KB LDS workspace.
#include "relmc_devfn.h"

namespace relmc {


// The first-dispatched wavefronts alternate between the lowest and the highest priority on this bit of the shader clock, sampled at the top of
// every interior-point iteration (~29 500 clocks on both test systems).  Round 3 sweep (profiles/r3_pf/c37_*.log, c38_*.log; RTS-24 / RTS-96 against
// bit 15, the round-1 choice): bits 9-11 +1..3 %, 12 -0.9 % / -1.0 %, 13, 14, 16 +1..2 % (a period close to the iteration's: the same wavefront wins
// iteration after iteration), 18-22 -0.7 % / -0.5 %, a coin per wavefront and iteration +0.4 % / +0.7 %, the iteration's parity +7 %.
// On top of bit 12: high level 2 instead of 3, or a second evaluation before the Newton step: -0.2 % / +0.2 .. +1 % (c39_*.log), not taken.
constexpr int kPrioBit = 12;
constexpr int kMinWaves = 2;       // waves per SIMD the register allocator must allow (<= 256 VGPRs)
#define DINF __builtin_inf()
// compiler-only fence: stops LICM from parking loop-invariant LDS table reads in VGPRs for the whole
// kernel (registers are the scarce resource here, LDS reads are cheap)
#define RELOAD_FENCE() __asm__ volatile("" ::: "memory")
// keeps the unrolled per-slot bodies from being interleaved (each body has ~20 live temporaries)
#define SLOT_FENCE() __builtin_amdgcn_sched_barrier(0)
// (round 3 re-measured every one of the six fence sites: removing any of them is neutral on the 16-lane tile and 0.3-2.7 % slower on the
// wide one, profiles/r3_pf/c11_*.log)
// LDS round trips of the vector phases taken off the wavefront's critical path (kPfMask, bit k = site k): table words are requested
// before the work that hides their latency instead of right where they are used.  Same arithmetic.
//   1 assembly: block offsets before the gathers                       2 gathers: incidence lists of both bus slots up front
//   3 step: the lines' and injections' solution entries as one batch   6 / 7 step / convergence test: solver options requested before the row reductions
// (sites 0, 4, 5 -- injection bounds / cost / lambda one slot ahead in the evaluation, the ratio tests, the update -- were measured neutral
//  to +1.8 % on both tiles and are gone from the source: profiles/r3_pf/c13_*.log)
constexpr int kPfMask = 0xc0;        // 16-lane tile: sites 6, 7 (-1 % kernel time; site 3 on top: -0.3 % for +44 B/lane of scratch = +40 % HBM traffic, not taken; single sites -0.3 .. +1.3 %, profiles/r3_pf/)
constexpr int kPfMaskWide = 0xc6;    // 64-lane tile: sites 1, 2, 6, 7 (-2.9 %)
#define PFSITE(k) (((PF_MASK >> (k)) & 1) != 0)
// optional per-phase cycle accounting (profiling builds only: -DRELMC_PHASE_TIMING)
#ifdef RELMC_PHASE_TIMING
#define PT_DECL unsigned long long pt_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long pt0_ = __builtin_readcyclecounter();
#define PT_MARK(k) { const unsigned long long n_ = __builtin_readcyclecounter(); pt_[k] += n_ - pt0_; pt0_ = n_; }
#define PT_FLUSH if (a.timing && lane == 0) { for (int k = 0; k < 8; ++k) a.timing[((size_t)blockIdx.x * WPB + (tid >> 6)) * 8 + k] = pt_[k]; }
#else
#define PT_DECL
#define PT_MARK(k)
#define PT_FLUSH
#endif
// -DRELMC_PHASE_TIMING -DRELMC_PT_INIT: the per-scenario-group setup split instead (0 window sampling, 1 state from the window / masks,
// 2 status -> model, 3 topology, 4 susceptance entries, 5 start point; 6 = the whole interior-point loop, 7 = output)
#if defined(RELMC_PHASE_TIMING) && defined(RELMC_PT_INIT)
#undef PT_MARK
#define PT_MARK(k) { const unsigned long long n_ = __builtin_readcyclecounter(); pt_[(k) == 7 ? 7 : 6] += n_ - pt0_; pt0_ = n_; }
#define PT_IMARK(k) { const unsigned long long n_ = __builtin_readcyclecounter(); pt_[k] += n_ - pt0_; pt0_ = n_; }
#else
#define PT_IMARK(k)
#endif

// MODE 0: fused non-sequential path (states from the counter-based sampler, accumulators only)
// MODE 1: explicit states (+ optional per-scenario load scale), per-scenario results written out
// MODE 2: sequential path: scenarios = compacted (year, hour) worklist, states from the chronology bit masks,
//         load scale from the hourly curve, curtailment written to curt[year][hour]
// MODE 3: distinct states of a sampled range with their multiplicities (the reference's unique-state database,
//         nsqMain.m:220-245, per launch): accumulators are weighted by the multiplicity
// MODE 5: MODE 0 instantiated a second time for relmc_case_load's order calibration, so that its one short launch does not enter
//         the statistics of the production kernel in a profile
// MODE 4: new rows of the persistent state database (nsqMain.m:257-278): scenario u = database row db_first + u, state from
//         the row's key words, results written into the row (dns, status/iterations, nodal shed); no accumulation
// MODE 6: MODE 4 with a dense, partially pivoted Newton solve in global scratch (a.dense): the last resort of the retry path
// MODE 7: the fused path behind the zero-curtailment pre-screen (relmc_screen.hip): scenario u = sample memo_perm[u] of the launch's range, the
//         ones the certificate did not cover, state = memo_keys[memo_perm[u]]; accumulators as MODE 0, per-sample dns at the sample's own index
template <int MODE_, class TL>
__global__ void __launch_bounds__(64 * TL::WPB, kMinWaves) relmc_eval_kernel(const DevCaseT<TL>* __restrict__ gcase, const EvalArgs a)
{
    constexpr int MODE = MODE_ == 5 ? 0 : (MODE_ == 6 ? 4 : MODE_);     // 5: the order-calibration probe = the fused path under a kernel name of its own (profiles stay clean)
    // MODE 6: MODE 4's rows with the Newton step solved DENSE with partial pivoting (the last resort for the units no static elimination
    // order converges on: MATLAB's `\` under mips, mc_simulation.m:41, pivots too).  The reduced system of order 2 nb lives in a global
    // scratch matrix per scenario row (RTS-96: 146 x 147 doubles do not fit LDS beside the tables); slow and rare by construction.
    constexpr bool DENSE = MODE_ == 6;
    constexpr int PAIR_MASK = TL::RW == 16 ? kRpairMask : kRpairMaskWide;      // which call sites share a reciprocal (frcp_pair)
#define PAIRSITE(k) (((PAIR_MASK >> (k)) & 1) != 0)
    constexpr int PF_MASK = TL::RW == 16 ? kPfMask : kPfMaskWide;                 // which LDS round trips are taken off the critical path (PFSITE)
    constexpr int RW = TL::RW, BS = TL::BS, LS = TL::LS, IS = TL::IS, NBT = TL::NBT, WPB = TL::WPB, SPW = TL::SPW, OW = TL::OW;
    using DevCase = DevCaseT<TL>;
    using Partial = PartialT<TL>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // solver options live in the first 128 bytes of LDS and are read where they are used: as kernel arguments they
    // would occupy ~20 SGPRs for the whole kernel and come back from spill lanes as 16-register tuples
    constexpr uint32_t OPT_BYTES = 128;
    double* const OPT = reinterpret_cast<double*>(smem);
#define A_(k, name) OPT[k]
    DevCase& C = *reinterpret_cast<DevCase*>(smem + OPT_BYTES);
    // Read-only tables that every iteration re-reads go through the vector-memory path (L1/L2 hits; its queue and `vmcnt` are otherwise
    // idle in this kernel) where that was measured to pay, instead of through the LDS pipe, which is the kernel's tightest resource:
    // the pass descriptors on both tiles (-2.0 % / -0.5 %), the injection table and the packed incidence lists on the wide tile (-1.9 %
    // each; on the narrow tile they cost +1.5 %: four scenarios' worth of dependent work waits on each load).
    const DevCase& TASKSRC = *gcase;
    const DevCase& TABL = C;
    const DevCase& TABI = (RW == 64) ? *gcase : C;
    const DevCase& TABG = (RW == 64) ? *gcase : C;
    const int tid = threadIdx.x, lane = tid & 63, rlane = lane & (RW - 1), row = tid / RW;
    const uint32_t case_bytes = a.case_bytes;     // = offsetof(DevCase, task): the pass schedule behind it stays in global memory (read through L1)
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(gcase);
        uint32_t* dst = reinterpret_cast<uint32_t*>(smem + OPT_BYTES);
        if (tid == 0) { OPT[0] = a.feastol; OPT[1] = a.gradtol; OPT[2] = a.comptol; OPT[3] = a.costtol; OPT[4] = a.xi; OPT[5] = a.sigma;
                        OPT[6] = a.z0; OPT[7] = a.alpha_min; OPT[8] = a.max_stepsize; OPT[9] = a.fail_threshold; }
        for (uint32_t i = tid; i < case_bytes / 4; i += 64 * WPB) dst[i] = src[i];
    }
    __syncthreads();
    double* const W = reinterpret_cast<double*>(smem + OPT_BYTES + ((case_bytes + 15u) & ~15u)) + (size_t)row * a.scen_doubles;
    // The evaluation arrays ALIAS the solver workspace: they are dead once the bus gathers have been
    // taken into registers, and only then are the KKT blocks written (see "assemble" below).
    // line record l = {g, lx, q, F} at LR + 4l; injection j: its p at IR[j], its {1/D, Np/D} in the stash pair j (the stash is indexed
    // by injection and survives the solve, so the pair is stored once for the gathers and for the step); record nl / entry ninj is an
    // all-zero dummy that unused gather slots point to.
    const int nlp = C.nl, nip = C.ninj;
    double* const LR = W;
    double* const IR = W + 4 * (nlp + 1);
    const int maxdeg0 = C.maxdeg_s[0], maxdeg1 = C.maxdeg_s[1], maxinj0 = C.maxinj_s[0], maxinj1 = C.maxinj_s[1];   // longest incidence lists per bus slot
    double* const Stash = W + a.stash_off + 2 * rlane;      // [IS][RW] pairs {1/D, Np/D} of this lane's injections (one b128 access each)
    const double* const StashJ = W + a.stash_off;          // pair j of the stash, read by the bus that gathers injection j (pair IS * RW: zeros)
    double* const Lam = W + a.stash_off + 2 * IS * RW + 2;   // [NBT]: bus multipliers lambda_i (kept across the solve)
    uint32_t* const OB = reinterpret_cast<uint32_t*>(Lam + NBT);   // [OW]: outage mask of the scenario

    const int ng = C.ng, ncomp = C.ncomp, nb = C.nb;
    const bool bwd_all_half = __builtin_amdgcn_readfirstlane((uint32_t)(C.bwd_half != 0 ? 1u : 0u)) != 0;      // all back-substitution passes in half form (relmc_dev.h)
    const int off_rhs = C.off_rhs, npu = C.npass_upd, npi = C.npass_inv, npass = C.npass, nzero = C.nzero, npuh = C.npass_upd - C.npass_updq, npuf = npuh - C.npass_updh;
    const double base = C.base_mva;
    const double eps = 2.220446049250313e-16;

    // ---- static per-lane tables -------------------------------------------------------
    uint32_t linfo[LS]; int lpart[LS];
#pragma unroll
    for (int s = 0; s < LS; ++s) { const int l = RW * s + rlane; linfo[s] = C.l_info[l]; lpart[s] = C.l_partner[l]; }
    // Loop-invariant per-lane table values kept in registers where that was measured to pay (the compiler spills colder values to scratch
    // instead, i.e. to the idle vector-memory path): susceptance and rating of the lane's lines on both tiles (-0.9 % / -1.9 %), the
    // injection bounds on the wide tile (-1.8 %; +0.6 % on the narrow one).  Cost and incidence lists in registers: neutral / +14 %.
    // ... per instantiation: the instantiations that write per-scenario results or walk the chronology (MODE 1, 2, 4) carry more live state
    // and lose 2-3 % with the line values in registers (measured on MODE 1 and 2), the fused ones gain 1-2 %
    // (A/Bs of round 3, DESIGN_HISTORY.md: every instantiation reading the line values from LDS halves the scratch traffic at the same time;
    //  the wide tile reading the injection bounds from the table does not buy a third wavefront per SIMD)
    constexpr bool LTAB_LDS = (MODE == 1 || MODE == 2 || MODE == 4);
    constexpr bool ITAB_REG = RW == 64;
    double lbv_[LS], lrv_[LS], ihi_[IS], ilo_[IS];
#pragma unroll
    for (int s = 0; s < LS; ++s) { lbv_[s] = LTAB_LDS ? 0.0 : TABL.l_b[RW * s + rlane]; lrv_[s] = LTAB_LDS ? 0.0 : TABL.l_rate[RW * s + rlane]; }
#pragma unroll
    for (int s = 0; s < IS; ++s) { ihi_[s] = ITAB_REG ? TABI.i_tab[RW * s + rlane][0] : 0.0; ilo_[s] = ITAB_REG ? TABI.i_tab[RW * s + rlane][1] : 0.0; }
#define lb(s) (LTAB_LDS ? TABL.l_b[RW * (s) + rlane] : lbv_[s])
#define lr(s) (LTAB_LDS ? TABL.l_rate[RW * (s) + rlane] : lrv_[s])
#define IHL(s, j) (ITAB_REG ? d2{ihi_[s], ilo_[s]} : ld2(TABI.i_tab[j]))
#define ICOST(s, j) TABI.i_tab[j][2]
#define PLIST(t, bi) TABG.b_line8[bi]
#define JLIST(t, bi) TABG.b_inj8[bi]
    uint32_t iinfo[IS];
#pragma unroll
    for (int s = 0; s < IS; ++s) iinfo[s] = C.i_info[RW * s + rlane];
    int vb[BS];                                   // the bus of this lane in bus slot t (>= nb: none)
#pragma unroll
    for (int t = 0; t < BS; ++t) vb[t] = C.b_lane[RW * t + rlane];

    // ---- accumulators (nsqMain.m:282-301 in per-sample form) ---------------------------
    // They live in this lane's Partial record in HBM (L2-resident, 104 B per lane) and are updated by a
    // read-modify-write once per scenario: ~25 VGPRs cheaper than carrying them through the solver.
    Partial& PA = reinterpret_cast<Partial*>(a.partial)[(size_t)blockIdx.x * (64 * WPB) + tid];
    {
        PA.dns = 0.0; PA.dns2 = 0.0;
#pragma unroll
        for (int s = 0; s < IS; ++s) { PA.shed[s] = 0.0; PA.cf_inj[s] = 0; }
#pragma unroll
        for (int s = 0; s < LS; ++s) PA.cf_line[s] = 0;
        PA.n = 0; PA.nfail = 0; PA.nsing = 0; PA.ninf = 0; PA.nnc = 0; PA.iters = 0; PA.pad = 0;
    }
    // the two counters every scenario touches stay in registers and reach the record once, at the end of the kernel: the
    // record is then written only by scenarios that shed load (8.5 % on RTS-24) instead of by all of them
    uint32_t acc_n = 0, acc_it = 0;

    PT_DECL
    const int64_t ngroups = (a.n + SPW - 1) / SPW;
    const int64_t gwave = (int64_t)blockIdx.x * WPB + (tid >> 6);
    const int64_t gstride = (int64_t)gridDim.x * WPB;
    // Issue-priority balancing.  A SIMD hosts two of these wavefronts and arbitrates oldest-first, so the wavefronts that were
    // dispatched first run ~20 % faster than the ones that joined them and the launch ends with half of the SIMD slots idle
    // (measured: per-wave times 0.905 / 1.093 of the mean).  The first-dispatched half (a.prio_mode 1: first half of the grid
    // when two workgroups share a CU; 2: first half of the waves of the workgroup) alternates between the lowest and the
    // highest user priority on a clock bit, the other half stays in between: each wavefront wins the arbitration half of
    // the TIME.  The work assignment stays static, so results are unchanged and reproducible.
    const bool prio_first = a.prio_mode == 1 ? blockIdx.x < (gridDim.x >> 1) : (a.prio_mode == 2 ? ((tid >> 8) & 1) == 0 : false);      // mode 2: the waves that came first on their SIMD (0-3; 8-11 behind 4-7)
    if (a.prio_mode != 0 && !prio_first) __builtin_amdgcn_s_setprio(1);
    // Fused non-sequential path on the 16-lane tile: every wavefront owns a contiguous range of scenario groups and walks
    // it in windows of 64 scenarios.  The window's states are sampled up front, one scenario per lane, and ordered so that
    // the states whose unit outages leave less capacity than the load (13-16 interior-point iterations instead of 12) share rows of the same groups: a
    // wavefront iterates until the slowest of its four rows has converged (mean of the maximum 12.9 vs mean 12.19).
    constexpr bool WINDOWED = (MODE == 0 && RW == 16);
    uint32_t* const WIN = reinterpret_cast<uint32_t*>(smem + OPT_BYTES + ((case_bytes + 15u) & ~15u)) +
                          (size_t)SPW * WPB * a.scen_doubles * 2 + (size_t)(tid >> 6) * 256;      // [64 slots][4 words] per wavefront
    // slot -> sampling lane of the window (the scenario's position in the sampled range), read only when per-scenario dns is asked for
    uint8_t* const WINL = reinterpret_cast<uint8_t*>(smem + OPT_BYTES + ((case_bytes + 15u) & ~15u)) +
                          ((size_t)SPW * WPB * a.scen_doubles * 2 + (size_t)WPB * 256) * 4 + (size_t)(tid >> 6) * 64;
    int64_t wb_begin = gwave, wb_end = ngroups, wb_step = gstride;
    if (WINDOWED) {
        wb_begin = ngroups * gwave / gstride; wb_end = ngroups * (gwave + 1) / gstride; wb_step = 16;
    }
    for (int64_t wb = wb_begin; wb < wb_end; wb += wb_step) {
    int win_groups = 1;
    uint32_t win_valid = 0;
    if (WINDOWED) {
        const int64_t si = wb * 4 + lane;                          // this lane's scenario of the window
        const bool valid = (wb + (lane >> 2)) < wb_end && si < a.n;
        uint32_t m0 = 0, m1 = 0, m2 = 0, m3 = 0;
        bool hard = false;
        if (valid) {
            const uint64_t gi = a.first_index + (uint64_t)si;
            const int nblk = (ncomp + 3) >> 2;
            for (int blk = 0; blk < nblk; ++blk) {
                uint32_t w[4], nib = 0;
                philox4x32_10((uint32_t)gi, (uint32_t)(gi >> 32), (uint32_t)blk, 0u, (uint32_t)a.seed, (uint32_t)(a.seed >> 32), w);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = blk * 4 + e;
                    if (k < ncomp && w[e] < C.thr[k]) nib |= 1u << e;   // strict '<', mc_sampling.m:35
                }
                const uint32_t sh = nib << ((blk & 7) * 4);
                const int wsel = blk >> 3;
                if (wsel == 0) m0 |= sh; else if (wsel == 1) m1 |= sh; else if (wsel == 2) m2 |= sh; else m3 |= sh;
            }
            // ordering key only (never a result): the unit outages leave less capacity than the load.  Measured on the
            // fixture: such states take 13-16 iterations, all others (line outages included) 12.
            double cap = 0.0;
            for (int k = 0; k < ng; ++k) {
                const uint32_t wsel = k < 64 ? (k < 32 ? m0 : m1) : (k < 96 ? m2 : m3);
                if (!((wsel >> (k & 31)) & 1u)) cap += C.i_tab[k][0];
            }
            hard = cap * base < C.total_load;
        }
        const uint64_t bv = __ballot(valid), bh = __ballot(valid && hard);
        const uint64_t below = (1ull << lane) - 1ull;
        const uint32_t n_valid = (uint32_t)__popcll(bv), n_easy = (uint32_t)__popcll(bv & ~bh);
        if (valid) {
            const uint32_t slot = hard ? n_easy + (uint32_t)__popcll(bh & below) : (uint32_t)__popcll(bv & ~bh & below);
            uint4 pk; pk.x = m0; pk.y = m1; pk.z = m2; pk.w = m3;
            *reinterpret_cast<uint4*>(WIN + 4 * slot) = pk;
            WINL[slot] = (uint8_t)lane;
        }
        RELOAD_FENCE();
        win_valid = n_valid;
        win_groups = (int)((n_valid + 3u) >> 2);
    }
    PT_IMARK(0)
    for (int wg = 0; wg < win_groups; ++wg) {
        const int64_t grp = WINDOWED ? wb + wg : wb;
        const int64_t sidx = grp * SPW + lane / RW;
        const bool live = WINDOWED ? (uint32_t)(wg * 4 + lane / RW) < win_valid : sidx < a.n;
        double lscale = 1.0;                 // load_scale_factor of seq_mcsimulation.m:38-42 (1 in the non-sequential path)
        uint32_t wgt = 1;                    // multiplicity of the state (MODE 3)
        int seq_year = 0, seq_hour = 0;
        RELOAD_FENCE();

        // per-scenario state ------------------------------------------------------------
        // per-lane status bits (one VGPR instead of lane masks in SGPRs):
        //   0-2 line slot in service, 3-5 line slot has a flow limit, 6-9 injection slot in service, 10-13 injection
        //   slot boxed (has inequality rows), 14-15 bus slot is an island's angle reference (identity theta row),
        //   16-17 bus slot's balance row dropped, 18-20 line slot touches a pinned bus, 21-23 / 24-26 the pair block's
        //   B entries K[th_hi][lam_lo] / K[lam_hi][th_lo] are removed (pinned column or dropped row)
        uint32_t sf = 0;
#define L_ON(s) (((sf >> (s)) & 1u) != 0)
#define L_ACT(s) (((sf >> (3 + (s))) & 1u) != 0)
#define I_ON(s) (((sf >> (6 + (s))) & 1u) != 0)
#define I_BOX(s) (((sf >> (10 + (s))) & 1u) != 0)
#define B_PIN(t) (((sf >> (14 + (t))) & 1u) != 0)
#define B_DROP(t) (((sf >> (16 + (t))) & 1u) != 0)
#define L_PIN(s) (((sf >> (18 + (s))) & 1u) != 0)
#define L_C1Z(s) (((sf >> (21 + (s))) & 1u) != 0)
#define L_C2Z(s) (((sf >> (24 + (s))) & 1u) != 0)
        double LFv[LS], LGv[LS], lzp[LS], lzm[LS], lmup[LS], lmum[LS], cBv[LS];
        double ip[IS], izp[IS], izm[IS], imup[IS], imum[IS];
        double bth[BS], bla[BS], cBd[BS];
        double gamma = 1.0, fval = 0.0, f0 = 0.0, alphap = 1.0, alphad = 1.0, zmu = 0.0;
        uint32_t niq = 0;
        int it = 0, status = 0;
        bool infeas = false, singular = false, iterating = false;
        uint32_t lozero = 0;                 // bit s: lower bound of injection slot s relaxed to 0 (island rules 3, 4)
#define ISC(s) ((RW * (s) + rlane >= ng) ? lscale : 1.0)      /* virtual generators (loads) scale with the hourly factor */
#define ILO(s) (((lozero >> (s)) & 1u) ? 0.0 : C.i_tab[RW * (s) + rlane][1] * ISC(s))
#define ILOV(s, lo_) (((lozero >> (s)) & 1u) ? 0.0 : (lo_) * ISC(s))      /* the same from a bound already loaded */
#pragma unroll
        for (int s = 0; s < LS; ++s) { LFv[s] = 0; LGv[s] = 0; lzp[s] = 1; lzm[s] = 1; lmup[s] = 1; lmum[s] = 1; cBv[s] = 0; }
#pragma unroll
        for (int s = 0; s < IS; ++s) { ip[s] = 0; izp[s] = 1; izm[s] = 1; imup[s] = 1; imum[s] = 1; }
#pragma unroll
        for (int t = 0; t < BS; ++t) { bth[t] = 0; bla[t] = 0; cBd[t] = 0; }

        if (live) {
            // ===== mc_sampling.m:24-41: Bernoulli outage state (1 = failed) -> OB (LDS words of this row) =====
            // LDS operations of one wavefront execute in order, so the row's lanes see each other's words after a
            // compiler-level fence; no barrier is needed (rows never share workspace).
            if (MODE == 2) {
                // (year, hour) of worklist entry sidx: binary search in the per-year offsets, seqMain.m:97-100
                int lo_ = 0, hi_ = a.seq_nyears;
                while (hi_ - lo_ > 1) { const int mid = (lo_ + hi_) >> 1; if ((int64_t)a.seq_offsets[mid] <= sidx) lo_ = mid; else hi_ = mid; }
                seq_year = lo_;
                seq_hour = a.seq_hours[(size_t)seq_year * a.seq_hpy + (size_t)(sidx - a.seq_offsets[seq_year])];
                const uint32_t* m = a.seq_masks + ((size_t)seq_year * a.seq_hpy + seq_hour) * OW;    // OW mask words per hour
                if (rlane < OW) OB[rlane] = m[rlane];
                lscale = a.load_factors[seq_hour];            // seqMain.m:114
            } else if (MODE == 3) {
                const uint32_t s0 = a.memo_start[sidx];
                wgt = a.memo_start[sidx + 1] - s0;
                if (rlane < OW) OB[rlane] = a.memo_keys[(size_t)a.memo_perm[s0] * OW + rlane];
            } else if (MODE == 4) {
                if (rlane < OW) OB[rlane] = a.memo_keys[(size_t)(a.db_first + (a.memo_perm ? (int64_t)a.memo_perm[sidx] : sidx)) * OW + rlane];
                if (a.load_scale) lscale = a.load_scale[sidx];          // retry rows of the scaled-load entry point
            } else if (MODE == 7) {
                if (rlane < OW) OB[rlane] = a.memo_keys[(size_t)a.memo_perm[sidx] * OW + rlane];
            } else {
                if (rlane < OW) OB[rlane] = 0u;
                RELOAD_FENCE();
                if (WINDOWED) {
                    if (rlane < OW) OB[rlane] = WIN[4 * (wg * 4 + lane / RW) + rlane];
                } else if (MODE == 0) {
                    const uint64_t gi = a.first_index + (uint64_t)sidx;
#pragma unroll
                    for (int h = 0; h < (TL::NCOMPMAX / 4 + RW - 1) / RW; ++h) {
                        const int blk = rlane + RW * h;
                        if (blk * 4 < ncomp) {
                            uint32_t w[4], nib = 0;
                            philox4x32_10((uint32_t)gi, (uint32_t)(gi >> 32), (uint32_t)blk, 0u, (uint32_t)a.seed, (uint32_t)(a.seed >> 32), w);
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const int k = blk * 4 + e;
                                if (k < ncomp && w[e] < C.thr[k]) nib |= 1u << e;   // strict '<', mc_sampling.m:35
                            }
                            if (nib) atomicOr(&OB[blk >> 3], nib << ((blk & 7) * 4));
                        }
                    }
                } else {
                    if (a.load_scale) lscale = a.load_scale[sidx];
                    const uint8_t* st = a.states + sidx * ncomp;
#pragma unroll
                    for (int q = 0; q < TL::NCOMPMAX / RW; ++q) {
                        const int k = rlane + RW * q;
                        if (k < ncomp && st[k]) atomicOr(&OB[k >> 5], 1u << (k & 31));
                    }
                }
            }
            RELOAD_FENCE();

            PT_IMARK(1)
            // ===== mc_simulation.m:32-37: component status -> model ==========================
#pragma unroll
            for (int s = 0; s < LS; ++s) {
                const int l = RW * s + rlane;
                const uint32_t fl = linfo[s] >> 24;
                const bool on = (fl & LF_EXISTS) && !outbit(OB, ng + l);
                if (on) sf |= 1u << s;
                if (on && (fl & LF_LIMITED)) sf |= 1u << (3 + s);
            }
#pragma unroll
            for (int s = 0; s < IS; ++s) {
                const int j = RW * s + rlane;
                const uint32_t kind = (iinfo[s] >> 8) & 0xff;
                if (kind == IK_VIRTUAL || (kind == IK_REAL && !outbit(OB, j))) sf |= 1u << (6 + s);
            }
            // lines out in this scenario?  If not (95 % of the RTS-24 scenarios) the network is one island and the
            // reachability sweeps are skipped; the island rules still run on it.
            bool lout = false;
#pragma unroll
            for (int s = 0; s < LS; ++s) lout = lout || (((linfo[s] >> 24) & LF_EXISTS) && !L_ON(s));
            const bool any_lout = C.base_connected == 0 || row_any<RW>(lout, lane);

            PT_IMARK(2)
            // ===== topology: isolated buses, islands, island rules (DESIGN.md "island policy") ==========
            if constexpr (RW == 16) {
                // 16-lane tile: bus sets are 32-bit masks in registers
                uint32_t pinned = 0, dropped = 0;
                uint32_t adjm[BS];
                bool iso = false;
#pragma unroll
                for (int t = 0; t < BS; ++t) {
                    const int i = vb[t];
                    uint32_t adj = 0;
                    if (i < nb && any_lout) {           // without a line outage no bus is isolated and the sweeps below are skipped
                        const int nlb = C.b_nline[i];
                        for (int e = 0; e < nlb; ++e) {
                            const uint32_t ent = C.b_line[i][e];
                            const int l = ent & 0x7f;
                            if (!outbit(OB, ng + l)) {
                                const uint32_t inf = C.l_info[l];
                                adj |= 1u << ((ent & 0x80) ? (inf & 0xff) : ((inf >> 8) & 0xff));
                            }
                        }
                        iso = iso || adj == 0;
                    }
                    adjm[t] = adj;
                }
                // a bus without any in-service branch makes MATPOWER's KKT matrix exactly singular; the
                // reference consumes the start point (mc_simulation.m:41,54)
                singular = (a.policy == 0) && row_any<RW>(iso, lane);
                if (!singular) {
                    uint32_t remaining = C.exist_mask;
                    for (int guard = 0; guard < NBT; ++guard) {
                        const bool more = remaining != 0;
                        if (!__any(more)) break;
                        if (more) {
                            uint32_t R = any_lout ? 1u << (__ffs((int)remaining) - 1) : remaining;
                            for (int sweep = 0; sweep < NBT; ++sweep) {
                                if (!__any(any_lout)) break;
                                uint32_t c = 0;
#pragma unroll
                                for (int t = 0; t < BS; ++t) if (vb[t] < nb && ((R >> vb[t]) & 1u)) c |= adjm[t];
                                const uint32_t Rn = R | row_or<RW>(c);
                                const bool ch = Rn != R;
                                R = Rn;
                                if (!__any(ch)) break;
                            }
                            // rule 1: the island's angle reference is its bus that is eliminated last =
                            // highest internal number (the host puts the reference bus at nb-1)
                            const int pin = 31 - __clz((int)R);
                            uint32_t cnt = 0; double losum = 0.0; bool inI[IS];
#pragma unroll
                            for (int s = 0; s < IS; ++s) {
                                const int j = RW * s + rlane;
                                const uint32_t kind = (iinfo[s] >> 8) & 0xff;
                                inI[s] = I_ON(s) && ((R >> (iinfo[s] & 0xff)) & 1u);
                                if (inI[s]) {
                                    cnt += 1u;
                                    if (kind == IK_VIRTUAL) cnt += 1u << 8; else if (C.i_tab[j][0] > 0.0) cnt += 1u << 16;
                                    losum += C.i_tab[j][3] * ISC(s);
                                }
                            }
                            cnt = row_add<RW>(cnt); losum = row_sum<RW>(losum);
                            const uint32_t n_inj = cnt & 0xff, n_load = (cnt >> 8) & 0xff, n_gen = cnt >> 16;
                            if (n_inj && !n_load) {                  // rule 2: no load -> decommit the island's units
#pragma unroll
                                for (int s = 0; s < IS; ++s) if (inI[s]) { sf &= ~(1u << (6 + s)); inI[s] = false; }
                                infeas = true;
                            } else if (n_load && !n_gen) {           // rule 3: no generation -> all load shed (p fixed 0)
#pragma unroll
                                for (int s = 0; s < IS; ++s) if (inI[s] && ((iinfo[s] >> 8) & 0xff) == IK_VIRTUAL) lozero |= 1u << s;
                            } else if (losum > 1e-9) {               // rule 4: over-generation -> relax Pmin
#pragma unroll
                                for (int s = 0; s < IS; ++s) if (inI[s] && ((iinfo[s] >> 8) & 0xff) == IK_REAL) lozero |= 1u << s;
                                infeas = true;
                            }
                            uint32_t nfree = 0;
#pragma unroll
                            for (int s = 0; s < IS; ++s) if (inI[s] && C.i_tab[RW * s + rlane][0] - ILO(s) > 0.0) nfree += 1u;
                            nfree = row_add<RW>(nfree);
                            if (!nfree) dropped |= 1u << pin;        // rule 5: dependent balance rows
                            pinned |= 1u << pin;
                            remaining &= ~R;
                        }
                    }
                }
#pragma unroll
                for (int t = 0; t < BS; ++t) {
                    const int i = vb[t];
                    if (i < nb && ((pinned >> i) & 1u)) sf |= 1u << (14 + t);
                    if (i < nb && ((dropped >> i) & 1u)) sf |= 1u << (16 + t);
                }
#pragma unroll
                for (int s = 0; s < LS; ++s) {
                    const int f = linfo[s] & 0xff, t = (linfo[s] >> 8) & 0xff;
                    const int lo_b = f < t ? f : t, hi_b = f < t ? t : f;    // block (hi, lo): rows of hi, columns of lo
                    if (((pinned >> f) | (pinned >> t)) & 1u) sf |= 1u << (18 + s);
                    if (((pinned >> hi_b) | (dropped >> lo_b)) & 1u) sf |= 1u << (21 + s);
                    if (((pinned >> lo_b) | (dropped >> hi_b)) & 1u) sf |= 1u << (24 + s);
                }
            } else {
                // one scenario per wavefront: bus sets live in LDS (the solver workspace is idle during this prologue):
                // LB[i] = label of bus i = highest internal bus number of its island, BF[i] = 1 pinned | 2 dropped
                static_assert(RW == 16 || RW == 64, "row width");
                int* const LB = reinterpret_cast<int*>(W);
                int* const BF = LB + NBT;
                bool iso = false;
#pragma unroll
                for (int t = 0; t < BS; ++t) {
                    const int i = vb[t];
                    if (i < nb) {
                        if (any_lout) {
                            bool anyon = false;
                            const int nlb = C.b_nline[i];
                            for (int e = 0; e < nlb; ++e) if (!outbit(OB, ng + (C.b_line[i][e] & 0x7f))) anyon = true;
                            iso = iso || !anyon;
                        }
                        LB[i] = any_lout ? i : nb - 1;
                        BF[i] = 0;
                    }
                }
                singular = (a.policy == 0) && row_any<RW>(iso, lane);
                RELOAD_FENCE();
                if (!singular) {
                    if (any_lout) {
                        for (int sweep = 0; sweep < NBT; ++sweep) {      // max-label propagation over the in-service lines
                            bool ch = false;
#pragma unroll
                            for (int s = 0; s < LS; ++s) {
                                if (L_ON(s)) {
                                    const int f = linfo[s] & 0xff, t = (linfo[s] >> 8) & 0xff;
                                    const int lf = LB[f], lt = LB[t];
                                    if (lf != lt) { ch = true; if (lf < lt) atomicMax(&LB[f], lt); else atomicMax(&LB[t], lf); }
                                }
                            }
                            RELOAD_FENCE();
                            if (!row_any<RW>(ch, lane)) break;
                        }
                    }
                    int ilab[IS];
#pragma unroll
                    for (int s = 0; s < IS; ++s) ilab[s] = LB[iinfo[s] & 0xff];
#pragma unroll
                    for (int t = 0; t < BS; ++t) {
                        const int i = vb[t];
                        uint64_t roots = __ballot(i < nb && LB[i] == i);
                        while (roots) {
                            const int pin = C.b_lane[RW * t + (int)__builtin_ctzll(roots)];      // rule 1: the island's highest bus (the bus that lane holds)
                            roots &= roots - 1;
                            uint32_t cnt = 0; double losum = 0.0; bool inI[IS];
#pragma unroll
                            for (int s = 0; s < IS; ++s) {
                                const int j = RW * s + rlane;
                                const uint32_t kind = (iinfo[s] >> 8) & 0xff;
                                inI[s] = I_ON(s) && ilab[s] == pin;
                                if (inI[s]) {
                                    cnt += 1u;
                                    if (kind == IK_VIRTUAL) cnt += 1u << 8; else if (C.i_tab[j][0] > 0.0) cnt += 1u << 16;
                                    losum += C.i_tab[j][3] * ISC(s);
                                }
                            }
                            cnt = row_add<RW>(cnt); losum = row_sum<RW>(losum);
                            const uint32_t n_inj = cnt & 0xff, n_load = (cnt >> 8) & 0xff, n_gen = cnt >> 16;
                            if (n_inj && !n_load) {                  // rule 2
#pragma unroll
                                for (int s = 0; s < IS; ++s) if (inI[s]) { sf &= ~(1u << (6 + s)); inI[s] = false; }
                                infeas = true;
                            } else if (n_load && !n_gen) {           // rule 3
#pragma unroll
                                for (int s = 0; s < IS; ++s) if (inI[s] && ((iinfo[s] >> 8) & 0xff) == IK_VIRTUAL) lozero |= 1u << s;
                            } else if (losum > 1e-9) {               // rule 4
#pragma unroll
                                for (int s = 0; s < IS; ++s) if (inI[s] && ((iinfo[s] >> 8) & 0xff) == IK_REAL) lozero |= 1u << s;
                                infeas = true;
                            }
                            uint32_t nfree = 0;
#pragma unroll
                            for (int s = 0; s < IS; ++s) if (inI[s] && C.i_tab[RW * s + rlane][0] - ILO(s) > 0.0) nfree += 1u;
                            nfree = row_add<RW>(nfree);
#pragma unroll
                            for (int u = 0; u < BS; ++u)
                                if (vb[u] == pin) { sf |= 1u << (14 + u); if (!nfree) sf |= 1u << (16 + u); }   // rules 1, 5
                        }
                    }
#pragma unroll
                    for (int t = 0; t < BS; ++t) if (vb[t] < nb) BF[vb[t]] = (B_PIN(t) ? 1 : 0) | (B_DROP(t) ? 2 : 0);
                    RELOAD_FENCE();
#pragma unroll
                    for (int s = 0; s < LS; ++s) {
                        if ((linfo[s] >> 24) & LF_EXISTS) {
                            const int f = linfo[s] & 0xff, t = (linfo[s] >> 8) & 0xff;
                            const int lo_b = f < t ? f : t, hi_b = f < t ? t : f;
                            const int flo = BF[lo_b], fhi = BF[hi_b];
                            if ((flo | fhi) & 1) sf |= 1u << (18 + s);
                            if ((fhi & 1) | (flo & 2)) sf |= 1u << (21 + s);
                            if ((flo & 1) | (fhi & 2)) sf |= 1u << (24 + s);
                        }
                    }
                    RELOAD_FENCE();
                }
            }

            PT_IMARK(3)
            // ===== constant (per scenario) susceptance entries of the KKT blocks, pins/drops masked ==
#pragma unroll
            for (int s = 0; s < LS; ++s) {
                const uint32_t inf = linfo[s];
                if ((inf >> 24) & LF_OWNER) {
                    double v = L_ON(s) ? lb(s) : 0.0;
                    const int pr = lpart[s];
                    if (pr >= 0 && !outbit(OB, ng + pr)) v += C.l_b[pr];
                    cBv[s] = -v;                 // -(b_l + b_partner) of the in-service lines of this bus pair
                }
            }
#pragma unroll
            for (int t = 0; t < BS; ++t) {
                const int i = vb[t];
                if (i < nb) {
                    double d = C.b_bsum[i];             // every line in service: the host's sum in the same order
                    if (any_lout) {
                        d = 0.0;
                        const int nlb = C.b_nline[i];
                        for (int e = 0; e < nlb; ++e) {
                            const int l = C.b_line[i][e] & 0x7f;
                            if (!outbit(OB, ng + l)) d += C.l_b[l];
                        }
                    }
                    cBd[t] = (B_PIN(t) || B_DROP(t)) ? 0.0 : d;
                }
            }

            PT_IMARK(4)
            // ===== dcopf_solver start point + mips initialisation (SURVEY.md Appendix B 2,4) =====
            double fl = 0.0;
            uint32_t nq = 0;
#pragma unroll
            for (int s = 0; s < LS; ++s) {
                if (L_ACT(s)) {
                    const double h = -lr(s);                       // x0: all angles 0 -> flow 0
                    double z = A_(6, z0); if (h < -A_(6, z0)) z = -h;
                    double mu = A_(6, z0); if (1.0 / z > A_(6, z0)) mu = 1.0 / z;
                    lzp[s] = z; lzm[s] = z; lmup[s] = mu; lmum[s] = mu;
                    nq += 2;
                }
            }
#pragma unroll
            for (int s = 0; s < IS; ++s) {
                const int j = RW * s + rlane;
                if (I_ON(s)) {
                    const double hi = C.i_tab[j][0], lo = ILO(s);
                    const bool box = hi - lo > eps;
                    if (box) sf |= 1u << (10 + s);
                    ip[s] = box ? 0.5 * (lo + hi) : hi;
                    if (box) {
                        const double h = -0.5 * (hi - lo);
                        double z = A_(6, z0); if (h < -A_(6, z0)) z = -h;
                        double mu = A_(6, z0); if (1.0 / z > A_(6, z0)) mu = 1.0 / z;
                        izp[s] = z; izm[s] = z; imup[s] = mu; imum[s] = mu;
                        nq += 2;
                    }
                    fl += C.i_tab[j][2] * ip[s];
                }
            }
#pragma unroll
            for (int t = 0; t < BS; ++t) if (vb[t] < nb) Lam[vb[t]] = 0.0;
            if (rlane == 0) st2(W + a.stash_off + 2 * IS * RW, 0.0, 0.0);     // the stash pair behind the last lane's: the dummy entry when every lane holds an injection
            niq = row_add<RW>(nq);
            fval = row_sum<RW>(fl);
            f0 = fval;
            iterating = !singular;
            status = singular ? 3 : 0;
        }

        PT_IMARK(5)
        PT_MARK(0)
        // ===== mips main loop =======================================
        while (__any(iterating)) {
            if (prio_first) {
                if ((__builtin_readcyclecounter() >> kPrioBit) & 1ull) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0);
            }
            RELOAD_FENCE();
            // The per-slot conditions (slot in service / boxed / owner / pinned ...) are loop invariant, so the compiler hoists them out of the
            // interior-point loop as 64-bit lane masks: ~45 masks, 163 SGPRs spilled to VGPR lanes, and the VGPRs those push out go to scratch.
            // Hiding the words they derive from once per iteration makes it re-derive them where they are used (a v_and + v_cmp instead of two
            // v_readlane): scratch 284 -> 164 B/lane and HBM traffic 530 -> 219 B per scenario at the same kernel time on the 16-lane tile
            // (round 3, profiles/r3_final/scratch_ab.log; results bit-identical).  The 64-lane tile loses 1.5 % with it and keeps the masks.
            if constexpr (RW == 16) {
                __asm__ volatile("" : "+v"(sf));
#pragma unroll
                for (int s = 0; s < LS; ++s) __asm__ volatile("" : "+v"(linfo[s]));
#pragma unroll
                for (int s = 0; s < IS; ++s) __asm__ volatile("" : "+v"(iinfo[s]));
#pragma unroll
                for (int t = 0; t < BS; ++t) __asm__ volatile("" : "+v"(vb[t]));
            }
            if (iterating) {
                // ---- evaluate h, Lx, barrier terms; scatter to LDS; convergence norms -------------
                double mx_gh = -DINF, mx_x = 0.0, mx_z = 0.0, mx_lx = 0.0, mx_lammu = 0.0;
                bool nanx = false;
                double gown[LS];
#pragma unroll
                for (int s = 0; s < LS; ++s) {
                    const int l = RW * s + rlane;
                    double g = 0.0, q = 0.0;
                    double lx = LGv[s];              // G_l = b_l (lambda_f - lambda_t), carried incrementally; stays 0 on a line out of service
                    if (L_ACT(s)) {
                        const double b = lb(s), rr = lr(s);
                        const double hp = LFv[s] - rr, hm = -LFv[s] - rr;
                        double rzp, rzm; frcp_pair<PAIRSITE(0)>(lzp[s], lzm[s], rzp, rzm);
                        g = b * b * (lmup[s] * rzp + lmum[s] * rzm);
                        lx = __builtin_fma(b, lmup[s] - lmum[s], lx);
                        q = b * ((lmup[s] * hp + gamma) * rzp - (lmum[s] * hm + gamma) * rzm);
                        mx_gh = vmax(mx_gh, vmax(hp, hm));
                        mx_z = vmax(mx_z, vmax(lzp[s], lzm[s]));
                        mx_lammu = vmax(mx_lammu, vmax(lmup[s], lmum[s]));
                    }
                    gown[s] = g;
                    if (l < nlp) { st2(LR + 4 * l, g, lx); st2(LR + 4 * l + 2, lx + q, LFv[s]); }   // nl / ninj records (they alias W)
                    SLOT_FENCE();
                }
#pragma unroll
                for (int s = 0; s < IS; ++s) {
                    const int j = RW * s + rlane;
                    double invD = 0.0, npd = 0.0;
                    const double pv = ip[s];         // stays 0 on an injection out of service
                    mx_x = vmax(mx_x, __builtin_fabs(pv));
                    nanx = nanx || pv != pv;
                    if (I_BOX(s)) {
                        const d2 hl = IHL(s, j);               // {upper, lower} bound
                        const double hp = pv - hl.x, hm = ILOV(s, hl.y) - pv;
                        const double lxp = ICOST(s, j) - Lam[iinfo[s] & 0xff] + (imup[s] - imum[s]);
                        if constexpr ((RW == 16 && kInjNform) || (RW == 64 && kInjNformWide)) {
                            // D = N / (z+ z-) with N = mu+ z- + mu- z+: 1/D and Np/D from ONE reciprocal (of N) instead of three (round 3: -1.9 %
                            // kernel time; no fixture state changes its iteration count, 6 of 1e6 sampled scenarios do by one).  The 64-lane tile keeps
                            // its round-2 arithmetic (see kRpairMaskWide).
                            const double N = __builtin_fma(imup[s], izm[s], imum[s] * izp[s]);
                            const double rN = frcp(N), zz = izp[s] * izm[s];
                            invD = zz * rN;
                            npd = __builtin_fma(lxp, zz, (imup[s] * hp + gamma) * izm[s] - (imum[s] * hm + gamma) * izp[s]) * rN;
                        } else {
                        double rzp, rzm; frcp_pair<PAIRSITE(1)>(izp[s], izm[s], rzp, rzm);
                        const double D = imup[s] * rzp + imum[s] * rzm;
                        const double np = lxp + (imup[s] * hp + gamma) * rzp - (imum[s] * hm + gamma) * rzm;
                        invD = frcp(D); npd = np * invD;
                        }
                        mx_lx = vmax(mx_lx, __builtin_fabs(lxp));
                        mx_gh = vmax(mx_gh, vmax(hp, hm));
                        mx_z = vmax(mx_z, vmax(izp[s], izm[s]));
                        mx_lammu = vmax(mx_lammu, vmax(imup[s], imum[s]));
                    }
                    if (j < nip) IR[j] = pv;
                    st2(Stash + 2 * RW * s, invD, npd);
                    SLOT_FENCE();
                }
                PT_MARK(1)
                if (rlane == 0) { st2(LR + 4 * nlp, 0.0, 0.0); st2(LR + 4 * nlp + 2, 0.0, 0.0); IR[nip] = 0.0; }
                // ---- assemble: gather everything the KKT blocks need into registers ... -------------
                uint32_t lblk_pf[LS]; uint32_t zo_pf = 0;
                if constexpr (PFSITE(1)) {
#pragma unroll
                    for (int s = 0; s < LS; ++s) lblk_pf[s] = C.l_blk[RW * s + rlane];
                    zo_pf = C.zero_off[rlane];
                }
                unsigned long long pl_pf[BS], pj_pf[BS];
                if constexpr (PFSITE(2)) {
#pragma unroll
                    for (int t = 0; t < BS; ++t) { const int bq = vb[t] < nb ? vb[t] : 0; pl_pf[t] = PLIST(t, bq); pj_pf[t] = JLIST(t, bq); }
                }
                double vown[LS];
                {   // the partner's g of every slot in one batch of loads (the all-zero record when there is none) instead of one LDS round trip per
                    // owner slot, each in a branch of its own (round 3: -1.7 % / -2.3 % kernel time on RTS-24 / RTS-96, bit-identical)
                    double gp[LS];
#pragma unroll
                    for (int s = 0; s < LS; ++s) gp[s] = LR[4 * (lpart[s] >= 0 ? lpart[s] : nlp)];
#pragma unroll
                    for (int s = 0; s < LS; ++s) vown[s] = (((linfo[s] >> 24) & LF_OWNER) && !L_PIN(s)) ? -(gown[s] + gp[s]) : 0.0;
                }
                double d00[BS], d11[BS], r0[BS], r1[BS];
#pragma unroll
                for (int t = 0; t < BS; ++t) {
                    const int bi = vb[t];
                    d00[t] = 0; d11[t] = 0; r0[t] = 0; r1[t] = 0;
                    if (bi < nb) {
                        double md = 0.0, lx = 0.0, nq_ = 0.0, bal = 0.0, E = 0.0, ssum = 0.0;
                        // incidence lists come packed (one 8-byte LDS read each); unused slots point at the zero
                        // records, so all record loads of a bus are independent and issue back to back
                        const unsigned long long pl = PFSITE(2) ? pl_pf[t] : PLIST(t, bi), pj = PFSITE(2) ? pj_pf[t] : JLIST(t, bi);
                        // two list entries per step: their four (three) record loads issue together and one wait covers both;
                        // an odd list's last step reads the all-zero record once more (adds exact zeros)
#pragma unroll
                        for (int e = 0; e < DEGMAX; e += 2) {
                            if (e >= (t == 0 ? maxdeg0 : maxdeg1)) break;
                            const uint32_t ent0 = (uint32_t)(pl >> (8 * e)) & 0xffu, ent1 = (uint32_t)(pl >> (8 * e + 8)) & 0xffu;
                            const int l0 = (int)(ent0 & 0x7f), l1 = (int)(ent1 & 0x7f);             // unused entries name the all-zero record nl
                            const double sg0 = (ent0 & 0x80) ? -1.0 : 1.0, sg1 = (ent1 & 0x80) ? -1.0 : 1.0;
                            const d2 ra0 = ld2(LR + 4 * l0), rb0 = ld2(LR + 4 * l0 + 2), ra1 = ld2(LR + 4 * l1), rb1 = ld2(LR + 4 * l1 + 2);
                            md += ra0.x;
                            lx = __builtin_fma(sg0, ra0.y, lx);
                            nq_ = __builtin_fma(sg0, rb0.x, nq_);
                            bal = __builtin_fma(sg0, rb0.y, bal);
                            md += ra1.x;
                            lx = __builtin_fma(sg1, ra1.y, lx);
                            nq_ = __builtin_fma(sg1, rb1.x, nq_);
                            bal = __builtin_fma(sg1, rb1.y, bal);
                        }
#pragma unroll
                        for (int e = 0; e < BINJMAX; e += 2) {
                            if (e >= (t == 0 ? maxinj0 : maxinj1)) break;
                            const int j0 = (int)((uint32_t)(pj >> (8 * e)) & 0xffu), j1 = (int)((uint32_t)(pj >> (8 * e + 8)) & 0xffu);   // unused entries name the all-zero record ninj
                            const d2 sh0 = ld2(StashJ + 2 * j0), sh1 = ld2(StashJ + 2 * j1);
                            const double p0 = IR[j0], p1 = IR[j1];
                            bal -= p0; E += sh0.x; ssum += sh0.y;
                            bal -= p1; E += sh1.x; ssum += sh1.y;
                        }
                        if (B_PIN(t)) {                   // fixed angle: identity row; its multiplier is -lx
                            d00[t] = 1.0; r0[t] = 0.0;
                            mx_lammu = vmax(mx_lammu, __builtin_fabs(lx));
                        } else {
                            d00[t] = md; r0[t] = -nq_;
                            mx_lx = vmax(mx_lx, __builtin_fabs(lx));
                        }
                        if (B_DROP(t)) {                  // dependent balance row
                            d11[t] = -1.0; r1[t] = 0.0;
                        } else {
                            d11[t] = -E; r1[t] = -bal - ssum;
                            mx_gh = vmax(mx_gh, __builtin_fabs(bal));
                            mx_lammu = vmax(mx_lammu, __builtin_fabs(bla[t]));
                        }
                        mx_x = vmax(mx_x, __builtin_fabs(bth[t]));
                        nanx = nanx || bth[t] != bth[t];
                    }
                }
                // ---- ... then overwrite the workspace (same LDS words as the evaluation arrays) ------
                RELOAD_FENCE();
                if constexpr (DENSE) {
                    // the same blocks into the dense matrix [n x (n + 1)], n = 2 nb, unknowns (theta_0, lambda_0, theta_1, ...), last column = rhs
                    const int n = 2 * nb, n1 = n + 1;
                    volatile double* A = a.dense + ((size_t)blockIdx.x * (WPB * SPW) + (size_t)(tid / RW)) * (size_t)a.dense_stride;
                    for (int e = rlane; e < n * n1; e += RW) A[e] = 0.0;
                    __threadfence_block();
#pragma unroll
                    for (int s = 0; s < LS; ++s) {
                        if ((linfo[s] >> 24) & LF_OWNER) {
                            const int f = linfo[s] & 0xff, t = (linfo[s] >> 8) & 0xff;
                            const int hi = f < t ? t : f, lo = f < t ? f : t;
                            const double c1 = L_C1Z(s) ? 0.0 : cBv[s], c2 = L_C2Z(s) ? 0.0 : cBv[s];
                            A[(2 * hi) * n1 + 2 * lo] = vown[s]; A[(2 * lo) * n1 + 2 * hi] = vown[s];
                            A[(2 * hi) * n1 + 2 * lo + 1] = c1; A[(2 * lo + 1) * n1 + 2 * hi] = c1;
                            A[(2 * hi + 1) * n1 + 2 * lo] = c2; A[(2 * lo) * n1 + 2 * hi + 1] = c2;
                        }
                    }
#pragma unroll
                    for (int t = 0; t < BS; ++t) {
                        const int bi = vb[t];
                        if (bi < nb) {
                            A[(2 * bi) * n1 + 2 * bi] = d00[t]; A[(2 * bi) * n1 + 2 * bi + 1] = cBd[t];
                            A[(2 * bi + 1) * n1 + 2 * bi] = cBd[t]; A[(2 * bi + 1) * n1 + 2 * bi + 1] = d11[t];
                            A[(2 * bi) * n1 + n] = r0[t]; A[(2 * bi + 1) * n1 + n] = r1[t];
                        }
                    }
                    __threadfence_block();
                } else {
#pragma unroll
                for (int s = 0; s < LS; ++s) {
                    if ((linfo[s] >> 24) & LF_OWNER) {
                        // block (hi, lo): rows of the later-eliminated bus, columns of the earlier one
                        const double c1 = L_C1Z(s) ? 0.0 : cBv[s];   // K[th_hi][lam_lo] = B(row lam_lo, col th_hi)
                        const double c2 = L_C2Z(s) ? 0.0 : cBv[s];   // K[lam_hi][th_lo] = B(row lam_hi, col th_lo)
                        double* blk = W + (PFSITE(1) ? lblk_pf[s] : (uint32_t)C.l_blk[RW * s + rlane]);
                        st2(blk, vown[s], c1); st2(blk + 2, c2, 0.0);
                    }
                }
                if constexpr (PFSITE(1)) {
                    for (int z = rlane; z < nzero; z += RW) {
                        const uint32_t zn = C.zero_off[z + RW < TL::MAXOFF ? z + RW : z];      // the next offset is on its way while this block is cleared
                        double* blk = W + zo_pf; st2(blk, 0.0, 0.0); st2(blk + 2, 0.0, 0.0);
                        zo_pf = zn;
                    }
                } else
                for (int z = rlane; z < nzero; z += RW) { double* blk = W + C.zero_off[z]; st2(blk, 0.0, 0.0); st2(blk + 2, 0.0, 0.0); }
#pragma unroll
                for (int t = 0; t < BS; ++t) {
                    const int bi = vb[t];
                    if (bi < nb) {
                        st2(W + 4 * bi, d00[t], cBd[t]); st2(W + 4 * bi + 2, cBd[t], d11[t]);
                        st2(W + off_rhs + 2 * bi, r0[t], r1[t]);
                    }
                }
                }

                PT_MARK(2)
                // ---- convergence test (mips.m feascond/gradcond/compcond/costcond) --------------
                // The four conditions must hold together, so the two that need only |x| (complementarity and
                // cost change) are tested first; the other three row reductions run only when some scenario
                // of the wavefront passes them (never before the last 2-3 iterations).
                double o_ct = 0.0, o_cc = 0.0, o_am = 0.0;
                if constexpr (PFSITE(7)) { o_ct = A_(2, comptol); o_cc = A_(3, costtol); o_am = A_(7, alpha_min); }
                mx_x = row_max<RW>(mx_x);
                const bool xnan = row_any<RW>(nanx, lane);
                if constexpr (!PFSITE(7)) { o_ct = A_(2, comptol); o_cc = A_(3, costtol); }
                bool conv = it > 0 && zmu < o_ct * (1.0 + mx_x) && __builtin_fabs(fval - f0) < o_cc * (1.0 + __builtin_fabs(f0));
#ifdef RELMC_TRACE
                {   // debug builds: per-iteration termination quantities of scenario 0 (first launch row) -> a.timing as doubles
                    const double t_gh = row_max<RW>(mx_gh), t_z = row_max<RW>(mx_z), t_lx = row_max<RW>(mx_lx), t_lm = row_max<RW>(mx_lammu);
                    if (a.timing && blockIdx.x == 0 && tid == 0 && it < 60) {
                        double* o = reinterpret_cast<double*>(a.timing) + 8 * it;
                        o[0] = t_gh / (1.0 + vmax(mx_x, t_z)); o[1] = t_lx / (1.0 + t_lm); o[2] = zmu / (1.0 + mx_x);
                        o[3] = __builtin_fabs(fval - f0) / (1.0 + __builtin_fabs(f0)); o[4] = alphap; o[5] = alphad; o[6] = gamma; o[7] = fval;
                    }
                }
#endif
                if (__any(conv)) {
                    mx_gh = row_max<RW>(mx_gh); mx_z = row_max<RW>(mx_z); mx_lx = row_max<RW>(mx_lx); mx_lammu = row_max<RW>(mx_lammu);
                    // feascond < feastol, gradcond < gradtol with the (positive) denominators multiplied out
                    conv = conv && mx_gh < A_(0, feastol) * (1.0 + vmax(mx_x, mx_z)) && mx_lx < A_(1, gradtol) * (1.0 + mx_lammu);
                }
                if (conv) { status = 0; iterating = false; }
                else if (it > 0 && (xnan || alphap < (PFSITE(7) ? o_am : A_(7, alpha_min)) || alphad < (PFSITE(7) ? o_am : A_(7, alpha_min)) || gamma < eps || gamma > 1.0 / eps)) { status = 2; iterating = false; }
                else if (it >= a.max_it) { status = 1; iterating = false; }
            }
            PT_MARK(3)
            RELOAD_FENCE();
            if (iterating) {
                f0 = fval;
                it += 1;
                if constexpr (DENSE) {
                    // ---- Newton step, last resort: Gaussian elimination with partial pivoting on the dense reduced system --------
                    // lane r of the row owns columns r, r + RW, ... (the rhs is column n); the multipliers of a step are computed by
                    // all lanes at once and parked in LDS (W is free: the factor is not built), so the row loop reads them there
                    const int n = 2 * nb, n1 = n + 1;
                    volatile double* A = a.dense + ((size_t)blockIdx.x * (WPB * SPW) + (size_t)(tid / RW)) * (size_t)a.dense_stride;
                    double* const Fm = W;                            // [n] multipliers of the current step, then the solution
                    bool singular_k = false;
                    for (int k = 0; k < n; ++k) {
                        double best = -1.0; int bi = n;
                        for (int i = k + rlane; i < n; i += RW) { const double v = __builtin_fabs(A[(size_t)i * n1 + k]); if (v > best) { best = v; bi = i; } }
                        const double mx = row_max<RW>(best);
                        const uint32_t cand = (best == mx) ? (uint32_t)bi : (uint32_t)n;
                        const int p = (int)row_min_u32<RW>(cand);                                     // lowest row index among the largest entries
                        if (!(mx > 0.0)) { singular_k = true; break; }                                // exactly singular (or NaN): numerically failed
                        if (p != k) for (int j = k + rlane; j <= n; j += RW) { const double u = A[(size_t)k * n1 + j], v = A[(size_t)p * n1 + j]; A[(size_t)k * n1 + j] = v; A[(size_t)p * n1 + j] = u; }
                        __threadfence_block();
                        const double rp = 1.0 / A[(size_t)k * n1 + k];
                        for (int i = k + 1 + rlane; i < n; i += RW) Fm[i] = A[(size_t)i * n1 + k] * rp;
                        RELOAD_FENCE();
                        for (int i = k + 1; i < n; ++i) {
                            const double f = Fm[i];
                            if (f != 0.0) for (int j = k + 1 + rlane; j <= n; j += RW) A[(size_t)i * n1 + j] = A[(size_t)i * n1 + j] - f * A[(size_t)k * n1 + j];
                        }
                        __threadfence_block();
                    }
                    // back substitution; x in LDS (Fm), then into the solution slots the step phase reads (X[2i] = dtheta_i, X[2i+1] = dlambda_i)
                    for (int k = n - 1; k >= 0 && !singular_k; --k) {
                        double sdot = 0.0;
                        for (int j = k + 1 + rlane; j < n; j += RW) sdot = __builtin_fma(A[(size_t)k * n1 + j], Fm[j], sdot);
                        sdot = row_sum<RW>(sdot);
                        if (rlane == 0) Fm[k] = (A[(size_t)k * n1 + n] - sdot) / A[(size_t)k * n1 + k];
                        RELOAD_FENCE();
                    }
                    double xs[(2 * NBT + RW - 1) / RW];
#pragma unroll
                    for (int q = 0; q < (2 * NBT + RW - 1) / RW; ++q) { const int e = rlane + RW * q; xs[q] = e < n ? (singular_k ? __builtin_nan("") : Fm[e]) : 0.0; }
                    RELOAD_FENCE();
#pragma unroll
                    for (int q = 0; q < (2 * NBT + RW - 1) / RW; ++q) { const int e = rlane + RW * q; if (e < n) W[off_rhs + e] = xs[q]; }
                    RELOAD_FENCE();
                } else {
                // ---- Newton step: sparse 2x2-block LDL' on the LDS workspace, static schedule --------
                // descriptors are prefetched one pass ahead (they do not depend on data); 0xffff = no task for this lane
                // descriptor fields are byte offsets into the scenario's workspace (bit 15 of the first = rhs task in the full-form passes)
                unsigned char* const W8 = reinterpret_cast<unsigned char*>(W);
#define WP(off) reinterpret_cast<double*>(W8 + (off))
                uint2 dsc = *reinterpret_cast<const uint2*>(&TASKSRC.task[0][rlane][0]);
                for (int p = 0; p < npuf; ++p) {             // T -= Wa * inv(D) * Wb'   (T: 2x2 block, or 1x2 rhs row)
                    const uint2 nxt = *reinterpret_cast<const uint2*>(&TASKSRC.task[p + 1][rlane][0]);
                    if ((dsc.x & 0xffffu) != 0xffffu) {
                        const bool vec = (dsc.x & 0x8000u) != 0;            // rhs pseudo-bus: Wa = [y_i'; 0], T = y_a'
                        double* T = WP(dsc.x & 0x7fffu);
                        const double* Wa = WP(dsc.x >> 16);
                        const double* Wb = WP(dsc.y & 0xffffu);
                        const double* D = WP(dsc.y >> 16);
                        const d2 dA = ld2(D); const double dBy = D[3];      // D = [[m, b], [b, -e]]: the second half is needed for -e only (b64 read)
                        const d2 a0 = ld2(Wa), b0 = ld2(Wb), b1 = ld2(Wb + 2);
                        d2 t0 = ld2(T);
                        // (all eight operand loads in one batch -- one LDS round trip less per pass -- is 0.7 % / 1.5 % SLOWER, profiles/r3_pf/c17_*.log:
                        //  these passes are bound by the LDS pipe, not by its latency, and the longer batch holds the pipe against the CU's other wavefronts)
                        const double pm = dA.x, pb = dA.y, pe = -dBy;
                        const double q = frcp(__builtin_fma(pm, pe, pb * pb));
                        const double P00 = pe * q, P01 = pb * q, P11 = -pm * q;
                        const double g00 = __builtin_fma(a0.x, P00, a0.y * P01), g01 = __builtin_fma(a0.x, P01, a0.y * P11);
                        t0.x -= __builtin_fma(g00, b0.x, g01 * b0.y); t0.y -= __builtin_fma(g00, b1.x, g01 * b1.y);
                        st2(T, t0.x, t0.y);
                        if (!vec) {
                            const d2 a1 = ld2(Wa + 2);
                            d2 t1 = ld2(T + 2);
                            const double g10 = __builtin_fma(a1.x, P00, a1.y * P01), g11 = __builtin_fma(a1.x, P01, a1.y * P11);
                            t1.x -= __builtin_fma(g10, b0.x, g11 * b0.y); t1.y -= __builtin_fma(g10, b1.x, g11 * b1.y);
                            st2(T + 2, t1.x, t1.y);
                        }
                    }
                    dsc = nxt;
                }
                for (int p = npuf; p < npuh; ++p) {          // the same update, one row of T per lane (passes filled to at most a half, relmc_dev.h)
                    const uint2 nxt = *reinterpret_cast<const uint2*>(&TASKSRC.task[p + 1][rlane][0]);
                    if ((dsc.x & 0xffffu) != 0xffffu) {
                        double* T = WP(dsc.x & 0xffffu);
                        const double* Wa = WP(dsc.x >> 16);
                        const double* Wb = WP(dsc.y & 0xffffu);
                        const double* D = WP(dsc.y >> 16);
                        const d2 dA = ld2(D); const double dBy = D[3];
                        const d2 a0 = ld2(Wa), b0 = ld2(Wb), b1 = ld2(Wb + 2);
                        d2 t0 = ld2(T);
                        const double pm = dA.x, pb = dA.y, pe = -dBy;
                        const double q = frcp(__builtin_fma(pm, pe, pb * pb));
                        const double P00 = pe * q, P01 = pb * q, P11 = -pm * q;
                        const double g00 = __builtin_fma(a0.x, P00, a0.y * P01), g01 = __builtin_fma(a0.x, P01, a0.y * P11);
                        t0.x -= __builtin_fma(g00, b0.x, g01 * b0.y); t0.y -= __builtin_fma(g00, b1.x, g01 * b1.y);
                        st2(T, t0.x, t0.y);
                    }
                    dsc = nxt;
                }
                for (int p = npuh; p < npu; ++p) {           // the same update, one element of T per lane (sparsely filled passes, relmc_dev.h)
                    const uint2 nxt = *reinterpret_cast<const uint2*>(&TASKSRC.task[p + 1][rlane][0]);
                    if ((dsc.x & 0xffffu) != 0xffffu) {
                        double* T = WP(dsc.x & 0xffffu);
                        const double* Wa = WP(dsc.x >> 16);
                        const double* Wb = WP(dsc.y & 0xffffu);
                        const double* D = WP(dsc.y >> 16);
                        const d2 dA = ld2(D); const double dBy = D[3];
                        const d2 a0 = ld2(Wa), b0 = ld2(Wb);
                        double t = *T;
                        const double pm = dA.x, pb = dA.y, pe = -dBy;
                        const double q = frcp(__builtin_fma(pm, pe, pb * pb));
                        const double P00 = pe * q, P01 = pb * q, P11 = -pm * q;
                        const double g0 = __builtin_fma(a0.x, P00, a0.y * P01), g1 = __builtin_fma(a0.x, P01, a0.y * P11);
                        t -= __builtin_fma(g0, b0.x, g1 * b0.y);
                        *T = t;
                    }
                    dsc = nxt;
                }
                PT_MARK(4)
                for (int p = npu; p < npu + npi; ++p) {      // D <- P = inv(D) in place; y <- P*y
                    const uint2 nxt = *reinterpret_cast<const uint2*>(&TASKSRC.task[p + 1][rlane][0]);
                    if ((dsc.x & 0xffffu) != 0xffffu) {
                        double* D = WP(dsc.x & 0xffffu);
                        double* Y = WP(dsc.x >> 16);
                        const d2 dA = ld2(D), y = ld2(Y); const double dBy = D[3];
                        const double pm = dA.x, pb = dA.y, pe = -dBy;
                        const double q = frcp(__builtin_fma(pm, pe, pb * pb));
                        const double P00 = pe * q, P01 = pb * q, P11 = -pm * q;
                        st2(D, P00, P01); st2(D + 2, P01, P11);
                        st2(Y, __builtin_fma(P00, y.x, P01 * y.y), __builtin_fma(P01, y.x, P11 * y.y));
                    }
                    dsc = nxt;
                }
                PT_MARK(5)
                if (RW == 64 && bwd_all_half) {               // every back-substitution pass is filled to at most a half: one component of y_i per lane (wide tile only)
                    for (int p = npu + npi; p < npass; ++p) {
                        const uint2 nxt = *reinterpret_cast<const uint2*>(&TASKSRC.task[p + 1][rlane][0]);
                        if ((dsc.x & 0xffffu) != 0xffffu) {
                            double* Yi = WP(dsc.x & 0xffffu);
                            const double* Wk = WP(dsc.x >> 16);
                            const double* Pr = WP(dsc.y & 0xffffu);
                            const double* Ya = WP(dsc.y >> 16);
                            const d2 w0 = ld2(Wk), w1 = ld2(Wk + 2), pr = ld2(Pr), x = ld2(Ya);
                            double y = *Yi;
                            const double u0 = __builtin_fma(w0.x, x.x, w1.x * x.y), u1 = __builtin_fma(w0.y, x.x, w1.y * x.y);
                            y -= __builtin_fma(pr.x, u0, pr.y * u1);
                            *Yi = y;
                        }
                        dsc = nxt;
                    }
                } else
                for (int p = npu + npi; p < npass; ++p) {    // y_i -= P_i * W' * x_a
                    const uint2 nxt = *reinterpret_cast<const uint2*>(&TASKSRC.task[p + 1][rlane][0]);
                    if ((dsc.x & 0xffffu) != 0xffffu) {
                        double* Yi = WP(dsc.x & 0xffffu);
                        const double* Wk = WP(dsc.x >> 16);
                        const double* P = WP(dsc.y & 0xffffu);
                        const double* Ya = WP(dsc.y >> 16);
                        const d2 w0 = ld2(Wk), w1 = ld2(Wk + 2), p0 = ld2(P), p1 = ld2(P + 2), x = ld2(Ya);
                        d2 y = ld2(Yi);
                        const double u0 = __builtin_fma(w0.x, x.x, w1.x * x.y), u1 = __builtin_fma(w0.y, x.x, w1.y * x.y);
                        y.x -= __builtin_fma(p0.x, u0, p0.y * u1); y.y -= __builtin_fma(p1.x, u0, p1.y * u1);
                        st2(Yi, y.x, y.y);
                    }
                    dsc = nxt;
                }
                PT_MARK(5)
                }
                RELOAD_FENCE();
                // ---- step lengths ---------------------------------------------------------------------
                const double* X = W + off_rhs;               // solution: X[2i] = dtheta_i, X[2i+1] = dlambda_i
                double step2 = 0.0;
                double dth[BS], dla[BS];
#pragma unroll
                for (int t = 0; t < BS; ++t) {
                    const int bi = vb[t];
                    dth[t] = 0; dla[t] = 0;
                    if (bi < nb) { const d2 x = ld2(X + 2 * bi); dth[t] = x.x; dla[t] = x.y; step2 = __builtin_fma(x.x, x.x, __builtin_fma(x.y, x.y, step2)); }
                }
                // Two passes over the slack/multiplier steps: pass 1 only finds the step lengths (ratio
                // tests), pass 2 recomputes dz, dmu and applies them.  Recomputing ~150 VALU instructions
                // keeps 72 VGPRs free, which is what lets two wavefronts share a SIMD.  Measured alternatives (round 3, profiles/r3_rcp/):
                // 1/z of the evaluation kept for both passes (14 doubles per lane that the allocator parks in scratch across the solver
                // passes): +11.6 % / +11.2 % kernel time on RTS-24 / RTS-96 -- the reloads sit on the wavefront's critical path; dmu of
                // pass 1 kept for pass 2: neutral (the 14 doubles are spilled again); the full-step-slack form dz = (r - (F + dF)) - z
                // (three adds less per pair): WRONG -- F + dF absorbs a step of 1e-12 into a flow of O(1), the slack of an active limit
                // (1e-10) then carries a relative error of 1e-6 and 1 % of the scenarios never converge; -h - z - dh keeps small with small.
                double tp = 0.0, td = 0.0;        // max over inequality rows of -dz/z and -dmu/mu
                double dF[LS], dG[LS];
                double dpv[IS];
                double o_ms = 0.0;
                if constexpr (PFSITE(6)) o_ms = A_(8, max_stepsize);
                if constexpr (PFSITE(3)) {
                    d2 xf[LS], xt[LS], sh[IS]; double dlb[IS];
#pragma unroll
                    for (int s = 0; s < LS; ++s) { xf[s] = ld2(X + 2 * (linfo[s] & 0xff)); xt[s] = ld2(X + 2 * ((linfo[s] >> 8) & 0xff)); }
#pragma unroll
                    for (int s = 0; s < IS; ++s) { dlb[s] = X[2 * (iinfo[s] & 0xff) + 1]; sh[s] = ld2(Stash + 2 * RW * s); }
#pragma unroll
                    for (int s = 0; s < LS; ++s) { dF[s] = L_ON(s) ? lb(s) * (xf[s].x - xt[s].x) : 0.0; dG[s] = L_ON(s) ? lb(s) * (xf[s].y - xt[s].y) : 0.0; }
#pragma unroll
                    for (int s = 0; s < IS; ++s) dpv[s] = I_BOX(s) ? __builtin_fma(dlb[s], sh[s].x, -sh[s].y) : 0.0;
                }
#pragma unroll
                for (int s = 0; s < LS; ++s) {
                    if constexpr (!PFSITE(3)) { dF[s] = 0; dG[s] = 0; }
                    if (L_ON(s)) {
                        if constexpr (!PFSITE(3)) {
                        const int f = linfo[s] & 0xff, t = (linfo[s] >> 8) & 0xff;
                        const d2 xf = ld2(X + 2 * f), xt = ld2(X + 2 * t);
                        dF[s] = lb(s) * (xf.x - xt.x);
                        dG[s] = lb(s) * (xf.y - xt.y);
                        }
                        if (L_ACT(s)) {
                            const double hp = LFv[s] - lr(s), hm = -LFv[s] - lr(s);
                            const double dzp = -hp - lzp[s] - dF[s], dzm = -hm - lzm[s] + dF[s];
                            double rzp, rzm; frcp_pair<PAIRSITE(2)>(lzp[s], lzm[s], rzp, rzm);
                            const double dmup = -lmup[s] + (gamma - lmup[s] * dzp) * rzp;
                            const double dmum = -lmum[s] + (gamma - lmum[s] * dzm) * rzm;
                            // ratio tests without divisions by the steps: min_k z_k/(-dz_k) = 1 / max_k(-dz_k/z_k)
                            tp = vmax(tp, vmax(-dzp * rzp, -dzm * rzm));
                            double rmp, rmm; frcp1_pair<PAIRSITE(3)>(lmup[s], lmum[s], rmp, rmm);
                            td = vmax(td, vmax(-dmup * rmp, -dmum * rmm));
                        }
                    }
                    SLOT_FENCE();
                }
#pragma unroll
                for (int s = 0; s < IS; ++s) {
                    const int j = RW * s + rlane;
                    if constexpr (!PFSITE(3)) dpv[s] = 0;
                    {
                        if (I_BOX(s)) {
                            if constexpr (!PFSITE(3)) {
                            const double dlb = X[2 * (iinfo[s] & 0xff) + 1];
                            const d2 sh = ld2(Stash + 2 * RW * s);
                            dpv[s] = __builtin_fma(dlb, sh.x, -sh.y);   // dp = (-Np + dlam)/D
                            }
                            const d2 hl = IHL(s, j);
                            const double hp = ip[s] - hl.x, hm = ILOV(s, hl.y) - ip[s];
                            const double dzp = -hp - izp[s] - dpv[s], dzm = -hm - izm[s] + dpv[s];
                            double rzp, rzm; frcp_pair<PAIRSITE(4)>(izp[s], izm[s], rzp, rzm);
                            const double dmup = -imup[s] + (gamma - imup[s] * dzp) * rzp;
                            const double dmum = -imum[s] + (gamma - imum[s] * dzm) * rzm;
                            tp = vmax(tp, vmax(-dzp * rzp, -dzm * rzm));
                            double rmp, rmm; frcp1_pair<PAIRSITE(5)>(imup[s], imum[s], rmp, rmm);
                            td = vmax(td, vmax(-dmup * rmp, -dmum * rmm));
                            step2 = __builtin_fma(dpv[s], dpv[s], step2);
                        }
                    }
                    SLOT_FENCE();
                }
                step2 = row_sum<RW>(step2);
#ifdef RELMC_TRACE
                if (a.timing && blockIdx.x == 0 && tid < RW && it < 40) {     // debug builds: the Newton step of scenario 0, per iteration
                    double* o2 = reinterpret_cast<double*>(a.timing) + 512 + 512 * it;
                    for (int t = 0; t < BS; ++t) if (vb[t] < nb) { o2[vb[t]] = dth[t]; o2[128 + vb[t]] = dla[t]; }
                    for (int s = 0; s < IS; ++s) o2[256 + RW * s + rlane] = dpv[s];
                    if (it == 1) for (int t = 0; t < BS; ++t) if (vb[t] < nb) (reinterpret_cast<double*>(a.timing) + 512 + 512 * 40)[vb[t]] = (double)C.b_ext[vb[t]];
                }
#endif
                if (!(PFSITE(6) ? step2 <= o_ms * o_ms : step2 <= A_(8, max_stepsize) * A_(8, max_stepsize))) {
                    // NaN or |dxdlam| > max_stepsize: "numerically failed", x is NOT updated
                    status = 2; iterating = false;
                } else {
                    double o_xi = 0.0, o_sg = 0.0;
                    if constexpr (PFSITE(6)) { o_xi = A_(4, xi); o_sg = A_(5, sigma); }
                    tp = row_max<RW>(tp); td = row_max<RW>(td);
                    if constexpr (!PFSITE(6)) o_xi = A_(4, xi);
                    alphap = tp > 0.0 ? vmin(o_xi * frcp(tp), 1.0) : 1.0;   // min(xi * min(z./-dz), 1)
                    alphad = td > 0.0 ? vmin(o_xi * frcp(td), 1.0) : 1.0;
                    double zl = 0.0, fl = 0.0;
#pragma unroll
                    for (int s = 0; s < LS; ++s) {
                        if (L_ACT(s)) {
                            const double hp = LFv[s] - lr(s), hm = -LFv[s] - lr(s);
                            const double dzp = -hp - lzp[s] - dF[s], dzm = -hm - lzm[s] + dF[s];
                            double rzp, rzm; frcp_pair<PAIRSITE(6)>(lzp[s], lzm[s], rzp, rzm);
                            const double dmup = -lmup[s] + (gamma - lmup[s] * dzp) * rzp;
                            const double dmum = -lmum[s] + (gamma - lmum[s] * dzm) * rzm;
                            lzp[s] = __builtin_fma(alphap, dzp, lzp[s]); lzm[s] = __builtin_fma(alphap, dzm, lzm[s]);
                            lmup[s] = __builtin_fma(alphad, dmup, lmup[s]); lmum[s] = __builtin_fma(alphad, dmum, lmum[s]);
                            zl = __builtin_fma(lzp[s], lmup[s], zl); zl = __builtin_fma(lzm[s], lmum[s], zl);
                        }
                        LFv[s] = __builtin_fma(alphap, dF[s], LFv[s]);      // dF = dG = 0 on a line out of service
                        LGv[s] = __builtin_fma(alphad, dG[s], LGv[s]);
                        SLOT_FENCE();
                    }
#pragma unroll
                    for (int s = 0; s < IS; ++s) {
                        if (I_BOX(s)) {
                            const int j = RW * s + rlane;
                            const d2 hl = IHL(s, j);
                            const double hp = ip[s] - hl.x, hm = ILOV(s, hl.y) - ip[s];
                            const double dzp = -hp - izp[s] - dpv[s], dzm = -hm - izm[s] + dpv[s];
                            double rzp, rzm; frcp_pair<PAIRSITE(7)>(izp[s], izm[s], rzp, rzm);
                            const double dmup = -imup[s] + (gamma - imup[s] * dzp) * rzp;
                            const double dmum = -imum[s] + (gamma - imum[s] * dzm) * rzm;
                            ip[s] = __builtin_fma(alphap, dpv[s], ip[s]);
                            izp[s] = __builtin_fma(alphap, dzp, izp[s]); izm[s] = __builtin_fma(alphap, dzm, izm[s]);
                            imup[s] = __builtin_fma(alphad, dmup, imup[s]); imum[s] = __builtin_fma(alphad, dmum, imum[s]);
                            zl = __builtin_fma(izp[s], imup[s], zl); zl = __builtin_fma(izm[s], imum[s], zl);
                        }
                        fl = __builtin_fma(ICOST(s, RW * s + rlane), ip[s], fl);      // p = 0 on an injection out of service
                        SLOT_FENCE();
                    }
#pragma unroll
                    for (int t = 0; t < BS; ++t) {
                        bth[t] = __builtin_fma(alphap, dth[t], bth[t]); bla[t] = __builtin_fma(alphad, dla[t], bla[t]);
                        if (vb[t] < nb) Lam[vb[t]] = bla[t];
                    }
                    zmu = row_sum<RW>(zl);
                    fval = row_sum<RW>(fl);
                    if (niq > 0) gamma = (PFSITE(6) ? o_sg : A_(5, sigma)) * zmu / (double)niq;
                }
                PT_MARK(6)
            }
        }

        PT_MARK(6)
        // ===== mc_simulation.m:54-99 (dns, noise filters, nodal shed) + nsqMain.m:270 =============
        if (live) {
            double dns = fval + C.total_load * lscale;            // mc_simulation.m:54 / seq_mcsimulation.m:67
            if (dns < 0.1) dns = 0.0;
            const bool fail = dns > a.fail_threshold;             // nsqMain.m:270 (1e-4) / seqMain.m:41,140 (0.01)
            // a unit the static pivot order did not converge on (6.7e-7 of the RTS-96 scenarios) goes to the retry list;
            // it is accumulated here only if the list is off or full
            bool defer = false;
            if (a.fail_list && (status == 1 || status == 2)) {
                uint32_t pos = 0;
                if (rlane == 0) pos = atomicAdd(a.fail_count, 1u);
                pos = __shfl(pos, lane & ~(RW - 1));
                if (pos < a.fail_cap) {
                    defer = true;
                    FailRec* fr = a.fail_list + pos;
                    if (rlane < OW) fr->mask[rlane] = OB[rlane];
                    if (rlane == 0) {
                        const int64_t u = (MODE == 0 && WINDOWED) ? wb * 4 + WINL[wg * 4 + lane / RW]
                                          : (MODE == 2 ? (int64_t)seq_year * a.seq_hpy + seq_hour                // MODE 2: the hour of the chronology
                                          : (MODE == 7 ? (int64_t)a.memo_perm[sidx]                              // MODE 7: the sample's offset in the range
                                          : (MODE == 4 && a.memo_perm ? (int64_t)a.memo_perm[sidx] : sidx)));      // MODE 4 behind the pre-screen: the row's offset
                        fr->unit = (unsigned long long)(u + a.unit_base); fr->weight = wgt; fr->pad = 0;
                    }
                }
            }
            double shed[IS];
#pragma unroll
            for (int s = 0; s < IS; ++s) {
                const int j = RW * s + rlane;
                shed[s] = 0.0;
                if (((iinfo[s] >> 8) & 0xff) == IK_VIRTUAL && dns > 0.0) {
                    const double v = ip[s] * base - C.i_tab[j][3] * lscale;     // Pg - Pmin, mc_simulation.m:86
                    if (v > 1e-3) shed[s] = v;                           // mc_simulation.m:90
                }
                if (MODE != 4 && !defer && shed[s] != 0.0) PA.shed[s] += MODE == 3 ? shed[s] * (double)wgt : shed[s];
                if (MODE != 4 && !defer && fail && ((iinfo[s] >> 8) & 0xff) == IK_REAL && outbit(OB, j)) PA.cf_inj[s] += wgt;
            }
#pragma unroll
            for (int s = 0; s < LS; ++s)
                if (MODE != 4 && !defer && fail && ((linfo[s] >> 24) & LF_EXISTS) && outbit(OB, ng + RW * s + rlane)) PA.cf_line[s] += wgt;
            if (MODE != 4 && !defer && rlane == 0) {          // row-uniform quantities: one lane per scenario row
                acc_n += wgt;
                if (dns != 0.0) {
                    if (MODE == 3) { PA.dns = __builtin_fma((double)wgt, dns, PA.dns); PA.dns2 = __builtin_fma((double)wgt * dns, dns, PA.dns2); }
                    else { PA.dns += dns; PA.dns2 = __builtin_fma(dns, dns, PA.dns2); }
                }
                if (fail) PA.nfail += wgt;
                if (status == 3) PA.nsing += wgt;
                if (status == 1 || status == 2) PA.nnc += wgt;
                if (infeas) PA.ninf += wgt;
                acc_it += (uint32_t)it * wgt;
            }
            if (MODE == 2 && rlane == 0) a.curt[(size_t)seq_year * a.seq_hpy + seq_hour] = dns;
            if (MODE == 0 && a.dns && rlane == 0)      // optional: dns of every sample in sampling order (checkpoint histories of small batches)
                a.dns[WINDOWED ? wb * 4 + WINL[wg * 4 + lane / RW] : sidx] = dns;
            if (MODE == 7 && a.dns && rlane == 0) a.dns[a.memo_perm[sidx]] = dns;
            if (MODE == 1 || MODE == 4) {
                const int64_t oidx = MODE == 4 ? a.db_first + (a.memo_perm ? (int64_t)a.memo_perm[sidx] : sidx) : sidx;       // MODE 4: the database row
#pragma unroll
                for (int s = 0; s < IS; ++s) if (RW * s + rlane < nip) IR[4 * (RW * s + rlane)] = shed[s];
                if (rlane == 0) {
                    a.dns[oidx] = dns;
                    // database rows pack what the estimators need beside dns: status | relaxed << 2 | iterations << 8
                    if (MODE == 4) a.status[oidx] = status | (infeas ? 4 : 0) | (it << 8);
                    else {
                        if (a.status) a.status[oidx] = status;
                        if (a.iters) a.iters[oidx] = it;
                    }
                }
                if (a.nodal) {
#pragma unroll
                    for (int t = 0; t < BS; ++t) {
                        const int i = vb[t];
                        if (i < nb) {
                            const int vj = C.b_vinj[i];
                            a.nodal[oidx * nb + C.b_ext[i]] = vj >= 0 ? IR[4 * vj] : 0.0;
                        }
                    }
                }
            }
        }
        PT_MARK(7)
    }
    }
    if (rlane == 0) { PA.n = acc_n; PA.iters = acc_it; }
    PT_FLUSH
}

// One workgroup of four wavefronts per output element.  Thread t sums the scenario rows t, t + 256, ... of its element's field (eight
// independent running sums, so eight loads are in flight per thread: the kernel is a chain of dependent L2 round trips, not bandwidth -- one
// wavefront per element took 61 us for 8 192 rows, round 5), the eight sums are combined in a fixed order, the wavefront by a fixed butterfly,
// the four wavefronts in index order: the accumulators are bit-reproducible for a launch geometry.
constexpr int FIN_THREADS = 256, FIN_UNROLL = 8;
template <class TL>
__global__ void __launch_bounds__(FIN_THREADS) relmc_finalize_kernel(const DevCaseT<TL>* __restrict__ C, const PartialT<TL>* __restrict__ part,
                                                                     int nrows, DevAcc* __restrict__ out)
{
    constexpr int RW = TL::RW;
    using Partial = PartialT<TL>;
    const int item = blockIdx.x, tid = threadIdx.x;
    int ln = 0; size_t foff = 0;              // lane of the scenario row that holds the element, byte offset of its field in the lane's record
    bool is_int = true, active = true;
    if (item < 6) foff = offsetof(Partial, n) + sizeof(uint32_t) * (size_t)item;             // n, nfail, nsing, ninf, nnc, iters
    else if (item < 8) { foff = item == 6 ? offsetof(Partial, dns) : offsetof(Partial, dns2); is_int = false; }
    else if (item < 8 + 256) {
        const int k = item - 8;
        if (k < C->ncomp) {
            const bool isgen = k < C->ng; const int idx = isgen ? k : k - C->ng; ln = idx % RW;
            foff = (isgen ? offsetof(Partial, cf_inj) : offsetof(Partial, cf_line)) + sizeof(uint32_t) * (size_t)(idx / RW);
        } else active = false;
    } else {
        const int i = item - 8 - 256;                  // external bus number
        const int ii = i < C->nb ? C->b_int[i] : -1;
        is_int = false;
        if (ii >= 0 && C->b_vinj[ii] >= 0) { const int j = C->b_vinj[ii]; ln = j % RW; foff = offsetof(Partial, shed) + sizeof(double) * (size_t)(j / RW); }
        else active = false;
    }
    static_assert(offsetof(Partial, nfail) == offsetof(Partial, n) + 4 && offsetof(Partial, iters) == offsetof(Partial, n) + 20, "the six counters are consecutive u32");
    long long si = 0; double sd = 0.0;
    if (active) {
        const unsigned char* base = reinterpret_cast<const unsigned char*>(part) + sizeof(Partial) * (size_t)ln + foff;
        constexpr size_t kRow = sizeof(Partial) * (size_t)RW, kStep = kRow * FIN_THREADS;
        const int full = nrows / (FIN_UNROLL * FIN_THREADS) * (FIN_UNROLL * FIN_THREADS);
        const unsigned char* q = base + kRow * (size_t)tid;
        if (is_int) {
            long long s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0, s6 = 0, s7 = 0;
            for (int r = 0; r < full; r += FIN_UNROLL * FIN_THREADS, q += FIN_UNROLL * kStep) {
                s0 += *reinterpret_cast<const uint32_t*>(q); s1 += *reinterpret_cast<const uint32_t*>(q + kStep);
                s2 += *reinterpret_cast<const uint32_t*>(q + 2 * kStep); s3 += *reinterpret_cast<const uint32_t*>(q + 3 * kStep);
                s4 += *reinterpret_cast<const uint32_t*>(q + 4 * kStep); s5 += *reinterpret_cast<const uint32_t*>(q + 5 * kStep);
                s6 += *reinterpret_cast<const uint32_t*>(q + 6 * kStep); s7 += *reinterpret_cast<const uint32_t*>(q + 7 * kStep);
            }
            for (int r = full + tid; r < nrows; r += FIN_THREADS, q += kStep) s0 += *reinterpret_cast<const uint32_t*>(q);
            si = ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7));
        } else {
            double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0, s4 = 0.0, s5 = 0.0, s6 = 0.0, s7 = 0.0;
            for (int r = 0; r < full; r += FIN_UNROLL * FIN_THREADS, q += FIN_UNROLL * kStep) {
                s0 += *reinterpret_cast<const double*>(q); s1 += *reinterpret_cast<const double*>(q + kStep);
                s2 += *reinterpret_cast<const double*>(q + 2 * kStep); s3 += *reinterpret_cast<const double*>(q + 3 * kStep);
                s4 += *reinterpret_cast<const double*>(q + 4 * kStep); s5 += *reinterpret_cast<const double*>(q + 5 * kStep);
                s6 += *reinterpret_cast<const double*>(q + 6 * kStep); s7 += *reinterpret_cast<const double*>(q + 7 * kStep);
            }
            for (int r = full + tid; r < nrows; r += FIN_THREADS, q += kStep) s0 += *reinterpret_cast<const double*>(q);
            sd = ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7));
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { si += __shfl_xor(si, off); sd += __shfl_xor(sd, off); }
    __shared__ long long wi[FIN_THREADS / 64];
    __shared__ double wd[FIN_THREADS / 64];
    if ((tid & 63) == 0) { wi[tid >> 6] = si; wd[tid >> 6] = sd; }
    __syncthreads();
    if (tid == 0) {
        si = (wi[0] + wi[1]) + (wi[2] + wi[3]); sd = (wd[0] + wd[1]) + (wd[2] + wd[3]);
        if (item < 6) { (&out->n)[item] = si; if (item == 0) out->n_screened = 0; }     // the pre-screen's count is the host's to add (the image may hold an all-reduce's or the database's)
        else if (item == 6) out->sum_dns = sd;
        else if (item == 7) out->sum_dns2 = sd;
        else if (item < 8 + 256) out->comp_fail[item - 8] = si;
        else out->sum_nodal[item - 8 - 256] = sd;
    }
}

// mc_sampling.m:2 materialised: eqstatus[n x ncomp] uint8, one thread per (scenario, 4-component block)
template <class TL>
__global__ void __launch_bounds__(256) relmc_sampling_kernel(const DevCaseT<TL>* __restrict__ C, uint64_t seed, uint64_t first_index,
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

// probe used by the unit tests: row broadcast / row all-reduce semantics the solver relies on
__global__ void relmc_dpp_probe_kernel(const double* __restrict__ in, double* __restrict__ out)
{
    const int t = threadIdx.x;
    const double v = in[t];
    out[t] = dppd<0x150 + 5>(v);
    out[64 + t] = row_sum<16>(v);
    out[128 + t] = row_max<16>(v);
    out[192 + t] = row_min<16>(v);
    out[256 + t] = (double)row_or<16>(1u << (t & 15));
    out[320 + t] = frcp(v);
    { const double r0 = __builtin_amdgcn_rcp(v); out[384 + t] = r0; const double e = __builtin_fma(-v, r0, 1.0); out[448 + t] = __builtin_fma(r0, e, r0); }
}

}  // namespace relmc
