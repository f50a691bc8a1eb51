// relmc_dev.h — device-side view of one study case (built on the host by relmc_case_load,
// resident in HBM, copied into LDS by every workgroup) and the tile constants the gfx950
// kernels are compiled for.
#pragma once
#include <stdint.h>

namespace relmc {

// ---- tile geometry: one scenario = one row of RW lanes of a wavefront (64 / RW scenarios per wavefront) -----
//   Tile24: 16-lane DPP rows, 4 scenarios per wavefront  (<= 32 buses, 48 lines, 64 injections: IEEE RTS-24)
//   Tile96: one scenario per wavefront                   (<= 128 buses, 128 lines, 192 injections: IEEE RTS-96)
template <int RW_, int BS_, int LS_, int IS_, int NCOMPMAX_, int MAXPASS_, int MAXOFF_, int WPB_>
struct TileT {
    static constexpr int RW = RW_;             // lanes per scenario
    static constexpr int BS = BS_;             // bus slots per lane
    static constexpr int LS = LS_;             // line slots per lane
    static constexpr int IS = IS_;             // injection slots per lane
    static constexpr int NBT = BS_ * RW_;
    static constexpr int NLT = LS_ * RW_;
    static constexpr int NIT = IS_ * RW_;
    static constexpr int SPW = 64 / RW_;       // scenarios per wavefront
    static constexpr int NCOMPMAX = NCOMPMAX_; // sampled components (generators + lines)
    static constexpr int OW = NCOMPMAX_ / 32;  // 32-bit words of the outage mask
    static constexpr int MAXPASS = MAXPASS_;   // passes of the static solver schedule
    static constexpr int MAXOFF = MAXOFF_;     // off-diagonal 2x2 blocks of the factor (lines + fill)
    static constexpr int WPB = WPB_;           // wavefronts per workgroup (they share the case tables)
    static_assert(BS_ <= 2 && LS_ <= 3 && IS_ <= 4, "the per-lane flag word has room for 2 bus, 3 line and 4 injection slots");
};
using Tile24 = TileT<16, 2, 3, 4, 128, 96, 160, 4>;
using Tile96 = TileT<64, 2, 2, 3, 256, 64, 320, 8>;
constexpr int DEGMAX = 8;          // lines per bus
constexpr int BINJMAX = 8;         // injections per bus

// l_info: from | to<<8 | flags<<24   (internal bus numbers = elimination positions)
constexpr uint32_t LF_EXISTS = 1u, LF_OWNER = 2u, LF_LIMITED = 4u;
// i_info: bus | kind<<8
constexpr uint32_t IK_NONE = 0u, IK_REAL = 1u, IK_VIRTUAL = 2u;

// Static solver schedule.  Passes [0, npass_upd) are PK_UPD, then npass_inv PK_INV, then PK_BWD.
// The last npass_updq PK_UPD passes hold at most RW / 4 tasks each (the top of the elimination tree) and come in QUARTER form: four lanes
// per task, lane 2r + c updating the single element T[r][c] -= (Wa[r] * inv(D)) . Wb[c] -- the same arithmetic per element, six LDS
// instructions per pass instead of ten, which is what a pass costs whatever its fill.  Quarter descriptor = (&T[r][c], &Wa[r][0], &Wb[c][0], D).
// In front of them, npass_updh passes of at most RW / 2 tasks in HALF form: two lanes per task, lane r updating the row T[r][:] (seven LDS
// instructions).  Half descriptor = (&T[r][0], &Wa[r][0], Wb, D).
//   PK_UPD  T -= Wa * inv(D) * Wb'         task = (T, Wa, Wb, D)      BYTE offsets into W of 2x2 blocks (bit 15 of T = rhs task; full form only)
//   PK_INV  P = inv(D); y <- P * y         task = (D, Y, P, -)
//   PK_BWD  y_i -= P_i * W' * x_a          task = (Y_i, W, P_i, Y_a)
//           half form (a pass of at most RW / 2 tasks): two lanes per task, lane r updating y_i[r] -= P_i[r][:] . (W' x_a) -- six LDS instructions
//           instead of seven.  Half descriptor = (&Y_i[r], W, &P_i[r][0], Y_a).
//
// Per-scenario solver workspace W (doubles), 2x2 blocks row-major [tt, tl, lt, ll]:
//   [0, 4*nb)                 diagonal blocks  D_i = [[M_ii, B_ii], [B_ii, -E_i]]
//   [4*nb, 4*(nb+noff))       off-diagonal blocks K(a, i), a eliminated after i (lines and fill)
//   [off_rhs, off_rhs+4*nb)   right-hand side as a pseudo-bus: block i = [[y_theta, y_lambda], [0, 0]]
//   [off_p, off_p+4*nb)       P_i = inv(D_i) after the factorisation
template <class TL>
struct DevCaseT {
    static constexpr int NBT = TL::NBT, NLT = TL::NLT, NIT = TL::NIT, NCOMPMAX = TL::NCOMPMAX, MAXPASS = TL::MAXPASS, MAXOFF = TL::MAXOFF, ROWL = TL::RW;
    int32_t nb, ng, nl, nd, ninj, ncomp, ref_bus, noff;
    double base_mva, total_load;
    uint32_t exist_mask;            // bit i = bus i exists (16-lane tile only)
    uint32_t nws;                   // doubles in W
    uint16_t off_rhs, off_p;
    uint16_t npass, npass_upd, npass_inv, nzero;
    uint16_t maxdeg, maxinj, base_connected, npass_updq;   // npass_updq: the last PK_UPD passes in quarter form (below)   // base_connected: the network with every line in service is one island   // largest number of lines / injections at one bus
    uint16_t maxdeg_s[2], maxinj_s[2];   // longest line / injection list among the buses of bus slot 0 / 1
    uint16_t npass_updh, pad3[3];   // PK_UPD passes in half form, in front of the quarter-form ones
    uint64_t bwd_half;              // bit k: PK_BWD pass k (counted from the first one) is in half form; all of them or none (relmc_case_load)
    uint64_t b_line8[NBT];          // the bus' line list packed one byte each (id | 0x80 = 'to' end), unused entries = nl (the all-zero record)
    uint64_t b_inj8[NBT];           // the bus' injection list packed one byte each, unused entries = ninj (the all-zero record)
    // lines
    double b_bsum[NBT];             // sum of the susceptances of the bus' lines, in list order (all lines in service)
    double l_b[NLT];
    double l_rate[NLT];             // p.u. (0 = unlimited)
    uint32_t l_info[NLT];
    int32_t l_partner[NLT];         // the other line of the same bus pair, -1 if none
    uint16_t l_blk[NLT];            // owner lines: W offset of the pair's off-diagonal block
    // injections (real generators then virtual generators = loads)
    // per injection {upper bound p.u., lower bound p.u., cost c1 * baseMVA (opf_setup), original Pmin in MW (nodal shed = Pg - Pmin,
    // mc_simulation.m:86)}: bounds and cost come in with one or two 16-byte LDS reads
    alignas(16) double i_tab[NIT][4];
    uint32_t i_info[NIT];
    // per-bus incidence
    uint8_t b_nline[NBT];
    uint8_t b_line[NBT][DEGMAX];    // line id | 0x80 when the bus is the line's 'to' end
    uint8_t b_ninj[NBT];
    uint8_t b_inj[NBT][BINJMAX];
    int16_t b_vinj[NBT];            // virtual generator at the bus, -1 if none
    uint8_t b_ext[NBT];             // internal bus -> external bus number
    uint8_t b_lane[NBT];            // bus held by lane r of bus slot t at [RW * t + r] (0xff = none): the partly filled last slot gets the buses with the
                                    // shortest incidence lists, because a slot's gather loops run to the longest list among its buses (relmc_case_load)
    uint8_t b_int[NBT];             // external bus number -> internal bus
    uint32_t thr[NCOMPMAX];         // Bernoulli thresholds floor(U*2^32)
    // static schedule of the sparse block LDL' (computed once per case on the host)
    uint16_t zero_off[MAXOFF];      // W offsets of fill-only blocks (cleared every iteration)
    uint8_t pass_ntask[MAXPASS];
    uint16_t task[MAXPASS][ROWL][4];    // byte offsets into the scenario's workspace (< 32 KiB), 0xffff in field 0 = no task for this lane
};

// per-lane partial accumulators written once per workgroup-row, reduced by relmc_finalize_kernel
template <class TL>
struct PartialT {
    double dns, dns2;
    double shed[TL::IS];
    uint32_t n, nfail, nsing, ninf, nnc, iters;
    uint32_t cf_inj[TL::IS];
    uint32_t cf_line[TL::LS];
    uint32_t pad;
};

// one unit the primary schedule did not converge on
struct FailRec {
    unsigned long long unit;        // launch-local index + unit_base: sample offset (MODE 0), state (1), year * hours + hour (2), distinct state (3), row offset (4)
    uint32_t weight, pad;
    uint32_t mask[8];               // outage mask words
};

struct EvalArgs {
    uint64_t seed, first_index;
    int64_t n;
    // solver options
    int32_t policy, max_it;
    double feastol, gradtol, comptol, costtol, xi, sigma, z0, alpha_min, max_stepsize;
    // explicit states in / per-scenario results out (materialised mode), device pointers
    const uint8_t* states;
    double* dns;
    double* nodal;
    int32_t* status;
    int32_t* iters;
    void* partial;                  // PartialT<tile>[gridDim.x * blockDim.x]
    uint32_t prio_mode;             // issue-priority balancing between co-resident wavefronts: 0 off, 1 by grid half, 2 by wave half
    uint32_t scen_doubles;          // per-scenario LDS doubles (workspace + stash), = 2 mod 4
    uint32_t stash_off;             // start of the per-lane stash behind the workspace
    uint32_t case_bytes;            // bytes of the case tables copied to LDS (everything before the pass schedule)
    unsigned long long* timing;     // profiling builds: [waves][8] phase cycle counters (else null)
    double fail_threshold;          // loss flag: dns > 1e-4 (nsqMain.m:270) / > 0.01 (seqMain.m:41)
    const double* load_scale;       // MODE 1: optional per-scenario load scale factor
    // MODE 2 (sequential): chronology masks [years][hours][OW x u32], compacted contingency hours, hourly load curve
    const uint32_t* seq_masks;
    const uint32_t* seq_offsets;    // [years + 1] exclusive scan of contingency-hour counts
    const uint16_t* seq_hours;      // [years][hours] contingency hours of each year, ascending
    const double* load_factors;     // [hours]
    double* curt;                   // [years][hours] hourly curtailment (MW), zero where not evaluated
    int32_t seq_nyears, seq_hpy;
    // MODE 3 (distinct states with multiplicities, nsqMain.m:220-245): scenario u = state memo_keys[memo_perm[memo_start[u]]]
    // counted memo_start[u+1] - memo_start[u] times
    const uint32_t* memo_keys;      // [n][OW] outage masks of the sampled range
    const uint32_t* memo_perm;      // [n] sample indices sorted by mask
    const uint32_t* memo_start;     // [n_distinct + 1] first sorted position of every distinct mask
    // MODE 4 (rows of the persistent state database): scenario u = row db_first + u, key words at memo_keys[row][OW],
    // results go to dns[row], status[row] (packed) and nodal[row][nb].  Behind the zero-curtailment pre-screen memo_perm lists the new rows the
    // certificate did not cover: scenario u = row db_first + memo_perm[u] (null = all rows)
    // MODE 7 (fused path behind the pre-screen): scenario u = sample memo_perm[u] of the range, state memo_keys[memo_perm[u]][OW]
    int64_t db_first;
    // Units (samples / states / rows) that end non-converged (status 1 or 2) are listed here instead of being accumulated; the host
    // evaluates them again under further elimination orders (relmc_retry.hip: fail_retry).  Null = off.
    uint32_t* fail_count;
    FailRec* fail_list;
    uint32_t fail_cap;
    int64_t unit_base;              // added to the launch-local unit index (chunked host-buffer pipeline)
    // MODE 6 (dense pivoted last resort): [scenario rows of the grid][dense_stride] doubles of global scratch, dense_stride >= 2 nb (2 nb + 1)
    double* dense;
    uint64_t dense_stride;
};

// Zero-curtailment certificate (relmc_screen.hip; SURVEY 8f rank 4, mc_simulation.m:57-59, 65): tables built by relmc_case_load on the host, resident in
// HBM, read through the vector-memory path.  A state is certified when the units in service, loaded proportionally between Pmin and Pmax,
// serve the whole load with every DC flow inside (or on) its rating: base-topology PTDF, one or two lines out through the outage system
// (I - H_MM) x = F_M, more lines out never.
struct ScreenTab {
    int32_t nl, ng, valid, pad;
    double total_load;              // MW at load scale 1
    double sum_pmin, sum_rng;       // MW over all units: sum Pmin, sum (Pmax - Pmin)
    const double* pmin;             // [ng] MW
    const double* rng;              // [ng] MW, Pmax - Pmin
    const double* f_min;            // [nl] MW flow of (all units at Pmin) through the PTDF
    const double* f_rng;            // [nl] MW flow of (all units' ranges)
    const double* f_load;           // [nl] MW flow of the bus loads at scale 1 (positive = the loads' own contribution, subtracted)
    const double* lim;              // [nl] MW rating + 1e-9 MW of rounding slack; +inf = no limit
    const double* gpair;            // [nl][ng][2] {PTDF[l, bus(k)] * Pmin_k, PTDF[l, bus(k)] * (Pmax_k - Pmin_k)}, 16-byte aligned pairs
    const double* hmat;             // [nl (line out m)][nl] H[m][l]: flow on l per MW sent from from(m) to to(m) in the base topology (the outage formulas' matrix)
    const uint8_t* bridge;          // [nl] 1 = taking the line out splits the network (1 - H[m][m] = 0; the certificate tests that itself)
};

// device image of relmc_acc (include/relmc.h): 6 + 256 + 1 int64, then 2 + 128 doubles
struct DevAcc {
    long long n, n_fail, n_singular, n_infeasible, n_nonconverged, sum_iters;
    long long comp_fail[256];
    long long n_screened;           // never written by the evaluation kernels: the pre-screen's count is added on the host (relmc_screen.hip)
    double sum_dns, sum_dns2;
    double sum_nodal[128];
};

constexpr int FIN_ITEMS = 8 + 256 + 128;

constexpr int NCOMPMAX = 128;       // unit capacity of the HL1 fleet tables
constexpr int SEQ_NCOMPMAX = 256;   // component capacity of the sequential chronology (both tiles)
struct SeqCase {
    int32_t ncomp, hpy, mw, pad;    // mw: 32-bit mask words per hour (= the tile's OW: 4 or 8)
    double mttf[SEQ_NCOMPMAX], mttr[SEQ_NCOMPMAX];
};

struct Hl1Case {
    int32_t ngen, nhours;
    uint32_t thr[NCOMPMAX];          // unit g down iff draw < thr[g]  (up iff rand() >= for_rate)
    double cap[NCOMPMAX];
};


}  // namespace relmc
