// relmc_ctx.h — what the translation units of librelmc.so share: the context behind the opaque relmc_ctx handle of include/relmc.h, the
// error helpers, and the internal entry points each unit offers the others.  Host code only orchestrates (case tables -> HBM, launches on
// the context's stream, HIP-event timing, deterministic partial reduction on the device, estimator arithmetic); there is no CPU
// evaluation path anywhere in the library.
//
//   relmc_schedule.hip   symbolic analysis of a case and the static solver schedule (host arithmetic only), the order tuner
//   relmc_core.hip       context lifetime, relmc_case_load + order calibration, the evaluation-kernel launcher, mc_sampling, estimators
//   relmc_retry.hip      units the primary elimination order does not converge on: further static orders, dense pivoted last resort
//   relmc_simulate.hip   mc_simulation (host-buffer pipeline), the fused nsq_accumulate, the nsqMain loop (relmc_nsq_run)
//   relmc_database.hip   the reference's dedupe and persistent unique-state database on the device
//   relmc_comm.hip       the path's single collective: RCCL (bound at run time) or a host-supplied all-reduce, with a wall-clock guard
//   relmc_seq.hip        sequential track (chronology, scaled-load hours, annual indices, the seqMain loop) and the HL1 copper sheet
//   relmc_screen.hip     the zero-curtailment pre-screen (relmc_solver_opts.screen): certificate tables, pre-pass kernels, worklists
//   relmc_debug.hip      introspection and test hooks that are not part of include/relmc.h
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <functional>
#include <string>
#include <vector>

#include "../../include/relmc.h"
#include "relmc_dev.h"

// Diagnosis switches of a context.  The default build sets them through relmc_debug_set only (tests); a -DRELMC_DEV_SWITCHES build
// (csrc/Makefile: ablate/librelmc_dev.so) also reads the RELMC_* environment variables of the same names once at relmc_ctx_create.
struct relmc_switches {
    bool no_retry = false;           // the first attempt's results as they are (no list of non-converged units)
    bool retry_dense_first = false;  // listed units straight to the dense pivoted solve
    bool nsq_no_stretch = false;     // relmc_nsq_run: one launch per batch
    bool db_no_probe = false;        // state database: every batch through the dedupe, no per-sample probe
};

struct relmc_ctx {
    int device = -1;
    relmc_switches sw;
    // relmc_seq_years' device buffers, kept between calls (six hipMalloc / hipFree pairs per call were 1 ms of a 17 ms step): grow-only
    uint32_t* sq_dm = nullptr; size_t sq_dm_words = 0;
    uint16_t* sq_hours = nullptr; double* sq_curt = nullptr; size_t sq_nh = 0;
    uint32_t* sq_counts = nullptr; uint32_t* sq_off = nullptr; double* sq_year = nullptr; int sq_years = 0;
    std::vector<int32_t> order_hint;     // relmc_case_order_hint: primary elimination order of the next relmc_case_load (external bus numbers), empty = the rule
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool has_case = false;
    int tile = 0;                        // 0: Tile24 (16-lane rows), 1: Tile96 (one scenario per wavefront)
    relmc::DevCaseT<relmc::Tile24> hcase24;
    relmc::DevCaseT<relmc::Tile96> hcase96;
    void* dcase = nullptr;               // device image of the active tile's case
    int nb = 0, ng = 0, nl = 0, ncomp = 0;
    void* dpartial = nullptr;
    size_t partial_bytes = 0;
    relmc::DevAcc* dacc = nullptr;
    struct HostStage { relmc_acc acc; uint32_t fail_cnt, pad; }* hstage = nullptr;      // pinned: accumulators + listed-unit count of a fused launch come back in one synchronisation
    int num_cu = 0;
    int blocks_per_cu = 0;
    uint32_t scen_doubles = 0, lds_bytes = 0, stash_off = 0;
    unsigned long long* dtiming = nullptr; int timing_waves = 0;
    double* dhist = nullptr; double* hhist = nullptr /* pinned */; int64_t hist_cap = 0;   // per-sample dns of one launch (checkpoint histories of small batches, relmc_nsq_run)
    // distinct-state path: device buffers sized for memo_cap samples
    int64_t memo_cap = 0; size_t memo_tmp_bytes = 0;
    uint32_t *mk = nullptr, *mperm0 = nullptr, *mperm1 = nullptr, *mhead = nullptr, *muid = nullptr, *mstart = nullptr, *mnu = nullptr;
    unsigned long long *mch0 = nullptr, *mch1 = nullptr; void* mtmp = nullptr;
    uint32_t *mmiss = nullptr, *mk2 = nullptr;   // probe-first database path: miss list (sample indices) and the masks of the misses
    // persistent state database (nsqMain.m:91-99): rows in HBM, open-addressing table of row ids
    int64_t db_cap = 0, db_n = 0, db_samples = 0; uint64_t db_tcap = 0;
    uint32_t* db_keys = nullptr; unsigned long long* db_count = nullptr; double* db_dns = nullptr; int32_t* db_meta = nullptr;
    double* db_nodal = nullptr; uint32_t* db_table = nullptr; relmc::DevAcc* db_partial = nullptr; int db_partial_cap = 0;
    bool db_has_opts = false; relmc_solver_opts db_opts;
    bool db_invalid = false;                 // an entry point failed between the count bumps of a batch and its bookkeeping: relmc_db_reset / relmc_case_load only
    // retry of the units the primary elimination order does not converge on: a second device image of the case built with
    // another static order (lazily, from a copy of the description), the kernel's list of such units, scratch rows for their re-evaluation
    struct CaseCopy {
        relmc_case_desc d; bool valid = false;
        std::vector<double> bus_pd, inj_pmin, inj_pmax, inj_cost, br_b, br_rate, unavail; std::vector<int32_t> inj_bus, br_from, br_to; std::vector<uint8_t> always_up;
    } case_copy;
    static constexpr int kAlt = 2;           // further static orders: [0] the primary rule with the ties broken the other way, [1] fill first
    int alt_state[kAlt] = {0, 0};            // 0 not built yet, 1 ready, -1 unavailable (that order does not fit the tile)
    void* dcase_alt[kAlt] = {nullptr, nullptr}; uint32_t alt_scen_doubles[kAlt] = {0, 0}, alt_lds_bytes[kAlt] = {0, 0}, alt_stash_off[kAlt] = {0, 0};
    relmc::FailRec* dfail = nullptr; uint32_t* dfail_count = nullptr; bool fail_dirty = false;
    uint32_t fail_cap = 0;                   // entries of dfail (grows with the size of the call, fail_arm)
    double* ddense = nullptr; size_t dense_bytes = 0;      // global scratch of the dense pivoted last resort (MODE 6)
    int64_t retry_dense_units = 0, retry_dense_converged = 0;    // units that went to it since the case was loaded
    int64_t retry_overflow = 0;              // units that did not fit the list and kept their first-attempt results (relmc_retry_overflow)
    std::vector<double> hlf;                 // host copy of the hourly load factors (load scale of a re-evaluated hour)
    // scratch rows of the re-evaluation, sized for rcap listed units of a case with rnb buses (twice: the third order's compact rows)
    uint32_t* rkeys = nullptr; double* rdns = nullptr; int32_t* rmeta = nullptr; double* rnodal = nullptr; double* rscale = nullptr; int64_t rcap = 0; int rnb = 0;
    int64_t retry_units = 0, retry_converged = 0;        // since the case was loaded
    int order_primary = 0; int32_t order_probe[3] = {-1, -1, -1};   // which static order runs first, and the calibration's failure counts (-1 = not probed)
    unsigned long long* db_snap = nullptr; int64_t db_snap_cap = 0;      // row counts before a stretch of small batches (relmc_nsq_run)
    // host-buffer entry points (relmc_mc_simulation, relmc_seq_mcsimulation): double-buffered chunk pipeline, device buffers
    // and pinned staging kept across calls
    struct HostPipe {
        bool ready = false; int ncomp = 0, nb = 0;
        hipStream_t up = nullptr, down = nullptr;
        hipEvent_t e_up[2] = {nullptr, nullptr}, e_ks[2] = {nullptr, nullptr}, e_ke[2] = {nullptr, nullptr}, e_down[2] = {nullptr, nullptr};
        uint8_t* d_st[2] = {nullptr, nullptr}; double* d_sc[2] = {nullptr, nullptr}; double* d_dns[2] = {nullptr, nullptr}; double* d_nod[2] = {nullptr, nullptr};
        int32_t* d_stat[2] = {nullptr, nullptr}; int32_t* d_it[2] = {nullptr, nullptr};
        uint8_t* h_st[2] = {nullptr, nullptr}; double* h_sc[2] = {nullptr, nullptr}; double* h_dns[2] = {nullptr, nullptr}; double* h_nod[2] = {nullptr, nullptr};
        int32_t* h_stat[2] = {nullptr, nullptr}; int32_t* h_it[2] = {nullptr, nullptr};
    } pipe;
    // communicator over the ranks of a multi-GPU run (optional; relmc_comm_*): RCCL, or the host's own collective
    void* comm = nullptr; int comm_nranks = 0, comm_rank = -1;
    relmc_allreduce_fn host_allreduce = nullptr; void* host_allreduce_user = nullptr;
    relmc_allreduce_f64_fn host_allreduce_f64 = nullptr; void* host_allreduce_f64_user = nullptr;   // optional vector transport of the host collective
    int64_t comm_calls = 0; double comm_seconds = 0.0;                                     // all-reduces of relmc_acc through this context, wall time in them
    double comm_timeout_s = 120.0;                                                         // wall-clock guard of communicator init and of every collective
    void* watchdog = nullptr;                                                              // the guard's thread (relmc_comm.hip), started at the first guarded call
    double* dgather = nullptr; size_t gather_doubles = 0;                                  // device staging of comm_allreduce_f64 (RCCL)
    // sequential track
    bool has_seq = false; relmc::SeqCase hseq; relmc::SeqCase* dseq = nullptr; double* dlf = nullptr;
    // HL1 copper-sheet model
    bool has_hl1 = false; relmc::Hl1Case* dhl1 = nullptr; double* dsorted = nullptr; double* dsuffix = nullptr; int hl1_hours = 0;
    double* h1_lole = nullptr; double* h1_eue = nullptr; int64_t h1_cap = 0; double* h1_part = nullptr; int64_t h1_part_cap = 0;   // relmc_hl1_nsq's device buffers, grow-only
    // zero-curtailment pre-screen (relmc_screen.hip): certificate tables of the case (device pointers inside tab), grow-only work buffers of a pre-pass
    struct Screen {
        relmc::ScreenTab tab = {};
        void* dtab = nullptr;
        uint32_t* keys = nullptr; uint8_t* flags = nullptr; uint32_t* idx = nullptr; uint32_t* dcount = nullptr; void* tmp = nullptr;
        size_t tmp_bytes = 0; int64_t cap = 0;
        hipEvent_t ev0 = nullptr, ev1 = nullptr;
    } screen;
    double last_kernel_ms = 0.0;
    long conflict_before = 0, conflict_after = 0;   // modelled extra LDS cycles per Newton step before / after the placement search
    long alt_conflict_before[kAlt] = {0, 0}, alt_conflict_after[kAlt] = {0, 0};      // the same of the further orders' images
    std::string err;
};

namespace relmc_host {

using namespace relmc;

extern const char* const kNoCtx;
int fail(relmc_ctx* ctx, int code, const std::string& msg);
bool verbose();                           // RELMC_VERBOSE in the environment (the library's only environment variable), read once

#define HIP_TRY(ctx, expr)                                                                                   \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess)                                                                                \
            return ::relmc_host::fail(ctx, RELMC_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

// ---- relmc_schedule.hip (no HIP call, no context) -----------------------------------------------------------------------------
struct SymOpts {
    const int32_t* order_hint = nullptr; int n_hint = 0;   // primary order from the host (external bus numbers, reference bus last); order_variant 0 only
    int place_moves = 100;               // moves per block of the LDS placement search (0: the order tuner's cost does not depend on the placement)
    long place_ww = 1;                   // weight of conflicts on operands that are written back
    bool no_quarter = false, no_half = false, no_bwd_half = false, no_bus_map = false;      // schedule forms off (ablation builds)
    int scen_pad4 = 0;                   // extra padding of a scenario row's LDS stride in units of 4 doubles (ablation builds)
    int model_leaf_free = -1;            // >= 0: scheduling MODEL with the pivots complete at assembly left out (relmc_debug_symbolic only: not a valid program)
};
struct SymGeom { uint32_t stash_off = 0, scen_doubles = 0, lds_bytes = 0; long conflict_before = 0, conflict_after = 0; };
SymOpts sym_opts_default();              // the shipped schedule; a RELMC_DEV_SWITCHES build reads the ablation variables here
template <class TL>
int case_symbolic(const relmc_case_desc* d, DevCaseT<TL>& C, int order_variant, const SymOpts& so, SymGeom& geom, std::string& err);
extern template int case_symbolic<Tile24>(const relmc_case_desc*, DevCaseT<Tile24>&, int, const SymOpts&, SymGeom&, std::string&);
extern template int case_symbolic<Tile96>(const relmc_case_desc*, DevCaseT<Tile96>&, int, const SymOpts&, SymGeom&, std::string&);
inline bool fits_tile24(const relmc_case_desc* d)
{
    return d->nb <= Tile24::NBT && d->nl <= Tile24::NLT && d->ng + d->nd <= Tile24::NIT && d->ng + d->nl <= Tile24::NCOMPMAX;
}

// ---- relmc_core.hip -----------------------------------------------------------------------------------------------------------
EvalArgs make_args(const relmc_solver_opts& o);
// launches relmc_eval_kernel<mode, active tile> on the context's stream between two events (default: ev0 / ev1); alt = 0 the primary image of
// the case, 1.. = the further orders'; *rows_out = scenario rows holding partial accumulators
int launch_eval(relmc_ctx* ctx, int mode, EvalArgs& a, int* rows_out, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr, int alt = 0);
int launch_finalize(relmc_ctx* ctx, int rows);     // deterministic reduction of the partial records into ctx->dacc
int finish_timing(relmc_ctx* ctx);                 // stream synchronised, ctx->last_kernel_ms = ev0 .. ev1
int case_load_image(relmc_ctx* ctx, const relmc_case_desc* d, int order_variant);   // device image of the case under a further order (retry levels)
inline int mask_words(const relmc_ctx* ctx) { return ctx->tile == 0 ? Tile24::OW : Tile96::OW; }

// ---- relmc_retry.hip ----------------------------------------------------------------------------------------------------------
constexpr uint32_t kFailCapMin = 4096, kFailCapSteady = 1u << 20, kFailCapMax = 1u << 26;
uint32_t fail_cap_for(int64_t call_units);
int fail_list_ensure(relmc_ctx* ctx, uint32_t cap);
struct RetryOut { std::vector<FailRec> rec; std::vector<double> dns, nodal; std::vector<int32_t> meta; };   // meta = status | relaxed << 2 | iterations << 8
int alt_ensure(relmc_ctx* ctx, int v);
int fail_arm(relmc_ctx* ctx, EvalArgs& a, int64_t unit_base, bool reset, int64_t call_units);
int fail_listed(relmc_ctx* ctx, uint32_t* cnt);
using ScaleFn = std::function<double(unsigned long long)>;          // unit -> load scale factor of a re-evaluated unit
// known_count (optional): the number of listed units if the caller has already copied it back with its results
int fail_retry(relmc_ctx* ctx, const relmc_solver_opts& o, double fail_threshold, const ScaleFn* scale, RetryOut& out, double* ms, const uint32_t* known_count = nullptr);
void acc_add_unit(relmc_acc* acc, const FailRec& rec, double dns, int32_t meta, const double* nodal, int nb, int ncomp, double fail_threshold);
void retry_free(relmc_ctx* ctx);

// ---- relmc_simulate.hip -------------------------------------------------------------------------------------------------------
void pipe_free(relmc_ctx* ctx);
int pipe_run(relmc_ctx* ctx, const uint8_t* states, const double* load_scale, int64_t n, const relmc_solver_opts& o, double fail_threshold,
             double* dns, double* nodal, int32_t* status, int32_t* iters);
int nsq_accumulate_impl(relmc_ctx* ctx, uint64_t seed, uint64_t first_index, int64_t n, const relmc_solver_opts* opts, relmc_acc* acc_out, double* dns_dev);

// ---- relmc_database.hip -------------------------------------------------------------------------------------------------------
void db_free(relmc_ctx* ctx);
void memo_free(relmc_ctx* ctx);
int db_accumulate(relmc_ctx* ctx, relmc_acc* acc_out);
int db_rewind(relmc_ctx* ctx, int64_t rows0, int64_t samples0);      // back to the first rows0 rows with the counts saved in ctx->db_snap
int db_snapshot(relmc_ctx* ctx);                                    // saves the counts of the present rows into ctx->db_snap
int db_sample_dns(relmc_ctx* ctx, uint64_t seed, uint64_t first_index, int64_t m, double* dns_dev);   // dns of every sample of a range from its row

// ---- relmc_comm.hip -----------------------------------------------------------------------------------------------------------
void comm_free(relmc_ctx* ctx);
inline int comm_ranks(const relmc_ctx* ctx) { return (ctx->comm || ctx->host_allreduce) ? ctx->comm_nranks : 1; }
// sum over the ranks of a vector of doubles, in place (the all-gather of the sequential loop: every rank fills its own slots, zeros elsewhere --
// x + 0 + ... + 0 is exact).  RCCL: one ncclAllReduce; host collective: through the registered relmc_acc all-reduce, 130 doubles per call
int comm_allreduce_f64(relmc_ctx* ctx, double* buf, int64_t count);

// ---- relmc_seq.hip ------------------------------------------------------------------------------------------------------------
void seq_free(relmc_ctx* ctx);

// ---- relmc_screen.hip ---------------------------------------------------------------------------------------------------------
void screen_free(relmc_ctx* ctx);
int screen_build(relmc_ctx* ctx, const relmc_case_desc* d);          // relmc_case_load: PTDF / LODF tables of the certificate
// samples [first_index, first_index + m): masks of the uncovered ones in ctx->screen.keys (own position), their ascending positions in ctx->screen.idx
int screen_prepass_nsq(relmc_ctx* ctx, uint64_t seed, uint64_t first_index, int64_t m, uint32_t* n_surv, double* ms);
int screen_gather_keys(relmc_ctx* ctx, uint32_t n_surv, uint32_t* keys_out);               // the uncovered samples' masks of the last screen_prepass_nsq, packed in sample order
int screen_prepass_rows(relmc_ctx* ctx, int64_t first, int64_t n, uint32_t* n_surv);      // new database rows: certified ones filled in, the others listed
int screen_seq_compact(relmc_ctx* ctx, const uint32_t* masks, int n_years, uint16_t* hours, uint32_t* counts, uint32_t* ncont);
constexpr int64_t kScreenChunk = (int64_t)1 << 22;                   // samples per pre-pass of the fused path (its buffers: 4 OW + 5 bytes per sample)

}  // namespace relmc_host
