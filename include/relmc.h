/*
 * relmc.h — C ABI of the MI355X-native non-sequential Monte Carlo HL2 engine.
 *
 * Drop-in boundary for the hot path of Matrixeigs/PowerSystemsReliabilityAssessment
 * (SURVEY.md §8b).  The reference has no FFI of its own: its "operator API" is three
 * MATLAB signatures, which the entry points below replace one for one:
 *
 *   relmc_case_load        <- loadcase + load model + failprob
 *                             (Montecarlo_nsq_single/nsqMain.m:42,121-153,167; failprob.m:1-41)
 *   relmc_mc_sampling      <- eqstatus = mc_sampling(U, n, Ng, Nl)          (mc_sampling.m:2)
 *   relmc_mc_simulation    <- [dns, nodal_dns] = mc_simulation(state, ...)  (mc_simulation.m:1),
 *                             batched over states (the parfor of nsqMain.m:257-263)
 *   relmc_nsq_accumulate   <- one pass of the main loop body, fused sample -> evaluate -> reduce
 *                             (nsqMain.m:212, 257-263, 269-278 folded into per-sample sums)
 *   relmc_nsq_indices      <- the estimators of nsqMain.m:282-301, 348-349, 366-376
 *   relmc_nsq_run          <- the whole `while beta > beta_limit` loop, nsqMain.m:208-318
 *
 * Conventions: plain C, no C++ or torch types; every buffer is owned by the caller and is
 * not retained after the call returns; row-major arrays; all floating point is fp64;
 * return value 0 = ok, <0 = relmc_status error (text via relmc_last_error); calls on one
 * context are blocking and not re-entrant, distinct contexts are independent.  One context
 * drives ONE GPU (one process per GPU; multi-GPU runs shard the global scenario index range
 * and all-reduce relmc_acc).  There is NO CPU fallback: if no HIP device
 * is usable every entry point fails with RELMC_ERR_NO_DEVICE.
 */
#ifndef RELMC_H
#define RELMC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RELMC_MAX_BUS 128
#define RELMC_MAX_COMP 256

typedef enum {
    RELMC_OK = 0,
    RELMC_ERR_INVALID = -1,      /* bad argument */
    RELMC_ERR_NO_DEVICE = -2,    /* no usable HIP device / HIP runtime error at create */
    RELMC_ERR_HIP = -3,          /* HIP runtime error (see relmc_last_error) */
    RELMC_ERR_UNSUPPORTED = -4,  /* case larger than the compiled kernel tiles */
    RELMC_ERR_NO_CASE = -5       /* relmc_case_load has not been called */
} relmc_status;

/* How states that leave a bus with no in-service branch are evaluated.
 * Known deviation: a MULTI-bus island without the reference bus also makes MATPOWER's KKT matrix singular (a constant angle
 * shift of the island is a null vector), but what MATLAB's `\` returns on that nearly singular matrix cannot be known
 * here; such states (about 1e-6 per RTS-24 sample) are solved island-aware under BOTH policies and are not claimed to
 * emulate the reference.  relmc_acc.n_infeasible / n_singular let a caller bound how many states any policy touched. */
typedef enum {
    RELMC_REFERENCE_EMULATE = 0, /* MATPOWER/MIPS fails in iteration 1, the start point is
                                    consumed unchecked (mc_simulation.m:41,54): dns = load/2 */
    RELMC_PHYSICAL = 1           /* island-aware LP: every island gets its own angle reference */
} relmc_singular_policy;

/* Per-scenario solver status. */
typedef enum {
    RELMC_ST_CONVERGED = 0,
    RELMC_ST_MAXIT = 1,          /* max_it reached (MIPS eflag 0)       */
    RELMC_ST_NUMFAIL = 2,        /* MIPS "numerically failed" (eflag -1) */
    RELMC_ST_SINGULAR = 3        /* isolated bus under RELMC_REFERENCE_EMULATE */
} relmc_scenario_status;

typedef struct relmc_ctx relmc_ctx;

/* Study case = MATPOWER case after the load model of nsqMain.m:121-153, 0-based indices.
 * Injections are the LP's generator columns in MATPOWER order: ng real generators then nd
 * virtual generators (one per load bus: Pmax = 0, Pmin = -Pd, cost 1). */
typedef struct {
    double base_mva;
    int32_t nb, ng, nl, nd;
    int32_t ref_bus;
    const double* bus_pd;     /* [nb]    MW, original bus loads                    */
    const int32_t* inj_bus;   /* [ng+nd]                                           */
    const double* inj_pmin;   /* [ng+nd] MW                                        */
    const double* inj_pmax;   /* [ng+nd] MW                                        */
    const double* inj_cost;   /* [ng+nd] linear cost c1 ($/MWh)                    */
    const int32_t* br_from;   /* [nl]                                              */
    const int32_t* br_to;     /* [nl]                                              */
    const double* br_b;       /* [nl]    1/(x*tap) p.u. (makeBdc)                  */
    const double* br_rate;    /* [nl]    MW, 0 = unconstrained                     */
    const double* unavail;    /* [ng+nl] failure probabilities (failprob.m:39)     */
    const uint8_t* always_up; /* [ng+nl] 1 = forced available (mc_sampling.m:40-41) */
    double total_load;        /* TestSystem.load, nsqMain.m:125                    */
} relmc_case_desc;

/* MIPS options as MATPOWER sets them for OPF_ALG_DC=200 (nsqMain.m:185-186). */
typedef struct {
    int32_t singular_policy;  /* relmc_singular_policy */
    int32_t max_it;           /* 150 */
    double feastol;           /* 5e-6 (opf.violation) */
    double gradtol;           /* 1e-6 */
    double comptol;           /* 1e-6 */
    double costtol;           /* 1e-6 */
    double xi;                /* 0.99995 */
    double sigma;             /* 0.1 */
    double z0;                /* 1 */
    double alpha_min;         /* 1e-8 */
    double max_stepsize;      /* 1e10 */
    int32_t screen;           /* 0 (default): every unit goes through the interior point.  1: zero-curtailment pre-screen: a unit
                                 for which an explicit dispatch serves all load -- units in service loaded proportionally between Pmin and Pmax, DC
                                 flows through the base-topology PTDF (one or two lines out: + the outage system's corrections) inside or on every rating -- has LP optimum 0, so the
                                 reference's outputs for it are exactly (0, zeros) (mc_simulation.m:57-59, 65) and it is counted without being solved.
                                 Every output but the iteration statistics is the one of screen = 0 (of its converged solves: a unit the interior point
                                 would have left non-converged is counted with the proven optimum); relmc_acc.n_screened counts the skipped units.
                                 Honoured by the fused pass (relmc_nsq_accumulate, relmc_nsq_run), by the per-batch dedupe
                                 (relmc_nsq_accumulate_distinct: only the uncovered samples are sorted, their distinct states solved), by the new
                                 rows of the state database (relmc_nsq_db_batch: a certified row carries 0 iterations) and by the sequential
                                 track's contingency hours at their own load factor (relmc_seq_years, relmc_seq_run); relmc_mc_simulation
                                 solves every state it is handed. */
    int32_t reserved;         /* 0 */
} relmc_solver_opts;

/* Additive accumulators of one scenario range (what is all-reduced across GPUs). */
typedef struct {
    int64_t n;                /* scenarios evaluated                                  */
    int64_t n_fail;           /* dns > 1e-4 (nsqMain.m:270)                            */
    int64_t n_singular;       /* RELMC_ST_SINGULAR                                     */
    int64_t n_infeasible;     /* states where an island needed Pmin relaxation / decommit */
    int64_t n_nonconverged;   /* RELMC_ST_MAXIT or RELMC_ST_NUMFAIL                    */
    int64_t sum_iters;        /* IPM iterations                                        */
    int64_t comp_fail[RELMC_MAX_COMP]; /* sum over failed scenarios of state_k (nsqMain.m:373-376) */
    int64_t n_screened;       /* of n: certified zero-curtailment by the pre-screen and not solved (relmc_solver_opts.screen) */
    double sum_dns;           /* sum dns      (nsqMain.m:286)                          */
    double sum_dns2;          /* sum dns^2    (for beta, nsqMain.m:299-301)            */
    double sum_nodal[RELMC_MAX_BUS];   /* sum nodal_dns (nsqMain.m:348)               */
} relmc_acc;

/* Reliability indices (nsqMain.m:282-301, 348-349, 366-376). */
typedef struct {
    int64_t n;
    double edns;              /* MW                       */
    double lole;              /* h/yr = plc*hours_per_year */
    double plc;
    double beta;              /* CoV of EDNS              */
    double eens;              /* MWh/yr = edns*hours_per_year */
    double mean_iters;
    double nodal_eens[RELMC_MAX_BUS];        /* MW (the reference's nodal_eens; x8760 in its CSV) */
    double comp_importance[RELMC_MAX_COMP];  /* P(component down | system failure) */
} relmc_indices;

typedef struct {
    double beta_limit;        /* nsqMain.m:60  (0.0017) */
    int64_t max_samples;      /* nsqMain.m:61  (1e5)    */
    int64_t batch;            /* scenarios per convergence check (nsqMain.m:62 uses 100) */
    uint64_t seed;
    double hours_per_year;    /* 8760, nsqMain.m:292 */
    relmc_solver_opts solver;
    /* optional history buffers (may be NULL); capacity in checkpoints */
    int64_t history_cap;
    double* beta_history;
    double* edns_history;
    double* lole_history;
    double* plc_history;
    int32_t distinct_states;  /* 0 (default): solve every sample.
                                 1: evaluate every distinct state of a batch once and count multiplicities (the
                                    reference's dedupe, nsqMain.m:220-229, per batch).
                                 2: the reference's persistent unique-state database (nsqMain.m:91-99, 220-278): states
                                    seen in an earlier batch only bump their count, only new states are solved, and the
                                    indices are recomputed from all rows after every batch (:282-301).  The run starts
                                    from an empty database (relmc_db_reset).
                                 The estimators are identical in all three. */
} relmc_nsq_opts;

typedef struct {
    relmc_acc acc;
    relmc_indices idx;
    int64_t checkpoints;      /* history entries written = min(batches, history_cap) */
    int32_t converged;        /* beta <= beta_limit */
    double wall_seconds;      /* host wall time of the loop */
    double kernel_seconds;    /* HIP-event time of the fused kernels */
    int64_t batches;          /* convergence checks made (passes of the nsqMain.m:208 loop) */
} relmc_nsq_result;

/* ---- lifetime ---------------------------------------------------------------------- */
int32_t relmc_ctx_create(int32_t device_id, relmc_ctx** out);
void relmc_ctx_destroy(relmc_ctx* ctx);
const char* relmc_last_error(const relmc_ctx* ctx);
const char* relmc_version(void);

/* ---- setup ------------------------------------------------------------------------- */
void relmc_solver_opts_default(relmc_solver_opts* opts);
void relmc_nsq_opts_default(relmc_nsq_opts* opts);
int32_t relmc_case_load(relmc_ctx* ctx, const relmc_case_desc* desc);
/* integer Bernoulli thresholds floor(U*2^32) the sampler compares against; out[ng+nl] */
int32_t relmc_case_thresholds(const relmc_ctx* ctx, uint32_t* out);

/* ---- mc_sampling (mc_sampling.m:2) -------------------------------------------------- */
/* eqstatus[n x (ng+nl)] row-major, 1 = failed.  Deterministic in (seed, global index). */
int32_t relmc_mc_sampling(relmc_ctx* ctx, uint64_t seed, uint64_t first_index, int64_t n,
                          uint8_t* eqstatus_host);
int32_t relmc_mc_sampling_dev(relmc_ctx* ctx, uint64_t seed, uint64_t first_index, int64_t n,
                              uint8_t* eqstatus_dev);

/* ---- mc_simulation (mc_simulation.m:1), batched ------------------------------------- */
/* states[n x (ng+nl)]; dns[n]; nodal[n x nb]; status[n] / iters[n] may be NULL. */
int32_t relmc_mc_simulation(relmc_ctx* ctx, const uint8_t* states_host, int64_t n,
                            const relmc_solver_opts* opts, double* dns_host, double* nodal_host,
                            int32_t* status_host, int32_t* iters_host);
/* same with every buffer already resident in this device's HBM */
int32_t relmc_mc_simulation_dev(relmc_ctx* ctx, const uint8_t* states_dev, int64_t n,
                                const relmc_solver_opts* opts, double* dns_dev, double* nodal_dev,
                                int32_t* status_dev, int32_t* iters_dev);

/* ---- fused sample -> evaluate -> reduce (nsqMain.m:208-318 loop body) ---------------- */
/* Evaluates global scenarios [first_index, first_index+n) and returns their accumulators. */
int32_t relmc_nsq_accumulate(relmc_ctx* ctx, uint64_t seed, uint64_t first_index, int64_t n,
                             const relmc_solver_opts* opts, relmc_acc* acc_out);
/* HIP-event duration (ms) of the most recent fused / simulation kernel on the ctx stream */
/* The same accumulators through the reference's dedupe (nsqMain.m:220-245): sample the range, sort the outage masks on
 * the device, evaluate each distinct state once weighted by its multiplicity.  *n_distinct_out (optional) = states solved.
 * With opts->screen = 1 the pre-screen's certificate runs first and only the uncovered samples are sorted and solved.
 * relmc_last_kernel_ms then covers sampling + sort + evaluation. */
int32_t relmc_nsq_accumulate_distinct(relmc_ctx* ctx, uint64_t seed, uint64_t first_index, int64_t n,
                                      const relmc_solver_opts* opts, relmc_acc* acc_out, int64_t* n_distinct_out);
int32_t relmc_last_kernel_ms(const relmc_ctx* ctx, double* ms);

/* ---- persistent unique-state database (nsqMain.m:91-99, 220-278) ------------------------- */
/* The reference keeps every distinct state it has ever sampled in `state_database` (columns: Ng+Nl component states |
 * count | dns | failure flag | Nb nodal values, nsqMain.m:91-99).  Per batch it removes duplicates (:220-229), bumps the
 * counts of states already in the database (:232-245), evaluates only the new ones (:257-263), appends them (:269-278)
 * and recomputes the indices from all rows (:282-301).  The same on the device: rows in HBM, an open-addressing table of
 * row ids (probed per sample once the database is warm; only the misses are sorted and deduplicated), new rows appended in
 * the order of first appearance in the sample stream (unique(...,'stable')), so the database and every sum over it depend
 * on (seed, samples drawn) only, not on the batch size.
 *   relmc_db_reset       empty database (relmc_case_load does the same)
 *   relmc_nsq_db_batch   one pass of the loop body over samples [first_index, first_index + n); acc_out (optional) =
 *                        accumulators of the WHOLE database afterwards (cumulative, not the batch's increment)
 *   relmc_db_accumulate  the accumulators of the whole database again (no sampling)
 *   relmc_db_size        rows and samples held
 *   relmc_db_export      rows [first_row, first_row + n_rows) in the reference's column layout; any output may be NULL:
 *                        states[n x (ng+nl)], count[n], dns[n], flag[n] (dns > 1e-4, :270), nodal[n x nb], status[n], iters[n],
 *                        relaxed[n] (bit 0 = an island rule relaxed Pmin or decommitted units: the row counts in n_infeasible; bit 1 = the row was
 *                        certified by the pre-screen and never solved: it counts in n_screened)
 *   relmc_db_import      resume: exported rows back into an EMPTY database, same order (the table of row ids is rebuilt); the next
 *                        relmc_nsq_db_batch / relmc_nsq_run(distinct_states = 2) continues as if the run had never stopped.
 *                        opts = the solver options the rows were computed under (NULL = defaults); status / iters / relaxed may be NULL
 * After an error return of relmc_nsq_db_batch on a non-empty database the counts of known rows may have advanced without their
 * samples: the database then refuses every call but relmc_db_reset (and relmc_case_load) with RELMC_ERR_INVALID. */
typedef struct {
    int64_t rows;             /* distinct states in the database (database_row_count)     */
    int64_t samples;          /* samples they stand for (sum of the count column)          */
    int64_t new_rows;         /* states evaluated by this call (num_new_states, :255)       */
    int64_t batch_distinct;   /* distinct states among this call's samples that went through the dedupe: all of
                                 them on an empty database, only the misses of the per-sample probe on a warm one */
} relmc_db_stats;
int32_t relmc_db_reset(relmc_ctx* ctx);
int32_t relmc_nsq_db_batch(relmc_ctx* ctx, uint64_t seed, uint64_t first_index, int64_t n,
                           const relmc_solver_opts* opts, relmc_acc* acc_out, relmc_db_stats* stats_out);
int32_t relmc_db_accumulate(relmc_ctx* ctx, relmc_acc* acc_out);
int32_t relmc_db_size(const relmc_ctx* ctx, int64_t* rows_out, int64_t* samples_out);
int32_t relmc_db_export(relmc_ctx* ctx, int64_t first_row, int64_t n_rows, uint8_t* states_host, int64_t* count_host,
                        double* dns_host, int32_t* flag_host, double* nodal_host, int32_t* status_host, int32_t* iters_host,
                        uint8_t* relaxed_host);
int32_t relmc_db_import(relmc_ctx* ctx, const relmc_solver_opts* opts, int64_t n_rows, const uint8_t* states_host,
                        const int64_t* count_host, const double* dns_host, const double* nodal_host, const int32_t* status_host,
                        const int32_t* iters_host, const uint8_t* relaxed_host);

/* ---- multi-GPU: the path's single collective (the reference's parfor, nsqMain.m:257-263) ------------ */
/* One process per GPU, one context per process.  Scenarios shard by global index (any contiguous split of
 * [first_index, first_index + n) over the ranks); at a convergence check every rank calls relmc_comm_allreduce_acc on the
 * accumulators of its slice: ONE grouped RCCL all-reduce(sum) over xGMI (int64 counters and fp64 sums in their own types).
 * The communicator is bootstrapped like any RCCL communicator: rank 0 calls relmc_comm_unique_id and hands the 128 bytes
 * to the other ranks by the host's own means (file, socket, MPI, Julia Distributed, torch store), then every rank calls
 * relmc_comm_init.  RCCL is bound at run time (dlopen): hosts that never call these need no RCCL installed. */
#define RELMC_COMM_ID_BYTES 128
int32_t relmc_comm_unique_id(uint8_t id_out[RELMC_COMM_ID_BYTES]);
int32_t relmc_comm_init(relmc_ctx* ctx, int32_t nranks, int32_t rank, const uint8_t id[RELMC_COMM_ID_BYTES]);
int32_t relmc_comm_allreduce_acc(relmc_ctx* ctx, relmc_acc* acc_inout);
int32_t relmc_comm_destroy(relmc_ctx* ctx);
/* A host that brings its own transport (MPI, Julia Distributed, torch.distributed over gloo) registers it instead of an RCCL
 * communicator: fn(user, acc) leaves the sum over all ranks in *acc on every rank (0 = ok) and is called in the same order on every
 * rank.  With either kind of communicator of more than one rank in the context, relmc_nsq_run IS the multi-rank loop: every batch of
 * the global sample stream is split contiguously over the ranks, one all-reduce per batch, every rank returns the same result
 * (nsqMain.m:208-318 around the parfor of :257-263); relmc_comm_allreduce_acc goes through the registered collective as well.
 * relmc_comm_info: kind 0 = none, 1 = RCCL (ranks / rank as ncclCommCount / ncclCommUserRank report them), 2 = host collective;
 * all-reduces issued through this context (of relmc_acc, and of vectors: relmc_comm_allreduce_f64 counts once per call) and the wall time
 * spent in them.  Any output may be NULL. */
typedef int32_t (*relmc_allreduce_fn)(void* user, relmc_acc* acc_inout);
int32_t relmc_comm_set_host_allreduce(relmc_ctx* ctx, int32_t nranks, int32_t rank, relmc_allreduce_fn fn, void* user);
/* Optional vector transport of a host collective (after relmc_comm_set_host_allreduce): fn(user, buf, count) leaves the sum over all ranks of
 * `count` doubles in buf on every rank (0 = ok).  relmc_comm_allreduce_f64 -- the all-gather of the sequential loop's annual indices
 * (seqMain.m:112-133) and the per-checkpoint sums of relmc_nsq_run's stretches -- is then ONE callback per call instead of count / 130 through the
 * relmc_acc callback.  NULL removes it. */
typedef int32_t (*relmc_allreduce_f64_fn)(void* user, double* buf_inout, int64_t count);
int32_t relmc_comm_set_host_allreduce_f64(relmc_ctx* ctx, relmc_allreduce_f64_fn fn, void* user);
int32_t relmc_comm_info(const relmc_ctx* ctx, int32_t* kind_out, int32_t* nranks_out, int32_t* rank_out, int64_t* calls_out,
                        double* seconds_out);
/* Wall-clock guard.  relmc_comm_init and every collective issued through the context (RCCL or the host's callback) must finish within
 * `seconds` (default 120; <= 0 switches the guard off): a peer that never arrives cannot be waited out from inside a collective, so on
 * expiry the rank prints its rank / rank count / pid / device / PCI bus id and what it was waiting for on stderr and ends the process with
 * exit code 86 -- a multi-GPU job that would hang for the launcher's own timeout becomes a diagnosis within two minutes. */
int32_t relmc_comm_set_timeout(relmc_ctx* ctx, double seconds);
/* Sum over the ranks of `count` doubles, in place (same result on every rank).  The sequential loop all-gathers its annual indices with it
 * (every rank fills its own slots of a zeroed vector: x + 0 + ... + 0 is exact).  RCCL: one ncclAllReduce; a host collective: one call of
 * the vector transport (relmc_comm_set_host_allreduce_f64) if there is one, else the registered relmc_acc all-reduce, 130 doubles per call. */
int32_t relmc_comm_allreduce_f64(relmc_ctx* ctx, double* buf_inout, int64_t count);
/* PCI bus id of the GPU the context drives ("0000:c1:00.0", cap >= 16): what a multi-rank host gathers to show its ranks sit on DISTINCT devices */
int32_t relmc_device_pci_bus_id(const relmc_ctx* ctx, char* out, int32_t cap);

/* ---- estimators (host arithmetic, no device) ---------------------------------------- */
void relmc_acc_zero(relmc_acc* acc);
void relmc_acc_merge(relmc_acc* dst, const relmc_acc* src);
void relmc_nsq_indices(const relmc_acc* acc, int32_t nb, int32_t ncomp, double hours_per_year,
                       relmc_indices* out);

/* ---- HL1 copper-sheet non-sequential MC --------------------------- */
/* Mirrors GeneratingAdequacy/PowerSystemAdequacy.jl:169-208 run_non_sequential_mc(gens, load,
 * iterations): per iteration one fleet state (unit g is down iff draw < FOR_g, i.e. up iff
 * rand() >= for_rate, :183-185), available capacity, then the whole hourly load curve is swept:
 * hours with cap < load and their deficit (:191-197). */
typedef struct {
    int64_t n;                /* iterations                                 */
    double sum_lole;          /* sum over iterations of loss hours          */
    double sum_eue;           /* sum over iterations of unserved energy MWh */
    double sum_lole2, sum_eue2;
} relmc_hl1_acc;
int32_t relmc_hl1_load(relmc_ctx* ctx, int32_t ngen, const double* capacity_mw, const double* for_rate,
                       int32_t nhours, const double* hourly_load_mw);
/* iter_lole / iter_eue: optional host buffers [n] with the per-iteration values (history, :202-204) */
int32_t relmc_hl1_nsq(relmc_ctx* ctx, uint64_t seed, uint64_t first_index, int64_t n, relmc_hl1_acc* acc,
                      double* iter_lole_host, double* iter_eue_host);

/* ---- sequential HL2 (/root/reference/Montecarlo_seq/) ----------------- */
/* relmc_seq_load          <- seqmeantime() [MTTF MTTR] (seqmeantime.m:21-36) + the hourly load factors of
 *                            anloducurve (anloducurve.m:24-88, seqMain.m:67)
 * relmc_seq_mcsampling    <- seq_mcsampling(reliability_data, Ng, Nl, 1, hours) per year (seq_mcsampling.m:2;
 *                            every year starts all-up as seqMain.m:91): out[years][hours][ng+nl], 1 = down
 * relmc_seq_mcsimulation  <- [curt, nodal] = seq_mcsimulation(status, load_scale, ...) (seq_mcsimulation.m:1), batched
 * relmc_seq_run           <- the loop itself with its stopping rule and post-processing, seqMain.m:85-249 (below)
 * relmc_seq_years         <- the body of seqMain.m:85-176 for a range of simulated years, fused on the GPU:
 *                            chronology -> contingency hours (:97) -> DC-OPF per hour (:112-133) -> annual
 *                            indices (:160-176, calnlc.m) + the post-processing accumulators (:142-158)     */
typedef struct {
    double ens;               /* MWh, seqMain.m:173              */
    double dlc;               /* loss hours, :169                */
    double nlc;               /* loss events, :166 (calnlc)      */
    int64_t n_contingency;    /* hours evaluated, :103           */
} relmc_seq_year;
int32_t relmc_seq_load(relmc_ctx* ctx, const double* mttf, const double* mttr, int32_t hours_per_year,
                       const double* load_factors);
int32_t relmc_seq_mcsampling(relmc_ctx* ctx, uint64_t seed, uint64_t first_year, int32_t num_years,
                             uint8_t* state_host);
int32_t relmc_seq_mcsimulation(relmc_ctx* ctx, const uint8_t* states_host, const double* load_scale_host, int64_t n,
                               const relmc_solver_opts* opts, double* dns_host, double* nodal_host,
                               int32_t* status_host, int32_t* iters_host);
/* acc_out: n = LPs solved, n_fail = loss hours (curtailment > curtail_threshold, seqMain.m:41,144),
 * comp_fail = component-down counts during loss hours (:155), sum_nodal = nodal MWh over loss hours (:149) */
int32_t relmc_seq_years(relmc_ctx* ctx, uint64_t seed, uint64_t first_year, int32_t n_years,
                        const relmc_solver_opts* opts, double curtail_threshold, relmc_seq_year* years_out,
                        relmc_acc* acc_out);

/* relmc_seq_run <- the whole yearly loop of seqMain.m:85-199 (CoV stop at :194) and its post-processing :211-249, below the C ABI like
 * relmc_nsq_run.  With a communicator of more than one rank in the context (relmc_comm_init / relmc_comm_set_host_allreduce) it IS the
 * multi-rank loop: the years of every batch shard contiguously over the ranks, the annual triples are all-gathered in year order, every rank
 * stops at the same year and returns the same result.  The years depend on (seed, global year) only: results_year, the stopping year and every
 * integer of the accumulators do not depend on the number of ranks or on batch_years. */
typedef struct {
    double cov_threshold;     /* seqMain.m:40 (0.05) */
    int32_t max_years;        /* seqMain.m:39 (4000) */
    int32_t batch_years;      /* simulated years per launch round over all ranks; 0 = 64 per rank */
    uint64_t seed;
    double curtail_threshold; /* seqMain.m:41 (0.01 MW) */
    relmc_solver_opts solver;
    /* optional history buffers (may be NULL), each with room for years_cap >= max_years entries */
    int64_t years_cap;
    relmc_seq_year* results_year;   /* results_year.ens / dlc / nlc (+ contingency hours), seqMain.m:162-176 */
    double* cum_eens;               /* results_cum.eens, :180 */
    double* cum_cov;                /* results_cum.cov, :183-185 (entry 0 = 0 as in the reference; NaN = 0/0 while no year has had curtailment, as :184) */
} relmc_seq_opts;
typedef struct {
    int32_t final_year;       /* years simulated = the stopping year (seqMain.m:203) */
    int32_t converged;        /* CoV < cov_threshold reached (:194) */
    double eens;              /* MWh/yr, results_cum.eens(end)     */
    double cov;               /* results_cum.cov(end); NaN (0 / 0, as seqMain.m:184) if no simulated year had curtailment */
    double lole;              /* h/yr, mean(results_year.dlc), :212 */
    double lolf;              /* occ/yr, mean(results_year.nlc), :213 */
    double plc;               /* mean(results_year.plc)            */
    int64_t n_contingency;    /* contingency hours of the years 1..final_year (= hourly OPFs the run stands for) */
    relmc_acc acc;            /* accumulators over exactly the years 1..final_year: n = hourly OPFs, n_fail = loss hours (total_loss_hours, :158),
                                 comp_fail (:155), sum_nodal in MWh (:149) */
    double nodal_eens_avg[RELMC_MAX_BUS];      /* MWh/yr, :218 */
    double comp_importance[RELMC_MAX_COMP];    /* P(component down | loss hour), :233 */
    double wall_seconds, kernel_seconds;
} relmc_seq_result;
void relmc_seq_opts_default(relmc_seq_opts* opts);
int32_t relmc_seq_run(relmc_ctx* ctx, const relmc_seq_opts* opts, relmc_seq_result* result);

/* Units (samples, states, database rows) that the solver's static elimination order ends non-converged (status MAXIT or NUMFAIL;
 * 6.7e-7 of the RTS-96 scenarios, 4e-10 on RTS-24) are evaluated again under further static orders by every entry point, the
 * sequential ones included; the later attempt's results replace the first's.  Counters since relmc_case_load:
 * units re-evaluated, and how many of them ended converged (or MATPOWER-singular).  (The library reads ONE environment variable, RELMC_VERBOSE:
 * schedule statistics on stderr.  Diagnosis switches -- no second attempt, dense level first, one launch per batch -- exist as test hooks
 * only, relmc_debug_set in csrc/relmc_debug.hip.) */
int32_t relmc_retry_stats(const relmc_ctx* ctx, int64_t* units_out, int64_t* converged_out);
/* The kernel's list of non-converged units holds 4096 + (units of the call) / 256 entries (at most 2^20 to begin with).  relmc_nsq_accumulate / relmc_nsq_run
 * (every sample solved) evaluate a chunk again with a longer list if it overflows; the other entry points leave the units beyond the
 * list with their first-attempt results and count them here (0 on every case relmc_case_load calibrated: its primary order fails on at
 * most 0.1 % of the states). */
int32_t relmc_retry_overflow(const relmc_ctx* ctx, int64_t* units_out);
/* A unit that no static elimination order converges on is evaluated once more with every Newton step solved by Gaussian elimination
 * with PARTIAL PIVOTING on the dense reduced system (one scenario row per system, global scratch): what MATLAB's `\` does under MIPS
 * (mc_simulation.m:41).  Slow and rare (2 units in 1e9 RTS-96 samples reach it); its result takes the earlier attempts' place like theirs.
 * units_out / converged_out: how many units went there since the case was loaded and how many it converged on. */
int32_t relmc_retry_dense_stats(const relmc_ctx* ctx, int64_t* units_out, int64_t* converged_out);
/* relmc_case_load evaluates 8192 sampled states under the primary static order; a case on which more than 0.1 % of them end
 * non-converged gets the further orders probed on the same sample and the best of the three as its primary.  primary_out: 0 = the
 * default (level-then-fill), 1 = the same with the ties broken the other way, 2 = fill first; probe_failures_out: failures of each
 * probed order among the 8192 (-1 = not probed).  RTS-24 and RTS-96: {0, -1, -1}, nothing changes. */
int32_t relmc_case_order(const relmc_ctx* ctx, int32_t* primary_out, int32_t probe_failures_out[3]);
/* The primary order as a tunable of the schedule.  The Newton step of every scenario runs a static pass program that relmc_case_load
 * derives from the elimination order of the buses (MATLAB's `\` under mips picks its own pivot order per call, mc_simulation.m:41); the
 * built-in rule (shallow elimination tree first, then little fill) is good, an order searched against the scheduler itself is better:
 * RTS-24 174 -> 168 LDS instructions per Newton step (-1.4 % kernel time), RTS-96 215 -> 200 and 32 -> 28 dependent passes (-2.9 %).
 *   relmc_tune_order       host only (no device, no context): simulated annealing over bus permutations, `evaluations` runs of the scheduler
 *                          (3 ms each on RTS-96), deterministic in `seed`; start = NULL begins at the rule's order.  order_out[nb] = external
 *                          bus numbers in elimination order, the reference bus last; stats_out (optional) = {LDS instructions per Newton step,
 *                          dependent passes} of the start order and of the result.  An order never changes WHAT is computed, only the
 *                          rounding of the factorisation (iteration counts of single states may move by one).
 *                          The figures quoted above are those of the orders tuned for RTS-24 / RTS-96, which the hosts ship with their case data
 *                          (case24.RTS24_ELIM_ORDER / case96.RTS96_ELIM_ORDER, the same constants in julia/RelMC.jl); a host that passes no hint --
 *                          a plain C client -- runs the rule's order: same results to the solver's tolerances, another rounding.
 *   relmc_case_order_hint  the order for the NEXT relmc_case_load of this context (copied; NULL / 0 = the rule); the load fails with
 *                          RELMC_ERR_INVALID if it is not a permutation of 0..nb-1 with the reference bus last.  The further orders of
 *                          the retry path (relmc_retry_stats) stay rule-made. */
int32_t relmc_tune_order(const relmc_case_desc* desc, int32_t evaluations, uint64_t seed, const int32_t* start, int32_t* order_out, int32_t stats_out[4]);
int32_t relmc_case_order_hint(relmc_ctx* ctx, const int32_t* order, int32_t n);

/* ---- nsqMain (nsqMain.m:208-318 + 345-376) ------------------------------------------- */
/* Batches of up to 8192 samples (the reference's is 100, nsqMain.m:60) are evaluated many at a time in the modes
 * distinct_states = 0 and 2; every checkpoint's history entry and the stopping point are those of the batch-by-batch loop
 *  up to fp64 summation order: inside a stretch the checkpoints' EDNS / beta come from host sums of the per-sample dns
 * (the last checkpoint of a stretch, and with it the result, from the device accumulators themselves), which differ from the
 * batch-by-batch device sums in the last bits; integers (samples, loss counts, the stopping batch on any beta not within 1e-12 of its
 * limit) are identical.  A stretch that is cut at the stopping batch is evaluated again over the shorter range; its first evaluation
 * leaves no trace in relmc_retry_stats or kernel_seconds.  */
int32_t relmc_nsq_run(relmc_ctx* ctx, const relmc_nsq_opts* opts, relmc_nsq_result* result);

#ifdef __cplusplus
}
#endif
#endif /* RELMC_H */
