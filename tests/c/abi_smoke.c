/* Plain-C client of include/relmc.h: proves the boundary is a C ABI (no C++ types, no Python).
 * Usage: abi_smoke <case.bin>   (case arrays written by tests/test_c_abi.py)  -> prints "n n_fail sum_dns n_distinct" */
#define _POSIX_C_SOURCE 200809L
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "relmc.h"

static void* rd(FILE* f, size_t bytes) { void* p = malloc(bytes); if (fread(p, 1, bytes, f) != bytes) { fprintf(stderr, "short read\n"); exit(2); } return p; }

/* "Transport" of the two-rank test: rank 0's all-reduce adds what rank 1 would have sent.  Rank 1's slice of the CURRENT batch is
 * evaluated here, on a second context, from the same (seed, batch, rank) arithmetic relmc_nsq_run documents. */
typedef struct { relmc_ctx* other; const relmc_nsq_opts* o; int64_t done; int calls; } two_rank_t;
static int32_t two_rank_allreduce(void* user, relmc_acc* acc)
{
    two_rank_t* t = (two_rank_t*)user;
    const int64_t m = (t->o->max_samples - t->done) < t->o->batch ? (t->o->max_samples - t->done) : t->o->batch;
    const int64_t lo = t->done + m * 1 / 2, cnt = t->done + m * 2 / 2 - lo;          /* rank 1 of 2 */
    relmc_acc part;
    if (relmc_nsq_accumulate(t->other, t->o->seed, (uint64_t)lo, cnt, &t->o->solver, &part) != RELMC_OK) return 1;
    relmc_acc_merge(acc, &part);
    t->done += m; t->calls++;
    return 0;
}
static int run_two_ranks(const relmc_case_desc* d, const int32_t* order, const relmc_nsq_opts* o, const relmc_nsq_result* plain)
{
    relmc_ctx *c0 = NULL, *c1 = NULL;
    if (relmc_ctx_create(0, &c0) != RELMC_OK || relmc_ctx_create(0, &c1) != RELMC_OK) return 1;
    if (relmc_case_order_hint(c0, order, d->nb) != RELMC_OK || relmc_case_order_hint(c1, order, d->nb) != RELMC_OK) return 2;
    if (relmc_case_load(c0, d) != RELMC_OK || relmc_case_load(c1, d) != RELMC_OK) return 2;
    two_rank_t t = {c1, o, 0, 0};
    if (relmc_comm_set_host_allreduce(c0, 2, 0, two_rank_allreduce, &t) != RELMC_OK) return 3;
    relmc_nsq_result r;
    if (relmc_nsq_run(c0, o, &r) != RELMC_OK) { fprintf(stderr, "two ranks: %s\n", relmc_last_error(c0)); return 4; }
    int32_t kind = 0, nr = 0;
    if (relmc_comm_info(c0, &kind, &nr, NULL, NULL, NULL) != RELMC_OK || kind != 2 || nr != 2) return 5;
    /* same integers as the one-rank run, sums up to their order, same stopping batch; one all-reduce per batch */
    if (r.acc.n != plain->acc.n || r.acc.n_fail != plain->acc.n_fail || r.acc.sum_iters != plain->acc.sum_iters || r.checkpoints != plain->checkpoints) return 6;
    if (memcmp(r.acc.comp_fail, plain->acc.comp_fail, sizeof(r.acc.comp_fail)) != 0) return 7;
    const double rel = (r.acc.sum_dns - plain->acc.sum_dns) / plain->acc.sum_dns;
    if (rel > 1e-12 || rel < -1e-12 || t.calls != (int)r.batches || t.done != r.acc.n) return 8;
    relmc_ctx_destroy(c0); relmc_ctx_destroy(c1);
    return 0;
}

/* ---- relmc_seq_run (seqMain.m:85-249 below the ABI): alone, with the one-rank RCCL communicator, and as TWO ranks of this process --
 * two threads, two contexts, and a host collective that is a two-party rendezvous (what an MPI / Julia Distributed host would register). */
enum { PAIR_VEC = 1 << 15 };
typedef struct { pthread_mutex_t m; pthread_cond_t cv; relmc_acc sum, result[2]; int arrived; unsigned generation; int calls;
                 double vsum[PAIR_VEC], vres[2][PAIR_VEC]; int varrived; unsigned vgeneration; int vcalls; } pair_t;
/* the vector transport of the same rendezvous (relmc_comm_set_host_allreduce_f64): ONE callback per all-gather of the annual indices */
static int32_t pair_allreduce_f64(void* user, double* buf, int64_t count)
{
    pair_t* p = (pair_t*)user;
    if (count > PAIR_VEC) return 1;
    pthread_mutex_lock(&p->m);
    const unsigned gen = p->vgeneration;
    if (p->varrived == 0) memcpy(p->vsum, buf, sizeof(double) * (size_t)count); else for (int64_t k = 0; k < count; ++k) p->vsum[k] += buf[k];
    if (++p->varrived == 2) { memcpy(p->vres[gen & 1u], p->vsum, sizeof(double) * (size_t)count); p->varrived = 0; p->vgeneration++; p->vcalls++; pthread_cond_broadcast(&p->cv); }
    else while (gen == p->vgeneration) pthread_cond_wait(&p->cv, &p->m);
    memcpy(buf, p->vres[gen & 1u], sizeof(double) * (size_t)count);
    pthread_mutex_unlock(&p->m);
    return 0;
}
static int32_t pair_allreduce(void* user, relmc_acc* acc)
{
    pair_t* p = (pair_t*)user;
    pthread_mutex_lock(&p->m);
    const unsigned gen = p->generation;
    if (p->arrived == 0) p->sum = *acc; else relmc_acc_merge(&p->sum, acc);
    if (++p->arrived == 2) { p->result[gen & 1u] = p->sum; p->arrived = 0; p->generation++; p->calls++; pthread_cond_broadcast(&p->cv); }
    else while (gen == p->generation) pthread_cond_wait(&p->cv, &p->m);
    *acc = p->result[gen & 1u];
    pthread_mutex_unlock(&p->m);
    return 0;
}
typedef struct { relmc_ctx* ctx; const relmc_seq_opts* o; relmc_seq_result r; relmc_seq_year* years; int rc; } seq_rank_t;
static void* seq_rank_main(void* arg)
{
    seq_rank_t* t = (seq_rank_t*)arg;
    relmc_seq_opts o = *t->o;
    o.results_year = t->years;
    t->rc = relmc_seq_run(t->ctx, &o, &t->r);
    return NULL;
}
static int same_seq(const relmc_seq_result* a, const relmc_seq_result* b, const relmc_seq_year* ya, const relmc_seq_year* yb, int bitwise_sums)
{
    if (a->final_year != b->final_year || a->converged != b->converged || a->eens != b->eens || a->cov != b->cov || a->lole != b->lole || a->lolf != b->lolf) return 0;
    if (memcmp(ya, yb, sizeof(relmc_seq_year) * (size_t)a->final_year) != 0) return 0;                     /* annual indices bit for bit */
    if (a->acc.n != b->acc.n || a->acc.n_fail != b->acc.n_fail || a->acc.sum_iters != b->acc.sum_iters || a->n_contingency != b->n_contingency) return 0;
    if (memcmp(a->acc.comp_fail, b->acc.comp_fail, sizeof(a->acc.comp_fail)) != 0 || memcmp(a->comp_importance, b->comp_importance, sizeof(a->comp_importance)) != 0) return 0;
    for (int i = 0; i < RELMC_MAX_BUS; ++i) {
        const double x = a->nodal_eens_avg[i], y = b->nodal_eens_avg[i], d = x - y;
        if (bitwise_sums ? x != y : (d > 1e-9 * (1.0 + x) || d < -1e-9 * (1.0 + x))) return 0;
    }
    return 1;
}
/* ---- relmc_nsq_run at the reference's checkpoint spacing (nsqMain.m:60: 100 samples) as TWO ranks of this process: the multi-rank loop walks stretches of
 * checkpoints, ONE vector all-reduce per stretch through the rendezvous above, and stops at the one-rank run's checkpoint with its history.  Twice: with and
 * without the vector transport (the per-130-doubles path through the relmc_acc callback), every sample solved and behind the pre-screen. */
typedef struct { relmc_ctx* ctx; relmc_nsq_opts o; relmc_nsq_result r; double* beta; int rc; } nsq_rank_t;
static void* nsq_rank_main(void* arg)
{
    nsq_rank_t* t = (nsq_rank_t*)arg;
    t->o.beta_history = t->beta;
    t->rc = relmc_nsq_run(t->ctx, &t->o, &t->r);
    return NULL;
}
static int run_nsq_stretches(relmc_ctx* ctx, const relmc_case_desc* d, const int32_t* order)
{
    enum { CAP = 12000 };
    static double b_plain[CAP], b_r0[CAP], b_r1[CAP];
    relmc_ctx* c1 = NULL;
    if (relmc_ctx_create(0, &c1) != RELMC_OK || relmc_case_order_hint(c1, order, d->nb) != RELMC_OK || relmc_case_load(c1, d) != RELMC_OK) return 1;
    for (int variant = 0; variant < 3; ++variant) {            /* 0: vector transport; 1: relmc_acc callback only; 2: vector transport + pre-screen */
        relmc_nsq_opts o; relmc_nsq_opts_default(&o);
        o.beta_limit = 0.02; o.max_samples = 1100000; o.batch = 100; o.seed = 6; o.history_cap = CAP; o.solver.screen = variant == 2;
        relmc_nsq_result plain;
        o.beta_history = b_plain;
        if (relmc_nsq_run(ctx, &o, &plain) != RELMC_OK || !plain.converged || plain.checkpoints < 300 || plain.checkpoints >= CAP) return 2;
        static pair_t pair; memset(&pair, 0, sizeof(pair));
        pthread_mutex_init(&pair.m, NULL); pthread_cond_init(&pair.cv, NULL);
        if (relmc_comm_set_host_allreduce(ctx, 2, 0, pair_allreduce, &pair) != RELMC_OK || relmc_comm_set_host_allreduce(c1, 2, 1, pair_allreduce, &pair) != RELMC_OK) return 3;
        if (variant != 1 && (relmc_comm_set_host_allreduce_f64(ctx, pair_allreduce_f64, &pair) != RELMC_OK || relmc_comm_set_host_allreduce_f64(c1, pair_allreduce_f64, &pair) != RELMC_OK)) return 3;
        nsq_rank_t t0 = {ctx, o, plain, b_r0, -1}, t1 = {c1, o, plain, b_r1, -1};
        pthread_t th;
        if (pthread_create(&th, NULL, nsq_rank_main, &t1) != 0) return 4;
        nsq_rank_main(&t0);
        pthread_join(th, NULL);
        if (t0.rc != RELMC_OK || t1.rc != RELMC_OK) { fprintf(stderr, "nsq two ranks: %s | %s\n", relmc_last_error(ctx), relmc_last_error(c1)); return 5; }
        if (getenv("RELMC_SMOKE_VERBOSE")) fprintf(stderr, "nsq stretches variant %d: %lld checkpoints, %d vector + %d relmc_acc collectives\n", variant, (long long)plain.checkpoints, pair.vcalls, pair.calls);
        if (t0.r.checkpoints != plain.checkpoints || t1.r.checkpoints != plain.checkpoints || t0.r.acc.n != plain.acc.n || t0.r.acc.n_fail != plain.acc.n_fail ||
            t0.r.acc.n_screened != plain.acc.n_screened || memcmp(t0.r.acc.comp_fail, plain.acc.comp_fail, sizeof(plain.acc.comp_fail)) != 0 ||
            memcmp(&t0.r.acc, &t1.r.acc, sizeof(relmc_acc)) != 0) return 6;
        for (int64_t k = 0; k < plain.checkpoints; ++k) {
            const double e = (b_r0[k] - b_plain[k]) / b_plain[k];
            if (e > 1e-10 || e < -1e-10 || b_r0[k] != b_r1[k]) return 7;
        }
        /* a stretch is one collective (+ one for the cut): a handful for the whole run, not one per checkpoint */
        if (variant != 1 ? (pair.vcalls < 2 || pair.vcalls > 16 || pair.calls != 0) : (pair.vcalls != 0 || pair.calls < 2)) return 8;
        if (variant == 2 && !(plain.acc.n_screened > plain.acc.n / 2)) return 9;
        relmc_comm_destroy(ctx); relmc_comm_destroy(c1);
        pthread_mutex_destroy(&pair.m); pthread_cond_destroy(&pair.cv);
    }
    relmc_ctx_destroy(c1);
    return 0;
}
static int run_seq(relmc_ctx* ctx, const relmc_case_desc* d, const int32_t* order, int hpy, const double* mttf, const double* mttr, const double* lf)
{
    enum { MAXY = 600 };
    if (relmc_seq_load(ctx, mttf, mttr, hpy, lf) != RELMC_OK) return 1;
    relmc_seq_opts o; relmc_seq_opts_default(&o);
    if (o.cov_threshold != 0.05 || o.max_years != 4000 || o.curtail_threshold != 0.01) return 2;                /* seqMain.m:39-41 */
    o.cov_threshold = 0.12; o.max_years = MAXY; o.seed = 3; o.years_cap = MAXY;
    static relmc_seq_year y_plain[MAXY], y_comm[MAXY], y_r0[MAXY], y_r1[MAXY];
    relmc_seq_result plain, with_comm;
    o.results_year = y_plain;
    if (relmc_seq_run(ctx, &o, &plain) != RELMC_OK) { fprintf(stderr, "seq: %s\n", relmc_last_error(ctx)); return 3; }
    if (!plain.converged || plain.final_year < 20 || plain.final_year >= MAXY || !(plain.cov > 0 && plain.cov < 0.12) || plain.acc.n_fail <= 0) return 4;
    /* batch size does not matter, nor does a one-rank RCCL communicator */
    uint8_t uid[RELMC_COMM_ID_BYTES];
    if (relmc_comm_unique_id(uid) != RELMC_OK || relmc_comm_init(ctx, 1, 0, uid) != RELMC_OK) return 5;
    o.results_year = y_comm; o.batch_years = 37;
    if (relmc_seq_run(ctx, &o, &with_comm) != RELMC_OK) return 6;
    relmc_comm_destroy(ctx);
    if (!same_seq(&plain, &with_comm, y_plain, y_comm, 0)) return 7;
    /* two ranks: two threads, two contexts, a rendezvous as the host's collective; both return the single-rank result */
    relmc_ctx* c1 = NULL;
    if (relmc_ctx_create(0, &c1) != RELMC_OK || relmc_case_order_hint(c1, order, d->nb) != RELMC_OK || relmc_case_load(c1, d) != RELMC_OK ||
        relmc_seq_load(c1, mttf, mttr, hpy, lf) != RELMC_OK) return 8;
    static pair_t pair; memset(&pair, 0, sizeof(pair));
    pthread_mutex_init(&pair.m, NULL); pthread_cond_init(&pair.cv, NULL);
    if (relmc_comm_set_host_allreduce_f64(ctx, pair_allreduce_f64, &pair) == RELMC_OK) return 9;                /* refused before the relmc_acc collective is there */
    if (relmc_comm_set_host_allreduce(ctx, 2, 0, pair_allreduce, &pair) != RELMC_OK || relmc_comm_set_host_allreduce(c1, 2, 1, pair_allreduce, &pair) != RELMC_OK) return 9;
    if (relmc_comm_set_host_allreduce_f64(ctx, pair_allreduce_f64, &pair) != RELMC_OK || relmc_comm_set_host_allreduce_f64(c1, pair_allreduce_f64, &pair) != RELMC_OK) return 9;
    o.batch_years = 0;
    seq_rank_t t0 = {ctx, &o, plain, y_r0, -1}, t1 = {c1, &o, plain, y_r1, -1};
    pthread_t th;
    if (pthread_create(&th, NULL, seq_rank_main, &t1) != 0) return 10;
    seq_rank_main(&t0);
    pthread_join(th, NULL);
    if (t0.rc != RELMC_OK || t1.rc != RELMC_OK) { fprintf(stderr, "seq two ranks: %s | %s\n", relmc_last_error(ctx), relmc_last_error(c1)); return 11; }
    if (!same_seq(&t0.r, &t1.r, y_r0, y_r1, 1) || !same_seq(&plain, &t0.r, y_plain, y_r0, 0) || pair.calls < 2) return 12;
    /* the annual indices went through the vector transport: one callback per batch of years, none through the 130-double path */
    if (pair.vcalls < 1 || pair.vcalls > pair.calls) return 13;
    relmc_comm_destroy(ctx); relmc_ctx_destroy(c1);
    pthread_mutex_destroy(&pair.m); pthread_cond_destroy(&pair.cv);
    return 0;
}

int main(int argc, char** argv)
{
    if (argc < 2) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    int32_t hdr[5];                                   /* nb ng nl nd ref_bus */
    double dd[2];                                     /* base_mva total_load */
    if (fread(hdr, sizeof(int32_t), 5, f) != 5 || fread(dd, sizeof(double), 2, f) != 2) return 2;
    relmc_case_desc d;
    memset(&d, 0, sizeof(d));
    d.nb = hdr[0]; d.ng = hdr[1]; d.nl = hdr[2]; d.nd = hdr[3]; d.ref_bus = hdr[4]; d.base_mva = dd[0]; d.total_load = dd[1];
    const int ninj = d.ng + d.nd, ncomp = d.ng + d.nl;
    d.bus_pd = rd(f, 8 * d.nb); d.inj_bus = rd(f, 4 * ninj); d.inj_pmin = rd(f, 8 * ninj); d.inj_pmax = rd(f, 8 * ninj); d.inj_cost = rd(f, 8 * ninj);
    d.br_from = rd(f, 4 * d.nl); d.br_to = rd(f, 4 * d.nl); d.br_b = rd(f, 8 * d.nl); d.br_rate = rd(f, 8 * d.nl);
    d.unavail = rd(f, 8 * ncomp); d.always_up = rd(f, ncomp);
    int32_t* order = rd(f, 4 * d.nb);                 /* primary elimination order of the solver schedule (the package's tuned order) */
    int32_t* hpy = rd(f, 4);                          /* sequential track: hours per year, [MTTF MTTR] (seqmeantime.m), hourly load factors (anloducurve.m) */
    double* mttf = rd(f, 8 * ncomp); double* mttr = rd(f, 8 * ncomp); double* lf = rd(f, 8 * (size_t)hpy[0]);
    fclose(f);
    relmc_ctx* ctx = NULL;
    if (relmc_ctx_create(0, &ctx) != RELMC_OK) { fprintf(stderr, "no device\n"); return 3; }
    /* the order hint: a non-permutation is refused by the load (and consumed: the next load is rule-made again), the real one accepted */
    int32_t first = order[0]; order[0] = order[1];
    if (relmc_case_order_hint(ctx, order, d.nb) != RELMC_OK || relmc_case_load(ctx, &d) != RELMC_ERR_INVALID) return 22;
    order[0] = first;
    if (relmc_case_order_hint(ctx, order, d.nb) != RELMC_OK) return 23;
    if (relmc_case_load(ctx, &d) != RELMC_OK) { fprintf(stderr, "%s\n", relmc_last_error(ctx)); return 4; }
    relmc_solver_opts o; relmc_solver_opts_default(&o);
    relmc_acc acc, acc2; int64_t nd = 0;
    if (relmc_nsq_accumulate(ctx, 1, 0, 100000, &o, &acc) != RELMC_OK) { fprintf(stderr, "%s\n", relmc_last_error(ctx)); return 5; }
    if (relmc_nsq_accumulate_distinct(ctx, 1, 0, 100000, &o, &acc2, &nd) != RELMC_OK) return 6;
    if (acc.n != acc2.n || acc.n_fail != acc2.n_fail) return 7;
    uint8_t st[4 * 256]; double dns[4];
    if (relmc_mc_sampling(ctx, 1, 0, 4, st) != RELMC_OK || relmc_mc_simulation(ctx, st, 4, &o, dns, NULL, NULL, NULL) != RELMC_OK) return 8;
    /* the reference's persistent unique-state database (nsqMain.m:220-278) in two batches: same accumulators */
    relmc_db_stats ds; relmc_acc acc3;
    if (relmc_db_reset(ctx) != RELMC_OK || relmc_nsq_db_batch(ctx, 1, 0, 60000, &o, NULL, &ds) != RELMC_OK ||
        relmc_nsq_db_batch(ctx, 1, 60000, 40000, &o, &acc3, &ds) != RELMC_OK) { fprintf(stderr, "%s\n", relmc_last_error(ctx)); return 9; }
    if (acc3.n != acc.n || acc3.n_fail != acc.n_fail || ds.samples != 100000 || ds.rows <= 0 || ds.rows > nd) return 10;
    /* solver bookkeeping: which static elimination order runs first (calibrated at load) and how many units went to a further one */
    int32_t primary = -1, probe[3]; int64_t units = -1, conv = -1;
    if (relmc_case_order(ctx, &primary, probe) != RELMC_OK || primary != 0 || probe[0] != 0) return 14;
    if (relmc_retry_stats(ctx, &units, &conv) != RELMC_OK || units != 0 || conv != 0) return 15;
    /* the path's one collective through the library's own RCCL communicator (a single rank here: the identity) */
    uint8_t uid[RELMC_COMM_ID_BYTES]; relmc_acc red = acc;
    if (relmc_comm_unique_id(uid) != RELMC_OK || relmc_comm_init(ctx, 1, 0, uid) != RELMC_OK) { fprintf(stderr, "comm: %s\n", relmc_last_error(ctx)); return 11; }
    if (relmc_comm_allreduce_acc(ctx, &red) != RELMC_OK) { fprintf(stderr, "comm: %s\n", relmc_last_error(ctx)); return 12; }
    if (memcmp(&red, &acc, sizeof(acc)) != 0) return 13;
    /* relmc_nsq_run with that communicator in the context: the no-communicator result bit for bit (one rank = nothing to split);
     * the communicator reports its own size */
    relmc_nsq_opts no; relmc_nsq_opts_default(&no);
    no.beta_limit = 0.02; no.max_samples = 200000; no.batch = 40000; no.seed = 4;      /* above 32 768: one launch and one all-reduce per batch (smaller batches run in stretches) */
    relmc_nsq_result r_comm, r_plain;
    int32_t kind = -1, nr = -1, rk = -1; int64_t calls = -1; double secs = -1.0;
    if (relmc_nsq_run(ctx, &no, &r_comm) != RELMC_OK) { fprintf(stderr, "%s\n", relmc_last_error(ctx)); return 16; }
    if (relmc_comm_info(ctx, &kind, &nr, &rk, &calls, &secs) != RELMC_OK || kind != 1 || nr != 1 || rk != 0 || calls != 1) return 17;
    relmc_comm_destroy(ctx);
    if (relmc_comm_info(ctx, &kind, &nr, NULL, NULL, NULL) != RELMC_OK || kind != 0 || nr != 1) return 18;
    if (relmc_nsq_run(ctx, &no, &r_plain) != RELMC_OK) return 19;
    if (memcmp(&r_comm.acc, &r_plain.acc, sizeof(relmc_acc)) != 0 || r_comm.idx.beta != r_plain.idx.beta || r_comm.checkpoints != r_plain.checkpoints) return 20;
    /* the multi-rank loop itself (relmc_nsq_run with R > 1: contiguous split of every batch, one all-reduce per batch) on this one GPU:
     * a host collective for "2 ranks" whose transport is this process evaluating the OTHER rank's slice on a second context */
    if (run_two_ranks(&d, order, &no, &r_plain) != 0) return 21;
    { const int rs = run_nsq_stretches(ctx, &d, order); if (rs != 0) { fprintf(stderr, "relmc_nsq_run stretches over two ranks: check %d failed\n", rs); return 25; } }
    { const int rs = run_seq(ctx, &d, order, hpy[0], mttf, mttr, lf); if (rs != 0) { fprintf(stderr, "relmc_seq_run check %d failed\n", rs); return 24; } }
    printf("%lld %lld %.9f %lld %s\n", (long long)acc.n, (long long)acc.n_fail, acc.sum_dns, (long long)nd, relmc_version());
    relmc_ctx_destroy(ctx);
    return 0;
}
