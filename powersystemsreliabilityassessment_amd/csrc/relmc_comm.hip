// relmc_comm.hip — multi-GPU: the one collective of the path (the reference's parfor, nsqMain.m:257-263), without any host
// framework.  RCCL is bound at run time (dlopen) so that the library has no link-time dependency on it and shares the copy a host such as
// PyTorch may already have loaded; ncclUniqueId is a 128-byte opaque struct, enums per rccl.h.  A host that brings its own transport
// registers an all-reduce callback instead.  Communicator init and every collective run under a wall-clock guard: a peer that never arrives
// ends the process with a diagnosis instead of hanging the job (relmc_comm_set_timeout).
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstddef>
#include <mutex>
#include <thread>

#include <dlfcn.h>
#include <unistd.h>

#include "relmc_ctx.h"

namespace relmc_host {

namespace {
struct RcclUid { char internal[128]; };
struct RcclApi {
    void* h = nullptr;
    int (*GetUniqueId)(RcclUid*) = nullptr;
    int (*CommInitRank)(void**, int, RcclUid, int) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int (*CommCount)(void*, int*) = nullptr;
    int (*CommUserRank)(void*, int*) = nullptr;
};
RcclApi g_rccl;
std::once_flag g_rccl_once;
const char* g_rccl_err = nullptr;
const char* rccl_load_once()
{
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        g_rccl.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (g_rccl.h) break;
    }
    if (!g_rccl.h) return "relmc_comm: librccl.so not found";
    g_rccl.GetUniqueId = reinterpret_cast<int (*)(RcclUid*)>(dlsym(g_rccl.h, "ncclGetUniqueId"));
    g_rccl.CommInitRank = reinterpret_cast<int (*)(void**, int, RcclUid, int)>(dlsym(g_rccl.h, "ncclCommInitRank"));
    g_rccl.AllReduce = reinterpret_cast<int (*)(const void*, void*, size_t, int, int, void*, hipStream_t)>(dlsym(g_rccl.h, "ncclAllReduce"));
    g_rccl.CommDestroy = reinterpret_cast<int (*)(void*)>(dlsym(g_rccl.h, "ncclCommDestroy"));
    g_rccl.GroupStart = reinterpret_cast<int (*)()>(dlsym(g_rccl.h, "ncclGroupStart"));
    g_rccl.GroupEnd = reinterpret_cast<int (*)()>(dlsym(g_rccl.h, "ncclGroupEnd"));
    g_rccl.GetErrorString = reinterpret_cast<const char* (*)(int)>(dlsym(g_rccl.h, "ncclGetErrorString"));
    g_rccl.CommCount = reinterpret_cast<int (*)(void*, int*)>(dlsym(g_rccl.h, "ncclCommCount"));
    g_rccl.CommUserRank = reinterpret_cast<int (*)(void*, int*)>(dlsym(g_rccl.h, "ncclCommUserRank"));
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce || !g_rccl.CommDestroy || !g_rccl.GroupStart || !g_rccl.GroupEnd) {
        return "relmc_comm: librccl.so lacks the expected entry points";
    }
    return nullptr;
}
// the table is filled exactly once per process, whichever context / host thread asks first (two contexts on two threads are a supported
// use of the ABI: tests/c/abi_smoke.c); a failed load stays failed with its reason
const char* rccl_load()
{
    std::call_once(g_rccl_once, []() { g_rccl_err = rccl_load_once(); });
    return g_rccl_err;
}

int rccl_fail(relmc_ctx* ctx, const char* what, int rc)
{
    return fail(ctx, RELMC_ERR_HIP, std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "RCCL error"));
}

// Wall-clock guard of one blocking step of the collective path.  A collective whose peer never arrives cannot be cancelled from inside
// (ncclCommInitRank has no communicator to abort yet; a host callback is the host's code), and a rank that hangs keeps every other rank
// and the launcher waiting: on expiry the rank says who it is, which GPU it drives and what it was waiting for, and leaves with exit
// code 86.  A fresh start is the launcher's business.  One watchdog thread per context, started at the first guarded call and parked on
// a condition variable between calls: arming and disarming it costs two mutex round trips, not a thread.
struct Watchdog {
    std::mutex m; std::condition_variable cv; std::thread th;
    bool armed = false, stop = false; unsigned long long epoch = 0;
    std::chrono::steady_clock::time_point deadline; double limit = 0.0;
    std::string what, pci; int device = -1, nranks = 1, rank = 0;
    void run()
    {
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            cv.wait(lk, [this]() { return armed || stop; });
            if (stop) return;
            const unsigned long long e = epoch;
            if (cv.wait_until(lk, deadline, [this, e]() { return stop || !armed || epoch != e; })) { if (stop) return; continue; }
            std::fprintf(stderr, "relmc_comm: rank %d of %d (pid %d, device %d, PCI %s) has waited %.0f s in %s: a peer never arrived (wrong rank count, a rank that "
                                 "died or took another path, two ranks on one GPU, or the fabric).  Leaving with exit code 86; relmc_comm_set_timeout changes the limit.\n",
                         rank, nranks, (int)getpid(), device, pci.c_str(), limit, what.c_str());
            std::fflush(stderr);
            _exit(86);
        }
    }
};
// (per context: a context is driven by one host thread at a time -- include/relmc.h -- so its watchdog needs no lock of its own)
Watchdog* watchdog_of(relmc_ctx* ctx)
{
    if (!ctx->watchdog) {
        Watchdog* w = new Watchdog();
        char pci[64] = "?";
        (void)hipDeviceGetPCIBusId(pci, (int)sizeof(pci), ctx->device);
        w->pci = pci; w->device = ctx->device;
        w->th = std::thread([w]() { w->run(); });
        ctx->watchdog = w;
    }
    return static_cast<Watchdog*>(ctx->watchdog);
}
struct Guard {
    Watchdog* w = nullptr;
    Guard(relmc_ctx* ctx, const char* what, int nranks, int rank)
    {
        const double limit = ctx->comm_timeout_s;
        if (!(limit > 0)) return;
        w = watchdog_of(ctx);
        std::lock_guard<std::mutex> lk(w->m);
        w->what = what; w->nranks = nranks; w->rank = rank; w->limit = limit;
        w->deadline = std::chrono::steady_clock::now() + std::chrono::duration_cast<std::chrono::steady_clock::duration>(std::chrono::duration<double>(limit));
        w->armed = true; w->epoch++;
        w->cv.notify_all();
    }
    ~Guard()
    {
        if (!w) return;
        std::lock_guard<std::mutex> lk(w->m);
        w->armed = false; w->epoch++;
        w->cv.notify_all();
    }
};
}  // namespace

void comm_free(relmc_ctx* ctx)
{
    if (ctx->watchdog) {
        Watchdog* w = static_cast<Watchdog*>(ctx->watchdog);
        { std::lock_guard<std::mutex> lk(w->m); w->stop = true; }
        w->cv.notify_all();
        if (w->th.joinable()) w->th.join();
        delete w;
        ctx->watchdog = nullptr;
    }
    if (ctx->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(ctx->comm);
    ctx->comm = nullptr;
    if (ctx->dgather) (void)hipFree(ctx->dgather);
    ctx->dgather = nullptr; ctx->gather_doubles = 0;
}

int comm_allreduce_f64(relmc_ctx* ctx, double* buf, int64_t count)
{
    if (count <= 0 || comm_ranks(ctx) <= 1) return RELMC_OK;
    const auto t0 = std::chrono::steady_clock::now();
    struct Tick { relmc_ctx* c; std::chrono::steady_clock::time_point t; ~Tick() { c->comm_calls++; c->comm_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count(); } } tick{ctx, t0};
    if (ctx->host_allreduce && ctx->host_allreduce_f64) {      // the host's vector transport: one callback
        Guard g(ctx, "the host's vector all-reduce callback", ctx->comm_nranks, ctx->comm_rank);
        const int32_t rc = ctx->host_allreduce_f64(ctx->host_allreduce_f64_user, buf, count);
        return rc == 0 ? RELMC_OK : fail(ctx, RELMC_ERR_HIP, "comm_allreduce_f64: the host's vector all-reduce returned " + std::to_string(rc));
    }
    if (ctx->host_allreduce) {
        // through the host's relmc_acc all-reduce, 130 doubles per call (sum_dns, sum_dns2, sum_nodal): the integers ride along as zeros
        constexpr int64_t kPer = 2 + RELMC_MAX_BUS;
        static_assert(offsetof(relmc_acc, sum_dns2) == offsetof(relmc_acc, sum_dns) + sizeof(double) &&
                      offsetof(relmc_acc, sum_nodal) == offsetof(relmc_acc, sum_dns) + 2 * sizeof(double) &&
                      sizeof(((relmc_acc*)nullptr)->sum_nodal) == sizeof(double) * RELMC_MAX_BUS,
                      "comm_allreduce_f64 carries 2 + RELMC_MAX_BUS doubles through the contiguous members sum_dns, sum_dns2, sum_nodal of relmc_acc");
        relmc_acc box;
        for (int64_t done = 0; done < count; done += kPer) {
            const int64_t m = (count - done) < kPer ? (count - done) : kPer;
            relmc_acc_zero(&box);
            std::memcpy(&box.sum_dns, buf + done, sizeof(double) * (size_t)m);
            Guard g(ctx, "the host's all-reduce callback (annual indices)", ctx->comm_nranks, ctx->comm_rank);
            const int32_t rc = ctx->host_allreduce(ctx->host_allreduce_user, &box);
            if (rc != 0) return fail(ctx, RELMC_ERR_HIP, "comm_allreduce_f64: the host's all-reduce returned " + std::to_string(rc));
            std::memcpy(buf + done, &box.sum_dns, sizeof(double) * (size_t)m);
        }
        return RELMC_OK;
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if ((size_t)count > ctx->gather_doubles) {
        if (ctx->dgather) (void)hipFree(ctx->dgather);
        ctx->dgather = nullptr; ctx->gather_doubles = 0;
        HIP_TRY(ctx, hipMalloc(&ctx->dgather, sizeof(double) * (size_t)count));
        ctx->gather_doubles = (size_t)count;
    }
    HIP_TRY(ctx, hipMemcpyAsync(ctx->dgather, buf, sizeof(double) * (size_t)count, hipMemcpyHostToDevice, ctx->stream));
    Guard g(ctx, "ncclAllReduce (annual indices)", ctx->comm_nranks, ctx->comm_rank);
    const int rc = g_rccl.AllReduce(ctx->dgather, ctx->dgather, (size_t)count, /*ncclFloat64*/ 8, /*ncclSum*/ 0, ctx->comm, ctx->stream);
    if (rc != 0) return rccl_fail(ctx, "ncclAllReduce", rc);
    HIP_TRY(ctx, hipMemcpyAsync(buf, ctx->dgather, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RELMC_OK;
}

}  // namespace relmc_host

using namespace relmc_host;

extern "C" {

int32_t relmc_comm_unique_id(uint8_t id_out[RELMC_COMM_ID_BYTES])
{
    if (!id_out) return RELMC_ERR_INVALID;
    if (rccl_load()) return RELMC_ERR_UNSUPPORTED;
    RcclUid u;
    if (g_rccl.GetUniqueId(&u) != 0) return RELMC_ERR_HIP;
    std::memcpy(id_out, u.internal, RELMC_COMM_ID_BYTES);
    return RELMC_OK;
}

int32_t relmc_comm_set_timeout(relmc_ctx* ctx, double seconds)
{
    if (!ctx) return RELMC_ERR_INVALID;
    ctx->comm_timeout_s = seconds > 0 ? seconds : 0.0;
    return RELMC_OK;
}

int32_t relmc_comm_init(relmc_ctx* ctx, int32_t nranks, int32_t rank, const uint8_t id[RELMC_COMM_ID_BYTES])
{
    if (!ctx) return RELMC_ERR_INVALID;
    if (!id || nranks < 1 || rank < 0 || rank >= nranks) return fail(ctx, RELMC_ERR_INVALID, "relmc_comm_init: bad arguments");
    if (ctx->comm || ctx->host_allreduce) return fail(ctx, RELMC_ERR_INVALID, "relmc_comm_init: the context already has a communicator");
    if (const char* e = rccl_load()) return fail(ctx, RELMC_ERR_UNSUPPORTED, e);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    RcclUid u;
    std::memcpy(u.internal, id, RELMC_COMM_ID_BYTES);
    void* comm = nullptr;
    int rc;
    {
        Guard g(ctx, "ncclCommInitRank", nranks, rank);
        rc = g_rccl.CommInitRank(&comm, nranks, u, rank);
    }
    if (rc != 0) return rccl_fail(ctx, "ncclCommInitRank", rc);
    ctx->comm = comm; ctx->comm_nranks = nranks; ctx->comm_rank = rank; ctx->comm_calls = 0; ctx->comm_seconds = 0.0;
    return RELMC_OK;
}

// The host's own collective in the place of RCCL (MPI, Julia Distributed, torch.distributed over gloo, ...): fn(user, acc) must leave the
// sum over all ranks in *acc on every rank and is called in the same order on every rank.  Everything else -- the sharding of every
// batch, the loop, the stopping rule -- is the library's (relmc_nsq_run), so a host supplies transport, not logic.
int32_t relmc_comm_set_host_allreduce(relmc_ctx* ctx, int32_t nranks, int32_t rank, relmc_allreduce_fn fn, void* user)
{
    if (!ctx) return RELMC_ERR_INVALID;
    if (!fn || nranks < 1 || rank < 0 || rank >= nranks) return fail(ctx, RELMC_ERR_INVALID, "relmc_comm_set_host_allreduce: bad arguments");
    if (ctx->comm || ctx->host_allreduce) return fail(ctx, RELMC_ERR_INVALID, "relmc_comm_set_host_allreduce: the context already has a communicator");
    ctx->host_allreduce = fn; ctx->host_allreduce_user = user; ctx->comm_nranks = nranks; ctx->comm_rank = rank; ctx->comm_calls = 0; ctx->comm_seconds = 0.0;
    return RELMC_OK;
}

int32_t relmc_comm_set_host_allreduce_f64(relmc_ctx* ctx, relmc_allreduce_f64_fn fn, void* user)
{
    if (!ctx) return RELMC_ERR_INVALID;
    if (!ctx->host_allreduce) return fail(ctx, RELMC_ERR_INVALID, "relmc_comm_set_host_allreduce_f64: register the relmc_acc collective first (relmc_comm_set_host_allreduce)");
    ctx->host_allreduce_f64 = fn; ctx->host_allreduce_f64_user = user;
    return RELMC_OK;
}

// what the communicator itself says: kind 0 none, 1 RCCL (ranks and rank from ncclCommCount / ncclCommUserRank), 2 host collective
int32_t relmc_comm_info(const relmc_ctx* ctx, int32_t* kind_out, int32_t* nranks_out, int32_t* rank_out, int64_t* calls_out, double* seconds_out)
{
    if (!ctx) return RELMC_ERR_INVALID;
    int kind = 0, n = 1, r = 0;
    if (ctx->comm) {
        kind = 1; n = ctx->comm_nranks; r = ctx->comm_rank;
        if (g_rccl.CommCount && g_rccl.CommUserRank) { int c = 0, u = 0; if (g_rccl.CommCount(ctx->comm, &c) == 0 && g_rccl.CommUserRank(ctx->comm, &u) == 0) { n = c; r = u; } }
    } else if (ctx->host_allreduce) { kind = 2; n = ctx->comm_nranks; r = ctx->comm_rank; }
    if (kind_out) *kind_out = kind;
    if (nranks_out) *nranks_out = n;
    if (rank_out) *rank_out = r;
    if (calls_out) *calls_out = ctx->comm_calls;
    if (seconds_out) *seconds_out = ctx->comm_seconds;
    return RELMC_OK;
}

// nsqMain.m:257-263's parfor gathers its slices implicitly; here: ONE grouped all-reduce(sum) over xGMI of the additive
// accumulators, int64 counters and fp64 sums each in their own type (exact integers)
int32_t relmc_comm_allreduce_acc(relmc_ctx* ctx, relmc_acc* acc)
{
    if (!ctx || !acc) return RELMC_ERR_INVALID;
    const auto t0 = std::chrono::steady_clock::now();
    struct Tick { relmc_ctx* c; std::chrono::steady_clock::time_point t; ~Tick() { c->comm_calls++; c->comm_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count(); } } tick{ctx, t0};
    if (ctx->host_allreduce) {
        Guard g(ctx, "the host's all-reduce callback (relmc_acc)", ctx->comm_nranks, ctx->comm_rank);
        const int32_t rc = ctx->host_allreduce(ctx->host_allreduce_user, acc);
        return rc == 0 ? RELMC_OK : fail(ctx, RELMC_ERR_HIP, "relmc_comm_allreduce_acc: the host's all-reduce returned " + std::to_string(rc));
    }
    if (!ctx->comm) return fail(ctx, RELMC_ERR_INVALID, "relmc_comm_allreduce_acc: relmc_comm_init has not been called");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->dacc, acc, sizeof(*acc), hipMemcpyHostToDevice, ctx->stream));
    constexpr size_t NI = 6 + RELMC_MAX_COMP + 1, ND = 2 + RELMC_MAX_BUS;
    static_assert(sizeof(relmc_acc) == 8 * (NI + ND), "relmc_acc = NI int64 then ND doubles");
    long long* di = reinterpret_cast<long long*>(ctx->dacc);
    double* dd = reinterpret_cast<double*>(di + NI);
    Guard g(ctx, "ncclAllReduce (relmc_acc)", ctx->comm_nranks, ctx->comm_rank);
    int rc = g_rccl.GroupStart();
    if (rc == 0) rc = g_rccl.AllReduce(di, di, NI, /*ncclInt64*/ 4, /*ncclSum*/ 0, ctx->comm, ctx->stream);
    if (rc == 0) rc = g_rccl.AllReduce(dd, dd, ND, /*ncclFloat64*/ 8, /*ncclSum*/ 0, ctx->comm, ctx->stream);
    const int rc2 = g_rccl.GroupEnd();
    if (rc != 0 || rc2 != 0) return rccl_fail(ctx, "ncclAllReduce", rc ? rc : rc2);
    HIP_TRY(ctx, hipMemcpyAsync(acc, ctx->dacc, sizeof(*acc), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RELMC_OK;
}

int32_t relmc_comm_allreduce_f64(relmc_ctx* ctx, double* buf_inout, int64_t count)
{
    if (!ctx || count < 0 || (count > 0 && !buf_inout)) return RELMC_ERR_INVALID;
    if (!ctx->comm && !ctx->host_allreduce) return fail(ctx, RELMC_ERR_INVALID, "relmc_comm_allreduce_f64: the context has no communicator");
    return comm_allreduce_f64(ctx, buf_inout, count);
}

int32_t relmc_comm_destroy(relmc_ctx* ctx)
{
    if (!ctx) return RELMC_ERR_INVALID;
    if (ctx->comm && g_rccl.CommDestroy) { (void)hipSetDevice(ctx->device); (void)g_rccl.CommDestroy(ctx->comm); }
    ctx->comm = nullptr; ctx->comm_nranks = 0; ctx->comm_rank = -1; ctx->host_allreduce = nullptr; ctx->host_allreduce_user = nullptr;
    ctx->host_allreduce_f64 = nullptr; ctx->host_allreduce_f64_user = nullptr;
    return RELMC_OK;
}

// PCI bus id of the GPU the context drives ("0000:c1:00.0"): what a multi-rank host gathers to show that its ranks sit on DISTINCT devices
int32_t relmc_device_pci_bus_id(const relmc_ctx* ctx, char* out, int32_t cap)
{
    if (!ctx || !out || cap < 16) return RELMC_ERR_INVALID;
    out[0] = 0;
    return hipDeviceGetPCIBusId(out, cap, ctx->device) == hipSuccess ? RELMC_OK : RELMC_ERR_HIP;
}

}  // extern "C"
