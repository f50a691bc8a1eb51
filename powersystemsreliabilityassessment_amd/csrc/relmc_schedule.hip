// relmc_schedule.hip — symbolic analysis of a study case and the static solver schedule the evaluation kernel interprets (host arithmetic
// only: no HIP call, no context), plus the offline tuner of the primary elimination order.  What nsqMain.m:42-167 prepares once before
// its Monte Carlo loop; MATLAB's `\` under MIPS picks its own pivot order per call (mc_simulation.m:41), here the order is fixed per case.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>

#include "relmc_ctx.h"
#ifdef RELMC_DEV_SWITCHES
#include "relmc_dev_switches.h"
#endif

namespace relmc_host {

bool verbose()
{
    static const bool v = std::getenv("RELMC_VERBOSE") != nullptr;      // the library's only environment variable in the default build
    return v;
}

SymOpts sym_opts_default()
{
    SymOpts so;
#ifdef RELMC_DEV_SWITCHES      // ablation builds (csrc/Makefile: ablate/librelmc_dev.so): schedule forms and search weights from the environment
    relmc_dev_switches_schedule(so);
#endif
    return so;
}

// Build the device tables from the plain case description: internal bus numbering = elimination
// order of the sparse block LDL' (level-then-min-fill, reference bus last), symbolic fill, the
// static task schedule the kernel interprets, incidence lists, thresholds.
// Pure host arithmetic: relmc_debug_symbolic runs it without a device, which is how the CPU test suite checks every schedule it
// produces by interpreting it against a dense solve (tests/test_schedule.py).
template <class TL>
int case_symbolic(const relmc_case_desc* d, DevCaseT<TL>& C, int order_variant, const SymOpts& so, SymGeom& geom, std::string& err)
{
    auto failx = [&](int code, const char* msg) { err = msg; return code; };
    constexpr int NBT = TL::NBT, NLT = TL::NLT, NIT = TL::NIT, NCOMPMAX = TL::NCOMPMAX, MAXOFF = TL::MAXOFF, MAXPASS = TL::MAXPASS,
                  ROWL = TL::RW, IS = TL::IS, WPB = TL::WPB, SPW = TL::SPW, OW = TL::OW;
    const int nb = d->nb, ng = d->ng, nl = d->nl, nd = d->nd, ninj = ng + nd, ncomp = ng + nl;
    if (nb > NBT || nl > NLT || ninj > NIT || ncomp > NCOMPMAX || nl > 126 || ninj > 254)
        return failx(RELMC_ERR_UNSUPPORTED, "relmc_case_load: case exceeds the compiled tiles (128 buses, 126 lines, 192 injections, 256 components)");
    std::memset(&C, 0, sizeof(C));
    C.nb = nb; C.ng = ng; C.nl = nl; C.nd = nd; C.ninj = ninj; C.ncomp = ncomp;
    C.base_mva = d->base_mva; C.total_load = d->total_load;
    C.exist_mask = nb >= 32 ? 0xffffffffu : ((1u << nb) - 1u);      // used by the 16-lane tile only (nb <= 32 there)

    // ---- elimination order on the bus graph (all lines in service = superset of every outage state)
    std::vector<std::vector<char>> A(nb, std::vector<char>(nb, 0));
    for (int l = 0; l < nl; ++l) {
        const int f = d->br_from[l], t = d->br_to[l];
        if (f < 0 || f >= nb || t < 0 || t >= nb || f == t || !(d->br_b[l] == d->br_b[l]))
            return failx(RELMC_ERR_INVALID, "relmc_case_load: bad branch end points");
        A[f][t] = A[t][f] = 1;
    }
    std::vector<int> ext2int(nb, -1), level(nb, -1);
    std::vector<char> gone(nb, 0);
    std::vector<std::vector<int>> hi_ext(nb);          // higher neighbours (external ids) at elimination time
    // The primary order may come from the host (relmc_case_order_hint: an order tuned offline by relmc_tune_order against this very
    // scheduler); the further orders of the retry path stay rule-made.
    std::vector<int> forced;
    if (order_variant == 0 && so.order_hint && so.n_hint > 0) forced.assign(so.order_hint, so.order_hint + so.n_hint);
    if (!forced.empty()) {
        std::vector<char> seen(nb, 0);
        bool ok = (int)forced.size() == nb && forced.back() == d->ref_bus;
        for (int v : forced) { if (v < 0 || v >= nb || seen[v]) ok = false; else seen[v] = 1; }
        if (!ok) return failx(RELMC_ERR_INVALID, "relmc_case_load: the elimination-order hint is not a permutation of the buses with the reference bus last");
    }
    for (int step = 0; step < nb; ++step) {
        int best = -1; long bestkey = 0;
        if (!forced.empty()) best = forced[step];
        else
        for (int b = 0; b < nb; ++b) {
            if (gone[b] || (b == d->ref_bus && step < nb - 1)) continue;
            int deg = 0, fillc = 0, lev = 0;
            for (int x = 0; x < nb; ++x) {
                if (x == b || !A[b][x]) continue;
                if (gone[x]) { if (level[x] + 1 > lev) lev = level[x] + 1; continue; }
                deg++;
                for (int y = x + 1; y < nb; ++y) if (!gone[y] && y != b && A[b][y] && !A[x][y]) fillc++;
            }
            // shallow elimination tree first (fewer dependent passes), then little fill
            long key = ((long)lev * 1000 + fillc) * 10000 + deg * 100 + b;
            if (order_variant == 1) key = ((long)lev * 1000 + fillc) * 10000 + deg * 100 + (nb - 1 - b);       // other tie-breaks
            else if (order_variant == 2) key = ((long)fillc * 1000 + lev) * 10000 + deg * 100 + b;            // fill first

            if (best < 0 || key < bestkey) { best = b; bestkey = key; }
        }
        int lev = 0;
        for (int x = 0; x < nb; ++x) if (x != best && A[best][x] && gone[x] && level[x] + 1 > lev) lev = level[x] + 1;
        level[best] = lev;
        for (int x = 0; x < nb; ++x) if (!gone[x] && x != best && A[best][x]) {
            hi_ext[best].push_back(x);
            for (int y = 0; y < nb; ++y) if (!gone[y] && y != best && y != x && A[best][y]) A[x][y] = A[y][x] = 1;
        }
        gone[best] = 1;
        ext2int[best] = step;
    }
    for (int i = 0; i < NBT; ++i) { C.b_ext[i] = 0xff; C.b_int[i] = 0xff; C.b_vinj[i] = -1; }
    for (int e = 0; e < nb; ++e) { C.b_ext[ext2int[e]] = (uint8_t)e; C.b_int[e] = (uint8_t)ext2int[e]; }
    C.ref_bus = ext2int[d->ref_bus];                    // == nb - 1
    std::vector<std::vector<int>> N(nb);                // higher neighbours, internal ids, ascending
    for (int e = 0; e < nb; ++e) {
        for (int x : hi_ext[e]) N[ext2int[e]].push_back(ext2int[x]);
        std::sort(N[ext2int[e]].begin(), N[ext2int[e]].end());
    }

    // ---- block storage: diagonal blocks, off-diagonal blocks (a, i) a > i, rhs blocks, P blocks
    std::vector<std::vector<int>> blk(nb, std::vector<int>(nb, -1));
    int noff = 0;
    for (int i = 0; i < nb; ++i) for (int a : N[i]) blk[a][i] = nb + noff++;
    if (noff > MAXOFF) return failx(RELMC_ERR_UNSUPPORTED, "relmc_case_load: too much fill for the solver workspace");
    C.noff = noff;
    C.off_rhs = (uint16_t)(4 * (nb + noff));          // rhs / solution: 2 doubles per bus
    C.off_p = 0;                                        // P = inv(D) overwrites D in place
    C.nws = (uint32_t)C.off_rhs + 2u * nb;
    if (C.nws >= 0x1000u) return failx(RELMC_ERR_UNSUPPORTED, "relmc_case_load: solver workspace too large");      // byte offsets of the pass descriptors keep bit 15 free
    auto OFFD = [&](int i) { return 4 * i; };
    std::vector<int> pos(nb + noff);                    // block id -> position in W (diagonal blocks stay at their bus index)
    for (int k = 0; k < nb + noff; ++k) pos[k] = k;
    auto OFFB = [&](int a, int i) { return 4 * pos[blk[a][i]]; };
    auto OFFY = [&](int i) { return (int)C.off_rhs + 2 * i; };
    auto OFFP = [&](int i) { return 4 * i; };

    // ---- task list in sequential (right-looking) order, then list scheduling into passes of 16
    struct Task { uint8_t kind; uint16_t o[4]; std::vector<int> rd, wr; };
    std::vector<Task> tasks;
    auto unit = [&](int off) { return off < (int)C.off_rhs ? off >> 2 : (int)C.off_rhs / 4 + ((off - (int)C.off_rhs) >> 1); };   // one block / one rhs pair
    // model only (SymOpts::model_leaf_free, with RELMC_VERBOSE): what would the update phase look like if the pivots that are complete at
    // assembly (no earlier-eliminated neighbour) were eliminated by the assembling lanes?  The printed schedule is NOT a valid program.
    std::vector<char> leaf_free(nb, 0);
    if (so.model_leaf_free >= 0) {
        const int maxdeg_free = so.model_leaf_free;
        std::vector<char> has_lower(nb, 0);
        for (int i = 0; i < nb; ++i) for (int a2 : N[i]) has_lower[a2] = 1;
        int nfree = 0, ntask_free = 0;
        for (int i = 0; i < nb; ++i) if (!has_lower[i] && (int)N[i].size() <= maxdeg_free) { leaf_free[i] = 1; nfree++; ntask_free += (int)(N[i].size() * (N[i].size() + 1) / 2 + N[i].size()); }
        fprintf(stderr, "relmc: model: %d pivots complete at assembly with <= %d higher neighbours, %d update tasks\n", nfree, maxdeg_free, ntask_free);
    }
    for (int i = 0; i < nb; ++i) {
        if (leaf_free[i]) continue;
        for (size_t ia = 0; ia < N[i].size(); ++ia)
            for (size_t ib = 0; ib <= ia; ++ib) {
                const int a2 = N[i][ia], b2 = N[i][ib];
                const int T = a2 == b2 ? OFFD(a2) : OFFB(a2, b2);
                if (a2 != b2 && blk[a2][b2] < 0) return failx(RELMC_ERR_INVALID, "relmc_case_load: symbolic factorisation inconsistent");
                Task t; t.kind = 0; t.o[0] = (uint16_t)T; t.o[1] = (uint16_t)OFFB(a2, i); t.o[2] = (uint16_t)OFFB(b2, i); t.o[3] = (uint16_t)OFFD(i);
                t.rd = {unit(OFFB(a2, i)), unit(OFFB(b2, i)), unit(OFFD(i)), unit(T)}; t.wr = {unit(T)};
                tasks.push_back(t);
            }
        for (int a2 : N[i]) {                             // right-hand side as a pseudo-bus: y_a' -= y_i' P W_a'
            Task t; t.kind = 0; t.o[0] = (uint16_t)(OFFY(a2) | 0x8000); t.o[1] = (uint16_t)OFFY(i); t.o[2] = (uint16_t)OFFB(a2, i); t.o[3] = (uint16_t)OFFD(i);
            t.rd = {unit(OFFY(i)), unit(OFFB(a2, i)), unit(OFFD(i)), unit(OFFY(a2))}; t.wr = {unit(OFFY(a2))};
            tasks.push_back(t);
        }
    }
    for (int i = 0; i < nb; ++i) {
        Task t; t.kind = 1; t.o[0] = (uint16_t)OFFD(i); t.o[1] = (uint16_t)OFFY(i); t.o[2] = 0; t.o[3] = 0;
        t.rd = {unit(OFFD(i)), unit(OFFY(i))}; t.wr = {unit(OFFD(i)), unit(OFFY(i))};
        tasks.push_back(t);
    }
    for (int a2 = nb - 1; a2 >= 0; --a2)
        for (int i = 0; i < a2; ++i) {
            if (blk[a2][i] < 0) continue;
            Task t; t.kind = 2; t.o[0] = (uint16_t)OFFY(i); t.o[1] = (uint16_t)OFFB(a2, i); t.o[2] = (uint16_t)OFFP(i); t.o[3] = (uint16_t)OFFY(a2);
            t.rd = {unit(OFFP(i)), unit(OFFB(a2, i)), unit(OFFY(a2)), unit(OFFY(i))}; t.wr = {unit(OFFY(i))};
            tasks.push_back(t);
        }
    {
        const int nunits = (int)C.nws / 2 + 2;
        std::vector<int> lastw(nunits, -1), lastr(nunits, -1), pkind, pcount;
        std::vector<std::vector<int>> pass_tasks;          // pass -> RW slots, task index or -1
        const auto t_sched0 = std::chrono::steady_clock::now();
        for (int q = 0; q < MAXPASS; ++q) for (int r = 0; r < ROWL; ++r) for (int k = 0; k < 4; ++k) C.task[q][r][k] = 0xffff;   // null task
        // List scheduling by longest remaining dependency path (round 2; the round-1 scheduler placed the tasks as soon as
        // possible in generation order and needed one more update pass on both test systems).  Dependencies in the sequential
        // order of `tasks`: read-after-write and write-after-write need a LATER pass, write-after-read allows the SAME pass
        // (all lanes load before any lane stores).  Phases stay contiguous: all PK_UPD passes, then PK_INV, then PK_BWD.
        {
            const int nt = (int)tasks.size();
            std::vector<std::vector<int>> strict(nt), weak(nt), succ(nt);
            std::vector<char> succ_w;                                    // parallel to the flattened succ lists: 1 = strict edge
            std::vector<std::vector<char>> succw(nt);
            {
                std::vector<int> lw(nunits, -1);
                std::vector<std::vector<int>> readers(nunits);
                for (int i = 0; i < nt; ++i) {
                    const Task& t = tasks[i];
                    for (int r : t.rd) if (lw[r] >= 0) strict[i].push_back(lw[r]);
                    for (int w : t.wr) {
                        if (lw[w] >= 0) strict[i].push_back(lw[w]);
                        for (int q : readers[w]) if (q != i) weak[i].push_back(q);
                    }
                    for (int r : t.rd) readers[r].push_back(i);
                    for (int w : t.wr) { lw[w] = i; readers[w].clear(); }
                }
                for (int i = 0; i < nt; ++i) {
                    for (int q : strict[i]) { succ[q].push_back(i); succw[q].push_back(1); }
                    for (int q : weak[i]) { succ[q].push_back(i); succw[q].push_back(0); }
                }
            }
            std::vector<int> depth(nt, 0), passof(nt, -1);
            for (int i = nt - 1; i >= 0; --i)
                for (size_t k = 0; k < succ[i].size(); ++k) {
                    const int j = succ[i][k];
                    if (tasks[j].kind == tasks[i].kind && depth[j] + succw[i][k] > depth[i]) depth[i] = depth[j] + succw[i][k];
                }
            if (verbose()) {         // longest dependency chain per phase = the fewest passes any packing could reach
                int md[3] = {0, 0, 0};
                for (int i = 0; i < nt; ++i) if (depth[i] + 1 > md[tasks[i].kind]) md[tasks[i].kind] = depth[i] + 1;
                fprintf(stderr, "relmc: order %d: critical path (passes) update %d, inversion %d, back substitution %d\n", order_variant, md[0], md[1], md[2]);
            }
            // A pass costs its LDS instructions whatever its fill, and an update pass filled to at most a half / a quarter runs in the
            // cheaper half / quarter form (relmc_dev.h): 10 / 7 / 6 instructions.  So the update phase is scheduled twice or more: with
            // the full row width throughout, and with only half of it from pass F on; the cheapest variant that needs no extra pass wins.
            auto upd_cost = [&](size_t first_pass) {
                long c = 0;
                for (size_t q = first_pass; q < pcount.size(); ++q) c += pcount[q] > ROWL / 2 ? 10 : (pcount[q] > ROWL / 4 ? 7 : 6);
                return c;
            };
            int best_f = 1 << 30;                                       // pass index from which the narrow capacity applies (none)
            {
                long best_cost = -1; size_t best_n = 0;
                for (int trial = -1; trial < MAXPASS; ++trial) {
                    const int f_try = trial < 0 ? (1 << 30) : trial;
                    std::vector<int> remaining;
                    for (int i = 0; i < nt; ++i) if (tasks[i].kind == 0) remaining.push_back(i);
                    std::vector<int> po(passof);
                    std::vector<int> cnt;
                    bool fits = true;
                    while (!remaining.empty()) {
                        const int cur = (int)cnt.size();
                        if (cur >= MAXPASS - 1) { fits = false; break; }
                        const int cap = cur >= f_try ? ROWL / 2 : ROWL;
                        std::vector<int> ready;
                        for (int i : remaining) {
                            bool ok = true;
                            for (int q : strict[i]) if (po[q] < 0 || po[q] >= cur) { ok = false; break; }
                            if (ok) ready.push_back(i);
                        }
                        std::stable_sort(ready.begin(), ready.end(), [&](int a2, int b2) { return depth[a2] != depth[b2] ? depth[a2] > depth[b2] : a2 < b2; });
                        std::vector<char> in_pass(nt, 0); int n_in = 0;
                        std::vector<int> chosen;
                        for (int i : ready) {
                            if (n_in >= cap) break;
                            bool ok = true;
                            for (int q : weak[i]) if (po[q] < 0 && !in_pass[q]) { ok = false; break; }
                            if (ok) { chosen.push_back(i); in_pass[i] = 1; n_in++; }
                        }
                        if (chosen.empty()) { fits = false; break; }
                        for (int i : chosen) po[i] = cur;
                        cnt.push_back(n_in);
                        std::vector<int> rest;
                        for (int i : remaining) if (po[i] < 0) rest.push_back(i);
                        remaining.swap(rest);
                    }
                    if (!fits) { if (trial < 0) break; else continue; }
                    pcount = cnt;
                    const long c = upd_cost(0);
                    pcount.clear();
                    if (trial < 0) { best_cost = c; best_n = cnt.size(); best_f = f_try; }
                    else if (cnt.size() <= best_n && c < best_cost) { best_cost = c; best_f = f_try; }
                    if (trial >= 0 && (size_t)trial >= best_n) break;
                }
                if (so.no_quarter) best_f = 1 << 30;
            }
            for (int kind = 0; kind < 3; ++kind) {
                std::vector<int> remaining;
                for (int i = 0; i < nt; ++i) if (tasks[i].kind == kind) remaining.push_back(i);
                while (!remaining.empty()) {
                    const int cur = (int)pkind.size();
                    if (cur >= MAXPASS - 1) return failx(RELMC_ERR_UNSUPPORTED, "relmc_case_load: solver schedule exceeds MAXPASS");
                    std::vector<int> ready;
                    for (int i : remaining) {
                        bool ok = true;
                        for (int q : strict[i]) if (passof[q] < 0 || passof[q] >= cur) { ok = false; break; }
                        if (ok) ready.push_back(i);
                    }
                    std::stable_sort(ready.begin(), ready.end(), [&](int a2, int b2) { return depth[a2] != depth[b2] ? depth[a2] > depth[b2] : a2 < b2; });
                    std::vector<int> chosen;
                    std::vector<char> in_pass(nt, 0);
                    const int cap = (kind == 0 && cur >= best_f) ? ROWL / 2 : ROWL;
                    for (int i : ready) {
                        if ((int)chosen.size() >= cap) break;
                        bool ok = true;                                   // readers of what this task overwrites: already placed, or in this pass
                        for (int q : weak[i]) if (passof[q] < 0 && !in_pass[q]) { ok = false; break; }
                        if (ok) { chosen.push_back(i); in_pass[i] = 1; }
                    }
                    if (chosen.empty()) return failx(RELMC_ERR_INVALID, "relmc_case_load: solver schedule has a dependency cycle");
                    pkind.push_back(kind); pcount.push_back((int)chosen.size()); pass_tasks.push_back(std::vector<int>(ROWL, -1));
                    for (size_t k = 0; k < chosen.size(); ++k) { pass_tasks[cur][k] = chosen[k]; passof[chosen[k]] = cur; }
                    std::vector<int> rest;
                    for (int i : remaining) if (passof[i] < 0) rest.push_back(i);
                    remaining.swap(rest);
                }
            }
        }
        (void)lastw; (void)lastr;
        const auto t_place0 = std::chrono::steady_clock::now();
        if (verbose()) fprintf(stderr, "relmc: scheduling %.1f ms\n", std::chrono::duration<double, std::milli>(t_place0 - t_sched0).count());
        // ---- LDS bank-conflict aware placement (host only; the passes and their dependencies are untouched).
        // ds_read_b128 serves a wavefront in four fixed 16-lane groups (MI355X_MICROARCH.md), bank = (byte address / 4) mod 64:
        // a group is conflict-free when its 16 lanes hit 16 different 16-byte bank slots.  Two degrees of freedom cost
        // nothing at run time: where the off-diagonal blocks live in W and which lane of the row carries which task of a
        // pass.  A seeded local search minimises the modelled extra LDS cycles of all operand reads of one Newton step.
        {
            const uint32_t eval_d = 4u * (nl + 1) + 4u * (ninj + 1);
            uint32_t stride = C.nws > eval_d ? C.nws : eval_d;
            stride = (stride + 1u) & ~1u;
            stride += 2u * IS * ROWL + 2u + NBT + OW / 2u;
            while ((stride & 3u) != 2u) stride += 1;            // = the per-scenario LDS stride computed below
            stride += 4u * (uint32_t)so.scen_pad4;
            static const int kGroupOfLane[64] = {0,0,0,0,1,1,1,1,1,1,1,1,0,0,0,0, 1,1,1,1,0,0,0,0,0,0,0,0,1,1,1,1,
                                                 2,2,2,2,3,3,3,3,3,3,3,3,2,2,2,2, 3,3,3,3,2,2,2,2,2,2,2,2,3,3,3,3};
            auto remap = [&](int o) {                             // offset under the identity placement -> current placement
                const int f = o & 0x8000; o &= 0x7fff;
                if (o >= 4 * nb && o < (int)C.off_rhs) o = 4 * pos[o >> 2] + (o & 3);
                return o | f;
            };
            // operand reads of a task: (offset index into t.o, +2 doubles?) per kind; rhs-row tasks skip the second halves of T and Wa
            // operands that are written back (T; D and y of an inversion; y_i of a back substitution) hit the same banks a second time with the
            // slower store instruction: their conflicts can be given more weight (SymOpts::place_ww, default 1 = reads only, as measured so far)
            const long ww = so.place_ww;
            auto pass_cost = [&](int p) {
                long cost = 0;
                const int kind = pkind[p];
                const int nins = kind == 0 ? 8 : (kind == 1 ? 3 : 6);
                for (int ins = 0; ins < nins; ++ins) {
                    int cnt[4][16]; int addr[4][16][16];
                    for (int g = 0; g < 4; ++g) for (int q = 0; q < 16; ++q) cnt[g][q] = 0;
                    for (int lane = 0; lane < 64; ++lane) {
                        const int slot = lane % ROWL, row = lane / ROWL;
                        const int ti = pass_tasks[p][slot];
                        if (ti < 0) continue;
                        const Task& t = tasks[ti];
                        int o = -1;
                        if (kind == 0) {
                            const bool vec = (t.o[0] & 0x8000) != 0;
                            switch (ins) {
                                case 0: o = remap(t.o[3]); break;            case 1: o = remap(t.o[3]) + 2; break;         // D
                                case 2: o = remap(t.o[1]); break;            case 3: o = remap(t.o[2]); break;             // Wa0, Wb0
                                case 4: o = remap(t.o[2]) + 2; break;        case 5: o = remap(t.o[0]) & 0x7fff; break;    // Wb1, T0
                                case 6: if (!vec) o = remap(t.o[1]) + 2; break;
                                default: if (!vec) o = (remap(t.o[0]) & 0x7fff) + 2; break;
                            }
                        } else if (kind == 1) {
                            o = ins == 0 ? remap(t.o[0]) : (ins == 1 ? remap(t.o[0]) + 2 : remap(t.o[1]));
                        } else {
                            switch (ins) {
                                case 0: o = remap(t.o[1]); break; case 1: o = remap(t.o[1]) + 2; break;
                                case 2: o = remap(t.o[2]); break; case 3: o = remap(t.o[2]) + 2; break;
                                case 4: o = remap(t.o[3]); break; default: o = remap(t.o[0]); break;
                            }
                        }
                        if (o < 0) continue;
                        const int ad = row * (int)stride + o, g = kGroupOfLane[lane], q = (ad >> 1) & 15;
                        bool seen = false;
                        for (int k = 0; k < cnt[g][q]; ++k) if (addr[g][q][k] == ad) { seen = true; break; }
                        if (!seen) addr[g][q][cnt[g][q]++] = ad;
                    }
                    const bool written = kind == 0 ? (ins == 5 || ins == 7) : (kind == 1 ? true : ins == 5);
                    for (int g = 0; g < 4; ++g) { int mx = 0; for (int q = 0; q < 16; ++q) if (cnt[g][q] > mx) mx = cnt[g][q]; if (mx > 1) cost += (mx - 1) * (written ? ww : 1); }
                }
                return cost;
            };
            const int np = (int)pkind.size();
            std::vector<long> pc(np);
            long total = 0;
            for (int p = 0; p < np; ++p) { pc[p] = pass_cost(p); total += pc[p]; }
            const long before = total;
            uint64_t rng = 0x9E3779B97F4A7C15ull;
            auto rnd = [&](int m) { rng = rng * 6364136223846793005ull + 1442695040888963407ull; return (int)((rng >> 33) % (uint64_t)m); };
            std::vector<std::vector<int>> passes_of_block(nb + noff);      // passes whose operands include the block
            for (int p = 0; p < np; ++p)
                for (int r = 0; r < ROWL; ++r) {
                    const int ti = pass_tasks[p][r];
                    if (ti < 0) continue;
                    for (int k = 0; k < 4; ++k) {
                        const int o = tasks[ti].o[k] & 0x7fff;
                        if (o >= 4 * nb && o < (int)C.off_rhs && !(tasks[ti].kind == 1 && k >= 2)) {
                            std::vector<int>& v = passes_of_block[o >> 2];
                            if (std::find(v.begin(), v.end(), p) == v.end()) v.push_back(p);
                        }
                    }
                }
            std::vector<int> touched; std::vector<long> newc;
            // 100 moves per block: 36 / 114 ms of relmc_case_load on RTS-24 / RTS-96; 400 (round 1) took 144 / 440 ms for kernel times within
            // run-to-run noise of these (19.0 vs 19.2 ms, 78.4 vs 78.2 ms per 1e6); no search at all: 19.2 / 79.3 ms
            const int moves = noff > 0 ? so.place_moves * (nb + noff) : 0;
            for (int it = 0; it < moves && total > 0; ++it) {
                if (rnd(10) < 6) {                                  // swap the positions of two off-diagonal blocks
                    const int i = nb + rnd(noff), j = nb + rnd(noff);
                    if (i == j) continue;
                    std::swap(pos[i], pos[j]);
                    touched.clear();
                    for (int p : passes_of_block[i]) touched.push_back(p);
                    for (int p : passes_of_block[j]) if (std::find(touched.begin(), touched.end(), p) == touched.end()) touched.push_back(p);
                    long delta = 0; newc.resize(touched.size());
                    for (size_t k = 0; k < touched.size(); ++k) { newc[k] = pass_cost(touched[k]); delta += newc[k] - pc[touched[k]]; }
                    if (delta <= 0) { total += delta; for (size_t k = 0; k < touched.size(); ++k) pc[touched[k]] = newc[k]; }
                    else std::swap(pos[i], pos[j]);
                } else {                                            // swap two lanes (tasks or holes) of one pass
                    const int p = rnd(np), i = rnd(ROWL), j = rnd(ROWL);
                    if (i == j) continue;
                    std::swap(pass_tasks[p][i], pass_tasks[p][j]);
                    const long c2 = pass_cost(p);
                    if (c2 <= pc[p]) { total += c2 - pc[p]; pc[p] = c2; } else std::swap(pass_tasks[p][i], pass_tasks[p][j]);
                }
            }
            geom.conflict_before = before; geom.conflict_after = total;
            if (verbose()) fprintf(stderr, "relmc: placement search %.1f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_place0).count());
            for (int p = 0; p < np; ++p)
                for (int r = 0; r < ROWL; ++r) {
                    const int ti = pass_tasks[p][r];
                    if (ti < 0) continue;
                    for (int k = 0; k < 4; ++k) C.task[p][r][k] = (uint16_t)remap(tasks[ti].o[k]);
                    if (tasks[ti].kind == 1) { C.task[p][r][2] = 0; C.task[p][r][3] = 0; }
                }
        }
        // the sparsely filled PK_UPD passes at the end of the phase in quarter form (relmc_dev.h)
        {
            int nup = 0;
            for (size_t q = 0; q < pkind.size(); ++q) if (pkind[q] == 0) nup++;
            int nq = 0;
            while (nq < nup && pcount[nup - 1 - nq] <= ROWL / 4 && !so.no_quarter) nq++;
            C.npass_updq = (uint16_t)nq;
            int nh = 0;
            while (nq + nh < nup && pcount[nup - 1 - nq - nh] <= ROWL / 2 && !so.no_half && !so.no_quarter) nh++;
            C.npass_updh = (uint16_t)nh;
            for (int p = nup - nq - nh; p < nup - nq; ++p) {
                uint16_t full[ROWL][4]; int nfull = 0;
                for (int r = 0; r < ROWL; ++r) if (C.task[p][r][0] != 0xffff) { for (int k = 0; k < 4; ++k) full[nfull][k] = C.task[p][r][k]; nfull++; }
                for (int r = 0; r < ROWL; ++r) for (int k = 0; k < 4; ++k) C.task[p][r][k] = 0xffff;
                for (int t = 0; t < nfull; ++t) {
                    const bool vec = (full[t][0] & 0x8000u) != 0;
                    const int T = full[t][0] & 0x7fff, Wa = full[t][1], Wb = full[t][2], D = full[t][3];
                    for (int r = 0; r < (vec ? 1 : 2); ++r) {
                        uint16_t* q = C.task[p][2 * t + r];
                        q[0] = (uint16_t)(T + 2 * r); q[1] = (uint16_t)(Wa + 2 * r); q[2] = (uint16_t)Wb; q[3] = (uint16_t)D;
                    }
                }
            }
            for (int p = nup - nq; p < nup; ++p) {
                uint16_t full[ROWL][4]; int nfull = 0;
                for (int r = 0; r < ROWL; ++r) if (C.task[p][r][0] != 0xffff) { for (int k = 0; k < 4; ++k) full[nfull][k] = C.task[p][r][k]; nfull++; }
                for (int r = 0; r < ROWL; ++r) for (int k = 0; k < 4; ++k) C.task[p][r][k] = 0xffff;
                for (int t = 0; t < nfull; ++t) {
                    const bool vec = (full[t][0] & 0x8000u) != 0;
                    const int T = full[t][0] & 0x7fff, Wa = full[t][1], Wb = full[t][2], D = full[t][3];
                    for (int r = 0; r < (vec ? 1 : 2); ++r) for (int c = 0; c < 2; ++c) {
                        uint16_t* q = C.task[p][4 * t + 2 * r + c];
                        q[0] = (uint16_t)(T + 2 * r + c); q[1] = (uint16_t)(Wa + 2 * r); q[2] = (uint16_t)(Wb + 2 * c); q[3] = (uint16_t)D;
                    }
                }
            }
        }
        // back-substitution passes filled to at most a half in half form (relmc_dev.h): one LDS instruction less per pass
        C.bwd_half = 0;
        if (!so.no_bwd_half) {
            int first_bwd = 0;
            while (first_bwd < (int)pkind.size() && pkind[first_bwd] != 2) first_bwd++;
            // all or nothing: a loop that switches form per pass costs more than the half form saves (measured: +1.6 % / +2.5 % against the
            // full form alone, profiles/r3_pf/c26_notes.txt), so the half form is taken when EVERY pass qualifies (RTS-96: 11 of 11; RTS-24: 5 of 7, stays full)
            // The 16-lane tile keeps the full form: its kernel is 0.35 % slower with the second loop compiled in, whatever runs.
            bool all_half = ROWL == 64 && (int)pkind.size() - first_bwd <= 64 && first_bwd < (int)pkind.size();
            for (int p = first_bwd; p < (int)pkind.size(); ++p) if (pcount[p] > ROWL / 2) all_half = false;
            for (int p = first_bwd; all_half && p < (int)pkind.size(); ++p) {
                uint16_t full[ROWL][4]; int nfull = 0;
                for (int r = 0; r < ROWL; ++r) if (C.task[p][r][0] != 0xffff) { for (int k = 0; k < 4; ++k) full[nfull][k] = C.task[p][r][k]; nfull++; }
                for (int r = 0; r < ROWL; ++r) for (int k = 0; k < 4; ++k) C.task[p][r][k] = 0xffff;
                for (int t = 0; t < nfull; ++t)
                    for (int r = 0; r < 2; ++r) {
                        uint16_t* q = C.task[p][2 * t + r];
                        q[0] = (uint16_t)(full[t][0] + r); q[1] = full[t][1]; q[2] = (uint16_t)(full[t][2] + 2 * r); q[3] = full[t][3];
                    }
                C.bwd_half |= 1ull << (p - first_bwd);
            }
        }
        // The kernel adds a descriptor field to the workspace's LDS address as it is (one VALU instruction per operand instead of two): the
        // table holds BYTE offsets (< 32 KiB: a scenario's workspace is a fraction of the 160 KiB of LDS), bit 15 of field 0 = rhs task as before.
        {
            int nup_all = 0;
            for (size_t q = 0; q < pkind.size(); ++q) if (pkind[q] == 0) nup_all++;
            const int nfull_upd = nup_all - (int)C.npass_updq - (int)C.npass_updh;          // only the full-form update passes carry the rhs flag
            for (size_t q = 0; q < pkind.size(); ++q)
                for (int r = 0; r < ROWL; ++r) {
                    if (C.task[q][r][0] == 0xffff) continue;
                    for (int k = 0; k < 4; ++k) {
                        const uint32_t v = C.task[q][r][k], f = (k == 0 && (int)q < nfull_upd) ? (v & 0x8000u) : 0u;
                        C.task[q][r][k] = (uint16_t)(((v & (f ? 0x7fffu : 0xffffu)) << 3) | f);
                    }
                }
        }
        C.npass = (uint16_t)pkind.size();
        int nu = 0, ni = 0;
        for (size_t q = 0; q < pkind.size(); ++q) {
            C.pass_ntask[q] = (uint8_t)pcount[q];
            if (pkind[q] == 0) nu++; else if (pkind[q] == 1) ni++;
            // kinds must appear as contiguous phases UPD.. INV.. BWD..
            if (q > 0 && pkind[q] < pkind[q - 1]) return failx(RELMC_ERR_INVALID, "relmc_case_load: schedule phases out of order");
        }
        C.npass_upd = (uint16_t)nu; C.npass_inv = (uint16_t)ni;
        if (verbose()) { fprintf(stderr, "relmc: order %d: %d + %d + %d passes, tasks per pass:", order_variant, nu, ni, (int)pkind.size() - nu - ni); for (size_t q = 0; q < pkind.size(); ++q) fprintf(stderr, " %d", pcount[q]); fprintf(stderr, "\n"); }
    }

    // ---- lines
    for (int l = 0; l < NLT; ++l) C.l_partner[l] = -1;
    std::vector<int> pair_owner((size_t)nb * nb, -1), pair_lines((size_t)nb * nb, 0);
    std::vector<char> has_line(nb + noff, 0);
    for (int l = 0; l < nl; ++l) {
        const int f = ext2int[d->br_from[l]], t = ext2int[d->br_to[l]];
        const int lo = f < t ? f : t, hi = f < t ? t : f;
        uint32_t flags = LF_EXISTS;
        if (d->br_rate[l] != 0.0) flags |= LF_LIMITED;
        const size_t key = (size_t)hi * nb + lo;
        if (pair_owner[key] < 0) {
            pair_owner[key] = l; flags |= LF_OWNER;
            if (blk[hi][lo] < 0) return failx(RELMC_ERR_INVALID, "relmc_case_load: line outside the symbolic pattern");
            C.l_blk[l] = (uint16_t)OFFB(hi, lo);
            has_line[blk[hi][lo]] = 1;
        } else {
            if (pair_lines[key] >= 2) return failx(RELMC_ERR_UNSUPPORTED, "relmc_case_load: more than two parallel lines");
            C.l_partner[pair_owner[key]] = l;
        }
        pair_lines[key]++;
        C.l_b[l] = d->br_b[l];
        C.l_rate[l] = d->br_rate[l] / d->base_mva;
        C.l_info[l] = (uint32_t)f | ((uint32_t)t << 8) | (flags << 24);
        for (int side = 0; side < 2; ++side) {
            const int bus = side ? t : f;
            if (C.b_nline[bus] >= DEGMAX) return failx(RELMC_ERR_UNSUPPORTED, "relmc_case_load: more than 8 lines at a bus");
            C.b_line[bus][C.b_nline[bus]++] = (uint8_t)(l | (side ? 0x80 : 0));
        }
    }
    for (int i = 0; i < nb; ++i) { double s_ = 0.0; for (int e = 0; e < C.b_nline[i]; ++e) s_ += C.l_b[C.b_line[i][e] & 0x7f]; C.b_bsum[i] = s_; }   // same order as the kernel's loop
    int maxdeg = 0, maxinj_ = 0;
    for (int i = 0; i < nb; ++i) {
        uint64_t pk = 0;
        for (int e = 0; e < 8; ++e) pk |= (uint64_t)(e < C.b_nline[i] ? C.b_line[i][e] : nl) << (8 * e);
        C.b_line8[i] = pk;
        if (C.b_nline[i] > maxdeg) maxdeg = C.b_nline[i];
    }
    int nzero = 0;
    for (int k = nb; k < nb + noff; ++k) if (!has_line[k]) C.zero_off[nzero++] = (uint16_t)(4 * pos[k]);
    C.nzero = (uint16_t)nzero;
    // ---- injections
    for (int j = 0; j < ninj; ++j) {
        if (d->inj_bus[j] < 0 || d->inj_bus[j] >= nb) return failx(RELMC_ERR_INVALID, "relmc_case_load: bad injection bus");
        const int bus = ext2int[d->inj_bus[j]];
        C.i_tab[j][0] = d->inj_pmax[j] / d->base_mva;
        C.i_tab[j][1] = d->inj_pmin[j] / d->base_mva;
        C.i_tab[j][2] = d->inj_cost[j] * d->base_mva;
        C.i_tab[j][3] = d->inj_pmin[j];
        C.i_info[j] = (uint32_t)bus | ((j < ng ? IK_REAL : IK_VIRTUAL) << 8);
        if (C.b_ninj[bus] >= BINJMAX) return failx(RELMC_ERR_UNSUPPORTED, "relmc_case_load: more than 8 injections at a bus");
        C.b_inj[bus][C.b_ninj[bus]++] = (uint8_t)j;
        if (j >= ng) {
            if (C.b_vinj[bus] >= 0) return failx(RELMC_ERR_UNSUPPORTED, "relmc_case_load: two virtual generators at one bus");
            C.b_vinj[bus] = (int16_t)j;
        }
    }
    for (int i = 0; i < nb; ++i) {
        uint64_t pk = 0;
        for (int e = 0; e < 8; ++e) pk |= (uint64_t)(e < C.b_ninj[i] ? C.b_inj[i][e] : ninj) << (8 * e);
        C.b_inj8[i] = pk;
        if (C.b_ninj[i] > maxinj_) maxinj_ = C.b_ninj[i];
    }
    C.maxdeg = (uint16_t)maxdeg; C.maxinj = (uint16_t)maxinj_;
    // ---- which lane of which bus slot holds which bus in the vector phases.  A slot's gather loops run to the longest line / injection
    // list among its buses (two entries per step), so the partly filled second slot is given the buses with the shortest lists: the pair of
    // list-length limits (a, b) with the fewest steps that still admits nb - RW buses.  Everything is indexed by bus, so this is free.
    {
        for (int q = 0; q < NBT; ++q) C.b_lane[q] = 0xff;
        std::vector<int> slot_of(nb, 0);
        const int n1 = nb > ROWL ? nb - ROWL : 0;
        if (n1 > 0 && TL::BS == 2 && !so.no_bus_map) {
            int best_a = DEGMAX, best_b = BINJMAX, best_steps = 1 << 30;
            for (int a2 = 0; a2 <= DEGMAX; ++a2) for (int b2 = 0; b2 <= BINJMAX; ++b2) {
                int cnt = 0;
                for (int i = 0; i < nb; ++i) if (C.b_nline[i] <= a2 && C.b_ninj[i] <= b2) cnt++;
                const int steps = (a2 + 1) / 2 + (b2 + 1) / 2;
                if (cnt >= n1 && steps < best_steps) { best_steps = steps; best_a = a2; best_b = b2; }
            }
            std::vector<int> cand;
            for (int i = 0; i < nb; ++i) if (C.b_nline[i] <= best_a && C.b_ninj[i] <= best_b) cand.push_back(i);
            std::stable_sort(cand.begin(), cand.end(), [&](int x, int y) { return C.b_nline[x] + C.b_ninj[x] < C.b_nline[y] + C.b_ninj[y]; });
            for (int k = 0; k < n1; ++k) slot_of[cand[k]] = 1;
        } else {
            for (int i = 0; i < nb; ++i) slot_of[i] = i / ROWL;
        }
        int fill[TL::BS] = {};
        for (int i = 0; i < nb; ++i) {
            const int t = slot_of[i];
            C.b_lane[ROWL * t + fill[t]++] = (uint8_t)i;
            if (C.b_nline[i] > C.maxdeg_s[t]) C.maxdeg_s[t] = C.b_nline[i];
            if (C.b_ninj[i] > C.maxinj_s[t]) C.maxinj_s[t] = C.b_ninj[i];
        }
    }
    {   // is the intact network connected?  (lets the kernel skip the island search when no line is out)
        std::vector<int> lab(nb); for (int i = 0; i < nb; ++i) lab[i] = i;
        auto find = [&](int x) { while (lab[x] != x) { lab[x] = lab[lab[x]]; x = lab[x]; } return x; };
        for (int l = 0; l < nl; ++l) { const int a2 = find(d->br_from[l]), b2 = find(d->br_to[l]); if (a2 != b2) lab[a2] = b2; }
        int roots = 0; for (int i = 0; i < nb; ++i) if (find(i) == i) roots++;
        C.base_connected = roots == 1 ? 1 : 0;
    }
    // Bernoulli thresholds: fail iff draw_u32 < floor(U * 2^32)   (mc_sampling.m:35, strict '<')
    for (int k = 0; k < ncomp; ++k) {
        double t = std::floor(d->unavail[k] * 4294967296.0);
        if (!(t > 0)) t = 0;
        if (t > 4294967295.0) t = 4294967295.0;
        C.thr[k] = d->always_up[k] ? 0u : (uint32_t)t;   // mc_sampling.m:40-41
    }
    // ---- launch geometry: dynamic LDS = case tables + schedule + one workspace per scenario row
    const uint32_t eval_doubles = 4u * (nl + 1) + 4u * (ninj + 1);   // line / injection records (+1 zero record each): alias the workspace
    uint32_t scen = C.nws > eval_doubles ? C.nws : eval_doubles;
    scen = (scen + 1u) & ~1u;
    const uint32_t stash_off = scen;
    scen += 2u * IS * ROWL + 2u + NBT + OW / 2u;           // stash: 1/D and Np/D per injection lane (+ one zero pair); lambda per bus; outage mask words
    while ((scen & 3u) != 2u) scen += 1;                  // 16-byte aligned rows (ds_read_b128!) whose 16-B slot index differs by an odd number
    scen += 4u * (uint32_t)so.scen_pad4;                   // (ablation builds: more padding between the rows, in steps of 32 bytes)
    const uint32_t case_bytes = (uint32_t)offsetof(DevCaseT<TL>, task);     // tables copied to LDS; the pass schedule is read from global memory
    const uint32_t lds_bytes = 128u + ((case_bytes + 15u) & ~15u) + (uint32_t)SPW * WPB * scen * (uint32_t)sizeof(double) + (ROWL == 16 ? (1024u + 64u) * WPB : 0u);   // + sampling window of the fused path   // 128: solver options
    if (lds_bytes > 160u * 1024u) return failx(RELMC_ERR_UNSUPPORTED, "relmc_case_load: case needs more than 160 KiB of LDS per workgroup");
    geom.stash_off = stash_off; geom.scen_doubles = scen; geom.lds_bytes = lds_bytes;
    return RELMC_OK;
}

template int case_symbolic<Tile24>(const relmc_case_desc*, DevCaseT<Tile24>&, int, const SymOpts&, SymGeom&, std::string&);
template int case_symbolic<Tile96>(const relmc_case_desc*, DevCaseT<Tile96>&, int, const SymOpts&, SymGeom&, std::string&);

// what the solver phase of one Newton step costs under a schedule: its LDS instructions (a pass costs them whatever its fill) + kPassWeight
// per dependent pass (the update passes are LDS-pipe-bound, and every pass is one more wait on the wavefront's chain)
constexpr long kPassWeight = 4;
template <class TL>
long schedule_cost(const DevCaseT<TL>& C, int32_t* lds_out, int32_t* passes_out)
{
    const int nbwd = (int)C.npass - (int)C.npass_upd - (int)C.npass_inv;
    const int nfull = (int)C.npass_upd - (int)C.npass_updh - (int)C.npass_updq;
    long lds = 10L * nfull + 7L * C.npass_updh + 6L * C.npass_updq + 6L * C.npass_inv;
    for (int k = 0; k < nbwd; ++k) lds += (k < 64 && ((C.bwd_half >> k) & 1ull)) ? 6 : 7;
    if (lds_out) *lds_out = (int32_t)lds;
    if (passes_out) *passes_out = (int32_t)C.npass;
    return lds + kPassWeight * (long)C.npass;
}
template <class TL>
int tune_order_impl(const relmc_case_desc* d, int32_t evaluations, uint64_t seed, const int32_t* start, int32_t* order_out, int32_t* stats)
{
    const int nb = d->nb;
    SymOpts so = sym_opts_default();
    so.place_moves = 0;                                 // the cost does not depend on the operand placement
    std::string err;
    auto C = std::make_unique<DevCaseT<TL>>();
    SymGeom g;
    std::vector<int32_t> cur(nb), best(nb), cand(nb);
    int32_t lds = 0, np = 0;
    if (start) cur.assign(start, start + nb);
    else {                                              // the rule's order: external buses by internal number
        const int rc = case_symbolic<TL>(d, *C, 0, so, g, err);
        if (rc) return rc;
        for (int e = 0; e < nb; ++e) cur[C->b_int[e]] = e;
    }
    auto eval = [&](const std::vector<int32_t>& o, int32_t* l, int32_t* p) -> long {
        so.order_hint = o.data(); so.n_hint = (int)o.size();
        if (case_symbolic<TL>(d, *C, 0, so, g, err) != RELMC_OK) return 1L << 40;      // too much fill / too many passes for the tile: never accepted
        return schedule_cost(*C, l, p);
    };
    long cc = eval(cur, &lds, &np);
    if (cc >= (1L << 40)) return RELMC_ERR_INVALID;     // a start order that is not a permutation with the reference bus last, or does not fit
    if (stats) { stats[0] = lds; stats[1] = np; }
    long bc = cc; best = cur; int32_t blds = lds, bnp = np;
    uint64_t rng = seed * 0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull;
    auto rnd = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; };
    auto unif = [&]() { return (double)(rnd() >> 11) * (1.0 / 9007199254740992.0); };
    double T = 1.2; int since = 0;
    for (int it = 0; it < evaluations && nb > 2; ++it) {
        const int i = (int)(rnd() % (uint64_t)(nb - 1)), j = (int)(rnd() % (uint64_t)(nb - 1));       // the reference bus stays last
        if (i == j) continue;
        cand = cur;
        if (unif() < 0.5) std::swap(cand[i], cand[j]);
        else { const int32_t v = cand[i]; cand.erase(cand.begin() + i); cand.insert(cand.begin() + j, v); }
        int32_t l2 = 0, p2 = 0;
        const long nc = eval(cand, &l2, &p2);
        if (nc <= cc || unif() < std::exp((double)(cc - nc) / T)) {
            cur = cand; cc = nc;
            if (nc < bc) { bc = nc; best = cand; blds = l2; bnp = p2; since = 0; }
        }
        T = T * 0.9995 > 0.12 ? T * 0.9995 : 0.12;
        if (++since > 1500) { cur = best; cc = bc; since = 0; }                                         // back to the best order found so far
    }
    for (int k = 0; k < nb; ++k) order_out[k] = best[k];
    if (stats) { stats[2] = blds; stats[3] = bnp; }
    return RELMC_OK;
}

}  // namespace relmc_host

using namespace relmc_host;

extern "C" {

int32_t relmc_tune_order(const relmc_case_desc* d, int32_t evaluations, uint64_t seed, const int32_t* start, int32_t* order_out, int32_t stats_out[4])
{
    if (!d || !order_out || evaluations < 0 || d->nb < 1) return RELMC_ERR_INVALID;
    if (fits_tile24(d)) return tune_order_impl<Tile24>(d, evaluations, seed, start, order_out, stats_out);
    return tune_order_impl<Tile96>(d, evaluations, seed, start, order_out, stats_out);
}

// Host-only introspection (no device, no context): the symbolic analysis and static solver schedule relmc_case_load would build for this
// case under elimination order `order_variant` (0 = primary).  tests/test_schedule.py interprets the schedule on the CPU against a dense
// solve, which is how every ordering / scheduling change is checked before it reaches a GPU.
//   hdr[24]: tile (0 = 16-lane rows, 1 = 64-lane rows), RW, nb, noff, nws, off_rhs, npass, npass_upd, npass_inv, npass_updh, npass_updq,
//            nzero, scen_doubles, lds_bytes, modelled LDS conflict cycles before / after the placement search, MAXPASS, nl, flags, bwd_half (2), longest line / injection
//            list per bus slot (4 bytes), 2 spare
//   tasks[npass][RW][4] (0xffff = no task), pass_ntask[npass], b_int[nb] (external -> internal bus), l_blk[nl] (W offset of the owner
//   line's block, 0xffff otherwise), l_info[nl] (from | to << 8 | flags << 24, internal bus numbers), zero_off[nzero]
int32_t relmc_debug_symbolic(const relmc_case_desc* d, int32_t order_variant, const int32_t* order_hint, int32_t n_hint, int32_t* hdr, uint16_t* tasks, int64_t tasks_cap, uint8_t* pass_ntask,
                             uint8_t* b_int, uint16_t* l_blk, uint32_t* l_info, uint16_t* zero_off, char* err, int32_t err_cap, int32_t model_leaf_free)
{
    if (!d || !hdr || !tasks || !pass_ntask || !b_int || !l_blk || !l_info || !zero_off) return RELMC_ERR_INVALID;
    SymOpts so = sym_opts_default();
    if (order_hint && n_hint > 0) { so.order_hint = order_hint; so.n_hint = n_hint; }
    if (model_leaf_free >= 0) so.model_leaf_free = model_leaf_free;
    std::string errs;
    SymGeom g;
    int rc;
    auto dump = [&](const auto& C, int tile, int rw, int maxpass) {
        for (int k = 0; k < 24; ++k) hdr[k] = 0;
        hdr[0] = tile; hdr[1] = rw; hdr[2] = C.nb; hdr[3] = C.noff; hdr[4] = (int)C.nws; hdr[5] = C.off_rhs; hdr[6] = C.npass; hdr[7] = C.npass_upd;
        hdr[8] = C.npass_inv; hdr[9] = C.npass_updh; hdr[10] = C.npass_updq; hdr[11] = C.nzero; hdr[12] = (int)g.scen_doubles; hdr[13] = (int)g.lds_bytes;
        hdr[14] = (int)g.conflict_before; hdr[15] = (int)g.conflict_after; hdr[16] = maxpass; hdr[17] = C.nl;
        hdr[19] = (int32_t)(uint32_t)(C.bwd_half & 0xffffffffull); hdr[20] = (int32_t)(uint32_t)(C.bwd_half >> 32);
        hdr[21] = (int32_t)((uint32_t)C.maxdeg_s[0] | ((uint32_t)C.maxdeg_s[1] << 8) | ((uint32_t)C.maxinj_s[0] << 16) | ((uint32_t)C.maxinj_s[1] << 24));   // longest line / injection list per bus slot
        if ((int64_t)C.npass * rw * 4 > tasks_cap) return (int)RELMC_ERR_INVALID;
        for (int p = 0; p < C.npass; ++p) { pass_ntask[p] = C.pass_ntask[p]; for (int r = 0; r < rw; ++r) for (int k = 0; k < 4; ++k) {
            // back to offsets in doubles (what tests/schedule_interp.py executes); bit 15 of field 0 of a full-form update pass is the rhs flag
            const uint16_t v = C.task[p][r][k];
            const bool full_upd = p < C.npass_upd - C.npass_updh - C.npass_updq;
            tasks[((size_t)p * rw + r) * 4 + k] = C.task[p][r][0] == 0xffff ? v : (uint16_t)((k == 0 && full_upd) ? (((v & 0x7fffu) >> 3) | (v & 0x8000u)) : (v >> 3));
        } }
        for (int i = 0; i < C.nb; ++i) b_int[i] = C.b_int[i];
        for (int l = 0; l < C.nl; ++l) { l_info[l] = C.l_info[l]; l_blk[l] = ((C.l_info[l] >> 24) & LF_OWNER) ? C.l_blk[l] : (uint16_t)0xffff; }
        for (int z = 0; z < C.nzero; ++z) zero_off[z] = C.zero_off[z];
        return (int)RELMC_OK;
    };
    if (fits_tile24(d)) {
        auto C = std::make_unique<DevCaseT<Tile24>>();
        rc = case_symbolic<Tile24>(d, *C, order_variant, so, g, errs);
        if (rc == RELMC_OK) rc = dump(*C, 0, Tile24::RW, Tile24::MAXPASS);
    } else {
        auto C = std::make_unique<DevCaseT<Tile96>>();
        rc = case_symbolic<Tile96>(d, *C, order_variant, so, g, errs);
        if (rc == RELMC_OK) rc = dump(*C, 1, Tile96::RW, Tile96::MAXPASS);
    }
    if (err && err_cap > 0) { std::strncpy(err, errs.c_str(), (size_t)err_cap - 1); err[err_cap - 1] = 0; }
    return rc;
}

}  // extern "C"
