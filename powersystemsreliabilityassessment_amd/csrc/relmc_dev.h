// relmc_dev.h — device-side view of one study case (built on the host by relmc_case_load,
// resident in HBM, copied into LDS by every workgroup) and the tile constants the gfx950
// kernels are compiled for.
#pragma once
#include <stdint.h>

namespace relmc {

// ---- tile geometry: one scenario = one 16-lane DPP row of a wavefront -----------------
constexpr int ROWL = 16;           // lanes per scenario
constexpr int NBT = 24;            // bus tile (KKT order 2*NBT = 48 = 3 row slots x 16 lanes)
constexpr int KS = 3;              // KKT row slots per lane
constexpr int LS = 3;              // line slots per lane   (NLT = 48 lines)
constexpr int IS = 4;              // injection slots/lane  (NIT = 64 injections)
constexpr int NLT = LS * ROWL;
constexpr int NIT = IS * ROWL;
constexpr int PMAX = 48;           // distinct bus pairs joined by >= 1 line
constexpr int DIAG0 = 2 * PMAX;    // value arrays: [0,PMAX) lower half, [PMAX,2PMAX) upper half,
constexpr int ZIDX = DIAG0 + NBT;  //               [DIAG0, DIAG0+NBT) diagonal, ZIDX = constant 0
constexpr int CARR = ZIDX + 1;
constexpr int DEGMAX = 8;          // lines per bus
constexpr int BINJMAX = 8;         // injections per bus
constexpr int NCOMPMAX = 128;      // sampled components (generators + lines)

// l_info: from | to<<8 | pair<<16 | flags<<24
constexpr uint32_t LF_EXISTS = 1u, LF_OWNER = 2u, LF_LIMITED = 4u;
// i_info: bus | kind<<8
constexpr uint32_t IK_NONE = 0u, IK_REAL = 1u, IK_VIRTUAL = 2u;

struct DevCase {
    int32_t nb, ng, nl, nd, ninj, ncomp, ref_bus, npair;
    double base_mva, total_load;
    uint32_t exist_mask;            // bit i = bus i exists
    uint32_t pad0;
    // lines
    double l_b[NLT];
    double l_rate[NLT];             // p.u. (0 = unlimited)
    uint32_t l_info[NLT];
    int32_t l_partner[NLT];         // the other line of the same bus pair, -1 if none
    // injections (real generators then virtual generators = loads)
    double i_lo[NIT];               // p.u.
    double i_hi[NIT];               // p.u.
    double i_cost[NIT];             // c1 * baseMVA (opf_setup)
    double i_pmin_mw[NIT];          // original Pmin in MW (nodal shed = Pg - Pmin, mc_simulation.m:86)
    uint32_t i_info[NIT];
    // per-bus incidence
    uint8_t b_nline[NBT];
    uint8_t b_line[NBT][DEGMAX];    // line id | 0x80 when the bus is the line's 'to' end
    uint8_t b_ninj[NBT];
    uint8_t b_inj[NBT][BINJMAX];
    int8_t b_vinj[NBT];             // virtual generator at the bus, -1 if none
    uint8_t b_ext[NBT];             // internal tile position -> external bus number
    uint8_t b_int[NBT];             // external bus number -> internal tile position
    uint8_t T[NBT][NBT];            // index of K(bus i, bus c) in the value arrays
    uint32_t fill[NBT];             // symbolic fill: bit j of fill[i] = block (i, j) can be non-zero when bus i is
                                    // eliminated (j later in the sequence); superset over all outage states
    uint32_t thr[NCOMPMAX];         // Bernoulli thresholds floor(U*2^32)
};

// per-lane partial accumulators written once per workgroup-row, reduced by relmc_finalize_kernel
struct Partial {
    double dns, dns2;
    double shed[IS];
    uint32_t n, nfail, nsing, ninf, nnc, iters;
    uint32_t cf_inj[IS];
    uint32_t cf_line[LS];
    uint32_t pad;
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
    Partial* partial;               // [gridDim.x * 64]
};

}  // namespace relmc
