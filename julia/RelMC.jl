# RelMC.jl — Julia host layer over the C ABI of include/relmc.h (north-star: "host code in Julia
# calling HIP through a thin C-ABI/ccall layer").  Keeps the reference's call surface:
#   mc_sampling(U, n, Ng, Nl)                     (Montecarlo_nsq_single/mc_sampling.m:2)
#   mc_simulation(state, TestSystem, mpopt, Ng, Nl) (mc_simulation.m:1)
#   nsqMain(; ...)                                (nsqMain.m:208-406)
# NOTE: Julia is not installed in the build image nor on the GPU box, so this file is NOT executed
# by the repository's tests; the Python mirror (powersystemsreliabilityassessment_amd/api.py) is.
module RelMC

const LIB = joinpath(@__DIR__, "..", "powersystemsreliabilityassessment_amd", "csrc", "librelmc.so")
const MAX_BUS = 128
const MAX_COMP = 256

struct CaseDesc
    base_mva::Cdouble
    nb::Int32; ng::Int32; nl::Int32; nd::Int32; ref_bus::Int32
    bus_pd::Ptr{Cdouble}
    inj_bus::Ptr{Int32}
    inj_pmin::Ptr{Cdouble}; inj_pmax::Ptr{Cdouble}; inj_cost::Ptr{Cdouble}
    br_from::Ptr{Int32}; br_to::Ptr{Int32}
    br_b::Ptr{Cdouble}; br_rate::Ptr{Cdouble}
    unavail::Ptr{Cdouble}
    always_up::Ptr{UInt8}
    total_load::Cdouble
end

mutable struct SolverOpts
    singular_policy::Int32; max_it::Int32
    feastol::Cdouble; gradtol::Cdouble; comptol::Cdouble; costtol::Cdouble
    xi::Cdouble; sigma::Cdouble; z0::Cdouble; alpha_min::Cdouble; max_stepsize::Cdouble
    screen::Int32; reserved::Int32            # screen = 1: zero-curtailment pre-screen (relmc.h), relmc_acc.n_screened counts the skipped units
    SolverOpts() = new()
end

mutable struct Acc
    n::Int64; n_fail::Int64; n_singular::Int64; n_infeasible::Int64; n_nonconverged::Int64; sum_iters::Int64
    comp_fail::NTuple{MAX_COMP,Int64}
    n_screened::Int64
    sum_dns::Cdouble; sum_dns2::Cdouble
    sum_nodal::NTuple{MAX_BUS,Cdouble}
    Acc() = new()
end

mutable struct Indices
    n::Int64
    edns::Cdouble; lole::Cdouble; plc::Cdouble; beta::Cdouble; eens::Cdouble; mean_iters::Cdouble
    nodal_eens::NTuple{MAX_BUS,Cdouble}
    comp_importance::NTuple{MAX_COMP,Cdouble}
    Indices() = new()
end

"isbits image of relmc_solver_opts, for embedding in NsqOpts"
struct SolverOptsC
    singular_policy::Int32; max_it::Int32
    feastol::Cdouble; gradtol::Cdouble; comptol::Cdouble; costtol::Cdouble
    xi::Cdouble; sigma::Cdouble; z0::Cdouble; alpha_min::Cdouble; max_stepsize::Cdouble
    screen::Int32; reserved::Int32
end
SolverOptsC(o::SolverOpts) = SolverOptsC(o.singular_policy, o.max_it, o.feastol, o.gradtol, o.comptol, o.costtol, o.xi, o.sigma, o.z0,
                                         o.alpha_min, o.max_stepsize, o.screen, o.reserved)

"relmc_nsq_opts (include/relmc.h): the options of the whole nsqMain loop"
struct NsqOpts
    beta_limit::Cdouble; max_samples::Int64; batch::Int64; seed::UInt64; hours_per_year::Cdouble
    solver::SolverOptsC
    history_cap::Int64
    beta_history::Ptr{Cdouble}; edns_history::Ptr{Cdouble}; lole_history::Ptr{Cdouble}; plc_history::Ptr{Cdouble}
    distinct_states::Int32
end

# relmc_nsq_result is read out of a byte buffer at these offsets (Acc and Indices are mutable mirrors and cannot be embedded)
const NSQ_RESULT_BYTES = 8 * (7 + MAX_COMP + 2 + MAX_BUS) + 8 * (7 + MAX_BUS + MAX_COMP) + 40
const NSQ_RESULT_IDX = 8 * (7 + MAX_COMP + 2 + MAX_BUS)
const NSQ_RESULT_TAIL = NSQ_RESULT_IDX + 8 * (7 + MAX_BUS + MAX_COMP)     # checkpoints, converged, wall_seconds, kernel_seconds, batches

"TestSystem after the load model of nsqMain.m:121-153 (0-based indices inside)."
struct TestSystem
    base_mva::Float64; nb::Int; ng::Int; nl::Int; nd::Int; ref_bus::Int
    bus_pd::Vector{Float64}
    inj_bus::Vector{Int32}; inj_pmin::Vector{Float64}; inj_pmax::Vector{Float64}; inj_cost::Vector{Float64}
    br_from::Vector{Int32}; br_to::Vector{Int32}; br_b::Vector{Float64}; br_rate::Vector{Float64}
    unavail::Vector{Float64}; always_up::Vector{UInt8}
    load::Float64                     # TestSystem.load, nsqMain.m:125
end

mutable struct Engine
    h::Ptr{Cvoid}
    sys::TestSystem
end

check(rc, h, what) = rc == 0 || error("$what failed ($rc): " * unsafe_string(ccall((:relmc_last_error, LIB), Cstring, (Ptr{Cvoid},), h)))

# The primary elimination orders this package ships for the two IEEE test systems (0-based bus numbers, reference bus last; tuned offline with
# relmc_tune_order, identical to case24.RTS24_ELIM_ORDER / case96.RTS96_ELIM_ORDER of the Python host, so that every host runs the same pass
# program -- an order changes the rounding of the factorisation: iteration counts of single states may move by one, DESIGN.md 3.2).
# Pass them as `Engine(sys; elim_order=RTS24_ELIM_ORDER)`; without a hint the library's rule decides.
const RTS24_ELIM_ORDER = Int32[4, 21, 18, 5, 19, 17, 3, 6, 22, 23, 1, 20, 2, 11, 13, 16, 10, 0, 14, 15, 7, 8, 9, 12]
const RTS96_ELIM_ORDER = Int32[69, 41, 27, 3, 61, 2, 67, 42, 4, 17, 47, 60, 11, 53, 66, 18, 36, 54, 43, 5, 52, 71, 24, 51, 59, 64, 30, 44, 21, 6, 62, 29, 37, 14, 58, 48, 35, 55, 7, 49, 28, 68, 1, 34, 13, 25, 19, 31, 45, 40, 23, 56, 16, 38, 72, 65, 0, 57, 10, 33, 32, 39, 63, 15, 50, 9, 8, 70, 26, 22, 20, 46, 12]

# `elim_order`: optional primary elimination order of the device solver's static schedule (0-based bus numbers, the reference bus last),
# e.g. one found by `tune_order` below; `nothing` = the library's rule (relmc_case_order_hint).
function Engine(sys::TestSystem; device::Integer=0, elim_order::Union{Nothing,Vector{Int32}}=nothing)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    rc = ccall((:relmc_ctx_create, LIB), Int32, (Int32, Ref{Ptr{Cvoid}}), device, h)
    rc == 0 || error("relmc_ctx_create failed ($rc): no usable HIP device (there is no CPU fallback)")
    eng = Engine(h[], sys)
    GC.@preserve sys begin
        d = CaseDesc(sys.base_mva, sys.nb, sys.ng, sys.nl, sys.nd, sys.ref_bus, pointer(sys.bus_pd),
                     pointer(sys.inj_bus), pointer(sys.inj_pmin), pointer(sys.inj_pmax), pointer(sys.inj_cost),
                     pointer(sys.br_from), pointer(sys.br_to), pointer(sys.br_b), pointer(sys.br_rate),
                     pointer(sys.unavail), pointer(sys.always_up), sys.load)
        if elim_order !== nothing
            check(ccall((:relmc_case_order_hint, LIB), Int32, (Ptr{Cvoid}, Ptr{Int32}, Int32), eng.h, elim_order, length(elim_order)), eng.h, "relmc_case_order_hint")
        end
        check(ccall((:relmc_case_load, LIB), Int32, (Ptr{Cvoid}, Ref{CaseDesc}), eng.h, Ref(d)), eng.h, "relmc_case_load")
    end
    finalizer(e -> ccall((:relmc_ctx_destroy, LIB), Cvoid, (Ptr{Cvoid},), e.h), eng)
    return eng
end

"mpoption('PF_DC',1,...,'OPF_ALG_DC',200,'OPF_FLOW_LIM',1) of nsqMain.m:185-186 = MIPS defaults.  screen = 1: the zero-curtailment pre-screen
(relmc.h: states with a proven LP optimum of 0 are counted, not solved; Acc.n_screened counts them; every other output unchanged)."
function mpoption(; singular_policy::Integer=0, screen::Integer=0)
    o = SolverOpts()
    ccall((:relmc_solver_opts_default, LIB), Cvoid, (Ref{SolverOpts},), o)
    o.singular_policy = singular_policy
    o.screen = screen
    return o
end

"eqstatus = mc_sampling(failure_probabilities, num_samples, numGenerators, numLines): n x (Ng+Nl), 1 = failed."
function mc_sampling(eng::Engine, failure_probabilities, num_samples::Integer, numGenerators::Integer, numLines::Integer;
                     seed::Integer=1, first_index::Integer=0)
    @assert numGenerators == eng.sys.ng && numLines == eng.sys.nl
    @assert failure_probabilities === nothing || failure_probabilities == eng.sys.unavail
    ncomp = numGenerators + numLines
    out = Matrix{UInt8}(undef, ncomp, num_samples)          # column-major: one scenario per column = row-major n x ncomp
    check(ccall((:relmc_mc_sampling, LIB), Int32, (Ptr{Cvoid}, UInt64, UInt64, Int64, Ptr{UInt8}),
                eng.h, seed, first_index, num_samples, out), eng.h, "relmc_mc_sampling")
    return permutedims(out)
end

"[dns, nodal_dns] = mc_simulation(component_states, TestSystem, mpopt, numGenerators, numLines); states: n x (Ng+Nl)."
function mc_simulation(eng::Engine, component_states::AbstractMatrix, mpopt::SolverOpts=mpoption(),
                       numGenerators::Integer=eng.sys.ng, numLines::Integer=eng.sys.nl)
    n = size(component_states, 1)
    st = Matrix{UInt8}(permutedims(component_states .!= 0))
    dns = Vector{Float64}(undef, n); nodal = Matrix{Float64}(undef, eng.sys.nb, n)
    status = Vector{Int32}(undef, n); iters = Vector{Int32}(undef, n)
    check(ccall((:relmc_mc_simulation, LIB), Int32,
                (Ptr{Cvoid}, Ptr{UInt8}, Int64, Ref{SolverOpts}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Int32}, Ptr{Int32}),
                eng.h, st, n, mpopt, dns, nodal, status, iters), eng.h, "relmc_mc_simulation")
    return dns, permutedims(nodal), status, iters
end

"One pass of the nsqMain loop body over global scenarios [first_index, first_index+n): additive accumulators."
function nsq_accumulate(eng::Engine, seed::Integer, first_index::Integer, n::Integer, mpopt::SolverOpts=mpoption())
    acc = Acc()
    check(ccall((:relmc_nsq_accumulate, LIB), Int32, (Ptr{Cvoid}, UInt64, UInt64, Int64, Ref{SolverOpts}, Ref{Acc}),
                eng.h, seed, first_index, n, mpopt, acc), eng.h, "relmc_nsq_accumulate")
    return acc
end

"The reference's unique-state database (nsqMain.m:220-245) per range: every distinct state solved once, weighted by its count."
function nsq_accumulate_distinct(eng::Engine, seed::Integer, first_index::Integer, n::Integer, mpopt::SolverOpts=mpoption())
    acc = Acc(); nd = Ref{Int64}(0)
    check(ccall((:relmc_nsq_accumulate_distinct, LIB), Int32, (Ptr{Cvoid}, UInt64, UInt64, Int64, Ref{SolverOpts}, Ref{Acc}, Ref{Int64}),
                eng.h, seed, first_index, n, mpopt, acc, nd), eng.h, "relmc_nsq_accumulate_distinct")
    return acc, nd[]
end

function indices(eng::Engine, acc::Acc; hours_per_year=8760.0)
    out = Indices()
    ccall((:relmc_nsq_indices, LIB), Cvoid, (Ref{Acc}, Int32, Int32, Cdouble, Ref{Indices}),
          acc, eng.sys.nb, eng.sys.ng + eng.sys.nl, hours_per_year, out)
    return out
end

# ---- the reference's persistent unique-state database (nsqMain.m:91-99, 220-278) on the device -------------------------

struct DbStats
    rows::Int64; samples::Int64; new_rows::Int64; batch_distinct::Int64
end

db_reset(eng::Engine) = check(ccall((:relmc_db_reset, LIB), Int32, (Ptr{Cvoid},), eng.h), eng.h, "relmc_db_reset")
"(units evaluated a second time under the alternate elimination order, how many of them then converged) since the case was loaded"
function retry_stats(eng::Engine)
    u = Ref{Int64}(0); c = Ref{Int64}(0)
    check(ccall((:relmc_retry_stats, LIB), Int32, (Ptr{Cvoid}, Ref{Int64}, Ref{Int64}), eng.h, u, c), eng.h, "relmc_retry_stats")
    return (u[], c[])
end
"units that did not fit the kernel's list of non-converged units and kept their first-attempt results (relmc_retry_overflow)"
function retry_overflow(eng::Engine)
    u = Ref{Int64}(0)
    check(ccall((:relmc_retry_overflow, LIB), Int32, (Ptr{Cvoid}, Ref{Int64}), eng.h, u), eng.h, "relmc_retry_overflow")
    return u[]
end
"(primary static elimination order 0/1/2, failures of each probed order among the 8192 calibration states; -1 = not probed)"
function case_order(eng::Engine)
    p = Ref{Int32}(0); f = zeros(Int32, 3)
    check(ccall((:relmc_case_order, LIB), Int32, (Ptr{Cvoid}, Ref{Int32}, Ptr{Int32}), eng.h, p, f), eng.h, "relmc_case_order")
    return (Int(p[]), f)
end
"""
    tune_order(sys; evaluations=20000, seed=1, start=nothing) -> (order, (lds_before, passes_before, lds_after, passes_after))

relmc_tune_order: host-only search of the primary elimination order against the library's own scheduler (no GPU needed); pass the result to
`Engine(sys; elim_order=order)`.
"""
function tune_order(sys::TestSystem; evaluations::Integer=20000, seed::Integer=1, start::Union{Nothing,Vector{Int32}}=nothing)
    order = zeros(Int32, sys.nb); st = zeros(Int32, 4)
    GC.@preserve sys begin
        d = CaseDesc(sys.base_mva, sys.nb, sys.ng, sys.nl, sys.nd, sys.ref_bus, pointer(sys.bus_pd),
                     pointer(sys.inj_bus), pointer(sys.inj_pmin), pointer(sys.inj_pmax), pointer(sys.inj_cost),
                     pointer(sys.br_from), pointer(sys.br_to), pointer(sys.br_b), pointer(sys.br_rate),
                     pointer(sys.unavail), pointer(sys.always_up), sys.load)
        rc = ccall((:relmc_tune_order, LIB), Int32, (Ref{CaseDesc}, Int32, UInt64, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}),
                   Ref(d), evaluations, seed, start === nothing ? C_NULL : start, order, st)
        rc == 0 || error("relmc_tune_order failed ($rc)")
    end
    return order, Tuple(Int.(st))
end
"(rows, samples) of the state database"
function db_size(eng::Engine)
    rows = Ref{Int64}(0); samples = Ref{Int64}(0)
    check(ccall((:relmc_db_size, LIB), Int32, (Ptr{Cvoid}, Ref{Int64}, Ref{Int64}), eng.h, rows, samples), eng.h, "relmc_db_size")
    return (rows[], samples[])
end

"One pass of the loop body (dedupe, count bumps of known states, evaluation of the new ones, indices from ALL rows): (Acc of the whole database, DbStats)."
function nsq_db_batch(eng::Engine, seed::Integer, first_index::Integer, n::Integer, mpopt::SolverOpts=mpoption())
    acc = Acc(); st = Ref(DbStats(0, 0, 0, 0))
    check(ccall((:relmc_nsq_db_batch, LIB), Int32, (Ptr{Cvoid}, UInt64, UInt64, Int64, Ref{SolverOpts}, Ref{Acc}, Ref{DbStats}),
                eng.h, seed, first_index, n, mpopt, acc, st), eng.h, "relmc_nsq_db_batch")
    return acc, st[]
end

"state_database rows [first_row, first_row + n_rows) in the reference's column layout (nsqMain.m:91-99)."
function db_export(eng::Engine, first_row::Integer, n_rows::Integer)
    ncomp = eng.sys.ng + eng.sys.nl; nb = eng.sys.nb
    states = Matrix{UInt8}(undef, ncomp, n_rows); count = Vector{Int64}(undef, n_rows); dns = Vector{Float64}(undef, n_rows)
    flag = Vector{Int32}(undef, n_rows); nodal = Matrix{Float64}(undef, nb, n_rows)
    check(ccall((:relmc_db_export, LIB), Int32,
                (Ptr{Cvoid}, Int64, Int64, Ptr{UInt8}, Ptr{Int64}, Ptr{Cdouble}, Ptr{Int32}, Ptr{Cdouble}, Ptr{Int32}, Ptr{Int32}, Ptr{UInt8}),
                eng.h, first_row, n_rows, states, count, dns, flag, nodal, C_NULL, C_NULL, C_NULL), eng.h, "relmc_db_export")
    return hcat(Float64.(permutedims(states)), Float64.(count), dns, Float64.(flag), permutedims(nodal))     # the reference's matrix
end

"Resume: a state_database matrix in the reference's column layout (as db_export returns it) back into the EMPTY database."
function db_import(eng::Engine, db::AbstractMatrix{Float64}; mpopt::SolverOpts = mpoption())
    ncomp = eng.sys.ng + eng.sys.nl; nb = eng.sys.nb; n = size(db, 1)
    states = Matrix{UInt8}(permutedims(db[:, 1:ncomp] .!= 0)); count = Vector{Int64}(round.(Int64, db[:, ncomp + 1]))
    dns = Vector{Float64}(db[:, ncomp + 2]); nodal = Matrix{Float64}(permutedims(db[:, ncomp + 4:ncomp + 3 + nb]))
    check(ccall((:relmc_db_import, LIB), Int32,
                (Ptr{Cvoid}, Ref{SolverOpts}, Int64, Ptr{UInt8}, Ptr{Int64}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Int32}, Ptr{Int32}, Ptr{UInt8}),
                eng.h, mpopt, n, states, count, dns, nodal, C_NULL, C_NULL, C_NULL), eng.h, "relmc_db_import")
end

# ---- multi-GPU: one process per GPU (Distributed / MPI.jl), ONE RCCL all-reduce of the accumulators per convergence check ---
const COMM_ID_BYTES = 128
function comm_unique_id()
    id = Vector{UInt8}(undef, COMM_ID_BYTES)
    rc = ccall((:relmc_comm_unique_id, LIB), Int32, (Ptr{UInt8},), id)
    rc == 0 || error("relmc_comm_unique_id failed ($rc)")
    return id                          # rank 0 broadcasts these 128 bytes by its own means (Distributed.remotecall, MPI.Bcast!, a file)
end
comm_init(eng::Engine, nranks::Integer, rank::Integer, id::Vector{UInt8}) =
    check(ccall((:relmc_comm_init, LIB), Int32, (Ptr{Cvoid}, Int32, Int32, Ptr{UInt8}), eng.h, nranks, rank, id), eng.h, "relmc_comm_init")
function comm_allreduce!(eng::Engine, acc::Acc)
    check(ccall((:relmc_comm_allreduce_acc, LIB), Int32, (Ptr{Cvoid}, Ref{Acc}), eng.h, acc), eng.h, "relmc_comm_allreduce_acc")
    return acc
end
comm_destroy(eng::Engine) = ccall((:relmc_comm_destroy, LIB), Int32, (Ptr{Cvoid},), eng.h)
"The host's own transport instead of RCCL: `fn = @cfunction(f, Int32, (Ptr{Cvoid}, Ptr{Acc}))` must leave the sum over all ranks in the
accumulators it is handed (0 = ok), e.g. MPI.Allreduce! on the two halves of the struct (relmc_comm_set_host_allreduce)."
comm_set_host_allreduce(eng::Engine, nranks::Integer, rank::Integer, fn::Ptr{Cvoid}, user::Ptr{Cvoid}=C_NULL) =
    check(ccall((:relmc_comm_set_host_allreduce, LIB), Int32, (Ptr{Cvoid}, Int32, Int32, Ptr{Cvoid}, Ptr{Cvoid}), eng.h, nranks, rank, fn, user),
          eng.h, "relmc_comm_set_host_allreduce")
"Vector transport of the host collective: `fn = @cfunction(f, Int32, (Ptr{Cvoid}, Ptr{Cdouble}, Int64))` leaves the sum over all ranks of `count` doubles
in the buffer (e.g. MPI.Allreduce!); the library's vector all-reduces (annual indices, per-checkpoint sums) are then one callback each
(relmc_comm_set_host_allreduce_f64)."
comm_set_host_allreduce_f64(eng::Engine, fn::Ptr{Cvoid}, user::Ptr{Cvoid}=C_NULL) =
    check(ccall((:relmc_comm_set_host_allreduce_f64, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), eng.h, fn, user), eng.h, "relmc_comm_set_host_allreduce_f64")
"(kind 0 none / 1 RCCL / 2 host, ranks, this rank, all-reduces issued, seconds in them) as the communicator itself reports (relmc_comm_info)"
function comm_info(eng::Engine)
    kind = Ref{Int32}(0); n = Ref{Int32}(0); r = Ref{Int32}(0); calls = Ref{Int64}(0); sec = Ref{Cdouble}(0.0)
    check(ccall((:relmc_comm_info, LIB), Int32, (Ptr{Cvoid}, Ref{Int32}, Ref{Int32}, Ref{Int32}, Ref{Int64}, Ref{Cdouble}), eng.h, kind, n, r, calls, sec),
          eng.h, "relmc_comm_info")
    return (kind=kind[], nranks=n[], rank=r[], allreduces=calls[], seconds=sec[])
end

"nsqMain: `while beta > beta_limit && n < max_iterations` (nsqMain.m:208-318) + post-processing (:345-393).
distinct_states: false = every sample solved; true = distinct states of each batch solved once; :database = the reference's
persistent unique-state database across batches.  More than one rank: give the engine a communicator first (comm_init = RCCL, or
comm_set_host_allreduce = the host's own transport) and call nsqMain on EVERY rank -- the library splits every batch over the
ranks, all-reduces once per batch and hands every rank the same result (relmc_nsq_run); there is no loop on the Julia side.
verbose: the reference's console output."
function nsqMain(eng::Engine; beta_limit=0.0017, max_iterations=100_000, samples_per_batch=100, seed=1, mpopt=mpoption(),
                 distinct_states=false, verbose=false)
    total = Acc(); ccall((:relmc_acc_zero, LIB), Cvoid, (Ref{Acc},), total)
    idx = Indices(); rows = 0
    t0 = time()
    # the loop runs inside the library (relmc_nsq_run), which also evaluates small batches such as the reference's 100 many checkpoints
    # per launch (DESIGN.md 6.8) and shards the batches over the ranks of the engine's communicator
    ncp = cld(max_iterations, samples_per_batch)
    beta_history = zeros(ncp); edns_history = zeros(ncp); lole_history = zeros(ncp); plc_history = zeros(ncp)
    mode = distinct_states === :database ? 2 : (distinct_states === true ? 1 : 0)
    buf = zeros(UInt8, NSQ_RESULT_BYTES)
    k = 0
    GC.@preserve beta_history edns_history lole_history plc_history buf total idx begin
        o = NsqOpts(beta_limit, max_iterations, samples_per_batch, seed, 8760.0, SolverOptsC(mpopt), ncp, pointer(beta_history),
                    pointer(edns_history), pointer(lole_history), pointer(plc_history), mode)
        check(ccall((:relmc_nsq_run, LIB), Int32, (Ptr{Cvoid}, Ref{NsqOpts}, Ptr{UInt8}), eng.h, Ref(o), buf), eng.h, "relmc_nsq_run")
        unsafe_copyto!(Ptr{UInt8}(pointer_from_objref(total)), pointer(buf), NSQ_RESULT_IDX)
        unsafe_copyto!(Ptr{UInt8}(pointer_from_objref(idx)), pointer(buf) + NSQ_RESULT_IDX, NSQ_RESULT_TAIL - NSQ_RESULT_IDX)
        k = unsafe_load(Ptr{Int64}(pointer(buf) + NSQ_RESULT_TAIL))
    end
    resize!(beta_history, k); resize!(edns_history, k); resize!(lole_history, k); resize!(plc_history, k)
    done = idx.n; beta = idx.beta
    mode == 2 && (rows = db_size(eng)[1])
    if verbose
        for c in 1:k
            n_c = min(c * samples_per_batch, done)
            n_c % 1000 == 0 && println("Iteration ", lpad(n_c, 6), ": Beta = ", round(beta_history[c], digits=6), ", EDNS = ",
                                       round(edns_history[c], digits=4), " MW, LOLE = ", round(lole_history[c], digits=4), " hr/yr")
        end
    end
    nb = eng.sys.nb; nc = eng.sys.ng + eng.sys.nl
    nodal = collect(idx.nodal_eens[1:nb]); imp = collect(idx.comp_importance[1:nc])
    if verbose                                                                                    # nsqMain.m:325-393
        println("Total simulation time: ", round(time() - t0, digits=2), " seconds\nTotal iterations: ", done)
        distinct_states === :database && println("Unique states evaluated: ", rows)
        println("Convergence achieved: ", beta <= beta_limit ? "YES" : "NO")
        println("EDNS (Expected Demand Not Supplied): ", round(idx.edns, digits=4), " MW\nLOLE (Loss of Load Expectation): ",
                round(idx.lole, digits=4), " hours/year\nPLC (Probability of Load Curtailment): ", round(idx.plc, digits=6),
                "\nBeta (Coefficient of Variation): ", round(idx.beta, digits=6), "\nTop 5 Buses by EENS (MWh/yr):")
        for k in sortperm(nodal, rev=true)[1:min(5, nb)]
            nodal[k] > 0 && println("  Bus ", lpad(k, 2), ": ", round(nodal[k] * 8760, digits=4), " MWh/yr")
        end
        if total.n_fail > 0
            println("Top 5 Critical Components (Prob. Down given System Failure):")
            for c in sortperm(imp, rev=true)[1:min(5, nc)]
                println("  ", c <= eng.sys.ng ? "Gen " * lpad(c, 2) : "Line " * lpad(c - eng.sys.ng, 2), ": ", round(imp[c] * 100, digits=2), "%")
            end
        else
            println("No failure events recorded to analyze weak points.")
        end
    end
    return (accumulated_edns=idx.edns, accumulated_lole=idx.lole, plc=idx.plc, current_beta=idx.beta,
            current_iteration=done, nodal_eens=nodal, comp_importance=imp, beta_history=beta_history, edns_history=edns_history,
            lole_history=lole_history, plc_history=plc_history, database_row_count=rows)
end

# ---- sequential track (Montecarlo_seq/) and HL1 copper sheet (GeneratingAdequacy/PowerSystemAdequacy.jl) ----------------

struct SeqYear
    ens::Cdouble; dlc::Cdouble; nlc::Cdouble; n_contingency::Int64
end

"relmc_seq_load: `reliability_data` = seqmeantime() [(Ng+Nl) x 2] = [MTTF MTTR]; `load_scale_factors` = anloducurve(hours)[3]."
function seq_load(eng::Engine, reliability_data::AbstractMatrix, load_scale_factors::Vector{Float64})
    mttf = Vector{Float64}(reliability_data[:, 1]); mttr = Vector{Float64}(reliability_data[:, 2])
    check(ccall((:relmc_seq_load, LIB), Int32, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Int32, Ptr{Cdouble}),
                eng.h, mttf, mttr, length(load_scale_factors), load_scale_factors), eng.h, "relmc_seq_load")
end

"state_duration_matrix = seq_mcsampling(reliability_data, Ng, Nl, num_years, hours_per_year): (Ng+Nl) x (years*hours), 1 = down."
function seq_mcsampling(eng::Engine, num_years::Integer, hours_per_year::Integer; seed::Integer=1, first_year::Integer=0)
    ncomp = eng.sys.ng + eng.sys.nl
    out = Matrix{UInt8}(undef, ncomp, num_years * hours_per_year)
    check(ccall((:relmc_seq_mcsampling, LIB), Int32, (Ptr{Cvoid}, UInt64, UInt64, Int32, Ptr{UInt8}),
                eng.h, seed, first_year, num_years, out), eng.h, "relmc_seq_mcsampling")
    return out
end

"[curtailment_mw, nodal] = seq_mcsimulation(component_status, load_scale_factor, ...), batched: states n x (Ng+Nl), scales n."
function seq_mcsimulation(eng::Engine, component_status::AbstractMatrix, load_scale_factor::Vector{Float64}, mpopt::SolverOpts=mpoption())
    n = size(component_status, 1)
    st = Matrix{UInt8}(permutedims(component_status .!= 0))
    dns = Vector{Float64}(undef, n); nodal = Matrix{Float64}(undef, eng.sys.nb, n)
    check(ccall((:relmc_seq_mcsimulation, LIB), Int32,
                (Ptr{Cvoid}, Ptr{UInt8}, Ptr{Cdouble}, Int64, Ref{SolverOpts}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Int32}, Ptr{Int32}),
                eng.h, st, load_scale_factor, n, mpopt, dns, nodal, C_NULL, C_NULL), eng.h, "relmc_seq_mcsimulation")
    return dns, permutedims(nodal)
end

"Body of the `for iYear` loop (seqMain.m:85-176) for years [first_year, first_year + n_years), fused on the GPU."
function seq_years(eng::Engine, seed::Integer, first_year::Integer, n_years::Integer, mpopt::SolverOpts=mpoption(); curtail_threshold=0.01)
    yrs = Vector{SeqYear}(undef, n_years); acc = Acc()
    check(ccall((:relmc_seq_years, LIB), Int32, (Ptr{Cvoid}, UInt64, UInt64, Int32, Ref{SolverOpts}, Cdouble, Ptr{SeqYear}, Ref{Acc}),
                eng.h, seed, first_year, n_years, mpopt, curtail_threshold, yrs, acc), eng.h, "relmc_seq_years")
    return yrs, acc
end

"relmc_seq_opts (include/relmc.h): the options of the whole seqMain loop"
struct SeqOpts
    cov_threshold::Cdouble; max_years::Int32; batch_years::Int32; seed::UInt64; curtail_threshold::Cdouble
    solver::SolverOptsC
    years_cap::Int64
    results_year::Ptr{SeqYear}; cum_eens::Ptr{Cdouble}; cum_cov::Ptr{Cdouble}
end

# relmc_seq_result is read out of a byte buffer at these offsets (like relmc_nsq_result)
const SEQ_RESULT_ACC = 56                                                   # final_year, converged, eens, cov, lole, lolf, plc, n_contingency
const SEQ_RESULT_NODAL = SEQ_RESULT_ACC + 8 * (7 + MAX_COMP + 2 + MAX_BUS)
const SEQ_RESULT_IMP = SEQ_RESULT_NODAL + 8 * MAX_BUS
const SEQ_RESULT_TAIL = SEQ_RESULT_IMP + 8 * MAX_COMP                       # wall_seconds, kernel_seconds
const SEQ_RESULT_BYTES = SEQ_RESULT_TAIL + 16

"""
seqMain (seqMain.m:85-262): annual indices until CoV(EENS) < cov_threshold, then LOLE / LOLF, nodal EENS (:218) and component importance
(:233).  The loop, its stopping rule and the post-processing run below the C ABI (relmc_seq_run); with a communicator in the engine's
context the same call on every rank is the multi-rank run.
"""
function seqMain(eng::Engine; max_sim_years=4000, cov_threshold=0.05, curtail_threshold=0.01, seed=1, mpopt=mpoption(), batch_years=0)
    yrs = Vector{SeqYear}(undef, max_sim_years); cum_eens = zeros(Float64, max_sim_years); cum_cov = zeros(Float64, max_sim_years)
    buf = zeros(UInt8, SEQ_RESULT_BYTES)
    acc = Acc()
    GC.@preserve yrs cum_eens cum_cov buf acc begin
        o = SeqOpts(cov_threshold, max_sim_years, batch_years, seed, curtail_threshold, SolverOptsC(mpopt), max_sim_years,
                    pointer(yrs), pointer(cum_eens), pointer(cum_cov))
        check(ccall((:relmc_seq_run, LIB), Int32, (Ptr{Cvoid}, Ref{SeqOpts}, Ptr{UInt8}), eng.h, Ref(o), buf), eng.h, "relmc_seq_run")
        unsafe_copyto!(Ptr{UInt8}(pointer_from_objref(acc)), pointer(buf) + SEQ_RESULT_ACC, SEQ_RESULT_NODAL - SEQ_RESULT_ACC)
    end
    rd(T, off) = GC.@preserve buf unsafe_load(Ptr{T}(pointer(buf) + off))
    k = Int(rd(Int32, 0)); nb = eng.sys.nb; nc = eng.sys.ng + eng.sys.nl
    resize!(yrs, k); resize!(cum_eens, k); resize!(cum_cov, k)
    return (final_year=k, converged=rd(Int32, 4) != 0, eens=rd(Cdouble, 8), cov=rd(Cdouble, 16), lole=rd(Cdouble, 24), lolf=rd(Cdouble, 32),
            results_year=(ens=[y.ens for y in yrs], dlc=[y.dlc for y in yrs], nlc=[y.nlc for y in yrs]), results_cum=(eens=cum_eens, cov=cum_cov),
            nodal_eens_avg=[rd(Cdouble, SEQ_RESULT_NODAL + 8 * (i - 1)) for i in 1:nb],
            comp_importance=[rd(Cdouble, SEQ_RESULT_IMP + 8 * (i - 1)) for i in 1:nc],
            total_loss_hours=acc.n_fail, n_lp=acc.n, elapsed_time=rd(Cdouble, SEQ_RESULT_TAIL), kernel_seconds=rd(Cdouble, SEQ_RESULT_TAIL + 8))
end

mutable struct Hl1Acc
    n::Int64; sum_lole::Cdouble; sum_eue::Cdouble; sum_lole2::Cdouble; sum_eue2::Cdouble
    Hl1Acc() = new()
end

"run_non_sequential_mc (PowerSystemAdequacy.jl:169-208): one fleet state per iteration swept over the hourly load curve."
function run_non_sequential_mc(eng::Engine, capacity::Vector{Float64}, for_rate::Vector{Float64}, hourly_load::Vector{Float64},
                               n_iterations::Integer; seed::Integer=1)
    check(ccall((:relmc_hl1_load, LIB), Int32, (Ptr{Cvoid}, Int32, Ptr{Cdouble}, Ptr{Cdouble}, Int32, Ptr{Cdouble}),
                eng.h, length(capacity), capacity, for_rate, length(hourly_load), hourly_load), eng.h, "relmc_hl1_load")
    acc = Hl1Acc()
    check(ccall((:relmc_hl1_nsq, LIB), Int32, (Ptr{Cvoid}, UInt64, UInt64, Int64, Ref{Hl1Acc}, Ptr{Cdouble}, Ptr{Cdouble}),
                eng.h, seed, 0, n_iterations, acc, C_NULL, C_NULL), eng.h, "relmc_hl1_nsq")
    return (lole_hours_yr=acc.sum_lole / acc.n, eue_mwh_yr=acc.sum_eue / acc.n)
end

# Layout table of the plain-C structs this file mirrors: tests/test_c_abi.py compiles a C program printing sizeof / offsetof of
# include/relmc.h's structs and compares with these numbers and with the ctypes mirror, so drift in either mirror is caught
# without a Julia installation.  (name, sizeof, [(field, offset) ...])
const LAYOUT = [
    ("relmc_case_desc", 128, [("base_mva", 0), ("nb", 8), ("ref_bus", 24), ("bus_pd", 32), ("always_up", 112), ("total_load", 120)]),
    ("relmc_solver_opts", 88, [("singular_policy", 0), ("max_it", 4), ("feastol", 8), ("max_stepsize", 72), ("screen", 80)]),
    ("relmc_acc", 8 * (7 + MAX_COMP + 2 + MAX_BUS), [("n", 0), ("comp_fail", 48), ("n_screened", 48 + 8 * MAX_COMP), ("sum_dns", 56 + 8 * MAX_COMP), ("sum_nodal", 72 + 8 * MAX_COMP)]),
    ("relmc_indices", 8 * (7 + MAX_BUS + MAX_COMP), [("n", 0), ("edns", 8), ("nodal_eens", 56), ("comp_importance", 56 + 8 * MAX_BUS)]),
    ("relmc_db_stats", 32, [("rows", 0), ("samples", 8), ("new_rows", 16), ("batch_distinct", 24)]),
    ("relmc_seq_year", 32, [("ens", 0), ("dlc", 8), ("nlc", 16), ("n_contingency", 24)]),
    ("relmc_hl1_acc", 40, [("n", 0), ("sum_lole", 8), ("sum_eue2", 32)]),
    ("relmc_nsq_opts", 176, [("beta_limit", 0), ("max_samples", 8), ("batch", 16), ("seed", 24), ("hours_per_year", 32), ("solver", 40), ("history_cap", 128), ("beta_history", 136), ("plc_history", 160), ("distinct_states", 168)]),
    ("relmc_seq_opts", 152, [("cov_threshold", 0), ("max_years", 8), ("batch_years", 12), ("seed", 16), ("curtail_threshold", 24), ("solver", 32), ("years_cap", 120), ("results_year", 128), ("cum_cov", 144)]),
    ("relmc_seq_result", SEQ_RESULT_BYTES, [("final_year", 0), ("converged", 4), ("eens", 8), ("plc", 40), ("n_contingency", 48), ("acc", SEQ_RESULT_ACC), ("nodal_eens_avg", SEQ_RESULT_NODAL), ("comp_importance", SEQ_RESULT_IMP), ("wall_seconds", SEQ_RESULT_TAIL), ("kernel_seconds", SEQ_RESULT_TAIL + 8)]),
    ("relmc_nsq_result", NSQ_RESULT_BYTES, [("acc", 0), ("idx", NSQ_RESULT_IDX), ("checkpoints", NSQ_RESULT_TAIL), ("converged", NSQ_RESULT_TAIL + 8), ("wall_seconds", NSQ_RESULT_TAIL + 16), ("kernel_seconds", NSQ_RESULT_TAIL + 24), ("batches", NSQ_RESULT_TAIL + 32)]),
]

end # module
