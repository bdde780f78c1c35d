# KatanaHIP.jl -- the reference-side binding of libkatana_hip.so (include/katana_hip.h).
#
# What a Katana.jl maintainer would add next to src/model.jl to put the MI355X engine behind the
# unchanged plugin surface (KatanaSolver / MathProgBase.NonlinearModel / loadproblem! / optimize! /
# status / getobjval / getsolution ...; src/solver.jl:34-46, src/model.jl:63-65,81-86,219,326-343).
# Julia 0.6 syntax, like the reference (REQUIRE:1).  Julia is not part of the build image of this
# repository, so this file is NOT executed by the test-suite; what IS tested (tests/test_julia_shim.py)
# is that its opcode table, struct mirrors and symbol names agree with include/katana_hip.h and with the
# Python binding (katana.jl_amd/_lib.py, expr.py), which is exercised on the GPU.
#
#     include("KatanaHIP.jl")                       # inside module Katana, after solver.jl (src/Katana.jl:21)
#
# With that one line the methods at the end of this file replace the two model factories of the reference
# (src/model.jl:63-65, src/solver.jl:46) and `KatanaHipSeparator()` becomes available as a `separator=` keyword of
# KatanaSolver (src/solver.jl:34-43) for users who keep the reference's own loop and LP solver.
module KatanaHIP

using MathProgBase
using JuMP
import ..KatanaSolver, ..AbstractKatanaSeparator, ..EpigraphNLPEvaluator          # src/solver.jl:6, src/separators.jl:8, src/nlpeval.jl:6
import ..initialize!, ..precompute!, ..gencut, ..isconstrsat                      # the separator API, src/separators.jl:23-53

export KatanaHipSeparator

const LIB = get(ENV, "KATANA_HIP_LIB", joinpath(dirname(@__FILE__), "..", "libkatana_hip.so"))

# ---- include/katana_hip.h: constants -------------------------------------------------------------------
const KTN_OP_CONST = Int32(0)
const KTN_OP_VAR   = Int32(1)
const KTN_OP_ADD   = Int32(2)
const KTN_OP_SUB   = Int32(3)
const KTN_OP_MUL   = Int32(4)
const KTN_OP_DIV   = Int32(5)
const KTN_OP_NEG   = Int32(6)
const KTN_OP_POWC  = Int32(7)
const KTN_OP_EXP   = Int32(8)
const KTN_OP_LOG   = Int32(9)
const KTN_OP_SQRT  = Int32(10)
const KTN_OP_SIN   = Int32(11)
const KTN_OP_COS   = Int32(12)
const KTN_ROW_SEP, KTN_ROW_TAPE, KTN_ROW_HOST = UInt8(0), UInt8(1), UInt8(2)
const STATUS = [:None, :Optimal, :Unbounded, :Infeasible, :UserLimit, :Error]     # KTN_STATUS_* + 1

# :call heads of a MathProgBase expression graph the device tape interpreter evaluates
const BINARY_OPS = Dict{Symbol,Int32}(:+ => KTN_OP_ADD, :- => KTN_OP_SUB, :* => KTN_OP_MUL, :/ => KTN_OP_DIV)
const UNARY_OPS  = Dict{Symbol,Int32}(:exp => KTN_OP_EXP, :log => KTN_OP_LOG, :sqrt => KTN_OP_SQRT,
                                      :sin => KTN_OP_SIN, :cos => KTN_OP_COS)

# ---- include/katana_hip.h: struct mirrors (field order and widths as in the header) ----------------------
struct KtnParams
    f_tol::Cdouble; cut_coef_rng::Cdouble; log_level::Int32; iter_cap::Int32; obj_eps::Cdouble
    vis_data::Int32; device::Int32; lp_max_iter::Int32; lp_check_every::Int32; lp_ruiz_iters::Int32
    lp_tol_scale::Cdouble; lp_tol_floor::Cdouble; lp_tol_cap::Cdouble; lp_gap_floor::Cdouble; lp_gap_cap::Cdouble
    lp_dual_inherit::Int32; profile::Int32
    purge_age::Int32; purge_margin::Cdouble; purge_min_frac::Cdouble; purge_min_rows::Int64
    lp_dense_after::Int32; cut_cap_factor::Cdouble; cut_cap_min::Int64; lp_stag_factor::Cdouble
    lp_ruiz_warm::Int32; lp_tiled_nnz::Int64; lp_near_check::Int32; dedupe_eps::Cdouble; polish_factor::Cdouble; polish_max_var::Int32; polish_max_iter::Int32
    epi_shift::Int32; obj_cert_tol::Cdouble; lp_mid_max_var::Int32
end

struct KtnNlpDesc
    num_var::Int64; num_constr::Int64
    rowptr::Ptr{Int64}; col::Ptr{Int32}
    row_kind::Ptr{UInt8}; row_linear::Ptr{UInt8}; rconst::Ptr{Cdouble}
    atom_kind::Ptr{UInt8}; p0::Ptr{Cdouble}; p1::Ptr{Cdouble}
    tape_ptr::Ptr{Int64}; tape_op::Ptr{Int32}; tape_arg::Ptr{Cdouble}
    obj_linear::Int32; obj_kind::Int32; obj_nnz::Int64
    obj_col::Ptr{Int32}; obj_atom_kind::Ptr{UInt8}; obj_p0::Ptr{Cdouble}; obj_p1::Ptr{Cdouble}
    obj_const::Cdouble; obj_tape_len::Int64; obj_tape_op::Ptr{Int32}; obj_tape_arg::Ptr{Cdouble}
    eval_rows::Ptr{Void}; eval_obj::Ptr{Void}; eval_user::Ptr{Void}
end

# the arrays a KtnNlpDesc points into; kept in the model so that they outlive every ccall
mutable struct NlpArrays
    rowptr::Vector{Int64}; col::Vector{Int32}; perm::Vector{Int}
    row_kind::Vector{UInt8}; row_linear::Vector{UInt8}; rconst::Vector{Cdouble}
    tape_ptr::Vector{Int64}; tape_op::Vector{Int32}; tape_arg::Vector{Cdouble}
    obj_tape_op::Vector{Int32}; obj_tape_arg::Vector{Cdouble}
end

function __init__()
    # layout check against the library, as katana.jl_amd/_lib.py does at load time
    sizeof(KtnParams) == ccall((:ktn_sizeof_params, LIB), Int64, ()) ||
        error("KtnParams does not mirror ktn_params of $LIB: rebuild the library or update the binding")
    sizeof(KtnNlpDesc) == ccall((:ktn_sizeof_nlp_desc, LIB), Int64, ()) ||
        error("KtnNlpDesc does not mirror ktn_nlp_desc of $LIB")
end

# ---- Expr -> postfix tape ---------------------------------------------------------------------------------
# Input: the body of MathProgBase.constr_expr(d, i) / obj_expr(d) (:ExprGraph): numbers, x[j] (Expr(:ref, :x, j)),
# and :call nodes.  n-ary + and * are folded left to right; a - with one argument is negation; ^ needs a
# constant exponent (the tape's KTN_OP_POWC).  Anything else is the reference's "Unsupported feature"
# (src/nlpeval.jl:28).
function emit_tape!(ops::Vector{Int32}, args::Vector{Cdouble}, ex)
    if isa(ex, Real)
        push!(ops, KTN_OP_CONST); push!(args, Float64(ex))
    elseif isa(ex, Expr) && ex.head == :ref
        push!(ops, KTN_OP_VAR); push!(args, Float64(ex.args[2] - 1))            # 0-based column
    elseif isa(ex, Expr) && ex.head == :call
        f = ex.args[1]
        a = ex.args[2:end]
        if f == :- && length(a) == 1
            emit_tape!(ops, args, a[1]); push!(ops, KTN_OP_NEG); push!(args, 0.0)
        elseif f == :+ && length(a) == 1
            emit_tape!(ops, args, a[1])
        elseif (f == :+ || f == :*) && length(a) >= 2
            emit_tape!(ops, args, a[1])
            for t in a[2:end]
                emit_tape!(ops, args, t); push!(ops, BINARY_OPS[f]); push!(args, 0.0)
            end
        elseif (f == :- || f == :/) && length(a) == 2
            emit_tape!(ops, args, a[1]); emit_tape!(ops, args, a[2]); push!(ops, BINARY_OPS[f]); push!(args, 0.0)
        elseif f == :^ && length(a) == 2 && isa(a[2], Real)
            emit_tape!(ops, args, a[1]); push!(ops, KTN_OP_POWC); push!(args, Float64(a[2]))
        elseif haskey(UNARY_OPS, f) && length(a) == 1
            emit_tape!(ops, args, a[1]); push!(ops, UNARY_OPS[f]); push!(args, 0.0)
        else
            error("Unsupported feature $f")
        end
    else
        error("Unsupported feature $(ex)")
    end
end

# body g(x) of a constraint expression: `g <= ub`, `g >= lb`, `g == c` or `lb <= g <= ub`
function constraint_body(ex::Expr)
    if ex.head == :comparison                      # lb <= g <= ub
        return ex.args[3]
    elseif ex.head == :call && ex.args[1] in (:<=, :>=, :(==))
        return ex.args[2]
    end
    error("Unsupported feature $(ex.head)")
end

# ---- d::AbstractNLPEvaluator -> ktn_nlp_desc ------------------------------------------------------------
function build_ktn_nlp_desc(d::MathProgBase.AbstractNLPEvaluator, num_var::Int, num_constr::Int)
    # Jacobian structure COO -> CSR exactly as initialize! does (src/separators.jl:92-104): row by row, the
    # entries of a row in COO order; perm[k] = COO index of CSR entry k (what sp_col_inds holds)
    sp_rows, sp_cols = MathProgBase.jac_structure(d)
    N = length(sp_rows)
    counts = zeros(Int64, num_constr)
    for ind in 1:N
        counts[sp_rows[ind]] += 1
    end
    rowptr = zeros(Int64, num_constr + 1)
    for i in 1:num_constr
        rowptr[i + 1] = rowptr[i] + counts[i]
    end
    fill_pos = copy(rowptr[1:num_constr])
    col = zeros(Int32, N)
    perm = zeros(Int, N)
    for ind in 1:N
        i = sp_rows[ind]
        fill_pos[i] += 1
        col[fill_pos[i]] = Int32(sp_cols[ind] - 1)
        perm[fill_pos[i]] = ind
    end
    # one tape per row (linear rows too: their tangent at the origin is the row itself, src/model.jl:115-118)
    tape_ptr = zeros(Int64, num_constr + 1)
    tape_op = Int32[]
    tape_arg = Cdouble[]
    row_linear = zeros(UInt8, num_constr)
    for i in 1:num_constr
        emit_tape!(tape_op, tape_arg, constraint_body(MathProgBase.constr_expr(d, i)))
        tape_ptr[i + 1] = length(tape_op)
        row_linear[i] = MathProgBase.isconstrlinear(d, i) ? UInt8(1) : UInt8(0)           # src/model.jl:116
    end
    obj_op = Int32[]
    obj_arg = Cdouble[]
    emit_tape!(obj_op, obj_arg, MathProgBase.obj_expr(d))
    arrs = NlpArrays(rowptr, col, perm, fill(KTN_ROW_TAPE, num_constr), row_linear, zeros(Cdouble, num_constr),
                     tape_ptr, tape_op, tape_arg, obj_op, obj_arg)
    desc = KtnNlpDesc(num_var, num_constr, pointer(arrs.rowptr), pointer(arrs.col),
                      pointer(arrs.row_kind), pointer(arrs.row_linear), pointer(arrs.rconst),
                      C_NULL, C_NULL, C_NULL,
                      pointer(arrs.tape_ptr), pointer(arrs.tape_op), pointer(arrs.tape_arg),
                      MathProgBase.isobjlinear(d) ? Int32(1) : Int32(0),                    # src/model.jl:125
                      Int32(KTN_ROW_TAPE), 0, C_NULL, C_NULL, C_NULL, C_NULL,
                      0.0, length(arrs.obj_tape_op), pointer(arrs.obj_tape_op), pointer(arrs.obj_tape_arg),
                      C_NULL, C_NULL, C_NULL)
    return desc, arrs
end

# ---- evaluators without :ExprGraph: the host-callback description (KTN_ROW_HOST) ---------------------------
# An evaluator that cannot hand over expressions (MathProgBase.features_available(d) lacks :ExprGraph -- the reference's own
# EpigraphNLPEvaluator is one, src/nlpeval.jl:23) is consumed the way the reference consumes it: eval_g + eval_jac_g
# (src/separators.jl:112-113) and eval_f + eval_grad_f (src/nlpeval.jl:35-41), called back from the engine once per
# sweep on the thread that is inside the ccall.  Constraint checks, cuts and the LP stay on the device.
mutable struct HostEval
    d::MathProgBase.AbstractNLPEvaluator
    num_var::Int
    num_constr::Int
    perm::Vector{Int}              # CSR entry k = COO entry perm[k]  (src/separators.jl:96-100)
    jcoo::Vector{Float64}
end

# ktn_eval_rows_cb (include/katana_hip.h): g[num_constr], jac in the CSR order of the description
function eval_rows_cb(user::Ptr{Void}, xp::Ptr{Cdouble}, gp::Ptr{Cdouble}, jp::Ptr{Cdouble})::Cint
    try
        h = unsafe_pointer_to_objref(user)::HostEval
        x = copy(unsafe_wrap(Array, xp, h.num_var))
        g = unsafe_wrap(Array, gp, h.num_constr)
        MathProgBase.eval_jac_g(h.d, h.jcoo, x)                                 # src/separators.jl:112
        MathProgBase.eval_g(h.d, g, x)                                          # src/separators.jl:113
        jac = unsafe_wrap(Array, jp, length(h.perm))
        for k in 1:length(h.perm)
            jac[k] = h.jcoo[h.perm[k]]
        end
        return Cint(0)
    catch
        return Cint(1)                                                          # never unwind through the C ABI: KTN_E_CALLBACK
    end
end

# ktn_eval_obj_cb: f and the dense gradient
function eval_obj_cb(user::Ptr{Void}, xp::Ptr{Cdouble}, fp::Ptr{Cdouble}, gradp::Ptr{Cdouble})::Cint
    try
        h = unsafe_pointer_to_objref(user)::HostEval
        x = copy(unsafe_wrap(Array, xp, h.num_var))
        unsafe_store!(fp, MathProgBase.eval_f(h.d, x))                          # src/nlpeval.jl:35
        MathProgBase.eval_grad_f(h.d, unsafe_wrap(Array, gradp, h.num_var), x)  # src/nlpeval.jl:36-39
        return Cint(0)
    catch
        return Cint(1)
    end
end

function build_host_nlp_desc(d::MathProgBase.AbstractNLPEvaluator, num_var::Int, num_constr::Int)
    MathProgBase.initialize(d, [:Grad, :Jac])                                   # src/separators.jl:88
    sp_rows, sp_cols = MathProgBase.jac_structure(d)
    N = length(sp_rows)
    counts = zeros(Int64, num_constr)
    for ind in 1:N
        counts[sp_rows[ind]] += 1
    end
    rowptr = zeros(Int64, num_constr + 1)
    for i in 1:num_constr
        rowptr[i + 1] = rowptr[i] + counts[i]
    end
    fill_pos = copy(rowptr[1:num_constr])
    col = zeros(Int32, N)
    perm = zeros(Int, N)
    for ind in 1:N
        i = sp_rows[ind]
        fill_pos[i] += 1
        col[fill_pos[i]] = Int32(sp_cols[ind] - 1)
        perm[fill_pos[i]] = ind
    end
    row_linear = UInt8[MathProgBase.isconstrlinear(d, i) ? 1 : 0 for i in 1:num_constr]     # src/model.jl:116
    arrs = NlpArrays(rowptr, col, perm, fill(KTN_ROW_HOST, num_constr), row_linear, zeros(Cdouble, num_constr),
                     zeros(Int64, num_constr + 1), Int32[], Cdouble[], Int32[], Cdouble[])
    host = HostEval(d, num_var, num_constr, perm, zeros(N))
    rows_c = cfunction(eval_rows_cb, Cint, (Ptr{Void}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}))
    obj_c = cfunction(eval_obj_cb, Cint, (Ptr{Void}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}))
    desc = KtnNlpDesc(num_var, num_constr, pointer(arrs.rowptr), pointer(arrs.col),
                      pointer(arrs.row_kind), pointer(arrs.row_linear), pointer(arrs.rconst),
                      C_NULL, C_NULL, C_NULL, C_NULL, C_NULL, C_NULL,
                      MathProgBase.isobjlinear(d) ? Int32(1) : Int32(0), Int32(KTN_ROW_HOST), 0, C_NULL, C_NULL, C_NULL, C_NULL,
                      0.0, 0, C_NULL, C_NULL,
                      rows_c, obj_c, pointer_from_objref(host))
    return desc, arrs, host
end

# expressions when the evaluator offers them, callbacks otherwise
function describe(d::MathProgBase.AbstractNLPEvaluator, num_var::Int, num_constr::Int)
    if :ExprGraph in MathProgBase.features_available(d)
        MathProgBase.initialize(d, [:Grad, :Jac, :ExprGraph])
        desc, arrs = build_ktn_nlp_desc(d, num_var, num_constr)
        return desc, arrs, nothing
    end
    return build_host_nlp_desc(d, num_var, num_constr)
end

# ---- the model type behind MathProgBase.NonlinearModel(s::KatanaSolver) ----------------------------------
mutable struct KatanaHipModel <: MathProgBase.AbstractNonlinearModel
    handle::Ptr{Void}
    params::KtnParams
    arrays::Union{NlpArrays,Void}
    host::Union{HostEval,Void}            # kept alive while the engine holds its address (eval_user)
end

function check(m::KatanaHipModel, code)
    code < 0 && error(unsafe_string(ccall((:ktn_last_error, LIB), Cstring, (Ptr{Void},), m.handle)))
    code
end

# `s` is the reference's KatanaSolver (src/solver.jl:6-10): lp_solver is ignored (the LP runs on the GPU),
# features and model_params are honoured (src/model.jl:46-58)
function KatanaHipModel(s)
    p = Ref{KtnParams}()
    ccall((:ktn_default_params, LIB), Void, (Ref{KtnParams},), p)
    d, mp = p[], s.model_params
    vis = Int32(0)
    for f in s.features
        f == :VisData || error("type KatanaFeatures has no field $f")                       # src/model.jl:50-52
        vis = Int32(1)
    end
    prm = KtnParams(mp.f_tol, mp.cut_coef_rng, Int32(mp.log_level), Int32(mp.iter_cap), mp.obj_eps, vis, d.device,
                    d.lp_max_iter, d.lp_check_every, d.lp_ruiz_iters, d.lp_tol_scale, d.lp_tol_floor, d.lp_tol_cap,
                    d.lp_gap_floor, d.lp_gap_cap, d.lp_dual_inherit, d.profile, d.purge_age, d.purge_margin,
                    d.purge_min_frac, d.purge_min_rows, d.lp_dense_after, d.cut_cap_factor, d.cut_cap_min,
                    d.lp_stag_factor, d.lp_ruiz_warm, d.lp_tiled_nnz, d.lp_near_check, d.dedupe_eps, d.polish_factor, d.polish_max_var, d.polish_max_iter,
                    d.epi_shift, d.obj_cert_tol, d.lp_mid_max_var)
    h = Ref{Ptr{Void}}(C_NULL)
    code = ccall((:ktn_create, LIB), Cint, (Ref{KtnParams}, Ref{Ptr{Void}}), Ref(prm), h)
    code == 0 || error("ktn_create failed ($code): no MI355X visible? the engine has no CPU path")
    m = KatanaHipModel(h[], prm, nothing, nothing)
    finalizer(m, x -> ccall((:ktn_destroy, LIB), Void, (Ptr{Void},), x.handle))
    m
end

function MathProgBase.loadproblem!(m::KatanaHipModel, num_var::Int, num_constr::Int,
        l_var::Vector{Float64}, u_var::Vector{Float64}, l_constr::Vector{Float64}, u_constr::Vector{Float64},
        sense::Symbol, d::MathProgBase.AbstractNLPEvaluator)                                  # src/model.jl:81-86
    desc, arrs, host = describe(d, num_var, num_constr)       # tapes from :ExprGraph, else eval_g / eval_jac_g callbacks
    m.arrays = arrs
    m.host = host
    check(m, ccall((:ktn_loadproblem, LIB), Cint,
        (Ptr{Void}, Int64, Int64, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Int32, Ref{KtnNlpDesc}),
        m.handle, num_var, num_constr, l_var, u_var, l_constr, u_constr, sense == :Max ? 1 : 0, Ref(desc)))
end

MathProgBase.optimize!(m::KatanaHipModel) =
    STATUS[check(m, ccall((:ktn_optimize, LIB), Cint, (Ptr{Void},), m.handle)) + 1]           # src/model.jl:219
MathProgBase.status(m::KatanaHipModel) = STATUS[ccall((:ktn_get_status, LIB), Cint, (Ptr{Void},), m.handle) + 1]
MathProgBase.getobjval(m::KatanaHipModel) = ccall((:ktn_get_objval, LIB), Cdouble, (Ptr{Void},), m.handle)
function MathProgBase.getsolution(m::KatanaHipModel)              # incl. the epigraph variable, src/model.jl:340-341
    n = ccall((:ktn_get_num_var, LIB), Int64, (Ptr{Void},), m.handle)
    x = Vector{Float64}(n)
    check(m, ccall((:ktn_get_solution, LIB), Cint, (Ptr{Void}, Ptr{Cdouble}, Int64), m.handle, x, n))
    x
end
MathProgBase.getsolvetime(m::KatanaHipModel) = ccall((:ktn_get_solvetime, LIB), Cdouble, (Ptr{Void},), m.handle)
MathProgBase.setwarmstart!(m::KatanaHipModel, x) = fill(0.0, length(x))                      # src/model.jl:335
numiters(m::KatanaHipModel) = ccall((:ktn_numiters, LIB), Int64, (Ptr{Void},), m.handle)    # src/model.jl:326
numcuts(m::KatanaHipModel)  = ccall((:ktn_numcuts,  LIB), Int64, (Ptr{Void},), m.handle)    # src/model.jl:333

# ---- the two model factories of the reference, now returning the HIP model -------------------------------------
MathProgBase.NonlinearModel(s::KatanaSolver) = KatanaHipModel(s)                                           # src/model.jl:63-65
MathProgBase.LinearQuadraticModel(s::KatanaSolver) = MathProgBase.NonlinearToLPQPBridge(MathProgBase.NonlinearModel(s))   # src/solver.jl:46

# ---- KatanaHipSeparator <: AbstractKatanaSeparator (src/separators.jl:8,23-53,58-120) ----------------------------
# For users who keep the reference's loop and LP solver (KatanaSolver(GLPKSolverLP(), separator = KatanaHipSeparator())):
# the separator plugin API served by the device.  initialize! loads the evaluator into an engine handle of its own;
# precompute! is ONE evaluation sweep of all rows on the GPU (ktn_sep_precompute); isconstrsat / gencut then answer per
# row from the device-resident g and J, like KatanaFirstOrderSeparator does from its host arrays.
mutable struct KatanaHipSeparator <: AbstractKatanaSeparator
    model::Union{KatanaHipModel,Void}
    linear_model::Union{JuMP.Model,Void}
    num_var::Int
    num_constr::Int
    cols::Vector{Int32}
    coefs::Vector{Float64}
    KatanaHipSeparator() = new(nothing, nothing, 0, 0, Int32[], Float64[])
end

function initialize!(sep::KatanaHipSeparator, linear_model::JuMP.Model, num_var::Int, num_constr::Int,
                     oracle::MathProgBase.AbstractNLPEvaluator)                                            # src/separators.jl:81-107
    sep.linear_model = linear_model
    sep.num_var, sep.num_constr = num_var, num_constr
    # the reference hands over its EpigraphNLPEvaluator when the objective is nonlinear (src/model.jl:166,171-172): the engine
    # appends the row f(x) - t itself (src/nlpeval.jl:42-63), so the wrapped evaluator is what gets loaded
    inner, nv, nc = isa(oracle, EpigraphNLPEvaluator) ? (oracle.nlpeval, num_var - 1, num_constr - 1) : (oracle, num_var, num_constr)
    p = Ref{KtnParams}()
    ccall((:ktn_default_params, LIB), Void, (Ref{KtnParams},), p)
    h = Ref{Ptr{Void}}(C_NULL)
    code = ccall((:ktn_create, LIB), Cint, (Ref{KtnParams}, Ref{Ptr{Void}}), p, h)
    code == 0 || error("ktn_create failed ($code): no MI355X visible? the engine has no CPU path")
    m = KatanaHipModel(h[], p[], nothing, nothing)
    finalizer(m, x -> ccall((:ktn_destroy, LIB), Void, (Ptr{Void},), x.handle))
    # bounds and sense play no part in the separator API (isconstrsat / gencut receive them per call): free variables,
    # one-sided rows
    MathProgBase.loadproblem!(m, nv, nc, fill(-Inf, nv), fill(Inf, nv), fill(-Inf, nc), zeros(nc), :Min, inner)
    sep.model = m
    kmax = ccall((:ktn_sep_jac_nnz, LIB), Int64, (Ptr{Void},), m.handle)
    sep.cols, sep.coefs = zeros(Int32, kmax), zeros(kmax)
end

function precompute!(sep::KatanaHipSeparator, xstar)                                                       # src/separators.jl:111-116
    x = convert(Vector{Float64}, xstar)
    check(sep.model, ccall((:ktn_sep_precompute, LIB), Cint, (Ptr{Void}, Ptr{Cdouble}, Int64), sep.model.handle, x, length(x)))
end

isconstrsat(sep::KatanaHipSeparator, i, lb, ub, f_tol) =                                                   # src/separators.jl:120
    check(sep.model, ccall((:ktn_sep_isconstrsat, LIB), Cint, (Ptr{Void}, Int64, Cdouble, Cdouble, Cdouble),
                           sep.model.handle, i - 1, lb, ub, f_tol)) == 1

function gencut(sep::KatanaHipSeparator, xstar, bounds, i)                                                 # src/separators.jl:118, src/algorithms.jl:3-18
    nnz = Ref{Int64}(length(sep.cols))
    b = Ref{Cdouble}(0.0)
    check(sep.model, ccall((:ktn_sep_gencut, LIB), Cint, (Ptr{Void}, Int64, Ptr{Int32}, Ptr{Cdouble}, Ref{Int64}, Ref{Cdouble}),
                           sep.model.handle, i - 1, sep.cols, sep.coefs, nnz, b))
    v = JuMP.Variable[JuMP.Variable(sep.linear_model, Int(sep.cols[k]) + 1) for k in 1:nnz[]]
    JuMP.AffExpr(v, sep.coefs[1:nnz[]], b[])
end

# ---- src/util.jl:16-36 ------------------------------------------------------------------------------------
function getKatanaCuts(m::KatanaHipModel)
    M = ccall((:ktn_lp_num_rows, LIB), Int64, (Ptr{Void},), m.handle)
    nnz = ccall((:ktn_lp_nnz, LIB), Int64, (Ptr{Void},), m.handle)
    N = ccall((:ktn_get_num_var, LIB), Int64, (Ptr{Void},), m.handle)
    rowptr, col, val = zeros(Int64, M + 1), zeros(Int32, max(nnz, 1)), zeros(max(nnz, 1))
    lo, hi = zeros(max(M, 1)), zeros(max(M, 1))
    check(m, ccall((:ktn_lp_get_rows, LIB), Cint, (Ptr{Void}, Ptr{Int64}, Ptr{Int32}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}),
                   m.handle, rowptr, col, val, lo, hi))
    A = zeros(M, N + 2)
    for i in 1:M
        for e in (rowptr[i] + 1):rowptr[i + 1]
            A[i, col[e] + 1] += val[e]
        end
        if isfinite(hi[i]) && !isfinite(lo[i])
            A[i, N + 1], A[i, N + 2] = hi[i], -1
        elseif isfinite(lo[i]) && !isfinite(hi[i])
            A[i, N + 1], A[i, N + 2] = lo[i], 1
        else
            error("range or free row: not an inequality")                                    # src/util.jl:27-29
        end
    end
    A
end
function getKatanaSols(m::KatanaHipModel)
    k = ccall((:ktn_num_lp_sols, LIB), Int64, (Ptr{Void},), m.handle)
    n = ccall((:ktn_get_num_var, LIB), Int64, (Ptr{Void},), m.handle)
    sols = Vector{Float64}[]
    for i in 0:(k - 1)
        x = Vector{Float64}(n)
        check(m, ccall((:ktn_get_lp_sol, LIB), Cint, (Ptr{Void}, Int64, Ptr{Cdouble}, Int64), m.handle, i, x, n))
        push!(sols, x)
    end
    sols
end

end # module
