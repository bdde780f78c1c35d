/* katana_hip.h -- C ABI of the MI355X-native Extended-Cutting-Plane engine.
 *
 * Drop-in boundary for the ONE hot path of lanl-ansi/Katana.jl:
 * the MathProgBase / KatanaSolver plugin surface and the separator API, re-expressed
 * as plain C so that a Julia `ccall` shim, the Python ctypes host
 * mirror (katana.jl_amd/solver.py) or any other FFI can bind it.  No torch types,
 * no C++ types: plain pointers and sizes.  All indices are 0-based.
 *
 * Every entry point names the reference interface it replaces (file:line relative
 * to the reference tree).  Functions return 0 on success and a negative KTN_E_* code
 * on failure (never throw, never abort); ktn_last_error() gives the message.
 * Host buffers are copied at the call (caller keeps ownership); the library owns all
 * device memory until ktn_destroy.  One handle <-> one host thread at a time; every
 * handle owns its HIP stream.  The library REQUIRES a gfx950 device: there is no CPU
 * fallback, ktn_create fails with KTN_E_NODEVICE when none is visible.
 */
#ifndef KATANA_HIP_H
#define KATANA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KTN_ABI_VERSION 1

/* ---- error codes ------------------------------------------------------------ */
#define KTN_OK            0
#define KTN_E_INVALID    -1   /* bad argument / call order                          */
#define KTN_E_NODEVICE   -2   /* no HIP device (the product has no CPU path)        */
#define KTN_E_HIP        -3   /* a HIP runtime call failed                           */
#define KTN_E_NOMEM      -4
#define KTN_E_UNSUPPORTED -5  /* e.g. unknown tape opcode ("Unsupported feature",
                                 src/nlpeval.jl:28)                                  */
#define KTN_E_CALLBACK   -6   /* the caller's evaluator callback reported a failure  */

/* ---- status vocabulary: MathProgBase.status(m) symbols ------------------------
 * :None src/model.jl:44, :Optimal, :Unbounded :246, LP pass-through e.g. :Infeasible
 * :261-263, :UserLimit :314, :Error :71 */
#define KTN_STATUS_NONE       0
#define KTN_STATUS_OPTIMAL    1
#define KTN_STATUS_UNBOUNDED  2
#define KTN_STATUS_INFEASIBLE 3
#define KTN_STATUS_USERLIMIT  4
#define KTN_STATUS_ERROR      5

#define KTN_MIN 0   /* sense :Min */
#define KTN_MAX 1   /* sense :Max */

/* ---- how a constraint row (or the objective) is evaluated on the device --------
 * Replaces the closures behind MathProgBase.eval_g / eval_jac_g / eval_f /
 * eval_grad_f (call sites src/separators.jl:112-113, src/nlpeval.jl:35-63). */
#define KTN_ROW_SEP  0   /* separable: g_i(x) = sum_e atom_e(x[col_e]) + rconst_i    */
#define KTN_ROW_TAPE 1   /* general: postfix expression tape, reverse-mode AD        */
#define KTN_ROW_HOST 2   /* fallback for evaluators that cannot hand over expressions (no :ExprGraph): values and
                            derivatives come from the caller's own eval_g / eval_jac_g / eval_f / eval_grad_f through
                            the callbacks below, once per sweep; isconstrsat, gencut, round_coefs, _addcut and the LP
                            still run on the device (SURVEY.md section 8b "Evaluator consumed")                     */

/* MathProgBase.eval_g + eval_jac_g (src/separators.jl:112-113) for the KTN_ROW_HOST rows: write g[i] and
 * jac[rowptr[i] .. rowptr[i+1]) (CSR order of ktn_nlp_desc) for every such row i; other entries are ignored.
 * x has num_var entries.  Return 0, or non-zero to make the running ktn_* call fail with KTN_E_CALLBACK. */
typedef int (*ktn_eval_rows_cb)(void* user, const double* x, double* g, double* jac);
/* MathProgBase.eval_f + eval_grad_f (src/nlpeval.jl:35-41) for a KTN_ROW_HOST objective: *f and the dense grad[num_var] */
typedef int (*ktn_eval_obj_cb)(void* user, const double* x, double* f, double* grad);

/* separable atoms, two f64 parameters per Jacobian entry */
#define KTN_ATOM_LIN    0   /* p0 * x                */
#define KTN_ATOM_QUAD   1   /* p0 * (x - p1)^2       */
#define KTN_ATOM_EXP    2   /* p0 * exp(p1 * x)      */
#define KTN_ATOM_NEGLOG 3   /* -p0 * log(x + p1)     */

/* postfix tape opcodes; `arg` is the constant for CONST/POWC and the 0-based
 * variable index (stored as a double) for VAR */
#define KTN_OP_CONST 0
#define KTN_OP_VAR   1
#define KTN_OP_ADD   2
#define KTN_OP_SUB   3
#define KTN_OP_MUL   4
#define KTN_OP_DIV   5
#define KTN_OP_NEG   6
#define KTN_OP_POWC  7   /* x ^ arg, arg constant */
#define KTN_OP_EXP   8
#define KTN_OP_LOG   9
#define KTN_OP_SQRT  10
#define KTN_OP_SIN   11
#define KTN_OP_COS   12

typedef struct ktn_handle_s* ktn_handle;

/* KatanaSolver keyword arguments and defaults, src/solver.jl:34-43 +
 * KatanaModelParams src/Katana.jl:12-19.  `lp_solver` has no counterpart: the LP is
 * solved on the GPU by the built-in first-order method; the lp_* fields tune it. */
typedef struct {
    double  f_tol;          /* 1e-6   feasibility tolerance                         */
    double  cut_coef_rng;   /* 1e9    max coefficient range per cut                 */
    int32_t log_level;      /* 10     print every log_level iterations, 0 = silent  */
    int32_t iter_cap;       /* 10000  iteration cap                                 */
    double  obj_eps;        /* -1.0   objective-delta stop (disabled when < 0)      */
    int32_t vis_data;       /* 0      feature :VisData (src/model.jl:1-4,29-31)     */
    int32_t device;         /* -1     HIP device ordinal; -1 = current device       */
    /* GPU LP (restarted reflected Halpern PDHG) */
    int32_t lp_max_iter;    /* 10000000 PDHG iterations per LP solve                */
    int32_t lp_check_every; /* 64      iterations between KKT checks                */
    int32_t lp_ruiz_iters;  /* 8   Ruiz (max-norm) passes before the Pock-Chambolle pass */
    double  lp_tol_scale;   /* 0.1     LP row tolerance = lp_tol_scale * max viol.  */
    double  lp_tol_floor;   /* 0.3     ... floored at lp_tol_floor * f_tol          */
    double  lp_tol_cap;     /* 10      ... capped                                   */
    double  lp_gap_floor;   /* 1e-7    relative duality-gap tolerance floor         */
    double  lp_gap_cap;     /* 1e-2                                                 */
    int32_t lp_dual_inherit;/* 1       new cut of NL row i inherits the dual of its
                                       previous cut (warm start)                    */
    int32_t profile;        /* 0       per-launch hipEvent timing of the hot kernels */
    /* cut-pool management (the reference never removes cuts, src/model.jl:215) */
    int32_t purge_age;      /* 2       drop a cut that had no multiplier and was slack at x* in this many
                                       consecutive LP solves; 0 = keep every cut like the reference      */
    double  purge_margin;   /* 1e-3    ... slack by more than purge_margin * max(1, |row bound|)          */
    double  purge_min_frac; /* 0.05    compact only when at least this fraction of the LP rows goes      */
    int64_t purge_min_rows; /* 2000    ... and only once the pool holds this many cuts: on small smooth
                                       problems (optimum on a curved face, e.g. test/2d.jl 107_01) dropping
                                       idle cuts makes Kelley's method cycle, so small pools are never purged */
    /* exact small-LP kernel (LPs of at most 32 columns; csrc/dense_lp.hpp) */
    int32_t lp_dense_after; /* 5000    when the first-order LP has not converged after this many iterations the
                                       LP is solved exactly by a dual active-set kernel (what the reference's
                                       simplex does on its small test models); 0 = never, < 0 = always exact */
    /* deepest-cut selection (the north star's "select deepest cuts"; the reference cuts every violated row, src/model.jl:272-283) */
    double  cut_cap_factor; /* 1.0     when more than max(cut_cap_factor * num_var, cut_cap_min) NL rows are violated, only that
                                       many of the deepest (largest g - ub / lb - g) get a cut in this iteration; the stop
                                       rule still needs EVERY row within f_tol.  0 = cut every violated row            */
    int64_t cut_cap_min;    /* 10000                                                                                    */
    double  lp_stag_factor; /* 300     primal-stagnation exit of the LP: rows feasible, primal objective flat over two checks,
                                       gap within lp_stag_factor * tolerance (the dual of a degenerate LP crawls long after
                                       the primal has converged), and a row violation that has stalled below 2x the row
                                       tolerance with everything else converged is accepted; 0 = only the full criteria      */
    int32_t lp_ruiz_warm;   /* 0       > 0: reuse the Ruiz equilibration of the previous LP solve for all but the appended rows and run
                                       only this many passes (measured on cfg3: the different scaling costs 1.8x the PDHG
                                       iterations, so the default stays lp_ruiz_iters passes from scratch)                    */
    int64_t lp_tiled_nnz;   /* 4000000 LPs with at least this many non-zeros run their SpMVs from tiled copies of the matrix
                                       (input vector staged through LDS in 64 KB blocks); 0 = never     */
    int32_t lp_near_check;  /* 7       once a check finds row violation, gap and dual residual within 4x their tolerances the next
                                       check comes after this many iterations instead of lp_check_every (a solve otherwise ends
                                       on average half a chunk after it converged: -17 % PDHG iterations on cfg3); 0 = off      */
    double  dedupe_eps;     /* 1e-6    at every purge pass, cuts of an NL row that agree with the row's newest cut within this relative
                                       tolerance (coefficients and bound, normalised by the largest coefficient) are dropped and
                                       their multipliers moved to the newest cut (the reference's TODO, src/model.jl:215); 0 = off */
    /* terminal refinement of small problems: once every NL row is within f_tol (the reference's stop rule, src/model.jl:257,273)
       the loop keeps cutting rows that are beyond polish_factor * f_tol, with the LP solved to the matching tolerance.  The
       reference's exact simplex vertices end Kelley's method far below f_tol on its small test models (its suite asserts the
       objective to 1e-6, test/3d.jl:124 to 1e-7); a first-order LP solution ends AT f_tol.  The point returned always
       satisfies the reference's rule; these passes are not counted in numiters (stat "polish_iters").                       */
    double  polish_factor;  /* 1e-3    0 or >= 1: off                                                                            */
    int32_t polish_max_var; /* 32      only for problems with at most this many LP columns (where the exact LP kernel applies) */
    int32_t polish_max_iter;/* 30      at most this many refinement passes                                                     */
    /* nonlinear objective (EpigraphNLPEvaluator, src/nlpeval.jl:42-63): every epigraph cut is dense and, close to the optimum, nearly
       parallel to every other one.  The first-order LP works on the exactly equivalent problem in which the epigraph variable
       is measured from the NEWEST cut, t = s + grad f(x_k)'x + b_k: the cuts' common part moves into the cost vector and the rows
       keep only their differences (csrc/kernels.hpp "epigraph reference shift").  The LP that getKatanaCuts exports is unchanged. */
    int32_t epi_shift;      /* 1       0 = solve the LP in the reference's own form                                              */
    /* objective certificate (problems with more than polish_max_var LP columns): at the point that meets the stop rule the engine
       evaluates  sum_i lambda_i (signed residual of NL row i) -- the convexity bound on  f* - objective  with the LP's duals as
       multipliers -- and keeps cutting below f_tol (at most polish_max_iter passes, not counted in numiters) while it exceeds
       obj_cert_tol * max(1, |objective|).  The reference's tests accept an objective within 1e-6 (test/runtests.jl:16-17); its exact
       simplex vertices end Kelley's method far below f_tol, a first-order LP ends AT f_tol times the multipliers.              */
    double  obj_cert_tol;   /* 1e-6    (refines while the sum exceeds half of it)  0 = stop at the reference's rule alone          */
    /* exact mid-size LP solver (csrc/mid_lp.hpp): LPs of 33 .. lp_mid_max_var columns are handed to a dual active-set method with
       the basis inverse in device memory when the first-order method stalls (same hand-over rule as lp_dense_after: the LPs of
       a cutting-plane run on a smooth-face optimum -- the reference's own regime, README.md:5 -- have many nearly parallel cuts
       active at once, which the reference's warm-started simplex re-solves in a handful of pivots, src/model.jl:259)            */
    int32_t lp_mid_max_var; /* 512     largest LP (columns) handed over; at most 4096 (n x n doubles of basis inverse); 0 = never.
                                       A pivot costs O(n^2) and a Kelley re-solve hundreds of pivots (the LP vertex jumps), so
                                       beyond a few hundred columns the hand-over stops paying (DESIGN.md section 5
                                       "Smooth-face optima")                                                                    */
} ktn_params;

/* NL-row blocks over several GPUs with a replicated LP: the exchange step of one cutting-plane round.
 * what = 0: the handle has just swept ITS block of NL rows; its new LP rows are the rows from `first_new_row` on.  The callback
 *           moves every rank's new rows into every rank's LP in rank order (ktn_lp_pack_rows_dev -> all-gather ->
 *           ktn_lp_truncate(first_new_row) -> ktn_lp_append_packed_dev per rank) and returns in scalars[0] the number of rows
 *           appended in total and in scalars[1..n) the MAXIMUM over the ranks of what it found there (largest violation, status
 *           flags, two values the loop's decisions are taken from);
 * what = 1: scalars[0..n) are replaced by their SUM over the ranks (the shares of the objective certificate).
 * Return 0, or non-zero to make the running ktn_* call fail with KTN_E_CALLBACK.  The callback may call ktn_lp_* on the handle. */
typedef int (*ktn_exchange_cb)(void* user, int32_t what, int64_t first_new_row, double* scalars, int32_t nscalars);

/* The device-evaluable statement of the NLP: replaces the
 * MathProgBase.AbstractNLPEvaluator `d` handed to loadproblem! (src/model.jl:86).
 * Jacobian structure is CSR over the num_constr constraints (what initialize! builds
 * from jac_structure, src/separators.jl:92-104). */
typedef struct {
    int64_t num_var;
    int64_t num_constr;
    /* Jacobian structure */
    const int64_t* rowptr;      /* [num_constr+1]                                    */
    const int32_t* col;         /* [nnz]                                             */
    /* per row */
    const uint8_t* row_kind;    /* [num_constr] KTN_ROW_*                            */
    const uint8_t* row_linear;  /* [num_constr] isconstrlinear(d,i), src/model.jl:116 */
    const double*  rconst;      /* [num_constr] constant of separable rows           */
    /* separable atoms, per Jacobian entry (unused entries of tape rows ignored)     */
    const uint8_t* atom_kind;   /* [nnz] KTN_ATOM_*                                  */
    const double*  p0;          /* [nnz]                                             */
    const double*  p1;          /* [nnz]                                             */
    /* tapes: row i owns ops [tape_ptr[i], tape_ptr[i+1]) (empty for separable rows) */
    const int64_t* tape_ptr;    /* [num_constr+1] or NULL when there are no tapes    */
    const int32_t* tape_op;
    const double*  tape_arg;
    /* objective f(x): isobjlinear src/model.jl:125; same two forms                  */
    int32_t obj_linear;
    int32_t obj_kind;           /* KTN_ROW_SEP, KTN_ROW_TAPE or KTN_ROW_HOST         */
    int64_t obj_nnz;            /* separable objective entries                       */
    const int32_t* obj_col;
    const uint8_t* obj_atom_kind;
    const double*  obj_p0;
    const double*  obj_p1;
    double         obj_const;
    int64_t        obj_tape_len;
    const int32_t* obj_tape_op;
    const double*  obj_tape_arg;
    /* host-evaluator fallback (KTN_ROW_HOST rows / objective); NULL when unused.  Called on the thread that is
     * inside ktn_loadproblem / ktn_optimize / ktn_ecp_step / ktn_sep_precompute / ktn_sep_sweep.            */
    ktn_eval_rows_cb eval_rows;
    ktn_eval_obj_cb  eval_obj;
    void*            eval_user;
} ktn_nlp_desc;

/* ---- plugin surface -------------------------------------------------------------- */

/* defaults of KatanaSolver(...) src/solver.jl:34-43 */
void ktn_default_params(ktn_params* p);

/* KatanaSolver(lp_solver; kwargs) + MathProgBase.NonlinearModel(s)
 * (src/solver.jl:34-43, src/model.jl:41-65) */
/* (on failure no handle exists, so ktn_last_error cannot be asked: the message goes to stderr, the code is returned) */
int ktn_create(const ktn_params* p, ktn_handle* out);
void ktn_destroy(ktn_handle h);
const char* ktn_last_error(ktn_handle h);
int ktn_abi_version(void);
/* sizeof(ktn_params) / sizeof(ktn_nlp_desc) as compiled into the library: a binding checks its own struct mirrors
 * against these before the first call (a layout mismatch would otherwise corrupt memory silently) */
int64_t ktn_sizeof_params(void);
int64_t ktn_sizeof_nlp_desc(void);

/* MathProgBase.loadproblem!(m, num_var, num_constr, l_var, u_var, l_constr, u_constr,
 * sense, d)  src/model.jl:81-173 */
int ktn_loadproblem(ktn_handle h, int64_t num_var, int64_t num_constr,
                    const double* l_var, const double* u_var,
                    const double* l_constr, const double* u_constr,
                    int32_t sense, const ktn_nlp_desc* d);

/* MathProgBase.optimize!(m)  src/model.jl:219-319; returns the status code (>= 0)
 * or a negative error */
int ktn_optimize(ktn_handle h);

/* one pass of the hot loop src/model.jl:258-308 (LP re-solve -> sweep -> cuts);
 * *done = 1 when the loop condition of :257 ends.  ktn_optimize == presolve + steps. */
int ktn_optimize_begin(ktn_handle h);           /* src/model.jl:227-256 (presolve)   */
int ktn_ecp_step(ktn_handle h, int32_t* done);
int ktn_optimize_end(ktn_handle h);             /* src/model.jl:311-318              */
/* forget all cuts and iterates: back to the state right after ktn_loadproblem */
int ktn_reset(ktn_handle h);

/* MathProgBase.status / getobjval / getsolution / getsolvetime  src/model.jl:337-343;
 * numiters / numcuts src/model.jl:326,333; setwarmstart! (a no-op) :335 */
int     ktn_get_status(ktn_handle h);
double  ktn_get_objval(ktn_handle h);
int64_t ktn_get_num_var(ktn_handle h);     /* incl. the epigraph variable, model.jl:138 */
int     ktn_get_solution(ktn_handle h, double* x_out, int64_t n);
double  ktn_get_solvetime(ktn_handle h);
int64_t ktn_numiters(ktn_handle h);
int64_t ktn_numcuts(ktn_handle h);
int     ktn_setwarmstart(ktn_handle h, const double* x, int64_t n);

/* ---- separator API (batched form of src/separators.jl:23-53,111-120) --------------
 * precompute!(sep, xstar): evaluate g and the sparse Jacobian of ALL rows of the
 * loaded (epigraph-lifted) problem at xstar [num_var incl. aux] on the device. */
int ktn_sep_precompute(ktn_handle h, const double* xstar, int64_t n);
int64_t ktn_sep_num_constr(ktn_handle h);  /* incl. the epigraph row                  */
int64_t ktn_sep_jac_nnz(ktn_handle h);
int ktn_sep_get_g(ktn_handle h, double* g_out, int64_t m);
int ktn_sep_get_jac(ktn_handle h, double* jac_out, int64_t nnz);
int ktn_sep_get_structure(ktn_handle h, int64_t* rowptr_out, int32_t* col_out);
/* isconstrsat(sep, i, lb, ub, f_tol) src/separators.jl:120 -> 1/0 */
int ktn_sep_isconstrsat(ktn_handle h, int64_t i, double lb, double ub, double f_tol);
/* gencut(sep, xstar, bounds, i) -> AffExpr  src/separators.jl:118 +
 * linear_oa_cut src/algorithms.jl:3-18; *nnz in: capacity, out: entries written */
int ktn_sep_gencut(ktn_handle h, int64_t i, int32_t* cols, double* coefs, int64_t* nnz,
                   double* constant);
/* {isconstrsat, gencut, round_coefs, _addcut} over all NL rows at the point of the last
 * precompute (src/model.jl:272-283, 200-207, 68-79): appends the cuts to the LP.
 * Outputs: number of violated rows, largest violation. */
int ktn_sep_sweep(ktn_handle h, double f_tol, int64_t* nviol, double* maxviol);

/* ---- LP introspection: getKatanaCuts / getKatanaSols, src/util.jl:16-36 ------------ */
int64_t ktn_lp_num_rows(ktn_handle h);
int64_t ktn_lp_nnz(ktn_handle h);
int ktn_lp_get_rows(ktn_handle h, int64_t* rowptr, int32_t* col, double* val,
                    double* lo, double* hi);
int ktn_lp_get_objective(ktn_handle h, double* c_out, int64_t n, double* c0);
int ktn_lp_get_duals(ktn_handle h, double* y_out, int64_t m);
/* solve the current LP to the given tolerances (tests / tools):
 * row_tol = max unscaled row violation, gap_tol = relative duality gap */
int ktn_lp_solve(ktn_handle h, double row_tol, double gap_tol, int32_t* lp_status,
                 int64_t* pdhg_iters);
/* exactly `iters` PDHG iterations from (x0, y0) with fixed eta/omega, no restart, no
 * rescaling (dr = dc = 1): kernel-level parity hook for tests */
int ktn_lp_pdhg_raw(ktn_handle h, const double* x0, const double* y0, double eta,
                    double omega, int64_t iters, double* x_out, double* y_out);
int64_t ktn_num_lp_sols(ktn_handle h);           /* :VisData lp_sols, src/model.jl:267 */
int ktn_get_lp_sol(ktn_handle h, int64_t k, double* x_out, int64_t n);

/* ---- statistics (not in the reference: measurement hooks) ----
 * names: "lp_time_s" "sep_time_s" "pdhg_iters" "lp_solves" "lp_restarts" "sweeps"
 * with params.profile = 1, per hot kernel K in {kx, ky, sweep_eval}:
 *        "K_time_s" "K_launches" "K_bytes"  from the start/stop hipEvents of hipExtLaunchKernelGGL
 *        on the engine's own stream (dispatch begin/end, as rocprofv3 --kernel-trace reports) */
double ktn_get_stat(ktn_handle h, const char* name);

/* ---- multi-GPU building blocks (no reference counterpart) -----
 * The NL rows shard by contiguous blocks: every rank loads the linear rows plus ITS block
 * of NL rows, sweeps it, and the generated cuts are exchanged (RCCL all-gather through
 * torch.distributed in katana.jl_amd/distributed.py) and appended in rank order so that
 * every rank holds the identical LP.  These calls are what that host loop is made of:
 *   ktn_sweep_lp_point   : the loop body of src/model.jl:268-283 at the current LP solution
 *   ktn_lp_get_rows_from : export the rows appended since `first_row` (rowptr rebased to 0)
 *   ktn_lp_truncate      : drop the rows >= nrows again (own cuts are re-appended in rank order)
 *   ktn_lp_append_rows   : append a CSR block of rows (rowptr 0-based), duals start at 0
 * Use lp_dual_inherit = 0 with truncate/append (row indices of earlier cuts change). */
int ktn_sweep_lp_point(ktn_handle h, double f_tol, int64_t* nviol, double* maxviol);
/* this handle's share of the objective certificate (sum over ITS NL rows of multiplier mass x signed residual at the last sweep's
 * point; the engine's own loop evaluates the same sum over all rows after the stop rule is met, src/model.jl:257 +
 * test/runtests.jl:16-17): the host loop adds the shares of all ranks and clamps at zero.  id_offset = global id of the
 * handle's first NL row (ktn_lp_enable_global_lists), 0 otherwise. */
int ktn_objective_certificate(ktn_handle h, int64_t id_offset, double* sum);
int64_t ktn_lp_nnz_from(ktn_handle h, int64_t first_row);
int ktn_lp_get_rows_from(ktn_handle h, int64_t first_row, int64_t* rowptr, int32_t* col, double* val,
                         double* lo, double* hi);
int ktn_lp_truncate(ktn_handle h, int64_t nrows);
/* Per-NL-row cut lists across ranks: after ktn_lp_enable_global_lists(h, total NL rows of the unsharded problem) rows
 * appended with ktn_lp_append_rows_nl carry the GLOBAL NL-row id of the row they cut (-1: none); the engine then gives them
 * the bookkeeping its own sweep gives local cuts -- the new cut inherits the multiplier of the row's previous cut
 * (lp_dual_inherit), stall consolidation and purging see the lists.  ktn_last_sweep_slots returns, for the cuts the last
 * sweep appended, the local NL slot (0-based position among this handle's NL rows) of each, in row order. */
int ktn_lp_enable_global_lists(ktn_handle h, int64_t nl_total);
int ktn_last_sweep_slots(ktn_handle h, int64_t* slots, int64_t cap, int64_t* count);
int ktn_lp_append_rows_nl(ktn_handle h, int64_t nrows, const int64_t* rowptr, const int32_t* col, const double* val,
                          const double* lo, const double* hi, const int64_t* nl_id);
/* The same exchange without leaving device memory (the north star's "RCCL all-gather of generated cuts over xGMI"): the rows
 * [first_row, M) -- the cuts a sweep just appended -- are written as ONE f64 block
 *     [rowptr[1:] rebased | col | val | lo | hi | global NL-row id = id_offset + local NL slot]      (4 nrows + 2 nnz doubles)
 * into a DEVICE buffer of the caller (dev_out == NULL: size query only), all-gathered by the caller (RCCL), and every rank's
 * block is appended from the device receive buffer by ktn_lp_append_packed_dev, with the bookkeeping of
 * ktn_lp_append_rows_nl.  Both calls return with the engine's stream idle; the caller synchronises its own stream between
 * the all-gather and the appends. */
int ktn_lp_pack_rows_dev(ktn_handle h, int64_t first_row, int64_t id_offset, double* dev_out, int64_t cap,
                         int64_t* nrows, int64_t* nnz);
int ktn_lp_append_packed_dev(ktn_handle h, int64_t nrows, int64_t nnz, const double* dev_in);
/* ONE implementation of the cutting-plane step for the NL-row-block layout (round 4): with an exchange callback installed
 * (after ktn_lp_enable_global_lists; first_nl_id = global id of this handle's first NL row) ktn_ecp_step / ktn_optimize run the
 * engine's own loop -- tolerance schedule, floor rule, purge, refinement by the objective certificate -- and call the callback
 * where the single-GPU loop sweeps: every decision of the loop is taken from what the callback returns, i.e. from the same
 * numbers on every rank.  (Until round 3 a host loop re-stated those rules on top of the building blocks above.)  cb = NULL
 * removes it. */
int ktn_set_cut_exchange(ktn_handle h, ktn_exchange_cb cb, void* user, int64_t first_nl_id);
/* cut-pool purge after an LP solve (what ktn_ecp_step does between the LP and the sweep); deterministic, so ranks that
 * hold identical LPs stay identical */
int ktn_lp_purge(ktn_handle h, int64_t* rows_removed);
int ktn_lp_append_rows(ktn_handle h, int64_t nrows, const int64_t* rowptr, const int32_t* col,
                       const double* val, const double* lo, const double* hi);

/* ---- throughput mode (BASELINE.json configs[4]; the loop being batched is src/model.jl:257-309) ---------------------
 * A batch of independent instances is loaded as ONE block-diagonal problem (variables, rows and the summed linear
 * objective side by side).  ktn_set_blocks(h, nblocks, col_offsets[nblocks + 1]) -- after ktn_loadproblem -- tells the
 * engine which column ranges are the instances: every LP re-solve then runs as one launch with one workgroup per instance
 * (iterates in LDS, __syncthreads() instead of kernel boundaries, restarts and termination decided per instance), while
 * the sweep and the cut bookkeeping keep serving the whole batch at once.  nblocks = 0 switches it off. */
int ktn_set_blocks(ktn_handle h, int64_t nblocks, const int64_t* col_offsets);
/* MathProgBase.optimize! for such a batch with the WHOLE loop of src/model.jl:257-309 of every instance inside its own
 * workgroup (csrc/batch_ecp.hpp): LP scaling, step-size estimate, PDHG with its checks and restarts, the sweep over the
 * instance's NL rows, cut append, column mirror, tolerance schedule and stop rule -- no instance waits for another.  Needs
 * separable rows, a linear :Min objective, finite variable bounds and the instances' rows grouped instance after instance
 * (instances.fuse_instances); `cut_capacity` = room for that many cuts per NL row (<= 0: 12).  Falls back to the ordinary
 * loop when the batch does not qualify or an instance runs out of room.  Returns the status like ktn_optimize; getters as usual
 * (numiters = the largest per-instance count, numcuts = the sum). */
int ktn_optimize_blocks(ktn_handle h, int32_t cut_capacity);

/* ---- row-sharded LP over several GPUs (no reference counterpart: it splits the LP re-solve of
 * src/model.jl:259 and the cut loop of :272-283 over the ranks) -------------------------------------------------------
 * One process per GPU.  Every rank creates a handle, joins the group with ONE of the two calls below BEFORE
 * ktn_loadproblem, and loads ITS shard: all variables and the objective, a block of the linear rows and a block of the NL
 * rows (katana.jl_amd/distributed.py::shard_rows).  ktn_optimize is then a collective call: x is replicated, every rank
 * keeps the cuts it generates (no cut exchange), A x is local and A'y is a local partial plus an all-reduce of an n-vector
 * per PDHG iteration; the stop rule and every restart decision use all-reduced quantities, so all ranks return the same
 * status, objective and solution.  With a nonlinear objective the epigraph row belongs to rank 0.
 *   ktn_dist_unique_id  : 128-byte RCCL id, made by ONE rank and sent to the others (e.g. torch.distributed.broadcast)
 *   ktn_dist_init_rccl  : collectives = ncclAllReduce on the engine's stream (RCCL over xGMI)
 *   ktn_dist_init_callback : collectives through the caller: cb(user, host_buf, count, op) must all-reduce host_buf in
 *                         place over the ranks (op 0: sum, 1: max) and return 0 -- the engine stages through the host.
 *                         For tests (gloo; several ranks sharing one GPU, which RCCL does not allow).
 *   ktn_dist_ipc_export + ktn_dist_init_ipc : peer-buffer transport.  Every rank exposes a buffer of 2 x `capacity` doubles
 *                         (capacity >= the LP's column count) and a page of flag words; export returns their two
 *                         hipIpcMemHandle_t (2 x 64 bytes: data, flags); the caller gathers the `world` pairs in rank order
 *                         (any byte transport: torch.distributed.all_gather, MPI, a file) and hands them to init, which maps
 *                         the peers' buffers (xGMI peer access).  An all-reduce is then: partial written into the exposed slot,
 *                         one single-workgroup kernel that signals every rank and waits for every rank (bounded spin:
 *                         KTN_IPC_TIMEOUT_S, default 20 s, then KTN_E_HIP), and the consumer adding up the world's slots in rank
 *                         order itself -- no ring, no intermediate copy, identical bits on every rank.  2 <= world <= 8, one
 *                         node.  Ranks must destroy their handles together (ktn_destroy runs one last barrier).
 *   ktn_dist_allreduce_probe : collective self-test / timing of whatever transport the handle has: all-reduces (sum, max) a
 *                         vector whose result every rank can compute itself and returns the largest deviation, then the
 *                         mean time of `reps` sum all-reduces of n doubles.                                              */
typedef int (*ktn_allreduce_cb)(void* user, double* host_buf, int64_t count, int32_t op);
int ktn_dist_unique_id(char* out128);
int ktn_dist_init_rccl(ktn_handle h, const char* uid128, int32_t rank, int32_t world);
int ktn_dist_init_callback(ktn_handle h, int32_t rank, int32_t world, ktn_allreduce_cb cb, void* user);
#define KTN_IPC_HANDLE_BYTES 64
int ktn_dist_ipc_export(ktn_handle h, int32_t rank, int32_t world, int64_t capacity, char* out_handles128);
int ktn_dist_init_ipc(ktn_handle h, int32_t rank, int32_t world, const char* all_handles /* world x 128 bytes */);
int ktn_dist_allreduce_probe(ktn_handle h, int64_t n, int32_t reps, double* usec_per_call, double* max_abs_err);
/* leave the peer-buffer transport again (before ktn_loadproblem), e.g. after a failed ktn_dist_allreduce_probe: the handle
 * can then be given another transport (ktn_dist_init_rccl / _callback) -- decided once, at init, in the same process */
int ktn_dist_release_ipc(ktn_handle h);

#ifdef __cplusplus
}
#endif
#endif /* KATANA_HIP_H */
