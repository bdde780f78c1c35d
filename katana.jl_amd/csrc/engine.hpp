// engine.hpp -- struct Engine: the state of one handle and the declarations of its methods.  No kernel is named here; the methods
// live in
//   sweep.hip   loadproblem, the separator sweep (precompute! + isconstrsat + gencut + round_coefs + _addcut), the host-evaluator rows
//   pool.hip    the cut pool: LP row storage, CSC mirror, working (epigraph-shifted) form, purging, append / pack / unpack
//   lp.hip      the LP solves: scaling, tiled copies, PDHG steps and checks, exact small / mid-size LPs, per-block LPs, recession ray
//   ecp.hip     the cutting-plane loop: begin / step / polish_step / end, objective certificate, device-side batch loop
//   dist.hip    collectives of the row-sharded LP: RCCL, peer buffers, host callback
//   abi.hip     the C ABI of include/katana_hip.h
// (until the middle of round 4 all of it was engine.hip, one translation unit of 4 100 lines).
//
// Mirrors, function by function, the reference's driver (src/model.jl) with every piece of
// arithmetic on the device:
//   Engine::loadproblem   <- loadproblem!   src/model.jl:81-173
//   Engine::boundroutine  <- boundroutine   src/model.jl:175-197
//   Engine::begin/step/end<- optimize!      src/model.jl:219-319
//   Engine::sweep         <- precompute! + isconstrsat + gencut + round_coefs + _addcut
//   Engine::lp_solve      <- solve(m.linear_model)  (GLPK in the reference) replaced by a
//                            restarted, reflected Halpern PDHG on the growing cut matrix
// The CPU mirror of the LP algorithm used by the tests is oracle/pdlp_mirror.py
// (solve_lp_halpern); it is test infrastructure and never linked or called from here.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cmath>
#include <limits>
#include <map>
#include <string>
#include <vector>

#include <rccl/rccl.h>

#include "common.hpp"
#include "types.hpp"
#include "prims.hpp"

namespace ktn {

static const double kInf = std::numeric_limits<double>::infinity();

// lanes per sparse row: the power of two nearest below the average row length (a row longer
// than G just loops), so that short-row matrices do not idle half of every wavefront
static inline int pick_group(double avg_len) {
    int g = 4;
    while (g < 64 && 2 * g <= avg_len) g <<= 1;
    return g;
}

struct LpResult {
    int status = KTN_STATUS_NONE;   // OPTIMAL / USERLIMIT
    int64_t iters = 0;
    double pobj = 0.0, dobj = 0.0, row_viol = 0.0, gap = 0.0;
    double dres_rel = 0.0;          // dual residual over (1 + ||c||): what the solve's gap tolerance is compared with
    bool exact = false;             // solved by the exact small-LP kernel: no tolerance tightening needed
    bool stag_exit = false;         // ended through the primal-stagnation exit (objective flat, gap within lp_stag_factor x tolerance)
};

// Development switches (tools/README.md), parsed ONCE PER HANDLE at ktn_create from the KTN_* environment of that moment: two
// handles made under different settings can be A/B'd in one process, nothing is `static`, and -- unlike the getenv calls that
// sat inside lp_solve_core until round 3 -- every one of them is listed here, next to ktn_params, with what ships.
struct DevParams {
    bool no_pinned_check = false, debug_load = false, no_csc_merge = false, tiled_general_build = false, debug_blocks = false,
         no_tiled_check = false, no_setup_reuse = false, debug_lp = false, no_packed = false, force_collective = false;
    int sweep_rows = 0;            // KTN_SWEEP_ROWS       NL rows per lane group of the sweep (0 = by size)
    int blk_cfg = -1;              // KTN_BLK_CFG          tuning variant of the column-blocked sweep
    int sweep_blocked = -1;        // KTN_SWEEP_BLOCKED    0 = row kernel instead of the column-blocked sweep
    int sweep_batched = -1;        // KTN_SWEEP_BATCHED    0 = row kernel instead of the batch-blocked sweep for many short rows, 1 = always
    int tiled_wg = 2;              // KTN_TILED_WG         workgroups per CU of k_spmv_tiled
    int ecp_power = 20;            // KTN_ECP_POWER        power passes of the device-side batch loop
    int grp_rows = 0, grp_cols = 0;// KTN_GRP_ROWS / COLS  lanes per LP row / column (0 = by average length)
    int tiled = -1;                // KTN_TILED            force the tiled SpMV off (0) / on (1)
    double smax_reuse = 0.0;       // KTN_SMAX_REUSE       reuse of the sigma_max estimate (measured: harmful)
    int power_passes = 0;          // KTN_POWER_PASSES     power-iteration passes (0 = 8)
    int omega_robust = 1;          // KTN_OMEGA_ROBUST     0 = initial primal weight from the plain 2-norm ratio
    int packed_trips = 0;          // KTN_PACKED_TRIPS     outputs per lane group of the packed steps (0 = default)
    int first_chunk = 31;          // KTN_FIRST_CHUNK      iterations before the first check after a restart
    int stag_chunk = 0;            // KTN_STAG_CHUNK       check cadence while only the objective is unsettled (measured, off)
    int near_chunk = -1;           // KTN_NEAR_CHUNK       overrides lp_near_check
    int stag_checks = 2;           // KTN_STAG_CHECKS      flat checks the stagnation exit asks for
    double flat_factor = 0.4;      // KTN_FLAT_FACTOR      "flat" = within this fraction of the gap tolerance
    int omega_art = 1;             // KTN_OMEGA_ART        0 = no primal-weight update on restarts the residual did not earn
    double omega_art_k = 256.0;    // KTN_OMEGA_ART_K      period length at which such a restart's ratio gets the full weight 0.5
    double omega_art_clamp = 0.0, omega_clamp = 0.0, omega_clamp_down = 0.0;   // KTN_OMEGA_ART_CLAMP / _CLAMP / _CLAMP_DOWN (measured, off)
    double ipc_timeout_s = 20.0;   // KTN_IPC_TIMEOUT_S    spin bound of the peer-buffer transport
    static bool flag(const char* k) { return std::getenv(k) != nullptr; }
    static int geti(const char* k, int d) { const char* v = std::getenv(k); return v ? std::atoi(v) : d; }
    static double getd(const char* k, double d) { const char* v = std::getenv(k); return v ? std::atof(v) : d; }
    void from_env() {
        no_pinned_check = flag("KTN_NO_PINNED_CHECK"); debug_load = flag("KTN_DEBUG_LOAD"); no_csc_merge = flag("KTN_NO_CSC_MERGE");
        tiled_general_build = flag("KTN_TILED_GENERAL_BUILD"); debug_blocks = flag("KTN_DEBUG_BLOCKS"); no_tiled_check = flag("KTN_NO_TILED_CHECK");
        no_setup_reuse = flag("KTN_NO_SETUP_REUSE"); debug_lp = flag("KTN_DEBUG_LP"); no_packed = flag("KTN_NO_PACKED");
        force_collective = flag("KTN_FORCE_COLLECTIVE");
        sweep_rows = geti("KTN_SWEEP_ROWS", sweep_rows); blk_cfg = geti("KTN_BLK_CFG", blk_cfg); sweep_blocked = geti("KTN_SWEEP_BLOCKED", sweep_blocked);
        sweep_batched = geti("KTN_SWEEP_BATCHED", sweep_batched);
        tiled_wg = geti("KTN_TILED_WG", tiled_wg); ecp_power = geti("KTN_ECP_POWER", ecp_power);
        grp_rows = geti("KTN_GRP_ROWS", grp_rows); grp_cols = geti("KTN_GRP_COLS", grp_cols); tiled = geti("KTN_TILED", tiled);
        smax_reuse = getd("KTN_SMAX_REUSE", smax_reuse); power_passes = geti("KTN_POWER_PASSES", power_passes);
        omega_robust = geti("KTN_OMEGA_ROBUST", omega_robust); packed_trips = geti("KTN_PACKED_TRIPS", packed_trips);
        first_chunk = geti("KTN_FIRST_CHUNK", first_chunk); stag_chunk = geti("KTN_STAG_CHUNK", stag_chunk); near_chunk = geti("KTN_NEAR_CHUNK", near_chunk);
        stag_checks = geti("KTN_STAG_CHECKS", stag_checks); flat_factor = getd("KTN_FLAT_FACTOR", flat_factor);
        omega_art = geti("KTN_OMEGA_ART", omega_art); omega_art_k = getd("KTN_OMEGA_ART_K", omega_art_k);
        omega_art_clamp = getd("KTN_OMEGA_ART_CLAMP", omega_art_clamp); omega_clamp = getd("KTN_OMEGA_CLAMP", omega_clamp);
        omega_clamp_down = getd("KTN_OMEGA_CLAMP_DOWN", omega_clamp_down); ipc_timeout_s = getd("KTN_IPC_TIMEOUT_S", ipc_timeout_s);
    }
};

struct Engine {
    ktn_params prm;
    DevParams dev;
    std::string err;
    hipStream_t stream = nullptr;
    int device = 0;

    // ---- problem (host) ----
    bool loaded = false;
    int64_t n0 = 0, m0 = 0;        // original sizes
    int64_t n_lp = 0;              // LP variables (n0, or n0+1 with the epigraph variable)
    int64_t m_ext = 0;             // rows of the extended structure = m0 + 1 (objective row last)
    int64_t nnz_ext = 0;
    int sense = KTN_MIN;
    bool obj_linear = true;
    bool has_inf_bound = false;
    std::vector<int64_t> h_rowptr;
    std::vector<int32_t> h_col;
    std::vector<uint8_t> h_rowkind;
    std::vector<double> h_lb, h_ub;      // per extended row
    std::vector<int32_t> h_nlrows;
    int64_t m_nl = 0, n_tape_nl = 0;
    int64_t m_nl_global = 0;       // NL rows over all ranks of a row-sharded LP (== m_nl otherwise): decisions that steer collectives use it
    int grp_sweep = 32;

    // ---- device NLP ----
    DBuf<int64_t> d_rowptr, d_nodeptr;
    DBuf<int32_t> d_col, d_nodeop, d_nodea, d_nodeb;
    DBuf<uint8_t> d_rowkind, d_padzero;
    DBuf<uint64_t> d_dkeys, d_dsorted;      // deepest-cut selection
    // host-evaluator fallback (KTN_ROW_HOST)
    ktn_eval_rows_cb cb_rows = nullptr;
    ktn_eval_obj_cb cb_obj = nullptr;
    void* cb_user = nullptr;
    int64_t n_host = 0, n_host_nl = 0;
    bool host_constr_rows = false, host_obj = false;
    std::vector<double> h_xh, h_gh, h_jh;
    DBuf<double> d_gh, d_jh;
    DBuf<int32_t> d_hostrows;
    void host_eval(const double* d_x);
    DBuf<int32_t> d_colk;
    DBuf<double2> d_pp;
    // block-major copy of the long rows for the column-blocked sweep (k_sep_eval_blk)
    bool blk_on = false;
    int blk_nb = 0, blk_cfg = 0, blk_cols = kBlkCols, blk_wg_per_cu = 2, num_cus = 256;
    DBuf<int32_t> d_bcolk;
    DBuf<double2> d_bpp;
    DBuf<int64_t> d_bseg;
    DBuf<int4> d_bkind;
    DBuf<SepSlot> d_slots;
    DBuf<SepPartial> d_part;
    DBuf<double> d_rconst, d_lb, d_ub, d_nodec, d_nodeval, d_nodeadj;
    DBuf<int32_t> d_nlrows, d_allrows, d_taperows_all, d_taperows_nl;
    // sweep state
    DBuf<double> d_g, d_jac, d_bconst, d_maxc, d_xs, d_ray, d_scal;
    DBuf<int32_t> d_nonfin, d_violslots, d_anynf;
    DBuf<int64_t> d_flag, d_cnt, d_rank, d_cntscan, d_lastcut, d_cutprev;
    DBuf<double> d_ones;
    DBuf<int32_t> d_age, d_age2;
    DBuf<int64_t> d_keep, d_keepnnz, d_newidx, d_newptr, d_cutprev2, lp_rowptr2;
    DBuf<int32_t> lp_col2;
    DBuf<double> lp_val2, lp_lo2, lp_hi2, lp_y2;
    DBuf<char> d_scantmp;
    bool have_precompute = false;

    // ---- LP ----
    DBuf<int64_t> lp_rowptr;
    DBuf<int32_t> lp_col;
    DBuf<double> lp_val, lp_lo, lp_hi, lp_y, lp_c, lp_l, lp_u, lp_x;
    double c0 = 0.0;
    int64_t M = 0, NNZ = 0, M_base = 0, NNZ_base = 0;
    int64_t numcuts = 0, numcuts_base = 0;
    bool lp_dirty = true;
    // matrix version: bumped wherever rows are appended or removed.  A re-solve of the SAME matrix (the floor-tolerance re-solve
    // of an iteration that found every row satisfied) reuses the scaling, the tiled copies and the sigma_max estimate.
    uint64_t lp_version = 1, scaled_version = 0, smax_version = 0;
    bool scaled_identity = false;
    // NL-row blocks over several GPUs with a replicated LP (the north star's design): the caller's callback exchanges the cuts of
    // a sweep (ktn_set_cut_exchange); the cutting-plane loop itself -- floor rule, refinement, certificate -- stays Engine::step
    ktn_exchange_cb exch_cb = nullptr;
    void* exch_user = nullptr;
    int64_t exch_lo = 0;         // global id of this handle's first NL row
    bool exchanging() const { return exch_cb != nullptr; }
    bool sharded_rows = false;   // rows were appended/truncated from the host: cut lists are not tracked ...
    bool glists = false;         // ... unless the host supplies global NL-row ids (ktn_lp_enable_global_lists)
    int64_t nl_total = 0;
    int64_t last_sweep_cuts = 0;
    DBuf<int64_t> d_glast, d_nlid;
    void append_link(int64_t nrows, const int64_t* nl_id_host);       // rows [M, M + nrows) just appended from the host
    void append_link_dev(int64_t nrows);                              // the same with the ids already in d_nlid (device-resident exchange)
    int64_t* list_heads() { return glists ? d_glast.p : d_lastcut.p; }
    int64_t list_count() const { return glists ? nl_total : m_nl; }
    bool lists_ok() const { return !sharded_rows || glists; }
    // CSC mirror + scaling + PDHG workspace
    DBuf<int64_t> c_ptr, c_cnt;
    DBuf<int32_t> c_row;
    DBuf<double> c_val, c_sval, r_sval;
    // per mirror position the CSR entry it came from; rows / entries the mirror covers; `lp_epoch` counts the changes that are
    // NOT appends (reset, purge, truncate) -- while it stands still the mirror is extended by a merge instead of a sort
    DBuf<uint32_t> c_perm, c_perm2;
    DBuf<int32_t> c_row2;
    DBuf<int64_t> c_ptr2, c_off;
    int64_t csc_M = -1, csc_NNZ = 0;
    uint64_t lp_epoch = 1, csc_epoch = 0;
    void csc_merge_appended();
    // Working form of the LP during a first-order solve with a nonlinear objective: the epigraph cuts relative to the newest one
    // (kernels.hpp "epigraph reference shift").  The stored LP keeps the reference's form; w_shift says that the CSC mirror,
    // the scaling and the w* arrays currently hold the shifted problem.
    bool w_shift = false;
    int64_t M_lin = 0;                          // LP rows [0, M_lin) are the pass-through linear rows: never epigraph cuts
    DBuf<double> wval, wlo, whi, wc, epi_ref, epi_scal;      // epi_scal: [0] b_ref, [1] a_ref'x
    DBuf<unsigned long long> epi_newest;
    const double* Wval() const { return w_shift ? wval.p : lp_val.p; }
    const double* Wlo() const { return w_shift ? wlo.p : lp_lo.p; }
    const double* Whi() const { return w_shift ? whi.p : lp_hi.p; }
    const double* Wc() const { return w_shift ? wc.p : lp_c.p; }
    bool want_shift(int mode) const { return prm.epi_shift != 0 && !obj_linear && mode == 0 && !row_sharded() && n_blocks == 0; }
    void build_working();
    void ensure_matrix(bool shift);
    void epi_dot(const double* x);              // epi_scal[1] = a_ref'x  (a_ref is zero at the epigraph variable)
    DBuf<uint64_t> k_in, k_out;
    DBuf<uint32_t> p_in, p_out;
    DBuf<char> d_sorttmp;
    DBuf<double> dr_r, dc_r;
    DBuf<ColRec> d_crec;
    DBuf<RowRec> d_rrec;
    DBuf<int2> d_cbl;
    bool packed_on = false;        // plain steps read packed per-column / per-row records (kernels.hpp)
    int packed_trips = 1;          // outputs per lane group in the packed kernels (KTN_PACKED_TRIPS: 1, 2, 4)
    DBuf<double> dr2, dc2;                      // ping-pong partners of dr / dc in the scaling passes
    DBuf<double> dr, dc, statr, statc, ch, lh, uh, loh, hih, xh, yh, x0h, y0h, xth, yth, xbar, pv, pw, box;
    DBuf<double> partials, chk_part, chkout, power_v, power_v0;
    int64_t power_v0_n = -1;                   // size the cached start vector of the power iteration was made for
    // the check sums of a one-GPU solve land in pinned, device-mapped host memory: k_chk_final writes them there and the host
    // reads them after the stream synchronisation -- no copy kernel (4.5 us + a boundary) per check
    double* h_chk = nullptr;
    double* h_chk_dev = nullptr;
    // exact small-LP path (dense_lp.hpp)
    int64_t lp_iter_budget = 0, dense_credit = 0, dense_run = 0;
    DBuf<int32_t> ds_W, ds_valid;
    DBuf<double> ds_dense, ds_out;
    // exact mid-size LP (mid_lp.hpp): basis inverse, working set, x and multipliers persist across the ECP iterations
    DBuf<double> md_Binv, md_hW, md_x, md_lam, md_u, md_d, md_r, md_pv, md_c, md_aug, md_prow, md_fcol;
    int64_t md_since_refactor = 0;
    int64_t mid_backoff = 0, mid_backoff_len = 0;      // after a failed exact solve the first-order method carries on alone for a while
    DBuf<int32_t> md_W, md_pi, md_lost;
    DBuf<MidState> md_st;
    bool md_valid = false;
    int64_t mid_credit = 0, mid_run = 0;
    DBuf<int32_t> d_longrows;
    // long COLUMNS of the mirror (a variable that every cut contains: min-max / epigraph-style models): found by find_long_cols
    // after the mirror is built; the column-side kernels then run in their vector form with a workgroup per long column
    DBuf<int32_t> d_longcols;
    int64_t n_longc = 0;
    int64_t col_gain_max = 0;                      // most NL rows sharing one column: what a column can gain per sweep
    int64_t col_len_max = -1, col_scan_rows = 0;   // longest column at the last scan, rows of the LP then (-1: never scanned)
    int64_t col_removed_rows = 0;                  // rows purged / truncated since that scan (appended since = M - col_scan_rows + this)
    void find_long_cols();
    void spmv_cols(const SpMat& AT, const double* v, double* out, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);
    static constexpr int kMaxChunk = 512;
    int64_t n_long = 0;
    static constexpr int64_t kLongRow = 2048;
    int64_t max_row_len = (int64_t)1 << 62;    // longest row the LP can hold (NLP structure rows + appended rows): no scan for long rows below kLongRow
    double omega = 1.0;
    bool have_omega = false;
    int grp_rows = 8, grp_cols = 8;
    // tiled copies of A^ (outputs = rows) and A^' (outputs = columns) for LPs beyond the caches (kernels.hpp "tiled SpMV")
    struct TiledBuf {
        DBuf<int64_t> segstart, segtot;
        DBuf<uint16_t> bptr, cur, idx;
        DBuf<int32_t> pcnt;
        DBuf<double> val;
        int nb_in = 0;
        int64_t tiles = 0, grid = 0, pieces = 1;     // persistent grid and the largest number of pieces of a tile
        TiledMat view() const { return TiledMat{segstart.p, bptr.p, idx.p, val.p, nb_in}; }
    } tA, tAT;
    DBuf<double> tpart;
    bool tiled_on = false, tiled_built = false;
    void launch_tiled(const TiledBuf& T, int64_t n_out, int64_t n_in, const double* in, hipEvent_t e0);
    bool build_tiled(TiledBuf& T, int64_t n_out, int64_t n_in, const int64_t* ptr, const int32_t* idx, const double* val, int64_t skip_longer);
    // throughput mode (batch_lp.hpp): the loaded problem is block-diagonal, one workgroup per block runs its LP
    int64_t n_blocks = 0, blocks_built_rows = -1;
    size_t lds_set_lp = 0, lds_set_ecp = 0;      // dynamic-LDS sizes already granted to the two per-instance kernels on this handle's device
    std::vector<int64_t> h_blkcol;
    DBuf<int64_t> d_blkcol;
    DBuf<int32_t> d_blkrowptr, d_blkrows, d_rowloc, d_crowl;
    DBuf<double> d_blkomega, d_blkres;
    int blk_nmax = 0, blk_mmax = 0;
    void build_blocks();
    bool optimize_blocks_device(int cap_mul);
    DBuf<EcpArena> d_ar;                          // arenas of the device-side loop
    DBuf<int64_t> d_blklin, d_blknl;
    DBuf<int32_t> e_rptr, e_cptr, e_last, e_prev;
    DBuf<uint16_t> e_rcol, e_crow;
    DBuf<double> e_xbest, e_ax, e_rval, e_rsval, e_lo, e_hi, e_y, e_dr, e_loh, e_hih, e_cval, e_csval, e_dc, e_ch, e_lh, e_uh, e_res;     // batch_ecp.hpp: the whole ECP loop of every instance in its own workgroup
    bool lp_solve_blocks(double tol_p, double tol_g, double eta, LpResult* R, int64_t max_it);
    double smax_prev = 0.0;
    int64_t smax_rows = 0;
    int64_t scal_rows = 0, scal_cols = 0;   // dr[0, scal_rows) / dc[0, scal_cols) hold the scaling of the last solve

    // ---- run state ----
    int status = KTN_STATUS_NONE;
    int lp_status = KTN_STATUS_OPTIMAL;
    int64_t iter = 0;
    double soltime = 0.0, objval = std::numeric_limits<double>::quiet_NaN();
    double last_maxviol = 1e300;
    double obj_prev = kInf;
    bool allsat = false, begun = false, tight_done = false;
    // terminal refinement (DESIGN.md section 5 "Polish"): after the stop rule of model.jl:257,273 holds, small problems keep
    // cutting at polish_factor * f_tol; the point returned is the best one that satisfies the reference's rule
    bool polishing = false, polish_done = false;
    int polish_count = 0;
    double polish_phi = 1e-3;       // the refinement cuts rows beyond polish_phi * f_tol
    double cert_target = 0.0;       // > 0: certificate-driven refinement (kernels.hpp "objective certificate"), ends when met
    double cert_gap = 0.0;          // relative LP gap tolerance of the refinement solves (a quarter of the objective target)
    DBuf<double> d_cert;
    double objective_certificate(int64_t id_offset = 0, bool raw = false);
    double certificate_all_ranks();
    bool sweep_all(const double* d_x, double f_cut, bool lp_ok, int lp_stat, int64_t* nviol, double* maxviol, double* extra0, double* extra1);
    double certificate_blocks(double* gap_tol);
    DBuf<double> d_certblk;
    double best_viol = kInf, best_obj = 0.0;
    DBuf<double> d_xbest;
    // print_header / print_stats bookkeeping  src/model.jl:209-217,252-254,284-303
    int64_t log_cuts_lastprnt = 0, log_max_viol = 0, purged_total = 0;
    bool logging() const { return prm.log_level > 0 && dist.rank == 0; }      // (row-sharded: one table, from rank 0)
    void print_header() const {
        std::printf("%-10s %-15s %-15s %-20s %-20s %-15s\n", "Iteration", "Total cuts", "Cuts added", "Max constr. viol.",
                    "Avg constr. viol.", "Current cuts");
    }
    // model.jl:213-217.  "Current cuts" is numcuts in the reference (it never removes a cut, :215 TODO); here it is the
    // number of cuts still in the LP after purging.
    void print_stats(int64_t iter_lastprnt) const {
        const double avg = (double)log_cuts_lastprnt / ((double)iter_lastprnt * (double)m_nl);
        std::printf("%-10lld %-15lld %-15lld %-20lld %-20.2f %-15lld\n", (long long)iter, (long long)numcuts,
                    (long long)log_cuts_lastprnt, (long long)log_max_viol, avg, (long long)(numcuts - purged_total));
        std::fflush(stdout);
    }
    std::chrono::steady_clock::time_point t_start;
    std::vector<std::vector<double>> lp_sols;
    std::map<std::string, double> stats;
    // profiling events
    std::vector<hipEvent_t> ev_pool;
    struct EvRec { int kind; size_t a, b; double bytes; };
    std::vector<EvRec> ev_recs;
    size_t ev_used = 0;

    explicit Engine(const ktn_params& p) : prm(p) {
        dev.from_env();
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
            throw Error(KTN_E_NODEVICE, "no HIP device visible: the Katana HIP engine has no CPU path");
        if (prm.device >= 0) {
            KTN_HIP(hipSetDevice(prm.device));
            device = prm.device;
        } else {
            KTN_HIP(hipGetDevice(&device));
        }
        {
            int cus = 0;
            if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) num_cus = cus;
        }
        KTN_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        partials.resize((size_t)kRedBlocks * kChkQ * 2, stream);
        chkout.resize(kChkQ * 2 + 16, stream);
        if (!dev.no_pinned_check &&
            hipHostMalloc((void**)&h_chk, sizeof(double) * (2 * kChkQ + 16), hipHostMallocMapped) == hipSuccess) {
            if (hipHostGetDevicePointer((void**)&h_chk_dev, h_chk, 0) != hipSuccess) { (void)hipHostFree(h_chk); h_chk = nullptr; h_chk_dev = nullptr; }
        } else {
            h_chk = nullptr;
            (void)hipGetLastError();
        }
        d_scal.resize(8, stream);
        d_anynf.resize(4, stream);
    }
    ~Engine() {
        if (dist.comm) (void)ncclCommDestroy(dist.comm);
        ipc_release();
        for (auto e : ev_pool) (void)hipEventDestroy(e);
        if (stream) (void)hipStreamDestroy(stream);
        if (h_chk) (void)hipHostFree(h_chk);
    }

    void sync() { KTN_HIP(hipStreamSynchronize(stream)); }
    bool chk_pinned() const { return h_chk_dev != nullptr && !row_sharded(); }       // (row-sharded: the sums are all-reduced on the device first)
    double* chk_target() { return chk_pinned() ? h_chk_dev : chkout.p; }
    void check_launch() { KTN_HIP(hipGetLastError()); }

    // ------------------------------------------------------- row-sharded LP over several GPUs ---
    // (kernels.hpp "row-sharded".)  world > 1: this handle holds a block of the linear rows and the
    // cuts of its block of NL rows; x is replicated, y local.  Collectives run on the engine's own stream: RCCL (xGMI)
    // when the communicator was made by ktn_dist_init_rccl, or a host callback (tests: gloo, ranks sharing one GPU).
    struct DistCtx {
        int rank = 0, world = 1;
        bool force = false;            // world == 1 but run the collectives anyway (one-rank RCCL test)
        ncclComm_t comm = nullptr;
        ktn_allreduce_cb cb = nullptr;
        void* user = nullptr;
        std::vector<double> hbuf;
        // peer-buffer transport (ktn_dist_ipc_export / ktn_dist_init_ipc; kernels.hpp "peer-buffer transport")
        struct Ipc {
            bool on = false;
            int64_t cap = 0;                          // doubles per slot
            double* data = nullptr;                   // this rank's exposed buffer: 2 slots
            unsigned long long* flags = nullptr;      // this rank's flag words (uncached)
            void* opened[2 * kIpcMaxRanks] = {};      // what hipIpcOpenMemHandle returned (to close)
            IpcPeers P = {};
            unsigned long long epoch = 0;
            long long timeout_ticks = 0;
            int* h_err = nullptr;                     // pinned, device-mapped: a timed-out spin reports here
            int* h_err_dev = nullptr;
        } ipc;
    } dist;
    // the slot the NEXT all-reduce publishes from: a producer may write its partial straight into it
    int64_t ipc_off() const { return (int64_t)(dist.ipc.epoch & 1ull) * dist.ipc.cap; }
    double* ipc_slot() const { return dist.ipc.data + ipc_off(); }
    void ipc_check() {
        if (dist.ipc.on && dist.ipc.h_err && *dist.ipc.h_err != 0) {
            const int code = *dist.ipc.h_err;           // 1 + r: rank r did not arrive in time; 101 + r: rank r reported its own failure
            if (code > 100)
                throw Error(KTN_E_HIP, "peer-buffer transport: rank " + std::to_string(code - 101) + " gave up (told rank " + std::to_string(dist.rank) + ")");
            throw Error(KTN_E_HIP, "peer-buffer transport: rank " + std::to_string(dist.rank) + " timed out waiting for rank " + std::to_string(code - 1));
        }
    }
    // signal "my slot of this epoch is complete" to every rank and wait for theirs; returns the slot offset to read
    int64_t ipc_barrier();
    void ipc_release();
    void probe_fill(int64_t n, double* v, double value);    // v[j] = value + 1e-3 (j mod 1000): contents of ktn_dist_allreduce_probe
    DBuf<double> d_red;            // small device scratch for scalar reductions
    bool row_sharded() const { return dist.world > 1 || dist.force; }
    void allreduce(double* d, size_t n, int op);          // in place; op 0: sum, 1: max
    // k values reduced over the ranks (host in, host out); every rank gets the identical result
    void allreduce_host(double* v, int k, int op);

    // ------------------------------------------------------------------ profiling ---
    size_t ev_get() {
        if (ev_used == ev_pool.size()) {
            hipEvent_t e;
            KTN_HIP(hipEventCreate(&e));
            ev_pool.push_back(e);
        }
        return ev_used++;
    }
    // profile mode: the timed launches go through hipExtLaunchKernelGGL, whose start/stop events
    // carry the dispatch's own begin/end timestamps (what rocprofv3 --kernel-trace reports)
    void ev_flush() {   // stream must be synchronised
        static const char* names[4] = {"kx", "ky", "sweep_eval", "allreduce"};
        for (auto& r : ev_recs) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, ev_pool[r.a], ev_pool[r.b]) == hipSuccess) {
                std::string k = names[r.kind];
                stats[k + "_time_s"] += ms * 1e-3;
                stats[k + "_launches"] += 1.0;
                stats[k + "_bytes"] += r.bytes;
            }
        }
        ev_recs.clear();
        ev_used = 0;
    }

    // --------------------------------------------------------------- reductions ---
    double dev_dot(int64_t n, const double* a, const double* b);
    double dev_finite_sq(int64_t n, const double* a);
    void exclusive_scan(const int64_t* in, int64_t* out, size_t n) {
        size_t need = scan_i64_temp_bytes(n);
        d_scantmp.resize(need + 16, stream);
        KTN_HIP(exclusive_scan_i64(d_scantmp.p, need, in, out, n, stream));
    }

    NlpDev nlp_view();
    SweepOut sweep_view();
    LpRows lp_view();
    // device side of ktn_lp_pack_rows_dev / ktn_lp_append_packed_dev (the cut exchange that stays in device memory)
    void pack_rows_launch(int64_t nr, int64_t nz, int64_t first_row, int64_t base, bool ids, int64_t id_offset, double* dev_out);
    void unpack_rows_launch(int64_t nrows, int64_t nnz, const double* dev_in);

    // ================================================================ loadproblem ===
    void loadproblem(int64_t num_var, int64_t num_constr, const double* l_var, const double* u_var,
                     const double* l_constr, const double* u_constr, int32_t sense_, const ktn_nlp_desc* d);
    void reset();

    // ================================================================ separator =====
    // precompute! for every row of the extended structure (jac materialised)
    void precompute_all(const double* d_x);

    // the batched {isconstrsat, gencut, round_coefs, _addcut} over the NL rows
    void sweep(const double* d_x, double f_tol, int64_t* nviol_out, double* maxviol_out, bool* nonfinite_out);
    // sweep + (row-sharded) the stop rule's quantities over all ranks: number of violated rows, largest violation, error flag
    void global_sweep(const double* d_x, double f_tol, int64_t* nviol, double* maxviol, bool* nonfinite);
    double sweep_bytes = 0.0;
    // very long separable rows (kernels.hpp k_sep_eval_long): all of them / those among the NL rows, with their NL slot
    DBuf<int32_t> d_longev_rows, d_longev_nlrows;
    DBuf<int64_t> d_longev_slots, d_longev_nlslots;
    int64_t n_longev = 0, n_longev_nl = 0;
    // batch-blocked sweep for many short rows (kernels.hpp k_sep_sweep_batch): the regrouped copy of the NL entries
    bool sb_on = false;
    int64_t sb_batches = 0;
    int sb_nb = 0;
    bool sb_lds_set = false;
    DBuf<uint16_t> d_sbck, d_sbrow;
    DBuf<double2> d_sbpp;
    DBuf<int64_t> d_sbseg;

    // ================================================================ LP ============
    void rebuild_csc();
    void purge_cuts();
    // Capacity for the cut pool up front: growing a buffer is hipMalloc + copy + hipFree (which synchronises the device),
    // and on a large instance the pool passes through a dozen sizes in the first iterations (cfg4: 2.5 of the 4.9 s of a
    // cold solve).  HBM is plentiful (288 GB): reserve for three sweeps' worth of cuts.
    void reserve_lp(int64_t rows, int64_t nnz);
    void find_long_rows();
    bool recession_ray_dense(bool* unbounded);
    void launch_y(const SpMat& A, double sigma, double w, double rho, hipEvent_t e0, hipEvent_t e1);
    void launch_x(const SpMat& AT, double tau, double w, double rho, bool update, hipEvent_t e0, hipEvent_t e1);
    // w_next >= 0: the check kernels also write the Halpern update of this iteration into xnext / ynext (returns true when they did:
    // the plain CSR form only); the caller swaps them in for x / y instead of launching k_halpern2
    bool launch_check(const SpMat& A, const SpMat& AT, double tau, double sigma, double w_next = -1.0, double rho = 1.0);
    DBuf<double> xnext, ynext;
    int chk_nrow = 0, chk_ncol = 0;     // partial blocks of the last check (rows | columns)
    void compute_scaling(bool identity);
    LpResult lp_solve(double tol_p, double tol_g, int mode, bool identity_scaling = false);
    LpResult lp_solve_core(double tol_p, double tol_g, int mode, bool identity_scaling);
    bool lp_solve_dense(LpResult* R);
    bool lp_solve_mid(LpResult* R);
    void pdhg_raw(const double* x0, const double* y0, double eta, double omega_, int64_t iters, double* x_out,
                  double* y_out);

    // ================================================================ ECP driver ====
    bool recession_ray();
    void boundroutine();
    void begin();
    void step(int32_t* done);
    void polish_step(int32_t* done);
    void end();
};

}  // namespace ktn
