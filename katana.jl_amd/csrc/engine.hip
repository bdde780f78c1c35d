// engine.hip -- host side of the Katana ECP engine + the C ABI of include/katana_hip.h.
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
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

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
#include "kernels.hpp"
#include "dense_lp.hpp"
#include "mid_lp.hpp"
#include "batch_lp.hpp"
#include "batch_ecp.hpp"
#include "prims.hpp"

namespace ktn {

static const double kInf = std::numeric_limits<double>::infinity();

__global__ __launch_bounds__(kBlock) void k_finite_sq_partial(int64_t n, const double* __restrict__ a,
                                                              double* __restrict__ partials) {
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const double v = a[i];
        if (isfinite(v)) acc += v * v;
    }
    __shared__ double sh[kBlock / 64];
    acc = group_sum<64>(acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double v = 0.0;
        for (int k = 0; k < kBlock / 64; ++k) v += sh[k];
        partials[blockIdx.x] = v;
    }
}

__global__ __launch_bounds__(kBlock) void k_hash_fill(int64_t n, double* __restrict__ z) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    uint64_t h = (uint64_t)i * 0x9E3779B97F4A7C15ULL + 0xD1B54A32D192ED03ULL;
    h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ULL; h ^= h >> 32;
    z[i] = 0.25 + (double)(h >> 11) * (1.0 / 9007199254740992.0);   // in [0.25, 1.25)
}

// lanes per sparse row: the power of two nearest below the average row length (a row longer
// than G just loops), so that short-row matrices do not idle half of every wavefront
static inline int pick_group(double avg_len) {
    int g = 4;
    while (g < 64 && 2 * g <= avg_len) g <<= 1;
    return g;
}

#define LAUNCH_G(G, KERNEL, count, stream, ...)                                                          \
    do {                                                                                                 \
        const int64_t cnt__ = (count);                                                                   \
        if (cnt__ > 0) {                                                                                 \
            switch (G) {                                                                                 \
                case 4: hipLaunchKernelGGL((KERNEL<4>), dim3(ceil_div(cnt__ * 4, kBlock)), dim3(kBlock), 0, stream, __VA_ARGS__); break;   \
                case 8: hipLaunchKernelGGL((KERNEL<8>), dim3(ceil_div(cnt__ * 8, kBlock)), dim3(kBlock), 0, stream, __VA_ARGS__); break;   \
                case 16: hipLaunchKernelGGL((KERNEL<16>), dim3(ceil_div(cnt__ * 16, kBlock)), dim3(kBlock), 0, stream, __VA_ARGS__); break; \
                case 32: hipLaunchKernelGGL((KERNEL<32>), dim3(ceil_div(cnt__ * 32, kBlock)), dim3(kBlock), 0, stream, __VA_ARGS__); break; \
                default: hipLaunchKernelGGL((KERNEL<64>), dim3(ceil_div(cnt__ * 64, kBlock)), dim3(kBlock), 0, stream, __VA_ARGS__); break; \
            }                                                                                            \
        }                                                                                                \
    } while (0)

#define LAUNCH_GB(G, KERNEL, B, count, stream, ...)                                                      \
    do {                                                                                                 \
        const int64_t cnt__ = (count);                                                                   \
        if (cnt__ > 0) {                                                                                 \
            switch (G) {                                                                                 \
                case 4: hipLaunchKernelGGL((KERNEL<4, B>), dim3(ceil_div(cnt__ * 4, kBlock)), dim3(kBlock), 0, stream, __VA_ARGS__); break;   \
                case 8: hipLaunchKernelGGL((KERNEL<8, B>), dim3(ceil_div(cnt__ * 8, kBlock)), dim3(kBlock), 0, stream, __VA_ARGS__); break;   \
                case 16: hipLaunchKernelGGL((KERNEL<16, B>), dim3(ceil_div(cnt__ * 16, kBlock)), dim3(kBlock), 0, stream, __VA_ARGS__); break; \
                case 32: hipLaunchKernelGGL((KERNEL<32, B>), dim3(ceil_div(cnt__ * 32, kBlock)), dim3(kBlock), 0, stream, __VA_ARGS__); break; \
                default: hipLaunchKernelGGL((KERNEL<64, B>), dim3(ceil_div(cnt__ * 64, kBlock)), dim3(kBlock), 0, stream, __VA_ARGS__); break; \
            }                                                                                            \
        }                                                                                                \
    } while (0)

// same, with the kernel's own start/stop timestamps recorded into (E0, E1)
#define LAUNCH_G_EV(G, KERNEL, count, stream, E0, E1, ...)                                               \
    do {                                                                                                 \
        const int64_t cnt__ = (count);                                                                   \
        if (cnt__ > 0) {                                                                                 \
            switch (G) {                                                                                 \
                case 4: hipExtLaunchKernelGGL((KERNEL<4>), dim3(ceil_div(cnt__ * 4, kBlock)), dim3(kBlock), 0, stream, E0, E1, 0, __VA_ARGS__); break;   \
                case 8: hipExtLaunchKernelGGL((KERNEL<8>), dim3(ceil_div(cnt__ * 8, kBlock)), dim3(kBlock), 0, stream, E0, E1, 0, __VA_ARGS__); break;   \
                case 16: hipExtLaunchKernelGGL((KERNEL<16>), dim3(ceil_div(cnt__ * 16, kBlock)), dim3(kBlock), 0, stream, E0, E1, 0, __VA_ARGS__); break; \
                case 32: hipExtLaunchKernelGGL((KERNEL<32>), dim3(ceil_div(cnt__ * 32, kBlock)), dim3(kBlock), 0, stream, E0, E1, 0, __VA_ARGS__); break; \
                default: hipExtLaunchKernelGGL((KERNEL<64>), dim3(ceil_div(cnt__ * 64, kBlock)), dim3(kBlock), 0, stream, E0, E1, 0, __VA_ARGS__); break; \
            }                                                                                            \
        }                                                                                                \
    } while (0)

#define LAUNCH_GB_EV(G, KERNEL, B, count, stream, E0, E1, ...)                                           \
    do {                                                                                                 \
        const int64_t cnt__ = (count);                                                                   \
        if (cnt__ > 0) {                                                                                 \
            switch (G) {                                                                                 \
                case 4: hipExtLaunchKernelGGL((KERNEL<4, B>), dim3(ceil_div(cnt__ * 4, kBlock)), dim3(kBlock), 0, stream, E0, E1, 0, __VA_ARGS__); break;   \
                case 8: hipExtLaunchKernelGGL((KERNEL<8, B>), dim3(ceil_div(cnt__ * 8, kBlock)), dim3(kBlock), 0, stream, E0, E1, 0, __VA_ARGS__); break;   \
                case 16: hipExtLaunchKernelGGL((KERNEL<16, B>), dim3(ceil_div(cnt__ * 16, kBlock)), dim3(kBlock), 0, stream, E0, E1, 0, __VA_ARGS__); break; \
                case 32: hipExtLaunchKernelGGL((KERNEL<32, B>), dim3(ceil_div(cnt__ * 32, kBlock)), dim3(kBlock), 0, stream, E0, E1, 0, __VA_ARGS__); break; \
                default: hipExtLaunchKernelGGL((KERNEL<64, B>), dim3(ceil_div(cnt__ * 64, kBlock)), dim3(kBlock), 0, stream, E0, E1, 0, __VA_ARGS__); break; \
            }                                                                                            \
        }                                                                                                \
    } while (0)

// G lanes per output, T outputs per lane group: grid = ceil(count * G / (kBlock * T))
#define LAUNCH_GT(G, T, KERNEL, count, stream, E0, E1, ...)                                              \
    do {                                                                                                 \
        const int64_t cnt__ = (count);                                                                   \
        if (cnt__ > 0) {                                                                                 \
            switch (G) {                                                                                 \
                case 4: hipExtLaunchKernelGGL((KERNEL<4, T>), dim3(ceil_div(cnt__ * 4, kBlock * T)), dim3(kBlock), 0, stream, E0, E1, 0, __VA_ARGS__); break;   \
                case 8: hipExtLaunchKernelGGL((KERNEL<8, T>), dim3(ceil_div(cnt__ * 8, kBlock * T)), dim3(kBlock), 0, stream, E0, E1, 0, __VA_ARGS__); break;   \
                case 16: hipExtLaunchKernelGGL((KERNEL<16, T>), dim3(ceil_div(cnt__ * 16, kBlock * T)), dim3(kBlock), 0, stream, E0, E1, 0, __VA_ARGS__); break; \
                case 32: hipExtLaunchKernelGGL((KERNEL<32, T>), dim3(ceil_div(cnt__ * 32, kBlock * T)), dim3(kBlock), 0, stream, E0, E1, 0, __VA_ARGS__); break; \
                default: hipExtLaunchKernelGGL((KERNEL<64, T>), dim3(ceil_div(cnt__ * 64, kBlock * T)), dim3(kBlock), 0, stream, E0, E1, 0, __VA_ARGS__); break; \
            }                                                                                            \
        }                                                                                                \
    } while (0)

#define LAUNCH_1(KERNEL, count, stream, ...)                                                             \
    do {                                                                                                 \
        const int64_t cnt__ = (count);                                                                   \
        if (cnt__ > 0) hipLaunchKernelGGL(KERNEL, dim3(ceil_div(cnt__, kBlock)), dim3(kBlock), 0, stream, __VA_ARGS__); \
    } while (0)

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
    void append_link(int64_t nrows, const int64_t* nl_id_host) {       // rows [M, M + nrows) just appended from the host
        d_nlid.upload(nl_id_host, (size_t)nrows, stream);
        LAUNCH_1(k_append_link, nrows, stream, nrows, M, d_nlid.p, nl_total, d_glast.p, d_cutprev.p, lp_y.p, (int)prm.lp_dual_inherit);
        check_launch();
    }
    void append_link_dev(int64_t nrows) {                              // the same with the ids already in d_nlid (device-resident exchange)
        LAUNCH_1(k_append_link, nrows, stream, nrows, M, d_nlid.p, nl_total, d_glast.p, d_cutprev.p, lp_y.p, (int)prm.lp_dual_inherit);
        check_launch();
    }
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
    void epi_dot(const double* x) {             // epi_scal[1] = a_ref'x  (a_ref is zero at the epigraph variable)
        hipLaunchKernelGGL(k_dot_partial, dim3(kRedBlocks), dim3(kBlock), 0, stream, n_lp, epi_ref.p, x, partials.p);
        hipLaunchKernelGGL(k_sum_final, dim3(1), dim3(kRedBlocks), 0, stream, partials.p, kRedBlocks, epi_scal.p + 1);
    }
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
    DBuf<double> partials, chk_part, chkout, power_v;
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
    void launch_tiled(const TiledBuf& T, int64_t n_out, int64_t n_in, const double* in, hipEvent_t e0) {
        hipExtLaunchKernelGGL(k_spmv_tiled, dim3((unsigned)T.grid), dim3(kTileThreads), 0, stream, e0, nullptr, 0, n_out, n_in, T.tiles,
                              T.view(), in, tpart.p);
    }
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
    int64_t ipc_barrier() {
        const int64_t off = ipc_off();
        dist.ipc.epoch += 1;
        hipLaunchKernelGGL(k_ipc_barrier, dim3(1), dim3(64), 0, stream, dist.ipc.P, dist.rank, dist.world, dist.ipc.epoch,
                           dist.ipc.timeout_ticks, dist.ipc.h_err_dev);
        return off;
    }
    void ipc_release();
    DBuf<double> d_red;            // small device scratch for scalar reductions
    bool row_sharded() const { return dist.world > 1 || dist.force; }
    void allreduce(double* d, size_t n, int op) {          // in place; op 0: sum, 1: max
        if (!row_sharded() || n == 0) return;
        stats["allreduce_calls"] += 1.0;
        stats["allreduce_bytes"] += 8.0 * (double)n;
        if (dist.ipc.on) {
            KTN_REQUIRE((int64_t)n <= dist.ipc.cap, "peer-buffer transport: vector longer than the exposed slot");
            size_t ea = 0, eb = 0;
            if (prm.profile) { ea = ev_get(); eb = ev_get(); KTN_HIP(hipEventRecord(ev_pool[ea], stream)); }
            if (d != ipc_slot()) KTN_HIP(hipMemcpyAsync(ipc_slot(), d, n * sizeof(double), hipMemcpyDeviceToDevice, stream));
            const int64_t off = ipc_barrier();
            if (op) hipLaunchKernelGGL((k_ipc_reduce<1>), dim3(ceil_div((int64_t)n, kBlock)), dim3(kBlock), 0, stream, (int64_t)n, dist.ipc.P, dist.world, off, d);
            else hipLaunchKernelGGL((k_ipc_reduce<0>), dim3(ceil_div((int64_t)n, kBlock)), dim3(kBlock), 0, stream, (int64_t)n, dist.ipc.P, dist.world, off, d);
            check_launch();
            if (prm.profile) { KTN_HIP(hipEventRecord(ev_pool[eb], stream)); ev_recs.push_back({3, ea, eb, 8.0 * (double)n}); }
        } else if (dist.comm) {
            size_t ea = 0, eb = 0;
            if (prm.profile) { ea = ev_get(); eb = ev_get(); KTN_HIP(hipEventRecord(ev_pool[ea], stream)); }
            const ncclResult_t r = ncclAllReduce(d, d, n, ncclDouble, op ? ncclMax : ncclSum, dist.comm, stream);
            if (r != ncclSuccess) throw Error(KTN_E_HIP, std::string("ncclAllReduce: ") + ncclGetErrorString(r));
            if (prm.profile) { KTN_HIP(hipEventRecord(ev_pool[eb], stream)); ev_recs.push_back({3, ea, eb, 8.0 * (double)n}); }
        } else {
            KTN_REQUIRE(dist.cb != nullptr, "row-sharded handle without a collective transport");
            dist.hbuf.resize(n);
            KTN_HIP(hipMemcpyAsync(dist.hbuf.data(), d, n * sizeof(double), hipMemcpyDeviceToHost, stream));
            sync();
            if (dist.cb(dist.user, dist.hbuf.data(), (int64_t)n, op) != 0) throw Error(KTN_E_CALLBACK, "all-reduce callback failed");
            KTN_HIP(hipMemcpyAsync(d, dist.hbuf.data(), n * sizeof(double), hipMemcpyHostToDevice, stream));
            sync();
        }
    }
    // k values reduced over the ranks (host in, host out); every rank gets the identical result
    void allreduce_host(double* v, int k, int op) {
        if (!row_sharded()) return;
        d_red.resize(64, stream);
        KTN_REQUIRE(k <= 64, "allreduce_host: too many values");
        KTN_HIP(hipMemcpyAsync(d_red.p, v, (size_t)k * sizeof(double), hipMemcpyHostToDevice, stream));
        allreduce(d_red.p, (size_t)k, op);
        KTN_HIP(hipMemcpyAsync(v, d_red.p, (size_t)k * sizeof(double), hipMemcpyDeviceToHost, stream));
        sync();
        ipc_check();
    }

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
    double dev_dot(int64_t n, const double* a, const double* b) {
        hipLaunchKernelGGL(k_dot_partial, dim3(kRedBlocks), dim3(kBlock), 0, stream, n, a, b, partials.p);
        hipLaunchKernelGGL(k_sum_final, dim3(1), dim3(kRedBlocks), 0, stream, partials.p, kRedBlocks, chkout.p + 2 * kChkQ);
        double v = 0.0;
        KTN_HIP(hipMemcpyAsync(&v, chkout.p + 2 * kChkQ, sizeof(double), hipMemcpyDeviceToHost, stream));
        sync();
        return v;
    }
    double dev_finite_sq(int64_t n, const double* a) {
        hipLaunchKernelGGL(k_finite_sq_partial, dim3(kRedBlocks), dim3(kBlock), 0, stream, n, a, partials.p);
        hipLaunchKernelGGL(k_sum_final, dim3(1), dim3(kRedBlocks), 0, stream, partials.p, kRedBlocks, chkout.p + 2 * kChkQ);
        double v = 0.0;
        KTN_HIP(hipMemcpyAsync(&v, chkout.p + 2 * kChkQ, sizeof(double), hipMemcpyDeviceToHost, stream));
        sync();
        return v;
    }
    void exclusive_scan(const int64_t* in, int64_t* out, size_t n) {
        size_t need = scan_i64_temp_bytes(n);
        d_scantmp.resize(need + 16, stream);
        KTN_HIP(exclusive_scan_i64(d_scantmp.p, need, in, out, n, stream));
    }

    NlpDev nlp_view() {
        NlpDev P;
        P.rowptr = d_rowptr.p; P.col = d_col.p; P.colk = d_colk.p; P.pp = d_pp.p;
        P.rconst = d_rconst.p; P.row_kind = d_rowkind.p; P.pad_zero = d_padzero.p; P.lb = d_lb.p; P.ub = d_ub.p;
        P.node_ptr = d_nodeptr.p; P.node_op = d_nodeop.p; P.node_a = d_nodea.p; P.node_b = d_nodeb.p;
        P.node_c = d_nodec.p; P.node_val = d_nodeval.p; P.node_adj = d_nodeadj.p;
        return P;
    }
    SweepOut sweep_view() {
        SweepOut O;
        O.g = d_g.p; O.jac = d_jac.p; O.bconst = d_bconst.p; O.maxc = d_maxc.p; O.nonfin = d_nonfin.p;
        O.flag = d_flag.p; O.cnt = d_cnt.p; O.maxviol = d_scal.p; O.any_nonfin = d_anynf.p;
        return O;
    }
    LpRows lp_view() {
        LpRows L;
        L.rowptr = lp_rowptr.p; L.col = lp_col.p; L.val = lp_val.p; L.lo = lp_lo.p; L.hi = lp_hi.p; L.y = lp_y.p;
        return L;
    }

    // ================================================================ loadproblem ===
    void loadproblem(int64_t num_var, int64_t num_constr, const double* l_var, const double* u_var,
                     const double* l_constr, const double* u_constr, int32_t sense_, const ktn_nlp_desc* d);
    void evaluate_all(const double* d_x);
    void reset();

    // ================================================================ separator =====
    // precompute! for every row of the extended structure (jac materialised)
    void precompute_all(const double* d_x) {
        NlpDev P = nlp_view();
        SweepOut O = sweep_view();
        // many short rows: the R-rows-per-lane-group form of the sweep with the Jacobian store (same sums, same bits); the
        // selection is the sweep's: once one row per group would make several times the resident wavefronts
        const int64_t waves1 = m_ext * grp_sweep / 64, resident = (int64_t)num_cus * 32;
        if (waves1 >= 16 * resident) {
#define KTN_PRE_LAUNCH(G) hipLaunchKernelGGL((k_sep_sweep<G, 4, true>), dim3(ceil_div(ceil_div(m_ext, (int64_t)4) * G, kBlock)), dim3(kBlock), 0, stream, P, d_allrows.p, m_ext, d_x, 0.0, O)
            switch (grp_sweep) {
                case 8: KTN_PRE_LAUNCH(8); break;
                case 16: KTN_PRE_LAUNCH(16); break;
                case 32: KTN_PRE_LAUNCH(32); break;
                default: KTN_PRE_LAUNCH(64); break;
            }
#undef KTN_PRE_LAUNCH
        } else {
            LAUNCH_G(grp_sweep, k_sep_eval, m_ext, stream, P, d_allrows.p, m_ext, d_x, 0.0, 1, 0, O);
        }
        if (n_longev > 0)
            hipLaunchKernelGGL(k_sep_eval_long, dim3((unsigned)n_longev), dim3(1024), 0, stream, P, d_longev_rows.p, d_longev_slots.p, d_x, 0.0, 0, O);
        LAUNCH_1(k_tape_eval, (int64_t)d_taperows_all.n, stream, P, d_taperows_all.p, (int64_t)d_taperows_all.n, d_x, O);
        if (n_host > 0) host_eval(d_x);
        // cut constants / maxima of tape rows from the materialised Jacobian (flags unused here)
        KTN_HIP(hipMemsetAsync(d_scal.p, 0, sizeof(double), stream));
        KTN_HIP(hipMemsetAsync(d_anynf.p, 0, sizeof(int32_t), stream));
        LAUNCH_1(k_gj_stats, m_ext, stream, P, d_allrows.p, m_ext, d_x, 0.0, (int)KTN_ROW_TAPE, O);
        check_launch();
    }

    // the batched {isconstrsat, gencut, round_coefs, _addcut} over the NL rows
    void sweep(const double* d_x, double f_tol, int64_t* nviol_out, double* maxviol_out, bool* nonfinite_out) {
        auto t0 = std::chrono::steady_clock::now();
        *nviol_out = 0;
        *maxviol_out = 0.0;
        *nonfinite_out = false;
        last_sweep_cuts = 0;
        stats["sweeps"] += 1.0;
        if (m_nl == 0) return;
        NlpDev P = nlp_view();
        SweepOut O = sweep_view();
        KTN_HIP(hipMemsetAsync(d_scal.p, 0, sizeof(double), stream));
        KTN_HIP(hipMemsetAsync(d_anynf.p, 0, sizeof(int32_t), stream));
        if (blk_on) {
            // long rows: column-blocked evaluation through LDS, then the block-order combination
            hipEvent_t e0 = nullptr, e1 = nullptr;
            if (prm.profile) {
                const size_t ea = ev_get(), eb = ev_get();
                e0 = ev_pool[ea]; e1 = ev_pool[eb];
                ev_recs.push_back({2, ea, eb, sweep_bytes});
            }
#define KTN_BLK_LAUNCH(G, BC, BS, U)                                                                                      \
    hipExtLaunchKernelGGL((k_sep_eval_blk<G, BC, BS, U>), dim3((unsigned)(num_cus * blk_wg_per_cu)), dim3(BS), 0, stream, e0, nullptr, 0, \
                          d_bcolk.p, d_bpp.p, d_bseg.p, d_bkind.p, m_nl, blk_nb, d_x, n_lp, d_part.p)
            switch (blk_cfg) {       // KTN_BLK_CFG: tuning variants kept for the next round's experiments
                case 1: KTN_BLK_LAUNCH(16, 8192, 512, 4); break;
                case 2: KTN_BLK_LAUNCH(16, 16384, 1024, 4); break;
                default: KTN_BLK_LAUNCH(8, 8192, 512, 4); break;
            }
#undef KTN_BLK_LAUNCH
            hipExtLaunchKernelGGL(k_sep_combine, dim3(ceil_div(m_nl, kBlock)), dim3(kBlock), 0, stream, nullptr, e1, 0, d_slots.p, m_nl, blk_nb,
                                  d_part.p, f_tol, O);
        } else {
            // many short rows: several rows per lane group (k_sep_sweep) once one row per group would make more wavefronts
            // than the chip holds several times over; small sweeps keep one row per group and all the parallelism
            const int rows_env = dev.sweep_rows;
            const int64_t waves1 = m_nl * grp_sweep / 64, resident = (int64_t)num_cus * 32;
            const int R = rows_env > 0 ? rows_env : (waves1 >= 16 * resident ? 4 : waves1 >= 8 * resident ? 2 : 1);
            hipEvent_t e0 = nullptr, e1 = nullptr;
            if (prm.profile) {
                const size_t ea = ev_get(), eb = ev_get();
                e0 = ev_pool[ea]; e1 = ev_pool[eb];
                ev_recs.push_back({2, ea, eb, sweep_bytes});
            }
            if (sb_on) {
                if (!sb_lds_set) {
                    KTN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_sep_sweep_batch), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSbLds));
                    sb_lds_set = true;
                }
                SbView V{d_sbck.p, d_sbrow.p, d_sbpp.p, d_sbseg.p, sb_nb};
                hipExtLaunchKernelGGL(k_sep_sweep_batch, dim3((unsigned)sb_batches), dim3(kSbThreads), kSbLds, stream, e0, e1, 0, V, P, d_nlrows.p, m_nl, d_x, n_lp, f_tol, O);
            } else
            if (R >= 4) LAUNCH_GB_EV(grp_sweep, k_sep_sweep, 4, ceil_div(m_nl, (int64_t)4), stream, e0, e1, P, d_nlrows.p, m_nl, d_x, f_tol, O);
            else if (R >= 2) LAUNCH_GB_EV(grp_sweep, k_sep_sweep, 2, ceil_div(m_nl, (int64_t)2), stream, e0, e1, P, d_nlrows.p, m_nl, d_x, f_tol, O);
            else LAUNCH_G_EV(grp_sweep, k_sep_eval, m_nl, stream, e0, e1, P, d_nlrows.p, m_nl, d_x, f_tol, 0, 1, O);
        }
        if (n_longev_nl > 0)
            hipLaunchKernelGGL(k_sep_eval_long, dim3((unsigned)n_longev_nl), dim3(1024), 0, stream, P, d_longev_nlrows.p, d_longev_nlslots.p, d_x, f_tol, 1, O);
        if (n_tape_nl > 0 || n_host_nl > 0) {
            LAUNCH_1(k_tape_eval, n_tape_nl, stream, P, d_taperows_nl.p, n_tape_nl, d_x, O);
            if (n_host_nl > 0) host_eval(d_x);
            LAUNCH_1(k_gj_stats, m_nl, stream, P, d_nlrows.p, m_nl, d_x, f_tol, (int)KTN_ROW_TAPE, O);
        }
        check_launch();
        exclusive_scan(d_flag.p, d_rank.p, (size_t)m_nl);
        exclusive_scan(d_cnt.p, d_cntscan.p, (size_t)m_nl);
        int64_t tail[4];
        double mv = 0.0;
        int32_t anynf = 0;
        if (h_chk_dev) {                               // one thread gathers the six scalars into pinned host memory
            double* ht = h_chk + 2 * kChkQ;
            hipLaunchKernelGGL(k_host_tail, dim3(1), dim3(1), 0, stream, h_chk_dev + 2 * kChkQ, d_flag.p + (m_nl - 1), d_rank.p + (m_nl - 1),
                               d_cnt.p + (m_nl - 1), d_cntscan.p + (m_nl - 1), (const double*)d_scal.p, (const int32_t*)d_anynf.p,
                               (const int32_t*)nullptr);
            sync();
            for (int k = 0; k < 4; ++k) tail[k] = (int64_t)ht[k];
            mv = ht[4]; anynf = (int32_t)ht[5];
        } else {
            KTN_HIP(hipMemcpyAsync(&tail[0], d_flag.p + (m_nl - 1), 8, hipMemcpyDeviceToHost, stream));
            KTN_HIP(hipMemcpyAsync(&tail[1], d_rank.p + (m_nl - 1), 8, hipMemcpyDeviceToHost, stream));
            KTN_HIP(hipMemcpyAsync(&tail[2], d_cnt.p + (m_nl - 1), 8, hipMemcpyDeviceToHost, stream));
            KTN_HIP(hipMemcpyAsync(&tail[3], d_cntscan.p + (m_nl - 1), 8, hipMemcpyDeviceToHost, stream));
            KTN_HIP(hipMemcpyAsync(&mv, d_scal.p, 8, hipMemcpyDeviceToHost, stream));
            KTN_HIP(hipMemcpyAsync(&anynf, d_anynf.p, 4, hipMemcpyDeviceToHost, stream));
            sync();
        }
        if (prm.profile) ev_flush();
        int64_t V = tail[0] + tail[1], nnzV = tail[2] + tail[3];
        *nviol_out = V;                // the stop rule counts EVERY violated row (model.jl:273-283)
        *maxviol_out = mv;
        // Deepest-cut selection: an LP vertex is supported by at most n_lp rows, so when far more rows than that are
        // violated only the cut_cap_factor * n_lp deepest get a cut this iteration (the reference cuts every violated
        // row; with 1e6 NL rows over 1e5 variables that makes the LP 10x larger than it needs to be).  Ties at the
        // threshold are all kept.  Never triggers on the reference's own test models.
        int64_t cap = (prm.cut_cap_factor > 0.0) ? std::max<int64_t>((int64_t)(prm.cut_cap_factor * (double)n_lp), prm.cut_cap_min) : 0;
        if (cap > 0 && row_sharded()) cap = std::max<int64_t>(cap / dist.world, 1);      // every rank selects among ITS rows
        if (cap > 0 && V > cap && !anynf) {
            d_dkeys.resize((size_t)m_nl, stream); d_dsorted.resize((size_t)m_nl, stream);
            LAUNCH_1(k_depth_keys, m_nl, stream, P, d_nlrows.p, m_nl, d_g.p, d_flag.p, d_dkeys.p);
            const size_t need = sort_keys_desc_temp_bytes((size_t)m_nl);
            d_sorttmp.resize(need + 16, stream);
            KTN_HIP(sort_keys_desc_u64(d_sorttmp.p, need, d_dkeys.p, d_dsorted.p, (size_t)m_nl, stream));
            LAUNCH_1(k_depth_reflag, m_nl, stream, m_nl, d_dkeys.p, d_dsorted.p, cap, d_flag.p, d_cnt.p);
            check_launch();
            exclusive_scan(d_flag.p, d_rank.p, (size_t)m_nl);
            exclusive_scan(d_cnt.p, d_cntscan.p, (size_t)m_nl);
            KTN_HIP(hipMemcpyAsync(&tail[0], d_flag.p + (m_nl - 1), 8, hipMemcpyDeviceToHost, stream));
            KTN_HIP(hipMemcpyAsync(&tail[1], d_rank.p + (m_nl - 1), 8, hipMemcpyDeviceToHost, stream));
            KTN_HIP(hipMemcpyAsync(&tail[2], d_cnt.p + (m_nl - 1), 8, hipMemcpyDeviceToHost, stream));
            KTN_HIP(hipMemcpyAsync(&tail[3], d_cntscan.p + (m_nl - 1), 8, hipMemcpyDeviceToHost, stream));
            sync();
            V = tail[0] + tail[1];
            nnzV = tail[2] + tail[3];
            stats["cut_selections"] += 1.0;
            stats["cuts_skipped"] += (double)(*nviol_out - V);
        }
        if (anynf) {   // model.jl:69-73: "Nonlinear constraint or objective likely undefined within domain"
            std::fprintf(stderr, "WARNING: Nonlinear constraint or objective likely undefined within domain\n");
            *nonfinite_out = true;
            return;
        }
        if (V > 0) {
            lp_rowptr.resize((size_t)(M + V + 1), stream);
            lp_lo.resize((size_t)(M + V), stream);
            lp_hi.resize((size_t)(M + V), stream);
            lp_y.resize((size_t)(M + V), stream);
            d_cutprev.resize((size_t)(M + V), stream);
            d_age.resize((size_t)(M + V), stream);
            KTN_HIP(hipMemsetAsync(d_age.p + M, 0, (size_t)V * sizeof(int32_t), stream));
            lp_col.resize((size_t)(NNZ + nnzV), stream);
            lp_val.resize((size_t)(NNZ + nnzV), stream);
            d_violslots.resize((size_t)V, stream);
            LpRows L = lp_view();
            LAUNCH_1(k_compact, m_nl, stream, P, d_nlrows.p, m_nl, d_flag.p, d_rank.p, d_cntscan.p, d_bconst.p, M, NNZ, L,
                     d_violslots.p, d_lastcut.p, d_cutprev.p, (int)(prm.lp_dual_inherit && !glists));
            LAUNCH_G(grp_sweep, k_emit, V, stream, P, d_nlrows.p, d_violslots.p, V, d_x, d_jac.p, d_maxc.p,
                     prm.cut_coef_rng, 1, M, L);
            check_launch();
            M += V;
            NNZ += nnzV;
            numcuts += V;
            last_sweep_cuts = V;
            lp_dirty = true; ++lp_version;
        }
        sync();
        stats["sep_time_s"] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    // sweep + (row-sharded) the stop rule's quantities over all ranks: number of violated rows, largest violation, error flag
    void global_sweep(const double* d_x, double f_tol, int64_t* nviol, double* maxviol, bool* nonfinite) {
        sweep(d_x, f_tol, nviol, maxviol, nonfinite);
        if (!row_sharded()) return;
        double v[2] = {(double)*nviol, *nonfinite ? 1.0 : 0.0};
        allreduce_host(v, 2, 0);
        double mv = *maxviol;
        allreduce_host(&mv, 1, 1);
        *nviol = (int64_t)(v[0] + 0.5);
        *nonfinite = v[1] > 0.0;
        *maxviol = mv;
    }
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
    void reserve_lp(int64_t rows, int64_t nnz) {
        const size_t r = (size_t)rows + 1, z = (size_t)nnz + 1;
        for (DBuf<double>* b : {&lp_lo, &lp_hi, &lp_y, &lp_lo2, &lp_hi2, &lp_y2, &dr, &dr2, &statr, &loh, &hih, &yh, &y0h, &yth, &pw}) b->reserve(r, stream);
        for (DBuf<int64_t>* b : {&lp_rowptr, &lp_rowptr2, &d_cutprev, &d_cutprev2, &d_keep, &d_keepnnz, &d_newidx, &d_newptr}) b->reserve(r, stream);
        for (DBuf<int32_t>* b : {&d_age, &d_age2, &d_longrows}) b->reserve(r, stream);
        for (DBuf<double>* b : {&lp_val, &lp_val2, &c_val, &c_sval, &r_sval}) b->reserve(z, stream);
        for (DBuf<int32_t>* b : {&lp_col, &lp_col2, &c_row, &c_row2}) b->reserve(z, stream);
        c_perm.reserve(z, stream); c_perm2.reserve(z, stream);
        k_in.reserve(z, stream); k_out.reserve(z, stream); p_in.reserve(z, stream); p_out.reserve(z, stream);
        // per-solve scratch that would otherwise grow (hipMalloc + copy + hipFree, a device synchronisation each) while the
        // first solve runs: packed row records, check partials (at most rows / 4 + columns / 4 blocks), sort / scan storage
        d_rrec.reserve(r, stream);
        d_crec.reserve((size_t)n_lp + 1, stream); d_cbl.reserve((size_t)n_lp + 1, stream);
        chk_part.reserve((r / 4 + (size_t)n_lp / 4 + 4096) * kChkQ, stream);
        d_sorttmp.reserve(sort_pairs_temp_bytes(z) + 16, stream);
        d_scantmp.reserve(scan_i64_temp_bytes(std::max(r, (size_t)n_lp + 2)) + 16, stream);
    }
    void find_long_rows();
    bool recession_ray_dense(bool* unbounded);
    void launch_y(const SpMat& A, double sigma, double w, double rho, hipEvent_t e0, hipEvent_t e1);
    void launch_x(const SpMat& AT, double tau, double w, double rho, bool update, hipEvent_t e0, hipEvent_t e1);
    void launch_check(const SpMat& A, const SpMat& AT, double tau, double sigma);
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

// ------------------------------------------------------------------------------------
// loadproblem!  src/model.jl:81-173
// ------------------------------------------------------------------------------------
static void postfix_to_nodes(const int32_t* op, const double* arg, int64_t len, const int32_t* rcols, int64_t rlen,
                             int64_t jac_base, std::vector<int32_t>& nop, std::vector<int32_t>& na,
                             std::vector<int32_t>& nb, std::vector<double>& nc) {
    std::vector<int32_t> st;
    const int64_t base = (int64_t)nop.size();
    for (int64_t t = 0; t < len; ++t) {
        const int o = op[t];
        int32_t a = 0, b = 0;
        double c = 0.0;
        switch (o) {
            case KTN_OP_CONST: c = arg[t]; break;
            case KTN_OP_VAR: {
                const int32_t v = (int32_t)arg[t];
                int64_t slot = -1;
                for (int64_t s = 0; s < rlen; ++s) if (rcols[s] == v) { slot = s; break; }
                if (slot < 0) throw Error(KTN_E_INVALID, "tape variable missing from the row's Jacobian structure");
                a = v;
                b = (int32_t)(jac_base + slot);
            } break;
            case KTN_OP_ADD: case KTN_OP_SUB: case KTN_OP_MUL: case KTN_OP_DIV:
                if (st.size() < 2) throw Error(KTN_E_INVALID, "malformed tape (binary op underflow)");
                b = st.back(); st.pop_back();
                a = st.back(); st.pop_back();
                break;
            case KTN_OP_POWC: c = arg[t];  // fallthrough
            case KTN_OP_NEG: case KTN_OP_EXP: case KTN_OP_LOG: case KTN_OP_SQRT: case KTN_OP_SIN: case KTN_OP_COS:
                if (st.empty()) throw Error(KTN_E_INVALID, "malformed tape (unary op underflow)");
                a = st.back(); st.pop_back();
                break;
            default: throw Error(KTN_E_UNSUPPORTED, "Unsupported tape opcode " + std::to_string(o));
        }
        nop.push_back(o); na.push_back(a); nb.push_back(b); nc.push_back(c);
        st.push_back((int32_t)((int64_t)nop.size() - 1 - base));
    }
    if (len > 0 && st.size() != 1) throw Error(KTN_E_INVALID, "malformed tape (stack not reduced to one value)");
}

// KTN_ROW_HOST: the caller's evaluator computes g and J of those rows at x (one call per sweep, like the reference's
// precompute!, src/separators.jl:111-116); the values are staged to the device, everything downstream is unchanged.
void Engine::host_eval(const double* d_x) {
    const int64_t nx = std::min<int64_t>(n_lp > 0 ? n_lp : n0, n0 + 1);
    KTN_HIP(hipMemcpyAsync(h_xh.data(), d_x, (size_t)nx * sizeof(double), hipMemcpyDeviceToHost, stream));
    sync();
    if (host_constr_rows) {
        const int rc = cb_rows(cb_user, h_xh.data(), h_gh.data(), h_jh.data());
        if (rc != 0) throw Error(KTN_E_CALLBACK, "eval_rows callback failed (" + std::to_string(rc) + ")");
    }
    if (host_obj) {
        double f = 0.0;
        double* grad = h_jh.data() + h_rowptr[m0];
        const int rc = cb_obj(cb_user, h_xh.data(), &f, grad);
        if (rc != 0) throw Error(KTN_E_CALLBACK, "eval_obj callback failed (" + std::to_string(rc) + ")");
        const double t = (nx > n0) ? h_xh[(size_t)n0] : 0.0;
        h_gh[(size_t)m0] = f - t;                       // f(x) - t, src/nlpeval.jl:45
        grad[n0] = -1.0;                                // src/nlpeval.jl:62
    }
    d_gh.upload(h_gh.data(), (size_t)m_ext, stream);
    d_jh.upload(h_jh.data(), (size_t)nnz_ext, stream);
    NlpDev P = nlp_view();
    SweepOut O = sweep_view();
    LAUNCH_1(k_host_scatter, n_host, stream, P, d_hostrows.p, n_host, d_gh.p, d_jh.p, O);
    check_launch();
    stats["host_evals"] += 1.0;
}

void Engine::loadproblem(int64_t num_var, int64_t num_constr, const double* l_var, const double* u_var,
                         const double* l_constr, const double* u_constr, int32_t sense_, const ktn_nlp_desc* d) {
    KTN_REQUIRE(d != nullptr, "nlp description is NULL");
    KTN_REQUIRE(num_var >= 0 && num_constr >= 0, "negative sizes");
    KTN_REQUIRE(d->num_var == num_var && d->num_constr == num_constr, "nlp description sizes disagree with loadproblem");
    KTN_REQUIRE(num_var + 1 < ((int64_t)1 << kKindShift), "num_var too large for the packed 29-bit column index");
    const bool dbg_load = dev.debug_load;
    auto tl0 = std::chrono::steady_clock::now();
    auto lapl = [&](const char* what) {
        if (!dbg_load) return;
        (void)hipStreamSynchronize(stream);
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[load] %-28s %.2f ms\n", what, 1e3 * std::chrono::duration<double>(now - tl0).count());
        tl0 = now;
    };
    loaded = false;
    n0 = num_var; m0 = num_constr; sense = sense_;
    obj_linear = d->obj_linear != 0;
    m_ext = m0 + 1;
    const int64_t nnz0 = m0 ? d->rowptr[m0] : 0;

    // ---- extended structure: rows 0..m0-1 + the objective row f(x) - t over n0+1 variables
    //      (EpigraphNLPEvaluator, src/nlpeval.jl:42-63; only structural non-zeros are stored,
    //       the reference's dense zeros are remembered in pad_zero for round_coefs)
    h_rowptr.assign(d->rowptr, d->rowptr + m0 + 1);
    h_col.assign(d->col, d->col + nnz0);
    // (bulk copies: a batch of 512 instances brings 3e6 entries through here)
    std::vector<uint8_t> akind, padzero(m_ext, 0);
    std::vector<double> p0, p1, rconst(m_ext, 0.0);
    if (d->atom_kind) akind.assign(d->atom_kind, d->atom_kind + nnz0); else akind.assign((size_t)nnz0, 0);
    if (d->p0) p0.assign(d->p0, d->p0 + nnz0); else p0.assign((size_t)nnz0, 0.0);
    if (d->p1) p1.assign(d->p1, d->p1 + nnz0); else p1.assign((size_t)nnz0, 0.0);
    {
        const size_t tail = (size_t)std::max<int64_t>(std::max<int64_t>(d->obj_nnz, d->obj_tape_len), (d->obj_kind == KTN_ROW_HOST ? n0 : 0)) + 2;
        h_col.reserve((size_t)nnz0 + tail); akind.reserve((size_t)nnz0 + tail); p0.reserve((size_t)nnz0 + tail); p1.reserve((size_t)nnz0 + tail);
    }
    h_rowkind.assign(m_ext, KTN_ROW_SEP);
    {
        int32_t cmin = 0, cmax = -1;
        for (int64_t e = 0; e < nnz0; ++e) { const int32_t c = h_col[e]; cmin = std::min(cmin, c); cmax = std::max(cmax, c); }
        KTN_REQUIRE(cmin >= 0 && cmax < n0, "column index out of range");
    }
    for (int64_t i = 0; i < m0; ++i) {
        h_rowkind[i] = d->row_kind ? d->row_kind[i] : KTN_ROW_SEP;
        rconst[i] = d->rconst ? d->rconst[i] : 0.0;
        KTN_REQUIRE(h_rowptr[i + 1] >= h_rowptr[i], "rowptr not monotone");
    }
    // objective row
    std::vector<int32_t> ocols;
    if (d->obj_kind == KTN_ROW_SEP) {
        for (int64_t e = 0; e < d->obj_nnz; ++e) {
            KTN_REQUIRE(d->obj_col[e] >= 0 && d->obj_col[e] < n0, "objective column out of range");
            h_col.push_back(d->obj_col[e]);
            akind.push_back(d->obj_atom_kind ? d->obj_atom_kind[e] : 0);
            p0.push_back(d->obj_p0[e]);
            p1.push_back(d->obj_p1 ? d->obj_p1[e] : 0.0);
        }
        h_col.push_back((int32_t)n0);   // - t
        akind.push_back(KTN_ATOM_LIN);
        p0.push_back(-1.0);
        p1.push_back(0.0);
        rconst[m0] = d->obj_const;
        h_rowkind[m0] = KTN_ROW_SEP;
    } else if (d->obj_kind == KTN_ROW_HOST) {
        // dense row, like the reference's own epigraph row (src/nlpeval.jl:49-54)
        KTN_REQUIRE(d->eval_obj != nullptr, "KTN_ROW_HOST objective without eval_obj callback");
        for (int64_t j = 0; j <= n0; ++j) { h_col.push_back((int32_t)j); akind.push_back(0); p0.push_back(0.0); p1.push_back(0.0); }
        rconst[m0] = 0.0;
        h_rowkind[m0] = KTN_ROW_HOST;
    } else {
        for (int64_t t = 0; t < d->obj_tape_len; ++t)
            if (d->obj_tape_op[t] == KTN_OP_VAR) ocols.push_back((int32_t)d->obj_tape_arg[t]);
        std::sort(ocols.begin(), ocols.end());
        ocols.erase(std::unique(ocols.begin(), ocols.end()), ocols.end());
        for (auto c : ocols) {
            KTN_REQUIRE(c >= 0 && c < n0, "objective tape variable out of range");
            h_col.push_back(c); akind.push_back(0); p0.push_back(0.0); p1.push_back(0.0);
        }
        h_col.push_back((int32_t)n0); akind.push_back(0); p0.push_back(0.0); p1.push_back(0.0);
        rconst[m0] = d->obj_const;
        h_rowkind[m0] = KTN_ROW_TAPE;
    }
    h_rowptr.push_back((int64_t)h_col.size());
    nnz_ext = (int64_t)h_col.size();
    padzero[m0] = (h_rowptr[m0 + 1] - h_rowptr[m0]) < (n0 + 1) ? 1 : 0;

    lapl("extended structure (host)");
    // ---- host-evaluated rows
    cb_rows = d->eval_rows; cb_obj = d->eval_obj; cb_user = d->eval_user;
    host_obj = h_rowkind[m0] == KTN_ROW_HOST;
    host_constr_rows = false;
    {
        std::vector<int32_t> hostrows;
        for (int64_t i = 0; i < m_ext; ++i) {
            KTN_REQUIRE(h_rowkind[i] <= KTN_ROW_HOST, "unknown row kind");
            if (h_rowkind[i] != KTN_ROW_HOST) continue;
            hostrows.push_back((int32_t)i);
            if (i < m0) host_constr_rows = true;
        }
        KTN_REQUIRE(!host_constr_rows || cb_rows != nullptr, "KTN_ROW_HOST rows without eval_rows callback");
        n_host = (int64_t)hostrows.size();
        d_hostrows.upload(hostrows, stream);
        h_gh.assign((size_t)m_ext, 0.0);
        h_jh.assign((size_t)nnz_ext + 1, 0.0);
        h_xh.assign((size_t)n0 + 1, 0.0);
    }

    // ---- tapes -> expression DAGs
    std::vector<int64_t> nodeptr(m_ext + 1, 0);
    std::vector<int32_t> nop, na, nb;
    std::vector<double> nc;
    std::vector<int32_t> tape_all;
    for (int64_t i = 0; i < m_ext; ++i) {
        nodeptr[i] = (int64_t)nop.size();
        if (h_rowkind[i] != KTN_ROW_TAPE) continue;
        tape_all.push_back((int32_t)i);
        const int32_t* rc = h_col.data() + h_rowptr[i];
        const int64_t rl = h_rowptr[i + 1] - h_rowptr[i];
        if (i < m0) {
            KTN_REQUIRE(d->tape_ptr != nullptr, "tape row without tape arrays");
            const int64_t tb = d->tape_ptr[i], te = d->tape_ptr[i + 1];
            postfix_to_nodes(d->tape_op + tb, d->tape_arg + tb, te - tb, rc, rl, h_rowptr[i], nop, na, nb, nc);
        } else {
            std::vector<int32_t> op(d->obj_tape_op, d->obj_tape_op + d->obj_tape_len);
            std::vector<double> arg(d->obj_tape_arg, d->obj_tape_arg + d->obj_tape_len);
            if (op.empty()) { op.push_back(KTN_OP_CONST); arg.push_back(0.0); }
            op.push_back(KTN_OP_VAR); arg.push_back((double)n0);
            op.push_back(KTN_OP_SUB); arg.push_back(0.0);
            postfix_to_nodes(op.data(), arg.data(), (int64_t)op.size(), rc, rl, h_rowptr[i], nop, na, nb, nc);
        }
    }
    nodeptr[m_ext] = (int64_t)nop.size();

    // ---- bounds per extended row; NL row list (model.jl:115-122,144-148)
    h_lb.assign(m_ext, 0.0);
    h_ub.assign(m_ext, 0.0);
    for (int64_t i = 0; i < m0; ++i) { h_lb[i] = l_constr[i]; h_ub[i] = u_constr[i]; }
    h_nlrows.clear();
    std::vector<int64_t> lin_rows;
    for (int64_t i = 0; i < m0; ++i) {
        if (d->row_linear && d->row_linear[i]) lin_rows.push_back(i);
        else h_nlrows.push_back((int32_t)i);
    }
    n_lp = n0;
    std::vector<double> lv(l_var, l_var + n0), uv(u_var, u_var + n0);
    if (!obj_linear) {
        n_lp = n0 + 1;                                  // @variable(m.linear_model, y)  model.jl:137-138
        lv.push_back(-kInf);
        uv.push_back(kInf);
        h_lb[m0] = (sense == KTN_MAX) ? 0.0 : -kInf;    // model.jl:144
        h_ub[m0] = (sense == KTN_MAX) ? kInf : 0.0;
        if (dist.rank == 0) h_nlrows.push_back((int32_t)m0);      // row-sharded: the epigraph row belongs to rank 0
    }
    m_nl = (int64_t)h_nlrows.size();
    has_inf_bound = false;
    for (int64_t j = 0; j < n_lp; ++j)
        if (!std::isfinite(lv[j]) || !std::isfinite(uv[j])) has_inf_bound = true;

    lapl("tapes, bounds, nl list");
    // ---- upload the NLP
    d_rowptr.upload(h_rowptr, stream); d_col.upload(h_col, stream);
    max_row_len = 2;                                   // (the bound-box vertex row and other engine-made rows are short)
    // (a LINEAR objective's row -- up to n entries -- is stored with the structure but never becomes an LP row: counting it made
    //  every LP solve of cfg3 scan for long rows, a launch and a host round trip each)
    for (int64_t i = 0; i < m0 + (obj_linear ? 0 : 1); ++i) max_row_len = std::max(max_row_len, h_rowptr[(size_t)i + 1] - h_rowptr[(size_t)i]);
    {
        // a cut has the sparsity of its NL row: the most entries ONE column can gain per sweep is the number of NL rows that
        // contain it (1-2 on the BASELINE shapes; m_nl for a variable every row shares -- min-max / epigraph-style models)
        std::vector<int32_t> cnt((size_t)n_lp + 1, 0);
        col_gain_max = 0;
        for (int32_t i : h_nlrows)
            for (int64_t e = h_rowptr[(size_t)i]; e < h_rowptr[(size_t)i + 1]; ++e)
                col_gain_max = std::max<int64_t>(col_gain_max, ++cnt[(size_t)h_col[(size_t)e]]);
    }
    {
        // packed row programs: the three arrays go up as they are and are packed on the device (k_pack_atoms)
        uint8_t kmax = 0;
        for (size_t e = 0; e < akind.size(); ++e) kmax = std::max(kmax, akind[e]);
        KTN_REQUIRE(kmax <= KTN_ATOM_NEGLOG, "unknown atom kind");
        const int64_t ne = (int64_t)h_col.size();
        DBuf<uint8_t> t_ak;
        DBuf<double> t_p0, t_p1;
        t_ak.upload(akind, stream); t_p0.upload(p0, stream); t_p1.upload(p1, stream);
        d_colk.resize((size_t)ne, stream); d_pp.resize((size_t)ne, stream);
        LAUNCH_1(k_pack_atoms, ne, stream, ne, d_col.p, t_ak.p, t_p0.p, t_p1.p, d_colk.p, d_pp.p);
        check_launch();
        sync();                                         // the temporaries are freed on leaving the scope
    }
    d_rconst.upload(rconst, stream);
    d_rowkind.upload(h_rowkind, stream); d_padzero.upload(padzero, stream);
    d_lb.upload(h_lb, stream); d_ub.upload(h_ub, stream);
    d_nodeptr.upload(nodeptr, stream); d_nodeop.upload(nop, stream); d_nodea.upload(na, stream);
    d_nodeb.upload(nb, stream); d_nodec.upload(nc, stream);
    d_nodeval.resize(nop.size() + 1, stream); d_nodeadj.resize(nop.size() + 1, stream);
    std::vector<int32_t> allrows(m_ext);
    for (int64_t i = 0; i < m_ext; ++i) allrows[i] = (int32_t)i;
    d_allrows.upload(allrows, stream);
    d_taperows_all.upload(tape_all, stream);
    std::vector<int32_t> tape_nl;
    int64_t nnz_nl = 0;
    for (auto r : h_nlrows) {
        if (h_rowkind[r] == KTN_ROW_TAPE) tape_nl.push_back(r);
        nnz_nl += h_rowptr[r + 1] - h_rowptr[r];
    }
    n_tape_nl = (int64_t)tape_nl.size();
    n_host_nl = 0;
    for (auto r : h_nlrows) n_host_nl += (h_rowkind[r] == KTN_ROW_HOST) ? 1 : 0;
    d_taperows_nl.upload(tape_nl, stream);
    d_nlrows.upload(h_nlrows, stream);
    grp_sweep = pick_group(m_nl ? (double)nnz_nl / (double)m_nl : 4.0);
    if (grp_sweep < 8) grp_sweep = 8;
    lapl("pack + upload NLP");
    // Long rows (hundreds of entries or more): block-major copy for the column-blocked sweep.  Needs every separable
    // NL row sorted by column (the segments are found by binary search).
    {
        // Measured on cfg3_hbm (2e7 entries, 2048 per row): exp/log atoms 156 -> 123 us, quadratic atoms 171 -> 114 us.
        // KTN_SWEEP_BLOCKED=0 switches it off (tests compare the two paths).
        if (dev.blk_cfg >= 0) blk_cfg = dev.blk_cfg;
        blk_cols = (blk_cfg == 2) ? 16384 : 8192;
        blk_wg_per_cu = (blk_cfg == 2) ? 1 : 2;
        const bool env = dev.sweep_blocked >= 0;
        blk_on = m_nl > 0 && n_lp >= 2 * blk_cols && (double)nnz_nl / (double)m_nl >= 256.0;
        if (env) blk_on = blk_on && dev.sweep_blocked != 0;
        blk_nb = ceil_div(n_lp, blk_cols);
        if (blk_on && (double)(m_nl + 1) * blk_nb > 4e8) blk_on = false;
        for (size_t si = 0; blk_on && si < h_nlrows.size(); ++si) {
            const int64_t r = h_nlrows[si];
            if (h_rowkind[r] != KTN_ROW_SEP) continue;
            for (int64_t e = h_rowptr[r] + 1; e < h_rowptr[r + 1]; ++e)
                if (h_col[e] < h_col[e - 1]) { blk_on = false; break; }
        }
        d_bcolk.release(); d_bpp.release(); d_bseg.release(); d_bkind.release(); d_part.release(); d_slots.release();
        if (blk_on) {
            std::vector<int64_t> bseg((size_t)(m_nl + 1) * blk_nb);
            std::vector<int4> bkind((size_t)(m_nl + 1) * blk_nb, make_int4(0, 0, 0, 0));
            std::vector<int32_t> bcolk((size_t)nnz_nl);
            std::vector<double2> bpp((size_t)nnz_nl);
            // cut[si * (NB + 1) + b]: first entry of row si with column >= b * blk_cols
            std::vector<int64_t> cut((size_t)m_nl * (blk_nb + 1));
            for (int64_t si = 0; si < m_nl; ++si) {
                const int64_t r = h_nlrows[si];
                const int32_t* cb = h_col.data() + h_rowptr[r];
                const int32_t* ce = (h_rowkind[r] == KTN_ROW_SEP) ? h_col.data() + h_rowptr[r + 1] : cb;   // tape rows: empty
                for (int b = 0; b <= blk_nb; ++b) {
                    const int64_t c0 = (int64_t)b * blk_cols;
                    cut[(size_t)si * (blk_nb + 1) + b] =
                        h_rowptr[r] + (std::lower_bound(cb, ce, c0, [](int32_t a, int64_t v) { return (int64_t)a < v; }) - cb);
                }
            }
            int64_t w = 0;
            for (int b = 0; b < blk_nb; ++b) {
                for (int64_t si = 0; si < m_nl; ++si) {
                    bseg[(size_t)b * (m_nl + 1) + si] = w;
                    const int64_t eb = cut[(size_t)si * (blk_nb + 1) + b], ee = cut[(size_t)si * (blk_nb + 1) + b + 1];
                    const int64_t w0 = w;
                    int32_t kstart[KTN_ATOM_NEGLOG + 1];
                    for (int kd = 0; kd <= KTN_ATOM_NEGLOG; ++kd) {      // segment grouped by atom kind (stable)
                        kstart[kd] = (int32_t)(w - w0);
                        for (int64_t e = eb; e < ee; ++e) {
                            if (akind[e] != kd) continue;
                            bcolk[(size_t)w] = h_col[e] | ((int32_t)kd << kKindShift);
                            bpp[(size_t)w] = make_double2(p0[e], p1[e]);
                            ++w;
                        }
                    }
                    bkind[(size_t)b * (m_nl + 1) + si] = make_int4(kstart[KTN_ATOM_QUAD], kstart[KTN_ATOM_EXP], kstart[KTN_ATOM_NEGLOG], 0);
                }
                bseg[(size_t)b * (m_nl + 1) + m_nl] = w;
            }
            d_bcolk.upload(bcolk.data(), (size_t)w, stream);
            d_bpp.upload(bpp.data(), (size_t)w, stream);
            std::vector<SepSlot> slots((size_t)m_nl);
            for (int64_t si = 0; si < m_nl; ++si) {
                const int64_t r = h_nlrows[si];
                SepSlot sl;
                sl.rconst = rconst[r]; sl.lb = h_lb[r]; sl.ub = h_ub[r];
                sl.row = (h_rowkind[r] == KTN_ROW_SEP) ? (int32_t)r : -1;
                sl.len_pad = (int32_t)(((h_rowptr[r + 1] - h_rowptr[r]) << 1) | (padzero[r] ? 1 : 0));
                slots[(size_t)si] = sl;
            }
            d_slots.upload(slots, stream);
            d_bseg.upload(bseg, stream);
            d_bkind.upload(bkind, stream);
            d_part.resize((size_t)m_nl * blk_nb, stream);
        }
    }
    // Very long separable rows get the device-side kind kRowSepLong and their own kernel (kernels.hpp k_sep_eval_long) -- not under
    // the column-blocked sweep, which is the long-row path of the NL rows and reads the host-side kinds
    {
        std::vector<uint8_t> dk(h_rowkind);
        std::vector<int32_t> lr, lnr;
        std::vector<int64_t> ls, lns;
        std::vector<int64_t> slot_of((size_t)m_ext, -1);
        for (int64_t si = 0; si < m_nl; ++si) slot_of[(size_t)h_nlrows[(size_t)si]] = si;
        for (int64_t i = 0; i < m_ext && !blk_on; ++i) {
            if (h_rowkind[i] != KTN_ROW_SEP || h_rowptr[(size_t)i + 1] - h_rowptr[(size_t)i] <= kLongEval) continue;
            dk[(size_t)i] = kRowSepLong;
            lr.push_back((int32_t)i); ls.push_back(slot_of[(size_t)i]);
            if (slot_of[(size_t)i] >= 0) { lnr.push_back((int32_t)i); lns.push_back(slot_of[(size_t)i]); }
        }
        n_longev = (int64_t)lr.size(); n_longev_nl = (int64_t)lnr.size();
        stats["sep_long_rows"] = (double)n_longev;
        if (n_longev > 0) {
            d_rowkind.upload(dk, stream);
            d_longev_rows.upload(lr, stream); d_longev_slots.upload(ls, stream);
            d_longev_nlrows.upload(lnr, stream); d_longev_nlslots.upload(lns, stream);
        }
    }
    // Many short rows: the batch-blocked copy (kernels.hpp k_sep_sweep_batch).  Built on the host in two counting passes over the
    // NL entries -- bucket (batch of 2 048 slots, block of 8 192 columns, atom kind), rows ascending inside a bucket because
    // the slots are visited in order -- 20 B per entry.
    {
        d_sbck.release(); d_sbrow.release(); d_sbpp.release(); d_sbseg.release();
        sb_on = !blk_on && 2 * m_nl >= (int64_t)3 * kSbRows * num_cus && n_tape_nl == 0 && n_host_nl == 0 && (double)nnz_nl / (double)std::max<int64_t>(m_nl, 1) <= 128.0 &&
                n_lp <= (int64_t)kSbCols * 64;
        if (dev.sweep_batched == 0) sb_on = false;
        if (dev.sweep_batched == 1) sb_on = m_nl > 0 && n_tape_nl == 0 && n_host_nl == 0 && n_lp <= (int64_t)kSbCols * 64 && !blk_on;
        if (sb_on) {
            sb_nb = ceil_div(n_lp, (int64_t)kSbCols);
            sb_batches = ceil_div(m_nl, (int64_t)kSbRows);
            const size_t nbuck = (size_t)sb_batches * sb_nb * 4;
            std::vector<int64_t> segs(nbuck + 4, 0);
            for (int64_t si = 0; si < m_nl; ++si) {
                const int64_t r = h_nlrows[(size_t)si];
                const size_t base = (size_t)(si / kSbRows) * sb_nb * 4;
                for (int64_t e = h_rowptr[r]; e < h_rowptr[r + 1]; ++e)
                    ++segs[base + (size_t)(h_col[e] / kSbCols) * 4 + akind[e] + 1];
            }
            for (size_t k = 1; k < segs.size(); ++k) segs[k] += segs[k - 1];
            std::vector<int64_t> cur(segs.begin(), segs.begin() + nbuck);
            std::vector<uint16_t> sck((size_t)nnz_nl), srw((size_t)nnz_nl);
            std::vector<double2> spp((size_t)nnz_nl);
            for (int64_t si = 0; si < m_nl; ++si) {
                const int64_t r = h_nlrows[(size_t)si];
                const size_t base = (size_t)(si / kSbRows) * sb_nb * 4;
                const uint16_t rl = (uint16_t)(si % kSbRows);
                for (int64_t e = h_rowptr[r]; e < h_rowptr[r + 1]; ++e) {
                    const int64_t bl = h_col[e] / kSbCols;
                    const int64_t w = cur[base + (size_t)bl * 4 + akind[e]]++;
                    sck[(size_t)w] = (uint16_t)(h_col[e] - bl * kSbCols);
                    srw[(size_t)w] = rl;
                    spp[(size_t)w] = make_double2(p0[e], p1[e]);
                }
            }
            segs.resize(nbuck + 1);
            d_sbck.upload(sck, stream); d_sbrow.upload(srw, stream); d_sbpp.upload(spp, stream); d_sbseg.upload(segs, stream);
            sync();
        }
        stats["sweep_batched"] = sb_on ? 1.0 : 0.0;
    }
    // algorithmic bytes of one evaluation pass over the NL rows (DESIGN.md "sweep bytes")
    sweep_bytes = (double)nnz_nl * (4 + 16) + 8.0 * (m_nl + 1) + 8.0 * n_lp + 8.0 * 4 * m_nl + 16.0 * m_nl;
    const size_t mm = (size_t)std::max<int64_t>(m_ext, 1);
    d_g.resize(mm, stream); d_bconst.resize(mm, stream); d_maxc.resize(mm, stream); d_nonfin.resize(mm, stream);
    d_jac.resize((size_t)nnz_ext + 1, stream);
    d_flag.resize(mm, stream); d_cnt.resize(mm, stream); d_rank.resize(mm, stream); d_cntscan.resize(mm, stream);
    d_lastcut.resize(mm, stream);
    d_xs.resize((size_t)n0 + 1, stream); d_ray.resize((size_t)n0 + 1, stream);
    d_flag.zero(stream); d_cnt.zero(stream);

    lapl("blocked copy + sweep buffers");
    // ---- tangent at the origin: linear rows and (linear) objective  model.jl:110-133
    d_xs.zero(stream);
    precompute_all(d_xs.p);
    // the LP rows of the linear constraints are written on the device (k_lin_rows): only the row pointers -- structural --
    // come from the host; the objective row's slice of (g, J) is all that travels back
    std::vector<int64_t> rp(1, 0);
    std::vector<int32_t> rc;                            // host-built rows (the epigraph cut at the vertex) follow the linear rows
    std::vector<double> rv, rlo, rhi;
    numcuts = 0;
    const int64_t n_lin = (int64_t)lin_rows.size();
    int64_t nnz_lin = 0;
    {
        rp.reserve((size_t)n_lin + 2);
        std::vector<int32_t> lr((size_t)n_lin);
        for (int64_t k = 0; k < n_lin; ++k) {
            const int64_t i = lin_rows[(size_t)k];
            lr[(size_t)k] = (int32_t)i;
            nnz_lin += h_rowptr[i + 1] - h_rowptr[i];
            rp.push_back(nnz_lin);
            numcuts += 1;                               // model.jl:77
        }
        lp_rowptr.resize((size_t)n_lin + 2, stream);
        KTN_HIP(hipMemcpyAsync(lp_rowptr.p, rp.data(), rp.size() * sizeof(int64_t), hipMemcpyHostToDevice, stream));
        lp_col.resize((size_t)nnz_lin + 1, stream); lp_val.resize((size_t)nnz_lin + 1, stream);
        lp_lo.resize((size_t)n_lin + 1, stream); lp_hi.resize((size_t)n_lin + 1, stream);
        DBuf<int32_t> t_lr;
        t_lr.upload(lr, stream);
        LAUNCH_1(k_lin_rows, n_lin, stream, n_lin, t_lr.p, d_rowptr.p, d_col.p, d_jac.p, d_g.p, d_lb.p, d_ub.p, lp_rowptr.p, lp_col.p,
                 lp_val.p, lp_lo.p, lp_hi.p);
        check_launch();
        sync();
    }
    // objective row at the origin: g0[m0] and its Jacobian entries
    const int64_t ob = h_rowptr[m0], ol = h_rowptr[m0 + 1] - ob;
    std::vector<double> j0obj((size_t)std::max<int64_t>(ol, 1));
    double g0obj = 0.0;
    if (ol > 0) KTN_HIP(hipMemcpyAsync(j0obj.data(), d_jac.p + ob, (size_t)ol * sizeof(double), hipMemcpyDeviceToHost, stream));
    KTN_HIP(hipMemcpyAsync(&g0obj, d_g.p + m0, sizeof(double), hipMemcpyDeviceToHost, stream));
    sync();
    lapl("tangent at origin + LP rows (host)");
    std::vector<double> cvec(n_lp, 0.0);
    c0 = 0.0;
    if (prm.log_level > 0) { std::printf(obj_linear ? "objective is linear\n" : "objective is nonlinear\n"); std::fflush(stdout); }   // model.jl:127,135
    if (obj_linear) {
        // gencut(fsep, pt, (0,0), num_constr+1), drop the fictitious aux variable  model.jl:129-133
        for (int64_t e = h_rowptr[m0]; e < h_rowptr[m0 + 1]; ++e)
            if (h_col[e] < n0) cvec[h_col[e]] += j0obj[(size_t)(e - ob)];
        c0 = g0obj;
    } else {
        cvec[n0] = 1.0;                                 // @objective(m.linear_model, sense, y)  model.jl:139
        // initial epigraph cut at the bound-box vertex  model.jl:93-97,156-164
        bool ok = true;
        std::vector<double> vtx(n0 + 1, 0.0);
        for (int64_t j = 0; j < n0; ++j) {
            const double lo = lv[j], hi = uv[j];
            if (lo > hi) ok = false;
            const bool lf = std::isfinite(lo), uf = std::isfinite(hi);
            if (lf && uf) vtx[j] = (std::fabs(lo) <= std::fabs(hi)) ? lo : hi;   // GLPK non-basic rule
            else if (lf) vtx[j] = lo;
            else if (uf) vtx[j] = hi;
            else vtx[j] = 0.0;
        }
        if (!ok) std::fprintf(stderr, "WARNING: Problem variables insufficiently bounded!\n");      // model.jl:156-157
        if (ok && dist.rank == 0) {
            KTN_HIP(hipMemcpyAsync(d_xs.p, vtx.data(), (n0 + 1) * sizeof(double), hipMemcpyHostToDevice, stream));
            precompute_all(d_xs.p);
            std::vector<double> g1 = d_g.to_host(stream);
            vtx[n0] = g1[m0];                           // push!(vertex, eval_f(d, vertex))
            KTN_HIP(hipMemcpyAsync(d_xs.p, vtx.data(), (n0 + 1) * sizeof(double), hipMemcpyHostToDevice, stream));
            precompute_all(d_xs.p);
            g1 = d_g.to_host(stream);
            std::vector<double> j1 = d_jac.to_host(stream);
            double b = g1[m0];
            std::vector<double> coef;
            double mx = -kInf;
            bool finite = true;
            for (int64_t e = h_rowptr[m0]; e < h_rowptr[m0 + 1]; ++e) {
                coef.push_back(j1[e]);
                b += -vtx[h_col[e]] * j1[e];
                if (!(j1[e] <= mx)) mx = (j1[e] != j1[e]) ? j1[e] : std::max(mx, j1[e]);
                if (!std::isfinite(j1[e])) finite = false;
            }
            if (padzero[m0] && !(mx != mx)) mx = std::max(mx, 0.0);
            for (auto& cf : coef) if (cf + prm.cut_coef_rng < mx) cf = 0.0;   // round_coefs
            if (!finite) {
                std::fprintf(stderr, "WARNING: Nonlinear constraint or objective likely undefined within domain\n");   // model.jl:70
                status = KTN_STATUS_ERROR;              // _addcut: warn + :Error, no row added
            } else {
                for (int64_t e = h_rowptr[m0]; e < h_rowptr[m0 + 1]; ++e) {
                    rc.push_back(h_col[e]);
                    rv.push_back(coef[e - h_rowptr[m0]]);
                }
                rp.push_back(nnz_lin + (int64_t)rc.size());
                rlo.push_back(h_lb[m0] - b);
                rhi.push_back(h_ub[m0] - b);
                numcuts += 1;
            }
        }
    }
    lapl("objective");
    // ---- LP: the linear rows are in place (device); rows built on the host (the epigraph cut at the vertex) are appended
    M = n_lin + (int64_t)rlo.size();
    NNZ = nnz_lin + (int64_t)rc.size();
    if (!rlo.empty()) {
        lp_rowptr.resize((size_t)M + 1, stream); lp_col.resize((size_t)NNZ + 1, stream); lp_val.resize((size_t)NNZ + 1, stream);
        lp_lo.resize((size_t)M, stream); lp_hi.resize((size_t)M, stream);
        KTN_HIP(hipMemcpyAsync(lp_rowptr.p + n_lin + 1, rp.data() + n_lin + 1, rlo.size() * sizeof(int64_t), hipMemcpyHostToDevice, stream));
        KTN_HIP(hipMemcpyAsync(lp_col.p + nnz_lin, rc.data(), rc.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream));
        KTN_HIP(hipMemcpyAsync(lp_val.p + nnz_lin, rv.data(), rv.size() * sizeof(double), hipMemcpyHostToDevice, stream));
        KTN_HIP(hipMemcpyAsync(lp_lo.p + n_lin, rlo.data(), rlo.size() * sizeof(double), hipMemcpyHostToDevice, stream));
        KTN_HIP(hipMemcpyAsync(lp_hi.p + n_lin, rhi.data(), rhi.size() * sizeof(double), hipMemcpyHostToDevice, stream));
        sync();
    }
    lp_rowptr.n = (size_t)M + 1; lp_col.n = lp_val.n = (size_t)NNZ; lp_lo.n = lp_hi.n = (size_t)M;
    lp_y.resize((size_t)M, stream); lp_y.zero(stream);
    lp_c.upload(cvec, stream); lp_l.upload(lv, stream); lp_u.upload(uv, stream);
    lp_x.resize((size_t)n_lp, stream); lp_x.zero(stream);
    M_base = M; NNZ_base = NNZ; numcuts_base = numcuts; M_lin = n_lin;
    n_longc = 0; col_len_max = -1; col_scan_rows = 0; col_removed_rows = 0;      // (a new problem: scan its columns at the first solve)
    {
        // room for three sweeps' worth of cuts (each sweep adds at most min(m_nl, cut cap) rows)
        int64_t per_sweep = m_nl;
        if (prm.cut_cap_factor > 0.0)
            per_sweep = std::min<int64_t>(m_nl, std::max<int64_t>((int64_t)(prm.cut_cap_factor * (double)n_lp), prm.cut_cap_min));
        const double avg_nl = m_nl ? (double)nnz_nl / (double)m_nl : 0.0;
        const int64_t rows = M + 3 * per_sweep;
        const int64_t nz = NNZ + (int64_t)(3.0 * (double)per_sweep * avg_nl);
        if ((double)rows * 200.0 + (double)nz * 60.0 < 64e9) reserve_lp(rows, nz);     // stay far below the 288 GB
        d_violslots.reserve((size_t)std::max<int64_t>(m_nl, 1), stream);
    }
    sync();
    lapl("LP upload + reserve");
    loaded = true;
    const int keep_status = status;
    reset();
    lapl("reset");
    if (keep_status == KTN_STATUS_ERROR) status = KTN_STATUS_ERROR;
    // row-sharded: what steers the sequence of collectives must be the same on every rank -- the number of NL rows (a rank
    // whose shard has none would otherwise take the pure-LP tolerance and leave the others' restart pattern) and the
    // load-time error (the vertex cut of the epigraph row is built on rank 0 only)
    m_nl_global = m_nl;
    if (row_sharded()) {
        double cnt = (double)m_nl, bad = (status == KTN_STATUS_ERROR) ? 1.0 : 0.0;
        allreduce_host(&cnt, 1, 0);
        allreduce_host(&bad, 1, 1);
        m_nl_global = (int64_t)(cnt + 0.5);
        if (bad > 0.0) status = KTN_STATUS_ERROR;
    }
}

void Engine::reset() {
    M = M_base; NNZ = NNZ_base; numcuts = numcuts_base;
    lp_rowptr.n = (size_t)M + 1; lp_col.n = (size_t)NNZ; lp_val.n = (size_t)NNZ;
    lp_lo.n = lp_hi.n = lp_y.n = (size_t)M;
    lp_y.zero(stream); lp_x.zero(stream);
    std::vector<int64_t> neg1((size_t)std::max<int64_t>(m_ext, 1), -1);
    d_lastcut.upload(neg1, stream);
    d_age.resize((size_t)std::max<int64_t>(M, 1), stream);
    d_age.zero(stream);
    lp_dirty = true; ++lp_version; ++lp_epoch; have_omega = false; have_precompute = false; sharded_rows = false; scal_rows = 0; smax_rows = 0; n_longc = 0; col_len_max = -1; col_scan_rows = 0; col_removed_rows = 0;
    blocks_built_rows = -1;
    if (d_blkomega.n) d_blkomega.zero(stream);
    if (ds_valid.n) ds_valid.zero(stream);
    dense_credit = dense_run = 0;
    md_valid = false; mid_credit = mid_run = 0; mid_backoff = mid_backoff_len = 0;
    if (glists) KTN_HIP(hipMemsetAsync(d_glast.p, 0xFF, d_glast.n * sizeof(int64_t), stream));
    last_sweep_cuts = 0;
    power_v.n = 0;
    status = KTN_STATUS_NONE; lp_status = KTN_STATUS_OPTIMAL;
    iter = 0; soltime = 0.0; objval = std::numeric_limits<double>::quiet_NaN();
    last_maxviol = 1e300; obj_prev = kInf; allsat = false; begun = false; tight_done = false;
    log_cuts_lastprnt = 0; log_max_viol = 0; purged_total = 0;
    polishing = false; polish_done = false; polish_count = 0; best_viol = kInf; best_obj = 0.0; cert_target = 0.0;
    lp_sols.clear();
    sync();
}

// ------------------------------------------------------------------------------------
// LP: column mirror, scaling, PDHG
// ------------------------------------------------------------------------------------
void Engine::rebuild_csc() {
    const bool no_merge = dev.no_csc_merge;       // (tests: the sort path for every solve)
    c_val.resize((size_t)NNZ + 1, stream);
    // (long columns: the merge orders a column's new entries by insertion -- fine for the 0.3 entries a column gains per sweep,
    //  quadratic for a column that gains one per cut; the radix sort does not care)
    // (host-appended rows: the other ranks' cuts come on top -- as many ranks as the global NL-row count says, else 8)
    const int64_t ranks = !sharded_rows ? 1 : (glists && m_nl > 0 ? (nl_total + m_nl - 1) / m_nl : 8);
    const bool few_per_col = col_gain_max * ranks <= 128;
    if (!no_merge && n_longc == 0 && few_per_col && csc_epoch == lp_epoch && csc_M >= 0 && M >= csc_M && NNZ >= csc_NNZ && NNZ < ((int64_t)1 << 32)) {
        if (M > csc_M) csc_merge_appended();              // (M == csc_M: same structure, only the values are gathered again)
        stats["lp_csc_merges"] += 1.0;
    } else {
        c_ptr.resize((size_t)n_lp + 1, stream);
        c_cnt.resize((size_t)n_lp + 1, stream);
        c_cnt.zero(stream);
        c_row.resize((size_t)NNZ + 1, stream);
        c_perm.resize((size_t)NNZ + 1, stream);
        if (NNZ > 0) {
            k_in.resize((size_t)NNZ, stream); k_out.resize((size_t)NNZ, stream);
            p_in.resize((size_t)NNZ, stream); p_out.resize((size_t)NNZ, stream);
            LAUNCH_G(pick_group((double)NNZ / (double)std::max<int64_t>(M, 1)), k_csc_keys, M, stream, M, lp_rowptr.p, lp_col.p, k_in.p,
                     p_in.p, c_cnt.p);
            check_launch();
        }
        exclusive_scan(c_cnt.p, c_ptr.p, (size_t)n_lp + 1);
        if (NNZ > 0) {
            int bits = 1;
            while (((int64_t)1 << bits) < n_lp + 1 && bits < 31) ++bits;
            size_t need = sort_pairs_temp_bytes((size_t)NNZ);
            d_sorttmp.resize(need + 16, stream);
            // keys are (col << 32 | row) in CSR order, i.e. already ascending in row: a STABLE sort on the column bits alone
            // gives (col, row) order in 3 radix passes instead of 7
            KTN_HIP(sort_pairs_u64_u32(d_sorttmp.p, need, k_in.p, k_out.p, p_in.p, p_out.p, (size_t)NNZ, 32, 32 + bits, stream));
            LAUNCH_1(k_csc_rows_perm, NNZ, stream, NNZ, k_out.p, p_out.p, c_row.p, c_perm.p);
            check_launch();
        }
        stats["lp_csc_sorts"] += 1.0;
    }
    LAUNCH_1(k_csc_vals, NNZ, stream, NNZ, c_perm.p, Wval(), c_val.p);
    check_launch();
    csc_epoch = lp_epoch; csc_M = M; csc_NNZ = NNZ;
    lp_dirty = false;
    blocks_built_rows = -1;
    find_long_cols();
}
// Columns longer than kLongRow.  A column gains at most one entry per appended row, so between scans the longest possible
// column is known on the host: no scan (and no round trip) while that bound stays below the threshold.
void Engine::find_long_cols() {
    if (col_len_max >= 0 && n_longc == 0 && col_len_max + (M - col_scan_rows) + col_removed_rows <= kLongRow) return;
    n_longc = 0;
    col_removed_rows = 0;
    if (M <= kLongRow || n_blocks > 0) { col_len_max = std::min<int64_t>(M, kLongRow); col_scan_rows = M; return; }
    d_longcols.resize((size_t)n_lp, stream);
    KTN_HIP(hipMemsetAsync(d_anynf.p + 1, 0, 2 * sizeof(int32_t), stream));
    LAUNCH_1(k_find_long_max, n_lp, stream, n_lp, c_ptr.p, kLongRow, d_longcols.p, d_anynf.p + 1);
    int32_t r[2] = {0, 0};
    KTN_HIP(hipMemcpyAsync(r, d_anynf.p + 1, 8, hipMemcpyDeviceToHost, stream));
    sync();
    n_longc = r[0];
    col_len_max = r[1];
    col_scan_rows = M;
    stats["lp_long_col_scans"] += 1.0;
    stats["lp_long_cols"] = (double)n_longc;
    stats["lp_long_cols_max"] = std::max(stats["lp_long_cols_max"], (double)n_longc);
    if (n_longc > 1) {                                  // list order = order of the workgroups' sums: make it reproducible
        std::vector<int32_t> tmp((size_t)n_longc);
        KTN_HIP(hipMemcpyAsync(tmp.data(), d_longcols.p, (size_t)n_longc * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
        sync();
        std::sort(tmp.begin(), tmp.end());
        KTN_HIP(hipMemcpyAsync(d_longcols.p, tmp.data(), (size_t)n_longc * sizeof(int32_t), hipMemcpyHostToDevice, stream));
        sync();
    }
}
// out = A'v over the mirror: G lanes per column, a 1024-thread workgroup per long column
void Engine::spmv_cols(const SpMat& AT, const double* v, double* out, hipEvent_t e0, hipEvent_t e1) {
    const int64_t n = n_lp;
    if (n_longc > 0) {
        LAUNCH_G(grp_cols, k_spmv_skip, n, stream, n, AT, v, out, kLongRow);
        hipLaunchKernelGGL(k_spmv_long, dim3((unsigned)n_longc), dim3(1024), 0, stream, d_longcols.p, AT, v, out);
    } else if (e0) {
        LAUNCH_G_EV(grp_cols, k_spmv, n, stream, e0, e1, n, AT, v, out);
    } else {
        LAUNCH_G(grp_cols, k_spmv, n, stream, n, AT, v, out);
    }
}
// rows [csc_M, M) were appended since the mirror was built (kernels.hpp "append-only update of the mirror")
void Engine::csc_merge_appended() {
    const int64_t n = n_lp, nnz_new = NNZ - csc_NNZ;
    c_off.resize((size_t)n + 1, stream);
    c_cnt.resize((size_t)n + 1, stream);
    c_cnt.zero(stream);
    LAUNCH_1(k_cscm_count, nnz_new, stream, csc_NNZ, NNZ, lp_col.p, c_cnt.p);
    exclusive_scan(c_cnt.p, c_off.p, (size_t)n + 1);
    c_ptr2.resize((size_t)n + 1, stream);
    c_row2.resize((size_t)NNZ + 1, stream);
    c_perm2.resize((size_t)NNZ + 1, stream);
    const int gc = pick_group((double)csc_NNZ / (double)std::max<int64_t>(n, 1));
    LAUNCH_G(gc, k_cscm_move, n + 1, stream, n, c_ptr.p, c_off.p, c_row.p, c_perm.p, c_ptr2.p, c_row2.p, c_perm2.p);
    c_cnt.zero(stream);
    LAUNCH_G(pick_group((double)nnz_new / (double)std::max<int64_t>(M - csc_M, 1)), k_cscm_place, M - csc_M, stream, csc_M, M, lp_rowptr.p, lp_col.p, c_ptr.p,
             c_off.p, c_cnt.p, c_row2.p, c_perm2.p);
    LAUNCH_1(k_cscm_order, n, stream, n, c_ptr.p, c_off.p, c_row2.p, c_perm2.p);
    check_launch();
    c_ptr.swap(c_ptr2); c_row.swap(c_row2); c_perm.swap(c_perm2);
}

// The matrix a solve works on: the stored LP, or its epigraph-shifted working form (kernels.hpp "epigraph reference
// shift"); rebuilt -- together with the column mirror -- when rows changed or the form toggles.
void Engine::ensure_matrix(bool shift) {
    if (shift != w_shift) { lp_dirty = true; ++lp_version; }
    if (!lp_dirty) return;
    w_shift = shift;
    if (shift) build_working();
    rebuild_csc();
}
void Engine::build_working() {
    const int32_t tcol = (int32_t)n0;
    const size_t mm = (size_t)std::max<int64_t>(M, 1);
    wval.resize((size_t)NNZ + 1, stream); wlo.resize(mm, stream); whi.resize(mm, stream);
    wc.resize((size_t)n_lp, stream); epi_ref.resize((size_t)n_lp, stream); epi_scal.resize(2, stream); epi_newest.resize(1, stream);
    if (NNZ > 0) KTN_HIP(hipMemcpyAsync(wval.p, lp_val.p, (size_t)NNZ * sizeof(double), hipMemcpyDeviceToDevice, stream));
    if (M > 0) {
        KTN_HIP(hipMemcpyAsync(wlo.p, lp_lo.p, (size_t)M * sizeof(double), hipMemcpyDeviceToDevice, stream));
        KTN_HIP(hipMemcpyAsync(whi.p, lp_hi.p, (size_t)M * sizeof(double), hipMemcpyDeviceToDevice, stream));
    }
    epi_ref.zero(stream); epi_scal.zero(stream); epi_newest.zero(stream);
    const int64_t rows = M - M_lin;
    LAUNCH_1(k_epi_newest, rows, stream, M_lin, M, lp_rowptr.p, lp_col.p, tcol, epi_newest.p);
    if (rows > 0) {
        hipLaunchKernelGGL(k_epi_setref, dim3(64), dim3(kBlock), 0, stream, epi_newest.p, lp_rowptr.p, lp_col.p, lp_val.p, lp_lo.p, lp_hi.p,
                           tcol, epi_ref.p, epi_scal.p);
        hipLaunchKernelGGL(k_epi_shift, dim3((unsigned)(rows * kEpiChunks)), dim3(kBlock), 0, stream, M_lin, M, lp_rowptr.p, lp_col.p,
                           lp_val.p, lp_lo.p, lp_hi.p, tcol, epi_ref.p, epi_scal.p, wval.p, wlo.p, whi.p);
    }
    LAUNCH_1(k_epi_cost, n_lp, stream, n_lp, lp_c.p, tcol, epi_ref.p, wc.p);
    check_launch();
    stats["lp_epi_shifts"] += 1.0;
}

// Cut-pool management after an LP solve (k_purge_mark / k_purge_copy / k_purge_relink).
void Engine::purge_cuts() {
    const int64_t m = M;
    d_age.resize((size_t)m, stream);
    d_keep.resize((size_t)m, stream); d_keepnnz.resize((size_t)m, stream);
    d_newidx.resize((size_t)m, stream); d_newptr.resize((size_t)m, stream);
    const int gp = pick_group((double)NNZ / (double)std::max<int64_t>(m, 1));      // lanes per row of the pool kernels
    LAUNCH_G(gp, k_purge_mark, m, stream, M_base, m, lp_rowptr.p, lp_col.p, lp_val.p, lp_x.p, lp_lo.p, lp_hi.p, lp_y.p, d_age.p,
             prm.purge_margin, (int)prm.purge_age, d_keep.p, d_keepnnz.p);
    bool deduped = false;
    if (prm.dedupe_eps > 0.0 && lists_ok() && list_count() > 0) {
        KTN_HIP(hipMemsetAsync(d_anynf.p + 1, 0, sizeof(int32_t), stream));
        LAUNCH_G(gp, k_dedupe_mark, list_count(), stream, list_count(), list_heads(), d_cutprev.p, lp_rowptr.p, lp_val.p, lp_lo.p, lp_hi.p, lp_y.p,
                 prm.dedupe_eps, d_keep.p, d_keepnnz.p, d_anynf.p + 1);
        deduped = true;
    }
    check_launch();
    exclusive_scan(d_keep.p, d_newidx.p, (size_t)m);
    exclusive_scan(d_keepnnz.p, d_newptr.p, (size_t)m);
    int64_t t[4];
    int32_t nd = 0;
    if (h_chk_dev) {                                   // the four scan tails and the dedupe count in one round trip
        double* ht = h_chk + 2 * kChkQ + 8;
        hipLaunchKernelGGL(k_host_tail, dim3(1), dim3(1), 0, stream, h_chk_dev + 2 * kChkQ + 8, d_keep.p + (m - 1), d_newidx.p + (m - 1),
                           d_keepnnz.p + (m - 1), d_newptr.p + (m - 1), (const double*)nullptr,
                           deduped ? (const int32_t*)(d_anynf.p + 1) : (const int32_t*)nullptr, (const int32_t*)nullptr);
        sync();
        for (int k = 0; k < 4; ++k) t[k] = (int64_t)ht[k];
        nd = (int32_t)ht[5];
    } else {
        if (deduped) KTN_HIP(hipMemcpyAsync(&nd, d_anynf.p + 1, 4, hipMemcpyDeviceToHost, stream));
        KTN_HIP(hipMemcpyAsync(&t[0], d_keep.p + (m - 1), 8, hipMemcpyDeviceToHost, stream));
        KTN_HIP(hipMemcpyAsync(&t[1], d_newidx.p + (m - 1), 8, hipMemcpyDeviceToHost, stream));
        KTN_HIP(hipMemcpyAsync(&t[2], d_keepnnz.p + (m - 1), 8, hipMemcpyDeviceToHost, stream));
        KTN_HIP(hipMemcpyAsync(&t[3], d_newptr.p + (m - 1), 8, hipMemcpyDeviceToHost, stream));
        sync();
    }
    stats["deduped_rows"] += (double)nd;
    const int64_t m_new = t[0] + t[1], nnz_new = t[2] + t[3];
    if (m - m_new < (int64_t)(prm.purge_min_frac * (double)m) || m_new == m) return;
    lp_rowptr2.resize((size_t)m_new + 1, stream); lp_col2.resize((size_t)nnz_new + 1, stream);
    lp_val2.resize((size_t)nnz_new + 1, stream); lp_lo2.resize((size_t)m_new, stream); lp_hi2.resize((size_t)m_new, stream);
    lp_y2.resize((size_t)m_new, stream); d_age2.resize((size_t)m_new, stream); d_cutprev2.resize((size_t)m_new, stream);
    LpRows Old = lp_view();
    LpRows New{lp_rowptr2.p, lp_col2.p, lp_val2.p, lp_lo2.p, lp_hi2.p, lp_y2.p};
    LAUNCH_G(gp, k_purge_copy, m, stream, m, d_keep.p, d_newidx.p, d_newptr.p, Old, d_age.p, New, d_age2.p);
    KTN_HIP(hipMemcpyAsync(lp_rowptr2.p + m_new, &nnz_new, 8, hipMemcpyHostToDevice, stream));
    if (lists_ok()) {
        LAUNCH_1(k_purge_relink, list_count(), stream, list_count(), list_heads(), d_cutprev.p, d_keep.p, d_newidx.p, d_cutprev2.p);
    } else {
        // rows appended from the host (multi-GPU exchange) are not threaded into the per-row cut lists, and the
        // lists are unused in that mode (no dual inheritance, no consolidation): void them
        KTN_HIP(hipMemsetAsync(d_lastcut.p, 0xFF, d_lastcut.n * sizeof(int64_t), stream));
        KTN_HIP(hipMemsetAsync(d_cutprev2.p, 0xFF, (size_t)m_new * sizeof(int64_t), stream));
    }
    if (scal_rows == m && prm.lp_ruiz_warm > 0 && dr_r.n >= (size_t)m) {   // keep the row scaling of the surviving rows (warm start of the next solve)
        statr.resize((size_t)m, stream);
        LAUNCH_1(k_compact_vec, m, stream, m, d_keep.p, d_newidx.p, dr_r.p, statr.p);
        dr_r.swap(statr);
        scal_rows = m_new;
    } else {
        scal_rows = 0;
    }
    check_launch();
    sync();
    lp_rowptr.swap(lp_rowptr2); lp_col.swap(lp_col2); lp_val.swap(lp_val2); lp_lo.swap(lp_lo2); lp_hi.swap(lp_hi2);
    lp_y.swap(lp_y2); d_age.swap(d_age2); d_cutprev.swap(d_cutprev2);
    lp_rowptr.n = (size_t)m_new + 1; lp_col.n = lp_val.n = (size_t)nnz_new;
    lp_lo.n = lp_hi.n = lp_y.n = d_age.n = d_cutprev.n = (size_t)m_new;
    if (ds_valid.n) ds_valid.zero(stream);          // row indices changed: the dense path's working set is void
    if (md_valid) {                                 // the mid-size solver's working rows move with the compaction (dropped one: cold start)
        md_lost.zero(stream);
        hipLaunchKernelGGL(k_mid_remap, dim3((unsigned)ceil_div(n_lp, 256)), dim3(256), 0, stream, (int)n_lp, md_W.p, d_keep.p, d_newidx.p, md_lost.p);
        int32_t lost = 0;
        KTN_HIP(hipMemcpyAsync(&lost, md_lost.p, 4, hipMemcpyDeviceToHost, stream));
        sync();
        if (lost) md_valid = false;
    }
    smax_rows = 0;
    stats["purged_rows"] += (double)(m - m_new);
    purged_total += m - m_new;
    stats["purges"] += 1.0;
    col_removed_rows += M - m_new;                     // (find_long_cols: a column may have gained as many entries as rows were appended)
    M = m_new; NNZ = nnz_new;
    lp_dirty = true; ++lp_version; ++lp_epoch;
}

void Engine::compute_scaling(bool identity) {
    dr.resize((size_t)std::max<int64_t>(M, 1), stream);
    dc.resize((size_t)n_lp, stream);
    statr.resize((size_t)std::max<int64_t>(M, 1), stream);
    statc.resize((size_t)n_lp, stream);
    // Warm start (lp_ruiz_warm > 0): rows are only ever appended, so the Ruiz (max-norm) equilibration of the previous solve
    // -- kept in dr_r / dc_r as it was BEFORE that solve's Pock-Chambolle pass -- already fits all but the new rows.  Those
    // start at 1 and lp_ruiz_warm passes replace the lp_ruiz_iters passes from scratch; the Pock-Chambolle pass (which
    // carries the ||A^||_2 <= 1 guarantee) is applied afresh.  (Warm-starting from the FINAL scaling instead compounds the
    // Pock-Chambolle passes of all earlier solves: cfg3 then needs 3.5x the PDHG iterations.)
    const bool warm = !identity && prm.lp_ruiz_warm > 0 && scal_rows > 0 && scal_rows <= M && scal_cols == n_lp;
    dr_r.resize((size_t)std::max<int64_t>(M, 1), stream);
    dc_r.resize((size_t)n_lp, stream);
    if (warm) {
        KTN_HIP(hipMemcpyAsync(dr.p, dr_r.p, (size_t)scal_rows * sizeof(double), hipMemcpyDeviceToDevice, stream));
        KTN_HIP(hipMemcpyAsync(dc.p, dc_r.p, (size_t)n_lp * sizeof(double), hipMemcpyDeviceToDevice, stream));
        LAUNCH_1(k_fill, M - scal_rows, stream, M - scal_rows, dr.p + scal_rows, 1.0);
        stats["lp_scaling_warm"] += 1.0;
    } else {
        LAUNCH_1(k_fill, M, stream, M, dr.p, 1.0);
        LAUNCH_1(k_fill, n_lp, stream, n_lp, dc.p, 1.0);
    }
    const int gr = pick_group((double)NNZ / (double)std::max<int64_t>(M, 1));
    const int gc = pick_group((double)NNZ / (double)std::max<int64_t>(n_lp, 1));
    const double cap_c = w_shift ? 1e3 : kInf;          // (kernels.hpp k_scale_apply2)
    if (!identity && (M > 0 || row_sharded())) {
        const int passes = warm ? prm.lp_ruiz_warm : prm.lp_ruiz_iters;
        for (int it = 0; it <= passes; ++it) {
            const int mode = (it == passes) ? 1 : 0;   // last pass: Pock-Chambolle (alpha = 1)
            if (mode == 1 && prm.lp_ruiz_warm > 0) {
                KTN_HIP(hipMemcpyAsync(dr_r.p, dr.p, (size_t)M * sizeof(double), hipMemcpyDeviceToDevice, stream));
                KTN_HIP(hipMemcpyAsync(dc_r.p, dc.p, (size_t)n_lp * sizeof(double), hipMemcpyDeviceToDevice, stream));
            }
            if (n_long == 0 && n_longc == 0 && !row_sharded() && M > 0) {
                // statistic + update in one launch per side, into new arrays that are swapped in (22 launches instead of 33)
                dr2.resize((size_t)M, stream); dc2.resize((size_t)n_lp, stream);
                LAUNCH_G(gr, k_scale_stat_upd, M, stream, M, lp_rowptr.p, lp_col.p, Wval(), dr.p, dc.p, mode, dr2.p, kInf);
                LAUNCH_G(gc, k_scale_stat_upd, n_lp, stream, n_lp, c_ptr.p, c_row.p, c_val.p, dc.p, dr.p, mode, dc2.p, cap_c);
                dr.swap(dr2); dc.swap(dc2);
                continue;
            }
            if (n_long > 0) {
                LAUNCH_G(gr, k_scale_stat_skip, M, stream, M, lp_rowptr.p, lp_col.p, Wval(), dr.p, dc.p, mode, statr.p, kLongRow);
                hipLaunchKernelGGL(k_scale_stat_long, dim3((unsigned)n_long), dim3(1024), 0, stream, d_longrows.p, lp_rowptr.p, lp_col.p,
                                   Wval(), dr.p, dc.p, mode, statr.p);
            } else {
                LAUNCH_G(gr, k_scale_stat, M, stream, M, lp_rowptr.p, lp_col.p, Wval(), dr.p, dc.p, mode, statr.p);
            }
            if (n_longc > 0) {
                LAUNCH_G(gc, k_scale_stat_skip, n_lp, stream, n_lp, c_ptr.p, c_row.p, c_val.p, dc.p, dr.p, mode, statc.p, kLongRow);
                hipLaunchKernelGGL(k_scale_stat_long, dim3((unsigned)n_longc), dim3(1024), 0, stream, d_longcols.p, c_ptr.p, c_row.p,
                                   c_val.p, dc.p, dr.p, mode, statc.p);
            } else {
                LAUNCH_G(gc, k_scale_stat, n_lp, stream, n_lp, c_ptr.p, c_row.p, c_val.p, dc.p, dr.p, mode, statc.p);
            }
            if (row_sharded()) {                       // a column's max / sum runs over the rows of every rank
                if (M == 0) LAUNCH_1(k_fill, n_lp, stream, n_lp, statc.p, 0.0);
                allreduce(statc.p, (size_t)n_lp, mode ? 0 : 1);
            }
            LAUNCH_1(k_scale_apply2, std::max(M, n_lp), stream, M, dr.p, statr.p, n_lp, dc.p, statc.p, cap_c);
        }
    }
    scal_rows = identity ? 0 : M;
    scal_cols = n_lp;
    r_sval.resize((size_t)NNZ + 1, stream);
    c_sval.resize((size_t)NNZ + 1, stream);
    LAUNCH_G(gr, k_scale_vals, M, stream, M, lp_rowptr.p, lp_col.p, Wval(), dr.p, dc.p, r_sval.p);
    // (long columns: the mirror's scaled values are gathered entry-parallel from the row copy instead of walked column by column)
    if (n_longc > 0) LAUNCH_1(k_csc_vals, NNZ, stream, NNZ, c_perm.p, r_sval.p, c_sval.p);
    else LAUNCH_G(gc, k_scale_vals, n_lp, stream, n_lp, c_ptr.p, c_row.p, c_val.p, dc.p, dr.p, c_sval.p);
    check_launch();
}

// Tiled copy of a sparse matrix given by (ptr, idx, val) over n_out outputs and n_in inputs; outputs longer than
// skip_longer are left out (long rows have their own kernel).  Returns false when a (tile, block) segment does not fit
// the 16-bit offsets (then the CSR kernels serve this solve).
bool Engine::build_tiled(TiledBuf& T, int64_t n_out, int64_t n_in, const int64_t* ptr, const int32_t* idx, const double* val,
                         int64_t skip_longer) {
    T.tiles = ceil_div(n_out, kTileOut);
    T.nb_in = ceil_div(n_in, kTileIn);
    const size_t cells = (size_t)T.tiles * (size_t)T.nb_in;
    T.bptr.resize(cells * (kTileOut + 1), stream);
    T.segtot.resize(cells + 1, stream);
    T.segstart.resize(cells + 1, stream);
    T.idx.resize((size_t)NNZ + 1, stream);
    T.val.resize((size_t)NNZ + 1, stream);
    int32_t ovf = 0;
    const bool no_sorted = dev.tiled_general_build;             // (tests: the general kernels)
    // first the run-based kernels (entries of an output in ascending input order: what the LP's rows and the mirror's columns
    // are); an output that is not ascending makes them give up (bit 1) and the general kernels build the copy
    for (int pass = no_sorted ? 1 : 0; pass < 2; ++pass) {
        T.bptr.zero(stream);
        T.segtot.zero(stream);
        KTN_HIP(hipMemsetAsync(d_anynf.p + 1, 0, sizeof(int32_t), stream));
        if (pass == 0) {
            LAUNCH_1(k_tile_count_sorted, n_out, stream, n_out, ptr, idx, T.nb_in, skip_longer, T.bptr.p, d_anynf.p + 1);
        } else {
            T.cur.resize(cells * (kTileOut + 1), stream);
            T.cur.zero(stream);
            LAUNCH_1(k_tile_count, n_out, stream, n_out, ptr, idx, T.nb_in, skip_longer, T.bptr.p, d_anynf.p + 1);
        }
        hipLaunchKernelGGL(k_tile_scan, dim3((unsigned)cells), dim3(kTileThreads), 0, stream, T.bptr.p, T.segtot.p, d_anynf.p + 1);
        check_launch();
        exclusive_scan(T.segtot.p, T.segstart.p, cells + 1);
        if (pass == 0) LAUNCH_1(k_tile_fill_sorted, n_out, stream, n_out, ptr, idx, val, T.nb_in, skip_longer, T.bptr.p, T.segstart.p, d_anynf.p + 1, T.idx.p, T.val.p);
        else LAUNCH_1(k_tile_fill, n_out, stream, n_out, ptr, idx, val, T.nb_in, skip_longer, T.bptr.p, T.cur.p, T.segstart.p, T.idx.p, T.val.p);
        check_launch();
        KTN_HIP(hipMemcpyAsync(&ovf, d_anynf.p + 1, 4, hipMemcpyDeviceToHost, stream));
        sync();
        if (pass == 0 && (ovf & 2)) { stats["lp_tiled_general_builds"] += 1.0; continue; }
        break;
    }
    // two 1024-thread workgroups per CU (80 KB of LDS each): a persistent grid over the (tile, block) units
    const int64_t U = T.tiles * T.nb_in;
    const int wg_per_cu = dev.tiled_wg;
    T.grid = std::max<int64_t>(std::min<int64_t>((int64_t)wg_per_cu * num_cus, U), 1);
    {
        std::vector<int32_t> pc((size_t)T.tiles);
        T.pieces = 1;
        for (int64_t tl = 0; tl < T.tiles; ++tl) {       // owner(u) = ((u + 1) G - 1) / U, as in the kernel
            const int64_t p0 = ((tl * T.nb_in + 1) * T.grid - 1) / U, p1 = (((tl + 1) * T.nb_in) * T.grid - 1) / U;
            pc[(size_t)tl] = (int32_t)(p1 - p0 + 1);
            T.pieces = std::max<int64_t>(T.pieces, p1 - p0 + 1);
        }
        T.pcnt.upload(pc, stream);
        sync();
    }
    return ovf == 0;
}

// Throughput mode: rows -> blocks (by the first column), block row lists in row order, local row positions of the CSC mirror.
void Engine::build_blocks() {
    const int nb = (int)n_blocks;
    d_blkrowptr.resize((size_t)nb + 1, stream);
    d_blkrows.resize((size_t)std::max<int64_t>(M, 1), stream);
    d_rowloc.resize((size_t)std::max<int64_t>(M, 1), stream);
    d_crowl.resize((size_t)NNZ + 1, stream);
    k_in.resize((size_t)std::max<int64_t>(M, 1), stream); k_out.resize((size_t)std::max<int64_t>(M, 1), stream);
    p_in.resize((size_t)std::max<int64_t>(M, 1), stream); p_out.resize((size_t)std::max<int64_t>(M, 1), stream);
    LAUNCH_1(k_row_block, M, stream, M, lp_rowptr.p, lp_col.p, d_blkcol.p, nb, k_in.p, p_in.p);
    int bits = 1;
    while ((1 << bits) < nb + 1 && bits < 30) ++bits;
    const size_t need = sort_pairs_temp_bytes((size_t)M);
    d_sorttmp.resize(need + 16, stream);
    KTN_HIP(sort_pairs_u64_u32(d_sorttmp.p, need, k_in.p, k_out.p, p_in.p, p_out.p, (size_t)M, 0, bits, stream));   // stable: rows ascending
    LAUNCH_1(k_block_rows, M, stream, M, k_out.p, p_out.p, nb, d_blkrowptr.p, d_blkrows.p, d_rowloc.p);
    LAUNCH_1(k_row_local, M, stream, M, k_out.p, p_out.p, d_blkrowptr.p, d_rowloc.p);
    LAUNCH_1(k_localize_rows, NNZ, stream, NNZ, c_row.p, d_rowloc.p, d_crowl.p);
    check_launch();
    std::vector<int32_t> rp = d_blkrowptr.to_host(stream);
    blk_mmax = 1;
    for (int b = 0; b < nb; ++b) blk_mmax = std::max(blk_mmax, rp[(size_t)b + 1] - rp[(size_t)b]);
    blocks_built_rows = M;
}

// One launch: every block's LP to the given tolerances.  Returns false when some block could not finish here (the caller
// then runs the ordinary loop).
bool Engine::lp_solve_blocks(double tol_p, double tol_g, double eta, LpResult* R, int64_t max_it) {
    const int nb = (int)n_blocks;
    const size_t lds = (size_t)(3 * blk_nmax + 3 * blk_mmax + (kBlkThreads / 64) * kBlkQ + kBlkQ + 8) * sizeof(double) +
                       (size_t)(blk_mmax + 2) * sizeof(int32_t);
    if (lds > 150 * 1024) return false;
    if (lds > lds_set_lp) {        // (per handle: handles live on different devices and host threads)
        KTN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_pdhg_blocks), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        lds_set_lp = lds;
    }
    d_blkres.resize((size_t)nb * 8, stream);
    BlkLp P;
    P.blk_col = d_blkcol.p; P.blk_rowptr = d_blkrowptr.p; P.blk_rows = d_blkrows.p;
    P.rptr = lp_rowptr.p; P.rcol = lp_col.p; P.rval = r_sval.p;
    P.cptr = c_ptr.p; P.crowl = d_crowl.p; P.cval = c_sval.p;
    P.c = ch.p; P.l = lh.p; P.u = uh.p; P.lo = loh.p; P.hi = hih.p; P.dr = dr.p; P.dc = dc.p;
    P.x = xh.p; P.y = yh.p; P.xt = xth.p; P.yt = yth.p; P.omega = d_blkomega.p; P.res = d_blkres.p;
    P.tol_p = tol_p; P.tol_g = tol_g; P.eta0 = eta; P.eta_safe = 0.998; P.stag_factor = prm.lp_stag_factor;
    P.stall_accept = (tol_p > prm.lp_tol_floor * prm.f_tol * (1.0 + 1e-9)) ? 10.0 : 3.0;
    P.check_every = std::max(2, prm.lp_check_every); P.first_chunk = 31; P.near_chunk = prm.lp_near_check;
    P.max_iter = (int)std::min<int64_t>(max_it, 2000000000);
    P.nmax = blk_nmax; P.mmax = blk_mmax;
    hipLaunchKernelGGL(k_pdhg_blocks, dim3((unsigned)nb), dim3(kBlkThreads), lds, stream, P);
    check_launch();
    std::vector<double> res = d_blkres.to_host(stream);
    bool all_ok = true;
    double pobj = 0.0, dobj = 0.0, pviol = 0.0, gap = 0.0, it_max = 0.0, it_sum = 0.0;
    for (int b = 0; b < nb; ++b) {
        const double* o = res.data() + (size_t)b * 8;
        if ((int)o[0] != KTN_STATUS_OPTIMAL) all_ok = false;
        pobj += o[2]; dobj += o[3]; pviol = std::max(pviol, o[4]); gap = std::max(gap, o[5]);
        it_max = std::max(it_max, o[1]); it_sum += o[1];
    }
    stats["blk_lp_launches"] += 1.0;
    stats["blk_pdhg_iters_sum"] += it_sum;
    stats["blk_pdhg_iters_max"] += it_max;
    if (!all_ok) { stats["blk_lp_fallbacks"] += 1.0; return false; }
    R->status = KTN_STATUS_OPTIMAL; R->iters = (int64_t)it_max; R->pobj = pobj; R->dobj = dobj; R->row_viol = pviol; R->gap = gap;
    return true;
}

// Throughput mode, device-side loop (batch_ecp.hpp).  Returns false when the problem does not qualify or an instance
// could not finish (arena overflow, LP status): the caller then runs the ordinary loop.
bool Engine::optimize_blocks_device(int cap_mul) {
    if (n_blocks <= 0 || !obj_linear || sense != KTN_MIN || has_inf_bound || n_tape_nl > 0 || n_host > 0 || prm.vis_data) return false;
    for (int64_t i = 0; i < m_ext - 1; ++i) if (h_rowkind[(size_t)i] != KTN_ROW_SEP) return false;
    const int nb = (int)n_blocks;
    auto block_of_col = [&](int64_t c) { return (int)(std::upper_bound(h_blkcol.begin(), h_blkcol.end(), c) - h_blkcol.begin()) - 1; };
    // linear rows of the loaded LP (in original row order) and NL slots must be grouped by instance, instance after instance
    std::vector<int64_t> lin_rows, blk_lin((size_t)nb + 1, 0), blk_nl((size_t)nb + 1, 0), nnz_lin((size_t)nb, 0), nnz_nl((size_t)nb, 0);
    {
        std::vector<char> is_nl((size_t)m0, 0);
        for (auto r : h_nlrows) if (r < m0) is_nl[(size_t)r] = 1;
        for (int64_t i = 0; i < m0; ++i) if (!is_nl[(size_t)i]) lin_rows.push_back(i);
    }
    if ((int64_t)lin_rows.size() != M_base) return false;
    int prev = 0;
    for (size_t k = 0; k < lin_rows.size(); ++k) {
        const int64_t r = lin_rows[k];
        if (h_rowptr[r + 1] == h_rowptr[r]) return false;
        const int bb = block_of_col(h_col[(size_t)h_rowptr[r]]);
        if (bb < prev || bb >= nb) return false;
        for (int64_t e = h_rowptr[r]; e < h_rowptr[r + 1]; ++e) if (block_of_col(h_col[(size_t)e]) != bb) return false;
        prev = bb;
        blk_lin[(size_t)bb + 1] += 1;
        nnz_lin[(size_t)bb] += h_rowptr[r + 1] - h_rowptr[r];
    }
    prev = 0;
    for (size_t k = 0; k < h_nlrows.size(); ++k) {
        const int64_t r = h_nlrows[k];
        if (h_rowptr[r + 1] == h_rowptr[r]) return false;
        const int bb = block_of_col(h_col[(size_t)h_rowptr[r]]);
        if (bb < prev || bb >= nb) return false;
        for (int64_t e = h_rowptr[r]; e < h_rowptr[r + 1]; ++e) if (block_of_col(h_col[(size_t)e]) != bb) return false;
        prev = bb;
        blk_nl[(size_t)bb + 1] += 1;
        nnz_nl[(size_t)bb] += h_rowptr[r + 1] - h_rowptr[r];
    }
    for (int bb = 0; bb < nb; ++bb) { blk_lin[(size_t)bb + 1] += blk_lin[(size_t)bb]; blk_nl[(size_t)bb + 1] += blk_nl[(size_t)bb]; }
    // arenas
    std::vector<EcpArena> ar((size_t)nb);
    int64_t row_tot = 0, nnz_tot = 0;
    int mmax = 1;
    for (int bb = 0; bb < nb; ++bb) {
        const int64_t ml = blk_lin[(size_t)bb + 1] - blk_lin[(size_t)bb], mn = blk_nl[(size_t)bb + 1] - blk_nl[(size_t)bb];
        EcpArena a;
        a.row0 = row_tot; a.nnz0 = nnz_tot;
        a.cap_rows = (int32_t)(ml + (int64_t)cap_mul * mn);
        a.cap_nnz = (int32_t)(nnz_lin[(size_t)bb] + (int64_t)cap_mul * nnz_nl[(size_t)bb]);
        row_tot += a.cap_rows; nnz_tot += a.cap_nnz;
        mmax = std::max<int>(mmax, std::max<int>(a.cap_rows, (int)mn));
        ar[(size_t)bb] = a;
    }
    const size_t lds = (size_t)(3 * blk_nmax + 3 * mmax + (kEcpThreads / 64) * kEcpQ + kEcpQ + 8 + 16) * sizeof(double) +
                       (size_t)(std::max(blk_nmax, mmax) + 4) * sizeof(int32_t);
    if (lds > 158 * 1024 || blk_nmax > 65535 || mmax > 65535) return false;      // 16-bit local indices in the arenas
    t_start = std::chrono::steady_clock::now();
    d_ar.upload(ar, stream); d_blklin.upload(blk_lin, stream); d_blknl.upload(blk_nl, stream);
    e_rptr.resize((size_t)row_tot + nb + 1, stream);
    e_rcol.resize((size_t)nnz_tot + 1, stream); e_rval.resize((size_t)nnz_tot + 1, stream); e_rsval.resize((size_t)nnz_tot + 1, stream);
    e_crow.resize((size_t)nnz_tot + 1, stream); e_cval.resize((size_t)nnz_tot + 1, stream); e_csval.resize((size_t)nnz_tot + 1, stream);
    for (DBuf<double>* v : {&e_lo, &e_hi, &e_y, &e_dr, &e_loh, &e_hih}) v->resize((size_t)row_tot + 1, stream);
    e_cptr.resize((size_t)n_lp + nb + 1, stream);
    for (DBuf<double>* v : {&e_dc, &e_ch, &e_lh, &e_uh}) v->resize((size_t)n_lp + 1, stream);
    e_last.resize((size_t)std::max<int64_t>(m_nl, 1), stream);
    e_prev.resize((size_t)row_tot + 1, stream); e_ax.resize((size_t)row_tot + 1, stream);
    e_res.resize((size_t)nb * 8, stream);
    EcpBatch B;
    B.blk_col = d_blkcol.p; B.blk_lin = d_blklin.p; B.blk_nl = d_blknl.p;
    B.lp_rowptr = lp_rowptr.p; B.lp_col = lp_col.p; B.lp_val = lp_val.p; B.lp_lo = lp_lo.p; B.lp_hi = lp_hi.p;
    B.c = lp_c.p; B.l = lp_l.p; B.u = lp_u.p;
    B.P = nlp_view(); B.nl_rows = d_nlrows.p;
    B.arena = d_ar.p;
    B.rptr = e_rptr.p; B.rcol = e_rcol.p; B.rval = e_rval.p; B.rsval = e_rsval.p; B.lo = e_lo.p; B.hi = e_hi.p; B.y = e_y.p; B.dr = e_dr.p;
    B.loh = e_loh.p; B.hih = e_hih.p;
    B.cptr = e_cptr.p; B.crow = e_crow.p; B.cval = e_cval.p; B.csval = e_csval.p;
    B.dc = e_dc.p; B.ch = e_ch.p; B.lh = e_lh.p; B.uh = e_uh.p;
    e_xbest.resize((size_t)n_lp + 1, stream);
    B.last_cut = e_last.p; B.cut_prev = e_prev.p; B.ax = e_ax.p; B.x = lp_x.p; B.xbest = e_xbest.p; B.res = e_res.p;
    B.cert_tol = prm.obj_cert_tol; B.polish_max_iter = prm.polish_max_iter;
    B.f_tol = prm.f_tol; B.cut_coef_rng = prm.cut_coef_rng; B.tol_scale = prm.lp_tol_scale; B.tol_floor = prm.lp_tol_floor;
    B.tol_cap = prm.lp_tol_cap; B.gap_floor = prm.lp_gap_floor; B.gap_cap = prm.lp_gap_cap; B.stag_factor = prm.lp_stag_factor;
    B.iter_cap = prm.iter_cap; B.lp_max_iter = prm.lp_max_iter; B.check_every = std::max(2, prm.lp_check_every);
    // (a lone workgroup's check is cheap and every instance stops on its own: a period of 24 halves the iterations of the
    //  slowest instance against 64 -- 512 x cfg5: max 15 842 -> 5 008, the launch 0.131 -> 0.097 s)
    B.check_every = std::min(B.check_every, 24);
    B.near_chunk = prm.lp_near_check; B.ruiz_iters = prm.lp_ruiz_iters; B.nmax = blk_nmax; B.mmax = mmax;
    B.power_passes = dev.ecp_power;
    if (lds > lds_set_ecp) {
        KTN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ecp_blocks), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        lds_set_ecp = lds;
    }
    // the loaded state: M == M_base rows (the linear rows), as after reset()
    hipLaunchKernelGGL(k_ecp_blocks, dim3((unsigned)nb), dim3(kEcpThreads), lds, stream, B);
    check_launch();
    std::vector<double> res = e_res.to_host(stream);
    bool ok = true;
    double obj = 0.0, it_max = 0.0, cuts = 0.0, pd = 0.0, rows = 0.0;
    for (int bb = 0; bb < nb; ++bb) {
        const double* o = res.data() + (size_t)bb * 8;
        if ((int)o[0] != KTN_STATUS_OPTIMAL) ok = false;
        obj += o[2]; it_max = std::max(it_max, o[1]); cuts += o[3]; pd += o[4]; rows += o[6];
    }
    if (dev.debug_blocks) {
        std::vector<std::pair<double, int>> v;
        for (int bb = 0; bb < nb; ++bb) v.push_back({res[(size_t)bb * 8 + 4], bb});
        std::sort(v.begin(), v.end());
        std::fprintf(stderr, "[ecp blocks] pdhg per instance: min %.0f median %.0f p90 %.0f p99 %.0f max %.0f (instance %d, %g ecp iterations, %g rows)\n",
                     v.front().first, v[v.size() / 2].first, v[v.size() * 9 / 10].first, v[v.size() * 99 / 100].first, v.back().first, v.back().second,
                     res[(size_t)v.back().second * 8 + 1], res[(size_t)v.back().second * 8 + 6]);
    }
    stats["ecp_blocks_launches"] += 1.0;
    stats["ecp_blocks_pdhg_sum"] += pd;
    stats["ecp_blocks_rows"] = rows;
    if (!ok) { stats["ecp_blocks_fallbacks"] += 1.0; return false; }
    status = KTN_STATUS_OPTIMAL; lp_status = KTN_STATUS_OPTIMAL;
    iter = (int64_t)it_max; numcuts = (int64_t)cuts; objval = obj + c0; allsat = true;
    soltime = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    return true;
}

void Engine::find_long_rows() {
    n_long = 0;
    if (M == 0 || max_row_len <= kLongRow) return;      // no row of this problem can be long: no scan, no round trip
    d_longrows.resize((size_t)M, stream);
    KTN_HIP(hipMemsetAsync(d_anynf.p + 1, 0, sizeof(int32_t), stream));
    LAUNCH_1(k_find_long, M, stream, M, lp_rowptr.p, kLongRow, d_longrows.p, d_anynf.p + 1);
    int32_t cnt = 0;
    KTN_HIP(hipMemcpyAsync(&cnt, d_anynf.p + 1, 4, hipMemcpyDeviceToHost, stream));
    sync();
    n_long = cnt;
    if (n_long > 1) {
        // k_find_long appends with an atomic counter: the ORDER of the list depends on scheduling, and the check kernels
        // accumulate the long rows in list order.  Sort it (a handful of entries) so that every sum -- and with it every
        // restart decision, on every rank of a sharded run -- is reproducible.
        std::vector<int32_t> tmp((size_t)n_long);
        KTN_HIP(hipMemcpyAsync(tmp.data(), d_longrows.p, (size_t)n_long * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
        sync();
        std::sort(tmp.begin(), tmp.end());
        KTN_HIP(hipMemcpyAsync(d_longrows.p, tmp.data(), (size_t)n_long * sizeof(int32_t), hipMemcpyHostToDevice, stream));
        sync();
    }
}

// y-step over all rows: G lanes per row for ordinary rows, a workgroup per row for the long ones.  Step sizes and the
// Halpern weight travel as kernel arguments (eager launches: at ~6 us per kernel the host stays ahead of the GPU, and a
// hipGraph of the chunk bought nothing measurable while its capture + instantiation cost every LP solve).
void Engine::launch_y(const SpMat& A, double sigma, double w, double rho, hipEvent_t e0, hipEvent_t e1) {
    const int64_t m = M;
    if (m == 0) return;                                 // (first LP of a model with NL rows only: no dual step, no zero-size grid)
    const int64_t thr = n_long > 0 ? kLongRow : (int64_t)1 << 62;
    if (tiled_on && m > 0) {
        launch_tiled(tA, m, n_lp, xbar.p, e0);
        hipExtLaunchKernelGGL(k_y_epilogue, dim3(ceil_div(m, kBlock)), dim3(kBlock), 0, stream, nullptr, e1, 0, m, tA.pcnt.p,
                              tpart.p, (n_long > 0 ? A.ptr : (const int64_t*)nullptr), thr, yh.p, y0h.p,
                              loh.p, hih.p, sigma, w, rho);
    } else if (packed_on) {
        const int thr32 = n_long > 0 ? (int)kLongRow : 0x7fffffff;
        const int32_t* none = nullptr;
        if (packed_trips == 2) LAUNCH_GT(grp_rows, 2, k_pdhg_y_packed, m, stream, e0, e1, m, A.idx, A.val, xbar.p, yh.p, d_rrec.p, sigma, w, rho, thr32, none, 0);
        else if (packed_trips == 4) LAUNCH_GT(grp_rows, 4, k_pdhg_y_packed, m, stream, e0, e1, m, A.idx, A.val, xbar.p, yh.p, d_rrec.p, sigma, w, rho, thr32, none, 0);
        else {
            // one launch for all rows: the regular lane groups plus one trailing workgroup per long row
            const unsigned grid = (unsigned)(ceil_div(m * grp_rows, (int64_t)kBlock) + n_long);
#define KTN_Y_PACKED(G) hipExtLaunchKernelGGL((k_pdhg_y_packed<G, 1>), dim3(grid), dim3(kBlock), 0, stream, e0, e1, 0, m, A.idx, A.val, xbar.p, yh.p, \
                                              d_rrec.p, sigma, w, rho, thr32, (const int32_t*)d_longrows.p, (int)n_long)
            switch (grp_rows) {
                case 4: KTN_Y_PACKED(4); break;
                case 8: KTN_Y_PACKED(8); break;
                case 16: KTN_Y_PACKED(16); break;
                case 32: KTN_Y_PACKED(32); break;
                default: KTN_Y_PACKED(64); break;
            }
#undef KTN_Y_PACKED
            return;
        }
    } else if (e0) {
        LAUNCH_G_EV(grp_rows, k_pdhg_y, m, stream, e0, e1, m, A, xbar.p, yh.p, y0h.p, loh.p, hih.p, sigma, w, rho, thr);
    } else {
        LAUNCH_G(grp_rows, k_pdhg_y, m, stream, m, A, xbar.p, yh.p, y0h.p, loh.p, hih.p, sigma, w, rho, thr);
    }
    if (n_long > 0)
        hipLaunchKernelGGL((k_pdhg_y_long<false>), dim3((unsigned)n_long), dim3(kLongBlock), 0, stream, d_longrows.p, A, xbar.p,
                           (const double*)nullptr, yh.p, y0h.p, yth.p, loh.p, hih.p, dr.p, sigma, w, rho, (double*)nullptr);
}
void Engine::launch_x(const SpMat& AT, double tau, double w, double rho, bool update, hipEvent_t e0, hipEvent_t e1) {
    const int64_t n = n_lp;
    if (row_sharded()) {
        // local partial of A'y, summed over the ranks, then the element-wise primal step on the replicated x
        // (peer-buffer transport: the partial is written straight into the exposed slot, and the primal step adds up the
        //  ranks' slots itself -- spmv, one single-workgroup barrier kernel, prox: no reduction pass, no copy)
        const bool ipc = dist.ipc.on && n <= dist.ipc.cap;
        double* part = ipc ? ipc_slot() : pv.p;
        if (M == 0) LAUNCH_1(k_fill, n, stream, n, part, 0.0);
        if (tiled_on && M > 0) {                          // this rank's block is large: its partial A_r'y_r from the tiled copy
            launch_tiled(tAT, n, M, yh.p, e0);
            hipExtLaunchKernelGGL(k_tile_vec, dim3(ceil_div(n, kBlock)), dim3(kBlock), 0, stream, nullptr, e1, 0, n, tAT.pcnt.p, tpart.p, part);
        }
        else spmv_cols(AT, yh.p, part, e0, e1);
        if (ipc) {
            stats["allreduce_calls"] += 1.0;
            stats["allreduce_bytes"] += 8.0 * (double)n;
            const int64_t off = ipc_barrier();
            if (update) LAUNCH_1(k_x_prox_ipc<true>, n, stream, n, dist.ipc.P, dist.world, off, xh.p, x0h.p, xth.p, xbar.p, ch.p, lh.p, uh.p, tau, w, rho);
            else LAUNCH_1(k_x_prox_ipc<false>, n, stream, n, dist.ipc.P, dist.world, off, xh.p, x0h.p, xth.p, xbar.p, ch.p, lh.p, uh.p, tau, w, rho);
            return;
        }
        allreduce(pv.p, (size_t)n, 0);
        if (update) LAUNCH_1(k_x_prox<true>, n, stream, n, pv.p, xh.p, x0h.p, xth.p, xbar.p, ch.p, lh.p, uh.p, tau, w, rho);
        else LAUNCH_1(k_x_prox<false>, n, stream, n, pv.p, xh.p, x0h.p, xth.p, xbar.p, ch.p, lh.p, uh.p, tau, w, rho);
        return;
    }
    if (n_longc > 0 && !(tiled_on && M > 0)) {
        // long columns: lane groups for the ordinary columns, a workgroup per long one -- the primal step fused into both
        if (update) {
            LAUNCH_GB(grp_cols, k_pdhg_x_skip, true, n, stream, n, AT, yh.p, xh.p, x0h.p, xth.p, xbar.p, ch.p, lh.p, uh.p, tau, w, rho, kLongRow);
            hipLaunchKernelGGL((k_pdhg_x_long<true>), dim3((unsigned)n_longc), dim3(1024), 0, stream, d_longcols.p, AT, yh.p, xh.p, x0h.p, xth.p,
                               xbar.p, ch.p, lh.p, uh.p, tau, w, rho);
        } else {
            LAUNCH_GB(grp_cols, k_pdhg_x_skip, false, n, stream, n, AT, yh.p, xh.p, x0h.p, xth.p, xbar.p, ch.p, lh.p, uh.p, tau, w, rho, kLongRow);
            hipLaunchKernelGGL((k_pdhg_x_long<false>), dim3((unsigned)n_longc), dim3(1024), 0, stream, d_longcols.p, AT, yh.p, xh.p, x0h.p, xth.p,
                               xbar.p, ch.p, lh.p, uh.p, tau, w, rho);
        }
        return;
    }
    if (update) {
        if (tiled_on && M > 0) {
            launch_tiled(tAT, n, M, yh.p, e0);
            hipExtLaunchKernelGGL(k_x_epilogue, dim3(ceil_div(n, kBlock)), dim3(kBlock), 0, stream, nullptr, e1, 0, n, tAT.pcnt.p,
                                  tpart.p, xh.p, x0h.p, xbar.p, ch.p, lh.p, uh.p, tau, w, rho);
        } else if (packed_on) {
            if (packed_trips == 2) LAUNCH_GT(grp_cols, 2, k_pdhg_x_packed, n, stream, e0, e1, n, d_cbl.p, AT.idx, AT.val, yh.p, xh.p, xbar.p, d_crec.p, tau, w, rho);
            else if (packed_trips == 4) LAUNCH_GT(grp_cols, 4, k_pdhg_x_packed, n, stream, e0, e1, n, d_cbl.p, AT.idx, AT.val, yh.p, xh.p, xbar.p, d_crec.p, tau, w, rho);
            else LAUNCH_GT(grp_cols, 1, k_pdhg_x_packed, n, stream, e0, e1, n, d_cbl.p, AT.idx, AT.val, yh.p, xh.p, xbar.p, d_crec.p, tau, w, rho);
        } else if (e0) {
            LAUNCH_GB_EV(grp_cols, k_pdhg_x, true, n, stream, e0, e1, n, AT, yh.p, xh.p, x0h.p, xth.p, xbar.p, ch.p, lh.p, uh.p, tau, w, rho);
        } else {
            LAUNCH_GB(grp_cols, k_pdhg_x, true, n, stream, n, AT, yh.p, xh.p, x0h.p, xth.p, xbar.p, ch.p, lh.p, uh.p, tau, w, rho);
        }
    } else {
        LAUNCH_GB(grp_cols, k_pdhg_x, false, n, stream, n, AT, yh.p, xh.p, x0h.p, xth.p, xbar.p, ch.p, lh.p, uh.p, tau, w, rho);
    }
}
// Check iteration: the PDHG step without update (xt, yt stored) and the KKT / fixed-point sums.  The row side rides on the
// y-step (k_pdhg_y_chk gathers xt and x anyway); the column side needs A'yt and is one G-lanes-per-column pass.
void Engine::launch_check(const SpMat& A, const SpMat& AT, double tau, double sigma) {
    const int64_t n = n_lp, m = M;
    const int64_t thr = n_long > 0 ? kLongRow : (int64_t)1 << 62;
    const int64_t brow = ceil_div(std::max<int64_t>(m, 1) * grp_rows, kBlock), bcol = ceil_div(n * grp_cols, kBlock);
    chk_part.resize((size_t)(brow + n_long + bcol) * kChkQ, stream);
    double* prow = chk_part.p;
    double* pcol = chk_part.p + (size_t)(brow + n_long) * kChkQ;
    const bool tiled_chk_off = dev.no_tiled_check;
    if (tiled_on && m > 0 && !tiled_chk_off) {
        // the four SpMV passes of a check through the tiled copy (kernels.hpp "check iteration on the tiled copy"); row-sharded:
        // the two column-side vectors are this rank's partials and are summed over the ranks -- the same sequence of
        // collectives as the CSR form below, so ranks may differ in which form they run
        const int64_t brow_t = ceil_div(m, (int64_t)kBlock), bcol_t = ceil_div(n, (int64_t)kBlock);      // <= brow, bcol
        launch_tiled(tAT, n, m, yh.p, nullptr);
        if (row_sharded()) {
            LAUNCH_1(k_tile_vec, n, stream, n, tAT.pcnt.p, tpart.p, pv.p);
            allreduce(pv.p, (size_t)n, 0);
            LAUNCH_1(k_x_prox<false>, n, stream, n, pv.p, xh.p, x0h.p, xth.p, xbar.p, ch.p, lh.p, uh.p, tau, 0.0, 1.0);
        } else {
            LAUNCH_1(k_x_epilogue_chk, n, stream, n, tAT.pcnt.p, tpart.p, xh.p, xth.p, ch.p, lh.p, uh.p, tau);
        }
        launch_tiled(tA, m, n, xth.p, nullptr);
        LAUNCH_1(k_tile_vec, m, stream, m, tA.pcnt.p, tpart.p, pw.p);
        launch_tiled(tA, m, n, xh.p, nullptr);
        LAUNCH_1(k_y_epilogue_chk, m, stream, m, tA.pcnt.p, tpart.p, pw.p, (n_long > 0 ? A.ptr : (const int64_t*)nullptr), thr, yh.p, y0h.p,
                 yth.p, loh.p, hih.p, dr.p, sigma, prow);
        if (n_long > 0)
            hipLaunchKernelGGL((k_pdhg_y_long<true>), dim3((unsigned)n_long), dim3(kLongBlock), 0, stream, d_longrows.p, A, xth.p, xh.p,
                               yh.p, y0h.p, yth.p, loh.p, hih.p, dr.p, sigma, 0.0, 1.0, prow + (size_t)brow_t * kChkQ);
        chk_nrow = (int)(brow_t + n_long);
        launch_tiled(tAT, n, m, yth.p, nullptr);
        LAUNCH_1(k_tile_vec, n, stream, n, tAT.pcnt.p, tpart.p, pv.p);
        if (row_sharded()) allreduce(pv.p, (size_t)n, 0);
        LAUNCH_1(k_chk_cols_vec, n, stream, n, pv.p, xh.p, xth.p, x0h.p, ch.p, lh.p, uh.p, dc.p, pcol);
        chk_ncol = (int)bcol_t;
        hipLaunchKernelGGL(k_chk_final, dim3(2 * kChkQ), dim3(kRedBlocks), 0, stream, prow, chk_nrow, pcol, chk_ncol, chk_target());
        if (row_sharded()) {                            // row sums: every rank's rows; column sums are identical already
            allreduce(chkout.p, 12, 0);
            allreduce(chkout.p + 12, 4, 1);
        }
        return;
    }
    launch_x(AT, tau, 0.0, 1.0, false, nullptr, nullptr);
    if (m > 0) {
        LAUNCH_G(grp_rows, k_pdhg_y_chk, m, stream, m, A, xth.p, xh.p, yh.p, y0h.p, yth.p, loh.p, hih.p, dr.p, sigma, thr, prow);
        if (n_long > 0)
            hipLaunchKernelGGL((k_pdhg_y_long<true>), dim3((unsigned)n_long), dim3(kLongBlock), 0, stream, d_longrows.p, A, xth.p, xh.p,
                               yh.p, y0h.p, yth.p, loh.p, hih.p, dr.p, sigma, 0.0, 1.0, prow + (size_t)brow * kChkQ);
    }
    chk_nrow = (m > 0) ? (int)(brow + n_long) : 0;
    if (row_sharded() || n_longc > 0) {
        if (m == 0) LAUNCH_1(k_fill, n, stream, n, pv.p, 0.0);
        spmv_cols(AT, yth.p, pv.p);
        allreduce(pv.p, (size_t)n, 0);
        chk_ncol = ceil_div(n, kBlock);             // <= bcol: the column partials fit the same region
        LAUNCH_1(k_chk_cols_vec, n, stream, n, pv.p, xh.p, xth.p, x0h.p, ch.p, lh.p, uh.p, dc.p, pcol);
    } else {
        LAUNCH_G(grp_cols, k_chk_cols, n, stream, n, AT, xh.p, xth.p, x0h.p, yth.p, ch.p, lh.p, uh.p, dc.p, pcol);
        chk_ncol = (int)bcol;
    }
    hipLaunchKernelGGL(k_chk_final, dim3(2 * kChkQ), dim3(kRedBlocks), 0, stream, prow, chk_nrow, pcol, chk_ncol, chk_target());   // (rows | columns) x quantity
    if (row_sharded()) {                            // row sums: every rank's rows; column sums are identical already
        allreduce(chkout.p, 12, 0);
        allreduce(chkout.p + 12, 4, 1);
    }
}

// LP dispatch.  The first-order method is the default: on the large sparse LPs of the hot path it is the only
// option, and on small ones its solutions sit in the middle of the optimal face, which Kelley's method likes
// (test/misc.jl 501: tens of iterations instead of thousands from simplex vertices).  Where it STALLS -- several
// nearly parallel cuts active at a curved optimum, DESIGN.md section 5 -- and the LP has at most kDenseMaxN
// columns, the exact kernel finishes the solve; each stall doubles the number of following solves that go to
// the exact kernel directly.
LpResult Engine::lp_solve(double tol_p, double tol_g, int mode, bool identity_scaling) {
    const bool dense_ok = mode == 0 && !identity_scaling && !row_sharded() && prm.lp_dense_after != 0 && n_lp >= 1 && n_lp <= kDenseMaxN &&
                          M * n_lp <= 8000000;
    // ... and LPs of 33 .. lp_mid_max_var columns by the exact mid-size solver (mid_lp.hpp), under the same hand-over rule
    const bool mid_ok = mode == 0 && !identity_scaling && !row_sharded() && prm.lp_dense_after != 0 && n_lp > kDenseMaxN &&
                        n_lp <= std::min<int64_t>(prm.lp_mid_max_var, kMidMaxN) && n_blocks == 0 && M < ((int64_t)1 << 29);
    if (mid_ok && mid_backoff > 0) {
        --mid_backoff;                                   // (a recent exact solve failed: cold starts cost ~n pivots each, do not repeat them at once)
    } else if (mid_ok) {
        auto failed = [&]() {
            stats["mid_lp_fallbacks"] += 1.0;
            mid_credit = 0;
            mid_backoff_len = std::min<int64_t>(2 * std::max<int64_t>(mid_backoff_len, 4), 256);
            mid_backoff = mid_backoff_len;
        };
        if (prm.lp_dense_after < 0 || mid_credit > 0) {
            if (mid_credit > 0) --mid_credit;
            LpResult R;
            if (lp_solve_mid(&R)) return R;
            failed();
            return lp_solve_core(tol_p, tol_g, mode, identity_scaling);
        }
        lp_iter_budget = prm.lp_dense_after;
        LpResult R = lp_solve_core(tol_p, tol_g, mode, identity_scaling);
        lp_iter_budget = 0;
        if (R.status != KTN_STATUS_USERLIMIT) return R;
        stats["lp_stalls"] += 1.0;
        LpResult D;
        if (lp_solve_mid(&D)) {
            mid_run = std::min<int64_t>(2 * std::max<int64_t>(mid_run, 1), 1 << 20);
            mid_credit = mid_run;
            mid_backoff_len = 0;
            return D;
        }
        failed();
        return lp_solve_core(tol_p, tol_g, mode, identity_scaling);
    }
    if (!dense_ok) return lp_solve_core(tol_p, tol_g, mode, identity_scaling);
    if (prm.lp_dense_after < 0 || dense_credit > 0) {
        if (dense_credit > 0) --dense_credit;
        LpResult R;
        if (lp_solve_dense(&R)) return R;
        stats["dense_lp_fallbacks"] += 1.0;
        return lp_solve_core(tol_p, tol_g, mode, identity_scaling);
    }
    lp_iter_budget = prm.lp_dense_after;
    LpResult R = lp_solve_core(tol_p, tol_g, mode, identity_scaling);
    lp_iter_budget = 0;
    if (R.status != KTN_STATUS_USERLIMIT) return R;
    stats["lp_stalls"] += 1.0;
    LpResult D;
    if (lp_solve_dense(&D)) {
        dense_run = std::min<int64_t>(2 * std::max<int64_t>(dense_run, 1), 1 << 20);
        dense_credit = dense_run;
        return D;
    }
    stats["dense_lp_fallbacks"] += 1.0;
    return lp_solve_core(tol_p, tol_g, mode, identity_scaling);
}

// Exact solve of a small LP by the dual active-set kernel (dense_lp.hpp).  Returns false when the kernel
// gives up (singular working set, pivot limit, or an artificial bound left in the optimal working set, i.e.
// the optimal face is unbounded in some zero-cost direction): the caller then runs the first-order method.
bool Engine::lp_solve_dense(LpResult* R) {
    auto t0 = std::chrono::steady_clock::now();
    const int n = (int)n_lp;
    const int64_t m = M;
    ds_dense.resize((size_t)std::max<int64_t>(m, 1) * n, stream);
    ds_out.resize(4, stream);
    if (ds_W.n != (size_t)n) {
        ds_W.resize(n, stream);
        ds_valid.resize(1, stream);
        ds_valid.zero(stream);
    }
    lp_y.resize((size_t)std::max<int64_t>(m, 1), stream);
    lp_y.n = (size_t)m;
    DenseLpIO P;
    P.n = n; P.m = m; P.rowptr = lp_rowptr.p; P.col = lp_col.p; P.val = lp_val.p; P.lo = lp_lo.p; P.hi = lp_hi.p;
    P.l = lp_l.p; P.u = lp_u.p; P.c = lp_c.p; P.sgn = (sense == KTN_MAX) ? -1.0 : 1.0;
    P.dense = ds_dense.p; P.W = ds_W.p; P.Wvalid = ds_valid.p; P.x = lp_x.p; P.y = lp_y.p; P.out = ds_out.p;
    P.max_pivots = 200 + 20 * n + (int)std::min<int64_t>(m, 100000);
    P.tol = 1e-9;
    hipLaunchKernelGGL(k_dense_lp, dim3(1), dim3(256), 0, stream, P);
    check_launch();
    double out[4];
    KTN_HIP(hipMemcpyAsync(out, ds_out.p, sizeof(out), hipMemcpyDeviceToHost, stream));
    sync();
    stats["dense_lp_solves"] += 1.0;
    stats["dense_lp_pivots"] += out[1];
    stats["lp_solves"] += 1.0;
    stats["lp_time_s"] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const int st = (int)out[0];
    if (st == 0) {
        R->status = KTN_STATUS_OPTIMAL;
        R->iters = (int64_t)out[1];
        R->pobj = R->dobj = out[2];
        R->row_viol = 0.0; R->gap = 0.0;
        R->exact = true;
        objval = P.sgn * out[2] + c0;
        return true;
    }
    if (st == 1) {
        R->status = KTN_STATUS_INFEASIBLE;
        R->iters = (int64_t)out[1];
        return true;
    }
    return false;
}

// Exact solve of a mid-size LP (mid_lp.hpp): batches of pivots enqueued without a host synchronisation, the device-resident
// state read back once per batch.  Returns false when the solver gives up (an artificial side of a free variable is still
// needed, pivot limit, a basis inverse that a cold restart does not repair): the caller then runs the first-order method.
bool Engine::lp_solve_mid(LpResult* R) {
    auto t0 = std::chrono::steady_clock::now();
    const int n = (int)n_lp;
    const int64_t m = M;
    if (md_W.n != (size_t)n) {
        md_Binv.resize((size_t)n * n, stream);
        md_W.resize(n, stream); md_hW.resize(n, stream); md_x.resize(n, stream); md_lam.resize(n, stream);
        md_u.resize(n, stream); md_d.resize(n, stream); md_r.resize(n, stream); md_c.resize(n, stream);
        md_pv.resize(kMidPriceBlocks, stream); md_pi.resize(kMidPriceBlocks, stream); md_st.resize(1, stream); md_lost.resize(1, stream);
        md_valid = false;
    }
    lp_y.resize((size_t)std::max<int64_t>(m, 1), stream);
    lp_y.n = (size_t)m;
    MidLpIO P;
    P.n = n; P.m = m; P.rowptr = lp_rowptr.p; P.col = lp_col.p; P.val = lp_val.p; P.lo = lp_lo.p; P.hi = lp_hi.p;
    P.l = lp_l.p; P.u = lp_u.p; P.c = lp_c.p; P.sgn = (sense == KTN_MAX) ? -1.0 : 1.0;
    P.Binv = md_Binv.p; P.W = md_W.p; P.hW = md_hW.p; P.x = md_x.p; P.lam = md_lam.p; P.uvec = md_u.p; P.dvec = md_d.p; P.rvec = md_r.p;
    P.part_val = md_pv.p; P.part_idx = md_pi.p; P.st = md_st.p; P.ctil = md_c.p;
    P.tol = 1e-9;
    // (a cold start from the bound vertex of a cutting-plane LP that the first-order method has already grown to a few thousand
    //  rows takes tens of pivots per column -- every variable leaves its box corner, many of them more than once; the warm
    //  re-solves that follow take tens to hundreds in total)
    P.max_pivots = 5000 + 200 * n + (int)std::min<int64_t>(20 * m, 4000000);
    const unsigned g_nn = (unsigned)ceil_div((int64_t)n * n, 256), g_n = (unsigned)ceil_div(n, 256);
    auto refine = [&]() {                               // x = B^-1 h_W + one step of iterative refinement; lambda = -B^-T c
        hipLaunchKernelGGL(k_mid_resid, dim3(g_n), dim3(256), 0, stream, P, 0);
        hipLaunchKernelGGL(k_mid_apply, dim3((unsigned)ceil_div(n, 4)), dim3(256), 0, stream, P, 0);
        hipLaunchKernelGGL(k_mid_resid, dim3(g_n), dim3(256), 0, stream, P, 1);
        hipLaunchKernelGGL(k_mid_apply, dim3((unsigned)ceil_div(n, 4)), dim3(256), 0, stream, P, 1);
        hipLaunchKernelGGL(k_mid_resid, dim3(g_n), dim3(256), 0, stream, P, 1);
        hipLaunchKernelGGL(k_mid_resid_norm, dim3(1), dim3(256), 0, stream, P, 1);
        hipLaunchKernelGGL(k_mid_lambda, dim3(g_n), dim3(256), 0, stream, P);
    };
    // B^-1 afresh from the working set (mid_lp.hpp "refactorisation"), then x and lambda from their definitions
    auto refactor = [&]() {
        md_aug.resize((size_t)n * 2 * n, stream); md_prow.resize((size_t)2 * n, stream); md_fcol.resize((size_t)n, stream);
        hipLaunchKernelGGL(k_mid_gj_build, dim3((unsigned)n), dim3(256), 0, stream, P, md_aug.p);
        const unsigned g_2n = (unsigned)ceil_div(2 * n, 256), g_aug = (unsigned)ceil_div((int64_t)n * 2 * n, 256);
        for (int col = 0; col < n; ++col) {
            hipLaunchKernelGGL(k_mid_gj_pivot, dim3(1), dim3(256), 0, stream, P, (const double*)md_aug.p, col);
            hipLaunchKernelGGL(k_mid_gj_swap, dim3(g_2n), dim3(256), 0, stream, P, md_aug.p, col, md_prow.p);
            hipLaunchKernelGGL(k_mid_gj_col, dim3(g_n), dim3(256), 0, stream, P, (const double*)md_aug.p, col, md_fcol.p);
            hipLaunchKernelGGL(k_mid_gj_elim, dim3(g_aug), dim3(256), 0, stream, P, md_aug.p, col, (const double*)md_prow.p, (const double*)md_fcol.p);
        }
        hipLaunchKernelGGL(k_mid_gj_store, dim3(g_nn), dim3(256), 0, stream, P, (const double*)md_aug.p);
        refine();
        md_since_refactor = 0;
        stats["mid_lp_refactors"] += 1.0;
    };
    MidState hs;
    int total_pivots = 0, status = 4;
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (!md_valid) {
            hipLaunchKernelGGL(k_mid_init, dim3(g_nn), dim3(256), 0, stream, P);
            stats["mid_lp_cold_starts"] += 1.0;
            md_since_refactor = 0;
        } else {
            hipLaunchKernelGGL(k_mid_rearm, dim3(1), dim3(1), 0, stream, md_st.p);
            if (md_since_refactor >= kMidRefactor) refactor();
        }
        int refined_at = -1, refinements = 0, pivots_seen = 0;
        bool bad_inverse = false;
        status = 4;
        for (;;) {
            for (int b = 0; b < 8; ++b) {
                hipLaunchKernelGGL(k_mid_price, dim3(kMidPriceBlocks), dim3(256), 0, stream, P);
                hipLaunchKernelGGL(k_mid_select, dim3(1), dim3(256), 0, stream, P);
                hipLaunchKernelGGL(k_mid_u, dim3(g_n), dim3(256), 0, stream, P);
                hipLaunchKernelGGL(k_mid_ratio, dim3(1), dim3(256), 0, stream, P);
                hipLaunchKernelGGL(k_mid_rank1, dim3(g_nn), dim3(256), 0, stream, P);
            }
            check_launch();
            KTN_HIP(hipMemcpyAsync(&hs, md_st.p, sizeof(hs), hipMemcpyDeviceToHost, stream));
            sync();
            md_since_refactor += hs.pivots - pivots_seen;
            pivots_seen = hs.pivots;
            if (hs.gj_singular) { bad_inverse = true; break; }            // the working set itself is (numerically) dependent: cold start
            if (hs.status == 0) {
                if (md_since_refactor >= kMidRefactor) refactor();
                continue;
            }
            if (hs.status == 1) {
                if (refined_at == hs.pivots) {          // the confirming price after the refinement found nothing either
                    if (!(hs.resid <= 1e-7 * hs.scale)) { bad_inverse = true; break; }
                    status = 0;
                    break;
                }
                if (++refinements > 50) break;
                refine();
                refined_at = hs.pivots;
                continue;
            }
            status = hs.status;                          // 3 infeasible, 4 failed
            break;
        }
        total_pivots += hs.pivots;
        if (status == 0 || status == 3) break;
        md_valid = false;                                // failed on a warm start (or a decayed inverse): once more from the bound vertex
        if (!bad_inverse && attempt == 0 && hs.pivots >= P.max_pivots) break;      // (a pivot limit is not repaired by a restart)
    }
    stats["mid_lp_solves"] += 1.0;
    stats["mid_lp_pivots"] += (double)total_pivots;
    stats["lp_solves"] += 1.0;
    bool ok = false;
    if (status == 0) {
        KTN_HIP(hipMemsetAsync(lp_y.p, 0, (size_t)std::max<int64_t>(m, 1) * sizeof(double), stream));
        hipLaunchKernelGGL(k_mid_final, dim3(1), dim3(256), 0, stream, P, lp_y.p);
        KTN_HIP(hipMemcpyAsync(lp_x.p, md_x.p, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, stream));
        KTN_HIP(hipMemcpyAsync(&hs, md_st.p, sizeof(hs), hipMemcpyDeviceToHost, stream));
        check_launch();
        sync();
        if (hs.art_left == 0) {
            R->status = KTN_STATUS_OPTIMAL;
            R->iters = total_pivots;
            R->pobj = R->dobj = hs.obj;
            R->row_viol = 0.0; R->gap = 0.0;
            R->exact = true;
            objval = P.sgn * hs.obj + c0;
            md_valid = true;
            ok = true;
        } else {
            md_valid = false;
        }
    } else if (status == 3) {
        R->status = KTN_STATUS_INFEASIBLE;
        R->iters = total_pivots;
        md_valid = false;
        ok = true;
    }
    stats["lp_time_s"] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return ok;
}

LpResult Engine::lp_solve_core(double tol_p, double tol_g, int mode, bool identity_scaling) {
    auto t0 = std::chrono::steady_clock::now();
    LpResult R;
    auto lap = [&](const char* key, std::chrono::steady_clock::time_point& tp) {     // setup breakdown (profile runs only)
        if (!prm.profile) return;
        sync();
        const auto now = std::chrono::steady_clock::now();
        stats[key] += std::chrono::duration<double>(now - tp).count();
        tp = now;
    };
    auto tp = t0;
    ensure_matrix(want_shift(mode));
    lap("lp_csc_time_s", tp);
    // (row-sharded: the versions are per rank while the scaling is a collective -- no reuse there)
    const bool no_reuse = dev.no_setup_reuse;
    const bool same_matrix = !no_reuse && !row_sharded() && scaled_version == lp_version && scaled_identity == identity_scaling;
    find_long_rows();                                   // (before the scaling: its row passes treat long rows separately)
    if (!same_matrix) compute_scaling(identity_scaling);
    else stats["lp_setup_reuses"] += 1.0;
    lap("lp_scaling_time_s", tp);
    const int64_t n = n_lp, m = M;
    const size_t mm = (size_t)std::max<int64_t>(m, 1);
    ch.resize(n, stream); lh.resize(n, stream); uh.resize(n, stream); xh.resize(n, stream);
    x0h.resize(n, stream); xth.resize(n, stream); xbar.resize(n, stream); pv.resize(n, stream);
    loh.resize(mm, stream); hih.resize(mm, stream); yh.resize(mm, stream); y0h.resize(mm, stream);
    yth.resize(mm, stream); pw.resize(mm, stream);
    const double sgn = (sense == KTN_MAX) ? -1.0 : 1.0;
    LAUNCH_1(k_prep_cols, n, stream, n, Wc(), lp_l.p, lp_u.p, dc.p, lp_x.p, (mode == 1 ? box.p : (double*)nullptr), sgn,
             mode, ch.p, lh.p, uh.p, xh.p);
    LAUNCH_1(k_prep_rows, m, stream, m, Wlo(), Whi(), dr.p, lp_y.p, mode, loh.p, hih.p, yh.p);
    if (w_shift) {                                      // the epigraph variable of the start: s = t - a_ref'x - b_ref
        epi_dot(lp_x.p);
        hipLaunchKernelGGL(k_epi_var, dim3(1), dim3(1), 0, stream, xh.p, (int32_t)n0, epi_scal.p, (const double*)dc.p, -1, have_omega ? 0 : 1, sgn, epi_newest.p);
    }
    check_launch();
    const double avg_r = m ? (double)NNZ / (double)m : 1.0, avg_c = n ? (double)NNZ / (double)n : 1.0;
    grp_rows = pick_group(avg_r);
    grp_cols = pick_group(avg_c);
    if (dev.grp_rows > 0) grp_rows = dev.grp_rows;
    if (dev.grp_cols > 0) grp_cols = dev.grp_cols;
    SpMat A{lp_rowptr.p, lp_col.p, r_sval.p};
    SpMat AT{c_ptr.p, c_row.p, c_sval.p};
    // LPs beyond the caches: tiled copies of A^ and A^' (kernels.hpp "tiled SpMV") serve the plain steps and the check
    // iterations; the power iteration keeps the CSR / CSC kernels
    {
        const bool tenv = dev.tiled >= 0;
        // ... and dense enough: every (tile, block) unit stages a 64 KB block of the input vector, so a matrix with few entries
        // per unit pays more for the staging than for its entries (n = 1e6, 5.8e6 entries: 350 per unit, 1.31 s tiled against
        // 0.49 s with the CSR kernels; cfg4: 9 500 per unit)
        const int64_t units_t = ceil_div(M, (int64_t)kTileOut) * ceil_div(n_lp, (int64_t)kTileIn);
        const int64_t units_tt = ceil_div(n_lp, (int64_t)kTileOut) * ceil_div(M, (int64_t)kTileIn);
        tiled_on = prm.lp_tiled_nnz > 0 && NNZ >= prm.lp_tiled_nnz && n_lp >= 2 * kTileIn && M >= 2 * kTileIn &&
                   NNZ >= 4096 * std::max(units_t, units_tt);
        if (tenv) tiled_on = dev.tiled != 0 && M > 0 && NNZ > 0;
        if (same_matrix) tiled_on = tiled_built;            // the copies of the previous solve (or their absence) still fit
        else if (tiled_on) {
            auto tt = std::chrono::steady_clock::now();
            tiled_on = build_tiled(tA, M, n_lp, lp_rowptr.p, lp_col.p, r_sval.p, kLongRow) &&
                       build_tiled(tAT, n_lp, M, c_ptr.p, c_row.p, c_sval.p, (int64_t)1 << 62);
            tpart.resize((size_t)std::max<int64_t>(tA.pieces * M, tAT.pieces * n_lp), stream);
            stats["lp_tiled_builds"] += 1.0;
            stats["lp_tiled_build_time_s"] += std::chrono::duration<double>(std::chrono::steady_clock::now() - tt).count();
            if (!tiled_on) stats["lp_tiled_overflows"] += 1.0;
        }
        tiled_built = tiled_on;
        scaled_version = lp_version; scaled_identity = identity_scaling;
    }

    // Step size eta = 0.998 / sigma_max(A^).  sigma_max comes from 8 power iterations (round 1: 20; hashed start
    // vector: a constant one can be orthogonal to every row).  The power iteration approaches sigma_max
    // from BELOW, and an estimate a few percent low makes PDHG stall in a limit cycle (seen on dense
    // epigraph cuts: constant fixed-point residual, 8e-6 row violation), so the main loop watches for
    // that stall and backs eta off towards eta_safe.  With the Pock-Chambolle (alpha = 1) pass applied
    // last ||A^||_2 <= 1 is guaranteed (Pock & Chambolle 2011, Lemma 2): eta_safe = 0.998; always using
    // it costs 40 % (cfg3) to 170 % (cfg2) more PDHG iterations than the estimate.
    double smax = 0.0;
    bool have_power = false;
    // sigma_max of the previous solve is reused while the matrix has grown by less than KTN_SMAX_REUSE (a fraction of its
    // rows) since the estimate was made (development switch, default off)
    const double smax_reuse = dev.smax_reuse;
    const bool reuse_smax = mode == 0 && smax_reuse > 0.0 && smax_rows > 0 && m >= smax_rows && !row_sharded() &&
                            (double)(m - smax_rows) <= smax_reuse * (double)smax_rows && smax_prev > 0.0;
    if (same_matrix && smax_version == lp_version && smax_prev > 0.0) smax = smax_prev;       // same matrix, same estimate
    else if (reuse_smax) { smax = smax_prev; stats["lp_smax_reused"] += 1.0; }
    else if ((m > 0 && NNZ > 0) || row_sharded()) {
        // 8 passes from a hashed start vector (a looser estimate is a larger step: 20 -> 8 passes saves 6 % on cfg3 and 8 %
        // on cfg4 beyond the passes themselves; the back-off safeguards of the main loop catch an estimate that is too low).  Norms stay on the device (k_normalize reads them): one host
        // round trip at the end instead of one per pass.  (Measured: warm-starting v from the previous LP makes
        // the estimate tighter and the step therefore smaller -- cfg3 then needs 14 700 instead of 7 800 PDHG
        // iterations; boosting eta by 5 % over the tight estimate stalls the method.  The slightly generous
        // cold estimate plus the back-off safeguard is the better operating point.)
        double* nrm = chkout.p + 2 * kChkQ + 1;
        power_v.resize(n, stream);
        LAUNCH_1(k_hash_fill, n, stream, n, power_v.p);
        auto dot_dev = [&](const double* a, double* out) {
            hipLaunchKernelGGL(k_dot_partial, dim3(kRedBlocks), dim3(kBlock), 0, stream, n, a, a, partials.p);
            hipLaunchKernelGGL(k_sum_final, dim3(1), dim3(kRedBlocks), 0, stream, partials.p, kRedBlocks, out);
        };
        dot_dev(power_v.p, nrm);
        LAUNCH_1(k_normalize, n, stream, n, power_v.p, nrm, pv.p);
        const int passes_env = dev.power_passes;
        const int iters = passes_env > 0 ? passes_env : 8;
        // The iterate is re-normalised only every fourth pass (and before the last, whose ||A'A v|| with ||v|| = 1 is the
        // estimate): with ||A^||_2 <= 1 after the Pock-Chambolle pass the un-normalised vector only shrinks slowly, and the
        // Rayleigh quotient does not depend on the scale -- 6 instead of 20 (dot, final sum, normalise) triples per LP solve.
        for (int it = 0; it < iters; ++it) {
            if (n_long > 0) {
                LAUNCH_G(grp_rows, k_spmv_skip, m, stream, m, A, pv.p, pw.p, kLongRow);
                hipLaunchKernelGGL(k_spmv_long, dim3((unsigned)n_long), dim3(1024), 0, stream, d_longrows.p, A, pv.p, pw.p);
            } else {
                LAUNCH_G(grp_rows, k_spmv, m, stream, m, A, pv.p, pw.p);
            }
            const bool norm_now = (it % 4 == 3) || it >= iters - 2;
            if (norm_now) {
                spmv_cols(AT, pw.p, xbar.p);
                allreduce(xbar.p, (size_t)n, 0);            // row-sharded: A'A v = sum over the ranks of A_r'(A_r v)
                dot_dev(xbar.p, nrm);                       // on the last pass: ||A'A v||^2 with ||v|| = 1
                LAUNCH_1(k_normalize, n, stream, n, xbar.p, nrm, pv.p);
            } else {
                spmv_cols(AT, pw.p, pv.p);
                allreduce(pv.p, (size_t)n, 0);
            }
        }
        have_power = true;
    }
    // The scalars the loop needs -- the power iteration's ||A'A v||, ||A^||_F^2, ||c^||^2 and the finite parts of ||lo^||^2, ||hi^||^2 --
    // are all queued into slots behind chkout's check sums and come back with ONE copy and ONE host synchronisation (they
    // were five round trips, each idling the GPU for ~30 us).
    double* slots = chkout.p + 2 * kChkQ;           // [1] power, [2] fro2, [3] nc2, [4] |lo|^2, [5] |hi|^2
    auto reduce_into = [&](bool finite_sq, int64_t cnt, const double* a, int slot) {
        if (finite_sq) hipLaunchKernelGGL(k_finite_sq_partial, dim3(kRedBlocks), dim3(kBlock), 0, stream, cnt, a, partials.p);
        else hipLaunchKernelGGL(k_dot_partial, dim3(kRedBlocks), dim3(kBlock), 0, stream, cnt, a, a, partials.p);
        hipLaunchKernelGGL(k_sum_final, dim3(1), dim3(kRedBlocks), 0, stream, partials.p, kRedBlocks, slots + slot);
    };
    if (NNZ > 0) reduce_into(false, NNZ, r_sval.p, 2);
    reduce_into(false, n, ch.p, 3);
    if (m > 0) { reduce_into(true, m, loh.p, 4); reduce_into(true, m, hih.p, 5); }
    if (w_shift) {                                      // ||c|| of the STORED cost vector: the scale of the (unscaled) dual-residual tolerance
        hipLaunchKernelGGL(k_dot_partial, dim3(kRedBlocks), dim3(kBlock), 0, stream, n, lp_c.p, lp_c.p, partials.p);
        hipLaunchKernelGGL(k_sum_final, dim3(1), dim3(kRedBlocks), 0, stream, partials.p, kRedBlocks, slots + 6);
    }
    // Initial primal weight of a solve that has none to inherit: ||c^|| / ||b^|| with each norm taken as sqrt(count) x geometric
    // mean of the magnitudes (KTN_OMEGA_ROBUST=0: the plain 2-norms).  A few columns whose only entries are ~1e-6 get column
    // factors of 1e5-1e6 and with them c^_j ~ 1e6: three such columns among 1e5 make ||c^||_2 a thousand times the typical
    // magnitude -- cfg3's first LP started at a weight of 2 686, settled at 2.7 six restarts later and took 744 iterations; with
    // this statistic it starts at 3.2 and takes 220.  64 / 48 / 16 seeds: cfg3 -5.5 %, cfg2 -10 %, cfg4 +4 % (-3 % iterations).
    const int omega_robust = dev.omega_robust;
    const bool robust = omega_robust && mode == 0 && !have_omega && !row_sharded() && m > 0;
    if (robust) {                                       // log-magnitude statistics of c^ and of the finite row bounds: slots 8..11
        auto logstat = [&](int64_t cnt, const double* a, const double* b, int slot) {
            hipLaunchKernelGGL(k_logabs_partial, dim3(kRedBlocks), dim3(kBlock), 0, stream, cnt, a, b, partials.p);
            hipLaunchKernelGGL(k_sum_final, dim3(1), dim3(kRedBlocks), 0, stream, partials.p, kRedBlocks, slots + slot);
            hipLaunchKernelGGL(k_sum_final, dim3(1), dim3(kRedBlocks), 0, stream, partials.p + kRedBlocks, kRedBlocks, slots + slot + 1);
        };
        logstat(n, ch.p, nullptr, 8);
        logstat(m, loh.p, hih.p, 10);
    }
    double hs[12] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    KTN_HIP(hipMemcpyAsync(hs, slots, sizeof(hs), hipMemcpyDeviceToHost, stream));
    double epi_b = 0.0;                                 // b_ref of the working form: the objective constant it carries
    if (w_shift) KTN_HIP(hipMemcpyAsync(&epi_b, epi_scal.p, sizeof(double), hipMemcpyDeviceToHost, stream));
    sync();
    const double obj_shift = sgn * epi_b;               // internal objective of the stored LP = working objective + obj_shift
    if (have_power) {
        const double nv2 = hs[1];
        smax = (nv2 > 0.0) ? std::sqrt(std::sqrt(nv2)) : 0.0;     // sigma_max^2 ~ ||A'A v||
        if (mode == 0) { smax_prev = smax; smax_rows = m; smax_version = lp_version; }
    }
    lap("lp_power_time_s", tp);
    double fro2 = (NNZ > 0) ? hs[2] : 0.0;                                              // ||A||_2 <= ||A||_F
    allreduce_host(&fro2, 1, 0);
    const double fro = std::sqrt(fro2);
    if (!(smax > 0.0)) smax = fro;
    const double eta_safe = 0.998 / std::max(identity_scaling ? fro : std::min(1.0, fro), 1e-12);
    double eta = std::max(0.998 / std::max(smax, 1e-12), eta_safe);
    int stall = 0, grow = 0, flat_rows = 0, consolidations = 0, infeas_hits = 0;
    double pobj_h[3] = {1e300, -1e300, 1e300}, pviol_h[3] = {1e300, -1e300, 1e300};
    double r_last_check = 0.0;
    stats["lp_setup_time_s"] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const double nc2 = hs[3];
    const double nc2_tol = w_shift ? hs[6] : nc2;       // (the working cost carries a_ref: not the scale the dual residual is judged on)
    double nb2 = (m > 0) ? hs[4] + hs[5] : 0.0;
    allreduce_host(&nb2, 1, 0);
    double omega_ref = (nc2 > 0.0 && nb2 > 0.0) ? std::sqrt(nc2 / nb2) : 1.0;
    if (robust && hs[9] > 0.0 && hs[11] > 0.0) {
        omega_ref = std::sqrt(hs[9] / hs[11]) * std::exp(hs[8] / hs[9] - hs[10] / hs[11]);
        stats["lp_omega_robust"] += 1.0;
    }
    double om = (have_omega && mode == 0) ? omega : omega_ref;
    const double rho = 1.0;
    const double cinf_scale = 1.0;
    if (dev.debug_lp)
        std::fprintf(stderr, "[lp setup mode %d] m %lld n %lld nnz %lld smax %.4g fro %.4g nc2 %.4g nc2_tol %.4g nb2 %.4g omega_ref %.4g om0 %.4g shift %d b_ref %.9g n_long %lld tol_p %.3g tol_g %.3g\n",
                     mode, (long long)m, (long long)n, (long long)NNZ, smax, fro, nc2, nc2_tol, nb2, omega_ref, om, (int)w_shift, epi_b, (long long)n_long, tol_p, tol_g);

    // anchors z0 = z; with the packed records of the plain steps (not for the tiled / row-sharded forms, whose steps are
    // split into SpMV + element-wise kernels)
    packed_on = !tiled_on && !row_sharded() && NNZ < ((int64_t)1 << 31) && !dev.no_packed;
    if (dev.packed_trips > 0) packed_trips = dev.packed_trips;
    if (packed_on) {
        d_crec.resize((size_t)n, stream); d_cbl.resize((size_t)n, stream); d_rrec.resize(mm, stream);
        LAUNCH_1(k_pack_cols, n, stream, n, c_ptr.p, ch.p, lh.p, uh.p, xh.p, x0h.p, d_crec.p, d_cbl.p);
        LAUNCH_1(k_pack_rows, m, stream, m, lp_rowptr.p, loh.p, hih.p, yh.p, y0h.p, d_rrec.p);
    } else {
        KTN_HIP(hipMemcpyAsync(x0h.p, xh.p, n * sizeof(double), hipMemcpyDeviceToDevice, stream));
        if (m > 0) KTN_HIP(hipMemcpyAsync(y0h.p, yh.p, m * sizeof(double), hipMemcpyDeviceToDevice, stream));
    }

    const double ky_bytes = (double)NNZ * 12 + 8.0 * (m + 1) + 8.0 * 5 * m + 8.0 * n;
    const double kx_bytes = (double)NNZ * 12 + 8.0 * (n + 1) + 8.0 * 7 * n + 8.0 * m;
    int64_t k = 0, it = 0;
    double r0 = 0.0, r_prev = 0.0;
    const bool dbg_lp = dev.debug_lp;
    R.status = KTN_STATUS_USERLIMIT;
    const int64_t max_it = lp_iter_budget > 0 ? std::min<int64_t>(lp_iter_budget, prm.lp_max_iter) : prm.lp_max_iter;
    const int chk = std::max(1, prm.lp_check_every);
    const int plain_len = std::min(chk - 1, (int)kMaxChunk);
    const int first_chunk = dev.first_chunk;
    bool plain_next = false, near_conv = false, primal_ok = false;
    const int stag_chunk = dev.stag_chunk;
    // throughput mode: one workgroup per block runs its LP to the end (batch_lp.hpp); the ordinary loop below only serves
    // as the fall-back when a block reports that it could not finish
    bool blocks_done = false;
    if (n_blocks > 0 && mode == 0 && !row_sharded() && n_long == 0 && m > 0 && !prm.profile) {
        if (blocks_built_rows != M) build_blocks();
        blocks_done = lp_solve_blocks(tol_p, tol_g, eta, &R, max_it);
        if (blocks_done) it = R.iters;
    }
    const int near_env = dev.near_chunk;
    const int near_chunk = near_env >= 0 ? near_env : prm.lp_near_check;
    while (!blocks_done && it < max_it) {
        const double tau = eta / om, sigma = eta * om;
        if (plain_next) {
            // ---- a chunk of plain (update) iterations between two checks
            plain_next = false;
            // the first chunk after a restart is shorter: the restarted iteration moves fastest there and an
            // early check catches the next restart / termination sooner
            int want = (k <= 1 && first_chunk > 0) ? std::min(first_chunk, plain_len) : plain_len;
            // close to the tolerances the next check comes sooner: a solve ends on average half a chunk after it converged
            if (near_conv && near_chunk > 0) want = std::min(want, near_chunk);
            if (primal_ok && stag_chunk > 0 && mode == 0) want = std::min(want, stag_chunk);
            const int np = (int)std::min<int64_t>(want, max_it - it);
            if (np <= 0) continue;
            for (int j = 0; j < np; ++j) {
                const double w = (double)(k + j + 1) / (double)(k + j + 2);
                if (prm.profile) {
                    const size_t e0 = ev_get(), e1 = ev_get(), e2 = ev_get(), e3 = ev_get();
                    launch_x(AT, tau, w, rho, true, ev_pool[e0], ev_pool[e1]);
                    launch_y(A, sigma, w, rho, ev_pool[e2], ev_pool[e3]);
                    ev_recs.push_back({0, e0, e1, kx_bytes});
                    if (m > 0) ev_recs.push_back({1, e2, e3, ky_bytes});
                } else {
                    launch_x(AT, tau, w, rho, true, nullptr, nullptr);
                    launch_y(A, sigma, w, rho, nullptr, nullptr);
                }
            }
            k += np; it += np;
            continue;
        }
        // ---- check iteration: PDHG step without update, KKT + fixed-point residual
        launch_check(A, AT, tau, sigma);
        check_launch();
        double q[2 * kChkQ];
        if (chk_pinned()) {
            sync();
            std::memcpy(q, h_chk, sizeof(q));
        } else {
            KTN_HIP(hipMemcpyAsync(q, chkout.p, sizeof(q), hipMemcpyDeviceToHost, stream));
            sync();
            ipc_check();
        }
        if (prm.profile) ev_flush();
        const double dyAdx = q[0], dy2 = q[1], dobj_rows = q[2], dy0sq = q[3], yt2 = q[4], pviol = q[12];
        const double dx2 = q[kChkQ + 5], pobj = q[kChkQ + 6] + obj_shift, dobj_cols = q[kChkQ + 7], dx0sq = q[kChkQ + 8],
                     xt2 = q[kChkQ + 9], dres = q[kChkQ + 13];
        const double dobj = dobj_rows + dobj_cols + obj_shift;
        const double r2 = om / eta * dx2 - 2.0 * dyAdx + dy2 / (eta * om);
        const double r = std::sqrt(std::max(r2, 0.0));
        const double gap = std::fabs(pobj - dobj) / (1.0 + std::fabs(pobj) + std::fabs(dobj));
        if (k == 0) { r0 = r; r_prev = r; }
        if (dbg_lp) std::fprintf(stderr, "[lp mode %d] it %7lld k %6lld r %.3e pviol %.3e dres %.3e gap %.3e pobj %.10g dobj %.10g om %.3g eta %.3g\n",
                                 mode, (long long)it, (long long)k, r, pviol, dres, gap, pobj, dobj, om, eta);
        R.pobj = pobj; R.dobj = dobj; R.row_viol = pviol; R.gap = gap;
        R.dres_rel = dres * cinf_scale / (1.0 + std::sqrt(nc2_tol));
        bool done = (pviol <= tol_p) && (gap <= tol_g) && (dres * cinf_scale <= tol_g * (1.0 + std::sqrt(nc2_tol)));
        near_conv = (pviol <= 4.0 * tol_p) && (gap <= 4.0 * tol_g) && (dres * cinf_scale <= 4.0 * tol_g * (1.0 + std::sqrt(nc2_tol)));
        primal_ok = (pviol <= tol_p) && (dres * cinf_scale <= tol_g * (1.0 + std::sqrt(nc2_tol)));
        // Primal-stagnation exit (lp_stag_factor).  On LPs with degenerate duals the primal part converges within a few
        // hundred iterations while the duality gap crawls for 10 000 more: stop when the rows are
        // feasible to tol_p, the dual residual is converged, the primal objective has not moved by more than 0.4 tol_g
        // over the last two checks, and the gap is certified to lp_stag_factor * tol_g.
        {
            // A model with free variables runs inside the box the presolve put around it (boundroutine, model.jl:175-197): a
            // loosely solved LP can then sit ANYWHERE in that box, and cuts taken at |x| ~ 1e13 have constants of 1e31 that no
            // first-order LP survives (fuzz model 2/109: primal weight 1e-30, objective 6e26, :Error).  Such models keep the
            // conservative exit of round 2: three flat checks within 0.1 of the gap tolerance, gap certified to 100 tolerances.
            const bool boxed_free = has_inf_bound;
            const double stag = boxed_free ? std::min(prm.lp_stag_factor, 100.0) : prm.lp_stag_factor;
            if (stag > 0.0 && mode == 0 && !done) {
                const double scale = 1.0 + std::fabs(pobj);
                // (flat over the last TWO checks; round 2 asked for three.  Most loose solves of the BASELINE shapes end here, and
                //  the third confirmation was 64 more iterations each: -5 ... -7 % PDHG iterations on cfg3 / cfg2 / cfg4 over
                //  96 / 32 / 8 seeds, objective errors, the 82 reference models, 240 fuzz models and the 48-shape matrix
                //  unchanged.  KTN_STAG_CHECKS=3 restores the longer window.)
                const int stag_checks = dev.stag_checks;
                // ("flat" = within 0.4 tol_g; 0.1 until round 3.  The exit decides whether x* is a good separation point, not
                //  the stop of the ECP loop, and tol_g itself is the accuracy asked of this solve: -12 % PDHG iterations on
                //  cfg3 over 96 seeds, -9 % on cfg4, cfg2 unchanged, worst objective error 5e-7 of the 1e-6 allowed, the GPU
                //  suite, fuzz set and shape matrix unchanged.  KTN_FLAT_FACTOR overrides.)
                const double flat_f = dev.flat_factor;
                const double ff = boxed_free ? std::min(flat_f, 0.1) : flat_f;
                const int nchk = boxed_free ? 3 : stag_checks;
                const bool flat = std::fabs(pobj - pobj_h[0]) <= ff * tol_g * scale && std::fabs(pobj - pobj_h[1]) <= ff * tol_g * scale &&
                                  (nchk < 3 || std::fabs(pobj - pobj_h[2]) <= ff * tol_g * scale);
                // (a row violation that sits on a plateau -- unchanged to 2 % over three checks -- within the stalled-row allowance
                //  below counts as feasible here: cfg4 seed 2 idled 23 000 iterations at 3.098e-7 against tol_p = 3.0e-7 with the
                //  objective flat and the gap at 3 tol_g, so that neither exit applied)
                const double accept0 = (tol_p > prm.lp_tol_floor * prm.f_tol * (1.0 + 1e-9)) ? 10.0 : 3.0;
                const bool plateau = pviol <= accept0 * tol_p && std::fabs(pviol - pviol_h[0]) <= 0.02 * pviol &&
                                     std::fabs(pviol - pviol_h[1]) <= 0.02 * pviol && std::fabs(pviol - pviol_h[2]) <= 0.02 * pviol;
                if (flat && (pviol <= tol_p || plateau) && gap <= stag * tol_g && dres * cinf_scale <= tol_g * (1.0 + std::sqrt(nc2_tol))) {
                    done = true;
                    R.stag_exit = true;
                    stats["lp_stagnation_exits"] += 1.0;
                }
            }
            // ... and its mirror image: objective, gap and dual residual converged, but ONE row stays violated by a hair more
            // than tol_p for millions of iterations (multiplier mass idling between nearly parallel cuts, seen with dense
            // epigraph cuts after the consolidation budget is spent: 3.47e-7 against tol_p = 3.0e-7 for 2.1e6 iterations).
            // tol_p's floor is 0.3 f_tol -- a safety factor, the stop rule itself is the sweep at f_tol -- so a violation that
            // has not moved by 2 % over three checks is accepted up to 3 tol_p (0.9 f_tol).  An INTERMEDIATE solve (tol_p above its floor:
            // its x* only has to be a useful separation point, cuts are valid anywhere) accepts up to 10 tol_p -- the new cuts
            // of the next sweep are what ends such a stall (263 000 iterations at 6.25e-2 against 3e-2 otherwise).
            const double stall_accept = (tol_p > prm.lp_tol_floor * prm.f_tol * (1.0 + 1e-9)) ? 10.0 : 3.0;
            if (stag > 0.0 && mode == 0 && !done && gap <= tol_g && dres * cinf_scale <= tol_g * (1.0 + std::sqrt(nc2_tol)) &&
                pviol <= stall_accept * tol_p && std::fabs(pviol - pviol_h[0]) <= 0.02 * pviol && std::fabs(pviol - pviol_h[1]) <= 0.02 * pviol &&
                std::fabs(pviol - pviol_h[2]) <= 0.02 * pviol) {
                done = true;
                stats["lp_stalled_row_exits"] += 1.0;
            }
            pviol_h[2] = pviol_h[1]; pviol_h[1] = pviol_h[0]; pviol_h[0] = pviol;
            pobj_h[2] = pobj_h[1]; pobj_h[1] = pobj_h[0]; pobj_h[0] = pobj;
        }
        if (done || !(r == r)) {
            R.status = done ? KTN_STATUS_OPTIMAL : KTN_STATUS_ERROR;
            ++it;
            break;
        }
        // primal infeasibility: yt is a Farkas certificate when the dual objective of the c = 0 problem is
        // positive (weak duality makes it <= 0 for every sign-valid y of a feasible LP).  Two checks in a row.
        if (mode == 0 && (m > 0 || row_sharded())) {     // (row-sharded: every quantity below is all-reduced, so all ranks agree)
            const double farkas = dobj_rows + q[kChkQ + 10];
            const double mag = q[10] + q[kChkQ + 11] + 1e-300;
            const bool cert = farkas > 1e-6 * mag && q[kChkQ + 14] <= 1e-9 * (1.0 + std::sqrt(yt2)) && pviol > tol_p;
            infeas_hits = cert ? infeas_hits + 1 : 0;
            if (infeas_hits >= 2 && it >= 2 * chk) {
                R.status = KTN_STATUS_INFEASIBLE;
                ++it;
                break;
            }
        }
        bool restart = k > 0 && (r <= 0.2 * r0 || (r <= 0.8 * r0 && r > r_prev) || (double)k >= 0.36 * (double)(it + 1));
        const bool decayed = r <= 0.8 * r0;             // (a restart that the residual earned; the artificial one below is by the clock)
        // step-size safeguard: a fixed-point residual that no longer moves (or a negative M-norm) while
        // the LP is not solved means eta * sigma_max > 1 -> shrink eta and restart from the current point
        if (k > 0 && eta > eta_safe * (1.0 + 1e-12)) {
            stall = (r2 < 0.0 || (r_last_check > 0.0 && r > 0.97 * r_last_check && r < 1.03 * r_last_check)) ? stall + 1 : 0;
            // ... and the other face of the same fault: the residual GROWS check after check, far above the residual the
            // period started with (a non-expansive step never does that for long; seen on cfg3 seeds 28/29: r0 = 6 -> 87 -> 530
            // -> 1e5 over 10 000 iterations until the flat-residual rule above finally fired).  Two growing checks above 5 r0.
            grow = (r > 5.0 * r0 && r_last_check > 0.0 && r > r_last_check) ? grow + 1 : 0;
            if (stall >= 3 || r2 < 0.0 || grow >= 2) {
                eta = std::max(eta_safe, 0.85 * eta);
                if (grow >= 2) stats["lp_divergence_backoffs"] += 1.0;
                stall = 0; grow = 0;
                restart = true;
                stats["lp_eta_backoffs"] += 1.0;
            }
        }
        // objective converged, rows not, residual flat: PDHG is idling between near-parallel cuts of one
        // NL row (k_consolidate).  Move the multiplier mass onto the tightest cut at the current point and
        // restart from there.
        // (row-sharded: the decision must not depend on what THIS rank holds -- a rank that consolidated while another did not
        //  would leave the sequence of collectives -- so the local conditions are dropped; k_consolidate on a rank without cuts
        //  is a no-op)
        const bool have_lists = row_sharded() ? (prm.lp_dual_inherit != 0) : (prm.lp_dual_inherit && lists_ok() && list_count() > 0 && m > M_base);
        if (mode == 0 && k > 0 && have_lists && gap <= tol_g &&
            dres * cinf_scale <= tol_g * (1.0 + std::sqrt(nc2_tol)) && pviol > tol_p) {
            flat_rows = (r_last_check > 0.0 && r > 0.98 * r_last_check) ? flat_rows + 1 : 0;
            if (flat_rows >= 3 && consolidations < 8) {
                flat_rows = 0;
                ++consolidations;
                stats["lp_consolidations"] += 1.0;
                LAUNCH_1(k_unscale, n, stream, n, xth.p, dc.p, pv.p);
                SpMat Au{lp_rowptr.p, lp_col.p, Wval()};
                LAUNCH_G(grp_rows, k_spmv, m, stream, m, Au, pv.p, pw.p);
                KTN_HIP(hipMemsetAsync(d_anynf.p + 1, 0, sizeof(int32_t), stream));
                LAUNCH_1(k_consolidate, list_count(), stream, list_count(), list_heads(), d_cutprev.p, pw.p, Wlo(), Whi(), dr.p, tol_p, yth.p,
                         d_anynf.p + 1);
                LAUNCH_1(k_restart_set, std::max(n, m), stream, n, m, xth.p, xh.p, x0h.p, yth.p, yh.p, y0h.p, packed_on ? d_crec.p : (ColRec*)nullptr,
                         packed_on ? d_rrec.p : (RowRec*)nullptr);
                k = 0;
                r_last_check = 0.0;
                ++it;
                continue;
            }
        } else {
            flat_rows = 0;
        }
        r_last_check = r;
        r_prev = r;
        if (restart) {
            const double dx = std::sqrt(dx0sq), dy = std::sqrt(dy0sq);
            if (dbg_lp) std::fprintf(stderr, "[lp restart] it %lld k %lld decayed %d dx %.3e dy %.3e |xt| %.3e |yt| %.3e om %.4g\n", (long long)it, (long long)k, (int)decayed, dx, dy, std::sqrt(xt2), std::sqrt(yt2), om);
            // guarded primal-weight update (oracle/pdlp_mirror.py solve_lp_halpern)
            const int om_art = dev.omega_art;
            // Unearned restarts carry little information about the weight (round 4).  The update reads the ratio
            // of the two movements since the last restart.  A solve's first restarts come "by the clock" (k >= 0.36 it: at the
            // first check of every solve, after 32 iterations), whether or not the residual has moved.  After a warm start whose
            // primal part is already converged (cfg2 seed 92: row violation 2e-7, dual objective 3e-4 away) x moves by 1e-6 of
            // its norm in such a period -- the size of the tolerance -- and the ratio of that movement to the dual's sent the
            // weight 603 -> 140 -> 13.8 -> 0.77 -> 0.16 in four restarts that had not reduced the residual at all; the solve then
            // needed 44 000 iterations to earn it back.  (Tried first, and harmful: skipping the update below a relative movement
            // of 10-100 gap tolerances -- any such floor also silences the early, loose solves, whose movements are small AND
            // informative: cfg3 13 -> 36-85 cutting-plane rounds, profiles/r04_omega_ab.txt.)  Instead the weight of the new
            // ratio in the geometric mean grows with the length of the period it was measured over: theta = 0.5 min(1, k / K)
            // for a restart the residual did not earn (K = KTN_OMEGA_ART_K, 0 = the plain 0.5), 0.5 for an earned one.
            const double om_art_k = dev.omega_art_k, om_art_clamp = dev.omega_art_clamp;
            if ((om_art || decayed) && dx > 1e-8 * (1.0 + std::sqrt(xt2)) && dy > 1e-8 * (1.0 + std::sqrt(yt2))) {
                const double om_clamp = dev.omega_clamp;
                const double om_old = om;
                // (models with free variables run inside the box the presolve put around them and keep the round-3 rule, like their
                //  exit rules above: fuzz model 2/109 ends `:Error` -- a cut taken 1e13 from the origin -- under the damped update)
                const double theta = (!decayed && om_art_k > 0.0 && !has_inf_bound) ? 0.5 * std::min(1.0, (double)k / om_art_k) : 0.5;
                om = std::exp(theta * std::log(dy / dx) + (1.0 - theta) * std::log(om));
                if (!decayed && om_art_clamp > 1.0) om = std::min(std::max(om, om_old / om_art_clamp), om_old * om_art_clamp);
                const double om_clamp_dn = dev.omega_clamp_down;
                if (om_clamp > 1.0) om = std::min(std::max(om, om_old / om_clamp), om_old * om_clamp);
                if (om_clamp_dn > 1.0) om = std::max(om, om_old / om_clamp_dn);
                om = std::min(std::max(om, omega_ref * 1e-3), omega_ref * 1e3);
            }
            LAUNCH_1(k_restart_set, std::max(n, m), stream, n, m, xth.p, xh.p, x0h.p, yth.p, yh.p, y0h.p, packed_on ? d_crec.p : (ColRec*)nullptr,
                         packed_on ? d_rrec.p : (RowRec*)nullptr);
            stats["lp_restarts"] += 1.0;
            k = 0;
            ++it;
            continue;
        }
        const double w = (double)(k + 1) / (double)(k + 2);
        LAUNCH_1(k_halpern2, std::max(n, m), stream, n, m, xh.p, xth.p, x0h.p, yh.p, yth.p, y0h.p, w, rho);
        ++k; ++it;
        plain_next = true;       // (after a restart k == 0 and the next pass is a check again: it needs r0)
    }
    R.iters = it;
    // un-scale the last PDHG point (xt, yt)
    if (mode == 0) {
        LAUNCH_1(k_unscale, n, stream, n, xth.p, dc.p, lp_x.p);
        LAUNCH_1(k_unscale, m, stream, m, yth.p, dr.p, lp_y.p);
        if (w_shift) {                                  // back to the epigraph variable of the stored LP: t = s + a_ref'x + b_ref
            epi_dot(lp_x.p);
            hipLaunchKernelGGL(k_epi_var, dim3(1), dim3(1), 0, stream, lp_x.p, (int32_t)n0, epi_scal.p, (const double*)nullptr, 1, 0, sgn, epi_newest.p);
        }
        omega = om;
        have_omega = true;
        objval = sgn * R.pobj + c0;
    } else {
        LAUNCH_1(k_unscale, n, stream, n, xth.p, dc.p, d_ray.p);
    }
    check_launch();
    sync();
    if (prm.profile) ev_flush();
    stats["pdhg_iters"] += (double)it;
    stats["lp_solves"] += 1.0;
    // (the quantities Engine::step's floor rule reads: a host-driven loop over the same entry points -- distributed.py -- needs them too)
    stats["lp_last_row_viol"] = R.row_viol; stats["lp_last_gap"] = R.gap; stats["lp_last_dres_rel"] = R.dres_rel;
    stats["lp_last_stag_exit"] = R.stag_exit ? 1.0 : 0.0;
    stats["lp_time_s"] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return R;
}

void Engine::pdhg_raw(const double* x0, const double* y0, double eta, double omega_, int64_t iters, double* x_out,
                      double* y_out) {
    ensure_matrix(false);
    compute_scaling(true);
    scaled_version = 0;                                  // (lp_solve_core must not take this identity scaling for its own)
    const int64_t n = n_lp, m = M;
    const size_t mm = (size_t)std::max<int64_t>(m, 1);
    ch.resize(n, stream); lh.resize(n, stream); uh.resize(n, stream); xh.resize(n, stream);
    x0h.resize(n, stream); xth.resize(n, stream); xbar.resize(n, stream); pv.resize(n, stream);      // (pv: vector form of the x-step)
    loh.resize(mm, stream); hih.resize(mm, stream); yh.resize(mm, stream); y0h.resize(mm, stream); yth.resize(mm, stream); pw.resize(mm, stream);
    const double sgn = (sense == KTN_MAX) ? -1.0 : 1.0;
    KTN_HIP(hipMemcpyAsync(lp_x.p, x0, n * sizeof(double), hipMemcpyHostToDevice, stream));
    if (m > 0) KTN_HIP(hipMemcpyAsync(lp_y.p, y0, m * sizeof(double), hipMemcpyHostToDevice, stream));
    LAUNCH_1(k_prep_cols, n, stream, n, lp_c.p, lp_l.p, lp_u.p, dc.p, lp_x.p, (double*)nullptr, sgn, 0, ch.p, lh.p, uh.p, xh.p);
    LAUNCH_1(k_prep_rows, m, stream, m, lp_lo.p, lp_hi.p, dr.p, lp_y.p, 0, loh.p, hih.p, yh.p);
    grp_rows = pick_group(m ? (double)NNZ / (double)m : 1.0);
    grp_cols = pick_group(n ? (double)NNZ / (double)n : 1.0);
    SpMat A{lp_rowptr.p, lp_col.p, r_sval.p};
    SpMat AT{c_ptr.p, c_row.p, c_sval.p};
    packed_on = false;
    KTN_HIP(hipMemcpyAsync(x0h.p, xh.p, n * sizeof(double), hipMemcpyDeviceToDevice, stream));
    if (m > 0) KTN_HIP(hipMemcpyAsync(y0h.p, yh.p, m * sizeof(double), hipMemcpyDeviceToDevice, stream));
    find_long_rows();
    // LPs beyond the caches: tiled copies of A^ and A^' (kernels.hpp "tiled SpMV") serve the plain steps and the check
    // iterations; the power iteration keeps the CSR / CSC kernels
    {
        const bool tenv = dev.tiled >= 0;
        // ... and dense enough: every (tile, block) unit stages a 64 KB block of the input vector, so a matrix with few entries
        // per unit pays more for the staging than for its entries (n = 1e6, 5.8e6 entries: 350 per unit, 1.31 s tiled against
        // 0.49 s with the CSR kernels; cfg4: 9 500 per unit)
        const int64_t units_t = ceil_div(M, (int64_t)kTileOut) * ceil_div(n_lp, (int64_t)kTileIn);
        const int64_t units_tt = ceil_div(n_lp, (int64_t)kTileOut) * ceil_div(M, (int64_t)kTileIn);
        tiled_on = prm.lp_tiled_nnz > 0 && NNZ >= prm.lp_tiled_nnz && n_lp >= 2 * kTileIn && M >= 2 * kTileIn &&
                   NNZ >= 4096 * std::max(units_t, units_tt);
        if (tenv) tiled_on = dev.tiled != 0 && M > 0 && NNZ > 0;
        if (tiled_on) {
            auto tt = std::chrono::steady_clock::now();
            tiled_on = build_tiled(tA, M, n_lp, lp_rowptr.p, lp_col.p, r_sval.p, kLongRow) &&
                       build_tiled(tAT, n_lp, M, c_ptr.p, c_row.p, c_sval.p, (int64_t)1 << 62);
            tpart.resize((size_t)std::max<int64_t>(tA.pieces * M, tAT.pieces * n_lp), stream);
            stats["lp_tiled_builds"] += 1.0;
            stats["lp_tiled_build_time_s"] += std::chrono::duration<double>(std::chrono::steady_clock::now() - tt).count();
            if (!tiled_on) stats["lp_tiled_overflows"] += 1.0;
        }
    }
    const double tau = eta / omega_, sigma = eta * omega_;
    const double ky_bytes = (double)NNZ * 12 + 8.0 * (m + 1) + 8.0 * 5 * m + 8.0 * n;
    const double kx_bytes = (double)NNZ * 12 + 8.0 * (n + 1) + 8.0 * 7 * n + 8.0 * m;
    for (int64_t k = 0; k < iters; ++k) {
        const double w = (double)(k + 1) / (double)(k + 2);
        if (prm.profile) {
            const size_t e0 = ev_get(), e1 = ev_get(), e2 = ev_get(), e3 = ev_get();
            launch_x(AT, tau, w, 1.0, true, ev_pool[e0], ev_pool[e1]);
            launch_y(A, sigma, w, 1.0, ev_pool[e2], ev_pool[e3]);
            ev_recs.push_back({0, e0, e1, kx_bytes});
            if (m > 0) ev_recs.push_back({1, e2, e3, ky_bytes});
        } else {
            launch_x(AT, tau, w, 1.0, true, nullptr, nullptr);
            launch_y(A, sigma, w, 1.0, nullptr, nullptr);
        }
        if ((k & 255) == 255) { sync(); if (prm.profile) ev_flush(); }
    }
    sync();
    if (prm.profile) ev_flush();
    check_launch();
    KTN_HIP(hipMemcpyAsync(x_out, xh.p, n * sizeof(double), hipMemcpyDeviceToHost, stream));
    if (m > 0) KTN_HIP(hipMemcpyAsync(y_out, yh.p, m * sizeof(double), hipMemcpyDeviceToHost, stream));
    sync();
}

// ------------------------------------------------------------------------------------
// ECP driver
// ------------------------------------------------------------------------------------
// "status == :Unbounded" + getunboundedray: solve the recession-cone LP (oracle/lp.py).
bool Engine::recession_ray() {
    box.resize((size_t)n_lp, stream);
    LAUNCH_1(k_fill, n_lp, stream, n_lp, box.p, 1.0);
    if (!obj_linear) {
        KTN_HIP(hipMemsetAsync(d_scal.p + 1, 0, sizeof(double), stream));
        LAUNCH_1(k_aux_box, M, stream, M, lp_rowptr.p, lp_col.p, lp_val.p, (int32_t)n0, d_scal.p + 1);
        double w = 0.0;
        KTN_HIP(hipMemcpyAsync(&w, d_scal.p + 1, 8, hipMemcpyDeviceToHost, stream));
        sync();
        allreduce_host(&w, 1, 1);
        LAUNCH_1(k_fill, 1, stream, (int64_t)1, box.p + n0, 1.0 + w);
    }
    // Small models (the reference's own tests: a handful of free variables): the recession LP by the exact kernel.  It is a
    // degenerate LP with tolerances of 1e-9 -- the first-order method can need more than its iteration limit for six rows and
    // five columns (fuzz model 13/142: 2e6 iterations, limit reached, "no ray" reported, and the main LP then ran along the ray
    // it had missed until ITS limit: :UserLimit where the oracle ends :Optimal) -- and the simplex ray is what the reference
    // hands to boundroutine (src/model.jl:233-236).
    if (n_lp <= kDenseMaxN && !row_sharded() && prm.lp_dense_after != 0 && M * n_lp <= 8000000) {
        bool unb = false;
        if (recession_ray_dense(&unb)) return unb;
    }
    LpResult R = lp_solve(1e-9, 1e-7, 1);
    return R.status == KTN_STATUS_OPTIMAL && R.pobj < -1e-6;
}
// min c'd over the recession cone of the LP's rows inside the box: variables with a finite bound keep that side at 0, free sides
// get -/+ box; finite row sides become 0 (k_prep_cols / k_prep_rows, mode 1, with unit scaling).  d -> d_ray.  Returns false when
// the exact kernel gives up (the caller falls back to the first-order solve).
bool Engine::recession_ray_dense(bool* unbounded) {
    const int n = (int)n_lp;
    const int64_t m = M;
    const size_t mm = (size_t)std::max<int64_t>(m, 1);
    const double sgn = (sense == KTN_MAX) ? -1.0 : 1.0;
    DBuf<double> ones, rl, ru, rc, rx, rlo, rhi, ry, out;
    DBuf<int32_t> W, Wv;
    ones.resize(std::max<size_t>(mm, (size_t)n), stream);
    LAUNCH_1(k_fill, (int64_t)ones.n, stream, (int64_t)ones.n, ones.p, 1.0);
    rl.resize(n, stream); ru.resize(n, stream); rc.resize(n, stream); rx.resize(n, stream);
    rlo.resize(mm, stream); rhi.resize(mm, stream); ry.resize(mm, stream); out.resize(4, stream);
    W.resize(n, stream); Wv.resize(1, stream); Wv.zero(stream);
    lp_x.resize((size_t)n, stream);
    LAUNCH_1(k_prep_cols, n, stream, (int64_t)n, lp_c.p, lp_l.p, lp_u.p, ones.p, lp_x.p, box.p, 1.0, 1, rc.p, rl.p, ru.p, rx.p);
    LAUNCH_1(k_prep_rows, m, stream, m, lp_lo.p, lp_hi.p, ones.p, ones.p, 1, rlo.p, rhi.p, ry.p);
    ds_dense.resize(mm * (size_t)n, stream);
    d_ray.resize((size_t)n, stream);
    DenseLpIO P;
    P.n = n; P.m = m; P.rowptr = lp_rowptr.p; P.col = lp_col.p; P.val = lp_val.p; P.lo = rlo.p; P.hi = rhi.p;
    P.l = rl.p; P.u = ru.p; P.c = lp_c.p; P.sgn = sgn;
    P.dense = ds_dense.p; P.W = W.p; P.Wvalid = Wv.p; P.x = d_ray.p; P.y = ry.p; P.out = out.p;
    P.max_pivots = 200 + 20 * n + (int)std::min<int64_t>(m, 100000);
    P.tol = 1e-9;
    hipLaunchKernelGGL(k_dense_lp, dim3(1), dim3(256), 0, stream, P);
    check_launch();
    double o[4];
    KTN_HIP(hipMemcpyAsync(o, out.p, sizeof(o), hipMemcpyDeviceToHost, stream));
    sync();
    stats["dense_recession_solves"] += 1.0;
    if ((int)o[0] != 0) { stats["dense_recession_fallbacks"] += 1.0; return false; }
    *unbounded = o[2] < -1e-6;
    return true;
}

// boundroutine  src/model.jl:175-197 with the ray in d_ray
void Engine::boundroutine() {
    for (int nn = 2; nn <= 1023; ++nn) {
        const double s = std::ldexp(1.0, nn);
        LAUNCH_1(k_axpy_scaled, n_lp, stream, n_lp, d_ray.p, s, d_xs.p);
        int64_t nviol = 0;
        double mv = 0.0;
        bool nonfin = false;
        global_sweep(d_xs.p, prm.f_tol, &nviol, &mv, &nonfin);
        if (nonfin) { status = KTN_STATUS_ERROR; return; }
        if (nviol > 0) break;   // !allsat -> stop searching in this direction
    }
}

void Engine::begin() {
    KTN_REQUIRE(loaded, "optimize! before loadproblem!");
    t_start = std::chrono::steady_clock::now();
    begun = true;
    lp_status = KTN_STATUS_OPTIMAL;
    if (status == KTN_STATUS_ERROR) return;
    if (has_inf_bound) {       // presolve: resolve an initially-unbounded LP  model.jl:228-247
        int64_t i = 0;
        bool unb = recession_ray();
        if (unb) std::fprintf(stderr, "WARNING: Automatically bounding unbounded LP\n");     // model.jl:229-231
        while (unb && i < n_lp) {
            if (prm.log_level > 0) {                                                          // model.jl:237
                std::vector<double> ray = d_ray.to_host(stream);
                std::printf("Unbounded ray along: [");
                for (int64_t j = 0; j < n_lp; ++j) std::printf(j ? ",%g" : "%g", ray[(size_t)j]);
                std::printf("]\n");
            }
            boundroutine();
            if (status == KTN_STATUS_ERROR) return;
            unb = recession_ray();
            ++i;
        }
        if (unb) {
            std::fprintf(stderr, "WARNING: Katana could not resolve unbounded LP\n");         // model.jl:244-247
            lp_status = KTN_STATUS_UNBOUNDED; status = KTN_STATUS_UNBOUNDED;
            return;
        }
    }
    if (logging()) { print_header(); std::fflush(stdout); }                                   // model.jl:249-251
}

void Engine::step(int32_t* done) {
    KTN_REQUIRE(begun, "ktn_ecp_step before ktn_optimize_begin");
    *done = 1;
    if (status == KTN_STATUS_ERROR || status == KTN_STATUS_UNBOUNDED) return;
    if (polishing) { polish_step(done); return; }
    if (allsat || iter >= prm.iter_cap) return;          // while !allsat && m.iter < iter_cap  model.jl:257
    iter += 1;
    const double floor_p = prm.lp_tol_floor * prm.f_tol;
    double tol_p = std::min(std::max(prm.lp_tol_scale * last_maxviol, floor_p), prm.lp_tol_cap);
    if (m_nl_global == 0) tol_p = floor_p;      // pure LP: one exact solve, like the reference (row-sharded: NL rows of ALL ranks)
    double tol_g = std::min(std::max(tol_p, prm.lp_gap_floor), prm.lp_gap_cap);
    LpResult R = lp_solve(tol_p, tol_g, 0);
    lp_status = R.status;
    int64_t nviol = 0;
    double mv = 0.0, ex_not_floor = 0.0, ex_obj = objval;
    if (R.status != KTN_STATUS_OPTIMAL) {                                // model.jl:261-263
        // (NL-row blocks: a rank whose LP failed still takes part in this iteration's exchange -- its flag makes every rank leave)
        if (exchanging()) (void)sweep_all(lp_x.p, prm.f_tol, false, R.status, &nviol, &mv, &ex_not_floor, &ex_obj);
        status = R.status;
        return;
    }
    if (prm.vis_data) lp_sols.push_back(lp_x.to_host(stream));          // model.jl:267
    // (NL-row blocks: every rank holds the identical LP, so the purge is identical too)
    if (prm.purge_age > 0 && !prm.vis_data && (!sharded_rows || exchanging()) && M - M_base >= std::max<int64_t>(prm.purge_min_rows, 1)) purge_cuts();
    const double floor_p0 = prm.lp_tol_floor * prm.f_tol, floor_g0 = std::min(std::max(floor_p0, prm.lp_gap_floor), prm.lp_gap_cap);
    ex_not_floor = (R.row_viol <= floor_p0 && R.dres_rel <= floor_g0 &&
                    (R.gap <= floor_g0 || (R.stag_exit && prm.lp_stag_factor > 0.0 && R.gap <= prm.lp_stag_factor * floor_g0))) ? 0.0 : 1.0;
    if (sweep_all(lp_x.p, prm.f_tol, true, R.status, &nviol, &mv, &ex_not_floor, &ex_obj)) return;      // model.jl:268-283 (true: status set, leave)
    last_maxviol = mv;
    stats["last_maxviol"] = mv; stats["last_nviol"] = (double)nviol;
    const bool sat_now = (nviol == 0);
    // inexact-LP rule (DESIGN.md "LP tolerance schedule"): all rows satisfied only counts once
    // the LP itself was solved to the floor tolerance -- by request, or because the last check of a looser solve
    // happens to meet the floor tolerances already (then the re-solve would return this very point)
    const double floor_g = std::min(std::max(floor_p, prm.lp_gap_floor), prm.lp_gap_cap);
    // (... or ended through the stagnation exit with a gap the floor-tolerance solve would accept through that same exit: it
    //  would return after its first two checks with this very point -- 34 iterations and a setup on cfg3)
    (void)floor_g;
    const bool at_floor = ex_not_floor == 0.0;          // (NL-row blocks: the ranks' verdicts agree; taken from the exchange all the same)
    if (sat_now && !R.exact && tol_p > floor_p * (1.0 + 1e-12) && !at_floor) last_maxviol = 0.0;
    else { allsat = sat_now; if (sat_now && tol_p > floor_p * (1.0 + 1e-12) && !R.exact) stats["floor_resolves_skipped"] += 1.0; }
    const double obj = objval;                                           // model.jl:287-289
    const double obj_delta = std::fabs((obj_prev - obj) / obj);
    obj_prev = obj;
    log_max_viol = std::max(log_max_viol, nviol);                        // model.jl:284-285
    log_cuts_lastprnt += last_sweep_cuts;
    if (logging()) {                                                     // model.jl:291-303
        const int64_t r = iter % prm.log_level;
        if (r == 0) {
            if (iter % ((int64_t)prm.log_level * 50) == 0) print_header();
            print_stats(prm.log_level);
            log_cuts_lastprnt = 0;
            log_max_viol = 0;
        } else if (allsat) {
            print_stats(r);                                              // print on last iteration also
        } else if (obj_delta <= prm.obj_eps) {
            print_stats(iter);
        }
    }
    const bool eps_stop = obj_delta <= prm.obj_eps;
    if (eps_stop) { allsat = true; }                                     // model.jl:306-308 (break)
    *done = (allsat || iter >= prm.iter_cap) ? 1 : 0;
    // Terminal refinement of small problems: the reference's simplex vertices end Kelley's method with the last
    // violation far below f_tol (its tests ask the objective to 1e-6 / 1e-7); a first-order LP ends AT f_tol.
    bool refine = false;
    // (row-sharded LP: every rank holds the cut lists of its own NL rows, so the certificate is the all-reduced sum of the ranks'
    //  shares and every decision below is taken from all-reduced numbers: all ranks refine, or none)
    if (allsat && !eps_stop && !polish_done && (!sharded_rows || exchanging()) && prm.polish_max_iter > 0 && (m_nl_global > 0 || exchanging())) {
        if (n_lp <= prm.polish_max_var && !row_sharded()) {
            refine = prm.polish_factor > 0.0 && prm.polish_factor < 1.0;
            polish_phi = prm.polish_factor;
            cert_target = 0.0;
        } else if (prm.obj_cert_tol > 0.0 && lists_ok()) {
            // Larger problems: refine only while the multiplier-weighted residual of the NL rows (the part of  f* - objective  the
            // stop rule leaves open; the LP's own accuracy is its gap tolerance) exceeds half the objective tolerance.
            // (A fused batch -- ktn_set_blocks -- owes the tolerance to EVERY instance: D is then the largest per-instance
            //  certificate in units of that instance's target, and the target is 1.)
            const double target = n_blocks > 0 ? 1.0 : prm.obj_cert_tol * std::max(1.0, std::fabs(ex_obj));
            cert_gap = 0.25 * target / (1.0 + 2.0 * std::fabs(ex_obj));
            const double D = n_blocks > 0 ? certificate_blocks(&cert_gap) : certificate_all_ranks();
            stats["cert_evals"] += 1.0;
            stats["cert_last"] = D;
            if (D > 0.5 * target) {
                refine = true;
                cert_target = target;
                polish_phi = std::min(std::max(0.25 * target / D, 0.05), 0.5);
                stats["cert_refinements"] += 1.0;
            }
        }
    }
    if (refine) {
        polishing = true;
        polish_count = 0;
        best_viol = kInf;
        d_xbest.resize((size_t)n_lp, stream);
        // The point that met the stop rule is a candidate for the answer; its largest violation among ALL rows (the
        // sweep above only measured rows beyond f_tol, i.e. none) comes from a sweep at the polish tolerance below.
        *done = 0;
    }
}

// One pass of the terminal refinement: LP at the polish tolerance, cuts for every row beyond polish_factor * f_tol.
// Ends when no such row is left, or after polish_max_iter passes; the answer is then the point with the smallest violation
// among those that satisfy the reference's stop rule (every row within f_tol).
void Engine::polish_step(int32_t* done) {
    const double f_eff = polish_phi * prm.f_tol;
    int64_t nviol = 0;
    double mv = 0.0;
    bool nonfin = false;
    auto consider = [&](double viol) {           // lp_x / objval hold a point whose largest violation is `viol` (<= f_tol)
        if (viol <= prm.f_tol && viol < best_viol) {
            best_viol = viol;
            best_obj = objval;
            KTN_HIP(hipMemcpyAsync(d_xbest.p, lp_x.p, (size_t)n_lp * sizeof(double), hipMemcpyDeviceToDevice, stream));
        }
    };
    auto finish = [&]() {
        if (best_viol < kInf) {
            KTN_HIP(hipMemcpyAsync(lp_x.p, d_xbest.p, (size_t)n_lp * sizeof(double), hipMemcpyDeviceToDevice, stream));
            objval = best_obj;
            sync();
        }
        polishing = false;
        polish_done = true;
        *done = 1;
    };
    *done = 0;
    double ex0 = 0.0, ex1 = 0.0;
    if (polish_count == 0) {
        // first pass: measure (and cut at) the point that met the stop rule
        if (sweep_all(lp_x.p, f_eff, true, KTN_STATUS_OPTIMAL, &nviol, &mv, &ex0, &ex1)) { polishing = false; *done = 1; return; }
        consider(mv);
        polish_count = 1;
        if (nviol == 0 && cert_target <= 0.0) finish();     // (certificate mode: the LP itself may be what is short -- solve it tighter)
        return;
    }
    if (polish_count > prm.polish_max_iter) { finish(); return; }
    ++polish_count;
    stats["polish_iters"] += 1.0;
    const double tol_p = prm.lp_tol_floor * f_eff;
    // gap tolerance of a refinement solve: scaled with the cut tolerance (small problems); in certificate mode a quarter of the
    // objective tolerance, as a relative gap
    const double tol_g = cert_target > 0.0 ? std::min(std::min(std::max(tol_p, prm.lp_gap_floor), prm.lp_gap_cap), cert_gap)
                                           : std::max(prm.lp_gap_floor * polish_phi, 1e-12);
    LpResult R = lp_solve(tol_p, tol_g, 0);
    if (R.status != KTN_STATUS_OPTIMAL) {                                // keep the point that met the stop rule
        // (NL-row blocks: the exchange of this pass still takes place, so that no rank waits in a collective the others left)
        if (exchanging()) { const int keep = status; (void)sweep_all(lp_x.p, f_eff, false, R.status, &nviol, &mv, &ex0, &ex1); status = keep; }
        finish();
        return;
    }
    { const int keep = status; if (sweep_all(lp_x.p, f_eff, true, KTN_STATUS_OPTIMAL, &nviol, &mv, &ex0, &ex1)) { status = keep; finish(); return; } }
    consider(mv);
    if (cert_target > 0.0 && mv <= prm.f_tol) {                          // certificate mode: done as soon as the bound holds
        double gap_now = 0.0;
        const double D = n_blocks > 0 ? certificate_blocks(&gap_now) : certificate_all_ranks();
        stats["cert_evals"] += 1.0;
        stats["cert_last"] = D;
        if (D <= 0.5 * cert_target) { finish(); return; }
    }
    if (nviol == 0) finish();
}

// max(sum_i lambda_i res_i, 0) over the NL rows at (lp_x, lp_y), with g of the last sweep (kernels.hpp "objective certificate")
// (id_offset: the global id of this handle's first NL row when the cut lists are global -- NL-row blocks over several GPUs;
//  raw: the signed sum, for the caller to add up over the ranks before clamping)
double Engine::objective_certificate(int64_t id_offset, bool raw) {
    if (m_nl <= 0) return 0.0;
    d_cert.resize((size_t)m_nl, stream);
    LAUNCH_1(k_cert_nl, m_nl, stream, m_nl, d_nlrows.p, list_heads() + id_offset, d_cutprev.p, lp_y.p, d_g.p, d_lb.p, d_ub.p, prm.f_tol, d_cert.p);
    hipLaunchKernelGGL(k_sum_partial, dim3(kRedBlocks), dim3(kBlock), 0, stream, m_nl, d_cert.p, partials.p);
    hipLaunchKernelGGL(k_sum_final, dim3(1), dim3(kRedBlocks), 0, stream, partials.p, kRedBlocks, d_scal.p);
    check_launch();
    double D = 0.0;
    KTN_HIP(hipMemcpyAsync(&D, d_scal.p, sizeof(double), hipMemcpyDeviceToHost, stream));
    sync();
    if (raw) return D;
    return (D == D) ? std::max(D, 0.0) : kInf;
}

// Fused batch: the largest per-instance certificate over that instance's own target (k_cert_blocks); *gap_tol = a quarter of the
// smallest per-instance (target / (1 + 2 |objective|)): the relative gap the refinement's LP solves are asked for
double Engine::certificate_blocks(double* gap_tol) {
    if (m_nl <= 0 || n_blocks <= 0) return 0.0;
    d_cert.resize((size_t)m_nl, stream);
    LAUNCH_1(k_cert_nl, m_nl, stream, m_nl, d_nlrows.p, list_heads(), d_cutprev.p, lp_y.p, d_g.p, d_lb.p, d_ub.p, prm.f_tol, d_cert.p);
    d_certblk.resize((size_t)(2 * n_blocks), stream);
    hipLaunchKernelGGL(k_cert_blocks, dim3((unsigned)n_blocks), dim3(kBlock), 0, stream, m_nl, d_nlrows.p, d_rowptr.p, d_col.p, d_cert.p,
                       d_blkcol.p, n_blocks, lp_c.p, lp_x.p, prm.obj_cert_tol, d_certblk.p);
    check_launch();
    std::vector<double> h = d_certblk.to_host(stream);
    double worst = 0.0, gap = kInf;
    for (int64_t b = 0; b < n_blocks; ++b) {
        const double r = h[(size_t)b];
        worst = (r == r) ? std::max(worst, r) : kInf;
        gap = std::min(gap, h[(size_t)(n_blocks + b)]);
    }
    if (gap_tol) *gap_tol = 0.25 * gap;
    return worst;
}

// The certificate of a solve whose NL rows (and their cut lists) are spread over the ranks of a row-sharded LP: the signed shares
// add up, the sum is clamped -- every rank gets the same number
// The stop rule's sweep in whichever form the handle runs: alone (sweep), row-sharded LP (all-reduced counts), or NL-row blocks
// with a replicated LP -- then this rank sweeps its block, the callback moves every rank's new rows into every rank's LP in rank
// order and returns the totals and the maxima of the flags.  Returns true when the loop has to end (status is set).
bool Engine::sweep_all(const double* d_x, double f_cut, bool lp_ok, int lp_stat, int64_t* nviol, double* maxviol, double* extra0, double* extra1) {
    bool nonfin = false;
    if (!exchanging()) {
        global_sweep(d_x, f_cut, nviol, maxviol, &nonfin);
        if (nonfin) { status = KTN_STATUS_ERROR; return true; }
        return false;
    }
    const int64_t m0 = M;
    int64_t nv = 0;
    double mv = 0.0;
    if (lp_ok) sweep(d_x, f_cut, &nv, &mv, &nonfin);
    double sc[5] = {0.0, mv, (lp_ok ? 0.0 : 2.0) + (nonfin ? 1.0 : 0.0), *extra0, *extra1};
    if (exch_cb(exch_user, 0, m0, sc, 5) != 0) throw Error(KTN_E_CALLBACK, "cut-exchange callback failed");
    *nviol = (int64_t)(sc[0] + 0.5);
    *maxviol = sc[1];
    *extra0 = sc[3]; *extra1 = sc[4];
    if (sc[2] >= 2.0) { status = lp_ok ? KTN_STATUS_ERROR : lp_stat; return true; }     // some rank's LP failed
    if (sc[2] >= 1.0) { status = KTN_STATUS_ERROR; return true; }                        // some rank's sweep met a non-finite cut
    return false;
}

double Engine::certificate_all_ranks() {
    if (exchanging()) {                                  // NL-row blocks: the ranks' shares through the callback (sum)
        double D = objective_certificate(exch_lo, true);
        if (!(D == D)) D = kInf;
        if (exch_cb(exch_user, 1, 0, &D, 1) != 0) throw Error(KTN_E_CALLBACK, "cut-exchange callback failed");
        return (D == D) ? std::max(D, 0.0) : kInf;
    }
    if (!row_sharded()) return objective_certificate();
    double D = objective_certificate(0, true);
    if (!(D == D)) D = kInf;
    allreduce_host(&D, 1, 0);
    return (D == D) ? std::max(D, 0.0) : kInf;
}

// Leave the peer-buffer transport: one last barrier (after it no peer kernel of an earlier epoch can still be reading this
// rank's slots, and this rank reads nobody's), then unmap the peers' buffers and free the exposed ones.
void Engine::ipc_release() {
    auto& I = dist.ipc;
    if (I.on) {
        (void)hipSetDevice(device);
        I.timeout_ticks = std::max<long long>(I.timeout_ticks / 10, 1);       // (a peer that is gone already must not hold the teardown up)
        ipc_barrier();
        (void)hipStreamSynchronize(stream);
        I.on = false;
    }
    for (void*& p : I.opened) if (p) { (void)hipIpcCloseMemHandle(p); p = nullptr; }
    if (I.data) { (void)hipFree(I.data); I.data = nullptr; }
    if (I.flags) { (void)hipFree(I.flags); I.flags = nullptr; }
    if (I.h_err) { (void)hipHostFree(I.h_err); I.h_err = nullptr; I.h_err_dev = nullptr; }
}

void Engine::end() {
    soltime = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();   // model.jl:311
    if (status == KTN_STATUS_ERROR || status == KTN_STATUS_UNBOUNDED) return;
    if (lp_status != KTN_STATUS_OPTIMAL) { status = lp_status; return; }
    status = (iter >= prm.iter_cap) ? KTN_STATUS_USERLIMIT : KTN_STATUS_OPTIMAL;                   // model.jl:313-317
}

}  // namespace ktn

// =====================================================================================
// C ABI
// =====================================================================================
using ktn::Engine;

struct ktn_handle_s {
    Engine* eng = nullptr;
    std::string err;
};

#define KTN_TRY(h, ...)                                                      \
    if (!(h) || !(h)->eng) return KTN_E_INVALID;                             \
    try {                                                                    \
        __VA_ARGS__                                                          \
    } catch (const ktn::Error& e) {                                          \
        (h)->err = e.what();                                                 \
        return e.code;                                                       \
    } catch (const std::bad_alloc&) {                                        \
        (h)->err = "host out of memory";                                     \
        return KTN_E_NOMEM;                                                  \
    } catch (const std::exception& e) {                                      \
        (h)->err = e.what();                                                 \
        return KTN_E_INVALID;                                                \
    }

extern "C" {

int ktn_abi_version(void) { return KTN_ABI_VERSION; }
int64_t ktn_sizeof_params(void) { return (int64_t)sizeof(ktn_params); }
int64_t ktn_sizeof_nlp_desc(void) { return (int64_t)sizeof(ktn_nlp_desc); }

void ktn_default_params(ktn_params* p) {
    if (!p) return;
    p->f_tol = 1e-6; p->cut_coef_rng = 1e9; p->log_level = 10; p->iter_cap = 10000; p->obj_eps = -1.0;
    p->vis_data = 0; p->device = -1;
    p->lp_max_iter = 10000000; p->lp_check_every = 64; p->lp_ruiz_iters = 8;
    p->lp_tol_scale = 0.1; p->lp_tol_floor = 0.3; p->lp_tol_cap = 10.0; p->lp_gap_floor = 1e-7; p->lp_gap_cap = 1e-2;
    p->lp_dual_inherit = 1; p->profile = 0;
    p->purge_age = 2; p->purge_margin = 1e-3; p->purge_min_frac = 0.05; p->purge_min_rows = 2000;
    p->lp_dense_after = 5000;
    p->cut_cap_factor = 1.0; p->cut_cap_min = 10000;
    p->lp_stag_factor = 300.0;
    p->lp_ruiz_warm = 0; p->lp_tiled_nnz = 4000000; p->lp_near_check = 7; p->dedupe_eps = 1e-6;
    p->polish_factor = 1e-3; p->polish_max_var = 32; p->polish_max_iter = 30;
    p->epi_shift = 1;
    p->obj_cert_tol = 1e-6;
    p->lp_mid_max_var = 512;
}

int ktn_create(const ktn_params* p, ktn_handle* out) {
    if (!out) return KTN_E_INVALID;
    *out = nullptr;
    ktn_params prm;
    if (p) prm = *p; else ktn_default_params(&prm);
    ktn_handle h = new (std::nothrow) ktn_handle_s();
    if (!h) return KTN_E_NOMEM;
    try {
        h->eng = new Engine(prm);
    } catch (const ktn::Error& e) {
        std::fprintf(stderr, "ktn_create: %s\n", e.what());
        int code = e.code;
        delete h;
        return code;
    } catch (...) {
        delete h;
        return KTN_E_INVALID;
    }
    *out = h;
    return KTN_OK;
}

void ktn_destroy(ktn_handle h) {
    if (!h) return;
    delete h->eng;
    delete h;
}

const char* ktn_last_error(ktn_handle h) { return h ? h->err.c_str() : "invalid handle"; }

int ktn_loadproblem(ktn_handle h, int64_t num_var, int64_t num_constr, const double* l_var, const double* u_var,
                    const double* l_constr, const double* u_constr, int32_t sense, const ktn_nlp_desc* d) {
    KTN_TRY(h, { h->eng->loadproblem(num_var, num_constr, l_var, u_var, l_constr, u_constr, sense, d); return KTN_OK; })
}

int ktn_optimize_begin(ktn_handle h) { KTN_TRY(h, { h->eng->begin(); return KTN_OK; }) }
int ktn_ecp_step(ktn_handle h, int32_t* done) {
    KTN_TRY(h, { int32_t d = 1; h->eng->step(&d); if (done) *done = d; return KTN_OK; })
}
int ktn_optimize_end(ktn_handle h) { KTN_TRY(h, { h->eng->end(); return h->eng->status; }) }

int ktn_optimize(ktn_handle h) {
    KTN_TRY(h, {
        Engine* e = h->eng;
        e->begin();
        int32_t done = (e->status == KTN_STATUS_ERROR || e->status == KTN_STATUS_UNBOUNDED) ? 1 : 0;
        while (!done) e->step(&done);
        e->end();
        return e->status;
    })
}

int ktn_reset(ktn_handle h) { KTN_TRY(h, { KTN_REQUIRE(h->eng->loaded, "reset before loadproblem"); h->eng->reset(); return KTN_OK; }) }

int ktn_get_status(ktn_handle h) { return (h && h->eng) ? h->eng->status : KTN_E_INVALID; }
double ktn_get_objval(ktn_handle h) { return (h && h->eng) ? h->eng->objval : NAN; }
int64_t ktn_get_num_var(ktn_handle h) { return (h && h->eng) ? h->eng->n_lp : -1; }
int ktn_get_solution(ktn_handle h, double* x_out, int64_t n) {
    KTN_TRY(h, {
        KTN_REQUIRE(h->eng->loaded && x_out && n >= h->eng->n_lp, "solution buffer too small");
        h->eng->lp_x.download(x_out, (size_t)h->eng->n_lp, h->eng->stream);
        return KTN_OK;
    })
}
double ktn_get_solvetime(ktn_handle h) { return (h && h->eng) ? h->eng->soltime : NAN; }
int64_t ktn_numiters(ktn_handle h) { return (h && h->eng) ? h->eng->iter : -1; }
int64_t ktn_numcuts(ktn_handle h) { return (h && h->eng) ? h->eng->numcuts : -1; }
int ktn_setwarmstart(ktn_handle h, const double* x, int64_t n) {   // src/model.jl:335: ignored
    (void)x; (void)n;
    return (h && h->eng) ? KTN_OK : KTN_E_INVALID;
}

// ---- separator API
int ktn_sep_precompute(ktn_handle h, const double* xstar, int64_t n) {
    KTN_TRY(h, {
        Engine* e = h->eng;
        KTN_REQUIRE(e->loaded && xstar && n >= e->n_lp, "precompute!: xstar too short");
        e->d_xs.zero(e->stream);
        KTN_HIP(hipMemcpyAsync(e->d_xs.p, xstar, (size_t)std::min<int64_t>(n, e->n0 + 1) * sizeof(double),
                               hipMemcpyHostToDevice, e->stream));
        e->precompute_all(e->d_xs.p);
        e->sync();
        e->have_precompute = true;
        return KTN_OK;
    })
}
int64_t ktn_sep_num_constr(ktn_handle h) { return (h && h->eng) ? (h->eng->obj_linear ? h->eng->m0 : h->eng->m_ext) : -1; }
int64_t ktn_sep_jac_nnz(ktn_handle h) {
    if (!h || !h->eng || !h->eng->loaded) return -1;
    Engine* e = h->eng;
    return e->obj_linear ? e->h_rowptr[e->m0] : e->nnz_ext;
}
int ktn_sep_get_g(ktn_handle h, double* g_out, int64_t m) {
    KTN_TRY(h, {
        KTN_REQUIRE(h->eng->have_precompute && g_out && m <= h->eng->m_ext, "get_g before precompute! or bad size");
        h->eng->d_g.download(g_out, (size_t)m, h->eng->stream);
        return KTN_OK;
    })
}
int ktn_sep_get_jac(ktn_handle h, double* jac_out, int64_t nnz) {
    KTN_TRY(h, {
        KTN_REQUIRE(h->eng->have_precompute && jac_out && nnz <= h->eng->nnz_ext, "get_jac before precompute! or bad size");
        h->eng->d_jac.download(jac_out, (size_t)nnz, h->eng->stream);
        return KTN_OK;
    })
}
int ktn_sep_get_structure(ktn_handle h, int64_t* rowptr_out, int32_t* col_out) {
    KTN_TRY(h, {
        Engine* e = h->eng;
        KTN_REQUIRE(e->loaded, "structure before loadproblem!");
        const int64_t m = e->obj_linear ? e->m0 : e->m_ext;
        std::memcpy(rowptr_out, e->h_rowptr.data(), (size_t)(m + 1) * sizeof(int64_t));
        std::memcpy(col_out, e->h_col.data(), (size_t)e->h_rowptr[m] * sizeof(int32_t));
        return KTN_OK;
    })
}
int ktn_sep_isconstrsat(ktn_handle h, int64_t i, double lb, double ub, double f_tol) {
    KTN_TRY(h, {
        Engine* e = h->eng;
        KTN_REQUIRE(e->have_precompute && i >= 0 && i < e->m_ext, "isconstrsat: bad row or no precompute!");
        double g = 0.0;
        KTN_HIP(hipMemcpyAsync(&g, e->d_g.p + i, 8, hipMemcpyDeviceToHost, e->stream));
        e->sync();
        return ((g >= lb - f_tol) && (g <= ub + f_tol)) ? 1 : 0;
    })
}
int ktn_sep_gencut(ktn_handle h, int64_t i, int32_t* cols, double* coefs, int64_t* nnz, double* constant) {
    KTN_TRY(h, {
        Engine* e = h->eng;
        KTN_REQUIRE(e->have_precompute && i >= 0 && i < e->m_ext && nnz, "gencut: bad row or no precompute!");
        const int64_t beg = e->h_rowptr[i], len = e->h_rowptr[i + 1] - beg;
        KTN_REQUIRE(*nnz >= len, "gencut: output capacity too small");
        std::memcpy(cols, e->h_col.data() + beg, (size_t)len * sizeof(int32_t));
        if (len) KTN_HIP(hipMemcpyAsync(coefs, e->d_jac.p + beg, (size_t)len * 8, hipMemcpyDeviceToHost, e->stream));
        KTN_HIP(hipMemcpyAsync(constant, e->d_bconst.p + i, 8, hipMemcpyDeviceToHost, e->stream));
        e->sync();
        *nnz = len;
        return KTN_OK;
    })
}
int ktn_sep_sweep(ktn_handle h, double f_tol, int64_t* nviol, double* maxviol) {
    KTN_TRY(h, {
        Engine* e = h->eng;
        KTN_REQUIRE(e->have_precompute, "sweep before precompute!");
        int64_t nv = 0; double mv = 0.0; bool nf = false;
        e->sweep(e->d_xs.p, f_tol, &nv, &mv, &nf);
        if (nviol) *nviol = nv;
        if (maxviol) *maxviol = mv;
        if (nf) e->status = KTN_STATUS_ERROR;
        return KTN_OK;
    })
}

// ---- LP introspection
int64_t ktn_lp_num_rows(ktn_handle h) { return (h && h->eng) ? h->eng->M : -1; }
int64_t ktn_lp_nnz(ktn_handle h) { return (h && h->eng) ? h->eng->NNZ : -1; }
int ktn_lp_get_rows(ktn_handle h, int64_t* rowptr, int32_t* col, double* val, double* lo, double* hi) {
    KTN_TRY(h, {
        Engine* e = h->eng;
        KTN_REQUIRE(e->loaded, "no problem loaded");
        e->lp_rowptr.download(rowptr, (size_t)e->M + 1, e->stream);
        e->lp_col.download(col, (size_t)e->NNZ, e->stream);
        e->lp_val.download(val, (size_t)e->NNZ, e->stream);
        e->lp_lo.download(lo, (size_t)e->M, e->stream);
        e->lp_hi.download(hi, (size_t)e->M, e->stream);
        return KTN_OK;
    })
}
int ktn_lp_get_objective(ktn_handle h, double* c_out, int64_t n, double* c0) {
    KTN_TRY(h, {
        Engine* e = h->eng;
        KTN_REQUIRE(e->loaded && n >= e->n_lp, "objective buffer too small");
        e->lp_c.download(c_out, (size_t)e->n_lp, e->stream);
        if (c0) *c0 = e->c0;
        return KTN_OK;
    })
}
int ktn_lp_get_duals(ktn_handle h, double* y_out, int64_t m) {
    KTN_TRY(h, {
        Engine* e = h->eng;
        KTN_REQUIRE(e->loaded && m >= e->M, "dual buffer too small");
        e->lp_y.download(y_out, (size_t)e->M, e->stream);
        return KTN_OK;
    })
}
int ktn_lp_solve(ktn_handle h, double row_tol, double gap_tol, int32_t* lp_status, int64_t* pdhg_iters) {
    KTN_TRY(h, {
        Engine* e = h->eng;
        KTN_REQUIRE(e->loaded, "no problem loaded");
        ktn::LpResult R = e->lp_solve(row_tol, gap_tol, 0);
        if (lp_status) *lp_status = R.status;
        if (pdhg_iters) *pdhg_iters = R.iters;
        return KTN_OK;
    })
}
int ktn_lp_pdhg_raw(ktn_handle h, const double* x0, const double* y0, double eta, double omega, int64_t iters,
                    double* x_out, double* y_out) {
    KTN_TRY(h, {
        KTN_REQUIRE(h->eng->loaded, "no problem loaded");
        h->eng->pdhg_raw(x0, y0, eta, omega, iters, x_out, y_out);
        return KTN_OK;
    })
}
int64_t ktn_num_lp_sols(ktn_handle h) { return (h && h->eng) ? (int64_t)h->eng->lp_sols.size() : -1; }
int ktn_get_lp_sol(ktn_handle h, int64_t k, double* x_out, int64_t n) {
    KTN_TRY(h, {
        Engine* e = h->eng;
        KTN_REQUIRE(k >= 0 && k < (int64_t)e->lp_sols.size() && n >= (int64_t)e->lp_sols[k].size(), "bad lp_sols index");
        std::memcpy(x_out, e->lp_sols[k].data(), e->lp_sols[k].size() * sizeof(double));
        return KTN_OK;
    })
}

double ktn_get_stat(ktn_handle h, const char* name) {
    if (!h || !h->eng || !name) return NAN;
    auto it = h->eng->stats.find(name);
    return it == h->eng->stats.end() ? 0.0 : it->second;
}

// ---- multi-GPU building blocks
int ktn_sweep_lp_point(ktn_handle h, double f_tol, int64_t* nviol, double* maxviol) {
    KTN_TRY(h, {
        Engine* e = h->eng;
        KTN_REQUIRE(e->loaded, "no problem loaded");
        int64_t nv = 0; double mv = 0.0; bool nf = false;
        e->sweep(e->lp_x.p, f_tol, &nv, &mv, &nf);
        if (nviol) *nviol = nv;
        if (maxviol) *maxviol = mv;
        if (nf) e->status = KTN_STATUS_ERROR;
        return KTN_OK;
    })
}
// sum over this handle's NL rows of (multiplier mass of the row's cuts) x (signed residual at the last sweep's point): the
// handle's share of the objective certificate (kernels.hpp "objective certificate") when the NL rows are split over several
// handles -- the caller adds the shares and clamps at zero.  id_offset: global id of the handle's first NL row (global lists).
int ktn_objective_certificate(ktn_handle h, int64_t id_offset, double* sum) {
    KTN_TRY(h, {
        Engine* e = h->eng;
        KTN_REQUIRE(e->loaded && sum && id_offset >= 0, "objective_certificate: bad arguments");
        KTN_REQUIRE(e->lists_ok() && (e->glists ? id_offset + e->m_nl <= e->list_count() : id_offset == 0),
                    "objective_certificate: no cut lists for these rows (ktn_lp_enable_global_lists)");
        *sum = e->objective_certificate(id_offset, true);
        return KTN_OK;
    })
}
int64_t ktn_lp_nnz_from(ktn_handle h, int64_t first_row) {
    if (!h || !h->eng || !h->eng->loaded) return -1;
    Engine* e = h->eng;
    if (first_row < 0 || first_row > e->M) return -1;
    int64_t base = 0;
    if (hipMemcpyAsync(&base, e->lp_rowptr.p + first_row, 8, hipMemcpyDeviceToHost, e->stream) != hipSuccess) return -1;
    if (hipStreamSynchronize(e->stream) != hipSuccess) return -1;
    return e->NNZ - base;
}
int ktn_lp_get_rows_from(ktn_handle h, int64_t first_row, int64_t* rowptr, int32_t* col, double* val, double* lo,
                         double* hi) {
    KTN_TRY(h, {
        Engine* e = h->eng;
        KTN_REQUIRE(e->loaded && first_row >= 0 && first_row <= e->M, "bad first_row");
        const int64_t nr = e->M - first_row;
        KTN_HIP(hipMemcpyAsync(rowptr, e->lp_rowptr.p + first_row, (size_t)(nr + 1) * 8, hipMemcpyDeviceToHost, e->stream));
        e->sync();
        const int64_t base = rowptr[0], nz = e->NNZ - base;
        for (int64_t i = 0; i <= nr; ++i) rowptr[i] -= base;
        if (nz > 0) {
            KTN_HIP(hipMemcpyAsync(col, e->lp_col.p + base, (size_t)nz * 4, hipMemcpyDeviceToHost, e->stream));
            KTN_HIP(hipMemcpyAsync(val, e->lp_val.p + base, (size_t)nz * 8, hipMemcpyDeviceToHost, e->stream));
        }
        if (nr > 0) {
            KTN_HIP(hipMemcpyAsync(lo, e->lp_lo.p + first_row, (size_t)nr * 8, hipMemcpyDeviceToHost, e->stream));
            KTN_HIP(hipMemcpyAsync(hi, e->lp_hi.p + first_row, (size_t)nr * 8, hipMemcpyDeviceToHost, e->stream));
        }
        e->sync();
        return KTN_OK;
    })
}
int ktn_lp_truncate(ktn_handle h, int64_t nrows) {
    KTN_TRY(h, {
        Engine* e = h->eng;
        KTN_REQUIRE(e->loaded && nrows >= e->M_base && nrows <= e->M, "truncate: nrows outside [base rows, current rows]");
        int64_t base = 0;
        KTN_HIP(hipMemcpyAsync(&base, e->lp_rowptr.p + nrows, 8, hipMemcpyDeviceToHost, e->stream));
        e->sync();
        e->numcuts -= (e->M - nrows);
        e->col_removed_rows += e->M - nrows;
        e->M = nrows; e->NNZ = base;
        e->scal_rows = std::min(e->scal_rows, nrows);
        e->sharded_rows = true;
        if (e->ds_valid.n) e->ds_valid.zero(e->stream);
        e->md_valid = false;
        e->lp_rowptr.n = (size_t)nrows + 1; e->lp_col.n = e->lp_val.n = (size_t)base;
        e->lp_lo.n = e->lp_hi.n = e->lp_y.n = (size_t)nrows;
        if (e->d_age.n > (size_t)nrows) e->d_age.n = (size_t)nrows;
        if (e->d_cutprev.n > (size_t)nrows) e->d_cutprev.n = (size_t)nrows;
        e->lp_dirty = true; ++e->lp_version; ++e->lp_epoch;
        return KTN_OK;
    })
}
int ktn_lp_enable_global_lists(ktn_handle h, int64_t nl_total) {
    KTN_TRY(h, {
        Engine* e = h->eng;
        KTN_REQUIRE(e->loaded && nl_total >= 0, "enable_global_lists: no problem loaded");
        KTN_REQUIRE(e->M == e->M_base, "enable_global_lists: call before the first cut");
        e->d_glast.resize((size_t)std::max<int64_t>(nl_total, 1), e->stream);
        KTN_HIP(hipMemsetAsync(e->d_glast.p, 0xFF, e->d_glast.n * sizeof(int64_t), e->stream));
        e->nl_total = nl_total;
        e->glists = true;
        {   // the gathered cuts of ALL ranks land in this LP: reserve for them (the load-time reserve only knew the local shard)
            const int64_t per_sweep = std::min<int64_t>(nl_total, std::max<int64_t>(2 * e->n_lp, 10000));
            int64_t nnz_loc = 0;
            for (auto r : e->h_nlrows) nnz_loc += e->h_rowptr[r + 1] - e->h_rowptr[r];
            const double avg_nl = e->m_nl ? (double)nnz_loc / (double)e->m_nl : 0.0;
            const int64_t rows = e->M + 3 * per_sweep;
            const int64_t nz = e->NNZ + (int64_t)(3.0 * (double)per_sweep * avg_nl);
            if ((double)rows * 200.0 + (double)nz * 60.0 < 64e9) e->reserve_lp(rows, nz);
        }
        e->sync();
        return KTN_OK;
    })
}
int ktn_last_sweep_slots(ktn_handle h, int64_t* slots, int64_t cap, int64_t* count) {
    KTN_TRY(h, {
        Engine* e = h->eng;
        KTN_REQUIRE(e->loaded && count, "last_sweep_slots: bad arguments");
        *count = e->last_sweep_cuts;
        if (slots && e->last_sweep_cuts > 0) {
            KTN_REQUIRE(cap >= e->last_sweep_cuts, "last_sweep_slots: buffer too small");
            std::vector<int32_t> tmp((size_t)e->last_sweep_cuts);
            e->d_violslots.download(tmp.data(), tmp.size(), e->stream);
            for (size_t i = 0; i < tmp.size(); ++i) slots[i] = tmp[i];
        }
        return KTN_OK;
    })
}
int ktn_set_cut_exchange(ktn_handle h, ktn_exchange_cb cb, void* user, int64_t first_nl_id) {
    KTN_TRY(h, {
        Engine* e = h->eng;
        KTN_REQUIRE(e->loaded && first_nl_id >= 0, "ktn_set_cut_exchange: after loadproblem");
        KTN_REQUIRE(cb == nullptr || e->glists, "ktn_set_cut_exchange: enable the global cut lists first (ktn_lp_enable_global_lists)");
        e->exch_cb = cb; e->exch_user = user; e->exch_lo = first_nl_id;
        return KTN_OK;
    })
}
int ktn_lp_purge(ktn_handle h, int64_t* rows_removed) {
    KTN_TRY(h, {
        Engine* e = h->eng;
        KTN_REQUIRE(e->loaded, "no problem loaded");
        const int64_t before = e->M;
        if (e->prm.purge_age > 0 && !e->prm.vis_data && e->M - e->M_base >= std::max<int64_t>(e->prm.purge_min_rows, 1)) e->purge_cuts();
        if (rows_removed) *rows_removed = before - e->M;
        return KTN_OK;
    })
}
// ---- throughput mode: the loaded problem is a block-diagonal batch of independent instances
int ktn_set_blocks(ktn_handle h, int64_t nblocks, const int64_t* col_offsets) {
    KTN_TRY(h, {
        Engine* e = h->eng;
        KTN_REQUIRE(e->loaded && nblocks >= 0, "ktn_set_blocks: after loadproblem");
        if (nblocks == 0) { e->n_blocks = 0; return KTN_OK; }
        KTN_REQUIRE(col_offsets && e->obj_linear, "ktn_set_blocks: needs a linear objective (no shared epigraph variable)");
        KTN_REQUIRE(col_offsets[0] == 0 && col_offsets[nblocks] == e->n_lp, "ktn_set_blocks: offsets must cover the columns");
        e->h_blkcol.assign(col_offsets, col_offsets + nblocks + 1);
        e->blk_nmax = 1;
        for (int64_t b = 0; b < nblocks; ++b) {
            KTN_REQUIRE(col_offsets[b + 1] >= col_offsets[b], "ktn_set_blocks: offsets not monotone");
            e->blk_nmax = std::max<int>(e->blk_nmax, (int)(col_offsets[b + 1] - col_offsets[b]));
        }
        e->d_blkcol.upload(e->h_blkcol, e->stream);
        e->d_blkomega.resize((size_t)nblocks, e->stream);
        e->d_blkomega.zero(e->stream);
        e->sync();
        e->n_blocks = nblocks;
        e->blocks_built_rows = -1;
        return KTN_OK;
    })
}

int ktn_optimize_blocks(ktn_handle h, int32_t cut_capacity) {
    KTN_TRY(h, {
        Engine* e = h->eng;
        KTN_REQUIRE(e->loaded && e->n_blocks > 0, "ktn_optimize_blocks: after ktn_loadproblem and ktn_set_blocks");
        if (e->M != e->M_base || e->iter != 0) e->reset();
        if (e->optimize_blocks_device(cut_capacity > 0 ? cut_capacity : 12)) return e->status;
        // an instance did not finish on the device (or the problem does not qualify): the ordinary loop, from the loaded state
        e->reset();
        e->begin();
        int32_t done = (e->status == KTN_STATUS_ERROR || e->status == KTN_STATUS_UNBOUNDED) ? 1 : 0;
        while (!done) e->step(&done);
        e->end();
        return e->status;
    })
}

// ---- row-sharded LP over several GPUs
int ktn_dist_unique_id(char* out128) {
    if (!out128) return KTN_E_INVALID;
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return KTN_E_HIP;
    static_assert(sizeof(id.internal) == 128, "ncclUniqueId is 128 bytes");
    std::memcpy(out128, id.internal, 128);
    return KTN_OK;
}
// Leave the peer-buffer transport again (before loadproblem): the handle can then be given another transport.  Used by the
// probe-at-init policy of the host side: peer buffers only where ktn_dist_allreduce_probe passed on every rank of THIS box.
int ktn_dist_release_ipc(ktn_handle h) {
    KTN_TRY(h, {
        Engine* e = h->eng;
        KTN_REQUIRE(!e->loaded, "ktn_dist_release_ipc: call before loadproblem");
        e->ipc_release();
        e->dist.ipc = decltype(e->dist.ipc)();
        e->dist.rank = 0; e->dist.world = 1;
        return KTN_OK;
    })
}
int ktn_dist_init_rccl(ktn_handle h, const char* uid128, int32_t rank, int32_t world) {
    KTN_TRY(h, {
        Engine* e = h->eng;
        KTN_REQUIRE(!e->loaded, "ktn_dist_init_*: call before loadproblem");
        KTN_REQUIRE(uid128 && world >= 1 && rank >= 0 && rank < world, "ktn_dist_init_rccl: bad rank / world");
        e->dist.rank = rank; e->dist.world = world;
        e->dist.force = world == 1 && e->dev.force_collective;
        if (world > 1 || e->dist.force) {
            ncclUniqueId id;
            std::memcpy(id.internal, uid128, 128);
            KTN_HIP(hipSetDevice(e->device));
            const ncclResult_t r = ncclCommInitRank(&e->dist.comm, world, id, rank);
            if (r != ncclSuccess) throw ktn::Error(KTN_E_HIP, std::string("ncclCommInitRank: ") + ncclGetErrorString(r));
        }
        return KTN_OK;
    })
}
int ktn_dist_init_callback(ktn_handle h, int32_t rank, int32_t world, ktn_allreduce_cb cb, void* user) {
    KTN_TRY(h, {
        Engine* e = h->eng;
        KTN_REQUIRE(!e->loaded, "ktn_dist_init_*: call before loadproblem");
        KTN_REQUIRE(world >= 1 && rank >= 0 && rank < world && (world == 1 || cb), "ktn_dist_init_callback: bad arguments");
        e->dist.rank = rank; e->dist.world = world; e->dist.cb = cb; e->dist.user = user;
        return KTN_OK;
    })
}

// peer-buffer transport: export this rank's buffers, then map everybody's (kernels.hpp "peer-buffer transport")
int ktn_dist_ipc_export(ktn_handle h, int32_t rank, int32_t world, int64_t capacity, char* out_handles128) {
    KTN_TRY(h, {
        Engine* e = h->eng;
        auto& I = e->dist.ipc;
        static_assert(sizeof(hipIpcMemHandle_t) == KTN_IPC_HANDLE_BYTES, "hipIpcMemHandle_t is 64 bytes");
        KTN_REQUIRE(!e->loaded, "ktn_dist_init_*: call before loadproblem");
        KTN_REQUIRE(out_handles128 && world >= 2 && world <= ktn::kIpcMaxRanks && rank >= 0 && rank < world && capacity >= 64,
                    "ktn_dist_ipc_export: bad arguments (2 <= world <= 8, capacity >= 64)");
        KTN_REQUIRE(I.data == nullptr, "ktn_dist_ipc_export: called twice");
        KTN_HIP(hipSetDevice(e->device));
        I.cap = capacity;
        KTN_HIP(hipMalloc((void**)&I.data, sizeof(double) * 2 * (size_t)capacity));
        // flag words: uncached device memory (every load and store goes to memory: what a peer wrote is what a spin reads)
        if (hipExtMallocWithFlags((void**)&I.flags, 4096, hipDeviceMallocUncached) != hipSuccess) {
            (void)hipGetLastError();
            KTN_HIP(hipExtMallocWithFlags((void**)&I.flags, 4096, hipDeviceMallocFinegrained));
        }
        KTN_HIP(hipMemsetAsync(I.data, 0, sizeof(double) * 2 * (size_t)capacity, e->stream));
        KTN_HIP(hipMemsetAsync(I.flags, 0, 4096, e->stream));
        KTN_HIP(hipHostMalloc((void**)&I.h_err, 64, hipHostMallocMapped));
        *I.h_err = 0;
        KTN_HIP(hipHostGetDevicePointer((void**)&I.h_err_dev, I.h_err, 0));
        e->sync();
        hipIpcMemHandle_t hd, hf;
        KTN_HIP(hipIpcGetMemHandle(&hd, I.data));
        KTN_HIP(hipIpcGetMemHandle(&hf, I.flags));
        std::memcpy(out_handles128, &hd, KTN_IPC_HANDLE_BYTES);
        std::memcpy(out_handles128 + KTN_IPC_HANDLE_BYTES, &hf, KTN_IPC_HANDLE_BYTES);
        e->dist.rank = rank; e->dist.world = world;
        return KTN_OK;
    })
}
int ktn_dist_init_ipc(ktn_handle h, int32_t rank, int32_t world, const char* all_handles) {
    KTN_TRY(h, {
        Engine* e = h->eng;
        auto& I = e->dist.ipc;
        KTN_REQUIRE(!e->loaded, "ktn_dist_init_*: call before loadproblem");
        KTN_REQUIRE(all_handles && I.data && rank == e->dist.rank && world == e->dist.world, "ktn_dist_init_ipc: ktn_dist_ipc_export first, same rank / world");
        KTN_HIP(hipSetDevice(e->device));
        for (int r = 0; r < world; ++r) {
            if (r == rank) { I.P.data[r] = I.data; I.P.flags[r] = I.flags; continue; }
            hipIpcMemHandle_t hd, hf;
            std::memcpy(&hd, all_handles + (size_t)r * 2 * KTN_IPC_HANDLE_BYTES, KTN_IPC_HANDLE_BYTES);
            std::memcpy(&hf, all_handles + (size_t)r * 2 * KTN_IPC_HANDLE_BYTES + KTN_IPC_HANDLE_BYTES, KTN_IPC_HANDLE_BYTES);
            void* pd = nullptr; void* pf = nullptr;
            KTN_HIP(hipIpcOpenMemHandle(&pd, hd, hipIpcMemLazyEnablePeerAccess));
            I.opened[2 * r] = pd;
            KTN_HIP(hipIpcOpenMemHandle(&pf, hf, hipIpcMemLazyEnablePeerAccess));
            I.opened[2 * r + 1] = pf;
            I.P.data[r] = (double*)pd; I.P.flags[r] = (unsigned long long*)pf;
        }
        int khz = 0;
        if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, e->device) != hipSuccess || khz <= 0) { (void)hipGetLastError(); khz = 100000; }
        const double secs = e->dev.ipc_timeout_s;
        I.timeout_ticks = (long long)(secs * 1e3 * (double)khz);
        I.epoch = 0;
        I.on = true;
        return KTN_OK;
    })
}
// One all-reduce of an n-vector through whatever transport the handle has, `reps` times: mean time per call and the largest
// deviation from the sum every rank can compute for itself (in round k rank r contributes k (r + 1) + 1e-3 (j mod 1000)).  Collective call.
int ktn_dist_allreduce_probe(ktn_handle h, int64_t n, int32_t reps, double* usec_per_call, double* max_abs_err) {
    KTN_TRY(h, {
        Engine* e = h->eng;
        KTN_REQUIRE(e->row_sharded() && n >= 1 && reps >= 1 && usec_per_call && max_abs_err, "ktn_dist_allreduce_probe: needs a transport (ktn_dist_init_*)");
        KTN_HIP(hipSetDevice(e->device));
        ktn::DBuf<double> v;
        v.resize((size_t)n, e->stream);
        const int w = e->dist.world;
        std::vector<double> host((size_t)n);
        double worst = 0.0;
        // six rounds with different contents, sum and max alternating: every slot of the peer-buffer transport is reused with
        // new data twice or more (a stale line anywhere on the way shows as a deviation)
        for (int round = 0; round < 6; ++round) {
            const int op = round & 1;
            const double scale = (double)(round + 1);
            hipLaunchKernelGGL(ktn::k_probe_fill, dim3(ktn::ceil_div(n, ktn::kBlock)), dim3(ktn::kBlock), 0, e->stream, n, v.p, scale * (double)(e->dist.rank + 1));
            e->allreduce(v.p, (size_t)n, op);
            KTN_HIP(hipMemcpyAsync(host.data(), v.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, e->stream));
            e->sync();
            e->ipc_check();
            for (int64_t j = 0; j < n; ++j) {
                const double t = 1e-3 * (double)(j % 1000);
                const double want = op ? scale * (double)w + t : scale * 0.5 * (double)w * (double)(w + 1) + (double)w * t;
                worst = std::max(worst, std::fabs(host[(size_t)j] - want));
            }
        }
        hipLaunchKernelGGL(ktn::k_probe_fill, dim3(ktn::ceil_div(n, ktn::kBlock)), dim3(ktn::kBlock), 0, e->stream, n, v.p, 0.0);
        e->allreduce(v.p, (size_t)n, 0);
        e->sync();
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < reps; ++i) e->allreduce(v.p, (size_t)n, 0);
        e->sync();
        e->ipc_check();
        *usec_per_call = 1e6 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / (double)reps;
        *max_abs_err = worst;
        return KTN_OK;
    })
}

int ktn_lp_append_rows(ktn_handle h, int64_t nrows, const int64_t* rowptr, const int32_t* col, const double* val,
                       const double* lo, const double* hi) {
    return ktn_lp_append_rows_nl(h, nrows, rowptr, col, val, lo, hi, nullptr);
}
int ktn_lp_append_rows_nl(ktn_handle h, int64_t nrows, const int64_t* rowptr, const int32_t* col, const double* val,
                          const double* lo, const double* hi, const int64_t* nl_id) {
    KTN_TRY(h, {
        Engine* e = h->eng;
        KTN_REQUIRE(e->loaded && nrows >= 0, "append: bad arguments");
        if (nrows == 0) return KTN_OK;
        const int64_t nz = rowptr[nrows] - rowptr[0];
        std::vector<int64_t> rp((size_t)nrows);
        for (int64_t i = 0; i < nrows; ++i) {
            KTN_REQUIRE(rowptr[i + 1] >= rowptr[i], "append: rowptr not monotone");
            rp[(size_t)i] = rowptr[i + 1] - rowptr[0] + e->NNZ;
            e->max_row_len = std::max(e->max_row_len, rowptr[i + 1] - rowptr[i]);
        }
        for (int64_t k = 0; k < nz; ++k) KTN_REQUIRE(col[rowptr[0] + k] >= 0 && col[rowptr[0] + k] < e->n_lp, "append: column out of range");
        hipStream_t s = e->stream;
        e->lp_rowptr.resize((size_t)(e->M + nrows + 1), s);
        e->lp_lo.resize((size_t)(e->M + nrows), s); e->lp_hi.resize((size_t)(e->M + nrows), s);
        e->lp_y.resize((size_t)(e->M + nrows), s);
        e->lp_col.resize((size_t)(e->NNZ + nz), s); e->lp_val.resize((size_t)(e->NNZ + nz), s);
        KTN_HIP(hipMemcpyAsync(e->lp_rowptr.p + e->M + 1, rp.data(), (size_t)nrows * 8, hipMemcpyHostToDevice, s));
        KTN_HIP(hipMemcpyAsync(e->lp_lo.p + e->M, lo, (size_t)nrows * 8, hipMemcpyHostToDevice, s));
        KTN_HIP(hipMemcpyAsync(e->lp_hi.p + e->M, hi, (size_t)nrows * 8, hipMemcpyHostToDevice, s));
        KTN_HIP(hipMemsetAsync(e->lp_y.p + e->M, 0, (size_t)nrows * 8, s));
        e->d_age.resize((size_t)(e->M + nrows), s); e->d_cutprev.resize((size_t)(e->M + nrows), s);
        KTN_HIP(hipMemsetAsync(e->d_age.p + e->M, 0, (size_t)nrows * sizeof(int32_t), s));
        KTN_HIP(hipMemsetAsync(e->d_cutprev.p + e->M, 0xFF, (size_t)nrows * sizeof(int64_t), s));
        if (nl_id && e->glists) e->append_link(nrows, nl_id);
        if (nz > 0) {
            KTN_HIP(hipMemcpyAsync(e->lp_col.p + e->NNZ, col + rowptr[0], (size_t)nz * 4, hipMemcpyHostToDevice, s));
            KTN_HIP(hipMemcpyAsync(e->lp_val.p + e->NNZ, val + rowptr[0], (size_t)nz * 8, hipMemcpyHostToDevice, s));
        }
        e->sync();
        e->M += nrows; e->NNZ += nz; e->numcuts += nrows;
        e->sharded_rows = true;
        e->lp_dirty = true; ++e->lp_version;
        return KTN_OK;
    })
}

// ---- device-resident cut exchange (replicated LP over several GPUs): the packed block of kernels.hpp "cut blocks for the
// exchange" is written into / read from DEVICE buffers of the caller (torch tensors handed to RCCL's all-gather)
int ktn_lp_pack_rows_dev(ktn_handle h, int64_t first_row, int64_t id_offset, double* dev_out, int64_t cap, int64_t* nrows, int64_t* nnz) {
    KTN_TRY(h, {
        using namespace ktn;
        Engine* e = h->eng;
        KTN_REQUIRE(e->loaded && first_row >= 0 && first_row <= e->M && nrows && nnz, "pack_rows_dev: bad arguments");
        const int64_t nr = e->M - first_row;
        int64_t base = e->NNZ;
        if (nr > 0) {
            KTN_HIP(hipMemcpyAsync(&base, e->lp_rowptr.p + first_row, 8, hipMemcpyDeviceToHost, e->stream));
            e->sync();
        }
        const int64_t nz = e->NNZ - base;
        *nrows = nr; *nnz = nz;
        if (!dev_out) return KTN_OK;                                   // size query
        KTN_REQUIRE(cap >= 4 * nr + 2 * nz, "pack_rows_dev: buffer too small");
        const bool ids = e->last_sweep_cuts == nr && nr > 0;          // the rows of the last sweep carry their NL slot
        // (with global cut lists a row packed WITHOUT its id would silently drop out of dual inheritance, list-based purging and
        //  the objective certificate on every rank)
        KTN_REQUIRE(ids || nr == 0 || !e->glists, "pack_rows_dev: the rows from first_row on are not exactly the cuts of the last sweep");
        LAUNCH_1(k_pack_rows, std::max(nr, nz), e->stream, nr, nz, e->lp_rowptr.p + first_row, e->lp_col.p + base, e->lp_val.p + base,
                 e->lp_lo.p + first_row, e->lp_hi.p + first_row, ids ? (const int32_t*)e->d_violslots.p : (const int32_t*)nullptr,
                 id_offset, dev_out);
        e->check_launch();
        e->sync();                                                     // the caller's stream may read the buffer now
        return KTN_OK;
    })
}
int ktn_lp_append_packed_dev(ktn_handle h, int64_t nrows, int64_t nnz, const double* dev_in) {
    KTN_TRY(h, {
        using namespace ktn;
        Engine* e = h->eng;
        KTN_REQUIRE(e->loaded && nrows >= 0 && nnz >= 0 && (dev_in || nrows == 0), "append_packed_dev: bad arguments");
        if (nrows == 0) return KTN_OK;
        hipStream_t s = e->stream;
        e->lp_rowptr.resize((size_t)(e->M + nrows + 1), s);
        e->lp_lo.resize((size_t)(e->M + nrows), s); e->lp_hi.resize((size_t)(e->M + nrows), s);
        e->lp_y.resize((size_t)(e->M + nrows), s);
        e->lp_col.resize((size_t)(e->NNZ + nnz), s); e->lp_val.resize((size_t)(e->NNZ + nnz), s);
        e->d_age.resize((size_t)(e->M + nrows), s); e->d_cutprev.resize((size_t)(e->M + nrows), s);
        e->d_nlid.resize((size_t)nrows, s);
        KTN_HIP(hipMemsetAsync(e->lp_y.p + e->M, 0, (size_t)nrows * 8, s));
        KTN_HIP(hipMemsetAsync(e->d_age.p + e->M, 0, (size_t)nrows * sizeof(int32_t), s));
        KTN_HIP(hipMemsetAsync(e->d_cutprev.p + e->M, 0xFF, (size_t)nrows * sizeof(int64_t), s));
        KTN_HIP(hipMemsetAsync(e->d_anynf.p + 1, 0, sizeof(int32_t), s));
        LAUNCH_1(k_unpack_rows, std::max(nrows, nnz), s, nrows, nnz, dev_in, e->NNZ, e->n_lp, e->lp_rowptr.p + e->M + 1, e->lp_col.p + e->NNZ,
                 e->lp_val.p + e->NNZ, e->lp_lo.p + e->M, e->lp_hi.p + e->M, e->d_nlid.p, e->d_anynf.p + 1);
        e->check_launch();
        if (e->glists) e->append_link_dev(nrows);
        int32_t bad = 0;
        KTN_HIP(hipMemcpyAsync(&bad, e->d_anynf.p + 1, 4, hipMemcpyDeviceToHost, s));
        e->sync();
        KTN_REQUIRE(bad == 0, "append_packed_dev: malformed block (row pointers not monotone or column out of range)");
        e->M += nrows; e->NNZ += nnz; e->numcuts += nrows;
        e->max_row_len = (int64_t)1 << 62;                             // (row lengths of other ranks' cuts are not known on the host)
        e->sharded_rows = true;
        e->lp_dirty = true; ++e->lp_version;
        return KTN_OK;
    })
}

}  // extern "C"
