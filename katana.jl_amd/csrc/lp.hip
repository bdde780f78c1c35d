// lp.hip -- solve(m.linear_model) (src/model.jl:259; GLPK in the reference): restarted reflected-Halpern PDHG and the exact small / mid-size LPs  (struct Engine: engine.hpp)
#include "engine.hpp"
#include "launch.hpp"
#include "kernels.hpp"
#include "dense_lp.hpp"
#include "mid_lp.hpp"
#include "batch_lp.hpp"

namespace ktn {

// --------------------------------------------------------------- reductions ---
double Engine::dev_dot(int64_t n, const double* a, const double* b) {
    hipLaunchKernelGGL(k_dot_partial, dim3(kRedBlocks), dim3(kBlock), 0, stream, n, a, b, partials.p);
    hipLaunchKernelGGL(k_sum_final, dim3(1), dim3(kRedBlocks), 0, stream, partials.p, kRedBlocks, chkout.p + 2 * kChkQ);
    double v = 0.0;
    KTN_HIP(hipMemcpyAsync(&v, chkout.p + 2 * kChkQ, sizeof(double), hipMemcpyDeviceToHost, stream));
    sync();
    return v;
}

double Engine::dev_finite_sq(int64_t n, const double* a) {
    hipLaunchKernelGGL(k_finite_sq_partial, dim3(kRedBlocks), dim3(kBlock), 0, stream, n, a, partials.p);
    hipLaunchKernelGGL(k_sum_final, dim3(1), dim3(kRedBlocks), 0, stream, partials.p, kRedBlocks, chkout.p + 2 * kChkQ);
    double v = 0.0;
    KTN_HIP(hipMemcpyAsync(&v, chkout.p + 2 * kChkQ, sizeof(double), hipMemcpyDeviceToHost, stream));
    sync();
    return v;
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
                // statistic + update of BOTH sides in one launch, into new arrays that are swapped in (9 launches; round 1: 33, round 2: 22)
                dr2.resize((size_t)M, stream); dc2.resize((size_t)n_lp, stream);
                const int br = ceil_div(M * gr, kBlock), bc = ceil_div(n_lp * gc, kBlock);
                hipLaunchKernelGGL(k_scale_stat_upd_both, dim3((unsigned)(br + bc)), dim3(kBlock), 0, stream, M, lp_rowptr.p, lp_col.p, Wval(), n_lp,
                                   c_ptr.p, c_row.p, c_val.p, dr.p, dc.p, mode, dr2.p, dc2.p, cap_c, gr, gc, br);
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
    // (long columns: the mirror's scaled values are gathered entry-parallel from the row copy instead of walked column by column)
    if (n_longc > 0 || M == 0) {
        LAUNCH_G(gr, k_scale_vals, M, stream, M, lp_rowptr.p, lp_col.p, Wval(), dr.p, dc.p, r_sval.p);
        if (n_longc > 0) LAUNCH_1(k_csc_vals, NNZ, stream, NNZ, c_perm.p, r_sval.p, c_sval.p);
        else LAUNCH_G(gc, k_scale_vals, n_lp, stream, n_lp, c_ptr.p, c_row.p, c_val.p, dc.p, dr.p, c_sval.p);
    } else {                                            // both copies in one launch
        const int br = ceil_div(M * gr, kBlock), bc = ceil_div(n_lp * gc, kBlock);
        hipLaunchKernelGGL(k_scale_vals_both, dim3((unsigned)(br + bc)), dim3(kBlock), 0, stream, M, lp_rowptr.p, lp_col.p, Wval(), n_lp, c_ptr.p, c_row.p,
                           c_val.p, dr.p, dc.p, r_sval.p, c_sval.p, gr, gc, br);
    }
    check_launch();
}

void Engine::launch_tiled(const TiledBuf& T, int64_t n_out, int64_t n_in, const double* in, hipEvent_t e0) {
    hipExtLaunchKernelGGL(k_spmv_tiled, dim3((unsigned)T.grid), dim3(kTileThreads), 0, stream, e0, nullptr, 0, n_out, n_in, T.tiles,
                          T.view(), in, tpart.p);
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
        LAUNCH_GB(grp_cols, k_pdhg_x, false, n, stream, n, AT, yh.p, xh.p, x0h.p, xth.p, (double*)nullptr, ch.p, lh.p, uh.p, tau, w, rho);
    }
}

// Check iteration: the PDHG step without update (xt, yt stored) and the KKT / fixed-point sums.  The row side rides on the
// y-step (k_pdhg_y_chk gathers xt and x anyway); the column side needs A'yt and is one G-lanes-per-column pass.
bool Engine::launch_check(const SpMat& A, const SpMat& AT, double tau, double sigma, double w_next, double rho) {
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
        return false;
    }
    // the plain CSR form on one GPU: the two step kernels of the check also leave the Halpern update in xnext / ynext
    const bool spec = w_next >= 0.0 && m > 0 && !row_sharded() && n_long == 0 && n_longc == 0;
    if (spec) {
        xnext.resize((size_t)n, stream); ynext.resize((size_t)m, stream);
        LAUNCH_GB(grp_cols, k_pdhg_x, false, n, stream, n, AT, yh.p, xh.p, x0h.p, xth.p, xnext.p, ch.p, lh.p, uh.p, tau, w_next, rho);
    } else {
        launch_x(AT, tau, 0.0, 1.0, false, nullptr, nullptr);
    }
    if (m > 0) {
        LAUNCH_G(grp_rows, k_pdhg_y_chk, m, stream, m, A, xth.p, xh.p, yh.p, y0h.p, yth.p, loh.p, hih.p, dr.p, sigma, thr, prow,
                 spec ? ynext.p : (double*)nullptr, w_next, rho);
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
    return spec;
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
    LAUNCH_1(k_prep_both, std::max(n, m), stream,
             PrepCols{n, Wc(), lp_l.p, lp_u.p, dc.p, lp_x.p, (mode == 1 ? box.p : (const double*)nullptr), sgn, mode, ch.p, lh.p, uh.p, xh.p},
             PrepRows{m, Wlo(), Whi(), dr.p, lp_y.p, mode, loh.p, hih.p, yh.p});
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
        double* nrm = (chk_pinned() ? h_chk_dev : chkout.p) + 2 * kChkQ + 1;
        auto dot_dev = [&](const double* a, double* out) {
            hipLaunchKernelGGL(k_dot_partial, dim3(kRedBlocks), dim3(kBlock), 0, stream, n, a, a, partials.p);
            hipLaunchKernelGGL(k_sum_final, dim3(1), dim3(kRedBlocks), 0, stream, partials.p, kRedBlocks, out);
        };
        auto normalize_into = [&](const double* a, double* out) {        // out = a / ||a||: partial sums, then k_normalize_sum adds them up itself
            hipLaunchKernelGGL(k_dot_partial, dim3(kRedBlocks), dim3(kBlock), 0, stream, n, a, a, partials.p);
            hipLaunchKernelGGL(k_normalize_sum, dim3(ceil_div(n, (int64_t)kRedBlocks)), dim3(kRedBlocks), 0, stream, n, a, partials.p, out);
        };
        // the start vector -- hashed, normalised -- depends on n alone: made once per handle (and size), read by the first pass in place
        if (power_v0_n != n) {
            power_v.resize(n, stream); power_v0.resize(n, stream);
            LAUNCH_1(k_hash_fill, n, stream, n, power_v.p);
            normalize_into(power_v.p, power_v0.p);
            power_v0_n = n;
        }
        const int passes_env = dev.power_passes;
        const int iters = passes_env > 0 ? passes_env : 8;
        // The iterate is re-normalised only every fourth pass (and before the last, whose ||A'A v|| with ||v|| = 1 is the
        // estimate): with ||A^||_2 <= 1 after the Pock-Chambolle pass the un-normalised vector only shrinks slowly, and the
        // Rayleigh quotient does not depend on the scale -- 6 instead of 20 (dot, final sum, normalise) triples per LP solve.
        for (int it = 0; it < iters; ++it) {
            const double* vin = (it == 0) ? power_v0.p : pv.p;
            if (n_long > 0) {
                LAUNCH_G(grp_rows, k_spmv_skip, m, stream, m, A, vin, pw.p, kLongRow);
                hipLaunchKernelGGL(k_spmv_long, dim3((unsigned)n_long), dim3(1024), 0, stream, d_longrows.p, A, vin, pw.p);
            } else {
                LAUNCH_G(grp_rows, k_spmv, m, stream, m, A, vin, pw.p);
            }
            const bool norm_now = (it % 4 == 3) || it >= iters - 2;
            if (norm_now) {
                spmv_cols(AT, pw.p, xbar.p);
                allreduce(xbar.p, (size_t)n, 0);            // row-sharded: A'A v = sum over the ranks of A_r'(A_r v)
                if (it == iters - 1) dot_dev(xbar.p, nrm);  // the last pass: ||A'A v||^2 with ||v|| = 1 is the estimate; the vector is not needed again
                else normalize_into(xbar.p, pv.p);
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
    // (one-GPU solves: the final sums are written straight into the pinned, device-mapped block the check sums use -- no copy kernel,
    //  the host reads them after the synchronisation)
    const bool pin_slots = chk_pinned();
    double* slots = (pin_slots ? h_chk_dev : chkout.p) + 2 * kChkQ;           // [1] power, [2] fro2, [3] nc2, [4] |lo|^2, [5] |hi|^2
    // slots 2..5 (||A^||_F^2, ||c^||^2, finite ||lo^||^2, ||hi^||^2) by one fused pair of launches
    hipLaunchKernelGGL(k_setup_norms_partial, dim3(kRedBlocks), dim3(kBlock), 0, stream, NNZ, r_sval.p, n, ch.p, m, loh.p, hih.p, partials.p);
    hipLaunchKernelGGL(k_sum_final_multi, dim3(4), dim3(kRedBlocks), 0, stream, partials.p, kRedBlocks, slots + 2);
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
    if (!pin_slots) KTN_HIP(hipMemcpyAsync(hs, slots, sizeof(hs), hipMemcpyDeviceToHost, stream));
    double epi_b = 0.0;                                 // b_ref of the working form: the objective constant it carries
    if (w_shift) KTN_HIP(hipMemcpyAsync(&epi_b, epi_scal.p, sizeof(double), hipMemcpyDeviceToHost, stream));
    sync();
    if (pin_slots) std::memcpy(hs, h_chk + 2 * kChkQ, sizeof(hs));
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
        LAUNCH_1(k_pack_both, std::max(n, m), stream, n, c_ptr.p, ch.p, lh.p, uh.p, xh.p, x0h.p, d_crec.p, d_cbl.p,
                 m, lp_rowptr.p, loh.p, hih.p, yh.p, y0h.p, d_rrec.p);
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
        const bool spec_next = launch_check(A, AT, tau, sigma, (double)(k + 1) / (double)(k + 2), rho);
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
        if (spec_next) {                                  // the check kernels left the update in xnext / ynext
            xh.swap(xnext); yh.swap(ynext);
        } else {
            const double w = (double)(k + 1) / (double)(k + 2);
            LAUNCH_1(k_halpern2, std::max(n, m), stream, n, m, xh.p, xth.p, x0h.p, yh.p, yth.p, y0h.p, w, rho);
        }
        ++k; ++it;
        plain_next = true;       // (after a restart k == 0 and the next pass is a check again: it needs r0)
    }
    R.iters = it;
    // un-scale the last PDHG point (xt, yt)
    if (mode == 0) {
        LAUNCH_1(k_unscale2, std::max(n, m), stream, n, xth.p, dc.p, lp_x.p, m, yth.p, dr.p, lp_y.p);
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

}  // namespace ktn
