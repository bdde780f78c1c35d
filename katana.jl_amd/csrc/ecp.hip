// ecp.hip -- optimize! (src/model.jl:219-319): presolve, the cutting-plane loop, the terminal refinement, the device-side batch loop  (struct Engine: engine.hpp)
#include "engine.hpp"
#include "launch.hpp"
#include "kernels.hpp"
#include "batch_lp.hpp"
#include "batch_ecp.hpp"

namespace ktn {

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

void Engine::end() {
    soltime = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();   // model.jl:311
    if (status == KTN_STATUS_ERROR || status == KTN_STATUS_UNBOUNDED) return;
    if (lp_status != KTN_STATUS_OPTIMAL) { status = lp_status; return; }
    status = (iter >= prm.iter_cap) ? KTN_STATUS_USERLIMIT : KTN_STATUS_OPTIMAL;                   // model.jl:313-317
}

}  // namespace ktn
