// pool.hip -- the cut pool: LP rows as the sweeps append them (src/model.jl:199-207), their column mirror, purging  (struct Engine: engine.hpp)
#include "engine.hpp"
#include "launch.hpp"
#include "kernels.hpp"
#include "mid_lp.hpp"

namespace ktn {

LpRows Engine::lp_view() {
    LpRows L;
    L.rowptr = lp_rowptr.p; L.col = lp_col.p; L.val = lp_val.p; L.lo = lp_lo.p; L.hi = lp_hi.p; L.y = lp_y.p;
    return L;
}

void Engine::append_link(int64_t nrows, const int64_t* nl_id_host) {       // rows [M, M + nrows) just appended from the host
    d_nlid.upload(nl_id_host, (size_t)nrows, stream);
    LAUNCH_1(k_append_link, nrows, stream, nrows, M, d_nlid.p, nl_total, d_glast.p, d_cutprev.p, lp_y.p, (int)prm.lp_dual_inherit);
    check_launch();
}

void Engine::append_link_dev(int64_t nrows) {                              // the same with the ids already in d_nlid (device-resident exchange)
    LAUNCH_1(k_append_link, nrows, stream, nrows, M, d_nlid.p, nl_total, d_glast.p, d_cutprev.p, lp_y.p, (int)prm.lp_dual_inherit);
    check_launch();
}

void Engine::pack_rows_launch(int64_t nr, int64_t nz, int64_t first_row, int64_t base, bool ids, int64_t id_offset, double* dev_out) {
    LAUNCH_1(k_pack_rows, std::max(nr, nz), stream, nr, nz, lp_rowptr.p + first_row, lp_col.p + base, lp_val.p + base,
             lp_lo.p + first_row, lp_hi.p + first_row, ids ? (const int32_t*)d_violslots.p : (const int32_t*)nullptr,
             id_offset, dev_out);
    check_launch();
}
void Engine::unpack_rows_launch(int64_t nrows, int64_t nnz, const double* dev_in) {
    LAUNCH_1(k_unpack_rows, std::max(nrows, nnz), stream, nrows, nnz, dev_in, NNZ, n_lp, lp_rowptr.p + M + 1, lp_col.p + NNZ,
             lp_val.p + NNZ, lp_lo.p + M, lp_hi.p + M, d_nlid.p, d_anynf.p + 1);
    check_launch();
}

// Capacity for the cut pool up front: growing a buffer is hipMalloc + copy + hipFree (which synchronises the device),
// and on a large instance the pool passes through a dozen sizes in the first iterations (cfg4: 2.5 of the 4.9 s of a
// cold solve).  HBM is plentiful (288 GB): reserve for three sweeps' worth of cuts.
void Engine::reserve_lp(int64_t rows, int64_t nnz) {
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

void Engine::epi_dot(const double* x) {             // epi_scal[1] = a_ref'x  (a_ref is zero at the epigraph variable)
    hipLaunchKernelGGL(k_dot_partial, dim3(kRedBlocks), dim3(kBlock), 0, stream, n_lp, epi_ref.p, x, partials.p);
    hipLaunchKernelGGL(k_sum_final, dim3(1), dim3(kRedBlocks), 0, stream, partials.p, kRedBlocks, epi_scal.p + 1);
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

}  // namespace ktn
