// abi.hip -- the C ABI of include/katana_hip.h over struct Engine (engine.hpp).  Launches no kernel and includes no kernel header:
// a change to a kernel does not rebuild this layer.
#include "engine.hpp"

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
            e->probe_fill(n, v.p, scale * (double)(e->dist.rank + 1));
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
        e->probe_fill(n, v.p, 0.0);
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
        e->pack_rows_launch(nr, nz, first_row, base, ids, id_offset, dev_out);
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
        e->unpack_rows_launch(nrows, nnz, dev_in);
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
