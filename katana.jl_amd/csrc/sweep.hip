// sweep.hip -- loadproblem! (src/model.jl:81-173) and the separator sweep (src/separators.jl:85-135, src/model.jl:272-283)  (struct Engine: engine.hpp)
#include "engine.hpp"
#include "launch.hpp"
#include "kernels.hpp"
#include "batch_lp.hpp"
#include "batch_ecp.hpp"

namespace ktn {

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

NlpDev Engine::nlp_view() {
    NlpDev P;
    P.rowptr = d_rowptr.p; P.col = d_col.p; P.colk = d_colk.p; P.pp = d_pp.p;
    P.rconst = d_rconst.p; P.row_kind = d_rowkind.p; P.pad_zero = d_padzero.p; P.lb = d_lb.p; P.ub = d_ub.p;
    P.node_ptr = d_nodeptr.p; P.node_op = d_nodeop.p; P.node_a = d_nodea.p; P.node_b = d_nodeb.p;
    P.node_c = d_nodec.p; P.node_val = d_nodeval.p; P.node_adj = d_nodeadj.p;
    return P;
}

SweepOut Engine::sweep_view() {
    SweepOut O;
    O.g = d_g.p; O.jac = d_jac.p; O.bconst = d_bconst.p; O.maxc = d_maxc.p; O.nonfin = d_nonfin.p;
    O.flag = d_flag.p; O.cnt = d_cnt.p; O.maxviol = d_scal.p; O.any_nonfin = d_anynf.p;
    return O;
}

// ================================================================ separator =====
// precompute! for every row of the extended structure (jac materialised)
void Engine::precompute_all(const double* d_x) {
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
void Engine::sweep(const double* d_x, double f_tol, int64_t* nviol_out, double* maxviol_out, bool* nonfinite_out) {
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
void Engine::global_sweep(const double* d_x, double f_tol, int64_t* nviol, double* maxviol, bool* nonfinite) {
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

}  // namespace ktn
