"""GPU tier: the separator kernels (precompute! / isconstrsat / gencut / round_coefs / _addcut)
against the CPU oracle on identical inputs, through the C ABI.

Tolerances (FP64): Jacobian entries are single formulas -> <= 4 ulp (device exp/log vs glibc);
g and the cut constant are k-term sums whose association differs (wavefront butterfly vs the
reference's left-to-right loop) -> |err| <= 1e-13 * sum|terms|.
"""
import numpy as np
import pytest

import katana_jl_amd as ktn
from helpers import hip_load_instance, hip_model_from_kat, oracle_evaluator, planted_obj_bound
from kat_util import load_kats
from oracle.evaluators import EpigraphNLPEvaluator, SexprNLPEvaluator
from oracle.katana import KatanaFirstOrderSeparator, linear_oa_cut, round_coefs

pytestmark = pytest.mark.gpu
ULP4 = 4 * np.finfo(float).eps


def _oracle_sep(inst, x):
    d = oracle_evaluator(inst)
    sep = KatanaFirstOrderSeparator()
    sep.initialize(None, inst.n, inst.num_constr, d)
    with np.errstate(all="ignore"):
        sep.precompute(np.asarray(x, dtype=float))
    return sep


@pytest.mark.parametrize("family,k", [("explog", 32), ("quad", 64), ("explog", 5), ("explog", 200)])
def test_precompute_and_gencut_match_oracle(family, k):
    inst = ktn.instances.make_instance(n=3000, m_nl=257, k=k, family=family, seed=11)
    rng = np.random.default_rng(1)
    x = rng.uniform(inst.l_var, inst.u_var)
    m = hip_load_instance(ktn, inst)
    sep = ktn.KatanaHipSeparator(m)
    sep.initialize()
    sep.precompute(x)
    osep = _oracle_sep(inst, x)
    assert np.array_equal(sep.rowptr, osep.csr_ptr) and np.array_equal(sep.col, osep.csr_col)
    jac_o = osep.jac[osep.csr_ind]
    assert np.all(np.abs(sep.jac - jac_o) <= ULP4 * np.abs(jac_o) + 1e-300)
    # g: sum tolerance relative to the magnitude of the terms
    from oracle.evaluators import _atoms
    val, _ = _atoms(inst.kind.astype(np.int64), inst.p0, inst.p1, x[inst.col])
    rows = np.repeat(np.arange(inst.num_constr), np.diff(inst.rowptr))
    mag = np.bincount(rows, weights=np.abs(val), minlength=inst.num_constr) + np.abs(inst.rconst)
    assert np.all(np.abs(sep.g - osep.g) <= 1e-13 * (mag + 1.0))
    for i in list(range(inst.m_lin, inst.m_lin + 5)) + [inst.num_constr - 1, 0]:
        cols, coefs, const = sep.gencut(x, (inst.l_constr[i], inst.u_constr[i]), i)
        cut = linear_oa_cut(osep, x, None, i)                       # src/algorithms.jl:3-18
        assert list(cols) == list(cut.vars)
        assert np.all(np.abs(coefs - cut.coeffs) <= ULP4 * np.abs(cut.coeffs) + 1e-300)
        dotmag = np.sum(np.abs(x[cols] * coefs)) + mag[i] + 1.0
        assert abs(const - cut.constant) <= 1e-13 * dotmag
        assert sep.isconstrsat(i, inst.l_constr[i], inst.u_constr[i], 1e-6) == \
            osep.isconstrsat(i, inst.l_constr[i], inst.u_constr[i], 1e-6)


def test_sweep_appends_exactly_the_oracle_cuts():
    inst = ktn.instances.make_instance(n=2000, m_nl=300, k=24, family="explog", seed=4)
    rng = np.random.default_rng(2)
    x = np.clip(inst.xhat + rng.normal(0, 0.8, inst.n), inst.l_var, inst.u_var)
    m = hip_load_instance(ktn, inst)
    sep = ktn.KatanaHipSeparator(m)
    sep.initialize()
    rp0, *_ = m.lp_rows()
    base_rows = len(rp0) - 1
    assert base_rows == inst.m_lin and m.numcuts() == inst.m_lin      # linear rows counted, model.jl:77,333
    sep.precompute(x)
    nviol, maxviol = sep.sweep(1e-6)
    osep = _oracle_sep(inst, x)
    viol = [i for i in range(inst.m_lin, inst.num_constr)
            if not osep.isconstrsat(i, inst.l_constr[i], inst.u_constr[i], 1e-6)]
    assert nviol == len(viol) > 0 and m.numcuts() == inst.m_lin + nviol
    assert abs(maxviol - max(osep.g[i] - inst.u_constr[i] for i in viol)) <= 1e-12 * (1 + maxviol)
    rowptr, col, val, lo, hi = m.lp_rows()
    assert len(lo) == base_rows + nviol
    for r, i in enumerate(viol):                                       # same order as `for i in nlconstr_ixs`
        cut = linear_oa_cut(osep, x, None, i)
        round_coefs(cut, 1e9)
        a, b = rowptr[base_rows + r], rowptr[base_rows + r + 1]
        assert list(col[a:b]) == list(cut.vars)
        assert np.all(np.abs(val[a:b] - cut.coeffs) <= ULP4 * np.abs(cut.coeffs) + 1e-300)
        scale = np.sum(np.abs(x[cut.vars] * np.asarray(cut.coeffs))) + abs(osep.g[i]) + 1
        assert lo[base_rows + r] == -np.inf
        assert abs(hi[base_rows + r] - (inst.u_constr[i] - cut.constant)) <= 1e-12 * scale   # model.jl:74-75


def _tiny_sep_model(kinds, p0, p1, cols, n, rconst=0.0, lb=-np.inf, ub=0.0, **kw):
    d = ktn.NLPDescription(n, [0, len(cols)], cols, [0], [0], [rconst], kinds, p0, p1,
                           obj_linear=True, obj_col=[0], obj_atom_kind=[0], obj_p0=[1.0], obj_p1=[0.0])
    m = ktn.NonlinearModel(ktn.KatanaSolver(log_level=0, **kw))
    m.loadproblem(n, 1, [-10.0] * n, [10.0] * n, [lb], [ub], "Min", d)
    return m


def test_round_coefs_uses_signed_max_and_keeps_constant():
    # coefficients at x=0: QUAD 2a(x-c): (-2e9*... ) -> [-4e9, 2.0, -6.0]; signed max = 2 -> first zeroed
    m = _tiny_sep_model([1, 1, 1], [1.0, 1.0, 1.0], [2e9, -1.0, 3.0], [0, 1, 2], 3, rconst=-1.0)
    sep = ktn.KatanaHipSeparator(m); sep.initialize()
    x = np.zeros(3)
    sep.precompute(x)
    assert list(sep.jac) == [-4e9, 2.0, -6.0]
    nviol, _ = sep.sweep(1e-6)
    assert nviol == 1
    rowptr, col, val, lo, hi = m.lp_rows()
    assert list(val) == [0.0, 2.0, -6.0]                              # src/model.jl:200-207
    g = 4e18 + 1.0 + 9.0 - 1.0
    assert hi[0] == 0.0 - (g - 0.0)                                   # b NOT recomputed after rounding


def test_nonfinite_coefficient_gives_error_status():
    # -p0*log(x + p1) at x = -p1: derivative -inf  -> _addcut warns and sets :Error (model.jl:69-73)
    m = _tiny_sep_model([3, 0], [1.0, 1.0], [0.0, 0.0], [0, 1], 2)
    sep = ktn.KatanaHipSeparator(m); sep.initialize()
    sep.precompute(np.zeros(2))
    assert not np.isfinite(sep.jac[0])
    sep.sweep(1e-6)
    assert m.status() == "Error" and len(m.lp_rows()[3]) == 0


def test_nan_value_with_finite_gradient_is_violated_and_row_gets_nan_bounds():
    # log of a negative number: g = NaN (violated, separators.jl:120), d/dx = -p0/(x+p1) finite
    m = _tiny_sep_model([3], [1.0], [0.0], [0], 1)
    sep = ktn.KatanaHipSeparator(m); sep.initialize()
    sep.precompute(np.array([-2.0]))
    assert np.isnan(sep.g[0]) and sep.jac[0] == 0.5
    assert not sep.isconstrsat(0, -np.inf, 0.0, 1e-6)
    nviol, _ = sep.sweep(1e-6)
    assert nviol == 1 and m.status() != "Error"
    *_, lo, hi = m.lp_rows()
    assert np.isnan(hi[0])


def test_ragged_and_empty_rows():
    # rows of length 0, 1, 70 (longer than a wavefront) in one description
    rng = np.random.default_rng(3)
    n = 100
    lens = [0, 1, 70, 3]
    rowptr = np.concatenate([[0], np.cumsum(lens)])
    col = np.concatenate([rng.choice(n, L, replace=False) for L in lens]).astype(np.int32)
    nnz = len(col)
    kind = rng.integers(1, 3, nnz).astype(np.uint8)
    p0, p1 = rng.uniform(0.1, 1, nnz), rng.uniform(-0.5, 0.5, nnz)
    rconst = np.array([0.5, -1.0, -2.0, 0.0])
    d = ktn.NLPDescription(n, rowptr, col, [0] * 4, [0] * 4, rconst, kind, p0, p1, obj_linear=True, obj_col=[0],
                           obj_atom_kind=[0], obj_p0=[1.0], obj_p1=[0.0])
    m = ktn.NonlinearModel(ktn.KatanaSolver(log_level=0))
    m.loadproblem(n, 4, [-2.0] * n, [2.0] * n, [-np.inf] * 4, [0.0] * 4, "Min", d)
    sep = ktn.KatanaHipSeparator(m); sep.initialize()
    x = rng.uniform(-2, 2, n)
    sep.precompute(x)
    from oracle.evaluators import SeparableNLPEvaluator
    od = SeparableNLPEvaluator(n, rowptr, col, kind, p0, p1, rconst, [0], [0], [1.0], [0.0])
    g = np.zeros(4); J = np.zeros(nnz)
    od.eval_g(g, x); od.eval_jac_g(J, x)
    assert np.allclose(sep.g, g, rtol=1e-13, atol=1e-13) and np.allclose(sep.jac, J, rtol=1e-14, atol=0)
    assert sep.g[0] == 0.5                                             # empty row: constant only
    nviol, _ = sep.sweep(1e-6)
    assert nviol == int(np.sum(g > 1e-6))


KATS = [m for m in load_kats() if not m["id"].startswith("501_0")] + \
       [m for m in load_kats() if m["id"] in ("501_01_n7", "501_02_n7", "501_02_n20")]


@pytest.mark.parametrize("m", KATS, ids=[m["id"] for m in KATS])
def test_tape_rows_match_oracle_ad_on_reference_models(m):
    """expression tapes (reverse-mode AD on the device) vs the oracle's forward-mode AD, on every
    constraint and on the epigraph row of every reference test model, at random points"""
    n = len(m["vars"])
    M = hip_model_from_kat(ktn, m)
    d = M.build()
    im = ktn.NonlinearModel(M.solver)
    im.loadproblem(n, len(M.cons), M.lb, M.ub, [c[1] for c in M.cons], [c[2] for c in M.cons], M.sense, d)
    sep = ktn.KatanaHipSeparator(im); sep.initialize()
    od = SexprNLPEvaluator(n, m["objective"], [c["expr"] for c in m["constraints"]],
                           [c["linear"] for c in m["constraints"]], m["objective_linear"])
    lifted = not m["objective_linear"]
    rng = np.random.default_rng(7)
    for trial in range(3):
        x = rng.uniform(0.2, 1.9, size=n + (1 if lifted else 0))
        sep.precompute(x)
        ev = EpigraphNLPEvaluator(od, n + 1, len(m["constraints"]) + 1) if lifted else od
        mrows = len(m["constraints"]) + (1 if lifted else 0)
        g = np.zeros(mrows)
        with np.errstate(all="ignore"):
            ev.eval_g(g, x)
        assert np.allclose(sep.g[:mrows], g, rtol=1e-13, atol=1e-13, equal_nan=True), (sep.g, g)
        # Jacobian: compare per row through dense rows (the oracle's epigraph row is dense, ours structural)
        rows_o, cols_o = ev.jac_structure()
        J = np.zeros(len(rows_o))
        with np.errstate(all="ignore"):
            ev.eval_jac_g(J, x)
        dense_o = np.zeros((mrows, len(x)))
        for r, c, v in zip(rows_o, cols_o, J):
            dense_o[r, c] += v
        dense_h = np.zeros((mrows, len(x)))
        for r in range(mrows):
            for e in range(sep.rowptr[r], sep.rowptr[r + 1]):
                dense_h[r, sep.col[e]] += sep.jac[e]
        assert np.allclose(dense_h, dense_o, rtol=1e-13, atol=1e-13, equal_nan=True)


@pytest.mark.parametrize("family", ["quad", "explog"])
def test_column_blocked_sweep_of_long_rows_matches_the_gather_sweep(family, monkeypatch):
    """rows with hundreds of entries: the block-major / LDS-staged evaluation (k_sep_eval_blk + k_sep_combine) finds the
    same violated rows and emits the same cuts as k_sep_eval (sums in a different, fixed order: 1e-13 relative)"""
    inst = ktn.instances.make_instance(n=20000, m_nl=120, k=700, family=family, seed=3)   # 3 column blocks of 8192
    x = np.clip(inst.xhat * 1.5 + 0.7, inst.l_var, inst.u_var)
    out = []
    for blocked in ("0", "1"):
        monkeypatch.setenv("KTN_SWEEP_BLOCKED", blocked)
        m = hip_load_instance(ktn, inst)
        sep = ktn.KatanaHipSeparator(m); sep.initialize()
        m0 = m.lp_num_rows()
        sep.precompute(x)
        nv, mv = sep.sweep(1e-6)
        out.append((nv, mv, m.lp_rows_from(m0)))
    (nv0, mv0, r0), (nv1, mv1, r1) = out
    assert nv0 == nv1 and nv0 > 0 and abs(mv0 - mv1) <= 1e-12 * max(1.0, abs(mv0))
    assert np.array_equal(r0[0], r1[0]) and np.array_equal(r0[1], r1[1])
    assert np.array_equal(r0[2], r1[2])                                     # coefficients: same arithmetic per entry
    for a, b in ((r0[3], r1[3]), (r0[4], r1[4])):
        a = np.asarray(a); b = np.asarray(b)
        fin = np.isfinite(a)
        assert np.array_equal(fin, np.isfinite(b))
        assert np.allclose(a[fin], b[fin], rtol=1e-12, atol=1e-12)


def test_ecp_solve_through_the_column_blocked_sweep(monkeypatch):
    """a whole ECP solve on long rows (the blocked sweep is the default there) agrees with the gather sweep"""
    inst = ktn.instances.make_instance(n=20000, m_nl=60, k=400, family="quad", seed=5)
    res = []
    for blocked in ("0", None):
        if blocked is None:
            monkeypatch.delenv("KTN_SWEEP_BLOCKED", raising=False)
        else:
            monkeypatch.setenv("KTN_SWEEP_BLOCKED", blocked)
        m = hip_load_instance(ktn, inst)
        assert m.optimize() == "Optimal"
        res.append((m.getobjval(), m.numiters(), m.numcuts()))
    assert abs(res[0][0] - res[1][0]) <= 1e-7 * max(1.0, abs(res[0][0]))
    assert abs(res[0][0] - inst.opt_obj) <= planted_obj_bound(inst)


def _load_traces():
    import json, os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kat_traces.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("tr", _load_traces(), ids=lambda t: t["id"])
def test_replay_of_committed_oracle_traces(tr):
    """tests/golden/kat_traces.json (made by tests/golden/make_trace_fixture.py): at every recorded LP point the device
    sweep finds the same violated rows and appends the same cuts (coefficients after round_coefs, row bounds)"""
    k = [m for m in load_kats() if m["id"] == tr["id"]][0]
    n = len(k["vars"])
    M = hip_model_from_kat(ktn, k)
    d = M.build()
    im = ktn.NonlinearModel(M.solver)
    im.loadproblem(n, len(M.cons), M.lb, M.ub, [c[1] for c in M.cons], [c[2] for c in M.cons], M.sense, d)
    sep = ktn.KatanaHipSeparator(im); sep.initialize()
    for it in tr["iterations"]:
        x = np.asarray(it["x"])
        m0 = im.lp_num_rows()
        sep.precompute(x)
        g_want = np.asarray(it["g"])
        g_got = sep.g[np.asarray(it["nl_rows"])]
        assert np.allclose(g_got, g_want, rtol=1e-13, atol=1e-13), (g_got, g_want)
        nviol, _ = sep.sweep(1e-6)
        assert nviol == len(it["cuts"])
        rowptr, col, val, lo, hi = im.lp_rows_from(m0)
        for r, cut in enumerate(it["cuts"]):
            dense = np.zeros(len(x))
            for e in range(rowptr[r], rowptr[r + 1]):
                dense[col[e]] += val[e]
            assert np.allclose(dense, cut["coefs"], rtol=1e-12, atol=1e-12), (tr["id"], cut["row"], dense, cut["coefs"])
            for got, want in ((lo[r], cut["lo"]), (hi[r], cut["hi"])):
                assert (got == want) if not np.isfinite(want) else abs(got - want) <= 1e-12 * max(1.0, abs(want))


def test_full_size_hbm_resident_sweep_blocked_equals_row_kernel(monkeypatch):
    """SURVEY.md section 8d's HBM-resident variant at FULL size (n = 1e5, 1e4 rows x 2048 entries = 2.05e7 Jacobian
    entries): the column-blocked sweep and the row kernel flag the same rows and emit the same cuts"""
    inst = ktn.instances.make_config("cfg3_hbm", seed=0, vertex=False)
    x = np.random.default_rng(5).uniform(inst.l_var, inst.u_var)
    out = []
    for blocked in ("0", "1"):
        monkeypatch.setenv("KTN_SWEEP_BLOCKED", blocked)
        m = hip_load_instance(ktn, inst, cut_cap_factor=0.0)
        sep = ktn.KatanaHipSeparator(m); sep.initialize()
        m0 = m.lp_num_rows()
        sep.precompute(x)
        nv, mv = sep.sweep(1e-6)
        rp, col, val, lo, hi = m.lp_rows_from(m0)
        out.append((nv, mv, np.asarray(rp), np.asarray(col), np.asarray(val), np.asarray(hi)))
        del m, sep
    a, b = out
    assert a[0] == b[0] and a[0] > 0
    assert abs(a[1] - b[1]) <= 1e-11 * max(1.0, abs(a[1]))
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4])
    assert np.allclose(a[5], b[5], rtol=1e-11, atol=1e-11)


def test_very_long_rows_go_through_the_long_row_kernel():
    """Round 4: a separable row beyond 8 192 entries (a linear objective stored as a row, the dense epigraph row of a nonlinear
    objective, a constraint over most variables) is evaluated by k_sep_eval_long -- one workgroup per row -- instead of one lane
    group walking it; every row kernel skips it, k_emit takes its derivatives from the materialised Jacobian.  Same tolerances
    against the oracle as the short rows: precompute!, gencut, and the cuts a sweep appends."""
    from oracle.evaluators import _atoms
    inst = ktn.instances.make_instance(n=12000, m_nl=4, k=9000, family="explog", seed=5, m_lin=50)
    rng = np.random.default_rng(2)
    x = rng.uniform(inst.l_var, inst.u_var)
    m = hip_load_instance(ktn, inst, purge_age=0, cut_cap_factor=0.0)
    assert m.stat("sep_long_rows") >= 4                               # the four NL rows (and the objective row, 1e4 entries)
    sep = ktn.KatanaHipSeparator(m)
    sep.initialize()
    sep.precompute(x)
    osep = _oracle_sep(inst, x)
    jac_o = osep.jac[osep.csr_ind]
    assert np.all(np.abs(sep.jac - jac_o) <= ULP4 * np.abs(jac_o) + 1e-300)
    val, _ = _atoms(inst.kind.astype(np.int64), inst.p0, inst.p1, x[inst.col])
    rows = np.repeat(np.arange(inst.num_constr), np.diff(inst.rowptr))
    mag = np.bincount(rows, weights=np.abs(val), minlength=inst.num_constr) + np.abs(inst.rconst)
    assert np.all(np.abs(sep.g - osep.g) <= 1e-13 * (mag + 1.0))
    nl = np.arange(inst.m_lin, inst.num_constr)
    viol_o = [i for i in nl if not osep.isconstrsat(i, inst.l_constr[i], inst.u_constr[i], 1e-6)]
    assert len(viol_o) >= 1
    m0 = m.lp_num_rows()
    nv, mv = sep.sweep(1e-6)
    assert nv == len(viol_o)
    rowptr, col, valr, lo, hi = m.lp_rows()
    for kcut, i in enumerate(viol_o):
        cut = linear_oa_cut(osep, x, None, i)
        round_coefs(cut, 1e9)
        beg, end = rowptr[m0 + kcut], rowptr[m0 + kcut + 1]
        assert list(col[beg:end]) == list(cut.vars)
        assert np.all(np.abs(valr[beg:end] - cut.coeffs) <= ULP4 * np.abs(cut.coeffs) + 1e-300)
        dotmag = np.sum(np.abs(x[cut.vars] * np.asarray(cut.coeffs))) + mag[i] + 1.0
        assert abs(hi[m0 + kcut] - (inst.u_constr[i] - cut.constant)) <= 1e-13 * dotmag


def test_batch_blocked_sweep_equals_the_row_kernel_and_is_bitwise_reproducible(monkeypatch):
    """Round 4: k_sep_sweep_batch (2 048 NL slots per workgroup, x* staged through LDS in 64 KB blocks, kind-uniform entry-parallel
    chunks, deterministic run sums) against the row kernel on the same instance and point, both in one process (the switch is read
    per handle): the same violated rows and cuts (structure exactly, values to the sum tolerance), and two sweeps of the batched
    kernel give the same bits.  (By default the kernel is used from 7.9e5 NL rows on -- the full-size cfg4 test runs through it.)"""
    inst = ktn.instances.make_instance(n=20000, m_nl=30000, k=32, family="explog", seed=8, m_lin=200)
    x = np.clip(inst.xhat + 0.3, inst.l_var, inst.u_var)
    out = {}
    for label, env in (("row", "0"), ("batch", "1")):
        monkeypatch.setenv("KTN_SWEEP_BATCHED", env)
        m = hip_load_instance(ktn, inst, purge_age=0, cut_cap_factor=0.0)
        assert m.stat("sweep_batched") == (1.0 if label == "batch" else 0.0)
        sep = ktn.KatanaHipSeparator(m); sep.initialize()
        sep.precompute(x)
        res = []
        for rep in range(2):
            nv, mv = sep.sweep(1e-6)
            res.append((nv, mv) + tuple(a.copy() for a in m.lp_rows()))
            m.reset(); sep.precompute(x)
        out[label] = res
    monkeypatch.delenv("KTN_SWEEP_BATCHED")
    (nv_r, mv_r, rp_r, col_r, val_r, lo_r, hi_r), (nv_b, mv_b, rp_b, col_b, val_b, lo_b, hi_b) = out["row"][0], out["batch"][0]
    assert nv_r == nv_b > 100 and abs(mv_r - mv_b) <= 1e-12 * (1.0 + abs(mv_r))
    assert np.array_equal(rp_r, rp_b) and np.array_equal(col_r, col_b)
    assert np.array_equal(val_r, val_b)                               # the coefficients come from k_emit either way
    fin = np.isfinite(hi_r)
    assert np.array_equal(fin, np.isfinite(hi_b)) and np.all(np.abs(hi_r[fin] - hi_b[fin]) <= 1e-12 * (1.0 + np.abs(hi_r[fin])))
    again = out["batch"][1]
    assert again[0] == nv_b and again[1] == mv_b and all(np.array_equal(p, q) for p, q in zip(again[2:], out["batch"][0][2:]))
