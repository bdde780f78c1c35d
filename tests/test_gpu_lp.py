"""GPU tier: the LP kernels.  (1) raw reflected-Halpern PDHG iterations against a dense numpy
statement of the same recurrences; (2) the full GPU LP solver against HiGHS on the same LP."""
import numpy as np
import pytest

import katana_jl_amd as ktn
from helpers import hip_load_instance

pytestmark = pytest.mark.gpu


def _dense_lp(m):
    rowptr, col, val, lo, hi = m.lp_rows()
    c, c0 = m.lp_objective()
    A = np.zeros((len(lo), m.num_var))
    for i in range(len(lo)):
        for e in range(rowptr[i], rowptr[i + 1]):
            A[i, col[e]] += val[e]
    return A, c, c0, lo, hi


def _halpern_numpy(A, c, l, u, lo, hi, x, y, eta, omega, iters):
    tau, sigma = eta / omega, eta * omega
    x0, y0 = x.copy(), y.copy()
    for k in range(iters):
        xt = np.clip(x - tau * (c - A.T @ y), l, u)
        v = y - sigma * (A @ (2 * xt - x))
        yt = v + sigma * np.clip(-v / sigma, lo, hi)
        w = (k + 1) / (k + 2)
        x = w * (2 * xt - x) + (1 - w) * x0
        y = w * (2 * yt - y) + (1 - w) * y0
    return x, y


@pytest.mark.parametrize("iters", [1, 2, 50])
def test_raw_pdhg_iterations_match_numpy(iters):
    inst = ktn.instances.make_instance(n=300, m_nl=40, k=8, family="quad", seed=9)
    m = hip_load_instance(ktn, inst)
    sep = ktn.KatanaHipSeparator(m); sep.initialize()
    sep.precompute(np.clip(inst.xhat + 0.7, inst.l_var, inst.u_var))
    assert sep.sweep(1e-6)[0] > 0                                  # LP = linear rows + a block of cuts
    A, c, c0, lo, hi = _dense_lp(m)
    rng = np.random.default_rng(0)
    x0 = rng.uniform(inst.l_var, inst.u_var)
    y0 = rng.normal(size=len(lo)) * (np.isfinite(lo) | np.isfinite(hi))
    y0 = np.where(np.isfinite(lo), y0, np.minimum(y0, 0.0))        # sign-feasible duals
    eta, omega = 0.05, 1.7
    xg, yg = m.lp_pdhg_raw(x0, y0, eta, omega, iters)
    xn, yn = _halpern_numpy(A, c, inst.l_var, inst.u_var, lo, hi, x0, y0, eta, omega, iters)
    assert np.max(np.abs(xg - xn)) <= 1e-11 * (1 + np.max(np.abs(xn)))
    assert np.max(np.abs(yg - yn)) <= 1e-11 * (1 + np.max(np.abs(yn)))


@pytest.mark.parametrize("seed", [0, 1])
def test_gpu_lp_matches_highs(seed):
    from oracle.lp import LinearModel
    inst = ktn.instances.make_instance(n=1500, m_nl=100, k=16, family="explog", seed=seed)
    m = hip_load_instance(ktn, inst)
    sep = ktn.KatanaHipSeparator(m); sep.initialize()
    sep.precompute(inst.u_var * 0.999)
    sep.sweep(1e-6)
    status, iters = m.lp_solve(row_tol=1e-8, gap_tol=1e-8)
    assert status == "Optimal" and iters > 0
    rowptr, col, val, lo, hi = m.lp_rows()
    c, c0 = m.lp_objective()
    lm = LinearModel()
    lm.add_variables(inst.l_var, inst.u_var)
    lm.set_objective("Min", np.arange(inst.n), c, c0)
    lm.add_rows(rowptr, col, val, lo, hi)
    assert lm.solve() == "Optimal"
    x = m.getsolution()
    assert abs(m.getobjval() - lm.getobjval()) <= 1e-6 * max(1.0, abs(lm.getobjval()))
    A, *_ = _dense_lp(m)
    ax = A @ x
    assert np.max(np.maximum(ax - hi, lo - ax)) <= 1e-7
    assert np.all(x >= inst.l_var - 1e-12) and np.all(x <= inst.u_var + 1e-12)
    # dual sign feasibility: y_i > 0 only where the row has a finite lower side
    y = m.lp_duals()
    assert np.all(y[~np.isfinite(lo)] <= 1e-12)


def test_long_column_lp_steps_match_numpy_and_the_solve_matches_highs():
    """A variable that EVERY nonlinear row contains (min-max / epigraph-style models: g_i(x) - t <= r_i) gets an entry from every
    cut: its column of the mirror is as long as the cut pool.  Beyond 2 048 entries the column side of the LP runs in its vector
    form -- A'y by lane groups plus one 1 024-thread workgroup per long column, then the element-wise primal step -- the scaling
    statistics and the power iteration likewise, and the mirror is rebuilt by the radix sort (the append-only merge orders a
    column's new entries by insertion).  Raw iterations against the dense numpy recurrences to 1e-11, the solve against HiGHS."""
    from oracle.lp import LinearModel
    inst = ktn.instances.make_instance(n=300, m_nl=2600, k=6, family="explog+t", seed=4, m_lin=100)
    m = hip_load_instance(ktn, inst, purge_age=0, cut_cap_factor=0.0)
    sep = ktn.KatanaHipSeparator(m); sep.initialize()
    x = np.clip(inst.xhat + 0.7, inst.l_var, inst.u_var)
    x[-1] = -5.0                                                   # t far below: every row violated (a cut is valid wherever it is taken)
    sep.precompute(x)
    assert sep.sweep(1e-6)[0] > 2048
    A, c, c0, lo, hi = _dense_lp(m)
    assert np.count_nonzero(A[:, -1]) > 2048                       # the shared column
    rng = np.random.default_rng(0)
    x0 = rng.uniform(inst.l_var, inst.u_var)
    y0 = rng.normal(size=len(lo)) * (np.isfinite(lo) | np.isfinite(hi))
    y0 = np.where(np.isfinite(lo), y0, np.minimum(y0, 0.0))
    eta, omega = 0.02, 1.3
    xg, yg = m.lp_pdhg_raw(x0, y0, eta, omega, 20)
    assert m.stat("lp_long_cols") == 1
    xn, yn = _halpern_numpy(A, c, inst.l_var, inst.u_var, lo, hi, x0, y0, eta, omega, 20)
    assert np.max(np.abs(xg - xn)) <= 1e-11 * (1 + np.max(np.abs(xn)))
    assert np.max(np.abs(yg - yn)) <= 1e-11 * (1 + np.max(np.abs(yn)))
    status, iters = m.lp_solve(row_tol=1e-8, gap_tol=1e-8)
    assert status == "Optimal" and iters > 0 and m.stat("lp_long_cols") == 1
    rowptr, col, val, lo, hi = m.lp_rows()
    lm = LinearModel()
    lm.add_variables(inst.l_var, inst.u_var)
    lm.set_objective("Min", np.arange(inst.n), c, c0)
    lm.add_rows(rowptr, col, val, lo, hi)
    assert lm.solve() == "Optimal"
    assert abs(m.getobjval() - lm.getobjval()) <= 1e-6 * max(1.0, abs(lm.getobjval()))
    ax = A @ m.getsolution()
    assert np.max(np.maximum(ax - hi, lo - ax)) <= 1e-7


def test_lp_with_no_rows_goes_to_the_bounds():
    d = ktn.NLPDescription(3, [0], [], [], [], [], [], [], [], obj_linear=True, obj_col=[0, 1, 2],
                           obj_atom_kind=[0, 0, 0], obj_p0=[1.0, -2.0, 0.0], obj_p1=[0, 0, 0], obj_const=0.5)
    m = ktn.NonlinearModel(ktn.KatanaSolver(log_level=0))
    m.loadproblem(3, 0, [-1.0, -1.0, -1.0], [2.0, 3.0, 4.0], [], [], "Min", d)
    assert m.optimize() == "Optimal"
    x = m.getsolution()
    assert x[0] == -1.0 and x[1] == 3.0 and abs(m.getobjval() - (-1.0 - 6.0 + 0.5)) < 1e-12
    assert m.numiters() == 1 and m.numcuts() == 0


def test_infeasible_lp_status_is_passed_through():
    """LP status other than :Optimal is returned as is (src/model.jl:261-263); the GPU LP certifies primal
    infeasibility with a Farkas multiplier"""
    x, y = ktn.var(0), ktn.var(1)
    M = ktn.Model(solver=ktn.KatanaSolver(log_level=0))
    M.variable(-5, 5); M.variable(-5, 5)
    M.objective("Min", x + y)
    M.constraint(x + y >= 3.0)
    M.constraint(x + 2 * y <= 1.0)
    M.constraint(x - 0 * y <= 0.5)          # x <= 0.5  => y >= 2.5 => x + 2y >= 5.5 > 1: empty
    M.constraint(x ** 2 + y ** 2 <= 100.0)
    assert M.solve() == "Infeasible"
    # and the oracle's LP says the same
    from oracle.lp import LinearModel
    lm = LinearModel(); lm.add_variables([-5, -5], [5, 5]); lm.set_objective("Min", [0, 1], [1.0, 1.0])
    lm.add_rows([0, 2, 4, 5], [0, 1, 0, 1, 0], [1, 1, 1, 2, 1], [3, -np.inf, -np.inf], [np.inf, 1, 0.5])
    assert lm.solve() == "Infeasible"


# ---- exact small-LP kernel (csrc/dense_lp.hpp); lp_dense_after < 0 sends every LP of <= 32 columns to it ----------

def _random_small_lp(rng, n):
    """bounded polytope around the origin: 6n inequality rows, a few range / equality rows, mixed variable bounds"""
    m_in, m_rg, m_eq = 6 * n, 3, min(2, n - 1)
    A = rng.normal(size=(m_in + m_rg + m_eq, n))
    A[rng.random(A.shape) < 0.3] = 0.0
    A[:, 0] += 0.1                                    # no empty row
    xf = rng.normal(size=n) * 0.3                     # a feasible point
    ax = A @ xf
    lo = np.full(len(A), -np.inf); hi = ax + rng.uniform(0.5, 2.0, len(A))
    lo[m_in:m_in + m_rg] = ax[m_in:m_in + m_rg] - rng.uniform(0.5, 2.0, m_rg)
    lo[m_in + m_rg:] = hi[m_in + m_rg:] = ax[m_in + m_rg:]
    l = np.full(n, -np.inf); u = np.full(n, np.inf)
    kind = rng.integers(0, 4, n)                      # 0 free, 1 lower only, 2 upper only, 3 boxed
    l[(kind == 1) | (kind == 3)] = xf[(kind == 1) | (kind == 3)] - 1.0
    u[(kind == 2) | (kind == 3)] = xf[(kind == 2) | (kind == 3)] + 1.0
    return A, lo, hi, l, u, rng.normal(size=n)


@pytest.mark.parametrize("n,sense,seed", [(1, "Min", 0), (3, "Min", 1), (3, "Max", 2), (12, "Min", 3), (32, "Max", 4), (32, "Min", 5)])
def test_exact_small_lp_kernel_matches_highs(n, sense, seed):
    from oracle.lp import LinearModel
    rng = np.random.default_rng(seed)
    A, lo, hi, l, u, c = _random_small_lp(rng, n)
    M = ktn.Model(solver=ktn.KatanaSolver(log_level=0, lp_dense_after=-1))
    V = [ktn.var(j) for j in range(n)]
    for j in range(n):
        M.variable(l[j], u[j])
    M.objective(sense, sum((float(c[j]) * V[j] for j in range(1, n)), float(c[0]) * V[0]))
    for i in range(len(A)):
        M.constraint((sum((float(A[i, j]) * V[j] for j in range(1, n)), float(A[i, 0]) * V[0]), lo[i], hi[i]))
    lm = LinearModel()
    lm.add_variables(l, u)
    lm.set_objective(sense, np.arange(n), c, 0.0)
    rp = np.arange(0, (len(A) + 1) * n, n)
    lm.add_rows(rp, np.tile(np.arange(n), len(A)), A.ravel(), lo, hi)
    want = lm.solve()
    got = M.solve()
    im = M.internal_model
    assert got == want
    if want != "Optimal":
        return
    # (pdhg_iters may be > 0: with unboxed variables the recession-cone LP is checked by the first-order path first)
    assert im.stat("dense_lp_solves") == 1 and im.stat("dense_lp_fallbacks") == 0
    assert abs(M.getobjectivevalue() - lm.getobjval()) <= 1e-9 * max(1.0, abs(lm.getobjval()))
    x = np.asarray(M.getvalue())
    ax = A @ x
    assert np.max(np.maximum(ax - hi, lo - ax)) <= 1e-8 and np.all(x >= l - 1e-9) and np.all(x <= u + 1e-9)
    # multipliers: dual objective equals the primal one (strong duality with exact arithmetic up to rounding)
    y = im.lp_duals()
    r = (c if sense == "Min" else -c) - A.T @ y
    dobj = np.sum(np.where(y > 0, lo, hi)[y != 0] * y[y != 0]) + np.sum(np.where(r > 0, l, u)[np.abs(r) > 1e-9] * r[np.abs(r) > 1e-9])
    pobj = float((c if sense == "Min" else -c) @ x)
    assert abs(dobj - pobj) <= 1e-7 * max(1.0, abs(pobj))


def test_exact_kernel_reports_infeasible_small_lp():
    x, y = ktn.var(0), ktn.var(1)
    M = ktn.Model(solver=ktn.KatanaSolver(log_level=0, lp_dense_after=-1))
    M.variable(-5, 5); M.variable(-5, 5)
    M.objective("Min", x + y)
    M.constraint(x + y >= 3.0)
    M.constraint(x + 2 * y <= 1.0)
    M.constraint(x - 0 * y <= 0.5)
    assert M.solve() == "Infeasible"
    assert M.internal_model.stat("dense_lp_fallbacks") == 0


@pytest.mark.parametrize("kid", ["101_01", "501_01_n4"])
def test_exact_lp_reproduces_the_simplex_trajectory(kid):
    """with exact vertex LP solutions the ECP iteration count equals the oracle's (HiGHS dual simplex) on models
    whose LPs have a unique optimal vertex"""
    from helpers import hip_model_from_kat, oracle_solve_kat
    from kat_util import load_kats
    k = [m for m in load_kats() if m["id"] == kid][0]
    om = oracle_solve_kat(k)
    M = hip_model_from_kat(ktn, k, lp_dense_after=-1, polish_factor=0.0)      # the reference's loop, nothing after the stop rule
    assert M.solve() == om.status == "Optimal"
    assert M.internal_model.numiters() == om.numiters()
    assert abs(M.getobjectivevalue() - om.getobjval()) <= 1e-7


def test_stalled_first_order_lp_hands_over_to_the_exact_kernel():
    """test/2d.jl 107_01-like flat optimum: the first-order LP exceeds a (small) iteration budget, the exact kernel
    finishes the solve and the answer is the reference's"""
    from helpers import hip_model_from_kat
    from kat_util import isapprox, load_kats
    k = [m for m in load_kats() if m["id"] == "108_01"][0]
    M = hip_model_from_kat(ktn, k, lp_dense_after=2000)
    assert M.solve() == "Optimal"
    im = M.internal_model
    assert im.stat("lp_stalls") >= 1 and im.stat("dense_lp_solves") >= 1
    assert isapprox(M.getobjectivevalue(), k["expect"]["obj"], 1e-6, 1e-6)


def test_tiled_spmv_steps_match_the_csr_steps():
    """LPs beyond the caches run k_pdhg_x / k_pdhg_y from tiled copies of the matrix (input staged through LDS, kernels.hpp
    "tiled SpMV"): the same PDHG iterates as the CSR kernels up to the summation order, and the same ECP answer"""
    import numpy as np
    from helpers import hip_load_instance, max_nl_violation, planted_obj_bound
    inst = ktn.instances.make_instance(n=20000, m_nl=60000, k=24, family="explog", seed=5)
    out = {}
    for name, thr in (("csr", 0), ("tiled", 1)):
        m = hip_load_instance(ktn, inst, lp_tiled_nnz=thr, cut_cap_factor=0.0, purge_age=0)
        sep = ktn.KatanaHipSeparator(m); sep.initialize()
        sep.precompute(np.clip(inst.xhat + 2.0, inst.l_var, inst.u_var))
        nviol, _ = sep.sweep(1e-6)
        assert m.lp_num_rows() > 2 * 8192
        rng = np.random.default_rng(1)
        x0, y0 = rng.uniform(-1, 1, m.num_var), np.abs(rng.standard_normal(m.lp_num_rows()))
        _, _, _, lo, hi = m.lp_rows()
        y0 = np.where(np.isfinite(lo), y0, -y0)                      # sign-feasible duals
        out[name] = m.lp_pdhg_raw(x0, y0, 2e-3, 1.0, 40) + (m.stat("lp_tiled_builds"),)
    assert out["csr"][2] == 0 and out["tiled"][2] == 1
    for a, b in zip(out["csr"][:2], out["tiled"][:2]):
        assert np.max(np.abs(a - b)) <= 1e-11 * max(1.0, np.max(np.abs(a)))
    # whole solve through the tiled path
    m = hip_load_instance(ktn, inst, lp_tiled_nnz=1, cut_cap_factor=0.0, purge_age=0)
    assert m.optimize() == "Optimal" and m.stat("lp_tiled_builds") >= 1
    assert abs(m.getobjval() - inst.opt_obj) <= planted_obj_bound(inst)
    assert max_nl_violation(inst, m.getsolution()) <= 1e-6 * (1 + 1e-6)


def test_run_based_tiled_build_equals_the_general_one_and_gives_way_to_it(monkeypatch):
    """the tiled copies are built by run-based kernels (an output's entries of one input block are one run when they come in
    ascending input order, as the LP's rows and the mirror's columns do): the same copy, bit for bit, as the general kernels
    with their per-cell cursors; a row appended by the host with its columns in DESCENDING order makes the run-based build give
    up and the general one take over -- same iterates as the CSR kernels either way"""
    import numpy as np
    from helpers import hip_load_instance
    inst = ktn.instances.make_instance(n=20000, m_nl=60000, k=24, family="explog", seed=6)

    def iterates(general, extra_row, tiled=1):
        if general:
            monkeypatch.setenv("KTN_TILED_GENERAL_BUILD", "1")
        else:
            monkeypatch.delenv("KTN_TILED_GENERAL_BUILD", raising=False)
        m = hip_load_instance(ktn, inst, lp_tiled_nnz=tiled, cut_cap_factor=0.0, purge_age=0)
        sep = ktn.KatanaHipSeparator(m); sep.initialize()
        sep.precompute(np.clip(inst.xhat + 2.0, inst.l_var, inst.u_var))
        sep.sweep(1e-6)
        if extra_row:
            cols = np.array([15000, 9000, 17, 3], dtype=np.int32)           # descending: crosses input blocks backwards
            m.lp_append_rows(np.array([0, 4], dtype=np.int64), cols, np.array([1.0, -2.0, 0.5, 3.0]), np.array([-np.inf]), np.array([4.0]))
        rng = np.random.default_rng(1)
        x0, y0 = rng.uniform(-1, 1, m.num_var), np.abs(rng.standard_normal(m.lp_num_rows()))
        _, _, _, lo, hi = m.lp_rows()
        y0 = np.where(np.isfinite(lo), y0, -y0)
        x, y = m.lp_pdhg_raw(x0, y0, 2e-3, 1.0, 25)
        return x, y, m.stat("lp_tiled_builds"), m.stat("lp_tiled_general_builds")

    xa, ya, ba, ga = iterates(False, False)
    xb, yb, bb, gb = iterates(True, False)
    assert ba == bb == 1 and ga == 0 and gb == 0
    assert np.array_equal(xa, xb) and np.array_equal(ya, yb)
    xc, yc, bc, gc = iterates(False, True)
    xd, yd, *_ = iterates(False, True, tiled=0)                             # the CSR kernels on the same LP
    assert bc == 1 and gc >= 1
    assert np.max(np.abs(xc - xd)) <= 1e-11 * max(1.0, np.max(np.abs(xd)))
    assert np.max(np.abs(yc - yd)) <= 1e-11 * max(1.0, np.max(np.abs(yd)))


def test_append_only_mirror_update_gives_the_sorted_mirror_bit_for_bit(monkeypatch):
    """the column mirror of the growing LP: between purges it is extended by a merge of the appended rows (kernels.hpp
    "append-only update of the mirror") instead of a sort of all non-zeros.  Both give every column its entries in row order, so
    the two solves run the same arithmetic: identical objective, PDHG iteration count and x -- with and without purging (which
    forces a fresh sort), and with a nonlinear objective (dense epigraph cuts whose working values change every solve)."""
    import katana_jl_amd as ktn
    from helpers import hip_load_instance
    for kw, solver in ((dict(n=3000, m_nl=300, k=16, family="explog", seed=4), {}),
                       (dict(n=3000, m_nl=300, k=16, family="explog", seed=4), dict(purge_min_rows=50)),
                       (dict(n=2500, m_nl=200, k=16, family="quad", seed=2, objective="quad"), {})):
        inst = ktn.instances.make_instance(**kw)
        res = []
        for off in (False, True):
            if off:
                monkeypatch.setenv("KTN_NO_CSC_MERGE", "1")
            else:
                monkeypatch.delenv("KTN_NO_CSC_MERGE", raising=False)
            m = hip_load_instance(ktn, inst, **solver)
            assert m.optimize() == "Optimal"
            res.append((m.getobjval(), m.stat("pdhg_iters"), m.numiters(), m.getsolution(), m.stat("lp_csc_merges"), m.stat("lp_csc_sorts")))
        a, b = res
        assert a[0] == b[0] and a[1] == b[1] and a[2] == b[2] and np.array_equal(a[3], b[3])
        assert a[4] >= a[2] - 3 and b[4] == 0 and b[5] >= b[2] - 3       # merges in the default build, sorts only when asked
        if not solver:
            assert a[5] <= 2                                              # (a sort only for the first mirror; the pool is never purged at this size)


@pytest.mark.parametrize("first", [0, 10, 20, 30, 40])
def test_lp_battery_against_highs_and_the_planted_optimum(first):
    """50 random sparse LPs, 1e3 ... 1e4 columns, degenerate vertices, rows scaled over six decades and
    dual-degenerate free columns included (instances.lp_battery_case), through ktn_lp_solve.  Every case: `:Optimal`, objective
    within 1e-7 relative of the planted primal-dual optimum, rows and bounds feasible.  The 15 cases below 1 700 columns also
    against HiGHS's committed status and objective (tests/golden/lp_battery_highs.json; on the larger ones HiGHS -- simplex
    and interior point -- does not finish within minutes here, while the engine needs 300 - 2 700 iterations, 3 - 20 ms)."""
    import json, os
    fx = {c["case"]: c for c in json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lp_battery_highs.json")))["cases"]}
    for i in range(first, first + 10):
        inst = ktn.instances.make_lp(**ktn.instances.lp_battery_case(i))
        m = hip_load_instance(ktn, inst)
        st, iters = m.lp_solve(row_tol=1e-8, gap_tol=1e-8)
        assert st == "Optimal" and iters > 0 and m.stat("mid_lp_solves") == 0, (i, st)
        scale = max(1.0, abs(inst.opt_obj))
        assert abs(m.getobjval() - inst.opt_obj) <= 1e-7 * scale, (i, m.getobjval(), inst.opt_obj)
        if i in fx:
            assert fx[i]["status"] == st and abs(m.getobjval() - fx[i]["objective"]) <= 1e-7 * scale, (i, m.getobjval(), fx[i])
        x = m.getsolution()
        rows = np.repeat(np.arange(inst.num_constr), np.diff(inst.rowptr))
        ax = np.bincount(rows, weights=inst.p0 * x[inst.col], minlength=inst.num_constr)
        assert np.max(ax - inst.u_constr) <= 1e-7                                            # (the solve's row tolerance was 1e-8, absolute)
        assert np.all(x >= inst.l_var - 1e-12) and np.all(x <= inst.u_var + 1e-12)


def test_exact_mid_size_lp_solver_matches_highs_when_forced():
    """csrc/mid_lp.hpp on its own: lp_dense_after < 0 sends an LP of 33 .. lp_mid_max_var columns straight to the dual
    active-set solver (basis inverse in device memory, cost perturbation against dual degeneracy).  Objective against HiGHS."""
    from oracle.lp import LinearModel
    for kw in (dict(n=120, m=150, seed=3), dict(n=300, m=260, seed=4, degenerate_frac=0.3), dict(n=200, m=400, seed=5, free_frac=0.3)):
        inst = ktn.instances.make_lp(**kw)
        m = hip_load_instance(ktn, inst, lp_dense_after=-1)
        st, pivots = m.lp_solve()
        assert st == "Optimal" and m.stat("mid_lp_solves") == 1 and m.stat("mid_lp_fallbacks") == 0 and pivots > 0
        lm = LinearModel()
        lm.add_variables(inst.l_var, inst.u_var)
        c = np.zeros(inst.n); c[inst.obj_col] = inst.obj_p0
        lm.set_objective("Min", np.arange(inst.n), c, 0.0)
        lm.add_rows(inst.rowptr, inst.col, inst.p0, inst.l_constr, inst.u_constr, assume_unique=True)
        assert lm.solve() == "Optimal"
        assert abs(m.getobjval() - lm.getobjval()) <= 1e-7 * max(1.0, abs(lm.getobjval()))
        x = m.getsolution()
        assert np.all(x >= inst.l_var - 1e-9) and np.all(x <= inst.u_var + 1e-9)
