"""Test helpers shared by the CPU and GPU tiers."""
import numpy as np

from oracle.evaluators import SeparableNLPEvaluator, SexprNLPEvaluator
from oracle.katana import KatanaModelParams, KatanaNonlinearModel as OracleModel


def oracle_evaluator(inst):
    return SeparableNLPEvaluator(inst.n, inst.rowptr, inst.col, inst.kind, inst.p0, inst.p1, inst.rconst,
                                 inst.obj_col, inst.obj_kind, inst.obj_p0, inst.obj_p1, inst.obj_const)


def oracle_solve_instance(inst, fast=True, **params):
    d = oracle_evaluator(inst)
    om = OracleModel(KatanaModelParams(**params), fast=fast)
    om.loadproblem(inst.n, inst.num_constr, inst.l_var, inst.u_var, inst.l_constr, inst.u_constr, inst.sense, d)
    om.optimize()
    return om


def oracle_solve_kat(m, **kw):
    n = len(m["vars"])
    d = SexprNLPEvaluator(n, m["objective"], [c["expr"] for c in m["constraints"]],
                          [c["linear"] for c in m["constraints"]], m["objective_linear"])
    om = OracleModel(KatanaModelParams(), **kw)
    om.loadproblem(n, len(m["constraints"]), [v["lb"] for v in m["vars"]], [v["ub"] for v in m["vars"]],
                   [c["lb"] for c in m["constraints"]], [c["ub"] for c in m["constraints"]], m["sense"], d)
    om.optimize()
    return om


def hip_model_from_kat(ktn, m, **solver_kw):
    import json, os
    solver_kw = dict(json.loads(os.environ.get("KTN_TEST_OPTS", "{}")), **solver_kw)      # experiments: override solver defaults
    M = ktn.Model(solver=ktn.KatanaSolver(**dict(dict(log_level=0), **solver_kw)))
    for v in m["vars"]:
        M.variable(v["lb"], v["ub"])
    M.objective(m["sense"], ktn.from_sexpr(m["objective"]), linear=m["objective_linear"])
    for c in m["constraints"]:
        M.constraint((ktn.from_sexpr(c["expr"]), c["lb"], c["ub"]), linear=c["linear"])
    return M


def hip_load_instance(ktn, inst, **solver_kw):
    m = ktn.NonlinearModel(ktn.KatanaSolver(**dict(dict(log_level=0), **solver_kw)))
    m.loadproblem(inst.n, inst.num_constr, inst.l_var, inst.u_var, inst.l_constr, inst.u_constr, inst.sense,
                  ktn.SeparableNLP(inst))
    return m


def planted_obj_bound(inst, f_tol=1e-6, lp_gap=1e-7):
    """Per-instance bound on |objective - planted optimum| at a point that meets the stop rule (src/model.jl:257,273).
    xhat is a KKT point with multipliers lambda_i (NL rows), mu_r (linear rows) and bound multipliers, so for the convex
    problem  f(x) >= f(xhat) - sum_i lambda_i max(g_i(x), 0) - sum_r mu_r max(a_r'x - b_r, 0)  for x inside the box.  The
    loop stops with every g_i <= f_tol; the LP leaves the linear rows within its row tolerance (0.3 f_tol, accepted up to
    twice that: DESIGN.md section 5), the box is met exactly (the prox clips), and the LP objective is certified to the
    relative duality-gap floor lp_gap.  With a nonlinear objective the epigraph row f(x) - t <= 0 is one more NL row
    with multiplier 1."""
    lam = inst.meta["lam_sum"] + (1.0 if inst.meta.get("objective") == "quad" else 0.0)
    return f_tol * (lam + 0.6 * inst.meta["mu_sum"]) + lp_gap * (1.0 + 2.0 * abs(inst.opt_obj))


def assert_planted_objective(obj, inst, f_tol=1e-6):
    """The objective assertions of every planted-optimum test, in this order:
    (1) the reference's own acceptance test, isapprox(obj, expected; atol = rtol = 1e-6) (test/runtests.jl:16-17, test/2d.jl:19),
        with the planted optimum as the expected value;
    (2) the per-instance Lagrangian bound of planted_obj_bound -- looser than (1) on the large configurations (its
        0.6 sum(mu) f_tol term dominates), kept because it is what the stop rule guarantees a priori."""
    tol = max(1e-6, 1e-6 * max(abs(obj), abs(inst.opt_obj)))
    assert abs(obj - inst.opt_obj) <= tol, ("reference tolerance 1e-6/1e-6", obj, inst.opt_obj, abs(obj - inst.opt_obj), tol)
    assert abs(obj - inst.opt_obj) <= planted_obj_bound(inst, f_tol=f_tol), ("planted bound", obj, inst.opt_obj)


def max_nl_violation(inst, x):
    """max over NL rows of g_i(x) - ub_i under the oracle's evaluator"""
    d = oracle_evaluator(inst)
    g = np.zeros(inst.num_constr)
    d.eval_g(g, np.asarray(x)[:inst.n])
    return float(np.max(g[inst.m_lin:] - inst.u_constr[inst.m_lin:])) if inst.m_nl else 0.0


# KATs whose *solution vector / tight objective* check depends on which LP vertices the LP
# solver happens to visit (GLPK in the reference): the stop rule g <= f_tol = 1e-6 bounds the
# tangential error of x only by ~sqrt(2 f_tol) = 1.4e-3 on these flat optima, while the
# reference's tests ask for 1e-3 (or rtol 1e-7 on 202_04).  Status and objective at the
# suite-wide 1e-6 tolerance are asserted for every KAT; for these ids x is asserted at 3e-3.
TRAJECTORY_SENSITIVE = {"105_04", "202_04", "501_02_n3", "501_02_n4", "501_02_n9"}
