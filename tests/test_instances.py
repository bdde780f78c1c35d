"""CPU tier: the instance generators the GPU tiers and bench.py run on -- the degeneracy dial of make_instance (round 4:
validation OFF the non-degenerate-vertex family), the LP-only battery, and the committed oracle results on the off-family
instances (tests/golden/offfamily_oracle.json)."""
import hashlib
import json
import os

import numpy as np
import pytest

import katana_jl_amd as ktn
from helpers import oracle_solve_instance, planted_obj_bound

HERE = os.path.dirname(os.path.abspath(__file__))


def _digest(i):
    h = hashlib.sha256()
    for a in (i.col, i.kind, i.p0, i.p1, i.rconst, i.l_var, i.u_var, i.u_constr, i.obj_col, i.obj_p0, i.obj_p1, i.xhat):
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()[:16]


def test_the_dial_at_one_is_the_generator_every_earlier_round_ran_on():
    """bound_frac = 1 (the default) must not move a single number of the vertex family: three rounds of seed sweeps, the bench
    workload and the committed profiles were made on it.  The digests are those of the round-3 generator."""
    assert _digest(ktn.instances.make_instance(n=400, m_nl=40, k=16, family="explog", seed=3)) == "1c1edd8cbff34019"
    assert _digest(ktn.instances.make_instance(n=300, m_nl=30, k=8, family="quad", seed=7, objective="quad")) == "73819feac492c58a"
    assert _digest(ktn.instances.make_instance(n=200, m_nl=20, k=8, family="explog+t", seed=1)) == "6baf610c6f70eb4c"
    # ... and the other end of the dial is the survey's smooth-face generator (what vertex=False always was)
    a = ktn.instances.make_instance(n=200, m_nl=20, k=8, family="quad", seed=2, bound_frac=0.0)
    assert _digest(a) == "6002ef4b9ff67f88" and _digest(ktn.instances.make_instance(n=200, m_nl=20, k=8, family="quad", seed=2, vertex=False)) == _digest(a)
    assert a.meta["vertex"] is False and a.meta["bound_frac"] == 0.0


@pytest.mark.parametrize("family", ["explog", "quad"])
@pytest.mark.parametrize("bound_frac", [0.0, 0.5, 1.0])
def test_the_planted_point_is_the_optimum_at_every_setting_of_the_dial(family, bound_frac):
    """xhat is planted as a KKT point; the oracle (Kelley's method on HiGHS vertices, src/model.jl:257-309) must end at its
    objective -- and needs more cutting-plane rounds the fewer variables are pinned (the smooth-face regime)."""
    inst = ktn.instances.make_instance(n=30, m_nl=4, k=6, family=family, seed=5, bound_frac=bound_frac)
    om = oracle_solve_instance(inst)
    assert om.status == "Optimal"
    assert abs(om.getobjval() - inst.opt_obj) <= 1e-6 * max(1.0, abs(inst.opt_obj))
    pinned = np.sum((inst.l_var == inst.xhat) | (inst.u_var == inst.xhat))
    assert (pinned == 0) if bound_frac == 0.0 else (pinned > 0)


def test_offfamily_fixture_is_consistent_with_the_generator():
    """tests/golden/offfamily_oracle.json: every case the oracle finished ended `:Optimal` within 1e-6 of the planted optimum
    (the reference's acceptance test, test/runtests.jl:16-17), the planted value recorded there is what the generator gives
    today, and the cases it did NOT finish within the limit are there as evidence (n = 1000: none finished in 40 minutes)."""
    fx = json.load(open(os.path.join(HERE, "golden", "offfamily_oracle.json")))
    done = [c for c in fx["cases"] if "status" in c]
    assert len(fx["cases"]) == 24 and len(done) == 9 and all(c["n"] == 200 for c in done)
    beyond = 0
    for c in done:
        inst = ktn.instances.make_instance(n=c["n"], m_nl=c["m_nl"], k=c["k"], family=c["family"], seed=c["seed"], bound_frac=c["bound_frac"])
        assert inst.opt_obj == c["planted"]
        # The stop rule (every NL row within f_tol, src/model.jl:257,273) leaves the objective short by up to f_tol times the
        # multipliers: on ONE of the nine cases (explog, bound_frac 0, seed 1) the oracle's own exact-vertex run ends 1.35e-6 below
        # the planted optimum -- outside the 1e-6 of test/runtests.jl:16-17, inside the a-priori bound of the stop rule.
        err = abs(c["objective"] - c["planted"])
        assert c["status"] == "Optimal" and c["objective"] <= c["planted"] + 1e-9 and err <= 1.5 * planted_obj_bound(inst)
        beyond += err > 1e-6 * max(1.0, abs(c["planted"]))
    assert beyond == 1
    assert all("oracle_timeout_s" in c or "not_run" in c for c in fx["cases"] if c["n"] == 1000)


def test_lp_battery_cases_have_the_planted_optimum():
    """make_lp plants a primal-dual optimal pair; HiGHS agrees on a small case of every kind (degenerate, badly scaled, free columns)"""
    from oracle.lp import LinearModel
    for kind in range(5):
        kw = dict(ktn.instances.lp_battery_case(kind), n=120, m=150)
        inst = ktn.instances.make_lp(**kw)
        lm = LinearModel()
        lm.add_variables(inst.l_var, inst.u_var)
        c = np.zeros(inst.n); c[inst.obj_col] = inst.obj_p0
        lm.set_objective("Min", np.arange(inst.n), c, 0.0)
        lm.add_rows(inst.rowptr, inst.col, inst.p0, inst.l_constr, inst.u_constr)
        assert lm.solve() == "Optimal"
        assert abs(lm.getobjval() - inst.opt_obj) <= 1e-7 * max(1.0, abs(inst.opt_obj)), (kind, lm.getobjval(), inst.opt_obj)
