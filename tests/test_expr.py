"""CPU tier: host logic -- expression graphs, tapes, linearity, descriptions, instances."""
import numpy as np

import katana_jl_amd as ktn
from kat_util import load_kats
from oracle import sexpr


def _eval_tape(ops, args, x):
    """reference stack machine for the postfix tape (test-side)"""
    L = ktn._lib
    st = []
    for o, a in zip(ops, args):
        if o == L.OP_CONST: st.append(a)
        elif o == L.OP_VAR: st.append(x[int(a)])
        elif o in (L.OP_ADD, L.OP_SUB, L.OP_MUL, L.OP_DIV):
            b = st.pop(); c = st.pop()
            st.append({L.OP_ADD: c + b, L.OP_SUB: c - b, L.OP_MUL: c * b, L.OP_DIV: c / b}[o])
        elif o == L.OP_NEG: st.append(-st.pop())
        elif o == L.OP_POWC: st.append(st.pop() ** a)
        else:
            v = st.pop()
            st.append({L.OP_EXP: np.exp, L.OP_LOG: np.log, L.OP_SQRT: np.sqrt, L.OP_SIN: np.sin, L.OP_COS: np.cos}[o](v))
    assert len(st) == 1
    return st[0]


def test_tapes_of_all_kat_expressions_evaluate_like_the_oracle():
    rng = np.random.default_rng(0)
    for m in load_kats():
        n = len(m["vars"])
        x = rng.uniform(0.3, 1.7, size=n)
        for s in [m["objective"]] + [c["expr"] for c in m["constraints"]]:
            e = ktn.from_sexpr(s)
            ops, args = e.tape()
            want = sexpr.eval_grad(s, x)[0]
            assert abs(_eval_tape(ops, args, x) - want) <= 1e-12 * max(1, abs(want))
            assert e.variables() == sexpr.variables(s)
            assert (e.affine() is not None) == sexpr.is_affine(s)


def test_linearity_flags_agree_with_the_fixture():
    for m in load_kats():
        for c in m["constraints"]:
            assert (ktn.from_sexpr(c["expr"]).affine() is not None) == c["linear"], m["id"]
        assert (ktn.from_sexpr(m["objective"]).affine() is not None) == m["objective_linear"], m["id"]


def test_affine_extraction():
    x, y = ktn.var(0), ktn.var(1)
    co, c0 = (3 * x - (y - 2) / 4 + 1).affine()
    assert co == {0: 3.0, 1: -0.25} and c0 == 1.5
    assert (x * y).affine() is None and (x ** 2).affine() is None


def test_exprnlp_description_shapes():
    x, y = ktn.var(0), ktn.var(1)
    d = ktn.ExprNLP(2, -x - y, [x ** 2 + y ** 2 - 1.0, x + y - 1.2])
    assert d.num_constr == 2 and list(d.row_kind) == [1, 0] and list(d.row_linear) == [0, 1]
    assert d.isobjlinear() and not d.isconstrlinear(0) and d.isconstrlinear(1)
    rows, cols = d.jac_structure()
    assert list(rows) == [0, 0, 1, 1] and list(cols) == [0, 1, 0, 1]
    cs = d.c_struct()
    assert cs.num_var == 2 and cs.obj_nnz == 2


def test_instance_generator_plants_a_kkt_vertex():
    inst = ktn.instances.make_instance(n=600, m_nl=60, k=16, family="explog", seed=5)
    val, der = ktn.instances.atom_value_deriv(inst.kind, inst.p0, inst.p1, inst.xhat[inst.col])
    rows = np.repeat(np.arange(inst.num_constr), np.diff(inst.rowptr))
    g = np.bincount(rows, weights=val, minlength=inst.num_constr) + inst.rconst
    assert np.all(g <= inst.u_constr + 1e-12)                     # xhat feasible
    act = np.abs(g - inst.u_constr) < 1e-12
    free = (inst.l_var < inst.xhat - 1e-12) & (inst.u_var > inst.xhat + 1e-12)
    assert act.sum() == free.sum() > 0                             # square active system
    assert np.all((inst.xhat >= inst.l_var) & (inst.xhat <= inst.u_var))
    c = np.zeros(inst.n); c[inst.obj_col] = inst.obj_p0
    assert abs(c @ inst.xhat - inst.opt_obj) < 1e-9
    # determinism
    again = ktn.instances.make_instance(n=600, m_nl=60, k=16, family="explog", seed=5)
    assert np.array_equal(again.p0, inst.p0) and np.array_equal(again.col, inst.col)


def test_fuse_instances_is_block_diagonal():
    a = ktn.instances.make_instance(n=50, m_nl=6, k=4, family="quad", seed=1)
    b = ktn.instances.make_instance(n=70, m_nl=9, k=5, family="explog", seed=2)
    f, offs = ktn.instances.fuse_instances([a, b])
    assert f.n == 120 and f.m_lin == a.m_lin + b.m_lin and f.m_nl == 15 and list(offs) == [0, 50, 120]
    assert abs(f.opt_obj - (a.opt_obj + b.opt_obj)) < 1e-12
    rp = f.rowptr
    # linear rows of a, then of b, then NL rows of a, then of b; columns shifted by the block offset
    la = a.rowptr[a.m_lin]
    assert np.array_equal(f.col[:la], a.col[:la])
    lb = b.rowptr[b.m_lin]
    assert np.array_equal(f.col[la:la + lb], b.col[:lb] + 50)
    first_nl_b = rp[f.m_lin + a.m_nl]
    assert np.array_equal(f.col[first_nl_b:], b.col[lb:] + 50) and np.array_equal(f.p0[first_nl_b:], b.p0[lb:])
    assert np.all(f.col[rp[f.m_lin]:first_nl_b] < 50)


def test_row_linearity_with_empty_rows_in_between_and_at_the_end():
    """SeparableNLP: a row is nonlinear iff one of its atoms is; empty rows (interior or trailing) are linear and must not
    cut the preceding row's reduction short (rp = [0,2,4,4], kinds [0,0,0,2]: row 1 IS nonlinear)."""
    from types import SimpleNamespace
    from katana_jl_amd.nlp import SeparableNLP
    inst = SimpleNamespace(n=3, rowptr=np.array([0, 2, 2, 4, 4, 4], dtype=np.int64), col=np.array([0, 1, 1, 2], dtype=np.int32),
                           kind=np.array([0, 0, 0, 2], dtype=np.uint8), p0=np.ones(4), p1=np.zeros(4), rconst=np.zeros(5),
                           obj_col=np.array([0], dtype=np.int32), obj_kind=np.array([0], dtype=np.uint8), obj_p0=np.ones(1),
                           obj_p1=np.zeros(1), obj_const=0.0)
    d = SeparableNLP(inst)
    assert list(d.row_linear) == [1, 1, 0, 1, 1]
    inst.rowptr = np.array([0, 2, 4, 4], dtype=np.int64); inst.rconst = np.zeros(3)
    assert list(SeparableNLP(inst).row_linear) == [1, 0, 1]
    inst.kind = np.array([3, 0, 0, 0], dtype=np.uint8)
    assert list(SeparableNLP(inst).row_linear) == [0, 1, 1]
