"""NLP evaluators for the oracle -- TEST INFRASTRUCTURE ONLY.

They expose the subset of the MathProgBase.AbstractNLPEvaluator interface the
reference consumes (SURVEY.md section 8b "Evaluator consumed"):
    initialize(features)            src/separators.jl:88
    jac_structure() -> rows, cols   src/separators.jl:92
    eval_g(g, x), eval_jac_g(J, x)  src/separators.jl:112-113
    eval_f(x), eval_grad_f(g, x)    src/model.jl:159, src/nlpeval.jl:35-39
    isconstrlinear(i), isobjlinear  src/model.jl:116,125
Indices are 0-based here (the reference is 1-based Julia).

* SexprNLPEvaluator    -- closed-form models of the reference's tests
                          (stands in for JuMP's AD evaluator).
* SeparableNLPEvaluator -- the synthetic families of SURVEY.md section 8d,
                          numpy-vectorised so the CPU baseline can run at
                          BASELINE.json sizes.
* EpigraphNLPEvaluator -- restatement of src/nlpeval.jl:6-63.
"""
import numpy as np

from . import sexpr

# separable atom kinds (shared vocabulary with include/katana_hip.h KTN_ATOM_*)
ATOM_LIN, ATOM_QUAD, ATOM_EXP, ATOM_NEGLOG = 0, 1, 2, 3


class SexprNLPEvaluator:
    def __init__(self, num_var, obj_expr, constr_exprs, constr_linear=None, obj_linear=None):
        self.num_var = int(num_var)
        self.obj_expr = obj_expr
        self.constr_exprs = list(constr_exprs)
        self.constr_linear = (list(constr_linear) if constr_linear is not None
                              else [sexpr.is_affine(e) for e in self.constr_exprs])
        self.obj_linear = sexpr.is_affine(obj_expr) if obj_linear is None else bool(obj_linear)
        self._cols = [sexpr.variables(e) for e in self.constr_exprs]

    def features_available(self):
        return ["Grad", "Jac"]

    def initialize(self, requested):
        for f in requested:
            if f not in self.features_available():
                raise ValueError("Unsupported feature %s" % f)

    def isobjlinear(self):
        return self.obj_linear

    def isconstrlinear(self, i):
        return self.constr_linear[i]

    def jac_structure(self):
        rows, cols = [], []
        for i, cs in enumerate(self._cols):
            rows.extend([i] * len(cs))
            cols.extend(cs)
        return rows, cols

    def eval_f(self, x):
        return sexpr.eval_grad(self.obj_expr, x)[0]

    def eval_grad_f(self, g, x):
        _, gr = sexpr.eval_grad(self.obj_expr, x)
        g[: self.num_var] = 0.0
        for j, v in gr.items():
            g[j] = v

    def eval_g(self, g, x):
        for i, e in enumerate(self.constr_exprs):
            g[i] = sexpr.eval_grad(e, x)[0]

    def eval_jac_g(self, J, x):
        k = 0
        for i, e in enumerate(self.constr_exprs):
            _, gr = sexpr.eval_grad(e, x)
            for j in self._cols[i]:
                J[k] = gr.get(j, 0.0)
                k += 1


def _atoms(kind, p0, p1, xv):
    """value and derivative of every separable atom (vectorised)."""
    with np.errstate(all="ignore"):
        val = np.empty_like(xv)
        der = np.empty_like(xv)
        m = kind == ATOM_LIN
        val[m] = p0[m] * xv[m]
        der[m] = p0[m]
        m = kind == ATOM_QUAD
        d = xv[m] - p1[m]
        val[m] = p0[m] * d * d
        der[m] = 2.0 * p0[m] * d
        m = kind == ATOM_EXP
        e = p0[m] * np.exp(p1[m] * xv[m])
        val[m] = e
        der[m] = p1[m] * e
        m = kind == ATOM_NEGLOG
        s = xv[m] + p1[m]
        val[m] = -p0[m] * np.log(s)
        der[m] = -p0[m] / s
    return val, der


class SeparableNLPEvaluator:
    """g_i(x) = sum_e atom_e(x[col_e]) + rconst_i over CSR rows; objective the
    same form.  Plain arrays in, so tests can hand the *same* arrays to the HIP
    path and to this oracle."""

    def __init__(self, num_var, rowptr, col, kind, p0, p1, rconst,
                 obj_col, obj_kind, obj_p0, obj_p1, obj_const=0.0):
        self.num_var = int(num_var)
        self.rowptr = np.asarray(rowptr, dtype=np.int64)
        self.col = np.asarray(col, dtype=np.int64)
        self.kind = np.asarray(kind, dtype=np.int64)
        self.p0 = np.asarray(p0, dtype=np.float64)
        self.p1 = np.asarray(p1, dtype=np.float64)
        self.rconst = np.asarray(rconst, dtype=np.float64)
        self.num_constr = len(self.rowptr) - 1
        self.obj_col = np.asarray(obj_col, dtype=np.int64)
        self.obj_kind = np.asarray(obj_kind, dtype=np.int64)
        self.obj_p0 = np.asarray(obj_p0, dtype=np.float64)
        self.obj_p1 = np.asarray(obj_p1, dtype=np.float64)
        self.obj_const = float(obj_const)
        self._rows = np.repeat(np.arange(self.num_constr), np.diff(self.rowptr))
        lin_entry = self.kind == ATOM_LIN
        nonlin_per_row = np.bincount(self._rows[~lin_entry], minlength=self.num_constr)
        self._row_linear = nonlin_per_row == 0

    def features_available(self):
        return ["Grad", "Jac"]

    def initialize(self, requested):
        for f in requested:
            if f not in self.features_available():
                raise ValueError("Unsupported feature %s" % f)

    def isobjlinear(self):
        return bool(np.all(self.obj_kind == ATOM_LIN))

    def isconstrlinear(self, i):
        return bool(self._row_linear[i])

    def jac_structure(self):
        return self._rows, self.col

    def eval_f(self, x):
        x = np.asarray(x, dtype=np.float64)
        val, _ = _atoms(self.obj_kind, self.obj_p0, self.obj_p1, x[self.obj_col])
        return float(val.sum() + self.obj_const)

    def eval_grad_f(self, g, x):
        x = np.asarray(x, dtype=np.float64)
        _, der = _atoms(self.obj_kind, self.obj_p0, self.obj_p1, x[self.obj_col])
        g[: self.num_var] = 0.0
        np.add.at(g, self.obj_col, der)

    def eval_g(self, g, x):
        x = np.asarray(x, dtype=np.float64)
        val, _ = _atoms(self.kind, self.p0, self.p1, x[self.col])
        g[: self.num_constr] = np.bincount(self._rows, weights=val, minlength=self.num_constr) + self.rconst

    def eval_jac_g(self, J, x):
        x = np.asarray(x, dtype=np.float64)
        _, der = _atoms(self.kind, self.p0, self.p1, x[self.col])
        J[: len(der)] = der


class EpigraphNLPEvaluator:
    """Restatement of src/nlpeval.jl:6-63: present `f(x) - t` as one extra,
    *dense* constraint row over num_var (= inner + 1) variables."""

    def __init__(self, d, num_var, num_constr):
        self.nlpeval = d
        self.num_var = num_var          # including the auxiliary variable
        self.num_constr = num_constr    # including the epigraph row
        self.grad_f = np.zeros(num_var)  # src/nlpeval.jl:11,14

    def isobjlinear(self):              # src/nlpeval.jl:17
        return self.nlpeval.isobjlinear()

    def isconstrlinear(self, i):        # src/nlpeval.jl:19
        return self.nlpeval.isconstrlinear(i)

    def features_available(self):       # src/nlpeval.jl:23
        return ["Grad", "Jac"]

    def initialize(self, requested):    # src/nlpeval.jl:25-32
        for f in requested:
            if f not in self.features_available():
                raise ValueError("Unsupported feature %s" % f)
        self.nlpeval.initialize(requested)

    def eval_f(self, x):                # src/nlpeval.jl:35
        return self.nlpeval.eval_f(x[:-1]) - x[-1]

    def eval_grad_f(self, g, x):        # src/nlpeval.jl:36-39
        self.nlpeval.eval_grad_f(g, x[:-1])
        g[-1] = -1.0

    def eval_g(self, g, x):             # src/nlpeval.jl:42-45
        self.nlpeval.eval_g(g, x[:-1])
        g[self.num_constr - 1] = self.eval_f(x)

    def jac_structure(self):            # src/nlpeval.jl:49-54
        rows, cols = self.nlpeval.jac_structure()
        rows = list(rows) + [self.num_constr - 1] * self.num_var
        cols = list(cols) + list(range(self.num_var))
        return rows, cols

    def eval_jac_g(self, J, x):         # src/nlpeval.jl:59-63
        self.nlpeval.eval_jac_g(J, x[:-1])
        self.eval_grad_f(self.grad_f, x)
        J[len(J) - self.num_var:] = self.grad_f
