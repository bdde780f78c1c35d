"""CPU restatement of the Katana ECP driver -- TEST INFRASTRUCTURE ONLY.

Follows, function by function (0-based indices here, 1-based in the Julia):
    KatanaModelParams          src/Katana.jl:12-19, defaults src/solver.jl:34-43
    linear_oa_cut              src/algorithms.jl:3-18
    KatanaFirstOrderSeparator  src/separators.jl:58-120
    _addcut                    src/model.jl:68-79
    loadproblem!               src/model.jl:81-173
    boundroutine               src/model.jl:175-197
    round_coefs                src/model.jl:200-207
    optimize!                  src/model.jl:219-319
    getters                    src/model.jl:326-343
The LP (`JuMP.Model` + GLPK in the reference) is oracle.lp.LinearModel.

`fast=True` replaces the per-row Python loops of the main loop with a
numpy-vectorised batch that produces the same cuts (tests/test_oracle.py checks
the two against each other); it exists so bench.py's cpu_baseline can run the
restatement at BASELINE.json sizes in seconds rather than minutes.
"""
import math
import time

import numpy as np

from .evaluators import EpigraphNLPEvaluator
from .lp import LinearModel

INF = float("inf")


class AffExpr:
    """vars/coeffs/constant triple (JuMP.AffExpr as used by the reference)."""

    def __init__(self, vars_, coeffs, constant):
        self.vars = list(vars_)
        self.coeffs = list(coeffs)
        self.constant = constant


# --------------------------------------------------------------------------
# separators.jl / algorithms.jl
# --------------------------------------------------------------------------
def linear_oa_cut(sep, a, b, i):
    """src/algorithms.jl:3-18 -- g_i(a) + (x - a).grad g_i(a), from precomputed g/jac."""
    v, coefs = [], []
    b = sep.g[i]
    for col, colix in zip(sep.sp_cols[i], sep.sp_col_inds[i]):
        v.append(col)
        partial = sep.jac[colix]
        coefs.append(partial)
        b += -sep.xstar[col] * partial
    return AffExpr(v, coefs, b)


class KatanaFirstOrderSeparator:
    """src/separators.jl:58-120."""

    def __init__(self, algo=linear_oa_cut):
        self.algo = algo

    def initialize(self, linear_model, num_var, num_constr, oracle):   # :81-107
        self.linear_model = linear_model
        oracle.initialize(["Grad", "Jac"])
        self.oracle = oracle
        sp_rows, sp_cols = oracle.jac_structure()
        sp_rows = np.asarray(sp_rows, dtype=np.int64)
        sp_cols = np.asarray(sp_cols, dtype=np.int64)
        N = len(sp_rows)
        # per-row lists in storage order == stable sort of the COO by row
        order = np.argsort(sp_rows, kind="stable")
        counts = np.bincount(sp_rows, minlength=num_constr)
        self.csr_ptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        self.csr_col = sp_cols[order]
        self.csr_ind = order
        self.sp_cols = _RowView(self.csr_ptr, self.csr_col)
        self.sp_col_inds = _RowView(self.csr_ptr, self.csr_ind)
        self.g = np.zeros(num_constr)
        self.jac = np.zeros(N)
        self.num_var = num_var
        self.num_constr = num_constr

    def precompute(self, xstar):                                        # :111-116
        self.oracle.eval_jac_g(self.jac, xstar)
        self.oracle.eval_g(self.g, xstar)
        self.xstar = xstar

    def gencut(self, xstar, bounds, i):                                 # :118
        return self.algo(self, xstar, bounds, i)

    def isconstrsat(self, i, lb, ub, f_tol):                            # :120
        return (self.g[i] >= lb - f_tol) and (self.g[i] <= ub + f_tol)


class _RowView:
    def __init__(self, ptr, data):
        self.ptr, self.data = ptr, data

    def __getitem__(self, i):
        return self.data[self.ptr[i]:self.ptr[i + 1]]


def round_coefs(cut, cut_coef_rng):
    """src/model.jl:200-207 -- SIGNED max; the constant is not recomputed."""
    max_coef = max(cut.coeffs) if cut.coeffs else -INF
    for i in range(len(cut.coeffs)):
        if cut.coeffs[i] + cut_coef_rng < max_coef:
            cut.coeffs[i] = 0.0


# --------------------------------------------------------------------------
# Katana.jl / solver.jl
# --------------------------------------------------------------------------
class KatanaModelParams:
    def __init__(self, f_tol=1e-6, iter_cap=10000, log_level=0, cut_coef_rng=1e9,
                 obj_eps=-1.0, separator=None):
        self.f_tol = f_tol
        self.iter_cap = iter_cap
        self.log_level = log_level
        self.cut_coef_rng = cut_coef_rng
        self.obj_eps = obj_eps
        self.separator = separator if separator is not None else KatanaFirstOrderSeparator()


def glpk_bound_vertex(l_var, u_var):
    """Solution of the objective-less, row-less LP of src/model.jl:92-97.

    GLPK (absent) leaves every column non-basic: at its lower bound when that is
    finite (and |lb| <= |ub| for a boxed column), else at its finite upper
    bound, else (free column) at 0, and reports :Optimal.  Unpinned convention,
    shared with the product (DESIGN.md "Bound-box vertex")."""
    l = np.asarray(l_var, dtype=np.float64)
    u = np.asarray(u_var, dtype=np.float64)
    lf, uf = np.isfinite(l), np.isfinite(u)
    v = np.zeros(len(l))
    both = lf & uf
    v[both] = np.where(np.abs(l[both]) <= np.abs(u[both]), l[both], u[both])
    v[lf & ~uf] = l[lf & ~uf]
    v[~lf & uf] = u[~lf & uf]
    return v


# --------------------------------------------------------------------------
# model.jl
# --------------------------------------------------------------------------
class KatanaNonlinearModel:
    def __init__(self, params=None, fast=False, lp_threads=1, vis_data=False, lp_factory=None):
        self.params = params if params is not None else KatanaModelParams()
        self.status = "None"                 # src/model.jl:44
        self.objval = float("nan")
        self.nlconstr_ixs = []
        self.iter = 0
        self.numcuts = 0
        self.soltime = 0.0
        self.fast = fast
        self.lp_threads = lp_threads
        self.lp_factory = lp_factory if lp_factory is not None else LinearModel
        self.vis_data = vis_data
        self.lp_sols = []
        self.linear_cuts = []
        self.lp_time = 0.0
        self.sep_time = 0.0

    # src/model.jl:68-79
    def _addcut(self, cut, lb, ub):
        if any(not math.isfinite(c) for c in cut.coeffs):
            self.status = "Error"
            return
        c = cut.constant
        self.linear_model.add_row(cut.vars, cut.coeffs, lb - c, ub - c)
        self.numcuts += 1
        if self.vis_data:
            self.linear_cuts.append((list(cut.vars), list(cut.coeffs), lb - c, ub - c))

    # src/model.jl:81-173
    def loadproblem(self, num_var, num_constr, l_var, u_var, l_constr, u_constr, sense, d):
        self.linear_model = self.lp_factory(threads=self.lp_threads)
        self.linear_model.add_variables(l_var, u_var)                      # :92
        vertex = glpk_bound_vertex(l_var, u_var)                            # :93-97
        if np.any(np.asarray(l_var) > np.asarray(u_var)):
            vertex = np.full(num_var, np.nan)

        self.num_var = num_var
        self.num_constr = num_constr
        self.l_constr = list(np.asarray(l_constr, dtype=np.float64))
        self.u_constr = list(np.asarray(u_constr, dtype=np.float64))

        fsep = KatanaFirstOrderSeparator()                                  # :110
        epi_d = EpigraphNLPEvaluator(d, num_var + 1, num_constr + 1)        # :111
        fsep.initialize(self.linear_model, num_var + 1, num_constr + 1, epi_d)
        pt = np.zeros(num_var + 1)
        with np.errstate(all="ignore"):
            fsep.precompute(pt)
        lin_rows = [i for i in range(num_constr) if d.isconstrlinear(i)]
        self.nlconstr_ixs = [i for i in range(num_constr) if not d.isconstrlinear(i)]
        if self.fast and lin_rows:
            self._add_rows_batch(fsep, np.asarray(lin_rows), round_=False)
        else:
            for i in lin_rows:                                              # :115-122
                cut = fsep.gencut(pt, (self.l_constr[i], self.u_constr[i]), i)
                self._addcut(cut, self.l_constr[i], self.u_constr[i])

        self.objislinear = d.isobjlinear()                                  # :125
        if self.objislinear:
            cut = fsep.gencut(pt, (0, 0), num_constr)                       # :129
            assert cut.vars[-1] == num_var
            cut.vars.pop()
            cut.coeffs.pop()
            self.linear_model.set_objective(sense, cut.vars, cut.coeffs, cut.constant)
        else:
            self.linear_model.add_variables([-INF], [INF])                  # :137
            self.num_var += 1
            self.linear_model.set_objective(sense, [self.num_var - 1], [1.0])
            l_obj, u_obj = (0.0, INF) if sense == "Max" else (-INF, 0.0)    # :144
            self.l_constr.append(l_obj)
            self.u_constr.append(u_obj)
            self.num_constr += 1
            self.nlconstr_ixs.append(self.num_constr - 1)
            if np.any(np.isnan(vertex)):                                    # :156
                pass  # "Problem variables insufficiently bounded!"
            else:
                vertex = np.append(vertex, d.eval_f(vertex))               # :159
                fsep.precompute(vertex)
                cut = fsep.gencut(vertex, (l_obj, u_obj), self.num_constr - 1)
                round_coefs(cut, self.params.cut_coef_rng)
                self._addcut(cut, l_obj, u_obj)
            d = EpigraphNLPEvaluator(d, num_var + 1, num_constr + 1)        # :166
        self.num_nlconstr = len(self.nlconstr_ixs)
        self.oracle = d
        self.params.separator.initialize(self.linear_model, self.num_var, self.num_constr, d)  # :171-172
        self._nl = np.asarray(self.nlconstr_ixs, dtype=np.int64)
        self._lc = np.asarray(self.l_constr, dtype=np.float64)
        self._uc = np.asarray(self.u_constr, dtype=np.float64)

    # src/model.jl:175-197
    def boundroutine(self, ray):
        sep = self.params.separator
        for n in range(2, 1024):
            x = (2.0 ** n) * ray
            allsat = True
            with np.errstate(all="ignore"):
                sep.precompute(x)
            for i in self.nlconstr_ixs:
                sat = sep.isconstrsat(i, self.l_constr[i], self.u_constr[i], self.params.f_tol)
                if not sat:
                    cut = sep.gencut(x, (self.l_constr[i], self.u_constr[i]), i)
                    round_coefs(cut, self.params.cut_coef_rng)
                    self._addcut(cut, self.l_constr[i], self.u_constr[i])
                allsat &= sat
            if not allsat:
                break

    # vectorised equivalent of {isconstrsat, gencut, round_coefs, _addcut} over rows `ixs`
    def _add_rows_batch(self, sep, ixs, round_=True):
        ptr = sep.csr_ptr
        lens = ptr[ixs + 1] - ptr[ixs]
        starts = ptr[ixs]
        tot = int(lens.sum())
        rowid = np.repeat(np.arange(len(ixs)), lens)
        offs = np.arange(tot) - np.repeat(np.cumsum(lens) - lens, lens)
        pos = np.repeat(starts, lens) + offs
        cols = sep.csr_col[pos]
        coefs = sep.jac[sep.csr_ind[pos]].copy()
        with np.errstate(all="ignore"):
            dot = np.bincount(rowid, weights=sep.xstar[cols] * coefs, minlength=len(ixs))
        b = sep.g[ixs] - dot
        if round_:
            mx = np.full(len(ixs), -INF)
            np.maximum.at(mx, rowid, coefs)
            coefs[coefs + self.params.cut_coef_rng < mx[rowid]] = 0.0
        if not np.all(np.isfinite(coefs)):
            self.status = "Error"
            return 0
        lo = self._lc_of(ixs) - b
        hi = self._uc_of(ixs) - b
        rp = np.concatenate([[0], np.cumsum(lens)])
        self.linear_model.add_rows(rp, cols, coefs, lo, hi, assume_unique=False)
        self.numcuts += len(ixs)
        return len(ixs)

    def _lc_of(self, ixs):
        return np.asarray(self.l_constr, dtype=np.float64)[ixs]

    def _uc_of(self, ixs):
        return np.asarray(self.u_constr, dtype=np.float64)[ixs]

    # src/model.jl:219-319
    def optimize(self):
        p = self.params
        sep = p.separator
        lm = self.linear_model
        start = time.time()
        status = lm.solve()                                                 # :228
        i = 0
        while status == "Unbounded" and i < self.num_var:                   # :235-242
            ray = lm.getunboundedray(aux=None if self.objislinear else self.num_var - 1)
            if ray is None:
                break
            self.boundroutine(ray)
            if self.status == "Error":
                return self.status
            status = lm.solve()
            i += 1
        if status == "Unbounded":                                           # :244-247
            self.status = status
            return self.status

        allsat = False
        obj_prev = INF
        while (not allsat) and self.iter < p.iter_cap:                      # :257
            self.iter += 1
            t0 = time.time()
            status = lm.solve()                                             # :259
            self.lp_time += time.time() - t0
            if status != "Optimal":
                self.status = status
                return self.status
            xstar = lm.getsolution()                                        # :265
            if self.vis_data:
                self.lp_sols.append(xstar.copy())
            t0 = time.time()
            with np.errstate(all="ignore"):
                sep.precompute(xstar)                                       # :268
            allsat = True
            if self.fast:
                gi = sep.g[self._nl]
                sat = (gi >= self._lc[self._nl] - p.f_tol) & (gi <= self._uc[self._nl] + p.f_tol)
                viol = self._nl[~sat]
                if len(viol):
                    self._add_rows_batch(sep, viol)
                    if self.status == "Error":
                        return self.status
                allsat = bool(np.all(sat))
            else:
                for i in self.nlconstr_ixs:                                 # :272-283
                    sat = sep.isconstrsat(i, self.l_constr[i], self.u_constr[i], p.f_tol)
                    if not sat:
                        cut = sep.gencut(xstar, (self.l_constr[i], self.u_constr[i]), i)
                        round_coefs(cut, p.cut_coef_rng)
                        self._addcut(cut, self.l_constr[i], self.u_constr[i])
                        if self.status == "Error":
                            return self.status
                    allsat &= sat
            self.sep_time += time.time() - t0
            obj = lm.getobjval()                                            # :287
            with np.errstate(all="ignore"):
                obj_delta = abs(np.float64(obj_prev - obj) / np.float64(obj))
            obj_prev = obj
            if p.log_level > 0 and (self.iter % p.log_level == 0 or allsat):
                print("oracle iter %d cuts %d obj %.10g lp_time %.2fs" % (self.iter, self.numcuts, obj, self.lp_time), flush=True)
            if obj_delta <= p.obj_eps:                                      # :306-308
                break
        self.soltime = time.time() - start                                  # :311
        if self.iter >= p.iter_cap:
            status = "UserLimit"
        self.status = status
        return self.status

    # src/model.jl:326-343
    def numiters(self):
        return self.iter

    def getnumcuts(self):
        return self.numcuts

    def getstatus(self):
        return self.status

    def getobjval(self):
        return self.linear_model.getobjval()

    def getsolution(self):
        return self.linear_model.getsolution()

    def getsolvetime(self):
        return self.soltime
