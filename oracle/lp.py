"""LP model used by the oracle -- TEST INFRASTRUCTURE ONLY.

Stands in for `JuMP.Model(solver=lp_solver)` + GLPK (third-party, absent; call
sites src/model.jl:89,94,96,228,233,236,240,259,265,287,338,341).  The LP is
held in a live HiGHS instance (SciPy 1.15.3's vendored `_highspy`): rows are
appended with addRows and every re-solve is a warm-started dual simplex, which
is how the reference's live GLPK model behaves.

    min/max  c'x + c0   s.t.  l <= x <= u,  lo_r <= a_r'x <= hi_r

Unbounded rays.  GLPK's `getunboundedray` is not available through the SciPy
binding, so when HiGHS reports an unbounded LP the ray is computed from the
recession-cone LP

    min  s*c'd   s.t.  a_r'd {>=,<=,=} 0 (by which side of row r is finite),
                       d_j in [0,1] / [-1,0] / {0} / [-1,1] (by finite bounds)

(s = +1 for Min, -1 for Max).  Its optimum is negative exactly when the LP is
unbounded and the minimiser is an (inf-norm <= 1) improving ray.  The product's
GPU path uses the same construction (DESIGN.md "Unbounded LPs"), so the two can
be compared; GLPK would return a simplex edge instead -- the reference tests
never pin the ray, only the final answer.
"""
import numpy as np
import scipy.optimize._highspy._core as _hc

INF = float("inf")
_KINF = _hc.kHighsInf


def _new_highs(threads=1):
    h = _hc._Highs()
    h.setOptionValue("output_flag", bool(__import__("os").environ.get("KTN_ORACLE_LPLOG")))
    h.setOptionValue("solver", "simplex")
    if threads == 0:
        pass                                     # HiGHS's own defaults: simplex strategy "choose", threads automatic
    else:
        h.setOptionValue("simplex_strategy", 1)  # dual simplex, serial
        if int(threads) > 1:
            h.setOptionValue("threads", int(threads))
    h.setOptionValue("primal_feasibility_tolerance", 1e-9)
    h.setOptionValue("dual_feasibility_tolerance", 1e-9)
    return h


def _clip_inf(a, nan_to=None):
    a = np.array(a, dtype=np.float64)
    if nan_to is not None:
        a[np.isnan(a)] = nan_to
    a[a >= 1e300] = _KINF
    a[a <= -1e300] = -_KINF
    return a


class LinearModel:
    def __init__(self, threads=1):
        self.h = _new_highs(threads)
        self.n = 0
        self.sense = "Min"
        self.c = np.zeros(0)
        self.c0 = 0.0
        self.l = np.zeros(0)
        self.u = np.zeros(0)
        # row store (kept for the recession LP and for introspection)
        self.blocks, self.row_lo, self.row_hi = [], [], []
        self.nnz = 0
        self.num_solves = 0
        self.simplex_iters = 0

    # -- model building ----------------------------------------------------
    def add_variables(self, l, u):
        l = np.asarray(l, dtype=np.float64)
        u = np.asarray(u, dtype=np.float64)
        k = len(l)
        self.h.addVars(k, _clip_inf(l), _clip_inf(u))
        self.l = np.concatenate([self.l, l])
        self.u = np.concatenate([self.u, u])
        self.c = np.concatenate([self.c, np.zeros(k)])
        self.n += k

    def set_objective(self, sense, cols, coefs, constant=0.0):
        self.sense = sense
        c = np.zeros(self.n)
        np.add.at(c, np.asarray(cols, dtype=np.int64), np.asarray(coefs, dtype=np.float64))
        self.c = c
        self.c0 = float(constant)
        self.h.changeColsCost(self.n, np.arange(self.n, dtype=np.int32), c)
        self.h.changeObjectiveSense(_hc.ObjSense.kMaximize if sense == "Max" else _hc.ObjSense.kMinimize)
        self.h.changeObjectiveOffset(self.c0)

    def add_rows(self, rowptr, cols, vals, lo, hi, assume_unique=False):
        """Append a CSR block of rows (duplicates inside a row are merged, as
        JuMP.addconstraint does for an AffExpr with repeated variables)."""
        rowptr = np.asarray(rowptr, dtype=np.int64)
        cols = np.asarray(cols, dtype=np.int64)
        vals = np.asarray(vals, dtype=np.float64)
        nrows = len(rowptr) - 1
        if nrows == 0:
            return
        if not assume_unique:
            rows = np.repeat(np.arange(nrows), np.diff(rowptr))
            key = rows * max(self.n, 1) + cols
            uk, inv = np.unique(key, return_inverse=True)
            if len(uk) != len(key):
                vals = np.bincount(inv, weights=vals, minlength=len(uk))
                rows = uk // max(self.n, 1)
                cols = uk % max(self.n, 1)
                rowptr = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=nrows))])
        self.blocks.append((rowptr.copy(), cols.copy(), vals.copy()))
        self.row_lo.extend(np.asarray(lo, dtype=np.float64).tolist())
        self.row_hi.extend(np.asarray(hi, dtype=np.float64).tolist())
        self.nnz += len(cols)
        # A NaN row bound (cut constant NaN: g undefined but gradient finite,
        # e.g. log at x < 0) never compares as violated inside a simplex code,
        # i.e. the side is vacuous -> +-inf.  Unpinned inference about GLPK,
        # needed for KAT 105_04 (test/2d.jl:338-354); see DESIGN.md.
        self.h.addRows(nrows, _clip_inf(lo, -_KINF), _clip_inf(hi, _KINF), len(cols),
                       rowptr[:-1].astype(np.int32), cols.astype(np.int32), vals)

    def add_row(self, cols, vals, lo, hi):
        self.add_rows([0, len(cols)], cols, vals, [lo], [hi])

    @property
    def num_rows(self):
        return len(self.row_lo)

    # -- solving -----------------------------------------------------------
    def solve(self):
        self.h.run()
        self.num_solves += 1
        st = self.h.getModelStatus()
        if st == _hc.HighsModelStatus.kUnboundedOrInfeasible:
            self.h.setOptionValue("presolve", "off")
            self.h.run()
            st = self.h.getModelStatus()
        self.simplex_iters += int(self.h.getInfo().simplex_iteration_count)
        # cache: HiGHS drops solution/info as soon as rows are appended, while
        # the reference reads x* and the objective after adding cuts (model.jl:265,287)
        self._x = np.array(self.h.getSolution().col_value, dtype=np.float64)
        self._obj = float(self.h.getInfo().objective_function_value)
        if st == _hc.HighsModelStatus.kOptimal:
            return "Optimal"
        if st == _hc.HighsModelStatus.kUnbounded:
            return "Unbounded"
        if st == _hc.HighsModelStatus.kInfeasible:
            return "Infeasible"
        return "Error"

    def getsolution(self):
        return self._x.copy()

    def getobjval(self):
        return self._obj

    def getunboundedray(self, aux=None):
        """`aux`: index of the epigraph variable (or None).  Its box in the
        recession LP is widened to 1 + max_r sum_{j != aux} |a_rj| / |a_r,aux|
        so that the unit box on the structural variables, not the box on the
        epigraph variable, is what normalises the ray."""
        row_cols, row_vals = [], []
        for ptr, cc, vv in self.blocks:
            for r in range(len(ptr) - 1):
                row_cols.append(cc[ptr[r]:ptr[r + 1]])
                row_vals.append(vv[ptr[r]:ptr[r + 1]])
        scale = np.ones(self.n)
        if aux is not None:
            w = 0.0
            for cc, vv in zip(row_cols, row_vals):
                hit = np.nonzero(cc == aux)[0]
                if len(hit) and vv[hit[0]] != 0.0:
                    w = max(w, float((np.abs(vv).sum() - abs(vv[hit[0]])) / abs(vv[hit[0]])))
            scale[aux] = 1.0 + w
        return recession_ray(self.n, self.sense, self.c, self.l, self.u,
                             row_cols, row_vals, self.row_lo, self.row_hi, scale)


def recession_ray(n, sense, c, l, u, row_cols, row_vals, row_lo, row_hi, scale=None):
    """Improving ray of the LP from the recession-cone LP (module docstring).
    Returns the ray, or None when the LP is not unbounded."""
    h = _new_highs()
    scale = np.ones(n) if scale is None else scale
    dl = np.where(np.isfinite(l), 0.0, -scale)
    du = np.where(np.isfinite(u), 0.0, scale)
    h.addVars(n, dl, du)
    s = -1.0 if sense == "Max" else 1.0
    h.changeColsCost(n, np.arange(n, dtype=np.int32), s * np.asarray(c, dtype=np.float64))
    lo = np.asarray(row_lo, dtype=np.float64)
    hi = np.asarray(row_hi, dtype=np.float64)
    keep = [r for r in range(len(lo)) if np.isfinite(lo[r]) or np.isfinite(hi[r])]
    if keep:
        ptr, cc, vv, rl, rh = [0], [], [], [], []
        for r in keep:
            cc.append(row_cols[r])
            vv.append(row_vals[r])
            ptr.append(ptr[-1] + len(row_cols[r]))
            rl.append(0.0 if np.isfinite(lo[r]) else -_KINF)
            rh.append(0.0 if np.isfinite(hi[r]) else _KINF)
        h.addRows(len(keep), np.array(rl), np.array(rh), ptr[-1],
                  np.asarray(ptr[:-1], dtype=np.int32),
                  np.concatenate(cc).astype(np.int32), np.concatenate(vv))
    h.run()
    if h.getModelStatus() != _hc.HighsModelStatus.kOptimal:
        return None
    val = float(h.getInfo().objective_function_value)
    if not val < -1e-9:
        return None
    return np.array(h.getSolution().col_value, dtype=np.float64)
