"""Host mirror of Katana's plugin surface over the C ABI (src/solver.jl, src/model.jl,
src/separators.jl, src/util.jl).  Names, argument meaning and status vocabulary follow the
reference; indices are 0-based."""
import ctypes as C
import os

import numpy as np

from . import _lib as L

STATUS_SYMBOLS = {L.STATUS_NONE: "None", L.STATUS_OPTIMAL: "Optimal", L.STATUS_UNBOUNDED: "Unbounded",
                  L.STATUS_INFEASIBLE: "Infeasible", L.STATUS_USERLIMIT: "UserLimit", L.STATUS_ERROR: "Error"}
_SUPPORTED_FEATURES = ("VisData",)   # src/solver.jl:31-32


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a, t=C.c_double):
    return a.ctypes.data_as(C.POINTER(t))


class KatanaSolver:
    """KatanaSolver(lp_solver; separator, features, f_tol=1e-6, cut_coef_rng=1e9, log_level=10,
    iter_cap=10000, obj_eps=-1.0)  -- src/solver.jl:34-43.

    `lp_solver` is accepted for call-site compatibility (`KatanaSolver(GLPKSolverLP(), ...)`,
    test/runtests.jl:24) and ignored: the LP is solved on the GPU.  Extra keywords `lp_*`,
    `device`, `profile` reach the GPU LP (ktn_params)."""

    def __init__(self, lp_solver=None, separator=None, features=(), f_tol=1e-6, cut_coef_rng=1e9, log_level=10,
                 iter_cap=10000, obj_eps=-1.0, **gpu_options):
        self.lp_solver = lp_solver
        self.features = list(features)
        for f in self.features:
            if f not in _SUPPORTED_FEATURES:
                raise ValueError("type KatanaFeatures has no field %s" % f)   # setfield! error, src/model.jl:50-52
        self.separator = separator
        self.model_params = dict(f_tol=float(f_tol), cut_coef_rng=float(cut_coef_rng), log_level=int(log_level),
                                 iter_cap=int(iter_cap), obj_eps=float(obj_eps))
        self.gpu_options = dict(gpu_options)


def NonlinearModel(s):
    """MathProgBase.NonlinearModel(s::KatanaSolver)  src/model.jl:63-65"""
    return KatanaNonlinearModel(s)


LinearQuadraticModel = NonlinearModel   # src/solver.jl:46: the LP/QP bridge presents the same model


class KatanaNonlinearModel:
    def __init__(self, solver):
        self.solver = solver
        lib = L.lib()
        p = L.KtnParams()
        lib.ktn_default_params(C.byref(p))
        for k, v in solver.model_params.items():
            setattr(p, k, v)
        p.vis_data = 1 if "VisData" in solver.features else 0
        for k, v in solver.gpu_options.items():
            if not hasattr(p, k):
                raise TypeError("unknown KatanaSolver option %r" % k)
            setattr(p, k, v)
        self.params = p
        self._h = C.c_void_p()
        code = lib.ktn_create(C.byref(p), C.byref(self._h))
        if code != L.KTN_OK:
            raise L.KatanaHipError(code, "ktn_create failed (no MI355X visible? the engine has no CPU path)")
        self._lib = lib
        self._pid = os.getpid()          # a handle belongs to the process that made it (a forked child must not touch the GPU state)
        self._desc = None
        self.num_var = 0
        self.num_constr = 0

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value and getattr(self, "_pid", None) == os.getpid():
                self._lib.ktn_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    # ---- MathProgBase.loadproblem!  src/model.jl:81-173 -----------------------------------
    def loadproblem(self, num_var, num_constr, l_var, u_var, l_constr, u_constr, sense, d):
        l_var, u_var, l_constr, u_constr = _f64(l_var), _f64(u_var), _f64(l_constr), _f64(u_constr)
        assert len(l_var) == num_var and len(u_var) == num_var
        assert len(l_constr) == num_constr and len(u_constr) == num_constr
        sense_code = {"Min": L.MIN, "Max": L.MAX}[sense]
        self._desc = d
        cd = d.c_struct()
        L.check(self._h, self._lib.ktn_loadproblem(self._h, num_var, num_constr, _p(l_var), _p(u_var), _p(l_constr),
                                                   _p(u_constr), sense_code, C.byref(cd)))
        self.num_var = int(self._lib.ktn_get_num_var(self._h))
        self.num_constr = int(self._lib.ktn_sep_num_constr(self._h))

    # ---- MathProgBase.optimize!  src/model.jl:219-319 ---------------------------------------
    def optimize(self):
        return STATUS_SYMBOLS[L.check(self._h, self._lib.ktn_optimize(self._h))]

    def optimize_begin(self):
        L.check(self._h, self._lib.ktn_optimize_begin(self._h))

    def ecp_step(self):
        done = C.c_int32(0)
        L.check(self._h, self._lib.ktn_ecp_step(self._h, C.byref(done)))
        return bool(done.value)

    def optimize_end(self):
        return STATUS_SYMBOLS[L.check(self._h, self._lib.ktn_optimize_end(self._h))]

    def reset(self):
        L.check(self._h, self._lib.ktn_reset(self._h))

    # ---- getters  src/model.jl:326-343 ---------------------------------------------------
    def status(self):
        return STATUS_SYMBOLS[self._lib.ktn_get_status(self._h)]

    def getobjval(self):
        return float(self._lib.ktn_get_objval(self._h))

    def getsolution(self):
        x = np.empty(self.num_var)
        L.check(self._h, self._lib.ktn_get_solution(self._h, _p(x), len(x)))
        return x

    def getsolvetime(self):
        return float(self._lib.ktn_get_solvetime(self._h))

    def numiters(self):
        return int(self._lib.ktn_numiters(self._h))

    def numcuts(self):
        return int(self._lib.ktn_numcuts(self._h))

    def setwarmstart(self, x):            # src/model.jl:335 -- ignores its input
        x = _f64(x)
        self._lib.ktn_setwarmstart(self._h, _p(x), len(x))
        return np.zeros(len(x))

    def set_blocks(self, col_offsets):
        """throughput mode: the loaded problem is block-diagonal with these column offsets (ktn_set_blocks)"""
        off = np.ascontiguousarray(col_offsets, dtype=np.int64)
        L.check(self._h, self._lib.ktn_set_blocks(self._h, len(off) - 1, _p(off, C.c_int64)))

    def optimize_blocks(self, cut_capacity=0):
        """optimize! of a block-diagonal batch with every instance's whole loop in its own workgroup (ktn_optimize_blocks)"""
        return STATUS_SYMBOLS[L.check(self._h, self._lib.ktn_optimize_blocks(self._h, int(cut_capacity)))]

    def stat(self, name):
        return float(self._lib.ktn_get_stat(self._h, name.encode()))

    # ---- LP introspection ---------------------------------------------------------------
    def lp_rows(self):
        m, nnz = int(self._lib.ktn_lp_num_rows(self._h)), int(self._lib.ktn_lp_nnz(self._h))
        rowptr, col = np.zeros(m + 1, dtype=np.int64), np.zeros(max(nnz, 1), dtype=np.int32)
        val, lo, hi = np.zeros(max(nnz, 1)), np.zeros(max(m, 1)), np.zeros(max(m, 1))
        L.check(self._h, self._lib.ktn_lp_get_rows(self._h, _p(rowptr, C.c_int64), _p(col, C.c_int32), _p(val), _p(lo),
                                                   _p(hi)))
        return rowptr, col[:nnz], val[:nnz], lo[:m], hi[:m]

    def lp_objective(self):
        c = np.zeros(self.num_var)
        c0 = C.c_double(0.0)
        L.check(self._h, self._lib.ktn_lp_get_objective(self._h, _p(c), len(c), C.byref(c0)))
        return c, c0.value

    def lp_duals(self):
        m = int(self._lib.ktn_lp_num_rows(self._h))
        y = np.zeros(max(m, 1))
        L.check(self._h, self._lib.ktn_lp_get_duals(self._h, _p(y), len(y)))
        return y[:m]

    def lp_solve(self, row_tol=1e-7, gap_tol=1e-7):
        st, it = C.c_int32(0), C.c_int64(0)
        L.check(self._h, self._lib.ktn_lp_solve(self._h, row_tol, gap_tol, C.byref(st), C.byref(it)))
        return STATUS_SYMBOLS[st.value], int(it.value)

    # ---- multi-GPU building blocks (include/katana_hip.h "multi-GPU building blocks") ------
    def lp_num_rows(self):
        return int(self._lib.ktn_lp_num_rows(self._h))

    def sweep_lp_point(self, f_tol):
        nv, mv = C.c_int64(0), C.c_double(0.0)
        L.check(self._h, self._lib.ktn_sweep_lp_point(self._h, f_tol, C.byref(nv), C.byref(mv)))
        return int(nv.value), float(mv.value)

    def objective_certificate(self, id_offset=0):
        """this handle's share of sum_i lambda_i * (signed residual of NL row i) at the last sweep's point (signed; the caller
        adds the shares of all handles and clamps at zero)"""
        v = C.c_double(0.0)
        L.check(self._h, self._lib.ktn_objective_certificate(self._h, int(id_offset), C.byref(v)))
        return float(v.value)

    def lp_rows_from(self, first_row):
        nr = self.lp_num_rows() - first_row
        nz = int(self._lib.ktn_lp_nnz_from(self._h, first_row))
        rowptr, col = np.zeros(nr + 1, dtype=np.int64), np.zeros(max(nz, 1), dtype=np.int32)
        val, lo, hi = np.zeros(max(nz, 1)), np.zeros(max(nr, 1)), np.zeros(max(nr, 1))
        L.check(self._h, self._lib.ktn_lp_get_rows_from(self._h, first_row, _p(rowptr, C.c_int64), _p(col, C.c_int32),
                                                        _p(val), _p(lo), _p(hi)))
        return rowptr, col[:nz], val[:nz], lo[:nr], hi[:nr]

    def lp_pack_rows_dev(self, first_row, id_offset, dev_ptr=0, cap=0):
        """(nrows, nnz) of the rows [first_row, M); with a device pointer the packed f64 block of those rows is written there
        (ktn_lp_pack_rows_dev: the device-resident half of the cut exchange)"""
        nr, nz = C.c_int64(0), C.c_int64(0)
        L.check(self._h, self._lib.ktn_lp_pack_rows_dev(self._h, first_row, id_offset, C.c_void_p(dev_ptr or None), cap,
                                                        C.byref(nr), C.byref(nz)))
        return int(nr.value), int(nz.value)

    def lp_append_packed_dev(self, nrows, nnz, dev_ptr):
        L.check(self._h, self._lib.ktn_lp_append_packed_dev(self._h, nrows, nnz, C.c_void_p(dev_ptr or None)))

    def lp_truncate(self, nrows):
        L.check(self._h, self._lib.ktn_lp_truncate(self._h, nrows))

    def lp_purge(self):
        """cut-pool purge between the LP solve and the sweep (as ktn_ecp_step does); returns the number of rows dropped"""
        n = C.c_int64(0)
        L.check(self._h, self._lib.ktn_lp_purge(self._h, C.byref(n)))
        return int(n.value)

    def lp_append_rows(self, rowptr, col, val, lo, hi, nl_id=None):
        """append rows; `nl_id` = global NL-row id of the row each cut belongs to (after lp_enable_global_lists)"""
        rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
        nr = len(rowptr) - 1
        if nr <= 0:
            return
        col = np.ascontiguousarray(col, dtype=np.int32)
        val, lo, hi = _f64(val), _f64(lo), _f64(hi)
        colp = col if len(col) else np.zeros(1, dtype=np.int32)
        valp = val if len(val) else np.zeros(1)
        if nl_id is None:
            idp = C.POINTER(C.c_int64)()
        else:
            nl_id = np.ascontiguousarray(nl_id, dtype=np.int64)
            assert len(nl_id) == nr
            idp = _p(nl_id, C.c_int64)
        L.check(self._h, self._lib.ktn_lp_append_rows_nl(self._h, nr, _p(rowptr, C.c_int64), _p(colp, C.c_int32), _p(valp),
                                                         _p(lo), _p(hi), idp))

    def set_cut_exchange(self, cb, first_nl_id):
        """ktn_set_cut_exchange: `cb` is an _lib.EXCHANGE_CB instance the caller keeps alive (None removes it)"""
        ptr = C.cast(cb, C.c_void_p) if cb is not None else None
        L.check(self._h, self._lib.ktn_set_cut_exchange(self._h, ptr, None, int(first_nl_id)))

    def lp_enable_global_lists(self, nl_total):
        L.check(self._h, self._lib.ktn_lp_enable_global_lists(self._h, int(nl_total)))

    def last_sweep_slots(self):
        """local NL slots of the cuts the last sweep appended, in row order"""
        n = C.c_int64(0)
        L.check(self._h, self._lib.ktn_last_sweep_slots(self._h, C.POINTER(C.c_int64)(), 0, C.byref(n)))
        out = np.zeros(max(int(n.value), 1), dtype=np.int64)
        if n.value:
            L.check(self._h, self._lib.ktn_last_sweep_slots(self._h, _p(out, C.c_int64), len(out), C.byref(n)))
        return out[:int(n.value)]

    def lp_pdhg_raw(self, x0, y0, eta, omega, iters):
        x0, y0 = _f64(x0), _f64(y0)
        m = int(self._lib.ktn_lp_num_rows(self._h))
        xo, yo = np.zeros(self.num_var), np.zeros(max(m, 1))
        y0p = y0 if len(y0) else np.zeros(1)
        L.check(self._h, self._lib.ktn_lp_pdhg_raw(self._h, _p(x0), _p(y0p), eta, omega, iters, _p(xo), _p(yo)))
        return xo, yo[:m]


class KatanaHipSeparator:
    """The first-order separator (KatanaFirstOrderSeparator, src/separators.jl:58-120) in batched,
    device-resident form.  precompute!/isconstrsat/gencut keep the reference's per-row meaning;
    `sweep` is the whole loop body of src/model.jl:272-283 in one call."""

    def __init__(self, model):
        self.m = model
        self.xstar = None

    def initialize(self):                         # src/separators.jl:81-107
        lib, h = self.m._lib, self.m._h
        mrows, nnz = int(lib.ktn_sep_num_constr(h)), int(lib.ktn_sep_jac_nnz(h))
        self.rowptr, self.col = np.zeros(mrows + 1, dtype=np.int64), np.zeros(max(nnz, 1), dtype=np.int32)
        L.check(h, lib.ktn_sep_get_structure(h, _p(self.rowptr, C.c_int64), _p(self.col, C.c_int32)))
        self.col = self.col[:nnz]
        self.num_constr, self.nnz = mrows, nnz
        self.sp_cols = [self.col[self.rowptr[i]:self.rowptr[i + 1]] for i in range(mrows)]

    def precompute(self, xstar):                  # src/separators.jl:111-116
        x = _f64(xstar)
        L.check(self.m._h, self.m._lib.ktn_sep_precompute(self.m._h, _p(x), len(x)))
        self.xstar = x
        self.g = np.zeros(max(self.num_constr, 1))
        self.jac = np.zeros(max(self.nnz, 1))
        L.check(self.m._h, self.m._lib.ktn_sep_get_g(self.m._h, _p(self.g), self.num_constr))
        L.check(self.m._h, self.m._lib.ktn_sep_get_jac(self.m._h, _p(self.jac), self.nnz))
        self.g, self.jac = self.g[:self.num_constr], self.jac[:self.nnz]

    def isconstrsat(self, i, lb, ub, f_tol):      # src/separators.jl:120
        return bool(L.check(self.m._h, self.m._lib.ktn_sep_isconstrsat(self.m._h, i, lb, ub, f_tol)))

    def gencut(self, xstar, bounds, i):           # src/separators.jl:118 -> (cols, coefs, constant)
        cap = int(self.rowptr[i + 1] - self.rowptr[i])
        cols, coefs = np.zeros(max(cap, 1), dtype=np.int32), np.zeros(max(cap, 1))
        nnz, const = C.c_int64(cap), C.c_double(0.0)
        L.check(self.m._h, self.m._lib.ktn_sep_gencut(self.m._h, i, _p(cols, C.c_int32), _p(coefs), C.byref(nnz),
                                                      C.byref(const)))
        return cols[:nnz.value], coefs[:nnz.value], const.value

    def sweep(self, f_tol):
        nv, mv = C.c_int64(0), C.c_double(0.0)
        L.check(self.m._h, self.m._lib.ktn_sep_sweep(self.m._h, f_tol, C.byref(nv), C.byref(mv)))
        return int(nv.value), float(mv.value)


# ---- src/util.jl ---------------------------------------------------------------------------
def getKatanaModel(m):                            # src/util.jl:3-5
    return m.internal_model if hasattr(m, "internal_model") else m


def getKatanaCuts(m):
    """src/util.jl:16-34: table M x (N+2): coefficients, constant, direction (-1 '<=', +1 '>=')
    of every LP row (the reference lists the rows recorded under :VisData)."""
    m = getKatanaModel(m)
    rowptr, col, val, lo, hi = m.lp_rows()
    M, N = len(lo), m.num_var
    table = np.zeros((M, N + 2))
    for i in range(M):
        for e in range(rowptr[i], rowptr[i + 1]):
            table[i, col[e]] += val[e]
        if np.isfinite(hi[i]) and not np.isfinite(lo[i]):
            table[i, N], table[i, N + 1] = hi[i], -1
        elif np.isfinite(lo[i]) and not np.isfinite(hi[i]):
            table[i, N], table[i, N + 1] = lo[i], 1
        else:
            raise AssertionError("range or free row: not an inequality (src/util.jl:27-29)")
    return table


def getKatanaSols(m):                             # src/util.jl:36
    m = getKatanaModel(m)
    k = int(m._lib.ktn_num_lp_sols(m._h))
    out = []
    for i in range(k):
        x = np.zeros(m.num_var)
        L.check(m._h, m._lib.ktn_get_lp_sol(m._h, i, _p(x), len(x)))
        out.append(x)
    return out
