"""Device-evaluable NLP descriptions: what `d::AbstractNLPEvaluator` is to the reference's
loadproblem! (src/model.jl:86).  An NLPDescription owns the numpy arrays behind a
`ktn_nlp_desc` (include/katana_hip.h) and answers the structural queries the reference asks
its evaluator (jac_structure, isconstrlinear, isobjlinear); values and derivatives are
computed on the device only."""
import ctypes as C

import numpy as np

from . import _lib as L
from .expr import Expr


def _ptr(a, ctype):
    return a.ctypes.data_as(C.POINTER(ctype)) if a is not None and a.size else C.POINTER(ctype)()


class NLPDescription:
    def __init__(self, num_var, rowptr, col, row_kind, row_linear, rconst, atom_kind, p0, p1,
                 tape_ptr=None, tape_op=None, tape_arg=None,
                 obj_linear=True, obj_kind=L.ROW_SEP, obj_col=None, obj_atom_kind=None, obj_p0=None, obj_p1=None,
                 obj_const=0.0, obj_tape_op=None, obj_tape_arg=None):
        i64, i32, u8, f64 = np.int64, np.int32, np.uint8, np.float64
        self.num_var = int(num_var)
        self.rowptr = np.ascontiguousarray(rowptr, dtype=i64)
        self.num_constr = len(self.rowptr) - 1
        self.col = np.ascontiguousarray(col, dtype=i32)
        nnz = len(self.col)
        self.row_kind = np.ascontiguousarray(row_kind, dtype=u8)
        self.row_linear = np.ascontiguousarray(row_linear, dtype=u8)
        self.rconst = np.ascontiguousarray(rconst, dtype=f64)
        self.atom_kind = np.ascontiguousarray(atom_kind if atom_kind is not None else np.zeros(nnz), dtype=u8)
        self.p0 = np.ascontiguousarray(p0 if p0 is not None else np.zeros(nnz), dtype=f64)
        self.p1 = np.ascontiguousarray(p1 if p1 is not None else np.zeros(nnz), dtype=f64)
        self.tape_ptr = np.ascontiguousarray(tape_ptr if tape_ptr is not None else np.zeros(self.num_constr + 1), dtype=i64)
        self.tape_op = np.ascontiguousarray(tape_op if tape_op is not None else [], dtype=i32)
        self.tape_arg = np.ascontiguousarray(tape_arg if tape_arg is not None else [], dtype=f64)
        self.obj_linear = bool(obj_linear)
        self.obj_kind = int(obj_kind)
        self.obj_col = np.ascontiguousarray(obj_col if obj_col is not None else [], dtype=i32)
        k = len(self.obj_col)
        self.obj_atom_kind = np.ascontiguousarray(obj_atom_kind if obj_atom_kind is not None else np.zeros(k), dtype=u8)
        self.obj_p0 = np.ascontiguousarray(obj_p0 if obj_p0 is not None else np.zeros(k), dtype=f64)
        self.obj_p1 = np.ascontiguousarray(obj_p1 if obj_p1 is not None else np.zeros(k), dtype=f64)
        self.obj_const = float(obj_const)
        self.obj_tape_op = np.ascontiguousarray(obj_tape_op if obj_tape_op is not None else [], dtype=i32)
        self.obj_tape_arg = np.ascontiguousarray(obj_tape_arg if obj_tape_arg is not None else [], dtype=f64)
        assert len(self.row_kind) == self.num_constr and len(self.row_linear) == self.num_constr
        assert len(self.rconst) == self.num_constr and len(self.p0) == nnz and len(self.p1) == nnz

    # ---- the structural part of the MathProgBase evaluator interface --------------------
    def isobjlinear(self):
        return self.obj_linear

    def isconstrlinear(self, i):
        return bool(self.row_linear[i])

    def jac_structure(self):
        rows = np.repeat(np.arange(self.num_constr), np.diff(self.rowptr))
        return rows, self.col

    def features_available(self):
        return ["Grad", "Jac"]

    def c_struct(self):
        d = L.KtnNlpDesc()
        d.num_var, d.num_constr = self.num_var, self.num_constr
        d.rowptr, d.col = _ptr(self.rowptr, C.c_int64), _ptr(self.col, C.c_int32)
        d.row_kind, d.row_linear = _ptr(self.row_kind, C.c_uint8), _ptr(self.row_linear, C.c_uint8)
        d.rconst, d.atom_kind = _ptr(self.rconst, C.c_double), _ptr(self.atom_kind, C.c_uint8)
        d.p0, d.p1 = _ptr(self.p0, C.c_double), _ptr(self.p1, C.c_double)
        d.tape_ptr, d.tape_op, d.tape_arg = (_ptr(self.tape_ptr, C.c_int64), _ptr(self.tape_op, C.c_int32),
                                             _ptr(self.tape_arg, C.c_double))
        d.obj_linear, d.obj_kind, d.obj_nnz = int(self.obj_linear), self.obj_kind, len(self.obj_col)
        d.obj_col, d.obj_atom_kind = _ptr(self.obj_col, C.c_int32), _ptr(self.obj_atom_kind, C.c_uint8)
        d.obj_p0, d.obj_p1, d.obj_const = _ptr(self.obj_p0, C.c_double), _ptr(self.obj_p1, C.c_double), self.obj_const
        d.obj_tape_len = len(self.obj_tape_op)
        d.obj_tape_op, d.obj_tape_arg = _ptr(self.obj_tape_op, C.c_int32), _ptr(self.obj_tape_arg, C.c_double)
        return d


def SeparableNLP(inst):
    """NLPDescription of a katana_jl_amd.instances.SeparableInstance (or any object with the
    same array attributes)."""
    m = len(inst.rowptr) - 1
    rp = np.asarray(inst.rowptr)
    kind = np.ascontiguousarray(inst.kind, dtype=np.uint8)                       # a row is nonlinear iff one of its atoms is (LIN = 0)
    nonlin = np.zeros(m, dtype=np.uint8)
    nz = np.flatnonzero(np.diff(rp) > 0)                                         # reduce over the non-empty rows only: reduceat
    if len(nz):                                                                  # runs a slice up to the NEXT start, so an empty
        nonlin[nz] = np.maximum.reduceat(kind, rp[:-1][nz])                      # row in between (or at the end) must not be a start
    return NLPDescription(
        inst.n, inst.rowptr, inst.col, np.zeros(m, dtype=np.uint8), (nonlin == 0).astype(np.uint8), inst.rconst,
        inst.kind, inst.p0, inst.p1,
        obj_linear=bool(np.all(np.asarray(inst.obj_kind) == L.ATOM_LIN)), obj_kind=L.ROW_SEP,
        obj_col=inst.obj_col, obj_atom_kind=inst.obj_kind, obj_p0=inst.obj_p0, obj_p1=inst.obj_p1,
        obj_const=inst.obj_const)


def ExprNLP(num_var, objective, constraints, constr_linear=None, obj_linear=None):
    """NLPDescription from expressions.  Affine rows become separable rows of LIN atoms (their
    tangent at the origin is the row itself, src/model.jl:115-118); every other row becomes a
    tape row."""
    rowptr, col, akind, p0, p1, rkind, rlin, rconst = [0], [], [], [], [], [], [], []
    tptr, top, targ = [0], [], []
    for i, e in enumerate(constraints):
        e = Expr.wrap(e)
        aff = e.affine()
        declared = constr_linear[i] if constr_linear is not None else (aff is not None)
        if aff is not None:
            co, c0 = aff
            for j in sorted(co):
                col.append(j); akind.append(L.ATOM_LIN); p0.append(co[j]); p1.append(0.0)
            rkind.append(L.ROW_SEP); rconst.append(c0)
        else:
            for j in e.variables():
                col.append(j); akind.append(0); p0.append(0.0); p1.append(0.0)
            o, a = e.tape()
            top.extend(o.tolist()); targ.extend(a.tolist())
            rkind.append(L.ROW_TAPE); rconst.append(0.0)
        rlin.append(1 if declared else 0)
        rowptr.append(len(col)); tptr.append(len(top))
    objective = Expr.wrap(objective)
    oaff = objective.affine()
    is_lin = obj_linear if obj_linear is not None else (oaff is not None)
    kw = {}
    if oaff is not None:
        co, c0 = oaff
        js = sorted(co)
        kw = dict(obj_kind=L.ROW_SEP, obj_col=js, obj_atom_kind=np.zeros(len(js)), obj_p0=[co[j] for j in js],
                  obj_p1=np.zeros(len(js)), obj_const=c0)
    else:
        o, a = objective.tape()
        kw = dict(obj_kind=L.ROW_TAPE, obj_tape_op=o, obj_tape_arg=a)
    return NLPDescription(num_var, rowptr, col, rkind, rlin, rconst, akind, p0, p1, tptr, top, targ,
                          obj_linear=is_lin, **kw)


class CallbackNLP(NLPDescription):
    """The fallback for evaluators that cannot hand over expressions (SURVEY.md section 8b "Evaluator consumed"): wraps an
    object with the MathProgBase.AbstractNLPEvaluator methods the reference calls -- `jac_structure()`,
    `eval_g(g, x)`, `eval_jac_g(J, x)`, `eval_f(x)`, `eval_grad_f(grad, x)`, `isconstrlinear(i)`, `isobjlinear()`
    (src/separators.jl:88-113, src/model.jl:116,125,159, src/nlpeval.jl:31-63) -- and declares every row KTN_ROW_HOST.
    Values and derivatives are then computed by that object on the host, once per sweep; constraint checks, cuts and the
    LP stay on the device.  The COO structure is converted to CSR exactly as initialize! does (src/separators.jl:92-104:
    row by row, entries in COO order)."""

    def __init__(self, evaluator, num_var, num_constr):
        if hasattr(evaluator, "initialize"):
            evaluator.initialize(["Grad", "Jac"])                      # src/separators.jl:88
        rows, cols = evaluator.jac_structure()
        rows = np.asarray(rows, dtype=np.int64); cols = np.asarray(cols, dtype=np.int64)
        order = np.argsort(rows, kind="stable")
        rowptr = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=num_constr))])
        nnz = len(rows)
        lin = [1 if evaluator.isconstrlinear(i) else 0 for i in range(num_constr)]
        super().__init__(num_var, rowptr, cols[order], np.full(num_constr, L.ROW_HOST), lin, np.zeros(num_constr),
                         None, None, None, obj_linear=bool(evaluator.isobjlinear()), obj_kind=L.ROW_HOST)
        self.evaluator = evaluator
        n, m = int(num_var), int(num_constr)

        def rows_cb(_user, xp, gp, jp):
            try:
                x = np.ctypeslib.as_array(xp, (n,)).copy()
                g = np.zeros(m); J = np.zeros(nnz)
                with np.errstate(all="ignore"):
                    evaluator.eval_g(g, x)                              # src/separators.jl:113
                    evaluator.eval_jac_g(J, x)                          # src/separators.jl:112
                if m:
                    np.ctypeslib.as_array(gp, (m,))[:] = g
                if nnz:
                    np.ctypeslib.as_array(jp, (nnz,))[:] = J[order]
                return 0
            except Exception:                                           # never unwind through the C ABI
                return 1

        def obj_cb(_user, xp, fp, gradp):
            try:
                x = np.ctypeslib.as_array(xp, (n,)).copy()
                grad = np.zeros(n)
                with np.errstate(all="ignore"):
                    f = evaluator.eval_f(x)                             # src/nlpeval.jl:35
                    evaluator.eval_grad_f(grad, x)                      # src/nlpeval.jl:37-41
                fp[0] = float(f)
                if n:
                    np.ctypeslib.as_array(gradp, (n,))[:] = grad
                return 0
            except Exception:
                return 1

        self._rows_cb = L.EVAL_ROWS_CB(rows_cb)       # keep the trampolines alive as long as the description
        self._obj_cb = L.EVAL_OBJ_CB(obj_cb)

    def c_struct(self):
        d = super().c_struct()
        d.eval_rows = C.cast(self._rows_cb, C.c_void_p)
        d.eval_obj = C.cast(self._obj_cb, C.c_void_p)
        d.eval_user = None
        return d
