"""Expression graphs -> device tapes.

Stands where JuMP's expression graph (`MathProgBase.constr_expr` / `obj_expr`, :ExprGraph)
stands for the reference: the host describes every nonlinear row as an expression, this
module flattens it to the postfix tape the HIP tape interpreter evaluates (opcodes
KTN_OP_* of include/katana_hip.h), finds the row's variables (= its Jacobian structure,
`jac_structure`, src/separators.jl:92) and recognises affine rows (`isconstrlinear` /
`isobjlinear`, src/model.jl:116,125).
"""
import numpy as np

from . import _lib as L

_UNARY = {"neg": L.OP_NEG, "exp": L.OP_EXP, "log": L.OP_LOG, "sqrt": L.OP_SQRT, "sin": L.OP_SIN, "cos": L.OP_COS}
_BINARY = {"+": L.OP_ADD, "-": L.OP_SUB, "*": L.OP_MUL, "/": L.OP_DIV}


class Expr:
    """Immutable expression node: ("const", c) | ("var", j) | (op, args...)."""
    __slots__ = ("op", "args")

    def __init__(self, op, *args):
        self.op = op
        self.args = args

    # -- operator overloading -------------------------------------------------
    @staticmethod
    def wrap(o):
        return o if isinstance(o, Expr) else Expr("const", float(o))

    def __add__(self, o): return Expr("+", self, Expr.wrap(o))
    def __radd__(self, o): return Expr("+", Expr.wrap(o), self)
    def __sub__(self, o): return Expr("-", self, Expr.wrap(o))
    def __rsub__(self, o): return Expr("-", Expr.wrap(o), self)
    def __mul__(self, o): return Expr("*", self, Expr.wrap(o))
    def __rmul__(self, o): return Expr("*", Expr.wrap(o), self)
    def __truediv__(self, o): return Expr("/", self, Expr.wrap(o))
    def __rtruediv__(self, o): return Expr("/", Expr.wrap(o), self)
    def __neg__(self): return Expr("neg", self)
    def __pow__(self, p): return Expr("^", self, float(p))

    # comparison operators build constraints (jump_like.Model.constraint)
    def __le__(self, o): return ("<=", self, Expr.wrap(o))
    def __ge__(self, o): return (">=", self, Expr.wrap(o))

    # -- analysis ---------------------------------------------------------------
    def variables(self):
        acc = set()
        stack = [self]
        while stack:
            e = stack.pop()
            if e.op == "var":
                acc.add(e.args[0])
            elif e.op != "const":
                stack.extend(a for a in e.args if isinstance(a, Expr))
        return sorted(acc)

    def affine(self):
        """Return (coef dict, constant) when the expression is affine, else None."""
        op = self.op
        if op == "const":
            return {}, self.args[0]
        if op == "var":
            return {self.args[0]: 1.0}, 0.0
        if op in ("+", "-"):
            a, b = self.args[0].affine(), self.args[1].affine()
            if a is None or b is None:
                return None
            s = 1.0 if op == "+" else -1.0
            co = dict(a[0])
            for k, v in b[0].items():
                co[k] = co.get(k, 0.0) + s * v
            return co, a[1] + s * b[1]
        if op == "neg":
            a = self.args[0].affine()
            return None if a is None else ({k: -v for k, v in a[0].items()}, -a[1])
        if op == "*":
            a, b = self.args[0].affine(), self.args[1].affine()
            if a is None or b is None:
                return None
            if not a[0]:
                return {k: a[1] * v for k, v in b[0].items()}, a[1] * b[1]
            if not b[0]:
                return {k: b[1] * v for k, v in a[0].items()}, a[1] * b[1]
            return None
        if op == "/":
            a, b = self.args[0].affine(), self.args[1].affine()
            if a is None or b is None or b[0]:
                return None
            return {k: v / b[1] for k, v in a[0].items()}, a[1] / b[1]
        if op == "^":
            a = self.args[0].affine()
            if a is not None and not a[0]:
                return {}, a[1] ** self.args[1]
            if a is not None and self.args[1] == 1.0:
                return a
            return None
        a = self.args[0].affine()          # unary function of a constant
        if a is not None and not a[0]:
            f = {"exp": np.exp, "log": np.log, "sqrt": np.sqrt, "sin": np.sin, "cos": np.cos}[op]
            return {}, float(f(a[1]))
        return None

    def tape(self):
        """Postfix tape: (ops int32[], args float64[])."""
        ops, args = [], []

        def emit(e):
            if e.op == "const":
                ops.append(L.OP_CONST); args.append(e.args[0])
            elif e.op == "var":
                ops.append(L.OP_VAR); args.append(float(e.args[0]))
            elif e.op in _BINARY:
                emit(e.args[0]); emit(e.args[1])
                ops.append(_BINARY[e.op]); args.append(0.0)
            elif e.op == "^":
                emit(e.args[0])
                ops.append(L.OP_POWC); args.append(float(e.args[1]))
            elif e.op in _UNARY:
                emit(e.args[0])
                ops.append(_UNARY[e.op]); args.append(0.0)
            else:
                raise ValueError("Unsupported operator %r" % (e.op,))
        emit(self)
        return np.asarray(ops, dtype=np.int32), np.asarray(args, dtype=np.float64)


def var(j): return Expr("var", int(j))
def const(c): return Expr("const", float(c))
def exp(a): return Expr("exp", Expr.wrap(a))
def log(a): return Expr("log", Expr.wrap(a))
def sqrt(a): return Expr("sqrt", Expr.wrap(a))
def sin(a): return Expr("sin", Expr.wrap(a))
def cos(a): return Expr("cos", Expr.wrap(a))


def from_sexpr(s):
    """Nested-list form (tests/golden/kat_models.json) -> Expr."""
    if isinstance(s, (int, float)):
        return const(s)
    op = s[0]
    if op == "var":
        return var(s[1])
    if op == "^":
        return from_sexpr(s[1]) ** float(s[2])
    if op in ("+", "*"):
        e = from_sexpr(s[1])
        for a in s[2:]:
            e = Expr(op, e, from_sexpr(a))
        return e
    if op in ("-", "/"):
        return Expr(op, from_sexpr(s[1]), from_sexpr(s[2]))
    if op in _UNARY:
        return Expr(op, from_sexpr(s[1]))
    raise ValueError("unknown op %r" % (op,))
