"""S-expression evaluation with first derivatives -- oracle side (TEST INFRASTRUCTURE).

Stands in for JuMP's NLPEvaluator (third-party, not in /root/reference; call
sites src/separators.jl:88,92,112,113 and src/nlpeval.jl:31,35,37,43,50,60):
given a closed-form expression it returns the value and the partial derivative
with respect to every variable that occurs in it.

The expression format is the neutral nested-list form stored in
tests/golden/kat_models.json:

    number | ["var", j] | [op, arg, ...]
    op in  + - neg * / ^ exp log sqrt sin cos

`^` takes a numeric constant exponent.  Forward-mode AD with a sparse gradient
dict; deliberately a different algorithm from the product's postfix tape with
reverse sweep (katana.jl_amd/expr.py, csrc/tape_eval.hip) so that the two do
not share bugs.  Domain errors follow JuMP's NaNMath convention: log/sqrt/pow
outside their domain give NaN, division by zero gives +-Inf (IEEE).
"""
import math

import numpy as np

_NAN = float("nan")


def _log(v):
    if v < 0 or v != v:
        return _NAN
    if v == 0:
        return -math.inf
    return math.log(v)


def _sqrt(v):
    if v < 0 or v != v:
        return _NAN
    return math.sqrt(v)


def _div(a, b):
    with np.errstate(all="ignore"):
        return float(np.float64(a) / np.float64(b))


def _mul(a, b):
    with np.errstate(all="ignore"):
        return float(np.float64(a) * np.float64(b))


def _pow(a, p):
    with np.errstate(all="ignore"):
        try:
            r = np.float64(a) ** np.float64(p)
        except Exception:
            return _NAN
        return float(r)


def _exp(v):
    try:
        return math.exp(v)
    except OverflowError:
        return math.inf


def _axpy(dst, s, src):
    """dst += s * src for sparse gradient dicts (IEEE semantics: 0*inf = nan)."""
    for k, v in src.items():
        dst[k] = dst.get(k, 0.0) + _mul(s, v)


def eval_grad(e, x):
    """Return (value, {var_index: partial}) of s-expression `e` at point `x`."""
    if isinstance(e, (int, float)):
        return float(e), {}
    op = e[0]
    if op == "var":
        j = int(e[1])
        return float(x[j]), {j: 1.0}
    if op == "+":
        val, g = 0.0, {}
        for a in e[1:]:
            v, ga = eval_grad(a, x)
            val += v
            _axpy(g, 1.0, ga)
        return val, g
    if op == "-":
        v1, g1 = eval_grad(e[1], x)
        v2, g2 = eval_grad(e[2], x)
        g = dict(g1)
        _axpy(g, -1.0, g2)
        return v1 - v2, g
    if op == "neg":
        v, g1 = eval_grad(e[1], x)
        g = {}
        _axpy(g, -1.0, g1)
        return -v, g
    if op == "*":
        val, g = eval_grad(e[1], x)
        g = dict(g)
        for a in e[2:]:
            v2, g2 = eval_grad(a, x)
            ng = {}
            _axpy(ng, v2, g)
            _axpy(ng, val, g2)
            val, g = _mul(val, v2), ng
        return val, g
    if op == "/":
        v1, g1 = eval_grad(e[1], x)
        v2, g2 = eval_grad(e[2], x)
        q = _div(v1, v2)
        g = {}
        _axpy(g, _div(1.0, v2), g1)
        _axpy(g, -_div(q, v2), g2)
        return q, g
    if op == "^":
        v, g1 = eval_grad(e[1], x)
        p = float(e[2])
        val = _pow(v, p)
        if p == 2.0:
            d = 2.0 * v
        elif p == 1.0:
            d = 1.0
        else:
            d = _mul(p, _pow(v, p - 1.0))
        g = {}
        _axpy(g, d, g1)
        return val, g
    v, g1 = eval_grad(e[1], x)
    if op == "exp":
        val = _exp(v)
        d = val
    elif op == "log":
        val = _log(v)
        d = _div(1.0, v)          # Calculus rule 1/x, finite for x < 0 (value is NaN there)
    elif op == "sqrt":
        val = _sqrt(v)
        d = _div(0.5, val) if val == val else _NAN
    elif op == "sin":
        val, d = math.sin(v), math.cos(v)
    elif op == "cos":
        val, d = math.cos(v), -math.sin(v)
    else:
        raise ValueError("unknown op %r" % (op,))
    g = {}
    _axpy(g, d, g1)
    return val, g


def variables(e, acc=None):
    """Sorted list of variable indices that occur in `e`."""
    top = acc is None
    if top:
        acc = set()
    if isinstance(e, list):
        if e[0] == "var":
            acc.add(int(e[1]))
        else:
            for a in e[1:]:
                variables(a, acc)
    return sorted(acc) if top else None


def is_affine(e):
    """True when `e` is syntactically affine (what JuMP reports through
    isconstrlinear / isobjlinear for @constraint/@objective with AffExpr)."""
    if isinstance(e, (int, float)):
        return True
    op = e[0]
    if op == "var":
        return True
    if op in ("+", "-", "neg"):
        return all(is_affine(a) for a in e[1:])
    if op == "*":
        nonconst = [a for a in e[1:] if variables(a)]
        return len(nonconst) <= 1 and all(is_affine(a) for a in nonconst)
    if op == "/":
        return is_affine(e[1]) and not variables(e[2])
    if op == "^":
        return (not variables(e[1])) or float(e[2]) == 1.0 and is_affine(e[1])
    return not variables(e)
