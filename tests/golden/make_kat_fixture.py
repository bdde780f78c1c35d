#!/usr/bin/env python3
"""Generate tests/golden/kat_models.json -- the reference's known-answer tests as DATA.

Every active @testset of the reference's test-suite (test/basic.jl, lpqp.jl,
2d.jl, 3d.jl, misc.jl; table in SURVEY.md section 4.1) is restated here as a
model description (variables with bounds, sense, objective and constraint
expressions in the nested-list form of oracle/sexpr.py) together with the
expected status / objective / solution and the tolerances of
test/runtests.jl:16-20.  Test cases the reference itself disables by comment
(105_02/03, 106_*, 108_02-04, 109_*, 204_01, 206_01, 203 norm form) are not
included.  Expected values are copied as numbers from the cited lines; the two
test/basic.jl cases whose expectation is a live Ipopt run use the closed forms
of the identical models 001_01 / 101_01.

Constraint normalisation: `lhs <= rhs` becomes g = lhs - rhs with bounds
(-inf, 0); `>=` gives (0, inf).  (JuMP keeps constants on the bound side for
linear rows; the LP row that results from the tangent at the origin,
src/model.jl:115-118, is identical.)

Run:  python tests/golden/make_kat_fixture.py   (writes kat_models.json next to it)
"""
import json
import math
import os

INF = float("inf")


class E:
    """expression DSL -> nested lists"""

    def __init__(self, s):
        self.s = s

    @staticmethod
    def w(o):
        return o.s if isinstance(o, E) else float(o)

    def __add__(self, o): return E(["+", self.s, E.w(o)])
    def __radd__(self, o): return E(["+", E.w(o), self.s])
    def __sub__(self, o): return E(["-", self.s, E.w(o)])
    def __rsub__(self, o): return E(["-", E.w(o), self.s])
    def __mul__(self, o): return E(["*", self.s, E.w(o)])
    def __rmul__(self, o): return E(["*", E.w(o), self.s])
    def __truediv__(self, o): return E(["/", self.s, E.w(o)])
    def __rtruediv__(self, o): return E(["/", E.w(o), self.s])
    def __neg__(self): return E(["neg", self.s])
    def __pow__(self, p): return E(["^", self.s, float(p)])


def exp(a): return E(["exp", E.w(a)])
def log(a): return E(["log", E.w(a)])
def sqrt(a): return E(["sqrt", E.w(a)])


def esum(items):
    items = list(items)
    return E(["+"] + [E.w(i) for i in items]) if len(items) > 1 else items[0]


MODELS = []


class Model:
    def __init__(self, id_, ref, sense="Min"):
        self.d = {"id": id_, "ref": ref, "vars": [], "sense": sense, "objective": 0.0,
                  "objective_linear": True, "constraints": [], "expect": {}}
        MODELS.append(self.d)

    def var(self, name, lb=-INF, ub=INF):
        self.d["vars"].append({"name": name, "lb": lb, "ub": ub})
        return E(["var", len(self.d["vars"]) - 1])

    def objective(self, sense, e, linear):
        self.d["sense"] = sense
        self.d["objective"] = E.w(e)
        self.d["objective_linear"] = linear

    def con(self, lhs, op, rhs, linear=False):
        lhs = lhs if isinstance(lhs, E) else E(float(lhs))
        g = E.w(lhs - rhs)
        lb, ub = (-INF, 0.0) if op == "<=" else (0.0, INF)
        self.d["constraints"].append({"expr": g, "lb": lb, "ub": ub, "linear": linear})

    def expect(self, obj, x=None, status="Optimal", obj_atol=1e-6, obj_rtol=1e-6):
        self.d["expect"] = {"status": status, "obj": obj, "obj_atol": obj_atol, "obj_rtol": obj_rtol,
                            "x": x, "sol_atol": 1e-3, "sol_rtol": 1e-3}


def five_linear(m, x, y):      # test/basic.jl:11-15 == test/lpqp.jl:14-18
    m.con(x + y, "<=", 5, linear=True)
    m.con(2 * x - y, "<=", 3, linear=True)
    m.con(3 * x + 9 * y, ">=", -10, linear=True)
    m.con(10 * x - y, ">=", -20, linear=True)
    m.con(-x + 2 * y, "<=", 8, linear=True)


def six_linear(m, x, y):       # test/lpqp.jl:60-65
    m.con(1 * x - 3 * y, "<=", 3, linear=True)
    m.con(1 * x - 5 * y, "<=", 0, linear=True)
    m.con(3 * x + 5 * y, ">=", 15, linear=True)
    m.con(7 * x + 2 * y, ">=", 20, linear=True)
    m.con(9 * x + 1 * y, ">=", 20, linear=True)
    m.con(3 * x + 7 * y, ">=", 17, linear=True)


r2 = math.sqrt(2.0)

# ---- test/basic.jl ---------------------------------------------------------
m = Model("basic_1", "test/basic.jl:4-42"); x = m.var("x"); y = m.var("y")
m.objective("Min", x, True); five_linear(m, x, y)
m.expect(-2.0430107680954848, [-2.0430107680954848, -0.4301075068564087])

m = Model("basic_2", "test/basic.jl:44-78"); x = m.var("x", -2, 2); y = m.var("y", -2, 2)
m.objective("Min", -x - y, True); m.con(x ** 2 + y ** 2, "<=", 1.0)
m.expect(-2 / r2, [1 / r2, 1 / r2])

m = Model("basic_3", "test/basic.jl:80-102"); x = m.var("x"); y = m.var("y")
m.objective("Min", (x - 1) ** 2 + (y - 2) ** 2, False); five_linear(m, x, y)
m.expect(0.0, [1.0, 2.0])

# ---- test/lpqp.jl ----------------------------------------------------------
m = Model("001_01", "test/lpqp.jl:7-27"); x = m.var("x"); y = m.var("y")
m.objective("Min", x, True); five_linear(m, x, y)
m.expect(-2.0430107680954848, [-2.0430107680954848, -0.4301075068564087])

m = Model("001_02", "test/lpqp.jl:30-50"); x = m.var("x"); y = m.var("y")
m.objective("Min", (x - 1) ** 2 + (y - 2) ** 2, False); five_linear(m, x, y)
m.expect(0.0, [1.0, 2.0])

m = Model("002_01", "test/lpqp.jl:53-73"); x = m.var("x"); y = m.var("y")
m.objective("Min", x + y, True); six_linear(m, x, y)
m.expect(3.9655172067026196, [2.4137930845761546, 1.5517241221264648])

m = Model("002_02", "test/lpqp.jl:76-97"); x = m.var("x"); y = m.var("y")
m.objective("Min", (x - 3) ** 2 + (y - 2) ** 2, False); six_linear(m, x, y)
m.expect(0.0, [3.0, 2.0])

# ---- test/2d.jl ------------------------------------------------------------
for id_, ref, sense, obj, eobj, ex in [
        ("101_01", "test/2d.jl:5-20", "Min", lambda x, y: -x - y, -2 / r2, [1 / r2, 1 / r2]),
        ("101_02", "test/2d.jl:23-38", "Min", lambda x, y: -x, -1.0, [1.0, 0.0]),
        ("101_03", "test/2d.jl:41-56", "Max", lambda x, y: 1 * x, 1.0, [1.0, 0.0])]:
    m = Model(id_, ref); x = m.var("x", -2, 2); y = m.var("y", -2, 2)
    m.objective(sense, obj(x, y), True); m.con(x ** 2 + y ** 2, "<=", 1.0)
    m.expect(eobj, ex)

for id_, ref, sense, obj, lin, eobj, ex in [
        ("102_01", "test/2d.jl:60-76", "Min", lambda x, y: -x, True, -0.974165743715913,
         [0.974165743715913, 0.2258342542139504]),
        ("102_02", "test/2d.jl:79-96", "Min", lambda x, y: x + y, True, 1.2, None),
        ("102_03", "test/2d.jl:99-115", "Max", lambda x, y: x + y, True, 2 / r2, [1 / r2, 1 / r2]),
        ("102_04", "test/2d.jl:118-134", "Min", lambda x, y: x ** 2 + y ** 2, False, 0.72, [0.6, 0.6]),
        ("102_05", "test/2d.jl:137-153", "Min", lambda x, y: (x - 0.65) ** 2 + (y - 0.65) ** 2, False, 0.0,
         [0.65, 0.65])]:
    m = Model(id_, ref); x = m.var("x"); y = m.var("y")
    m.objective(sense, obj(x, y), lin)
    m.con(x ** 2 + y ** 2, "<=", 1.0)
    m.con(x + y, ">=", 1.2, linear=True)
    m.expect(eobj, ex)

for id_, ref, obj, eobj, ex in [
        ("103_01", "test/2d.jl:157-173", lambda x, y: 1 * y, 0.0, [0.0, 0.0]),
        ("103_02", "test/2d.jl:176-192", lambda x, y: -y, -1.0, [0.0, 1.0]),
        ("103_03", "test/2d.jl:195-211", lambda x, y: -x - y, -5 / 4, [2 / 4, 3 / 4]),
        ("103_04", "test/2d.jl:214-230", lambda x, y: x + y, -1 / 4, [-2 / 4, 1 / 4]),
        ("103_05", "test/2d.jl:233-249", lambda x, y: -x, -1 / r2, [1 / r2, 1 / 2])]:
    m = Model(id_, ref); x = m.var("x"); y = m.var("y")
    m.objective("Min", obj(x, y), True)
    m.con(x ** 2, "<=", y)
    m.con(-(x ** 2) + 1, ">=", y)
    m.expect(eobj, ex)

m = Model("104_01", "test/2d.jl:253-271"); x = m.var("x"); y = m.var("y")
m.objective("Min", -x, True)
m.con(x ** 2, "<=", y); m.con(-(x ** 2) + 1, ">=", y); m.con(x ** 2 + (y - 0.5) ** 2, "<=", 1.0)
m.expect(-1 / r2, [1 / r2, 1 / 2])

for id_, ref, obj, eobj, ex in [
        ("105_01", "test/2d.jl:275-291", lambda x, y: -x - y, -4.176004405036646,
         [2.687422019398147, 1.488582385638499]),
        ("105_04", "test/2d.jl:338-354", lambda x, y: -x + y, -3 / 2, [2.0, 1 / 2])]:
    m = Model(id_, ref); x = m.var("x"); y = m.var("y")
    m.objective("Min", obj(x, y), True)
    m.con(exp(x - 2.0) - 0.5, "<=", y)
    m.con(log(x) + 0.5, ">=", y)
    m.expect(eobj, ex)

for id_, ref, a, eobj, ex in [
        ("107_01", "test/2d.jl:405-420", 0.5, 0.0, [0.5, 0.5]),
        ("107_02", "test/2d.jl:423-438", 1.0, 0.17157287363083387, [1 / r2, 1 / r2]),
        ("107_03", "test/2d.jl:441-456", 1.0, 0.17157287363083387, [1 / r2, 1 / r2])]:
    m = Model(id_, ref); x = m.var("x"); y = m.var("y")
    m.objective("Min", (x - a) ** 2 + (y - a) ** 2, False)
    m.con(x ** 2 + y ** 2, "<=", 1)
    m.expect(eobj, ex)

m = Model("108_01", "test/2d.jl:460-476"); x = m.var("x", 0.0, INF); y = m.var("y", 0.0, INF)
m.objective("Min", (x - 1.0) ** 2 + (y - 0.75) ** 2, False)
m.con(2 * x ** 2 - 4 * x * y - 4 * x + 4, "<=", y)
m.con(y ** 2, "<=", -x + 2)
m.expect(0.0, [1.0, 0.75])

for id_, ref, obj, eobj, ex in [
        ("110_01", "test/2d.jl:603-618", lambda x, y: exp(x), math.exp(-1), [-1.0, 0.0]),
        ("110_02", "test/2d.jl:621-636", lambda x, y: exp(x) + exp(y), 2 * math.exp(-1 / r2), [-1 / r2, -1 / r2]),
        ("110_03", "test/2d.jl:639-654", lambda x, y: exp(x + y), math.exp(-2 / r2), [-1 / r2, -1 / r2])]:
    m = Model(id_, ref); x = m.var("x"); y = m.var("y")
    m.objective("Min", obj(x, y), False)
    m.con(x ** 2 + y ** 2, "<=", 1.0)
    m.expect(eobj, ex)

# ---- test/3d.jl ------------------------------------------------------------
r3 = math.sqrt(3.0)
for id_, ref, obj, eobj, ex in [
        ("201_01", "test/3d.jl:5-23", lambda x, y, z: -(x + y + z), -3 / r3, [1 / r3] * 3),
        ("201_02", "test/3d.jl:26-42", lambda x, y, z: -x, -1.0, [1.0, 0.0, 0.0])]:
    m = Model(id_, ref); x = m.var("x"); y = m.var("y"); z = m.var("z")
    m.objective("Min", obj(x, y, z), True)
    m.con(x ** 2 + y ** 2 + z ** 2, "<=", 1.0)
    m.expect(eobj, ex)

for id_, ref, obj, eobj, ex, tol in [
        ("202_01", "test/3d.jl:46-64", lambda x, y, z: -z, -1.0, [0.0, 0.0, 1.0], None),
        ("202_02", "test/3d.jl:67-85", lambda x, y, z: 1 * z, 0.0, [0.0, 0.0, 0.0], None),
        ("202_03", "test/3d.jl:88-106", lambda x, y, z: -(x + y + 2 * z), -9 / 4, [1 / 4, 1 / 4, 7 / 8], None),
        ("202_04", "test/3d.jl:109-128", lambda x, y, z: x + y + 2 * z, -1 / 4, [-1 / 4, -1 / 4, 1 / 8], "rtol1e-7"),
        ("202_05", "test/3d.jl:131-149", lambda x, y, z: x + y, -1.0, [-1 / 2, -1 / 2, 1 / 2], None)]:
    m = Model(id_, ref); x = m.var("x"); y = m.var("y"); z = m.var("z")
    m.objective("Min", obj(x, y, z), True)
    m.con(x ** 2 + y ** 2, "<=", z)
    m.con(x ** 2 + y ** 2, "<=", -z + 1)
    if tol:
        m.expect(eobj, ex, obj_atol=0.0, obj_rtol=1e-7)   # test/3d.jl:124 isapprox(..., rtol=1e-7)
    else:
        m.expect(eobj, ex)

m = Model("203_01", "test/3d.jl:153-171"); x = m.var("x"); y = m.var("y"); z = m.var("z")
m.objective("Min", x + y, True)
m.con(sqrt(x ** 2 + y ** 2), "<=", z - 0.25)
m.con(x ** 2 + y ** 2, "<=", -z + 1)
m.expect(-1 / r2, [-math.sqrt(1 / 8), -math.sqrt(1 / 8), 3 / 4])

m = Model("205_01", "test/3d.jl:221-240"); x = m.var("x"); y = m.var("y", 0.0, INF); z = m.var("z")
m.objective("Max", 1 * y, True)
m.con(y * exp(x / y), "<=", z)
m.con(y * exp(-x / y), "<=", z)
m.con(x ** 2 + y ** 2, "<=", -z + 5)
m.expect(1.7912878443121907, [0.0, 1.7912878443121907, 1.7912878443121907])

for id_, ref, a, eobj, ex in [
        ("210_01", "test/3d.jl:271-288", 0.5, 0.0, [0.5] * 3),
        ("210_02", "test/3d.jl:291-308", 1.0, 0.535898380052066, [1 / r3] * 3),
        ("210_03", "test/3d.jl:311-328", 1.0, 0.535898380052066, [1 / r3] * 3)]:
    m = Model(id_, ref); x = m.var("x"); y = m.var("y"); z = m.var("z")
    m.objective("Min", (x - a) ** 2 + (y - a) ** 2 + (z - a) ** 2, False)
    m.con(x ** 2 + y ** 2 + z ** 2, "<=", 1.0)
    m.expect(eobj, ex)

# ---- test/misc.jl ----------------------------------------------------------
for n in range(1, 21):
    m = Model("501_01_n%d" % n, "test/misc.jl:4-30")
    vs = [m.var("x%d" % i) for i in range(n)]
    m.objective("Min", esum([-v for v in vs]), True)
    m.con(esum([v ** 2 for v in vs]), "<=", 1.0)
    m.expect(-n / math.sqrt(n), [1 / math.sqrt(n)] * n)
for n in range(1, 21):
    m = Model("501_02_n%d" % n, "test/misc.jl:33-57")
    vs = [m.var("x%d" % i) for i in range(n)]
    m.objective("Min", esum([-v for v in vs]), True)
    m.con(sqrt(esum([v ** 2 for v in vs])), "<=", 1.0)
    m.expect(-n / math.sqrt(n), [1 / math.sqrt(n)] * n)

# ---- extension of the test/misc.jl family beyond the reference's loop bound ----------
# The reference runs `for n in 1:20` (test/misc.jl:6,35); the family has the closed form obj = -sqrt(n), x = 1/sqrt(n)
# at every n.  These sizes are NOT among the reference's 82 tests: they exist so that the reference's tolerances are also
# asserted on smooth-face optima with more LP columns than the exact small-LP kernel takes (32).  They go to their own file
# (kat_family_ext.json); kat_models.json keeps exactly the reference's cases.
N_REFERENCE = len(MODELS)
for n in (33, 64, 128):
    m = Model("501_01_n%d" % n, "test/misc.jl:4-30 (family, n beyond the reference's 1:20)")
    vs = [m.var("x%d" % i) for i in range(n)]
    m.objective("Min", esum([-v for v in vs]), True)
    m.con(esum([v ** 2 for v in vs]), "<=", 1.0)
    m.expect(-n / math.sqrt(n), [1 / math.sqrt(n)] * n)
for n in (33, 64, 128):
    m = Model("501_02_n%d" % n, "test/misc.jl:33-57 (family, n beyond the reference's 1:20)")
    vs = [m.var("x%d" % i) for i in range(n)]
    m.objective("Min", esum([-v for v in vs]), True)
    m.con(sqrt(esum([v ** 2 for v in vs])), "<=", 1.0)
    m.expect(-n / math.sqrt(n), [1 / math.sqrt(n)] * n)


def _enc(o):
    if isinstance(o, float):
        if o == INF:
            return "inf"
        if o == -INF:
            return "-inf"
    return o


def _walk(o):
    if isinstance(o, dict):
        return {k: _walk(v) for k, v in o.items()}
    if isinstance(o, list):
        return [_walk(v) for v in o]
    return _enc(o)


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    tol = {"ref": "test/runtests.jl:16-20", "opt_atol": 1e-6, "opt_rtol": 1e-6, "sol_atol": 1e-3, "sol_rtol": 1e-3}
    out = os.path.join(here, "kat_models.json")
    with open(out, "w") as f:
        json.dump({"tolerances": tol, "models": _walk(MODELS[:N_REFERENCE])}, f, indent=1)
    print("wrote", out, N_REFERENCE, "models")
    out = os.path.join(here, "kat_family_ext.json")
    with open(out, "w") as f:
        json.dump({"tolerances": tol, "note": "closed-form members of the test/misc.jl family at sizes the reference does not run",
                   "models": _walk(MODELS[N_REFERENCE:])}, f, indent=1)
    print("wrote", out, len(MODELS) - N_REFERENCE, "models")
