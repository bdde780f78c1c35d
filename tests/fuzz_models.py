"""Random small convex models in the fixture format of tests/golden/kat_models.json (tape rows: convex quadratics, exp-sums,
sqrt-norms, linear rows; linear or quadratic objective; boxed or free variables).  tests/tools/fuzz_small.py runs batches of
them against the CPU oracle; tests/test_gpu_kats.py replays the members that once failed."""
import numpy as np


def rand_model(rng, nv, boxed):
    V = [["var", j] for j in range(nv)]
    cons = []
    for _ in range(rng.integers(1, 4)):
        # convex quadratic  sum a_j (x_j - c_j)^2 <= r   or  exp-sum  or  sqrt-norm
        kind = rng.integers(0, 3)
        if kind == 0:
            e = ["+"] + [["*", float(rng.uniform(0.5, 2)), ["^", ["-", v, float(rng.normal())], 2.0]] for v in V]
            cons.append({"expr": ["-", e, float(rng.uniform(1.0, 4.0) * nv)], "lb": -np.inf, "ub": 0.0, "linear": False})
        elif kind == 1:
            e = ["+"] + [["exp", ["*", float(rng.uniform(-1, 1)), v]] for v in V]
            cons.append({"expr": ["-", e, float(nv * rng.uniform(1.5, 3.0))], "lb": -np.inf, "ub": 0.0, "linear": False})
        else:
            e = ["sqrt", ["+"] + [["^", ["-", v, float(rng.normal() * 0.3)], 2.0] for v in V] + [0.01]]
            cons.append({"expr": ["-", e, float(rng.uniform(1.0, 3.0))], "lb": -np.inf, "ub": 0.0, "linear": False})
    for _ in range(rng.integers(0, 3)):
        a = rng.normal(size=nv)
        e = ["+"] + [["*", float(a[j]), V[j]] for j in range(nv)]
        cons.append({"expr": ["-", e, float(abs(rng.normal()) + 0.5)], "lb": -np.inf, "ub": 0.0, "linear": True})
    c = rng.normal(size=nv)
    if rng.random() < 0.4:
        obj, lin = ["+"] + [["^", ["-", V[j], float(rng.normal())], 2.0] for j in range(nv)], False
    else:
        obj, lin = ["+"] + [["*", float(c[j]), V[j]] for j in range(nv)], True
    b = 5.0 if boxed else np.inf
    return {"id": "fuzz", "vars": [{"lb": -b, "ub": b}] * nv, "sense": "Min" if rng.random() < 0.8 or not lin else "Max",
            "objective": obj, "objective_linear": lin, "constraints": cons, "expect": {}}



def model_at(seed, index):
    """the index-th model of fuzz_small.py's stream for `seed` (the generator is replayed up to it)"""
    rng = np.random.default_rng(seed)
    for t in range(index + 1):
        m = rand_model(rng, int(rng.integers(2, 7)), boxed=rng.random() < 0.5)
    if m["sense"] == "Max":
        m["sense"] = "Min"; m["objective"] = ["neg", m["objective"]]
    m["id"] = "fuzz_%d_%d" % (seed, index)
    return m
