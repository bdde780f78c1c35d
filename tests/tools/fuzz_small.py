"""Development aid: random small convex models (tape rows) -- HIP engine vs the CPU oracle."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import katana_jl_amd as ktn
from helpers import hip_model_from_kat, oracle_solve_kat

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

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
ONLY = int(sys.argv[3]) if len(sys.argv) > 3 else -1
MAXIT = int(sys.argv[4]) if len(sys.argv) > 4 else 2000000
for t in range(N):
    m = rand_model(rng, int(rng.integers(2, 7)), boxed=rng.random() < 0.5)
    if ONLY >= 0 and t != ONLY:
        continue
    if m["sense"] == "Max":
        m["sense"] = "Min"; m["objective"] = ["neg", m["objective"]]
    om = oracle_solve_kat(m)
    M = hip_model_from_kat(ktn, m, lp_max_iter=MAXIT, **({"lp_dense_after": int(os.environ["LPDENSE"])} if "LPDENSE" in os.environ else {}))
    t0 = time.time(); st = M.solve(); w = time.time() - t0
    so = om.getstatus()
    ok = (st == so) and (st != "Optimal" or abs(M.getobjectivevalue() - om.getobjval()) <= 1e-5 * max(1, abs(om.getobjval())))
    bad += not ok
    print("%2d nv=%d cons=%d boxed=%s objlin=%s | hip %s %.8f it=%d pdhg=%d %.2fs | oracle %s %.8f it=%d %s" % (
        t, len(m["vars"]), len(m["constraints"]), np.isfinite(m["vars"][0]["ub"]), m["objective_linear"], st, M.getobjectivevalue(),
        M.internal_model.numiters(), M.internal_model.stat("pdhg_iters"), w, so, om.getobjval() if so == "Optimal" else float("nan"),
        om.numiters(), "" if ok else "  <<< MISMATCH"), flush=True)
    if ONLY >= 0:
        im = M.internal_model
        rowptr, col, val, lo, hi = im.lp_rows()
        c, c0 = im.lp_objective()
        np.savez(os.path.join(ROOT, "gpurun_out", "fuzz_lp.npz"), rowptr=rowptr, col=col, val=val, lo=lo, hi=hi, c=c, c0=c0,
                 l=np.array([v["lb"] for v in m["vars"]] + ([-np.inf] if not m["objective_linear"] else [])),
                 u=np.array([v["ub"] for v in m["vars"]] + ([np.inf] if not m["objective_linear"] else [])),
                 x=im.getsolution(), y=im.lp_duals())
        print("consolidations", im.stat("lp_consolidations"), "backoffs", im.stat("lp_eta_backoffs"))
print("mismatches:", bad, "of", N)
