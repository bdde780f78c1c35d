"""dev: every reference KAT on the GPU under one or more option variants; per model the objective error, the worst
solution error, the worst nonlinear-row value at the returned point and the iteration count, then a summary of what
misses the reference's own tolerances (test/runtests.jl:16-20, test/3d.jl:124).

    python tests/tools/dev_kat_strict.py '{}' '{"f_tol": 1e-7}' ...
"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import katana_jl_amd as ktn
from kat_util import isapprox, load_kats
from helpers import hip_model_from_kat
from oracle import sexpr

variants = [json.loads(a) for a in sys.argv[1:]] or [{}]
only = os.environ.get("KATS")
for var in variants:
    print("==== variant", var, flush=True)
    bad_obj, bad_x, bad_st, tot = [], [], [], 0.0
    for k in load_kats():
        if only and k["id"] not in only.split(","):
            continue
        M = hip_model_from_kat(ktn, k, **var)
        t0 = time.time()
        st = M.solve()
        dt = time.time() - t0
        tot += dt
        im = M.internal_model
        e = k["expect"]
        obj = M.getobjectivevalue()
        xs = np.asarray(M.getvalue())
        oerr = abs(obj - e["obj"])
        xerr = max((abs(a - b) for a, b in zip(xs, e["x"])), default=0.0) if e["x"] is not None else 0.0
        gmax = -np.inf
        for c in k["constraints"]:
            if c["linear"]:
                continue
            with np.errstate(all="ignore"):
                g = sexpr.eval_grad(c["expr"], xs)[0]
            gmax = max(gmax, g - c["ub"], c["lb"] - g)
        ok_o = isapprox(obj, e["obj"], e["obj_atol"], e["obj_rtol"])
        ok_x = e["x"] is None or all(isapprox(a, b, e["sol_atol"], e["sol_rtol"]) for a, b in zip(xs, e["x"]))
        if st != e["status"]: bad_st.append(k["id"])
        if not ok_o: bad_obj.append((k["id"], oerr))
        if not ok_x: bad_x.append((k["id"], xerr))
        print("%-12s %-9s oerr %.2e xerr %.2e gmax %+.2e it %5d cuts %5d %.2fs dense %d stalls %d pdhg %d polish %d%s" % (
            k["id"], st, oerr, xerr, gmax, im.numiters(), im.numcuts(), dt, im.stat("dense_lp_solves"), im.stat("lp_stalls"),
            im.stat("pdhg_iters"), im.stat("polish_iters"), "" if (ok_o and ok_x) else "   <-- MISS"), flush=True)
    print("---- variant", var, "total %.1fs" % tot)
    print("status misses:", bad_st)
    print("objective misses:", [(i, "%.2e" % v) for i, v in bad_obj])
    print("solution misses:", [(i, "%.2e" % v) for i, v in bad_x], flush=True)
