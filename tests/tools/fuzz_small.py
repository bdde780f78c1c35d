"""Development aid: random small convex models (tape rows) -- HIP engine vs the CPU oracle."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import katana_jl_amd as ktn
from helpers import hip_model_from_kat, oracle_solve_kat

from fuzz_models import rand_model

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
ONLY = int(sys.argv[3]) if len(sys.argv) > 3 else -1
MAXIT = int(sys.argv[4]) if len(sys.argv) > 4 else 2000000
NV_LO, NV_HI = (int(v) for v in os.environ.get("FUZZ_NV", "2,7").split(","))     # (the replayed regressions use the default range)
for t in range(N):
    m = rand_model(rng, int(rng.integers(NV_LO, NV_HI)), boxed=rng.random() < 0.5)
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
