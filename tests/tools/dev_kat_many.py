"""dev: run a list of KAT ids (or 'all') on the GPU, one line each: status objective expected iters time stats."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import katana_jl_amd as ktn
from kat_util import load_kats
from helpers import hip_model_from_kat
ids = sys.argv[1:]
for k in load_kats():
    if ids != ["all"] and k["id"] not in ids:
        continue
    M = hip_model_from_kat(ktn, k, lp_max_iter=int(os.environ.get("LPMAX", "2000000")), **({"lp_dense_after": int(os.environ["LPDENSE"])} if "LPDENSE" in os.environ else {}))
    t0 = time.time()
    st = M.solve()
    im = M.internal_model
    exp = k.get("objective")
    print(k["id"], st, "obj=%.9g" % M.getobjectivevalue(), "exp=%s" % exp, "it=%d" % im.numiters(),
          "%.2fs" % (time.time() - t0), "dense=%d fb=%d pdhg=%d" % (im.stat("dense_lp_solves"), im.stat("dense_lp_fallbacks"), im.stat("pdhg_iters")), flush=True)
