import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import katana_jl_amd as ktn
from kat_util import load_kats
from helpers import hip_model_from_kat
m = [k for k in load_kats() if k["id"] == sys.argv[1]][0]
M = hip_model_from_kat(ktn, m, lp_max_iter=int(sys.argv[2]) if len(sys.argv) > 2 else 20000)
print(M.solve(), M.getobjectivevalue(), M.getvalue(), M.internal_model.numiters())
print(M.internal_model.lp_rows())
