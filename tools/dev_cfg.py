import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import katana_jl_amd as ktn
name = sys.argv[1]
t = time.time(); inst = ktn.instances.make_config(name, seed=0); tg = time.time() - t
m = ktn.NonlinearModel(ktn.KatanaSolver(log_level=1, **({"cut_cap_factor": float(os.environ["CUTCAP"])} if "CUTCAP" in os.environ else {})))
t = time.time()
m.loadproblem(inst.n, inst.num_constr, inst.l_var, inst.u_var, inst.l_constr, inst.u_constr, inst.sense, ktn.SeparableNLP(inst))
tl = time.time() - t; t = time.time()
st = m.optimize(); ts = time.time() - t
x = m.getsolution()
print("%s: gen %.1fs load %.2fs solve %.3fs %s iters=%d cuts=%d lp_rows=%d obj=%.9f opt=%.9f relerr=%.2e pdhg=%d lp=%.3fs sep=%.3fs xerr=%.1e" % (
    name, tg, tl, ts, st, m.numiters(), m.numcuts(), m.lp_num_rows(), m.getobjval(), inst.opt_obj,
    abs(m.getobjval() - inst.opt_obj) / max(1, abs(inst.opt_obj)), m.stat("pdhg_iters"), m.stat("lp_time_s"), m.stat("sep_time_s"),
    np.max(np.abs(x - inst.xhat))), flush=True)
print("lp setup %.4fs of lp %.4fs over %d solves" % (m.stat("lp_setup_time_s"), m.stat("lp_time_s"), m.stat("lp_solves")))
