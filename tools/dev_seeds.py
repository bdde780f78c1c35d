import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import katana_jl_amd as ktn
for name in sys.argv[1:]:
    for seed in range(5):
        inst = ktn.instances.make_config(name, seed=seed)
        m = ktn.NonlinearModel(ktn.KatanaSolver(log_level=0))
        m.loadproblem(inst.n, inst.num_constr, inst.l_var, inst.u_var, inst.l_constr, inst.u_constr, inst.sense, ktn.SeparableNLP(inst))
        t = time.time(); st = m.optimize(); w = time.time() - t
        print("%s seed %d: %s iters=%d wall=%.3fs pdhg=%d relerr=%.1e xerr=%.1e consol=%d backoffs=%d" % (
            name, seed, st, m.numiters(), w, m.stat("pdhg_iters"), abs(m.getobjval() - inst.opt_obj) / max(1, abs(inst.opt_obj)),
            np.max(np.abs(m.getsolution()[:inst.n] - inst.xhat)), m.stat("lp_consolidations"), m.stat("lp_eta_backoffs")), flush=True)
