"""dev: a matrix of vertex-planted instance shapes (family x objective x size x seed) -- status, accuracy, time.
(The smooth-face regime, vertex=False, needs hundreds to thousands of ECP iterations by nature of Kelley's method.)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import katana_jl_amd as ktn
worst = 0.0
for fam in ("explog", "quad"):
    for obj in ("linear", "quad"):
        for (n, m_nl, k) in ((300, 30, 8), (3000, 300, 16), (10000, 1000, 32)):
            for seed in range(4):
                inst = ktn.instances.make_instance(n=n, m_nl=m_nl, k=k, family=fam, seed=seed, objective=obj)
                m = ktn.NonlinearModel(ktn.KatanaSolver(log_level=0, lp_max_iter=400000))
                m.loadproblem(inst.n, inst.num_constr, inst.l_var, inst.u_var, inst.l_constr, inst.u_constr, inst.sense, ktn.SeparableNLP(inst))
                t = time.time(); st = m.optimize(); w = time.time() - t
                worst = max(worst, w)
                print("%-6s obj=%-6s n=%-5d m_nl=%-4d seed=%d: %-9s iters=%-4d wall=%6.2fs pdhg=%-8d relerr=%.1e" % (
                    fam, obj, n, m_nl, seed, st, m.numiters(), w, m.stat("pdhg_iters"),
                    abs(m.getobjval() - inst.opt_obj) / max(1, abs(inst.opt_obj))), flush=True)
print("worst wall %.2fs" % worst)
