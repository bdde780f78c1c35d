"""dev: a matrix of instance shapes (family x objective x vertex / smooth-face optimum) -- status, accuracy, time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import katana_jl_amd as ktn
cases = []
for fam in ("explog", "quad"):
    for obj in ("linear", "quad"):
        for vertex in (True, False):
            for (n, m_nl, k) in ((300, 30, 8), (3000, 300, 16)):
                cases.append((fam, obj, vertex, n, m_nl, k))
for fam, obj, vertex, n, m_nl, k in cases:
    inst = ktn.instances.make_instance(n=n, m_nl=m_nl, k=k, family=fam, seed=1, objective=obj, vertex=vertex)
    m = ktn.NonlinearModel(ktn.KatanaSolver(log_level=0, iter_cap=(10000 if vertex else 400), lp_max_iter=300000))
    m.loadproblem(inst.n, inst.num_constr, inst.l_var, inst.u_var, inst.l_constr, inst.u_constr, inst.sense, ktn.SeparableNLP(inst))
    t = time.time(); st = m.optimize(); w = time.time() - t
    print("%-6s obj=%-6s vertex=%-5s n=%-5d m_nl=%-4d: %-9s iters=%-5d wall=%7.2fs pdhg=%-9d relerr=%.1e" % (
        fam, obj, vertex, n, m_nl, st, m.numiters(), w, m.stat("pdhg_iters"), abs(m.getobjval() - inst.opt_obj) / max(1, abs(inst.opt_obj))), flush=True)
