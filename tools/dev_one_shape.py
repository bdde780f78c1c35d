import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import katana_jl_amd as ktn
fam, obj, n, m_nl, k, seed = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
kw = {}
for a in sys.argv[7:]:
    kk, v = a.split("="); kw[kk] = float(v) if ("." in v or "e" in v) else int(v)
inst = ktn.instances.make_instance(n=n, m_nl=m_nl, k=k, family=fam, seed=seed, objective=obj)
m = ktn.NonlinearModel(ktn.KatanaSolver(log_level=1, **kw))
m.loadproblem(inst.n, inst.num_constr, inst.l_var, inst.u_var, inst.l_constr, inst.u_constr, inst.sense, ktn.SeparableNLP(inst))
t = time.time(); st = m.optimize(); w = time.time() - t
print(st, "iters", m.numiters(), "wall %.2fs" % w, "pdhg", m.stat("pdhg_iters"), "relerr %.2e" % (abs(m.getobjval() - inst.opt_obj) / max(1, abs(inst.opt_obj))),
      "stag exits", m.stat("lp_stagnation_exits"), "consol", m.stat("lp_consolidations"), "backoffs", m.stat("lp_eta_backoffs"), flush=True)
