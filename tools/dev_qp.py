import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import katana_jl_amd as ktn
n, m_nl, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
inst = ktn.instances.make_instance(n=n, m_nl=m_nl, k=k, family="quad", seed=0, objective="quad")
m = ktn.NonlinearModel(ktn.KatanaSolver(log_level=1))
m.loadproblem(inst.n, inst.num_constr, inst.l_var, inst.u_var, inst.l_constr, inst.u_constr, inst.sense, ktn.SeparableNLP(inst))
t = time.time(); st = m.optimize()
print(st, "iters", m.numiters(), "obj", m.getobjval(), "opt", inst.opt_obj, "relerr", abs(m.getobjval()-inst.opt_obj)/max(1,abs(inst.opt_obj)),
      "wall %.2fs pdhg %d lp %.2fs sep %.3fs" % (time.time()-t, m.stat("pdhg_iters"), m.stat("lp_time_s"), m.stat("sep_time_s")))
