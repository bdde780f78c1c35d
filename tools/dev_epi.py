import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import katana_jl_amd as ktn
inst = ktn.instances.make_instance(n=600, m_nl=60, k=16, family="explog", seed=2, objective="quad")
m = ktn.NonlinearModel(ktn.KatanaSolver(log_level=1, lp_max_iter=int(sys.argv[1]) if len(sys.argv)>1 else 200000))
m.loadproblem(inst.n, inst.num_constr, inst.l_var, inst.u_var, inst.l_constr, inst.u_constr, inst.sense, ktn.SeparableNLP(inst))
print(m.optimize(), m.numiters(), m.getobjval(), inst.opt_obj)
import numpy as np
rowptr, col, val, lo, hi = m.lp_rows()
xh = np.append(inst.xhat, inst.opt_obj)
ax = np.array([val[rowptr[i]:rowptr[i+1]] @ xh[col[rowptr[i]:rowptr[i+1]]] for i in range(len(lo))])
viol = np.maximum(ax - hi, lo - ax)
bad = np.argsort(-viol)[:5]
print("rows", len(lo), "m_lin", inst.m_lin, "worst violations of the planted point:", [(int(i), float(viol[i]), int(rowptr[i+1]-rowptr[i])) for i in bad])
x = m.getsolution()
ax = np.array([val[rowptr[i]:rowptr[i+1]] @ x[col[rowptr[i]:rowptr[i+1]]] for i in range(len(lo))])
viol = np.maximum(ax - hi, lo - ax)
bad = np.argsort(-viol)[:5]
print("worst violations of the returned point:", [(int(i), float(viol[i]), int(rowptr[i+1]-rowptr[i])) for i in bad])
print("bound viol", np.max(np.maximum(inst.l_var - x[:-1], x[:-1] - inst.u_var)))
c, c0 = m.lp_objective()
np.savez("gpurun_out/epi_lp.npz", rowptr=rowptr, col=col, val=val, lo=lo, hi=hi, c=c, c0=c0, l=np.append(inst.l_var, -np.inf),
         u=np.append(inst.u_var, np.inf), x=m.getsolution(), y=m.lp_duals(), xhat=xh)
print("consolidations", m.stat("lp_consolidations"), "eta backoffs", m.stat("lp_eta_backoffs"), "pdhg", m.stat("pdhg_iters"))
