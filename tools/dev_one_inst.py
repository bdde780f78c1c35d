"""dev: per-ECP-step LP counters for one generated instance.  usage: dev_one_inst.py family objective n m_nl k seed [opt=val ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import katana_jl_amd as ktn
fam, obj, n, m_nl, k, seed = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
kw = {}
for a in sys.argv[7:]:
    kk, v = a.split("="); kw[kk] = float(v) if ("." in v or "e" in v) else int(v)
inst = ktn.instances.make_instance(n=n, m_nl=m_nl, k=k, family=fam, seed=seed, objective=obj)
m = ktn.NonlinearModel(ktn.KatanaSolver(log_level=0, **kw))
m.loadproblem(inst.n, inst.num_constr, inst.l_var, inst.u_var, inst.l_constr, inst.u_constr, inst.sense, ktn.SeparableNLP(inst))
keys = ("pdhg_iters", "lp_restarts", "lp_consolidations", "lp_eta_backoffs", "lp_divergence_backoffs", "lp_stagnation_exits", "lp_stalled_row_exits")
prev = {kk: 0.0 for kk in keys}
m.optimize_begin()
step = 0
while True:
    t = time.time(); fin = m.ecp_step(); dt = time.time() - t
    step += 1
    cur = {kk: m.stat(kk) for kk in keys}
    print("step %3d %.3fs rows %d " % (step, dt, m.lp_num_rows()) + " ".join("%s=%d" % (kk.replace("lp_", ""), cur[kk] - prev[kk]) for kk in keys), flush=True)
    prev = cur
    if fin: break
print("final", m.optimize_end(), m.getobjval(), inst.opt_obj)
