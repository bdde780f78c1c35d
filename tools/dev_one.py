"""dev: one config / seed with solver options; prints per-ECP-step LP counters (and KTN_DEBUG_LP=1 gives the per-check LP trace)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import katana_jl_amd as ktn
name, seed = sys.argv[1], int(sys.argv[2])
kw = {}
for a in sys.argv[3:]:
    k, v = a.split("="); kw[k] = float(v) if ("." in v or "e" in v) else int(v)
inst = ktn.instances.make_config(name, seed=seed)
m = ktn.NonlinearModel(ktn.KatanaSolver(log_level=0, **kw))
m.loadproblem(inst.n, inst.num_constr, inst.l_var, inst.u_var, inst.l_constr, inst.u_constr, inst.sense, ktn.SeparableNLP(inst))
m.optimize_begin()
keys = ("pdhg_iters", "lp_restarts", "lp_consolidations", "lp_eta_backoffs", "lp_stagnation_exits", "lp_stalled_row_exits", "lp_time_s")
prev = {k: m.stat(k) for k in keys}
done = False
while not done:
    t = time.time(); done = m.ecp_step(); dt = time.time() - t
    cur = {k: m.stat(k) for k in keys}
    print("step %2d %.3fs rows %d " % (m.numiters(), dt, m.lp_num_rows()) + " ".join("%s=%g" % (k.replace("lp_", ""), cur[k] - prev[k]) for k in keys), flush=True)
    prev = cur
print(m.optimize_end(), m.getobjval(), inst.opt_obj)
