import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import katana_jl_amd as ktn
name, n0, n1 = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
kw = {}
for a in sys.argv[4:]:
    k, v = a.split("="); kw[k] = float(v) if ("." in v or "e" in v) else int(v)
tot_w = tot_p = 0; ws = []; errs = []
for seed in range(n0, n1):
    inst = ktn.instances.make_config(name, seed=seed)
    m = ktn.NonlinearModel(ktn.KatanaSolver(log_level=0, **kw))
    m.loadproblem(inst.n, inst.num_constr, inst.l_var, inst.u_var, inst.l_constr, inst.u_constr, inst.sense, ktn.SeparableNLP(inst))
    m.optimize(); m.reset()              # warm buffers: the timed solve allocates nothing (as in bench.py)
    p0 = m.stat("pdhg_iters")
    t = time.time(); st = m.optimize(); w = time.time() - t
    assert st == "Optimal"
    ws.append(w); tot_p += m.stat("pdhg_iters") - p0; errs.append(abs(m.getobjval() - inst.opt_obj) / max(1.0, abs(inst.opt_obj)))
print("%s seeds %d-%d %s: mean %.3fs median %.3fs max %.3fs total pdhg %d max relerr %.1e | %s" % (name, n0, n1 - 1, kw, np.mean(ws), np.median(ws), np.max(ws), tot_p, max(errs), " ".join("%.3f" % w for w in ws)))
