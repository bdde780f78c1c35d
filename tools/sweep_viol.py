"""Sweep kernel time when (nearly) every NL row is violated -- the first cutting-plane rounds of cfg4 -- against the
near-optimal point tools/sweep_bench.py uses.  usage: sweep_viol.py [config] [reps]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import katana_jl_amd as ktn
name = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
inst = ktn.instances.make_config(name, seed=0)
m = ktn.NonlinearModel(ktn.KatanaSolver(log_level=0, profile=1))
m.loadproblem(inst.n, inst.num_constr, inst.l_var, inst.u_var, inst.l_constr, inst.u_constr, inst.sense, ktn.SeparableNLP(inst))
sep = ktn.KatanaHipSeparator(m); sep.initialize()
out = {}
for label, x in (("near_optimum", np.clip(inst.xhat + 0.05, inst.l_var, inst.u_var)), ("upper_bounds", np.where(np.isfinite(inst.u_var), inst.u_var, inst.xhat + 10.0))):
    m.reset(); sep.precompute(x)
    t0, n0 = m.stat("sweep_eval_time_s"), m.stat("sweep_eval_launches")
    for r in range(reps):
        nv, mv = sep.sweep(1e-6)
        m.reset(); sep.precompute(x)
    tt, nl = m.stat("sweep_eval_time_s") - t0, m.stat("sweep_eval_launches") - n0
    out[label] = {"violated_rows": int(nv), "avg_launch_us": round(1e6 * tt / nl, 1)}
print(json.dumps(out))
