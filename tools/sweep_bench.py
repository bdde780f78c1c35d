"""Separator-sweep microbenchmark on an HBM-resident instance (cfg3_hbm: k = 2048 nnz per NL row,
2.05e7 Jacobian entries, 430 MB of row data): algorithmic bytes / kernel time of k_sep_eval."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import katana_jl_amd as ktn

name = sys.argv[1] if len(sys.argv) > 1 else "cfg3_hbm"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
t = time.time()
fam = sys.argv[3] if len(sys.argv) > 3 else None
kw = dict(family=fam) if fam else {}
inst = ktn.instances.make_config(name, seed=0, vertex=False, **kw) if name == "cfg3_hbm" else ktn.instances.make_config(name, seed=0, **kw)
print("generated %s in %.1fs" % (name, time.time() - t), flush=True)
m = ktn.NonlinearModel(ktn.KatanaSolver(log_level=0, profile=1))
m.loadproblem(inst.n, inst.num_constr, inst.l_var, inst.u_var, inst.l_constr, inst.u_constr, inst.sense, ktn.SeparableNLP(inst))
sep = ktn.KatanaHipSeparator(m); sep.initialize()
x = np.clip(inst.xhat + 0.05, inst.l_var, inst.u_var)
sep.precompute(x)
out = []
for r in range(reps):
    nv, mv = sep.sweep(1e-6)
    if (r + 1) % 4 == 0:
        m.reset(); sep.precompute(x)
tt, nl, by = m.stat("sweep_eval_time_s"), m.stat("sweep_eval_launches"), m.stat("sweep_eval_bytes")
res = {"workload": name, "family": inst.meta["family"], "nnz_nl": int(inst.rowptr[-1] - inst.rowptr[inst.m_lin]), "violated_rows": nv, "launches": nl,
       "avg_launch_us": 1e6 * tt / nl, "algorithmic_bytes_per_launch": by / nl, "achieved_GBps": by / tt / 1e9,
       "frac_of_8TBps": by / tt / 8e12}
print(json.dumps(res))
