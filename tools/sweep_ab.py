"""The sweep on many short rows: the row kernel (k_sep_sweep<G, 4>) against the batch-blocked kernel (k_sep_sweep_batch) on the
same instance and point, in ONE process (the switch is read per handle): results compared (g of every NL row, violated rows,
largest violation, the cuts appended), then timed.  usage: sweep_ab.py [config=cfg4] [reps=8]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C
import numpy as np
import katana_jl_amd as ktn
name = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
inst = ktn.instances.make_config(name, seed=0) if name in ktn.instances.CONFIGS else None
if inst is None:
    fam, n, m_nl, k = name.split(":")
    inst = ktn.instances.make_instance(n=int(n), m_nl=int(m_nl), k=int(k), family=fam, seed=0)
models = {}
for label, env in (("row", "0"), ("batch", "1")):
    os.environ["KTN_SWEEP_BATCHED"] = env
    m = ktn.NonlinearModel(ktn.KatanaSolver(log_level=0, profile=1, purge_age=0, cut_cap_factor=0.0))
    m.loadproblem(inst.n, inst.num_constr, inst.l_var, inst.u_var, inst.l_constr, inst.u_constr, inst.sense, ktn.SeparableNLP(inst))
    sep = ktn.KatanaHipSeparator(m); sep.initialize()
    models[label] = (m, sep)
del os.environ["KTN_SWEEP_BATCHED"]
out = {}
for pt, x in (("near_optimum", np.clip(inst.xhat + 0.05, inst.l_var, inst.u_var)), ("mid", np.clip(inst.xhat + 0.5, inst.l_var, inst.u_var))):
    res = {}
    for label, (m, sep) in models.items():
        m.reset(); sep.precompute(x)
        t0, n0 = m.stat("sweep_eval_time_s"), m.stat("sweep_eval_launches")
        for r in range(reps):
            nv, mv = sep.sweep(1e-6)
            if r == 0:
                g = np.zeros(sep.num_constr)
                assert m._lib.ktn_sep_get_g(m._h, g.ctypes.data_as(C.POINTER(C.c_double)), sep.num_constr) == 0
                rows = m.lp_rows()
            m.reset(); sep.precompute(x)
        tt, nl = m.stat("sweep_eval_time_s") - t0, m.stat("sweep_eval_launches") - n0
        res[label] = dict(nv=nv, mv=mv, g=g, rows=rows, us=1e6 * tt / nl)
    a, b = res["row"], res["batch"]
    gdiff = float(np.max(np.abs(a["g"][inst.m_lin:] - b["g"][inst.m_lin:]) / (1.0 + np.abs(a["g"][inst.m_lin:]))))
    same_rows = all(np.array_equal(p, q) for p, q in zip(a["rows"], b["rows"]))
    out[pt] = dict(violated=(a["nv"], b["nv"]), maxviol=(a["mv"], b["mv"]), g_max_rel_diff=gdiff, same_lp_rows=bool(same_rows),
                   row_kernel_us=round(a["us"], 1), batch_kernel_us=round(b["us"], 1))
print(json.dumps(out))
