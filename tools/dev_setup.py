"""dev: breakdown of the per-solve LP setup on a config (profile=1 adds syncs)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import katana_jl_amd as ktn
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
inst = ktn.instances.make_config(name, seed=0)
for prof in (0, 1):
    m = ktn.NonlinearModel(ktn.KatanaSolver(log_level=0, profile=prof))
    m.loadproblem(inst.n, inst.num_constr, inst.l_var, inst.u_var, inst.l_constr, inst.u_constr, inst.sense, ktn.SeparableNLP(inst))
    t = time.time(); st = m.optimize(); ts = time.time() - t
    print("profile=%d %s %.3fs iters=%d pdhg=%d lp=%.4f setup=%.4f csc=%.4f scaling=%.4f power=%.4f sep=%.4f solves=%d" % (
        prof, st, ts, m.numiters(), m.stat("pdhg_iters"), m.stat("lp_time_s"), m.stat("lp_setup_time_s"), m.stat("lp_csc_time_s"),
        m.stat("lp_scaling_time_s"), m.stat("lp_power_time_s"), m.stat("sep_time_s"), m.stat("lp_solves")))

# per-step breakdown through the stepping API
m = ktn.NonlinearModel(ktn.KatanaSolver(log_level=0, profile=1))
m.loadproblem(inst.n, inst.num_constr, inst.l_var, inst.u_var, inst.l_constr, inst.u_constr, inst.sense, ktn.SeparableNLP(inst))
m.optimize_begin()
keys = ("lp_csc_time_s", "lp_scaling_time_s", "lp_power_time_s", "lp_setup_time_s", "lp_time_s", "sep_time_s", "pdhg_iters")
prev = {k: m.stat(k) for k in keys}
done = False
while not done:
    done = m.ecp_step()
    cur = {k: m.stat(k) for k in keys}
    print("step %2d " % m.numiters() + " ".join("%s=%.3fms" % (k.replace("_time_s", ""), 1e3 * (cur[k] - prev[k])) if k != "pdhg_iters" else "pdhg=%d" % (cur[k] - prev[k]) for k in keys))
    prev = cur
m.optimize_end()
