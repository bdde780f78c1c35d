"""Development aid: wall-clock of one cfg3 solve under different GPU-LP parameters."""
import os, sys, time, itertools, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import katana_jl_amd as ktn
inst = ktn.instances.make_config(sys.argv[1] if len(sys.argv) > 1 else "cfg3", seed=0)
d = ktn.SeparableNLP(inst)
grid = json.loads(sys.argv[2]) if len(sys.argv) > 2 else [{}]
for kw in grid:
    m = ktn.NonlinearModel(ktn.KatanaSolver(log_level=0, **kw))
    m.loadproblem(inst.n, inst.num_constr, inst.l_var, inst.u_var, inst.l_constr, inst.u_constr, inst.sense, d)
    m.optimize(); m.reset()
    t = time.perf_counter(); st = m.optimize(); w = time.perf_counter() - t
    print("%-70s %s wall=%.3fs iters=%d pdhg=%d relerr=%.1e" % (json.dumps(kw), st, w, m.numiters(), m.stat("pdhg_iters") / 2,
          abs(m.getobjval() - inst.opt_obj) / max(1, abs(inst.opt_obj))), flush=True)
