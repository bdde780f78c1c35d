import sys
sys.path.insert(0,'/root/repo')
import katana_jl_amd as ktn
for cfg, seed in (("cfg3",28),("cfg3",29),("cfg3",0)):
    inst = ktn.instances.make_config(cfg, seed=seed)
    m = ktn.NonlinearModel(ktn.KatanaSolver(log_level=0))
    m.loadproblem(inst.n, inst.num_constr, inst.l_var, inst.u_var, inst.l_constr, inst.u_constr, inst.sense, ktn.SeparableNLP(inst))
    st = m.optimize()
    print(cfg, seed, st, m.numiters(), m.stat("pdhg_iters"), "div", m.stat("lp_divergence_backoffs"), "eta", m.stat("lp_eta_backoffs"), "reuse", m.stat("lp_setup_reuses"), "stag", m.stat("lp_stagnation_exits"), "stalled", m.stat("lp_stalled_row_exits"))
