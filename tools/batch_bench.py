"""cfg5: a batch of independent 1e3-variable convex NLPs, throughput mode."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import katana_jl_amd as ktn
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 512
insts = [ktn.instances.make_config("cfg5_one", seed=s) for s in range(nb)]
for thr in [int(a) for a in sys.argv[2:]] or [1, 8, 32]:
    res, wall = ktn.solve_batch(ktn.KatanaSolver(log_level=0), insts[: (nb if thr > 1 else min(nb, 64))], threads=thr)
    ok = sum(r["status"] == "Optimal" for r in res)
    err = max(abs(r["objval"] - i.opt_obj) / max(1, abs(i.opt_obj)) for r, i in zip(res, insts))
    print(json.dumps({"threads": thr, "instances": len(res), "optimal": ok, "wall_s": wall, "instances_per_s": len(res) / wall,
                      "max_obj_relerr": err, "ecp_iters_mean": sum(r["iters"] for r in res) / len(res)}), flush=True)

for per_inst in (True, False, True):
    res, wall = ktn.solve_batch(ktn.KatanaSolver(log_level=0), insts, fused=True, per_instance_lp=per_inst)
    ok = sum(r["status"] == "Optimal" for r in res)
    err = max(abs(r["objval"] - i.opt_obj) / max(1, abs(i.opt_obj)) for r, i in zip(res, insts))
    print(json.dumps({"fused": True, "per_instance_lp": per_inst, "instances": len(res), "optimal": ok, "wall_s_incl_load": wall,
                      "instances_per_s": len(res) / wall, "max_obj_relerr": err, "ecp_iters": res[0]["iters"],
                      "pdhg_iters": res[0]["pdhg_iters"], "blk_lp_launches": res[0]["blk_lp_launches"],
                      "blk_lp_fallbacks": res[0]["blk_lp_fallbacks"], "blk_pdhg_iters_sum": res[0]["blk_pdhg_iters_sum"]}), flush=True)
