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

for mode in ("device_loop", "global_lp", "per_instance_lp", "device_loop"):
    kw = dict(device_loop=(mode == "device_loop"), per_instance_lp=(mode == "per_instance_lp"))
    res, wall = ktn.solve_batch(ktn.KatanaSolver(log_level=0), insts, fused=True, **kw)
    ok = sum(r["status"] == "Optimal" for r in res)
    err = max(abs(r["objval"] - i.opt_obj) / max(1, abs(i.opt_obj)) for r, i in zip(res, insts))
    print(json.dumps({"fused": True, "mode": mode, "instances": len(res), "optimal": ok, "wall_s_incl_load": wall,
                      "instances_per_s": len(res) / wall, "max_obj_relerr": err, "ecp_iters": res[0]["iters"],
                      "pdhg_iters": res[0]["pdhg_iters"], "blk_lp_launches": res[0]["blk_lp_launches"],
                      "ecp_blocks_launches": res[0]["ecp_blocks_launches"], "ecp_blocks_fallbacks": res[0]["ecp_blocks_fallbacks"],
                      "ecp_blocks_pdhg_sum": res[0]["ecp_blocks_pdhg_sum"]}), flush=True)
