"""dev: 512 x cfg5 as G fused groups, each on its own engine handle / stream / host thread (per-instance LP kernel or
global loop inside each group)."""
import json, os, sys, time
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import katana_jl_amd as ktn
from katana_jl_amd.batch import _solve_fused
nb = 512
insts = [ktn.instances.make_config("cfg5_one", seed=s) for s in range(nb)]
solver = ktn.KatanaSolver(log_level=0)
_solve_fused(solver, insts[:8], True)          # warm the process
for groups, per_inst in [(1, False), (1, True), (4, True), (8, True), (16, True), (32, True), (8, False), (16, False), (16, True)]:
    size = nb // groups
    parts = [insts[g * size:(g + 1) * size] for g in range(groups)]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=groups) as ex:
        outs = list(ex.map(lambda p: _solve_fused(solver, p, per_inst), parts))
    wall = time.perf_counter() - t0
    res = [r for o, _ in outs for r in o]
    ok = sum(r["status"] == "Optimal" for r in res)
    err = max(abs(r["objval"] - i.opt_obj) / max(1, abs(i.opt_obj)) for r, i in zip(res, insts))
    print(json.dumps({"groups": groups, "per_instance_lp": per_inst, "optimal": ok, "wall_s_incl_load": wall, "instances_per_s": nb / wall,
                      "max_obj_relerr": err, "rounds": [o[0]["iters"] for o, _ in outs][:8]}), flush=True)
