"""dev: time split of the fused batch mode (512 x cfg5): instance fusion, loadproblem, optimize (LP / sweep / setup)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import katana_jl_amd as ktn
from katana_jl_amd.instances import fuse_instances
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 512
insts = [ktn.instances.make_config("cfg5_one", seed=s) for s in range(nb)]
for per_inst in ("device", False, "device", True):
    t0 = time.perf_counter(); big, offs = fuse_instances(insts); t1 = time.perf_counter()
    m = ktn.NonlinearModel(ktn.KatanaSolver(log_level=0, **{k: (float(v) if "." in v or "e" in v else int(v)) for k, v in (a.split("=") for a in sys.argv[2:])}))
    desc = ktn.SeparableNLP(big); t2 = time.perf_counter()
    m.loadproblem(big.n, big.num_constr, big.l_var, big.u_var, big.l_constr, big.u_constr, big.sense, desc); t3 = time.perf_counter()
    if per_inst:
        m.set_blocks(offs)
    st = m.optimize_blocks() if per_inst == "device" else m.optimize(); t4 = time.perf_counter()
    x = m.getsolution(); t5 = time.perf_counter()
    print(json.dumps({"per_instance_lp": per_inst, "status": st, "fuse_s": t1 - t0, "describe_s": t2 - t1, "load_s": t3 - t2, "optimize_s": t4 - t3,
                      "getsolution_s": t5 - t4, "total_s": t5 - t0, "instances_per_s": nb / (t5 - t0), "ecp_iters": m.numiters(),
                      "lp_s": m.stat("lp_time_s"), "lp_setup_s": m.stat("lp_setup_time_s"), "sep_s": m.stat("sep_time_s"),
                      "lp_solves": m.stat("lp_solves"), "pdhg_iters": m.stat("pdhg_iters"), "blk_iters_sum": m.stat("blk_pdhg_iters_sum"),
                      "blk_iters_max_sum": m.stat("blk_pdhg_iters_max"), "lp_rows": m.lp_num_rows(), "ecp_pdhg_sum": m.stat("ecp_blocks_pdhg_sum"), "ecp_fallbacks": m.stat("ecp_blocks_fallbacks"), "ecp_rows": m.stat("ecp_blocks_rows")}), flush=True)
