"""dev: cost of the host-driven sharded loop (1 rank, RCCL collectives forced) against the engine's own loop on cfg3."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
os.environ["KTN_FORCE_COLLECTIVE"] = "1"
import torch, torch.distributed as dist
import katana_jl_amd as ktn
from katana_jl_amd.distributed import ShardedKatanaModel
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
inst = ktn.instances.make_config("cfg3", seed=0)
a = ShardedKatanaModel(ktn.KatanaSolver(log_level=0, device=0), inst, 0, 1, dist)
b = ktn.NonlinearModel(ktn.KatanaSolver(log_level=0, device=0))
b.loadproblem(inst.n, inst.num_constr, inst.l_var, inst.u_var, inst.l_constr, inst.u_constr, inst.sense, ktn.SeparableNLP(inst))
for name, m in (("sharded loop (1 rank, nccl)", a), ("engine loop", b)):
    m.optimize(); m.reset()
    ts = []
    for _ in range(3):
        t = time.perf_counter(); st = m.optimize(); ts.append(time.perf_counter() - t); it = m.numiters(); m.reset()
    print("%-30s %s iters=%d  %.4f s (best of 3)" % (name, st, it, min(ts)))
dist.barrier(); dist.destroy_process_group()
