"""Development aid: exercise the RCCL code path of the exchange with a one-rank nccl group."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ["KTN_FORCE_COLLECTIVE"] = "1"
import torch, torch.distributed as dist
import katana_jl_amd as ktn
from katana_jl_amd.distributed import ShardedKatanaModel
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
inst = ktn.instances.make_instance(n=4000, m_nl=400, k=16, family="explog", seed=21)
m = ShardedKatanaModel(ktn.KatanaSolver(log_level=0, device=0), inst, 0, 1, dist)
print("exchange device:", m.exchange_device)
print(m.optimize(), m.getobjval(), inst.opt_obj, m.numiters(), "exchanged rows", m.exchanged_rows)
dist.barrier(); dist.destroy_process_group()
