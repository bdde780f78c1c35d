import sys, time
sys.path.insert(0,'/root/repo')
import numpy as np
import katana_jl_amd as ktn
from katana_jl_amd.batch import FusedBatch
from katana_jl_amd.instances import fuse_instances
from katana_jl_amd.nlp import SeparableNLP
insts_a=[ktn.instances.make_config("cfg5_one", seed=s) for s in range(512)]
insts_b=[ktn.instances.make_config("cfg5_one", seed=1000+s) for s in range(512)]
fb=FusedBatch(ktn.KatanaSolver(log_level=0), insts_a); r=fb.solve()
for rep in range(3):
    for insts in (insts_b, insts_a):
        t0=time.perf_counter()
        big, offs = fuse_instances(insts); t1=time.perf_counter()
        d=SeparableNLP(big); t2=time.perf_counter()
        fb.m.loadproblem(big.n, big.num_constr, big.l_var, big.u_var, big.l_constr, big.u_constr, big.sense, d); t3=time.perf_counter()
        fb.m.set_blocks(offs); st=fb.m.optimize_blocks(); x=fb.m.getsolution(); t4=time.perf_counter()
        print("fuse %.3f describe %.3f load %.3f solve %.3f total %.3f %s" % (t1-t0,t2-t1,t3-t2,t4-t3,t4-t0,st), flush=True)
    t0=time.perf_counter(); r2=FusedBatch(ktn.KatanaSolver(log_level=0), insts_a).solve(); print("fresh handle total %.3f"%(time.perf_counter()-t0))
