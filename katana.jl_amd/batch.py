"""Throughput mode: a batch of independent convex NLPs (BASELINE.json configs[4]; SURVEY.md section 8e
"replicas only": no communication).

Every instance gets its own engine handle, hence its own HIP stream; `threads` host threads drive
the handles concurrently (ctypes releases the GIL for the duration of every ktn_* call), so the GPU
sees up to `threads` independent kernel streams at once -- small instances (n ~ 1e3) occupy a few
workgroups each and many of them fit on the 256 CUs side by side."""
import time
from concurrent.futures import ThreadPoolExecutor

from .nlp import SeparableNLP
from .solver import NonlinearModel


def solve_batch(solver, instances, threads=16, describe=SeparableNLP):
    """Solve every instance; returns (results, wall_seconds).  results[i] = dict(status, objval, iters,
    numcuts, x) in the order of `instances`."""
    def work(inst):
        m = NonlinearModel(solver)
        m.loadproblem(inst.n, inst.num_constr, inst.l_var, inst.u_var, inst.l_constr, inst.u_constr, inst.sense,
                      describe(inst))
        status = m.optimize()
        return dict(status=status, objval=m.getobjval(), iters=m.numiters(), numcuts=m.numcuts(), x=m.getsolution(),
                    pdhg_iters=m.stat("pdhg_iters"))

    t0 = time.perf_counter()
    if threads <= 1:
        out = [work(i) for i in instances]
    else:
        with ThreadPoolExecutor(max_workers=threads) as ex:
            out = list(ex.map(work, instances))
    return out, time.perf_counter() - t0
