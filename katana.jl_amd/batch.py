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


def solve_batch(solver, instances, threads=16, describe=SeparableNLP, fused=False, per_instance_lp=False, device_loop=None):
    """Solve every instance; returns (results, wall_seconds).  results[i] = dict(status, objval, iters,
    numcuts, x) in the order of `instances`.

    fused=True solves the block-diagonal union of the instances as ONE problem (instances.fuse_instances): the
    cutting-plane loop, the sweep and every PDHG launch then serve the whole batch at once, and the stop rule --
    every nonlinear row of every instance within f_tol -- is the conjunction of the per-instance stop rules.
    Iteration counts are then those of the union (the slowest instance), objectives are split per instance.
    With per_instance_lp every LP re-solve of the fused problem is ONE launch with one workgroup per instance
    (ktn_set_blocks, csrc/batch_lp.hpp: iterates in LDS, restarts and termination per instance) instead of the global
    first-order loop.  Measured on 512 x cfg5: 2.5x fewer instance-iterations in total, but every
    cutting-plane round still waits for its slowest instance, so the batch takes 0.26 s against 0.22 s -- hence off by default.
    With device_loop (the default for fused batches) the WHOLE cutting-plane loop of every instance runs inside its own
    workgroup (ktn_optimize_blocks, csrc/batch_ecp.hpp): no instance waits for another; 0.10 s for 512 x cfg5.  Batches it does
    not cover (tape rows, nonlinear objective, free variables) fall back to the host-driven loop inside the call."""
    if device_loop is None:
        device_loop = fused and not per_instance_lp
    if fused:
        return _solve_fused(solver, instances, per_instance_lp, device_loop)
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


class FusedBatch:
    """A batch loaded ONCE as one block-diagonal problem (instances.fuse_instances + ktn_loadproblem [+ ktn_set_blocks]); solve()
    can then be called repeatedly -- every call after the first starts from the loaded state (ktn_reset) -- with the batch's
    data resident in HBM.  bench.py --workload cfg5 times solve() alone; solve_batch(fused=True) is load + one solve()."""

    def __init__(self, solver, instances, per_instance_lp=False, device_loop=True):
        self.device_loop = device_loop
        self.per_instance_lp = per_instance_lp
        self.m = NonlinearModel(solver)
        self.load(instances)

    def load(self, instances):
        """(re)load a batch on this handle: a process that serves batch after batch keeps its handle -- and with it the
        engine's device buffers -- instead of making a new one per batch (512 x cfg5: 0.14 s per batch against 0.20 s)"""
        import numpy as np
        from .instances import fuse_instances
        self.instances = instances
        self.big, self.offs = fuse_instances(instances)
        big = self.big
        self.m.loadproblem(big.n, big.num_constr, big.l_var, big.u_var, big.l_constr, big.u_constr, big.sense, SeparableNLP(big))
        if self.per_instance_lp or self.device_loop:
            self.m.set_blocks(self.offs)
        self._obj = (np.asarray(big.obj_kind), np.asarray(big.obj_p0), np.asarray(big.obj_p1), np.asarray(big.obj_col))
        self._solved = False
        return self

    def solve(self, cut_capacity=0):
        """cut_capacity: cuts per NL row an instance's arena has room for in the device-side loop (0: the engine's default, 12);
        an instance that outgrows it sends the batch to the host-driven loop (stat ecp_blocks_fallbacks)"""
        import numpy as np
        from .instances import atom_value_deriv
        m, big, offs = self.m, self.big, self.offs
        if self._solved:
            m.reset()
        self._solved = True
        status = m.optimize_blocks(cut_capacity) if self.device_loop else m.optimize()
        x = m.getsolution()
        # per-instance objectives: all atoms of the fused objective at once, then sums by instance
        kind, p0, p1, col = self._obj
        val, _ = atom_value_deriv(kind, p0, p1, x[col])
        optr = big.meta["obj_ptr"]
        seg = np.zeros(len(self.instances))
        nz = np.flatnonzero(np.diff(optr) > 0)                    # (instances with an empty objective are not reduceat starts)
        if len(nz):
            seg[nz] = np.add.reduceat(val, optr[:-1][nz])
        common = dict(status=status, iters=m.numiters(), numcuts=None, pdhg_iters=m.stat("pdhg_iters"),
                      blk_lp_launches=m.stat("blk_lp_launches"), blk_lp_fallbacks=m.stat("blk_lp_fallbacks"),
                      blk_pdhg_iters_sum=m.stat("blk_pdhg_iters_sum"), ecp_blocks_launches=m.stat("ecp_blocks_launches"),
                      ecp_blocks_fallbacks=m.stat("ecp_blocks_fallbacks"), ecp_blocks_pdhg_sum=m.stat("ecp_blocks_pdhg_sum"))
        return [dict(common, objval=float(seg[k] + inst.obj_const), x=x[offs[k]:offs[k + 1]]) for k, inst in enumerate(self.instances)]


def _solve_fused(solver, instances, per_instance_lp=False, device_loop=False):
    t0 = time.perf_counter()
    out = FusedBatch(solver, instances, per_instance_lp, device_loop).solve()
    return out, time.perf_counter() - t0


def shard_range(count, rank, world):
    """contiguous block of `count` items owned by `rank` (sizes differ by at most one)"""
    return (count * rank) // world, (count * (rank + 1)) // world


def solve_batch_sharded(solver, instances, rank, world, dist=None, gather=True, **kw):
    """Throughput mode over several GPUs (SURVEY.md section 8e "batch mode: replicas only, no communication"): rank r solves
    the contiguous block shard_range(len(instances), r, world) of the batch on ITS device as one fused batch (solve_batch,
    fused=True by default: every instance's loop inside its own workgroup).  The data path has no collective.  With
    gather=True the per-instance results are exchanged afterwards through `dist.all_gather_object` (host objects: statuses,
    objectives, solutions) so that every rank returns the results of the whole batch, in the order of `instances`; with
    gather=False a rank returns only its own block.  Returns (results, wall_seconds of this rank's solve)."""
    lo, hi = shard_range(len(instances), rank, world)
    kw.setdefault("fused", True)
    mine, wall = solve_batch(solver, instances[lo:hi], **kw) if hi > lo else ([], 0.0)
    if not gather or dist is None or world == 1:
        return mine, wall
    parts = [None] * world
    dist.all_gather_object(parts, mine)
    return [r for part in parts for r in part], wall
