#!/usr/bin/env python3
"""bench.py -- ECP iterations/sec and wall-clock to f_tol = 1e-6 on BASELINE.json's workload.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the reference's hot loop (src/model.jl:258-308): re-solve the growing LP,
evaluate every nonlinear row at x*, append one tangent cut per violated row.  The workload at
N = 1 is BASELINE.json configs[2] ("cfg3": 1e5 variables, 5e4 linear rows, 1e4 exp/log rows with
32 non-zeros, seed 0, synthetic, planted optimum); when an instance converges (all rows within
f_tol) the engine is reset to its post-loadproblem! state and the next step starts the next solve,
so K steps are K consecutive iterations of back-to-back solves.  Inputs are resident in HBM before
the timed region (ktn_loadproblem copies them once).

Prints ONE JSON line (rank 0).  `value` = ECP iterations / second, whole job.
  roofline     -- the dominant kernel (k_pdhg_y: the A x SpMV + dual prox of the GPU LP), algorithmic
                  bytes per launch / mean launch duration from the start/stop hipEvents of
                  hipExtLaunchKernelGGL on the engine's own stream, in a second, identical pass over the
                  same K steps (profile=1).
  sweep_roofline -- the separator sweep (k_sep_eval_blk + k_sep_combine) on the HBM-resident variant of the workload
                  (cfg3_hbm: 2048 instead of 32 entries per NL row, 411 MB per pass), same timing.
  cpu_baseline -- the CPU oracle (serial restatement of the reference + HiGHS dual simplex, 1 core)
                  on a bounded sample: the same family at half scale, full solve to f_tol.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=33)
    ap.add_argument("--warmup", type=int, default=11)
    ap.add_argument("--workload", default="cfg3")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-sweep-roofline", action="store_true")
    return ap.parse_args()


def run_steps(model, nsteps, solve_log=None):
    """nsteps hot-loop passes; returns when exactly nsteps have run."""
    done_steps = 0
    t_solve = time.perf_counter()
    if not getattr(model, "_bench_begun", False):
        model.optimize_begin()
        model._bench_begun = True
    while done_steps < nsteps:
        finished = model.ecp_step()
        done_steps += 1
        if finished:
            status = model.optimize_end()
            if solve_log is not None:
                solve_log.append(dict(status=status, iters=model.numiters(), obj=model.getobjval(),
                                      wall=time.perf_counter() - t_solve, pdhg=model.stat("pdhg_iters")))
            model.reset()
            model.optimize_begin()
            t_solve = time.perf_counter()
    return done_steps


def cpu_baseline(args):
    import katana_jl_amd as ktn
    from oracle.evaluators import SeparableNLPEvaluator
    from oracle.katana import KatanaModelParams, KatanaNonlinearModel as OracleModel
    cfg = dict(ktn.instances.CONFIGS[args.workload])
    cfg["n"] //= 2
    cfg["m_nl"] //= 2
    inst = ktn.instances.make_instance(seed=args.seed, **cfg)
    d = SeparableNLPEvaluator(inst.n, inst.rowptr, inst.col, inst.kind, inst.p0, inst.p1, inst.rconst, inst.obj_col,
                              inst.obj_kind, inst.obj_p0, inst.obj_p1, inst.obj_const)
    om = OracleModel(KatanaModelParams(), fast=True)
    om.loadproblem(inst.n, inst.num_constr, inst.l_var, inst.u_var, inst.l_constr, inst.u_constr, inst.sense, d)
    t0 = time.perf_counter()
    status = om.optimize()
    wall = time.perf_counter() - t0
    # the same sample on the GPU, for a like-for-like ratio
    m = ktn.NonlinearModel(ktn.KatanaSolver(log_level=0))
    m.loadproblem(inst.n, inst.num_constr, inst.l_var, inst.u_var, inst.l_constr, inst.u_constr, inst.sense,
                  ktn.SeparableNLP(inst))
    m.optimize()
    t0 = time.perf_counter()
    m.reset()
    gstatus = m.optimize()
    gwall = time.perf_counter() - t0
    return {
        "value": om.numiters() / wall, "unit": "ECP iterations/s", "cores": 1, "kind": "port",
        "sample": "%s family at half scale (n=%d, m_lin=%d, m_nl=%d, k=%d, seed %d): one full solve to f_tol=1e-6; "
                  "oracle = serial CPU restatement of src/model.jl:219-319 + HiGHS dual simplex (SciPy 1.15.3), "
                  "warm-started, 1 thread" % (args.workload, inst.n, inst.m_lin, inst.m_nl, cfg["k"], args.seed),
        "status": status, "ecp_iters": om.numiters(), "wall_s": wall, "obj": om.getobjval(), "planted_obj": inst.opt_obj,
        "host_cores_available": os.cpu_count(),
        "gpu_on_same_sample": {"value": m.numiters() / gwall, "wall_s": gwall, "ecp_iters": m.numiters(),
                               "status": gstatus, "obj": m.getobjval()},
    }


def main():
    args = parse()
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        assert world == 1, "launch with torch.distributed.run --nproc-per-node == --gpus"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the Katana HIP engine has no CPU path")
    # KTN_BENCH_BACKEND=gloo rehearses the N > 1 path on a one-GPU box (all ranks share cuda:0)
    backend = os.environ.get("KTN_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    import katana_jl_amd as ktn
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    inst = ktn.instances.make_config(args.workload, seed=args.seed)
    if world > 1:
        from katana_jl_amd.distributed import ShardedKatanaModel
        model = ShardedKatanaModel(ktn.KatanaSolver(log_level=0, device=local_rank), inst, rank, world, dist)
    else:
        model = ktn.NonlinearModel(ktn.KatanaSolver(log_level=0, device=local_rank))
        model.loadproblem(inst.n, inst.num_constr, inst.l_var, inst.u_var, inst.l_constr, inst.u_constr, inst.sense,
                          ktn.SeparableNLP(inst))

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    run_steps(model, args.warmup)
    barrier()
    solves = []
    p0 = model.stat("pdhg_iters")
    t0 = time.perf_counter()
    run_steps(model, args.steps, solves)
    barrier()
    elapsed = time.perf_counter() - t0
    pdhg_timed = model.stat("pdhg_iters") - p0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # one complete solve from the loaded state: wall-clock to f_tol
    model.reset()
    model._bench_begun = False
    barrier()
    t1 = time.perf_counter()
    status = model.optimize()
    barrier()
    wall_to_ftol = time.perf_counter() - t1
    obj = model.getobjval()
    iters_to_ftol = model.numiters()

    roofline = None
    if world == 1 and not args.no_roofline:
        # identical second pass with per-launch hipEvent timing on the engine's stream
        prof = ktn.NonlinearModel(ktn.KatanaSolver(log_level=0, device=local_rank, profile=1))
        prof.loadproblem(inst.n, inst.num_constr, inst.l_var, inst.u_var, inst.l_constr, inst.u_constr, inst.sense,
                         ktn.SeparableNLP(inst))
        run_steps(prof, args.warmup)
        keys = [p + s for p in ("ky", "kx", "sweep_eval") for s in ("_time_s", "_launches", "_bytes")]
        base = {k: prof.stat(k) for k in keys}
        run_steps(prof, args.steps)
        d = {k: prof.stat(k) - v for k, v in base.items()}

        def rf(prefix, kernel):
            n = max(d[prefix + "_launches"], 1.0)
            avg_t = d[prefix + "_time_s"] / n
            avg_b = d[prefix + "_bytes"] / n
            ach = avg_b / avg_t / 1e9 if avg_t > 0 else 0.0
            return {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                    "traffic": None, "kernel": kernel, "launches": int(n), "avg_launch_us": avg_t * 1e6,
                    "algorithmic_bytes_per_launch": avg_b,
                    "timing": "per launch, hipExtLaunchKernelGGL start/stop hipEvents on the engine's own stream "
                              "(the dispatch's begin/end timestamps, as rocprofv3 --kernel-trace reports them)"}
        roofline = rf("ky", "k_pdhg_y (A x SpMV + dual prox + Halpern update)")
        # HBM traffic from the PMC counters cannot be collected in-process; the per-launch figure of the
        # committed rocprofv3 --pmc passes over this same command is reported (profiles/r01_traffic.json)
        tpath = os.path.join(ROOT, "profiles", "r01_traffic.json")
        if os.path.exists(tpath) and args.workload == "cfg3":
            t = json.load(open(tpath))
            roofline["traffic"] = t["k_pdhg_y"]["bytes_per_launch"]
            roofline["traffic_source"] = "profiles/r01_traffic.json (rocprofv3 --pmc FETCH_SIZE, --pmc WRITE_SIZE; raw sum, see its _note)"
        roofline["other_kernels"] = {
            "k_pdhg_x": rf("kx", "k_pdhg_x (A'y SpMV + primal prox + Halpern update)"),
            "k_sep_eval": rf("sweep_eval", "k_sep_eval (separator sweep: g, cut constant, violation)"),
        }

    # the separator sweep on the HBM-resident variant of the same workload (k = 2048 entries per
    # NL row, 411 MB per pass); at cfg3's own k = 32 the sweep is a 10 us launch-latency-bound kernel.
    sweep_roofline = None
    if world == 1 and not args.no_roofline and not args.no_sweep_roofline and args.workload == "cfg3":
        hb = ktn.instances.make_config("cfg3_hbm", seed=args.seed, vertex=False)
        sm = ktn.NonlinearModel(ktn.KatanaSolver(log_level=0, device=local_rank, profile=1))
        sm.loadproblem(hb.n, hb.num_constr, hb.l_var, hb.u_var, hb.l_constr, hb.u_constr, hb.sense, ktn.SeparableNLP(hb))
        sep = ktn.KatanaHipSeparator(sm); sep.initialize()
        xs = np.clip(hb.xhat + 0.05, hb.l_var, hb.u_var)
        sep.precompute(xs)
        for _ in range(3):
            sep.sweep(1e-6)
        b0 = {k: sm.stat(k) for k in ("sweep_eval_time_s", "sweep_eval_launches", "sweep_eval_bytes")}
        for _ in range(20):
            sep.sweep(1e-6)
        dt, dn, db = (sm.stat(k) - b0[k] for k in ("sweep_eval_time_s", "sweep_eval_launches", "sweep_eval_bytes"))
        ach = db / dt / 1e9
        sweep_roofline = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                          "kernel": "k_sep_eval_blk + k_sep_combine (column-blocked separator sweep)",
                          "workload": "cfg3_hbm: n=%d, m_nl=%d exp/log rows, k=%d" % (hb.n, hb.m_nl, hb.meta["k"]),
                          "launches": int(dn), "avg_launch_us": 1e6 * dt / dn, "algorithmic_bytes_per_launch": db / dn}
        tpath = os.path.join(ROOT, "profiles", "r01_traffic.json")
        if os.path.exists(tpath):
            t = json.load(open(tpath)).get("k_sep_eval_blk")
            if t:
                sweep_roofline["traffic"] = t["bytes_per_launch"]
                sweep_roofline["traffic_source"] = "profiles/r01_traffic.json (rocprofv3 --pmc FETCH_SIZE, x2 gfx950 correction for 16-B/lane streams)"
        del sm, sep, hb

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args)

    if rank == 0:
        out = {
            "metric": "ECP iterations/sec + wall-clock to f_tol=1e-6, 1e5-var synthetic convex NLP",
            "value": args.steps / elapsed, "unit": "ECP iterations/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s: n=%d variables, m_lin=%d linear rows (8 nnz), m_nl=%d %s rows (k=%d nnz), seed %d, "
                                   "planted non-degenerate vertex optimum, f_tol=1e-6" % (
                                       args.workload, inst.n, inst.m_lin, inst.m_nl, inst.meta["family"], inst.meta["k"],
                                       args.seed),
                       "parallelism": "1 GPU" if world == 1 else "nl-rows sharded x%d, replicated LP, all-gather of cuts" % world},
            "wall_to_ftol_s": wall_to_ftol, "ecp_iters_to_ftol": iters_to_ftol, "status": status, "objective": obj,
            "planted_objective": inst.opt_obj, "objective_relerr": abs(obj - inst.opt_obj) / max(1.0, abs(inst.opt_obj)),
            "pdhg_iters_per_step": pdhg_timed / args.steps, "solves_in_timed_region": len(solves),
            "roofline": roofline, "sweep_roofline": sweep_roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
