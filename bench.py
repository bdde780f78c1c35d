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

With --gpus N > 1 (one rank per GPU, torch.distributed.run) the workload stays the SAME (strong scaling: the per-N values
are one curve).  How the path is split depends on the workload (--lp-layout auto):
  replicated   cfg3 and smaller: the NL rows are block-sharded over the ranks, every rank sweeps its block at x*, the cuts of
               a sweep travel in ONE device-resident RCCL all-gather per ECP iteration and every rank appends all of them, in
               rank order, to its own copy of the LP (katana.jl_amd/distributed.py::ShardedKatanaModel; north_star's design).
               A cfg3 PDHG iteration is two launch-floor-bound kernels (14 us): there is nothing in it a second GPU can take.
  row-sharded  --workload cfg4 (BASELINE.json configs[3], 1e6 exp/log rows): the LP ROW-SHARDED over the ranks
               (RowShardedKatanaModel): every rank keeps its own cuts, A'y costs one RCCL all-reduce of an n-vector per
               PDHG iteration.
The line then also carries the per-phase split, the measured latency of one all-reduce of an n-vector on the fabric and the
same workload timed on rank 0's GPU alone in the same run.

Prints ONE JSON line (rank 0).  `value` = ECP iterations / second, whole job.
  roofline     -- the dominant kernel = whichever of k_pdhg_x / k_pdhg_y has the larger total time in the run
                  (algorithmic bytes per launch / mean launch duration from the start/stop hipEvents of
                  hipExtLaunchKernelGGL on the engine's own stream, in a second, identical pass over the same K steps
                  with profile=1); the other LP kernel and the sweep kernel are listed under other_kernels.
  sweep_roofline -- the separator sweep (k_sep_eval_blk + k_sep_combine) on the HBM-resident variant of the workload
                  (cfg3_hbm: 2048 instead of 32 entries per NL row, 411 MB per pass), same timing.
  spmv_roofline -- the LP SpMV steps on an HBM-resident cut matrix (cfg4's LP after one un-capped sweep: >= 1.2e7
                  non-zeros, CSR + CSC mirror beyond the Infinity Cache), same timing.
  --workload cfg5 -- BASELINE.json configs[4] instead: the batch of 512 independent instances in throughput mode, split into one
                  contiguous block per GPU with no communication; `value` is then instances/s (its own `metric` string).
  stream_ceiling -- torch.sum / copy_ over 1 GiB in the same run: the box's practical read and copy rates next to the 8 TB/s
                  spec figure the fractions are quoted against.
  cpu_baseline -- the CPU oracle (serial restatement of the reference + HiGHS dual simplex, 1 core) on the SAME
                  configuration as `value` (full cfg3: one solve to f_tol, about 100 s of one host core).
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
PROFILE_ROUND = "r04"      # profiles/<round>_kernel_stats.csv, <round>_traffic.json: the rocprofv3 summaries of THIS command


def csrc_sha16():
    """Content hash of the kernel sources (katana.jl_amd/csrc/*.hip, *.hpp): profiles/<round>_source.json records the hash (and
    the git commit) the committed rocprofv3 summaries were made from -- tools/stamp_profiles.py -- and a summary whose hash is
    not the hash of the sources this run was built from is STALE and is not quoted."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "katana.jl_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "katana.jl_amd", "csrc", "*.hpp"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def profile_stamp():
    path = os.path.join(ROOT, "profiles", PROFILE_ROUND + "_source.json")
    if not os.path.exists(path):
        return None
    st = json.load(open(path))
    st["fresh"] = st.get("csrc_sha16") == csrc_sha16()
    return st


def rocprof_avg_ns(csv_name, kernel_prefix):
    """Call-weighted mean duration (ns) of the kernels whose name starts with `kernel_prefix` in a committed
    rocprofv3 --kernel-trace --stats summary (profiles/<csv_name>); None when the file or the kernel is absent."""
    import csv
    path = os.path.join(ROOT, "profiles", csv_name)
    if not os.path.exists(path):
        return None
    calls = total = 0.0
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row.get("Name", "")
            if name.startswith("void ktn::" + kernel_prefix) or name.startswith("ktn::" + kernel_prefix) or name.startswith(kernel_prefix):
                calls += float(row["Calls"])
                total += float(row["TotalDurationNs"])
    return total / calls if calls else None


def with_rocprof(r, csv_name, kernel_prefix):
    """`achieved` / `frac` of a roofline record stay what THIS run measured (per-launch hipEvents on the engine's stream).  Beside
    them, as `frac_rocprof_committed`, the same algorithmic bytes over the mean duration of the kernel in the committed
    rocprofv3 summary -- only when profiles/<round>_source.json says the summary was made from the very sources this run was
    built from (csrc_sha16); a stale summary is named as such and not quoted."""
    r["achieved_inprocess"], r["frac_inprocess"] = r["achieved"], r["frac"]      # (the names of rounds 2-3, kept for the record)
    r["frac_source"] = "this run: algorithmic bytes per launch / mean launch duration, hipExtLaunchKernelGGL start/stop events on the engine's stream"
    st = profile_stamp()
    ns = rocprof_avg_ns(csv_name, kernel_prefix)
    r["frac_rocprof_committed"] = None
    if ns and st and st["fresh"]:
        r["avg_launch_us_rocprof"] = ns / 1e3
        r["frac_rocprof_committed"] = r["algorithmic_bytes_per_launch"] / (ns * 1e-9) / 1e9 / HBM_PEAK_GBS
        r["rocprof_source"] = "profiles/%s, made at commit %s from sources %s (= this build)" % (csv_name, st.get("git_head"), st.get("csrc_sha16"))
    elif ns and st:
        r["rocprof_source"] = "profiles/%s is STALE: made from sources %s (commit %s), this build is %s -- not quoted" % (
            csv_name, st.get("csrc_sha16"), st.get("git_head"), csrc_sha16())
    else:
        r["rocprof_source"] = "no profiles/%s (or no %s_source.json stamp)" % (csv_name, PROFILE_ROUND)
    return r


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=33)
    ap.add_argument("--warmup", type=int, default=11)
    ap.add_argument("--workload", default="cfg3", help="cfg3 (BASELINE.json's metric), cfg2, cfg4, cfg5, cfg3_qp ...")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cpu-baseline-mt", action="store_true", help="skip the second CPU line (HiGHS default threading)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-sweep-roofline", action="store_true")
    ap.add_argument("--no-spmv-roofline", action="store_true")
    ap.add_argument("--cpu-baseline-scale", type=float, default=1.0, help="1.0 = the configuration of `value`; 0.5 = half scale (quick)")
    ap.add_argument("--lp-layout", choices=("auto", "replicated", "row-sharded"), default="auto",
                    help="N > 1: replicated LP + all-gather of cuts, or row-sharded LP + all-reduce per PDHG iteration; "
                         "auto = row-sharded for cfg4, replicated otherwise")
    ap.add_argument("--replicated-lp", action="store_true", help="same as --lp-layout replicated")
    ap.add_argument("--transport", choices=("auto", "rccl", "ipc", "callback"), default="auto",
                    help="row-sharded layout: RCCL all-reduce (default over nccl), peer-buffer transport (ktn_dist_init_ipc), or the "
                         "host callback (gloo rehearsals)")
    a = ap.parse_args()
    if a.lp_layout == "auto":
        a.lp_layout = "row-sharded" if a.workload == "cfg4" else "replicated"
    a.replicated_lp = a.replicated_lp or a.lp_layout == "replicated"
    return a


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
    cfg["n"] = int(cfg["n"] * args.cpu_baseline_scale)
    cfg["m_nl"] = int(cfg["m_nl"] * args.cpu_baseline_scale)
    inst = ktn.instances.make_instance(seed=args.seed, **cfg)
    d = SeparableNLPEvaluator(inst.n, inst.rowptr, inst.col, inst.kind, inst.p0, inst.p1, inst.rconst, inst.obj_col,
                              inst.obj_kind, inst.obj_p0, inst.obj_p1, inst.obj_const)
    om = OracleModel(KatanaModelParams(), fast=True)
    om.loadproblem(inst.n, inst.num_constr, inst.l_var, inst.u_var, inst.l_constr, inst.u_constr, inst.sense, d)
    t0 = time.perf_counter()
    status = om.optimize()
    wall = time.perf_counter() - t0
    # "also report HiGHS with its default threading as a second, more favourable CPU line"
    mt = None
    if not args.no_cpu_baseline_mt:
        om2 = OracleModel(KatanaModelParams(), fast=True, lp_threads=0)
        om2.loadproblem(inst.n, inst.num_constr, inst.l_var, inst.u_var, inst.l_constr, inst.u_constr, inst.sense, d)
        t0 = time.perf_counter()
        st2 = om2.optimize()
        w2 = time.perf_counter() - t0
        mt = {"value": om2.numiters() / w2, "unit": "ECP iterations/s", "cores": os.cpu_count(), "kind": "port",
              "sample": "the same solve with HiGHS left to its default threading and simplex strategy (threads = 0: automatic, "
                        "%d host cores visible)" % os.cpu_count(),
              "status": st2, "ecp_iters": om2.numiters(), "wall_s": w2, "obj": om2.getobjval()}
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
        "sample": "%s %s (n=%d, m_lin=%d, m_nl=%d, k=%d, seed %d): one full solve to f_tol=1e-6; "
                  "oracle = serial CPU restatement of src/model.jl:219-319 + HiGHS dual simplex (SciPy 1.15.3), "
                  "warm-started, 1 thread" % (args.workload, "at full size: the configuration of `value`" if args.cpu_baseline_scale == 1.0
                                              else "family at scale %g" % args.cpu_baseline_scale, inst.n, inst.m_lin, inst.m_nl,
                                              cfg["k"], args.seed),
        "status": status, "ecp_iters": om.numiters(), "wall_s": wall, "obj": om.getobjval(), "planted_obj": inst.opt_obj,
        "host_cores_available": os.cpu_count(), "default_threading": mt,
        "gpu_on_same_sample": {"value": m.numiters() / gwall, "wall_s": gwall, "ecp_iters": m.numiters(),
                               "status": gstatus, "obj": m.getobjval()},
    }


def main_batch(args, ktn, torch, dist, rank, world, local_rank, backend):
    """--workload cfg5: BASELINE.json configs[4], the batch of 512 independent 1e3-variable instances in throughput mode.  With
    N > 1 ranks the batch is split into N contiguous blocks, one per GPU, with no communication on the data path (SURVEY.md
    section 8e "replicas only").  Every rank loads its block ONCE (instance fusion + ktn_loadproblem + ktn_set_blocks: the
    batch's data is resident in HBM when the timed region starts); a step is one solve of the whole batch from the loaded
    state -- ktn_reset, the device-side loop of every instance, the solution read back and split per instance.  `value` =
    instances solved per second by the whole job; `incl_load` reports the same with the host-side load inside the clock."""
    nb = 512
    insts = [ktn.instances.make_config("cfg5_one", seed=s) for s in range(nb)]
    from katana_jl_amd.batch import FusedBatch, shard_range
    lo, hi = shard_range(nb, rank, world)
    mine = insts[lo:hi]

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    t_load = time.perf_counter()
    fb = FusedBatch(ktn.KatanaSolver(log_level=0, device=local_rank), mine) if mine else None
    load_s = time.perf_counter() - t_load
    for _ in range(args.warmup):
        if fb: fb.solve()
    barrier()
    t0 = time.perf_counter()
    ok, worst = True, 0.0
    for _ in range(args.steps):
        res = fb.solve() if fb else []
        for r, inst in zip(res, mine):
            ok = ok and r["status"] == "Optimal"
            worst = max(worst, abs(r["objval"] - inst.opt_obj) / max(1.0, abs(inst.opt_obj)))
    barrier()
    elapsed = time.perf_counter() - t0
    # the same with the load inside the clock (three passes: fusion + description + ktn_loadproblem + solve)
    barrier()
    t1 = time.perf_counter()
    for _ in range(3):
        if fb: fb.load(mine).solve()                  # (a serving process keeps its handle: FusedBatch.load)
    barrier()
    incl = (time.perf_counter() - t1) / 3
    flags = torch.tensor([elapsed, 0.0 if ok else 1.0, worst, incl, load_s], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
    if dist is not None:
        dist.all_reduce(flags, op=dist.ReduceOp.MAX)
    elapsed, bad, worst, incl, load_s = (float(v) for v in flags.tolist())
    if rank == 0:
        print(json.dumps({
            "metric": "instances/s, batch of 512 independent 1e3-var convex NLPs to f_tol=1e-6 (throughput mode, BASELINE.json configs[4])",
            "value": nb * args.steps / elapsed, "unit": "instances/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "cfg5: 512 x (n=1000 variables, m_lin=500 linear rows, m_nl=100 explog rows, k=16), seeds 0..511, "
                                   "planted optima; one fused batch per GPU, loaded once, every instance's loop in its own workgroup",
                       "parallelism": "1 GPU" if world == 1 else "%d contiguous blocks of the batch, one per GPU, no communication" % world},
            "status": "Optimal" if bad == 0.0 else "some instance not Optimal", "max_objective_relerr": worst,
            "incl_load": {"instances_per_s": nb / incl, "s_per_batch": incl, "first_load_s": load_s,
                          "what": "instance fusion + row-program description + ktn_loadproblem + ktn_set_blocks on the same handle + solve, per batch"},
            "roofline": None, "cpu_baseline": None}))
    if dist is not None:
        dist.destroy_process_group()


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
    # KTN_FORCE_COLLECTIVE=1 with one rank: the N > 1 code path (process group, sharded model, collectives, multi_gpu block) on
    # a one-GPU box -- every gather returns the rank's own block; a rehearsal of the nccl branch, not a measurement
    multi_path = world > 1 or bool(os.environ.get("KTN_FORCE_COLLECTIVE"))
    if multi_path:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    if args.workload == "cfg5":
        return main_batch(args, ktn, torch, dist, rank, world, local_rank, backend)
    inst = ktn.instances.make_config(args.workload, seed=args.seed)
    if multi_path and args.replicated_lp:
        from katana_jl_amd.distributed import ShardedKatanaModel
        model = ShardedKatanaModel(ktn.KatanaSolver(log_level=0, device=local_rank), inst, rank, world, dist)
    elif multi_path:
        from katana_jl_amd.distributed import RowShardedKatanaModel
        model = RowShardedKatanaModel(ktn.KatanaSolver(log_level=0, device=local_rank), inst, rank, world, dist, transport=args.transport)
    else:
        # What a user of the reference pays (soltime spans all of optimize!, src/model.jl:227,311, and a JuMP `solve` pays
        # loadproblem! every time, src/model.jl:81-173): a fresh handle, the load and the first solve, COLD -- before any warm-up,
        # first touch of the device allocator included.  The timed steps below then run from this handle's loaded state.
        torch.cuda.synchronize()
        tc0 = time.perf_counter()
        nlp = ktn.SeparableNLP(inst)
        tc1 = time.perf_counter()
        model = ktn.NonlinearModel(ktn.KatanaSolver(log_level=0, device=local_rank))
        model.loadproblem(inst.n, inst.num_constr, inst.l_var, inst.u_var, inst.l_constr, inst.u_constr, inst.sense, nlp)
        tc2 = time.perf_counter()
        st_first = model.optimize()
        tc3 = time.perf_counter()
        cold = {"first_solve_s": tc3 - tc1, "first_load_s": tc2 - tc1, "first_optimize_s": tc3 - tc2, "describe_s": tc1 - tc0,
                "status": st_first, "ecp_iters": model.numiters(),
                "what": "fresh ktn_create + ktn_loadproblem + ktn_optimize on a cold process, host buffers in, before any warm-up"}
        # ... and the same load with the allocator warm (a second problem on a live process): median of 3 on a second handle
        m2 = ktn.NonlinearModel(ktn.KatanaSolver(log_level=0, device=local_rank))
        loads = []
        for _ in range(3):
            torch.cuda.synchronize()
            tl = time.perf_counter()
            m2.loadproblem(inst.n, inst.num_constr, inst.l_var, inst.u_var, inst.l_constr, inst.u_constr, inst.sense, nlp)
            loads.append(time.perf_counter() - tl)
        cold["load_s"] = sorted(loads)[1]
        del m2
        # ... and what a JuMP user pays per solve (a NEW model, src/model.jl:63-65): a fresh handle in this warm process (~100
        # device buffers allocated anew; a caching block pool was tried in round 4 and was SLOWER -- 28 against 15 ms -- because
        # giving a block back needs the device synchronisation hipFree does implicitly, for every buffer)
        fresh = []
        for _ in range(3):
            torch.cuda.synchronize()
            tl = time.perf_counter()
            m3 = ktn.NonlinearModel(ktn.KatanaSolver(log_level=0, device=local_rank))
            m3.loadproblem(inst.n, inst.num_constr, inst.l_var, inst.u_var, inst.l_constr, inst.u_constr, inst.sense, nlp)
            fresh.append(time.perf_counter() - tl)
            del m3
        cold["load_new_handle_s"] = sorted(fresh)[1]
        model.reset()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if multi_path:
        cold = None
    run_steps(model, args.warmup)
    barrier()
    solves = []
    phase_keys = ("pdhg_iters", "lp_time_s", "sep_time_s", "lp_setup_time_s", "allreduce_calls", "allreduce_bytes")
    ph0 = {k: model.stat(k) for k in phase_keys}
    t0 = time.perf_counter()
    run_steps(model, args.steps, solves)
    barrier()
    elapsed = time.perf_counter() - t0
    phases = {k: model.stat(k) - ph0[k] for k in phase_keys}
    pdhg_timed = phases["pdhg_iters"]
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
    if not multi_path and not args.no_roofline:
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
        names = {"ky": "k_pdhg_y (A x SpMV + dual prox + Halpern update)",
                 "kx": "k_pdhg_x (A'y SpMV + primal prox + Halpern update)"}
        dom = "kx" if d["kx_time_s"] >= d["ky_time_s"] else "ky"          # the kernel with the largest total time in the run
        oth = "ky" if dom == "kx" else "kx"
        roofline = rf(dom, names[dom])
        roofline["total_time_s_in_run"] = d[dom + "_time_s"]
        if args.workload == "cfg3":
            with_rocprof(roofline, PROFILE_ROUND + "_kernel_stats.csv", "k_pdhg_x_packed" if dom == "kx" else "k_pdhg_y_packed")
        # HBM traffic from the PMC counters cannot be collected in-process; the per-launch figure of the committed
        # rocprofv3 --pmc passes over this same command is reported (profiles/r02_traffic.json, regenerated every round)
        tpath = os.path.join(ROOT, "profiles", PROFILE_ROUND + "_traffic.json")
        if os.path.exists(tpath) and args.workload == "cfg3" and (profile_stamp() or {}).get("fresh"):
            t = json.load(open(tpath))
            key = "k_pdhg_x" if dom == "kx" else "k_pdhg_y"
            if key in t:
                roofline["traffic"] = t[key]["bytes_per_launch"]
                roofline["traffic_source"] = "profiles/%s_traffic.json (rocprofv3 --pmc FETCH_SIZE, --pmc WRITE_SIZE, separate passes; see its _note)" % PROFILE_ROUND
        roofline["other_kernels"] = {
            ("k_pdhg_x" if oth == "kx" else "k_pdhg_y"): rf(oth, names[oth]),
            "k_sep_eval": rf("sweep_eval", "k_sep_eval (separator sweep: g, cut constant, violation)"),
        }

    # the separator sweep on the HBM-resident variant of the same workload (k = 2048 entries per
    # NL row, 411 MB per pass); at cfg3's own k = 32 the sweep is a 10 us launch-latency-bound kernel.
    sweep_roofline = None
    if not multi_path and not args.no_roofline and not args.no_sweep_roofline and args.workload == "cfg3":
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
                          "launches": int(dn), "avg_launch_us": 1e6 * dt / dn, "algorithmic_bytes_per_launch": db / dn, "traffic": None}
        tpath = os.path.join(ROOT, "profiles", PROFILE_ROUND + "_traffic.json")
        if os.path.exists(tpath):
            t = json.load(open(tpath)).get("k_sep_eval_blk")
            if t:
                sweep_roofline["traffic"] = t["bytes_per_launch"]
                sweep_roofline["traffic_source"] = "profiles/%s_traffic.json (rocprofv3 --pmc FETCH_SIZE, x2 gfx950 correction for 16-B/lane streams)" % PROFILE_ROUND
        # the sweep is two kernels per pass (blocked evaluation + combination): the rocprofv3 figure is the sum of their means
        ns_blk, ns_cmb = rocprof_avg_ns(PROFILE_ROUND + "_sweep_hbm_kernel_stats.csv", "k_sep_eval_blk"), rocprof_avg_ns(PROFILE_ROUND + "_sweep_hbm_kernel_stats.csv", "k_sep_combine")
        sweep_roofline["achieved_inprocess"], sweep_roofline["frac_inprocess"] = sweep_roofline["achieved"], sweep_roofline["frac"]
        sweep_roofline["frac_source"] = "this run (per-launch hipEvents)"
        st = profile_stamp()
        sweep_roofline["frac_rocprof_committed"] = None
        if ns_blk and ns_cmb and st and st["fresh"]:
            sweep_roofline["avg_launch_us_rocprof"] = (ns_blk + ns_cmb) / 1e3
            sweep_roofline["frac_rocprof_committed"] = sweep_roofline["algorithmic_bytes_per_launch"] / ((ns_blk + ns_cmb) * 1e-9) / 1e9 / HBM_PEAK_GBS
            sweep_roofline["rocprof_source"] = "profiles/%s_sweep_hbm_kernel_stats.csv (k_sep_eval_blk + k_sep_combine), commit %s" % (PROFILE_ROUND, st.get("git_head"))
        del sm, sep, hb

    # The box's practical streaming ceilings next to the 8 TB/s spec figure `peak` ("quote the copy-kernel
    # ceiling measured in the same run"): a read-only reduction and a copy over 1 GiB, plain torch kernels
    stream_ceiling = None
    if not multi_path and not args.no_roofline and rank == 0:
        nel = (1 << 30) // 8
        xs_ = torch.ones(nel, dtype=torch.float64, device="cuda")
        ys_ = torch.empty_like(xs_)
        stream_ceiling = {"unit": "GB/s", "working_set": "1 GiB f64 (beyond the 256 MiB Infinity Cache)", "kernels": "torch.sum / Tensor.copy_"}
        for name, fn, nbytes in (("read", lambda: xs_.sum(), nel * 8), ("copy", lambda: ys_.copy_(xs_), 2 * nel * 8)):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record()
            torch.cuda.synchronize()
            stream_ceiling[name] = nbytes * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del xs_, ys_

    # The LP SpMV steps in the HBM regime (tools/spmv_bench.py): cfg4's LP after one un-capped sweep
    spmv_roofline = None
    sweep_short = None
    if not multi_path and not args.no_roofline and not args.no_spmv_roofline and args.workload == "cfg3":
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import spmv_bench
        sinst, sm, nviol = spmv_bench.build(1_000_000, device=local_rank)
        M, nnz = sm.lp_num_rows(), int(sm._lib.ktn_lp_nnz(sm._h))
        x0, y0 = np.zeros(sm.num_var), np.zeros(M)
        sm.lp_pdhg_raw(x0, y0, 1e-3, 1.0, 4)
        keys = [p + q for p in ("kx", "ky") for q in ("_time_s", "_launches", "_bytes")]
        b0 = {k: sm.stat(k) for k in keys}
        sm.lp_pdhg_raw(x0, y0, 1e-3, 1.0, 40)
        dd = {k: sm.stat(k) - b0[k] for k in keys}
        spmv_roofline = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": None,
                         "workload": "cfg4 LP after one un-capped sweep: n=%d, rows=%d (%d cuts), nnz=%d: CSR + CSC mirror %.0f MB" % (
                             sinst.n, M, nviol, nnz, 2 * nnz * 12 / 1e6),
                         "tiled": bool(sm.stat("lp_tiled_builds")), "kernels": {}}
        for pfx, name in (("kx", "x-step (A'y): k_spmv_tiled + k_x_epilogue"), ("ky", "y-step (A x): k_spmv_tiled + k_y_epilogue")):
            tt, nl, by = dd[pfx + "_time_s"], max(dd[pfx + "_launches"], 1), dd[pfx + "_bytes"]
            spmv_roofline["kernels"][name] = {"avg_step_us": 1e6 * tt / nl, "algorithmic_bytes_per_step": by / nl,
                                              "achieved": by / tt / 1e9, "frac": by / tt / 1e9 / HBM_PEAK_GBS}
        # the same two fractions from the committed rocprofv3 summary of tools/spmv_bench.py (k_spmv_tiled + its epilogue per step)
        ns_t = rocprof_avg_ns(PROFILE_ROUND + "_spmv_hbm_kernel_stats.csv", "k_spmv_tiled")
        for name, epi in (("x-step (A'y): k_spmv_tiled + k_x_epilogue", "k_x_epilogue("), ("y-step (A x): k_spmv_tiled + k_y_epilogue", "k_y_epilogue(")):
            rec = spmv_roofline["kernels"][name]
            rec["achieved_inprocess"], rec["frac_inprocess"] = rec["achieved"], rec["frac"]
            ns_e = rocprof_avg_ns(PROFILE_ROUND + "_spmv_hbm_kernel_stats.csv", epi)
            st = profile_stamp()
            rec["frac_rocprof_committed"] = None
            if ns_t and ns_e and spmv_roofline["tiled"] and st and st["fresh"]:
                rec["avg_step_us_rocprof"] = (ns_t + ns_e) / 1e3
                rec["frac_rocprof_committed"] = rec["algorithmic_bytes_per_step"] / ((ns_t + ns_e) * 1e-9) / 1e9 / HBM_PEAK_GBS
        worst = min(spmv_roofline["kernels"].values(), key=lambda r: r["frac"])
        spmv_roofline["achieved"], spmv_roofline["frac"] = worst["achieved"], worst["frac"]
        spmv_roofline["frac_inprocess"] = min(r["frac_inprocess"] for r in spmv_roofline["kernels"].values())
        spmv_roofline["frac_source"] = "this run (per-launch hipEvents, the worse of the two steps); frac_rocprof_committed per step from profiles/%s_spmv_hbm_kernel_stats.csv when its stamp matches this build" % PROFILE_ROUND
        tpath = os.path.join(ROOT, "profiles", PROFILE_ROUND + "_traffic.json")
        if os.path.exists(tpath):
            t = json.load(open(tpath)).get("k_spmv_tiled")
            if t:
                spmv_roofline["traffic"] = t["bytes_per_launch"]
                spmv_roofline["traffic_source"] = "profiles/%s_traffic.json (k_spmv_tiled alone; rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)" % PROFILE_ROUND
        # The separator sweep on the SAME instance: 1e6 SHORT rows (k = 32), the kernel that shards over the GPUs under the
        # north star's NL-row split (DESIGN.md section 4 "the sweep on 1e6 short rows").  Near the optimum: few violated rows.
        if not args.no_sweep_roofline:
            sep2 = ktn.KatanaHipSeparator(sm); sep2.initialize()
            sm.reset()
            xs2 = np.clip(sinst.xhat + 0.05, sinst.l_var, sinst.u_var)
            sep2.precompute(xs2)
            for _ in range(2):
                sep2.sweep(1e-6); sm.reset(); sep2.precompute(xs2)
            skeys = ("sweep_eval_time_s", "sweep_eval_launches", "sweep_eval_bytes")
            b1 = {k: sm.stat(k) for k in skeys}
            for _ in range(8):
                nv2, _mv = sep2.sweep(1e-6); sm.reset(); sep2.precompute(xs2)
            dt, dn, db = (sm.stat(k) - b1[k] for k in skeys)
            ach = db / dt / 1e9
            sweep_short = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                           "kernel": ("k_sep_sweep_batch (batch-blocked: 2 048 rows per workgroup, x* staged through LDS in 64 KB blocks, "
                                      "kind-uniform entry-parallel evaluation)" if sm.stat("sweep_batched") else
                                      "k_sep_sweep (row kernel, several short rows per lane group)"),
                           "workload": "cfg4-shaped: n=%d, m_nl=%d exp/log rows, k=%d" % (sinst.n, sinst.m_nl, sinst.meta["k"]),
                           "launches": int(dn), "avg_launch_us": 1e6 * dt / dn, "algorithmic_bytes_per_launch": db / dn,
                           "violated_rows_at_the_point": int(nv2)}
            with_rocprof(sweep_short, PROFILE_ROUND + "_sweep_short_kernel_stats.csv", "k_sep_sweep_batch" if sm.stat("sweep_batched") else "k_sep_sweep<")
        del sm, sinst

    cpu = None
    if rank == 0 and not multi_path and not args.no_cpu_baseline:
        cpu = cpu_baseline(args)

    # N > 1: the per-phase split of the timed region, the latency of one all-reduce of an n-vector on this fabric, and the
    # SAME workload on rank 0's GPU alone (what a strong-scaling ratio has to be taken against)
    multi = None
    if multi_path:
        t = torch.zeros(inst.n + 1, dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        for _ in range(5):
            dist.all_reduce(t)
        barrier()
        ta = time.perf_counter()
        for _ in range(50):
            dist.all_reduce(t)
        barrier()
        ar_us = 1e6 * (time.perf_counter() - ta) / 50
        multi = {"lp_s": phases["lp_time_s"], "sweep_s": phases["sep_time_s"], "lp_setup_s": phases["lp_setup_time_s"],
                 "allreduce_calls": phases["allreduce_calls"], "allreduce_MB": phases["allreduce_bytes"] / 1e6,
                 "allreduce_n_vector_us": ar_us, "exchange": "none (row-sharded LP: every rank keeps its own cuts)"
                 if not args.replicated_lp else "all-gather of cut blocks", "lp_rows_rank0": model.lp_num_rows()}
        if not args.replicated_lp:
            # the same vector through the engine's own transport (RCCL on its stream / peer buffers / host callback), and its
            # self-check: the sum and the max every rank can compute for itself
            eng_us, eng_dev = model.allreduce_probe(inst.n + 1, 50)
            multi.update({"transport": model.transport, "engine_allreduce_n_vector_us": eng_us, "engine_allreduce_deviation": eng_dev})
        if rank == 0:
            one = ktn.NonlinearModel(ktn.KatanaSolver(log_level=0, device=local_rank))
            one.loadproblem(inst.n, inst.num_constr, inst.l_var, inst.u_var, inst.l_constr, inst.u_constr, inst.sense,
                            ktn.SeparableNLP(inst))
            run_steps(one, args.warmup)
            torch.cuda.synchronize()
            t1g = time.perf_counter()
            run_steps(one, args.steps)
            torch.cuda.synchronize()
            e1 = time.perf_counter() - t1g
            multi["same_workload_one_gpu"] = {"value": args.steps / e1, "ms_per_step": 1e3 * e1 / args.steps}
            del one
        barrier()

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
                       "parallelism": "1 GPU" if not multi_path else (
                           "nl-rows sharded x%d, replicated LP, all-gather of cuts" % world if args.replicated_lp else
                           "rows sharded x%d (linear rows and NL rows by blocks, cuts stay on their rank), x replicated, "
                           "one all-reduce of an n-vector per PDHG iteration (transport: %s)" % (world, model.transport))},
            "wall_to_ftol_s": wall_to_ftol, "ecp_iters_to_ftol": iters_to_ftol, "status": status, "objective": obj,
            "planted_objective": inst.opt_obj, "objective_relerr": abs(obj - inst.opt_obj) / max(1.0, abs(inst.opt_obj)),
            "pdhg_iters_per_step": pdhg_timed / args.steps, "solves_in_timed_region": len(solves),
            "first_solve_s": cold["first_solve_s"] if cold else None, "load_s": cold["load_s"] if cold else None,
            "load_new_handle_s": cold["load_new_handle_s"] if cold else None, "cold": cold,
            "job_rate_incl_load": (iters_to_ftol / (cold["load_s"] + wall_to_ftol)) if cold else None,
            "roofline": roofline, "sweep_roofline": sweep_roofline, "sweep_roofline_short_rows": sweep_short,
            "spmv_roofline": spmv_roofline, "stream_ceiling": stream_ceiling,
            "cpu_baseline": cpu,
            "multi_gpu": multi,
        }
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
