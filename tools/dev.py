"""Development drivers for the engine, one file, one subcommand each (none of this is product code).

    python tools/dev.py cfg <config> [opt=val ...]                    one solve of a BASELINE config: time split, accuracy
    python tools/dev.py steps <config|shape> <seed> [opt=val ...]      per-ECP-step LP counters (KTN_DEBUG_LP=1: per-check trace)
    python tools/dev.py seeds <config|shape> <s0> <s1> [opt=val ...]   solve-time spread over seeds, warm buffers (as bench.py)
    python tools/dev.py knobs <config> <s0> <s1> "<ENV=.. opt=..>" ... `seeds` per environment / option variant, fresh process each
    python tools/dev.py setup [config]                                 per-solve LP setup breakdown (CSC mirror, scaling, power)
    python tools/dev.py shapes                                         2 families x 2 objectives x 3 sizes x 4 seeds robustness matrix
    python tools/dev.py kernel_us [n ...]                              mean launch duration of the LP step kernels against LP size
    python tools/dev.py reload                                         512 x cfg5 batches reloaded on one handle vs fresh handles

    python tools/dev.py offfamily <wall_s> <s0> <s1> <shape> ... [opt=val ...]  off-family battery: each solve stepwise under a wall budget

A <shape> is family:objective:n:m_nl:k[:bound_frac[:pivot_boost]], e.g. explog:quad:100000:10000:32 or explog:linear:1000:100:32:0.5:0
(instances.make_instance; `cfg3:::::0.5` = a BASELINE config with another bound_frac); a <config> is a key of
instances.CONFIGS.  opt=val are KatanaSolver keywords (ktn_params fields).
"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _opts(args):
    kw = {}
    for a in args:
        k, v = a.split("=")
        kw[k] = float(v) if ("." in v or "e" in v) else int(v)
    return kw


def _make(ktn, spec, seed):
    if ":" in spec:
        f = spec.split(":")
        fam, obj, n, m_nl, k = f[:5]
        if fam in ktn.instances.CONFIGS:                 # cfg3::::: 0.5 -> a BASELINE config with another bound_frac
            kw = dict(ktn.instances.CONFIGS[fam])
        else:
            kw = dict(n=int(n), m_nl=int(m_nl), k=int(k), family=fam, objective=obj)
        if len(f) > 5 and f[5] != "":
            kw["bound_frac"] = float(f[5])
        if len(f) > 6 and f[6] != "":
            kw["pivot_boost"] = bool(int(f[6]))
        return ktn.instances.make_instance(seed=seed, **kw)
    return ktn.instances.make_config(spec, seed=seed)


def _load(ktn, inst, **kw):
    m = ktn.NonlinearModel(ktn.KatanaSolver(**kw))
    m.loadproblem(inst.n, inst.num_constr, inst.l_var, inst.u_var, inst.l_constr, inst.u_constr, inst.sense, ktn.SeparableNLP(inst))
    return m


def _relerr(m, inst):
    return abs(m.getobjval() - inst.opt_obj) / max(1.0, abs(inst.opt_obj))


def cmd_cfg(ktn, args):
    import numpy as np
    name, kw = args[0], _opts(args[1:])
    t = time.time(); inst = _make(ktn, name, 0); tg = time.time() - t
    t = time.time(); m = _load(ktn, inst, log_level=1, **kw); tl = time.time() - t
    t = time.time(); st = m.optimize(); ts = time.time() - t
    x = m.getsolution()
    print("%s: gen %.1fs load %.2fs solve %.3fs %s iters=%d cuts=%d lp_rows=%d obj=%.9f opt=%.9f relerr=%.2e pdhg=%d lp=%.3fs sep=%.3fs xerr=%.1e" % (
        name, tg, tl, ts, st, m.numiters(), m.numcuts(), m.lp_num_rows(), m.getobjval(), inst.opt_obj, _relerr(m, inst),
        m.stat("pdhg_iters"), m.stat("lp_time_s"), m.stat("sep_time_s"), np.max(np.abs(x[:inst.n] - inst.xhat))), flush=True)
    print("lp setup %.4fs of lp %.4fs over %d solves" % (m.stat("lp_setup_time_s"), m.stat("lp_time_s"), m.stat("lp_solves")))


STEP_KEYS = ("pdhg_iters", "lp_restarts", "lp_consolidations", "lp_eta_backoffs", "lp_divergence_backoffs", "lp_stagnation_exits",
             "lp_stalled_row_exits", "lp_time_s")


def cmd_steps(ktn, args):
    inst = _make(ktn, args[0], int(args[1]))
    m = _load(ktn, inst, log_level=0, **_opts(args[2:]))
    m.optimize_begin()
    prev = {k: m.stat(k) for k in STEP_KEYS}
    done, t0 = False, time.time()
    max_steps = int(os.environ.get("DEV_MAX_STEPS", "1000000"))
    while not done and m.numiters() < max_steps:
        t = time.time(); done = m.ecp_step(); dt = time.time() - t
        cur = {k: m.stat(k) for k in STEP_KEYS}
        print("step %3d %.3fs rows %d cuts %d obj %.9g nviol %d maxviol %.3e " % (m.numiters(), dt, m.lp_num_rows(), m.numcuts(), m.getobjval(), m.stat("last_nviol"), m.stat("last_maxviol")) +
              " ".join("%s=%g" % (k.replace("lp_", ""), cur[k] - prev[k]) for k in STEP_KEYS), flush=True)
        prev = cur
    print(m.optimize_end(), "wall %.3fs" % (time.time() - t0), "obj", m.getobjval(), "planted", inst.opt_obj, "relerr %.2e" % _relerr(m, inst))


def cmd_seeds(ktn, args):
    import numpy as np
    name, n0, n1, kw = args[0], int(args[1]), int(args[2]), _opts(args[3:])
    ws, errs, its, tot_p, refine = [], [], [], 0, 0
    for seed in range(n0, n1):
        inst = _make(ktn, name, seed)
        m = _load(ktn, inst, log_level=0, **kw)
        m.optimize(); m.reset()              # warm buffers: the timed solve allocates nothing (as in bench.py)
        p0 = m.stat("pdhg_iters")
        t = time.time(); st = m.optimize(); w = time.time() - t
        assert st == "Optimal", (seed, st)
        ws.append(w); tot_p += m.stat("pdhg_iters") - p0; errs.append(_relerr(m, inst)); its.append(m.numiters())
        refine += int(m.stat("cert_refinements") > 1)          # (two solves per seed: > 1 = the timed one refined too)
        print("  seed %d: %s %.3fs iters %d relerr %.1e" % (seed, st, w, m.numiters(), errs[-1]), flush=True)
    print("%s seeds %d-%d %s: mean %.3fs median %.3fs max %.3fs total pdhg %d refined %d ecp iters %s max relerr %.1e | %s" % (
        name, n0, n1 - 1, kw, np.mean(ws), np.median(ws), np.max(ws), tot_p, refine, its, max(errs), " ".join("%.3f" % w for w in ws)))


def cmd_offfamily(ktn, args):
    """Solves off the non-degenerate-vertex family (the degeneracy dial of instances.make_instance), each stepwise so that a
    solve that crawls is cut off at the wall budget and reported instead of eating the GPU budget."""
    import numpy as np
    wall, n0, n1 = float(args[0]), int(args[1]), int(args[2])
    shapes = [a for a in args[3:] if ":" in a]
    kw = _opts([a for a in args[3:] if ":" not in a])
    for spec in shapes:
        for seed in range(n0, n1):
            inst = _make(ktn, spec, seed)
            m = _load(ktn, inst, log_level=0, **kw)
            t0 = time.time()
            m.optimize_begin()
            done = False
            while not done and time.time() - t0 < wall:
                done = m.ecp_step()
            st = m.optimize_end() if done else "WALL"
            w = time.time() - t0
            x = m.getsolution()[:inst.n]
            err = m.getobjval() - inst.opt_obj
            tol = max(1e-6, 1e-6 * max(abs(m.getobjval()), abs(inst.opt_obj)))
            print("%-40s seed %d: %-9s wall %7.2fs ecp %5d polish %3d pdhg %9d rows %7d obj %.9g planted %.9g err %+.2e (%s) dense %d mid %d piv %d cold %d fb %d lp_s %.2f" % (
                spec, seed, st, w, m.numiters(), m.stat("polish_iters"), m.stat("pdhg_iters"), m.lp_num_rows(), m.getobjval(), inst.opt_obj,
                err, "ok" if abs(err) <= tol else "MISS", m.stat("dense_lp_solves"), m.stat("mid_lp_solves"), m.stat("mid_lp_pivots"),
                m.stat("mid_lp_cold_starts"), m.stat("mid_lp_fallbacks"), m.stat("lp_time_s")), flush=True)


def cmd_knobs(ktn, args):
    name, n0, n1 = args[:3]
    for var in args[3:] or [""]:
        env, opts = dict(os.environ), []
        for tok in var.split():
            k, v = tok.split("=")
            if k.startswith("KTN_"):
                env[k] = v
            else:
                opts.append(tok)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "seeds", name, n0, n1] + opts, env=env, capture_output=True, text=True)
        print("[%s] %s" % (var, (r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1]), flush=True)


def cmd_setup(ktn, args):
    inst = _make(ktn, args[0] if args else "cfg3", 0)
    for prof in (0, 1):
        m = _load(ktn, inst, log_level=0, profile=prof)
        t = time.time(); st = m.optimize(); ts = time.time() - t
        print("profile=%d %s %.3fs iters=%d pdhg=%d lp=%.4f setup=%.4f csc=%.4f scaling=%.4f power=%.4f sep=%.4f solves=%d" % (
            prof, st, ts, m.numiters(), m.stat("pdhg_iters"), m.stat("lp_time_s"), m.stat("lp_setup_time_s"), m.stat("lp_csc_time_s"),
            m.stat("lp_scaling_time_s"), m.stat("lp_power_time_s"), m.stat("sep_time_s"), m.stat("lp_solves")))
    m = _load(ktn, inst, log_level=0, profile=1)
    m.optimize_begin()
    keys = ("lp_csc_time_s", "lp_scaling_time_s", "lp_power_time_s", "lp_setup_time_s", "lp_time_s", "sep_time_s", "pdhg_iters")
    prev = {k: m.stat(k) for k in keys}
    done = False
    while not done:
        done = m.ecp_step()
        cur = {k: m.stat(k) for k in keys}
        print("step %2d " % m.numiters() + " ".join(("%s=%.3fms" % (k.replace("_time_s", ""), 1e3 * (cur[k] - prev[k]))) if k != "pdhg_iters"
                                                    else "pdhg=%d" % (cur[k] - prev[k]) for k in keys))
        prev = cur
    m.optimize_end()


def cmd_shapes(ktn, args):
    """(The smooth-face regime, vertex=False, needs hundreds to thousands of ECP iterations by nature of Kelley's method.)"""
    kw = _opts(args)
    worst = 0.0
    for fam in ("explog", "quad"):
        for obj in ("linear", "quad"):
            for (n, m_nl, k) in ((300, 30, 8), (3000, 300, 16), (10000, 1000, 32)):
                for seed in range(4):
                    inst = ktn.instances.make_instance(n=n, m_nl=m_nl, k=k, family=fam, seed=seed, objective=obj)
                    m = _load(ktn, inst, log_level=0, lp_max_iter=400000, **kw)
                    t = time.time(); st = m.optimize(); w = time.time() - t
                    worst = max(worst, w)
                    print("%-6s obj=%-6s n=%-5d m_nl=%-4d seed=%d: %-9s iters=%-4d wall=%6.2fs pdhg=%-8d relerr=%.1e" % (
                        fam, obj, n, m_nl, seed, st, m.numiters(), w, m.stat("pdhg_iters"), _relerr(m, inst)), flush=True)
    print("worst wall %.2fs" % worst)


def cmd_kernel_us(ktn, args):
    for n in [int(a) for a in args] or [25000, 50000, 100000, 200000]:
        inst = ktn.instances.make_instance(n=n, m_nl=n // 10, k=32, family="explog", seed=0)
        m = _load(ktn, inst, log_level=0, profile=1)
        st = m.optimize()
        kx, ky = m.stat("kx_time_s") / max(m.stat("kx_launches"), 1), m.stat("ky_time_s") / max(m.stat("ky_launches"), 1)
        print("n=%7d rows=%7d nnz=%8d %s  k_pdhg_x %.2f us (%.1f MB)  k_pdhg_y %.2f us (%.1f MB)" % (
            n, m.lp_num_rows(), int(m._lib.ktn_lp_nnz(m._h)), st, 1e6 * kx, m.stat("kx_bytes") / max(m.stat("kx_launches"), 1) / 1e6,
            1e6 * ky, m.stat("ky_bytes") / max(m.stat("ky_launches"), 1) / 1e6), flush=True)


def cmd_reload(ktn, args):
    from katana_jl_amd.batch import FusedBatch
    from katana_jl_amd.instances import fuse_instances
    from katana_jl_amd.nlp import SeparableNLP
    insts_a = [ktn.instances.make_config("cfg5_one", seed=s) for s in range(512)]
    insts_b = [ktn.instances.make_config("cfg5_one", seed=1000 + s) for s in range(512)]
    fb = FusedBatch(ktn.KatanaSolver(log_level=0), insts_a)
    fb.solve()
    for _ in range(3):
        for insts in (insts_b, insts_a):
            t0 = time.perf_counter()
            big, offs = fuse_instances(insts); t1 = time.perf_counter()
            d = SeparableNLP(big); t2 = time.perf_counter()
            fb.m.loadproblem(big.n, big.num_constr, big.l_var, big.u_var, big.l_constr, big.u_constr, big.sense, d); t3 = time.perf_counter()
            fb.m.set_blocks(offs); st = fb.m.optimize_blocks(); fb.m.getsolution(); t4 = time.perf_counter()
            print("fuse %.3f describe %.3f load %.3f solve %.3f total %.3f %s" % (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t4 - t0, st), flush=True)
        t0 = time.perf_counter(); FusedBatch(ktn.KatanaSolver(log_level=0), insts_a).solve()
        print("fresh handle total %.3f" % (time.perf_counter() - t0))


def main():
    if len(sys.argv) < 2 or ("cmd_" + sys.argv[1]) not in globals():
        print(__doc__)
        sys.exit(2)
    import katana_jl_amd as ktn
    globals()["cmd_" + sys.argv[1]](ktn, sys.argv[2:])


if __name__ == "__main__":
    main()
