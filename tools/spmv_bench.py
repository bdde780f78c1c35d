"""LP SpMV microbenchmark on an HBM-resident cut matrix (SURVEY.md section 8d "LP, per PDHG iteration").

    python tools/spmv_bench.py [rows_nl] [iters]

Builds cfg4's LP after ONE un-capped sweep at a far-away point: n = 1e5 columns, 5e4 linear rows + `rows_nl` cut rows of 32
entries (default 1e6 -> 3.2e7 non-zeros, 384 MB CSR + 384 MB CSC mirror: far beyond the 256 MiB Infinity Cache), then runs
`iters` raw PDHG iterations (k_pdhg_x + k_pdhg_y, fixed step) with per-launch hipEvent timing and prints one JSON line with
algorithmic bytes / mean launch duration against the 8 TB/s HBM peak for both kernels."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import katana_jl_amd as ktn


def build(rows_nl, profile=1, device=-1, **kw):
    inst = ktn.instances.make_instance(n=100_000, m_nl=rows_nl, k=32, family="explog", seed=0)
    m = ktn.NonlinearModel(ktn.KatanaSolver(log_level=0, profile=profile, device=device, cut_cap_factor=0.0, purge_age=0, **kw))
    m.loadproblem(inst.n, inst.num_constr, inst.l_var, inst.u_var, inst.l_constr, inst.u_constr, inst.sense, ktn.SeparableNLP(inst))
    sep = ktn.KatanaHipSeparator(m); sep.initialize()
    x = np.clip(inst.xhat + 2.0, inst.l_var, inst.u_var)          # nearly every NL row violated -> one cut per row
    sep.precompute(x)
    nviol, _ = sep.sweep(1e-6)
    return inst, m, nviol


def main():
    rows_nl = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    tiled = int(sys.argv[3]) if len(sys.argv) > 3 else -1            # -1: engine default, 0: CSR kernels, 1: tiled kernels
    inst, m, nviol = build(rows_nl, **({} if tiled < 0 else {"lp_tiled_nnz": tiled}))
    M, nnz = m.lp_num_rows(), int(m._lib.ktn_lp_nnz(m._h))
    x0, y0 = np.zeros(m.num_var), np.zeros(M)
    m.lp_pdhg_raw(x0, y0, 1e-3, 1.0, 4)                            # warm-up (builds the mirror and the scaling)
    keys = [p + s for p in ("kx", "ky") for s in ("_time_s", "_launches", "_bytes")]
    b0 = {k: m.stat(k) for k in keys}
    t0 = time.perf_counter()
    m.lp_pdhg_raw(x0, y0, 1e-3, 1.0, iters)
    wall = time.perf_counter() - t0
    d = {k: m.stat(k) - b0[k] for k in keys}
    out = {"workload": "cfg4 LP after one un-capped sweep: n=%d, rows=%d (%d cuts), nnz=%d" % (inst.n, M, nviol, nnz),
           "csr_plus_csc_MB": 2 * nnz * 12 / 1e6, "iters": iters, "wall_per_iter_us": 1e6 * wall / iters,
           "tiled": bool(m.stat("lp_tiled_builds")), "tiled_build_ms": 1e3 * m.stat("lp_tiled_build_time_s") / max(m.stat("lp_tiled_builds"), 1)}
    for p, name in (("kx", "k_pdhg_x (A'y)"), ("ky", "k_pdhg_y (A x)")):
        t, nl, by = d[p + "_time_s"], max(d[p + "_launches"], 1), d[p + "_bytes"]
        out[name] = {"avg_launch_us": 1e6 * t / nl, "algorithmic_MB": by / nl / 1e6, "GBps": by / t / 1e9, "frac_of_8TBps": by / t / 8e12}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
