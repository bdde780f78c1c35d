"""LP-only battery: the engine's LP (ktn_lp_solve on a problem without nonlinear rows) against the planted optimum of
instances.make_lp -- sizes 1e3 ... 1e4, degenerate vertices and badly scaled rows included.  (The comparison with HiGHS is
tests/test_gpu_lp.py::test_lp_battery_against_highs; this script only needs the planted objective, so it imports no oracle.)

    python tools/lp_battery.py [count] [first_seed] [opt=val ...]
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import katana_jl_amd as ktn


battery_case = ktn.instances.lp_battery_case


def main():
    cnt = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    kw = {}
    for a in sys.argv[3:]:
        k, v = a.split("=")
        kw[k] = float(v) if ("." in v or "e" in v) else int(v)
    bad = 0
    for i in range(s0, s0 + cnt):
        c = battery_case(i)
        inst = ktn.instances.make_lp(**c)
        m = ktn.NonlinearModel(ktn.KatanaSolver(log_level=0, **kw))
        m.loadproblem(inst.n, inst.num_constr, inst.l_var, inst.u_var, inst.l_constr, inst.u_constr, inst.sense, ktn.SeparableNLP(inst))
        t = time.time(); st, it = m.lp_solve(row_tol=1e-8, gap_tol=1e-8); w = time.time() - t
        err = abs(m.getobjval() - inst.opt_obj) / max(1.0, abs(inst.opt_obj))
        ok = st == "Optimal" and err <= 1e-7
        bad += not ok
        print("case %2d n %5d m %5d nnz/row %2d deg %.1f scale %.0f free %.1f: %-9s iters %8d %.3fs relerr %.1e mid %d %s" % (
            i, c["n"], c["m"], c["nnz_row"], c.get("degenerate_frac", 0), c.get("bad_scale_decades", 0), c.get("free_frac", 0), st, it, w, err,
            m.stat("mid_lp_solves"), "" if ok else "MISS"), flush=True)
    print("misses:", bad)


if __name__ == "__main__":
    main()
