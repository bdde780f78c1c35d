"""Generator of tests/golden/offfamily_oracle.json: the CPU oracle (oracle/katana.py on HiGHS vertices) on the instance
families OFF the non-degenerate-vertex family the engine was tuned on -- the degeneracy dial `bound_frac` of
katana.jl_amd/instances.make_instance (0.0 = SURVEY.md section 8d's smooth-face generator, the regime of the reference's
own claim README.md:5 and of test/misc.jl:4-57; 0.5 = half of the non-pivot variables pinned).

On these families Kelley's method on simplex vertices needs hundreds to thousands of iterations (src/model.jl:257-309 never
drops a cut), so the oracle is run ONCE here, each case in its own process under a time limit, and its (status, objective,
iterations, cuts, seconds) committed as data; a case the oracle does not finish within the limit is recorded as such and is
pinned by the planted optimum alone (xhat is a KKT point of a convex problem: its objective is exact).

    python tests/golden/make_offfamily_fixture.py [limit_s] [workers]

These are self-consistency vectors made by the restatement in this container, not outputs of the Julia reference.
"""
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

CASES = [dict(family=fam, n=n, m_nl=n // 10, k=k, bound_frac=bf, seed=seed)
         for (n, k) in ((200, 16), (1000, 32)) for fam in ("explog", "quad") for bf in (0.5, 0.0) for seed in (0, 1, 2)]


def _run(case, q):
    import katana_jl_amd as ktn
    from tests.helpers import oracle_solve_instance
    inst = ktn.instances.make_instance(**case)
    t = time.time()
    om = oracle_solve_instance(inst)
    q.put(dict(status=om.status, objective=float(om.getobjval()), iters=int(om.numiters()), numcuts=int(om.numcuts),
               seconds=round(time.time() - t, 1), planted=float(inst.opt_obj)))


def one(args):
    case, limit = args
    q = mp.Queue()
    p = mp.Process(target=_run, args=(case, q))
    p.start()
    p.join(limit)
    out = dict(case)
    if p.is_alive():
        p.terminate()
        p.join()
        out.update(oracle_timeout_s=limit)
    else:
        out.update(q.get())
    print(out, flush=True)
    return out


if __name__ == "__main__":
    limit = float(sys.argv[1]) if len(sys.argv) > 1 else 1800.0
    workers = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(workers) as ex:
        res = list(ex.map(one, [(c, limit) for c in CASES]))
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "offfamily_oracle.json"), "w") as f:
        json.dump(dict(note="oracle (oracle/katana.py, HiGHS dual simplex) on the off-family instances; generator make_offfamily_fixture.py",
                       limit_s=limit, cases=res), f, indent=1)
