"""Development aid: throughput of cfg5 with P worker PROCESSES (each with T threads)."""
import json, os, sys, time
import multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def worker(args):
    lo, hi, threads = args
    import katana_jl_amd as ktn
    insts = [ktn.instances.make_config("cfg5_one", seed=s) for s in range(lo, hi)]
    t0 = time.perf_counter()
    res, wall = ktn.solve_batch(ktn.KatanaSolver(log_level=0), insts, threads=threads)
    return sum(r["status"] == "Optimal" for r in res), time.perf_counter() - t0

if __name__ == "__main__":
    nb, P, T = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    chunks = [(nb * p // P, nb * (p + 1) // P, T) for p in range(P)]
    ctx = mp.get_context("spawn")
    t0 = time.perf_counter()
    with ctx.Pool(P) as pool:
        out = pool.map(worker, chunks)
    wall = time.perf_counter() - t0
    print(json.dumps({"instances": nb, "processes": P, "threads": T, "optimal": sum(o[0] for o in out), "wall_incl_startup_s": wall,
                      "max_worker_solve_s": max(o[1] for o in out), "instances_per_s_solve": nb / max(o[1] for o in out)}))
