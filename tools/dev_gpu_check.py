"""Development aid: quick end-to-end checks on a GPU box (not a test)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import katana_jl_amd as ktn
from kat_util import load_kats, isapprox

def run_kats(ids=None):
    ok = 0; tot = 0
    for m in load_kats():
        if ids and m["id"] not in ids: continue
        tot += 1
        M = ktn.Model(solver=ktn.KatanaSolver(log_level=0))
        for v in m["vars"]: M.variable(v["lb"], v["ub"])
        M.objective(m["sense"], ktn.from_sexpr(m["objective"]), linear=m["objective_linear"])
        for c in m["constraints"]:
            M.constraint((ktn.from_sexpr(c["expr"]), c["lb"], c["ub"]), linear=c["linear"])
        t = time.time()
        try:
            st = M.solve()
            im = M.internal_model
            e = m["expect"]; obj = M.getobjectivevalue(); x = M.getvalue()
            good = st == e["status"] and isapprox(obj, e["obj"], 1e-6, 1e-6)
            xerr = max(abs(a - b) for a, b in zip(x, e["x"])) if e["x"] else 0.0
            ok += good
            print("%-12s %-9s %s obj=%.9f want=%.9f xerr=%.1e it=%d cuts=%d pdhg=%d %.2fs" % (
                m["id"], st, "ok " if good else "BAD", obj, e["obj"], xerr, im.numiters(), im.numcuts(),
                im.stat("pdhg_iters"), time.time() - t), flush=True)
        except Exception as ex:
            print("%-12s EXC %r" % (m["id"], ex), flush=True)
    print("KATs ok %d / %d" % (ok, tot))

def run_syn(n, m_nl, k, fam, seed=0, **kw):
    inst = ktn.instances.make_instance(n=n, m_nl=m_nl, k=k, family=fam, seed=seed)
    m = ktn.NonlinearModel(ktn.KatanaSolver(log_level=1, **kw))
    t = time.time()
    m.loadproblem(inst.n, inst.num_constr, inst.l_var, inst.u_var, inst.l_constr, inst.u_constr, inst.sense, ktn.SeparableNLP(inst))
    tl = time.time() - t; t = time.time()
    st = m.optimize()
    print("syn n=%d m_nl=%d k=%d %s: %s iters=%d cuts=%d obj=%.9f opt=%.9f relerr=%.2e load=%.2fs solve=%.3fs lp=%.3fs sep=%.3fs pdhg=%d restarts=%d" % (
        n, m_nl, k, fam, st, m.numiters(), m.numcuts(), m.getobjval(), inst.opt_obj,
        abs(m.getobjval() - inst.opt_obj) / max(1, abs(inst.opt_obj)), tl, time.time() - t, m.stat("lp_time_s"),
        m.stat("sep_time_s"), m.stat("pdhg_iters"), m.stat("lp_restarts")), flush=True)

if __name__ == "__main__":
    what = sys.argv[1:] or ["small"]
    if "small" in what:
        run_syn(400, 40, 16, "explog")
        run_syn(1000, 100, 16, "quad")
    if "kats" in what:
        run_kats()
    if "kats_few" in what:
        run_kats({"101_01", "001_01", "basic_3", "105_01", "203_01", "501_01_n5"})
    if "mid" in what:
        run_syn(10000, 1000, 32, "explog")
    if "cfg3" in what:
        run_syn(100000, 10000, 32, "explog")
    if "cfg2" in what:
        run_syn(10000, 1000, 64, "quad")
