"""Generates tests/golden/kat_traces.json: per-iteration traces of the CPU oracle (oracle/katana.py) on four of the
reference's test models (golden vectors (2)).

For every ECP iteration: the LP point x*, the constraint values g at it, the violated nonlinear rows and the cut each
of them gets (dense coefficient vector after round_coefs, row bounds lo/hi) -- src/model.jl:265-283, src/separators.jl:
111-120, src/algorithms.jl:3-18.  These are SELF-CONSISTENCY vectors (restatement <-> kernels), produced by the oracle in
this container; they are not outputs of the Julia reference, which cannot run here.

Run from the repo root:  python tests/golden/make_trace_fixture.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from kat_util import load_kats                                            # noqa: E402
from oracle.evaluators import SexprNLPEvaluator                            # noqa: E402
from oracle.katana import (KatanaModelParams, KatanaNonlinearModel, linear_oa_cut, round_coefs)   # noqa: E402

IDS = ["101_01", "105_01", "203_01", "501_01_n5"]
MAX_ITERS = 12          # iterations recorded per model (the fixture stays small)


def trace(m):
    n = len(m["vars"])
    d = SexprNLPEvaluator(n, m["objective"], [c["expr"] for c in m["constraints"]],
                          [c["linear"] for c in m["constraints"]], m["objective_linear"])
    om = KatanaNonlinearModel(KatanaModelParams(), vis_data=True)
    om.loadproblem(n, len(m["constraints"]), [v["lb"] for v in m["vars"]], [v["ub"] for v in m["vars"]],
                   [c["lb"] for c in m["constraints"]], [c["ub"] for c in m["constraints"]], m["sense"], d)
    om.optimize()
    sep = om.params.separator
    its = []
    for x in om.lp_sols[:MAX_ITERS]:
        with np.errstate(all="ignore"):
            sep.precompute(np.asarray(x, dtype=float))
        cuts = []
        for i in om.nlconstr_ixs:
            lb, ub = om.l_constr[i], om.u_constr[i]
            if sep.isconstrsat(i, lb, ub, om.params.f_tol):
                continue
            cut = linear_oa_cut(sep, x, None, i)
            round_coefs(cut, om.params.cut_coef_rng)
            dense = np.zeros(om.num_var)
            for c, v in zip(cut.vars, cut.coeffs):
                dense[c] += v
            cuts.append({"row": int(i), "coefs": dense.tolist(), "lo": float(lb - cut.constant), "hi": float(ub - cut.constant)})
        its.append({"x": [float(v) for v in x], "g": [float(sep.g[i]) for i in om.nlconstr_ixs],
                    "nl_rows": [int(i) for i in om.nlconstr_ixs], "cuts": cuts})
    return {"id": m["id"], "status": om.status, "numiters": om.numiters(), "objective": om.getobjval(), "iterations": its}


if __name__ == "__main__":
    kats = {k["id"]: k for k in load_kats()}
    out = [trace(kats[i]) for i in IDS]
    path = os.path.join(HERE, "kat_traces.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0)
    print("wrote", path, os.path.getsize(path), "bytes;", [(t["id"], t["numiters"], len(t["iterations"])) for t in out])
