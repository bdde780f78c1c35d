"""Generator of tests/golden/lp_battery_highs.json: HiGHS (the oracle's LP, oracle/lp.py: dual simplex, tolerances 1e-9) on the
cases of the LP-only battery (katana.jl_amd/instances.lp_battery_case) with fewer than 1 700 columns.  On the larger cases of
the battery HiGHS -- simplex and interior point alike -- does not finish within minutes in this container (random sparse rows:
fill-in; measured 120 - 210 s and more at 3 300 - 8 300 columns), so those are pinned by the planted primal-dual pair of
instances.make_lp alone, which HiGHS confirms to 1e-10 on every case it did solve.

    python tests/golden/make_lp_battery_fixture.py
"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import katana_jl_amd as ktn
from oracle.lp import LinearModel

out = []
for i in range(50):
    kw = ktn.instances.lp_battery_case(i)
    if kw["n"] >= 1700:
        continue
    inst = ktn.instances.make_lp(**kw)
    lm = LinearModel()
    lm.add_variables(inst.l_var, inst.u_var)
    c = np.zeros(inst.n); c[inst.obj_col] = inst.obj_p0
    lm.set_objective("Min", np.arange(inst.n), c, 0.0)
    lm.add_rows(inst.rowptr, inst.col, inst.p0, inst.l_constr, inst.u_constr, assume_unique=True)
    t = time.time(); st = lm.solve(); w = time.time() - t
    out.append(dict(case=i, n=kw["n"], m=kw["m"], status=st, objective=float(lm.getobjval()), planted=float(inst.opt_obj), seconds=round(w, 2)))
    print(out[-1], flush=True)
json.dump(dict(note="HiGHS dual simplex (SciPy 1.15.3, oracle/lp.py) on the small cases of the LP battery; generator make_lp_battery_fixture.py",
               cases=out), open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "lp_battery_highs.json"), "w"), indent=1)
