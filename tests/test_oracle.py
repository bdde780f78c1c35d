"""CPU tier: the oracle (CPU restatement of the reference) against the reference's own
known-answer tests, and its vectorised path against its literal path."""
import numpy as np
import pytest

import katana_jl_amd as ktn
from helpers import TRAJECTORY_SENSITIVE, oracle_solve_instance, oracle_solve_kat
from kat_util import isapprox, load_family_ext, load_kats

KATS = load_kats()
# the n-D sphere family is exercised at every second size on the CPU tier to keep it quick
CPU_KATS = [m for m in KATS if not (m["id"].startswith("501_") and int(m["id"].split("_n")[1]) % 2 == 0
                                    and int(m["id"].split("_n")[1]) > 4)]


def test_fixture_covers_the_reference_suite():
    ids = {m["id"] for m in KATS}
    assert len(KATS) == 82
    for must in ("basic_1", "001_01", "101_01", "105_01", "203_01", "205_01", "210_03", "501_01_n20", "501_02_n20"):
        assert must in ids


@pytest.mark.parametrize("m", CPU_KATS, ids=[m["id"] for m in CPU_KATS])
def test_oracle_passes_reference_kat(m):
    om = oracle_solve_kat(m)
    e = m["expect"]
    assert om.getstatus() == e["status"]
    obj, x = om.getobjval(), om.getsolution()
    if m["id"] in TRAJECTORY_SENSITIVE:
        assert isapprox(obj, e["obj"], 1e-6, 1e-6)                  # suite tolerance, test/runtests.jl:16-17
        if e["x"] is not None:
            assert np.max(np.abs(np.asarray(x[:len(e["x"])]) - e["x"])) <= 3e-3
    else:
        assert isapprox(obj, e["obj"], e["obj_atol"], e["obj_rtol"])
        if e["x"] is not None:
            for got, want in zip(x, e["x"]):
                assert isapprox(got, want, e["sol_atol"], e["sol_rtol"]), (list(x), e["x"])


def test_oracle_on_the_ball_family_beyond_the_reference_sizes():
    """tests/golden/kat_family_ext.json: the smallest member (n = 33, quadratic form) through the oracle -- 2 711 ECP iterations
    on HiGHS vertices, 20 s; the larger ones and the norm form (10 000 iterations without meeting the stop rule on exact
    vertices) are pinned by their closed form only."""
    ext = {m["id"]: m for m in load_family_ext()}
    assert sorted(ext) == sorted("501_0%d_n%d" % (f, n) for f in (1, 2) for n in (33, 64, 128))
    for m in ext.values():
        n = len(m["vars"])
        assert abs(m["expect"]["obj"] + np.sqrt(n)) < 1e-12 and np.allclose(m["expect"]["x"], 1 / np.sqrt(n), rtol=0, atol=1e-15)
    m = ext["501_01_n33"]
    om = oracle_solve_kat(m)
    assert om.getstatus() == "Optimal"
    assert isapprox(om.getobjval(), m["expect"]["obj"], 1e-6, 1e-6)
    assert np.max(np.abs(np.asarray(om.getsolution()[:33]) - m["expect"]["x"])) <= 1e-3


def test_fast_path_equals_literal_path():
    inst = ktn.instances.make_instance(n=300, m_nl=30, k=8, family="explog", seed=1)
    a = oracle_solve_instance(inst, fast=True)
    b = oracle_solve_instance(inst, fast=False)
    assert a.getstatus() == b.getstatus() == "Optimal"
    assert a.numiters() == b.numiters() and a.getnumcuts() == b.getnumcuts()
    assert abs(a.getobjval() - b.getobjval()) <= 1e-9 * max(1, abs(b.getobjval()))


@pytest.mark.parametrize("family", ["explog", "quad", "explog+t", "quad+t"])
def test_oracle_finds_planted_optimum(family):
    inst = ktn.instances.make_instance(n=500, m_nl=50, k=16, family=family, seed=2)
    om = oracle_solve_instance(inst)
    assert om.getstatus() == "Optimal"
    assert abs(om.getobjval() - inst.opt_obj) <= 1e-5 * max(1.0, abs(inst.opt_obj))


@pytest.mark.parametrize("seed,index,nv,ncons,status,obj,iters", [
    (2, 109, 4, 3, "Optimal", -5.129864387562447e-07, 75), (13, 142, 4, 2, "Optimal", 0.010341655863652294, 56),
    (55, 106, 3, 1, "Optimal", 1.4213188826617347, 33), (144, 51, 4, 4, "Optimal", 0.9446566311142093, 30),
    (2, 62, 5, 4, "Infeasible", None, 9), (2, 138, 6, 4, "Infeasible", None, 17)])
def test_fuzz_regression_models_are_pinned(seed, index, nv, ncons, status, obj, iters):
    """the six members of the random small-model stream (tests/fuzz_models.py) that the GPU suite replays: the generator gives the
    same models as when they were found, and the oracle the same status / objective / iteration count"""
    from fuzz_models import model_at
    from helpers import oracle_solve_kat
    m = model_at(seed, index)
    assert len(m["vars"]) == nv and len(m["constraints"]) == ncons and all(v["ub"] == float("inf") for v in m["vars"])
    om = oracle_solve_kat(m)
    assert om.getstatus() == status and om.numiters() == iters
    if obj is not None:
        assert abs(om.getobjval() - obj) <= 1e-9 * max(1.0, abs(obj))


def test_round_coefs_signed_max_and_constant_untouched():
    from oracle.katana import AffExpr, round_coefs
    cut = AffExpr([0, 1, 2], [-2e9, 1.0, -5.0], 7.0)
    round_coefs(cut, 1e9)
    assert cut.coeffs == [0.0, 1.0, -5.0] and cut.constant == 7.0       # src/model.jl:200-207
    cut = AffExpr([0, 1], [-3e9, -1.0], 1.0)                            # signed max = -1
    round_coefs(cut, 1e9)
    assert cut.coeffs == [0.0, -1.0]


def test_pdlp_mirror_solves_small_lp():
    import scipy.sparse as sp
    from oracle import pdlp_mirror as pm
    A = sp.csr_matrix(np.array([[4.0, 4.0], [1.0, -1.0]]))
    r = pm.solve_lp_halpern(A, np.array([-1.0, -1.0]), np.array([-2.0, -2.0]), np.array([2.0, 2.0]),
                            np.array([-np.inf, -0.5]), np.array([9.0, 0.5]), params=pm.PdlpParams(eps=1e-9))
    assert r["status"] == "Optimal" and abs(r["pobj"] + 2.25) < 1e-6


def test_committed_trace_fixture_is_what_the_oracle_produces():
    """tests/golden/kat_traces.json is regenerated in-process for two models and compared with the committed file"""
    import importlib.util, json, os
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("make_trace_fixture", os.path.join(here, "golden", "make_trace_fixture.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    with open(os.path.join(here, "golden", "kat_traces.json")) as f:
        committed = {t["id"]: t for t in json.load(f)}
    from kat_util import load_kats
    kats = {k["id"]: k for k in load_kats()}
    for kid in ("101_01", "105_01"):
        fresh = json.loads(json.dumps(mod.trace(kats[kid])))
        assert fresh["numiters"] == committed[kid]["numiters"] and fresh["status"] == "Optimal"
        assert len(fresh["iterations"]) == len(committed[kid]["iterations"])
        for a, b in zip(fresh["iterations"], committed[kid]["iterations"]):
            assert np.allclose(a["x"], b["x"], rtol=0, atol=1e-12) and a["nl_rows"] == b["nl_rows"]
            assert [c["row"] for c in a["cuts"]] == [c["row"] for c in b["cuts"]]
            for ca, cb in zip(a["cuts"], b["cuts"]):
                assert np.allclose(ca["coefs"], cb["coefs"], rtol=1e-12, atol=1e-12)
