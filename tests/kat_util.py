"""Shared helpers for the known-answer tests (tests/golden/kat_models.json)."""
import json
import math
import os

HERE = os.path.dirname(os.path.abspath(__file__))
KAT_PATH = os.path.join(HERE, "golden", "kat_models.json")


def _dec(o):
    if o == "inf":
        return math.inf
    if o == "-inf":
        return -math.inf
    if isinstance(o, dict):
        return {k: _dec(v) for k, v in o.items()}
    if isinstance(o, list):
        return [_dec(v) for v in o]
    return o


def load_kats():
    with open(KAT_PATH) as f:
        return _dec(json.load(f))["models"]


def load_family_ext():
    """members of the test/misc.jl family at sizes the reference does not run (closed-form expectations; make_kat_fixture.py)"""
    with open(os.path.join(HERE, "golden", "kat_family_ext.json")) as f:
        return _dec(json.load(f))["models"]


def isapprox(a, b, atol, rtol):
    """Julia's isapprox(a, b; atol, rtol): |a-b| <= max(atol, rtol*max(|a|,|b|))."""
    return abs(a - b) <= max(atol, rtol * max(abs(a), abs(b)))


def check_expectation(model, status, obj, x):
    e = model["expect"]
    assert status == e["status"], (model["id"], status)
    assert isapprox(obj, e["obj"], e["obj_atol"], e["obj_rtol"]), (model["id"], obj, e["obj"])
    if e["x"] is not None:
        for got, want in zip(x, e["x"]):
            assert isapprox(got, want, e["sol_atol"], e["sol_rtol"]), (model["id"], list(x), e["x"])
