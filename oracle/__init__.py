"""CPU oracle for the Katana ECP hot path -- TEST INFRASTRUCTURE ONLY.

This package is a plain CPU (numpy / pure-Python) restatement of the reference
algorithm in lanl-ansi/Katana.jl (src/model.jl, src/separators.jl,
src/algorithms.jl, src/nlpeval.jl).  It exists to *check* the HIP product path.

Rules (enforced by tests/test_no_oracle_in_product.py):
  * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
    import anything under oracle/;
  * nothing under katana.jl_amd/ imports, links, calls or executes it;
  * the product path fails loudly when the HIP library is missing.

Pinning status: the reference's arithmetic for g/J lives in JuMP's AD
evaluator and its LP solves live in GLPK -- neither is in /root/reference nor in
this image (REQUIRE:3-5, test/REQUIRE:3-6; no lockfile, so no pinned version
exists).  The restatement is therefore pinned END-TO-END by the reference's own
known-answer tests (test/basic.jl, lpqp.jl, 2d.jl, 3d.jl, misc.jl; SURVEY.md
section 4.1), committed as tests/golden/kat_models.json: status, objective to
1e-6 and solution to 1e-3, exactly the acceptance the reference's test-suite
applies (test/runtests.jl:16-20).  Intermediate quantities (per-iteration x*,
cut rows, iteration counts) are parity-UNPINNED: no reference test reads them.

LP substitute: GLPK (absent) -> HiGHS dual simplex as vendored in SciPy 1.15.3
(scipy.optimize._highspy), warm-started across ECP iterations like the
reference's live GLPK model.  Unbounded rays (GLPK getunboundedray, absent in
the SciPy binding) come from the recession-cone LP described in oracle/lp.py.
"""
