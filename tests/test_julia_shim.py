"""CPU tier: the reference-side Julia binding (katana.jl_amd/julia/KatanaHIP.jl) cannot be executed here (no julia in the
image), so what CAN be checked is checked statically: its opcode table, constants, struct mirrors and ccall symbol
names against include/katana_hip.h and the Python binding that the GPU tests exercise, and that the operator set its
Expr -> postfix walker accepts covers every expression of the reference's test models (tests/golden/kat_models.json)
-- the same set the Python tape compiler (katana.jl_amd/expr.py) emits for them."""
import os
import re

import numpy as np

import katana_jl_amd as ktn
from katana_jl_amd import _lib as L
from kat_util import load_kats

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JL = open(os.path.join(ROOT, "katana.jl_amd", "julia", "KatanaHIP.jl")).read()
HDR = open(os.path.join(ROOT, "include", "katana_hip.h")).read()


def header_defines(prefix):
    return {m.group(1): int(m.group(2)) for m in re.finditer(r"#define\s+(%s\w+)\s+(-?\d+)" % prefix, HDR)}


def julia_struct_fields(name):
    body = re.search(r"^struct %s\n(.*?)^end" % name, JL, re.S | re.M).group(1)
    return [(m.group(1), m.group(2)) for m in re.finditer(r"(\w+)::([\w{}]+)", body)]


def test_opcode_constants_equal_the_header():
    ops = header_defines("KTN_OP_")
    assert len(ops) == 13
    for name, val in ops.items():
        m = re.search(r"const %s\s*=\s*Int32\((\d+)\)" % name, JL)
        assert m and int(m.group(1)) == val, name
        assert getattr(L, name[4:]) == val                     # and the Python binding agrees
    for name, val in header_defines("KTN_ROW_").items():
        assert getattr(L, name[4:]) == val
    assert re.search(r"const KTN_ROW_SEP, KTN_ROW_TAPE, KTN_ROW_HOST = UInt8\(0\), UInt8\(1\), UInt8\(2\)", JL)
    assert re.search(r"const STATUS = \[:None, :Optimal, :Unbounded, :Infeasible, :UserLimit, :Error\]", JL)
    assert [ktn.solver.STATUS_SYMBOLS[i] for i in range(6)] == ["None", "Optimal", "Unbounded", "Infeasible", "UserLimit", "Error"]


JL2C = {"Cdouble": "c_double", "Int32": "c_int", "Int64": "c_long", "Ptr{Int64}": "LP_c_long", "Ptr{Int32}": "LP_c_int",
        "Ptr{UInt8}": "LP_c_ubyte", "Ptr{Cdouble}": "LP_c_double", "Ptr{Void}": "c_void_p"}


def test_struct_mirrors_have_the_fields_of_the_python_binding_in_order():
    for jl_name, cls in (("KtnParams", L.KtnParams), ("KtnNlpDesc", L.KtnNlpDesc)):
        jf = julia_struct_fields(jl_name)
        pf = [(n, t.__name__) for n, t in cls._fields_]
        assert [n for n, _ in jf] == [n for n, _ in pf], jl_name
        for (n, jt), (_, pt) in zip(jf, pf):
            assert JL2C[jt] == pt, (jl_name, n, jt, pt)


def test_every_ccall_names_a_symbol_the_library_binding_declares():
    syms = set(re.findall(r"ccall\(\(:(\w+),\s+LIB\)", JL))
    assert syms and syms <= set(L.PROTOTYPES), syms - set(L.PROTOTYPES)
    # the plugin surface of src/model.jl:63-65,81-86,219,326-343 is bound
    for s in ("ktn_create", "ktn_destroy", "ktn_loadproblem", "ktn_optimize", "ktn_get_status", "ktn_get_objval",
              "ktn_get_solution", "ktn_get_solvetime", "ktn_numiters", "ktn_numcuts", "ktn_last_error",
              "ktn_sizeof_params", "ktn_sizeof_nlp_desc"):
        assert s in syms, s


def sexpr_ops(s, acc):
    if isinstance(s, list):
        acc.add(s[0])
        for a in s[1:]:
            sexpr_ops(a, acc)
    return acc


def test_walker_operator_set_covers_the_reference_models_like_the_python_compiler():
    binary = dict(re.findall(r":(\S+) => (KTN_OP_\w+)", re.search(r"const BINARY_OPS = Dict.*?\)", JL, re.S).group(0)))
    unary = dict(re.findall(r":(\w+) => (KTN_OP_\w+)", re.search(r"const UNARY_OPS\s*= Dict.*?\)\n", JL, re.S).group(0)))
    assert binary == {"+": "KTN_OP_ADD", "-": "KTN_OP_SUB", "*": "KTN_OP_MUL", "/": "KTN_OP_DIV"}
    assert unary == {"exp": "KTN_OP_EXP", "log": "KTN_OP_LOG", "sqrt": "KTN_OP_SQRT", "sin": "KTN_OP_SIN", "cos": "KTN_OP_COS"}
    # ... and the Python compiler uses the same tables (katana.jl_amd/expr.py)
    from katana_jl_amd import expr as E
    assert {k: "KTN_OP_" + {v: n for n, v in vars(L).items() if n.startswith("OP_")}[v][3:] for k, v in E._BINARY.items()} == binary
    assert {k: v for k, v in unary.items()} == {k: "KTN_OP_" + {v: n for n, v in vars(L).items() if n.startswith("OP_")}[v][3:]
                                                for k, v in E._UNARY.items() if k != "neg"}
    # operators of every expression of the reference's test models
    used = set()
    for k in load_kats():
        sexpr_ops(k["objective"], used)
        for c in k["constraints"]:
            sexpr_ops(c["expr"], used)
    walker = set(binary) | set(unary) | {"^", "neg", "var"}          # :ref -> VAR, unary minus -> NEG, ^ const -> POWC
    assert used <= walker, used - walker
    # the opcodes the Python compiler emits over the same fixture are exactly those the walker can emit
    emitted = set()
    for k in load_kats():
        for s in [k["objective"]] + [c["expr"] for c in k["constraints"]]:
            emitted |= set(int(o) for o in ktn.from_sexpr(s).tape()[0])
    can_emit = {L.OP_CONST, L.OP_VAR, L.OP_NEG, L.OP_POWC} | set(E._BINARY.values()) | {v for k, v in E._UNARY.items()}
    assert emitted <= can_emit
    # every branch of emit_tape! is reachable from the fixture except sin/cos/div, which the reference's models never use
    names = {v: n for n, v in vars(L).items() if n.startswith("OP_")}
    assert {names[o] for o in can_emit - emitted} <= {"OP_SIN", "OP_COS", "OP_DIV", "OP_CONST"}


def test_coo_to_csr_conversion_matches_initialize():
    """build_ktn_nlp_desc's COO -> CSR loop restated: row by row, entries of a row in COO order (src/separators.jl:92-104);
    the Python callback binding (nlp.CallbackNLP) must produce the same permutation."""
    rng = np.random.default_rng(0)
    m, N = 7, 40
    rows, cols = rng.integers(0, m, N), rng.integers(0, 9, N)
    # the Julia loop (1-based there)
    counts = np.bincount(rows, minlength=m)
    rowptr = np.concatenate([[0], np.cumsum(counts)])
    fill = rowptr[:-1].copy()
    col, perm = np.zeros(N, dtype=int), np.zeros(N, dtype=int)
    for ind in range(N):
        i = rows[ind]
        col[fill[i]] = cols[ind]; perm[fill[i]] = ind
        fill[i] += 1
    # reference: sp_cols[i] in push order
    sp_cols = [[] for _ in range(m)]
    for ind in range(N):
        sp_cols[rows[ind]].append(cols[ind])
    assert [list(col[rowptr[i]:rowptr[i + 1]]) for i in range(m)] == sp_cols
    order = np.argsort(rows, kind="stable")                       # what nlp.CallbackNLP uses
    assert np.array_equal(order, perm)


def _code_only(src):
    """Julia source without comments, string literals and the inside of [...] (where `end` is an index, not a block end)"""
    out = []
    for line in src.splitlines():
        line = re.sub(r'"(?:[^"\\]|\\.)*"', '""', line)
        line = line.split("#", 1)[0]
        prev = None
        while prev != line:
            prev = line
            line = re.sub(r"\[[^\[\]]*\]", "", line)
        out.append(line)
    return "\n".join(out)


def test_blocks_are_balanced():
    """no julia here to parse the file: at least every block opener has its `end`"""
    code = _code_only(JL)
    openers = len(re.findall(r"(?m)^\s*(?:mutable struct|struct|module|function|if|for|while|try|let|begin)\b", code))
    openers += len(re.findall(r"\bdo\b", code))
    ends = len(re.findall(r"\bend\b", code))
    assert openers == ends, (openers, ends)
    assert code.count("(") == code.count(")")


def test_every_plugin_method_of_the_reference_has_a_method_here():
    """src/model.jl:63-65,81-86,219,326-343 and src/solver.jl:46: the MathProgBase methods (and numiters / numcuts) the reference
    defines on its model, and the two model factories on KatanaSolver"""
    for name in ("loadproblem!", "optimize!", "setwarmstart!", "status", "getobjval", "getsolution", "getsolvetime"):
        assert re.search(r"MathProgBase\.%s\(m::KatanaHipModel" % re.escape(name), JL), name
    for name in ("numiters", "numcuts"):
        assert re.search(r"(?m)^%s\(m::KatanaHipModel\)" % name, JL), name
    assert re.search(r"(?m)^MathProgBase\.NonlinearModel\(s::KatanaSolver\) = KatanaHipModel\(s\)", JL)
    assert re.search(r"(?m)^MathProgBase\.LinearQuadraticModel\(s::KatanaSolver\) = MathProgBase\.NonlinearToLPQPBridge\(MathProgBase\.NonlinearModel\(s\)\)", JL)
    assert "import ..KatanaSolver, ..AbstractKatanaSeparator, ..EpigraphNLPEvaluator" in JL
    # the loadproblem! signature is the reference's (src/model.jl:81-86)
    sig = re.search(r"function MathProgBase\.loadproblem!\(m::KatanaHipModel,(.*?)\)\s+#", JL, re.S).group(1)
    assert [t.strip() for t in re.findall(r"::([\w.{}]+)", sig)] == ["Int", "Int", "Vector{Float64}", "Vector{Float64}", "Vector{Float64}",
                                                                       "Vector{Float64}", "Symbol", "MathProgBase.AbstractNLPEvaluator"]


def test_separator_plugin_api_is_implemented_over_the_sep_entry_points():
    """src/separators.jl:8,23-53: a subtype of AbstractKatanaSeparator with the four API methods, each over its ktn_sep_* call"""
    assert re.search(r"mutable struct KatanaHipSeparator <: AbstractKatanaSeparator", JL)
    assert "import ..initialize!, ..precompute!, ..gencut, ..isconstrsat" in JL
    body = JL[JL.index("mutable struct KatanaHipSeparator"):JL.index("# ---- src/util.jl")]
    for method, sym in (("initialize!", "ktn_sep_jac_nnz"), ("precompute!", "ktn_sep_precompute"), ("isconstrsat", "ktn_sep_isconstrsat"),
                        ("gencut", "ktn_sep_gencut")):
        m = re.search(r"(?ms)^(?:function )?%s\(sep::KatanaHipSeparator.*?(?=^(?:function |isconstrsat\(|# ----)|\Z)" % re.escape(method), body)
        assert m and sym in m.group(0), (method, sym)
    # initialize! has the reference's signature (src/separators.jl:81-85) and unwraps the epigraph evaluator the reference passes
    assert re.search(r"initialize!\(sep::KatanaHipSeparator, linear_model::JuMP\.Model, num_var::Int, num_constr::Int,\s+oracle::MathProgBase\.AbstractNLPEvaluator\)", JL)
    assert "isa(oracle, EpigraphNLPEvaluator)" in JL and "oracle.nlpeval" in JL
    # gencut returns what linear_oa_cut returns: AffExpr over JuMP.Variable(linear_model, col) (src/algorithms.jl:6-17), 1-based columns
    assert "JuMP.Variable(sep.linear_model, Int(sep.cols[k]) + 1)" in body and "JuMP.AffExpr(v, sep.coefs[1:nnz[]], b[])" in body
    assert "i - 1" in body                                   # 1-based row of the plugin API -> 0-based row of the ABI


def test_host_callback_path_matches_the_header_typedefs():
    """no :ExprGraph -> KTN_ROW_HOST rows with cfunction trampolines of the two callback types of include/katana_hip.h"""
    for cb in ("ktn_eval_rows_cb", "ktn_eval_obj_cb"):
        m = re.search(r"typedef int \(\*%s\)\(void\* user, const double\* x, double\* \w+, double\* \w+\);" % cb, HDR)
        assert m, cb
    assert len(re.findall(r"cfunction\(eval_(?:rows|obj)_cb, Cint, \(Ptr\{Void\}, Ptr\{Cdouble\}, Ptr\{Cdouble\}, Ptr\{Cdouble\}\)\)", JL)) == 2
    assert ":ExprGraph in MathProgBase.features_available(d)" in JL
    host = JL[JL.index("function build_host_nlp_desc"):JL.index("# expressions when the evaluator offers them")]
    assert "fill(KTN_ROW_HOST, num_constr)" in host and "Int32(KTN_ROW_HOST)" in host and "pointer_from_objref(host)" in host
    assert "rows_c, obj_c, pointer_from_objref(host))" in host          # eval_rows, eval_obj, eval_user: the last three fields of KtnNlpDesc
    assert [f for f, _ in julia_struct_fields("KtnNlpDesc")][-3:] == ["eval_rows", "eval_obj", "eval_user"]
    # the callbacks use exactly the evaluator calls the reference makes (src/separators.jl:112-113, src/nlpeval.jl:35-39) and never throw
    cbs = JL[JL.index("function eval_rows_cb"):JL.index("function build_host_nlp_desc")]
    for call in ("MathProgBase.eval_jac_g(h.d, h.jcoo, x)", "MathProgBase.eval_g(h.d, g, x)", "MathProgBase.eval_f(h.d, x)", "MathProgBase.eval_grad_f(h.d,"):
        assert call in cbs, call
    assert cbs.count("catch") == 2 and cbs.count("return Cint(1)") == 2
    # KatanaHipModel keeps the HostEval object alive (its address is the engine's eval_user)
    assert "m.host = host" in JL and re.search(r"host::Union\{HostEval,Void\}", JL)
