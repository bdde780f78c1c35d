"""katana.jl_amd -- MI355X-native Extended-Cutting-Plane engine behind Katana's plugin surface.

Host-side mirror (Python, ctypes over the C ABI of include/katana_hip.h) of the reference's
MathProgBase interface for the ONE hot path of lanl-ansi/Katana.jl:

    KatanaSolver(...)                      src/solver.jl:6-43
    NonlinearModel(s) -> KatanaNonlinearModel   src/model.jl:9-65
    loadproblem / optimize / status / getobjval / getsolution / getsolvetime   src/model.jl:81-343
    KatanaHipSeparator: initialize / precompute / isconstrsat / gencut          src/separators.jl
    getKatanaModel / getKatanaCuts / getKatanaSols                              src/util.jl

All arithmetic runs in libkatana_hip.so (hand-written HIP for gfx950).  There is no CPU
fallback: importing works without a GPU (so the CPU test tier can check the ABI), creating
a model without one raises.
"""
from . import _lib
from .expr import Expr, var, const, exp, log, sqrt, sin, cos, from_sexpr
from .nlp import NLPDescription, SeparableNLP, ExprNLP, CallbackNLP
from .solver import (KatanaSolver, KatanaNonlinearModel, KatanaHipSeparator, NonlinearModel,
                     getKatanaModel, getKatanaCuts, getKatanaSols, STATUS_SYMBOLS)
from .jump_like import Model
from . import instances
from .batch import solve_batch, solve_batch_sharded

__all__ = ["KatanaSolver", "KatanaNonlinearModel", "KatanaHipSeparator", "NonlinearModel", "getKatanaModel",
           "getKatanaCuts", "getKatanaSols", "NLPDescription", "SeparableNLP", "ExprNLP", "CallbackNLP", "Model", "Expr", "var",
           "const", "exp", "log", "sqrt", "sin", "cos", "from_sexpr", "instances", "STATUS_SYMBOLS", "solve_batch", "solve_batch_sharded"]
