"""A minimal JuMP-shaped front end so that models read like the reference's tests
(test/2d.jl etc.): Model(solver=...), variable, objective, constraint, solve, getvalue.
It only assembles the arguments of loadproblem! (src/model.jl:81-86): nothing numerical
happens here."""
import math

import numpy as np

from .expr import Expr, var
from .nlp import ExprNLP
from .solver import NonlinearModel


class Model:
    def __init__(self, solver=None):
        self.solver = solver
        self.lb, self.ub = [], []
        self.sense, self.obj = "Min", Expr.wrap(0.0)
        self.obj_linear = None
        self.cons = []          # (expr g, lb, ub, declared_linear)
        self.internal_model = None
        self._status = "None"

    def variable(self, lb=-math.inf, ub=math.inf, start=None):
        self.lb.append(float(lb)); self.ub.append(float(ub))
        return var(len(self.lb) - 1)

    def variables(self, n, lb=-math.inf, ub=math.inf):
        return [self.variable(lb, ub) for _ in range(n)]

    def objective(self, sense, e, linear=None):
        self.sense, self.obj, self.obj_linear = sense, Expr.wrap(e), linear

    def constraint(self, rel, linear=None):
        """`rel` is `lhs <= rhs` / `lhs >= rhs` built from Expr operators, or (g, lb, ub)."""
        if len(rel) == 3 and rel[0] in ("<=", ">="):
            op, lhs, rhs = rel
            g = lhs - rhs
            lb, ub = (-math.inf, 0.0) if op == "<=" else (0.0, math.inf)
        else:
            g, lb, ub = rel
        self.cons.append((Expr.wrap(g), float(lb), float(ub), linear))

    NLconstraint = constraint

    def build(self):
        n = len(self.lb)
        lin = [c[3] for c in self.cons]
        lin = None if any(v is None for v in lin) else lin
        return ExprNLP(n, self.obj, [c[0] for c in self.cons], lin, self.obj_linear)

    def solve(self):
        d = self.build()
        self.internal_model = NonlinearModel(self.solver)
        self.internal_model.loadproblem(len(self.lb), len(self.cons), self.lb, self.ub, [c[1] for c in self.cons],
                                        [c[2] for c in self.cons], self.sense, d)
        self._status = self.internal_model.optimize()
        return self._status

    def getobjectivevalue(self):
        return self.internal_model.getobjval()

    def getvalue(self, v=None):
        x = self.internal_model.getsolution()
        if v is None:
            return x[:len(self.lb)]
        if isinstance(v, (list, tuple)):
            return np.array([x[e.args[0]] for e in v])
        return float(x[v.args[0]])
