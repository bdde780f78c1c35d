"""Seeded synthetic convex NLPs with a planted optimum.

The reference ships no benchmark instances; these are
the instances BASELINE.json's configs are measured on.  Everything is plain
numpy arrays in the layout `ktn_load_problem` takes (include/katana_hip.h), so
the very same arrays feed the HIP path, the CPU oracle and the fixtures.

    variables   x in [-B, B]^n                        (box => LP never unbounded)
    linear rows a_r'x <= a_r'xhat + slack             m_lin rows, 8 nnz each
    NL rows     g_i(x) = sum_{j in S_i} atom_ij(x_j) - r_i <= 0,   |S_i| = k
                  quad family :  a (x - c)^2
                  explog family: w exp(a x)  (half of S_i)  and  -v log(x + B + 1)
    objective   c = -sum_{i active} lambda_i grad g_i(xhat)   => xhat is KKT-optimal,
                optimum c'xhat known in closed form (no Ipopt needed)

A fraction `active_frac` of the NL rows is active at xhat (g_i(xhat) = 0), the
rest have slack U(0.1, 1).

`vertex=True` (default; deviation from SURVEY.md section 8d, see DESIGN.md
"Instances"): xhat is made a NON-DEGENERATE VERTEX of the feasible set.  A
fraction `lin_active_frac` of the linear rows is active too, every active row
is matched to a private pivot column, and every non-pivot variable sits on one
of its bounds (l_j = xhat_j or u_j = xhat_j) with a positive multiplier, so
that #active constraints = n with a structurally nonsingular active Jacobian.
Kelley's method then has its Newton-like local convergence (tens of ECP
iterations).  With vertex=False the optimum lies on a smooth face, where the
cutting-plane method needs thousands of iterations already at n ~ 20 (the
reference's own test/misc.jl behaviour) -- that regime is covered by the KATs.
"""
from dataclasses import dataclass, field

import numpy as np

ATOM_LIN, ATOM_QUAD, ATOM_EXP, ATOM_NEGLOG = 0, 1, 2, 3

# BASELINE.json configs (k = nnz per nonlinear row fixed in SURVEY.md section 8)
CONFIGS = {
    "cfg2": dict(n=10_000, m_nl=1_000, k=64, family="quad"),
    # BASELINE.json configs[1] says "convex QP": the same rows under the quadratic objective 1/2 |x - x0|_D^2, which enters
    # through the (dense) epigraph row f(x) - t <= 0 (src/nlpeval.jl:42-63; SURVEY.md section 8d "cfg2 QP variant")
    "cfg2_qp": dict(n=10_000, m_nl=1_000, k=64, family="quad", objective="quad"),
    "cfg3": dict(n=100_000, m_nl=10_000, k=32, family="explog"),
    # the north star's shape with the nonlinear-objective path switched on (src/nlpeval.jl:42-63): every epigraph cut has 1e5 entries
    "cfg3_qp": dict(n=100_000, m_nl=10_000, k=32, family="explog", objective="quad"),
    "cfg3_hbm": dict(n=100_000, m_nl=10_000, k=2048, family="explog"),
    "cfg4": dict(n=100_000, m_nl=1_000_000, k=32, family="explog"),
    "cfg5_one": dict(n=1_000, m_nl=100, k=16, family="explog"),
}


@dataclass
class SeparableInstance:
    n: int
    l_var: np.ndarray
    u_var: np.ndarray
    sense: str
    # all constraint rows (linear rows first), CSR with per-entry atoms
    rowptr: np.ndarray
    col: np.ndarray
    kind: np.ndarray
    p0: np.ndarray
    p1: np.ndarray
    rconst: np.ndarray
    l_constr: np.ndarray
    u_constr: np.ndarray
    # separable objective
    obj_col: np.ndarray
    obj_kind: np.ndarray
    obj_p0: np.ndarray
    obj_p1: np.ndarray
    obj_const: float
    # planted solution
    xhat: np.ndarray
    opt_obj: float
    m_lin: int
    m_nl: int
    meta: dict = field(default_factory=dict)

    @property
    def num_constr(self):
        return len(self.rowptr) - 1


def _distinct_sorted_cols(rng, rows, k, n):
    """k distinct, ascending column indices per row (vectorised)."""
    base = np.sort(rng.integers(0, n - k + 1, size=(rows, k)), axis=1)
    return base + np.arange(k)[None, :]


def atom_value_deriv(kind, p0, p1, xv):
    val = np.empty_like(xv)
    der = np.empty_like(xv)
    m = kind == ATOM_LIN
    val[m], der[m] = p0[m] * xv[m], p0[m]
    m = kind == ATOM_QUAD
    d = xv[m] - p1[m]
    val[m], der[m] = p0[m] * d * d, 2.0 * p0[m] * d
    m = kind == ATOM_EXP
    e = p0[m] * np.exp(p1[m] * xv[m])
    val[m], der[m] = e, p1[m] * e
    m = kind == ATOM_NEGLOG
    s = xv[m] + p1[m]
    val[m], der[m] = -p0[m] * np.log(s), -p0[m] / s
    return val, der


def make_instance(n, m_nl, k, family="explog", seed=0, m_lin=None, lin_nnz=8, B=10.0,
                  active_frac=0.05, objective="linear", vertex=True, lin_active_frac=0.3,
                  pivot_boost=True, shared_column=False, bound_frac=None):
    """`bound_frac` is the *degeneracy dial*: the fraction of the non-pivot variables that is pinned on a bound with a
    positive multiplier.  1.0 (default with `vertex=True`) is the non-degenerate vertex family; 0.0 is SURVEY.md section
    8d's smooth-face generator itself (same as `vertex=False`: no variable pinned, every linear row slack, c = -sum lambda
    grad g); values in between keep the active linear rows and the pivot matching and leave the un-pinned variables inside
    the box with no bound multiplier, so the optimum lies on a curved face of dimension ~ (1 - bound_frac) * #non-pivot
    variables: xhat stays a KKT point (the planted objective stays exact) but need no longer be the only minimiser, and
    Kelley's method no longer has a vertex to lock on to.  `pivot_boost=False` additionally drops the conditioning help
    (pivot entries as random as the others).  The random stream is that of the previous generator when bound_frac is 1.

    `shared_column` (or a family name ending in "+t"): one more variable t that EVERY nonlinear row contains,
    g_i(x) - t <= r_i - that (the shape of min-max and epigraph models: minimise t subject to g_i(x) <= t).  Every cut then
    has an entry in t's column, which grows by up to m_nl entries per sweep -- the LP's long-column case.  t sits on its
    lower bound that with multiplier 1 and cost sum(lambda) + 1, so the planted point stays a non-degenerate KKT vertex."""
    if family.endswith("+t"):
        family, shared_column = family[:-2], True
    if bound_frac is None:
        bound_frac = 1.0 if vertex else 0.0
    if not 0.0 <= bound_frac <= 1.0:
        raise ValueError("bound_frac must lie in [0, 1]")
    if bound_frac == 0.0:
        vertex = False
    rng = np.random.default_rng(seed)
    m_lin = n // 2 if m_lin is None else m_lin
    lin_nnz = min(lin_nnz, n)
    k = min(k, n)
    xhat = rng.uniform(-1.0, 1.0, size=n)

    # ---- structure, active sets, pivot matching -----------------------------
    lcol2 = _distinct_sorted_cols(rng, m_lin, lin_nnz, n)
    ncol2 = _distinct_sorted_cols(rng, m_nl, k, n)
    active = rng.random(m_nl) < active_frac
    if m_nl and not active.any():
        active[0] = True
    lin_active = np.zeros(m_lin, dtype=bool)
    used = np.zeros(n, dtype=bool)
    lin_pivot = np.full(m_lin, -1)      # position (0..lin_nnz-1) of the pivot entry in the row
    nl_pivot = np.full(m_nl, -1)
    if vertex:
        lin_active = rng.random(m_lin) < lin_active_frac
        cand = [(0, r) for r in np.nonzero(lin_active)[0]] + [(1, i) for i in np.nonzero(active)[0]]
        for idx in rng.permutation(len(cand)):
            which, r = cand[idx]
            cols_r = lcol2[r] if which == 0 else ncol2[r]
            free = np.nonzero(~used[cols_r])[0]
            if len(free):
                pos = free[rng.integers(len(free))]
                used[cols_r[pos]] = True
                if which == 0:
                    lin_pivot[r] = pos
                else:
                    nl_pivot[r] = pos
            elif which == 0:
                lin_active[r] = False
            else:
                active[r] = False

    # ---- linear block ------------------------------------------------------
    lval2 = rng.standard_normal((m_lin, lin_nnz))
    if vertex and pivot_boost:
        ra = np.nonzero(lin_active)[0]
        lval2[ra, lin_pivot[ra]] = np.sign(lval2[ra, lin_pivot[ra]]) * rng.uniform(2.0, 4.0, size=len(ra))
    lcol, lval = lcol2.reshape(-1), lval2.reshape(-1)
    lrow = np.repeat(np.arange(m_lin), lin_nnz)
    ax = np.bincount(lrow, weights=lval * xhat[lcol], minlength=m_lin)
    lin_ub = ax + np.where(lin_active, 0.0, rng.uniform(0.1, 1.0, size=m_lin))

    # ---- nonlinear block ---------------------------------------------------
    if family == "quad":
        nkind = np.full((m_nl, k), ATOM_QUAD, dtype=np.uint8)
        np0 = rng.uniform(0.5, 2.0, size=(m_nl, k))
        np1 = rng.standard_normal((m_nl, k))
        if vertex and pivot_boost:
            # pivot atom: derivative 2a(x-c) of magnitude U(4,8) -> dominates the row's basis entries
            ra = np.nonzero(active)[0]
            xa = xhat[ncol2[ra, nl_pivot[ra]]]
            np0[ra, nl_pivot[ra]] = 2.0
            np1[ra, nl_pivot[ra]] = xa - rng.choice([-1.0, 1.0], size=len(ra)) * rng.uniform(1.0, 2.0, size=len(ra))
    elif family == "explog":
        nkind = np.full((m_nl, k), ATOM_EXP, dtype=np.uint8)
        neg = rng.random((m_nl, k)) < 0.5
        np0 = rng.uniform(0.1, 1.0, size=(m_nl, k))
        np1 = rng.uniform(-0.5, 0.5, size=(m_nl, k))
        if vertex and pivot_boost:
            # pivot atom: w exp(a x) with |d/dx| = U(1,2) at xhat
            ra = np.nonzero(active)[0]
            pa = nl_pivot[ra]
            neg[ra, pa] = False
            a = rng.choice([-0.5, 0.5], size=len(ra))
            np1[ra, pa] = a
            np0[ra, pa] = rng.uniform(1.0, 2.0, size=len(ra)) / (0.5 * np.exp(a * xhat[ncol2[ra, pa]]))
        nkind[neg] = ATOM_NEGLOG
        np1[neg] = B + 1.0
    else:
        raise ValueError(family)
    ncol, nkind, np0, np1 = ncol2.reshape(-1), nkind.reshape(-1), np0.reshape(-1), np1.reshape(-1)
    nrow = np.repeat(np.arange(m_nl), k)
    val, der = atom_value_deriv(nkind, np0, np1, xhat[ncol])
    gx = np.bincount(nrow, weights=val, minlength=m_nl)
    slack = np.where(active, 0.0, rng.uniform(0.1, 1.0, size=m_nl))
    r = gx + slack
    lam = np.where(active, rng.uniform(0.5, 1.5, size=m_nl), 0.0)
    mu = np.where(lin_active, rng.uniform(0.5, 1.5, size=m_lin), 0.0)
    l_var = np.full(n, -B)
    u_var = np.full(n, B)
    c = -np.bincount(ncol, weights=(lam[nrow] * der), minlength=n)
    if vertex:
        at_bound = ~used
        if bound_frac < 1.0:
            at_bound &= rng.random(n) < bound_frac
        lower = at_bound & (rng.random(n) < 0.5)
        upper = at_bound & ~lower
        l_var[lower] = xhat[lower]
        u_var[upper] = xhat[upper]
        nu = rng.uniform(0.5, 1.5, size=n)
        c -= np.bincount(lcol, weights=(mu[lrow] * lval), minlength=n)
        c[lower] += nu[lower]
        c[upper] -= nu[upper]

    # ---- objective ---------------------------------------------------------
    if objective == "linear":
        nz = np.nonzero(c)[0]
        obj_col, obj_kind = nz, np.zeros(len(nz), dtype=np.uint8)
        obj_p0, obj_p1 = c[nz], np.zeros(len(nz))
        opt = float(c @ xhat)
    elif objective == "quad":
        # f(x) = sum_j 0.5 d_j (x_j - x0_j)^2 with grad f(xhat) = c  (dense epigraph row)
        dj = rng.uniform(0.5, 2.0, size=n)
        x0 = xhat - c / dj
        obj_col, obj_kind = np.arange(n), np.full(n, ATOM_QUAD, dtype=np.uint8)
        obj_p0, obj_p1 = 0.5 * dj, x0
        opt = float(np.sum(0.5 * dj * (xhat - x0) ** 2))
    else:
        raise ValueError(objective)

    if shared_column:
        if objective != "linear":
            raise ValueError("shared_column needs the linear objective")
        that = 0.25
        ncol = np.concatenate([ncol2, np.full((m_nl, 1), n)], axis=1).reshape(-1)
        nkind = np.concatenate([nkind.reshape(m_nl, k), np.full((m_nl, 1), ATOM_LIN, dtype=np.uint8)], axis=1).reshape(-1)
        np0 = np.concatenate([np0.reshape(m_nl, k), np.full((m_nl, 1), -1.0)], axis=1).reshape(-1)
        np1 = np.concatenate([np1.reshape(m_nl, k), np.zeros((m_nl, 1))], axis=1).reshape(-1)
        r = r - that                                   # g_i(xhat) - that - r' = -slack_i
        ct = float(lam.sum()) + 1.0
        obj_col, obj_kind = np.append(obj_col, n), np.append(obj_kind, np.uint8(ATOM_LIN))
        obj_p0, obj_p1 = np.append(obj_p0, ct), np.append(obj_p1, 0.0)
        opt += ct * that
        xhat = np.append(xhat, that)
        l_var, u_var = np.append(l_var, that), np.append(u_var, B)
        n, k = n + 1, k + 1
    rowptr = np.concatenate([np.arange(m_lin + 1) * lin_nnz,
                             m_lin * lin_nnz + np.arange(1, m_nl + 1) * k]).astype(np.int64)
    inst = SeparableInstance(
        n=n, l_var=l_var, u_var=u_var, sense="Min",
        rowptr=rowptr,
        col=np.concatenate([lcol, ncol]).astype(np.int32),
        kind=np.concatenate([np.zeros(len(lcol), dtype=np.uint8), nkind]).astype(np.uint8),
        p0=np.concatenate([lval, np0]), p1=np.concatenate([np.zeros(len(lcol)), np1]),
        rconst=np.concatenate([np.zeros(m_lin), -r]),
        l_constr=np.full(m_lin + m_nl, -np.inf),
        u_constr=np.concatenate([lin_ub, np.zeros(m_nl)]),
        obj_col=obj_col.astype(np.int32), obj_kind=obj_kind, obj_p0=obj_p0, obj_p1=obj_p1, obj_const=0.0,
        xhat=xhat, opt_obj=opt, m_lin=m_lin, m_nl=m_nl,
        meta=dict(n=n, m_nl=m_nl, k=k, family=family, seed=seed, m_lin=m_lin, lin_nnz=lin_nnz, B=B,
                  active_frac=active_frac, objective=objective, n_active=int(active.sum()),
                  vertex=bool(vertex), bound_frac=float(bound_frac), pivot_boost=bool(pivot_boost), n_lin_active=int(lin_active.sum()), shared_column=bool(shared_column),
                  # planted multipliers: an x with g_i(x) <= eps on the NL rows, a_r'x <= b_r + delta on the linear rows and the
                  # bounds met exactly has  c'x >= c'xhat - eps * lam_sum - delta * mu_sum  (Lagrangian bound at xhat)
                  lam_sum=float(lam.sum()), mu_sum=float(mu.sum())))
    return inst


def make_config(name, seed=0, **overrides):
    kw = dict(CONFIGS[name])
    kw.update(overrides)
    return make_instance(seed=seed, **kw)


def fuse_instances(insts):
    """Block-diagonal union of independent instances: one NLP whose variables, rows and (summed) objective are
    those of all the instances side by side (linear rows of every instance first, then the NL rows).  Solving it is
    solving all of them at once -- every kernel launch of the engine then serves the whole batch
    (katana_jl_amd.batch.solve_batch(..., fused=True)).  Returns (fused instance, variable offsets)."""
    offs = np.concatenate([[0], np.cumsum([i.n for i in insts])]).astype(np.int64)
    rps = [np.asarray(i.rowptr) for i in insts]
    cut = [int(rp[i.m_lin]) for rp, i in zip(rps, insts)]                # first entry of the NL rows of each instance

    def cat(attr, dtype=None):
        """linear-row part of every instance, then the NL-row part of every instance (per entry)"""
        a = [np.asarray(getattr(i, attr)) for i in insts]
        out = np.concatenate([x[:c] for x, c in zip(a, cut)] + [x[c:] for x, c in zip(a, cut)])
        return out if dtype is None else out.astype(dtype, copy=False)

    def cat_rows(attr):
        a = [np.asarray(getattr(i, attr)) for i in insts]
        return np.concatenate([x[:i.m_lin] for x, i in zip(a, insts)] + [x[i.m_lin:] for x, i in zip(a, insts)])

    lens = np.concatenate([np.diff(rp[:i.m_lin + 1]) for rp, i in zip(rps, insts)] + [np.diff(rp[i.m_lin:]) for rp, i in zip(rps, insts)])
    cols = [np.asarray(i.col, dtype=np.int32) for i in insts]
    o32 = offs.astype(np.int32)
    col = np.concatenate([x[:c] + o for x, c, o in zip(cols, cut, o32)] + [x[c:] + o for x, c, o in zip(cols, cut, o32)])
    fused = SeparableInstance(
        n=int(offs[-1]), l_var=np.concatenate([i.l_var for i in insts]), u_var=np.concatenate([i.u_var for i in insts]),
        sense=insts[0].sense, rowptr=np.concatenate([[0], np.cumsum(lens)]).astype(np.int64),
        col=col, kind=cat("kind", np.uint8), p0=cat("p0"), p1=cat("p1"), rconst=cat_rows("rconst"),
        l_constr=cat_rows("l_constr"), u_constr=cat_rows("u_constr"),
        obj_col=np.concatenate([np.asarray(i.obj_col, dtype=np.int32) + o for i, o in zip(insts, o32)]),
        obj_kind=np.concatenate([i.obj_kind for i in insts]).astype(np.uint8, copy=False),
        obj_p0=np.concatenate([i.obj_p0 for i in insts]), obj_p1=np.concatenate([i.obj_p1 for i in insts]),
        obj_const=float(sum(i.obj_const for i in insts)), xhat=np.concatenate([i.xhat for i in insts]),
        opt_obj=float(sum(i.opt_obj for i in insts)), m_lin=int(sum(i.m_lin for i in insts)),
        m_nl=int(sum(i.m_nl for i in insts)),
        meta=dict(fused=len(insts), lam_sum=float(sum(i.meta.get("lam_sum", 0.0) for i in insts)),
                  mu_sum=float(sum(i.meta.get("mu_sum", 0.0) for i in insts)),
                  obj_ptr=np.concatenate([[0], np.cumsum([len(i.obj_col) for i in insts])]).astype(np.int64)))
    assert all(i.sense == fused.sense for i in insts)
    return fused, offs


def make_lp(n, m, seed=0, nnz_row=8, B=10.0, active_frac=0.3, degenerate_frac=0.0, bad_scale_decades=0.0, free_frac=0.0):
    """A random sparse LP with a planted primal-dual optimal pair, in the layout of `make_instance` (every row a separable row
    of LIN atoms, no nonlinear row): the LP-only battery of tests/test_gpu_lp.py (engine LP against HiGHS).

        min c'x   s.t.  a_r'x <= b_r (m rows, nnz_row entries each),   l <= x <= u

    A fraction `active_frac` of the rows is tight at xhat and carries a multiplier; `degenerate_frac` more rows are tight WITHOUT a
    multiplier (a primal-degenerate vertex: more tight constraints than the vertex needs); every variable that is in no
    multiplier-carrying row sits on a bound with a positive reduced cost, except a fraction `free_frac` that stays strictly
    inside a wide box with zero reduced cost (dual degeneracy).  `bad_scale_decades` = d scales every row by 10^U(-d, d)."""
    rng = np.random.default_rng(seed)
    nnz_row = min(nnz_row, n)
    xhat = rng.uniform(-1.0, 1.0, size=n)
    col2 = _distinct_sorted_cols(rng, m, nnz_row, n)
    val2 = rng.standard_normal((m, nnz_row))
    active = rng.random(m) < active_frac
    tight = active | (rng.random(m) < degenerate_frac)
    mu = np.where(active, rng.uniform(0.5, 1.5, size=m), 0.0)
    col, val = col2.reshape(-1), val2.reshape(-1)
    row = np.repeat(np.arange(m), nnz_row)
    ax = np.bincount(row, weights=val * xhat[col], minlength=m)
    ub = ax + np.where(tight, 0.0, rng.uniform(0.1, 1.0, size=m))
    c = -np.bincount(col, weights=mu[row] * val, minlength=n)
    l_var, u_var = np.full(n, -B), np.full(n, B)
    in_active = np.zeros(n, dtype=bool)
    in_active[col2[active].reshape(-1)] = True
    pin = ~in_active & (rng.random(n) >= free_frac)
    lower = pin & (rng.random(n) < 0.5)
    upper = pin & ~lower
    l_var[lower] = xhat[lower]
    u_var[upper] = xhat[upper]
    nu = rng.uniform(0.5, 1.5, size=n)
    c[lower] += nu[lower]
    c[upper] -= nu[upper]
    if bad_scale_decades > 0.0:
        s = 10.0 ** rng.uniform(-bad_scale_decades, bad_scale_decades, size=m)
        val = val * s[row]
        ub = ub * s
    nz = np.nonzero(c)[0]
    return SeparableInstance(
        n=n, l_var=l_var, u_var=u_var, sense="Min", rowptr=(np.arange(m + 1) * nnz_row).astype(np.int64),
        col=col.astype(np.int32), kind=np.zeros(len(col), dtype=np.uint8), p0=val, p1=np.zeros(len(col)),
        rconst=np.zeros(m), l_constr=np.full(m, -np.inf), u_constr=ub,
        obj_col=nz.astype(np.int32), obj_kind=np.zeros(len(nz), dtype=np.uint8), obj_p0=c[nz], obj_p1=np.zeros(len(nz)),
        obj_const=0.0, xhat=xhat, opt_obj=float(c @ xhat), m_lin=m, m_nl=0,
        meta=dict(n=n, m_lin=m, seed=seed, lp=True, active_frac=active_frac, degenerate_frac=degenerate_frac,
                  bad_scale_decades=bad_scale_decades, free_frac=free_frac, lam_sum=0.0, mu_sum=float(mu.sum())))


def lp_battery_case(i):
    """Case i of the LP-only battery (tools/lp_battery.py, tests/test_gpu_lp.py): sizes 1e3 ... 1e4 columns, 0.5 ... 2 rows per
    column, and five kinds in turn -- plain, primal-degenerate vertex, rows scaled over six decades, degenerate + scaled +
    dual-degenerate (free columns with zero reduced cost), heavily degenerate.  Returns the keyword arguments of make_lp."""
    rng = np.random.default_rng(1000 + i)
    n = int(10 ** rng.uniform(3.0, 4.0))
    m = int(n * rng.uniform(0.5, 2.0))
    kw = dict(n=n, m=m, seed=i, nnz_row=int(rng.integers(4, 17)))
    kind = i % 5
    if kind == 1: kw.update(degenerate_frac=0.3)
    if kind == 2: kw.update(bad_scale_decades=3.0)
    if kind == 3: kw.update(degenerate_frac=0.2, bad_scale_decades=2.0, free_frac=0.2)
    if kind == 4: kw.update(active_frac=0.6, degenerate_frac=0.4)
    return kw
