"""CPU mirror of the product's GPU LP solver -- TEST INFRASTRUCTURE ONLY.

The reference delegates its LP to GLPK; the product replaces it by a restarted
primal-dual hybrid gradient method (PDLP family) on the GPU
(katana.jl_amd/csrc/pdlp.hip).  This file is a numpy statement of the *same*
algorithm, with the same interface as oracle.lp.LinearModel, used by tests to
(a) check single PDHG steps / KKT residuals of the kernels and (b) rehearse the
ECP + first-order-LP combination on CPU.  It is not on any product path.

    min c'x  s.t.  l <= x <= u,  lo <= Ax <= hi          (Max: c -> -c)
    saddle:  L(x,y) = c'x - y'Ax + sum_i (lo_i y_i^+ - hi_i y_i^-)
    x+ = proj_[l,u](x - tau (c - A'y))
    v  = y - sigma A(2x+ - x);  y+ = v + sigma clip(-v/sigma, lo, hi)
with tau = eta/omega, sigma = eta*omega, diagonal (Ruiz + Pock-Chambolle)
preconditioning, iterate averaging, KKT-based adaptive restarts and primal
weight updates (Applegate et al. 2021; Lu & Yang 2023 for the KKT restart).
"""
import numpy as np
import scipy.sparse as sp

INF = float("inf")


class PdlpParams:
    def __init__(self, eps=1e-8, max_iter=200000, check_every=64, ruiz_iters=10,
                 beta_suff=0.2, beta_nec=0.8, beta_art=0.36, theta=0.5, verbose=False):
        self.eps = eps
        self.max_iter = max_iter
        self.check_every = check_every
        self.ruiz_iters = ruiz_iters
        self.beta_suff, self.beta_nec, self.beta_art = beta_suff, beta_nec, beta_art
        self.theta = theta
        self.verbose = verbose


def _proj(v, lo, hi):
    return np.minimum(np.maximum(v, lo), hi)


def scale_matrix(A, ruiz_iters):
    """Ruiz (inf-norm) then Pock-Chambolle (alpha=1) diagonal scaling.
    Returns (dr, dc) with A_hat = diag(dr) A diag(dc)."""
    m, n = A.shape
    dr = np.ones(m)
    dc = np.ones(n)
    Ah = A.copy().tocsr()
    absA = abs(Ah)
    for _ in range(ruiz_iters):
        rmax = np.sqrt(np.asarray(absA.max(axis=1).todense()).ravel()) if m else np.ones(0)
        cmax = np.sqrt(np.asarray(absA.max(axis=0).todense()).ravel()) if m else np.ones(n)
        rmax[rmax == 0] = 1.0
        cmax[cmax == 0] = 1.0
        dr /= rmax
        dc /= cmax
        absA = sp.diags(1.0 / rmax) @ absA @ sp.diags(1.0 / cmax)
    if m:
        r1 = np.sqrt(np.asarray(absA.sum(axis=1)).ravel())
        c1 = np.sqrt(np.asarray(absA.sum(axis=0)).ravel())
        r1[r1 == 0] = 1.0
        c1[c1 == 0] = 1.0
        dr /= r1
        dc /= c1
    return dr, dc


def kkt(A, AT, c, l, u, lo, hi, x, y):
    """(primal residual, dual residual, primal obj, dual obj), unscaled norms."""
    ax = A @ x
    pres = np.linalg.norm(ax - _proj(ax, lo, hi))
    r = c - AT @ y
    rp = np.maximum(r, 0.0)
    rm = np.maximum(-r, 0.0)
    # reduced cost that cannot be absorbed by a finite bound is dual residual
    bad = np.where(np.isfinite(l), 0.0, rp) + np.where(np.isfinite(u), 0.0, rm)
    dres = np.linalg.norm(bad)
    pobj = float(c @ x)
    yp = np.maximum(y, 0.0)
    ym = np.maximum(-y, 0.0)
    with np.errstate(invalid="ignore"):
        dobj = float(np.sum(np.where(yp > 0, lo * yp, 0.0)) - np.sum(np.where(ym > 0, hi * ym, 0.0))
                     + np.sum(np.where(np.isfinite(l) & (rp > 0), l * rp, 0.0))
                     - np.sum(np.where(np.isfinite(u) & (rm > 0), u * rm, 0.0)))
    return pres, dres, pobj, dobj


def solve_lp(A, c, l, u, lo, hi, x0=None, y0=None, params=None, omega0=None):
    """Restarted averaged PDHG on the scaled problem; returns dict with unscaled x, y."""
    P = params or PdlpParams()
    m, n = A.shape
    A = A.tocsr()
    dr, dc = scale_matrix(A, P.ruiz_iters)
    Ah = (sp.diags(dr) @ A @ sp.diags(dc)).tocsr()
    AhT = Ah.T.tocsr()
    ch = c * dc
    lh, uh = l / dc, u / dc
    loh, hih = lo * dr, hi * dr
    x = np.zeros(n) if x0 is None else x0 / dc
    x = _proj(x, lh, uh)
    y = np.zeros(m) if y0 is None else y0 / dr
    # step size from a power-iteration estimate of ||A_hat||_2
    v = np.ones(n) / np.sqrt(max(n, 1))
    smax = 1.0
    for _ in range(30):
        w = Ah @ v
        v2 = AhT @ w
        nv = np.linalg.norm(v2)
        if nv == 0:
            break
        smax = np.sqrt(nv / max(np.linalg.norm(v), 1e-300))
        v = v2 / nv
    eta = 0.99 / max(smax, 1e-12)
    nc, nb = np.linalg.norm(ch), np.linalg.norm(np.where(np.isfinite(loh), loh, 0.0) + np.where(np.isfinite(hih) & ~np.isfinite(loh), hih, 0.0))
    if omega0 is not None:
        omega = omega0
    else:
        omega = nc / nb if nc > 0 and nb > 0 else 1.0
    bnorm = np.linalg.norm(np.concatenate([np.where(np.isfinite(lo), lo, 0.0), np.where(np.isfinite(hi), hi, 0.0)]))
    cnorm = np.linalg.norm(c)

    def rel_err(xs, ys):
        pres, dres, pobj, dobj = kkt(A, A.T, c, l, u, lo, hi, xs * dc, ys * dr)
        return (pres / (1 + bnorm), dres / (1 + cnorm), abs(pobj - dobj) / (1 + abs(pobj) + abs(dobj)), pobj, dobj)

    def kkt_err(xs, ys):
        pres, dres, pobj, dobj = kkt(Ah, AhT, ch, lh, uh, loh, hih, xs, ys)
        return np.sqrt(omega * pres * pres + dres * dres / omega + (pobj - dobj) ** 2)

    xs, ys = x.copy(), y.copy()          # restart point
    xsum, ysum, cnt = np.zeros(n), np.zeros(m), 0
    err_restart = kkt_err(x, y)
    err_prev_cand = err_restart
    k_restart = 0
    aty = AhT @ y
    it = 0
    status = "IterLimit"
    while it < P.max_iter:
        tau, sigma = eta / omega, eta * omega
        xn = _proj(x - tau * (ch - aty), lh, uh)
        vv = y - sigma * (Ah @ (2 * xn - x))
        with np.errstate(invalid="ignore"):
            yn = vv + sigma * _proj(-vv / sigma, loh, hih)
        yn = np.where(np.isfinite(yn), yn, 0.0)
        x, y = xn, yn
        aty = AhT @ y
        xsum += x
        ysum += y
        cnt += 1
        it += 1
        if it % P.check_every:
            continue
        xa, ya = xsum / cnt, ysum / cnt
        e_avg, e_cur = kkt_err(xa, ya), kkt_err(x, y)
        if e_avg <= e_cur:
            xc, yc, ec = xa, ya, e_avg
        else:
            xc, yc, ec = x, y, e_cur
        rp, rd, rg, pobj, dobj = rel_err(xc, yc)
        if P.verbose:
            print("it %6d  pres %.2e dres %.2e gap %.2e  pobj %.10g  omega %.3g" % (it, rp, rd, rg, pobj, omega))
        if max(rp, rd, rg) <= P.eps:
            x, y = xc, yc
            status = "Optimal"
            break
        do_restart = (ec <= P.beta_suff * err_restart
                      or (ec <= P.beta_nec * err_restart and ec > err_prev_cand)
                      or (it - k_restart) >= P.beta_art * it)
        err_prev_cand = ec
        if do_restart:
            dx, dy = np.linalg.norm(xc - xs), np.linalg.norm(yc - ys)
            if dx > 1e-300 and dy > 1e-300:
                omega = np.exp(P.theta * np.log(dy / dx) + (1 - P.theta) * np.log(omega))
            x, y = xc.copy(), yc.copy()
            xs, ys = x.copy(), y.copy()
            xsum[:] = 0
            ysum[:] = 0
            cnt = 0
            aty = AhT @ y
            err_restart = kkt_err(x, y)
            err_prev_cand = err_restart
            k_restart = it
    rp, rd, rg, pobj, dobj = rel_err(x, y)
    return dict(x=x * dc, y=y * dr, status=status, iters=it, pobj=pobj, dobj=dobj,
                rel=(rp, rd, rg), omega=omega)


class PdlpLinearModel:
    """Drop-in for oracle.lp.LinearModel backed by the PDHG mirror (dev/test only)."""

    def __init__(self, params=None, threads=1):
        self.params = params or PdlpParams()
        self.n = 0
        self.sense = "Min"
        self.c = np.zeros(0)
        self.c0 = 0.0
        self.l = np.zeros(0)
        self.u = np.zeros(0)
        self.blocks, self.row_lo, self.row_hi = [], [], []
        self.nnz = 0
        self.x = np.zeros(0)
        self.y = np.zeros(0)
        self.omega = None
        self.num_solves = 0
        self.pdhg_iters = 0
        self.iters_log = []

    def add_variables(self, l, u):
        l = np.asarray(l, dtype=np.float64)
        u = np.asarray(u, dtype=np.float64)
        self.l = np.concatenate([self.l, l])
        self.u = np.concatenate([self.u, u])
        self.c = np.concatenate([self.c, np.zeros(len(l))])
        self.x = np.concatenate([self.x, np.zeros(len(l))])
        self.n += len(l)

    def set_objective(self, sense, cols, coefs, constant=0.0):
        self.sense = sense
        c = np.zeros(self.n)
        np.add.at(c, np.asarray(cols, dtype=np.int64), np.asarray(coefs, dtype=np.float64))
        self.c, self.c0 = c, float(constant)

    def add_rows(self, rowptr, cols, vals, lo, hi, assume_unique=False):
        rowptr = np.asarray(rowptr, dtype=np.int64)
        if len(rowptr) <= 1:
            return
        self.blocks.append((rowptr.copy(), np.asarray(cols, dtype=np.int64).copy(),
                            np.asarray(vals, dtype=np.float64).copy()))
        lo = np.asarray(lo, dtype=np.float64).copy()
        hi = np.asarray(hi, dtype=np.float64).copy()
        lo[np.isnan(lo)] = -INF
        hi[np.isnan(hi)] = INF
        self.row_lo.extend(lo.tolist())
        self.row_hi.extend(hi.tolist())
        self.y = np.concatenate([self.y, np.zeros(len(lo))])
        self.nnz += len(cols)

    def add_row(self, cols, vals, lo, hi):
        self.add_rows([0, len(cols)], cols, vals, [lo], [hi])

    @property
    def num_rows(self):
        return len(self.row_lo)

    def matrix(self):
        m = self.num_rows
        if m == 0:
            return sp.csr_matrix((0, self.n))
        ptr = [np.zeros(1, dtype=np.int64)]
        off = 0
        cc, vv = [], []
        for p, c, v in self.blocks:
            ptr.append(p[1:] + off)
            off += p[-1]
            cc.append(c)
            vv.append(v)
        A = sp.csr_matrix((np.concatenate(vv), np.concatenate(cc), np.concatenate(ptr)), shape=(m, self.n))
        A.sum_duplicates()
        return A

    def solve(self, eps=None):
        A = self.matrix()
        s = -1.0 if self.sense == "Max" else 1.0
        P = self.params
        if eps is not None:
            P = PdlpParams(**{**self.params.__dict__, "eps": eps})
        if len(self.x) < self.n:
            self.x = np.concatenate([self.x, np.zeros(self.n - len(self.x))])
        res = solve_lp(A, s * self.c, self.l, self.u, np.asarray(self.row_lo), np.asarray(self.row_hi),
                       x0=self.x, y0=self.y, params=P, omega0=self.omega)
        self.x, self.y, self.omega = res["x"], res["y"], res["omega"]
        self.num_solves += 1
        self.pdhg_iters += res["iters"]
        self.iters_log.append(res["iters"])
        self.last = res
        self._obj = float(self.c @ self.x) + self.c0
        return "Optimal" if res["status"] == "Optimal" else "Error"

    def getsolution(self):
        return self.x.copy()

    def getobjval(self):
        return self._obj


def solve_lp_halpern(A, c, l, u, lo, hi, x0=None, y0=None, params=None, omega0=None, rho=1.0):
    """Reflected restarted Halpern PDHG (Lu & Yang 2024; cuPDLPx 2025), fixed step.

        zt      = PDHG(z_k)                       (one x-step, one y-step)
        z_{k+1} = (k+1)/(k+2) ((1+rho) zt - rho z_k) + 1/(k+2) z_0
    restart on the fixed-point residual r(z) = ||z - PDHG(z)||_M,
    M = [[omega/eta I, -A'], [-A, 1/(eta omega) I]].

    The product adds two safeguards that need state this mirror does not have (csrc/lp.hip
    lp_solve): backing eta off when the residual stops moving, and moving multiplier mass between
    the cuts of one nonlinear row when PDHG idles between two near-parallel cuts (k_consolidate).
    """
    P = params or PdlpParams()
    m, n = A.shape
    A = A.tocsr()
    dr, dc = scale_matrix(A, P.ruiz_iters)
    Ah = (sp.diags(dr) @ A @ sp.diags(dc)).tocsr()
    AhT = Ah.T.tocsr()
    ch = c * dc
    lh, uh = l / dc, u / dc
    loh, hih = lo * dr, hi * dr
    x = _proj(np.zeros(n) if x0 is None else x0 / dc, lh, uh)
    y = np.zeros(m) if y0 is None else y0 / dr
    v = np.ones(n) / np.sqrt(max(n, 1))
    smax = 1.0
    for _ in range(40):
        v2 = AhT @ (Ah @ v)
        nv = np.linalg.norm(v2)
        if nv == 0:
            break
        smax = np.sqrt(nv / max(np.linalg.norm(v), 1e-300))
        v = v2 / nv
    eta = 0.998 / max(smax, 1e-12)
    fin_b = np.concatenate([np.where(np.isfinite(loh), loh, 0.0), np.where(np.isfinite(hih), hih, 0.0)])
    nc, nb = np.linalg.norm(ch), np.linalg.norm(fin_b)
    omega_ref = nc / nb if nc > 0 and nb > 0 else 1.0
    omega = omega0 if omega0 is not None else omega_ref
    bnorm = np.linalg.norm(np.concatenate([np.where(np.isfinite(lo), lo, 0.0), np.where(np.isfinite(hi), hi, 0.0)]))
    cnorm = np.linalg.norm(c)

    def rel_err(xs, ys):
        pres, dres, pobj, dobj = kkt(A, A.T, c, l, u, lo, hi, xs * dc, ys * dr)
        return (pres / (1 + bnorm), dres / (1 + cnorm), abs(pobj - dobj) / (1 + abs(pobj) + abs(dobj)), pobj, dobj)

    def pdhg(x, y, omega):
        tau, sigma = eta / omega, eta * omega
        xn = _proj(x - tau * (ch - AhT @ y), lh, uh)
        vv = y - sigma * (Ah @ (2 * xn - x))
        with np.errstate(invalid="ignore"):
            yn = vv + sigma * _proj(-vv / sigma, loh, hih)
        return xn, np.where(np.isfinite(yn), yn, 0.0)

    def fp_res(x, y, xt, yt, omega):
        dx, dy = xt - x, yt - y
        val = omega / eta * (dx @ dx) - 2.0 * (dy @ (Ah @ dx)) + (dy @ dy) / (eta * omega)
        return np.sqrt(max(val, 0.0))

    x0_, y0_ = x.copy(), y.copy()
    k = 0
    it = 0
    k_restart_total = 0
    r0 = None
    r_prev = None
    status = "IterLimit"
    e_prev, e_sum = 0.0, 0.0
    while it < P.max_iter:
        xt, yt = pdhg(x, y, omega)
        check = (it % P.check_every == 0) or k == 0
        if check:
            r = fp_res(x, y, xt, yt, omega)
            if k == 0:
                r0 = r
                r_prev = r
            rp, rd, rg, pobj, dobj = rel_err(xt, yt)
            if P.verbose:
                print("it %6d k %5d r %.3e pres %.2e dres %.2e gap %.2e pobj %.10g omega %.3g" % (it, k, r, rp, rd, rg, pobj, omega))
            if getattr(P, "tol_p", None) is not None:
                axu = A @ (xt * dc)
                pinf = np.max(np.abs(axu - _proj(axu, lo, hi))) if m else 0.0
                done = pinf <= P.tol_p and rg <= P.tol_g and rd <= P.tol_g
            else:
                done = max(rp, rd, rg) <= P.eps
            if done:
                x, y = xt, yt
                status = "Optimal"
                break
            do_restart = k > 0 and (r <= P.beta_suff * r0 or (r <= P.beta_nec * r0 and r > r_prev)
                                    or k >= P.beta_art * (it + 1))
            r_prev = r
            if do_restart:
                dx, dy = np.linalg.norm(xt - x0_), np.linalg.norm(yt - y0_)
                # primal-weight update, guarded: a warm start that is already
                # primal- (or dual-) converged gives dx -> 0 and would send omega to
                # infinity; skip tiny moves and keep omega within 1e3 of omega_ref
                if dx > 1e-8 * (1 + np.linalg.norm(xt)) and dy > 1e-8 * (1 + np.linalg.norm(yt)):
                    omega = np.exp(P.theta * np.log(dy / dx) + (1 - P.theta) * np.log(omega))
                    omega = min(max(omega, omega_ref * 1e-3), omega_ref * 1e3)
                x, y = xt.copy(), yt.copy()
                x0_, y0_ = x.copy(), y.copy()
                k = 0
                it += 1
                continue
        w = (k + 1.0) / (k + 2.0)
        x = w * ((1 + rho) * xt - rho * x) + (1 - w) * x0_
        y = w * ((1 + rho) * yt - rho * y) + (1 - w) * y0_
        k += 1
        it += 1
    rp, rd, rg, pobj, dobj = rel_err(x, y)
    return dict(x=x * dc, y=y * dr, status=status, iters=it, pobj=pobj, dobj=dobj, rel=(rp, rd, rg), omega=omega)
