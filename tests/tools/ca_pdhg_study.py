"""Feasibility study of communication-avoiding PDHG on the row-sharded LP, in numpy on the CPU mirror
of the product's LP method (oracle/pdlp_mirror.py: restarted reflected-Halpern PDHG).  Test-side tooling, not product code.

Row-sharded layout: rank r holds a block of rows A_r with its duals y_r; x is replicated; per iteration
the partial products p_r = A_r'y_r are summed over the ranks (one all-reduce of an n-vector).  "K local steps per exchange":
between two exchanges a rank keeps using the OTHER ranks' partials of the last exchange,

    xt_r = P( x_r - tau (c - p_r(y_r) - S_r) ),    S_r = sum_{s != r} p_s at the last exchange,
    yt_r = dual prox( y_r, A_r (2 xt_r - x_r) ),   Halpern update of (x_r, y_r) towards the common anchor,

so the copies x_r drift apart; at an exchange (every K iterations) the partials are refreshed and x is replaced by the mean of
the copies (one all-reduce of 2 n doubles instead of K all-reduces of n).  Checks -- KKT errors, fixed-point residual, restart
and termination decisions -- are exact PDHG steps on the synchronised point, as in the product.  K = 1 is the product's
iteration.  Printed per LP: iterations and exchanges to the same tolerances for the product's iteration (rho = 1), for the
unreflected one (rho = 0) and for 2 x 2, 8 x 2, 2 x 3, 2 x 4 (ranks x K) at rho = 0 -- under reflection every local-step variant
runs into the iteration limit (solve(..., ranks=2, K=2, rho=1.0) shows it).  Result of the round-4 run: profiles/r04_ca_pdhg_study.txt.

    python tests/tools/ca_pdhg_study.py [first_case last_case]      (cases of instances.lp_battery_case with < 2500 columns)
"""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import katana_jl_amd as ktn                                   # noqa: E402
from oracle.pdlp_mirror import PdlpParams, _proj, kkt, scale_matrix      # noqa: E402


def solve(A, c, l, u, lo, hi, ranks=1, K=1, tol=1e-6, max_iter=200000, check_every=64, rho=1.0):
    P = PdlpParams()
    m, n = A.shape
    A = A.tocsr()
    dr, dc = scale_matrix(A, P.ruiz_iters)
    Ah = (sp.diags(dr) @ A @ sp.diags(dc)).tocsr()
    AhT = Ah.T.tocsr()
    ch, lh, uh, loh, hih = c * dc, l / dc, u / dc, lo * dr, hi * dr
    v = np.ones(n) / np.sqrt(n)
    smax = 1.0
    for _ in range(40):
        v2 = AhT @ (Ah @ v)
        nv = np.linalg.norm(v2)
        smax = np.sqrt(nv / max(np.linalg.norm(v), 1e-300))
        v = v2 / nv
    eta = 0.998 / smax
    fin_b = np.concatenate([np.where(np.isfinite(loh), loh, 0.0), np.where(np.isfinite(hih), hih, 0.0)])
    nc, nb = np.linalg.norm(ch), np.linalg.norm(fin_b)
    omega_ref = nc / nb if nc > 0 and nb > 0 else 1.0
    omega = omega_ref
    bnorm = np.linalg.norm(np.concatenate([np.where(np.isfinite(lo), lo, 0.0), np.where(np.isfinite(hi), hi, 0.0)]))
    cnorm = np.linalg.norm(c)
    # row blocks
    cuts = [(m * r) // ranks for r in range(ranks + 1)]
    Ab = [Ah[cuts[r]:cuts[r + 1]] for r in range(ranks)]
    AbT = [b.T.tocsr() for b in Ab]
    sl = [slice(cuts[r], cuts[r + 1]) for r in range(ranks)]

    def dual_prox(yv, ax, lo_, hi_, sigma):
        vv = yv - sigma * ax
        with np.errstate(invalid="ignore"):
            yn = vv + sigma * _proj(-vv / sigma, lo_, hi_)
        return np.where(np.isfinite(yn), yn, 0.0)

    def pdhg(x, y):                               # the exact step on a synchronised point
        tau, sigma = eta / omega, eta * omega
        xn = _proj(x - tau * (ch - AhT @ y), lh, uh)
        return xn, dual_prox(y, Ah @ (2 * xn - x), loh, hih, sigma)

    x = _proj(np.zeros(n), lh, uh)
    y = np.zeros(m)
    x0_, y0_ = x.copy(), y.copy()
    k = it = 0
    r0 = r_prev = None
    exchanges = 0
    status = "IterLimit"
    while it < max_iter:
        # ---- check iteration (exact): every check_every iterations and right after a restart
        xt, yt = pdhg(x, y)
        exchanges += 1
        dx, dy = xt - x, yt - y
        r = np.sqrt(max(omega / eta * (dx @ dx) - 2.0 * (dy @ (Ah @ dx)) + (dy @ dy) / (eta * omega), 0.0))
        if k == 0:
            r0 = r_prev = r
        pres, dres, pobj, dobj = kkt(A, A.T, c, l, u, lo, hi, xt * dc, yt * dr)
        rp, rd, rg = pres / (1 + bnorm), dres / (1 + cnorm), abs(pobj - dobj) / (1 + abs(pobj) + abs(dobj))
        if not np.isfinite(r) or r > 1e30:
            status = "Diverged"
            break
        if max(rp, rd, rg) <= tol:
            status = "Optimal"
            it += 1
            break
        restart = k > 0 and (r <= 0.2 * r0 or (r <= 0.8 * r0 and r > r_prev) or k >= 0.36 * (it + 1))
        r_prev = r
        if restart:
            ddx, ddy = np.linalg.norm(xt - x0_), np.linalg.norm(yt - y0_)
            if ddx > 1e-8 * (1 + np.linalg.norm(xt)) and ddy > 1e-8 * (1 + np.linalg.norm(yt)):
                omega = np.exp(0.5 * np.log(ddy / ddx) + 0.5 * np.log(omega))
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
        # ---- a chunk of plain iterations up to the next check
        chunk = min(31 if k <= 1 else check_every - 1, max_iter - it)
        tau, sigma = eta / omega, eta * omega
        if K == 1 or ranks == 1:
            for _ in range(chunk):
                xt, yt = pdhg(x, y)
                w = (k + 1.0) / (k + 2.0)
                x = w * ((1 + rho) * xt - rho * x) + (1 - w) * x0_
                y = w * ((1 + rho) * yt - rho * y) + (1 - w) * y0_
                k += 1
            it += chunk
            exchanges += chunk
            continue
        xr = [x.copy() for _ in range(ranks)]
        done_in_chunk = 0
        while done_in_chunk < chunk:
            # exchange: fresh partials, x <- mean of the copies
            xm = sum(xr) / ranks
            p = [AbT[r] @ y[sl[r]] for r in range(ranks)]
            tot = sum(p)
            exchanges += 1
            xr = [xm.copy() for _ in range(ranks)]
            steps = min(K, chunk - done_in_chunk)
            for _ in range(steps):
                w = (k + 1.0) / (k + 2.0)
                for r in range(ranks):
                    pr = AbT[r] @ y[sl[r]]
                    aty = pr + (tot - p[r])                              # own partial fresh, the others' from the exchange
                    xtr = _proj(xr[r] - tau * (ch - aty), lh, uh)
                    ytr = dual_prox(y[sl[r]], Ab[r] @ (2 * xtr - xr[r]), loh[sl[r]], hih[sl[r]], sigma)
                    xr[r] = w * ((1 + rho) * xtr - rho * xr[r]) + (1 - w) * x0_
                    y[sl[r]] = w * ((1 + rho) * ytr - rho * y[sl[r]]) + (1 - w) * y0_[sl[r]]
                k += 1
            done_in_chunk += steps
        x = sum(xr) / ranks
        it += chunk
    return dict(status=status, iters=it, exchanges=exchanges, pobj=pobj)


def main():
    a = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    b = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    print("iterations / exchanges to 1e-6 (KKT, relative).  product = reflected Halpern (rho = 1), every iteration exact; the local-step\n"
          "variants diverge under reflection (rho = 1 and 0.5: iteration limit in every case tried), so they run UNREFLECTED (rho = 0)", flush=True)
    print("%-5s %-6s %-6s %-5s | %-12s | %-12s | %s" % ("case", "n", "m", "kind", "product", "rho=0, K=1", "ranks x K at rho = 0: iterations (ratio to product) / exchanges (ratio to product)"), flush=True)
    for i in range(a, b):
        kw = ktn.instances.lp_battery_case(i)
        if kw["n"] >= 2500:
            continue
        inst = ktn.instances.make_lp(**kw)
        A = sp.csr_matrix((inst.p0, inst.col, inst.rowptr), shape=(inst.num_constr, inst.n))
        c = np.zeros(inst.n); c[inst.obj_col] = inst.obj_p0
        args = (A, c, inst.l_var, inst.u_var, inst.l_constr, inst.u_constr)
        t0 = time.time()
        base = solve(*args)
        b0 = solve(*args, rho=0.0)
        line = "%-5d %-6d %-6d %-5d | %-7s %4d | %-7s %4d | " % (i, inst.n, inst.num_constr, i % 5, base["status"], base["iters"], b0["status"], b0["iters"])
        cap = max(12 * base["iters"], 6000)
        for ranks, K in ((2, 2), (8, 2), (2, 3), (2, 4)):
            r = solve(*args, ranks=ranks, K=K, max_iter=cap, rho=0.0)
            if r["status"] == "Optimal":
                line += "%dx%d: %d (%.2fx) / %d (%.2fx)   " % (ranks, K, r["iters"], r["iters"] / base["iters"], r["exchanges"], r["exchanges"] / base["exchanges"])
            else:
                line += "%dx%d: %s after %d   " % (ranks, K, r["status"], r["iters"])
        print(line + "[%.0f s]" % (time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
