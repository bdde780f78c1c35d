// dense_lp.hpp -- exact LP kernel for SMALL cutting-plane LPs (a few dozen columns).
//
// On an optimum that lies on a curved face Kelley's method ends with several nearly parallel cuts of
// one row active at once; a first-order LP method then needs millions of iterations (DESIGN.md
// section 5, "small smooth problems").  For LPs with at most kDenseMaxN columns -- the size of the
// reference's own test models -- this kernel solves the LP exactly, the way the reference's simplex code
// does: a dual active-set (dual simplex in constraint form) method in ONE 256-thread workgroup, with
// the n x n basis inverse in LDS.
//
//   min s*c'x   s.t.  every "side" k:  g_k'x <= h_k
//   sides:  row i upper (a_i'x <= hi_i, id 2i), row i lower (-a_i'x <= -lo_i, id 2i+1), and with NEGATIVE ids
//           (-1 - (2j + lower)), so that they stay valid when rows are appended, var j upper / lower
//           (an infinite variable bound becomes an ARTIFICIAL side at 0 so that a dual-feasible start always exists)
//   working set W (n sides, linearly independent):  x = B^-1 h_W,  lambda = -B^-T c >= 0
//   pivot: most violated side q enters; u = B^-T g_q; ratio test theta = min_{u_r>0} lambda_r/u_r
//          (Bland tie-break) picks the side that leaves; dual objective rises monotonically.
// W persists across the ECP iterations (rows are only appended), so a re-solve after new cuts is a
// few pivots -- the warm-started dual simplex of the reference's live GLPK model.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace ktn {

constexpr int kDenseMaxN = 32;
constexpr double kDenseBig = 1e7;

struct DenseLpIO {
    int n;
    int64_t m;
    const int64_t* rowptr; const int32_t* col; const double* val;
    const double* lo; const double* hi; const double* l; const double* u; const double* c;
    double sgn;
    double* dense;       // m x n scratch (densified rows)
    int32_t* W;          // [n] working set (side ids), in/out
    int32_t* Wvalid;     // [1] in: 1 when W holds a dual-feasible working set of an earlier solve
    double* x;           // [n] out
    double* y;           // [m] out: row multipliers in the engine's sign convention (>0 lower side, <0 upper side)
    double* out;         // [4] status (0 optimal, 1 infeasible, 2 failed/limit, 3 artificial bound active), pivots, objective
    int max_pivots;
    double tol;
};

// normal and right-hand side of side k (dense normal into g[0..n))
__device__ __forceinline__ double dense_side(const DenseLpIO& P, int64_t k, double* g, const double* art) {
    const int n = P.n;
    if (k >= 0) {
        const int64_t i = k >> 1;
        const double sg = (k & 1) ? -1.0 : 1.0;
        for (int q = 0; q < n; ++q) g[q] = sg * P.dense[i * n + q];
        double b = (k & 1) ? -P.lo[i] : P.hi[i];
        if (b != b) b = __builtin_inf();
        return b;
    }
    const int64_t j = (-1 - k) >> 1;
    const bool lower = (-1 - k) & 1;
    for (int q = 0; q < n; ++q) g[q] = 0.0;
    g[j] = lower ? -1.0 : 1.0;
    double b = lower ? -P.l[j] : P.u[j];
    if (!(b < kDenseBig)) b = art[-1 - k];        // infinite (or NaN) bound -> artificial side, starts at 0
    return b;
}
__device__ __forceinline__ double dense_side_rhs(const DenseLpIO& P, int64_t k) {
    if (k >= 0) {
        const int64_t i = k >> 1;
        double b = (k & 1) ? -P.lo[i] : P.hi[i];
        if (b != b) b = __builtin_inf();
        return b;
    }
    const int64_t j = (-1 - k) >> 1;
    const bool lower = (-1 - k) & 1;
    double b = lower ? -P.l[j] : P.u[j];
    if (!(b < kDenseBig)) b = __builtin_inf();    // artificial sides only ever LEAVE the working set
    return b;
}
__device__ __forceinline__ double dense_side_dot(const DenseLpIO& P, int64_t k, const double* x, double* nrm) {
    const int n = P.n;
    if (k >= 0) {
        const int64_t i = k >> 1;
        double acc = 0.0, n2 = 0.0;
        for (int q = 0; q < n; ++q) { const double a = P.dense[i * n + q]; acc += a * x[q]; n2 += a * a; }
        *nrm = sqrt(n2);
        return (k & 1) ? -acc : acc;
    }
    const int64_t j = (-1 - k) >> 1;
    *nrm = 1.0;
    return ((-1 - k) & 1) ? -x[j] : x[j];
}

static __global__ __launch_bounds__(256) void k_dense_lp(DenseLpIO P) {
    constexpr int N = kDenseMaxN;
    __shared__ double Aug[N][2 * N];     // [B | I] -> [I | B^-1]
    __shared__ double hW[N], xs[N], lam[N], uvec[N], gq[N], cs[N];
    __shared__ int Ws[N];
    __shared__ double red_val[256];
    __shared__ long long red_idx[256];
    __shared__ int s_flag, s_p;
    __shared__ double art[2 * N];        // rhs of the artificial sides of infinite variable bounds
    const int n = P.n, t = threadIdx.x;
    const int64_t m = P.m, K = 2 * m + 2 * (int64_t)n;      // sides: ids -2n .. 2m-1

    // densify rows
    for (int64_t i = t; i < m; i += 256) {
        double* d = P.dense + i * n;
        for (int q = 0; q < n; ++q) d[q] = 0.0;
        for (int64_t e = P.rowptr[i]; e < P.rowptr[i + 1]; ++e) d[P.col[e]] += P.val[e];
    }
    // A variable without a finite bound in the direction its cost pushes it starts on an ARTIFICIAL side at 0 --
    // where a simplex code keeps a non-basic free column.  Artificial sides never enter the working set; one
    // that is still in it at the optimum with a zero multiplier is harmless (the point is optimal for the real
    // LP), one with a positive multiplier is relaxed (below) and the iteration continues.
    for (int v = t; v < 2 * n; v += 256) art[v] = 0.0;
    if (t < n) {
        cs[t] = P.sgn * P.c[t];
        // start: previous working set, else the bound side the cost pushes each variable to
        Ws[t] = P.Wvalid[0] ? P.W[t] : -1 - (2 * t + (cs[t] >= 0.0 ? 1 : 0));
    }
    __syncthreads();

    int status = 2, pivots = 0;
    for (; pivots <= P.max_pivots; ++pivots) {
        // ---- B (rows = normals of W) and its inverse by Gauss-Jordan with partial pivoting
        if (t < n) {
            double g[N];
            hW[t] = dense_side(P, Ws[t], g, art);
            for (int q = 0; q < n; ++q) { Aug[t][q] = g[q]; Aug[t][n + q] = (q == t) ? 1.0 : 0.0; }
        }
        __syncthreads();
        if (t == 0) s_flag = 0;
        __syncthreads();
        for (int col = 0; col < n; ++col) {
            if (t == 0) {
                int piv = col; double best = fabs(Aug[col][col]);
                for (int r = col + 1; r < n; ++r) if (fabs(Aug[r][col]) > best) { best = fabs(Aug[r][col]); piv = r; }
                if (!(best > 1e-13)) s_flag = 1;
                s_p = piv;
            }
            __syncthreads();
            if (s_flag) break;
            const int piv = s_p;
            if (piv != col) {
                // NOTE: swapping rows of [B | I] permutes the equations, not the unknowns: B^-1 stays B^-1
                for (int q = t; q < 2 * n; q += 256) { const double tmp = Aug[col][q]; Aug[col][q] = Aug[piv][q]; Aug[piv][q] = tmp; }
            }
            __syncthreads();
            const double d = Aug[col][col];
            __syncthreads();
            for (int q = t; q < 2 * n; q += 256) Aug[col][q] /= d;
            __syncthreads();
            for (int rq = t; rq < n * 2 * n; rq += 256) {
                const int r = rq / (2 * n), q = rq % (2 * n);
                if (r != col && q != col) Aug[r][q] -= Aug[r][col] * Aug[col][q];
            }
            __syncthreads();
            for (int r = t; r < n; r += 256) if (r != col) Aug[r][col] = 0.0;
            __syncthreads();
        }
        if (s_flag) { status = 2; break; }
        // Row swaps were applied to the equations [B | I]; the right block now holds B^-1 with its COLUMNS in the
        // original row order of B only if no swap happened.  Track nothing: recompute through the identity
        // B^-1 = right block (Gauss-Jordan with row interchanges on the augmented matrix yields exactly B^-1).
        // x = B^-1 hW ; lambda = -B^-T c
        if (t < n) {
            double acc = 0.0;
            for (int q = 0; q < n; ++q) acc += Aug[t][n + q] * hW[q];
            xs[t] = acc;
            double l2 = 0.0;
            for (int q = 0; q < n; ++q) l2 += Aug[q][n + t] * cs[q];
            lam[t] = -l2;
        }
        __syncthreads();
        // ---- most violated side (normalised), Bland: smallest id among (near) ties is not needed for entering
        const long long kNone = 0x7fffffffffffffffLL;
        double bestv = 0.0; long long besti = kNone;
        for (int64_t kk = t; kk < K; kk += 256) {
            const int64_t k = kk - 2 * (int64_t)n;
            const double h = dense_side_rhs(P, k);
            if (!(h < __builtin_inf())) continue;
            double nrm;
            const double act = dense_side_dot(P, k, xs, &nrm);
            const double v = (act - h) / fmax(nrm, 1e-300);
            if (v > bestv) { bestv = v; besti = k; }
        }
        red_val[t] = bestv; red_idx[t] = besti;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (t < s) {
                if (red_val[t + s] > red_val[t] || (red_val[t + s] == red_val[t] && red_idx[t + s] < red_idx[t])) {
                    red_val[t] = red_val[t + s]; red_idx[t] = red_idx[t + s];
                }
            }
            __syncthreads();
        }
        const double viol = red_val[0];
        const long long q_in = red_idx[0];
        double scale = 1.0;
        for (int q = 0; q < n; ++q) scale = fmax(scale, fabs(xs[q]));
        if (!(viol > P.tol * scale) || q_in == kNone) {
            // primal feasible: done unless an artificial side still carries a multiplier.  Such a side is RELAXED:
            // x moves along d = B^-1 e_r (the objective falls at rate lambda_r) past the first real side that blocks,
            // which then enters by the ordinary dual pivot.  No blocking side: the LP is unbounded (status 3).
            if (t == 0) {
                double cmax = 1.0, lbest = 0.0;
                int rbest = -1;
                for (int q = 0; q < n; ++q) cmax = fmax(cmax, fabs(cs[q]));
                for (int r = 0; r < n; ++r) {
                    const int k = Ws[r];
                    if (k >= 0) continue;
                    const int j = (-1 - k) >> 1;
                    const bool lower = (-1 - k) & 1;
                    const double b = lower ? -P.l[j] : P.u[j];
                    if (b < kDenseBig) continue;
                    if (lam[r] > 1e-9 * cmax && lam[r] > lbest) { lbest = lam[r]; rbest = r; }
                }
                s_p = rbest;
            }
            __syncthreads();
            const int r_rel = s_p;
            if (r_rel < 0) { status = 0; break; }
            if (t < n) uvec[t] = Aug[t][n + r_rel];         // d
            __syncthreads();
            double best_s = __builtin_inf();
            for (int64_t kk = t; kk < K; kk += 256) {
                const int64_t k = kk - 2 * (int64_t)n;
                const double h = dense_side_rhs(P, k);
                if (!(h < __builtin_inf())) continue;
                double nrm;
                const double ad = dense_side_dot(P, k, uvec, &nrm);
                if (!(ad > 1e-12 * fmax(nrm, 1e-300))) continue;
                const double ax = dense_side_dot(P, k, xs, &nrm);
                best_s = fmin(best_s, fmax(h - ax, 0.0) / ad);
            }
            red_val[t] = best_s;
            __syncthreads();
            for (int s = 128; s > 0; s >>= 1) {
                if (t < s) red_val[t] = fmin(red_val[t], red_val[t + s]);
                __syncthreads();
            }
            const double step = red_val[0];
            __syncthreads();
            if (!(step < __builtin_inf())) { status = 3; break; }
            if (t == 0) art[-1 - Ws[r_rel]] += 2.0 * step + 1.0;
            __syncthreads();
            continue;
        }
        // ---- u = B^-T g_q ; ratio test
        if (t < n) {
            double g[N];
            (void)dense_side(P, q_in, g, art);
            gq[t] = g[t];
        }
        __syncthreads();
        if (t < n) {
            double acc = 0.0;
            for (int q = 0; q < n; ++q) acc += Aug[q][n + t] * gq[q];
            uvec[t] = acc;
        }
        __syncthreads();
        if (t == 0) {
            int p = -1; double theta = __builtin_inf();
            for (int r = 0; r < n; ++r) {
                if (uvec[r] > 1e-11) {
                    const double ratio = fmax(lam[r], 0.0) / uvec[r];
                    if (ratio < theta - 1e-14 || (ratio <= theta + 1e-14 && p >= 0 && Ws[r] < Ws[p])) { theta = ratio; p = r; }
                }
            }
            s_p = p;
        }
        __syncthreads();
        if (s_p < 0) {
            // No side can leave: the dual ray -u proves infeasibility -- of the LP INSIDE its artificial sides.  The stand-in bound
            // of a free variable starts at 0 and is no constraint of the LP, so a ray that puts weight on one (u_r < 0) proves
            // nothing yet (fuzz models 55/106, 144/51: feasible LPs with free variables reported infeasible).  Those sides are
            // pushed outwards -- far enough to satisfy the entering side, at least tenfold, at most to kDenseBig -- and the
            // iteration goes on with the same working set and multipliers.  A ray that survives bounds of 1e7 is taken as the
            // certificate (fuzz models 2/62, 2/138: infeasible, and said so by the oracle's simplex).
            if (t == 0) {
                double nrm;
                const double excess = fmax(dense_side_dot(P, q_in, xs, &nrm) - dense_side_rhs(P, q_in), 0.0);
                int grown = 0;
                for (int r = 0; r < n; ++r) {
                    const int k = Ws[r];
                    if (k >= 0 || !(uvec[r] < -1e-11)) continue;
                    const int j = (-1 - k) >> 1;
                    const double b = ((-1 - k) & 1) ? -P.l[j] : P.u[j];
                    if (b < kDenseBig) continue;               // a real bound
                    double& a = art[-1 - k];
                    if (a >= kDenseBig) continue;
                    a = fmin(fmax(fmax(10.0 * a, 1.0), a + 2.0 * excess / (-uvec[r]) + 1.0), kDenseBig);
                    grown = 1;
                }
                s_flag = grown;
            }
            __syncthreads();
            if (!s_flag) { status = 1; break; }        // a certificate on real sides (or on artificial ones at 1e7): infeasible
            __syncthreads();
            continue;
        }
        if (t == 0) Ws[s_p] = (int)q_in;
        __syncthreads();
    }
    if (pivots > P.max_pivots && status == 2) status = 2;
    // ---- outputs
    if (status == 0) {
        for (int64_t i = t; i < m; i += 256) P.y[i] = 0.0;
        __syncthreads();
        if (t < n) {
            P.x[t] = xs[t];
            P.W[t] = Ws[t];
            const int k = Ws[t];
            if (k >= 0) P.y[k >> 1] = (k & 1) ? fmax(lam[t], 0.0) : -fmax(lam[t], 0.0);
        }
        if (t == 0) {
            double obj = 0.0;
            for (int q = 0; q < n; ++q) obj += cs[q] * xs[q];
            P.out[0] = 0.0;
            P.out[1] = (double)pivots;
            P.out[2] = obj;
            P.Wvalid[0] = 1;
        }
    } else if (t == 0) {
        P.out[0] = (double)status;
        P.out[1] = (double)pivots;
        P.out[2] = 0.0;
        P.Wvalid[0] = 0;
    }
}

}  // namespace ktn
