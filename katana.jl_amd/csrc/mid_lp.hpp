// mid_lp.hpp -- exact LP solves for MID-SIZE cutting-plane LPs (33 ... kMidMaxN columns).
//
// Why it exists (round 4, DESIGN.md section 5 "Smooth-face optima").  Off the non-degenerate-vertex family -- optimum on a
// curved face, the regime of the reference's own claim (README.md:5) and of test/misc.jl:4-57 -- Kelley's method
// (src/model.jl:257-309) needs hundreds to thousands of iterations, and every LP of the sequence has many nearly parallel
// cuts of one nonlinear row active at once.  A first-order LP method converges on such LPs at a rate set by that
// conditioning: 1 000 - 20 000 PDHG iterations per re-solve on a 200-column LP, where the reference's warm-started dual
// simplex (GLPK behind `solve(m.linear_model)`, src/model.jl:259) needs a handful of pivots.  dense_lp.hpp gives LPs of at
// most 32 columns an exact solve inside one workgroup; this file does the same for LPs up to kMidMaxN columns with the
// basis inverse in HBM / L2:
//
//   min s*c'x   s.t.  every side k:  g_k'x <= h_k          (side ids as in dense_lp.hpp: row i upper 2i, lower 2i+1,
//                                                            variable j upper / lower -1 - (2j + lower))
//   working set W (n linearly independent sides):  B x = h_W,  B'lambda = -c,  lambda >= 0   (dual feasible throughout)
//   pivot:  the most violated side q (normalised) enters;  u = B^-T g_q  (only the rows of B^-1 in the support of g_q are
//           read: O(n k));  ratio test theta = min_{u_r > 0} lambda_r / u_r picks the side p that leaves;
//           x <- x - B^-1 e_p (g_q'x - h_q) / u_p,  lambda <- lambda - theta u,  lambda_p = theta,
//           B^-1 <- B^-1 - (B^-1 e_p)(u - e_p)' / u_p                                   (rank one, O(n^2), one launch)
// B^-1 is n x n doubles, row-major (8 MB at n = 1 000: L2 / Infinity-Cache resident), and PERSISTS across the ECP iterations
// together with W, x and lambda: rows are only appended, so a re-solve after a sweep is a few pivots per new cut -- the
// reference's live GLPK model.  One pivot is five launches (price -> select -> u -> ratio -> rank-one update) whose control
// flow lives in a device-resident state block: the host enqueues batches of pivots without synchronising and reads the
// state back once per batch; kernels of a batch that come after the end of the solve return at once.  Before a solve is
// declared optimal x is recomputed from its definition (B^-1 h_W plus one step of iterative refinement against the sparse
// rows themselves) and lambda as -B^-T c, so the accumulated error of the rank-one updates never reaches the answer; a
// residual that does not come down makes the solve restart cold, from the bound vertex.
//
// Infinite variable bounds get artificial sides at 0 like in dense_lp.hpp, but here they are only allowed to LEAVE: one that
// is still needed at the end (or blocks an infeasibility proof) makes the solve fail and the first-order method takes over.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "common.hpp"
#include "dense_lp.hpp"

namespace ktn {

constexpr int kMidMaxN = 4096;
constexpr int kMidPriceBlocks = 128;

struct MidState {            // device-resident control block of a solve
    int32_t status;          // 0 running, 1 primal feasible, 3 infeasible, 4 failed
    int32_t pivots;
    int32_t q;               // entering side
    int32_t p;               // leaving position
    int32_t degen_run;       // consecutive pivots with theta = 0
    int32_t art_left;        // (final) artificial sides still carrying a multiplier
    int32_t gj_p;            // refactorisation: pivot row of the current column
    int32_t gj_singular;     // ... a pivot below 1e-13 of the column's scale: the working set is (numerically) dependent
    double viol, excess, hq, up, theta, scale, obj, resid, gj_piv;
};

struct MidLpIO {
    int n;
    int64_t m;
    const int64_t* rowptr; const int32_t* col; const double* val;
    const double* lo; const double* hi; const double* l; const double* u; const double* c;
    double sgn;
    double* Binv;            // n x n, row-major: x = Binv h_W
    int32_t* W;              // [n] side ids
    double* hW;              // [n] right-hand sides of the working sides
    double* x; double* lam; double* uvec; double* dvec; double* rvec;
    double* ctil;            // [n] s * c with the anti-degeneracy perturbation (k_mid_init): the cost the pivoting works with
    double* part_val; int32_t* part_idx;
    MidState* st;
    double tol;
    int max_pivots;
};

__device__ __forceinline__ double mid_bound_rhs(const MidLpIO& P, int k) {       // variable side: +-bound, inf when absent
    const int j = (-1 - k) >> 1;
    const bool lower = (-1 - k) & 1;
    const double b = lower ? -P.l[j] : P.u[j];
    return (b < kDenseBig) ? b : __builtin_inf();
}
__device__ __forceinline__ double mid_row_rhs(const MidLpIO& P, int k) {
    const int64_t i = k >> 1;
    double b = (k & 1) ? -P.lo[i] : P.hi[i];
    if (b != b) b = __builtin_inf();                 // NaN row bound: the side is vacuous
    return b;
}

// bound vertex the cost pushes to: B = diag(+-1) is its own inverse
static __global__ __launch_bounds__(256) void k_mid_init(MidLpIO P) {
    const int n = P.n;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx < (int64_t)n * n) {
        const int j = (int)(idx / n), r = (int)(idx % n);
        double v = 0.0;
        if (j == r) {
            const double cs = P.sgn * P.c[j];
            bool lower = cs >= 0.0;
            const bool lo_ok = -P.l[j] < kDenseBig, up_ok = P.u[j] < kDenseBig;
            if (cs == 0.0 && !lo_ok && up_ok) lower = false;
            v = lower ? -1.0 : 1.0;
            const double b = lower ? (lo_ok ? -P.l[j] : 0.0) : (up_ok ? P.u[j] : 0.0);     // (absent bound: artificial side at 0)
            P.W[j] = -1 - (2 * j + (lower ? 1 : 0));
            P.hW[j] = b;
            P.x[j] = lower ? -b : b;
            // Cost perturbation.  Cutting-plane LPs are massively dual degenerate (every variable that no active row contains
            // has reduced cost 0), and a dual method then stalls or cycles in zero-length steps (seen: period-42 cycle on a
            // 200-column LP; numpy mirror of this file).  Every cost is pushed AWAY from zero by 1e-9 (1 + |c_j|) x U(0.5, 1)
            // (hashed, deterministic) in the direction that keeps the starting vertex dual feasible; the perturbation stays
            // for the life of the working set.  The vertex returned is optimal for the perturbed cost: among the optimal
            // vertices of a degenerate LP it is one of them, and otherwise its objective (reported with the TRUE cost) is
            // off by at most sum_j 1e-9 (1 + |c_j|) |dx_j| -- orders below the 1e-6 the reference's tests ask for.
            uint64_t hsh = (uint64_t)j * 0x9E3779B97F4A7C15ULL + 0xD1B54A32D192ED03ULL;
            hsh ^= hsh >> 29; hsh *= 0xBF58476D1CE4E5B9ULL; hsh ^= hsh >> 32;
            const double xi = 1e-9 * (1.0 + fabs(cs)) * (0.5 + 0.5 * (double)(hsh >> 11) * (1.0 / 9007199254740992.0));
            const double ct = lower ? cs + xi : cs - xi;
            P.ctil[j] = ct;
            P.lam[j] = fabs(ct);
        }
        P.Binv[idx] = v;
    }
    if (idx == 0) {
        P.st->status = 0; P.st->pivots = 0; P.st->degen_run = 0; P.st->art_left = 0; P.st->resid = 0.0; P.st->p = -1; P.st->gj_singular = 0; P.st->gj_p = 0;
    }
}

// pricing: most violated side, normalised by the side's norm.  Rows: 8 lanes per row, coalesced entry loads, xor-butterfly (one
// thread per row walked 16 - 32 entries in as many dependent round trips: a pivot took 36 us, with this 29); variables: one
// thread each.  (Measured and NOT kept, round 4: the whole pivot in one 1 024-thread workgroup -- 35 us, pricing 2.7 MB through one
// CU's L2 port; pricing + one fused single-workgroup launch for the rest -- 42 us with the rank-one update inside it (a 50-trip
// dependent loop per thread), 26 - 29 us with the rank-one update as a third launch: no better than these five small launches.)
static __global__ __launch_bounds__(256) void k_mid_price(MidLpIO P) {
    __shared__ double sv[256];
    __shared__ int si[256];
    const int t = threadIdx.x;
    double bestv = 0.0;
    int besti = 0x7fffffff;
    if (P.st->status == 0) {
        const int lane = t & 7;
        const int64_t g0 = ((int64_t)blockIdx.x * 256 + t) >> 3, ng = ((int64_t)gridDim.x * 256) >> 3;
        for (int64_t it = g0; it < P.m; it += ng) {
            double act = 0.0, n2 = 0.0;
            for (int64_t e = P.rowptr[it] + lane; e < P.rowptr[it + 1]; e += 8) { const double a = P.val[e]; act += a * P.x[P.col[e]]; n2 += a * a; }
            act = group_sum<8>(act); n2 = group_sum<8>(n2);
            if (lane == 0) {
                const double nrm = fmax(sqrt(n2), 1e-300);
                const double hi = P.hi[it], lo = P.lo[it];
                const double vu = (hi == hi && hi < __builtin_inf()) ? (act - hi) / nrm : -1.0;
                const double vl = (lo == lo && lo > -__builtin_inf()) ? (lo - act) / nrm : -1.0;
                const int k = (vl > vu) ? (int)(2 * it + 1) : (int)(2 * it);
                const double v = fmax(vu, vl);
                if (v > bestv || (v == bestv && v > 0.0 && k < besti)) { bestv = v; besti = k; }
            }
        }
        for (int64_t j = (int64_t)blockIdx.x * 256 + t; j < P.n; j += (int64_t)gridDim.x * 256) {
            const double xj = P.x[j];
            const double vu = (P.u[j] < kDenseBig) ? xj - P.u[j] : -1.0;
            const double vl = (-P.l[j] < kDenseBig) ? P.l[j] - xj : -1.0;
            const int k = -1 - (2 * (int)j + ((vl > vu) ? 1 : 0));
            const double v = fmax(vu, vl);
            if (v > bestv || (v == bestv && v > 0.0 && k < besti)) { bestv = v; besti = k; }
        }
    }
    sv[t] = bestv; si[t] = besti;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (t < s && (sv[t + s] > sv[t] || (sv[t + s] == sv[t] && si[t + s] < si[t]))) { sv[t] = sv[t + s]; si[t] = si[t + s]; }
        __syncthreads();
    }
    if (t == 0) { P.part_val[blockIdx.x] = sv[0]; P.part_idx[blockIdx.x] = si[0]; }
    if (t == 0 && blockIdx.x == 0) P.st->p = -1;             // (nobody reads `p` in this launch; k_mid_ratio sets it when it pivots)
}

// final argmax, the stop test, and the entering side's excess g_q'x - h_q
static __global__ __launch_bounds__(256) void k_mid_select(MidLpIO P) {
    __shared__ double sv[256];
    __shared__ int si[256];
    MidState* st = P.st;
    if (st->status != 0) return;
    const int t = threadIdx.x;
    sv[t] = (t < kMidPriceBlocks) ? P.part_val[t] : 0.0;
    si[t] = (t < kMidPriceBlocks) ? P.part_idx[t] : 0x7fffffff;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (t < s && (sv[t + s] > sv[t] || (sv[t + s] == sv[t] && si[t + s] < si[t]))) { sv[t] = sv[t + s]; si[t] = si[t + s]; }
        __syncthreads();
    }
    const double viol = sv[0];
    const int q = si[0];
    __syncthreads();
    double mx = 1.0;
    for (int j = t; j < P.n; j += 256) mx = fmax(mx, fabs(P.x[j]));
    sv[t] = mx;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (t < s) sv[t] = fmax(sv[t], sv[t + s]); __syncthreads(); }
    const double scale = sv[0];
    __syncthreads();
    if (!(viol > P.tol * scale) || q == 0x7fffffff) {
        if (t == 0) { st->status = 1; st->viol = viol; st->scale = scale; }
        return;
    }
    double acc = 0.0, h;
    if (q >= 0) {
        const int64_t i = q >> 1;
        const double sg = (q & 1) ? -1.0 : 1.0;
        for (int64_t e = P.rowptr[i] + t; e < P.rowptr[i + 1]; e += 256) acc += sg * P.val[e] * P.x[P.col[e]];
        h = mid_row_rhs(P, q);
    } else {
        const int j = (-1 - q) >> 1;
        if (t == 0) acc = ((-1 - q) & 1) ? -P.x[j] : P.x[j];
        h = mid_bound_rhs(P, q);
    }
    sv[t] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (t < s) sv[t] += sv[t + s]; __syncthreads(); }
    if (t == 0) { st->q = q; st->hq = h; st->excess = sv[0] - h; st->viol = viol; st->scale = scale; }
}

// u = B^-T g_q: only the rows of B^-1 in the support of g_q
static __global__ __launch_bounds__(256) void k_mid_u(MidLpIO P) {
    if (P.st->status != 0) return;
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= P.n) return;
    const int q = P.st->q, n = P.n;
    double acc = 0.0;
    if (q >= 0) {
        const int64_t i = q >> 1;
        const double sg = (q & 1) ? -1.0 : 1.0;
        for (int64_t e = P.rowptr[i]; e < P.rowptr[i + 1]; ++e) acc += sg * P.val[e] * P.Binv[(int64_t)P.col[e] * n + r];
    } else {
        const int j = (-1 - q) >> 1;
        acc = (((-1 - q) & 1) ? -1.0 : 1.0) * P.Binv[(int64_t)j * n + r];
    }
    P.uvec[r] = acc;
}

// ratio test, then the updates of x, lambda, W (the rank-one update of B^-1 follows in k_mid_rank1)
static __global__ __launch_bounds__(256) void k_mid_ratio(MidLpIO P) {
    __shared__ double sv[256];
    __shared__ int si[256];
    __shared__ int s_art;
    MidState* st = P.st;
    if (st->status != 0) return;
    const int t = threadIdx.x, n = P.n;
    if (t == 0) s_art = 0;
    double um = 0.0;
    for (int r = t; r < n; r += 256) um = fmax(um, fabs(P.uvec[r]));
    sv[t] = um;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (t < s) sv[t] = fmax(sv[t], sv[t + s]); __syncthreads(); }
    const double ptol = fmax(1e-11, 1e-9 * sv[0]);
    __syncthreads();
    // phase 1: the smallest ratio
    double th = __builtin_inf();
    for (int r = t; r < n; r += 256) {
        const double ur = P.uvec[r];
        if (ur > ptol) th = fmin(th, fmax(P.lam[r], 0.0) / ur);
    }
    sv[t] = th;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (t < s) sv[t] = fmin(sv[t], sv[t + s]); __syncthreads(); }
    const double theta0 = sv[0];
    __syncthreads();
    if (!(theta0 < __builtin_inf())) {
        // no side can leave: a dual ray.  With weight on an artificial side it proves nothing (dense_lp.hpp) -> give up
        for (int r = t; r < n; r += 256) {
            const int k = P.W[r];
            if (k < 0 && P.uvec[r] < -ptol && !(mid_bound_rhs(P, k) < __builtin_inf())) s_art = 1;
        }
        __syncthreads();
        if (t == 0) st->status = s_art ? 4 : 3;
        return;
    }
    // phase 2: among the (near) ties the largest pivot element; after a run of degenerate pivots the smallest side id (Bland)
    const bool bland = st->degen_run > 40;
    const double cut = theta0 + 1e-12 * (1.0 + theta0);
    double bu = -1.0;
    int bi = -1;
    for (int r = t; r < n; r += 256) {
        const double ur = P.uvec[r];
        if (ur > ptol && fmax(P.lam[r], 0.0) / ur <= cut) {
            const double key = bland ? -(double)P.W[r] : ur;
            if (bi < 0 || key > bu) { bu = key; bi = r; }
        }
    }
    sv[t] = bu; si[t] = bi;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (t < s && si[t + s] >= 0 && (si[t] < 0 || sv[t + s] > sv[t] || (sv[t + s] == sv[t] && si[t + s] < si[t]))) { sv[t] = sv[t + s]; si[t] = si[t + s]; }
        __syncthreads();
    }
    const int p = si[0];
    const double up = P.uvec[p];
    const double theta = fmax(P.lam[p], 0.0) / up;
    const double delta = st->excess / up;
    __syncthreads();
    for (int j = t; j < n; j += 256) {
        const double d = P.Binv[(int64_t)j * n + p];
        P.dvec[j] = d;
        P.x[j] -= delta * d;
    }
    for (int r = t; r < n; r += 256) P.lam[r] = (r == p) ? theta : fmax(P.lam[r] - theta * P.uvec[r], 0.0);
    if (t == 0) {
        P.W[p] = st->q;
        P.hW[p] = st->hq;
        st->p = p; st->up = up; st->theta = theta;
        st->degen_run = (theta <= 1e-13) ? st->degen_run + 1 : 0;
        st->pivots += 1;
        if (st->pivots >= P.max_pivots) st->status = 4;      // (k_mid_rank1 of this pivot still runs: it tests `p`)
    }
}

// B^-1 <- B^-1 - d (u - e_p)' / u_p
static __global__ __launch_bounds__(256) void k_mid_rank1(MidLpIO P) {
    const MidState* st = P.st;
    // runs exactly when k_mid_ratio of this pivot pivoted: k_mid_price cleared `p`, only a pivot sets it
    if (st->p < 0) return;
    const int n = P.n;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)n * n) return;
    const int j = (int)(idx / n), r = (int)(idx % n);
    const int p = st->p;
    const double f = (P.uvec[r] - (r == p ? 1.0 : 0.0)) / st->up;
    P.Binv[idx] -= P.dvec[j] * f;
}

// ---- refinement: x and lambda from their definitions ------------------------------------------------------------
// rvec_r = h_r - b_r'x over the working sides (x = nullptr: rvec = h_W)
static __global__ __launch_bounds__(256) void k_mid_resid(MidLpIO P, int use_x) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= P.n) return;
    const int k = P.W[r];
    double acc = 0.0;
    if (use_x) {
        if (k >= 0) {
            const int64_t i = k >> 1;
            const double sg = (k & 1) ? -1.0 : 1.0;
            for (int64_t e = P.rowptr[i]; e < P.rowptr[i + 1]; ++e) acc += sg * P.val[e] * P.x[P.col[e]];
        } else {
            const int j = (-1 - k) >> 1;
            acc = ((-1 - k) & 1) ? -P.x[j] : P.x[j];
        }
    }
    P.rvec[r] = P.hW[r] - acc;
}
// x_j (+)= sum_r Binv[j][r] rvec_r: one wavefront per j
static __global__ __launch_bounds__(256) void k_mid_apply(MidLpIO P, int accumulate) {
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (j >= P.n) return;
    const double* row = P.Binv + (int64_t)j * P.n;
    double acc = 0.0;
    for (int r = lane; r < P.n; r += 64) acc += row[r] * P.rvec[r];
    acc = group_sum<64>(acc);
    if (lane == 0) P.x[j] = accumulate ? P.x[j] + acc : acc;
}
// lambda_r = -sum_j Binv[j][r] s c_j (coalesced over r), clamped at 0
static __global__ __launch_bounds__(256) void k_mid_lambda(MidLpIO P) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= P.n) return;
    double acc = 0.0;
    for (int j = 0; j < P.n; ++j) acc += P.Binv[(int64_t)j * P.n + r] * P.ctil[j];
    P.lam[r] = fmax(-acc, 0.0);
}
static __global__ void k_mid_rearm(MidState* st) { st->status = 0; st->pivots = 0; st->degen_run = 0; st->art_left = 0; st->p = -1; st->gj_singular = 0; }
// max |rvec| into st->resid, and back to "running" for the confirming price
static __global__ __launch_bounds__(256) void k_mid_resid_norm(MidLpIO P, int rearm) {
    __shared__ double sv[256];
    const int t = threadIdx.x;
    double m = 0.0;
    for (int r = t; r < P.n; r += 256) { const double v = fabs(P.rvec[r]); m = (v == v) ? fmax(m, v) : __builtin_inf(); }
    sv[t] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (t < s) sv[t] = fmax(sv[t], sv[t + s]); __syncthreads(); }
    if (t == 0) { P.st->resid = sv[0]; if (rearm && P.st->status == 1) P.st->status = 0; }
}

// outputs: row multipliers in the engine's convention (> 0 lower side, < 0 upper side), objective, leftover artificial sides
static __global__ __launch_bounds__(256) void k_mid_final(MidLpIO P, double* y) {
    __shared__ double sv[256];
    __shared__ int s_art;
    const int t = threadIdx.x, n = P.n;
    if (t == 0) s_art = 0;
    __syncthreads();
    double cmax = 1.0;
    for (int j = t; j < n; j += 256) cmax = fmax(cmax, fabs(P.c[j]));
    sv[t] = cmax;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (t < s) sv[t] = fmax(sv[t], sv[t + s]); __syncthreads(); }
    cmax = sv[0];
    __syncthreads();
    double acc = 0.0;
    for (int j = t; j < n; j += 256) acc += P.sgn * P.c[j] * P.x[j];
    sv[t] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (t < s) sv[t] += sv[t + s]; __syncthreads(); }
    for (int r = t; r < n; r += 256) {
        const int k = P.W[r];
        const double lm = fmax(P.lam[r], 0.0);
        if (k >= 0) y[k >> 1] = (k & 1) ? lm : -lm;
        else if (!(mid_bound_rhs(P, k) < __builtin_inf()) && lm > 1e-9 * cmax) atomicAdd(&s_art, 1);
    }
    __syncthreads();
    if (t == 0) { P.st->obj = sv[0]; P.st->art_left = s_art; }
}

// ---- refactorisation: B^-1 afresh from the working set ----------------------------------------------------------------
// The rank-one updates accumulate error (measured on the numpy mirror: ||B B^-1 - I|| = 8e-5 after a few thousand pivots on a
// cutting-plane LP), and a long cold start is thousands of pivots.  Every kMidRefactor pivots -- and before a warm solve
// whose inverse is older than that -- B is rebuilt from W into [B | I] and inverted by Gauss-Jordan with partial pivoting,
// one column per step (four small launches: pivot search, row swap + scaling, column extract, elimination: n = 512 -> 10 ms).
constexpr int kMidRefactor = 1500;
static __global__ __launch_bounds__(256) void k_mid_gj_build(MidLpIO P, double* __restrict__ aug) {      // aug: n x 2n, row r = [normal of W_r | e_r]
    const int n = P.n;
    const int r = blockIdx.x;
    double* row = aug + (int64_t)r * 2 * n;
    for (int q = threadIdx.x; q < 2 * n; q += 256) row[q] = (q == n + r) ? 1.0 : 0.0;
    __syncthreads();
    const int k = P.W[r];
    if (k >= 0) {
        const int64_t i = k >> 1;
        const double sg = (k & 1) ? -1.0 : 1.0;
        if (threadIdx.x == 0)                                  // (serial: a row may repeat a column; rows are short)
            for (int64_t e = P.rowptr[i]; e < P.rowptr[i + 1]; ++e) row[P.col[e]] += sg * P.val[e];
    } else if (threadIdx.x == 0) {
        row[(-1 - k) >> 1] = ((-1 - k) & 1) ? -1.0 : 1.0;
    }
    if (r == 0 && threadIdx.x == 0) P.st->gj_singular = 0;
}
static __global__ __launch_bounds__(256) void k_mid_gj_pivot(MidLpIO P, const double* __restrict__ aug, int col) {
    __shared__ double sv[256];
    __shared__ int si[256];
    const int n = P.n, t = threadIdx.x;
    double bv = -1.0; int bi = col;
    for (int r = col + t; r < n; r += 256) { const double v = fabs(aug[(int64_t)r * 2 * n + col]); if (v > bv) { bv = v; bi = r; } }
    sv[t] = bv; si[t] = bi;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (t < s && (sv[t + s] > sv[t] || (sv[t + s] == sv[t] && si[t + s] < si[t]))) { sv[t] = sv[t + s]; si[t] = si[t + s]; }
        __syncthreads();
    }
    if (t == 0) {
        P.st->gj_p = si[0];
        P.st->gj_piv = aug[(int64_t)si[0] * 2 * n + col];
        if (!(sv[0] > 1e-13)) P.st->gj_singular = 1;
    }
}
// swap rows col <-> p, and the scaled pivot row into prow
static __global__ __launch_bounds__(256) void k_mid_gj_swap(MidLpIO P, double* __restrict__ aug, int col, double* __restrict__ prow) {
    const int n = P.n;
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= 2 * n || P.st->gj_singular) return;
    const int p = P.st->gj_p;
    const double a = aug[(int64_t)col * 2 * n + q], b = aug[(int64_t)p * 2 * n + q];
    if (p != col) aug[(int64_t)p * 2 * n + q] = a;
    prow[q] = b / P.st->gj_piv;
}
static __global__ __launch_bounds__(256) void k_mid_gj_col(MidLpIO P, const double* __restrict__ aug, int col, double* __restrict__ fcol) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= P.n || P.st->gj_singular) return;
    fcol[r] = aug[(int64_t)r * 2 * P.n + col];
}
static __global__ __launch_bounds__(256) void k_mid_gj_elim(MidLpIO P, double* __restrict__ aug, int col, const double* __restrict__ prow,
                                                     const double* __restrict__ fcol) {
    const int n = P.n;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)n * 2 * n || P.st->gj_singular) return;
    const int r = (int)(idx / (2 * n)), q = (int)(idx % (2 * n));
    aug[idx] = (r == col) ? prow[q] : aug[idx] - fcol[r] * prow[q];
}
static __global__ __launch_bounds__(256) void k_mid_gj_store(MidLpIO P, const double* __restrict__ aug) {
    const int n = P.n;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)n * n || P.st->gj_singular) return;
    const int j = (int)(idx / n), r = (int)(idx % n);
    P.Binv[idx] = aug[(int64_t)j * 2 * n + n + r];
}

// after a purge: working rows move to their new indices; a working row that was dropped voids the warm start
static __global__ __launch_bounds__(256) void k_mid_remap(int n, int32_t* W, const int64_t* __restrict__ keep, const int64_t* __restrict__ newidx,
                                                   int32_t* __restrict__ lost) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    const int k = W[r];
    if (k < 0) return;
    const int64_t i = k >> 1;
    if (!keep[i]) { atomicAdd(lost, 1); return; }
    W[r] = (int32_t)(2 * newidx[i] + (k & 1));
}

}  // namespace ktn
