// batch_lp.hpp -- throughput mode (BASELINE.json configs[4]): ONE workgroup per instance runs the whole restarted
// reflected-Halpern PDHG of that instance's LP -- every iteration, every KKT check, every restart decision -- with
// __syncthreads() where the single-problem path has kernel boundaries and host round trips.
//
// The batch is loaded as one block-diagonal problem (instances.fuse_instances): sweep, cut bookkeeping, CSC mirror and
// diagonal scaling are the engine's ordinary kernels and serve all instances at once.  Only the LP iteration differs:
// block b owns the columns [blk_col[b], blk_col[b+1]) and the rows whose entries lie in them (listed in blk_rows, in row
// order).  Its iterates x, x0, xt|xbar, y, y0, yt live in LDS (6 vectors of ~1e3 doubles: 50 KB, three instances per CU);
// the matrix entries stream from L2 / Infinity Cache (24 B per non-zero and iteration for the two products -- a cfg5 LP is
// ~3e5 B, the 512 of them are 150 MB: they do not fit the 160 KB of LDS next to the vectors).
// The control logic is the host loop of Engine::lp_solve_core restated per block (termination, primal-stagnation exit,
// stalled-row acceptance, the three restart rules, the guarded primal-weight update, the step-size back-off); what a
// block cannot do here -- dual-mass consolidation, infeasibility certificates -- makes it report USERLIMIT, and the
// engine then finishes the solve with the ordinary loop from the point reached.
//
// Replaces, per instance, the loop of src/model.jl:257-309 around solve(m.linear_model) (:259).
#pragma once
#include "kernels.hpp"

namespace ktn {

constexpr int kBlkThreads = 1024;
constexpr int kBlkQ = 12;       // sums 0..9 + maxima 10, 11

struct BlkLp {
    // block structure
    const int64_t* blk_col;     // [nb + 1]
    const int32_t* blk_rowptr;  // [nb + 1] into blk_rows
    const int32_t* blk_rows;    // global row ids, block after block, ascending inside a block
    // matrix: CSR rows (global column ids) and CSC mirror with LOCAL row positions
    const int64_t* rptr; const int32_t* rcol; const double* rval;
    const int64_t* cptr; const int32_t* crowl; const double* cval;
    // scaled problem vectors (global indexing)
    const double* c; const double* l; const double* u; const double* lo; const double* hi; const double* dr; const double* dc;
    double* x; double* y;       // in: warm start (scaled); untouched on exit
    double* xt; double* yt;     // out: T(z) of the last check (the point the engine un-scales)
    double* omega;              // [nb] in/out primal weights (<= 0: start from the reference weight)
    double* res;                // [nb * 8] out: status, iterations, pobj, dobj, pviol, gap, restarts, dres
    double tol_p, tol_g, eta0, eta_safe, stag_factor, stall_accept;
    int check_every, first_chunk, near_chunk, max_iter, nmax, mmax;
};

template <int NQ>
__device__ __forceinline__ void blk_reduce(double (&v)[NQ], int nsum, double* red, double* out) {
    // v[0..nsum) are summed, v[nsum..NQ) maximised; fixed shape: butterfly per wavefront, wavefronts in order
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        double s = (q < nsum) ? group_sum<64>(v[q]) : group_max<64>(v[q]);
        if (lane == 0) red[wv * NQ + q] = s;
    }
    __syncthreads();
    if (threadIdx.x < NQ) {
        const int q = threadIdx.x;
        double s = red[q];
        for (int k = 1; k < kBlkThreads / 64; ++k) s = (q < nsum) ? s + red[k * NQ + q] : fmax(s, red[k * NQ + q]);
        out[q] = s;
    }
    __syncthreads();
}


// Sparse dot products of up to T consecutive "trips" of a thread (trip t serves output o0 + t * stride), G lanes per
// output, with ALL first-level loads (entry ranges), then ALL second-level loads (the first two entries per lane of every
// trip) issued before anything is used: a phase costs two memory latencies, not two per trip.  Outputs past `count` and
// entries past a range contribute nothing; ranges longer than 2 G entries finish in a tail loop.
template <int G, int T, bool TWO>
__device__ __forceinline__ void blk_dots(int o0, int stride, int count, int lane, const int64_t* __restrict__ ptr, const int64_t* gidx /* global output id per trip */,
                                         const int32_t* __restrict__ idx, const double* __restrict__ val, int idx_off,
                                         const double* v1, const double* v2, double (&acc1)[T], double (&acc2)[T]) {
    int64_t beg[T], end[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const bool on = o0 + t * stride < count;
        beg[t] = on ? ptr[gidx[t]] : 0;
        end[t] = on ? ptr[gidx[t] + 1] : 0;
    }
    double a0[T], a1[T];
    int i0[T], i1[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int64_t e0 = beg[t] + lane, e1 = e0 + G;
        const bool on0 = e0 < end[t], on1 = e1 < end[t];
        a0[t] = on0 ? val[e0] : 0.0; i0[t] = on0 ? idx[e0] - idx_off : 0;
        a1[t] = on1 ? val[e1] : 0.0; i1[t] = on1 ? idx[e1] - idx_off : 0;
    }
#pragma unroll
    for (int t = 0; t < T; ++t) {
        double s1 = a0[t] * v1[i0[t]] + a1[t] * v1[i1[t]];
        double s2 = TWO ? a0[t] * v2[i0[t]] + a1[t] * v2[i1[t]] : 0.0;
        for (int64_t e = beg[t] + lane + 2 * G; e < end[t]; e += G) {
            const double vv = val[e];
            const int ii = idx[e] - idx_off;
            s1 += vv * v1[ii];
            if (TWO) s2 += vv * v2[ii];
        }
        acc1[t] = group_sum<G>(s1);
        if (TWO) acc2[t] = group_sum<G>(s2);
    }
}

static __global__ __launch_bounds__(kBlkThreads) void k_pdhg_blocks(BlkLp P) {
    extern __shared__ double sm[];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int64_t c0 = P.blk_col[b];
    const int nb = (int)(P.blk_col[b + 1] - c0);
    const int r0 = P.blk_rowptr[b], mb = P.blk_rowptr[b + 1] - r0;
    double* xs = sm;                 // x
    double* x0s = xs + P.nmax;       // anchor
    double* xts = x0s + P.nmax;      // xbar in plain iterations, xt in check iterations
    double* ys = xts + P.nmax;
    double* y0s = ys + P.mmax;
    double* yts = y0s + P.mmax;
    double* red = yts + P.mmax;                       // [8 waves][kBlkQ]
    double* q = red + (kBlkThreads / 64) * kBlkQ;     // [kBlkQ] reduced quantities
    double* ctl = q + kBlkQ;                          // [8] broadcast of thread 0's decision
    int32_t* rows_l = reinterpret_cast<int32_t*>(ctl + 8);   // [mmax] global ids of the block's rows
    constexpr int GX = 4, GY = 4;
    constexpr int TX = 4, TY = 5;      // trips per thread and pass: 1024 columns / 1280 rows per pass (a cfg5 block: one pass each)

    // ---- start: iterates into LDS, ||c||^2 and ||(lo, hi) finite||^2 of the block
    double v2[2] = {0.0, 0.0};
    for (int j = tid; j < nb; j += kBlkThreads) {
        const double xv = P.x[c0 + j];
        xs[j] = xv; x0s[j] = xv; xts[j] = xv;
        const double cj = P.c[c0 + j];
        v2[0] += cj * cj;
    }
    for (int r = tid; r < mb; r += kBlkThreads) {
        const int gi = P.blk_rows[r0 + r];
        rows_l[r] = gi;
        const double yv = P.y[gi];
        ys[r] = yv; y0s[r] = yv; yts[r] = yv;
        const double a = P.lo[gi], bb = P.hi[gi];
        if (isfinite(a)) v2[1] += a * a;
        if (isfinite(bb)) v2[1] += bb * bb;
    }
    __syncthreads();
    blk_reduce<2>(v2, 2, red, q);
    const double nc2 = q[0], nb2 = q[1];
    __syncthreads();
    const double omega_ref = (nc2 > 0.0 && nb2 > 0.0) ? sqrt(nc2 / nb2) : 1.0;
    const double dres_tol_scale = 1.0 + sqrt(nc2);

    // ---- control state (held redundantly by every thread; decisions are taken by thread 0 and broadcast through ctl)
    double om = P.omega[b] > 0.0 ? P.omega[b] : omega_ref;
    double eta = P.eta0;
    int k = 0, it = 0, stall = 0, restarts = 0;
    double rr0 = 0.0, r_prev = 0.0, r_last = 0.0;
    double pobj_h0 = 1e300, pobj_h1 = -1e300, pobj_h2 = 1e300, pv_h0 = 1e300, pv_h1 = -1e300, pv_h2 = 1e300;
    bool plain_next = false, near_conv = false;
    int status = KTN_STATUS_USERLIMIT;
    double pobj = 0.0, dobj = 0.0, pviol = 0.0, gap = 0.0, dres = 0.0;
    const int plain_len = P.check_every - 1;

    while (it < P.max_iter) {
        const double tau = eta / om, sigma = eta * om;
        if (plain_next) {
            plain_next = false;
            int want = (k <= 1 && P.first_chunk > 0) ? min(P.first_chunk, plain_len) : plain_len;
            if (near_conv && P.near_chunk > 0) want = min(want, P.near_chunk);
            const int np = min(want, P.max_iter - it);
            for (int s = 0; s < np; ++s) {
                const double w = (double)(k + s + 1) / (double)(k + s + 2);
                // x-step: G lanes per column, y gathered from LDS through the local row positions
                for (int jb = 0; jb < nb; jb += TX * (kBlkThreads / GX)) {
                    const int lane = tid & (GX - 1), j0 = jb + tid / GX;
                    int64_t gj[TX];
#pragma unroll
                    for (int t = 0; t < TX; ++t) gj[t] = c0 + j0 + t * (kBlkThreads / GX);
                    double cj[TX], lj[TX], uj[TX];
#pragma unroll
                    for (int t = 0; t < TX; ++t) {
                        const bool on = j0 + t * (kBlkThreads / GX) < nb;
                        cj[t] = on ? P.c[gj[t]] : 0.0; lj[t] = on ? P.l[gj[t]] : 0.0; uj[t] = on ? P.u[gj[t]] : 0.0;
                    }
                    double acc[TX], dummy[TX];
                    blk_dots<GX, TX, false>(j0, kBlkThreads / GX, nb, lane, P.cptr, gj, P.crowl, P.cval, 0, ys, ys, acc, dummy);
                    if (lane == 0) {
#pragma unroll
                        for (int t = 0; t < TX; ++t) {
                            const int j = j0 + t * (kBlkThreads / GX);
                            if (j < nb) {
                                const double xv = xs[j];
                                const double xtv = clampd(xv - tau * (cj[t] - acc[t]), lj[t], uj[t]);
                                xts[j] = 2.0 * xtv - xv;
                                xs[j] = w * (2.0 * xtv - xv) + (1.0 - w) * x0s[j];
                            }
                        }
                    }
                }
                __syncthreads();
                // y-step: G lanes per row, xbar gathered from LDS
                for (int rb = 0; rb < mb; rb += TY * (kBlkThreads / GY)) {
                    const int lane = tid & (GY - 1), rr = rb + tid / GY;
                    int64_t gi[TY];
#pragma unroll
                    for (int t = 0; t < TY; ++t) { const int r = rr + t * (kBlkThreads / GY); gi[t] = r < mb ? rows_l[r] : 0; }
                    double loi[TY], hii[TY];
#pragma unroll
                    for (int t = 0; t < TY; ++t) {
                        const bool on = rr + t * (kBlkThreads / GY) < mb;
                        loi[t] = on ? P.lo[gi[t]] : 0.0; hii[t] = on ? P.hi[gi[t]] : 0.0;
                    }
                    double acc[TY], dummy[TY];
                    blk_dots<GY, TY, false>(rr, kBlkThreads / GY, mb, lane, P.rptr, gi, P.rcol, P.rval, (int)c0, xts, xts, acc, dummy);
                    if (lane == 0) {
#pragma unroll
                        for (int t = 0; t < TY; ++t) {
                            const int r = rr + t * (kBlkThreads / GY);
                            if (r < mb) {
                                const double yv = ys[r];
                                const double v = yv - sigma * acc[t];
                                const double ytv = v + sigma * clampd(-v / sigma, loi[t], hii[t]);
                                ys[r] = w * (2.0 * ytv - yv) + (1.0 - w) * y0s[r];
                            }
                        }
                    }
                }
                __syncthreads();
            }
            k += np; it += np;
            continue;
        }
        // ---- check iteration: T(z) without update, KKT and fixed-point sums
        double a[kBlkQ];
#pragma unroll
        for (int i = 0; i < kBlkQ; ++i) a[i] = 0.0;
        //   a0 dy*(A dx)  a1 dy^2  a2 dual obj (rows)  a3 (yt-y0)^2  a4 dx^2  a5 primal obj  a6 dual obj (bounds)  a7 (xt-x0)^2
        //   a8 xt^2  a9 yt^2  a10 max row violation (unscaled)  a11 max dual residual (unscaled)
        for (int j = tid / GX; j < nb; j += kBlkThreads / GX) {
            const int lane = tid & (GX - 1);
            const int64_t gj = c0 + j;
            const int64_t beg = P.cptr[gj], end = P.cptr[gj + 1];
            double acc = 0.0;
            for (int64_t e = beg + lane; e < end; e += GX) acc += P.cval[e] * ys[P.crowl[e]];
            acc = group_sum<GX>(acc);
            if (lane == 0) xts[j] = clampd(xs[j] - tau * (P.c[gj] - acc), P.l[gj], P.u[gj]);
        }
        __syncthreads();
        for (int r = tid / GY; r < mb; r += kBlkThreads / GY) {
            const int lane = tid & (GY - 1);
            const int gi = rows_l[r];
            const int64_t beg = P.rptr[gi], end = P.rptr[gi + 1];
            double axt = 0.0, axk = 0.0;
            for (int64_t e = beg + lane; e < end; e += GY) {
                const int cl = P.rcol[e] - (int)c0;
                const double vv = P.rval[e];
                axt += vv * xts[cl];
                axk += vv * xs[cl];
            }
            axt = group_sum<GY>(axt);
            axk = group_sum<GY>(axk);
            if (lane == 0) {
                const double loi = P.lo[gi], hii = P.hi[gi], yv = ys[r];
                const double v = yv - sigma * (2.0 * axt - axk);
                const double ytv = v + sigma * clampd(-v / sigma, loi, hii);
                yts[r] = ytv;
                const double dy = ytv - yv;
                a[0] += dy * (axt - axk);
                a[1] += dy * dy;
                if (ytv > 0.0) { if (loi > -__builtin_inf()) a[2] += loi * ytv; }
                else if (ytv < 0.0) { if (hii < __builtin_inf()) a[2] += hii * ytv; }
                const double d0 = ytv - y0s[r];
                a[3] += d0 * d0;
                a[9] += ytv * ytv;
                a[10] = fmax(a[10], fmax(fmax(loi - axt, axt - hii), 0.0) / P.dr[gi]);
            }
        }
        __syncthreads();
        for (int j = tid / GX; j < nb; j += kBlkThreads / GX) {
            const int lane = tid & (GX - 1);
            const int64_t gj = c0 + j;
            const int64_t beg = P.cptr[gj], end = P.cptr[gj + 1];
            double aty = 0.0;
            for (int64_t e = beg + lane; e < end; e += GX) aty += P.cval[e] * yts[P.crowl[e]];
            aty = group_sum<GX>(aty);
            if (lane == 0) {
                const double xtv = xts[j], cj = P.c[gj], lj = P.l[gj], uj = P.u[gj];
                const double dx = xtv - xs[j];
                a[4] += dx * dx;
                a[5] += cj * xtv;
                const double rc = cj - aty;
                double bad = 0.0;
                if (rc > 0.0) { if (isfinite(lj)) a[6] += lj * rc; else bad = rc; }
                else if (rc < 0.0) { if (isfinite(uj)) a[6] += uj * rc; else bad = -rc; }
                const double d0 = xtv - x0s[j];
                a[7] += d0 * d0;
                a[8] += xtv * xtv;
                a[11] = fmax(a[11], bad / P.dc[gj]);
            }
        }
        blk_reduce<kBlkQ>(a, 10, red, q);
        if (tid == 0) {
            const double dyAdx = q[0], dy2 = q[1], dy0sq = q[3], dx2 = q[4], dx0sq = q[7], xt2 = q[8], yt2 = q[9];
            pobj = q[5]; dobj = q[2] + q[6]; pviol = q[10]; dres = q[11];
            const double r2 = om / eta * dx2 - 2.0 * dyAdx + dy2 / (eta * om);
            const double r = sqrt(fmax(r2, 0.0));
            gap = fabs(pobj - dobj) / (1.0 + fabs(pobj) + fabs(dobj));
            if (k == 0) { rr0 = r; r_prev = r; }
            const bool dres_ok = dres <= P.tol_g * dres_tol_scale;
            bool done = (pviol <= P.tol_p) && (gap <= P.tol_g) && dres_ok;
            const bool near = (pviol <= 4.0 * P.tol_p) && (gap <= 4.0 * P.tol_g) && (dres <= 4.0 * P.tol_g * dres_tol_scale);
            if (P.stag_factor > 0.0 && !done) {
                const double scale = 1.0 + fabs(pobj), f = 0.1 * P.tol_g * scale;
                const bool flat = fabs(pobj - pobj_h0) <= f && fabs(pobj - pobj_h1) <= f && fabs(pobj - pobj_h2) <= f;
                const bool plateau = pviol <= P.stall_accept * P.tol_p && fabs(pviol - pv_h0) <= 0.02 * pviol &&
                                     fabs(pviol - pv_h1) <= 0.02 * pviol && fabs(pviol - pv_h2) <= 0.02 * pviol;
                if (flat && (pviol <= P.tol_p || plateau) && gap <= P.stag_factor * P.tol_g && dres_ok) done = true;
                if (!done && gap <= P.tol_g && dres_ok && plateau) done = true;
            }
            pv_h2 = pv_h1; pv_h1 = pv_h0; pv_h0 = pviol;
            pobj_h2 = pobj_h1; pobj_h1 = pobj_h0; pobj_h0 = pobj;
            int action = 0;                                   // 0 continue (Halpern update), 1 restart, 2 done, 3 give up
            if (done) action = 2;
            else if (!(r == r)) action = 3;
            else {
                bool restart = k > 0 && (r <= 0.2 * rr0 || (r <= 0.8 * rr0 && r > r_prev) || (double)k >= 0.36 * (double)(it + 1));
                if (k > 0 && eta > P.eta_safe * (1.0 + 1e-12)) {
                    stall = (r2 < 0.0 || (r_last > 0.0 && r > 0.97 * r_last && r < 1.03 * r_last)) ? stall + 1 : 0;
                    if (stall >= 3 || r2 < 0.0) { eta = fmax(P.eta_safe, 0.85 * eta); stall = 0; restart = true; }
                }
                r_last = r; r_prev = r;
                if (restart) {
                    const double dx = sqrt(dx0sq), dy = sqrt(dy0sq);
                    if (dx > 1e-8 * (1.0 + sqrt(xt2)) && dy > 1e-8 * (1.0 + sqrt(yt2))) {
                        om = exp(0.5 * log(dy / dx) + 0.5 * log(om));
                        om = fmin(fmax(om, omega_ref * 1e-3), omega_ref * 1e3);
                    }
                    action = 1;
                }
            }
            ctl[0] = (double)action; ctl[1] = om; ctl[2] = eta; ctl[3] = near ? 1.0 : 0.0;
        }
        __syncthreads();
        const int action = (int)ctl[0];
        om = ctl[1]; eta = ctl[2]; near_conv = ctl[3] != 0.0;
        __syncthreads();
        ++it;
        if (action == 2) { status = KTN_STATUS_OPTIMAL; break; }
        if (action == 3) { status = KTN_STATUS_ERROR; break; }
        if (action == 1) {
            for (int j = tid; j < nb; j += kBlkThreads) { const double v = xts[j]; xs[j] = v; x0s[j] = v; }
            for (int r = tid; r < mb; r += kBlkThreads) { const double v = yts[r]; ys[r] = v; y0s[r] = v; }
            k = 0; ++restarts;
            __syncthreads();
            continue;                                         // the next pass is a check again: it sets rr0
        }
        {
            const double w = (double)(k + 1) / (double)(k + 2);
            for (int j = tid; j < nb; j += kBlkThreads) xs[j] = w * (2.0 * xts[j] - xs[j]) + (1.0 - w) * x0s[j];
            for (int r = tid; r < mb; r += kBlkThreads) ys[r] = w * (2.0 * yts[r] - ys[r]) + (1.0 - w) * y0s[r];
            ++k;
            plain_next = true;
            __syncthreads();
        }
    }
    // ---- result: T(z) of the last check (or the current iterate when no check completed)
    for (int j = tid; j < nb; j += kBlkThreads) P.xt[c0 + j] = xts[j];
    for (int r = tid; r < mb; r += kBlkThreads) P.yt[rows_l[r]] = yts[r];
    if (tid == 0) {
        P.omega[b] = om;
        double* o = P.res + (int64_t)b * 8;
        o[0] = (double)status; o[1] = (double)it; o[2] = pobj; o[3] = dobj; o[4] = pviol; o[5] = gap; o[6] = (double)restarts; o[7] = dres;
    }
}

// row -> block (binary search of the row's first column in blk_col); rows without entries go to block 0
static __global__ __launch_bounds__(kBlock) void k_row_block(int64_t m, const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                      const int64_t* __restrict__ blk_col, int nblk, uint64_t* __restrict__ keys,
                                                      uint32_t* __restrict__ vals) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= m) return;
    int b = 0;
    if (rowptr[i + 1] > rowptr[i]) {
        const int64_t c = col[rowptr[i]];
        int lo = 0, hi = nblk;                     // blk_col[lo] <= c < blk_col[hi]
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (blk_col[mid] <= c) lo = mid; else hi = mid; }
        b = lo;
    }
    keys[i] = (uint64_t)b;
    vals[i] = (uint32_t)i;
}
// sorted (block, row) pairs -> block row pointers, row list, local position of every row
static __global__ __launch_bounds__(kBlock) void k_block_rows(int64_t m, const uint64_t* __restrict__ skeys, const uint32_t* __restrict__ svals,
                                                       int nblk, int32_t* __restrict__ blk_rowptr, int32_t* __restrict__ blk_rows,
                                                       int32_t* __restrict__ row_loc) {
    const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (p >= m) return;
    const int b = (int)skeys[p];
    blk_rows[p] = (int32_t)svals[p];
    const int prevb = p > 0 ? (int)skeys[p - 1] : -1;
    for (int bb = prevb + 1; bb <= b; ++bb) blk_rowptr[bb] = (int32_t)p;      // first position of every block up to b
    if (p == m - 1) for (int bb = b + 1; bb <= nblk; ++bb) blk_rowptr[bb] = (int32_t)m;
}
static __global__ __launch_bounds__(kBlock) void k_row_local(int64_t m, const uint64_t* __restrict__ skeys, const uint32_t* __restrict__ svals,
                                                      const int32_t* __restrict__ blk_rowptr, int32_t* __restrict__ row_loc) {
    const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (p >= m) return;
    row_loc[svals[p]] = (int32_t)(p - blk_rowptr[(int)skeys[p]]);
}
static __global__ __launch_bounds__(kBlock) void k_localize_rows(int64_t nnz, const int32_t* __restrict__ crow, const int32_t* __restrict__ row_loc,
                                                          int32_t* __restrict__ crowl) {
    const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (e < nnz) crowl[e] = row_loc[crow[e]];
}

}  // namespace ktn
