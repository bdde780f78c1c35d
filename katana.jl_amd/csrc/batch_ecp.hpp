// batch_ecp.hpp -- throughput mode, second stage (BASELINE.json configs[4] "one instance per CU group"): ONE workgroup runs
// the WHOLE cutting-plane loop of one instance -- src/model.jl:257-309 with everything inside it: LP scaling, step-size
// estimate, the restarted reflected-Halpern PDHG with its checks and restarts, the separator sweep over the instance's
// nonlinear rows (precompute! + isconstrsat + gencut + round_coefs + _addcut), the column mirror of the grown matrix, the
// tolerance schedule and the stop rule -- so that no instance ever waits for another (with a host-driven round per
// cutting-plane iteration every round waits for its slowest instance: csrc/batch_lp.hpp).
//
// The batch is loaded as one block-diagonal problem (instances.fuse_instances) of SEPARABLE rows with a linear objective;
// ktn_loadproblem builds the fused LP of the linear rows as usual.  k_ecp_blocks then works on a per-instance ARENA in
// global memory (the instance's LP in local indices with room for its cuts): CSR rows (unscaled + scaled values), CSC
// mirror (unscaled + scaled), row bounds, duals, diagonal scalings, scaled problem vectors.  The PDHG iterates live in
// LDS; matrix entries stream from L2 / Infinity Cache.  Everything is deterministic: sums have a fixed order, the column
// mirror is sorted by row after the counting-sort scatter.
#pragma once
#include "kernels.hpp"

namespace ktn {

constexpr int kEcpThreads = 1024;
constexpr int kEcpQ = 12;

struct EcpArena {            // per-instance offsets into the flat arrays below (all local indices are instance-relative)
    int64_t row0, nnz0;      // first row slot / first entry slot of the instance
    int32_t cap_rows, cap_nnz;
};

struct EcpBatch {
    // ---- fused problem as loaded by the engine
    const int64_t* blk_col;          // [nb + 1] column offsets
    const int64_t* blk_lin;          // [nb + 1] offsets of the instances' linear rows in the fused LP (rows of the loaded LP)
    const int64_t* blk_nl;           // [nb + 1] offsets of the instances' NL slots (positions in nl_rows)
    const int64_t* lp_rowptr; const int32_t* lp_col; const double* lp_val; const double* lp_lo; const double* lp_hi;   // loaded LP
    const double* c; const double* l; const double* u;                                                                // LP columns (unscaled)
    NlpDev P; const int32_t* nl_rows;                                                                                 // NL rows (global ids)
    // ---- arenas
    const EcpArena* arena;
    int32_t* rptr; uint16_t* rcol; double* rval; double* rsval; double* lo; double* hi; double* y; double* dr; double* loh; double* hih;
    int32_t* cptr; uint16_t* crow; double* cval; double* csval;    // cptr: [ncols_total + nb] (instance b at blk_col[b] + b); 16-bit local
                                                                   // indices: 20 instead of 24 bytes per non-zero and PDHG iteration
    double* dc; double* ch; double* lh; double* uh;                // per column (global column indexing)
    int32_t* last_cut;                                             // per NL slot: local row of its newest cut (-1)
    int32_t* cut_prev;                                             // per row: previous cut of the same NL slot (-1): lists for the stall handler
    double* ax;                                                    // per row: scaled activity A^ xt of the last check
    double* x;                                                     // [ncols_total] out: solution (unscaled); in: ignored
    double* xbest;                                                 // [ncols_total] scratch: best point of the certificate refinement
    double* res;                                                   // [nb * 8] out: status, ecp iterations, objective, cuts, pdhg iterations, max violation, lp rows, -
    // ---- parameters
    double f_tol, cut_coef_rng, tol_scale, tol_floor, tol_cap, gap_floor, gap_cap, stag_factor;
    double cert_tol;                 // obj_cert_tol: per-instance objective certificate (below); 0 = stop at the reference's rule alone
    int iter_cap, lp_max_iter, check_every, near_chunk, ruiz_iters, power_passes, nmax, mmax, polish_max_iter;
};

// ---------------------------------------------------------------------------------------------------------------------
template <int NQ>
__device__ __forceinline__ void ecp_reduce(double (&v)[NQ], int nsum, double* red, double* out) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const double s = (q < nsum) ? group_sum<64>(v[q]) : group_max<64>(v[q]);
        if (lane == 0) red[wv * NQ + q] = s;
    }
    __syncthreads();
    if (threadIdx.x < NQ) {
        const int q = threadIdx.x;
        double s = red[q];
        for (int k = 1; k < kEcpThreads / 64; ++k) s = (q < nsum) ? s + red[k * NQ + q] : fmax(s, red[k * NQ + q]);
        out[q] = s;
    }
    __syncthreads();
}

// sparse dot products of the block's outputs (rows or columns) with an LDS vector: out(o, acc) called by lane 0 of each
// group.  T outputs per lane group and pass: the entry ranges of all T, then the first two entries per lane of all T, are
// requested before anything is used -- a pass costs two memory latencies instead of two per output (a lone workgroup, the
// tail of the batch, is latency-bound).
struct EcpPre { double a, b, c; };
template <int G, int T, class PF, class F>
__device__ __forceinline__ void ecp_spmv(int count, const int32_t* __restrict__ ptr, const uint16_t* __restrict__ idx,
                                         const double* __restrict__ val, const double* v, PF&& pre, F&& out) {
    constexpr int kGroups = kEcpThreads / G;
    const int lane = threadIdx.x & (G - 1), g0 = threadIdx.x / G;
    static_assert(T <= G, "one output per lane of the group");
    for (int base = 0; base < count; base += T * kGroups) {
        // the scalars the step's element-wise tail needs for the output THIS lane will finish (bounds, cost) are requested
        // first: their latency overlaps the gather chain
        const int omine = base + g0 + lane * kGroups;
        const bool fin = lane < T && omine < count;
        EcpPre pf{0.0, 0.0, 0.0};
        if (fin) pf = pre(omine);
        int beg[T], end[T];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int o = base + g0 + t * kGroups;
            const bool on = o < count;
            beg[t] = on ? ptr[o] : 0;
            end[t] = on ? ptr[o + 1] : 0;
        }
        double a0[T], a1[T];
        int i0[T], i1[T];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int e0 = beg[t] + lane, e1 = e0 + G;
            const bool on0 = e0 < end[t], on1 = e1 < end[t];
            a0[t] = on0 ? val[e0] : 0.0; i0[t] = on0 ? idx[e0] : 0;
            a1[t] = on1 ? val[e1] : 0.0; i1[t] = on1 ? idx[e1] : 0;
        }
        // the butterfly leaves every lane of the group with the sum: lane t finishes output t (T <= G), so the element-wise
        // tail of the step (clamps, the Halpern update) runs once per group with all lanes busy instead of T times with one
        // lane in G
        double mine = 0.0;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            double acc = a0[t] * v[i0[t]] + a1[t] * v[i1[t]];
            for (int e = beg[t] + lane + 2 * G; e < end[t]; e += G) acc += val[e] * v[idx[e]];
            acc = group_sum<G>(acc);
            if (lane == t) mine = acc;
        }
        if (fin) out(omine, mine, pf);
    }
}

static __global__ __launch_bounds__(kEcpThreads) void k_ecp_blocks(EcpBatch B) {
    extern __shared__ double sm[];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int64_t c0 = B.blk_col[b];
    const int nb = (int)(B.blk_col[b + 1] - c0);
    const int64_t lin0 = B.blk_lin[b];
    const int m_lin = (int)(B.blk_lin[b + 1] - lin0);
    const int64_t nl0 = B.blk_nl[b];
    const int m_nl = (int)(B.blk_nl[b + 1] - nl0);
    const EcpArena A = B.arena[b];
    int32_t* rptr = B.rptr + A.row0 + b;          // cap_rows + 1 slots per instance
    uint16_t* rcol = B.rcol + A.nnz0;
    double* rval = B.rval + A.nnz0;
    double* rsval = B.rsval + A.nnz0;
    double* lo = B.lo + A.row0; double* hi = B.hi + A.row0; double* yv = B.y + A.row0; double* dr = B.dr + A.row0;
    double* loh = B.loh + A.row0; double* hih = B.hih + A.row0;
    int32_t* cptr = B.cptr + c0 + b;              // nb + 1 slots per instance
    uint16_t* crow = B.crow + A.nnz0;
    double* cval = B.cval + A.nnz0;
    double* csval = B.csval + A.nnz0;
    double* dc = B.dc + c0; double* ch = B.ch + c0; double* lh = B.lh + c0; double* uh = B.uh + c0;
    const double* cc = B.c + c0; const double* ll = B.l + c0; const double* uu = B.u + c0;
    int32_t* last_cut = B.last_cut + nl0;
    int32_t* cut_prev = B.cut_prev + A.row0;
    double* axv = B.ax + A.row0;
    double* xg = B.x + c0;
    double* xbest = B.xbest + c0;

    double* xs = sm;                   // x (scaled)
    double* x0s = xs + B.nmax;
    double* xts = x0s + B.nmax;        // xbar | xt ; also scratch
    double* ys = xts + B.nmax;
    double* y0s = ys + B.mmax;
    double* yts = y0s + B.mmax;
    double* red = yts + B.mmax;                        // [waves][kEcpQ]
    double* q = red + (kEcpThreads / 64) * kEcpQ;      // [kEcpQ]
    double* ctl = q + kEcpQ;                           // [8]
    double* st = ctl + 8;                              // [16] thread 0's control state of the LP loop (kept out of every thread's registers)
    int32_t* icnt = reinterpret_cast<int32_t*>(st + 16);   // [max(nmax, mmax) + 2] integer scratch (column counts, flags)

    // ================================================================ initial LP: the instance's linear rows =========
    int M = m_lin, NNZ = 0;
    {
        const int64_t e0 = B.lp_rowptr[lin0];
        NNZ = (int)(B.lp_rowptr[lin0 + m_lin] - e0);
        for (int r = tid; r <= m_lin; r += kEcpThreads) rptr[r] = (int32_t)(B.lp_rowptr[lin0 + r] - e0);
        for (int r = tid; r < m_lin; r += kEcpThreads) { lo[r] = B.lp_lo[lin0 + r]; hi[r] = B.lp_hi[lin0 + r]; yv[r] = 0.0; }
        for (int e = tid; e < NNZ; e += kEcpThreads) { rcol[e] = (uint16_t)(B.lp_col[e0 + e] - (int32_t)c0); rval[e] = B.lp_val[e0 + e]; }
        for (int i = tid; i < m_nl; i += kEcpThreads) last_cut[i] = -1;
        for (int j = tid; j < nb; j += kEcpThreads) xg[j] = 0.0;
    }
    __syncthreads();
    int status = KTN_STATUS_NONE, iter = 0, numcuts = m_lin;
    long pdhg_total = 0;
    double last_maxviol = 1e300, objval = 0.0, om_keep = -1.0;
    bool allsat = false, overflow = false;
    // Objective certificate per instance (Engine::objective_certificate / kernels.hpp "objective certificate", round 4): at the
    // point that meets the reference's stop rule the workgroup adds up  D = sum_i lambda_i * (signed residual of NL row i)  with
    // lambda_i the LP duals summed over the cuts of row i -- the part of  f* - objective  the stop rule leaves open -- and while
    // D exceeds half of cert_tol * max(1, |objective|) it keeps cutting at  f_eff = phi * f_tol  with the LP gap tolerance at a
    // quarter of the target (passes not counted in the iteration number, at most polish_max_iter).  The point returned is the
    // one with the smallest violation among those that satisfy the reference's rule.
    double f_eff = B.f_tol, gap_cert = 1e300, best_viol = 1e300, best_obj = 0.0;
    int passes = 0;
    bool refining = false;
  for (;;) {
    const double floor_p = B.tol_floor * f_eff;

    while (!allsat && (refining ? passes <= B.polish_max_iter : iter < B.iter_cap)) {
        if (refining) ++passes; else ++iter;
        double tol_p = fmin(fmax(B.tol_scale * last_maxviol, floor_p), B.tol_cap);
        if (m_nl == 0 || refining) tol_p = floor_p;
        const double tol_g = fmin(fmin(fmax(tol_p, B.gap_floor), B.gap_cap), gap_cert);

        // ============================================================ column mirror of the current rows ================
        // counting sort by column, then every column's entries sorted by row (fixed summation order of A'y)
        for (int j = tid; j <= nb; j += kEcpThreads) icnt[j] = 0;
        __syncthreads();
        for (int e = tid; e < NNZ; e += kEcpThreads) atomicAdd(&icnt[rcol[e] + 1], 1);
        __syncthreads();
        if (tid == 0) { int run = 0; for (int j = 0; j <= nb; ++j) { run += icnt[j]; cptr[j] = run; } }   // nb ~ 1e3: a serial scan is cheap
        __syncthreads();
        for (int j = tid; j < nb; j += kEcpThreads) icnt[j] = cptr[j];                 // cursors
        __syncthreads();
        for (int r = tid; r < M; r += kEcpThreads)
            for (int e = rptr[r]; e < rptr[r + 1]; ++e) {
                const int p = atomicAdd(&icnt[rcol[e]], 1);
                crow[p] = (uint16_t)r; cval[p] = rval[e];
            }
        __syncthreads();
        for (int j = tid; j < nb; j += kEcpThreads) {                                   // insertion sort by row (short lists)
            const int beg = cptr[j], end = cptr[j + 1];
            for (int a = beg + 1; a < end; ++a) {
                const uint16_t rr = crow[a]; const double vv = cval[a];
                int p = a - 1;
                while (p >= beg && crow[p] > rr) { crow[p + 1] = crow[p]; cval[p + 1] = cval[p]; --p; }
                crow[p + 1] = rr; cval[p + 1] = vv;
            }
        }
        __syncthreads();

        // ============================================================ diagonal scaling: Ruiz passes + Pock-Chambolle ===
        for (int r = tid; r < M; r += kEcpThreads) dr[r] = 1.0;
        for (int j = tid; j < nb; j += kEcpThreads) dc[j] = 1.0;
        __syncthreads();
        for (int pass = 0; pass <= B.ruiz_iters; ++pass) {
            const bool pc = pass == B.ruiz_iters;
            // stats with the CURRENT dr, dc (rows into yts, columns into xts), then both applied
            {
                const int lane = tid & 3;
                for (int r = tid / 4; r < M; r += kEcpThreads / 4) {
                    double acc = 0.0;
                    for (int e = rptr[r] + lane; e < rptr[r + 1]; e += 4) { const double v = fabs(rval[e]) * dc[rcol[e]]; acc = pc ? acc + v : fmax(acc, v); }
                    acc = pc ? group_sum<4>(acc) : group_max<4>(acc);
                    if (lane == 0) yts[r] = dr[r] * acc;
                }
                for (int j = tid / 4; j < nb; j += kEcpThreads / 4) {
                    double acc = 0.0;
                    for (int e = cptr[j] + lane; e < cptr[j + 1]; e += 4) { const double v = fabs(cval[e]) * dr[crow[e]]; acc = pc ? acc + v : fmax(acc, v); }
                    acc = pc ? group_sum<4>(acc) : group_max<4>(acc);
                    if (lane == 0) xts[j] = dc[j] * acc;
                }
            }
            __syncthreads();
            for (int r = tid; r < M; r += kEcpThreads) { const double s = yts[r]; if (s > 0.0 && isfinite(s)) dr[r] /= sqrt(s); }
            for (int j = tid; j < nb; j += kEcpThreads) { const double s = xts[j]; if (s > 0.0 && isfinite(s)) dc[j] /= sqrt(s); }
            __syncthreads();
        }
        for (int r = tid; r < M; r += kEcpThreads) {
            const double d = dr[r];
            for (int e = rptr[r]; e < rptr[r + 1]; ++e) rsval[e] = d * rval[e] * dc[rcol[e]];
            double a = lo[r], bb = hi[r];
            if (a != a) a = -__builtin_inf();
            if (bb != bb) bb = __builtin_inf();
            loh[r] = a * d; hih[r] = bb * d;
            const double yh = yv[r] / d;
            ys[r] = yh; y0s[r] = yh; yts[r] = yh;
        }
        for (int j = tid; j < nb; j += kEcpThreads) {
            const double d = dc[j];
            for (int e = cptr[j]; e < cptr[j + 1]; ++e) csval[e] = d * cval[e] * dr[crow[e]];
            ch[j] = cc[j] * d; lh[j] = ll[j] / d; uh[j] = uu[j] / d;
            const double xh = clampd(xg[j] / d, ll[j] / d, uu[j] / d);
            xs[j] = xh; x0s[j] = xh; xts[j] = xh;
        }
        __syncthreads();

        // ============================================================ sigma_max: 20 power passes (hashed start vector) ==
        double smax = 0.0;
        {
            double a1[1];
            a1[0] = 0.0;
            for (int j = tid; j < nb; j += kEcpThreads) {
                uint64_t h = (uint64_t)(c0 + j) * 0x9E3779B97F4A7C15ULL + 0xD1B54A32D192ED03ULL;
                h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ULL; h ^= h >> 32;
                const double v = 0.25 + (double)(h >> 11) * (1.0 / 9007199254740992.0);
                xts[j] = v; a1[0] += v * v;
            }
            ecp_reduce<1>(a1, 1, red, q);
            double nrm = sqrt(q[0]);
            __syncthreads();
            for (int j = tid; j < nb; j += kEcpThreads) xts[j] = nrm > 0.0 ? xts[j] / nrm : 0.0;
            __syncthreads();
            const int npow = B.power_passes;
            for (int pass = 0; pass < npow; ++pass) {
                ecp_spmv<4, 4>(M, rptr, rcol, rsval, xts, [&](int) { return EcpPre{0.0, 0.0, 0.0}; }, [&](int r, double acc, const EcpPre&) { yts[r] = acc; });
                __syncthreads();
                a1[0] = 0.0;
                // A'(A v) into icnt-free scratch: reuse ch? no -- write into x0s? x0s holds the anchor: use the LDS vector xts after the
                // products are taken from yts only (columns read yts, write xts: no hazard)
                ecp_spmv<4, 4>(nb, cptr, crow, csval, yts, [&](int) { return EcpPre{0.0, 0.0, 0.0}; },
                               [&](int j, double acc, const EcpPre&) { xts[j] = acc; a1[0] += acc * acc; });
                ecp_reduce<1>(a1, 1, red, q);
                nrm = sqrt(q[0]);
                __syncthreads();
                if (pass < npow - 1) for (int j = tid; j < nb; j += kEcpThreads) xts[j] = nrm > 0.0 ? xts[j] / nrm : 0.0;
                __syncthreads();
            }
            smax = sqrt(nrm);                    // ||A'A v|| with ||v|| = 1 -> sigma_max^2
            // restore the iterates the passes overwrote
            for (int j = tid; j < nb; j += kEcpThreads) xts[j] = xs[j];
            for (int r = tid; r < M; r += kEcpThreads) yts[r] = ys[r];
            __syncthreads();
        }
        const double eta_safe = 0.998;
        double eta = fmax(0.998 / fmax(smax, 1e-12), eta_safe);

        // ============================================================ LP: restarted reflected-Halpern PDHG =============
        double v2[2] = {0.0, 0.0};
        for (int j = tid; j < nb; j += kEcpThreads) v2[0] += ch[j] * ch[j];
        for (int r = tid; r < M; r += kEcpThreads) { const double a = loh[r], bb = hih[r]; if (isfinite(a)) v2[1] += a * a; if (isfinite(bb)) v2[1] += bb * bb; }
        ecp_reduce<2>(v2, 2, red, q);
        const double nc2 = q[0], nb2 = q[1];
        __syncthreads();
        const double omega_ref = (nc2 > 0.0 && nb2 > 0.0) ? sqrt(nc2 / nb2) : 1.0;
        const double dres_scale = 1.0 + sqrt(nc2);
        double om = om_keep > 0.0 ? om_keep : omega_ref;
        int k = 0, it = 0;
        // st: 0 rr0, 1 r_prev, 2 r_last, 3-5 primal objective history, 6-8 row violation history, 9 stall, 10 flat_rows, 11 consolidations
        if (tid == 0) { st[0] = 0.0; st[1] = 0.0; st[2] = 0.0; st[3] = 1e300; st[4] = -1e300; st[5] = 1e300; st[6] = 1e300; st[7] = -1e300; st[8] = 1e300;
                        st[9] = 0.0; st[10] = 0.0; st[11] = 0.0; }
        bool plain_next = false, near_conv = false;
        int lp_status = KTN_STATUS_USERLIMIT;
        double pobj = 0.0;
        const int plain_len = B.check_every - 1;
        const double stall_accept = (tol_p > floor_p * (1.0 + 1e-9)) ? 10.0 : 3.0;
        while (it < B.lp_max_iter) {
            const double tau = eta / om, sigma = eta * om, inv_sigma = 1.0 / sigma;
            if (plain_next) {
                plain_next = false;
                int want = (k <= 1) ? min(31, plain_len) : plain_len;
                if (near_conv && B.near_chunk > 0) want = min(want, B.near_chunk);
                const int np = min(want, B.lp_max_iter - it);
                for (int s = 0; s < np; ++s) {
                    const double w = (double)(k + s + 1) / (double)(k + s + 2);
                    ecp_spmv<4, 4>(nb, cptr, crow, csval, ys, [&](int j) { return EcpPre{ch[j], lh[j], uh[j]}; },
                                   [&](int j, double acc, const EcpPre& p) {
                        const double xv = xs[j];
                        const double xtv = clampd(xv - tau * (p.a - acc), p.b, p.c);
                        xts[j] = 2.0 * xtv - xv;
                        xs[j] = w * (2.0 * xtv - xv) + (1.0 - w) * x0s[j];
                    });
                    __syncthreads();
                    ecp_spmv<4, 4>(M, rptr, rcol, rsval, xts, [&](int r) { return EcpPre{loh[r], hih[r], 0.0}; },
                                   [&](int r, double acc, const EcpPre& p) {
                        const double y1 = ys[r];
                        const double v = y1 - sigma * acc;
                        const double ytv = v + sigma * clampd(-v * inv_sigma, p.a, p.b);
                        ys[r] = w * (2.0 * ytv - y1) + (1.0 - w) * y0s[r];
                    });
                    __syncthreads();
                }
                k += np; it += np;
                continue;
            }
            double a[kEcpQ];
#pragma unroll
            for (int i = 0; i < kEcpQ; ++i) a[i] = 0.0;
            ecp_spmv<4, 4>(nb, cptr, crow, csval, ys, [&](int j) { return EcpPre{ch[j], lh[j], uh[j]}; },
                           [&](int j, double acc, const EcpPre& p) { xts[j] = clampd(xs[j] - tau * (p.a - acc), p.b, p.c); });
            __syncthreads();
            {
                const int lane = tid & 3;
                for (int r = tid / 4; r < M; r += kEcpThreads / 4) {
                    double axt = 0.0, axk = 0.0;
                    for (int e = rptr[r] + lane; e < rptr[r + 1]; e += 4) { const int cl = rcol[e]; const double vv = rsval[e]; axt += vv * xts[cl]; axk += vv * xs[cl]; }
                    axt = group_sum<4>(axt); axk = group_sum<4>(axk);
                    if (lane == 0) {
                        const double loi = loh[r], hii = hih[r], y1 = ys[r];
                        const double v = y1 - sigma * (2.0 * axt - axk);
                        const double ytv = v + sigma * clampd(-v / sigma, loi, hii);
                        yts[r] = ytv;
                        axv[r] = axt;
                        const double dy = ytv - y1;
                        a[0] += dy * (axt - axk); a[1] += dy * dy;
                        if (ytv > 0.0) { if (loi > -__builtin_inf()) a[2] += loi * ytv; }
                        else if (ytv < 0.0) { if (hii < __builtin_inf()) a[2] += hii * ytv; }
                        const double d0 = ytv - y0s[r];
                        a[3] += d0 * d0; a[9] += ytv * ytv;
                        a[10] = fmax(a[10], fmax(fmax(loi - axt, axt - hii), 0.0) / dr[r]);
                    }
                }
            }
            __syncthreads();
            ecp_spmv<4, 4>(nb, cptr, crow, csval, yts, [&](int j) { return EcpPre{ch[j], lh[j], uh[j]}; },
                           [&](int j, double aty, const EcpPre& p) {
                const double xtv = xts[j], cj = p.a, lj = p.b, uj = p.c;
                const double dx = xtv - xs[j];
                a[4] += dx * dx; a[5] += cj * xtv;
                const double rc = cj - aty;
                double bad = 0.0;
                if (rc > 0.0) { if (isfinite(lj)) a[6] += lj * rc; else bad = rc; }
                else if (rc < 0.0) { if (isfinite(uj)) a[6] += uj * rc; else bad = -rc; }
                const double d0 = xtv - x0s[j];
                a[7] += d0 * d0; a[8] += xtv * xtv;
                a[11] = fmax(a[11], bad / dc[j]);
            });
            ecp_reduce<kEcpQ>(a, 10, red, q);
            if (tid == 0) {
                const double dyAdx = q[0], dy2 = q[1], dy0sq = q[3], dx2 = q[4], dx0sq = q[7], xt2 = q[8], yt2 = q[9];
                const double po = q[5], dobj = q[2] + q[6], pviol = q[10], dres = q[11];
                double om1 = om, eta1 = eta;
                const double r2 = om1 / eta1 * dx2 - 2.0 * dyAdx + dy2 / (eta1 * om1);
                const double r = sqrt(fmax(r2, 0.0));
                const double gap = fabs(po - dobj) / (1.0 + fabs(po) + fabs(dobj));
                if (k == 0) { st[0] = r; st[1] = r; }
                const bool dres_ok = dres <= tol_g * dres_scale;
                bool done = (pviol <= tol_p) && (gap <= tol_g) && dres_ok;
                const bool near = (pviol <= 4.0 * tol_p) && (gap <= 4.0 * tol_g) && (dres <= 4.0 * tol_g * dres_scale);
                if (B.stag_factor > 0.0 && !done) {
                    const double f = 0.1 * tol_g * (1.0 + fabs(po));
                    const bool flat = fabs(po - st[3]) <= f && fabs(po - st[4]) <= f && fabs(po - st[5]) <= f;
                    const bool plateau = pviol <= stall_accept * tol_p && fabs(pviol - st[6]) <= 0.02 * pviol &&
                                         fabs(pviol - st[7]) <= 0.02 * pviol && fabs(pviol - st[8]) <= 0.02 * pviol;
                    if (flat && (pviol <= tol_p || plateau) && gap <= B.stag_factor * tol_g && dres_ok) done = true;
                    if (!done && gap <= tol_g && dres_ok && plateau) done = true;
                }
                st[8] = st[7]; st[7] = st[6]; st[6] = pviol; st[5] = st[4]; st[4] = st[3]; st[3] = po;
                int action = 0;
                if (done) action = 2;
                else if (!(r == r)) action = 3;
                else {
                    const double rr0 = st[0], r_prev = st[1], r_last = st[2];
                    bool restart = k > 0 && (r <= 0.2 * rr0 || (r <= 0.8 * rr0 && r > r_prev) || (double)k >= 0.36 * (double)(it + 1));
                    if (k > 0 && eta1 > eta_safe * (1.0 + 1e-12)) {
                        const int stall = (r2 < 0.0 || (r_last > 0.0 && r > 0.97 * r_last && r < 1.03 * r_last)) ? (int)st[9] + 1 : 0;
                        st[9] = (double)stall;
                        if (stall >= 3 || r2 < 0.0) { eta1 = fmax(eta_safe, 0.85 * eta1); st[9] = 0.0; restart = true; }
                    }
                    // objective converged, rows not, residual flat: multiplier mass idles between near-parallel cuts of one NL
                    // row (kernels.hpp k_consolidate): move it onto the tightest cut and restart there
                    bool consolidate = false;
                    if (k > 0 && M > m_lin && gap <= tol_g && dres_ok && pviol > tol_p) {
                        const int flat_rows = (r_last > 0.0 && r > 0.98 * r_last) ? (int)st[10] + 1 : 0;
                        st[10] = (double)flat_rows;
                        if (flat_rows >= 3 && st[11] < 8.0) { st[10] = 0.0; st[11] += 1.0; consolidate = true; }
                    } else {
                        st[10] = 0.0;
                    }
                    st[2] = r; st[1] = r;
                    if (consolidate) { action = 4; st[2] = 0.0; }
                    else if (restart) {
                        const double dx = sqrt(dx0sq), dy = sqrt(dy0sq);
                        if (dx > 1e-8 * (1.0 + sqrt(xt2)) && dy > 1e-8 * (1.0 + sqrt(yt2))) {
                            om1 = exp(0.5 * log(dy / dx) + 0.5 * log(om1));
                            om1 = fmin(fmax(om1, omega_ref * 1e-3), omega_ref * 1e3);
                        }
                        action = 1;
                    }
                }
                ctl[0] = (double)action; ctl[1] = om1; ctl[2] = eta1; ctl[3] = near ? 1.0 : 0.0; ctl[4] = po;
            }
            __syncthreads();
            const int action = (int)ctl[0];
            om = ctl[1]; eta = ctl[2]; near_conv = ctl[3] != 0.0; pobj = ctl[4];
            __syncthreads();
            ++it;
            if (action == 2) { lp_status = KTN_STATUS_OPTIMAL; break; }
            if (action == 3) { lp_status = KTN_STATUS_ERROR; break; }
            if (action == 4) {
                // one thread per NL slot walks the slot's cuts (newest first): unscaled activity a'x = ax / dr, unscaled multiplier y dr
                for (int i = tid; i < m_nl; i += kEcpThreads) {
                    int best_u = -1, best_l = -1, ncuts = 0;
                    double res_u = -__builtin_inf(), res_l = -__builtin_inf();
                    for (int r = last_cut[i]; r >= 0; r = cut_prev[r]) {
                        ++ncuts;
                        const double axu = axv[r] / dr[r];
                        const double ru = axu - hi[r], rl = lo[r] - axu;
                        if (ru > res_u) { res_u = ru; best_u = r; }
                        if (rl > res_l) { res_l = rl; best_l = r; }
                    }
                    if (ncuts < 2) continue;
                    double mass_u = 0.0, mass_l = 0.0;
                    for (int r = last_cut[i]; r >= 0; r = cut_prev[r]) {
                        const double yu = yts[r] * dr[r];
                        const double axu = axv[r] / dr[r];
                        if (yu < 0.0 && best_u >= 0 && r != best_u && (axu - hi[r]) < -tol_p) { mass_u += yu; yts[r] = 0.0; }
                        else if (yu > 0.0 && best_l >= 0 && r != best_l && (lo[r] - axu) < -tol_p) { mass_l += yu; yts[r] = 0.0; }
                    }
                    if (mass_u != 0.0) yts[best_u] += mass_u / dr[best_u];
                    if (mass_l != 0.0) yts[best_l] += mass_l / dr[best_l];
                }
                __syncthreads();
            }
            if (action == 1 || action == 4) {
                for (int j = tid; j < nb; j += kEcpThreads) { const double v = xts[j]; xs[j] = v; x0s[j] = v; }
                for (int r = tid; r < M; r += kEcpThreads) { const double v = yts[r]; ys[r] = v; y0s[r] = v; }
                k = 0;
                __syncthreads();
                continue;
            }
            {
                const double w = (double)(k + 1) / (double)(k + 2);
                for (int j = tid; j < nb; j += kEcpThreads) xs[j] = w * (2.0 * xts[j] - xs[j]) + (1.0 - w) * x0s[j];
                for (int r = tid; r < M; r += kEcpThreads) ys[r] = w * (2.0 * yts[r] - ys[r]) + (1.0 - w) * y0s[r];
                ++k;
                plain_next = true;
                __syncthreads();
            }
        }
        pdhg_total += it;
        om_keep = om;
        if (lp_status != KTN_STATUS_OPTIMAL) { status = lp_status; break; }          // model.jl:261-263
        // un-scale T(z): x* (kept in LDS xs for the sweep) and the duals
        for (int j = tid; j < nb; j += kEcpThreads) { const double v = xts[j] * dc[j]; xg[j] = v; xs[j] = v; }
        for (int r = tid; r < M; r += kEcpThreads) yv[r] = yts[r] * dr[r];
        objval = pobj;
        __syncthreads();

        // ============================================================ separator sweep over the instance's NL rows ======
        // precompute! + isconstrsat (src/separators.jl:111-120): 16 lanes per row; flags and row lengths into icnt
        double mv[1] = {0.0};
        int bad_nf = 0;
        {
            const int lane = tid & 15;
            for (int i = tid / 16; i < m_nl; i += kEcpThreads / 16) {
                const int32_t gr = B.nl_rows[nl0 + i];
                const int64_t beg = B.P.rowptr[gr], end = B.P.rowptr[gr + 1];
                double g = 0.0;
                for (int64_t e = beg + lane; e < end; e += 16) {
                    const int ck = B.P.colk[e];
                    const double2 pp = B.P.pp[e];
                    double val, der;
                    atom_eval((unsigned)ck >> kKindShift, pp.x, pp.y, xs[(ck & kColMask) - (int)c0], val, der);
                    g += val;
                }
                g = group_sum<16>(g);
                if (lane == 0) {
                    g += B.P.rconst[gr];
                    const double lb = B.P.lb[gr], ub = B.P.ub[gr];
                    const bool sat = (g >= lb - f_eff) && (g <= ub + f_eff);           // NaN -> violated
                    icnt[i] = sat ? 0 : (int)(end - beg);
                    yts[i] = g;                                                         // keep g for the emit pass (m_nl <= mmax)
                    if (!sat) { double d = fmax(g - ub, lb - g); if (!(d == d)) d = __builtin_inf(); mv[0] = fmax(mv[0], d); }
                }
            }
        }
        __syncthreads();
        {
            double t1[1] = {mv[0]};
            ecp_reduce<1>(t1, 0, red, q);
            mv[0] = q[0];
        }
        __syncthreads();
        // ranks of the violated rows (serial scan over m_nl ~ 1e2 slots)
        if (tid == 0) {
            int rows = 0, nz = 0;
            for (int i = 0; i < m_nl; ++i) { const int len = icnt[i]; icnt[i] = len > 0 ? ((rows << 1) | 1) : 0; if (len > 0) { ++rows; } }
            // second pass for the entry offsets (row lengths come from the structure again)
            ctl[5] = (double)rows;
            (void)nz;
        }
        __syncthreads();
        const int V = (int)ctl[5];
        int nnzV = 0;
        if (V > 0) {
            // entry offsets: thread 0 walks the violated rows in order
            // (capacity is tested BEFORE anything is stored: the row pointers of an arena have cap_rows + 1 slots and the
            //  next instance's arena follows -- a dry pass for the entry count, the stores only when rows and entries fit)
            if (tid == 0) {
                int64_t nz = NNZ;
                for (int i = 0; i < m_nl; ++i) if (icnt[i] & 1) {
                    const int32_t gr = B.nl_rows[nl0 + i];
                    nz += B.P.rowptr[gr + 1] - B.P.rowptr[gr];
                }
                const bool fits = (M + V <= A.cap_rows) && (nz <= A.cap_nnz);
                if (fits) {
                    int at = NNZ;
                    for (int i = 0; i < m_nl; ++i) if (icnt[i] & 1) {
                        const int32_t gr = B.nl_rows[nl0 + i];
                        rptr[M + (icnt[i] >> 1)] = at;
                        at += (int)(B.P.rowptr[gr + 1] - B.P.rowptr[gr]);
                    }
                    rptr[M + V] = at;
                }
                ctl[6] = (double)(nz - NNZ);
                ctl[7] = fits ? 0.0 : 1.0;
            }
            __syncthreads();
            nnzV = (int)ctl[6];
            if (ctl[7] != 0.0) { overflow = true; status = KTN_STATUS_ERROR; break; }
            // gencut + round_coefs + _addcut (src/algorithms.jl:3-18, src/model.jl:68-79,200-207): 16 lanes per violated row
            const int lane = tid & 15;
            for (int i = tid / 16; i < m_nl; i += kEcpThreads / 16) {
                if (!(icnt[i] & 1)) continue;
                const int rnew = M + (icnt[i] >> 1);
                const int32_t gr = B.nl_rows[nl0 + i];
                const int64_t beg = B.P.rowptr[gr], end = B.P.rowptr[gr + 1];
                const int dst = rptr[rnew];
                double dot = 0.0, mx = -__builtin_inf();
                int nf = 0;
                for (int64_t e = beg + lane; e < end; e += 16) {
                    const int ck = B.P.colk[e];
                    const double2 pp = B.P.pp[e];
                    const int cl = (ck & kColMask) - (int)c0;
                    double val, der;
                    atom_eval((unsigned)ck >> kKindShift, pp.x, pp.y, xs[cl], val, der);
                    dot += xs[cl] * der;
                    mx = nanmax(mx, der);
                    nf |= !isfinite(der);
                    rcol[dst + (int)(e - beg)] = (uint16_t)cl;
                    rval[dst + (int)(e - beg)] = der;
                }
                dot = group_sum<16>(dot);
                mx = group_nanmax<16>(mx);
                nf = group_or<16>(nf);
                if (B.P.pad_zero[gr]) mx = nanmax(mx, 0.0);
                for (int64_t e = beg + lane; e < end; e += 16) {                          // round_coefs (signed max)
                    const double der = rval[dst + (int)(e - beg)];
                    if (der + B.cut_coef_rng < mx) rval[dst + (int)(e - beg)] = 0.0;
                }
                if (lane == 0) {
                    const double bconst = yts[i] - dot;
                    lo[rnew] = B.P.lb[gr] - bconst;
                    hi[rnew] = B.P.ub[gr] - bconst;
                    const int prev = last_cut[i];
                    double y0 = 0.0;
                    if (prev >= 0) { y0 = yv[prev]; yv[prev] = 0.0; }
                    yv[rnew] = y0;
                    cut_prev[rnew] = prev;
                    last_cut[i] = rnew;
                    if (nf) bad_nf = 1;
                }
            }
            {
                double t1[1] = {(double)bad_nf};
                ecp_reduce<1>(t1, 0, red, q);
                if (q[0] != 0.0) { status = KTN_STATUS_ERROR; __syncthreads(); break; }    // model.jl:69-73
            }
            M += V; NNZ += nnzV; numcuts += V;
        }
        __syncthreads();
        last_maxviol = mv[0];
        const bool sat_now = V == 0;
        if (sat_now && tol_p > floor_p * (1.0 + 1e-12)) last_maxviol = 0.0;        // inexact-LP rule: re-solve at the floor tolerance
        else allsat = sat_now;
        if (refining && mv[0] <= B.f_tol && mv[0] < best_viol) {                    // a candidate for the answer (uniform: mv is reduced)
            best_viol = mv[0]; best_obj = objval;
            for (int j = tid; j < nb; j += kEcpThreads) xbest[j] = xg[j];
        }
    }
    // ---- certificate: refine, or done
    if (!allsat && !refining) break;                                               // iteration cap
    if (status != KTN_STATUS_NONE || !(B.cert_tol > 0.0) || m_nl == 0 || B.polish_max_iter <= 0) break;
    if (refining && passes > B.polish_max_iter) break;
    {
        double d1[1] = {0.0};
        for (int i = tid; i < m_nl; i += kEcpThreads) {
            double lam = 0.0;
            for (int r = last_cut[i]; r >= 0; r = cut_prev[r]) lam += fabs(yv[r]);
            const int32_t gr = B.nl_rows[nl0 + i];
            const double lb = B.P.lb[gr], ub = B.P.ub[gr], g = yts[i];
            double v = -__builtin_inf();
            if (isfinite(ub)) v = fmax(v, g - ub);
            if (isfinite(lb)) v = fmax(v, lb - g);
            if (v >= -10.0 * B.f_tol) d1[0] += lam * v;
        }
        ecp_reduce<1>(d1, 1, red, q);
        const double D = (q[0] == q[0]) ? fmax(q[0], 0.0) : __builtin_inf();
        const double target = B.cert_tol * fmax(1.0, fabs(objval));
        __syncthreads();
        if (allsat && D <= 0.5 * target) break;                                     // the reference's objective tolerance holds
        if (!refining) {                                                            // the point that met the stop rule is the first candidate
            refining = true;
            best_viol = B.f_tol; best_obj = objval;                                 // (any later point that is strictly inside replaces it)
            for (int j = tid; j < nb; j += kEcpThreads) xbest[j] = xg[j];
        }
        f_eff = fmin(fmax(0.25 * target / D, 0.05), 0.5) * B.f_tol;
        gap_cert = 0.25 * target / (1.0 + 2.0 * fabs(objval));
        allsat = false;
        last_maxviol = 0.0;
    }
  }
    if (refining) {                                                                 // the best point that satisfies the reference's rule
        __syncthreads();
        for (int j = tid; j < nb; j += kEcpThreads) xg[j] = xbest[j];
        objval = best_obj;
        allsat = true;
        last_maxviol = best_viol;
    }
    if (status == KTN_STATUS_NONE) status = (iter >= B.iter_cap && !allsat) ? KTN_STATUS_USERLIMIT : (allsat ? KTN_STATUS_OPTIMAL : KTN_STATUS_USERLIMIT);
    if (tid == 0) {
        double* o = B.res + (int64_t)b * 8;
        o[0] = (double)status; o[1] = (double)iter; o[2] = objval; o[3] = (double)numcuts; o[4] = (double)pdhg_total;
        o[5] = last_maxviol; o[6] = (double)M; o[7] = overflow ? 1.0 : 0.0;
    }
}

}  // namespace ktn
