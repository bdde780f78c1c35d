// kernels.hpp -- CDNA4 (gfx950) device kernels of the Katana ECP engine.
//
// All arithmetic is FP64 (SURVEY.md section 7 hard part 5) and HBM/L2-bandwidth bound:
// sparse gather-multiply-reduce, no dense contraction, hence no MFMA.  Work is mapped
// as "G lanes per sparse row" with G in {4..64} a power of two chosen from the average
// row length, so that a 64-wide wavefront serves 64/G rows, consecutive lanes read
// consecutive CSR entries (coalesced) and the row reduction is a __shfl_xor butterfly
// inside the wavefront (no LDS, no barriers).
//
//   separator sweep   k_sep_eval / k_tape_eval / k_gj_stats / k_compact / k_emit
//                     == precompute! + isconstrsat + gencut(linear_oa_cut) + round_coefs
//                        + _addcut   (src/separators.jl:111-120, src/algorithms.jl:3-18,
//                                     src/model.jl:68-79,200-207,272-283)
//   LP (replaces GLPK) k_pdhg_x / k_pdhg_y / k_chk_* / scaling kernels
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

#include "../../include/katana_hip.h"
#include "types.hpp"

namespace ktn {

constexpr int kKindShift = 29;    // packed (col | kind << 29): columns < 2^29
constexpr int kColMask = (1 << kKindShift) - 1;

// ---------------------------------------------------------------- small helpers ----
template <int G>
__device__ __forceinline__ double group_sum(double v) {
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
template <int G>
__device__ __forceinline__ double group_max(double v) {
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
    return v;
}
template <int G>
__device__ __forceinline__ int group_or(int v) {
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) v |= __shfl_xor(v, off, 64);
    return v;
}
// max with NaN poisoning (Julia's maximum() propagates NaN; fmax would drop it)
__device__ __forceinline__ double nanmax(double a, double b) {
    return (a != a || b != b) ? __builtin_nan("") : (a > b ? a : b);
}
template <int G>
__device__ __forceinline__ double group_nanmax(double v) {
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) v = nanmax(v, __shfl_xor(v, off, 64));
    return v;
}
// order-independent (hence deterministic) atomic max for non-negative doubles
// The plain load in front filters the atomics: the target only grows, so a value that does not exceed what this CU last saw
// cannot change it (a stale, smaller reading merely costs an atomic that loses).  Without the filter the first sweeps of
// cfg4 -- 1e6 violated rows, one atomic each on ONE address -- took 5.7 ms instead of 0.5 ms.
__device__ __forceinline__ void atomic_max_nonneg(double* addr, double v) {
    if (v != v) v = __builtin_inf();
    if (v > 0.0) {
        const unsigned long long b = (unsigned long long)__double_as_longlong(v);
        if (b > *reinterpret_cast<const unsigned long long*>(addr)) atomicMax(reinterpret_cast<unsigned long long*>(addr), b);
    }
}
// Block-wide form for kernels in which EVERY thread of the workgroup reaches the call (no early returns): the lanes' values
// meet in an LDS cell first and one thread publishes the workgroup's maximum -- one global atomic per workgroup instead of
// one per violated row.  (cfg4's first sweeps, 1e6 violated rows: 2.6 ms -> 0.45 ms; a max is order-independent, so the
// result is the same.)
__device__ __forceinline__ void block_max_nonneg(double* addr, double v) {
    __shared__ unsigned long long s_blockmax;
    if (threadIdx.x == 0) s_blockmax = 0ull;
    __syncthreads();
    if (v != v) v = __builtin_inf();
    if (v > 0.0) atomicMax(&s_blockmax, (unsigned long long)__double_as_longlong(v));
    __syncthreads();
    if (threadIdx.x == 0 && s_blockmax != 0ull && s_blockmax > *reinterpret_cast<const unsigned long long*>(addr))
        atomicMax(reinterpret_cast<unsigned long long*>(addr), s_blockmax);
}
__device__ __forceinline__ double clampd(double v, double lo, double hi) { return fmin(fmax(v, lo), hi); }

// Six scalars a sweep / a purge hands back to the host -- the last elements of two (flag, exclusive scan) pairs, the
// largest violation and two int flags -- written by ONE thread into pinned, device-mapped host memory instead of six
// device-to-host copies (a copy kernel of ~4.5 us each).  Counts are exact in a double below 2^53.
static __global__ void k_host_tail(double* __restrict__ out, const int64_t* __restrict__ a, const int64_t* __restrict__ b,
                            const int64_t* __restrict__ c, const int64_t* __restrict__ d, const double* __restrict__ mv,
                            const int32_t* __restrict__ f0, const int32_t* __restrict__ f1) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    out[0] = (double)*a; out[1] = (double)*b; out[2] = (double)*c; out[3] = (double)*d;
    out[4] = mv ? *mv : 0.0;
    out[5] = f0 ? (double)*f0 : 0.0;
    out[6] = f1 ? (double)*f1 : 0.0;
}

// ------------------------------------------------------------- separable atoms ----
__device__ __forceinline__ void atom_eval(int kind, double a, double b, double x, double& val, double& der) {
    switch (kind) {
        case KTN_ATOM_LIN: val = a * x; der = a; break;
        case KTN_ATOM_QUAD: { double d = x - b; val = a * d * d; der = 2.0 * a * d; } break;
        case KTN_ATOM_EXP: { double e = a * exp(b * x); val = e; der = b * e; } break;
        default: { double s = x + b; val = -a * log(s); der = -a / s; } break;   // KTN_ATOM_NEGLOG
    }
}

// Device view of the (epigraph-lifted) NLP: CSR Jacobian structure + row programs.
struct NlpDev {
    const int64_t* rowptr;
    const int32_t* col;
    // separable atoms, packed for the sweep: 20 B per Jacobian entry in two coalesced streams
    const int32_t* colk;     // col | kind << 29
    const double2* pp;       // (p0, p1)
    const double* rconst;
    const uint8_t* row_kind;
    const uint8_t* pad_zero;   // row has implicit zero coefficients (dense epigraph row, src/nlpeval.jl:49-54)
    const double* lb;
    const double* ub;
    // expression DAG of the tape rows (nodes in evaluation order)
    const int64_t* node_ptr;
    const int32_t* node_op;
    const int32_t* node_a;
    const int32_t* node_b;
    const double* node_c;
    double* node_val;
    double* node_adj;
};

// Per-NL-row outputs of the evaluation stage
struct SweepOut {
    double* g;         // [m]   constraint values  (separators.jl:113)
    double* jac;       // [nnz] Jacobian values    (separators.jl:112); separable rows write it only if materialize
    double* bconst;    // [m]   cut constant b = g - sum x*_c J_c   (algorithms.jl:8,15)
    double* maxc;      // [m]   signed max coefficient (model.jl:201)
    int32_t* nonfin;   // [m]   any non-finite coefficient (model.jl:69)
    int64_t* flag;     // [m_nl] 1 if violated (by NL slot)
    int64_t* cnt;      // [m_nl] row nnz if violated else 0
    double* maxviol;   // [1]
    int32_t* any_nonfin;  // [1] some violated row has a non-finite coefficient
};

// precompute! + isconstrsat for separable rows: G lanes per row.
template <int G>
__global__ __launch_bounds__(kBlock) void k_sep_eval(NlpDev P, const int32_t* __restrict__ nl_rows, int64_t m_nl,
                                                     const double* __restrict__ x, double f_tol, int materialize,
                                                     int only_flagged_nl, SweepOut O) {
    const int64_t gid = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / G;
    const int lane = threadIdx.x & (G - 1);
    // (no early return: every thread takes part in block_max_nonneg at the end; an inactive group has an empty entry range)
    const int32_t r = gid < m_nl ? nl_rows[gid] : 0;
    const bool live = gid < m_nl && P.row_kind[r] == KTN_ROW_SEP;
    const int64_t beg = live ? P.rowptr[r] : 0, end = live ? P.rowptr[r + 1] : 0;
    double acc_g = 0.0, acc_dot = 0.0, mx = -__builtin_inf();
    int nf = 0;
    // kU entries per lane and trip: all (colk, pp) loads and all x gathers of a trip are issued
    // before any arithmetic, so each wavefront keeps kU * G * 20 B (+ gathers) in flight
    constexpr int kU = 2;
    for (int64_t e = beg + lane; e < end; e += kU * G) {
        int ck[kU];
        double2 q[kU];
        double xv[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int64_t eu = e + (int64_t)u * G;
            const bool on = eu < end;
            ck[u] = on ? P.colk[eu] : -1;
            q[u] = on ? P.pp[eu] : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) xv[u] = (ck[u] >= 0) ? x[ck[u] & kColMask] : 0.0;
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            if (ck[u] >= 0) {
                double val, der;
                atom_eval((unsigned)ck[u] >> kKindShift, q[u].x, q[u].y, xv[u], val, der);
                acc_g += val;
                acc_dot += xv[u] * der;
                mx = nanmax(mx, der);
                nf |= !isfinite(der);
                if (materialize) O.jac[e + (int64_t)u * G] = der;
            }
        }
    }
    acc_g = group_sum<G>(acc_g);
    acc_dot = group_sum<G>(acc_dot);
    mx = group_nanmax<G>(mx);
    nf = group_or<G>(nf);
    double viol = 0.0;
    if (lane == 0 && live) {
        const double g = acc_g + P.rconst[r];
        if (P.pad_zero[r]) mx = nanmax(mx, 0.0);
        O.g[r] = g;
        O.bconst[r] = g - acc_dot;
        O.maxc[r] = mx;
        O.nonfin[r] = nf;
        if (only_flagged_nl) {
            const double lb = P.lb[r], ub = P.ub[r];
            const bool sat = (g >= lb - f_tol) && (g <= ub + f_tol);   // separators.jl:120 (NaN -> violated)
            O.flag[gid] = sat ? 0 : 1;
            O.cnt[gid] = sat ? 0 : (end - beg);
            if (!sat) {
                viol = fmax(g - ub, lb - g);
                if (viol != viol) viol = __builtin_inf();
                if (nf) { if (*O.any_nonfin == 0) atomicOr(O.any_nonfin, 1); };
            }
        }
    }
    if (only_flagged_nl) block_max_nonneg(O.maxviol, viol);
}

// The sweep's form of k_sep_eval (materialize = 0, flags on) for instances with MANY SHORT rows (cfg4: 1e6 rows of 32
// entries): a lane group takes R consecutive NL slots and issues the loads of all R rows level by level -- R row
// numbers, R row records, R x (colk, pp) entry pairs, R gathers -- before any arithmetic.  With one row per group a
// wavefront has four dependent memory round trips and 64 entries to show for them, and the kernel is latency-bound at
// 4 % of the HBM peak; here the same four round trips serve R times as many entries.  Same arithmetic and the same
// summation order per row as k_sep_eval (lane-strided partial sums, xor-butterfly), hence the same bits.
// MAT = true is the literal precompute! (src/separators.jl:111-116) in the same form: the Jacobian values are stored, the
// isconstrsat tail is left out (round 4: ktn_sep_precompute on 1e6 rows of 32 entries ran one row per group, 1.86 ms).
template <int G, int R, bool MAT = false>
__global__ __launch_bounds__(kBlock) void k_sep_sweep(NlpDev P, const int32_t* __restrict__ nl_rows, int64_t m_nl,
                                                      const double* __restrict__ x, double f_tol, SweepOut O) {
    const int64_t s0 = (((int64_t)blockIdx.x * kBlock + threadIdx.x) / G) * R;
    const int lane = threadIdx.x & (G - 1);
    int32_t r[R];
    int64_t beg[R];
    int len[R];
#pragma unroll
    for (int j = 0; j < R; ++j) r[j] = (s0 + j < m_nl) ? nl_rows[s0 + j] : -1;
    int maxlen = 0;
    {
        uint8_t kd[R];
        int64_t en[R];
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const int32_t rr = r[j] >= 0 ? r[j] : 0;
            kd[j] = P.row_kind[rr]; beg[j] = P.rowptr[rr]; en[j] = P.rowptr[rr + 1];
        }
#pragma unroll
        for (int j = 0; j < R; ++j) {
            if (r[j] >= 0 && kd[j] != KTN_ROW_SEP) r[j] = -1;
            len[j] = r[j] >= 0 ? (int)(en[j] - beg[j]) : 0;
            maxlen = len[j] > maxlen ? len[j] : maxlen;
        }
    }
    double acc_g[R], acc_dot[R], mx[R];
    int nf[R];
#pragma unroll
    for (int j = 0; j < R; ++j) { acc_g[j] = 0.0; acc_dot[j] = 0.0; mx[j] = -__builtin_inf(); nf[j] = 0; }
    for (int off = lane; off < maxlen; off += G) {
        int ck[R];
        double2 q[R];
        double xv[R];
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const bool on = off < len[j];
            ck[j] = on ? P.colk[beg[j] + off] : -1;
            q[j] = on ? P.pp[beg[j] + off] : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int j = 0; j < R; ++j) xv[j] = (ck[j] >= 0) ? x[ck[j] & kColMask] : 0.0;
#pragma unroll
        for (int j = 0; j < R; ++j) {
            if (ck[j] >= 0) {
                double val, der;
                atom_eval((unsigned)ck[j] >> kKindShift, q[j].x, q[j].y, xv[j], val, der);
                acc_g[j] += val;
                acc_dot[j] += xv[j] * der;
                mx[j] = nanmax(mx[j], der);
                nf[j] |= !isfinite(der);
                if (MAT) O.jac[beg[j] + off] = der;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < R; ++j) {
        acc_g[j] = group_sum<G>(acc_g[j]);
        acc_dot[j] = group_sum<G>(acc_dot[j]);
        mx[j] = group_nanmax<G>(mx[j]);
        nf[j] = group_or<G>(nf[j]);
    }
    // lane j finishes row j (the butterflies leave every lane with the totals)
    double viol = 0.0;
#pragma unroll
    for (int j = 0; j < R; ++j) {
        if (lane == (j & (G - 1)) && r[j] >= 0) {
            const int32_t rr = r[j];
            const double g = acc_g[j] + P.rconst[rr];
            double m = mx[j];
            if (P.pad_zero[rr]) m = nanmax(m, 0.0);
            O.g[rr] = g;
            O.bconst[rr] = g - acc_dot[j];
            O.maxc[rr] = m;
            O.nonfin[rr] = nf[j];
            if (!MAT) {
                const double lb = P.lb[rr], ub = P.ub[rr];
                const bool sat = (g >= lb - f_tol) && (g <= ub + f_tol);   // separators.jl:120 (NaN -> violated)
                O.flag[s0 + j] = sat ? 0 : 1;
                O.cnt[s0 + j] = sat ? 0 : len[j];
                if (!sat) {
                    double v = fmax(g - ub, lb - g);
                    if (v != v) v = __builtin_inf();
                    viol = fmax(viol, v);
                    if (nf[j]) { if (*O.any_nonfin == 0) atomicOr(O.any_nonfin, 1); };
                }
            }
        }
    }
    if (!MAT) block_max_nonneg(O.maxviol, viol);
}

// ---- ONE VERY LONG separable row (round 4) -----------------------------------------------------------------------------
// A linear objective is stored as a row of the structure (up to n entries), a nonlinear one as the epigraph row (n + 1 entries:
// src/nlpeval.jl:49-54).  With G lanes per row ONE lane group walked it: 3 125 dependent trips for 1e5 entries -- the 1.9 ms that
// `ktn_sep_precompute` took on a 1e6-row instance were this one row, not the million short ones (profiles/r04_*), and every sweep
// of a model with a nonlinear objective paid it too.  Rows beyond kLongEval entries carry the device-side row kind kRowSepLong:
// every row kernel skips them (they test for KTN_ROW_SEP), k_emit takes their derivatives from the materialised Jacobian, and this
// kernel evaluates them, one 1 024-thread workgroup per row (thread-strided partial sums, wavefront butterflies, the block's
// wavefronts in order: a fixed summation order).
constexpr int kLongEval = 8192;
constexpr uint8_t kRowSepLong = 3;
static __global__ __launch_bounds__(1024) void k_sep_eval_long(NlpDev P, const int32_t* __restrict__ rows, const int64_t* __restrict__ slots,
                                                        const double* __restrict__ x, double f_tol, int flags_on, SweepOut O) {
    __shared__ double sh[16][3];
    __shared__ int shn[16];
    const int32_t r = rows[blockIdx.x];
    const int64_t beg = P.rowptr[r], end = P.rowptr[r + 1];
    double acc_g = 0.0, acc_dot = 0.0, mx = -__builtin_inf();
    int nf = 0;
    for (int64_t e = beg + threadIdx.x; e < end; e += 1024) {
        const int ck = P.colk[e];
        const double2 q = P.pp[e];
        const double xv = x[ck & kColMask];
        double val, der;
        atom_eval((unsigned)ck >> kKindShift, q.x, q.y, xv, val, der);
        acc_g += val; acc_dot += xv * der; mx = nanmax(mx, der); nf |= !isfinite(der);
        O.jac[e] = der;
    }
    acc_g = group_sum<64>(acc_g); acc_dot = group_sum<64>(acc_dot); mx = group_nanmax<64>(mx); nf = group_or<64>(nf);
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sh[wv][0] = acc_g; sh[wv][1] = acc_dot; sh[wv][2] = mx; shn[wv] = nf; }
    __syncthreads();
    if (threadIdx.x != 0) return;
    for (int k = 1; k < 16; ++k) { acc_g += sh[k][0]; acc_dot += sh[k][1]; mx = nanmax(mx, sh[k][2]); nf |= shn[k]; }
    const double g = acc_g + P.rconst[r];
    if (P.pad_zero[r]) mx = nanmax(mx, 0.0);
    O.g[r] = g; O.bconst[r] = g - acc_dot; O.maxc[r] = mx; O.nonfin[r] = nf;
    const int64_t slot = slots[blockIdx.x];
    if (flags_on && slot >= 0) {
        const double lb = P.lb[r], ub = P.ub[r];
        const bool sat = (g >= lb - f_tol) && (g <= ub + f_tol);   // separators.jl:120 (NaN -> violated)
        O.flag[slot] = sat ? 0 : 1;
        O.cnt[slot] = sat ? 0 : (end - beg);
        if (!sat) {
            double v = fmax(g - ub, lb - g);
            if (v != v) v = __builtin_inf();
            atomicMax(reinterpret_cast<unsigned long long*>(O.maxviol), (unsigned long long)__double_as_longlong(v));
            if (nf) atomicOr(O.any_nonfin, 1);
        }
    }
}

// ---- batch-blocked sweep for MANY SHORT rows (round 4; cfg4: 1e6 rows of 32 entries) ---------------
// The row kernel above gathers x*[col] from L2 once per ENTRY: 3.2e7 eight-byte gathers, each its own L1 miss that drags a
// 128-byte line -- 4 GB of L2->L1 traffic per sweep, which is what bounds it at 19 % of the HBM peak.  A short row cannot be
// split over column blocks the way k_sep_eval_blk splits long ones (a 32-byte partial per 2.5 entries), and a thread-per-row
// form with the block of x* in LDS (round 3) ends up FP64-bound at 2.4x the useful work, because exp and log atoms mix inside
// every wavefront and every lane runs the longest row's entry slots.  This form keeps what both had right:
//   * a workgroup owns a BATCH of kSbRows consecutive NL slots and keeps their accumulators (g, x.grad g, max, non-finite) in
//     LDS for the whole pass;
//   * the batch's entries are regrouped at load time block-major (blocks of kSbCols columns) and, inside a (batch, block)
//     unit, by atom kind and then by row: 20 B per entry -- 16-bit local column, 16-bit local row, two parameters;
//   * per unit the workgroup stages 64 KB of x* in LDS ONCE (13 B of L2 traffic per entry instead of 128) and evaluates the
//     unit's entries ENTRY-parallel in chunks of 1 024: every wavefront runs exactly one atom kind on 64 useful entries;
//   * the per-entry results go through an LDS scratch; the first lane of every run of equal rows (runs are contiguous: sorted)
//     adds its run up in storage order and updates the row's accumulator.  Chunks never mix kinds, so a row has at most one
//     run per chunk and one lane owns it: no atomics, a fixed summation order, bitwise reproducible sweeps.
constexpr int kSbRows = 2048;
constexpr int kSbCols = 8192;
constexpr int kSbThreads = 1024;
constexpr size_t kSbLds = (size_t)kSbCols * 8 + (size_t)kSbRows * (3 * 8 + 4) + (size_t)kSbThreads * (3 * 8 + 2);
struct SbView {
    const uint16_t* ck;      // local column
    const uint16_t* row;     // local row (NL slot - batch * kSbRows)
    const double2* pp;       // (p0, p1)
    const int64_t* seg;      // [(batches * nb + 1) * 4]: first entry of (batch, block, kind); the last slot closes the list
    int nb;                  // column blocks
};
static __global__ __launch_bounds__(kSbThreads) void k_sep_sweep_batch(SbView V, NlpDev P, const int32_t* __restrict__ nl_rows, int64_t m_nl,
                                                                const double* __restrict__ x, int64_t n_x, double f_tol, SweepOut O) {
    extern __shared__ double sb_sm[];
    double* xs = sb_sm;
    double* acc_g = xs + kSbCols;
    double* acc_d = acc_g + kSbRows;
    double* acc_m = acc_d + kSbRows;
    double* sval = acc_m + kSbRows;
    double* sxd = sval + kSbThreads;
    double* sder = sxd + kSbThreads;
    int32_t* acc_nf = reinterpret_cast<int32_t*>(sder + kSbThreads);
    uint16_t* srow = reinterpret_cast<uint16_t*>(acc_nf + kSbRows);
    const int tid = threadIdx.x;
    const int64_t batch = blockIdx.x, s0 = batch * kSbRows;
    const int nrows = (int)((m_nl - s0) < kSbRows ? (m_nl - s0) : kSbRows);
    __shared__ int64_t sseg[64 * 4 + 1];                           // the batch's segment table (nb <= 64 blocks)
    for (int i = tid; i < kSbRows; i += kSbThreads) { acc_g[i] = 0.0; acc_d[i] = 0.0; acc_m[i] = -__builtin_inf(); acc_nf[i] = 0; }
    const int nseg = V.nb * 4;
    for (int i = tid; i <= nseg; i += kSbThreads) sseg[i] = V.seg[batch * nseg + i];
    __syncthreads();
    // The batch's chunks -- at most kSbThreads entries of ONE (block, kind) segment each -- form one stream; the loads of the
    // next chunk are issued (into registers) before the current one is evaluated, so the HBM latency of the 20 KB a chunk
    // reads is covered by a chunk's worth of FP64 work instead of being paid 78 times per batch.  (Measured on cfg4's sweep,
    // tools/sweep_ab.py: plain 475 us, with this prefetch 398, with the next block of x* requested a chunk ahead 380; letting
    // the kinds of a unit share chunks -- 17 % fewer chunks, heads added up kind after kind -- 469: the per-lane kind costs
    // more than the fuller chunks save.  Row kernel: 451.)
    struct Cur { int sg; int64_t cb, e1; };                         // segment index (block * 4 + kind), chunk start, segment end
    auto settle = [&](Cur& c) {                                     // skip empty segments
        while (c.sg < nseg && c.cb >= c.e1) { ++c.sg; if (c.sg < nseg) { c.cb = sseg[c.sg]; c.e1 = sseg[c.sg + 1]; } }
    };
    Cur cur{0, sseg[0], sseg[1]}, nxt;
    settle(cur);
    int nck = 0; uint16_t nrw = 0xFFFF; double2 npp = make_double2(0.0, 0.0);
    auto fetch = [&](const Cur& c) {
        const int64_t e = c.cb + tid;
        const bool live = c.sg < nseg && e < c.e1;
        nck = live ? (int)V.ck[e] : 0;
        nrw = live ? V.row[e] : (uint16_t)0xFFFF;
        npp = live ? V.pp[e] : make_double2(0.0, 0.0);
    };
    fetch(cur);
    int staged = -1;
    double xpre[kSbCols / kSbThreads];                              // the NEXT block's slice of x*, requested one chunk ahead
    int xpre_b = -1;
    auto xfetch = [&](int b) {
        const int64_t c0 = (int64_t)b * kSbCols;
#pragma unroll
        for (int i = 0; i < kSbCols / kSbThreads; ++i) {
            const int64_t c = c0 + i * kSbThreads + tid;
            xpre[i] = c < n_x ? x[c] : 0.0;
        }
        xpre_b = b;
    };
    if (cur.sg < nseg) xfetch(cur.sg >> 2);
    while (cur.sg < nseg) {
        const int b = cur.sg >> 2, kind = cur.sg & 3;
        if (b != staged) {                                          // a new column block: its 64 KB of x* into LDS, once
            if (xpre_b != b) xfetch(b);
            __syncthreads();
#pragma unroll
            for (int i = 0; i < kSbCols / kSbThreads; ++i) xs[i * kSbThreads + tid] = xpre[i];
            staged = b;
            __syncthreads();
        }
        const int ck = nck; const uint16_t rw = nrw; const double2 pp = npp;
        const bool live = rw != (uint16_t)0xFFFF;
        nxt = cur; nxt.cb += kSbThreads; settle(nxt);
        fetch(nxt);                                                 // the next chunk's loads are in flight while this one is evaluated
        if (nxt.sg < nseg && (nxt.sg >> 2) != b) xfetch(nxt.sg >> 2);   // ... and, on a unit's last chunk, the next block of x*
        const double xv = xs[ck];
        double val = 0.0, der = 0.0;
        if (live) {
            switch (kind) {                                         // (uniform over the workgroup: a chunk holds one kind)
                case KTN_ATOM_LIN: atom_eval(KTN_ATOM_LIN, pp.x, pp.y, xv, val, der); break;
                case KTN_ATOM_QUAD: atom_eval(KTN_ATOM_QUAD, pp.x, pp.y, xv, val, der); break;
                case KTN_ATOM_EXP: atom_eval(KTN_ATOM_EXP, pp.x, pp.y, xv, val, der); break;
                default: atom_eval(KTN_ATOM_NEGLOG, pp.x, pp.y, xv, val, der); break;
            }
        }
        sval[tid] = val; sxd[tid] = xv * der; sder[tid] = der; srow[tid] = rw;
        __syncthreads();
        if (live && (tid == 0 || srow[tid - 1] != rw)) {            // head of a run of equal rows: add the run up in storage order
            double g = 0.0, d = 0.0, mx = -__builtin_inf();
            int nf = 0;
            for (int j = tid; j < kSbThreads && srow[j] == rw; ++j) {
                g += sval[j]; d += sxd[j]; mx = nanmax(mx, sder[j]); nf |= !isfinite(sder[j]);
            }
            acc_g[rw] += g; acc_d[rw] += d; acc_m[rw] = nanmax(acc_m[rw], mx); acc_nf[rw] |= nf;
        }
        __syncthreads();
        cur = nxt;
    }
    __syncthreads();
    // the isconstrsat tail of k_sep_sweep, one thread per row of the batch
    double viol = 0.0;
    for (int i = tid; i < nrows; i += kSbThreads) {
        const int32_t rr = nl_rows[s0 + i];
        if (P.row_kind[rr] != KTN_ROW_SEP) continue;
        const double g = acc_g[i] + P.rconst[rr];
        double m = acc_m[i];
        if (P.pad_zero[rr]) m = nanmax(m, 0.0);
        const int nf = acc_nf[i];
        O.g[rr] = g;
        O.bconst[rr] = g - acc_d[i];
        O.maxc[rr] = m;
        O.nonfin[rr] = nf;
        const double lb = P.lb[rr], ub = P.ub[rr];
        const bool sat = (g >= lb - f_tol) && (g <= ub + f_tol);   // separators.jl:120 (NaN -> violated)
        O.flag[s0 + i] = sat ? 0 : 1;
        O.cnt[s0 + i] = sat ? 0 : (P.rowptr[rr + 1] - P.rowptr[rr]);
        if (!sat) {
            double v = fmax(g - ub, lb - g);
            if (v != v) v = __builtin_inf();
            viol = fmax(viol, v);
            if (nf) { if (*O.any_nonfin == 0) atomicOr(O.any_nonfin, 1); };
        }
    }
    block_max_nonneg(O.maxviol, viol);
}

// ---- column-blocked evaluation for LONG rows (HBM-resident Jacobians) ------------------
// k_sep_eval gathers x*[col] from L2: with thousands of entries per row every 8-byte gather pulls its own 128-byte
// line into the L1 and the kernel is bound by that traffic at a third of the HBM peak.  For such instances the
// engine keeps a second, BLOCK-MAJOR copy of the packed Jacobian: for each block of kBlkCols columns, the entries
// of all rows that fall into the block, row after row.  A workgroup owns (block b, a tile of rows): it stages
// x*[b*BC .. (b+1)*BC) in LDS once, streams one contiguous run of entries and gathers from LDS.  Per (row, block)
// partials are combined in block order by k_sep_combine, so every sum keeps a fixed order (bitwise reproducible).
struct SepPartial { double g, dot, mx, nf; };

// one kind-uniform run [beg, end) of a (row, block) segment; x* comes from the LDS copy of the block
template <int G, int KIND, int kU>
__device__ __forceinline__ void blk_run(const int32_t* __restrict__ bcolk, const double2* __restrict__ bpp, int64_t beg,
                                        int64_t end, int lane, const double* xs, int64_t c0, double& acc_g, double& acc_dot,
                                        double& mx, int& nf) {
    for (int64_t e = beg + lane; e < end; e += kU * G) {
        int ck[kU];
        double2 q[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int64_t eu = e + (int64_t)u * G;
            const bool on = eu < end;
            ck[u] = on ? bcolk[eu] : -1;
            q[u] = on ? bpp[eu] : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            if (ck[u] >= 0) {
                const double xv = xs[(ck[u] & kColMask) - c0];
                double val, der;
                atom_eval(KIND, q[u].x, q[u].y, xv, val, der);
                acc_g += val;
                acc_dot += xv * der;
                mx = nanmax(mx, der);
                nf |= !isfinite(der);
            }
        }
    }
}

// Persistent grid (as many workgroups as fit the chip at once): the (block, row-tile) units are dealt out in contiguous,
// equal chunks in block-major order, so the chip is balanced to within one tile whatever m_nl and n are, and a
// workgroup reloads its x* block at most twice.
template <int G, int BC, int BS, int kU>
__global__ __launch_bounds__(BS) void k_sep_eval_blk(const int32_t* __restrict__ bcolk, const double2* __restrict__ bpp,
                                                     const int64_t* __restrict__ bseg, const int4* __restrict__ bkind,
                                                     int64_t m_nl, int NB, const double* __restrict__ x, int64_t n,
                                                     SepPartial* __restrict__ part) {
    __shared__ double xs[BC];
    const int grp = threadIdx.x / G, lane = threadIdx.x & (G - 1);
    constexpr int NG = BS / G;
    const int64_t tiles = (m_nl + NG - 1) / NG;
    const int64_t units = tiles * NB;
    const int64_t per = (units + gridDim.x - 1) / gridDim.x;
    const int64_t u0 = (int64_t)blockIdx.x * per;
    const int64_t u1 = (u0 + per < units) ? u0 + per : units;
    int cur_b = -1;
    int64_t c0 = 0;
    for (int64_t u = u0; u < u1; ++u) {
        const int b = (int)(u / tiles);
        const int64_t s = (u - (int64_t)b * tiles) * NG + grp;     // consecutive groups: consecutive rows, adjacent segments
        if (b != cur_b) {                                          // (uniform over the workgroup)
            __syncthreads();
            c0 = (int64_t)b * BC;
            for (int i = threadIdx.x; i < BC; i += BS) xs[i] = (c0 + i < n) ? x[c0 + i] : 0.0;
            __syncthreads();
            cur_b = b;
        }
        if (s >= m_nl) continue;
        const int64_t* sp = bseg + (int64_t)b * (m_nl + 1);
        const int64_t beg = sp[s], end = sp[s + 1];
        const int4 kb = bkind[(int64_t)b * (m_nl + 1) + s];   // the segment is sorted by atom kind: starts of the QUAD / EXP / NEGLOG runs
        double acc_g = 0.0, acc_dot = 0.0, mx = -__builtin_inf();
        int nf = 0;
        blk_run<G, KTN_ATOM_LIN, kU>(bcolk, bpp, beg, beg + kb.x, lane, xs, c0, acc_g, acc_dot, mx, nf);
        blk_run<G, KTN_ATOM_QUAD, kU>(bcolk, bpp, beg + kb.x, beg + kb.y, lane, xs, c0, acc_g, acc_dot, mx, nf);
        blk_run<G, KTN_ATOM_EXP, kU>(bcolk, bpp, beg + kb.y, beg + kb.z, lane, xs, c0, acc_g, acc_dot, mx, nf);
        blk_run<G, KTN_ATOM_NEGLOG, kU>(bcolk, bpp, beg + kb.z, end, lane, xs, c0, acc_g, acc_dot, mx, nf);
        acc_g = group_sum<G>(acc_g);
        acc_dot = group_sum<G>(acc_dot);
        mx = group_nanmax<G>(mx);
        nf = group_or<G>(nf);
        if (lane == 0) {
            SepPartial o;
            o.g = acc_g; o.dot = acc_dot; o.mx = mx; o.nf = nf ? 1.0 : 0.0;
            part[(int64_t)b * m_nl + s] = o;             // [block][row]: k_sep_combine reads it coalesced
        }
    }
}

// block-order combination of the partials + the isconstrsat tail of k_sep_eval.  Everything a slot needs sits in one
// 32-byte record (no nl_rows -> row_kind -> bounds pointer chase: the kernel is 10 000 threads of pure latency).
struct SepSlot { double rconst, lb, ub; int32_t row; int32_t len_pad; };   // len_pad = row length << 1 | pad_zero; row < 0: not separable

static __global__ __launch_bounds__(kBlock) void k_sep_combine(const SepSlot* __restrict__ slots, int64_t m_nl, int NB,
                                                        const SepPartial* __restrict__ part, double f_tol, SweepOut O) {
    const int64_t s = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (s >= m_nl) return;
    const SepSlot sl = slots[s];
    if (sl.row < 0) return;
    double acc_g = 0.0, acc_dot = 0.0, mx = -__builtin_inf();
    int nf = 0;
#pragma unroll 8
    for (int b = 0; b < NB; ++b) {
        const SepPartial q = part[(int64_t)b * m_nl + s];
        acc_g += q.g;
        acc_dot += q.dot;
        mx = nanmax(mx, q.mx);
        nf |= (q.nf != 0.0);
    }
    const int32_t r = sl.row;
    const double g = acc_g + sl.rconst;
    if (sl.len_pad & 1) mx = nanmax(mx, 0.0);
    O.g[r] = g;
    O.bconst[r] = g - acc_dot;
    O.maxc[r] = mx;
    O.nonfin[r] = nf;
    const bool sat = (g >= sl.lb - f_tol) && (g <= sl.ub + f_tol);   // separators.jl:120 (NaN -> violated)
    O.flag[s] = sat ? 0 : 1;
    O.cnt[s] = sat ? 0 : (int64_t)(sl.len_pad >> 1);
    if (!sat) {
        atomic_max_nonneg(O.maxviol, fmax(g - sl.ub, sl.lb - g));
        if (nf) { if (*O.any_nonfin == 0) atomicOr(O.any_nonfin, 1); };
    }
}

// ---- deepest-cut selection (cut_cap): depth keys of the violated rows, then re-flagging against the threshold ----
// key = bit pattern of the violation depth max(g - ub, lb - g) (non-negative doubles order like unsigned integers;
// NaN counts as +inf), 0 for satisfied rows.
static __global__ __launch_bounds__(kBlock) void k_depth_keys(NlpDev P, const int32_t* __restrict__ nl_rows, int64_t m_nl,
                                                       const double* __restrict__ g, const int64_t* __restrict__ flag,
                                                       uint64_t* __restrict__ keys) {
    const int64_t s = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (s >= m_nl) return;
    uint64_t k = 0;
    if (flag[s]) {
        const int32_t r = nl_rows[s];
        double d = fmax(g[r] - P.ub[r], P.lb[r] - g[r]);
        if (!(d == d)) d = __builtin_inf();
        k = (d > 0.0) ? (uint64_t)__double_as_longlong(d) : 1ULL;     // violated within f_tol slack only: smallest key
    }
    keys[s] = k;
}
static __global__ __launch_bounds__(kBlock) void k_depth_reflag(int64_t m_nl, const uint64_t* __restrict__ keys,
                                                         const uint64_t* __restrict__ sorted_desc, int64_t keep,
                                                         int64_t* __restrict__ flag, int64_t* __restrict__ cnt) {
    const int64_t s = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (s >= m_nl) return;
    const uint64_t thr = sorted_desc[keep - 1];
    if (flag[s] && keys[s] < thr) { flag[s] = 0; cnt[s] = 0; }
}

// precompute! for tape rows: one thread per row, forward sweep then reverse sweep over
// the row's expression DAG.  Derivative conventions follow the oracle (oracle/sexpr.py):
// log' = 1/v, sqrt' = 0.5/sqrt(v), pow: 2 -> 2v, 1 -> 1, else p v^(p-1).
static __global__ __launch_bounds__(kBlock) void k_tape_eval(NlpDev P, const int32_t* __restrict__ tape_rows, int64_t n_tape,
                                                      const double* __restrict__ x, SweepOut O) {
    const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (t >= n_tape) return;
    const int32_t r = tape_rows[t];
    const int64_t nb = P.node_ptr[r], ne = P.node_ptr[r + 1];
    for (int64_t e = P.rowptr[r]; e < P.rowptr[r + 1]; ++e) O.jac[e] = 0.0;
    if (ne == nb) { O.g[r] = P.rconst[r]; return; }
    double* val = P.node_val;
    double* adj = P.node_adj;
    for (int64_t i = nb; i < ne; ++i) {
        const int op = P.node_op[i];
        const double a = (op >= KTN_OP_ADD) ? val[nb + P.node_a[i]] : 0.0;
        const double b = (op >= KTN_OP_ADD && op <= KTN_OP_DIV) ? val[nb + P.node_b[i]] : 0.0;
        double v;
        switch (op) {
            case KTN_OP_CONST: v = P.node_c[i]; break;
            case KTN_OP_VAR: v = x[P.node_a[i]]; break;
            case KTN_OP_ADD: v = a + b; break;
            case KTN_OP_SUB: v = a - b; break;
            case KTN_OP_MUL: v = a * b; break;
            case KTN_OP_DIV: v = a / b; break;
            case KTN_OP_NEG: v = -a; break;
            case KTN_OP_POWC: v = pow(a, P.node_c[i]); break;
            case KTN_OP_EXP: v = exp(a); break;
            case KTN_OP_LOG: v = log(a); break;
            case KTN_OP_SQRT: v = sqrt(a); break;
            case KTN_OP_SIN: v = sin(a); break;
            default: v = cos(a); break;
        }
        val[i] = v;
        adj[i] = 0.0;
    }
    adj[ne - 1] = 1.0;
    for (int64_t i = ne - 1; i >= nb; --i) {
        const int op = P.node_op[i];
        const double w = adj[i];
        if (op == KTN_OP_CONST) continue;
        if (op == KTN_OP_VAR) { O.jac[P.node_b[i]] += w; continue; }
        const int64_t ia = nb + P.node_a[i];
        const double a = val[ia];
        switch (op) {
            case KTN_OP_ADD: adj[ia] += w; adj[nb + P.node_b[i]] += w; break;
            case KTN_OP_SUB: adj[ia] += w; adj[nb + P.node_b[i]] -= w; break;
            case KTN_OP_MUL: { const int64_t ib = nb + P.node_b[i]; const double b = val[ib];
                               adj[ia] += w * b; adj[ib] += w * a; } break;
            case KTN_OP_DIV: { const int64_t ib = nb + P.node_b[i]; const double b = val[ib];
                               adj[ia] += w * (1.0 / b); adj[ib] += w * (-(val[i] / b)); } break;
            case KTN_OP_NEG: adj[ia] -= w; break;
            case KTN_OP_POWC: { const double p = P.node_c[i];
                                const double d = (p == 2.0) ? 2.0 * a : (p == 1.0 ? 1.0 : p * pow(a, p - 1.0));
                                adj[ia] += w * d; } break;
            case KTN_OP_EXP: adj[ia] += w * val[i]; break;
            case KTN_OP_LOG: adj[ia] += w * (1.0 / a); break;
            case KTN_OP_SQRT: adj[ia] += w * (0.5 / val[i]); break;
            case KTN_OP_SIN: adj[ia] += w * cos(a); break;
            default: adj[ia] += w * (-sin(a)); break;
        }
    }
    O.g[r] = val[ne - 1] + P.rconst[r];
}

// linear_oa_cut constant / round_coefs max / finite check / isconstrsat from a
// materialised Jacobian row (tape rows; also the host-evaluator fallback of section 8b).
// One thread per row, entries in storage order == the reference's left-to-right order.
static __global__ __launch_bounds__(kBlock) void k_gj_stats(NlpDev P, const int32_t* __restrict__ nl_rows, int64_t m_nl,
                                                     const double* __restrict__ x, double f_tol, int kind_filter,
                                                     SweepOut O) {
    const int64_t gid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (gid >= m_nl) return;
    const int32_t r = nl_rows[gid];
    if (kind_filter >= 0 && (P.row_kind[r] == KTN_ROW_SEP || P.row_kind[r] == 3)) return;     // tape rows and host-evaluated rows (3 = kRowSepLong: k_sep_eval_long's)
    const int64_t beg = P.rowptr[r], end = P.rowptr[r + 1];
    const double g = O.g[r];
    double b = g, mx = -__builtin_inf();
    int nf = 0;
    for (int64_t e = beg; e < end; ++e) {
        const double der = O.jac[e];
        b += -x[P.col[e]] * der;
        mx = nanmax(mx, der);
        nf |= !isfinite(der);
    }
    if (P.pad_zero[r]) mx = nanmax(mx, 0.0);
    O.bconst[r] = b;
    O.maxc[r] = mx;
    O.nonfin[r] = nf;
    const double lb = P.lb[r], ub = P.ub[r];
    const bool sat = (g >= lb - f_tol) && (g <= ub + f_tol);
    O.flag[gid] = sat ? 0 : 1;
    O.cnt[gid] = sat ? 0 : (end - beg);
    if (!sat) {
        atomic_max_nonneg(O.maxviol, fmax(g - ub, lb - g));
        if (nf) { if (*O.any_nonfin == 0) atomicOr(O.any_nonfin, 1); };
    }
}

// KTN_ROW_HOST rows: values and Jacobian entries computed by the caller's evaluator, staged in (gh, jh)
static __global__ __launch_bounds__(kBlock) void k_host_scatter(NlpDev P, const int32_t* __restrict__ host_rows, int64_t n_host,
                                                         const double* __restrict__ gh, const double* __restrict__ jh, SweepOut O) {
    const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (t >= n_host) return;
    const int32_t r = host_rows[t];
    O.g[r] = gh[r];
    for (int64_t e = P.rowptr[r]; e < P.rowptr[r + 1]; ++e) O.jac[e] = jh[e];
}

// loadproblem!: the packed row programs (col | kind << 29, (p0, p1)) from the caller's separate arrays
static __global__ __launch_bounds__(kBlock) void k_pack_atoms(int64_t nnz, const int32_t* __restrict__ col, const uint8_t* __restrict__ kind,
                                                       const double* __restrict__ p0, const double* __restrict__ p1,
                                                       int32_t* __restrict__ colk, double2* __restrict__ pp) {
    const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (e >= nnz) return;
    colk[e] = col[e] | ((int32_t)kind[e] << kKindShift);
    pp[e] = make_double2(p0[e], p1[e]);
}
// loadproblem!: the linear rows of the LP from the tangent at the origin (src/model.jl:110-122): row r of the LP is
// constraint lin_rows[r]; coefficients = its Jacobian entries at 0, bounds [l - b, u - b] with b = g(0) - sum 0 * J.
// lp_rowptr is already in place (the row lengths are structural).
static __global__ __launch_bounds__(kBlock) void k_lin_rows(int64_t nlin, const int32_t* __restrict__ lin_rows, const int64_t* __restrict__ rowptr,
                                                     const int32_t* __restrict__ col, const double* __restrict__ jac,
                                                     const double* __restrict__ g, const double* __restrict__ lb, const double* __restrict__ ub,
                                                     const int64_t* __restrict__ lp_rowptr, int32_t* __restrict__ lp_col,
                                                     double* __restrict__ lp_val, double* __restrict__ lp_lo, double* __restrict__ lp_hi) {
    const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (r >= nlin) return;
    const int32_t i = lin_rows[r];
    const int64_t beg = rowptr[i], len = rowptr[i + 1] - beg, dst = lp_rowptr[r];
    double b = g[i];
    for (int64_t e = 0; e < len; ++e) {
        const double jv = jac[beg + e];
        lp_col[dst + e] = col[beg + e];
        lp_val[dst + e] = jv;
        b += -0.0 * jv;                              // NaN / Inf coefficients poison the constant, as in linear_oa_cut
    }
    lp_lo[r] = lb[i] - b;
    lp_hi[r] = ub[i] - b;
}

// Growing row-sparse LP  lo <= A x <= hi  (CSR, rows only ever appended).
struct LpRows {
    int64_t* rowptr;
    int32_t* col;
    double* val;
    double* lo;
    double* hi;
    double* y;   // duals (unscaled), warm start across ECP iterations
};

// _addcut bookkeeping after the scans: row bounds (lb - b, ub - b), new rowptr entries,
// violated-row list, and the dual warm start (new cut inherits the dual of the previous
// cut of the same NL row).
static __global__ __launch_bounds__(kBlock) void k_compact(NlpDev P, const int32_t* __restrict__ nl_rows, int64_t m_nl,
                                                    const int64_t* __restrict__ flag, const int64_t* __restrict__ rank,
                                                    const int64_t* __restrict__ cnt_scan, const double* __restrict__ bconst,
                                                    int64_t base_row, int64_t base_nnz, LpRows L,
                                                    int32_t* __restrict__ viol_slots, int64_t* __restrict__ last_cut,
                                                    int64_t* __restrict__ cut_prev, int inherit) {
    const int64_t gid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (gid >= m_nl) return;
    if (!flag[gid]) return;
    const int32_t r = nl_rows[gid];
    const int64_t v = rank[gid];
    const int64_t R = base_row + v;
    const int64_t len = P.rowptr[r + 1] - P.rowptr[r];
    L.rowptr[R + 1] = base_nnz + cnt_scan[gid] + len;   // rowptr[base_row] already == base_nnz
    const double b = bconst[r];
    L.lo[R] = P.lb[r] - b;   // model.jl:74-75; a NaN constant gives NaN bounds -> vacuous side in the LP
    L.hi[R] = P.ub[r] - b;
    viol_slots[v] = (int32_t)gid;
    double y0 = 0.0;
    const int64_t prev = last_cut[gid];
    if (inherit && prev >= 0) { y0 = L.y[prev]; L.y[prev] = 0.0; }
    L.y[R] = y0;
    cut_prev[R] = prev;      // linked list of the cuts of this NL row, newest first (k_consolidate)
    last_cut[gid] = R;
}

// Rows appended from the host (multi-GPU exchange) join the per-NL-row cut lists through their GLOBAL NL-row id:
// the same bookkeeping k_compact does for the rows of a local sweep (dual inheritance, list threading).
static __global__ __launch_bounds__(kBlock) void k_append_link(int64_t nrows, int64_t base_row, const int64_t* __restrict__ nl_id,
                                                       int64_t nl_total, int64_t* __restrict__ glast,
                                                       int64_t* __restrict__ cut_prev, double* __restrict__ y, int inherit) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= nrows) return;
    const int64_t R = base_row + i, g = nl_id[i];
    double y0 = 0.0;
    int64_t prev = -1;
    if (g >= 0 && g < nl_total) {
        prev = glast[g];
        if (inherit && prev >= 0) { y0 = y[prev]; y[prev] = 0.0; }
        glast[g] = R;                      // one cut per NL row and append: no two threads share g
    }
    y[R] = y0;
    cut_prev[R] = prev;
}

// Dual-mass consolidation among the cuts of ONE nonlinear row (stall handler of the GPU LP).
// Near the optimum successive cuts of a row are nearly parallel; PDHG moves multiplier mass between
// two of them only at a rate proportional to the (tiny) violation and idles with the iterate stuck
// between the two: one cut violated by eps, the other slack by eps and still holding the mass.  By
// complementary slackness a cut that is slack at the (objective-converged) point carries no multiplier:
// move the mass of every cut of the row that is slack by more than `thresh` onto the row's tightest
// cut (unscaled duals, so A'y changes only by mass * (difference of two nearly equal rows)).  Cuts
// that are tight keep their multipliers, so a vertex formed by several cuts of one row is untouched.
// One thread per NL slot walks that slot's list; sequential => deterministic.
static __global__ __launch_bounds__(kBlock) void k_consolidate(int64_t m_nl, const int64_t* __restrict__ last_cut,
                                                        const int64_t* __restrict__ cut_prev, const double* __restrict__ ax,
                                                        const double* __restrict__ lo, const double* __restrict__ hi,
                                                        const double* __restrict__ dr, double thresh,
                                                        double* __restrict__ y, int32_t* __restrict__ moved) {
    const int64_t s = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (s >= m_nl) return;
    int64_t best_u = -1, best_l = -1;
    double res_u = -__builtin_inf(), res_l = -__builtin_inf();
    int ncuts = 0;
    for (int64_t r = last_cut[s]; r >= 0; r = cut_prev[r]) {
        ++ncuts;
        const double ru = ax[r] - hi[r], rl = lo[r] - ax[r];      // -inf on an infinite side, NaN on a vacuous one
        if (ru > res_u) { res_u = ru; best_u = r; }
        if (rl > res_l) { res_l = rl; best_l = r; }
    }
    if (ncuts < 2) return;
    double mass_u = 0.0, mass_l = 0.0;                             // unscaled multipliers: y_unscaled = y * dr
    for (int64_t r = last_cut[s]; r >= 0; r = cut_prev[r]) {
        const double yu = y[r] * dr[r];
        if (yu < 0.0 && best_u >= 0 && r != best_u && (ax[r] - hi[r]) < -thresh) { mass_u += yu; y[r] = 0.0; }
        else if (yu > 0.0 && best_l >= 0 && r != best_l && (lo[r] - ax[r]) < -thresh) { mass_l += yu; y[r] = 0.0; }
    }
    if (mass_u != 0.0) { y[best_u] += mass_u / dr[best_u]; atomicAdd(moved, 1); }
    if (mass_l != 0.0) { y[best_l] += mass_l / dr[best_l]; atomicAdd(moved, 1); }
}

// ------------------------------------------------------------------ cut-pool management ----
// (the reference never removes cuts, src/model.jl:215 TODO.)  A cut that carries
// no multiplier and is slack at x* for `max_age` consecutive LP solves is dropped: every cut is a valid
// inequality of the convex feasible set, so dropping one keeps the LP an outer approximation, and the sweep
// regenerates it should its row become violated again.
// (G lanes per row in the three pool kernels below: a thread per row walked the 1e4-entry epigraph cuts of a nonlinear
//  objective serially -- k_dedupe_mark 2.9 ms, k_purge_mark / k_purge_copy 1.3 ms per call on cfg2's QP variant)
template <int G>
__global__ __launch_bounds__(kBlock) void k_purge_mark(int64_t m_base, int64_t m, const int64_t* __restrict__ rowptr,
                                                       const int32_t* __restrict__ col, const double* __restrict__ val,
                                                       const double* __restrict__ x, const double* __restrict__ lo,
                                                       const double* __restrict__ hi, const double* __restrict__ y,
                                                       int32_t* __restrict__ age, double margin, int max_age,
                                                       int64_t* __restrict__ keep, int64_t* __restrict__ keepnnz) {
    const int64_t r = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / G;
    const int lane = threadIdx.x & (G - 1);
    if (r >= m) return;
    const int64_t beg = rowptr[r], end = rowptr[r + 1], len = end - beg;
    bool k = true;
    if (r >= m_base) {
        double ax = 0.0;
        for (int64_t e = beg + lane; e < end; e += G) ax += val[e] * x[col[e]];
        ax = group_sum<G>(ax);
        double slack = __builtin_inf(), scale = 1.0;
        if (isfinite(hi[r])) { slack = fmin(slack, hi[r] - ax); scale = fmax(scale, fabs(hi[r])); }
        if (isfinite(lo[r])) { slack = fmin(slack, ax - lo[r]); scale = fmax(scale, fabs(lo[r])); }
        const bool idle = (y[r] == 0.0) && (slack > margin * scale);
        const int a = idle ? age[r] + 1 : 0;
        if (lane == 0) age[r] = a;
        k = a < max_age;
    }
    if (lane == 0) { keep[r] = k ? 1 : 0; keepnnz[r] = k ? len : 0; }
}
// Near-duplicate cuts (second half; the reference's TODO at src/model.jl:215).  Close to the
// optimum successive iterates differ by ~1e-6, so successive tangent cuts of an active NL row are the same inequality up
// to that order -- rows that cost bytes in every PDHG iteration and are what makes the LP's duals degenerate.  For every NL
// row the cuts older than its NEWEST one are compared with it after normalising by the largest coefficient: a cut whose
// coefficients and bound agree within `eps` is dropped (dropping a cut only loosens the outer approximation, by O(eps)
// here) and its multiplier moves to the newest cut.  One thread per NL slot walks the slot's list (deterministic).
template <int G>
__global__ __launch_bounds__(kBlock) void k_dedupe_mark(int64_t nslots, const int64_t* __restrict__ last_cut,
                                                        const int64_t* __restrict__ cut_prev, const int64_t* __restrict__ rowptr,
                                                        const double* __restrict__ val, const double* __restrict__ lo,
                                                        const double* __restrict__ hi, double* __restrict__ y, double eps,
                                                        int64_t* __restrict__ keep, int64_t* __restrict__ keepnnz,
                                                        int32_t* __restrict__ dropped) {
    const int64_t s = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / G;
    const int lane = threadIdx.x & (G - 1);
    if (s >= nslots) return;
    const int64_t head = last_cut[s];
    if (head < 0 || !keep[head]) return;
    const int64_t hb = rowptr[head], hl = rowptr[head + 1] - hb;
    double nh = 0.0;
    for (int64_t e = lane; e < hl; e += G) nh = fmax(nh, fabs(val[hb + e]));
    nh = group_nanmax<G>(nh);                               // (every lane of the group takes the same branches below)
    if (!(nh > 0.0) || !isfinite(nh)) return;
    const bool up = isfinite(hi[head]);                     // a cut has one finite side (src/model.jl:74-75)
    const double bh = (up ? hi[head] : lo[head]) / nh;
    if (!isfinite(bh)) return;
    int nd = 0;
    for (int64_t r = cut_prev[head]; r >= 0; r = cut_prev[r]) {
        if (!keep[r]) continue;
        const int64_t rb = rowptr[r];
        if (rowptr[r + 1] - rb != hl || isfinite(hi[r]) != up) continue;
        double nr = 0.0;
        for (int64_t e = lane; e < hl; e += G) nr = fmax(nr, fabs(val[rb + e]));
        nr = group_nanmax<G>(nr);
        if (!(nr > 0.0) || !isfinite(nr)) continue;
        const double br = (up ? hi[r] : lo[r]) / nr;
        if (!(fabs(br - bh) <= eps * (1.0 + fabs(bh)))) continue;
        double diff = 0.0;
        for (int64_t e = lane; e < hl; e += G) diff = nanmax(diff, fabs(val[rb + e] / nr - val[hb + e] / nh));
        diff = group_nanmax<G>(diff);
        if (!(diff <= eps)) continue;
        if (lane == 0) {
            keep[r] = 0; keepnnz[r] = 0;
            y[head] += y[r] * (nr / nh);                    // A'y changes by O(eps |y|)
            y[r] = 0.0;
        }
        ++nd;
    }
    if (lane == 0 && nd) atomicAdd(dropped, nd);
}
template <int G>
__global__ __launch_bounds__(kBlock) void k_purge_copy(int64_t m, const int64_t* __restrict__ keep, const int64_t* __restrict__ newidx,
                                                       const int64_t* __restrict__ newptr, LpRows Old, const int32_t* __restrict__ age_old,
                                                       LpRows New, int32_t* __restrict__ age_new) {
    const int64_t r = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / G;
    const int lane = threadIdx.x & (G - 1);
    if (r >= m || !keep[r]) return;
    const int64_t nr = newidx[r], dst = newptr[r], src = Old.rowptr[r], len = Old.rowptr[r + 1] - src;
    for (int64_t e = lane; e < len; e += G) { New.col[dst + e] = Old.col[src + e]; New.val[dst + e] = Old.val[src + e]; }
    if (lane == 0) {
        New.rowptr[nr] = dst;
        New.lo[nr] = Old.lo[r]; New.hi[nr] = Old.hi[r]; New.y[nr] = Old.y[r];
        age_new[nr] = age_old[r];
    }
}
// re-thread the per-NL-row cut lists (k_consolidate) through the kept rows
static __global__ __launch_bounds__(kBlock) void k_purge_relink(int64_t m_nl, int64_t* __restrict__ last_cut,
                                                         const int64_t* __restrict__ prev_old, const int64_t* __restrict__ keep,
                                                         const int64_t* __restrict__ newidx, int64_t* __restrict__ prev_new) {
    const int64_t s = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (s >= m_nl) return;
    int64_t head = -1, tail = -1;
    for (int64_t r = last_cut[s]; r >= 0; r = prev_old[r]) {
        if (!keep[r]) continue;
        const int64_t nr = newidx[r];
        if (head < 0) head = nr; else prev_new[tail] = nr;
        tail = nr;
    }
    if (tail >= 0) prev_new[tail] = -1;
    last_cut[s] = head;
}

// gencut + round_coefs + row append: G lanes per violated row.
template <int G>
__global__ __launch_bounds__(kBlock) void k_emit(NlpDev P, const int32_t* __restrict__ nl_rows,
                                                 const int32_t* __restrict__ viol_slots, int64_t n_viol,
                                                 const double* __restrict__ x, const double* __restrict__ jac,
                                                 const double* __restrict__ maxc, double cut_coef_rng, int round_coefs,
                                                 int64_t base_row, LpRows L) {
    const int64_t v = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / G;
    const int lane = threadIdx.x & (G - 1);
    if (v >= n_viol) return;
    const int32_t r = nl_rows[viol_slots[v]];
    const int64_t beg = P.rowptr[r], end = P.rowptr[r + 1];
    const int64_t dst = L.rowptr[base_row + v];
    const bool sep = P.row_kind[r] == KTN_ROW_SEP;
    const double mx = maxc[r];
    for (int64_t e = beg + lane; e < end; e += G) {
        const int c = P.col[e];
        double der;
        if (sep) { double val; const double2 q = P.pp[e]; atom_eval((unsigned)P.colk[e] >> kKindShift, q.x, q.y, x[c], val, der); }
        else der = jac[e];
        if (round_coefs && (der + cut_coef_rng < mx)) der = 0.0;   // model.jl:202-206 (signed max)
        L.col[dst + (e - beg)] = c;
        L.val[dst + (e - beg)] = der;
    }
}

// ================================================================ LP: PDHG ========
// Scaled problem  A^ = Dr A Dc.  CSR (rows) serves A x, the CSC mirror serves A'y.
struct SpMat {
    const int64_t* ptr;
    const int32_t* idx;
    const double* val;   // scaled values
};

// Block-level accumulation of kChkQ quantities: [0..11] sums, [12..15] maxima.
struct ChkAcc {
    double s[kChkQ];
    __device__ void init() {
#pragma unroll
        for (int q = 0; q < kChkQ; ++q) s[q] = 0.0;
    }
};
// every thread of the block must call this (it holds a barrier); fixed-shape reduction: butterfly per wavefront,
// then the block's wavefronts in order.  MASK: the quantities this side of the check accumulates (the others are
// written as zeros without being reduced).
constexpr unsigned kChkRowMask = 0x141Fu;     // s0-s4, s10, m12
constexpr unsigned kChkColMask = 0x6FE0u;     // s5-s11, m13, m14
// G: only the first lane of every G-lane group carries a value (the others hold the zeros of init()), so the butterfly's last log2 G
// steps would add zeros and are left out.  The steps run quantity-interleaved -- one step of all chains, then the next -- so that
// the cross-lane latencies of the (up to nine) chains overlap instead of queueing behind each other; same sums, same order.
template <int BLOCK, unsigned MASK, int G = 1>
__device__ __forceinline__ void chk_block_store(ChkAcc& a, double* partials) {
    __shared__ double sh[kChkQ][BLOCK / 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int off = 32; off >= G; off >>= 1) {
#pragma unroll
        for (int q = 0; q < kChkQ; ++q) {
            if (!((MASK >> q) & 1u)) continue;
            const double o = __shfl_xor(a.s[q], off, 64);
            a.s[q] = (q < 12) ? a.s[q] + o : fmax(a.s[q], o);
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < kChkQ; ++q)
            if ((MASK >> q) & 1u) sh[q][wv] = a.s[q];
    }
    __syncthreads();
    if (threadIdx.x < kChkQ) {
        const int q = threadIdx.x;
        double v = 0.0;
        if ((MASK >> q) & 1u) {
            v = sh[q][0];
            for (int k = 1; k < BLOCK / 64; ++k) v = (q < 12) ? v + sh[q][k] : fmax(v, sh[q][k]);
        }
        partials[(int64_t)blockIdx.x * kChkQ + q] = v;
    }
}

// x-step + reflected Halpern update.  One group per column (CSC gather of y).
//   xt = clip(x - tau (c - A'y), l, u);  xbar = 2 xt - x
//   UPDATE: x <- w ((1+rho) xt - rho x) + (1-w) x0     else: store xt  (check iteration; xbar is not needed there,
//   the check form of the y-step gathers xt and x itself)
template <int G, bool UPDATE>
__global__ __launch_bounds__(kBlock) void k_pdhg_x(int64_t n, SpMat AT, const double* __restrict__ y,
                                                   double* __restrict__ x, const double* __restrict__ x0,
                                                   double* __restrict__ xt, double* __restrict__ xbar,
                                                   const double* __restrict__ c, const double* __restrict__ l,
                                                   const double* __restrict__ u, double tau, double w, double rho) {
    const int64_t j = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / G;
    const int lane = threadIdx.x & (G - 1);
    if (j >= n) return;
    const int64_t beg = AT.ptr[j], end = AT.ptr[j + 1];
    // the per-column scalars are requested up front by every lane of the group (same address: one
    // transaction), so their latency overlaps the gather chain instead of following the reduction
    // (check form with xbar != nullptr: the slot receives the Halpern update the iteration WOULD make -- if the host decides to go
    //  on, it swaps that array in for x instead of launching k_halpern2; same expression, same bits)
    const double xv = x[j], cj = c[j], lj = l[j], uj = u[j], x0j = (UPDATE || xbar) ? x0[j] : 0.0;
    double acc = 0.0;
    for (int64_t e = beg + lane; e < end; e += G) acc += AT.val[e] * y[AT.idx[e]];
    acc = group_sum<G>(acc);
    if (lane == 0) {
        const double xtv = clampd(xv - tau * (cj - acc), lj, uj);
        if (UPDATE) { xbar[j] = 2.0 * xtv - xv; x[j] = w * ((1.0 + rho) * xtv - rho * xv) + (1.0 - w) * x0j; }
        else {
            xt[j] = xtv;
            if (xbar) xbar[j] = w * ((1.0 + rho) * xtv - rho * xv) + (1.0 - w) * x0j;
        }
    }
}

// The x-step of an LP with long columns (a variable every cut contains): lane groups for the ordinary columns, one
// 1024-thread workgroup per long column (fixed-shape reduction: butterfly per wavefront, wavefronts in order).
template <int G, bool UPDATE>
__global__ __launch_bounds__(kBlock) void k_pdhg_x_skip(int64_t n, SpMat AT, const double* __restrict__ y,
                                                        double* __restrict__ x, const double* __restrict__ x0,
                                                        double* __restrict__ xt, double* __restrict__ xbar,
                                                        const double* __restrict__ c, const double* __restrict__ l,
                                                        const double* __restrict__ u, double tau, double w, double rho, int64_t skip_longer) {
    const int64_t j = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / G;
    const int lane = threadIdx.x & (G - 1);
    if (j >= n) return;
    const int64_t beg = AT.ptr[j], end = AT.ptr[j + 1];
    if (end - beg > skip_longer) return;                  // k_pdhg_x_long
    const double xv = x[j], cj = c[j], lj = l[j], uj = u[j], x0j = UPDATE ? x0[j] : 0.0;
    double acc = 0.0;
    for (int64_t e = beg + lane; e < end; e += G) acc += AT.val[e] * y[AT.idx[e]];
    acc = group_sum<G>(acc);
    if (lane == 0) {
        const double xtv = clampd(xv - tau * (cj - acc), lj, uj);
        if (UPDATE) { xbar[j] = 2.0 * xtv - xv; x[j] = w * ((1.0 + rho) * xtv - rho * xv) + (1.0 - w) * x0j; }
        else xt[j] = xtv;
    }
}
template <bool UPDATE>
__global__ __launch_bounds__(1024) void k_pdhg_x_long(const int32_t* __restrict__ cols, SpMat AT, const double* __restrict__ y,
                                                      double* __restrict__ x, const double* __restrict__ x0,
                                                      double* __restrict__ xt, double* __restrict__ xbar,
                                                      const double* __restrict__ c, const double* __restrict__ l,
                                                      const double* __restrict__ u, double tau, double w, double rho) {
    __shared__ double sh[1024 / 64];
    const int64_t j = cols[blockIdx.x];
    double acc = 0.0;
    for (int64_t e = AT.ptr[j] + threadIdx.x; e < AT.ptr[j + 1]; e += 1024) acc += AT.val[e] * y[AT.idx[e]];
    acc = group_sum<64>(acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0;
        for (int k = 0; k < 1024 / 64; ++k) a += sh[k];
        const double xv = x[j];
        const double xtv = clampd(xv - tau * (c[j] - a), l[j], u[j]);
        if (UPDATE) { xbar[j] = 2.0 * xtv - xv; x[j] = w * ((1.0 + rho) * xtv - rho * xv) + (1.0 - w) * x0[j]; }
        else xt[j] = xtv;
    }
}

// ---- packed operands of the plain (update) steps ----------------------------------------------------------------
// Per column the x-step reads (beg, len) as one 8-byte pair and (c, l, u, x0) as one 32-byte record instead of seven
// separate arrays; per row the y-step reads (lo, hi, y0, beg | len) as one 32-byte record.  The records mirror ch / lh /
// uh / x0h and loh / hih / y0h (which the check kernels keep using) and are rewritten where those change: at the start
// of a solve (k_pack_cols / k_pack_rows) and at restarts (k_restart_set).
struct __attribute__((aligned(32))) ColRec { double c, l, u, x0; };
struct __attribute__((aligned(32))) RowRec { double lo, hi, y0; int32_t beg, len; };

static __global__ __launch_bounds__(kBlock) void k_pack_cols(int64_t n, const int64_t* __restrict__ ptr, const double* __restrict__ c,
                                                      const double* __restrict__ l, const double* __restrict__ u,
                                                      const double* __restrict__ x, double* __restrict__ x0,
                                                      ColRec* __restrict__ rec, int2* __restrict__ bl) {
    const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= n) return;
    const double xv = x[j];
    x0[j] = xv;
    ColRec r; r.c = c[j]; r.l = l[j]; r.u = u[j]; r.x0 = xv;
    rec[j] = r;
    bl[j] = make_int2((int)ptr[j], (int)(ptr[j + 1] - ptr[j]));
}
static __global__ __launch_bounds__(kBlock) void k_pack_rows(int64_t m, const int64_t* __restrict__ ptr, const double* __restrict__ lo,
                                                      const double* __restrict__ hi, const double* __restrict__ y,
                                                      double* __restrict__ y0, RowRec* __restrict__ rec) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= m) return;
    const double yv = y[i];
    y0[i] = yv;
    RowRec r; r.lo = lo[i]; r.hi = hi[i]; r.y0 = yv; r.beg = (int32_t)ptr[i]; r.len = (int32_t)(ptr[i + 1] - ptr[i]);
    rec[i] = r;
}
// the packed records of the columns and of the rows in one launch (the bodies of k_pack_cols / k_pack_rows)
static __global__ __launch_bounds__(kBlock) void k_pack_both(int64_t n, const int64_t* __restrict__ cptr, const double* __restrict__ c,
                                                      const double* __restrict__ l, const double* __restrict__ u, const double* __restrict__ x,
                                                      double* __restrict__ x0, ColRec* __restrict__ crec, int2* __restrict__ bl, int64_t m,
                                                      const int64_t* __restrict__ rptr, const double* __restrict__ lo, const double* __restrict__ hi,
                                                      const double* __restrict__ y, double* __restrict__ y0, RowRec* __restrict__ rrec) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) {
        const double xv = x[i];
        x0[i] = xv;
        ColRec r; r.c = c[i]; r.l = l[i]; r.u = u[i]; r.x0 = xv;
        crec[i] = r;
        bl[i] = make_int2((int)cptr[i], (int)(cptr[i + 1] - cptr[i]));
    }
    if (i < m) {
        const double yv = y[i];
        y0[i] = yv;
        RowRec r; r.lo = lo[i]; r.hi = hi[i]; r.y0 = yv; r.beg = (int32_t)rptr[i]; r.len = (int32_t)(rptr[i + 1] - rptr[i]);
        rrec[i] = r;
    }
}
// T outputs per lane group (output g, g + groups, ...): the records and the first entries of all T are requested before any
// is used, so a wavefront keeps T times the loads in flight and the grid is T times smaller (shorter ramp-up and drain of
// a ~6 us kernel whose boundary costs ~1.5 us).
template <int G, int T>
__global__ __launch_bounds__(kBlock) void k_pdhg_x_packed(int64_t n, const int2* __restrict__ bl, const int32_t* __restrict__ idx,
                                                          const double* __restrict__ val, const double* __restrict__ y,
                                                          double* __restrict__ x, double* __restrict__ xbar,
                                                          const ColRec* __restrict__ rec, double tau, double w, double rho) {
    const int64_t groups = (int64_t)gridDim.x * (kBlock / G);
    const int64_t g0 = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / G;
    const int lane = threadIdx.x & (G - 1);
    int2 b[T];
    ColRec r[T];
    double xv[T], acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int64_t j = g0 + t * groups;
        const bool on = j < n;
        b[t] = on ? bl[j] : make_int2(0, 0);
        if (on) r[t] = rec[j];
        xv[t] = on ? x[j] : 0.0;
        acc[t] = 0.0;
    }
#pragma unroll
    for (int t = 0; t < T; ++t)
        for (int e = b[t].x + lane; e < b[t].x + b[t].y; e += G) acc[t] += val[e] * y[idx[e]];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int64_t j = g0 + t * groups;
        const double a = group_sum<G>(acc[t]);
        if (lane == 0 && j < n) {
            const double xtv = clampd(xv[t] - tau * (r[t].c - a), r[t].l, r[t].u);
            xbar[j] = 2.0 * xtv - xv[t];
            x[j] = w * ((1.0 + rho) * xtv - rho * xv[t]) + (1.0 - w) * r[t].x0;
        }
    }
}
template <int G, int T>
__global__ __launch_bounds__(kBlock) void k_pdhg_y_packed(int64_t m, const int32_t* __restrict__ idx, const double* __restrict__ val,
                                                          const double* __restrict__ xbar, double* __restrict__ y,
                                                          const RowRec* __restrict__ rec, double sigma, double w, double rho,
                                                          int long_thresh, const int32_t* __restrict__ long_rows, int n_long) {
    // The last n_long workgroups of the grid take one LONG row each (dense epigraph cuts: n + 1 entries) -- the rows the lane
    // groups below skip.  They used to be a launch of their own (k_pdhg_y_long): 6 us + a boundary per PDHG iteration, a
    // quarter of the GPU time of cfg2's QP variant.  Sixteen entries per thread are requested before the first is used;
    // fixed-shape reduction (butterfly per wavefront, wavefronts in order) => deterministic.
    const int nreg = (int)gridDim.x - n_long;
    if ((int)blockIdx.x >= nreg) {
        __shared__ double sh[kBlock / 64];
        const int i = long_rows[(int)blockIdx.x - nreg];
        const RowRec rr = rec[i];
        const int end = rr.beg + rr.len;
        double acc = 0.0;
        int e = rr.beg + (int)threadIdx.x;
        for (; e + 15 * kBlock < end; e += 16 * kBlock) {
            double v[16], xg[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = val[e + k * kBlock];
#pragma unroll
            for (int k = 0; k < 16; ++k) xg[k] = xbar[idx[e + k * kBlock]];
#pragma unroll
            for (int k = 0; k < 16; ++k) acc += v[k] * xg[k];
        }
        for (; e + 3 * kBlock < end; e += 4 * kBlock) {
            double v[4], xg[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = val[e + k * kBlock];
#pragma unroll
            for (int k = 0; k < 4; ++k) xg[k] = xbar[idx[e + k * kBlock]];
#pragma unroll
            for (int k = 0; k < 4; ++k) acc += v[k] * xg[k];
        }
        for (; e < end; e += kBlock) acc += val[e] * xbar[idx[e]];
        acc = group_sum<64>(acc);
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            double a = 0.0;
            for (int k = 0; k < kBlock / 64; ++k) a += sh[k];
            const double yv = y[i];
            const double v = yv - sigma * a;
            const double ytv = v + sigma * clampd(-v / sigma, rr.lo, rr.hi);
            y[i] = w * ((1.0 + rho) * ytv - rho * yv) + (1.0 - w) * rr.y0;
        }
        return;
    }
    const int64_t groups = (int64_t)nreg * (kBlock / G);
    const int64_t g0 = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / G;
    const int lane = threadIdx.x & (G - 1);
    RowRec r[T];
    double yv[T], acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int64_t i = g0 + t * groups;
        const bool on = i < m;
        if (on) r[t] = rec[i]; else { r[t].beg = 0; r[t].len = 0; }
        if (r[t].len > long_thresh) r[t].len = -1;      // served by k_pdhg_y_long (a workgroup per row)
        yv[t] = on ? y[i] : 0.0;
        acc[t] = 0.0;
    }
#pragma unroll
    for (int t = 0; t < T; ++t)
        for (int e = r[t].beg + lane; e < r[t].beg + r[t].len; e += G) acc[t] += val[e] * xbar[idx[e]];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int64_t i = g0 + t * groups;
        const double a = group_sum<G>(acc[t]);
        if (lane == 0 && i < m && r[t].len >= 0) {
            const double v = yv[t] - sigma * a;
            const double ytv = v + sigma * clampd(-v / sigma, r[t].lo, r[t].hi);
            y[i] = w * ((1.0 + rho) * ytv - rho * yv[t]) + (1.0 - w) * r[t].y0;
        }
    }
}

// KKT / fixed-point quantities, row side, accumulated by the check form of the y-step.
//  s0 sum dy*(A dx)   s1 sum dy^2      s2 dual objective (rows)  s3 sum (yt-y0)^2  s4 sum yt^2
//  s10 abs terms of the dual objective   m12 max unscaled row violation
__device__ __forceinline__ void chk_row_accumulate(ChkAcc& a, double ytv, double yv, double y0v, double axt, double axk,
                                                   double lo, double hi, double dri) {
    const double dy = ytv - yv;
    a.s[0] += dy * (axt - axk);
    a.s[1] += dy * dy;
    // The projection makes yt sign-feasible (yt > 0 only where lo is finite) up to ROUNDING: v + sigma * (-v / sigma) can
    // leave a residue of a few 1e-17 with the wrong sign, and -inf * 1e-17 = -inf would void the duality gap of this
    // check (seen at every second check of a slowly converging solve).  Such residues carry no dual objective.
    if (ytv > 0.0) { if (lo > -__builtin_inf()) { a.s[2] += lo * ytv; a.s[10] += fabs(lo * ytv); } }
    else if (ytv < 0.0) { if (hi < __builtin_inf()) { a.s[2] += hi * ytv; a.s[10] += fabs(hi * ytv); } }
    const double d0 = ytv - y0v;
    a.s[3] += d0 * d0;
    a.s[4] += ytv * ytv;
    const double viol = fmax(fmax(lo - axt, axt - hi), 0.0) / dri;
    a.s[12] = fmax(a.s[12], viol);
}

// y-step + reflected Halpern update.  One group per row (CSR gather of xbar).
//   v = y - sigma A xbar;  yt = v + sigma clip(-v/sigma, lo, hi)
template <int G>
__global__ __launch_bounds__(kBlock) void k_pdhg_y(int64_t m, SpMat A, const double* __restrict__ xbar,
                                                   double* __restrict__ y, const double* __restrict__ y0,
                                                   const double* __restrict__ lo, const double* __restrict__ hi,
                                                   double sigma, double w, double rho, int64_t long_thresh) {
    const int64_t i = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / G;
    const int lane = threadIdx.x & (G - 1);
    if (i >= m) return;
    const int64_t beg = A.ptr[i], end = A.ptr[i + 1];
    if (end - beg > long_thresh) return;          // served by k_pdhg_y_long (a workgroup per row)
    const double yv = y[i], loi = lo[i], hii = hi[i], y0i = y0[i];
    double acc = 0.0;
    for (int64_t e = beg + lane; e < end; e += G) acc += A.val[e] * xbar[A.idx[e]];
    acc = group_sum<G>(acc);
    if (lane == 0) {
        const double v = yv - sigma * acc;
        const double ytv = v + sigma * clampd(-v / sigma, loi, hii);
        y[i] = w * ((1.0 + rho) * ytv - rho * yv) + (1.0 - w) * y0i;
    }
}

// Check form of the y-step: no update; gathers xt and x (A xbar = 2 A xt - A x), stores yt and accumulates the row side of
// the KKT / fixed-point sums -- the two extra SpMV passes a separate check kernel would need ride on the gathers the step
// does anyway.  No early return: every thread reaches the block reduction.
template <int G>
__global__ __launch_bounds__(kBlock) void k_pdhg_y_chk(int64_t m, SpMat A, const double* __restrict__ xt,
                                                       const double* __restrict__ x, const double* __restrict__ y,
                                                       const double* __restrict__ y0, double* __restrict__ yt,
                                                       const double* __restrict__ lo, const double* __restrict__ hi,
                                                       const double* __restrict__ dr, double sigma, int64_t long_thresh,
                                                       double* __restrict__ partials, double* __restrict__ ynext, double w, double rho) {
    const int64_t i = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / G;
    const int lane = threadIdx.x & (G - 1);
    ChkAcc a; a.init();
    const bool on = i < m;
    const int64_t beg = on ? A.ptr[i] : 0, end = on ? A.ptr[i + 1] : 0;
    const bool mine = on && (end - beg <= long_thresh);
    double yv = 0.0, loi = 0.0, hii = 0.0, y0i = 0.0, dri = 1.0;
    if (mine) { yv = y[i]; loi = lo[i]; hii = hi[i]; y0i = y0[i]; dri = dr[i]; }
    double axt = 0.0, axk = 0.0;
    if (mine)
        for (int64_t e = beg + lane; e < end; e += G) {
            const int c = A.idx[e];
            const double v = A.val[e];
            axt += v * xt[c];
            axk += v * x[c];
        }
    axt = group_sum<G>(axt);
    axk = group_sum<G>(axk);
    if (mine && lane == 0) {
        const double v = yv - sigma * (2.0 * axt - axk);
        const double ytv = v + sigma * clampd(-v / sigma, loi, hii);
        yt[i] = ytv;
        if (ynext) ynext[i] = w * ((1.0 + rho) * ytv - rho * yv) + (1.0 - w) * y0i;      // (see k_pdhg_x: the update the iteration would make)
        chk_row_accumulate(a, ytv, yv, y0i, axt, axk, loi, hii, dri);
    }
    chk_block_store<kBlock, kChkRowMask, G>(a, partials);
}

// Long rows (dense epigraph cuts: n+1 entries): one 1024-thread workgroup per row, fixed-shape
// reduction (butterfly per wavefront, then the partials through LDS) => deterministic.
constexpr int kLongBlock = 1024;
template <bool CHECK>
__global__ __launch_bounds__(kLongBlock) void k_pdhg_y_long(const int32_t* __restrict__ rows, SpMat A,
                                                        const double* __restrict__ xa, const double* __restrict__ xb,
                                                        double* __restrict__ y, const double* __restrict__ y0,
                                                        double* __restrict__ yt, const double* __restrict__ lo,
                                                        const double* __restrict__ hi, const double* __restrict__ dr,
                                                        double sigma, double w, double rho, double* __restrict__ partials) {
    // plain: xa = xbar.  CHECK: xa = xt, xb = x; partials = this row's slot after the regular blocks' partials
    const int64_t i = rows[blockIdx.x];
    const int64_t beg = A.ptr[i], end = A.ptr[i + 1];
    double acc = 0.0, acc2 = 0.0;
    for (int64_t e = beg + threadIdx.x; e < end; e += kLongBlock) {
        const int c = A.idx[e];
        const double v = A.val[e];
        acc += v * xa[c];
        if (CHECK) acc2 += v * xb[c];
    }
    __shared__ double sh[2][kLongBlock / 64];
    acc = group_sum<64>(acc);
    if (CHECK) acc2 = group_sum<64>(acc2);
    if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = acc; sh[1][threadIdx.x >> 6] = acc2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double ax = 0.0, ax2 = 0.0;
        for (int k = 0; k < kLongBlock / 64; ++k) { ax += sh[0][k]; ax2 += sh[1][k]; }
        const double yv = y[i];
        if (CHECK) {
            const double v = yv - sigma * (2.0 * ax - ax2);
            const double ytv = v + sigma * clampd(-v / sigma, lo[i], hi[i]);
            yt[i] = ytv;
            ChkAcc a; a.init();
            chk_row_accumulate(a, ytv, yv, y0[i], ax, ax2, lo[i], hi[i], dr[i]);
#pragma unroll
            for (int q = 0; q < kChkQ; ++q) partials[(int64_t)blockIdx.x * kChkQ + q] = a.s[q];
        } else {
            const double v = yv - sigma * ax;
            const double ytv = v + sigma * clampd(-v / sigma, lo[i], hi[i]);
            y[i] = w * ((1.0 + rho) * ytv - rho * yv) + (1.0 - w) * y0[i];
        }
    }
}
static __global__ __launch_bounds__(kBlock) void k_find_long(int64_t m, const int64_t* __restrict__ rowptr, int64_t thresh,
                                                      int32_t* __restrict__ list, int32_t* __restrict__ count) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= m) return;
    if (rowptr[i + 1] - rowptr[i] > thresh) list[atomicAdd(count, 1)] = (int32_t)i;
}

// the same for the columns of the mirror, with the longest length reported too (count[0]: long ones, count[1]: max length)
static __global__ __launch_bounds__(kBlock) void k_find_long_max(int64_t n, const int64_t* __restrict__ ptr, int64_t thresh,
                                                          int32_t* __restrict__ list, int32_t* __restrict__ count) {
    const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    int len = 0;
    if (j < n) {
        len = (int)(ptr[j + 1] - ptr[j]);
        if (len > thresh) list[atomicAdd(count, 1)] = (int32_t)j;
    }
    __shared__ int s_max;
    if (threadIdx.x == 0) s_max = 0;
    __syncthreads();
    if (len > 0) atomicMax(&s_max, len);
    __syncthreads();
    if (threadIdx.x == 0 && s_max > 0) atomicMax(count + 1, s_max);
}

// Restart: z <- T(z), anchor z0 <- T(z), primal and dual part in one launch (instead of four device-to-device copies);
// the packed records of the plain steps carry x0 / y0 too.
static __global__ __launch_bounds__(kBlock) void k_restart_set(int64_t n, int64_t m, const double* __restrict__ xt, double* __restrict__ x,
                                                       double* __restrict__ x0, const double* __restrict__ yt,
                                                       double* __restrict__ y, double* __restrict__ y0,
                                                       ColRec* __restrict__ crec, RowRec* __restrict__ rrec) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) { const double v = xt[i]; x[i] = v; x0[i] = v; if (crec) crec[i].x0 = v; }
    if (i < m) { const double v = yt[i]; y[i] = v; y0[i] = v; if (rrec) rrec[i].y0 = v; }
}
// Halpern update of the primal and the dual part in one launch (after a check iteration that neither terminated nor
// restarted); also refreshes xbar = 2 xt - x_old for nobody: the next x-step recomputes it
static __global__ __launch_bounds__(kBlock) void k_halpern2(int64_t n, int64_t m, double* __restrict__ x, const double* __restrict__ xt,
                                                    const double* __restrict__ x0, double* __restrict__ y,
                                                    const double* __restrict__ yt, const double* __restrict__ y0, double w, double rho) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) x[i] = w * ((1.0 + rho) * xt[i] - rho * x[i]) + (1.0 - w) * x0[i];
    if (i < m) y[i] = w * ((1.0 + rho) * yt[i] - rho * y[i]) + (1.0 - w) * y0[i];
}

// second stage of the deterministic reductions: one block of kRedBlocks threads per (side, quantity) -- blockIdx.x =
// side * kChkQ + q, side 0: rows, 1: columns -- thread b owns the partial blocks b, b + kRedBlocks, ... in that order,
// then a fixed-shape butterfly + LDS tree -> run-to-run identical sums.
static __global__ __launch_bounds__(kRedBlocks) void k_chk_final(const double* __restrict__ prow, int nrow, const double* __restrict__ pcol,
                                                          int ncol, double* __restrict__ out) {
    __shared__ double sh[kRedBlocks / 64];
    const int side = blockIdx.x / kChkQ, q = blockIdx.x - side * kChkQ;
    const double* partials = side ? pcol : prow;
    const int nblocks = side ? ncol : nrow;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double v = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += kRedBlocks) {
        const double p = partials[(int64_t)b * kChkQ + q];
        v = (q < 12) ? v + p : fmax(v, p);
    }
    v = (q < 12) ? group_sum<64>(v) : group_max<64>(v);
    if (lane == 0) sh[wv] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = sh[0];
        for (int k = 1; k < kRedBlocks / 64; ++k) r = (q < 12) ? r + sh[k] : fmax(r, sh[k]);
        out[side * kChkQ + q] = r;
    }
}

//  s5 sum dx^2  s6 primal objective  s7 dual objective (bounds)  s8 sum (xt-x0)^2  s9 sum xt^2
//  s10 Farkas value (bounds part)  s11 its absolute terms  m14 max reduced cost of the c=0 problem on an infinite bound
//  m13 max unscaled dual residual (reduced cost not absorbable by a finite bound)
// Column side of the check: A'yt needs yt, which only exists after the y-step, so this one SpMV pass stays a kernel of
// its own -- G lanes per column over the whole chip, like the x-step.
template <int G>
__global__ __launch_bounds__(kBlock) void k_chk_cols(int64_t n, SpMat AT, const double* __restrict__ x,
                                                     const double* __restrict__ xt, const double* __restrict__ x0,
                                                     const double* __restrict__ yt, const double* __restrict__ c,
                                                     const double* __restrict__ l, const double* __restrict__ u,
                                                     const double* __restrict__ dc, double* __restrict__ partials) {
    const int64_t j = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / G;
    const int lane = threadIdx.x & (G - 1);
    ChkAcc a; a.init();
    const bool on = j < n;
    const int64_t beg = on ? AT.ptr[j] : 0, end = on ? AT.ptr[j + 1] : 0;
    double xtv = 0.0, xv = 0.0, x0v = 0.0, cj = 0.0, lj = 0.0, uj = 0.0, dcj = 1.0;
    if (on) { xtv = xt[j]; xv = x[j]; x0v = x0[j]; cj = c[j]; lj = l[j]; uj = u[j]; dcj = dc[j]; }
    double aty = 0.0;
    for (int64_t e = beg + lane; e < end; e += G) aty += AT.val[e] * yt[AT.idx[e]];
    aty = group_sum<G>(aty);
    if (on && lane == 0) {
        const double dx = xtv - xv;
        a.s[5] += dx * dx;
        a.s[6] += cj * xtv;
        const double r = cj - aty;
        double bad = 0.0;
        if (r > 0.0) { if (isfinite(lj)) a.s[7] += lj * r; else bad = r; }
        else if (r < 0.0) { if (isfinite(uj)) a.s[7] += uj * r; else bad = -r; }
        const double d0 = xtv - x0v;
        a.s[8] += d0 * d0;
        a.s[9] += xtv * xtv;
        a.s[13] = fmax(a.s[13], bad / dcj);
        // Farkas value of yt: the dual objective with c = 0 (positive <=> the rows + bounds are infeasible)
        const double r0 = -aty;
        if (r0 > 0.0) { if (isfinite(lj)) { a.s[10] += lj * r0; a.s[11] += fabs(lj * r0); } else a.s[14] = fmax(a.s[14], r0); }
        else if (r0 < 0.0) { if (isfinite(uj)) { a.s[10] += uj * r0; a.s[11] += fabs(uj * r0); } else a.s[14] = fmax(a.s[14], -r0); }
    }
    chk_block_store<kBlock, kChkColMask, G>(a, partials);
}

// ================================================== LP: row-sharded over several GPUs ===============================
//   Every rank holds a block of the linear rows and the cuts of ITS block of NL rows; x is
// replicated, y is local.  A x is local; A'y is the local partial k_spmv(A'_r, y_r) summed over the ranks by ONE
// all-reduce of an n-vector per PDHG iteration, after which every rank runs the same element-wise primal step:
template <bool UPDATE>
__global__ __launch_bounds__(kBlock) void k_x_prox(int64_t n, const double* __restrict__ aty, double* __restrict__ x,
                                                   const double* __restrict__ x0, double* __restrict__ xt, double* __restrict__ xbar,
                                                   const double* __restrict__ c, const double* __restrict__ l,
                                                   const double* __restrict__ u, double tau, double w, double rho) {
    const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= n) return;
    const double xv = x[j];
    const double xtv = clampd(xv - tau * (c[j] - aty[j]), l[j], u[j]);
    if (UPDATE) { xbar[j] = 2.0 * xtv - xv; x[j] = w * ((1.0 + rho) * xtv - rho * xv) + (1.0 - w) * x0[j]; }
    else xt[j] = xtv;
}
// column side of the check from the all-reduced A'yt (same sums as k_chk_cols; identical on every rank)
static __global__ __launch_bounds__(kBlock) void k_chk_cols_vec(int64_t n, const double* __restrict__ atyv, const double* __restrict__ x,
                                                         const double* __restrict__ xt, const double* __restrict__ x0,
                                                         const double* __restrict__ c, const double* __restrict__ l,
                                                         const double* __restrict__ u, const double* __restrict__ dc,
                                                         double* __restrict__ partials) {
    const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    ChkAcc a; a.init();
    if (j < n) {
        const double xtv = xt[j], xv = x[j], x0v = x0[j], cj = c[j], lj = l[j], uj = u[j], dcj = dc[j], aty = atyv[j];
        const double dx = xtv - xv;
        a.s[5] += dx * dx;
        a.s[6] += cj * xtv;
        const double r = cj - aty;
        double bad = 0.0;
        if (r > 0.0) { if (isfinite(lj)) a.s[7] += lj * r; else bad = r; }
        else if (r < 0.0) { if (isfinite(uj)) a.s[7] += uj * r; else bad = -r; }
        const double d0 = xtv - x0v;
        a.s[8] += d0 * d0;
        a.s[9] += xtv * xtv;
        a.s[13] = fmax(a.s[13], bad / dcj);
        const double r0 = -aty;
        if (r0 > 0.0) { if (isfinite(lj)) { a.s[10] += lj * r0; a.s[11] += fabs(lj * r0); } else a.s[14] = fmax(a.s[14], r0); }
        else if (r0 < 0.0) { if (isfinite(uj)) { a.s[10] += uj * r0; a.s[11] += fabs(uj * r0); } else a.s[14] = fmax(a.s[14], -r0); }
    }
    chk_block_store<kBlock, kChkColMask>(a, partials);
}

// ---- peer-buffer transport for the row-sharded LP (ktn_dist_init_ipc) -------------------------------------------------
// Instead of a ring all-reduce (RCCL: 2 (w - 1) dependent hops for 0.8 MB) every rank EXPOSES its partial vector in a buffer
// the other ranks have mapped (hipIpcOpenMemHandle: xGMI peer loads), tells them so, and each rank adds up all w partials
// itself, in rank order -- the same sum, bit for bit, on every rank.  Per all-reduce: the producer writes straight into the
// exposed slot, ONE single-workgroup kernel signals + waits (k_ipc_barrier), and the consumer (the primal step itself,
// k_x_prox_ipc) reads the w slots.  Rules the protocol rests on:
//   * data is only ever read by a kernel that STARTS after the barrier kernel of that epoch has ended, and was written by a
//     kernel that ENDED before it started (stream order): visibility rides on kernel boundaries, never on a fence inside a
//     running grid; peer reads are system-scope loads on top (no stale line of this GPU's L2s can serve them);
//   * flags are monotone epoch counters in uncached memory, one word per (receiver, source): no reset, no ABA;
//   * two slots alternate: a rank overwrites slot s at epoch e + 2 only after passing the barrier of epoch e + 1, which
//     every peer enters after its reads of epoch e (stream order);
//   * every spin is bounded (wall clock); a timeout is reported through *err (host-mapped) and turns into an error status.
// (kIpcMaxRanks, IpcPeers: types.hpp)
__device__ __forceinline__ double ipc_load(const double* p) {
    return __longlong_as_double((long long)__hip_atomic_load((unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
}
static __global__ __launch_bounds__(64) void k_ipc_barrier(IpcPeers P, int rank, int world, unsigned long long epoch, long long timeout_ticks,
                                                    int* __restrict__ err) {
    // A failure is STICKY and FAST: once *err is set (a peer did not arrive in time, or a peer said it failed) every later
    // barrier of this rank returns at once instead of spinning the full timeout again -- the host only looks at *err at the next
    // KKT check, up to lp_check_every barriers later -- and tells the peers by writing the poison epoch into their flag words,
    // so that they fail within one spin too instead of waiting for a rank that has given up.
    constexpr unsigned long long kPoison = ~0ULL;
    const int t = threadIdx.x;
    if (t >= world) return;
    volatile int* verr = err;
    if (*verr != 0) {
        if (t != rank) __hip_atomic_store(P.flags[t] + rank, kPoison, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    __hip_atomic_store(P.flags[t] + rank, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);       // "my slot of this epoch is complete"
    const long long t0 = (long long)wall_clock64();
    for (;;) {
        const unsigned long long v = __hip_atomic_load(P.flags[rank] + t, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
        if (v == kPoison) { *verr = 101 + t; break; }                                  // rank t gave up
        if (v >= epoch) break;
        if ((long long)wall_clock64() - t0 > timeout_ticks) { *verr = 1 + t; break; }  // rank t did not arrive
        if (*verr != 0) break;                                                         // another lane of this barrier already failed
        __builtin_amdgcn_s_sleep(4);
    }
    if (*verr != 0 && t != rank) __hip_atomic_store(P.flags[t] + rank, kPoison, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// out = sum (OP 0) or max (OP 1) over the ranks of their slot, in rank order
template <int OP>
__global__ __launch_bounds__(kBlock) void k_ipc_reduce(int64_t n, IpcPeers P, int world, int64_t off, double* __restrict__ out) {
    const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= n) return;
    double s = ipc_load(P.data[0] + off + j);
    for (int r = 1; r < world; ++r) {
        const double v = ipc_load(P.data[r] + off + j);
        s = OP ? fmax(s, v) : s + v;
    }
    out[j] = s;
}
// k_x_prox with A'y = the sum of the ranks' exposed partials
template <bool UPDATE>
__global__ __launch_bounds__(kBlock) void k_x_prox_ipc(int64_t n, IpcPeers P, int world, int64_t off, double* __restrict__ x,
                                                       const double* __restrict__ x0, double* __restrict__ xt, double* __restrict__ xbar,
                                                       const double* __restrict__ c, const double* __restrict__ l,
                                                       const double* __restrict__ u, double tau, double w, double rho) {
    const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= n) return;
    double aty = ipc_load(P.data[0] + off + j);
    for (int r = 1; r < world; ++r) aty += ipc_load(P.data[r] + off + j);
    const double xv = x[j];
    const double xtv = clampd(xv - tau * (c[j] - aty), l[j], u[j]);
    if (UPDATE) { xbar[j] = 2.0 * xtv - xv; x[j] = w * ((1.0 + rho) * xtv - rho * xv) + (1.0 - w) * x0[j]; }
    else xt[j] = xtv;
}
static __global__ __launch_bounds__(kBlock) void k_probe_fill(int64_t n, double* __restrict__ v, double base) {
    const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j < n) v[j] = base + 1e-3 * (double)(j % 1000);
}

// ================================================== LP: tiled SpMV for LPs beyond the caches =========================
// (DESIGN.md section 4 "HBM-regime SpMV".)  In CSR form every 8-byte gather of the input vector is its own L1 miss and
// drags a 128-byte line from L2; with tens of millions of entries that traffic, not the 12 B/entry matrix stream, is
// what the kernel waits for (measured: 20 % of the HBM peak).  The tiled form cuts the INPUT index range into blocks of
// kTileIn entries (64 KB of doubles: one LDS image) and the OUTPUT index range into tiles of kTileOut (one output per
// thread of a 1024-thread workgroup).  A workgroup owns one output tile (and one of S ranges of input blocks): for each
// input block it stages the block in LDS, then every thread walks the entries of ITS output that fall into the block --
// stored contiguously per (tile, block), output after output, as (uint16 local input index, f64 value): 10 B per entry
// instead of 12 -- and gathers from LDS.  Sums keep a fixed order (block after block, entries in storage order), so
// results stay bitwise reproducible.  The same kernel serves A x (outputs = rows, from the CSR copy) and A'y (outputs =
// columns, from the CSC mirror); partial sums of the S ranges are added in order by the step's epilogue kernel.
constexpr int kTileThreads = 1024;
constexpr int kTilePer = 4;                              // outputs per thread
constexpr int kTileOut = kTileThreads * kTilePer;        // outputs per tile
constexpr int kTileIn = 8192;

// (TiledMat: types.hpp)

// Per input block the workgroup (a) stages the block of the input vector in LDS, (b) multiplies the segment's entries
// with their LDS-gathered inputs ENTRY-parallel -- thread t takes entries t, t + 1024, ...: perfectly coalesced, eight
// independent loads per array and lane in flight, no divergence over row lengths -- leaving the products in an LDS chunk,
// and (c) adds up each output's products OUTPUT-parallel from LDS (thread t owns the outputs t, t + 1024, ... of the
// tile; a cell holds 2-3 entries on the cut matrices).  (An earlier form walked each output's entries straight from
// global memory: its divergent, dependent tail loads left the kernel latency-bound at 27 % of the HBM peak.)
constexpr int kTileChunk = 2048;                         // products per LDS chunk (16 KB): 80 KB of LDS per workgroup, two per CU
constexpr int kTileCE = kTileChunk / kTileThreads;       // entries per lane and chunk
// The (tile, block) units, tile-major, are dealt out to the persistent grid in contiguous, equal ranges (two workgroups per
// CU whatever the number of tiles); a tile's partial sums come from the consecutive workgroups whose ranges touch it
// ("pieces") and are added in that order by the epilogue.
__device__ __forceinline__ int64_t tile_unit_begin(int64_t w, int64_t U, int64_t G) { return w * U / G; }
__device__ __forceinline__ int64_t tile_unit_owner(int64_t u, int64_t U, int64_t G) { return ((u + 1) * G - 1) / U; }

constexpr int kTileRing = 4;                             // sub-chunks requested ahead of their use (register ring)

// position in the WG's stream of sub-chunks: unit u, first entry ch of the sub-chunk inside the unit's segment
struct TileCursor {
    int64_t u, seg;
    int ch, nseg;
    __device__ __forceinline__ void open(const int64_t* __restrict__ segstart, int64_t u_, int64_t u_end) {
        u = u_; ch = 0;
        if (u < u_end) { seg = segstart[u]; nseg = (int)(segstart[u + 1] - seg); } else { seg = 0; nseg = 0; }
    }
    // every unit has at least one (possibly empty) sub-chunk
    __device__ __forceinline__ void next(const int64_t* __restrict__ segstart, int64_t u_end) {
        ch += kTileChunk;
        if (ch >= nseg) open(segstart, u + 1, u_end);
    }
};

// The matrix entries of a workgroup's unit range are ONE contiguous stream (units are stored tile-major); it is consumed
// in sub-chunks of kTileChunk entries that never straddle a unit.  A ring of kTileRing register slots keeps the next
// sub-chunks' loads in flight (issued right after a slot's products are written), so the ~2 us HBM latency is covered by
// four sub-steps of work instead of stalling every one of them.
static __global__ __launch_bounds__(kTileThreads, 8) void k_spmv_tiled(int64_t n_out, int64_t n_in, int64_t tiles, TiledMat Tm,
                                                                const double* __restrict__ in, double* __restrict__ part) {
    __shared__ double xs[kTileIn];
    __shared__ double prod[kTileChunk];
    const int64_t U = tiles * Tm.nb_in, G = gridDim.x, w = blockIdx.x;
    const int64_t u_beg = tile_unit_begin(w, U, G), u_end = tile_unit_begin(w + 1, U, G);
    if (u_beg >= u_end) return;
    const int t = threadIdx.x;
    double acc[kTilePer];
    int64_t cur = -1;
    double rv[kTileRing][kTileCE];
    int rix[kTileRing][kTileCE];
    TileCursor pf, cs;
    pf.open(Tm.segstart, u_beg, u_end);
    cs.open(Tm.segstart, u_beg, u_end);
    auto request = [&](auto slot_c) {                     // loads of the sub-chunk at the prefetch cursor into ring slot `slot`
        constexpr int slot = decltype(slot_c)::value;
        const bool live = pf.u < u_end;
#pragma unroll
        for (int i = 0; i < kTileCE; ++i) {
            const int e = pf.ch + i * kTileThreads + t;
            const bool on = live && e < pf.nseg;
            rv[slot][i] = on ? Tm.val[pf.seg + e] : 0.0;
            rix[slot][i] = on ? (int)Tm.idx[pf.seg + e] : 0;
        }
        if (live) pf.next(Tm.segstart, u_end);
    };
    request(std::integral_constant<int, 0>{});
    request(std::integral_constant<int, 1>{});
    request(std::integral_constant<int, 2>{});
    request(std::integral_constant<int, 3>{});
    uint32_t ee[kTilePer];                                // (first entry | end entry << 16) of this thread's outputs in the unit
    auto flush = [&]() {
        const int64_t piece = w - tile_unit_owner(cur * Tm.nb_in, U, G);
#pragma unroll
        for (int k = 0; k < kTilePer; ++k) {
            const int64_t o = cur * kTileOut + k * kTileThreads + t;
            if (o < n_out) part[piece * n_out + o] = acc[k];
        }
    };
    auto step = [&](auto slot_c) {
        constexpr int slot = decltype(slot_c)::value;
        if (cs.ch == 0) {                                 // first sub-chunk of unit cs.u: stage its block of the input vector
            const int64_t tile = cs.u / Tm.nb_in;
            const int b = (int)(cs.u - tile * Tm.nb_in);
            if (tile != cur) {
                if (cur >= 0) flush();
                cur = tile;
#pragma unroll
                for (int k = 0; k < kTilePer; ++k) acc[k] = 0.0;
            }
            // xs is only read between the two barriers of a sub-step, so it is free here
            const int64_t c0 = (int64_t)b * kTileIn;
#pragma unroll
            for (int h = 0; h < 2; ++h) {                                // two batches of four loads in flight (register budget: 64)
                double xf[kTileIn / kTileThreads / 2];
#pragma unroll
                for (int i = 0; i < kTileIn / kTileThreads / 2; ++i) {
                    const int64_t c = c0 + (h * 4 + i) * kTileThreads + t;
                    xf[i] = in[c < n_in ? c : n_in - 1];
                }
#pragma unroll
                for (int i = 0; i < kTileIn / kTileThreads / 2; ++i)
                    xs[(h * 4 + i) * kTileThreads + t] = (c0 + (h * 4 + i) * kTileThreads + t < n_in) ? xf[i] : 0.0;
            }
            const uint16_t* bp = Tm.bptr + cs.u * (kTileOut + 1);
            uint16_t q0[kTilePer], q1[kTilePer];
#pragma unroll
            for (int k = 0; k < kTilePer; ++k) { q0[k] = bp[k * kTileThreads + t]; q1[k] = bp[k * kTileThreads + t + 1]; }
#pragma unroll
            for (int k = 0; k < kTilePer; ++k) ee[k] = (uint32_t)q0[k] | ((uint32_t)q1[k] << 16);
        }
        __syncthreads();                                  // xs staged; products of the previous sub-step consumed
#pragma unroll
        for (int i = 0; i < kTileCE; ++i) prod[i * kTileThreads + t] = rv[slot][i] * xs[rix[slot][i]];
        request(slot_c);                                  // the slot is free: next sub-chunk, kTileRing sub-steps ahead
        __syncthreads();
        const int ch = cs.ch;
#pragma unroll
        for (int k = 0; k < kTilePer; ++k) {
            const int e0 = (int)(ee[k] & 0xFFFFu), e1 = (int)(ee[k] >> 16);
            const int lo = e0 > ch ? e0 : ch;
            const int hi = e1 < ch + kTileChunk ? e1 : ch + kTileChunk;
            for (int e = lo; e < hi; ++e) acc[k] += prod[e - ch];
        }
        cs.next(Tm.segstart, u_end);
    };
    while (cs.u < u_end) {
        step(std::integral_constant<int, 0>{});
        if (cs.u >= u_end) break;
        step(std::integral_constant<int, 1>{});
        if (cs.u >= u_end) break;
        step(std::integral_constant<int, 2>{});
        if (cs.u >= u_end) break;
        step(std::integral_constant<int, 3>{});
    }
    if (cur >= 0) flush();
}

// sum of the pieces of output o, in piece order (pcnt[tile] = number of workgroups whose unit range touches the tile)
__device__ __forceinline__ double tile_pieces_sum(const double* __restrict__ part, int64_t o, int64_t n_out, const int32_t* __restrict__ pcnt) {
    const int np = pcnt[o / kTileOut];
    double acc = 0.0;
    int p = 0;
    // eight pieces requested before the first is added (the x-step has ~20 pieces per output and only 1.5 wavefronts per SIMD:
    // one round trip per piece made the epilogue latency-bound); the additions keep the piece order
    for (; p + 8 <= np; p += 8) {
        double v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = part[(int64_t)(p + k) * n_out + o];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += v[k];
    }
    for (; p < np; ++p) acc += part[(int64_t)p * n_out + o];
    return acc;
}
// epilogues of the tiled steps: the pieces in order, then exactly the arithmetic of k_pdhg_x / k_pdhg_y
static __global__ __launch_bounds__(kBlock) void k_x_epilogue(int64_t n, const int32_t* __restrict__ pcnt, const double* __restrict__ part,
                                                       double* __restrict__ x, const double* __restrict__ x0, double* __restrict__ xbar,
                                                       const double* __restrict__ c, const double* __restrict__ l,
                                                       const double* __restrict__ u, double tau, double w, double rho) {
    const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= n) return;
    const double xv = x[j], cj = c[j], lj = l[j], uj = u[j], x0j = x0[j];
    const double acc = tile_pieces_sum(part, j, n, pcnt);
    const double xtv = clampd(xv - tau * (cj - acc), lj, uj);
    xbar[j] = 2.0 * xtv - xv;
    x[j] = w * ((1.0 + rho) * xtv - rho * xv) + (1.0 - w) * x0j;
}
static __global__ __launch_bounds__(kBlock) void k_y_epilogue(int64_t m, const int32_t* __restrict__ pcnt, const double* __restrict__ part,
                                                       const int64_t* __restrict__ rowptr, int64_t long_thresh, double* __restrict__ y,
                                                       const double* __restrict__ y0, const double* __restrict__ lo,
                                                       const double* __restrict__ hi, double sigma, double w, double rho) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= m) return;
    if (rowptr && rowptr[i + 1] - rowptr[i] > long_thresh) return;      // served by k_pdhg_y_long (rowptr == NULL: no long rows)
    const double yv = y[i], loi = lo[i], hii = hi[i], y0i = y0[i];
    const double acc = tile_pieces_sum(part, i, m, pcnt);
    const double v = yv - sigma * acc;
    const double ytv = v + sigma * clampd(-v / sigma, loi, hii);
    y[i] = w * ((1.0 + rho) * ytv - rho * yv) + (1.0 - w) * y0i;
}

// ---- check iteration on the tiled copy: four tiled passes (A'y, A xt, A x, A'yt) with element-wise epilogues; the sums are
// those of k_pdhg_x<G, false> / k_pdhg_y_chk / k_chk_cols (the CSR check kernels cost 310 us per check on cfg4's large
// LPs, 4.5 plain iterations; this form 150 us) ------------------------------------------------------------------------
static __global__ __launch_bounds__(kBlock) void k_x_epilogue_chk(int64_t n, const int32_t* __restrict__ pcnt, const double* __restrict__ part,
                                                           const double* __restrict__ x, double* __restrict__ xt,
                                                           const double* __restrict__ c, const double* __restrict__ l,
                                                           const double* __restrict__ u, double tau) {
    const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= n) return;
    xt[j] = clampd(x[j] - tau * (c[j] - tile_pieces_sum(part, j, n, pcnt)), l[j], u[j]);
}
static __global__ __launch_bounds__(kBlock) void k_tile_vec(int64_t n_out, const int32_t* __restrict__ pcnt, const double* __restrict__ part,
                                                     double* __restrict__ out) {
    const int64_t o = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (o < n_out) out[o] = tile_pieces_sum(part, o, n_out, pcnt);
}
// part holds the pieces of A x, axt_v the vector A xt
static __global__ __launch_bounds__(kBlock) void k_y_epilogue_chk(int64_t m, const int32_t* __restrict__ pcnt, const double* __restrict__ part,
                                                           const double* __restrict__ axt_v, const int64_t* __restrict__ rowptr,
                                                           int64_t long_thresh, const double* __restrict__ y, const double* __restrict__ y0,
                                                           double* __restrict__ yt, const double* __restrict__ lo,
                                                           const double* __restrict__ hi, const double* __restrict__ dr, double sigma,
                                                           double* __restrict__ partials) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    ChkAcc a; a.init();
    const bool mine = i < m && !(rowptr && rowptr[i + 1] - rowptr[i] > long_thresh);      // long rows: k_pdhg_y_long<true>
    if (mine) {
        const double yv = y[i], loi = lo[i], hii = hi[i], axt = axt_v[i];
        const double axk = tile_pieces_sum(part, i, m, pcnt);
        const double v = yv - sigma * (2.0 * axt - axk);
        const double ytv = v + sigma * clampd(-v / sigma, loi, hii);
        yt[i] = ytv;
        chk_row_accumulate(a, ytv, yv, y0[i], axt, axk, loi, hii, dr[i]);
    }
    chk_block_store<kBlock, kChkRowMask>(a, partials);
}

// ---- building the tiled copy (once per LP solve, after the scaling): count -> scan per (tile, block) -> fill ----------
// cnt is zeroed, shaped like bptr; thread o owns the shorts [.. + t + 1] of its tile's blocks (no atomics needed)
static __global__ __launch_bounds__(kBlock) void k_tile_count(int64_t n_out, const int64_t* __restrict__ ptr, const int32_t* __restrict__ idx,
                                                       int nb_in, int64_t skip_longer, uint16_t* __restrict__ cnt,
                                                       int32_t* __restrict__ overflow) {
    const int64_t o = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (o >= n_out) return;
    const int64_t beg = ptr[o], end = ptr[o + 1];
    if (end - beg > skip_longer) return;
    const int64_t tile = o / kTileOut;
    const int t = (int)(o - tile * kTileOut);
    for (int64_t e = beg; e < end; ++e) {
        const int b = idx[e] / kTileIn;
        uint16_t* p = cnt + ((tile * nb_in + b) * (kTileOut + 1) + t + 1);
        const uint16_t c = *p;
        if (c == 0xFFFF) { atomicOr(overflow, 1); return; }
        *p = (uint16_t)(c + 1);
    }
}
// one workgroup per (tile, block): in-place inclusive scan of the kTileOut counts -> offsets; segment total out.
// Thread t scans the kTilePer CONSECUTIVE counts [t * kTilePer, (t + 1) * kTilePer), then the thread totals are scanned.
static __global__ __launch_bounds__(kTileThreads) void k_tile_scan(uint16_t* __restrict__ cnt, int64_t* __restrict__ segtot,
                                                            int32_t* __restrict__ overflow) {
    __shared__ uint32_t wsum[kTileThreads / 64];
    const int64_t tb = blockIdx.x;
    uint16_t* p = cnt + tb * (kTileOut + 1);
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    uint32_t c[kTilePer];
    uint32_t tot = 0;
#pragma unroll
    for (int k = 0; k < kTilePer; ++k) { tot += p[t * kTilePer + k + 1]; c[k] = tot; }
    uint32_t v = tot;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t u = __shfl_up(v, off, 64);
        if (lane >= off) v += u;
    }
    if (lane == 63) wsum[wv] = v;
    __syncthreads();
    uint32_t base = 0;
    for (int k = 0; k < wv; ++k) base += wsum[k];
    v += base;                      // inclusive over threads
    const uint32_t excl = v - tot;
    if (v > 0xFFFFu) atomicOr(overflow, 1);
#pragma unroll
    for (int k = 0; k < kTilePer; ++k) p[t * kTilePer + k + 1] = (uint16_t)(excl + c[k]);
    if (t == kTileThreads - 1) segtot[tb] = (int64_t)v;
}
static __global__ __launch_bounds__(kBlock) void k_tile_fill(int64_t n_out, const int64_t* __restrict__ ptr, const int32_t* __restrict__ idx,
                                                      const double* __restrict__ val, int nb_in, int64_t skip_longer,
                                                      const uint16_t* __restrict__ bptr, uint16_t* __restrict__ cur,
                                                      const int64_t* __restrict__ segstart, uint16_t* __restrict__ tidx,
                                                      double* __restrict__ tval) {
    const int64_t o = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (o >= n_out) return;
    const int64_t beg = ptr[o], end = ptr[o + 1];
    if (end - beg > skip_longer) return;
    const int64_t tile = o / kTileOut;
    const int t = (int)(o - tile * kTileOut);
    for (int64_t e = beg; e < end; ++e) {
        const int c = idx[e];
        const int b = c / kTileIn;
        const int64_t tb = tile * nb_in + b;
        uint16_t* q = cur + (tb * (kTileOut + 1) + t);
        const uint16_t k = *q;
        *q = (uint16_t)(k + 1);
        const int64_t pos = segstart[tb] + bptr[tb * (kTileOut + 1) + t] + k;
        tidx[pos] = (uint16_t)(c - b * kTileIn);
        tval[pos] = val[e];
    }
}

// The same two passes for outputs whose entries come in ASCENDING input order (the rows of the LP -- linear rows and cuts are
// stored by ascending column -- and the columns of the mirror, by ascending row): an output's entries of one input block are
// then one RUN, so its count is a register (one store per run) and the fill needs no per-cell cursor -- the count pass had a
// global read-modify-write per entry, the fill one plus two dependent loads (204 + 653 us per orientation on cfg4's LP).  An
// output that is not ascending sets bit 1 of *overflow; the fill then does nothing and the host repeats the build with the
// general kernels above.
static __global__ __launch_bounds__(kBlock) void k_tile_count_sorted(int64_t n_out, const int64_t* __restrict__ ptr, const int32_t* __restrict__ idx,
                                                              int nb_in, int64_t skip_longer, uint16_t* __restrict__ cnt,
                                                              int32_t* __restrict__ overflow) {
    const int64_t o = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (o >= n_out) return;
    const int64_t beg = ptr[o], end = ptr[o + 1];
    if (end - beg > skip_longer) return;
    const int64_t tile = o / kTileOut;
    const int t = (int)(o - tile * kTileOut);
    int prev = -1;
    uint32_t run = 0;
    for (int64_t e = beg; e < end; ++e) {
        const int b = idx[e] / kTileIn;
        if (b != prev) {
            if (b < prev) { atomicOr(overflow, 2); return; }
            if (prev >= 0) cnt[(tile * nb_in + prev) * (kTileOut + 1) + t + 1] = (uint16_t)run;
            prev = b;
            run = 0;
        }
        if (++run > 0xFFFFu) { atomicOr(overflow, 1); return; }
    }
    if (prev >= 0) cnt[(tile * nb_in + prev) * (kTileOut + 1) + t + 1] = (uint16_t)run;
}
static __global__ __launch_bounds__(kBlock) void k_tile_fill_sorted(int64_t n_out, const int64_t* __restrict__ ptr, const int32_t* __restrict__ idx,
                                                             const double* __restrict__ val, int nb_in, int64_t skip_longer,
                                                             const uint16_t* __restrict__ bptr, const int64_t* __restrict__ segstart,
                                                             const int32_t* __restrict__ overflow, uint16_t* __restrict__ tidx,
                                                             double* __restrict__ tval) {
    const int64_t o = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (o >= n_out || *overflow != 0) return;              // (counts of a non-ascending or overflowing build are not to be trusted)
    const int64_t beg = ptr[o], end = ptr[o + 1];
    if (end - beg > skip_longer) return;
    const int64_t tile = o / kTileOut;
    const int t = (int)(o - tile * kTileOut);
    int prev = -1;
    int64_t pos = 0;
    for (int64_t e = beg; e < end; ++e) {
        const int c = idx[e];
        const int b = c / kTileIn;
        if (b != prev) {
            prev = b;
            const int64_t tb = tile * nb_in + b;
            pos = segstart[tb] + bptr[tb * (kTileOut + 1) + t];
        }
        tidx[pos] = (uint16_t)(c - b * kTileIn);
        tval[pos] = val[e];
        ++pos;
    }
}

// ---------------------------------------------------------- diagonal scaling ------
// Ruiz / Pock-Chambolle passes on the UNSCALED matrix with the current dr, dc:
//   mode 0: out_i = dr_i * max_e |a_e| dc_col(e)     mode 1: out_i = dr_i * sum_e |a_e| dc_col(e)
// (the same kernel serves columns through the CSC mirror with the roles of dr/dc swapped)
template <int G>
__global__ __launch_bounds__(kBlock) void k_scale_stat(int64_t m, const int64_t* __restrict__ ptr,
                                                       const int32_t* __restrict__ idx, const double* __restrict__ val,
                                                       const double* __restrict__ dself, const double* __restrict__ dother,
                                                       int mode, double* __restrict__ out) {
    const int64_t i = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / G;
    const int lane = threadIdx.x & (G - 1);
    if (i >= m) return;
    double acc = 0.0;
    for (int64_t e = ptr[i] + lane; e < ptr[i + 1]; e += G) {
        const double v = fabs(val[e]) * dother[idx[e]];
        acc = mode ? acc + v : fmax(acc, v);
    }
    acc = mode ? group_sum<G>(acc) : group_max<G>(acc);
    if (lane == 0) out[i] = dself[i] * acc;
}
// statistic and update in one launch: dnew_i = dself_i / sqrt(dself_i * stat_i) (unchanged where the statistic is 0 or not
// finite) -- the arithmetic of k_scale_stat followed by k_scale_apply2, without the third launch of every pass.  Rows and
// columns both read the OLD scalings and write new arrays, which the host swaps in after the pass.
// ... and both sides of a pass in ONE launch: they do not depend on each other -- blocks [0, br) take the rows with gr lanes each, the
// rest the columns with gc lanes (lane counts at run time; the butterfly runs in the order of group_sum / group_max: the same bits as
// the one-side-per-launch form of rounds 2-4).  Nine launches and nine kernel boundaries less per LP solve.  (cap: see k_scale_apply2.)
__device__ __forceinline__ void scale_stat_upd_side(int64_t t, int g, int64_t m, const int64_t* __restrict__ ptr,
                                                    const int32_t* __restrict__ idx, const double* __restrict__ val,
                                                    const double* __restrict__ dself, const double* __restrict__ dother,
                                                    int mode, double* __restrict__ dnew, double cap) {
    const int64_t i = t / g;
    const int lane = (int)(t & (g - 1));
    if (i >= m) return;
    double acc = 0.0;
    for (int64_t e = ptr[i] + lane; e < ptr[i + 1]; e += g) {
        const double v = fabs(val[e]) * dother[idx[e]];
        acc = mode ? acc + v : fmax(acc, v);
    }
    for (int off = g >> 1; off > 0; off >>= 1) {
        const double o = __shfl_xor(acc, off, 64);
        acc = mode ? acc + o : fmax(acc, o);
    }
    if (lane == 0) {
        double d = dself[i];
        const double st = d * acc;
        if (st > 0.0 && isfinite(st)) d /= sqrt(st);
        dnew[i] = fmin(d, cap);
    }
}
static __global__ __launch_bounds__(kBlock) void k_scale_stat_upd_both(int64_t m, const int64_t* __restrict__ rptr, const int32_t* __restrict__ ridx,
                                                                const double* __restrict__ rval, int64_t n, const int64_t* __restrict__ cptr,
                                                                const int32_t* __restrict__ cidx, const double* __restrict__ cval,
                                                                const double* __restrict__ dr, const double* __restrict__ dc, int mode,
                                                                double* __restrict__ dr_new, double* __restrict__ dc_new, double cap_c,
                                                                int gr, int gc, int br) {
    if ((int)blockIdx.x < br)
        scale_stat_upd_side((int64_t)blockIdx.x * kBlock + threadIdx.x, gr, m, rptr, ridx, rval, dr, dc, mode, dr_new, __builtin_inf());
    else
        scale_stat_upd_side((int64_t)(blockIdx.x - br) * kBlock + threadIdx.x, gc, n, cptr, cidx, cval, dc, dr, mode, dc_new, cap_c);
}
// the same with the long rows left to k_scale_stat_long (one 1024-thread workgroup per long row)
template <int G>
__global__ __launch_bounds__(kBlock) void k_scale_stat_skip(int64_t m, const int64_t* __restrict__ ptr,
                                                            const int32_t* __restrict__ idx, const double* __restrict__ val,
                                                            const double* __restrict__ dself, const double* __restrict__ dother,
                                                            int mode, double* __restrict__ out, int64_t skip_longer) {
    const int64_t i = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / G;
    const int lane = threadIdx.x & (G - 1);
    if (i >= m) return;
    const int64_t beg = ptr[i], end = ptr[i + 1];
    if (end - beg > skip_longer) return;
    double acc = 0.0;
    for (int64_t e = beg + lane; e < end; e += G) {
        const double v = fabs(val[e]) * dother[idx[e]];
        acc = mode ? acc + v : fmax(acc, v);
    }
    acc = mode ? group_sum<G>(acc) : group_max<G>(acc);
    if (lane == 0) out[i] = dself[i] * acc;
}
static __global__ __launch_bounds__(1024) void k_scale_stat_long(const int32_t* __restrict__ rows, const int64_t* __restrict__ ptr,
                                                          const int32_t* __restrict__ idx, const double* __restrict__ val,
                                                          const double* __restrict__ dself, const double* __restrict__ dother,
                                                          int mode, double* __restrict__ out) {
    __shared__ double sh[1024 / 64];
    const int64_t i = rows[blockIdx.x];
    double acc = 0.0;
    for (int64_t e = ptr[i] + threadIdx.x; e < ptr[i + 1]; e += 1024) {
        const double v = fabs(val[e]) * dother[idx[e]];
        acc = mode ? acc + v : fmax(acc, v);
    }
    acc = mode ? group_sum<64>(acc) : group_max<64>(acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0;
        for (int k = 0; k < 1024 / 64; ++k) a = mode ? a + sh[k] : fmax(a, sh[k]);
        out[i] = dself[i] * a;
    }
}
static __global__ __launch_bounds__(kBlock) void k_scale_apply(int64_t m, double* __restrict__ d, const double* __restrict__ stat) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= m) return;
    const double s = stat[i];
    if (s > 0.0 && isfinite(s)) d[i] /= sqrt(s);
}
// rows and columns in one launch.  cap_c bounds the column factors: in the epigraph-shifted working form a column that
// occurs only in the dense cuts holds nothing but DIFFERENCES of nearly equal derivatives (1e-7 ... rounding noise); the
// equilibration would blow such a column up by that factor and its cost with it (||c^|| 1e17 seen: primal weight and
// tolerances meaningless).  A smaller factor than Pock-Chambolle's keeps ||A^||_2 <= 1.  (inf for every other solve.)
static __global__ __launch_bounds__(kBlock) void k_scale_apply2(int64_t m, double* __restrict__ dr, const double* __restrict__ sr, int64_t n,
                                                        double* __restrict__ dc, const double* __restrict__ sc, double cap_c) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < m) { const double s = sr[i]; if (s > 0.0 && isfinite(s)) dr[i] /= sqrt(s); }
    if (i < n) { const double s = sc[i]; if (s > 0.0 && isfinite(s)) dc[i] = fmin(dc[i] / sqrt(s), cap_c); }
}
// out[newidx[r]] = in[r] for the kept rows (purge)
static __global__ __launch_bounds__(kBlock) void k_compact_vec(int64_t m, const int64_t* __restrict__ keep, const int64_t* __restrict__ newidx,
                                                       const double* __restrict__ in, double* __restrict__ out) {
    const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (r < m && keep[r]) out[newidx[r]] = in[r];
}
// sval_e = dself_i * a_e * dother_idx(e)
template <int G>
__global__ __launch_bounds__(kBlock) void k_scale_vals(int64_t m, const int64_t* __restrict__ ptr,
                                                       const int32_t* __restrict__ idx, const double* __restrict__ val,
                                                       const double* __restrict__ dself, const double* __restrict__ dother,
                                                       double* __restrict__ sval) {
    const int64_t i = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / G;
    const int lane = threadIdx.x & (G - 1);
    if (i >= m) return;
    const double di = dself[i];
    for (int64_t e = ptr[i] + lane; e < ptr[i + 1]; e += G) sval[e] = di * val[e] * dother[idx[e]];
}

// the scaled values of the row copy and of the column mirror in one launch (blocks [0, br): rows with gr lanes, the rest: columns)
static __global__ __launch_bounds__(kBlock) void k_scale_vals_both(int64_t m, const int64_t* __restrict__ rptr, const int32_t* __restrict__ ridx,
                                                            const double* __restrict__ rval, int64_t n, const int64_t* __restrict__ cptr,
                                                            const int32_t* __restrict__ cidx, const double* __restrict__ cval,
                                                            const double* __restrict__ dr, const double* __restrict__ dc,
                                                            double* __restrict__ rsval, double* __restrict__ csval, int gr, int gc, int br) {
    const bool rows = (int)blockIdx.x < br;
    const int g = rows ? gr : gc;
    const int64_t t = (int64_t)(rows ? blockIdx.x : blockIdx.x - br) * kBlock + threadIdx.x;
    const int64_t i = t / g;
    const int lane = (int)(t & (g - 1));
    if (i >= (rows ? m : n)) return;
    const int64_t* ptr = rows ? rptr : cptr;
    const int32_t* idx = rows ? ridx : cidx;
    const double* val = rows ? rval : cval;
    const double* dother = rows ? dc : dr;
    double* sval = rows ? rsval : csval;
    const double di = rows ? dr[i] : dc[i];
    for (int64_t e = ptr[i] + lane; e < ptr[i + 1]; e += g) sval[e] = di * val[e] * dother[idx[e]];
}

// scaled problem vectors
//   mode 0 (LP):        ch = s c dc,  lh = l/dc, uh = u/dc, xh = clip(x/dc)
//   mode 1 (recession): ch = s c dc,  box = finite? 0 : -+scale_j  (oracle/lp.py recession_ray)
struct PrepCols { int64_t n; const double *c, *l, *u, *dc, *x, *box; double sgn; int mode; double *ch, *lh, *uh, *xh; };
struct PrepRows { int64_t m; const double *lo, *hi, *dr, *y; int mode; double *loh, *hih, *yh; };
__device__ __forceinline__ void prep_col(int64_t j, const PrepCols& a) {
    if (j >= a.n) return;
    const double d = a.dc[j];
    a.ch[j] = a.sgn * a.c[j] * d;
    double lo = a.l[j], hi = a.u[j];
    if (a.mode == 1) {
        const double b = a.box ? a.box[j] : 1.0;
        lo = isfinite(lo) ? 0.0 : -b;
        hi = isfinite(hi) ? 0.0 : b;
    }
    a.lh[j] = lo / d;
    a.uh[j] = hi / d;
    a.xh[j] = clampd((a.mode == 1 ? 0.0 : a.x[j]) / d, lo / d, hi / d);
}
__device__ __forceinline__ void prep_row(int64_t i, const PrepRows& r) {
    if (i >= r.m) return;
    double a = r.lo[i], b = r.hi[i];
    if (a != a) a = -__builtin_inf();   // NaN bound = vacuous side (oracle/lp.py add_rows)
    if (b != b) b = __builtin_inf();
    if (r.mode == 1) {
        a = isfinite(a) ? 0.0 : -__builtin_inf();
        b = isfinite(b) ? 0.0 : __builtin_inf();
    }
    const double d = r.dr[i];
    r.loh[i] = a * d;
    r.hih[i] = b * d;
    r.yh[i] = (r.mode == 1) ? 0.0 : r.y[i] / d;
}
static __global__ __launch_bounds__(kBlock) void k_prep_cols(int64_t n, const double* __restrict__ c, const double* __restrict__ l,
                                                      const double* __restrict__ u, const double* __restrict__ dc,
                                                      const double* __restrict__ x, const double* __restrict__ box,
                                                      double sgn, int mode, double* __restrict__ ch,
                                                      double* __restrict__ lh, double* __restrict__ uh,
                                                      double* __restrict__ xh) {
    prep_col((int64_t)blockIdx.x * kBlock + threadIdx.x, PrepCols{n, c, l, u, dc, x, box, sgn, mode, ch, lh, uh, xh});
}
static __global__ __launch_bounds__(kBlock) void k_prep_rows(int64_t m, const double* __restrict__ lo, const double* __restrict__ hi,
                                                      const double* __restrict__ dr, const double* __restrict__ y, int mode,
                                                      double* __restrict__ loh, double* __restrict__ hih,
                                                      double* __restrict__ yh) {
    prep_row((int64_t)blockIdx.x * kBlock + threadIdx.x, PrepRows{m, lo, hi, dr, y, mode, loh, hih, yh});
}
// columns and rows of a solve's start in one launch (thread i takes column i and row i)
static __global__ __launch_bounds__(kBlock) void k_prep_both(PrepCols C, PrepRows R) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    prep_col(i, C);
    prep_row(i, R);
}
// x = xh . dc and y = yh . dr in one launch
static __global__ __launch_bounds__(kBlock) void k_unscale2(int64_t n, const double* __restrict__ xh, const double* __restrict__ dc, double* __restrict__ x,
                                                     int64_t m, const double* __restrict__ yh, const double* __restrict__ dr, double* __restrict__ y) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) x[i] = xh[i] * dc[i];
    if (i < m) y[i] = yh[i] * dr[i];
}
static __global__ __launch_bounds__(kBlock) void k_unscale(int64_t n, const double* __restrict__ zh, const double* __restrict__ d,
                                                    double* __restrict__ z) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) z[i] = zh[i] * d[i];
}

static __global__ __launch_bounds__(kBlock) void k_div_vec(int64_t n, const double* __restrict__ a, const double* __restrict__ d,
                                                    double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) out[i] = a[i] / d[i];
}

// ---------------------------------------------------------- generic vector ops ----
template <int G>
__global__ __launch_bounds__(kBlock) void k_spmv(int64_t m, SpMat A, const double* __restrict__ v, double* __restrict__ out) {
    const int64_t i = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / G;
    const int lane = threadIdx.x & (G - 1);
    if (i >= m) return;
    double acc = 0.0;
    for (int64_t e = A.ptr[i] + lane; e < A.ptr[i + 1]; e += G) acc += A.val[e] * v[A.idx[e]];
    acc = group_sum<G>(acc);
    if (lane == 0) out[i] = acc;
}
// Row-side products of a matrix with a few LONG rows (dense epigraph cuts): the lane groups skip them and one 1024-thread
// workgroup per long row does them (k_spmv_long) -- a 1e4-entry row given to a 32-lane group made the whole launch wait 60 us.
template <int G>
__global__ __launch_bounds__(kBlock) void k_spmv_skip(int64_t m, SpMat A, const double* __restrict__ v, double* __restrict__ out,
                                                      int64_t skip_longer) {
    const int64_t i = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / G;
    const int lane = threadIdx.x & (G - 1);
    if (i >= m) return;
    const int64_t beg = A.ptr[i], end = A.ptr[i + 1];
    if (end - beg > skip_longer) return;
    double acc = 0.0;
    for (int64_t e = beg + lane; e < end; e += G) acc += A.val[e] * v[A.idx[e]];
    acc = group_sum<G>(acc);
    if (lane == 0) out[i] = acc;
}
static __global__ __launch_bounds__(1024) void k_spmv_long(const int32_t* __restrict__ rows, SpMat A, const double* __restrict__ v,
                                                    double* __restrict__ out) {
    __shared__ double sh[1024 / 64];
    const int64_t i = rows[blockIdx.x];
    double acc = 0.0;
    for (int64_t e = A.ptr[i] + threadIdx.x; e < A.ptr[i + 1]; e += 1024) acc += A.val[e] * v[A.idx[e]];
    acc = group_sum<64>(acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0;
        for (int k = 0; k < 1024 / 64; ++k) a += sh[k];
        out[i] = a;
    }
}
// partials[b] = sum over the block's grid-stride share of a_i * b_i  (b may alias a)
static __global__ __launch_bounds__(kBlock) void k_dot_partial(int64_t n, const double* __restrict__ a, const double* __restrict__ b,
                                                        double* __restrict__ partials) {
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) acc += a[i] * b[i];
    __shared__ double sh[kBlock / 64];
    acc = group_sum<64>(acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double v = 0.0;
        for (int k = 0; k < kBlock / 64; ++k) v += sh[k];
        partials[blockIdx.x] = v;
    }
}
// The four sums every LP solve needs before its first step -- ||A^||_F^2, ||c^||^2 and the finite parts of ||lo^||^2, ||hi^||^2 -- in ONE
// launch (+ one final): partials[q * gridDim + b].  Each sum runs over its array exactly as k_dot_partial / k_finite_sq_partial
// would (same grid, same stride, same reduction shape): the same bits, six launches less per solve.
static __global__ __launch_bounds__(kBlock) void k_setup_norms_partial(int64_t nnz, const double* __restrict__ aval, int64_t n,
                                                                const double* __restrict__ c, int64_t m, const double* __restrict__ lo,
                                                                const double* __restrict__ hi, double* __restrict__ partials) {
    const int64_t t0 = (int64_t)blockIdx.x * kBlock + threadIdx.x, stride = (int64_t)gridDim.x * kBlock;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int64_t i = t0; i < nnz; i += stride) acc[0] += aval[i] * aval[i];
    for (int64_t i = t0; i < n; i += stride) acc[1] += c[i] * c[i];
    for (int64_t i = t0; i < m; i += stride) { const double v = lo[i]; if (isfinite(v)) acc[2] += v * v; }
    for (int64_t i = t0; i < m; i += stride) { const double v = hi[i]; if (isfinite(v)) acc[3] += v * v; }
    __shared__ double sh[4][kBlock / 64];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const double v = group_sum<64>(acc[q]);
        if ((threadIdx.x & 63) == 0) sh[q][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        double v = 0.0;
        for (int k = 0; k < kBlock / 64; ++k) v += sh[threadIdx.x][k];
        partials[(int64_t)threadIdx.x * gridDim.x + blockIdx.x] = v;
    }
}
// block q: out[q] = sum of partials[q * nblocks .. ), in the order of k_sum_final
static __global__ __launch_bounds__(kRedBlocks) void k_sum_final_multi(const double* __restrict__ partials, int nblocks, double* __restrict__ out) {
    __shared__ double sh[kRedBlocks / 64];
    const double* p = partials + (int64_t)blockIdx.x * nblocks;
    double v = ((int)threadIdx.x < nblocks) ? p[threadIdx.x] : 0.0;
    v = group_sum<64>(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int k = 0; k < kRedBlocks / 64; ++k) t += sh[k];
        out[blockIdx.x] = t;
    }
}
// partials[b] = sum of log|a_i| (and |b_i|, b optional) over the finite non-zero entries, partials[gridDim + b] = their count:
// a magnitude statistic that a handful of outliers cannot move (see Engine::lp_solve_core, the initial primal weight)
static __global__ __launch_bounds__(kBlock) void k_logabs_partial(int64_t n, const double* __restrict__ a, const double* __restrict__ b,
                                                           double* __restrict__ partials) {
    double ls = 0.0, cnt = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const double u = fabs(a[i]);
        if (u > 0.0 && isfinite(u)) { ls += log(u); cnt += 1.0; }
        if (b) { const double v = fabs(b[i]); if (v > 0.0 && isfinite(v)) { ls += log(v); cnt += 1.0; } }
    }
    __shared__ double sh[2][kBlock / 64];
    ls = group_sum<64>(ls); cnt = group_sum<64>(cnt);
    if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = ls; sh[1][threadIdx.x >> 6] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double u = 0.0, v = 0.0;
        for (int k = 0; k < kBlock / 64; ++k) { u += sh[0][k]; v += sh[1][k]; }
        partials[blockIdx.x] = u; partials[gridDim.x + blockIdx.x] = v;
    }
}
// partials[b] = sum of (a_i d_i)^2: the squared norm of a vector in scaled coordinates without materialising it
static __global__ __launch_bounds__(kBlock) void k_scaled_sq_partial(int64_t n, const double* __restrict__ a, const double* __restrict__ d,
                                                              double* __restrict__ partials) {
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) { const double v = a[i] * d[i]; acc += v * v; }
    __shared__ double sh[kBlock / 64];
    acc = group_sum<64>(acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double v = 0.0;
        for (int k = 0; k < kBlock / 64; ++k) v += sh[k];
        partials[blockIdx.x] = v;
    }
}
static __global__ __launch_bounds__(kRedBlocks) void k_sum_final(const double* __restrict__ partials, int nblocks,
                                                          double* __restrict__ out) {
    __shared__ double sh[kRedBlocks / 64];
    double v = ((int)threadIdx.x < nblocks) ? partials[threadIdx.x] : 0.0;
    v = group_sum<64>(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int k = 0; k < kRedBlocks / 64; ++k) t += sh[k];
        out[0] = t;
    }
}
static __global__ __launch_bounds__(kBlock) void k_scale_vec(int64_t n, double* __restrict__ z, double s) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) z[i] *= s;
}
// out = a / sqrt(*normsq)  (normalisation of the power iteration without a host round trip)
static __global__ __launch_bounds__(kBlock) void k_normalize(int64_t n, const double* __restrict__ a, const double* __restrict__ normsq,
                                                      double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const double s = normsq[0];
    if (i < n) out[i] = (s > 0.0) ? a[i] / sqrt(s) : 0.0;
}
// the same with the second stage of the reduction folded in: every block adds up the kRedBlocks partials of ||a||^2 itself, in the
// order of k_sum_final (same bits), instead of reading the sum a launch of its own would have left
static __global__ __launch_bounds__(kRedBlocks) void k_normalize_sum(int64_t n, const double* __restrict__ a, const double* __restrict__ partials,
                                                              double* __restrict__ out) {
    __shared__ double sh[kRedBlocks / 64];
    __shared__ double s_tot;
    double v = partials[threadIdx.x];
    v = group_sum<64>(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int k = 0; k < kRedBlocks / 64; ++k) t += sh[k];
        s_tot = t;
    }
    __syncthreads();
    const double s = s_tot;
    const int64_t i = (int64_t)blockIdx.x * kRedBlocks + threadIdx.x;
    if (i < n) out[i] = (s > 0.0) ? a[i] / sqrt(s) : 0.0;
}
static __global__ __launch_bounds__(kBlock) void k_fill(int64_t n, double* __restrict__ z, double v) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) z[i] = v;
}
static __global__ __launch_bounds__(kBlock) void k_axpy_scaled(int64_t n, const double* __restrict__ a, double s, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) out[i] = a[i] * s;
}

// ---------------------------------------------------------- CSC mirror build ------
// key = (col << 32) | row, value = entry index; a radix sort by key orders every column
// by row index, so the column sums are summed in a fixed order (deterministic).
template <int G>
__global__ __launch_bounds__(kBlock) void k_csc_keys(int64_t m, const int64_t* __restrict__ rowptr,
                                                     const int32_t* __restrict__ col, uint64_t* __restrict__ keys,
                                                     uint32_t* __restrict__ vals, int64_t* __restrict__ colcount) {
    const int64_t i = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / G;
    const int lane = threadIdx.x & (G - 1);
    if (i >= m) return;
    for (int64_t e = rowptr[i] + lane; e < rowptr[i + 1]; e += G) {
        const uint32_t c = (uint32_t)col[e];
        keys[e] = ((uint64_t)c << 32) | (uint64_t)(uint32_t)i;
        vals[e] = (uint32_t)e;
        atomicAdd(reinterpret_cast<unsigned long long*>(&colcount[c]), 1ULL);
    }
}
// ---- append-only update of the mirror ------------------------------------------------------------------------------
// Between two purges rows are only ever APPENDED (src/model.jl:74-77: cuts are added, never changed), and an appended row has
// a larger index than every row already mirrored: column j of the new mirror is column j of the old one followed by j's
// entries in the new rows, in row order.  So instead of sorting all non-zeros again (three radix passes over 5e5 pairs per LP
// solve on cfg3) the old columns are shifted by the running count of new entries and the new entries dropped in behind
// them.  The mirror keeps, per position, the CSR entry it came from (perm): values are gathered through it, which is also
// what lets the epigraph-shifted working form refresh its values without touching the structure.
static __global__ __launch_bounds__(kBlock) void k_cscm_count(int64_t e0, int64_t nnz, const int32_t* __restrict__ col, int64_t* __restrict__ cnt) {
    const int64_t e = e0 + (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (e < nnz) atomicAdd(reinterpret_cast<unsigned long long*>(&cnt[col[e]]), 1ULL);
}
// old column j moves to old_ptr[j] + off[j] (off = exclusive scan of the new entries per column); G lanes per column
template <int G>
__global__ __launch_bounds__(kBlock) void k_cscm_move(int64_t n, const int64_t* __restrict__ old_ptr, const int64_t* __restrict__ off,
                                                      const int32_t* __restrict__ old_row, const uint32_t* __restrict__ old_perm,
                                                      int64_t* __restrict__ new_ptr, int32_t* __restrict__ new_row,
                                                      uint32_t* __restrict__ new_perm) {
    const int64_t j = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / G;
    const int lane = threadIdx.x & (G - 1);
    if (j > n) return;
    if (j == n) { if (lane == 0) new_ptr[n] = old_ptr[n] + off[n]; return; }
    const int64_t src = old_ptr[j], len = old_ptr[j + 1] - src, dst = src + off[j];
    for (int64_t k = lane; k < len; k += G) { new_row[dst + k] = old_row[src + k]; new_perm[dst + k] = old_perm[src + k]; }
    if (lane == 0) new_ptr[j] = dst;
}
// the entries of the new rows [m0, m) behind their columns' old entries, in arrival order (cursor zeroed) ...
template <int G>
__global__ __launch_bounds__(kBlock) void k_cscm_place(int64_t m0, int64_t m, const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                       const int64_t* __restrict__ old_ptr, const int64_t* __restrict__ off,
                                                       int64_t* __restrict__ cursor, int32_t* __restrict__ new_row,
                                                       uint32_t* __restrict__ new_perm) {
    const int64_t r = m0 + ((int64_t)blockIdx.x * kBlock + threadIdx.x) / G;
    const int lane = threadIdx.x & (G - 1);
    if (r >= m) return;
    for (int64_t e = rowptr[r] + lane; e < rowptr[r + 1]; e += G) {        // (a thread per row walked its 32 atomics one after the other: 57 us)
        const int32_t j = col[e];
        const int64_t k = (int64_t)atomicAdd(reinterpret_cast<unsigned long long*>(&cursor[j]), 1ULL);
        const int64_t pos = old_ptr[j + 1] + off[j] + k;
        new_row[pos] = (int32_t)r;
        new_perm[pos] = (uint32_t)e;
    }
}
// ... and put into row order column by column (a column gets 0.3 new entries per sweep on average: an insertion sort of a
// handful), so that every column sum keeps its fixed order whatever the arrival order was
static __global__ __launch_bounds__(kBlock) void k_cscm_order(int64_t n, const int64_t* __restrict__ old_ptr, const int64_t* __restrict__ off,
                                                       int32_t* __restrict__ new_row, uint32_t* __restrict__ new_perm) {
    const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= n) return;
    const int64_t cnt = off[j + 1] - off[j];
    if (cnt < 2) return;
    const int64_t b = old_ptr[j + 1] + off[j];
    for (int64_t a = 1; a < cnt; ++a) {
        const int32_t rr = new_row[b + a];
        const uint32_t pp = new_perm[b + a];
        int64_t q = a - 1;
        while (q >= 0 && (new_row[b + q] > rr || (new_row[b + q] == rr && new_perm[b + q] > pp))) {
            new_row[b + q + 1] = new_row[b + q]; new_perm[b + q + 1] = new_perm[b + q]; --q;
        }
        new_row[b + q + 1] = rr; new_perm[b + q + 1] = pp;
    }
}
static __global__ __launch_bounds__(kBlock) void k_csc_vals(int64_t nnz, const uint32_t* __restrict__ perm, const double* __restrict__ val,
                                                     double* __restrict__ cval) {
    const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (p < nnz) cval[p] = val[perm[p]];
}
static __global__ __launch_bounds__(kBlock) void k_csc_rows_perm(int64_t nnz, const uint64_t* __restrict__ keys, const uint32_t* __restrict__ perm,
                                                          int32_t* __restrict__ crow, uint32_t* __restrict__ cperm) {
    const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (p >= nnz) return;
    crow[p] = (int32_t)(keys[p] & 0xffffffffULL);
    cperm[p] = perm[p];
}

// ---- cut blocks for the exchange between GPUs, packed and unpacked on the device ------------------------------------
// One f64 block per rank: [rowptr[1:] (rebased) | col | val | lo | hi | global NL-row id], integers exact in f64 -- the
// layout katana_jl_amd/distributed.py::pack_block uses on the host path.  The block never leaves device memory: the engine
// writes it into the caller's send buffer, RCCL all-gathers it, the engine appends every rank's block from the receive
// buffers (ktn_lp_pack_rows_dev / ktn_lp_append_packed_dev).
static __global__ __launch_bounds__(kBlock) void k_pack_rows(int64_t nr, int64_t nz, const int64_t* __restrict__ rowptr /* at first_row */,
                                                      const int32_t* __restrict__ col, const double* __restrict__ val /* at base */,
                                                      const double* __restrict__ lo, const double* __restrict__ hi,
                                                      const int32_t* __restrict__ slots, int64_t id_offset, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t base = rowptr[0];
    if (i < nr) {
        out[i] = (double)(rowptr[i + 1] - base);
        out[nr + 2 * nz + i] = lo[i];
        out[2 * nr + 2 * nz + i] = hi[i];
        out[3 * nr + 2 * nz + i] = slots ? (double)(id_offset + slots[i]) : -1.0;
    }
    if (i < nz) {
        out[nr + i] = (double)col[i];
        out[nr + nz + i] = val[i];
    }
}
static __global__ __launch_bounds__(kBlock) void k_unpack_rows(int64_t nr, int64_t nz, const double* __restrict__ in, int64_t nnz0,
                                                        int64_t n_lp, int64_t* __restrict__ rowptr /* at M + 1 */,
                                                        int32_t* __restrict__ col, double* __restrict__ val /* at nnz0 */,
                                                        double* __restrict__ lo, double* __restrict__ hi /* at M */,
                                                        int64_t* __restrict__ nl_id, int32_t* __restrict__ bad) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < nr) {
        const int64_t end = (int64_t)in[i], beg = i ? (int64_t)in[i - 1] : 0;
        if (end < beg || end > nz) atomicOr(bad, 1);
        rowptr[i] = nnz0 + end;
        lo[i] = in[nr + 2 * nz + i];
        hi[i] = in[2 * nr + 2 * nz + i];
        nl_id[i] = (int64_t)in[3 * nr + 2 * nz + i];
    }
    if (i < nz) {
        const int64_t c = (int64_t)in[nr + i];
        if (c < 0 || c >= n_lp) atomicOr(bad, 2);
        col[i] = (int32_t)c;
        val[i] = in[nr + nz + i];
    }
}

// recession-LP box of the epigraph variable: 1 + max_r sum_{j != aux} |a_rj| / |a_r,aux|
static __global__ __launch_bounds__(kBlock) void k_aux_box(int64_t m, const int64_t* __restrict__ rowptr,
                                                    const int32_t* __restrict__ col, const double* __restrict__ val,
                                                    int32_t aux, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= m) return;
    double tot = 0.0, av = 0.0;
    for (int64_t e = rowptr[i]; e < rowptr[i + 1]; ++e) {
        const double a = fabs(val[e]);
        tot += a;
        if (col[e] == aux) av += a;
    }
    if (av > 0.0) atomic_max_nonneg(out, (tot - av) / av);
}

// ---------------------------------------------------------- objective certificate ----
// At a point that meets the reference's stop rule (every NL row within f_tol, src/model.jl:257,273) the LP objective is a
// lower bound of the optimum f* of the NLP, and convexity bounds it from the other side by the multiplier-weighted
// violation of the nonlinear rows,
//     f* - obj  <=  sum_i lambda_i res_i  +  LP duality gap            (Lagrangian: f(x) >= f* - sum_i lambda_i g_i(x) for every x),
// with the LP's own duals as the multipliers: lambda_i = sum of |y| over the cuts of NL row i, res_i its SIGNED residual
// g_i - ub_i (lb_i - g_i for a >= row) at the LP point -- rows that end a hair inside cancel rows that end a hair outside, as
// they do in the objective.  (The one-sided sum, positive parts only, is useless as a trigger: 500 active rows at 5e-7 each
// add up to 2.5e-4 while the objective is right to 1e-5; the same holds for the linear rows, which are the LP's own and whose
// residuals sit inside its duality gap.)  Rows further inside than 10 f_tol are left out: a multiplier there is a
// complementarity error of the LP solve, which its gap accounts for.
// The engine evaluates this sum and keeps cutting below f_tol while it exceeds the objective tolerance the reference's tests
// ask for (test/runtests.jl:16-17) -- the refinement that used to be tied to problems of at most 32 columns.
static __global__ __launch_bounds__(kBlock) void k_cert_nl(int64_t m_nl, const int32_t* __restrict__ nl_rows, const int64_t* __restrict__ last_cut,
                                                    const int64_t* __restrict__ cut_prev, const double* __restrict__ y,
                                                    const double* __restrict__ g, const double* __restrict__ lb,
                                                    const double* __restrict__ ub, double f_tol, double* __restrict__ out) {
    const int64_t s = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (s >= m_nl) return;
    double lam = 0.0;
    for (int64_t r = last_cut[s]; r >= 0; r = cut_prev[r]) lam += fabs(y[r]);
    const int32_t i = nl_rows[s];
    double v = -__builtin_inf();                                   // the tighter side's signed residual
    if (isfinite(ub[i])) v = fmax(v, g[i] - ub[i]);
    if (isfinite(lb[i])) v = fmax(v, lb[i] - g[i]);
    out[s] = (v >= -10.0 * f_tol) ? lam * v : 0.0;
}
// Per-BLOCK certificate of a fused batch (ktn_set_blocks: independent instances side by side, one objective tolerance EACH):
// workgroup b adds up the shares of the NL slots whose row lives in block b's columns (fixed order: deterministic) and the
// block's own objective c_b'x_b, and writes  out[b] = D_b / target_b  (target_b = tol * max(1, |obj_b|))  and
// out[nb + b] = target_b / (1 + 2 |obj_b|)  (the relative LP gap a quarter of which the refinement solves ask for).
static __global__ __launch_bounds__(kBlock) void k_cert_blocks(int64_t m_nl, const int32_t* __restrict__ nl_rows, const int64_t* __restrict__ rowptr,
                                                        const int32_t* __restrict__ col, const double* __restrict__ cert,
                                                        const int64_t* __restrict__ blk_col, int64_t nb, const double* __restrict__ c,
                                                        const double* __restrict__ x, double tol, double* __restrict__ out) {
    __shared__ double sh[kBlock];
    const int64_t b = blockIdx.x;
    const int64_t c0 = blk_col[b], c1 = blk_col[b + 1];
    double d = 0.0, o = 0.0;
    for (int64_t s = threadIdx.x; s < m_nl; s += kBlock) {
        const int32_t r = nl_rows[s];
        const int64_t e = rowptr[r];
        if (rowptr[r + 1] > e) { const int32_t cc = col[e]; if (cc >= c0 && cc < c1) d += cert[s]; }
    }
    for (int64_t j = c0 + threadIdx.x; j < c1; j += kBlock) o += c[j] * x[j];
    sh[threadIdx.x] = d;
    __syncthreads();
    for (int k = kBlock / 2; k > 0; k >>= 1) { if ((int)threadIdx.x < k) sh[threadIdx.x] += sh[threadIdx.x + k]; __syncthreads(); }
    const double D = sh[0];
    __syncthreads();
    sh[threadIdx.x] = o;
    __syncthreads();
    for (int k = kBlock / 2; k > 0; k >>= 1) { if ((int)threadIdx.x < k) sh[threadIdx.x] += sh[threadIdx.x + k]; __syncthreads(); }
    if (threadIdx.x == 0) {
        const double ob = fabs(sh[0]), target = tol * fmax(1.0, ob);
        out[b] = (D == D) ? fmax(D, 0.0) / target : __builtin_inf();
        out[nb + b] = target / (1.0 + 2.0 * ob);
    }
}
// partials[b] = sum over the block's grid-stride share of a_i
static __global__ __launch_bounds__(kBlock) void k_sum_partial(int64_t n, const double* __restrict__ a, double* __restrict__ partials) {
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) acc += a[i];
    __shared__ double sh[kBlock / 64];
    acc = group_sum<64>(acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double v = 0.0;
        for (int k = 0; k < kBlock / 64; ++k) v += sh[k];
        partials[blockIdx.x] = v;
    }
}

// ---------------------------------------------------------- epigraph reference shift ----
// With a nonlinear objective every epigraph cut  grad f(x_k)'x - t {<=,>=} -b_k  (src/nlpeval.jl:49-63, src/model.jl:137-164)
// is dense, and close to the optimum the cuts are nearly parallel: their common part grad f(x*) dominates every row.  A
// first-order LP method then idles (multiplier mass moves between two such rows at a rate proportional to the violation;
// the free variable t couples to them only through a 1/sqrt(n) share of the row norm).  The LP solve therefore works on
// the column-transformed problem  t = s + a_ref'x + b_ref  with (a_ref, b_ref) the NEWEST epigraph cut:
//     rows   (a_k - a_ref)'x - s {<=,>=} -(b_k - b_ref)        cost  c_x + c_t a_ref  (+ constant c_t b_ref)
// an exact change of variables -- same duals, same x, t recovered afterwards -- in which the common part sits in the cost
// vector and the rows hold only the differences.  The stored LP (lp_val, lp_lo, lp_hi, lp_c: what getKatanaCuts exports,
// what purging and the exact small-LP kernel read) stays in the reference's form; the solve reads the working copies.
// An epigraph cut is recognised by its last entry: the epigraph variable has the largest column index (engine.hip, loadproblem).
static __global__ __launch_bounds__(kBlock) void k_epi_newest(int64_t first, int64_t m, const int64_t* __restrict__ rowptr,
                                                       const int32_t* __restrict__ col, int32_t tcol,
                                                       unsigned long long* __restrict__ newest_plus1) {
    const int64_t r = first + (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (r >= m) return;
    const int64_t end = rowptr[r + 1];
    if (end > rowptr[r] && col[end - 1] == tcol) atomicMax(newest_plus1, (unsigned long long)(r + 1));
}
// a_ref (dense, zero-initialised by the caller) and b_ref from the newest epigraph cut; scal[0] = b_ref, scal[1] = a_ref'x (later)
static __global__ __launch_bounds__(kBlock) void k_epi_setref(const unsigned long long* __restrict__ newest_plus1,
                                                       const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                       const double* __restrict__ val, const double* __restrict__ lo,
                                                       const double* __restrict__ hi, int32_t tcol, double* __restrict__ aref,
                                                       double* __restrict__ scal) {
    const int64_t r = (int64_t)newest_plus1[0] - 1;
    if (r < 0) {
        if (blockIdx.x == 0 && threadIdx.x == 0) scal[0] = 0.0;
        return;
    }
    const int64_t beg = rowptr[r], end = rowptr[r + 1];
    for (int64_t e = beg + (int64_t)blockIdx.x * kBlock + threadIdx.x; e < end; e += (int64_t)gridDim.x * kBlock) {
        const int32_t c = col[e];
        if (c != tcol) aref[c] = val[e];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const double h = hi[r], l = lo[r];
        const double b = isfinite(h) ? -h : (isfinite(l) ? -l : 0.0);     // a cut has one finite side (src/model.jl:74-75)
        scal[0] = b;
    }
}
// working copies of the epigraph cuts: kEpiChunks workgroups per LP row from `first` on (all others return at once)
constexpr int kEpiChunks = 16;
static __global__ __launch_bounds__(kBlock) void k_epi_shift(int64_t first, int64_t m, const int64_t* __restrict__ rowptr,
                                                      const int32_t* __restrict__ col, const double* __restrict__ val,
                                                      const double* __restrict__ lo, const double* __restrict__ hi, int32_t tcol,
                                                      const double* __restrict__ aref, const double* __restrict__ scal,
                                                      double* __restrict__ wval, double* __restrict__ wlo, double* __restrict__ whi) {
    const int64_t r = first + (int64_t)blockIdx.x / kEpiChunks;
    const int chunk = (int)(blockIdx.x % kEpiChunks);
    if (r >= m) return;
    const int64_t beg = rowptr[r], end = rowptr[r + 1];
    if (end <= beg || col[end - 1] != tcol) return;
    for (int64_t e = beg + (int64_t)chunk * kBlock + threadIdx.x; e < end; e += (int64_t)kEpiChunks * kBlock) {
        const int32_t c = col[e];
        if (c != tcol) {
            const double a = val[e], b = aref[c], d = a - b;
            wval[e] = (fabs(d) <= 1e-14 * (fabs(a) + fabs(b))) ? 0.0 : d;      // rounding noise of two equal derivatives
        }
    }
    if (chunk == 0 && threadIdx.x == 0) { wlo[r] = lo[r] + scal[0]; whi[r] = hi[r] + scal[0]; }
}
// working cost: c_x + c_t a_ref  (a_ref[tcol] == 0)
static __global__ __launch_bounds__(kBlock) void k_epi_cost(int64_t n, const double* __restrict__ c, int32_t tcol,
                                                     const double* __restrict__ aref, double* __restrict__ wc) {
    const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j < n) wc[j] = c[j] + c[tcol] * aref[j];
}
// t <-> s around a solve (scal[1] = a_ref'x of the unscaled x):
//   dir = -1: xh[tcol] = s / dc[tcol] with s = t - a_ref'x - b_ref, put on the feasible side of the reference cut -- whose
//             working row is simply s >= 0 (Min) / s <= 0 (Max) whatever x is: a NEW reference cut is violated at the
//             previous LP point by f(x) - t (1e6 early on), and started from there PDHG over-shoots by as much and then
//             drifts back at the pace of the primal step.  first = 1 (no earlier solve): s = 0, i.e. t starts AT the
//             reference cut's value -- from t = 0 it would have to travel |a_ref'x + b_ref| the same way.
//   dir = +1: x[tcol]  += a_ref'x + b_ref                (x unscaled, after the solve)
static __global__ void k_epi_var(double* __restrict__ x, int32_t tcol, const double* __restrict__ scal, const double* __restrict__ d, int dir,
                          int first, double sgn, const unsigned long long* __restrict__ newest_plus1) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    const double off = scal[1] + scal[0];
    if (dir < 0) {
        double s = x[tcol] - off / d[tcol];
        if (newest_plus1[0] > 0) {                       // (without any epigraph cut there is no reference row: s = t)
            if (first) s = 0.0;
            else s = (sgn > 0.0) ? fmax(s, 0.0) : fmin(s, 0.0);
        }
        x[tcol] = s;
    } else {
        x[tcol] += off;
    }
}

// ---- two small helpers of the LP setup (sum of squares over the finite entries; hashed start vector of the power iteration)
static __global__ __launch_bounds__(kBlock) void k_finite_sq_partial(int64_t n, const double* __restrict__ a,
                                                              double* __restrict__ partials) {
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const double v = a[i];
        if (isfinite(v)) acc += v * v;
    }
    __shared__ double sh[kBlock / 64];
    acc = group_sum<64>(acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double v = 0.0;
        for (int k = 0; k < kBlock / 64; ++k) v += sh[k];
        partials[blockIdx.x] = v;
    }
}

static __global__ __launch_bounds__(kBlock) void k_hash_fill(int64_t n, double* __restrict__ z) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    uint64_t h = (uint64_t)i * 0x9E3779B97F4A7C15ULL + 0xD1B54A32D192ED03ULL;
    h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ULL; h ^= h >> 32;
    z[i] = 0.25 + (double)(h >> 11) * (1.0 / 9007199254740992.0);   // in [0.25, 1.25)
}

}  // namespace ktn
