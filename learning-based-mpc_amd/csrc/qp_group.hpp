// qp_group.hpp -- Goldfarb-Idnani dual active-set solver, one 16-lane DPP row per QP (4 QPs per wave).
//
//   min z'Pz  s.t.  alpha S_b + S_A (q + s0) >= 0 (nS rows),  (1-h) - H z >= 0 (nH rows)      UQRobustMPC.m:149-155
//
// Mapping inside a row (i = lane & 15):  lane i strides the constraint rows i, i+16, ... in the pivot scan;
// lane a < nq carries active constraint a (row id, multiplier u_a, r_a, 1/L[a][a], and row/column a of the Cholesky
// factor of M = C_A Ginv C_A' in LL[]); lane k < nz carries z_k, g_k and column k of Wm = Ginv C_A' in WmK[].
// Every cross-lane move is a row-local DPP op (row_newbcast / quad_perm / mirrors / row_shl): no LDS round trips in
// the algebra, and the four rows of a wave run four independent QPs through one instruction stream; all per-QP
// scalars (nq, p, sp, t, ...) are row-uniform VGPR values and control flow is predicated per row.
// Arithmetic, operation order and tie rules are those of the oracle's qp_solve (oracle/lbmpc_oracle.c).
#pragma once
#include <type_traits>

namespace lbmpc {

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
template <int I, class F>
__device__ __forceinline__ void static_for_down(F&& f) {
    if constexpr (I >= 0) {
        f(std::integral_constant<int, I>{});
        static_for_down<I - 1>(f);
    }
}

template <int K> __device__ __forceinline__ double rowb(double v) { return dpp_f64<0x150 + K>(v); }  // row_newbcast:K
template <int K> __device__ __forceinline__ int rowb_i(int v) { return dpp_i32<0x150 + K>(v); }

__device__ __forceinline__ double row_min(double v) {
    v = __builtin_fmin(v, dpp_f64<0xB1>(v));
    v = __builtin_fmin(v, dpp_f64<0x4E>(v));
    v = __builtin_fmin(v, dpp_f64<0x141>(v));
    return __builtin_fmin(v, dpp_f64<0x140>(v));
}
__device__ __forceinline__ double row_max(double v) {
    v = __builtin_fmax(v, dpp_f64<0xB1>(v));
    v = __builtin_fmax(v, dpp_f64<0x4E>(v));
    v = __builtin_fmax(v, dpp_f64<0x141>(v));
    return __builtin_fmax(v, dpp_f64<0x140>(v));
}
__device__ __forceinline__ int row_min_i32(int v) {
    int o;
    o = dpp_i32<0xB1>(v); v = o < v ? o : v;
    o = dpp_i32<0x4E>(v); v = o < v ? o : v;
    o = dpp_i32<0x141>(v); v = o < v ? o : v;
    o = dpp_i32<0x140>(v); v = o < v ? o : v;
    return v;
}
__device__ __forceinline__ int row_max_i32(int v) {
    int o;
    o = dpp_i32<0xB1>(v); v = o > v ? o : v;
    o = dpp_i32<0x4E>(v); v = o > v ? o : v;
    o = dpp_i32<0x141>(v); v = o > v ? o : v;
    o = dpp_i32<0x140>(v); v = o > v ? o : v;
    return v;
}
// (value, index) arg-min inside the row, ties to the lowest index
__device__ __forceinline__ void row_argmin(double& v, int& idx) {
    const double best = row_min(v);
    idx = row_min_i32(v == best ? idx : 0x7fffffff);
    v = best;
}

// per-group LDS slot (doubles): [0,16) 1-h_k ; [16,32) z out ; [32,40) act out (16 ints) ; [40] status, iters ; [41] nact
#define GWS_DOUBLES 48
#define GWS_PER_WAVE (4 * GWS_DOUBLES)

// coefficients of QP row p (c_p of c_p.z <= d_p) for a per-lane row id: S rows are [-S_A[j,:], 0], H rows F_bar Psi^i.
// cr[k] = 0 beyond len.  All loads are issued together.
template <int NX>
__device__ __forceinline__ void load_row(const CtxDev* __restrict__ C, const LdsView& L, int p, double* cr, int& len) {
    const bool srow = p < C->nS;
    const double* row = srow ? (L.Srow + p * C->RL + 1) : (L.Hm + (p - C->nS) * C->nzp);
    len = srow ? NX : C->nz;
#pragma unroll
    for (int k = 0; k < NZMAX; ++k) {
        const double cv = (k < len) ? row[k] : 0.0;
        cr[k] = srow ? -cv : cv;
    }
}

// Append row `a` (row-uniform) to the factor: lane b <= a forms M[a][b] = c_b . g_a (c_b = its own active row, g_a
// broadcast from the lanes holding it), the row solve is a column sweep, lane a gets the diagonal.
template <int NX>
__device__ __forceinline__ void chol_append_row(const CtxDev* __restrict__ C, const LdsView& L, bool pred, int a,
                                                int my_act, double ga /* lane k: g_a[k] */, int i, double* LL,
                                                double& my_invd) {
    double cr[NZMAX];
    int len;
    load_row<NX>(C, L, (pred && i <= a) ? my_act : 0, cr, len);
    double acc = 0.0;
    static_for<0, NZMAX>([&](auto K) {
        constexpr int k = K;
        const double gk = rowb<k>(ga);
        acc = (k < len) ? fma(cr[k], gk, acc) : acc;
    });
    acc = (pred && i <= a) ? acc : 0.0;
    static_for<0, NZMAX>([&](auto K) {
        constexpr int k = K;
        const double yk = rowb<k>(acc) * rowb<k>(my_invd);
        if (k < a) {
            if (i == k) acc = yk;
            else if (i > k && i < a) acc = fma(-yk, LL[k], acc);
            else if (i == a) acc = fma(-yk, yk, acc);
        }
    });
    if (pred && i == a) {
        acc = sqrt(acc > 0.0 ? acc : 0.0);
        my_invd = 1.0 / acc;
    }
    // store: lane b < a keeps L[a][b] as its column entry a; lane a gathers its row
    static_for<0, NZMAX>([&](auto K) {
        constexpr int k = K;
        const double lab = rowb<k>(acc);  // L[a][k] for k < a
        if (pred) {
            if (i < a && k == a) LL[k] = acc;
            if (i == a && k < a) LL[k] = lab;
        }
    });
}

// Solve the QPs of the four rows of this wave (live = this row has a problem).  alpha, q: row-uniform.
// gws: this ROW's LDS slot (omh filled by the caller; results written at the end).
template <int NX>
__device__ __forceinline__ void group_qp_solve(const CtxDev* __restrict__ C, const LdsView& L, double* gws, bool live,
                                               double alpha, const double* q) {
    const int i = lane_id() & 15;
    const int nz = C->nz, nS = C->nS, nH = C->nH, nc = C->nc, RL = C->RL, nzp = C->nzp;
    const int mS = (nS + 15) >> 4, mH = (nH + 15) >> 4;
    bool done = !live, have_p = false;
    int nq = 0, iters = 0, status = 0, p = 0;
    double sp = 0.0, cg = 0.0;
    uint64_t actS0 = 0, actS1 = 0, actH0 = 0, actH1 = 0;  // bit m: my m-th S / H row (i + 16 m) is active
    double my_u = 0.0, my_z = 0.0, my_g = 0.0, my_invd = 1.0;
    int my_act = 0;
    double LL[NZMAX], WmK[NZMAX];
#pragma unroll
    for (int k = 0; k < NZMAX; ++k) { LL[k] = 0.0; WmK[k] = 0.0; }

    while (__builtin_amdgcn_ballot_w64(!done) != 0) {
        // ---- Step 1: most violated row at the current z (rows without a pivot) ----
        const bool need_scan = !done && !have_p;
        if (__builtin_amdgcn_ballot_w64(need_scan) != 0) {
            double zr[NZMAX], e[NX];
            static_for<0, NZMAX>([&](auto K) { constexpr int k = K; zr[k] = rowb<k>(my_z); });
#pragma unroll
            for (int k = 0; k < NX; ++k) e[k] = q[k] + zr[k];
            double best = -QP_TOL_VIOL;
            int bj = 0x7fffffff;
#pragma unroll 4
            for (int m = 0; m < mS; ++m) {
                const int j = i + 16 * m;
                const double* row = L.Srow + (j < nS ? j : 0) * RL;
                double acc = alpha * row[0];
#pragma unroll
                for (int k = 0; k < NX; ++k) acc = fma(row[1 + k], e[k], acc);
                const uint64_t mk = m < 64 ? actS0 : actS1;
                if (j < nS && !((mk >> (m & 63)) & 1) && acc < best) { best = acc; bj = j; }
            }
            int r = i % nc;
            const int rstep = 16 % nc;
#pragma unroll 2
            for (int m = 0; m < mH; ++m) {
                const int h = i + 16 * m;
                const double* row = L.Hm + (h < nH ? h : 0) * nzp;
                double acc = gws[r];
#pragma unroll
                for (int k = 0; k < NZMAX; ++k)
                    if (k < nz) acc = fma(-row[k], zr[k], acc);
                const uint64_t mk = m < 64 ? actH0 : actH1;
                if (h < nH && !((mk >> (m & 63)) & 1) && acc < best) { best = acc; bj = nS + h; }
                r += rstep;
                if (r >= nc) r -= nc;
            }
            row_argmin(best, bj);
            const bool optimal = need_scan && bj == 0x7fffffff;
            const bool newp = need_scan && !optimal;
            done = done || optimal;
            // c_p, g = Ginv c_p (lane k holds g_k), cg = c_p . g
            double cp[NZMAX], gi[NZMAX];
            int len;
            load_row<NX>(C, L, newp ? bj : 0, cp, len);
            const int gl = i < nz ? i : 0;
#pragma unroll
            for (int k = 0; k < NZMAX; ++k) gi[k] = (k < nz) ? L.Ginv[gl * nz + k] : 0.0;
            double g = 0.0;
#pragma unroll
            for (int k = 0; k < NZMAX; ++k) g = (k < len) ? fma(gi[k], cp[k], g) : g;
            g = i < nz ? g : 0.0;
            double cgn = 0.0;
            static_for<0, NZMAX>([&](auto K) {
                constexpr int k = K;
                const double gk = rowb<k>(g);
                cgn = (k < len) ? fma(cp[k], gk, cgn) : cgn;
            });
            if (newp) {
                p = bj; sp = best; cg = cgn; my_g = g; have_p = true;
                if (i == nq) my_u = 0.0;
            }
        }
        // ---- Step 2 for every row that is not done ----
        const bool act2 = !done;
        if (act2) {
            iters += 1;
            if (iters > QP_MAX_ITERS) { status = LBMPC_ST_ITER_CAP; done = true; }
        }
        // rhs_a = c_a . g (lane a), forward / backward column sweeps with the stored reciprocals
        double my_rhs = 0.0;
        {
            double cr[NZMAX];
            int len;
            load_row<NX>(C, L, (act2 && i < nq) ? my_act : 0, cr, len);
            static_for<0, NZMAX>([&](auto K) {
                constexpr int k = K;
                const double gk = rowb<k>(my_g);
                my_rhs = (k < len) ? fma(cr[k], gk, my_rhs) : my_rhs;
            });
            my_rhs = (act2 && i < nq) ? my_rhs : 0.0;
        }
        double acc = my_rhs;
        static_for<0, NZMAX>([&](auto K) {  // L y = rhs
            constexpr int k = K;
            const double yk = rowb<k>(acc) * rowb<k>(my_invd);
            if (k < nq) {
                if (i == k) acc = yk;
                else if (i > k && i < nq) acc = fma(-LL[k], yk, acc);
            }
        });
        static_for_down<NZMAX - 1>([&](auto K) {  // L' r = y
            constexpr int k = K;
            const double rk = rowb<k>(acc) * rowb<k>(my_invd);
            if (k < nq) {
                if (i == k) acc = rk;
                else if (i < k) acc = fma(-LL[k], rk, acc);
            }
        });
        const double my_r = (i < nq) ? acc : 0.0;
        double delta = cg, zacc = my_g;
        static_for<0, NZMAX>([&](auto K) {
            constexpr int a = K;
            const double ra = rowb<a>(my_r), rh = rowb<a>(my_rhs);
            if (a < nq) {
                delta = fma(-rh, ra, delta);
                zacc = fma(-WmK[a], ra, zacc);
            }
        });
        const double zd = -zacc;  // primal direction, component i: -(g - Wm' r)
        const bool dependent = (nq >= nz) || !(delta > QP_TOL_DEP * cg);
        const double rmax = row_max((i < nq) ? fabs(my_r) : 0.0);
        double t1 = LB_INF;
        int kdrop = 0x7fffffff;
        if (i < nq && my_r > QP_TOL_R * rmax) { t1 = my_u / my_r; kdrop = i; }
        row_argmin(t1, kdrop);
        const double t2 = dependent ? LB_INF : -sp / delta;
        const double t = t1 < t2 ? t1 : t2;
        if (act2 && !done && t == LB_INF) { status = LBMPC_ST_QP_INFEASIBLE; done = true; }
        const bool upd = act2 && !done;
        if (upd) {
            if (i < nq) my_u = fma(-t, my_r, my_u);
            if (i == nq) my_u += t;
            if (!dependent) {
                if (i < nz) my_z = fma(t, zd, my_z);
                sp = fma(t, delta, sp);
            }
        }
        const bool full = upd && !dependent && t2 <= t1;
        const bool drop = upd && !full;
        if (__builtin_amdgcn_ballot_w64(full) != 0) {
            // full step: constraint p joins the active set as number nq
            if (full && i == nq) my_act = p;
#pragma unroll
            for (int a = 0; a < NZMAX; ++a) WmK[a] = (full && a == nq) ? my_g : WmK[a];
            chol_append_row<NX>(C, L, full, nq, my_act, my_g, i, LL, my_invd);
            if (full) {
                const int hh = p < nS ? p : p - nS;
                const int m = hh >> 4;
                if ((hh & 15) == i) {
                    const uint64_t bit = 1ull << (m & 63);
                    if (p < nS) { if (m < 64) actS0 |= bit; else actS1 |= bit; }
                    else { if (m < 64) actH0 |= bit; else actH1 |= bit; }
                }
                nq += 1;
                have_p = false;
            }
        }
        if (__builtin_amdgcn_ballot_w64(drop) != 0) {
            // partial / dual-only step: drop active constraint kdrop, stay in step 2 with the same pivot
            const int pd = row_max_i32((drop && i == kdrop) ? my_act : -1);
            if (drop) {
                const int hh = pd < nS ? pd : pd - nS;
                const int m = hh >> 4;
                if ((hh & 15) == i) {
                    const uint64_t bit = ~(1ull << (m & 63));
                    if (pd < nS) { if (m < 64) actS0 &= bit; else actS1 &= bit; }
                    else { if (m < 64) actH0 &= bit; else actH1 &= bit; }
                }
            }
            const int nxt_act = dpp_i32<0x101>(my_act);   // row_shl:1 : lane i receives lane i+1
            const double nxt_u = dpp_f64<0x101>(my_u);
            if (drop && i >= kdrop && i < nq) { my_act = nxt_act; my_u = nxt_u; }
#pragma unroll
            for (int a = 0; a < NZMAX - 1; ++a) WmK[a] = (drop && a >= kdrop && a < nq - 1) ? WmK[a + 1] : WmK[a];
            if (drop) nq -= 1;
            // rebuild the factor rows kdrop .. nq-1 (their g is recomputed exactly as when they were added)
            static_for<0, NZMAX>([&](auto A) {
                constexpr int a = A;
                const bool pr = drop && a >= kdrop && a < nq;
                if (__builtin_amdgcn_ballot_w64(pr) != 0) {
                    const int pa = rowb_i<a>(my_act);
                    double cp[NZMAX], gi[NZMAX];
                    int len;
                    load_row<NX>(C, L, pr ? pa : 0, cp, len);
                    const int gl = i < nz ? i : 0;
#pragma unroll
                    for (int k = 0; k < NZMAX; ++k) gi[k] = (k < nz) ? L.Ginv[gl * nz + k] : 0.0;
                    double g = 0.0;
#pragma unroll
                    for (int k = 0; k < NZMAX; ++k) g = (k < len) ? fma(gi[k], cp[k], g) : g;
                    g = i < nz ? g : 0.0;
                    chol_append_row<NX>(C, L, pr, a, my_act, g, i, LL, my_invd);
                }
            });
        }
    }
    // results -> this row's LDS slot
    gws[16 + i] = my_z;
    reinterpret_cast<int*>(gws + 32)[i] = (i < nq) ? my_act : -1;
    if (i == 0) {
        reinterpret_cast<int*>(gws + 40)[0] = status;
        reinterpret_cast<int*>(gws + 40)[1] = iters;
        reinterpret_cast<int*>(gws + 41)[0] = nq;
    }
}

}  // namespace lbmpc
