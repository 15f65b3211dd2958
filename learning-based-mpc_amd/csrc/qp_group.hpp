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
template <int NX, int KB /* compile-time bound on the row length in flight */>
__device__ __forceinline__ void load_row(const CtxDev* __restrict__ C, const LdsView& L, int p, double* cr, int& len) {
    const bool srow = p < C->nS;
    const double* row = srow ? (L.Srow + p * C->RL + 1) : (L.Hm + (p - C->nS) * C->nzp);
    len = srow ? NX : C->nz;
#pragma unroll
    for (int k = 0; k < KB; ++k) {
        const double cv = (k < len) ? row[k] : 0.0;
        cr[k] = srow ? -cv : cv;
    }
}

// Set / clear the scan bit of row p of QP g (p, flag: row-uniform in row g) in the lane that scans that row.
__device__ __forceinline__ void mark_rows(bool flag, int p, int nS, int lane, uint32_t* actS, uint32_t* actH, bool set) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if (__builtin_amdgcn_readlane((int)flag, 16 * g)) {  // wave-uniform branch
            const int pg = __builtin_amdgcn_readlane(p, 16 * g);
            const bool srow = pg < nS;
            const int hh = srow ? pg : pg - nS;
            const uint32_t bit = ((hh & 63) == lane) ? (1u << (hh >> 6)) : 0u;
            if (set) { actS[g] |= srow ? bit : 0u; actH[g] |= srow ? 0u : bit; }
            else { actS[g] &= ~(srow ? bit : 0u); actH[g] &= ~(srow ? 0u : bit); }
        }
    }
}

// Pivot scan for the four QPs of the wave at once.  Lane l evaluates rows l, l+64, ... for ALL four QPs: every row
// is fetched from LDS once per wave (64 distinct rows per ds_read, next rows in flight during the arithmetic), the
// per-QP operands (alpha_g, e_g = q_g + s0_g, z_g) are wave-uniform.  Returns this ROW's (best, bj).
// slack_S[j] = alpha S_b[j] + S_A[j,:] e ; slack_H[h] = (1-h_k)[h % nc] - H[h,:] z ; strict <, ties -> lowest index.
template <int NX>
__device__ __forceinline__ void scan_rows(const CtxDev* __restrict__ C, const LdsView& L, const double* wws,
                                          double alpha, const double* q, double my_z, bool need, const uint32_t* actS,
                                          const uint32_t* actH, double& best_out, int& bj_out, unsigned* hpasses) {
    constexpr int RLc = NX + 1;
    const int lane = lane_id(), grp = lane >> 4;
    const int nS = C->nS, nH = C->nH, nc = C->nc, nzp = C->nzp, nz = C->nz;
    double ea[4], ee[4][NX], best[4];
    int bj[4];
    {
        double e[NX];
        static_for<0, NX>([&](auto K) { constexpr int k = K; e[k] = q[k] + rowb<k>(my_z); });
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            ea[g] = readlane_f64(alpha, 16 * g);
#pragma unroll
            for (int k = 0; k < NX; ++k) ee[g][k] = readlane_f64(e[k], 16 * g);
            best[g] = -QP_TOL_VIOL; bj[g] = 0x7fffffff;
        }
    }
    // ---- S rows (padded to a multiple of 128 with never-violated rows) ----
    {
        const int nch = C->nSp >> 6;  // even
        const double* base = L.Srow + lane * RLc;
        double A[RLc], B[RLc];
        auto eval = [&](const double* cf, int m) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                double acc = ea[g] * cf[0];
#pragma unroll
                for (int k = 0; k < NX; ++k) acc = fma(cf[1 + k], ee[g][k], acc);
                const bool ok = !((actS[g] >> m) & 1u) && acc < best[g];
                best[g] = ok ? acc : best[g];
                bj[g] = ok ? lane + 64 * m : bj[g];
            }
        };
#pragma unroll
        for (int k = 0; k < RLc; ++k) A[k] = base[k];
        for (int m = 0; m < nch; m += 2) {
            const double* pb = base + (m + 1) * 64 * RLc;
#pragma unroll
            for (int k = 0; k < RLc; ++k) B[k] = pb[k];
            __builtin_amdgcn_sched_barrier(0);
            eval(A, m);
            const double* pa = base + (m + 2 < nch ? m + 2 : 0) * 64 * RLc;
#pragma unroll
            for (int k = 0; k < RLc; ++k) A[k] = pa[k];
            __builtin_amdgcn_sched_barrier(0);
            eval(B, m + 1);
        }
    }
    // ---- H rows (row index clamped for the loads; rows >= nH are skipped), two QPs per pass ----
    {
        const int nchH = ((nH + 127) >> 7) << 1;  // even
        const int nz4 = (nz + 3) >> 2;
        // Certificate that no H row of this QP can be violated at z: slack_h = (1-h_k)[r] - H[h,:].z >=
        // min_r (1-h_k)[r] - sum_k max_h|H[h,k]| |z_k| >= 1e-9 (far above the tolerance and the rounding of the exact
        // evaluation).  A pass over the H rows is skipped when it holds for both of its QPs (rows that do not need a
        // scan count as certified).
        bool hskip;
        {
            const int i = lane & 15;
            double t = (i < nz) ? C->cmH[i] * fabs(my_z) : 0.0;
            t += dpp_f64<0xB1>(t); t += dpp_f64<0x4E>(t); t += dpp_f64<0x141>(t); t += dpp_f64<0x140>(t);
            const double omin = row_min((i < nc) ? wws[grp * GWS_DOUBLES + i] : LB_INF);
            hskip = !need || (t <= omin - 1e-9);
        }
#pragma unroll
        for (int g0 = 0; g0 < 4; g0 += 2) {
            if (__builtin_amdgcn_readlane((int)hskip, 16 * g0) && __builtin_amdgcn_readlane((int)hskip, 16 * (g0 + 1))) continue;
            *hpasses += 1;
            double zu[2][NZMAX];
#pragma unroll
            for (int gg = 0; gg < 2; ++gg)
#pragma unroll
                for (int k = 0; k < NZMAX; ++k) zu[gg][k] = readlane_f64(my_z, 16 * (g0 + gg) + k);  // lanes >= nz hold 0
            int r = lane % nc;
            const int rstep = 64 % nc;
            const double* hb = L.Hm;
            double rowA[NZMAX], rowB[NZMAX];
            auto fetch = [&](double* dst, int m) {
                const int hr = lane + 64 * m;
                const double* hp = hb + (hr < nH ? hr : nH - 1) * nzp;
#pragma unroll
                for (int k4 = 0; k4 < NZMAX / 4; ++k4)
                    if (k4 < nz4) {
#pragma unroll
                        for (int k = 4 * k4; k < 4 * k4 + 4; ++k) dst[k] = hp[k];
                    }
            };
            auto eval = [&](const double* row, int m, double o0, double o1) {
                const int h = lane + 64 * m;
                double acc0 = o0, acc1 = o1;
#pragma unroll
                for (int k4 = 0; k4 < NZMAX / 4; ++k4)
                    if (k4 < nz4) {
#pragma unroll
                        for (int k = 4 * k4; k < 4 * k4 + 4; ++k) {
                            acc0 = fma(-row[k], zu[0][k], acc0);
                            acc1 = fma(-row[k], zu[1][k], acc1);
                        }
                    }
                const bool in = h < nH;
                const bool ok0 = in && !((actH[g0] >> m) & 1u) && acc0 < best[g0];
                const bool ok1 = in && !((actH[g0 + 1] >> m) & 1u) && acc1 < best[g0 + 1];
                best[g0] = ok0 ? acc0 : best[g0];         bj[g0] = ok0 ? nS + h : bj[g0];
                best[g0 + 1] = ok1 ? acc1 : best[g0 + 1]; bj[g0 + 1] = ok1 ? nS + h : bj[g0 + 1];
            };
            auto omh = [&](int gg, int& rr) {  // (1 - h_k)[h % nc] of QP g0+gg for the row of this m-step
                return wws[(g0 + gg) * GWS_DOUBLES + rr];
            };
            fetch(rowA, 0);
            for (int m = 0; m < nchH; m += 2) {  // nchH is even
                fetch(rowB, m + 1);
                const double a0 = omh(0, r), a1 = omh(1, r);
                r += rstep; r = r >= nc ? r - nc : r;
                const double b0 = omh(0, r), b1 = omh(1, r);
                r += rstep; r = r >= nc ? r - nc : r;
                __builtin_amdgcn_sched_barrier(0);
                eval(rowA, m, a0, a1);
                fetch(rowA, m + 2 < nchH ? m + 2 : 0);
                __builtin_amdgcn_sched_barrier(0);
                eval(rowB, m + 1, b0, b1);
            }
        }
    }
    best_out = 0.0; bj_out = 0x7fffffff;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        wave_argmin(best[g], bj[g]);
        best_out = grp == g ? best[g] : best_out;
        bj_out = grp == g ? bj[g] : bj_out;
    }
}

// Most violated candidate row of this QP (row-local): lane i evaluates cand0 and cand1 with the arithmetic of the full
// scan (so the value equals the oracle's row_slack bit for bit), rows already active are skipped, (value, row) arg-min
// with ties to the lowest row.
template <int NX>
__device__ __forceinline__ void candidate_pivot(const CtxDev* __restrict__ C, const LdsView& L, const double* gws, double alpha,
                                                const double* q, double my_z, int my_act, int nq, bool need, int cand0,
                                                int cand1, double& best, int& bj) {
    const int nS = C->nS, nc = C->nc, nz = C->nz, nzp = C->nzp;
    double e[NX];
    static_for<0, NX>([&](auto K) { constexpr int k = K; e[k] = q[k] + rowb<k>(my_z); });
    bool act0 = false, act1 = false;
    static_for<0, NZMAX>([&](auto A) {
        constexpr int a = A;
        const int ra = rowb_i<a>(my_act);
        act0 = act0 || (a < nq && ra == cand0);
        act1 = act1 || (a < nq && ra == cand1);
    });
    const bool v0 = need && cand0 >= 0 && !act0, v1 = need && cand1 >= 0 && !act1;
    double s0v = LB_INF, s1v = LB_INF;
    {   // S-type candidate (cand1 is never an S row)
        const int j = (v0 && cand0 < nS) ? cand0 : 0;
        const double* row = L.Srow + j * (NX + 1);
        double acc = alpha * row[0];
#pragma unroll
        for (int k = 0; k < NX; ++k) acc = fma(row[1 + k], e[k], acc);
        if (v0 && cand0 < nS) s0v = acc;
    }
    if (__builtin_amdgcn_ballot_w64((v0 && cand0 >= nS) || v1) != 0) {  // H-type candidates
        double zr[NZMAX];
        static_for<0, NZMAX>([&](auto K) { constexpr int k = K; zr[k] = rowb<k>(my_z); });
        const int h0 = (v0 && cand0 >= nS) ? cand0 - nS : 0, h1 = v1 ? cand1 - nS : 0;
        const double* r0 = L.Hm + h0 * nzp;
        const double* r1 = L.Hm + h1 * nzp;
        double a0 = gws[h0 % nc], a1 = gws[h1 % nc];
#pragma unroll
        for (int k = 0; k < NZMAX; ++k)
            if (k < nz) { a0 = fma(-r0[k], zr[k], a0); a1 = fma(-r1[k], zr[k], a1); }
        if (v0 && cand0 >= nS) s0v = a0;
        if (v1) s1v = a1;
    }
    if (v0 && s0v < best) { best = s0v; bj = cand0; }
    if (v1 && (s1v < best || (s1v == best && cand1 < bj))) { best = s1v; bj = cand1; }
    row_argmin(best, bj);
    if (!(best < -QP_TOL_VIOL)) bj = 0x7fffffff;
}

// Append row `a` (row-uniform) to the factor: lane b <= a forms M[a][b] = c_b . g_a (c_b = its own active row, g_a
// broadcast from the lanes holding it), the row solve is a column sweep, lane a gets the diagonal.
template <int NX, int KB>
__device__ __forceinline__ void chol_append_row(const CtxDev* __restrict__ C, const LdsView& L, bool pred, int a,
                                                int my_act, double ga /* lane k: g_a[k] */, int i, double* LL,
                                                double& my_invd) {
    double cr[KB];
    int len;
    load_row<NX, KB>(C, L, (pred && i <= a) ? my_act : 0, cr, len);
    double acc = 0.0;
    static_for<0, KB>([&](auto K) {
        constexpr int k = K;
        const double gk = rowb<k>(ga);
        acc = (k < len) ? fma(cr[k], gk, acc) : acc;
    });
    acc = (pred && i <= a) ? acc : 0.0;
    int amax;  // largest row index being appended among the four rows (wave-uniform)
    {
        const int ap = pred ? a : 0;
        const int n0 = __builtin_amdgcn_readlane(ap, 0), n1 = __builtin_amdgcn_readlane(ap, 16);
        const int n2 = __builtin_amdgcn_readlane(ap, 32), n3 = __builtin_amdgcn_readlane(ap, 48);
        const int n01 = n0 > n1 ? n0 : n1, n23 = n2 > n3 ? n2 : n3;
        amax = n01 > n23 ? n01 : n23;
    }
    static_for<0, NZMAX>([&](auto K) {
        constexpr int k = K;
        if (k < amax) {
            const double yk = rowb<k>(acc) * rowb<k>(my_invd);
            if (k < a) {
                if (i == k) acc = yk;
                else if (i > k && i < a) acc = fma(-yk, LL[k], acc);
                else if (i == a) acc = fma(-yk, yk, acc);
            }
        }
    });
    if (pred && i == a) {
        acc = sqrt(acc > 0.0 ? acc : 0.0);
        my_invd = 1.0 / acc;
    }
    // store: lane b < a keeps L[a][b] as its column entry a; lane a gathers its row
    static_for<0, NZMAX>([&](auto K) {
        constexpr int k = K;
        if (k <= amax) {
            const double lab = rowb<k>(acc);  // L[a][k] for k < a
            if (pred) {
                if (i < a && k == a) LL[k] = acc;
                if (i == a && k < a) LL[k] = lab;
            }
        }
    });
}

// Solve the QPs of the four rows of this wave (live = this row has a problem).  alpha, q: row-uniform.
// gws: this ROW's LDS slot (omh filled by the caller; results written at the end).
template <int NX>
__device__ __forceinline__ void group_qp_solve(const CtxDev* __restrict__ C, const LdsView& L, double* gws, bool live,
                                               double alpha, const double* q, unsigned* ctr /* [0] trips, [1] full scans, [2] H passes (wave-uniform) */,
                                               int* hint_rows /* this row's 16 candidate slots in HBM, or null */,
                                               int* hint_n /* their count (in: previous step, out: this solve) */,
                                               long long* prof = nullptr) {
#ifdef LBMPC_PROFILE_FULL
#define GP_T(var) const long long var = clock64()
#define GP_ADD(slot, t0, t1) prof[slot] += (t1) - (t0)
#define GP_INC(slot) prof[slot] += 1
#else
#define GP_T(var)
#define GP_ADD(slot, t0, t1)
#define GP_INC(slot)
#endif
    const int lane = lane_id(), i = lane & 15, grp = lane >> 4;
    const int nz = C->nz, nS = C->nS;
    bool done = !live, have_p = false;
    int nq = 0, iters = 0, status = 0, p = 0;
    double sp = 0.0, cg = 0.0;
    // scan bookkeeping is per LANE (not per row): bit m of actS[g] / actH[g] = row lane + 64 m is active in QP g
    uint32_t actS[4] = {0, 0, 0, 0}, actH[4] = {0, 0, 0, 0};
    double my_u = 0.0, my_z = 0.0, my_g = 0.0, my_invd = 1.0;
    int my_act = 0;
    double LL[NZMAX], WmK[NZMAX];
#pragma unroll
    for (int k = 0; k < NZMAX; ++k) { LL[k] = 0.0; WmK[k] = 0.0; }
    // S rows only touch s0: while no H row is active or pivot in any of the four QPs, every row vector, g = Ginv c
    // (Ginv is block diagonal) and column of Wm is zero beyond nx, and the k-loops stop at kmax = nx (exact zeros
    // are skipped, nothing else changes)
    int hact = 0;  // active H rows of this QP (row-uniform)
    int kmax = NX;
    // Pivot candidates (oracle qp_solve): the final active set of the trial's previous closed-loop step; lane i holds
    // candidate i and, for an H row, its one-step-earlier twin.
    int cand0 = -1, cand1 = -1;
    if (live && hint_rows) {
        const int nh = *hint_n;
        cand0 = i < nh ? hint_rows[i] : -1;
        cand1 = cand0 >= nS + C->nc ? cand0 - C->nc : -1;
    }

    while (__builtin_amdgcn_ballot_w64(!done) != 0) {
        // ---- Step 1: most violated row at the current z (rows without a pivot) ----
        const bool need_scan = !done && !have_p;
        GP_T(g0);
        GP_INC(4);
        ctr[0] += 1;
        if (__builtin_amdgcn_ballot_w64(need_scan) != 0) {
            double best = -QP_TOL_VIOL;
            int bj = 0x7fffffff;
            // candidates first: while one of them is violated no pass over all the rows is needed
            bool need_full = need_scan;
            if (__builtin_amdgcn_ballot_w64(need_scan && cand0 >= 0) != 0) {
                double cb = -QP_TOL_VIOL;
                int cj = 0x7fffffff;
                candidate_pivot<NX>(C, L, gws, alpha, q, my_z, my_act, nq, need_scan, cand0, cand1, cb, cj);
                if (need_scan && cj != 0x7fffffff) { best = cb; bj = cj; need_full = false; }
            }
            if (__builtin_amdgcn_ballot_w64(need_full) != 0) {
                GP_INC(5);
                ctr[1] += 1;
                double fb = -QP_TOL_VIOL;
                int fj = 0x7fffffff;
                scan_rows<NX>(C, L, gws - grp * GWS_DOUBLES, alpha, q, my_z, need_full, actS, actH, fb, fj, ctr + 2);
                if (need_full) { best = fb; bj = fj; }
            }
            const bool optimal = need_scan && bj == 0x7fffffff;
            const bool newp = need_scan && !optimal;
            done = done || optimal;
            // c_p, g = Ginv c_p (lane k holds g_k), cg = c_p . g
            if (newp) { p = bj; sp = best; have_p = true; }
            kmax = (__builtin_amdgcn_ballot_w64(!done && (hact > 0 || (have_p && p >= nS))) != 0) ? nz : NX;
            double g = 0.0, cgn = 0.0;
            auto pivot_g = [&](auto KBc) {
                constexpr int KB = KBc;
                double cp[KB], gi[KB];
                int len;
                load_row<NX, KB>(C, L, newp ? bj : 0, cp, len);
                const int gl = i < nz ? i : 0;
#pragma unroll
                for (int k = 0; k < KB; ++k) gi[k] = (k < nz) ? L.Ginv[gl * nz + k] : 0.0;
#pragma unroll
                for (int k = 0; k < KB; ++k) g = (k < len) ? fma(gi[k], cp[k], g) : g;
                g = i < nz ? g : 0.0;
                static_for<0, KB>([&](auto K) {
                    constexpr int k = K;
                    const double gk = rowb<k>(g);
                    cgn = (k < len) ? fma(cp[k], gk, cgn) : cgn;
                });
            };
            if (kmax <= NX) pivot_g(std::integral_constant<int, NX>{}); else pivot_g(std::integral_constant<int, NZMAX>{});
            if (newp) {
                cg = cgn; my_g = g;
                if (i == nq) my_u = 0.0;
            }
        }
        // ---- Step 2 for every row that is not done ----
        GP_T(g1);
        GP_ADD(0, g0, g1);
        const bool act2 = !done;
        if (act2) {
            iters += 1;
            if (iters > QP_MAX_ITERS) { status = LBMPC_ST_ITER_CAP; done = true; }
        }
        // rhs_a = c_a . g (lane a), forward / backward column sweeps with the stored reciprocals
        double my_rhs = 0.0;
        {
            auto rhs_dot = [&](auto KBc) {
                constexpr int KB = KBc;
                double cr[KB];
                int len;
                load_row<NX, KB>(C, L, (act2 && i < nq) ? my_act : 0, cr, len);
                static_for<0, KB>([&](auto K) {
                    constexpr int k = K;
                    const double gk = rowb<k>(my_g);
                    my_rhs = (k < len) ? fma(cr[k], gk, my_rhs) : my_rhs;
                });
            };
            if (kmax <= NX) rhs_dot(std::integral_constant<int, NX>{}); else rhs_dot(std::integral_constant<int, NZMAX>{});
            my_rhs = (act2 && i < nq) ? my_rhs : 0.0;
        }
        // largest active-set size among the four rows (wave-uniform): the sweeps stop there
        int nqmax;
        {
            const int n0 = __builtin_amdgcn_readlane(nq, 0), n1 = __builtin_amdgcn_readlane(nq, 16);
            const int n2 = __builtin_amdgcn_readlane(nq, 32), n3 = __builtin_amdgcn_readlane(nq, 48);
            const int n01 = n0 > n1 ? n0 : n1, n23 = n2 > n3 ? n2 : n3;
            nqmax = n01 > n23 ? n01 : n23;
        }
        double acc = my_rhs;
        static_for<0, NZMAX>([&](auto K) {  // L y = rhs
            constexpr int k = K;
            if (k < nqmax) {
                const double yk = rowb<k>(acc) * rowb<k>(my_invd);
                if (k < nq) {
                    if (i == k) acc = yk;
                    else if (i > k && i < nq) acc = fma(-LL[k], yk, acc);
                }
            }
        });
        static_for_down<NZMAX - 1>([&](auto K) {  // L' r = y
            constexpr int k = K;
            if (k < nqmax) {
                const double rk = rowb<k>(acc) * rowb<k>(my_invd);
                if (k < nq) {
                    if (i == k) acc = rk;
                    else if (i < k) acc = fma(-LL[k], rk, acc);
                }
            }
        });
        const double my_r = (i < nq) ? acc : 0.0;
        double delta = cg, zacc = my_g;
        static_for<0, NZMAX>([&](auto K) {
            constexpr int a = K;
            if (a < nqmax) {
                const double ra = rowb<a>(my_r), rh = rowb<a>(my_rhs);
                if (a < nq) {
                    delta = fma(-rh, ra, delta);
                    zacc = fma(-WmK[a], ra, zacc);
                }
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
        GP_T(g2);
        GP_ADD(1, g1, g2);
        if (__builtin_amdgcn_ballot_w64(full) != 0) {
            // full step: constraint p joins the active set as number nq
            if (full && i == nq) my_act = p;
#pragma unroll
            for (int a = 0; a < NZMAX; ++a) WmK[a] = (full && a == nq) ? my_g : WmK[a];
            if (kmax <= NX) chol_append_row<NX, NX>(C, L, full, nq, my_act, my_g, i, LL, my_invd);
            else chol_append_row<NX, NZMAX>(C, L, full, nq, my_act, my_g, i, LL, my_invd);
            mark_rows(full, p, nS, lane, actS, actH, true);
            if (full) {
                nq += 1;
                have_p = false;
                hact += p >= nS ? 1 : 0;
            }
        }
        GP_T(g3);
        GP_ADD(2, g2, g3);
        if (__builtin_amdgcn_ballot_w64(drop) != 0) {
            GP_INC(6);
            // partial / dual-only step: drop active constraint kdrop, stay in step 2 with the same pivot
            const int pd = row_max_i32((drop && i == kdrop) ? my_act : -1);
            mark_rows(drop, pd, nS, lane, actS, actH, false);
            if (drop) hact -= pd >= nS ? 1 : 0;
            const int nxt_act = dpp_i32<0x101>(my_act);   // row_shl:1 : lane i receives lane i+1
            const double nxt_u = dpp_f64<0x101>(my_u);
            if (drop && i >= kdrop && i < nq) { my_act = nxt_act; my_u = nxt_u; }
#pragma unroll
            for (int a = 0; a < NZMAX - 1; ++a) WmK[a] = (drop && a >= kdrop && a < nq - 1) ? WmK[a + 1] : WmK[a];
            if (drop) nq -= 1;
            // rebuild the factor rows kdrop .. nq-1; g_a = Ginv c_a of row a is the stored column a of Wm (lane k: WmK[a])
            static_for<0, NZMAX>([&](auto A) {
                constexpr int a = A;
                const bool pr = drop && a >= kdrop && a < nq;
                if (__builtin_amdgcn_ballot_w64(pr) != 0) {
                    if (kmax <= NX) chol_append_row<NX, NX>(C, L, pr, a, my_act, WmK[a], i, LL, my_invd);
                    else chol_append_row<NX, NZMAX>(C, L, pr, a, my_act, WmK[a], i, LL, my_invd);
                }
            });
        }
        GP_T(g4);
        GP_ADD(3, g3, g4);
    }
    if (live && hint_rows) {  // this solve's active set = the next step's candidates
        if (i < nq) hint_rows[i] = my_act;
        if (i == 0) *hint_n = nq;
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
