// qp_lane_gen.hpp -- Goldfarb-Idnani for the tube QP with rows of F_bar Psi^i in play: ONE QP PER LANE, up to 64 QPs
// per wave in lock-step (second tier: qp_lane.hpp handles the S-row regime in registers, the 16-lane-row solver of
// qp_group.hpp is the last resort).
//
//   min z'Pz  s.t.  alpha S_b + S_A (q + s0) >= 0 (nS rows),  (1-h) - H z >= 0 (nH rows)      UQRobustMPC.m:149-155
//
// The constraint matrix is constant over trials and steps, so a lane keeps row IDS only: the coefficients of an
// active row are gathered from the shared LDS copy whenever the algebra needs them (odd row pitch: distinct rows
// of a gather fall on distinct banks), and Wm[a] = Ginv c_a is recomputed from them -- Ginv = (2 blkdiag(Px, Pc))^-1
// must be block diagonal with a diagonal Pc block (checked when the context is created; every reference configuration:
// Pc = (B'PxB + R) I, offline_parameters_computation_nxn.m:313-317), so that takes nx^2 + (nz - nx) products.  The only
// per-lane state that does not fit registers is the Cholesky factor of M = C_A Ginv C_A' (<= LG_NQL rows, packed, the
// diagonal stored as its reciprocal) and the nc values of 1 - h_k: they live in LDS, lane-interleaved
// (slot * lanes + lane: conflict-free ds_read_b64 / ds_write_b64).
// The factor is kept between iterations instead of being rebuilt (oracle qp_solve rebuilds it every trip): a row that
// joins is the forward-substitution vector of the same trip (identical operations: M[nq][b] = c_b . g = rhs[b], and
// fma(-L[nq][k], L[b][k], .) = fma(-L[b][k], y[k], .)), rows below a dropped one keep their leading entries and have
// the others recomputed with the operations of the rebuild.  Arithmetic, operation order and tie rules are the
// oracle's, so results, active sets and iteration counts are bit-identical.  A QP leaves this path (`fallback`,
// re-solved from scratch by group_qp_solve) when its candidate list or its active set would exceed LG_NQL rows, or a
// full scan returns an active row (never seen: an active row's slack is an exact zero up to rounding, far above -tol).
// Full scans are cooperative as in qp_lane.hpp: rows spread over the lanes, four QPs per pass over the S rows, two
// per pass over the H rows (skipped under the certificate of scan_rows).
#pragma once

namespace lbmpc {

#ifndef LG_NQL
#define LG_NQL 8                                   // active rows a lane can hold (C1: <= 8, C3: <= 4 observed)
#endif
#define LG_LP (LG_NQL * (LG_NQL + 1) / 2)          // packed factor
#define LG_SLOTS (LG_LP + LBMPC_MAX_NC)            // + (1 - h_k)[r]
__host__ __device__ inline int lanegen_doubles(int lanes) { return LG_SLOTS * lanes; }

// coefficients c_j (c_j . z <= d_j) of per-lane row j: S rows [-S_A[j,:], 0], H rows F_bar Psi^i.  The tail of an S row
// is read from the row of zeros stored behind the H rows (CtxDev::nHs), its sign flipped by one xor on the high word.
template <int NX, int NZ>
__device__ __forceinline__ void lg_row(const CtxDev* __restrict__ C, const LdsView& L, int j, double* c) {
    const bool isS = j < C->nS;
    const double* ph = L.Hm + (isS ? C->nH : j - C->nS) * C->nzp;
    const double* p0 = isS ? L.Srow + j * (NX + 1) + 1 : ph;
    const long long sm = isS ? (long long)0x8000000000000000ull : 0ll;
#pragma unroll
    for (int k = 0; k < NX; ++k) c[k] = __longlong_as_double(__double_as_longlong(p0[k]) ^ sm);
#pragma unroll
    for (int k = NX; k < NZ; ++k) c[k] = ph[k];
}

// g = Ginv c for the block-diagonal Ginv (the skipped terms of the oracle's chains are exact zeros); the blocks are
// read as scalars (CtxDev::GiX, Gd)
template <int NX, int NZ>
__device__ __forceinline__ void lg_ginv(const CtxDev* __restrict__ C, const double* c, double* g) {
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < NX; ++k) acc = fma(C->GiX[i * NX + k], c[k], acc);
        g[i] = acc;
    }
#pragma unroll
    for (int i = NX; i < NZ; ++i) g[i] = fma(C->Gd[i], c[i], 0.0);
}

// slack of per-lane row j at z (e = q + s0): row_slack of the oracle
template <int NX, int NZ>
__device__ __forceinline__ double lg_slack(const CtxDev* __restrict__ C, const LdsView& L, const double* omh_l /* this lane's 1-h_k, stride lstr */,
                                           int lstr, unsigned ncmagic, int j, double alpha, const double* e, const double* z) {
    const bool isS = j < C->nS;
    const int h = isS ? 0 : j - C->nS;
    const double* ps = L.Srow + (isS ? j : 0) * (NX + 1);
    const double* ph = L.Hm + h * C->nzp;
    const int r = h - (int)__umulhi((unsigned)h, ncmagic) * C->nc;   // h % nc
    double sv[NX + 1], hv[NZ];
#pragma unroll
    for (int k = 0; k <= NX; ++k) sv[k] = ps[k];
#pragma unroll
    for (int k = 0; k < NZ; ++k) hv[k] = ph[k];
    double accS = alpha * sv[0];
#pragma unroll
    for (int k = 0; k < NX; ++k) accS = fma(sv[1 + k], e[k], accS);
    double accH = omh_l[r * lstr];
#pragma unroll
    for (int k = 0; k < NZ; ++k) accH = fma(-hv[k], z[k], accH);
    return isS ? accS : accH;
}

// Full scan for up to four QPs whose operands live in lanes src[0..3] (wave-uniform lane ids, -1 = empty slot): the S
// rows for all of them, the H rows for those with wantH set.  Lane l evaluates rows l, l+64, ...; strict <, ties to the
// lowest row (S rows come first).  Active rows are not excluded (see the header).  Results are wave-uniform.
template <int NX, int NZ>
__device__ __forceinline__ void lg_scan(const CtxDev* __restrict__ C, const LdsView& L, const double* ls, int lstr,
                                        unsigned ncmagic, const int* src, double alpha, const double* e, const double* z,
                                        unsigned hmask /* chunks of 64 H rows this lane's QP needs */, double* best_out, int* bj_out, unsigned* hpasses) {
    scan_S_lanes<NX>(C, L, src, alpha, e, best_out, bj_out);
    const int lane = lane_id();
    const int nS = C->nS, nH = C->nH, nc = C->nc, nzp = C->nzp;
    const int nchH = (nH + 63) >> 6;
#pragma unroll
    for (int g0 = 0; g0 < 4; g0 += 2) {
        const unsigned m0 = src[g0] >= 0 ? (unsigned)__builtin_amdgcn_readlane((int)hmask, src[g0] < 0 ? 0 : src[g0]) : 0u;
        const unsigned m1 = src[g0 + 1] >= 0 ? (unsigned)__builtin_amdgcn_readlane((int)hmask, src[g0 + 1] < 0 ? 0 : src[g0 + 1]) : 0u;
        const bool w0 = m0 != 0, w1 = m1 != 0;
        if (!w0 && !w1) continue;
        *hpasses += 1;
        const int s0l = src[g0] < 0 ? 0 : src[g0], s1l = src[g0 + 1] < 0 ? 0 : src[g0 + 1];
        const int o0 = s0l & (lstr - 1), o1 = s1l & (lstr - 1);   // their slots in the LDS region
        double zu0[NZ], zu1[NZ];
#pragma unroll
        for (int k = 0; k < NZ; ++k) { zu0[k] = readlane_f64(z[k], s0l); zu1[k] = readlane_f64(z[k], s1l); }
        double b0 = -QP_TOL_VIOL, b1 = -QP_TOL_VIOL;
        int j0 = 0x7fffffff, j1 = 0x7fffffff;
        for (int m = 0; m < nchH; ++m) {
            if ((((m0 | m1) >> m) & 1u) == 0) continue;   // both QPs certify this chunk
            hpasses[4] += 1;
            const int h = lane + 64 * m;
            const bool in = h < nH;
            const int hc = in ? h : nH - 1;
            const double* hp = L.Hm + hc * nzp;
            const int r = hc - (int)__umulhi((unsigned)hc, ncmagic) * nc;
            double row[NZ];
#pragma unroll
            for (int k = 0; k < NZ; ++k) row[k] = hp[k];
            double a0 = ls[(LG_LP + r) * lstr + o0], a1 = ls[(LG_LP + r) * lstr + o1];
#pragma unroll
            for (int k = 0; k < NZ; ++k) { a0 = fma(-row[k], zu0[k], a0); a1 = fma(-row[k], zu1[k], a1); }
            if (in && ((m0 >> m) & 1u) && a0 < b0) { b0 = a0; j0 = nS + h; }
            if (in && ((m1 >> m) & 1u) && a1 < b1) { b1 = a1; j1 = nS + h; }
        }
        wave_argmin(b0, j0);
        wave_argmin(b1, j1);
        if (w0 && j0 != 0x7fffffff && b0 < best_out[g0]) { best_out[g0] = b0; bj_out[g0] = j0; }
        if (w1 && j1 != 0x7fffffff && b1 < best_out[g0 + 1]) { best_out[g0 + 1] = b1; bj_out[g0 + 1] = j1; }
    }
}


// lowest row among the lanes' (value, row) pairs with the smallest value; val < -tol where a lane has a row.  The
// common case -- one lane holds the minimum -- costs one fp64 reduction, a ballot and a readlane.
__device__ __forceinline__ void lg_reduce(double val, int idx, double& best_out, int& bj_out) {
    const double bmin = wave_min(val);
    int j = 0x7fffffff;
    if (bmin < -QP_TOL_VIOL) {
        const uint64_t holders = __builtin_amdgcn_ballot_w64(val == bmin);
        if ((holders & (holders - 1)) == 0) j = __builtin_amdgcn_readlane(idx, (int)__builtin_ctzll(holders));
        else j = wave_min_i32(val == bmin ? idx : 0x7fffffff);
    }
    best_out = bmin; bj_out = j;
}

// lg_scan for tube sets of a few 64-row chunks (2x2: 128 padded rows): every lane keeps (slack, row) of its best S
// and H row for each of the four QPs, so that one reduction per QP finishes the scan -- no second pass over the rows,
// and the four reductions are independent instruction streams.  Same values, same tie rule.
template <int NX, int NZ>
__device__ __forceinline__ void lg_scan_small(const CtxDev* __restrict__ C, const LdsView& L, const double* ls, int lstr,
                                              unsigned ncmagic, const int* src, double alpha, const double* e, const double* z,
                                              unsigned hmask, double* best_out, int* bj_out, unsigned* hpasses) {
    constexpr int RLc = NX + 1;
    const int lane = lane_id();
    const int nS = C->nS, nH = C->nH, nc = C->nc, nzp = C->nzp;
    double best[4];
    int bj[4], sln[4];
    unsigned wh[4];   // chunks of H rows QP g needs
    {
        double ea[4], ee[4][NX];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            sln[g] = src[g] < 0 ? 0 : src[g];
            ea[g] = readlane_f64(alpha, sln[g]);
#pragma unroll
            for (int k = 0; k < NX; ++k) ee[g][k] = readlane_f64(e[k], sln[g]);
            best[g] = -QP_TOL_VIOL; bj[g] = 0x7fffffff;
            wh[g] = src[g] >= 0 ? (unsigned)__builtin_amdgcn_readlane((int)hmask, sln[g]) : 0u;
        }
        const int nch = C->nSp >> 6;
        for (int m = 0; m < nch; ++m) {   // pad rows (b = 1e300) are never violated
            const double* row = L.Srow + (lane + 64 * m) * RLc;
            double cf[RLc];
#pragma unroll
            for (int k = 0; k < RLc; ++k) cf[k] = row[k];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                double acc = ea[g] * cf[0];
#pragma unroll
                for (int k = 0; k < NX; ++k) acc = fma(cf[1 + k], ee[g][k], acc);
                const bool lt = acc < best[g];
                best[g] = lt ? acc : best[g];
                bj[g] = lt ? lane + 64 * m : bj[g];
            }
        }
    }
    if ((wh[0] | wh[1] | wh[2] | wh[3]) != 0) {
        const int nchH = (nH + 63) >> 6;
#pragma unroll
        for (int g0 = 0; g0 < 4; g0 += 2) {
            if ((wh[g0] | wh[g0 + 1]) == 0) continue;
            *hpasses += 1;
            const int o0 = sln[g0] & (lstr - 1), o1 = sln[g0 + 1] & (lstr - 1);
            double zu0[NZ], zu1[NZ];
#pragma unroll
            for (int k = 0; k < NZ; ++k) { zu0[k] = readlane_f64(z[k], sln[g0]); zu1[k] = readlane_f64(z[k], sln[g0 + 1]); }
            for (int m = 0; m < nchH; ++m) {
                if ((((wh[g0] | wh[g0 + 1]) >> m) & 1u) == 0) continue;   // both QPs certify this chunk
                hpasses[4] += 1;
                const int h = lane + 64 * m;
                const bool in = h < nH;
                const int hc = in ? h : nH - 1;
                const double* hp = L.Hm + hc * nzp;
                const int r = hc - (int)__umulhi((unsigned)hc, ncmagic) * nc;
                double row[NZ];
#pragma unroll
                for (int k = 0; k < NZ; ++k) row[k] = hp[k];
                double a0 = ls[(LG_LP + r) * lstr + o0], a1 = ls[(LG_LP + r) * lstr + o1];
#pragma unroll
                for (int k = 0; k < NZ; ++k) { a0 = fma(-row[k], zu0[k], a0); a1 = fma(-row[k], zu1[k], a1); }
                // strict <: an S row of this lane with the same slack keeps its (lower) index
                const bool l0 = in && ((wh[g0] >> m) & 1u) && a0 < best[g0], l1 = in && ((wh[g0 + 1] >> m) & 1u) && a1 < best[g0 + 1];
                best[g0] = l0 ? a0 : best[g0];             bj[g0] = l0 ? nS + h : bj[g0];
                best[g0 + 1] = l1 ? a1 : best[g0 + 1];     bj[g0 + 1] = l1 ? nS + h : bj[g0 + 1];
            }
        }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) lg_reduce(best[g], bj[g], best_out[g], bj_out[g]);
}

// Solve the QPs of the lanes flagged in `need` (one per lane; lanes >= lstr are never flagged).  On return, for every
// flagged lane either fallback = true (nothing else is meaningful, the hint list is untouched) or z[0..NZ), status,
// iters, nact, act[0..nact) (insertion order) are those of the oracle's qp_solve and the lane's hint list in HBM holds
// this solve's active set.  ls: this wave's LG_SLOTS x lstr doubles of LDS.
// ctr[0] += loop trips, ctr[1] += scan passes (S rows), ctr[2] += H-row passes, ctr[3] += QPs served by scans, ctr[4] += factor rebuilds, ctr[6] += chunks of 64 H rows evaluated (wave-uniform).
template <int NX, int NZ>
__device__ __forceinline__ void lanegen_qp_solve(const CtxDev* __restrict__ C, const LdsView& L, double* ls, int lstr, bool need,
                                                 double alpha, const double* v, const double* q, double min_omh,
                                                 int* hint_rows /* this lane's 16 slots or null */, int* hint_n, double* z,
                                                 int& status, int& iters, int& nact, int* act /* LG_NQL */, bool& fallback,
                                                 unsigned* ctr, long long* prof = nullptr /* LBMPC_PROFILE: 10 phase slots */) {
#ifdef LBMPC_PROFILE_FULL
#define LGP_T(var) const long long var = clock64()
#define LGP_ADD(slot, t0, t1) do { if (prof) prof[slot] += (t1) - (t0); } while (0)
#else
#define LGP_T(var)
#define LGP_ADD(slot, t0, t1)
#endif
    const int lane = lane_id();
    const int sl = lane & (lstr - 1);                // the flagged lanes of a call have distinct slots (lstr consecutive lanes); the others only ever read
    double* lsl = ls + sl;                           // element `slot` of this lane: lsl[slot * lstr]
    const int nS = C->nS, nc = C->nc, nz = C->nz;
    const unsigned ncmagic = 0xFFFFFFFFu / (unsigned)nc + 1u;   // h % nc for h < 2^16
    bool done = !need, have_p = false, pending = false;
    unsigned pend_hm = 0;   // chunks of 64 H rows the pending full scan has to evaluate
    fallback = false;
    int nq = 0, p = 0, nh = 0;
    iters = 0; status = 0;
    double sp = 0.0, cg = 0.0;
    double g[NZ], u[LG_NQL + 1];
    int cand[LG_NQL];
#pragma unroll
    for (int k = 0; k < NZ; ++k) { z[k] = 0.0; g[k] = 0.0; }
#pragma unroll
    for (int a = 0; a < LG_NQL; ++a) { u[a] = 0.0; act[a] = -1; cand[a] = -1; }
    u[LG_NQL] = 0.0;
    if (need) {   // 1 - h_k of this lane's (alpha, v)
        for (int r = 0; r < nc; ++r) lsl[(LG_LP + r) * lstr] = 1.0 - hk_value<NX>(C, alpha, v, r);
        // factor entries beyond the active rows are read and multiplied by exact zeros: they must be finite
#pragma unroll
        for (int i = 0; i < LG_LP; ++i) lsl[i * lstr] = 0.0;
    }
    if (need && hint_rows) {
        nh = *hint_n;
        if (nh > LG_NQL) { fallback = true; done = true; nh = 0; }
#pragma unroll
        for (int i = 0; i < LG_NQL; ++i) cand[i] = i < nh ? hint_rows[i] : -1;
    }
    wave_sync();

    auto isactive = [&](int j) {
        bool f = false;
#pragma unroll
        for (int a = 0; a < LG_NQL; ++a) f = f || (a < nq && act[a] == j);
        return f;
    };

    while (__builtin_amdgcn_ballot_w64(!done) != 0) {
        ctr[0] += 1;
        prio_tick<(NX == 4 ? LBMPC_PRIO_SHIFT4 : LBMPC_PRIO_SHIFT)>(-1);
        LGP_T(t_a);
        // ---- Step 1: a violated row at the current z for the lanes without a pivot ----
        const bool s1 = !done && !have_p && !pending;
        double best = -QP_TOL_VIOL;
        int bj = 0x7fffffff;
        bool got = false;
        if (__builtin_amdgcn_ballot_w64(s1) != 0) {
            double e[NX];
#pragma unroll
            for (int k = 0; k < NX; ++k) e[k] = q[k] + z[k];
            // candidates first: the previous step's active rows and, for an H row, its one-step-earlier twin
#pragma unroll
            for (int i = 0; i < LG_NQL; ++i) {
                if (__builtin_amdgcn_ballot_w64(s1 && i < nh) == 0) break;
#pragma unroll
                for (int tw = 0; tw < 2; ++tw) {
                    int j = cand[i];
                    if (tw) j = j >= nS + nc ? j - nc : -1;
                    const bool ev = s1 && i < nh && j >= 0 && !isactive(j);
                    if (__builtin_amdgcn_ballot_w64(ev) == 0) continue;
                    const double s = lg_slack<NX, NZ>(C, L, lsl + LG_LP * lstr, lstr, ncmagic, ev ? j : 0, alpha, e, z);
                    if (ev && (s < best || (s == best && bj != 0x7fffffff && j < bj))) { best = s; bj = j; }
                }
            }
            const bool need_full = s1 && bj == 0x7fffffff;
            if (__builtin_amdgcn_ballot_w64(need_full) != 0) {
                // certificate that no H row of a chunk is violated at z (as scan_rows, per chunk of 64 rows): slack_h >=
                // min_r (1-h_k)[r] - sum_k max_{h in chunk} |H[h,k]| |z_k| >= 1e-9; certified chunks are not evaluated
                unsigned hm = 0;
                const int nchH = (C->nH + 63) >> 6;
                for (int m = 0; m < nchH; ++m) {
                    double t = 0.0;
#pragma unroll
                    for (int k = 0; k < NZ; ++k) t = fma(C->cmHc[m][k], fabs(z[k]), t);
                    hm |= (t <= min_omh - 1e-9) ? 0u : (1u << m);
                }
                if (need_full) { pending = true; pend_hm = hm; }
            }
            got = s1 && !pending;
        }
        LGP_T(t_b);
        LGP_ADD(0, t_a, t_b);
        uint64_t pm = __builtin_amdgcn_ballot_w64(pending);
        if (pm != 0) {
            const bool others = __builtin_amdgcn_ballot_w64(!done && !pending && (have_p || got)) != 0;
            const int np = __builtin_popcountll(pm);
            int nscan = others ? (np & ~3) : np;
            if (nscan > 0) {
                double e[NX];
#pragma unroll
                for (int k = 0; k < NX; ++k) e[k] = q[k] + z[k];
                bool scanned = false;
                while (nscan > 0) {
                    ctr[1] += 1;
                    int src[4];
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const bool take = pm != 0 && nscan > 0;
                        src[gq] = take ? (int)__builtin_ctzll(pm) : -1;
                        if (take) { pm &= pm - 1; nscan -= 1; ctr[3] += 1; }
                    }
                    double fb[4];
                    int fj[4];
                    if (C->nSp <= 256) lg_scan_small<NX, NZ>(C, L, ls, lstr, ncmagic, src, alpha, e, z, pend_hm, fb, fj, ctr + 2);
                    else lg_scan<NX, NZ>(C, L, ls, lstr, ncmagic, src, alpha, e, z, pend_hm, fb, fj, ctr + 2);
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq)
                        if (lane == src[gq]) { best = fb[gq]; bj = fj[gq]; scanned = true; }
                }
                if (scanned) {
                    pending = false;
                    if (isactive(bj)) { fallback = true; done = true; }
                    got = !done;
                }
            }
        }
        LGP_T(t_c);
        LGP_ADD(1, t_b, t_c);
        {
            const bool optimal = got && !done && bj == 0x7fffffff;
            done = done || optimal;
            const bool newp = got && !done;
            if (__builtin_amdgcn_ballot_w64(newp) != 0) {
                // c_p, g = Ginv c_p, cg = c_p . g
                double cpn[NZ], gn[NZ], cgn = 0.0;
                lg_row<NX, NZ>(C, L, newp ? bj : 0, cpn);
                lg_ginv<NX, NZ>(C, cpn, gn);
#pragma unroll
                for (int k = 0; k < NZ; ++k) cgn = fma(cpn[k], gn[k], cgn);
                if (newp) {
                    p = bj; sp = best; have_p = true; cg = cgn;
#pragma unroll
                    for (int k = 0; k < NZ; ++k) g[k] = gn[k];
#pragma unroll
                    for (int a = 0; a <= LG_NQL; ++a) u[a] = a == nq ? 0.0 : u[a];
                }
            }
        }
        LGP_T(t_d);
        LGP_ADD(2, t_c, t_d);
        // ---- Step 2 for every lane that has a pivot ----
        const bool act2 = !done && !pending;
        if (act2) {
            iters += 1;
            if (iters > QP_MAX_ITERS) { status = LBMPC_ST_ITER_CAP; done = true; }
        }
        if (__builtin_amdgcn_ballot_w64(act2) == 0) continue;
        int nqmax = 0;   // wave-uniform bound of the loops over active rows
#pragma unroll
        for (int a = 0; a < LG_NQL; ++a) nqmax = __builtin_amdgcn_ballot_w64(act2 && nq > a) != 0 ? a + 1 : nqmax;
        // rhs = C_A g ; L y = rhs ; L' r = y   (rows a >= nq of a lane compute unused values)
        double rhs[LG_NQL], y[LG_NQL], r[LG_NQL];
#pragma unroll
        for (int a = 0; a < LG_NQL; ++a) {
            rhs[a] = 0.0; y[a] = 0.0; r[a] = 0.0;
            if (a < nqmax) {
                double c[NZ];
                lg_row<NX, NZ>(C, L, a < nq ? act[a] : 0, c);
                double acc = 0.0;
#pragma unroll
                for (int k = 0; k < NZ; ++k) acc = fma(c[k], g[k], acc);
                rhs[a] = a < nq ? acc : 0.0;
            }
        }
#pragma unroll
        for (int a = 0; a < LG_NQL; ++a)
            if (a < nqmax) {   // forward
                double acc = rhs[a];
#pragma unroll
                for (int k = 0; k < a; ++k) acc = fma(-lsl[(a * (a + 1) / 2 + k) * lstr], y[k], acc);
                y[a] = a < nq ? acc * lsl[(a * (a + 1) / 2 + a) * lstr] : 0.0;
            }
#pragma unroll
        for (int a = LG_NQL - 1; a >= 0; --a)
            if (a < nqmax) {   // backward, terms in descending k
                double acc = y[a];
#pragma unroll
                for (int k = LG_NQL - 1; k > a; --k)
                    if (k < nqmax) acc = fma(-lsl[(k * (k + 1) / 2 + a) * lstr], r[k], acc);   // r[k] = 0 for k >= nq, the entry is finite
                r[a] = a < nq ? acc * lsl[(a * (a + 1) / 2 + a) * lstr] : 0.0;
            }
        LGP_T(t_e);
        LGP_ADD(3, t_d, t_e);
        double delta = cg;
#pragma unroll
        for (int a = 0; a < LG_NQL; ++a) delta = fma(-rhs[a], r[a], delta);   // zeros beyond nq
        // primal direction zd = -(g - Wm' r), Wm[a] = Ginv c_a recomputed from the row
        double zd[NZ];
#pragma unroll
        for (int k = 0; k < NZ; ++k) zd[k] = g[k];
#pragma unroll
        for (int a = 0; a < LG_NQL; ++a)
            if (a < nqmax) {
                double c[NZ], wm[NZ];
                lg_row<NX, NZ>(C, L, a < nq ? act[a] : 0, c);
                lg_ginv<NX, NZ>(C, c, wm);
#pragma unroll
                for (int k = 0; k < NZ; ++k) zd[k] = fma(-wm[k], r[a], zd[k]);   // r[a] = 0 for a >= nq
            }
#pragma unroll
        for (int k = 0; k < NZ; ++k) zd[k] = -zd[k];
        LGP_T(t_f);
        LGP_ADD(4, t_e, t_f);
        const bool dependent = (nq >= nz) || !(delta > QP_TOL_DEP * cg);
        double rmax = 0.0;
#pragma unroll
        for (int a = 0; a < LG_NQL; ++a) rmax = (a < nq && fabs(r[a]) > rmax) ? fabs(r[a]) : rmax;
        double t1 = LB_INF;
        int kdrop = -1;
#pragma unroll
        for (int a = 0; a < LG_NQL; ++a)
            if (a < nqmax) {
                const bool ok = a < nq && r[a] > QP_TOL_R * rmax;
                const double tt = u[a] / (ok ? r[a] : 1.0);
                if (ok && tt < t1) { t1 = tt; kdrop = a; }
            }
        const double t2 = dependent ? LB_INF : -sp / delta;
        const double t = t1 < t2 ? t1 : t2;
        if (act2 && !done && t == LB_INF) { status = LBMPC_ST_QP_INFEASIBLE; done = true; }
        const bool upd = act2 && !done;
        bool full = upd && !dependent && t2 <= t1;
        bool drop = upd && !full;
        if (full && nq >= LG_NQL) { fallback = true; done = true; full = false; }   // beyond this path's capacity: general solver
        if (upd && !done) {
#pragma unroll
            for (int a = 0; a < LG_NQL; ++a) u[a] = fma(-t, r[a], u[a]);   // r[a] = 0 for a >= nq, t finite here
#pragma unroll
            for (int a = 0; a <= LG_NQL; ++a) u[a] = a == nq ? u[a] + t : u[a];
            if (!dependent) {
#pragma unroll
                for (int k = 0; k < NZ; ++k) z[k] = fma(t, zd[k], z[k]);
                sp = fma(t, delta, sp);
            }
        }
        LGP_T(t_g);
        LGP_ADD(5, t_f, t_g);
        if (__builtin_amdgcn_ballot_w64(full) != 0) {
            // constraint p joins the active set as number nq: its factor row is y, the diagonal from cg
            double dacc = cg;
#pragma unroll
            for (int k = 0; k < LG_NQL; ++k) dacc = fma(-y[k], y[k], dacc);   // y[k] = 0 for k >= nq
            const double dd = sqrt(dacc > 0.0 ? dacc : 0.0);
            const double inv = 1.0 / dd;
#pragma unroll
            for (int a = 0; a < LG_NQL; ++a)
                if (a <= nqmax) {
                    if (full && a == nq) {
#pragma unroll
                        for (int k = 0; k < a; ++k) lsl[(a * (a + 1) / 2 + k) * lstr] = y[k];
                        lsl[(a * (a + 1) / 2 + a) * lstr] = inv;
                    }
                }
            if (full) {
#pragma unroll
                for (int a = 0; a < LG_NQL; ++a) act[a] = a == nq ? p : act[a];
                nq += 1;
                have_p = false;
            }
        }
        LGP_T(t_h);
        LGP_ADD(6, t_g, t_h);
        if (__builtin_amdgcn_ballot_w64(drop) != 0) {
            ctr[4] += 1;
            // drop active constraint kdrop, stay in step 2 with the same pivot
#pragma unroll
            for (int a = 0; a < LG_NQL; ++a) {
                const bool mv = drop && a >= kdrop && a < nq;
                if (a + 1 < LG_NQL) act[a] = (mv && a < nq - 1) ? act[a + 1] : act[a];
                u[a] = mv ? u[a + 1] : u[a];
            }
            if (drop) nq -= 1;
            // factor rows a >= kdrop: entries b < kdrop are those of the old row a+1, the others are recomputed
            int kdmin = LG_NQL, nq2max = 0;   // wave-uniform loop bounds over the dropping lanes
#pragma unroll
            for (int a = LG_NQL - 1; a >= 0; --a) kdmin = __builtin_amdgcn_ballot_w64(drop && kdrop <= a) != 0 ? a : kdmin;
#pragma unroll
            for (int a = 0; a < LG_NQL; ++a) nq2max = __builtin_amdgcn_ballot_w64(drop && nq > a) != 0 ? a + 1 : nq2max;
#pragma unroll
            for (int a = 0; a < LG_NQL - 1; ++a)
                if (a >= kdmin && a < nq2max) {
                    const bool pr = drop && a >= kdrop && a < nq;
#pragma unroll
                    for (int b = 0; b <= a; ++b) {
                        const double o = lsl[((a + 1) * (a + 2) / 2 + b) * lstr];
                        if (pr && b < kdrop) lsl[(a * (a + 1) / 2 + b) * lstr] = o;
                    }
                }
#pragma unroll
            for (int a = 0; a < LG_NQL - 1; ++a)
                if (a >= kdmin && a < nq2max) {
                    const bool pr = drop && a >= kdrop && a < nq;
                    double ca[NZ], wm[NZ];
                    lg_row<NX, NZ>(C, L, pr ? act[a] : 0, ca);
                    lg_ginv<NX, NZ>(C, ca, wm);
#pragma unroll
                    for (int b = 0; b <= a; ++b)
                        if (b >= kdmin) {
                            const bool pb = pr && b >= kdrop;
                            double cb[NZ];
                            if (b < a) lg_row<NX, NZ>(C, L, pb ? act[b] : 0, cb);
                            double acc = 0.0;
#pragma unroll
                            for (int k = 0; k < NZ; ++k) acc = fma(b < a ? cb[k] : ca[k], wm[k], acc);
#pragma unroll
                            for (int k = 0; k < b; ++k)
                                acc = fma(-lsl[(a * (a + 1) / 2 + k) * lstr], lsl[(b * (b + 1) / 2 + k) * lstr], acc);
                            if (b == a) {
                                const double dd = sqrt(acc > 0.0 ? acc : 0.0);
                                const double inv = 1.0 / dd;
                                if (pb) lsl[(a * (a + 1) / 2 + a) * lstr] = inv;
                            } else {
                                const double lab = acc * lsl[(b * (b + 1) / 2 + b) * lstr];
                                if (pb) lsl[(a * (a + 1) / 2 + b) * lstr] = lab;
                            }
                        }
                }
        }
        LGP_T(t_i);
        LGP_ADD(7, t_h, t_i);
    }
    nact = nq;
    if (need && !fallback && hint_rows) {   // this solve's active set = the next step's candidates
#pragma unroll
        for (int a = 0; a < LG_NQL; ++a) if (a < nq) hint_rows[a] = act[a];
        *hint_n = nq;
    }
}

}  // namespace lbmpc
