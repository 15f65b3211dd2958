// qp_lane.hpp -- Goldfarb-Idnani for the S-row regime of the tube QP: ONE QP PER LANE, 64 QPs per wave in lock-step.
//
// While no row of F_bar Psi^i is active, pivot or violated, the QP only moves s0 (the S rows touch the first nx
// variables and (2P)^-1 is block diagonal): at most nx rows are active and the whole solver state -- active rows,
// multipliers, C_A, Ginv C_A', the Cholesky factor -- is 4x4.  It fits the registers of one lane, so the algebra of
// up to 64 QPs advances in one instruction stream (the 16-lane-row solver of qp_group.hpp runs 4), and the pivot
// scans, which stay "row per lane" with the QP's operands as wave-uniform scalars, serve exactly the QPs that need
// one, four per pass over the rows.  Arithmetic, operation order and tie rules are those of the oracle's qp_solve
// restricted to k < nx (the skipped terms are exact zeros there), so results, active sets and iteration counts are
// bit-identical; a QP leaves this path (`fallback`, re-solved from scratch by group_qp_solve) as soon as
//   * its pivot candidates contain a row of F_bar Psi^i, or more than nx rows,
//   * the certificate "no H row can be violated at z" (see scan_rows) fails at a full scan,
//   * a (nx+1)-th row would join the active set, or a full scan returns an active row (never seen; exclusion of
//     active rows is not tested per row here, their slack is an exact zero up to rounding, far above -tol).
#pragma once

namespace lbmpc {

// Full scan of the S rows for up to four QPs whose operands live in lanes src[0..3] (wave-uniform lane ids, -1 =
// empty slot).  Lane l evaluates rows l, l+64, ...; slack = alpha S_b[j] + S_A[j,:].e with the arithmetic of
// row_slack; strict <, ties to the lowest row.  Results are wave-uniform.
template <int NX>
__device__ __forceinline__ void scan_S_lanes(const CtxDev* __restrict__ C, const LdsView& L, const int* src, double alpha,
                                             const double* e, double* best_out, int* bj_out) {
    constexpr int RLc = NX + 1;
    const int lane = lane_id();
    double ea[4], ee[4][NX], best[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int sl = src[g] < 0 ? 0 : src[g];
        ea[g] = readlane_f64(alpha, sl);
#pragma unroll
        for (int k = 0; k < NX; ++k) ee[g][k] = readlane_f64(e[k], sl);
        best[g] = -QP_TOL_VIOL;
    }
    const int nch = C->nSp >> 6;  // even: nSp is a multiple of 128; pad rows are never violated
    const double* base = L.Srow + lane * RLc;
    double A[RLc], B[RLc];
    // Pass 1 keeps the running minimum only (one v_min_f64 per row and QP; carrying the row index along costs a compare
    // and three selects more).  The index comes from pass 2 below.
    auto eval = [&](const double* cf) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            double acc = ea[g] * cf[0];
#pragma unroll
            for (int k = 0; k < NX; ++k) acc = fma(cf[1 + k], ee[g][k], acc);
            // v_min_f64 as written: the builtin adds a canonicalising v_max_f64 on the loop-carried operand
            asm("v_min_f64 %0, %1, %2" : "=v"(best[g]) : "v"(acc), "v"(best[g]));
        }
    };
#pragma unroll
    for (int k = 0; k < RLc; ++k) A[k] = base[k];
    for (int m = 0; m < nch; m += 2) {
        const double* pb = base + (m + 1) * 64 * RLc;
#pragma unroll
        for (int k = 0; k < RLc; ++k) B[k] = pb[k];
        __builtin_amdgcn_sched_barrier(0);
        eval(A);
        const double* pa = base + (m + 2 < nch ? m + 2 : 0) * 64 * RLc;
#pragma unroll
        for (int k = 0; k < RLc; ++k) A[k] = pa[k];
        __builtin_amdgcn_sched_barrier(0);
        eval(B);
    }
    // Pass 2: the lowest row whose slack equals the wave minimum.  A lane that holds the minimum owns rows
    // lane0 + 64 m; lane m re-evaluates row lane0 + 64 m (same operations, same bits) and the lowest hit is that
    // lane's first row at the minimum; over the lanes that hold it (one, barring exact ties) the lowest row wins.
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const double bmin = wave_min(best[g]);
        int jmin = 0x7fffffff;
        if (bmin < -QP_TOL_VIOL) {
            uint64_t holders = __builtin_amdgcn_ballot_w64(best[g] == bmin);
            while (holders != 0) {
                const int l0 = (int)__builtin_ctzll(holders);
                holders &= holders - 1;
                const int m = lane < nch ? lane : nch - 1;
                const double* row = L.Srow + (l0 + 64 * m) * RLc;
                double acc = ea[g] * row[0];
#pragma unroll
                for (int k = 0; k < NX; ++k) acc = fma(row[1 + k], ee[g][k], acc);
                const uint64_t hit = __builtin_amdgcn_ballot_w64(lane < nch && acc == bmin);
                const int j = l0 + 64 * (int)__builtin_ctzll(hit | (1ull << 63));
                jmin = (hit != 0 && j < jmin) ? j : jmin;
            }
        }
        best_out[g] = bmin; bj_out[g] = jmin;
    }
}

// Solve the QPs of the lanes flagged in `need` (one per lane).  On return, for every flagged lane either
// fallback = true (nothing else is meaningful, the hint list is untouched) or z[0..NX) = s0 (the rest of z is 0),
// status, iters, nact, act[0..nact) (insertion order) are those of the oracle's qp_solve, and the lane's hint list in
// HBM holds this solve's active set.  ctr[0] += loop trips, ctr[1] += scan passes, ctr[3] += QPs served by them (wave-uniform).
template <int NX>
__device__ __forceinline__ void lane_qp_solve(const CtxDev* __restrict__ C, const LdsView& L, bool need, double alpha,
                                              const double* q, double min_omh, int* hint_rows /* this lane's 16 slots or null */,
                                              int* hint_n, double* z, int& status, int& iters, int& nact, int* act,
                                              bool& fallback, unsigned* ctr, long long* prof = nullptr) {
    const int lane = lane_id();
    const int nS = C->nS, nz = C->nz;
    bool done = !need, have_p = false;
    fallback = false;
    int nq = 0, p = 0, nh = 0;
    iters = 0; status = 0;
    double sp = 0.0, cg = 0.0;
    double g[NX], cp[NX], u[NX + 1], CA[NX][NX], Wm[NX][NX];
    int cand[NX];
#pragma unroll
    for (int k = 0; k < NX; ++k) {
        z[k] = 0.0; g[k] = 0.0; cp[k] = 0.0; u[k] = 0.0; act[k] = -1; cand[k] = -1;
#pragma unroll
        for (int j = 0; j < NX; ++j) { CA[k][j] = 0.0; Wm[k][j] = 0.0; }
    }
    u[NX] = 0.0;
    if (need && hint_rows) {
        nh = *hint_n;
        if (nh > NX) { fallback = true; done = true; nh = 0; }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            cand[i] = i < nh ? hint_rows[i] : -1;
            if (cand[i] >= nS) { fallback = true; done = true; }
        }
    }
    // (2 Px)^-1: the leading NX x NX block of Ginv (wave-uniform)
    double Gi[NX][NX];
#pragma unroll
    for (int i = 0; i < NX; ++i)
#pragma unroll
        for (int k = 0; k < NX; ++k) Gi[i][k] = L.Ginv[i * nz + k];

    bool pending = false;   // this lane waits for a full scan (its z does not move meanwhile)
    while (__builtin_amdgcn_ballot_w64(!done) != 0) {
        ctr[0] += 1;
        prio_tick<(NX == 4 ? LBMPC_PRIO_SHIFT4 : LBMPC_PRIO_SHIFT)>(-1);
#ifdef LBMPC_PROFILE_FULL
        const long long pt0 = clock64();
        long long pscan = 0;
        if (prof) prof[4] += 1;
#endif
        // ---- Step 1: a violated row at the current z for the lanes without a pivot ----
        // Candidates are evaluated by the lane itself; a lane that needs the full scan waits until four such lanes can
        // share one pass over the rows (or until no other lane of the wave has a step 2 to do): a pass costs the same
        // for one QP as for four, and per lane the sequence of operations is unchanged.
        const bool s1 = !done && !have_p && !pending;
        double best = -QP_TOL_VIOL;
        int bj = 0x7fffffff;
        bool got = false;   // the lane has a step-1 result to act on in this trip
        if (__builtin_amdgcn_ballot_w64(s1) != 0) {
            double e[NX];
#pragma unroll
            for (int k = 0; k < NX; ++k) e[k] = q[k] + z[k];
            if (__builtin_amdgcn_ballot_w64(s1 && nh > 0) != 0) {   // candidates first, in list order
#pragma unroll
                for (int i = 0; i < NX; ++i) {
                    const int j = cand[i];
                    bool isact = false;
#pragma unroll
                    for (int a = 0; a < NX; ++a) isact = isact || (a < nq && act[a] == j);
                    const bool ev = s1 && i < nh && !isact;
                    const double* row = L.Srow + (ev ? j : 0) * (NX + 1);
                    double acc = alpha * row[0];
#pragma unroll
                    for (int k = 0; k < NX; ++k) acc = fma(row[1 + k], e[k], acc);
                    if (ev && (acc < best || (acc == best && bj != 0x7fffffff && j < bj))) { best = acc; bj = j; }
                }
            }
            const bool need_full = s1 && bj == 0x7fffffff;
            if (__builtin_amdgcn_ballot_w64(need_full) != 0) {
                // certificate that no H row is violated at z (z is zero beyond NX)
                double t = 0.0;
#pragma unroll
                for (int k = 0; k < NX; ++k) t = fma(C->cmH[k], fabs(z[k]), t);
                if (need_full) {
                    if (!(t <= min_omh - 1e-9)) { fallback = true; done = true; }
                    else pending = true;
                }
            }
            got = s1 && !pending && !done;
        }
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
#ifdef LBMPC_PROFILE_FULL
                    const long long ps0 = clock64();
#endif
                    scan_S_lanes<NX>(C, L, src, alpha, e, fb, fj);
#ifdef LBMPC_PROFILE_FULL
                    pscan += clock64() - ps0;
                    if (prof) prof[5] += 1;
#endif
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq)
                        if (lane == src[gq]) { best = fb[gq]; bj = fj[gq]; scanned = true; }
                }
                if (scanned) {   // an active row never wins a scan (its slack is ~0 > -tol); if one does, leave this path
                    pending = false;
                    bool isact = false;
#pragma unroll
                    for (int a = 0; a < NX; ++a) isact = isact || (a < nq && act[a] == bj);
                    if (isact) { fallback = true; done = true; }
                    got = !done;
                }
            }
        }
        {
            const bool optimal = got && !done && bj == 0x7fffffff;
            done = done || optimal;
            const bool newp = got && !done;
            if (__builtin_amdgcn_ballot_w64(newp) != 0) {
                // c_p = -S_A[p,:], g = Ginv c_p, cg = c_p . g
                const double* row = L.Srow + (newp ? bj : 0) * (NX + 1) + 1;
                double cpn[NX], gn[NX], cgn = 0.0;
#pragma unroll
                for (int k = 0; k < NX; ++k) cpn[k] = -row[k];
#pragma unroll
                for (int i = 0; i < NX; ++i) {
                    double acc = 0.0;
#pragma unroll
                    for (int k = 0; k < NX; ++k) acc = fma(Gi[i][k], cpn[k], acc);
                    gn[i] = acc;
                }
#pragma unroll
                for (int k = 0; k < NX; ++k) cgn = fma(cpn[k], gn[k], cgn);
                if (newp) {
                    p = bj; sp = best; have_p = true; cg = cgn;
#pragma unroll
                    for (int k = 0; k < NX; ++k) { cp[k] = cpn[k]; g[k] = gn[k]; }
#pragma unroll
                    for (int a = 0; a <= NX; ++a) u[a] = a == nq ? 0.0 : u[a];
                }
            }
        }
#ifdef LBMPC_PROFILE_FULL
        const long long pt1 = clock64();
        if (prof) { prof[0] += pscan; prof[2] += (pt1 - pt0) - pscan; }
#endif
        // ---- Step 2 for every lane that has a pivot ----
        const bool act2 = !done && !pending;
        if (act2) {
            iters += 1;
            if (iters > QP_MAX_ITERS) { status = LBMPC_ST_ITER_CAP; done = true; }
        }
        if (__builtin_amdgcn_ballot_w64(act2) == 0) continue;
        // M = C_A Ginv C_A', Cholesky, r = M^-1 C_A g  (rows a >= nq compute unused values)
        double rhs[NX], r[NX], Lm[NX][NX], invd[NX];
#pragma unroll
        for (int a = 0; a < NX; ++a) {
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < NX; ++k) acc = fma(CA[a][k], g[k], acc);
            rhs[a] = a < nq ? acc : 0.0;
        }
#pragma unroll
        for (int a = 0; a < NX; ++a)
#pragma unroll
            for (int b = 0; b <= a; ++b) {
                double acc = 0.0;
#pragma unroll
                for (int k = 0; k < NX; ++k) acc = fma(CA[b][k], Wm[a][k], acc);
#pragma unroll
                for (int k = 0; k < b; ++k) acc = fma(-Lm[a][k], Lm[b][k], acc);
                if (a == b) { Lm[a][a] = sqrt(acc > 0.0 ? acc : 0.0); invd[a] = 1.0 / Lm[a][a]; }
                else Lm[a][b] = acc * invd[b];
            }
#pragma unroll
        for (int a = 0; a < NX; ++a) {   // forward
            double acc = rhs[a];
#pragma unroll
            for (int k = 0; k < a; ++k) acc = fma(-Lm[a][k], r[k], acc);
            r[a] = a < nq ? acc * invd[a] : 0.0;
        }
#pragma unroll
        for (int a = NX - 1; a >= 0; --a) {   // backward, terms in descending k
            double acc = r[a];
#pragma unroll
            for (int k = NX - 1; k > a; --k) acc = k < nq ? fma(-Lm[k][a], r[k], acc) : acc;
            r[a] = a < nq ? acc * invd[a] : 0.0;
        }
        double delta = cg;
#pragma unroll
        for (int a = 0; a < NX; ++a) delta = a < nq ? fma(-rhs[a], r[a], delta) : delta;
        double zd[NX];
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            double acc = g[k];
#pragma unroll
            for (int a = 0; a < NX; ++a) acc = a < nq ? fma(-Wm[a][k], r[a], acc) : acc;
            zd[k] = -acc;
        }
        const bool dependent = (nq >= nz) || !(delta > QP_TOL_DEP * cg);
        double rmax = 0.0;
#pragma unroll
        for (int a = 0; a < NX; ++a) rmax = (a < nq && fabs(r[a]) > rmax) ? fabs(r[a]) : rmax;
        double t1 = LB_INF;
        int kdrop = -1;
#pragma unroll
        for (int a = 0; a < NX; ++a)
            if (a < nq && r[a] > QP_TOL_R * rmax) {
                const double tt = u[a] / r[a];
                if (tt < t1) { t1 = tt; kdrop = a; }
            }
        const double t2 = dependent ? LB_INF : -sp / delta;
        const double t = t1 < t2 ? t1 : t2;
        if (act2 && !done && t == LB_INF) { status = LBMPC_ST_QP_INFEASIBLE; done = true; }
        const bool upd = act2 && !done;
        const bool full = upd && !dependent && t2 <= t1;
        const bool drop = upd && !full;
        if (full && nq >= NX) { fallback = true; done = true; }   // a row beyond the S regime's rank: general solver
        if (upd && !done) {
#pragma unroll
            for (int a = 0; a < NX; ++a) u[a] = a < nq ? fma(-t, r[a], u[a]) : u[a];
#pragma unroll
            for (int a = 0; a <= NX; ++a) u[a] = a == nq ? u[a] + t : u[a];
            if (!dependent) {
#pragma unroll
                for (int k = 0; k < NX; ++k) z[k] = fma(t, zd[k], z[k]);
                sp = fma(t, delta, sp);
            }
            if (full) {   // constraint p joins the active set as number nq
#pragma unroll
                for (int a = 0; a < NX; ++a) {
                    const bool here = a == nq;
                    act[a] = here ? p : act[a];
#pragma unroll
                    for (int k = 0; k < NX; ++k) { CA[a][k] = here ? cp[k] : CA[a][k]; Wm[a][k] = here ? g[k] : Wm[a][k]; }
                }
                nq += 1;
                have_p = false;
            } else if (drop) {   // drop active constraint kdrop, stay in step 2 with the same pivot
#pragma unroll
                for (int a = 0; a < NX; ++a) {
                    const bool mv = a >= kdrop && a < nq;
                    if (a + 1 < NX) {
                        act[a] = (mv && a < nq - 1) ? act[a + 1] : act[a];
#pragma unroll
                        for (int k = 0; k < NX; ++k) {
                            CA[a][k] = (mv && a < nq - 1) ? CA[a + 1][k] : CA[a][k];
                            Wm[a][k] = (mv && a < nq - 1) ? Wm[a + 1][k] : Wm[a][k];
                        }
                    }
                    u[a] = mv ? u[a + 1] : u[a];
                }
                nq -= 1;
            }
        }
#ifdef LBMPC_PROFILE_FULL
        if (prof) prof[1] += clock64() - pt1;
#endif
    }
    nact = nq;
    if (need && !fallback && hint_rows) {   // this solve's active set = the next step's candidates
#pragma unroll
        for (int a = 0; a < NX; ++a) if (a < nq) hint_rows[a] = act[a];
        *hint_n = nq;
    }
}

}  // namespace lbmpc
