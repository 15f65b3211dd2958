// lbmpc_mex.cpp -- MEX gateway: MATLAB <-> liblbmpc_hip.so (include/lbmpc.h).
//
//   out = lbmpc_mex(cmd, args...)
//
// Built where MATLAB exists with
//   mex -R2018a lbmpc_mex.cpp -I../../include -L../csrc -llbmpc_hip
// The build image has no MATLAB: there this file is compiled against the stand-in mex.h of tests/mex_stub/ and its
// mexFunction is driven by tests/test_mex_gateway.py (argument order, plhs handling, struct marshalling, error
// mapping) -- the real mex build is unverified.  MATLAB owns every input mxArray (read-only via mxGetDoubles);
// outputs are allocated with mxCreateDoubleMatrix and filled by the library.  Controller contexts are returned as
// uint64 handles, freed by 'destroy' / mexAtExit; the MEX file stays locked while any context lives.
//   lbmpc_mex('init', device)            optional, before anything else: GPU for the single-device calls (default 0)
//   h = lbmpc_mex('create', p [, ngpu])  ngpu > 1: 'mc_run' on this handle shards its trials over GPUs 0..ngpu-1
#include <cstdint>
#include <cstring>
#include <set>
#include <string>

#include "lbmpc.h"
#include "mex.h"

struct Handle {            // what a uint64 handle points to
    lbmpc_ctx* ctx = nullptr;      // single-device context ('step', 'mc_run' with one GPU)
    lbmpc_multi* multi = nullptr;  // one context per device ('mc_run' with ngpu > 1)
};
static std::set<Handle*> g_live;
static bool g_init = false;

static void fail(const char* id, int rc) {
    mexErrMsgIdAndTxt(id, "lbmpc error %d: %s", rc, lbmpc_last_error());
}
static void release(Handle* h) {
    if (h->ctx) lbmpc_destroy(h->ctx);
    if (h->multi) lbmpc_multi_destroy(h->multi);
    delete h;
}
static void at_exit() {
    for (Handle* h : g_live) release(h);
    g_live.clear();
}
static const double* dbl(const mxArray* a) { return mxGetDoubles(a); }
static int rows(const mxArray* a) { return (int)mxGetM(a); }
static int cols(const mxArray* a) { return (int)mxGetN(a); }
static const mxArray* field(const mxArray* s, const char* name) {
    const mxArray* f = mxGetField(s, 0, name);
    if (!f) mexErrMsgIdAndTxt("lbmpc:field", "missing field %s", name);
    return f;
}
static int ifield(const mxArray* s, const char* name) { return (int)mxGetScalar(field(s, name)); }
static Handle* handle(const mxArray* a) {
    if (!a || mxGetClassID(a) != mxUINT64_CLASS || mxIsEmpty(a)) mexErrMsgIdAndTxt("lbmpc:handle", "expected a uint64 context handle");
    Handle* h = reinterpret_cast<Handle*>(*reinterpret_cast<uint64_t*>(mxGetData(a)));
    if (!g_live.count(h)) mexErrMsgIdAndTxt("lbmpc:handle", "stale or foreign context handle");
    return h;
}
static void need_args(int nrhs, int n, const char* usage) {
    if (nrhs < n) mexErrMsgIdAndTxt("lbmpc:usage", "%s", usage);
}

static void fill_sampler(const mxArray* s, lbmpc_sampler* o) {
    o->nx = ifield(s, "nx"); o->nu = ifield(s, "nu");
    o->EV_A = dbl(field(s, "EV_A")); o->EV_b = dbl(field(s, "EV_b")); o->nEV = rows(field(s, "EV_A"));
    o->EV_lo = dbl(field(s, "EV_lo")); o->EV_hi = dbl(field(s, "EV_hi"));
    o->LV_A = dbl(field(s, "LV_A")); o->LV_b = dbl(field(s, "LV_b")); o->nLV = rows(field(s, "LV_A"));
    o->LV_lo = dbl(field(s, "LV_lo")); o->LV_hi = dbl(field(s, "LV_hi"));
    o->B = dbl(field(s, "B")); o->min_u_LV = dbl(field(s, "min_u_LV")); o->max_u_LV = dbl(field(s, "max_u_LV"));
    const mxArray* eu = mxGetField(s, 0, "EV_U");   // optional oriented proposal boxes
    const mxArray* lu = mxGetField(s, 0, "LV_U");
    o->EV_U = (eu && !mxIsEmpty(eu)) ? dbl(eu) : nullptr;
    o->LV_U = (lu && !mxIsEmpty(lu)) ? dbl(lu) : nullptr;
}

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    if (nrhs < 1 || !mxIsChar(prhs[0])) mexErrMsgIdAndTxt("lbmpc:usage", "lbmpc_mex(cmd, ...)");
    char cmdbuf[64];
    if (mxGetString(prhs[0], cmdbuf, sizeof cmdbuf)) mexErrMsgIdAndTxt("lbmpc:usage", "command name too long");
    const std::string cmd(cmdbuf);
    if (!g_init || cmd == "init") {
        const int device = (cmd == "init" && nrhs > 1) ? (int)mxGetScalar(prhs[1]) : 0;
        int rc = lbmpc_init(device);
        if (rc) fail("lbmpc:init", rc);
        if (!g_init) mexAtExit(at_exit);
        g_init = true;
        if (cmd == "init") return;
    }
    if (cmd == "support_batch") {  // h = lbmpc_mex('support_batch', A, Vw, s, alpha, D, transpose)
        need_args(nrhs, 7, "h = lbmpc_mex('support_batch', A, Vw, s, alpha, D, transpose)");
        const int nx = rows(prhs[1]);
        const int64_t nd = cols(prhs[5]);
        plhs[0] = mxCreateDoubleMatrix(nd, 1, mxREAL);
        int rc = lbmpc_support_batch(dbl(prhs[1]), nx, dbl(prhs[2]), rows(prhs[2]), (int)mxGetScalar(prhs[3]),
                                     mxGetScalar(prhs[4]), dbl(prhs[5]), nd, (int)mxGetScalar(prhs[6]), mxGetDoubles(plhs[0]));
        if (rc < 0) fail("lbmpc:support_batch", rc);
    } else if (cmd == "mrpi_s_alpha") {  // [s, alpha, Ms] = lbmpc_mex('mrpi_s_alpha', A, Vw, WA, Wb, Dirs, eps, mode)
        need_args(nrhs, 8, "[s, alpha, Ms] = lbmpc_mex('mrpi_s_alpha', A, Vw, WA, Wb, Dirs, eps, mode)");
        int s = 0; double alpha = 0, Ms = 0;
        int rc = lbmpc_mrpi_s_alpha(dbl(prhs[1]), rows(prhs[1]), dbl(prhs[2]), rows(prhs[2]), dbl(prhs[3]), dbl(prhs[4]),
                                    rows(prhs[3]), dbl(prhs[5]), cols(prhs[5]), mxGetScalar(prhs[6]), (int)mxGetScalar(prhs[7]),
                                    60, &s, &alpha, &Ms);
        if (rc < 0) fail("lbmpc:mrpi_s_alpha", rc);
        if (rc > 0) mexErrMsgIdAndTxt("lbmpc:mrpi_s_alpha", "the (s, alpha) search hit its iteration cap at s = %d, alpha = %g", s, alpha);
        plhs[0] = mxCreateDoubleScalar(s);
        if (nlhs > 1) plhs[1] = mxCreateDoubleScalar(alpha);
        if (nlhs > 2) plhs[2] = mxCreateDoubleScalar(Ms);
    } else if (cmd == "learn_set") {  // [alpha, v, status] = lbmpc_mex('learn_set', WA, Wb, samples)  samples nx x N
        need_args(nrhs, 4, "[alpha, v, status] = lbmpc_mex('learn_set', WA, Wb, samples)");
        const int nW = rows(prhs[1]), nx = cols(prhs[1]);
        if (nW > LBMPC_MAX_NW) mexErrMsgIdAndTxt("lbmpc:learn_set", "W has %d rows (limit %d)", nW, LBMPC_MAX_NW);
        double m[LBMPC_MAX_NW];
        int rc = lbmpc_halfspace_max(dbl(prhs[1]), nW, nx, dbl(prhs[3]), cols(prhs[3]), 1, 0, m);
        if (rc < 0) fail("lbmpc:halfspace_max", rc);
        plhs[0] = mxCreateDoubleScalar(0);
        mxArray* v = mxCreateDoubleMatrix(nx, 1, mxREAL);
        int st = 0;
        rc = lbmpc_homothety_lp(dbl(prhs[1]), dbl(prhs[2]), nW, nx, m, 1, LBMPC_TIE_CENTER, mxGetDoubles(plhs[0]), mxGetDoubles(v), nullptr, &st);
        if (rc < 0) fail("lbmpc:homothety_lp", rc);
        if (nlhs > 1) plhs[1] = v;
        if (nlhs > 2) plhs[2] = mxCreateDoubleScalar(st);
    } else if (cmd == "sample_sets") {  // [samples, m] = lbmpc_mex('sample_sets', sampler, WA, N, seed, set0)
        need_args(nrhs, 6, "[samples, m] = lbmpc_mex('sample_sets', sampler, WA, N, seed, set0)");
        lbmpc_sampler smp; fill_sampler(prhs[1], &smp);
        const int nW = rows(prhs[2]);
        const int64_t N = (int64_t)mxGetScalar(prhs[3]);
        plhs[0] = mxCreateDoubleMatrix(smp.nx, N, mxREAL);
        mxArray* m = mxCreateDoubleMatrix(nW, 1, mxREAL);
        int rc = lbmpc_sample_information_sets(&smp, dbl(prhs[2]), nW, N, 1, (uint64_t)mxGetScalar(prhs[4]),
                                               (uint64_t)mxGetScalar(prhs[5]), 100, mxGetDoubles(plhs[0]), mxGetDoubles(m), nullptr);
        if (rc < 0) fail("lbmpc:sample_sets", rc);
        if (nlhs > 1) plhs[1] = m;
    } else if (cmd == "sample_polytope") {
        // [x, status] = lbmpc_mex('sample_polytope', A, b, lo, hi, n, seed, stream, index0, purpose)
        // n points uniform on {x : A x <= b} by rejection from the box [lo, hi]; point i is draw (seed, stream, index0 + i,
        // purpose) of the library's Philox stream (ModelingCar.m:47-91 consume MATLAB's global rand instead)
        need_args(nrhs, 10, "[x, status] = lbmpc_mex('sample_polytope', A, b, lo, hi, n, seed, stream, index0, purpose)");
        const int nr = rows(prhs[1]), nx = cols(prhs[1]);
        const int64_t n = (int64_t)mxGetScalar(prhs[5]);
        if (rows(prhs[2]) * cols(prhs[2]) != nr || rows(prhs[3]) * cols(prhs[3]) != nx || rows(prhs[4]) * cols(prhs[4]) != nx || n < 0)
            mexErrMsgIdAndTxt("lbmpc:sample_polytope", "A nr x nx, b nr, lo and hi nx, n >= 0");
        plhs[0] = mxCreateDoubleMatrix(nx, n, mxREAL);
        mxArray* st = mxCreateNumericMatrix(n, 1, mxINT32_CLASS, mxREAL);
        int rc = lbmpc_sample_polytope(dbl(prhs[1]), dbl(prhs[2]), nr, nx, dbl(prhs[3]), dbl(prhs[4]), n, (uint64_t)mxGetScalar(prhs[6]),
                                       (uint64_t)mxGetScalar(prhs[7]), (uint32_t)mxGetScalar(prhs[8]), (int)mxGetScalar(prhs[9]), 1000,
                                       mxGetDoubles(plhs[0]), (int*)mxGetData(st));
        if (rc < 0) fail("lbmpc:sample_polytope", rc);
        if (nlhs > 1) plhs[1] = st;
    } else if (cmd == "create") {  // h = lbmpc_mex('create', p)  p: struct with the lbmpc_params fields (A,b unpacked)
        need_args(nrhs, 2, "h = lbmpc_mex('create', p [, ngpu])");
        const mxArray* s = prhs[1];
        const int ngpu = nrhs > 2 ? (int)mxGetScalar(prhs[2]) : 1;
        lbmpc_params p;
        p.nx = ifield(s, "nx"); p.nu = ifield(s, "nu"); p.N = ifield(s, "N"); p.nc = ifield(s, "nc"); p.Nu = ifield(s, "Nu");
        p.nS = rows(field(s, "S_A")); p.nW = rows(field(s, "W_A"));
        p.A = dbl(field(s, "A")); p.B = dbl(field(s, "B")); p.K = dbl(field(s, "K")); p.Phi = dbl(field(s, "Phi"));
        p.Px = dbl(field(s, "Px")); p.Pc = dbl(field(s, "Pc")); p.F = dbl(field(s, "F")); p.G = dbl(field(s, "G"));
        p.F_bar = dbl(field(s, "F_bar")); p.Psi = dbl(field(s, "Psi")); p.S_A = dbl(field(s, "S_A")); p.S_b = dbl(field(s, "S_b"));
        p.W_A = dbl(field(s, "W_A")); p.W_b = dbl(field(s, "W_b")); p.hs = dbl(field(s, "hs"));
        Handle* h = new Handle();
        int rc = lbmpc_uqmpc_create(&p, &h->ctx);
        if (!rc && ngpu > 1) rc = lbmpc_multi_create(&p, nullptr, ngpu, &h->multi);
        if (rc) { release(h); fail("lbmpc:create", rc); }
        g_live.insert(h);
        if (g_live.size() == 1) mexLock();
        plhs[0] = mxCreateNumericMatrix(1, 1, mxUINT64_CLASS, mxREAL);
        *reinterpret_cast<uint64_t*>(mxGetData(plhs[0])) = reinterpret_cast<uint64_t>(h);
    } else if (cmd == "destroy") {
        need_args(nrhs, 2, "lbmpc_mex('destroy', h)");
        Handle* h = handle(prhs[1]);
        g_live.erase(h);
        release(h);
        if (g_live.empty()) mexUnlock();
    } else if (cmd == "step") {
        // [s0, u, alpha, v, hk, Sb_hat, Wb_hat, status] = lbmpc_mex('step', h, uq, dims, x, alpha_prev, v_prev, w_new)
        need_args(nrhs, 7, "[s0, u, alpha, v, hk, Sb_hat, Wb_hat, status] = lbmpc_mex('step', h, uq, dims, x, alpha_prev, v_prev [, w_new])");
        lbmpc_ctx* c = handle(prhs[1])->ctx;
        const int uq = (int)mxGetScalar(prhs[2]);
        if (uq && nrhs < 8) mexErrMsgIdAndTxt("lbmpc:usage", "'step' with uq = 1 needs w_new");
        const double* dims = dbl(prhs[3]);  // [nx nu nc nS nW]
        const int nx = (int)dims[0], nu = (int)dims[1], nc = (int)dims[2], nS = (int)dims[3], nW = (int)dims[4];
        plhs[0] = mxCreateDoubleMatrix(nx, 1, mxREAL);
        mxArray* u = mxCreateDoubleMatrix(nu, 1, mxREAL); mxArray* al = mxCreateDoubleScalar(0);
        mxArray* v = mxCreateDoubleMatrix(nx, 1, mxREAL); mxArray* hk = mxCreateDoubleMatrix(nc, 1, mxREAL);
        mxArray* Sb = mxCreateDoubleMatrix(nS, 1, mxREAL); mxArray* Wb = mxCreateDoubleMatrix(nW, 1, mxREAL);
        int st = 0;
        const double ap = mxGetScalar(prhs[5]);
        int rc = lbmpc_uqmpc_step(c, 1, uq, dbl(prhs[4]), &ap, dbl(prhs[6]), uq ? dbl(prhs[7]) : nullptr, mxGetDoubles(plhs[0]),
                                  nullptr, mxGetDoubles(u), mxGetDoubles(al), mxGetDoubles(v), mxGetDoubles(hk), mxGetDoubles(Sb),
                                  mxGetDoubles(Wb), &st, nullptr, nullptr, nullptr);
        if (rc < 0) fail("lbmpc:step", rc);
        // numerical statuses do not throw: the reference has none (UQRobustMPC.m:156-158); warn and return the iterate
        if (st) mexWarnMsgIdAndTxt("lbmpc:status", "UQ-RMPC step status %d (1 QP infeasible, 2 LP infeasible, 3 iteration cap)", st);
        mxArray* outs[] = {u, al, v, hk, Sb, Wb, mxCreateDoubleScalar(st)};
        for (int i = 1; i < nlhs && i <= 7; ++i) plhs[i] = outs[i - 1];
    } else if (cmd == "mc_run") {
        // R = lbmpc_mex('mc_run', h, sampler, cfg)  cfg: trials, steps, card_I0, seed, x0, acc_bound, distance_error, trial0
        need_args(nrhs, 4, "R = lbmpc_mex('mc_run', h, sampler, cfg)");
        Handle* h = handle(prhs[1]);
        lbmpc_sampler smp; fill_sampler(prhs[2], &smp);
        const mxArray* s = prhs[3];
        lbmpc_mc_cfg cfg;
        std::memset(&cfg, 0, sizeof cfg);
        cfg.trials = (int64_t)mxGetScalar(field(s, "trials")); cfg.trial0 = (int64_t)mxGetScalar(field(s, "trial0"));
        cfg.steps = ifield(s, "steps"); cfg.card_I0 = ifield(s, "card_I0"); cfg.seed = (uint64_t)mxGetScalar(field(s, "seed"));
        cfg.x0 = dbl(field(s, "x0")); cfg.acc_bound = mxGetScalar(field(s, "acc_bound"));
        cfg.distance_error = mxGetScalar(field(s, "distance_error")); cfg.sampler_max_attempts = 1000;
        const mxArray* tt = mxGetField(s, 0, "traj_trials");   // optional: record State_RM / Control_EV of the first trials
        cfg.traj_trials = tt ? (int32_t)mxGetScalar(tt) : 0;
        if (cfg.traj_trials > cfg.trials) cfg.traj_trials = (int32_t)cfg.trials;
        const char* names[] = {"Infeasible_Index", "status", "alpha_K", "v_K", "alpha_ini", "qp_iters", "State_RM", "Control_EV", "Alpha", "Nominal_RM_State"};
        plhs[0] = mxCreateStructMatrix(1, 1, 10, names);
        mxArray* inf = mxCreateNumericMatrix(cfg.trials, 1, mxINT32_CLASS, mxREAL);
        mxArray* st = mxCreateNumericMatrix(cfg.trials, 1, mxINT32_CLASS, mxREAL);
        mxArray* aK = mxCreateDoubleMatrix(cfg.trials, 1, mxREAL);
        mxArray* vK = mxCreateDoubleMatrix(smp.nx, cfg.trials, mxREAL);
        mxArray* a0 = mxCreateDoubleMatrix(cfg.trials, 1, mxREAL);
        mxArray* it = mxCreateNumericMatrix(cfg.trials, 1, mxINT32_CLASS, mxREAL);
        const size_t R = (size_t)cfg.traj_trials, K = (size_t)cfg.steps;
        // trajectories as nx x (K+1) x R etc. flattened to 2-D (MATLAB reshapes): column block r is trial r
        mxArray* tx = mxCreateDoubleMatrix(smp.nx, (K + 1) * R, mxREAL);
        mxArray* tu = mxCreateDoubleMatrix(smp.nu, K * R, mxREAL);
        mxArray* ta = mxCreateDoubleMatrix(K + 1, R, mxREAL);
        mxArray* ts = mxCreateDoubleMatrix(smp.nx, K * R, mxREAL);
        lbmpc_mc_out out;
        std::memset(&out, 0, sizeof out);
        out.infeasible = (int32_t*)mxGetData(inf); out.status = (int32_t*)mxGetData(st);
        out.alpha_final = mxGetDoubles(aK); out.v_final = mxGetDoubles(vK);
        out.alpha0 = mxGetDoubles(a0); out.qp_iters = (int32_t*)mxGetData(it);
        if (R) { out.traj_x = mxGetDoubles(tx); out.traj_u = mxGetDoubles(tu); out.traj_alpha = mxGetDoubles(ta); out.traj_s0 = mxGetDoubles(ts); }
        int rc = h->multi ? lbmpc_multi_mc_run(h->multi, &smp, &cfg, &out) : lbmpc_mc_run(h->ctx, &smp, &cfg, &out);
        if (rc < 0) fail("lbmpc:mc_run", rc);
        mxArray* vals[] = {inf, st, aK, vK, a0, it, tx, tu, ta, ts};
        for (int i = 0; i < 10; ++i) mxSetField(plhs[0], 0, names[i], vals[i]);
    } else if (cmd == "polytope_support") {
        // [h, z, status] = lbmpc_mex('polytope_support', A, b, C)   A m x n, b m x nb (> 0), C n x nd; h nd x nb
        need_args(nrhs, 4, "[h, z, status] = lbmpc_mex('polytope_support', A, b, C)");
        const int m = rows(prhs[1]), n = cols(prhs[1]);
        const int64_t nb = cols(prhs[2]), nd = cols(prhs[3]);
        if (rows(prhs[2]) != m || rows(prhs[3]) != n) mexErrMsgIdAndTxt("lbmpc:polytope_support", "b must be m x nb and C n x nd");
        plhs[0] = mxCreateDoubleMatrix(nd, nb, mxREAL);
        mxArray* z = mxCreateDoubleMatrix(n, nd * nb, mxREAL);
        mxArray* st = mxCreateNumericMatrix(nd, nb, mxINT32_CLASS, mxREAL);
        int rc = lbmpc_polytope_support(dbl(prhs[1]), m, n, dbl(prhs[2]), nb, dbl(prhs[3]), nd, mxGetDoubles(plhs[0]),
                                        mxGetDoubles(z), (int*)mxGetData(st));
        if (rc < 0) fail("lbmpc:polytope_support", rc);
        if (nlhs > 1) plhs[1] = z;
        if (nlhs > 2) plhs[2] = st;
    } else if (cmd == "polytope_redundant") {
        // [slack, status] = lbmpc_mex('polytope_redundant', A, b)   slack(j) >= 0: row j is implied by the others
        need_args(nrhs, 3, "[slack, status] = lbmpc_mex('polytope_redundant', A, b)");
        const int m = rows(prhs[1]), n = cols(prhs[1]);
        if (rows(prhs[2]) * cols(prhs[2]) != m) mexErrMsgIdAndTxt("lbmpc:polytope_redundant", "b must have one entry per row of A");
        plhs[0] = mxCreateDoubleMatrix(m, 1, mxREAL);
        mxArray* st = mxCreateNumericMatrix(m, 1, mxINT32_CLASS, mxREAL);
        int rc = lbmpc_polytope_redundant(dbl(prhs[1]), m, n, dbl(prhs[2]), mxGetDoubles(plhs[0]), (int*)mxGetData(st));
        if (rc < 0) fail("lbmpc:polytope_redundant", rc);
        if (nlhs > 1) plhs[1] = st;
    } else if (cmd == "scmpc_mc_run") {
        // R = lbmpc_mex('scmpc_mc_run', p, sampler, cfg)   p: N, N_SC, A, B, Q, R, F, G (ScenarioMPC.m:30-48); cfg as 'mc_run'
        need_args(nrhs, 4, "R = lbmpc_mex('scmpc_mc_run', p, sampler, cfg)");
        const mxArray* ps = prhs[1];
        lbmpc_scmpc_params p;
        p.A = dbl(field(ps, "A")); p.B = dbl(field(ps, "B")); p.Q = dbl(field(ps, "Q")); p.R = dbl(field(ps, "R"));
        p.F = dbl(field(ps, "F")); p.G = dbl(field(ps, "G"));
        p.nx = rows(field(ps, "A")); p.nu = cols(field(ps, "B")); p.nc = rows(field(ps, "F"));
        p.N = ifield(ps, "N"); p.N_SC = ifield(ps, "N_SC");
        lbmpc_sampler smp; fill_sampler(prhs[2], &smp);
        const mxArray* s = prhs[3];
        lbmpc_mc_cfg cfg;
        std::memset(&cfg, 0, sizeof cfg);
        cfg.trials = (int64_t)mxGetScalar(field(s, "trials")); cfg.trial0 = (int64_t)mxGetScalar(field(s, "trial0"));
        cfg.steps = ifield(s, "steps"); cfg.card_I0 = 1; cfg.seed = (uint64_t)mxGetScalar(field(s, "seed"));
        cfg.x0 = dbl(field(s, "x0")); cfg.acc_bound = mxGetScalar(field(s, "acc_bound"));
        cfg.distance_error = mxGetScalar(field(s, "distance_error")); cfg.sampler_max_attempts = 1000;
        const mxArray* tt = mxGetField(s, 0, "traj_trials");
        cfg.traj_trials = tt ? (int32_t)mxGetScalar(tt) : 0;
        if (cfg.traj_trials > cfg.trials) cfg.traj_trials = (int32_t)cfg.trials;
        const char* names[] = {"Infeasible_Index", "status", "qp_iters", "x_K", "State_RM_SCMPC", "Control_EV_SCMPC"};
        plhs[0] = mxCreateStructMatrix(1, 1, 6, names);
        const size_t R = (size_t)cfg.traj_trials, K = (size_t)cfg.steps;
        mxArray* inf = mxCreateNumericMatrix(cfg.trials, 1, mxINT32_CLASS, mxREAL);
        mxArray* st = mxCreateNumericMatrix(cfg.trials, 1, mxINT32_CLASS, mxREAL);
        mxArray* it = mxCreateNumericMatrix(cfg.trials, 1, mxINT32_CLASS, mxREAL);
        mxArray* xK = mxCreateDoubleMatrix(p.nx, cfg.trials, mxREAL);
        mxArray* tx = mxCreateDoubleMatrix(p.nx, (K + 1) * R, mxREAL);
        mxArray* tu = mxCreateDoubleMatrix(p.nu, K * R, mxREAL);
        lbmpc_scmpc_out out;
        std::memset(&out, 0, sizeof out);
        out.infeasible = (int32_t*)mxGetData(inf); out.status = (int32_t*)mxGetData(st); out.qp_iters = (int32_t*)mxGetData(it);
        out.x_final = mxGetDoubles(xK);
        if (R) { out.traj_x = mxGetDoubles(tx); out.traj_u = mxGetDoubles(tu); }
        int rc = lbmpc_scmpc_mc_run(&p, &smp, &cfg, &out);
        if (rc < 0) fail("lbmpc:scmpc_mc_run", rc);
        mxArray* vals[] = {inf, st, it, xK, tx, tu};
        for (int i = 0; i < 6; ++i) mxSetField(plhs[0], 0, names[i], vals[i]);
    } else {
        mexErrMsgIdAndTxt("lbmpc:usage", "unknown command %s", cmdbuf);
    }
}
