// Micro-benchmark: fp64 DPP (row_newbcast) FMA rate and semantics, LDS read patterns.  Diagnostic only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int K> __device__ __forceinline__ double fmac_bc(double acc, double coef, double x) {
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(coef), "v"(x), "n"(K));
    return acc;
}

__global__ void sem(const double* a, const double* q, double* o) {
    double c = a[threadIdx.x], x = q[threadIdx.x];
    double acc = 1.0;
    acc = fmac_bc<3>(acc, c, x);   // 1 + a[row*16+3]*q[lane]
    acc = fmac_bc<11>(acc, c, x);
    o[threadIdx.x] = acc;
}

// mode 0: plain v_fma_f64 ; mode 1: v_fmac_f64_dpp
template <int MODE>
__global__ void rate(double* o, long long* cyc, int iters, double seed) {
    double c = seed + threadIdx.x, x = 1.0 + 1e-9 * threadIdx.x;
    double a0 = 0, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
            asm volatile("v_fma_f64 %0, %8, %9, %0\n v_fma_f64 %1, %8, %9, %1\n v_fma_f64 %2, %8, %9, %2\n v_fma_f64 %3, %8, %9, %3\n"
                         "v_fma_f64 %4, %8, %9, %4\n v_fma_f64 %5, %8, %9, %5\n v_fma_f64 %6, %8, %9, %6\n v_fma_f64 %7, %8, %9, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(x));
        } else {
            asm volatile("v_fmac_f64_dpp %0, %8, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %1, %8, %9 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
                         "v_fmac_f64_dpp %2, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %3, %8, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
                         "v_fmac_f64_dpp %4, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %5, %8, %9 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
                         "v_fmac_f64_dpp %6, %8, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %7, %8, %9 row_newbcast:8 row_mask:0xf bank_mask:0xf\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(x));
        }
    }
    long long t1 = clock64();
    o[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x % 64 == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = t1 - t0;
}

// LDS read patterns: mode 0 uniform b128 broadcast; 1: b64 lane%16 distinct (x4 replicated); 2: b64 64 distinct; 3: b128 64 distinct; 4: uniform b64
template <int MODE>
__global__ void ldsrate(double* o, long long* cyc, int iters) {
    __shared__ double sm[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) sm[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    double acc = 0.0;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        const int base = (i * 64) & 4095;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0) { double2 v = *reinterpret_cast<const double2*>(&sm[base + 2 * u]); acc += v.x + v.y; }
            if (MODE == 1) { acc += sm[base + 16 * u + (lane & 15)]; }
            if (MODE == 2) { acc += sm[base + 64 * u + lane]; }
            if (MODE == 3) { double2 v = *reinterpret_cast<const double2*>(&sm[base + 128 * u + 2 * lane]); acc += v.x + v.y; }
            if (MODE == 4) { acc += sm[base + u]; }
        }
    }
    long long t1 = clock64();
    o[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (lane == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = t1 - t0;
}

template <class F>
static double run(F launch, long long* dcyc, int nw) {
    launch();
    hipDeviceSynchronize();
    std::vector<long long> h(nw);
    hipMemcpy(h.data(), dcyc, nw * sizeof(long long), hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += v;
    return s / nw;
}

int main() {
    double *a, *q, *o; long long* cyc;
    CK(hipMalloc(&a, 64 * 8)); CK(hipMalloc(&q, 64 * 8)); CK(hipMalloc(&o, 1 << 24)); CK(hipMalloc(&cyc, 1 << 20));
    double ha[64], hq[64], ho[64];
    for (int i = 0; i < 64; ++i) { ha[i] = 100 + i; hq[i] = 0.5 + i; }
    CK(hipMemcpy(a, ha, sizeof ha, hipMemcpyHostToDevice)); CK(hipMemcpy(q, hq, sizeof hq, hipMemcpyHostToDevice));
    sem<<<1, 64>>>(a, q, o);
    CK(hipMemcpy(ho, o, sizeof ho, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < 64; ++i) {
        const int r = i / 16;
        double ref = fma(ha[r * 16 + 11], hq[i], fma(ha[r * 16 + 3], hq[i], 1.0));
        if (ref != ho[i]) { if (bad < 4) printf("sem mismatch lane %d got %g want %g\n", i, ho[i], ref); bad++; }
    }
    printf("semantics: %s\n", bad ? "MISMATCH" : "ok (src0 broadcast from lane K of each 16-lane row)");
    const int iters = 20000;
    for (int wpb : {64, 256, 512, 1024}) {  // 1, 4(1/SIMD), 8 (2/SIMD), 16 waves per block, one block per CU (256 blocks)
        const int nw = 256 * wpb / 64;
        double c0 = run([&] { rate<0><<<256, wpb>>>(o, cyc, iters, 1.0); }, cyc, nw);
        double c1 = run([&] { rate<1><<<256, wpb>>>(o, cyc, iters, 1.0); }, cyc, nw);
        printf("waves/CU %2d: fma %.2f cyc/instr/wave   fmac_dpp %.2f cyc/instr/wave\n", wpb / 64, c0 / (8.0 * iters), c1 / (8.0 * iters));
    }
    const char* names[5] = {"b128 uniform bcast", "b64 16-distinct x4", "b64 64-distinct", "b128 64-distinct", "b64 uniform bcast"};
    for (int wpb : {64, 512}) {
        const int nw = 256 * wpb / 64;
        double c[5];
        c[0] = run([&] { ldsrate<0><<<256, wpb>>>(o, cyc, 4000); }, cyc, nw);
        c[1] = run([&] { ldsrate<1><<<256, wpb>>>(o, cyc, 4000); }, cyc, nw);
        c[2] = run([&] { ldsrate<2><<<256, wpb>>>(o, cyc, 4000); }, cyc, nw);
        c[3] = run([&] { ldsrate<3><<<256, wpb>>>(o, cyc, 4000); }, cyc, nw);
        c[4] = run([&] { ldsrate<4><<<256, wpb>>>(o, cyc, 4000); }, cyc, nw);
        for (int m = 0; m < 5; ++m) printf("waves/CU %2d: lds %-20s %.2f cyc per read instr per wave (incl. add)\n", wpb / 64, names[m], c[m] / (8.0 * 4000));
    }
    return 0;
}
