// Micro-benchmark: v_mfma_f64_16x16x4_f64 on gfx950 -- (1) is D = C + sum_k A[i][k] B[k][j] the fma chain in
// ascending k (one rounding per term, C as the initial value)?  That is the arithmetic contract of the pivot scans
// (qp_lane.hpp: acc = alpha b; acc = fma(a_k, e_k, acc), k ascending), so a scan through the matrix pipe would be
// bit-identical only in that case.  (2) issue interval of the instruction at one and two waves per SIMD, alone and with
// the v_mul_f64 / v_min_f64 work of a scan block interleaved (do the two pipes overlap?).  Diagnostic only.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off mfma64.hip -o mfma64 && ./mfma64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cmath>
#include <vector>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef double v4d __attribute__((ext_vector_type(4)));

// one 16x16x4 product per wave: A[16][4], B[4][16], C[16][16] row-major in memory
__global__ void exact_kernel(const double* A, const double* B, const double* C, double* D, int ntiles) {
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (tile >= ntiles) return;
    const double* a = A + (size_t)tile * 64; const double* b = B + (size_t)tile * 64;
    const double* c = C + (size_t)tile * 256; double* d = D + (size_t)tile * 256;
    const int i = lane & 15, k = lane >> 4;
    v4d acc;
    for (int r = 0; r < 4; ++r) acc[r] = c[(k + 4 * r) * 16 + i];   // row = (lane >> 4) + 4 r, col = lane & 15
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i * 4 + k], b[k * 16 + i], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d[(k + 4 * r) * 16 + i] = acc[r];
}

template <int MODE>
__global__ void rate_kernel(double* o, long long* cyc, int iters) {
    const int lane = threadIdx.x & 63;
    double a = 1.0 + 1e-9 * lane, b = 1.0 - 1e-9 * lane, al = 0.5 + 1e-3 * lane;
    v4d d0 = {0, 0, 0, 0}, d1 = {1, 1, 1, 1}, d2 = {2, 2, 2, 2}, d3 = {3, 3, 3, 3};
    v4d m = {1e300, 1e300, 1e300, 1e300};
    v4d bb = {1.0 + lane, 2.0 + lane, 3.0 + lane, 4.0 + lane};
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {          // four independent accumulators, MFMA only
            d0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, d1, 0, 0, 0);
            d2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, d2, 0, 0, 0);
            d3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, d3, 0, 0, 0);
        } else if (MODE == 1) {   // a scan block: C = alpha * b (4 mul), MFMA, 4 min on the PREVIOUS block's result
            v4d c0, c1;
            for (int r = 0; r < 4; ++r) c0[r] = al * bb[r];
            v4d n0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            for (int r = 0; r < 4; ++r) asm("v_min_f64 %0, %1, %2" : "=v"(m[r]) : "v"(d0[r]), "v"(m[r]));
            for (int r = 0; r < 4; ++r) c1[r] = al * bb[3 - r];
            v4d n1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
            for (int r = 0; r < 4; ++r) asm("v_min_f64 %0, %1, %2" : "=v"(m[r]) : "v"(d1[r]), "v"(m[r]));
            for (int r = 0; r < 4; ++r) c0[r] = al * bb[(r + 1) & 3];
            v4d n2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            for (int r = 0; r < 4; ++r) asm("v_min_f64 %0, %1, %2" : "=v"(m[r]) : "v"(d2[r]), "v"(m[r]));
            for (int r = 0; r < 4; ++r) c1[r] = al * bb[(r + 2) & 3];
            v4d n3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
            for (int r = 0; r < 4; ++r) asm("v_min_f64 %0, %1, %2" : "=v"(m[r]) : "v"(d3[r]), "v"(m[r]));
            d0 = n0; d1 = n1; d2 = n2; d3 = n3;
            al += 1e-9;
        } else if (MODE == 2) {   // the same 256 row-QP pairs on the vector pipe: 4 x (mul + 4 fma + min) per 64 pairs
            for (int q = 0; q < 16; ++q) {
                double acc = al * bb[q & 3];
                acc = fma(a, b, acc); acc = fma(b, a, acc); acc = fma(a, al, acc); acc = fma(b, al, acc);
                asm("v_min_f64 %0, %1, %2" : "=v"(m[q & 3]) : "v"(acc), "v"(m[q & 3]));
            }
            al += 1e-9;
        }
    }
    long long t1 = clock64();
    double s = 0;
    for (int r = 0; r < 4; ++r) s += d0[r] + d1[r] + d2[r] + d3[r] + m[r];
    o[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = t1 - t0;
}

static double chain(const double* a, const double* b, double c, int i, int j, bool asc) {
    double acc = c;
    if (asc) for (int k = 0; k < 4; ++k) acc = fma(a[i * 4 + k], b[k * 16 + j], acc);
    else for (int k = 3; k >= 0; --k) acc = fma(a[i * 4 + k], b[k * 16 + j], acc);
    return acc;
}

int main() {
    // ---- exactness ----
    const int nt = 4096;
    std::vector<double> A((size_t)nt * 64), B((size_t)nt * 64), C((size_t)nt * 256), D((size_t)nt * 256);
    std::mt19937_64 g(12345);
    std::uniform_real_distribution<double> U(-1.0, 1.0);
    for (int t = 0; t < nt; ++t) {
        const int kind = t % 4;
        for (int x = 0; x < 64; ++x) {
            double sa = kind == 1 ? std::ldexp(1.0, (int)(g() % 40) - 20) : 1.0;
            A[(size_t)t * 64 + x] = U(g) * sa; B[(size_t)t * 64 + x] = U(g) * (kind == 2 ? 1e-8 : 1.0);
        }
        for (int x = 0; x < 256; ++x) C[(size_t)t * 256 + x] = kind == 3 ? 0.0 : U(g);
        if (kind == 2)   // cancellation: C close to minus the exact sum
            for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
                double s = 0; for (int k = 0; k < 4; ++k) s += A[(size_t)t * 64 + i * 4 + k] * B[(size_t)t * 64 + k * 16 + j];
                C[(size_t)t * 256 + i * 16 + j] = -s * (1.0 + 1e-12 * U(g));
            }
    }
    double *dA, *dB, *dC, *dD;
    CK(hipMalloc(&dA, A.size() * 8)); CK(hipMalloc(&dB, B.size() * 8)); CK(hipMalloc(&dC, C.size() * 8)); CK(hipMalloc(&dD, D.size() * 8));
    CK(hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dC, C.data(), C.size() * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(exact_kernel, dim3(nt / 4), dim3(256), 0, 0, dA, dB, dC, dD, nt);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(D.data(), dD, D.size() * 8, hipMemcpyDeviceToHost));
    long bad_asc = 0, bad_desc = 0, n = 0;
    for (int t = 0; t < nt; ++t)
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
            const double got = D[(size_t)t * 256 + i * 16 + j], c = C[(size_t)t * 256 + i * 16 + j];
            const double ea = chain(&A[(size_t)t * 64], &B[(size_t)t * 64], c, i, j, true);
            const double ed = chain(&A[(size_t)t * 64], &B[(size_t)t * 64], c, i, j, false);
            bad_asc += std::memcmp(&got, &ea, 8) != 0; bad_desc += std::memcmp(&got, &ed, 8) != 0; ++n;
        }
    printf("exactness over %ld outputs: mismatches vs ascending fma chain %ld, vs descending %ld\n", n, bad_asc, bad_desc);

    // ---- rates ----
    const int iters = 2000;
    const char* names[] = {"4 mfma_f64_16x16x4 (independent)", "4 scan blocks: 4 mul + mfma + 4 min each", "vector form of 256 pairs x4: 16 x (mul + 4 fma + min)"};
    for (int wps : {1, 2, 4}) {
        const int blocks = 256 * wps, threads = 256, nw = blocks * threads / 64;
        double* o; long long* cyc;
        CK(hipMalloc(&o, (size_t)blocks * threads * 8)); CK(hipMalloc(&cyc, (size_t)nw * 8));
        for (int mode = 0; mode < 3; ++mode) {
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0));
                if (mode == 0) hipLaunchKernelGGL(rate_kernel<0>, dim3(blocks), dim3(threads), 0, 0, o, cyc, iters);
                if (mode == 1) hipLaunchKernelGGL(rate_kernel<1>, dim3(blocks), dim3(threads), 0, 0, o, cyc, iters);
                if (mode == 2) hipLaunchKernelGGL(rate_kernel<2>, dim3(blocks), dim3(threads), 0, 0, o, cyc, iters);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
            }
            std::vector<long long> h(nw);
            CK(hipMemcpy(h.data(), cyc, (size_t)nw * 8, hipMemcpyDeviceToHost));
            double mean = 0; for (long long c : h) mean += (double)c; mean /= nw;
            // clock64 ticks at 100 MHz on this part: report both the tick count and the time-derived SIMD cycles at 2.4 GHz
            const double simd_cycles = ms * 1e-3 * 2.4e9 / (iters * 4.0 * wps);   // SIMD cycles per block of 256 pairs (2.4 GHz assumed)
            printf("waves/SIMD %d  %-58s %8.3f ms  %7.1f SIMD cycles per 256-pair block  ticks/wave %.0f\n", wps, names[mode], ms, simd_cycles, mean);
        }
        CK(hipFree(o)); CK(hipFree(cyc));
    }
    return 0;
}
