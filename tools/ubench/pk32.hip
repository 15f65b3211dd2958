// Micro-benchmark: v_pk_fma_f32 / v_fma_f32 / v_fma_f64 issue rate, independent vs dependent chains.  Diagnostic only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
// MODE 0: pk_fma 8 independent ; 1: pk_fma dependent chain ; 2: fma_f32 8 indep ; 3: fma_f64 dependent ; 4: pk_fma 2 indep ; 5: pk_fma 4 indep
template <int MODE>
__global__ void rate(float* o, long long* cyc, int iters, float seed) {
    f2 c = {seed + threadIdx.x, seed}, x = {1.0f + 1e-6f * threadIdx.x, 0.5f};
    f2 a0 = {0, 0}, a1 = {1, 1}, a2 = {2, 2}, a3 = {3, 3}, a4 = {4, 4}, a5 = {5, 5}, a6 = {6, 6}, a7 = {7, 7};
    double d0 = seed, dc = seed * 0.5, dx = 1.0 + 1e-9 * threadIdx.x;
    float s0 = 0, s1 = 1, s2 = 2, s3 = 3, s4 = 4, s5 = 5, s6 = 6, s7 = 7, sc = seed, sx = 1.0f;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0)
            asm volatile("v_pk_fma_f32 %0, %8, %9, %0\n v_pk_fma_f32 %1, %8, %9, %1\n v_pk_fma_f32 %2, %8, %9, %2\n v_pk_fma_f32 %3, %8, %9, %3\n"
                         "v_pk_fma_f32 %4, %8, %9, %4\n v_pk_fma_f32 %5, %8, %9, %5\n v_pk_fma_f32 %6, %8, %9, %6\n v_pk_fma_f32 %7, %8, %9, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(x));
        if (MODE == 1)
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0\n s_nop 0\n v_pk_fma_f32 %0, %1, %2, %0\n s_nop 0\n v_pk_fma_f32 %0, %1, %2, %0\n s_nop 0\n v_pk_fma_f32 %0, %1, %2, %0\n s_nop 0\n"
                         "v_pk_fma_f32 %0, %1, %2, %0\n s_nop 0\n v_pk_fma_f32 %0, %1, %2, %0\n s_nop 0\n v_pk_fma_f32 %0, %1, %2, %0\n s_nop 0\n v_pk_fma_f32 %0, %1, %2, %0\n s_nop 0\n"
                         : "+v"(a0) : "v"(c), "v"(x));
        if (MODE == 2)
            asm volatile("v_fma_f32 %0, %8, %9, %0\n v_fma_f32 %1, %8, %9, %1\n v_fma_f32 %2, %8, %9, %2\n v_fma_f32 %3, %8, %9, %3\n"
                         "v_fma_f32 %4, %8, %9, %4\n v_fma_f32 %5, %8, %9, %5\n v_fma_f32 %6, %8, %9, %6\n v_fma_f32 %7, %8, %9, %7\n"
                         : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "+v"(s4), "+v"(s5), "+v"(s6), "+v"(s7) : "v"(sc), "v"(sx));
        if (MODE == 3)
            asm volatile("v_fma_f64 %0, %1, %2, %0\n v_fma_f64 %0, %1, %2, %0\n v_fma_f64 %0, %1, %2, %0\n v_fma_f64 %0, %1, %2, %0\n"
                         "v_fma_f64 %0, %1, %2, %0\n v_fma_f64 %0, %1, %2, %0\n v_fma_f64 %0, %1, %2, %0\n v_fma_f64 %0, %1, %2, %0\n"
                         : "+v"(d0) : "v"(dc), "v"(dx));
        if (MODE == 4)
            asm volatile("v_pk_fma_f32 %0, %2, %3, %0\n v_pk_fma_f32 %1, %2, %3, %1\n v_pk_fma_f32 %0, %2, %3, %0\n v_pk_fma_f32 %1, %2, %3, %1\n"
                         "v_pk_fma_f32 %0, %2, %3, %0\n v_pk_fma_f32 %1, %2, %3, %1\n v_pk_fma_f32 %0, %2, %3, %0\n v_pk_fma_f32 %1, %2, %3, %1\n"
                         : "+v"(a0), "+v"(a1) : "v"(c), "v"(x));
        if (MODE == 5)
            asm volatile("v_pk_fma_f32 %0, %4, %5, %0\n v_pk_fma_f32 %1, %4, %5, %1\n v_pk_fma_f32 %2, %4, %5, %2\n v_pk_fma_f32 %3, %4, %5, %3\n"
                         "v_pk_fma_f32 %0, %4, %5, %0\n v_pk_fma_f32 %1, %4, %5, %1\n v_pk_fma_f32 %2, %4, %5, %2\n v_pk_fma_f32 %3, %4, %5, %3\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c), "v"(x));
    }
    long long t1 = clock64();
    f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    o[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y + (float)d0 + s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7;
    if (threadIdx.x % 64 == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = t1 - t0;
}
template <class F> static double run(F launch, long long* dcyc, int nw) {
    launch(); hipDeviceSynchronize();
    std::vector<long long> h(nw);
    hipMemcpy(h.data(), dcyc, nw * sizeof(long long), hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += v;
    return s / nw;
}
int main() {
    float* o; long long* cyc;
    hipMalloc(&o, 1 << 24); hipMalloc(&cyc, 1 << 20);
    const int iters = 20000;
    const char* names[6] = {"pk_fma_f32 x8 indep", "pk_fma_f32 dependent", "fma_f32 x8 indep", "fma_f64 dependent", "pk_fma_f32 x2 indep", "pk_fma_f32 x4 indep"};
    for (int wpb : {256, 512, 1024}) {
        const int nw = 256 * wpb / 64;
        double c[6];
        c[0] = run([&] { rate<0><<<256, wpb>>>(o, cyc, iters, 1.0f); }, cyc, nw);
        c[1] = run([&] { rate<1><<<256, wpb>>>(o, cyc, iters, 1.0f); }, cyc, nw);
        c[2] = run([&] { rate<2><<<256, wpb>>>(o, cyc, iters, 1.0f); }, cyc, nw);
        c[3] = run([&] { rate<3><<<256, wpb>>>(o, cyc, iters, 1.0f); }, cyc, nw);
        c[4] = run([&] { rate<4><<<256, wpb>>>(o, cyc, iters, 1.0f); }, cyc, nw);
        c[5] = run([&] { rate<5><<<256, wpb>>>(o, cyc, iters, 1.0f); }, cyc, nw);
        for (int m = 0; m < 6; ++m) printf("waves/SIMD %d: %-22s %.2f cyc/instr/wave\n", wpb / 256, names[m], c[m] / (8.0 * iters));
    }
    return 0;
}
