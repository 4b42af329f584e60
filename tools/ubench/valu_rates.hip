// Micro-benchmark: issue cost (cycles per wave-instruction per SIMD) of the VALU ops the cUDE kernels use.
// One wave per SIMD (grid = 256 CUs x 4 waves of 64), independent chains (8 accumulators) => throughput.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP 512
#define CHK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s\n",hipGetErrorString(e)); return 1;}}while(0)

template <int OP>
__global__ __launch_bounds__(64) void k(double* out, long long* cyc, double seed, int iters) {
    double a[8];
    for (int i = 0; i < 8; i++) a[i] = seed + threadIdx.x * 1e-3 + i;
    int n = (int)seed;
    long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < REP / 8; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (OP == 0) a[i] = fma(a[i], 1.0000001, 0.5);
                if (OP == 1) a[i] = __builtin_amdgcn_rcp(a[i]);
                if (OP == 2) a[i] = ldexp(a[i], n);
                if (OP == 3) a[i] = rint(fma(a[i], 1.0000001, 0.3));        // fma + rndne
                if (OP == 4) a[i] = (double)(int)fma(a[i], 1.0000001, 0.3); // fma + cvt_i32_f64 + cvt_f64_i32
                if (OP == 5) a[i] = fmin(fma(a[i], 1.0000001, 0.3), 40.0);  // fma + min
                if (OP == 6) a[i] = a[i] + 1.5;
                if (OP == 7) a[i] = a[i] * 1.0000001;
                if (OP == 8) a[i] = copysign(fma(a[i], 1.0000001, 0.3), -a[(i + 1) & 7]);   // fma + bfi
                if (OP == 9) { float f = (float)a[i]; f = fmaf(f, 1.0001f, 0.5f); a[i] = f; }   // cvt+fma32+cvt
                if (OP == 10) a[i] = __shfl_xor(a[i], 1, 64);
                if (OP == 11) a[i] = __builtin_amdgcn_readfirstlane((int)n + i) + a[i];
            }
        }
    }
    long long t1 = clock64();
    double s = 0; for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP> int run(const char* name, double* out, long long* cyc, int waves_per_simd) {
    int nblk = 256 * 4 * waves_per_simd, iters = 4000;
    hipLaunchKernelGGL(k<OP>, dim3(nblk), dim3(64), 0, 0, out, cyc, 1.0, iters);
    CHK(hipDeviceSynchronize());
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); hipLaunchKernelGGL(k<OP>, dim3(nblk), dim3(64), 0, 0, out, cyc, 1.0, iters); hipEventRecord(e1);
    CHK(hipDeviceSynchronize());
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(nblk); hipMemcpy(h.data(), cyc, nblk * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= nblk;
    double ninst = (double)REP * iters;
    printf("%-28s waves/SIMD=%d  %.2f clk64/inst/wave  wall %.3f ms => %.3f ns per wave-inst per SIMD = %.2f cyc@2.4GHz\n", name, waves_per_simd,
           avg / ninst, ms, ms * 1e6 / ninst / waves_per_simd, ms * 1e6 / ninst / waves_per_simd * 2.4);
    return 0;
}

int main() {
    double* out; long long* cyc;
    CHK(hipMalloc(&out, 8 * 64 * 8192)); CHK(hipMalloc(&cyc, 8 * 8192));
    for (int w : {1, 2, 4}) {
        run<0>("v_fma_f64", out, cyc, w); run<1>("v_rcp_f64", out, cyc, w); run<2>("v_ldexp_f64", out, cyc, w);
        run<3>("v_rndne_f64", out, cyc, w); run<4>("cvt_i32_f64+cvt_f64_i32", out, cyc, w); run<5>("v_min_f64", out, cyc, w);
        run<6>("v_add_f64", out, cyc, w); run<7>("v_mul_f64", out, cyc, w); run<8>("copysign(bfi)", out, cyc, w);
        run<9>("cvt+fma_f32+cvt", out, cyc, w); run<10>("shfl_xor f64", out, cyc, w); run<11>("readfirstlane+cvt+add", out, cyc, w);
    }
    return 0;
}
