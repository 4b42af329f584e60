// fp64 FMA issue/latency on gfx950: NC independent dependent-chains per wave, W waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s\n",hipGetErrorString(e)); return 1;}}while(0)
template <int NC, bool SGPRC>
__global__ __launch_bounds__(64) void k(double* out, double seed, double c0, double c1, int iters) {
    double a[NC];
    for (int i = 0; i < NC; i++) a[i] = seed + threadIdx.x * 1e-3 + i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 256 / NC; r++) {
#pragma unroll
            for (int i = 0; i < NC; i++) a[i] = SGPRC ? fma(a[i], a[i], c0) : fma(a[i], 1.0000001, a[i]);
        }
    }
    double s = 0; for (int i = 0; i < NC; i++) s += a[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int NC, bool SG> int run(double* out, int w) {
    int nblk = 256 * 4 * w, iters = 4000;
    hipLaunchKernelGGL((k<NC, SG>), dim3(nblk), dim3(64), 0, 0, out, 1e-3, 0.25, 0.5, 10);
    CHK(hipDeviceSynchronize());
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); hipLaunchKernelGGL((k<NC, SG>), dim3(nblk), dim3(64), 0, 0, out, 1e-3, 0.25, 0.5, iters); hipEventRecord(e1);
    CHK(hipDeviceSynchronize());
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double ninst = 256.0 * iters;
    printf("chains=%d sgpr_const=%d waves/SIMD=%d : %.3f ns per wave-inst per SIMD, %.3f ns per inst per wave\n", NC, (int)SG, w,
           ms * 1e6 / ninst / w, ms * 1e6 / ninst);
    return 0;
}
int main() {
    double* out; CHK(hipMalloc(&out, 8 * 64 * 8192));
    for (int w : {1, 2, 3, 4}) { run<1, true>(out, w); run<2, true>(out, w); run<4, true>(out, w); run<8, true>(out, w); }
    run<1, false>(out, 2); run<8, false>(out, 2);
    return 0;
}
