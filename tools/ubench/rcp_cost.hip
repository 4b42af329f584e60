// Cost of one v_rcp_f64 (+ cubic Newton step) embedded in a stream of independent fp64 FMAs, 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s\n",hipGetErrorString(e)); return 1;}}while(0)
template <int MODE>
__global__ __launch_bounds__(64) void k(double* out, double seed, int iters) {
    double a[8], q = seed + 2.0 + threadIdx.x * 1e-3;
    for (int i = 0; i < 8; i++) a[i] = seed + threadIdx.x * 1e-3 + i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 12; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) a[i] = fma(a[i], 0.999999, 1e-7);
        }
        if (MODE == 1) {            // hardware seed + cubic Newton
            double r0 = __builtin_amdgcn_rcp(q);
            double e = fma(-q, r0, 1.0); double t = fma(e, e, e); q = fma(r0, t, r0) + 1.5;
        } else if (MODE == 2) {     // integer seed + three cubic Newton steps
            double r0 = __longlong_as_double(0x7FDE6238502484BAll - __double_as_longlong(q));
#pragma unroll
            for (int s = 0; s < 3; s++) { double e = fma(-q, r0, 1.0); double t = fma(e, e, e); r0 = fma(r0, t, r0); }
            q = r0 + 1.5;
        } else if (MODE == 3) {     // 4 extra FMAs instead (same instruction count as MODE 1 without the rcp)
            double e = fma(-q, q, 1.0); double t = fma(e, e, e); q = fma(q, t, q) * 1e-3 + 1.5;
        }
    }
    double s = q; for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int MODE> float run(double* out, const char* name) {
    int nblk = 256 * 4 * 2, iters = 20000;
    hipLaunchKernelGGL((k<MODE>), dim3(nblk), dim3(64), 0, 0, out, 1e-3, 10);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); hipLaunchKernelGGL((k<MODE>), dim3(nblk), dim3(64), 0, 0, out, 1e-3, iters); hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s %.3f ms  => %.2f ns per loop body per wave\n", name, ms, ms * 1e6 / iters);
    return ms;
}
int main() {
    double* out; CHK(hipMalloc(&out, 8 * 64 * 4096));
    for (int rep = 0; rep < 2; rep++) {
        run<0>(out, "96 FMA");
        run<1>(out, "96 FMA + rcp + newton(3) + add");
        run<2>(out, "96 FMA + int seed + newton(9) + add");
        run<3>(out, "96 FMA + 4 FMA + mul-add");
    }
    return 0;
}
