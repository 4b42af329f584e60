// What costs a wave (and a pair of co-resident waves) its fp64 VALU issue rate on gfx950: streams of independent
// v_fma_f64 chains mixed with the other instruction kinds the gradient kernels contain.  1 and 2 waves per SIMD.
//   A  fma only (4 chains)                    B  + 2 s_mov_b32 per 8 fma            C  + s_load_dwordx8 and an immediate wait per 48 fma
//   D  the same load issued 24 fma earlier     E  + ds_read_b64 and an immediate wait per 48 fma   F  + a taken branch per 24 fma
//   G  fma with ONE dependent chain            H  B + C + E + F together
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s\n",hipGetErrorString(e)); return 1;}}while(0)

#define FMA4 "v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n"
#define FMA8 FMA4 FMA4
#define FMA24 FMA8 FMA8 FMA8
#define SMOV2 "s_mov_b32 s20, 0x3ff00000\n s_mov_b32 s21, 0x12345678\n"
#define FMA1x4 "v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %0, %0, %4, %5\n"

template <int MODE>
__global__ __launch_bounds__(64) void k(double* out, const double* tab, double m, double c, int iters) {
    __shared__ double lds[64];
    lds[threadIdx.x] = 1.0 + threadIdx.x * 1e-9;
    __syncthreads();
    double a0 = 1.0 + threadIdx.x * 1e-6, a1 = a0 + 1e-3, a2 = a0 + 2e-3, a3 = a0 + 3e-3;
    double t = 0.0;
    const double* lp = &lds[threadIdx.x];
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) asm volatile(FMA24 FMA24 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(m), "v"(c));
        if (MODE == 1)
            asm volatile(FMA8 SMOV2 FMA8 SMOV2 FMA8 SMOV2 FMA8 SMOV2 FMA8 SMOV2 FMA8 SMOV2
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(m), "v"(c) : "s20", "s21");
        if (MODE == 2)
            asm volatile("s_load_dwordx8 s[20:27], %6, 0x0\n s_waitcnt lgkmcnt(0)\n" FMA24 FMA24
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(m), "v"(c), "s"(tab)
                         : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
        if (MODE == 3)
            asm volatile("s_load_dwordx8 s[20:27], %6, 0x0\n" FMA24 "s_waitcnt lgkmcnt(0)\n" FMA24
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(m), "v"(c), "s"(tab)
                         : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
        if (MODE == 4)
            asm volatile("ds_read_b64 %6, %7\n s_waitcnt lgkmcnt(0)\n" FMA24 FMA24
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(m), "v"(c), "v"(t), "v"((unsigned)(size_t)lp));
        if (MODE == 5)
            asm volatile(FMA24 "s_branch 1f\n s_nop 0\n s_nop 0\n 1:\n" FMA24 "s_branch 2f\n s_nop 0\n s_nop 0\n 2:\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(m), "v"(c));
        if (MODE == 6)
            asm volatile(FMA1x4 FMA1x4 FMA1x4 FMA1x4 FMA1x4 FMA1x4 FMA1x4 FMA1x4 FMA1x4 FMA1x4 FMA1x4 FMA1x4
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(m), "v"(c));
        if (MODE == 7)
            asm volatile("s_load_dwordx8 s[20:27], %6, 0x0\n s_waitcnt lgkmcnt(0)\n" FMA8 SMOV2 FMA8 SMOV2 FMA8 SMOV2
                         "s_branch 1f\n s_nop 0\n 1:\n ds_read_b64 %7, %8\n s_waitcnt lgkmcnt(0)\n" FMA8 SMOV2 FMA8 SMOV2 FMA8 SMOV2
                         "s_branch 2f\n s_nop 0\n 2:\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(m), "v"(c), "s"(tab), "v"(t), "v"((unsigned)(size_t)lp)
                         : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + t;
}

template <int MODE> int run(double* out, const double* tab, int w, const char* name) {
    const int nblk = 256 * 4 * w, iters = 20000;
    hipLaunchKernelGGL((k<MODE>), dim3(nblk), dim3(64), 0, 0, out, tab, 0.999999, 1e-7, 2000);
    CHK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<MODE>), dim3(nblk), dim3(64), 0, 0, out, tab, 0.999999, 1e-7, iters);
    CHK(hipEventRecord(e1));
    CHK(hipDeviceSynchronize());
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    const double nfma = 48.0 * iters;
    printf("%-44s waves/SIMD=%d : %.3f ns per fma per wave, %.3f ns per fma per SIMD\n", name, w, ms * 1e6 / nfma, ms * 1e6 / nfma / w);
    return 0;
}
int main() {
    double *out, *tab;
    CHK(hipMalloc(&out, 8 * 64 * 8192)); CHK(hipMalloc(&tab, 4096)); CHK(hipMemset(tab, 0, 4096));
    for (int rep = 0; rep < 2; rep++)
    for (int w : {1, 2, 3}) {
        run<0>(out, tab, w, "A fma x4 chains");
        run<1>(out, tab, w, "B + 2 s_mov per 8 fma");
        run<2>(out, tab, w, "C + s_load, immediate wait, per 48 fma");
        run<3>(out, tab, w, "D + s_load issued 24 fma before its wait");
        run<4>(out, tab, w, "E + ds_read, immediate wait, per 48 fma");
        run<5>(out, tab, w, "F + taken branch per 24 fma");
        run<6>(out, tab, w, "G one dependent chain");
        run<7>(out, tab, w, "H B+C+E+F together");
    }
    return 0;
}
