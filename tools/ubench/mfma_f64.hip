// fp64 MFMA (v_mfma_f64_16x16x4_f64) on gfx950: layout check, issue cost, and how much it slows a concurrent
// fp64 VALU stream (the question behind moving the gradient outer products onto the matrix pipe).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#define CHK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s\n",hipGetErrorString(e)); return 1;}}while(0)
typedef double v4d __attribute__((ext_vector_type(4)));

// D = A(16x4) * B(4x16): lane l supplies A[l%16][l/16] and B[l/16][l%16]; receives D[4*(l/16)+r][l%16], r=0..3 ?
__global__ void layout_kernel(const double* A, const double* B, double* D) {
    const int l = threadIdx.x;
    const double a = A[(l % 16) * 4 + l / 16];
    const double b = B[(l / 16) * 16 + l % 16];
    v4d c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; r++) D[l * 4 + r] = c[r];
}

template <int NMFMA, int NFMA>
__global__ __launch_bounds__(64) void mix_kernel(double* out, double seed, int iters) {
    double a[8];
    for (int i = 0; i < 8; i++) a[i] = seed + threadIdx.x * 1e-3 + i;
    v4d c = {0, 0, 0, 0};
    double ma = seed + threadIdx.x, mb = seed * 2 + threadIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int m = 0; m < NMFMA; m++) c = __builtin_amdgcn_mfma_f64_16x16x4f64(ma, mb, c, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < NFMA / 8; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) a[i] = fma(a[i], 0.999999, 1e-7);
        }
    }
    double s = c[0] + c[1] + c[2] + c[3];
    for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int NMFMA, int NFMA> void run(double* out, int waves, const char* name) {
    int nblk = 256 * 4 * waves, iters = 2000;
    hipLaunchKernelGGL((mix_kernel<NMFMA, NFMA>), dim3(nblk), dim3(64), 0, 0, out, 1e-3, 10);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); hipLaunchKernelGGL((mix_kernel<NMFMA, NFMA>), dim3(nblk), dim3(64), 0, 0, out, 1e-3, iters); hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s waves/SIMD=%d: %.1f ns per iteration per SIMD (%.1f ns per wave-iteration)\n", name, waves,
           ms * 1e6 / iters / waves, ms * 1e6 / iters);
}

int main() {
    std::vector<double> A(64), B(64), D(256), ref(256, 0.0);
    for (int i = 0; i < 64; i++) { A[i] = 0.1 * i + 1; B[i] = 0.01 * i * i - 2; }
    for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) for (int k = 0; k < 4; k++) ref[i * 16 + j] += A[i * 4 + k] * B[k * 16 + j];
    double *dA, *dB, *dD, *out;
    CHK(hipMalloc(&dA, 512)); CHK(hipMalloc(&dB, 512)); CHK(hipMalloc(&dD, 2048)); CHK(hipMalloc(&out, 8 * 64 * 8192));
    hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost);
    double err = 0;
    for (int l = 0; l < 64; l++) for (int r = 0; r < 4; r++) err = fmax(err, fabs(D[l * 4 + r] - ref[(4 * (l / 16) + r) * 16 + l % 16]));
    printf("layout D[4*(l/16)+r][l%%16]: max err %.3g\n", err);
    for (int w : {2, 4}) {
        run<0, 464>(out, w, "464 FMA");
        run<16, 0>(out, w, "16 MFMA");
        run<16, 464>(out, w, "16 MFMA + 464 FMA");
        run<4, 464>(out, w, "4 MFMA + 464 FMA");
    }
    return 0;
}
