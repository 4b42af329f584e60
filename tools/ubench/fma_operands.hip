// fp64 FMA issue rate on gfx950 as a function of WHERE the operands come from: number of distinct VGPR pairs read,
// their bank parity (a 64-bit operand occupies banks {0,1} or {2,3} of the 4-bank VGPR file), SGPR / inline constants.
// 4 independent chains, 1..3 waves per SIMD.  Registers are hard-coded so the allocator cannot rearrange them.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s\n",hipGetErrorString(e)); return 1;}}while(0)

// chains: v[10:11] v[14:15] v[18:19] v[22:23]  (all "odd" pairs: banks 2,3)   alt chains on even pairs: v[12:13] v[16:17] v[20:21] v[24:25]
// sources: even pairs v[40:41] v[44:45] ; odd pairs v[42:43] v[46:47]
#define INIT "v_mov_b32 v10, 0\n v_mov_b32 v11, 0x3ff00000\n v_mov_b32 v14, 0\n v_mov_b32 v15, 0x3ff00000\n v_mov_b32 v18, 0\n v_mov_b32 v19, 0x3ff00000\n v_mov_b32 v22, 0\n v_mov_b32 v23, 0x3ff00000\n" \
             "v_mov_b32 v12, 0\n v_mov_b32 v13, 0x3ff00000\n v_mov_b32 v16, 0\n v_mov_b32 v17, 0x3ff00000\n v_mov_b32 v20, 0\n v_mov_b32 v21, 0x3ff00000\n v_mov_b32 v24, 0\n v_mov_b32 v25, 0x3ff00000\n" \
             "v_mov_b32 v40, 0\n v_mov_b32 v41, 0x3fefffff\n v_mov_b32 v42, 0\n v_mov_b32 v43, 0x3fefffff\n v_mov_b32 v44, 0\n v_mov_b32 v45, 0x3e000000\n v_mov_b32 v46, 0\n v_mov_b32 v47, 0x3e000000\n" \
             "s_mov_b32 s20, 0\n s_mov_b32 s21, 0x3fefffff\n"
#define CLOB "v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v40","v41","v42","v43","v44","v45","v46","v47","s20","s21"
#define R4(X) X X X X
#define R16(X) R4(R4(X))
// one group = 4 fma, one per chain (odd-pair chains)
#define G_1PAIR      "v_fma_f64 v[10:11], v[10:11], s[20:21], v[10:11]\n v_fma_f64 v[14:15], v[14:15], s[20:21], v[14:15]\n v_fma_f64 v[18:19], v[18:19], s[20:21], v[18:19]\n v_fma_f64 v[22:23], v[22:23], s[20:21], v[22:23]\n"
#define G_2PAIR_DIFF "v_fma_f64 v[10:11], v[10:11], s[20:21], v[44:45]\n v_fma_f64 v[14:15], v[14:15], s[20:21], v[44:45]\n v_fma_f64 v[18:19], v[18:19], s[20:21], v[44:45]\n v_fma_f64 v[22:23], v[22:23], s[20:21], v[44:45]\n"
#define G_2PAIR_SAME "v_fma_f64 v[10:11], v[10:11], s[20:21], v[46:47]\n v_fma_f64 v[14:15], v[14:15], s[20:21], v[46:47]\n v_fma_f64 v[18:19], v[18:19], s[20:21], v[46:47]\n v_fma_f64 v[22:23], v[22:23], s[20:21], v[46:47]\n"
#define G_2PAIR_MUL  "v_fma_f64 v[10:11], v[40:41], s[20:21], v[10:11]\n v_fma_f64 v[14:15], v[40:41], s[20:21], v[14:15]\n v_fma_f64 v[18:19], v[40:41], s[20:21], v[18:19]\n v_fma_f64 v[22:23], v[40:41], s[20:21], v[22:23]\n"
#define G_3PAIR_A    "v_fma_f64 v[10:11], v[10:11], v[40:41], v[44:45]\n v_fma_f64 v[14:15], v[14:15], v[40:41], v[44:45]\n v_fma_f64 v[18:19], v[18:19], v[40:41], v[44:45]\n v_fma_f64 v[22:23], v[22:23], v[40:41], v[44:45]\n"
#define G_3PAIR_B    "v_fma_f64 v[10:11], v[10:11], v[40:41], v[46:47]\n v_fma_f64 v[14:15], v[14:15], v[40:41], v[46:47]\n v_fma_f64 v[18:19], v[18:19], v[40:41], v[46:47]\n v_fma_f64 v[22:23], v[22:23], v[40:41], v[46:47]\n"
#define G_3PAIR_C    "v_fma_f64 v[10:11], v[10:11], v[42:43], v[46:47]\n v_fma_f64 v[14:15], v[14:15], v[42:43], v[46:47]\n v_fma_f64 v[18:19], v[18:19], v[42:43], v[46:47]\n v_fma_f64 v[22:23], v[22:23], v[42:43], v[46:47]\n"
#define G_ACC        "v_fma_f64 v[10:11], v[40:41], v[44:45], v[10:11]\n v_fma_f64 v[14:15], v[40:41], v[44:45], v[14:15]\n v_fma_f64 v[18:19], v[40:41], v[44:45], v[18:19]\n v_fma_f64 v[22:23], v[40:41], v[44:45], v[22:23]\n"
#define G_ACC_MIX    "v_fma_f64 v[10:11], v[40:41], v[46:47], v[10:11]\n v_fma_f64 v[14:15], v[40:41], v[46:47], v[14:15]\n v_fma_f64 v[18:19], v[40:41], v[46:47], v[18:19]\n v_fma_f64 v[22:23], v[40:41], v[46:47], v[22:23]\n"
#define G_FMAC       "v_fmac_f64 v[10:11], v[40:41], v[44:45]\n v_fmac_f64 v[14:15], v[40:41], v[44:45]\n v_fmac_f64 v[18:19], v[40:41], v[44:45]\n v_fmac_f64 v[22:23], v[40:41], v[44:45]\n"
#define G_MUL2       "v_mul_f64 v[10:11], v[10:11], v[40:41]\n v_mul_f64 v[14:15], v[14:15], v[40:41]\n v_mul_f64 v[18:19], v[18:19], v[40:41]\n v_mul_f64 v[22:23], v[22:23], v[40:41]\n"
#define G_MUL1       "v_mul_f64 v[10:11], v[10:11], s[20:21]\n v_mul_f64 v[14:15], v[14:15], s[20:21]\n v_mul_f64 v[18:19], v[18:19], s[20:21]\n v_mul_f64 v[22:23], v[22:23], s[20:21]\n"
#define G_INLINE     "v_fma_f64 v[10:11], v[10:11], 1.0, v[44:45]\n v_fma_f64 v[14:15], v[14:15], 1.0, v[44:45]\n v_fma_f64 v[18:19], v[18:19], 1.0, v[44:45]\n v_fma_f64 v[22:23], v[22:23], 1.0, v[44:45]\n"

#define KERNEL(NAME, G) __global__ __launch_bounds__(64) void NAME(double* out, int iters) { \
    asm volatile(INIT ::: CLOB); \
    for (int it = 0; it < iters; it++) asm volatile(R16(G) ::: CLOB); \
    double r; asm volatile("v_add_f64 %0, v[10:11], v[14:15]\n v_add_f64 %0, %0, v[18:19]\n v_add_f64 %0, %0, v[22:23]\n" : "=v"(r) :: CLOB); \
    out[blockIdx.x * 64 + threadIdx.x] = r; }
KERNEL(k_1pair, G_1PAIR) KERNEL(k_2diff, G_2PAIR_DIFF) KERNEL(k_2same, G_2PAIR_SAME) KERNEL(k_2mul, G_2PAIR_MUL)
KERNEL(k_3a, G_3PAIR_A) KERNEL(k_3b, G_3PAIR_B) KERNEL(k_3c, G_3PAIR_C) KERNEL(k_acc, G_ACC) KERNEL(k_accmix, G_ACC_MIX)
KERNEL(k_fmac, G_FMAC) KERNEL(k_mul2, G_MUL2) KERNEL(k_mul1, G_MUL1) KERNEL(k_inline, G_INLINE)

template <class K> int run(K kern, double* out, int w, const char* name) {
    const int nblk = 256 * 4 * w, iters = 30000;
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(64), 0, 0, out, 3000);
    CHK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(64), 0, 0, out, iters);
    CHK(hipEventRecord(e1));
    CHK(hipDeviceSynchronize());
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-58s waves/SIMD=%d : %.3f ns per instruction per SIMD\n", name, w, ms * 1e6 / (64.0 * iters) / w);
    return 0;
}
int main() {
    double* out; CHK(hipMalloc(&out, 8 * 64 * 8192));
    for (int w : {1, 2, 3, 4}) {
        run(k_1pair, out, w, "fma d,d,s,d        (1 VGPR pair)");
        run(k_2diff, out, w, "fma d,d,s,c        (2 pairs, c other bank half)");
        run(k_2same, out, w, "fma d,d,s,c        (2 pairs, c same bank half)");
        run(k_2mul, out, w, "fma d,a,s,d        (2 pairs, a other bank half)");
        run(k_inline, out, w, "fma d,d,1.0,c      (2 pairs + inline constant)");
        run(k_3a, out, w, "fma d,d,b,c        (3 pairs: d odd, b even, c even)");
        run(k_3b, out, w, "fma d,d,b,c        (3 pairs: d odd, b even, c odd)");
        run(k_3c, out, w, "fma d,d,b,c        (3 pairs: all odd)");
        run(k_acc, out, w, "fma d,a,b,d        (accumulate, a b even, d odd)");
        run(k_accmix, out, w, "fma d,a,b,d        (accumulate, a even, b odd, d odd)");
        run(k_fmac, out, w, "fmac d,a,b         (VOP2 accumulate)");
        run(k_mul2, out, w, "mul d,d,a          (2 pairs)");
        run(k_mul1, out, w, "mul d,d,s          (1 pair)");
    }
    return 0;
}
