// Micro-experiment for the peer-write exchange (csrc/cude_xchg.hip): two PROCESSES on one GPU, each owning a mailbox in
// uncached / fine-grained / plain device memory shared through hipIpcGetMemHandle, exchange 69 doubles per round with
// 8-byte {sequence, half} words written by system-scope atomic stores and polled by system-scope atomic loads.
// Answers: does hipIpcGetMemHandle accept hipExtMallocWithFlags memory; do two processes' kernels run side by side
// (a spinning kernel of one must not keep the other's from starting); what a round costs.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/xchg_ipc.hip -o tools/ubench/xchg_ipc && tools/ubench/xchg_ipc [kind] [rounds]
//   kind: 0 = hipMalloc, 1 = fine-grained, 3 = uncached
// The parent forks BEFORE any HIP call and never touches the GPU itself.
#include <hip/hip_runtime.h>
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x)                                                                                       \
    do {                                                                                            \
        hipError_t e_ = (x);                                                                        \
        if (e_ != hipSuccess) {                                                                     \
            fprintf(stderr, "[rank %d] %s: %s\n", g_rank, #x, hipGetErrorString(e_));               \
            _exit(3);                                                                               \
        }                                                                                           \
    } while (0)

static int g_rank = -1;
constexpr int kCols = 69, kRanks = 2;

struct Args {
    unsigned long long* peers[kRanks];
    unsigned* seq;      // [kCols]
    int* status;
    double* in;         // [kCols]
    double* out;        // [kCols]
    int rank;
    long long timeout;  // wall_clock64 ticks (100 MHz)
};

__global__ __launch_bounds__(64) void xchg_kernel(Args a) {
    if (threadIdx.x != 0) return;
    const int col = blockIdx.x;
    const unsigned s = a.seq[col] + 1u;
    const unsigned long long bits = (unsigned long long)__double_as_longlong(a.in[col] + (double)s);
    const unsigned long long w0 = ((unsigned long long)s << 32) | (bits & 0xffffffffull);
    const unsigned long long w1 = ((unsigned long long)s << 32) | (bits >> 32);
    const size_t mine = (((size_t)(s & 1u) * kRanks + a.rank) * kCols + col) * 2;
    for (int r = 0; r < kRanks; r++) {
        __hip_atomic_store(a.peers[r] + mine, w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(a.peers[r] + mine + 1, w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    const long long t0 = wall_clock64();
    double acc = 0.0;
    for (int r = 0; r < kRanks; r++) {
        const unsigned long long* p = a.peers[a.rank] + (((size_t)(s & 1u) * kRanks + r) * kCols + col) * 2;
        unsigned long long v0, v1;
        for (;;) {
            v0 = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            v1 = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if ((unsigned)(v0 >> 32) == s && (unsigned)(v1 >> 32) == s) break;
            if (wall_clock64() - t0 > a.timeout) {
                *a.status = 1;
                a.out[col] = __builtin_nan("");
                a.seq[col] = s;
                return;
            }
            __builtin_amdgcn_s_sleep(4);
        }
        const double vr = __longlong_as_double((long long)((v1 << 32) | (v0 & 0xffffffffull)));
        acc = r == 0 ? vr : acc + vr;
    }
    a.out[col] = acc;
    a.seq[col] = s;
}

// a long kernel in front of the exchange in ONE of the two processes: the other's exchange kernel must spin meanwhile
__global__ void busy_kernel(long long ticks, int* sink) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (sink) *sink = 1;
}

static void child(int rank, int kind, int rounds, int rd, int wr) {
    g_rank = rank;
    CK(hipSetDevice(0));
    const size_t words = (size_t)2 * kRanks * kCols * 2;
    unsigned long long* box = nullptr;
    if (kind == 0) CK(hipMalloc((void**)&box, words * 8));
    else CK(hipExtMallocWithFlags((void**)&box, words * 8, (unsigned)kind));
    CK(hipMemset(box, 0, words * 8));
    hipIpcMemHandle_t mine, theirs;
    CK(hipIpcGetMemHandle(&mine, box));
    if (write(wr, &mine, sizeof(mine)) != (ssize_t)sizeof(mine)) _exit(4);
    if (read(rd, &theirs, sizeof(theirs)) != (ssize_t)sizeof(theirs)) _exit(4);
    unsigned long long* peer = nullptr;
    CK(hipIpcOpenMemHandle((void**)&peer, theirs, hipIpcMemLazyEnablePeerAccess));
    Args a{};
    a.peers[rank] = box;
    a.peers[1 - rank] = peer;
    a.rank = rank;
    a.timeout = 300000000LL;  // 3 s
    CK(hipMalloc((void**)&a.seq, kCols * sizeof(unsigned)));
    CK(hipMemset(a.seq, 0, kCols * sizeof(unsigned)));
    CK(hipMalloc((void**)&a.status, sizeof(int)));
    CK(hipMemset(a.status, 0, sizeof(int)));
    CK(hipMalloc((void**)&a.in, kCols * 8));
    CK(hipMalloc((void**)&a.out, kCols * 8));
    double h_in[kCols], h_out[kCols];
    for (int q = 0; q < kCols; q++) h_in[q] = (rank + 1) * 1000.0 + q * 0.125;
    CK(hipMemcpy(a.in, h_in, sizeof(h_in), hipMemcpyHostToDevice));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    // round-trip barrier through the pipes so both start together
    char c = 1;
    if (write(wr, &c, 1) != 1 || read(rd, &c, 1) != 1) _exit(4);
    // (1) skewed arrival: rank 1 is busy for 200 ms first
    if (rank == 1) hipLaunchKernelGGL(busy_kernel, dim3(1), dim3(64), 0, s, 20000000LL, (int*)nullptr);
    hipLaunchKernelGGL(xchg_kernel, dim3(kCols), dim3(64), 0, s, a);
    CK(hipStreamSynchronize(s));
    int st = 0;
    CK(hipMemcpy(&st, a.status, sizeof(int), hipMemcpyDeviceToHost));
    CK(hipMemcpy(h_out, a.out, sizeof(h_out), hipMemcpyDeviceToHost));
    int bad = 0;
    for (int q = 0; q < kCols; q++)
        if (h_out[q] != (1000.0 + q * 0.125 + 1.0) + (2000.0 + q * 0.125 + 1.0)) bad++;
    printf("[rank %d] kind %d skewed round: status %d, wrong columns %d\n", rank, kind, st, bad);
    fflush(stdout);
    if (st || bad) _exit(5);
    // (2) throughput: `rounds` exchanges back to back, plain launches
    auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < rounds; k++) hipLaunchKernelGGL(xchg_kernel, dim3(kCols), dim3(64), 0, s, a);
    CK(hipStreamSynchronize(s));
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / rounds;
    CK(hipMemcpy(&st, a.status, sizeof(int), hipMemcpyDeviceToHost));
    CK(hipMemcpy(h_out, a.out, sizeof(h_out), hipMemcpyDeviceToHost));
    bad = 0;
    const double sq = rounds + 1.0;
    for (int q = 0; q < kCols; q++)
        if (h_out[q] != (1000.0 + q * 0.125 + sq) + (2000.0 + q * 0.125 + sq)) bad++;
    printf("[rank %d] kind %d: %d rounds, %.2f us per round (plain launches), status %d, wrong columns %d\n", rank, kind,
           rounds, us, st, bad);
    // (3) the same inside a captured graph of 8 rounds
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int k = 0; k < 8; k++) hipLaunchKernelGGL(xchg_kernel, dim3(kCols), dim3(64), 0, s, a);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < rounds / 8; k++) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (rounds / 8 * 8);
    CK(hipMemcpy(&st, a.status, sizeof(int), hipMemcpyDeviceToHost));
    printf("[rank %d] kind %d: %.2f us per round inside 8-round graphs, status %d\n", rank, kind, us, st);
    fflush(stdout);
    CK(hipIpcCloseMemHandle(peer));
    _exit(st || bad ? 5 : 0);
}

int main(int argc, char** argv) {
    const int kind = argc > 1 ? atoi(argv[1]) : 3;
    const int rounds = argc > 2 ? atoi(argv[2]) : 2000;
    int p01[2], p10[2];
    if (pipe(p01) || pipe(p10)) return 2;
    pid_t kids[2];
    for (int r = 0; r < 2; r++) {
        kids[r] = fork();
        if (kids[r] == 0) {
            if (r == 0) child(0, kind, rounds, p10[0], p01[1]);
            else child(1, kind, rounds, p01[0], p10[1]);
        }
    }
    int rc = 0;
    for (int r = 0; r < 2; r++) {
        int st = 0;
        waitpid(kids[r], &st, 0);
        if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) rc = 1;
    }
    printf("kind %d: %s\n", kind, rc ? "FAILED" : "ok");
    return rc;
}
