// libcude_hip.so -- multi-GPU: subjects are sharded, one context per GPU; the shared network's gradient (with the loss
// sum and the failure count: P+2 doubles) is summed over the ranks once per optimiser step.  Two transports:
//   * RCCL (dlopen'ed; ncclAllReduce on the context's stream), cude_comm_*;
//   * the peer-write exchange (cude_xchg.h): mailboxes in device memory mapped into the peers through HIP IPC, the
//     reduction kernel itself writes and collects the words -- capturable, bitwise reproducible, cude_xchg_*.
#include <unistd.h>

#include "cude_ctx.h"

namespace cude {
namespace api {

// ---------------------------------------------------------------------------------- RCCL (dlopen)
typedef struct { char internal[CUDE_UNIQUE_ID_BYTES]; } nccl_uid;
struct Rccl {
    void* handle = nullptr;
    int (*GetUniqueId)(nccl_uid*) = nullptr;
    int (*CommInitRank)(void**, int, nccl_uid, int) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int (*CommCount)(void*, int*) = nullptr;
    int (*CommUserRank)(void*, int*) = nullptr;
    int (*GetVersion)(int*) = nullptr;
    const char* (*GetLastError)(void*) = nullptr;      // NCCL >= 2.13: the library's own description of what went wrong
};
Rccl g_rccl;

int32_t load_rccl() {
    if (g_rccl.handle) return CUDE_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names) {   // reuse a copy already in the process (e.g. PyTorch's) first
        h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        if (h) break;
    }
    if (!h)
        for (const char* n : names) {
            h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (h) break;
        }
    if (!h) return fail(CUDE_ERR_COMM, std::string("cannot load librccl: ") + dlerror());
    g_rccl.GetUniqueId = (int (*)(nccl_uid*))dlsym(h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (int (*)(void**, int, nccl_uid, int))dlsym(h, "ncclCommInitRank");
    g_rccl.AllReduce =
        (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(h, "ncclAllReduce");
    g_rccl.CommDestroy = (int (*)(void*))dlsym(h, "ncclCommDestroy");
    g_rccl.GetErrorString = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
    g_rccl.CommCount = (int (*)(void*, int*))dlsym(h, "ncclCommCount");
    g_rccl.CommUserRank = (int (*)(void*, int*))dlsym(h, "ncclCommUserRank");
    g_rccl.GetVersion = (int (*)(int*))dlsym(h, "ncclGetVersion");
    g_rccl.GetLastError = (const char* (*)(void*))dlsym(h, "ncclGetLastError");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce || !g_rccl.CommDestroy)
        return fail(CUDE_ERR_COMM, "librccl lacks a required symbol");
    g_rccl.handle = h;
    return CUDE_OK;
}

inline std::string rccl_diagnosis(int r) {
    std::string m = g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "rccl error";
    if (g_rccl.GetLastError) {
        const char* last = g_rccl.GetLastError(nullptr);
        if (last && last[0]) m += std::string(" [") + last + "]";
    }
    return m;
}
#define RCCL_TRY(expr)                                                                             \
    do {                                                                                           \
        int _r = (expr);                                                                           \
        if (_r != 0) return fail(CUDE_ERR_COMM, std::string(#expr) + ": " + rccl_diagnosis(_r));   \
    } while (0)

// The enumerators of nccl.h this file needs (librccl is dlopen'ed: its header is not compiled against).  They are not
// trusted: cude_comm_init runs comm_self_test(), which fails unless a sum and a max of known doubles come back right.
constexpr int kNcclFloat64 = 8, kNcclSum = 0, kNcclMax = 2;

cude::XchgArgs xchg_args(const cude_ctx* c) {
    cude::XchgArgs x{};
    const Exchange& e = c->xchg;
    if (!e.attached) return x;
    for (int r = 0; r < c->n_ranks; r++) x.peers[r] = reinterpret_cast<unsigned long long*>(e.peers[r]);
    x.seq = e.seq;
    x.status = e.status;
    x.n_ranks = c->n_ranks; x.rank = c->rank; x.cols = e.cols;
    x.timeout = (long long)(e.timeout_s * 1e8);         // wall_clock64 counts at 100 MHz
    return x;
}

// After a synchronisation of the context's stream: did a device-side wait of the exchange give up?  The status word
// lives in page-locked coherent host memory (the kernels hold its device address): reading it is a host load, so every
// call that synchronises behind an exchange launch asks -- not only those whose loss came back NaN.
int32_t xchg_check(cude_ctx* c) {
    if (!c->xchg.attached || !c->xchg.status_host) return CUDE_OK;
    volatile int32_t* st = c->xchg.status_host;
    if (*st == 0) return CUDE_OK;
    c->xchg_timeouts += 1;
    *st = 0;
    return fail(CUDE_ERR_COMM, "peer-write exchange: a rank's contribution did not arrive within the time limit "
                               "(a peer died, or the ranks are not making the same sequence of calls)");
}

// op: 0 = sum, 1 = max.  Exchange when it is on, else the communicator, else (one rank) nothing.
int32_t allreduce_dev(cude_ctx* c, double* buf, size_t count, int op) {
    if (c->xchg.ready) {
        HIP_TRY(cude::launch_xchg_allreduce(xchg_args(c), buf, (int64_t)count, op, c->stream));
        return CUDE_OK;
    }
    if (!c->comm) return CUDE_OK;
    RCCL_TRY(g_rccl.AllReduce(buf, buf, count, kNcclFloat64, op == 1 ? kNcclMax : kNcclSum, c->comm, c->stream));
    return CUDE_OK;
}

// sum / max of a small host vector over all ranks through the context's transport (identity on one rank)
int32_t comm_reduce_host(cude_ctx* c, double* values, int32_t count, int op) {
    if (!distributed(c)) return CUDE_OK;
    HIP_TRY(c->red_tmp.reserve((size_t)count));
    HIP_TRY(hipMemcpyAsync(c->red_tmp.p, values, count * sizeof(double), hipMemcpyHostToDevice, c->stream));
    int32_t rc = allreduce_dev(c, c->red_tmp.p, (size_t)count, op);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(values, c->red_tmp.p, count * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return c->xchg.ready ? xchg_check(c) : CUDE_OK;
}

// Every rank contributes [1, 2, rank + 1]: the sum must be [n, 2n, n(n+1)/2] and the max [1, 2, n].  A wrong datatype
// or operator enumerator (or a communicator that silently spans fewer ranks) cannot produce both.
int32_t comm_self_test(cude_ctx* c) {
    const double n = (double)c->n_ranks;
    double v[3] = {1.0, 2.0, (double)c->rank + 1.0};
    int32_t rc = comm_reduce_host(c, v, 3, 0);
    if (rc) return rc;
    if (v[0] != n || v[1] != 2.0 * n || v[2] != 0.5 * n * (n + 1.0))
        return fail(CUDE_ERR_COMM, "transport self-test: sum all-reduce of doubles returned a wrong result");
    double w[3] = {1.0, 2.0, (double)c->rank + 1.0};
    if ((rc = comm_reduce_host(c, w, 3, 1))) return rc;
    if (w[0] != 1.0 || w[1] != 2.0 || w[2] != n)
        return fail(CUDE_ERR_COMM, "transport self-test: max all-reduce of doubles returned a wrong result");
    return CUDE_OK;
}

// The exchange's own self-test: kXchgTestRounds sum rounds over MORE values than the mailbox has columns (every column,
// both parities, every slot written at least twice: a stale word from the previous round would be a wrong sum, since the
// values change with the round), then one max round.  Rank r contributes (r + 1) * (j + 1) + round to value j.
constexpr int kXchgTestRounds = 4;
int32_t xchg_self_test(cude_ctx* c) {
    const double n = (double)c->n_ranks;
    const int count = 2 * c->xchg.cols + 1;
    std::vector<double> v((size_t)count);
    for (int round = 0; round < kXchgTestRounds; round++) {
        for (int j = 0; j < count; j++) v[(size_t)j] = ((double)c->rank + 1.0) * (j + 1) + round;
        int32_t rc = comm_reduce_host(c, v.data(), count, 0);
        if (rc) return rc;
        for (int j = 0; j < count; j++)
            if (v[(size_t)j] != 0.5 * n * (n + 1.0) * (j + 1) + n * round)
                return fail(CUDE_ERR_COMM, "round " + std::to_string(round) + ", value " + std::to_string(j) +
                                               ": the sum over the ranks came back wrong (stale or torn mailbox words)");
    }
    double w[3] = {1.0, 2.0, (double)c->rank + 1.0};
    int32_t rc = comm_reduce_host(c, w, 3, 1);
    if (rc) return rc;
    if (w[0] != 1.0 || w[1] != 2.0 || w[2] != n) return fail(CUDE_ERR_COMM, "the max over the ranks came back wrong");
    return CUDE_OK;
}

// Multi-process RCCL and HIP IPC need dmabuf IPC on hosts whose driver has no legacy IPC: without
// HSA_ENABLE_IPC_MODE_LEGACY=0 in the environment BEFORE the first HIP call, ncclCommInitRank / hipIpcGetMemHandle die
// much later with "invalid argument".  Too late to set it when a communicator is asked for, so it is checked then.
bool ipc_mode_ok() {
    const char* ipc = getenv("HSA_ENABLE_IPC_MODE_LEGACY");
    if ((ipc && std::strcmp(ipc, "0") == 0) || getenv("CUDE_ALLOW_LEGACY_IPC")) return true;
    fail(CUDE_ERR_COMM, "export HSA_ENABLE_IPC_MODE_LEGACY=0 before the process touches the GPU (dmabuf IPC for RCCL and "
                        "for the exchange's mailboxes); set CUDE_ALLOW_LEGACY_IPC=1 to skip this check");
    return false;
}

void comm_release(cude_ctx* c) {
    if (c->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);
    c->comm = nullptr;
}

// What a rank tells its peers about its mailbox (CUDE_XCHG_HANDLE_BYTES = 128): the HIP IPC handle for other processes,
// the plain address for contexts of the same process (which cannot open their own process's handle).
struct XchgHandle {
    uint64_t magic;             // "CUDEXCH1"
    int64_t pid;
    uint64_t address;           // of the mailbox in the exporting process
    int32_t n_ranks, rank, cols, device;
    hipIpcMemHandle_t ipc;      // 64 bytes
    int32_t pci[3];             // domain, bus, device of the GPU the mailbox lives on (device ordinals are per process)
    int32_t kind_index;         // 0 uncached, 1 fine-grained, 2 ordinary device memory
};
static_assert(sizeof(XchgHandle) <= CUDE_XCHG_HANDLE_BYTES, "handle layout");
constexpr uint64_t kXchgMagic = 0x3148435845445543ull;

void xchg_release(cude_ctx* c) {
    Exchange& e = c->xchg;
    for (int r = 0; r < CUDE_XCHG_MAX_RANKS; r++) {
        if (e.opened[r] && e.peers[r]) (void)hipIpcCloseMemHandle(e.peers[r]);
        e.opened[r] = false;
        e.peers[r] = nullptr;
    }
    if (e.box) (void)hipFree(e.box);
    if (e.seq) (void)hipFree(e.seq);
    if (e.status_host) (void)hipHostFree(e.status_host);
    e = Exchange{};
}

// cude::ReduceFn over the context's communicator (the L-BFGS stage of cude_train_restarts on a sharded population)
int32_t lbfgs_comm_reduce(double* values, int32_t count, int32_t op, void* user) {
    return comm_reduce_host(static_cast<cude_ctx*>(user), values, count, op);
}

}  // namespace api
}  // namespace cude

using namespace cude::api;

extern "C" {

int32_t cude_comm_unique_id(uint8_t id[CUDE_UNIQUE_ID_BYTES]) {
    if (!id) return fail(CUDE_ERR_ARG, "null id");
    int32_t rc = load_rccl();
    if (rc) return rc;
    nccl_uid u;
    RCCL_TRY(g_rccl.GetUniqueId(&u));
    std::memcpy(id, u.internal, CUDE_UNIQUE_ID_BYTES);
    return CUDE_OK;
}

int32_t cude_comm_init(cude_ctx* c, int32_t n_ranks, int32_t rank, const uint8_t id[CUDE_UNIQUE_ID_BYTES]) {
    int32_t rc = bind(c);
    if (rc) return rc;
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks || !id) return fail(CUDE_ERR_ARG, "bad communicator arguments");
    if (c->comm) return fail(CUDE_ERR_STATE, "communicator already attached");
    if (c->xchg.attached && (c->n_ranks != n_ranks || c->rank != rank))
        return fail(CUDE_ERR_ARG, "the communicator must span the same ranks as the attached exchange");
    // Multi-process RCCL needs dmabuf IPC on hosts whose driver has no legacy IPC: without
    // HSA_ENABLE_IPC_MODE_LEGACY=0 in the environment BEFORE the first HIP call, ncclCommInitRank dies much later
    // with "hipIpcGetMemHandle: invalid argument".  Too late to set it here, so say so now.
    if (n_ranks > 1 && !ipc_mode_ok()) return CUDE_ERR_COMM;
    if ((rc = load_rccl())) return rc;
    nccl_uid u;
    std::memcpy(u.internal, id, CUDE_UNIQUE_ID_BYTES);
    RCCL_TRY(g_rccl.CommInitRank(&c->comm, n_ranks, u, rank));
    const bool xon = c->xchg.ready;
    c->xchg.ready = false;                   // the self-test is to go through the communicator
    c->n_ranks = n_ranks;
    c->rank = rank;
    rc = comm_self_test(c);
    c->xchg.ready = xon;
    if (rc) {          // (collective: every rank runs it, every rank sees the same verdict)
        comm_release(c);
        if (!c->xchg.attached) { c->n_ranks = 1; c->rank = 0; }
        return rc;
    }
    return CUDE_OK;
}

int32_t cude_comm_allreduce_host(cude_ctx* c, double* values, int32_t count) {
    int32_t rc = bind(c);
    if (rc) return rc;
    if (!values || count < 1) return fail(CUDE_ERR_ARG, "bad buffer");
    return comm_reduce_host(c, values, count, 0);   // single rank: identity
}

// The kinds of memory a mailbox can live in, in order of preference: uncached device memory (what RCCL keeps its flags
// in: peers' writes and the owner's polls meet in HBM, not in an L2), fine-grained device memory, ordinary device memory
// + system-scope accesses (all three measured equal on ONE GPU, tools/ubench/xchg_ipc.hip; the last is refused when a
// peer sits on another device: its owner's polls may be served from an L2 that the peers' xGMI writes do not pass).
constexpr unsigned kXchgKinds[3] = {hipDeviceMallocUncached, hipDeviceMallocFinegrained, 0u};

int32_t cude_xchg_export(cude_ctx* c, int32_t n_ranks, int32_t rank, uint8_t handle[CUDE_XCHG_HANDLE_BYTES]) {
    int32_t rc = bind(c);
    if (rc) return rc;
    if (n_ranks < 1 || n_ranks > CUDE_XCHG_MAX_RANKS || rank < 0 || rank >= n_ranks || !handle)
        return fail(CUDE_ERR_ARG, "bad exchange arguments (1 <= n_ranks <= 16)");
    if (c->xchg.box) return fail(CUDE_ERR_STATE, "exchange already exported on this context (cude_xchg_detach releases it)");
    if (c->comm && (c->n_ranks != n_ranks || c->rank != rank))
        return fail(CUDE_ERR_ARG, "the exchange must span the same ranks as the attached communicator");
    if (n_ranks > 1 && !ipc_mode_ok()) return CUDE_ERR_COMM;
    if (c->xchg_next_kind > 2)
        return fail(CUDE_ERR_COMM, "exchange mailbox: every kind of device memory has been tried on this context");
    Exchange& e = c->xchg;
    e.cols = c->P + 2;
    e.box_words = (size_t)2 * n_ranks * e.cols * 2;
    XchgHandle h{};
    // a kind counts only if the block can be allocated AND cleared AND exported: a failure of any of the three moves on
    // to the next kind (the clearing is ordered on the context's stream, which does not wait for the null stream)
    std::string why;
    for (int k = c->xchg_next_kind; k < 3 && !e.box; k++) {
        void* p = nullptr;
        hipError_t he = kXchgKinds[k] ? hipExtMallocWithFlags(&p, e.box_words * 8, kXchgKinds[k]) : hipMalloc(&p, e.box_words * 8);
        const char* step = "allocation";
        if (he == hipSuccess) {
            step = "clearing";
            he = hipMemsetAsync(p, 0, e.box_words * 8, c->stream);
            if (he == hipSuccess) he = hipStreamSynchronize(c->stream);
        }
        if (he == hipSuccess) {
            step = "hipIpcGetMemHandle";
            he = hipIpcGetMemHandle(&h.ipc, p);
        }
        if (he == hipSuccess) {
            e.box = static_cast<uint64_t*>(p);
            e.kind = (int)kXchgKinds[k];
            e.kind_index = k;
        } else {
            why += std::string(why.empty() ? "" : "; ") + "kind " + std::to_string(k) + ": " + step + ": " + hipGetErrorString(he);
            if (p) (void)hipFree(p);
            (void)hipGetLastError();
        }
    }
    if (!e.box) {
        c->xchg_next_kind = 3;
        e = Exchange{};
        return fail(CUDE_ERR_HIP, "exchange mailbox: " + why);
    }
    hipError_t he;
    if ((he = hipMalloc((void**)&e.seq, e.cols * sizeof(uint32_t))) != hipSuccess ||
        (he = hipHostMalloc((void**)&e.status_host, sizeof(int32_t), hipHostMallocCoherent | hipHostMallocMapped)) != hipSuccess ||
        (he = hipHostGetDevicePointer((void**)&e.status, e.status_host, 0)) != hipSuccess ||
        (he = hipMemsetAsync(e.seq, 0, e.cols * sizeof(uint32_t), c->stream)) != hipSuccess ||
        (he = hipStreamSynchronize(c->stream)) != hipSuccess) {
        xchg_release(c);
        return fail(CUDE_ERR_HIP, std::string("exchange mailbox: ") + hipGetErrorString(he));
    }
    *e.status_host = 0;
    h.magic = kXchgMagic;
    h.pid = (int64_t)getpid();
    h.address = (uint64_t)(uintptr_t)e.box;
    h.n_ranks = n_ranks; h.rank = rank; h.cols = e.cols; h.device = c->cfg.device;
    h.kind_index = e.kind_index;
    int v = 0;
    h.pci[0] = hipDeviceGetAttribute(&v, hipDeviceAttributePciDomainID, c->cfg.device) == hipSuccess ? v : -1;
    h.pci[1] = hipDeviceGetAttribute(&v, hipDeviceAttributePciBusId, c->cfg.device) == hipSuccess ? v : -1;
    h.pci[2] = hipDeviceGetAttribute(&v, hipDeviceAttributePciDeviceId, c->cfg.device) == hipSuccess ? v : -1;
    (void)hipGetLastError();
    std::memset(handle, 0, CUDE_XCHG_HANDLE_BYTES);
    std::memcpy(handle, &h, sizeof(h));
    c->n_ranks = n_ranks;
    c->rank = rank;
    return CUDE_OK;
}

// a failed attach: everything released, the next export starts one kind further down the order of preference.  The
// ranks move through these LEVELS in lock step (every rank detaches after a failed attempt); a rank whose own export had
// to skip a kind simply offers the same kind again at the next level.
static int32_t xchg_attach_failed(cude_ctx* c, int32_t code, const std::string& msg) {
    c->xchg_next_kind += 1;         // (one LEVEL per attempt on every rank, whatever kind this rank's export had ended up with)
    (void)hipStreamSynchronize(c->stream);
    xchg_release(c);
    if (!c->comm) { c->n_ranks = 1; c->rank = 0; }
    return fail(code, msg);
}

int32_t cude_xchg_attach(cude_ctx* c, const uint8_t* handles, double timeout_s) {
    int32_t rc = bind(c);
    if (rc) return rc;
    Exchange& e = c->xchg;
    if (!e.box) return fail(CUDE_ERR_STATE, "call cude_xchg_export first");
    if (e.attached) return fail(CUDE_ERR_STATE, "exchange already attached");
    if (!handles || !(timeout_s > 0) || !(timeout_s <= 3600)) return fail(CUDE_ERR_ARG, "null handles / bad time limit");
    const int n = c->n_ranks;
    XchgHandle own;
    std::memcpy(&own, handles + (size_t)c->rank * CUDE_XCHG_HANDLE_BYTES, sizeof(own));
    bool other_device = false;
    int same_device_same_process = 0;
    for (int r = 0; r < n; r++) {
        XchgHandle h;
        std::memcpy(&h, handles + (size_t)r * CUDE_XCHG_HANDLE_BYTES, sizeof(h));
        if (h.magic != kXchgMagic || h.n_ranks != n || h.rank != r || h.cols != e.cols)
            return xchg_attach_failed(c, CUDE_ERR_ARG, "exchange handle " + std::to_string(r) + " does not describe rank " +
                                                           std::to_string(r) + " of " + std::to_string(n) +
                                                           " with the same network shape (a rank whose export failed sends zeros)");
        const bool same_gpu = h.pci[0] == own.pci[0] && h.pci[1] == own.pci[1] && h.pci[2] == own.pci[2] && h.pci[1] >= 0;
        if (!same_gpu) other_device = true;
        if (r == c->rank) {
            e.peers[r] = e.box;
        } else if (h.pid == (int64_t)getpid()) {
            // a context of this process: its address is ours too (another device of the node: peer access)
            if (h.device != c->cfg.device) {
                hipError_t pe = hipDeviceEnablePeerAccess(h.device, 0);
                if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled)
                    return xchg_attach_failed(c, CUDE_ERR_HIP, std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(pe));
                (void)hipGetLastError();
            } else {
                same_device_same_process++;
            }
            e.peers[r] = reinterpret_cast<uint64_t*>((uintptr_t)h.address);
        } else {
            void* p = nullptr;
            hipError_t oe = hipIpcOpenMemHandle(&p, h.ipc, hipIpcMemLazyEnablePeerAccess);
            if (oe != hipSuccess) {
                (void)hipGetLastError();
                return xchg_attach_failed(c, CUDE_ERR_HIP, std::string("hipIpcOpenMemHandle (rank ") + std::to_string(r) +
                                                               ", memory kind " + std::to_string(h.kind_index) + "): " +
                                                               hipGetErrorString(oe));
            }
            e.peers[r] = static_cast<uint64_t*>(p);
            e.opened[r] = true;
        }
    }
    // Ranks of ONE process on ONE device wait for each other in spinning kernels on separate streams: the runtime maps
    // streams onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default), and a peer's kernels queued behind a spinning
    // one never run.  A rehearsal configuration; bounded so that it cannot stall until the time limit.
    if (same_device_same_process + 1 > 4)
        return xchg_attach_failed(c, CUDE_ERR_UNSUPPORTED, "more than 4 ranks of one process on one device: their spinning "
                                                           "reduction kernels would share hardware queues (one rank per device "
                                                           "is the supported configuration)");
    if (other_device && e.kind_index == 2 && !c->opt.xchg_allow_plain)
        return xchg_attach_failed(c, CUDE_ERR_COMM, "only ordinary device memory was left for the mailbox and peers sit on other "
                                                    "devices: the owner's polls could be served from its L2 (option "
                                                    "\"xchg_allow_plain\" / CUDE_ALLOW_PLAIN_MAILBOX=1 to try all the same)");
    e.timeout_s = timeout_s;
    e.attached = true;
    e.ready = true;
    drop_graph(c);                              // captured iterations carry the reduction kernels' arguments
    rc = xchg_self_test(c);                     // (collective: every rank runs it; a rank whose peers never write times out)
    if (rc == CUDE_OK && ((c->opt.xchg_fail_kinds >> e.kind_index) & 1))
        rc = fail(CUDE_ERR_COMM, "failure of memory kind " + std::to_string(e.kind_index) + " forced by option xchg_fail_kinds");
    if (rc) return xchg_attach_failed(c, rc, "exchange self-test (memory kind " + std::to_string(e.kind_index) + "): " +
                                                 cude_last_error());
    return CUDE_OK;
}

int32_t cude_xchg_detach(cude_ctx* c) {
    int32_t rc = bind(c);
    if (rc) return rc;
    if (!c->xchg.box) return CUDE_OK;           // (a failed attach has released it and moved on to the next kind already)
    HIP_TRY(hipStreamSynchronize(c->stream));
    drop_graph(c);
    c->xchg_next_kind += 1;
    xchg_release(c);
    if (!c->comm) { c->n_ranks = 1; c->rank = 0; }
    return CUDE_OK;
}

int32_t cude_xchg_enable(cude_ctx* c, int32_t enabled) {
    int32_t rc = bind(c);
    if (rc) return rc;
    if (!c->xchg.attached) return fail(CUDE_ERR_STATE, "no exchange attached");
    if (!enabled && !c->comm && c->n_ranks > 1)
        return fail(CUDE_ERR_STATE, "switching the exchange off needs a communicator to take over (cude_comm_init)");
    HIP_TRY(hipStreamSynchronize(c->stream));
    drop_graph(c);
    c->xchg.ready = enabled != 0;
    return CUDE_OK;
}

int32_t cude_xchg_info(cude_ctx* c, int32_t* n_ranks, int32_t* rank, int32_t* memory_kind, int32_t* timeouts) {
    if (!c) return fail(CUDE_ERR_ARG, "null context");
    const bool on = c->xchg.attached;
    if (n_ranks) *n_ranks = on ? c->n_ranks : 1;
    if (rank) *rank = on ? c->rank : 0;
    if (memory_kind) *memory_kind = on ? c->xchg.kind : 0;
    if (timeouts) *timeouts = c->xchg_timeouts;
    return CUDE_OK;
}

int32_t cude_comm_info(cude_ctx* c, int32_t* n_ranks, int32_t* rank, int32_t* version) {
    int32_t rc = bind(c);
    if (rc) return rc;
    if (!n_ranks || !rank || !version) return fail(CUDE_ERR_ARG, "null output");
    *n_ranks = 1; *rank = 0; *version = 0;
    if (!c->comm) return CUDE_OK;
    if (!g_rccl.CommCount || !g_rccl.CommUserRank) return fail(CUDE_ERR_COMM, "librccl lacks ncclCommCount/ncclCommUserRank");
    int n = 0, r = 0, v = 0;
    RCCL_TRY(g_rccl.CommCount(c->comm, &n));
    RCCL_TRY(g_rccl.CommUserRank(c->comm, &r));
    if (g_rccl.GetVersion) RCCL_TRY(g_rccl.GetVersion(&v));
    *n_ranks = n; *rank = r; *version = v;
    return CUDE_OK;
}

}  // extern "C"
