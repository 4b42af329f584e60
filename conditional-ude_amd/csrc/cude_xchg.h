// The peer-write exchange: the sum of one double per column over the ranks of a node, formed by every rank for itself
// from words its peers wrote straight into its mailbox (host side: cude_comm.hip; used by reduce_partials_kernel and
// xchg_allreduce_kernel in cude_common.hip).
//
// One lane exchanges one double: it writes two 8-byte words {sequence, half of the double} into the slot
// [parity][own rank][column] of EVERY rank's mailbox (atomic 8-byte stores: one fabric write each, never torn), then
// polls the n_ranks slots of that column in its OWN mailbox until each carries the sequence, and combines the values in
// rank order -- the same order on every rank, hence the same bits.  A word is valid exactly when its sequence matches,
// so nothing depends on the order in which writes arrive and no fence is involved.
//
// Why two parities suffice: a rank writes sequence s+1 of a column only after its wait for s has ended, i.e. after
// every peer wrote s; a peer writes s+1 only after it has finished READING s.  So while a rank still reads s, a fast
// peer can be at s+1 (other parity) but not at s+2.
//
// The wait is bounded (XchgArgs::timeout): a lane that sees no peer sets *status, returns NaN and still advances its
// counter -- every wave of a launch that uses the exchange terminates whatever the peers do.
//
// The protocol is written once, over a memory policy: XchgDeviceMem (system-scope atomics, the 100 MHz wall clock) in
// the kernels; tests/cpp/xchg_protocol.cpp runs the SAME function with host threads as ranks (std::atomic accesses,
// under ThreadSanitizer) -- that is where orderings, late peers and time-outs of 2, 3 and 8 ranks are exercised
// without 8 GPUs.  This header therefore includes nothing from HIP unless a HIP compiler reads it.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace cude {

constexpr int kXchgMaxRanks = 16;      // = CUDE_XCHG_MAX_RANKS

// Mailbox of a rank: words [parity][source rank][column][half], each (sequence << 32) | 32 bits of the double;
// sequence = exchanges of that column so far + 1 (device-resident counter, so a captured graph needs no per-iteration
// argument), parity = its low bit.
struct XchgArgs {
    unsigned long long* peers[kXchgMaxRanks] = {};   // every rank's mailbox as mapped into this process; [rank] = the own one
    unsigned* seq = nullptr;           // [cols] sequence counters; nullptr: no exchange
    int* status = nullptr;             // set to 1 by a wait that ran out of time
    int n_ranks = 1, rank = 0, cols = 0;
    long long timeout = 0;             // ticks of the policy's clock (device: wall_clock64, 100 MHz)
};

#ifdef __HIPCC__
#define CUDE_XCHG_HD __host__ __device__
#else
#define CUDE_XCHG_HD
#endif

CUDE_XCHG_HD inline size_t xchg_slot(const XchgArgs& x, unsigned seq, int src, int col) {
    return (((size_t)(seq & 1u) * (size_t)x.n_ranks + (size_t)src) * (size_t)x.cols + (size_t)col) * 2;
}

CUDE_XCHG_HD inline unsigned long long xchg_bits(double v) {
    union { double d; unsigned long long u; } c;
    c.d = v;
    return c.u;
}
CUDE_XCHG_HD inline double xchg_value(unsigned long long u) {
    union { double d; unsigned long long u; } c;
    c.u = u;
    return c.d;
}

// The two halves of an exchange, so that a lane that exchanges several columns (the tail workgroup of the reduction: loss
// sum and failure count) has all its writes in flight before it starts to wait.
// xchg_push: writes this rank's value of `col` into every mailbox; returns the sequence number the wait must see.
// Mem: store(p, word), load(p) -- untorn 8-byte accesses that reach / come from the memory all ranks see --, now(), pause().
template <class Mem>
CUDE_XCHG_HD inline unsigned xchg_push(const XchgArgs& x, int col, double v, Mem mem) {
    const unsigned s = x.seq[col] + 1u;
    const unsigned long long bits = xchg_bits(v);
    const unsigned long long w0 = ((unsigned long long)s << 32) | (bits & 0xffffffffull);
    const unsigned long long w1 = ((unsigned long long)s << 32) | (bits >> 32);
    const size_t mine = xchg_slot(x, s, x.rank, col);
    for (int r = 0; r < x.n_ranks; r++) {
        mem.store(x.peers[r] + mine, w0);
        mem.store(x.peers[r] + mine + 1, w1);
    }
    return s;
}

// xchg_wait: polls the n_ranks slots of `col` in the own mailbox until each carries sequence s and combines the values
// in rank order (op: 0 = sum, 1 = max); advances the column's counter.  t0: the time the caller started waiting
// (one time limit for all the columns a lane waits for).
template <class Mem>
CUDE_XCHG_HD inline double xchg_wait(const XchgArgs& x, int col, unsigned s, int op, long long t0, Mem mem) {
    double acc = 0.0;
    bool lost = false;
    for (int r = 0; r < x.n_ranks && !lost; r++) {
        const unsigned long long* p = x.peers[x.rank] + xchg_slot(x, s, r, col);
        unsigned long long a0, a1;
        for (;;) {
            a0 = mem.load(p);
            a1 = mem.load(p + 1);
            if ((unsigned)(a0 >> 32) == s && (unsigned)(a1 >> 32) == s) break;
            if (mem.now() - t0 > x.timeout) { lost = true; break; }
            mem.pause();
        }
        if (lost) break;
        const double vr = xchg_value((a1 << 32) | (a0 & 0xffffffffull));
        acc = r == 0 ? vr : (op == 1 ? (vr > acc || acc != acc ? vr : acc) : acc + vr);
    }
    x.seq[col] = s;
    if (lost) {
        *x.status = 1;
        return xchg_value(0x7ff8000000000000ull);
    }
    return acc;
}

// one column: push, then wait.  Called by ONE lane / thread per column; columns are independent of each other.
template <class Mem>
CUDE_XCHG_HD inline double xchg_combine(const XchgArgs& x, int col, double v, int op, Mem mem) {
    const unsigned s = xchg_push(x, col, v, mem);
    return xchg_wait(x, col, s, op, mem.now(), mem);
}

#ifdef __HIPCC__
struct XchgDeviceMem {
    __device__ __forceinline__ void store(unsigned long long* p, unsigned long long w) const {
        __hip_atomic_store(p, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __device__ __forceinline__ unsigned long long load(const unsigned long long* p) const {
        return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __device__ __forceinline__ long long now() const { return wall_clock64(); }
    __device__ __forceinline__ void pause() const { __builtin_amdgcn_s_sleep(2); }
};
__device__ __forceinline__ double xchg_combine(const XchgArgs& x, int col, double v, int op = 0) {
    return xchg_combine(x, col, v, op, XchgDeviceMem{});
}
// two columns at once (col and col + 1): both values written before either is waited for
__device__ __forceinline__ void xchg_combine2(const XchgArgs& x, int col, double& v0, double& v1) {
    const XchgDeviceMem mem{};
    const unsigned s0 = xchg_push(x, col, v0, mem), s1 = xchg_push(x, col + 1, v1, mem);
    const long long t0 = mem.now();
    v0 = xchg_wait(x, col, s0, 0, t0, mem);
    v1 = xchg_wait(x, col + 1, s1, 0, t0, mem);
}
#endif

}  // namespace cude
