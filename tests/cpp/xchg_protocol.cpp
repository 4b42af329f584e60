// CPU-only driver for the peer-write exchange's protocol (conditional-ude_amd/csrc/cude_xchg.h: xchg_combine over a
// memory policy).  Host threads stand in for the ranks of a node: every thread owns a mailbox, sees all mailboxes, and
// runs the SAME function the kernels run, with std::atomic accesses as the policy.  Built by tests/test_xchg_protocol.py
// (plain and with -fsanitize=thread).  Checks, for 1, 2, 3 and 8 ranks and several columns:
//   * every rank ends every round with bit-identical sums, equal to the sum taken in rank order;
//   * that holds with ranks arriving in random order, one rank a whole round late, and a rank running one round ahead
//     (the two-parity argument in the header);
//   * max works like sum;
//   * a rank whose peer never writes gives up after its time limit, sets the status word, returns NaN, and has still
//     advanced its sequence counter.
// Prints one line per case: name, ok flag.
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

#include "cude_xchg.h"

using cude::XchgArgs;

struct HostMem {
    void store(unsigned long long* p, unsigned long long w) const {
        reinterpret_cast<std::atomic<unsigned long long>*>(p)->store(w, std::memory_order_relaxed);
    }
    unsigned long long load(const unsigned long long* p) const {
        return reinterpret_cast<const std::atomic<unsigned long long>*>(p)->load(std::memory_order_relaxed);
    }
    long long now() const {
        return std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
    }
    void pause() const { std::this_thread::yield(); }
};

struct Node {
    int n, cols;
    std::vector<std::vector<unsigned long long>> box;
    std::vector<std::vector<unsigned>> seq;
    std::vector<int> status;
    Node(int n_, int cols_) : n(n_), cols(cols_), box(n_), seq(n_), status(n_, 0) {
        for (int r = 0; r < n; r++) {
            box[r].assign((size_t)2 * n * cols * 2, 0ull);
            seq[r].assign(cols, 0u);
        }
    }
    XchgArgs args(int rank, long long timeout_us) {
        XchgArgs x{};
        for (int r = 0; r < n; r++) x.peers[r] = box[r].data();
        x.seq = seq[rank].data();
        x.status = &status[rank];
        x.n_ranks = n; x.rank = rank; x.cols = cols;
        x.timeout = timeout_us;
        return x;
    }
};

static double contribution(int rank, int col, int round) {
    // values whose sum depends on the order of the additions
    return std::ldexp(1.0 + 0.37 * rank + 0.011 * col, (rank * 7 + round) % 40 - 20) * ((rank + col + round) % 3 == 0 ? -1.0 : 1.0);
}

// `rounds` exchanges of every column by every rank; delay(rank, round) microseconds before a rank's round
template <class Delay>
static bool run_case(int n, int cols, int rounds, int op, Delay delay) {
    Node node(n, cols);
    std::vector<std::vector<double>> got(n, std::vector<double>((size_t)rounds * cols));
    std::vector<std::thread> th;
    for (int r = 0; r < n; r++)
        th.emplace_back([&, r] {
            const XchgArgs x = node.args(r, 20 * 1000 * 1000);
            for (int k = 0; k < rounds; k++) {
                const int d = delay(r, k);
                if (d > 0) std::this_thread::sleep_for(std::chrono::microseconds(d));
                for (int c = 0; c < cols; c++)
                    got[r][(size_t)k * cols + c] = cude::xchg_combine(x, c, contribution(r, c, k), op, HostMem{});
            }
        });
    for (auto& t : th) t.join();
    bool ok = true;
    for (int k = 0; k < rounds; k++)
        for (int c = 0; c < cols; c++) {
            double ref = contribution(0, c, k);
            for (int r = 1; r < n; r++) ref = op == 1 ? std::fmax(ref, contribution(r, c, k)) : ref + contribution(r, c, k);
            for (int r = 0; r < n; r++)
                if (std::memcmp(&got[r][(size_t)k * cols + c], &ref, sizeof(double)) != 0) ok = false;
        }
    for (int r = 0; r < n; r++) {
        if (node.status[r] != 0) ok = false;
        for (int c = 0; c < cols; c++)
            if (node.seq[r][c] != (unsigned)rounds) ok = false;
    }
    return ok;
}

int main() {
    std::mt19937 gen(12345);
    for (int n : {1, 2, 3, 8}) {
        std::vector<int> jitter((size_t)n * 64);
        for (auto& j : jitter) j = (int)(gen() % 200);
        char name[64];
        std::snprintf(name, sizeof name, "sum_ranks%d", n);
        std::printf("%s %d\n", name, (int)run_case(n, 5, 64, 0, [&](int r, int k) { return jitter[(size_t)r * 64 + k]; }));
        std::snprintf(name, sizeof name, "max_ranks%d", n);
        std::printf("%s %d\n", name, (int)run_case(n, 3, 32, 1, [&](int r, int k) { return jitter[(size_t)r * 64 + k] / 4; }));
        // the last rank is 3 ms late in every round (everybody waits for it, then it is the one that runs ahead)
        std::snprintf(name, sizeof name, "late_rank%d", n);
        std::printf("%s %d\n", name, (int)run_case(n, 4, 12, 0, [&](int r, int) { return r == n - 1 ? 3000 : 0; }));
        // rank 0 is slow to come back every other round: its peers are a round ahead of what it still reads
        std::snprintf(name, sizeof name, "ahead_ranks%d", n);
        std::printf("%s %d\n", name, (int)run_case(n, 4, 40, 0, [&](int r, int k) { return (r == 0 && (k & 1)) ? 500 : 0; }));
    }
    {   // a peer that never writes: bounded wait, status word, NaN, counter advanced
        Node node(2, 3);
        XchgArgs x = node.args(0, 50 * 1000);          // 50 ms
        const auto t0 = std::chrono::steady_clock::now();
        const double v = cude::xchg_combine(x, 1, 42.0, 0, HostMem{});
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        const bool ok = std::isnan(v) && node.status[0] == 1 && node.seq[0][1] == 1u && ms >= 50.0 && ms < 5000.0;
        std::printf("timeout %d\n", (int)ok);
        // the peer shows up afterwards with the same sequence number: the NEXT round pairs up again
        node.status[0] = 0;
        XchgArgs y = node.args(1, 20 * 1000 * 1000);
        std::thread t([&] { (void)cude::xchg_combine(y, 1, 1.0, 0, HostMem{}); });
        t.join();      // rank 1's round 1 completes: rank 0's word of round 1 is in its mailbox
        std::thread a([&] { (void)cude::xchg_combine(y, 1, 2.5, 0, HostMem{}); });
        const double w = cude::xchg_combine(x, 1, 4.0, 0, HostMem{});
        a.join();
        std::printf("recovered %d\n", (int)(w == 6.5 && node.status[0] == 0));
    }
    return 0;
}
