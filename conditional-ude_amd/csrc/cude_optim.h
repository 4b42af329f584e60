// Host-side optimisers of the training drivers (no device code): L-BFGS with backtracking line search as a
// RESUMABLE state machine, so that K independent runs can be advanced in lock step with one batched device
// evaluation per round (cude_train_restarts), and the Adam rule vectorised over restarts.
//
// Replaces (reference repo paths; the algorithms themselves are third-party packages, restated from their
// documentation and defaults):
//   Optimization.solve(prob, LBFGS(linesearch = BackTracking()), maxiters)   src/parameter-estimation.jl:179-180,
//                                                                            suppression/src/suppression_model.jl:168
//       Optim.jl L-BFGS: memory m = 10, scaleinvH0 (initial inverse-Hessian scaling s'y / y'y), g_tol = 1e-8
//       (max-norm), alphaguess = InitialStatic(alpha = 1, scaled = false): EVERY line search starts at step 1,
//       a (dx, dg) pair is stored whatever the sign of dx'dg (only dx'dg == 0 resets the history), a direction
//       that is not a descent direction is replaced by -g within the same iteration (reset_search_direction!)
//       LineSearches.BackTracking: c_1 = 1e-4, rho_hi = 0.5, rho_lo = 0.1, order 3 (quadratic, then cubic
//       interpolation), at most 52 halvings to reach a finite value (-log2(eps)), at most 1000 shrinks; running out
//       of shrinks is a LineSearchException: Optim takes the step it has and stops
// Sharded use (cude_train_restarts on a context with a communicator, cude_lbfgs_minimize_sharded): the vector is
// [shared (replicated on every rank); local (this rank's conditional parameters)]; inner products and max-norms
// take their local part through a reducer (sum / max over ranks), so every rank walks the same path.
//   Optimisers.Adam                                                           parameter-estimation.jl:175-176
// The Python mirror (cude/lbfgs.py) states the same algorithm as a generator; tests compare the two.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>

namespace cude {

// sum (op 0) or max (op 1) of `count` doubles over all ranks, in place; returns < 0 on failure
typedef int32_t (*ReduceFn)(double* values, int32_t count, int32_t op, void* user);

class Lbfgs {
public:
    struct Result {
        double f = std::numeric_limits<double>::quiet_NaN();
        int iterations = 0, f_calls = 0;
        bool converged = false;
    };

    // n_shared: leading entries that are replicated on every rank (all n when there is no reducer)
    Lbfgs(const double* x0, int n, int maxiters, int m = 10, double g_tol = 1e-8, int n_shared = -1,
          ReduceFn reduce = nullptr, void* reduce_user = nullptr)
        : n_(n), m_(m), maxiters_(maxiters), n_shared_(reduce ? std::max(0, std::min(n_shared, n)) : n),
          g_tol_(g_tol), reduce_(reduce), reduce_user_(reduce_user), x_(x0, x0 + n), g_(n), d_(n), trial_(x0, x0 + n),
          S_((size_t)m * n), Y_((size_t)m * n), rho_(m) {}

    // the point whose (f, g) the machine is waiting for; nullptr once the run has finished
    const double* pending() const { return done_ ? nullptr : trial_.data(); }
    bool done() const { return done_; }
    bool comm_failed() const { return comm_failed_; }
    const std::vector<double>& x() const { return x_; }
    Result result() const { return {f_, it_, calls_, converged_}; }
    int accepted_steps() const { return accepted_; }     // successful iterations so far (where Optim calls back)
    double current_f() const { return f_; }

    // hand over f and g at pending(); advances to the next request (or finishes)
    void feed(double f, const double* g) {
        if (done_) return;
        switch (state_) {
            case FIRST:
                f_ = f;
                std::copy(g, g + n_, g_.begin());
                calls_ = 1;
                converged_ = std::isfinite(f_) && max_abs(g_.data()) <= g_tol_;
                start_iteration();
                break;
            case LS_FINITE:                       // first trial of a line search, halving until finite
                n_eval_++;
                if (!std::isfinite(f) && ls_it_ < kMaxFinite) {
                    a1_ = a2_;
                    a2_ *= 0.5;
                    ls_it_++;
                    set_trial();
                    break;
                }
                phi1_ = f0_;
                ls_it_ = 0;
                state_ = LS_ARMIJO;
                armijo(f, g);
                break;
            case LS_ARMIJO:
                n_eval_++;
                armijo(f, g);
                break;
        }
    }

private:
    enum State { FIRST, LS_FINITE, LS_ARMIJO };
    static constexpr int kMaxLs = 1000, kMaxFinite = 52;
    static constexpr double kC1 = 1e-4, kRhoHi = 0.5, kRhoLo = 0.1;

    double reduced(double shared, double local, int op) {
        if (!reduce_) return shared;              // n_shared_ == n_: everything was summed as "shared"
        if (reduce_(&local, 1, op, reduce_user_) < 0) comm_failed_ = true;
        return op == 0 ? shared + local : std::max(shared, local);
    }
    double dot(const double* a, const double* b) {
        double s = 0.0, l = 0.0;
        for (int i = 0; i < n_shared_; i++) s += a[i] * b[i];
        for (int i = n_shared_; i < n_; i++) l += a[i] * b[i];
        return reduced(s, l, 0);
    }
    double max_abs(const double* a) {
        double s = 0.0, l = 0.0;
        for (int i = 0; i < n_shared_; i++) s = std::max(s, std::fabs(a[i]));
        for (int i = n_shared_; i < n_; i++) l = std::max(l, std::fabs(a[i]));
        return reduced(s, l, 1);
    }
    // history ring addressed like Optim's: pair `index` (1-based count of stored steps) lives in slot (index-1) % m
    const double* hs(int index) const { return S_.data() + (size_t)((index - 1) % m_) * n_; }
    const double* hy(int index) const { return Y_.data() + (size_t)((index - 1) % m_) * n_; }
    double hr(int index) const { return rho_[(index - 1) % m_]; }

    void set_trial() {
        for (int i = 0; i < n_; i++) trial_[i] = x_[i] + a2_ * d_[i];
    }

    void finish() {
        done_ = true;
        trial_ = x_;
    }

    // top of the main loop: either stop, or compute the direction and request the first line-search trial
    void start_iteration() {
        if (!(it_ < maxiters_) || converged_ || !std::isfinite(f_) || comm_failed_) { finish(); return; }
        pseudo_++;
        // two-loop recursion over the pairs lower..upper (Optim twoloop!)
        const int upper = pseudo_ - 1, lower = std::max(1, pseudo_ - m_);
        std::vector<double> q(g_), alpha(m_ + 1);
        for (int k = upper; k >= lower; k--) {
            alpha[k - lower] = hr(k) * dot(hs(k), q.data());
            const double* y = hy(k);
            for (int i = 0; i < n_; i++) q[i] -= alpha[k - lower] * y[i];
        }
        if (pseudo_ > 1) {                        // scaleinvH0
            const double sc = dot(hs(upper), hy(upper)) / dot(hy(upper), hy(upper));
            for (int i = 0; i < n_; i++) q[i] *= sc;
        }
        for (int k = lower; k <= upper; k++) {
            const double b = hr(k) * dot(hy(k), q.data());
            const double* s = hs(k);
            for (int i = 0; i < n_; i++) q[i] += (alpha[k - lower] - b) * s[i];
        }
        for (int i = 0; i < n_; i++) d_[i] = -q[i];
        // line search set-up
        f0_ = f_;
        dphi0_ = dot(g_.data(), d_.data());
        if (!(dphi0_ < 0)) {                      // corrupted direction: restart from steepest descent (reset_search_direction!)
            pseudo_ = 1;
            for (int i = 0; i < n_; i++) d_[i] = -g_[i];
            dphi0_ = dot(g_.data(), d_.data());
            if (!(dphi0_ < 0)) { finish(); return; }   // zero (or non-finite) gradient
        }
        a1_ = a2_ = 1.0;                          // InitialStatic(alpha = 1)
        phi1_ = f0_;
        n_eval_ = 0;
        ls_it_ = 0;
        state_ = LS_FINITE;
        set_trial();
    }

    void armijo(double f, const double* g) {
        if (f > f0_ + kC1 * a2_ * dphi0_) {       // (NaN compares false: accepted, the main loop then stops)
            ls_it_++;
            if (ls_it_ > kMaxLs) {                // LineSearchException(alpha = the last step tried): Optim's
                calls_ += n_eval_;                // perform_linesearch! takes that step (state.x += alpha s), then
                x_ = trial_;                      // update_state! reports failure and the main loop breaks: the
                f_ = f;                           // result is the last TRIAL point and its value, not the iterate
                std::copy(g, g + n_, g_.begin()); // the search started from
                it_++;
                finish();
                return;
            }
            double a_tmp;
            if (ls_it_ == 1) {
                a_tmp = -(dphi0_ * a2_ * a2_) / (2.0 * (f - f0_ - dphi0_ * a2_));
            } else {
                const double div = 1.0 / (a1_ * a1_ * a2_ * a2_ * (a2_ - a1_));
                const double A = (a1_ * a1_ * (f - f0_ - dphi0_ * a2_) - a2_ * a2_ * (phi1_ - f0_ - dphi0_ * a1_)) * div;
                const double B = (-a1_ * a1_ * a1_ * (f - f0_ - dphi0_ * a2_) + a2_ * a2_ * a2_ * (phi1_ - f0_ - dphi0_ * a1_)) * div;
                if (std::fabs(A) <= 2.220446049250313e-16) {
                    a_tmp = dphi0_ / (2.0 * B);
                } else {
                    const double disc = std::max(B * B - 3.0 * A * dphi0_, 0.0);
                    a_tmp = (-B + std::sqrt(disc)) / (3.0 * A);
                }
            }
            a1_ = a2_;
            // NaNMath.min / NaNMath.max: a NaN candidate is ignored
            double a_new = std::isnan(a_tmp) ? a2_ * kRhoHi : std::min(a_tmp, a2_ * kRhoHi);
            a_new = std::max(a_new, a2_ * kRhoLo);
            a2_ = a_new;
            phi1_ = f;
            set_trial();
            return;
        }
        // accepted: x <- x + alpha d, history update, convergence test
        calls_ += n_eval_;
        std::vector<double> s(n_), y(n_);
        double moved = 0.0, moved_l = 0.0;
        for (int i = 0; i < n_; i++) {
            s[i] = a2_ * d_[i];
            y[i] = g[i] - g_[i];
            const double xn = x_[i] + s[i];
            (i < n_shared_ ? moved : moved_l) = std::max(i < n_shared_ ? moved : moved_l, std::fabs(xn - x_[i]));
            x_[i] = xn;
        }
        const bool x_same = reduced(moved, moved_l, 1) == 0.0;
        const double f_prev = f_;
        f_ = f;
        std::copy(g, g + n_, g_.begin());
        it_++;
        accepted_++;
        f_flat_ = (std::fabs(f_prev - f_) == 0.0) ? f_flat_ + 1 : 0;       // successive_f_tol = 1
        converged_ = x_same || max_abs(g_.data()) <= g_tol_ || f_flat_ > 1;
        if (!converged_) {                        // update_h!
            const double sy = dot(s.data(), y.data());
            const double rho = 1.0 / sy;
            if (std::isinf(rho)) {
                pseudo_ = 0;
            } else {
                const int slot = (pseudo_ - 1) % m_;
                std::copy(s.begin(), s.end(), S_.begin() + (size_t)slot * n_);
                std::copy(y.begin(), y.end(), Y_.begin() + (size_t)slot * n_);
                rho_[slot] = rho;
            }
        }
        start_iteration();
    }

    int n_, m_, maxiters_, n_shared_;
    double g_tol_;
    ReduceFn reduce_;
    void* reduce_user_;
    std::vector<double> x_, g_, d_, trial_, S_, Y_, rho_;
    int pseudo_ = 0;
    double f_ = std::numeric_limits<double>::quiet_NaN();
    int it_ = 0, calls_ = 0, accepted_ = 0, f_flat_ = 0;
    bool converged_ = false, done_ = false, comm_failed_ = false;
    State state_ = FIRST;
    double f0_ = 0, dphi0_ = 0, a1_ = 0, a2_ = 0, phi1_ = 0;
    int ls_it_ = 0, n_eval_ = 0;
};

// Optimisers.jl Adam over a flat vector: returns nothing, updates x, m, v in place (t = 1-based step count)
inline void adam_update(double* x, const double* g, double* m, double* v, int64_t n, int t, double lr, double b1 = 0.9,
                        double b2 = 0.999, double eps = 1e-8) {
    const double c1 = 1.0 - std::pow(b1, t), c2 = 1.0 - std::pow(b2, t);
    for (int64_t i = 0; i < n; i++) {
        m[i] = b1 * m[i] + (1.0 - b1) * g[i];
        v[i] = b2 * v[i] + (1.0 - b2) * g[i] * g[i];
        x[i] -= lr * (m[i] / c1) / (std::sqrt(v[i] / c2) + eps);
    }
}

}  // namespace cude
