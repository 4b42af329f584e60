// Host-side optimisers of the training drivers (no device code): L-BFGS with backtracking line search as a
// RESUMABLE state machine, so that K independent runs can be advanced in lock step with one batched device
// evaluation per round (cude_train_restarts), and the Adam rule vectorised over restarts.
//
// Replaces (reference repo paths; the algorithms themselves are third-party packages, restated from their
// documentation and defaults):
//   Optimization.solve(prob, LBFGS(linesearch = BackTracking()), maxiters)   src/parameter-estimation.jl:179-180,
//                                                                            suppression/src/suppression_model.jl:168
//       Optim.jl L-BFGS: memory m = 10, initial inverse-Hessian scaling s'y / y'y, g_tol = 1e-8 (max-norm)
//       LineSearches.BackTracking: c_1 = 1e-4, rho_hi = 0.5, rho_lo = 0.1, order 3 (quadratic, then cubic
//       interpolation), initial step 1 (first iteration: min(1, 1/|g|)), at most 50 shrinks
//   Optimisers.Adam                                                           parameter-estimation.jl:175-176
// The Python mirror (cude/lbfgs.py) states the same algorithm as a generator; tests compare the two.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>

namespace cude {

class Lbfgs {
public:
    struct Result {
        double f = std::numeric_limits<double>::quiet_NaN();
        int iterations = 0, f_calls = 0;
        bool converged = false;
    };

    Lbfgs(const double* x0, int n, int maxiters, int m = 10, double g_tol = 1e-8)
        : n_(n), m_(m), maxiters_(maxiters), g_tol_(g_tol), x_(x0, x0 + n), g_(n), d_(n), trial_(x0, x0 + n),
          S_((size_t)m * n), Y_((size_t)m * n), rho_(m) {}

    // the point whose (f, g) the machine is waiting for; nullptr once the run has finished
    const double* pending() const { return done_ ? nullptr : trial_.data(); }
    bool done() const { return done_; }
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
            case LS_FINITE:                       // first trial of a line search, shrinking until finite
                n_eval_++;
                if (!std::isfinite(f) && ls_it_ < kMaxLs) {
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
    static constexpr int kMaxLs = 50;
    static constexpr double kC1 = 1e-4, kRhoHi = 0.5, kRhoLo = 0.1;

    double dot(const double* a, const double* b) const {
        double s = 0.0;
        for (int i = 0; i < n_; i++) s += a[i] * b[i];
        return s;
    }
    double max_abs(const double* a) const {
        double s = 0.0;
        for (int i = 0; i < n_; i++) s = std::max(s, std::fabs(a[i]));
        return s;
    }
    const double* hs(int k) const { return S_.data() + (size_t)((head_ + k) % m_) * n_; }     // k-th oldest pair
    const double* hy(int k) const { return Y_.data() + (size_t)((head_ + k) % m_) * n_; }
    double hr(int k) const { return rho_[(head_ + k) % m_]; }

    void set_trial() {
        for (int i = 0; i < n_; i++) trial_[i] = x_[i] + a2_ * d_[i];
    }

    void finish() {
        done_ = true;
        trial_ = x_;
    }

    // top of the main loop: either stop, or compute the direction and request the first line-search trial
    void start_iteration() {
        while (true) {
            if (!(it_ < maxiters_) || converged_ || !std::isfinite(f_)) { finish(); return; }
            // two-loop recursion
            std::vector<double> q(g_), alpha(hist_);
            for (int k = hist_ - 1; k >= 0; k--) {
                alpha[k] = hr(k) * dot(hs(k), q.data());
                const double* y = hy(k);
                for (int i = 0; i < n_; i++) q[i] -= alpha[k] * y[i];
            }
            if (hist_ > 0) {
                const double sc = dot(hs(hist_ - 1), hy(hist_ - 1)) / dot(hy(hist_ - 1), hy(hist_ - 1));
                for (int i = 0; i < n_; i++) q[i] *= sc;
            }
            for (int k = 0; k < hist_; k++) {
                const double b = hr(k) * dot(hy(k), q.data());
                const double* s = hs(k);
                for (int i = 0; i < n_; i++) q[i] += (alpha[k] - b) * s[i];
            }
            for (int i = 0; i < n_; i++) d_[i] = -q[i];
            // line search set-up
            f0_ = f_;
            dphi0_ = dot(g_.data(), d_.data());
            if (!(dphi0_ < 0)) {                  // not a descent direction: the search "fails"
                if (!line_search_failed()) return;
                continue;
            }
            const double alpha0 = hist_ > 0 ? 1.0 : std::min(1.0, 1.0 / std::max(std::sqrt(dot(g_.data(), g_.data())), 1e-300));
            a1_ = a2_ = alpha0;
            phi1_ = f0_;
            n_eval_ = 0;
            ls_it_ = 0;
            state_ = LS_FINITE;
            set_trial();
            return;
        }
    }

    // returns true when the main loop should go on (history reset), false when the run is over
    bool line_search_failed() {
        if (hist_ == 0) { finish(); return false; }
        hist_ = 0;                                // reset to steepest descent once, as Optim does
        head_ = 0;
        it_++;
        return true;
    }

    void armijo(double f, const double* g) {
        if (f > f0_ + kC1 * a2_ * dphi0_) {       // (NaN compares false: accepted, the main loop then stops)
            ls_it_++;
            if (ls_it_ > kMaxLs) {                // no sufficient decrease found (its evaluations are not counted,
                if (line_search_failed()) start_iteration();   // as in the Python statement of the algorithm)
                return;
            }
            double a_tmp;
            if (ls_it_ == 1 || !std::isfinite(phi1_)) {
                a_tmp = -(dphi0_ * a2_ * a2_) / (2.0 * (f - f0_ - dphi0_ * a2_));
            } else {
                const double div = 1.0 / (a1_ * a1_ * a2_ * a2_ * (a2_ - a1_));
                const double A = (a1_ * a1_ * (f - f0_ - dphi0_ * a2_) - a2_ * a2_ * (phi1_ - f0_ - dphi0_ * a1_)) * div;
                const double B = (-a1_ * a1_ * a1_ * (f - f0_ - dphi0_ * a2_) + a2_ * a2_ * a2_ * (phi1_ - f0_ - dphi0_ * a1_)) * div;
                if (std::fabs(A) < 1e-300) {
                    a_tmp = dphi0_ / (2.0 * B);
                } else {
                    const double disc = std::max(B * B - 3.0 * A * dphi0_, 0.0);
                    a_tmp = (-B + std::sqrt(disc)) / (3.0 * A);
                }
            }
            a1_ = a2_;
            if (!std::isfinite(a_tmp)) a_tmp = a2_ * kRhoHi;
            a2_ = std::min(std::max(a_tmp, a2_ * kRhoLo), a2_ * kRhoHi);
            phi1_ = f;
            set_trial();
            return;
        }
        // accepted: x <- x + alpha d, history update, convergence test
        calls_ += n_eval_;
        std::vector<double> s(n_), y(n_);
        for (int i = 0; i < n_; i++) {
            s[i] = a2_ * d_[i];
            y[i] = g[i] - g_[i];
            x_[i] += s[i];
        }
        const double sy = dot(s.data(), y.data());
        if (sy > 1e-300) {                        // keep the pair; beyond m pairs the oldest one is overwritten
            const int slot = hist_ < m_ ? (head_ + hist_) % m_ : head_;
            std::copy(s.begin(), s.end(), S_.begin() + (size_t)slot * n_);
            std::copy(y.begin(), y.end(), Y_.begin() + (size_t)slot * n_);
            rho_[slot] = 1.0 / sy;
            if (hist_ < m_) hist_++;
            else head_ = (head_ + 1) % m_;
        }
        const double f_prev = f_;
        f_ = f;
        std::copy(g, g + n_, g_.begin());
        it_++;
        accepted_++;
        converged_ = max_abs(g_.data()) <= g_tol_ || std::fabs(f_prev - f_) == 0.0;
        start_iteration();
    }

    int n_, m_, maxiters_;
    double g_tol_;
    std::vector<double> x_, g_, d_, trial_, S_, Y_, rho_;
    int hist_ = 0, head_ = 0;
    double f_ = std::numeric_limits<double>::quiet_NaN();
    int it_ = 0, calls_ = 0, accepted_ = 0;
    bool converged_ = false, done_ = false;
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
