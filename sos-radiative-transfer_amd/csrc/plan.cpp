#include "plan.hpp"

#include <cmath>

namespace sosrt {

int fix_bucket(double tau_ref) {
    if (tau_ref <= 0.0625) return 0;
    if (tau_ref <= 1) return 1;
    if (tau_ref < 4) return 2;
    return 3;
}

int fix_count_of_bucket(int bucket, int N) {
    static const double c[4] = {0.005, 0.02, 0.04, 0.06};
    return (int)(c[bucket] * (double)N);   // Python: int(c * nb_angles)
}

int fix_count(double tau_ref, int N) { return fix_count_of_bucket(fix_bucket(tau_ref), N); }

void Plan::set_grid(int N_, const double* mu_) {
    N = N_;
    D = 2 * N_;
    mu.assign(mu_, mu_ + D);
    w.assign(D, 0.0);
    for (int k = 0; k + 1 < D; ++k) {
        double d = mu[k + 1] - mu[k];
        w[k] += d / 2;
        w[k + 1] += d / 2;
    }
    wflux_dn.assign(N, 0.0);
    wflux_up.assign(N, 0.0);
    for (int k = 0; k + 1 < N; ++k) {
        double d = mu[k + 1] - mu[k];
        wflux_dn[k] += d / 2;
        wflux_dn[k + 1] += d / 2;
        double u = mu[N + k + 1] - mu[N + k];
        wflux_up[k] += u / 2;
        wflux_up[k + 1] += u / 2;
    }
    for (int k = 0; k < N; ++k) {
        wflux_dn[k] *= mu[k];
        wflux_up[k] *= mu[N + k];
    }
    small_lanes.clear();
    for (int m = 0; m <= N - 2; ++m)
        if (std::fabs(mu[m]) < kMuThreshold) small_lanes.push_back(m);
    for (int b = 0; b < 4; ++b) fix[b] = make_fix_table(fix_count_of_bucket(b, N));
}

void Plan::fold(const double* P, std::vector<double>& W) const {
    W.assign((size_t)D * D, 0.0);
    for (int k = 0; k < D; ++k)
        for (int m = 0; m < D; ++m) W[(size_t)k * D + m] = w[k] * P[(size_t)m * D + (D - 1 - k)];
}

namespace {
void inv3(const long double G[3][3], long double R[3][3]) {
    long double a = G[0][0], b = G[0][1], c = G[0][2], d = G[1][0], e = G[1][1], f = G[1][2], g = G[2][0],
                h = G[2][1], i = G[2][2];
    long double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
    R[0][0] = (e * i - f * h) / det; R[0][1] = (c * h - b * i) / det; R[0][2] = (b * f - c * e) / det;
    R[1][0] = (f * g - d * i) / det; R[1][1] = (a * i - c * g) / det; R[1][2] = (c * d - a * f) / det;
    R[2][0] = (d * h - e * g) / det; R[2][1] = (b * g - a * h) / det; R[2][2] = (a * e - b * d) / det;
}
}  // namespace

// In_limit:113-141.  The reference fits a polynomial to n = min(5, idx) fixed angles and evaluates
// it at the idx rewritten angles; for a fixed grid that is a fixed linear map of the n source
// values.  The least-squares case (n >= 3, degree 2) is solved in centred / scaled coordinates
// in extended precision, which reproduces np.polyfit to its own rounding error.
FixTable Plan::make_fix_table(int idx) const {
    FixTable t;
    t.idx = idx;
    if (idx <= 0) return t;
    const int n = idx < kFixMaxSrc ? idx : kFixMaxSrc;
    auto X = [&](int lane) { return (long double)mu[lane]; };
    if (n < 2) {
        // line through In_down[-idx-2], In_down[-idx-1]   (In_limit:120-124)
        t.s0 = N - idx - 2;
        t.ns = 2;
        t.C.assign((size_t)idx * 2, 0.0);
        long double x3 = X(t.s0), x2 = X(t.s0 + 1);
        for (int i = 0; i < idx; ++i) {
            long double c3 = (X(N - 1 - i) - x2) / (x3 - x2);
            t.C[i * 2 + 0] = (double)c3;
            t.C[i * 2 + 1] = (double)(1 - c3);
        }
        return t;
    }
    t.s0 = N - idx - n;
    t.ns = n;
    t.C.assign((size_t)idx * n, 0.0);
    if (n == 2) {
        // line through the first and last source point (In_limit:139-141)
        long double x0 = X(t.s0), x1 = X(t.s0 + 1);
        for (int i = 0; i < idx; ++i) {
            long double c1 = (X(N - 1 - i) - x0) / (x1 - x0);
            t.C[i * 2 + 0] = (double)(1 - c1);
            t.C[i * 2 + 1] = (double)c1;
        }
        return t;
    }
    long double xb = 0;
    for (int j = 0; j < n; ++j) xb += X(t.s0 + j);
    xb /= n;
    long double h = X(t.s0 + 1) - X(t.s0);
    long double u[kFixMaxSrc], V[kFixMaxSrc][3];
    long double G[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, Gi[3][3];
    for (int j = 0; j < n; ++j) {
        u[j] = (X(t.s0 + j) - xb) / h;
        V[j][0] = 1; V[j][1] = u[j]; V[j][2] = u[j] * u[j];
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) G[a][b] += V[j][a] * V[j][b];
    }
    inv3(G, Gi);
    for (int i = 0; i < idx; ++i) {
        long double ue = (X(N - 1 - i) - xb) / h;
        long double e[3] = {1, ue, ue * ue};
        for (int j = 0; j < n; ++j) {
            long double s = 0;
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) s += e[a] * Gi[a][b] * V[j][b];
            t.C[(size_t)i * n + j] = (double)s;
        }
    }
    return t;
}

}  // namespace sosrt
