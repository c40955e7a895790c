// Host-side plan: everything that depends only on the direction grid and the phase
// matrices, built once per sweep and shared by every column (no GPU needed).
#pragma once
#include <vector>

namespace sosrt {

constexpr double kMuThreshold = 0.01;           // gva:5  MU_THRESHOLD
constexpr double kMuVerySmall = 0.001;          // gva:7  MU_VERY_SMALL_THRESHOLD
constexpr double kMuExtreme = 1e-8;             // gva:6  MU_EXTREME_THRESHOLD (same branch as kMuVerySmall, In_limit:79-93)
constexpr int kFixMaxIdx = 64;                  // rewritten angles next to mu=0-: int(0.06 N) <= 64 -> N <= 1066
constexpr int kFixMaxSrc = 5;                   // In_limit:118  n_points = min(5, idx)

// Number of downward angles rewritten next to mu = 0- (I1_In:124-127, spec:342-345).
int fix_count(double tau_ref, int N);
// bucket 0..3 of the piecewise-constant factor above
int fix_bucket(double tau_ref);
int fix_count_of_bucket(int bucket, int N);

// Linear map that replaces improved_limit_mu_down (In_limit:113-141): for a row of
// downward radiances, rewritten lane N-1-i = sum_j C[i*ns + j] * row[s0 + j].
struct FixTable {
    int idx = 0, s0 = 0, ns = 0;
    std::vector<double> C;
};

struct Plan {
    int N = 0, D = 0;
    std::vector<double> mu;          // [D]
    std::vector<double> w;           // [D]   np.trapz weights on mu, including the zero-width mu=0 interval (I1_In:73)
    std::vector<double> wflux_dn;    // [N]   trapz weights on mu[:N] times mu   (graphe:157)
    std::vector<double> wflux_up;    // [N]   trapz weights on mu[N:] times mu   (graphe:158)
    std::vector<int> small_lanes;    // downward lanes m <= N-2 with |mu| < MU_THRESHOLD (spec:333)
    FixTable fix[4];                 // one table per tau_ref bucket

    void set_grid(int N, const double* mu);
    // W[k*D + m] = w_k * P[m*D + (D-1-k)]  so that  trapz(P[:, ::-1] * x, mu, axis=1) = x @ W  (I1_In:73)
    void fold(const double* P, std::vector<double>& W) const;
    FixTable make_fix_table(int idx) const;
};

}  // namespace sosrt
