// Host-only driver for certifiedgpt_amd/csrc/stats.h, built by tests/test_stats_sanitizers.py with
// -fsanitize=address,undefined (GPU sanitizers are not available on the pool; the statistics are host code anyway).
// Protocol on stdin, one case per line:
//   C K n alpha sigma  sel[0..K)  est[0..K)   -> "label radius"
//   P K alpha  counts[0..K)                   -> "label"
//   L NA N alpha                              -> "bound"
//   B x n p                                   -> "pvalue"
//   Q p                                       -> "norm_ppf"
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../certifiedgpt_amd/csrc/stats.h"

int main() {
    char kind;
    while (std::scanf(" %c", &kind) == 1) {
        if (kind == 'C') {
            int K; long long n; double alpha, sigma;
            if (std::scanf("%d %lld %lf %lf", &K, &n, &alpha, &sigma) != 4) return 2;
            std::vector<int64_t> sel(K), est(K);
            for (auto& v : sel) { long long t; if (std::scanf("%lld", &t) != 1) return 2; v = t; }
            for (auto& v : est) { long long t; if (std::scanf("%lld", &t) != 1) return 2; v = t; }
            int label; double radius;
            cgpt_stats::certify_from_counts(sel.data(), est.data(), K, n, alpha, sigma, &label, &radius);
            std::printf("%d %.17g\n", label, radius);
        } else if (kind == 'P') {
            int K; double alpha;
            if (std::scanf("%d %lf", &K, &alpha) != 2) return 2;
            std::vector<int64_t> c(K);
            for (auto& v : c) { long long t; if (std::scanf("%lld", &t) != 1) return 2; v = t; }
            std::printf("%d\n", cgpt_stats::predict_from_counts(c.data(), K, alpha));
        } else if (kind == 'L') {
            long long NA, N; double alpha;
            if (std::scanf("%lld %lld %lf", &NA, &N, &alpha) != 3) return 2;
            std::printf("%.17g\n", cgpt_stats::cp_lower_bound(NA, N, alpha));
        } else if (kind == 'B') {
            long long x, n; double p;
            if (std::scanf("%lld %lld %lf", &x, &n, &p) != 3) return 2;
            std::printf("%.17g\n", cgpt_stats::binom_test_two_sided(x, n, p));
        } else if (kind == 'Q') {
            double p;
            if (std::scanf("%lf", &p) != 1) return 2;
            std::printf("%.17g\n", cgpt_stats::norm_ppf(p));
        } else {
            return 3;
        }
    }
    return 0;
}
