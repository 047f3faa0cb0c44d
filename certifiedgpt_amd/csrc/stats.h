// stats.h -- float64 statistics of Smooth.certify / Smooth.predict.
//
// Replaces the three third-party calls on the reference hot path
// (randomized_smoothing/smoothing.py):
//   :117 statsmodels proportion_confint(NA, N, alpha=2*alpha, method="beta")[0]
//        == Beta.ppf(alpha; NA, N-NA+1), 0 when NA == 0            -> cp_lower_bound
//   :76  scipy.stats.binom_test(x, n, p) (two-sided, scipy 1.7)      -> binom_test_two_sided
//   :55  scipy.stats.norm.ppf                                        -> norm_ppf
// Written as host/device inline functions so the same arithmetic can run in a finalize kernel.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define CGPT_HD __host__ __device__
#else
#define CGPT_HD
#endif

namespace cgpt_stats {

// Continued fraction of the incomplete beta function (modified Lentz).
CGPT_HD inline double beta_cf(double a, double b, double x) {
    const double tiny = 1e-300, eps = 1e-16;
    const double qab = a + b, qap = a + 1.0, qam = a - 1.0;
    double c = 1.0, d = 1.0 - qab * x / qap;
    if (fabs(d) < tiny) d = tiny;
    d = 1.0 / d;
    double h = d;
    for (int m = 1; m <= 20000; ++m) {
        const double m2 = 2.0 * m;
        double aa = m * (b - m) * x / ((qam + m2) * (a + m2));
        d = 1.0 + aa * d; if (fabs(d) < tiny) d = tiny;
        c = 1.0 + aa / c; if (fabs(c) < tiny) c = tiny;
        d = 1.0 / d; h *= d * c;
        aa = -(a + m) * (qab + m) * x / ((a + m2) * (qap + m2));
        d = 1.0 + aa * d; if (fabs(d) < tiny) d = tiny;
        c = 1.0 + aa / c; if (fabs(c) < tiny) c = tiny;
        d = 1.0 / d;
        const double del = d * c;
        h *= del;
        if (fabs(del - 1.0) < eps) break;
    }
    return h;
}

// log of x^a (1-x)^b / B(a,b)
CGPT_HD inline double beta_log_front(double a, double b, double x) {
    return a * log(x) + b * log1p(-x) + lgamma(a + b) - lgamma(a) - lgamma(b);
}

// Regularised incomplete beta I_x(a, b).
CGPT_HD inline double beta_inc(double a, double b, double x) {
    if (x <= 0.0) return 0.0;
    if (x >= 1.0) return 1.0;
    const double front = exp(beta_log_front(a, b, x));
    if (x < (a + 1.0) / (a + b + 2.0)) return front * beta_cf(a, b, x) / a;
    return 1.0 - front * beta_cf(b, a, 1.0 - x) / b;
}

// Solve I_x(a,b) = q for x (Beta.ppf): bracketed Newton on the monotone CDF.
CGPT_HD inline double beta_ppf(double q, double a, double b) {
    if (q <= 0.0) return 0.0;
    if (q >= 1.0) return 1.0;
    if (b == 1.0) return exp(log(q) / a);           // I_x(a,1) = x^a
    if (a == 1.0) return -expm1(log1p(-q) / b);     // I_x(1,b) = 1-(1-x)^b
    double lo = 0.0, hi = 1.0;
    double x = a / (a + b);                          // start at the mean
    for (int it = 0; it < 200; ++it) {
        const double f = beta_inc(a, b, x) - q;
        if (f > 0.0) hi = x; else lo = x;
        // pdf = x^(a-1) (1-x)^(b-1) / B(a,b)
        const double lpdf = beta_log_front(a, b, x) - log(x) - log1p(-x);
        const double pdf = exp(lpdf);
        double xn = x - f / pdf;
        if (!(xn > lo && xn < hi) || !(pdf > 0.0) || isinf(pdf)) xn = 0.5 * (lo + hi);
        if (fabs(xn - x) <= 4e-16 * x || hi - lo <= 2e-16 * hi) { x = xn; break; }
        x = xn;
    }
    return x;
}

// Smooth._lower_confidence_bound, smoothing.py:107-117.
CGPT_HD inline double cp_lower_bound(int64_t NA, int64_t N, double alpha) {
    if (NA <= 0) return 0.0;                          // statsmodels: ci_low = 0 when count == 0
    return beta_ppf(alpha, (double)NA, (double)(N - NA) + 1.0);
}

CGPT_HD inline double binom_logpmf(int64_t k, int64_t n, double p) {
    if (k < 0 || k > n) return -INFINITY;
    double lp = lgamma((double)n + 1.0) - lgamma((double)k + 1.0) - lgamma((double)(n - k) + 1.0);
    if (k > 0) lp += (double)k * log(p);
    if (n - k > 0) lp += (double)(n - k) * log1p(-p);
    return lp;
}
CGPT_HD inline double binom_pmf(int64_t k, int64_t n, double p) { return exp(binom_logpmf(k, n, p)); }

// P(X <= k) and P(X > k) as sums of positive pmf terms (smallest first: no cancellation).
CGPT_HD inline double binom_cdf(int64_t k, int64_t n, double p) {
    if (k < 0) return 0.0;
    if (k >= n) return 1.0;
    double s = 0.0;
    if ((double)k <= p * (double)n) { for (int64_t i = 0; i <= k; ++i) s += binom_pmf(i, n, p); return s; }
    for (int64_t i = n; i > k; --i) s += binom_pmf(i, n, p);
    return 1.0 - s;
}
CGPT_HD inline double binom_sf(int64_t k, int64_t n, double p) {
    if (k < 0) return 1.0;
    if (k >= n) return 0.0;
    double s = 0.0;
    if ((double)k >= p * (double)n) { for (int64_t i = n; i > k; --i) s += binom_pmf(i, n, p); return s; }
    for (int64_t i = 0; i <= k; ++i) s += binom_pmf(i, n, p);
    return 1.0 - s;
}

// scipy 1.7 stats.binom_test(x, n, p, alternative='two-sided') ("algorithm from R's binom.test").
CGPT_HD inline double binom_test_two_sided(int64_t x, int64_t n, double p) {
    const double d = binom_pmf(x, n, p);
    const double rerr = 1.0 + 1e-7;
    const double pn = p * (double)n;
    double pval;
    if ((double)x == pn) {
        pval = 1.0;
    } else if ((double)x < pn) {
        int64_t y = 0;
        for (int64_t i = (int64_t)ceil(pn); i <= n; ++i) y += (binom_pmf(i, n, p) <= d * rerr) ? 1 : 0;
        pval = binom_cdf(x, n, p) + binom_sf(n - y, n, p);
    } else {
        int64_t y = 0;
        for (int64_t i = 0; i <= (int64_t)floor(pn); ++i) y += (binom_pmf(i, n, p) <= d * rerr) ? 1 : 0;
        pval = binom_cdf(y - 1, n, p) + binom_sf(x - 1, n, p);
    }
    return pval < 1.0 ? pval : 1.0;
}

// Phi^-1: Acklam's rational start + Halley refinement on the tail that keeps precision.
CGPT_HD inline double norm_ppf(double p) {
    if (p <= 0.0) return -INFINITY;
    if (p >= 1.0) return INFINITY;
    if (p == 0.5) return 0.0;
    const bool upper = p > 0.5;
    const double q = upper ? 1.0 - p : p;            // lower-tail mass, q in (0, 0.5)
    const double a[6] = {-3.969683028665376e+01, 2.209460984245205e+02, -2.759285104469687e+02,
                         1.383577518672690e+02, -3.066479806614716e+01, 2.506628277459239e+00};
    const double b[5] = {-5.447609879822406e+01, 1.615858368580409e+02, -1.556989798598866e+02,
                         6.680131188771972e+01, -1.328068155288572e+01};
    const double c[6] = {-7.784894002430293e-03, -3.223964580411365e-01, -2.400758277161838e+00,
                         -2.549732539343734e+00, 4.374664141464968e+00, 2.938163982698783e+00};
    const double dd[4] = {7.784695709041462e-03, 3.224671290700398e-01, 2.445134137142996e+00,
                          3.754408661907416e+00};
    double x;
    if (q < 0.02425) {
        const double t = sqrt(-2.0 * log(q));
        x = (((((c[0] * t + c[1]) * t + c[2]) * t + c[3]) * t + c[4]) * t + c[5]) /
            ((((dd[0] * t + dd[1]) * t + dd[2]) * t + dd[3]) * t + 1.0);
    } else {
        const double t = q - 0.5, r = t * t;
        x = (((((a[0] * r + a[1]) * r + a[2]) * r + a[3]) * r + a[4]) * r + a[5]) * t /
            (((((b[0] * r + b[1]) * r + b[2]) * r + b[3]) * r + b[4]) * r + 1.0);
    }
    // x < 0 with Phi(x) ~= q; refine on Phi(x) = erfc(-x/sqrt2)/2 (accurate in the lower tail)
    for (int it = 0; it < 4; ++it) {
        const double e = 0.5 * erfc(-x * 0.70710678118654752440) - q;
        const double u = e * 2.50662827463100050242 * exp(0.5 * x * x);
        const double dx = u / (1.0 + 0.5 * x * u);
        x -= dx;
        if (fabs(dx) <= 1e-16 * fabs(x)) break;
    }
    return upper ? -x : x;
}

// Smooth.certify, smoothing.py:46-56, given the two histograms.
CGPT_HD inline void certify_from_counts(const int64_t* csel, const int64_t* cest, int K, int64_t n, double alpha,
                                        double sigma, int* label, double* radius) {
    int cAHat = 0;                                    // ndarray.argmax(): first maximal index
    for (int i = 1; i < K; ++i) if (csel[i] > csel[cAHat]) cAHat = i;
    const int64_t nA = cest[cAHat];
    const double pABar = cp_lower_bound(nA, n, alpha);
    if (pABar < 0.5) { *label = -1; *radius = 0.0; return; }
    *label = cAHat;
    *radius = sigma * norm_ppf(pABar);
}

// Smooth.predict, smoothing.py:73-79, given the histogram.  Ties cannot change the outcome: a first-place
// tie has p-value 1 (abstain); a second-place tie has equal count2.
CGPT_HD inline int predict_from_counts(const int64_t* counts, int K, double alpha) {
    int i1 = 0;
    for (int i = 1; i < K; ++i) if (counts[i] > counts[i1]) i1 = i;
    int i2 = (i1 == 0) ? 1 : 0;
    for (int i = 0; i < K; ++i) if (i != i1 && counts[i] > counts[i2]) i2 = i;
    const int64_t c1 = counts[i1], c2 = counts[i2];
    if (binom_test_two_sided(c1, c1 + c2, 0.5) > alpha) return -1;
    return i1;
}

}  // namespace cgpt_stats
