// capi_stats.cpp -- host statistics entry points of the C-ABI (include/cgpt.h).
#include "../../include/cgpt.h"
#include "common_host.h"
#include "stats.h"

extern "C" {

cgpt_status cgpt_certify_from_counts(const int64_t* counts_selection, const int64_t* counts_estimation,
                                     int32_t num_classes, int64_t n, double alpha, double sigma,
                                     int32_t* label_out, double* radius_out) {
    if (!counts_selection || !counts_estimation || !label_out || !radius_out || num_classes < 1 || n < 1 ||
        !(alpha > 0.0 && alpha < 1.0))
        return cgpt_fail(CGPT_ERR_INVALID, "cgpt_certify_from_counts: bad argument");
    int label; double radius;
    cgpt_stats::certify_from_counts(counts_selection, counts_estimation, num_classes, n, alpha, sigma, &label, &radius);
    *label_out = label; *radius_out = radius;
    return CGPT_OK;
}

cgpt_status cgpt_certify_many_from_counts(const int64_t* counts, int64_t num_images, int32_t num_classes, int64_t n, double alpha,
                                          double sigma, int32_t* labels_out, double* radii_out) {
    if (!counts || !labels_out || !radii_out || num_images < 0 || num_classes < 1 || n < 1 || !(alpha > 0.0 && alpha < 1.0))
        return cgpt_fail(CGPT_ERR_INVALID, "cgpt_certify_many_from_counts: bad argument");
    for (int64_t i = 0; i < num_images; ++i) {
        const int64_t* sel = counts + i * 2 * (int64_t)num_classes;
        int label; double radius;
        cgpt_stats::certify_from_counts(sel, sel + num_classes, num_classes, n, alpha, sigma, &label, &radius);
        labels_out[i] = label; radii_out[i] = radius;
    }
    return CGPT_OK;
}

cgpt_status cgpt_predict_from_counts(const int64_t* counts, int32_t num_classes, double alpha, int32_t* label_out) {
    // the reference indexes top2[1] (smoothing.py:75): it needs at least two classes
    if (!counts || !label_out || num_classes < 2 || !(alpha > 0.0 && alpha < 1.0))
        return cgpt_fail(CGPT_ERR_INVALID, "cgpt_predict_from_counts: bad argument (num_classes >= 2 required)");
    *label_out = cgpt_stats::predict_from_counts(counts, num_classes, alpha);
    return CGPT_OK;
}

double cgpt_lower_confidence_bound(int64_t NA, int64_t N, double alpha) { return cgpt_stats::cp_lower_bound(NA, N, alpha); }
double cgpt_binom_test(int64_t x, int64_t n, double p) { return cgpt_stats::binom_test_two_sided(x, n, p); }
double cgpt_norm_ppf(double p) { return cgpt_stats::norm_ppf(p); }

}  // extern "C"
