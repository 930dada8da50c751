// spherical_sfm_amd -- SphericalEstimator over the C ABI (see spherical_estimator.h).
#include "spherical_estimator.h"
#include <cstdlib>
#include <iostream>

namespace sphericalsfm {

ssfm_ctx* default_context() {
    static ssfm_ctx* ctx = nullptr;
    if (!ctx && ssfm_ctx_create(-1, nullptr, &ctx) != SSFM_OK) { std::cout << "error: " << ssfm_last_error(nullptr) << "\n"; std::exit(1); }
    return ctx;
}

static void die(ssfm_ctx* ctx) { std::cout << "error: " << ssfm_last_error(ctx) << "\n"; std::exit(1); }

SphericalEstimator::SphericalEstimator(const RayPairList& _correspondences, const bool _use_poly_solver, const bool _inward)
    : correspondences(_correspondences), use_poly_solver(_use_poly_solver), inward(_inward), handle(nullptr), have_cache(false) {
    const size_t n = correspondences.size();
    std::vector<double> u(3 * n), v(3 * n);
    for (size_t i = 0; i < n; i++) for (int k = 0; k < 3; k++) { u[3 * i + k] = correspondences[i].first[k]; v[3 * i + k] = correspondences[i].second[k]; }
    if (ssfm_estimator_create(default_context(), (int32_t)n, u.data(), v.data(), use_poly_solver ? 1 : 0, inward ? 1 : 0, &handle) != SSFM_OK) die(default_context());
}
SphericalEstimator::~SphericalEstimator() { ssfm_estimator_destroy(handle); }

int SphericalEstimator::MinimalSolver(const std::vector<int>& sample, std::vector<Mat3>* Es) const {
    double buf[36]; int32_t k = 0;
    if (ssfm_estimator_minimal_solver(handle, sample.data(), (int32_t)sample.size(), buf, &k) != SSFM_OK) die(default_context());
    Es->resize(k);
    for (int m = 0; m < k; m++) for (int q = 0; q < 9; q++) (*Es)[m][q] = buf[9 * m + q];
    return k;
}

int SphericalEstimator::NonMinimalSolver(const std::vector<int>& sample, Mat3* E) const {
    int32_t ok = 0;
    if (ssfm_estimator_non_minimal_solver(handle, sample.data(), (int32_t)sample.size(), E->data(), &ok) != SSFM_OK) die(default_context());
    return ok;
}

double SphericalEstimator::EvaluateModelOnPoint(const Mat3& E, int i) const {
    if ((size_t)i > correspondences.size()) std::cout << "error: " << i << " / " << correspondences.size() << std::endl;    // src/spherical_estimator.cpp:69
    if (!have_cache || E != cached_model) {
        cached_errors.resize(correspondences.size());
        if (ssfm_estimator_evaluate_model(handle, E.data(), cached_errors.data()) != SSFM_OK) die(default_context());
        cached_model = E; have_cache = true;
    }
    return cached_errors[i];
}

void SphericalEstimator::LeastSquares(const std::vector<int>& sample, Mat3* E) const {
    if (ssfm_estimator_least_squares(handle, sample.data(), (int32_t)sample.size(), E->data()) != SSFM_OK) die(default_context());
}

void SphericalEstimator::Decompose(const Mat3& E, const std::vector<int>& /*inliers*/, Mat3* R, Vec3* t) const {
    double tv[3];
    if (ssfm_estimator_decompose(handle, E.data(), R->data(), tv) != SSFM_OK) die(default_context());
    *t = Vec3(tv[0], tv[1], tv[2]);
}

}  // namespace sphericalsfm
