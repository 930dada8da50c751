// spherical_sfm_amd -- mirror of sphericalsfm::SphericalEstimator (include/sphericalsfm/spherical_estimator.h:8-35,
// src/spherical_estimator.cpp:67-164) over the C ABI (include/ssfm.h: ssfm_estimator_*): same constructor, same seven virtuals + Decompose,
// every one of them evaluated by the HIP library on rays that stay resident on the device.  EvaluateModelOnPoint(E, i) is called for
// i = 0 .. n-1 with the same model by every RANSAC driver (ScoreModel / GetInliers), so the errors of all rays are fetched in one launch
// when a new model arrives and served from that vector afterwards.
#pragma once
#include <utility>
#include <vector>
#include "estimator.h"

namespace sphericalsfm {

typedef Vec3 Ray;                                   // include/sphericalsfm/ray.h:8-10
typedef std::pair<Ray, Ray> RayPair;
typedef std::vector<RayPair> RayPairList;

class SphericalEstimator : public EssentialEstimator {
    const RayPairList& correspondences;
    const bool use_poly_solver;
    const bool inward;
    ssfm_estimator* handle;
    mutable Mat3 cached_model; mutable bool have_cache; mutable std::vector<double> cached_errors;
public:
    SphericalEstimator(const RayPairList& _correspondences, const bool _use_poly_solver, const bool _inward);
    ~SphericalEstimator();
    SphericalEstimator(const SphericalEstimator&) = delete;
    SphericalEstimator& operator=(const SphericalEstimator&) = delete;

    inline int min_sample_size() const { return 3; }
    inline int non_minimal_sample_size() const { return 4; }
    inline int num_data() const { return (int)correspondences.size(); }
    virtual int MinimalSolver(const std::vector<int>& sample, std::vector<Mat3>* Es) const;
    int NonMinimalSolver(const std::vector<int>& sample, Mat3* E) const;        // 0 if no model could be estimated and 1 otherwise
    double EvaluateModelOnPoint(const Mat3& E, int i) const;
    void LeastSquares(const std::vector<int>& sample, Mat3* E) const;
    void Decompose(const Mat3& E, const std::vector<int>& inliers, Mat3* R, Vec3* t) const;
};

}  // namespace sphericalsfm
