// spherical_sfm_amd -- mirror of the reference's estimator interface (include/sphericalsfm/estimator.h:7-29): the seven virtuals a
// RANSAC driver calls, and EssentialEstimator::Decompose.  Eigen::Matrix3d -> Mat3 (column-major, tools.h), Eigen::Vector3d -> Vec3 (sfm.h).
#pragma once
#include <vector>
#include "tools.h"

namespace sphericalsfm {

template <typename SolutionType>
class Estimator {
public:
    virtual ~Estimator() {}
    virtual int min_sample_size() const = 0;
    virtual int non_minimal_sample_size() const = 0;
    virtual int num_data() const = 0;
    virtual int MinimalSolver(const std::vector<int>& sample, std::vector<SolutionType>* Es) const = 0;
    virtual int NonMinimalSolver(const std::vector<int>& sample, SolutionType* E) const = 0;
    virtual double EvaluateModelOnPoint(const SolutionType& E, int i) const = 0;
    virtual void LeastSquares(const std::vector<int>& sample, SolutionType* E) const = 0;
};

class EssentialEstimator : public Estimator<Mat3> {
public:
    virtual void Decompose(const Mat3& E, const std::vector<int>& inliers, Mat3* R, Vec3* t) const = 0;
};

// the library context the signature-compatible free functions and estimators of this shim share (the reference's signatures carry none)
ssfm_ctx* default_context();

}  // namespace sphericalsfm
