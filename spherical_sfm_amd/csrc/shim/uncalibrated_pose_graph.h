// spherical_sfm_amd -- include/sphericalsfm/uncalibrated_pose_graph.h:8-19 with the reference's signatures.
#pragma once
#include "rotation_averaging.h"

namespace sphericalsfm {

double get_cost(std::vector<Mat3>& rotations, const std::vector<RelativeRotation>& relative_rotations);

double optimize_rotations_and_focal_length(std::vector<Mat3>& rotations, const std::vector<RelativeRotation>& relative_rotations, double& focal_length,
                                           const double min_focal, const double max_focal, const double focal_guess, bool inward);

#ifdef SSFM_WITH_EIGEN
#include <Eigen/Core>       // (named here, not only through sfm.h: ADVICE r5)
#endif
#ifdef SSFM_WITH_EIGEN      // the reference's own signatures (uncalibrated_pose_graph.h:8-19); not compiled in this image (no Eigen), see sfm.h
inline double get_cost(std::vector<Eigen::Matrix3d>& rotations, const std::vector<RelativeRotationEigen>& relative_rotations) {
    std::vector<Mat3> R; R.reserve(rotations.size());
    for (const auto& M : rotations) R.push_back(mat3_from_eigen(M));
    return get_cost(R, relative_rotations_from_eigen(relative_rotations));
}
inline double optimize_rotations_and_focal_length(std::vector<Eigen::Matrix3d>& rotations, const std::vector<RelativeRotationEigen>& relative_rotations, double& focal_length,
                                                  const double min_focal, const double max_focal, const double focal_guess, bool inward) {
    std::vector<Mat3> R; R.reserve(rotations.size());
    for (const auto& M : rotations) R.push_back(mat3_from_eigen(M));
    const double cost = optimize_rotations_and_focal_length(R, relative_rotations_from_eigen(relative_rotations), focal_length, min_focal, max_focal, focal_guess, inward);
    for (size_t i = 0; i < R.size(); i++) rotations[i] = mat3_to_eigen(R[i]);
    return cost;
}
#endif

}  // namespace sphericalsfm
