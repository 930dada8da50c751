// spherical_sfm_amd -- include/sphericalsfm/rotation_averaging.h:9-16 with the reference's signature; Eigen::Matrix3d -> Mat3 (column-major).
#pragma once
#include <vector>
#include "estimator.h"

namespace sphericalsfm {

struct RelativeRotation {
    int index0, index1;
    Mat3 R;   // R1 * R0^T
    RelativeRotation(const int _index0, const int _index1, const Mat3& _R) : index0(_index0), index1(_index1), R(_R) {}
};

double optimize_rotations(std::vector<Mat3>& rotations, const std::vector<RelativeRotation>& relative_rotations);

}  // namespace sphericalsfm
