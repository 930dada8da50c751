// spherical_sfm_amd -- the part of examples/spherical_sfm_tools.{h,cpp} that sits on the optimisation hot path, over the C ABI:
// find_best_focal_length_random (spherical_sfm_tools.cpp:1418-1496) with its run_optimization (:1160-1188).
// POD stand-ins for the Eigen types, as in sfm.h.  Host code only.
#pragma once
#include <array>
#include <cstddef>
#include <map>
#include <vector>
#include "../../../include/ssfm.h"

namespace sphericalsfm {

typedef std::array<double, 9> Mat3;                       // column-major like Eigen::Matrix3d::data()
typedef std::map<std::size_t, std::size_t> Matches;                 // spherical_sfm_tools.h:20

struct ImageMatch {                                       // spherical_sfm_tools.h:41-50
    int index0, index1;
    Matches matches;
    Mat3 R;
    ImageMatch(int _index0, int _index1, const Matches& _matches, const Mat3& _R) : index0(_index0), index1(_index1), matches(_matches), R(_R) {}
};

// The reference seeds std::mt19937 from std::random_device and draws inside an OpenMP loop; here the draw is sequential from
// `seed` (deterministic), everything after it follows the reference: costs of all trials in one GPU launch, first minimum,
// sequential rotations at the best focal, then the joint rotation + focal refinement.  Only sequential = true is supported
// (the GraphOptim initialisation is outside the scope of this build).  Returns false on an error of the library.
bool find_best_focal_length_random(ssfm_ctx* ctx, int num_cameras, std::vector<ImageMatch>& image_matches, bool inward, bool sequential,
                                   double focal_guess, double min_focal, double max_focal, int num_trials, std::vector<Mat3>& rotations,
                                   double& best_focal, unsigned seed = 0, const char* costs_path = "costs.txt");

}  // namespace sphericalsfm
