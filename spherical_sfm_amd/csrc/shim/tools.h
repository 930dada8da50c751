// spherical_sfm_amd -- the part of examples/spherical_sfm_tools.{h,cpp} that sits on the optimisation hot path, over the C ABI:
// find_best_focal_length_random (spherical_sfm_tools.cpp:1418-1496) with its run_optimization (:1160-1188).
// POD stand-ins for the Eigen types, as in sfm.h.  Host code only.
#pragma once
#include <array>
#include <cstddef>
#include <map>
#include <string>
#include <vector>
#include "../../../include/ssfm.h"
#include "sfm.h"

namespace sphericalsfm {

typedef std::array<double, 9> Mat3;                       // column-major like Eigen::Matrix3d::data()
typedef std::map<std::size_t, std::size_t> Matches;                 // spherical_sfm_tools.h:20

struct ImageMatch {                                       // spherical_sfm_tools.h:41-50
    int index0, index1;
    Matches matches;
    Mat3 R;
    ImageMatch(int _index0, int _index1, const Matches& _matches, const Mat3& _R) : index0(_index0), index1(_index1), matches(_matches), R(_R) {}
};

struct Point2f { float x, y; };
struct Features {                                         // spherical_sfm_tools.h:19-28 (cv::Mat descs -> flat 128 floats per feature)
    std::vector<int> tracks;
    std::vector<Point2f> points;
    std::vector<std::array<unsigned char, 3>> colors;
    std::vector<float> descs;
    int size() const { return (int)points.size(); }
    bool empty() const { return points.empty(); }
};
struct Keyframe {                                         // spherical_sfm_tools.h:30-39 (images are not part of this path)
    int index; std::string name; Features features;
    Keyframe(int _index, const std::string& _name, const Features& _features) : index(_index), name(_name), features(_features) {}
};

// keyframes.txt / features.dat / matches.dat (examples/spherical_sfm_io.cpp:10-120).  The reference prints the keyframe name with
// "%s" of a std::string object and reads back only the indices; here the name is written as text and skipped on input.
void write_feature_tracks(const std::string& outputpath, const std::vector<Keyframe>& keyframes, const std::vector<ImageMatch>& image_matches);
bool read_feature_tracks(const std::string& outputpath, std::vector<Keyframe>& keyframes, std::vector<ImageMatch>& image_matches);

// estimate_pairwise (spherical_sfm_tools.cpp:309-431): every candidate pair (index0 < index1) with at least min_num_inliers matches goes
// through the spherical 3-point LO-MSAC -- all pairs in ONE ssfm_ransac_batch launch instead of the OpenMP loop; pairs with more than
// min_num_inliers inliers come back with their inlier matches and R = so3exp(decompose(E)).  Returns the number of loop closures
// (accepted pairs that are not consecutive).  COLLECTIVE if ctx carries a communicator: every rank calls it with the same arguments (pairs are sharded, one all-reduce
// returns all results everywhere); a context without a communicator runs all pairs locally.
int estimate_pairwise(ssfm_ctx* ctx, const Intrinsics& intrinsics, const std::vector<Keyframe>& keyframes, const std::vector<ImageMatch>& image_matches,
                      double inlier_threshold, int min_num_inliers, bool inward, std::vector<ImageMatch>& image_matches_out);

void initialize_rotations_sequential(int num_cameras, const std::vector<ImageMatch>& image_matches, std::vector<Mat3>& rotations);   // tools.cpp:794-813
double refine_rotations(ssfm_ctx* ctx, int num_cameras, const std::vector<ImageMatch>& image_matches, std::vector<Mat3>& rotations); // tools.cpp:851-860
// tools.cpp:862-955: tracks (ssfm_build_tracks, ids bit-exact with the reference's AddPoint sequence), cameras, observations, Retriangulate
void build_sfm(std::vector<Keyframe>& keyframes, const std::vector<ImageMatch>& image_matches, const std::vector<Mat3>& rotations, SfM& sfm,
               bool spherical, bool merge, bool inward, int fix_camera = 0);

// The reference seeds std::mt19937 from std::random_device and draws inside an OpenMP loop; here the draw is sequential from
// `seed` (deterministic), everything after it follows the reference: costs of all trials in one GPU launch, first minimum,
// sequential rotations at the best focal, then the joint rotation + focal refinement.  Only sequential = true is supported
// (the GraphOptim initialisation is outside the scope of this build).  Returns false on an error of the library.
bool find_best_focal_length_random(ssfm_ctx* ctx, int num_cameras, std::vector<ImageMatch>& image_matches, bool inward, bool sequential,
                                   double focal_guess, double min_focal, double max_focal, int num_trials, std::vector<Mat3>& rotations,
                                   double& best_focal, unsigned seed = 0, const char* costs_path = "costs.txt");

}  // namespace sphericalsfm
