// spherical_sfm_amd -- the reference's pose-graph entry points with their own signatures (src/rotation_averaging.cpp:44-91,
// src/uncalibrated_pose_graph.cpp:116-203), over ssfm_rotavg_solve / ssfm_rotavg_cost / ssfm_posegraph_focal_solve.
#include <cstdlib>
#include <iostream>
#include "uncalibrated_pose_graph.h"

namespace sphericalsfm {

namespace {
struct FlatGraph {
    std::vector<int32_t> i0, i1; std::vector<double> rel, rot;
    FlatGraph(const std::vector<Mat3>& rotations, const std::vector<RelativeRotation>& rr) : i0(rr.size()), i1(rr.size()), rel(9 * rr.size()), rot(9 * rotations.size()) {
        for (size_t e = 0; e < rr.size(); e++) { i0[e] = rr[e].index0; i1[e] = rr[e].index1; for (int k = 0; k < 9; k++) rel[9 * e + k] = rr[e].R[k]; }
        for (size_t i = 0; i < rotations.size(); i++) for (int k = 0; k < 9; k++) rot[9 * i + k] = rotations[i][k];
    }
    void store(std::vector<Mat3>& rotations) const { for (size_t i = 0; i < rotations.size(); i++) for (int k = 0; k < 9; k++) rotations[i][k] = rot[9 * i + k]; }
};
void check(int rc, const ssfm_ba_summary& S) {
    if (rc != SSFM_OK) { std::cout << "error: " << ssfm_last_error(default_context()) << "\n"; std::exit(1); }
    if (S.termination == SSFM_FAILURE) { std::cout << "error: ceres failed.\n"; std::exit(1); }       // src/rotation_averaging.cpp:82-86
}
}  // namespace

double optimize_rotations(std::vector<Mat3>& rotations, const std::vector<RelativeRotation>& relative_rotations) {
    FlatGraph G(rotations, relative_rotations); ssfm_ba_summary S;
    const int rc = ssfm_rotavg_solve(default_context(), (int32_t)rotations.size(), G.rot.data(), (int32_t)G.i0.size(), G.i0.data(), G.i1.data(), G.rel.data(), nullptr, &S);
    check(rc, S); G.store(rotations);
    return S.final_cost;
}

double get_cost(std::vector<Mat3>& rotations, const std::vector<RelativeRotation>& relative_rotations) {
    FlatGraph G(rotations, relative_rotations); double cost = 0;
    if (ssfm_rotavg_cost(default_context(), (int32_t)rotations.size(), G.rot.data(), (int32_t)G.i0.size(), G.i0.data(), G.i1.data(), G.rel.data(), &cost) != SSFM_OK) {
        std::cout << "error: " << ssfm_last_error(default_context()) << "\n"; std::exit(1); }
    return cost;
}

// focal_guess and inward are part of the reference's signature and unused by its body (src/uncalibrated_pose_graph.cpp:147-203)
double optimize_rotations_and_focal_length(std::vector<Mat3>& rotations, const std::vector<RelativeRotation>& relative_rotations, double& focal_length,
                                           const double min_focal, const double max_focal, const double /*focal_guess*/, bool /*inward*/) {
    FlatGraph G(rotations, relative_rotations); ssfm_ba_summary S;
    const int rc = ssfm_posegraph_focal_solve(default_context(), (int32_t)rotations.size(), G.rot.data(), (int32_t)G.i0.size(), G.i0.data(), G.i1.data(), G.rel.data(),
                                              &focal_length, min_focal, max_focal, nullptr, &S);
    check(rc, S); G.store(rotations);
    return S.final_cost;
}

}  // namespace sphericalsfm
