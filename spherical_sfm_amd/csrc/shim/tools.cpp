// see tools.h
#include "tools.h"
#include <cstdio>
#include <iostream>
#include <random>

namespace sphericalsfm {

bool find_best_focal_length_random(ssfm_ctx* ctx, int num_cameras, std::vector<ImageMatch>& image_matches, bool inward, bool sequential,
                                   double focal_guess, double min_focal, double max_focal, int num_trials, std::vector<Mat3>& rotations,
                                   double& best_focal, unsigned seed, const char* costs_path) {
    if (!sequential) { std::cout << "error: only the sequential rotation initialisation is available\n"; return false; }
    const int E = (int)image_matches.size();
    std::vector<int32_t> i0(E), i1(E); std::vector<double> rel((size_t)9 * E);
    for (int e = 0; e < E; e++) { i0[e] = image_matches[e].index0; i1[e] = image_matches[e].index1; for (int k = 0; k < 9; k++) rel[9 * (size_t)e + k] = image_matches[e].R[k]; }
    std::mt19937 gen(seed);
    std::uniform_real_distribution<double> dist(min_focal, max_focal);                    // spherical_sfm_tools.cpp:1449-1451
    std::vector<double> focals(num_trials), costs(num_trials);
    for (int t = 0; t < num_trials; t++) focals[t] = dist(gen);
    int32_t best = 0;
    std::vector<double> rot((size_t)9 * num_cameras), rel_best((size_t)9 * E);
    if (ssfm_focal_search(ctx, num_cameras, E, i0.data(), i1.data(), rel.data(), inward ? 1 : 0, focal_guess, num_trials, focals.data(), costs.data(),
                          &best, rot.data(), rel_best.data()) != SSFM_OK) { std::cout << "error: " << ssfm_last_error(ctx) << "\n"; return false; }
    if (costs_path) {                                                                     // :1463-1468
        if (FILE* f = std::fopen(costs_path, "w")) { for (int t = 0; t < num_trials; t++) std::fprintf(f, "%d %lf %lf\n", t, focals[t], costs[t]); std::fclose(f); }
    }
    best_focal = focals[best];
    std::cout << "before optimization: " << best_focal << "\n";
    // run_optimization (:1160-1188): matches at the best focal, joint refinement of rotations and focal inside [min, max]
    ssfm_ba_summary S;
    if (ssfm_posegraph_focal_solve(ctx, num_cameras, rot.data(), E, i0.data(), i1.data(), rel_best.data(), &best_focal, min_focal, max_focal, nullptr, &S) != SSFM_OK
        || S.termination == SSFM_FAILURE) { std::cout << "error: ceres failed.\n"; return false; }
    rotations.resize(num_cameras);
    for (int i = 0; i < num_cameras; i++) for (int k = 0; k < 9; k++) rotations[i][k] = rot[9 * (size_t)i + k];
    std::cout << "after optimization: " << best_focal << "\n";
    return true;
}

}  // namespace sphericalsfm
