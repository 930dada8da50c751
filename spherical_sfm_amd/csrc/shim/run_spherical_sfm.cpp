// Drop-in driver for the calibrated pipeline from the feature tracks on (examples/run_spherical_sfm.cpp:71-121, the flow the
// reference intends past its debugging exit(0) at :81): read keyframes.txt / features.dat / matches.dat, sequential rotation
// initialisation, rotation averaging, build_sfm, spherical BA -> Retriangulate -> BA, general BA -> Normalize -> Retriangulate -> BA
// -> Normalize, then poses.txt, points.obj, cameras.obj and the COLMAP text model.  Everything numerical runs in libssfm_hip.so.
// Feature detection / matching / pairwise RANSAC over images (the OpenCV front end) is outside this build; ssfm_ransac_batch is
// the GPU replacement of the RANSAC step for callers that have the matches.
//   run_spherical_sfm -intrinsics <file: focal cx cy> -output <dir with the feature tracks> [-inward] [-width W -height H]
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include "tools.h"
using namespace sphericalsfm;

int main(int argc, char** argv) {
    std::string intrinsics_path, output; bool inward = false, pairwise = false; int width = 1920, height = 1080, mininliers = 100; double inlierthresh = 2.0;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        if (a == "-intrinsics" && i + 1 < argc) intrinsics_path = argv[++i];
        else if (a == "-output" && i + 1 < argc) output = argv[++i];
        else if (a == "-width" && i + 1 < argc) width = std::atoi(argv[++i]);
        else if (a == "-height" && i + 1 < argc) height = std::atoi(argv[++i]);
        else if (a == "-inward") inward = true;
        else if (a == "-pairwise") pairwise = true;                         // matches.dat holds raw matches: run estimate_pairwise (GPU RANSAC) first
        else if (a == "-inlierthresh" && i + 1 < argc) inlierthresh = std::atof(argv[++i]);
        else if (a == "-mininliers" && i + 1 < argc) mininliers = std::atoi(argv[++i]);
        else if (a == "-sequential") {}                                     // the only rotation initialisation available here
        else { std::cout << "unknown argument " << a << "\n"; return 2; }
    }
    if (intrinsics_path.empty() || output.empty()) { std::cout << "usage: run_spherical_sfm -intrinsics <file> -output <dir> [-inward]\n"; return 2; }
    double focal, centerx, centery;
    std::ifstream intrinsicsf(intrinsics_path);
    if (!(intrinsicsf >> focal >> centerx >> centery)) { std::cout << "error: could not read " << intrinsics_path << "\n"; return 1; }
    std::cout << "intrinsics : " << focal << ", " << centerx << ", " << centery << "\n";
    Intrinsics intrinsics(focal, centerx, centery);

    std::vector<Keyframe> keyframes; std::vector<ImageMatch> image_matches;
    if (!read_feature_tracks(output, keyframes, image_matches)) { std::cout << "error: no feature tracks in " << output << "\n"; return 1; }
    if (image_matches.empty()) { std::cout << "error: no loop closures found\n"; return 1; }

    SfM sfm(intrinsics);
    int loop_closures = -1;
    if (pairwise) {                                                          // run_spherical_sfm.cpp:56-63
        std::cout << "detecting loop closures\n";
        std::vector<ImageMatch> all_image_matches; all_image_matches.swap(image_matches);
        loop_closures = estimate_pairwise(sfm.GetContext(), intrinsics, keyframes, all_image_matches, inlierthresh, mininliers, inward, image_matches);
        if (loop_closures == 0) { std::cout << "error: no loop closures found\n"; return 1; }
        std::cout << "kept " << image_matches.size() << " of " << all_image_matches.size() << " image pairs, " << loop_closures << " loop closures\n";
    }
    std::cout << "initializing rotations\n";
    std::vector<Mat3> rotations;
    initialize_rotations_sequential((int)keyframes.size(), image_matches, rotations);

    std::cout << "refining rotations\n";
    const double rot_cost = refine_rotations(sfm.GetContext(), (int)keyframes.size(), image_matches, rotations);

    std::cout << "building sfm\n";
    build_sfm(keyframes, image_matches, rotations, sfm, true, true, inward);
    sfm.WritePointsOBJ(output + "/points-pre-spherical-ba.obj");
    sfm.WriteCameraCentersOBJ(output + "/cameras-pre-spherical-ba.obj");

    const bool ok1 = sfm.Optimize();
    sfm.Retriangulate();
    const bool ok2 = sfm.Optimize();
    const double cost_spherical = sfm.LastSummary().final_cost;
    sfm.WritePointsOBJ(output + "/points-pre-ba.obj");
    sfm.WriteCameraCentersOBJ(output + "/cameras-pre-ba.obj");

    // --- general SFM ---
    for (int i = 1; i < sfm.GetNumCameras(); i++) sfm.SetTranslationFixed(i, false);
    std::cout << "running general optimization\n";
    const bool ok3 = sfm.Optimize();
    sfm.Normalize(inward);
    sfm.Retriangulate();
    const bool ok4 = sfm.Optimize();
    sfm.Normalize(inward);
    std::cout << "done.\n";

    std::vector<int> keyframe_indices(keyframes.size());
    for (size_t i = 0; i < keyframes.size(); i++) keyframe_indices[i] = keyframes[i].index;
    sfm.WritePoses(output + "/poses.txt", keyframe_indices);
    sfm.WritePointsOBJ(output + "/points.obj");
    sfm.WriteCameraCentersOBJ(output + "/cameras.obj");
    sfm.WriteCOLMAP(output, width, height);
    std::printf("PAIRWISE_RESULT pairs=%zu loop_closures=%d\n", image_matches.size(), loop_closures);
    std::printf("PIPELINE_RESULT ok=%d%d%d%d cameras=%d points=%d rot_cost=%.6e cost_spherical=%.6e cost_general=%.6e residuals=%lld\n", ok1, ok2, ok3, ok4,
                sfm.GetNumCameras(), sfm.GetNumPoints(), rot_cost, cost_spherical, sfm.LastSummary().final_cost, (long long)sfm.LastSummary().num_residual_blocks);
    return 0;
}
