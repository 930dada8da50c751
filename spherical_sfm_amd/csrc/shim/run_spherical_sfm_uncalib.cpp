// Drop-in driver for the uncalibrated pipeline from the feature tracks on (examples/run_spherical_sfm_uncalib.cpp:101-228):
// focal guess (width + height) / 2, matches whose rotations were estimated at that guess, 1024-trial focal search around the pose
// graph + joint refinement (find_best_focal_length_random), build_sfm at the found focal, shared focal FREE in every bundle
// adjustment: spherical BA -> Retriangulate -> BA, with -generalba: unfix translations -> BA -> Normalize -> Retriangulate -> BA ->
// Normalize; poses.txt, OBJ files, COLMAP text model, calib.txt.  Everything numerical runs in libssfm_hip.so.
//   run_spherical_sfm_uncalib -output <dir with keyframes.txt, features.dat, matches.dat> -width W -height H [-generalba] [-inward]
#include <cstdio>
#include <iostream>
#include "tools.h"
using namespace sphericalsfm;

int main(int argc, char** argv) {
    std::string output; bool inward = false, generalba = false; int width = 0, height = 0, num_trials = 1024; unsigned seed = 0;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        if (a == "-output" && i + 1 < argc) output = argv[++i];
        else if (a == "-width" && i + 1 < argc) width = std::atoi(argv[++i]);
        else if (a == "-height" && i + 1 < argc) height = std::atoi(argv[++i]);
        else if (a == "-trials" && i + 1 < argc) num_trials = std::atoi(argv[++i]);
        else if (a == "-seed" && i + 1 < argc) seed = (unsigned)std::atoi(argv[++i]);
        else if (a == "-inward") inward = true;
        else if (a == "-generalba") generalba = true;
        else if (a == "-sequential") {}
        else { std::cout << "unknown argument " << a << "\n"; return 2; }
    }
    if (output.empty() || width <= 0 || height <= 0) { std::cout << "usage: run_spherical_sfm_uncalib -output <dir> -width W -height H [-generalba] [-inward]\n"; return 2; }
    std::vector<Keyframe> keyframes; std::vector<ImageMatch> image_matches;
    if (!read_feature_tracks(output, keyframes, image_matches) || image_matches.empty()) { std::cout << "error: no matches found\n"; return 1; }
    const double focal_guess = (width + height) / 2, centerx = width / 2, centery = height / 2;     // :101-103 (integer division as in the reference)
    std::cout << "initial focal: " << focal_guess << "\n";
    const double min_focal = focal_guess / 4, max_focal = focal_guess * 2;                          // :141-142

    SfM sfm_probe(Intrinsics(focal_guess, centerx, centery));                                       // owns the library context for the search
    std::vector<Mat3> rotations; double focal_new = focal_guess;
    if (!find_best_focal_length_random(sfm_probe.GetContext(), (int)keyframes.size(), image_matches, inward, true, focal_guess, min_focal, max_focal,
                                       num_trials, rotations, focal_new, seed, (output + "/costs.txt").c_str())) {
        std::cout << "ERROR: could not find any acceptable focal length\n"; return 1;
    }
    std::cout << " best focal: " << focal_new << "\n";
    const double focal_search = focal_new;
    Intrinsics intrinsics(focal_new, centerx, centery);

    std::cout << "building sfm\n";
    SfM sfm(intrinsics);
    build_sfm(keyframes, image_matches, rotations, sfm, true, true, inward);
    sfm.SetFocalFixed(false);
    sfm.WriteCOLMAP(output + "/sparse-pre-spherical-ba", width, height);
    sfm.WritePointsOBJ(output + "/points-pre-spherical-ba.obj");
    sfm.WriteCameraCentersOBJ(output + "/cameras-pre-spherical-ba.obj");
    const bool ok1 = sfm.Optimize();
    sfm.Retriangulate();
    const bool ok2 = sfm.Optimize();
    sfm.WriteCOLMAP(output + "/sparse-pre-general-ba", width, height);
    sfm.WritePointsOBJ(output + "/points-pre-general-ba.obj");
    sfm.WriteCameraCentersOBJ(output + "/cameras-pre-general-ba.obj");
    const double focal_spherical = sfm.GetFocal();
    std::cout << "focal after spherical BA: " << focal_spherical << "\n";
    bool ok3 = true, ok4 = true;
    if (generalba) {
        for (int i = 1; i < sfm.GetNumCameras(); i++) sfm.SetTranslationFixed(i, false);
        std::cout << "running general optimization\n";
        ok3 = sfm.Optimize();
        sfm.Normalize(inward);
        sfm.Retriangulate();
        ok4 = sfm.Optimize();
        sfm.Normalize(inward);
        std::cout << "done.\n";
        std::cout << "focal after general BA: " << sfm.GetFocal() << "\n";
    }
    std::vector<int> keyframe_indices(keyframes.size());
    for (size_t i = 0; i < keyframes.size(); i++) keyframe_indices[i] = keyframes[i].index;
    sfm.WritePoses(output + "/poses.txt", keyframe_indices);
    sfm.WritePointsOBJ(output + "/points.obj");
    sfm.WriteCameraCentersOBJ(output + "/cameras.obj");
    sfm.WriteCOLMAP(output + "/sparse", width, height);
    sfm.WriteCalib(output + "/calib.txt");
    std::printf("PIPELINE_RESULT ok=%d%d%d%d cameras=%d focal_guess=%.3f focal_search=%.6f focal_spherical=%.6f focal_final=%.6f cost=%.6e residuals=%lld\n", ok1, ok2, ok3, ok4,
                sfm.GetNumCameras(), focal_guess, focal_search, focal_spherical, sfm.GetFocal(), sfm.LastSummary().final_cost, (long long)sfm.LastSummary().num_residual_blocks);
    return 0;
}
