// see tools.h
#include "tools.h"
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <random>

#include "../ssfm_math.h"

namespace sphericalsfm {

void write_feature_tracks(const std::string& outputpath, const std::vector<Keyframe>& keyframes, const std::vector<ImageMatch>& image_matches) {
    if (FILE* f = std::fopen((outputpath + "/keyframes.txt").c_str(), "w")) {
        std::fprintf(f, "%d\n", (int)keyframes.size());
        for (const Keyframe& k : keyframes) std::fprintf(f, "%d %s\n", k.index, k.name.c_str());
        std::fclose(f);
    }
    if (FILE* f = std::fopen((outputpath + "/features.dat").c_str(), "w")) {
        for (const Keyframe& k : keyframes) {
            const int nfeatures = k.features.size();
            std::fwrite(&nfeatures, sizeof(int), 1, f);
            for (int j = 0; j < nfeatures; j++) {
                std::fwrite(&k.features.points[j].x, sizeof(float), 1, f); std::fwrite(&k.features.points[j].y, sizeof(float), 1, f);
                static const float zeros[128] = {0};
                std::fwrite(k.features.descs.size() >= (size_t)(j + 1) * 128 ? &k.features.descs[(size_t)j * 128] : zeros, sizeof(float), 128, f);
            }
        }
        std::fclose(f);
    }
    if (FILE* f = std::fopen((outputpath + "/matches.dat").c_str(), "w")) {
        const int n = (int)image_matches.size(); std::fwrite(&n, sizeof(int), 1, f);
        for (const ImageMatch& m : image_matches) {
            std::fwrite(&m.index0, sizeof(int), 1, f); std::fwrite(&m.index1, sizeof(int), 1, f);
            const int nm = (int)m.matches.size(); std::fwrite(&nm, sizeof(int), 1, f);
            for (auto& kv : m.matches) { const int a = (int)kv.first, b = (int)kv.second; std::fwrite(&a, sizeof(int), 1, f); std::fwrite(&b, sizeof(int), 1, f); }
            std::fwrite(m.R.data(), sizeof(double), 9, f);                              // Eigen column-major
        }
        std::fclose(f);
    }
}

bool read_feature_tracks(const std::string& outputpath, std::vector<Keyframe>& keyframes, std::vector<ImageMatch>& image_matches) {
    FILE* kf = std::fopen((outputpath + "/keyframes.txt").c_str(), "r");
    if (!kf) return false;
    int nkeyframes = 0;
    if (std::fscanf(kf, "%d\n", &nkeyframes) != 1 || nkeyframes < 0) { std::fclose(kf); return false; }
    std::vector<int> indices(nkeyframes);
    for (int i = 0; i < nkeyframes; i++) {
        if (std::fscanf(kf, "%d", &indices[i]) != 1) { std::fclose(kf); return false; }
        int ch; while ((ch = std::fgetc(kf)) != EOF && ch != '\n') {}                  // the rest of the line is the name
    }
    std::fclose(kf);
    std::cout << "read " << indices.size() << " indices\n";
    FILE* ff = std::fopen((outputpath + "/features.dat").c_str(), "r");
    if (!ff) return false;
    for (int i = 0; i < nkeyframes; i++) {
        int nfeatures = 0;
        if (std::fread(&nfeatures, sizeof(int), 1, ff) != 1 || nfeatures < 0) { std::fclose(ff); return false; }
        Features features; features.points.resize(nfeatures); features.descs.resize((size_t)nfeatures * 128);
        for (int j = 0; j < nfeatures; j++) {
            if (std::fread(&features.points[j].x, sizeof(float), 1, ff) != 1 || std::fread(&features.points[j].y, sizeof(float), 1, ff) != 1 ||
                std::fread(&features.descs[(size_t)j * 128], sizeof(float), 128, ff) != 128) { std::fclose(ff); return false; }
        }
        char name[1024]; std::snprintf(name, sizeof name, "%06d.jpg", indices[i] + 1);
        keyframes.push_back(Keyframe(indices[i], name, features));
    }
    std::fclose(ff);
    FILE* mf = std::fopen((outputpath + "/matches.dat").c_str(), "r");
    if (!mf) return false;
    int nmatches = 0;
    if (std::fread(&nmatches, sizeof(int), 1, mf) != 1) { std::fclose(mf); return false; }
    for (int i = 0; i < nmatches; i++) {
        int index0, index1, nm;
        if (std::fread(&index0, sizeof(int), 1, mf) != 1 || std::fread(&index1, sizeof(int), 1, mf) != 1 || std::fread(&nm, sizeof(int), 1, mf) != 1) { std::fclose(mf); return false; }
        Matches m;
        for (int j = 0; j < nm; j++) { int a, b; if (std::fread(&a, sizeof(int), 1, mf) != 1 || std::fread(&b, sizeof(int), 1, mf) != 1) { std::fclose(mf); return false; } m[a] = b; }
        Mat3 R; if (std::fread(R.data(), sizeof(double), 9, mf) != 9) { std::fclose(mf); return false; }
        image_matches.push_back(ImageMatch(index0, index1, m, R));
    }
    std::fclose(mf);
    return true;
}

int estimate_pairwise(ssfm_ctx* ctx, const Intrinsics& intrinsics, const std::vector<Keyframe>& keyframes, const std::vector<ImageMatch>& image_matches,
                      double inlier_threshold, int min_num_inliers, bool inward, std::vector<ImageMatch>& image_matches_out) {
    const double kinv = 1.0 / intrinsics.focal;                                           // Kinv(0,0)
    const double sq_thresh = inlier_threshold * inlier_threshold * kinv * kinv;           // :315
    // the reference enumerates all (index0 < index1) and takes the FIRST stored match set of each pair
    std::map<std::pair<int, int>, const ImageMatch*> first;
    for (const ImageMatch& m : image_matches) if (m.index0 < m.index1) first.emplace(std::make_pair(m.index0, m.index1), &m);
    std::vector<const ImageMatch*> cand;
    for (auto& kv : first) if ((int)kv.second->matches.size() >= min_num_inliers && !kv.second->matches.empty()) cand.push_back(kv.second);   // :353
    if (cand.empty()) return 0;
    // per-frame feature rays Kinv * (x, y, 1) (:362-376, once per feature) + per-pair match lists: ssfm_ransac_batch_indexed gathers the ray pairs on the device
    const int nf = (int)keyframes.size();
    std::vector<int32_t> feat_ptr(nf + 1, 0);
    for (int f = 0; f < nf; f++) feat_ptr[f + 1] = feat_ptr[f] + (int32_t)keyframes[f].features.points.size();
    std::vector<double> rays((size_t)3 * std::max(1, (int)feat_ptr[nf]));
    for (int f = 0; f < nf; f++) {
        const Features& ft = keyframes[f].features;
        for (size_t k = 0; k < ft.points.size(); k++) {
            double* r = &rays[3 * ((size_t)feat_ptr[f] + k)];
            r[0] = (ft.points[k].x - intrinsics.centerx) * kinv; r[1] = (ft.points[k].y - intrinsics.centery) * kinv; r[2] = 1.0;
        }
    }
    std::vector<int32_t> pair_ptr(1, 0), pf0, pf1, m0, m1;
    for (const ImageMatch* m : cand) {
        pf0.push_back(m->index0); pf1.push_back(m->index1);
        for (auto& kv : m->matches) { m0.push_back((int32_t)kv.first); m1.push_back((int32_t)kv.second); }
        pair_ptr.push_back((int32_t)m0.size());
    }
    ssfm_ransac_options O; ssfm_ransac_default_options(&O);
    O.min_num_inliers = min_num_inliers; O.inward = inward ? 1 : 0; O.final_least_squares = 1;                                 // :316-318
    const int P = (int)cand.size();
    std::vector<double> R((size_t)9 * P); std::vector<uint8_t> mask(std::max<size_t>(1, m0.size())); std::vector<int32_t> nin(P);
    // COLLECTIVE when the context carries a communicator (ssfm_comm_init / ssfm_comm_init_host): every rank of the job must make this call with the same
    // keyframes and matches, or the ranks that did wait in the result all-reduce forever -- a rank-0-only pairwise stage needs a context of its own without a
    // communicator.  Without one this is the plain single-GPU indexed batch.  (The result table is summed: a -0.0 entry of a rotation comes back as +0.0 on the
    // ranks that did not compute it; every other bit equals the single-GPU result.)
    if (ssfm_ransac_batch_indexed_sharded(ctx, nf, feat_ptr.data(), rays.data(), P, pf0.data(), pf1.data(), pair_ptr.data(), m0.data(), m1.data(), sq_thresh, &O, nullptr, R.data(),
                                  mask.data(), nin.data(), nullptr, nullptr) != SSFM_OK) {
        std::cout << "error: " << ssfm_last_error(ctx) << "\n"; std::exit(1);
    }
    int loop_closure_count = 0;
    for (int k = 0; k < P; k++) {
        if (!(nin[k] > min_num_inliers)) continue;                                        // :410
        Matches inl; size_t j = (size_t)pair_ptr[k];
        for (auto& kv : cand[k]->matches) { if (mask[j++]) inl[kv.first] = kv.second; }
        if (inl.empty()) continue;
        Mat3 Rk; for (int q = 0; q < 9; q++) Rk[q] = R[9 * (size_t)k + q];
        if (cand[k]->index0 + 1 != cand[k]->index1) loop_closure_count++;
        image_matches_out.push_back(ImageMatch(cand[k]->index0, cand[k]->index1, inl, Rk));
    }
    return loop_closure_count;
}

void initialize_rotations_sequential(int num_cameras, const std::vector<ImageMatch>& image_matches, std::vector<Mat3>& rotations) {
    const Mat3 I = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    rotations.assign(num_cameras, I);
    Mat3 R = I;
    for (int index = 1; index < num_cameras; index++)
        for (size_t i = 0; i < image_matches.size(); i++)
            if (image_matches[i].index0 == index - 1 && image_matches[i].index1 == index) {
                Mat3 Rn;                                                              // R = match.R * R, column-major 3x3
                for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { double s = 0; for (int k = 0; k < 3; k++) s += image_matches[i].R[r + 3 * k] * R[k + 3 * c]; Rn[r + 3 * c] = s; }
                R = Rn; rotations[index] = R;
                break;
            }
}

double refine_rotations(ssfm_ctx* ctx, int num_cameras, const std::vector<ImageMatch>& image_matches, std::vector<Mat3>& rotations) {
    const int E = (int)image_matches.size();
    std::vector<int32_t> i0(E), i1(E); std::vector<double> rel((size_t)9 * E), rot((size_t)9 * num_cameras);
    for (int e = 0; e < E; e++) { i0[e] = image_matches[e].index0; i1[e] = image_matches[e].index1; for (int k = 0; k < 9; k++) rel[9 * (size_t)e + k] = image_matches[e].R[k]; }
    for (int i = 0; i < num_cameras; i++) for (int k = 0; k < 9; k++) rot[9 * (size_t)i + k] = rotations[i][k];
    ssfm_ba_summary S;
    if (ssfm_rotavg_solve(ctx, num_cameras, rot.data(), E, i0.data(), i1.data(), rel.data(), nullptr, &S) != SSFM_OK || S.termination == SSFM_FAILURE) {
        std::cout << "error: ceres failed.\n"; std::exit(1);                          // src/rotation_averaging.cpp:82-86
    }
    for (int i = 0; i < num_cameras; i++) for (int k = 0; k < 9; k++) rotations[i][k] = rot[9 * (size_t)i + k];
    return S.final_cost;
}

void build_sfm(std::vector<Keyframe>& keyframes, const std::vector<ImageMatch>& image_matches, const std::vector<Mat3>& rotations, SfM& sfm,
               bool spherical, bool merge, bool inward, int fix_camera) {
    std::cout << "building tracks\n";
    const int nk = (int)keyframes.size();
    std::vector<int32_t> feat_ptr(nk + 1, 0);
    for (int i = 0; i < nk; i++) feat_ptr[i + 1] = feat_ptr[i] + keyframes[i].features.size();
    std::vector<double> feat_xy((size_t)feat_ptr[nk] * 2);
    for (int i = 0; i < nk; i++) for (int j = 0; j < keyframes[i].features.size(); j++) {
        feat_xy[2 * ((size_t)feat_ptr[i] + j)] = keyframes[i].features.points[j].x; feat_xy[2 * ((size_t)feat_ptr[i] + j) + 1] = keyframes[i].features.points[j].y; }
    std::vector<int32_t> ms0, ms1, ms_ptr(1, 0), f0, f1;
    for (const ImageMatch& m : image_matches) {
        ms0.push_back(m.index0); ms1.push_back(m.index1);
        for (auto& kv : m.matches) { f0.push_back((int32_t)kv.first); f1.push_back((int32_t)kv.second); }
        ms_ptr.push_back((int32_t)f0.size());
    }
    const size_t nm = f0.size();
    std::vector<int32_t> tracks(feat_ptr[nk]), ocam(2 * nm + 1), opt(2 * nm + 1); std::vector<uint8_t> alive(nm + 1); std::vector<double> oxy(4 * nm + 2);
    int32_t npts = 0; int64_t nobs = 0;
    std::cout << "adding cameras\n";
    for (int index = 0; index < nk; index++) {
        double Rm[9], r[3];                                                          // so3ln of the column-major rotation
        for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) Rm[3 * a + b] = rotations[index][a + 3 * b];
        ssfm::so3ln(Rm, r);
        const int camera = sfm.AddCamera(Pose(Vec3(0, 0, inward ? 1 : -1), Vec3(r[0], r[1], r[2])), keyframes[index].name);
        sfm.SetRotationFixed(camera, index == fix_camera);
        sfm.SetTranslationFixed(camera, spherical ? true : (index == fix_camera));
    }
    std::cout << "adding tracks\nnumber of keyframes is " << nk << "\n";
    ssfm_build_tracks(nk, feat_ptr.data(), feat_xy.data(), (int32_t)image_matches.size(), ms0.data(), ms1.data(), ms_ptr.data(), f0.data(), f1.data(),
                      sfm.GetIntrinsics().centerx, sfm.GetIntrinsics().centery, merge ? 1 : 0, tracks.data(), &npts, alive.data(), &nobs, ocam.data(), opt.data(), oxy.data());
    for (int i = 0; i < nk; i++) keyframes[i].features.tracks.assign(tracks.begin() + feat_ptr[i], tracks.begin() + feat_ptr[i + 1]);
    for (int j = 0; j < npts; j++) { const int p = sfm.AddPoint(Point(0, 0, 0)); sfm.SetPointFixed(p, false); }
    for (int j = 0; j < npts; j++) if (!alive[j]) sfm.RemovePoint(j);                    // points consumed by MergePoint
    for (int64_t o = 0; o < nobs; o++) sfm.AddObservation(ocam[o], opt[o], Observation(oxy[2 * o], oxy[2 * o + 1]));
    std::cout << "retriangulating...\n";
    sfm.Retriangulate();
}

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
