// C++ mirror of sphericalsfm::SfM over the C ABI (see sfm.h).  Host-only code; the solve runs in libssfm_hip.so.
#include "sfm.h"
#include <algorithm>
#include <initializer_list>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <sys/stat.h>
#include <sys/types.h>
#include "../ssfm_math.h"

namespace sphericalsfm {

double Vec3::norm() const { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }

Pose::Pose() : R{1, 0, 0, 0, 1, 0, 0, 0, 1} {}
Pose::Pose(const Vec3& _t, const Vec3& _r) : t(_t), r(_r) { ssfm::so3exp(r.v, R); }          // src/sfm_types.cpp:14-19
Pose Pose::inverse() const {                                                                  // src/sfm_types.cpp:21-29
    Pose q;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) q.R[3 * i + j] = R[3 * j + i];
    double mt[3] = {-t.v[0], -t.v[1], -t.v[2]};
    ssfm::mat3_vec(q.R, mt, q.t.v);
    q.r = Vec3(-r.v[0], -r.v[1], -r.v[2]);
    return q;
}
void Pose::postMultiply(const Pose& pose) {                                                   // src/sfm_types.cpp:31-36: P = P * pose.P
    double Rn[9], tn[3];
    ssfm::mat3_mul(R, pose.R, Rn);
    ssfm::mat3_vec(R, pose.t.v, tn);
    for (int i = 0; i < 3; i++) t.v[i] += tn[i];
    for (int i = 0; i < 9; i++) R[i] = Rn[i];
    ssfm::so3ln(R, r.v);
}
Point Pose::apply(const Point& p) const { Point q; ssfm::mat3_vec(R, p.v, q.v); for (int i = 0; i < 3; i++) q.v[i] += t.v[i]; return q; }
Point Pose::applyInverse(const Point& p) const { double d[3] = {p.v[0] - t.v[0], p.v[1] - t.v[1], p.v[2] - t.v[2]}; Point q; ssfm::mat3_tvec(R, d, q.v); return q; }
Vec3 Pose::getCenter() const { double mt[3] = {-t.v[0], -t.v[1], -t.v[2]}; Vec3 c; ssfm::mat3_tvec(R, mt, c.v); return c; }

SfM::SfM(const Intrinsics& _intrinsics)
    : intrinsics(_intrinsics), focalFixed(true), numCameras(0), numPoints(0), nextCamera(-1), nextPoint(0), ctx(nullptr), last_summary() {}
SfM::~SfM() { if (ctx) ssfm_ctx_destroy(ctx); }
ssfm_ctx* SfM::GetContext() {
    if (!ctx && ssfm_ctx_create(-1, nullptr, &ctx) != SSFM_OK) { std::cout << "error: " << ssfm_last_error(nullptr) << "\n"; exit(1); }
    return ctx;
}

int SfM::AddCamera(const Pose& pose, const std::string& path) {                               // src/sfm.cpp:99-111
    nextCamera++; numCameras++; obs_cache.stale = true;
    cameras[nextCamera] = Camera{pose.t.v[0], pose.t.v[1], pose.t.v[2], pose.r.v[0], pose.r.v[1], pose.r.v[2]};
    paths[nextCamera] = path; rotationFixed[nextCamera] = false; translationFixed[nextCamera] = false;
    return nextCamera;
}
int SfM::AddPoint(const Point& X) { numPoints++; obs_cache.stale = true; points[nextPoint] = X; pointFixed[nextPoint] = false; return nextPoint++; }   // src/sfm.cpp:113-127
int SfM::AddPoint(const Point& X, const std::array<unsigned char, 3>& color_bgr) { const int p = AddPoint(X); colors[p] = color_bgr; return p; }
int SfM::AddPoint(const Point& X, const std::vector<float>& descriptor, const std::array<unsigned char, 3>& color_bgr) { const int p = AddPoint(X, color_bgr); descriptors[p] = descriptor; return p; }
std::vector<float> SfM::GetDescriptor(int point) { auto it = descriptors.find(point); return it == descriptors.end() ? std::vector<float>() : it->second; }
std::array<unsigned char, 3> SfM::GetColor(int point) { auto it = colors.find(point); return it == colors.end() ? std::array<unsigned char, 3>{0, 0, 0} : it->second; }
void SfM::AddObservation(int camera, int point, const Observation& o) { observations[camera][point] = o; obs_cache.stale = true; }                 // src/sfm.cpp:143-146
bool SfM::GetObservation(int camera, int point, Observation& o) {
    auto r = observations.find(camera); if (r == observations.end()) return false;
    auto c = r->second.find(point); if (c == r->second.end()) return false;
    o = c->second; return true;
}
bool SfM::GetMeasurement(int i, int j, Pose& m) {                                            // same lookup as GetObservation (src/sfm.cpp:148-154) on the m x m map
    auto r = measurements.find(i); if (r == measurements.end()) return false;
    auto c = r->second.find(j); if (c == r->second.end()) return false;
    m = c->second; return true;
}
void SfM::MergePoint(int point1, int point2) {                                                // src/sfm.cpp:129-141
    obs_cache.stale = true;
    for (auto& row : observations) {
        if (row.first < 0 || row.first >= numCameras || !cameras.count(row.first)) continue;
        auto it = row.second.find(point2);
        if (it != row.second.end()) { const Observation o = it->second; row.second[point1] = o; }    // (a copy first: the insert may move the row's storage)
    }
    RemovePoint(point2);
}
void SfM::RemovePoint(int point) {                                                            // src/sfm.cpp:435-444
    obs_cache.stale = true;
    for (auto& row : observations) if (row.first >= 0 && row.first < numCameras) row.second.erase(point);
    points.erase(point); colors.erase(point); descriptors.erase(point);
}
void SfM::RemoveCamera(int camera) {                                                          // src/sfm.cpp:446-461
    obs_cache.stale = true;
    cameras.erase(camera); observations.erase(camera);
    for (int j = 0; j < numPoints; j++) {
        bool seen = false;
        for (auto& row : observations) if (row.first >= 0 && row.first < numCameras && row.second.count(j)) { seen = true; break; }
        if (!seen) points.erase(j);
    }
}
Pose SfM::GetPose(int camera) {
    auto it = cameras.find(camera); if (it == cameras.end()) return Pose();
    const Camera& c = it->second; return Pose(Vec3(c[0], c[1], c[2]), Vec3(c[3], c[4], c[5]));
}
void SfM::SetPose(int camera, const Pose& p) {
    auto it = cameras.find(camera); const Camera c{p.t.v[0], p.t.v[1], p.t.v[2], p.r.v[0], p.r.v[1], p.r.v[2]};
    if (it == cameras.end()) { obs_cache.stale = true; cameras[camera] = c; } else it->second = c;      // a NEW key changes which observations count
}
Point SfM::GetPoint(int point) { auto it = points.find(point); return it == points.end() ? Point(0, 0, 0) : it->second; }
void SfM::SetPoint(int point, const Point& X) { auto it = points.find(point); if (it == points.end()) { obs_cache.stale = true; points[point] = X; } else it->second = X; }

// dense index spaces [0,numCameras) x [0,numPoints) as the reference's loops use them; absent entries stay absent
void SfM::Flatten(FlatProblem& F) {
    F.cam.assign((size_t)numCameras * 6, 0.0); F.pts.assign((size_t)numPoints * 3, 0.0);
    F.rf.assign(numCameras, 1); F.tf.assign(numCameras, 1); F.pf.assign(numPoints, 0);
    std::vector<char> cam_ok(numCameras, 0), pt_ok(numPoints, 0);                             // dense existence masks: one map walk instead of a lookup per observation
    for (auto& kv : cameras) if (kv.first >= 0 && kv.first < numCameras) {
        for (int k = 0; k < 6; k++) F.cam[(size_t)kv.first * 6 + k] = kv.second[k];
        cam_ok[kv.first] = 1;
    }
    for (auto& kv : rotationFixed) if (kv.first >= 0 && kv.first < numCameras && cam_ok[kv.first]) F.rf[kv.first] = kv.second;
    for (auto& kv : translationFixed) if (kv.first >= 0 && kv.first < numCameras && cam_ok[kv.first]) F.tf[kv.first] = kv.second;
    for (int c = 0; c < numCameras; c++) if (cam_ok[c]) { if (!rotationFixed.count(c)) F.rf[c] = 0; if (!translationFixed.count(c)) F.tf[c] = 0; }   // (operator[] of the flags: absent = false)
    for (auto&& kv : points) if (kv.first >= 0 && kv.first < numPoints) {
        for (int k = 0; k < 3; k++) F.pts[(size_t)kv.first * 3 + k] = kv.second.v[k];
        pt_ok[kv.first] = 1;
    }
    for (auto&& kv : pointFixed) if (kv.first >= 0 && kv.first < numPoints && pt_ok[kv.first]) F.pf[kv.first] = kv.second;
    ObsCache& C = obs_cache;
    if (C.stale) {
        // counting sort by point over the camera-major maps: rows come in ascending camera order and every row in ascending point order, so the observations of a
        // point land in ascending camera order -- exactly the order the reference's build loop visits them in (src/sfm.cpp:240-263)
        std::vector<int64_t> start((size_t)numPoints + 1, 0);
        for (auto& row : observations) {
            if (row.first < 0 || row.first >= numCameras || !cam_ok[row.first]) continue;      // src/sfm.cpp:249 / :167
            for (auto& kv : row.second) if (kv.first >= 0 && kv.first < numPoints && pt_ok[kv.first]) start[(size_t)kv.first + 1]++;   // src/sfm.cpp:242 / :161
        }
        for (int j = 0; j < numPoints; j++) start[(size_t)j + 1] += start[j];
        const size_t M = (size_t)start[numPoints];
        C.xy.resize(2 * M); C.oc.resize(M); C.op.resize(M);
        for (auto& row : observations) {
            if (row.first < 0 || row.first >= numCameras || !cam_ok[row.first]) continue;
            for (auto& kv : row.second) {
                if (kv.first < 0 || kv.first >= numPoints || !pt_ok[kv.first]) continue;
                const size_t w = (size_t)start[kv.first]++;
                C.xy[2 * w] = kv.second.x; C.xy[2 * w + 1] = kv.second.y; C.oc[w] = row.first; C.op[w] = kv.first;
            }
        }
        C.stale = false;
    }
    GetContext();
    ssfm_ba_problem& P = F.P;
    P.num_cameras = numCameras; P.num_points = numPoints; P.num_observations = (int64_t)C.oc.size();
    P.cameras = F.cam.data(); P.points = F.pts.data(); P.focal = &intrinsics.focal;
    P.obs_xy = C.xy.data(); P.obs_cam = C.oc.data(); P.obs_pt = C.op.data();
    P.rot_fixed = F.rf.data(); P.trans_fixed = F.tf.data(); P.pt_fixed = F.pf.data(); P.focal_fixed = focalFixed ? 1 : 0;
}

// src/sfm.cpp:156-192: every existing point is re-estimated from its observations (one GPU lane per point instead of the
// cv::parallel_for_ over LO-MSAC objects); < 3 observations or < 3 inliers -> (0,0,0)
void SfM::Retriangulate() {
    if (numCameras == 0 || numPoints == 0) return;
    FlatProblem F; Flatten(F);
    int rc = retriangulateMode < 0 ? ssfm_retriangulate(ctx, &F.P, nullptr) : ssfm_retriangulate_mode(ctx, &F.P, retriangulateMode, nullptr, nullptr, nullptr);
    if (rc != SSFM_OK) { std::cout << "error: " << ssfm_last_error(ctx) << "\n"; exit(1); }
    for (auto&& kv : points) if (kv.first >= 0 && kv.first < numPoints) for (int k = 0; k < 3; k++) kv.second.v[k] = F.pts[(size_t)kv.first * 3 + k];
}

bool SfM::Optimize() {
    if (numCameras == 0 || numPoints == 0) return false;                                      // src/sfm.cpp:230
    std::cout << "\tBuilding BA problem...\n";
    const bool timing = std::getenv("SSFM_PLAN_TIMING") != nullptr;
    const auto t_0 = std::chrono::steady_clock::now();
    FlatProblem F; Flatten(F);
    const auto t_1 = std::chrono::steady_clock::now();
    ssfm_ba_problem& P = F.P; std::vector<double>&cam = F.cam, &pts = F.pts;
    ssfm_ba_options O; ssfm_ba_default_options(&O);                                            // src/sfm.cpp:194-212
    O.verbose = 1;                                                                            // minimizer_progress_to_stdout
    int rc = ssfm_ba_solve(ctx, &P, &O, &last_summary);
    const auto t_2 = std::chrono::steady_clock::now();
    if (rc != SSFM_OK) { std::cout << "error: " << ssfm_last_error(ctx) << "\n"; exit(1); }
    if (last_summary.termination == SSFM_NOTHING_TO_DO) { std::cout << "didn't add any cameras\n"; return false; }   // src/sfm.cpp:265-268
    std::cout << "Running optimizer...\n\t" << 2 * last_summary.num_residual_blocks << " residuals\n";
    std::printf("iterations %d  initial cost %.6e  final cost %.6e  termination %d  solve %.3f s\n", last_summary.iterations,
                last_summary.initial_cost, last_summary.final_cost, last_summary.termination, last_summary.t_solve_s);
    if (last_summary.termination == SSFM_FAILURE) { std::cout << "error: ceres failed.\n"; exit(1); }             // src/sfm.cpp:278-282
    for (auto& kv : cameras) if (kv.first >= 0 && kv.first < numCameras) for (int k = 0; k < 6; k++) kv.second[k] = cam[(size_t)kv.first * 6 + k];
    for (auto&& kv : points) if (kv.first >= 0 && kv.first < numPoints) for (int k = 0; k < 3; k++) kv.second.v[k] = pts[(size_t)kv.first * 3 + k];
    if (timing) { auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
                  std::fprintf(stderr, "[mirror] Optimize: flatten %.2f ms, ssfm_ba_solve %.2f ms, write-back %.2f ms\n", ms(t_0, t_1), ms(t_1, t_2), ms(t_2, std::chrono::steady_clock::now())); }
    return last_summary.termination == SSFM_CONVERGENCE;                                      // src/sfm.cpp:289
}

void SfM::Apply(const Pose& pose) {                                                           // src/sfm.cpp:341-362
    Pose inv = pose.inverse();
    for (int i = 0; i < numCameras; i++) { Pose c = GetPose(i); c.postMultiply(inv); SetPose(i, c); }
    // (the reference probes j = 0 .. numPoints-1 with exists(): the same points in the same order as walking the map)
    for (auto&& kv : points) {
        if (kv.first < 0 || kv.first >= numPoints || kv.second.norm() == 0) continue;
        kv.second = pose.apply(kv.second);
    }
}
void SfM::Apply(double scale) {                                                               // src/sfm.cpp:364-382
    for (int i = 0; i < numCameras; i++) { Pose c = GetPose(i); for (int k = 0; k < 3; k++) c.t.v[k] *= scale; SetPose(i, c); }
    for (auto&& kv : points) {
        if (kv.first < 0 || kv.first >= numPoints || kv.second.norm() == 0) continue;
        for (int k = 0; k < 3; k++) kv.second.v[k] *= scale;
    }
}
void SfM::Unapply(const Pose& pose) {                                                         // src/sfm.cpp:384-402
    for (int i = 0; i < numCameras; i++) { Pose c = GetPose(i); c.postMultiply(pose); SetPose(i, c); }
    Pose inv = pose.inverse();
    for (int j = 0; j < numPoints; j++) SetPoint(j, inv.apply(GetPoint(j)));
}
void SfM::Normalize(bool inward) {                                                            // src/sfm.cpp:535-571
    Vec3 centroid;
    for (int i = 0; i < numCameras; i++) { Vec3 c = GetPose(i).getCenter(); for (int k = 0; k < 3; k++) centroid.v[k] += c.v[k]; }
    for (int k = 0; k < 3; k++) centroid.v[k] /= numCameras;
    Apply(Pose(Vec3(-centroid.v[0], -centroid.v[1], -centroid.v[2]), Vec3(0, 0, 0)));
    double avg = 0; for (int i = 0; i < numCameras; i++) avg += GetPose(i).getCenter().norm();
    avg /= numCameras;
    Apply(1.0 / avg);
    const double tz = GetPose(0).t.v[2];
    if ((inward && tz < 0) || (!inward && tz > 0)) { std::cout << "inverted! flipping to correct\n"; Apply(-1.0); }
}
void SfM::WritePoses(const std::string& path, const std::vector<int>& indices) {              // src/sfm.cpp:463-480
    FILE* f = std::fopen(path.c_str(), "w"); if (!f) return;
    for (int i = 0; i < numCameras; i++) {
        std::fprintf(f, "%d ", indices[i]);
        const Camera c = cameras[i];
        for (int j = 0; j < 6; j++) std::fprintf(f, "%.15lf ", c[j]);
        std::fprintf(f, "\n");
    }
    std::fclose(f);
}

void SfM::WritePointsOBJ(const std::string& path) {                                           // src/sfm.cpp:482-519
    FILE* f = std::fopen(path.c_str(), "w"); if (!f) return;
    std::vector<double> distances(numPoints, 0.0);             // distance to the LAST camera that observes the point, as the reference's loop leaves it
    for (int i = 0; i < numCameras; i++) {
        const Vec3 center = GetPose(i).getCenter();
        auto row = observations.find(i); if (row == observations.end()) continue;
        for (auto& kv : row->second) {
            const int j = kv.first; if (j < 0 || j >= numPoints || !points.count(j)) continue;
            const Point X = GetPoint(j);
            distances[j] = Vec3(X.v[0] - center.v[0], X.v[1] - center.v[1], X.v[2] - center.v[2]).norm();
        }
    }
    for (int i = 0; i < numPoints; i++) {
        if (!points.count(i)) continue;
        if (distances[i] > 2000.) continue;
        const Point X = GetPoint(i);
        if (X.norm() == 0) continue;
        std::fprintf(f, "v %0.15lf %0.15lf %0.15lf\n", X.v[0], X.v[1], X.v[2]);
    }
    std::fclose(f);
}
void SfM::WriteCameraCentersOBJ(const std::string& path) {                                    // src/sfm.cpp:521-533
    FILE* f = std::fopen(path.c_str(), "w"); if (!f) return;
    for (int i = 0; i < numCameras; i++) { const Vec3 c = GetPose(i).getCenter(); std::fprintf(f, "v %0.15lf %0.15lf %0.15lf\n", c.v[0], c.v[1], c.v[2]); }
    std::fclose(f);
}
// COLMAP text model (the byte layout of src/sfm.cpp:573-647: header comments, "%lf" fields, 1-based ids, pixel coordinates with the
// principal point added back, colours stored BGR and written RGB, points at the origin skipped everywhere).  Written from one pass over
// the sparse observation rows: each image line is emitted while its row is walked, and the (image, index-in-line) pairs of a point's
// track are collected as flat records that a stable sort by point id turns into the TRACK[] lists of points3D.txt.
namespace {
struct TextOut {
    FILE* f;
    explicit TextOut(const std::string& path) : f(std::fopen(path.c_str(), "w")) {}
    ~TextOut() { if (f) std::fclose(f); }
    void lines(std::initializer_list<const char*> ls) { for (const char* l : ls) std::fputs(l, f); }
};
struct TrackEntry { int point, image, slot; };
// Eigen::Quaterniond(Eigen::AngleAxisd(|r|, r / |r|)): (cos(th/2), sin(th/2) r/|r|); the identity for r = 0
void angle_axis_to_quaternion(const Vec3& r, double q[4]) {
    const double th = r.norm();
    q[0] = 1; q[1] = q[2] = q[3] = 0;
    if (th == 0) return;
    const double k = std::sin(0.5 * th) / th;
    q[0] = std::cos(0.5 * th); q[1] = k * r.v[0]; q[2] = k * r.v[1]; q[3] = k * r.v[2];
}
}  // namespace

void SfM::WriteCOLMAP(const std::string& sparse_dir, int width, int height) {
    mkdir(sparse_dir.c_str(), 0777);
    {
        TextOut out(sparse_dir + "/cameras.txt"); if (!out.f) return;
        out.lines({"# Camera list with one line of data per camera:\n", "#   CAMERA_ID, MODEL, WIDTH, HEIGHT, PARAMS[]\n", "# Number of cameras: 1\n"});
        std::fprintf(out.f, "1 SIMPLE_PINHOLE %d %d %lf %lf %lf\n", width, height, intrinsics.focal, intrinsics.centerx, intrinsics.centery);
    }
    const int nc = GetNumCameras(), np = GetNumPoints();
    std::vector<char> live(np, 0);                                  // points that exist and are not at the origin
    for (const auto& kv : points) if (kv.first >= 0 && kv.first < np && kv.second.norm() != 0) live[kv.first] = 1;
    std::vector<TrackEntry> tracks;
    {
        TextOut out(sparse_dir + "/images.txt"); if (!out.f) return;
        out.lines({"# Image list with two lines of data per image:\n", "#   IMAGE_ID, QW, QX, QY, QZ, TX, TY, TZ, CAMERA_ID, NAME\n", "#   POINTS2D[] as (X, Y, POINT3D_ID)\n"});
        std::fprintf(out.f, "# Number of images: %d, mean observations per image:\n", nc);
        for (int image = 0; image < nc; image++) {
            const Pose pose = GetPose(image);
            double q[4]; angle_axis_to_quaternion(pose.r, q);
            std::fprintf(out.f, "%d %lf %lf %lf %lf %lf %lf %lf 1 %s\n", image + 1, q[0], q[1], q[2], q[3], pose.t.v[0], pose.t.v[1], pose.t.v[2], paths[image].c_str());
            const auto row = observations.find(image);
            if (row != observations.end() && cameras.count(image)) {
                int slot = 0;
                for (const auto& ob : row->second) {                  // ascending point id
                    if (ob.first < 0 || ob.first >= np || !live[ob.first]) continue;
                    std::fprintf(out.f, "%lf %lf %d ", ob.second.x + intrinsics.centerx, ob.second.y + intrinsics.centery, ob.first + 1);
                    tracks.push_back(TrackEntry{ob.first, image + 1, slot++});
                }
            }
            std::fputc('\n', out.f);
        }
    }
    std::stable_sort(tracks.begin(), tracks.end(), [](const TrackEntry& x, const TrackEntry& y) { return x.point < y.point; });   // images stay ascending inside a point
    TextOut out(sparse_dir + "/points3D.txt"); if (!out.f) return;
    out.lines({"# 3D point list with one line of data per point:\n", "#   POINT3D_ID, X, Y, Z, R, G, B, ERROR, TRACK[] as (IMAGE_ID, POINT2D_IDX)\n"});
    std::fprintf(out.f, "# Number of points: %d, mean track length: \n", np);
    size_t cursor = 0;
    for (int id = 0; id < np; id++) {
        if (!live[id]) { while (cursor < tracks.size() && tracks[cursor].point == id) cursor++; continue; }
        const Point X = GetPoint(id); const std::array<unsigned char, 3> bgr = GetColor(id);
        std::fprintf(out.f, "%d %lf %lf %lf %d %d %d 0 ", id + 1, X.v[0], X.v[1], X.v[2], bgr[2], bgr[1], bgr[0]);
        for (; cursor < tracks.size() && tracks[cursor].point == id; cursor++) std::fprintf(out.f, "%d %d ", tracks[cursor].image, tracks[cursor].slot);
        std::fputc('\n', out.f);
    }
}
void SfM::WriteCalib(const std::string& path) {                                               // run_spherical_sfm_uncalib.cpp:225-228
    FILE* f = std::fopen(path.c_str(), "w"); if (!f) return;
    std::fprintf(f, "%0.15f %0.15f %0.15f\n", GetFocal(), intrinsics.centerx, intrinsics.centery);
    std::fclose(f);
}
// Drops every observation whose reprojection error exceeds thresh pixels, for points that are seen by at least three cameras and are not
// at the origin; a point that loses all its observations is moved to the origin (so later Optimize() calls skip it).  Same outcome as
// src/sfm.cpp:297-339 -- the removals of a point do not depend on each other, so instead of probing every (point, camera) key this counts
// the observations of all points in one sweep over the sparse rows and tests them in a second one.
void SfM::FilterObservations(double thresh) {
    obs_cache.stale = true;
    std::vector<int> seen(numPoints, 0);
    auto usable_row = [&](int cam) { return cam >= 0 && cam < numCameras && cameras.count(cam) != 0; };
    for (const auto& row : observations)
        if (usable_row(row.first)) for (const auto& ob : row.second) if (ob.first >= 0 && ob.first < numPoints) seen[ob.first]++;
    std::vector<char> tested(numPoints, 0);
    for (const auto& kv : points) if (kv.first >= 0 && kv.first < numPoints && kv.second.norm() != 0 && seen[kv.first] >= 3) tested[kv.first] = 1;
    std::vector<int> left(seen);
    int dropped = 0;
    for (auto& row : observations) {
        if (!usable_row(row.first)) continue;
        const Camera& cam = cameras[row.first];
        double R[9]; ssfm::angle_axis_to_matrix(&cam[3], R);                  // ReprojectionError without the loss (src/sfm.cpp:38-63)
        for (auto it = row.second.begin(); it != row.second.end();) {
            const int id = it->first;
            bool drop = false;
            if (id >= 0 && id < numPoints && tested[id]) {
                double p[3]; ssfm::mat3_vec(R, points[id].v, p);
                p[0] += cam[0]; p[1] += cam[1]; p[2] += cam[2];
                const double ex = intrinsics.focal * (p[0] / p[2]) - it->second.x, ey = intrinsics.focal * (p[1] / p[2]) - it->second.y;
                drop = std::sqrt(ex * ex + ey * ey) > thresh;
            }
            if (drop) { it = row.second.erase(it); left[id]--; dropped++; } else ++it;
        }
    }
    for (int id = 0; id < numPoints; id++) if (tested[id] && left[id] == 0) SetPoint(id, Point(0, 0, 0));
    std::cout << "removed " << dropped << " observations\n";
}

}  // namespace sphericalsfm
