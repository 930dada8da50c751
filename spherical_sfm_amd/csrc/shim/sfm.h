// spherical_sfm_amd -- C++ host-side mirror of the reference's BA container `sphericalsfm::SfM`
// (reference include/sphericalsfm/sfm.h:15-105, src/sfm.cpp), written against the C ABI of include/ssfm.h.
//
// Same class name, method names, argument meaning and error behaviour for the hot path, so that a caller such as
// build_sfm / run_spherical_sfm_uncalib (examples/spherical_sfm_tools.cpp:862-955, examples/run_spherical_sfm_uncalib.cpp:176-211)
// compiles against it unchanged apart from the vector types: Eigen / OpenCV / Ceres are not available in this build,
// so Vec3 / Pose below are small PODs with the member names the reference uses (t, r, getCenter, ...).
// Storage is map-based like the reference's SparseVector / SparseMatrix (include/sphericalsfm/sparse.hpp), keyed
// observations[camera][point]; Optimize() flattens by walking the sparse rows (not the O(Np*Nc) probe loop of
// src/sfm.cpp:240-263) and hands caller-owned buffers to ssfm_ba_solve.
#pragma once
#include <array>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <map>
#include <string>
#include <vector>
#include "../../../include/ssfm.h"
// SSFM_WITH_EIGEN (SURVEY 8b "Build constraint"; reference include/sphericalsfm/sfm.h:4-13, sfm_types.h:3-29): a maintainer who HAS Eigen defines it and the
// PODs below convert to and from the Eigen types of the reference's signatures implicitly -- sfm.AddPoint(Eigen::Vector3d(...)), Eigen::Vector3d X(sfm.GetPoint(j)) / sfm.GetPoint(j).eigen() (POD -> Eigen is explicit),
// Pose(Eigen::Vector3d t, Eigen::Vector3d r), pose.P() -- so a caller written against the reference's headers compiles against these.  Eigen is not in this
// image: this block is written from Eigen's documented interface and has NOT been compiled here (INTEGRATION.md says so too).
#ifdef SSFM_WITH_EIGEN
#include <Eigen/Core>
#endif

namespace sphericalsfm {

struct Vec3 {
    double v[3];
    Vec3() : v{0, 0, 0} {}
    Vec3(double x, double y, double z) : v{x, y, z} {}
#ifdef SSFM_WITH_EIGEN
    Vec3(const Eigen::Vector3d& e) : v{e(0), e(1), e(2)} {}                                 // Eigen -> POD implicit (arguments of AddPoint, SetPoint, Pose(...))
    explicit operator Eigen::Vector3d() const { return Eigen::Vector3d(v[0], v[1], v[2]); }   // POD -> Eigen explicit: one implicit direction only, so that mixed expressions are not ambiguous
    Eigen::Vector3d eigen() const { return Eigen::Vector3d(v[0], v[1], v[2]); }
#endif
    double& operator()(int i) { return v[i]; }
    double operator()(int i) const { return v[i]; }
    double& operator[](int i) { return v[i]; }
    double operator[](int i) const { return v[i]; }
    double norm() const;
};
typedef Vec3 Point;                       // include/sphericalsfm/sfm_types.h:8
typedef std::array<double, 6> Camera;     // [t;r], sfm_types.h:9
struct Observation {
    double x, y;
    Observation() : x(0), y(0) {}
    Observation(double _x, double _y) : x(_x), y(_y) {}
#ifdef SSFM_WITH_EIGEN                    // typedef Eigen::Vector2d Observation, sfm_types.h:12
    Observation(const Eigen::Vector2d& e) : x(e(0)), y(e(1)) {}
    explicit operator Eigen::Vector2d() const { return Eigen::Vector2d(x, y); }
    Eigen::Vector2d eigen() const { return Eigen::Vector2d(x, y); }
    double operator()(int i) const { return i == 0 ? x : y; }
#endif
};

struct Pose {                             // sfm_types.h:14-29 / src/sfm_types.cpp
    Vec3 t, r;
    double R[9];                          // row-major rotation so3exp(r) (the 3x3 block of the reference's P)
    Pose();
    Pose(const Vec3& _t, const Vec3& _r);
#ifdef SSFM_WITH_EIGEN                    // the reference's member `Eigen::Matrix4d P` (sfm_types.h:18) as a function: [R t; 0 0 0 1]
    Eigen::Matrix4d P() const { Eigen::Matrix4d M = Eigen::Matrix4d::Identity(); for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) M(i, j) = R[3 * i + j]; M(i, 3) = t(i); } return M; }
#endif
    Pose inverse() const;
    void postMultiply(const Pose& pose);
    Point apply(const Point& point) const;
    Point applyInverse(const Point& point) const;
    Vec3 getCenter() const;
};

struct Intrinsics {                       // sfm_types.h:31-47
    double focal, centerx, centery;
    Intrinsics(double _focal, double _centerx, double _centery) : focal(_focal), centerx(_centerx), centery(_centery) {}
};

// The slice of std::map<int, T>'s interface the mirror uses -- operator[], find / end, count, erase, iteration in ascending key order with .first / .second -- over
// DENSE storage: point ids are small non-negative integers issued in order (AddPoint), and at 170 000 points every walk over a std::map (Flatten, the write-back after
// Optimize / Retriangulate, Apply: ten per driver run) cost ~5 ms of pointer chasing against 0.1 ms over a vector (the reference's SparseVector is a std::map too:
// sparse.hpp:9-59; what callers see -- which ids exist, in which order they are visited -- is the same).
template <class T> class IndexedMap {
    std::vector<T> val; std::vector<char> has; size_t n = 0;
    static void check(int k) { if (k < 0) { std::fprintf(stderr, "IndexedMap: negative id %d\n", k); std::abort(); } }
public:
    struct Ref { const int first; T& second; };
    struct CRef { const int first; const T& second; };
    template <class M, class R> class Iter {
        M* m; int i;
        void skip() { while (i < (int)m->has.size() && !m->has[i]) i++; }
        struct Arrow { R r; R* operator->() { return &r; } };
    public:
        Iter(M* m_, int i_) : m(m_), i(i_) { skip(); }
        R operator*() const { return R{i, m->val[i]}; }
        Arrow operator->() const { return Arrow{R{i, m->val[i]}}; }
        Iter& operator++() { i++; skip(); return *this; }
        bool operator==(const Iter& o) const { return i == o.i; }
        bool operator!=(const Iter& o) const { return i != o.i; }
    };
    using iterator = Iter<IndexedMap, Ref>; using const_iterator = Iter<const IndexedMap, CRef>;
    friend iterator; friend const_iterator;
    iterator begin() { return iterator(this, 0); }
    iterator end() { return iterator(this, (int)has.size()); }
    const_iterator begin() const { return const_iterator(this, 0); }
    const_iterator end() const { return const_iterator(this, (int)has.size()); }
    iterator find(int k) { return (k >= 0 && k < (int)has.size() && has[k]) ? iterator(this, k) : end(); }
    const_iterator find(int k) const { return (k >= 0 && k < (int)has.size() && has[k]) ? const_iterator(this, k) : end(); }
    size_t count(int k) const { return (k >= 0 && k < (int)has.size() && has[k]) ? 1 : 0; }
    T& operator[](int k) {
        check(k);
        if (k >= (int)has.size()) { const size_t want = std::max<size_t>((size_t)k + 1, has.size() + has.size() / 2); val.resize(want); has.resize(want, 0); }
        if (!has[k]) { has[k] = 1; val[k] = T(); n++; }
        return val[k];
    }
    size_t erase(int k) { if (!count(k)) return 0; has[k] = 0; val[k] = T(); n--; return 1; }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    void clear() { val.clear(); has.clear(); n = 0; }
};

// One row of the observation table (the points a camera sees): the same slice of std::map<int, T>'s interface over a SORTED VECTOR of (id, value) pairs.  A row is written in
// ascending point order by the synthetic drivers (push_back); anything else goes through a small merge buffer (below).  The first Flatten of a run walks all rows
// twice: 2 x 1.02 M std::map nodes took ~60 ms at configs[2] size, the vectors ~5.
//
// WHERE THESE TWO CONTAINERS PROMISE LESS THAN std::map (the reference's sparse.hpp): (1) a reference returned by operator[] is valid only until the NEXT insert
// into the same container (IndexedMap grows its vectors, FlatMap merges its buffer) -- copy the value before inserting, as MergePoint does; std::map references
// stay valid for the node's life.  (2) const access is NOT safe for concurrent readers: FlatMap's begin / end / find / count merge the pending buffer first
// (`mutable`), so two threads reading one row at once race unless the row was frozen before -- call freeze() on every row (SfM::FreezeObservations) ahead of a
// parallel read-only phase; after it the const accessors touch nothing until the next insert.  The mirror itself is single-threaded, like the reference's SfM.
template <class T> class FlatMap {
    // Out-of-order inserts (build_sfm hands a keyframe's observations over in feature order, i.e. in random track order) wait in a small unsorted buffer and are merged
    // in when it holds 64 of them: a random insert into a 2000-entry row costs a search + 1/64 of a merge instead of moving half the row.
    mutable std::vector<std::pair<int, T>> v;          // sorted by id, unique
    mutable std::vector<std::pair<int, T>> pend;       // ids not in v, unique, unsorted
    struct KeyLess { bool operator()(const std::pair<int, T>& a, int k) const { return a.first < k; } };
    void flush() const {
        if (pend.empty()) return;
        std::sort(pend.begin(), pend.end(), [](const std::pair<int, T>& a, const std::pair<int, T>& b) { return a.first < b.first; });
        const size_t mid = v.size();
        v.insert(v.end(), pend.begin(), pend.end());
        std::inplace_merge(v.begin(), v.begin() + (std::ptrdiff_t)mid, v.end(), [](const std::pair<int, T>& a, const std::pair<int, T>& b) { return a.first < b.first; });
        pend.clear();
    }
public:
    using iterator = typename std::vector<std::pair<int, T>>::iterator;
    using const_iterator = typename std::vector<std::pair<int, T>>::const_iterator;
    iterator begin() { flush(); return v.begin(); }
    iterator end() { flush(); return v.end(); }
    const_iterator begin() const { flush(); return v.begin(); }
    const_iterator end() const { flush(); return v.end(); }
    iterator find(int k) { flush(); auto it = std::lower_bound(v.begin(), v.end(), k, KeyLess()); return (it != v.end() && it->first == k) ? it : v.end(); }
    const_iterator find(int k) const { flush(); auto it = std::lower_bound(v.begin(), v.end(), k, KeyLess()); return (it != v.end() && it->first == k) ? it : v.end(); }
    size_t count(int k) const { flush(); auto it = std::lower_bound(v.begin(), v.end(), k, KeyLess()); return (it != v.end() && it->first == k) ? 1 : 0; }
    T& operator[](int k) {
        if (pend.empty() && (v.empty() || v.back().first < k)) { v.emplace_back(k, T()); return v.back().second; }      // ids in ascending order: the common case
        auto it = std::lower_bound(v.begin(), v.end(), k, KeyLess());
        if (it != v.end() && it->first == k) return it->second;
        for (auto& e : pend) if (e.first == k) return e.second;
        pend.emplace_back(k, T());
        if (pend.size() > 64) { flush(); return std::lower_bound(v.begin(), v.end(), k, KeyLess())->second; }
        return pend.back().second;
    }
    size_t erase(int k) { auto it = find(k); if (it == v.end()) return 0; v.erase(it); return 1; }
    iterator erase(iterator it) { return v.erase(it); }          // (it comes from begin() / find(): the buffer is merged)
    void freeze() { flush(); }                                   // merge the pending inserts now: const access is then read-only (see the note above)
    bool frozen() const { return pend.empty(); }
    size_t size() const { return v.size() + pend.size(); }
    bool empty() const { return v.empty() && pend.empty(); }
};

class SfM {
protected:
    Intrinsics intrinsics;
    std::map<int, Camera> cameras;
    IndexedMap<Point> points;
    std::map<int, FlatMap<Observation>> observations;          // [camera][point]: rows as sorted vectors (FlatMap above)
    std::map<int, std::map<int, Pose>> measurements;           // [camera][camera], include/sphericalsfm/sfm.h:27 (the reference never writes it either)
    std::map<int, std::string> paths;
    std::map<int, std::array<unsigned char, 3>> colors;        // BGR like cv::Vec3b (the reference's SparseVector<cv::Vec3b>)
    std::map<int, std::vector<float>> descriptors;             // SparseVector<cv::Mat>, one row of floats per point
    bool focalFixed;
    std::map<int, bool> rotationFixed, translationFixed;
    IndexedMap<char> pointFixed;
    int numCameras, numPoints, nextCamera, nextPoint;
    ssfm_ctx* ctx;                        // created on first Optimize()
    ssfm_ba_summary last_summary;
    struct FlatProblem {                  // the flat arrays the C ABI takes + the struct pointing at them
        std::vector<double> cam, pts; std::vector<uint8_t> rf, tf, pf; ssfm_ba_problem P;
    };
    // The observation arrays of the flat problem, point-major / cameras ascending (what ssfm_ba_solve plans fastest: no sort), kept between calls: the drivers call
    // Optimize / Retriangulate up to six times on one observation set, and walking a million std::map nodes per call cost more than the solve itself
    // (bench.py side_paths.pipeline_configs2).  Any call that changes which cameras, points or observations exist marks them stale.
    struct ObsCache { std::vector<double> xy; std::vector<int32_t> oc, op; bool stale = true; } obs_cache;
    void Flatten(FlatProblem& F);
public:
    explicit SfM(const Intrinsics& _intrinsics);
    ~SfM();
    SfM(const SfM&) = delete;
    SfM& operator=(const SfM&) = delete;

    Intrinsics GetIntrinsics() const { return intrinsics; }
    int AddCamera(const Pose& initial_pose, const std::string& path = "");
    int AddPoint(const Point& initial_position);
    int AddPoint(const Point& initial_position, const std::array<unsigned char, 3>& color_bgr);
    // cv::Mat descriptor -> a flat float vector (128 floats for SIFT): opaque to this path, kept per point and erased with it (src/sfm.cpp:113-127,443)
    int AddPoint(const Point& initial_position, const std::vector<float>& descriptor, const std::array<unsigned char, 3>& color_bgr = {{0, 0, 0}});
    std::vector<float> GetDescriptor(int point);                                                // include/sphericalsfm/sfm.h:72
    // (GetMeasurement, include/sphericalsfm/sfm.h:71, is declared by the reference and defined nowhere in it: nothing to mirror)
    std::array<unsigned char, 3> GetColor(int point);
    void AddObservation(int camera, int point, const Observation& observation);
    void RemoveCamera(int camera);
    void RemovePoint(int point);
    void MergePoint(int point1, int point2);     // point2 will be removed
    int GetNumCameras() { return numCameras; }
    int GetNumPoints() { return numPoints; }
    bool GetObservation(int camera, int point, Observation& observation);
    bool GetMeasurement(int i, int j, Pose& measurement);      // include/sphericalsfm/sfm.h:71 -- declared there and defined nowhere in the reference; here: lookup in `measurements`

    void Retriangulate();                         // src/sfm.cpp:156-192
    int retriangulateMode = -1;                   // -1: the library default (trace replay); SSFM_RETRI_MODE_TRACE / SSFM_RETRI_MODE_ENUMERATE (not in the reference)
    bool Optimize();                              // src/sfm.cpp:228-290: true iff CONVERGENCE; exit(1) on FAILURE

    void Apply(const Pose& pose);
    void Apply(double scale);
    void Unapply(const Pose& pose);
    void Normalize(bool inward);
    double GetFocal() { return intrinsics.focal; }
    Pose GetPose(int camera);
    void SetPose(int camera, const Pose& pose);
    Point GetPoint(int point);
    void SetPoint(int point, const Point& position);
    void SetFocalFixed(bool fixed) { focalFixed = fixed; }
    void SetRotationFixed(int camera, bool fixed) { rotationFixed[camera] = fixed; }
    void SetTranslationFixed(int camera, bool fixed) { translationFixed[camera] = fixed; }
    void SetPointFixed(int point, bool fixed) { pointFixed[point] = fixed; }
    void WritePoses(const std::string& path, const std::vector<int>& indices);        // src/sfm.cpp:463-480
    void WritePointsOBJ(const std::string& path);                                    // src/sfm.cpp:482-519
    void WriteCameraCentersOBJ(const std::string& path);                             // src/sfm.cpp:521-533
    void WriteCOLMAP(const std::string& sparse_dir, int width, int height);          // src/sfm.cpp:573-647
    void WriteCalib(const std::string& path);                                        // run_spherical_sfm_uncalib.cpp:225-228
    void FilterObservations(double thresh);                                          // src/sfm.cpp:297-339
    void FreezeObservations() { for (auto& row : observations) row.second.freeze(); }   // before a parallel read-only phase over this object (see FlatMap)
    const ssfm_ba_summary& LastSummary() const { return last_summary; }
    ssfm_ctx* GetContext();                       // the library context of this object (created on first use); for the tools around it
};

}  // namespace sphericalsfm
