// On-disk formats of the drop-in on a small deterministic scene, no GPU needed:
// poses.txt, points.obj, cameras.obj, sparse/{cameras,images,points3D}.txt, calib.txt, plus FilterObservations.
// tests/test_formats_cpu.py rebuilds the same scene in Python and compares the files byte for byte with the
// reference's format strings (src/sfm.cpp:463-533,573-647; examples/run_spherical_sfm_uncalib.cpp:215-228).
#include <cmath>
#include <cstdio>
#include <string>
#include "sfm.h"
using namespace sphericalsfm;

int main(int argc, char** argv) {
    if (argc < 2) { std::printf("usage: demo_formats <output dir>\n"); return 2; }
    const std::string out = argv[1];
    const int Nc = 5, Np = 12;
    const double focal = 800.0;
    SfM sfm(Intrinsics(focal, 320.0, 240.0));
    for (int i = 0; i < Nc; i++) {
        const Vec3 r(0.01 * i, 0.2 * i - 0.3, i == 0 ? 0.0 : -0.02 * i);          // camera 0: r has a zero z, camera index 0 keeps |r| != 0
        char name[64]; std::snprintf(name, sizeof name, "images/%06d.jpg", 10 * i + 1);
        sfm.AddCamera(Pose(Vec3(0.05 * i, -0.01 * i, -1.0), i == 2 ? Vec3(0, 0, 0) : r), name);   // camera 2: identity rotation (quaternion branch)
    }
    for (int j = 0; j < Np; j++) {
        const Point X(0.4 * std::cos(0.7 * j), 0.3 * std::sin(1.3 * j), 4.0 + 0.25 * j);
        sfm.AddPoint(j == 7 ? Point(0, 0, 0) : X, std::array<unsigned char, 3>{{(unsigned char)(10 * j), (unsigned char)(255 - 10 * j), (unsigned char)(3 * j)}});   // point 7 is "removed" (zero)
        for (int i = 0; i < Nc; i++) {
            if ((i + j) % 4 == 3) continue;                                       // ragged tracks
            const Pose P = sfm.GetPose(i);
            const Point q = P.apply(X);
            double ox = focal * q.v[0] / q.v[2], oy = focal * q.v[1] / q.v[2];
            if (j == 3 && i == 1) { ox += 25.0; oy -= 40.0; }                      // one gross outlier for FilterObservations
            if (j == 5) { ox += 0.5 * (i - 2); }                                   // small inlier noise
            sfm.AddObservation(i, j, Observation(ox, oy));
        }
    }
    sfm.RemovePoint(10);                                                          // an absent point id
    // raw state for the test: it formats these numbers itself with the reference's format strings
    auto dump = [&](const std::string& path) {
        FILE* f = std::fopen(path.c_str(), "wb"); if (!f) return;
        int hdr[2] = {Nc, Np}; std::fwrite(hdr, 4, 2, f);
        double intr[3] = {sfm.GetFocal(), 320.0, 240.0}; std::fwrite(intr, 8, 3, f);
        for (int i = 0; i < Nc; i++) { Pose P = sfm.GetPose(i); std::fwrite(P.t.v, 8, 3, f); std::fwrite(P.r.v, 8, 3, f); Vec3 c = P.getCenter(); std::fwrite(c.v, 8, 3, f); }
        for (int j = 0; j < Np; j++) { Point X = sfm.GetPoint(j); std::fwrite(X.v, 8, 3, f); auto c = sfm.GetColor(j); std::fwrite(c.data(), 1, 3, f); }
        for (int i = 0; i < Nc; i++) for (int j = 0; j < Np; j++) { Observation o; const int has = sfm.GetObservation(i, j, o) ? 1 : 0; std::fwrite(&has, 4, 1, f); double xy[2] = {o.x, o.y}; std::fwrite(xy, 8, 2, f); }
        std::fclose(f);
    };
    dump(out + "/state.bin");
    std::vector<int> indices(Nc); for (int i = 0; i < Nc; i++) indices[i] = 10 * i + 1;
    sfm.WritePoses(out + "/poses.txt", indices);
    sfm.WritePointsOBJ(out + "/points.obj");
    sfm.WriteCameraCentersOBJ(out + "/cameras.obj");
    sfm.WriteCOLMAP(out + "/sparse", 640, 480);
    sfm.WriteCalib(out + "/calib.txt");
    sfm.FilterObservations(10.0);                                                 // prints "removed 1 observations"
    sfm.WriteCOLMAP(out + "/sparse_filtered", 640, 480);
    dump(out + "/state_filtered.bin");
    return 0;
}
