// Drop-in demonstration: the call sequence of examples/run_spherical_sfm_uncalib.cpp:177-211 / run_spherical_sfm.cpp:93-112
// (spherical BA, Retriangulate, BA; then general BA + Normalize + Retriangulate + BA + Normalize) on a small synthetic circle, through the sphericalsfm::SfM mirror.
// Prints a machine-readable summary line consumed by tests/test_cpp_shim_gpu.py.
#include <cmath>
#include <cstdio>
#include <random>
#include "sfm.h"
#include "../ssfm_math.h"
using namespace sphericalsfm;

int main(int argc, char** argv) {
    const int Nc = 60, Np = argc > 1 ? std::atoi(argv[1]) : 1500, K = 6;
    std::mt19937_64 rng(1234);
    std::uniform_real_distribution<double> uxy(-0.3, 0.3), udepth(4.0, 8.0);
    std::normal_distribution<double> n01(0.0, 1.0);
    const double focal = 1000.0;
    SfM sfm(Intrinsics(focal * 1.1, 960, 540));
    std::vector<std::array<double, 9>> Rgt(Nc);
    for (int i = 0; i < Nc; i++) {
        double ang = 2 * M_PI * i / Nc; if (ang > M_PI) ang -= 2 * M_PI;
        double r[3] = {0, ang, 0}; ssfm::so3exp(r, Rgt[i].data());
        Vec3 rn(r[0] + (i ? n01(rng) * 0.00873 : 0), r[1] + (i ? n01(rng) * 0.00873 : 0), r[2] + (i ? n01(rng) * 0.00873 : 0));
        int c = sfm.AddCamera(Pose(Vec3(0, 0, -1), rn));
        sfm.SetRotationFixed(c, i == 0); sfm.SetTranslationFixed(c, true);                    // spherical (tools.cpp:882-883)
    }
    for (int j = 0; j < Np; j++) {
        const int a = (int)((long long)j * Nc / Np);
        const double d = udepth(rng), pc[3] = {uxy(rng) * d, uxy(rng) * d, d + 1.0};           // pc - t, t = (0,0,-1)
        double X[3]; ssfm::mat3_tvec(Rgt[a].data(), pc, X);
        const double s = 1.0 + 0.01 * n01(rng);
        int p = sfm.AddPoint(Point(X[0] * s, X[1] * s, X[2] * s));
        for (int k = 0; k < K; k++) {
            const int c = ((a + k - K / 2) % Nc + Nc) % Nc;
            double q[3]; ssfm::mat3_vec(Rgt[c].data(), X, q); q[2] -= 1.0;
            sfm.AddObservation(c, p, Observation(focal * q[0] / q[2] + 0.5 * n01(rng), focal * q[1] / q[2] + 0.5 * n01(rng)));
        }
    }
    sfm.SetFocalFixed(false);
    // optional dump of the problem before / after the first Optimize(), so a test can replay it through the oracle
    FILE* dump = argc > 2 ? std::fopen(argv[2], "wb") : nullptr;
    auto dump_state = [&]() {
        if (!dump) return;
        for (int i = 0; i < Nc; i++) { Pose q = sfm.GetPose(i); std::fwrite(q.t.v, 8, 3, dump); std::fwrite(q.r.v, 8, 3, dump); }
        for (int j = 0; j < Np; j++) { Point X = sfm.GetPoint(j); std::fwrite(X.v, 8, 3, dump); }
        double f = sfm.GetFocal(); std::fwrite(&f, 8, 1, dump);
    };
    if (dump) {
        int hdr[3] = {Nc, Np, K}; std::fwrite(hdr, 4, 3, dump);
        for (int j = 0; j < Np; j++) for (int c = 0; c < Nc; c++) { Observation o; if (sfm.GetObservation(c, j, o)) { int ids[2] = {c, j}; std::fwrite(ids, 4, 2, dump); double xy[2] = {o.x, o.y}; std::fwrite(xy, 8, 2, dump); } }
    }
    dump_state();
    const bool ok1 = sfm.Optimize();                                                          // spherical BA
    dump_state();
    const double f1 = sfm.GetFocal(), c1 = sfm.LastSummary().final_cost; const int it1 = sfm.LastSummary().iterations;
    auto count_zero = [&]() { int z = 0; for (int j = 0; j < Np; j++) { Point X = sfm.GetPoint(j); if (X.v[0] == 0 && X.v[1] == 0 && X.v[2] == 0) z++; } return z; };
    sfm.Retriangulate();                                                                      // run_spherical_sfm.cpp:93-95
    dump_state();
    if (dump) std::fclose(dump);
    const int zero1 = count_zero();
    const bool ok1b = sfm.Optimize();
    const double c1b = sfm.LastSummary().final_cost;
    for (int i = 1; i < sfm.GetNumCameras(); i++) sfm.SetTranslationFixed(i, false);          // general BA (run_spherical_sfm.cpp:101-112)
    const bool ok2 = sfm.Optimize();
    const int it2 = sfm.LastSummary().iterations; const double c2 = sfm.LastSummary().final_cost, f2 = sfm.GetFocal();
    sfm.Normalize(false);
    sfm.Retriangulate();
    const int zero2 = count_zero();
    const bool ok3 = sfm.Optimize();
    sfm.Normalize(false);
    double mean_radius = 0; for (int i = 0; i < Nc; i++) mean_radius += sfm.GetPose(i).getCenter().norm();
    std::printf("SHIM_RESULT ok1=%d ok1b=%d ok2=%d ok3=%d it1=%d it2=%d focal1=%.9f focal2=%.9f focal3=%.9f cost1=%.9e cost1b=%.9e cost2=%.9e cost3=%.9e dof2=%d "
                "zero1=%d zero2=%d mean_radius=%.12f\n", ok1, ok1b, ok2, ok3, it1, it2, f1, f2, sfm.GetFocal(), c1, c1b, c2, sfm.LastSummary().final_cost,
                sfm.LastSummary().camera_dof, zero1, zero2, mean_radius / Nc);
    return (ok1 && ok1b && ok2 && ok3) ? 0 : 1;
}
