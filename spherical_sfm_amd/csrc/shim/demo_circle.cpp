// Drop-in demonstration: the call sequence of examples/run_spherical_sfm_uncalib.cpp:177-211 / run_spherical_sfm.cpp:93-112
// (spherical BA, Retriangulate, BA; then general BA + Normalize + Retriangulate + BA + Normalize) on a synthetic circle, through the sphericalsfm::SfM mirror.
// Prints a machine-readable summary line consumed by tests/test_cpp_shim_gpu.py and tests/test_pipeline_gpu.py.
//   demo_circle [Np = 1500] [dump file] [Nc = 60] [K = 6] [stride = 1] [focal free = 1] [outlier fraction = 0]
// The dump holds the problem and the state after EVERY stage (tests replay each stage through the oracle from the state before it):
//   int32 Nc, Np, M, stages | M x {int32 camera, int32 point, double x, double y} | state 0 | stages x { int32 kind, int32 iterations, int32 ok, int32 pad,
//   double final_cost, state }   with state = Nc x [t; r], Np x X, focal   and kind 0 = Optimize, 1 = Retriangulate, 2 = Normalize, 3 = translations unfixed.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <random>
#include "sfm.h"
#include "../ssfm_math.h"
using namespace sphericalsfm;

int main(int argc, char** argv) {
    const int Np = argc > 1 ? std::atoi(argv[1]) : 1500;
    const int Nc = argc > 3 ? std::atoi(argv[3]) : 60, K = argc > 4 ? std::atoi(argv[4]) : 6, stride = argc > 5 ? std::atoi(argv[5]) : 1;
    const bool focal_free = argc > 6 ? std::atoi(argv[6]) != 0 : true;
    const double outlier_frac = argc > 7 ? std::atof(argv[7]) : 0.0;
    std::mt19937_64 rng(1234);
    const double xy_range = (K - 1) * stride * 360.0 / Nc <= 24.5 ? 0.45 : 0.30;             // as spherical_sfm_amd/synth.py: wide spans need a narrower anchor window
    std::uniform_real_distribution<double> uxy(-std::min(xy_range, 0.3), std::min(xy_range, 0.3)), udepth(4.0, 8.0), u01(0.0, 1.0), upx(-40.0, 40.0);
    std::normal_distribution<double> n01(0.0, 1.0);
    const double focal = 1000.0;
    SfM sfm(Intrinsics(focal_free ? focal * 1.1 : focal, 960, 540));
    std::vector<std::array<double, 9>> Rgt(Nc);
    for (int i = 0; i < Nc; i++) {
        double ang = 2 * M_PI * i / Nc; if (ang > M_PI) ang -= 2 * M_PI;
        double r[3] = {0, ang, 0}; ssfm::so3exp(r, Rgt[i].data());
        Vec3 rn(r[0] + (i ? n01(rng) * 0.00873 : 0), r[1] + (i ? n01(rng) * 0.00873 : 0), r[2] + (i ? n01(rng) * 0.00873 : 0));
        int c = sfm.AddCamera(Pose(Vec3(0, 0, -1), rn));
        sfm.SetRotationFixed(c, i == 0); sfm.SetTranslationFixed(c, true);                    // spherical (tools.cpp:882-883)
    }
    int64_t M = 0;
    for (int j = 0; j < Np; j++) {
        const int a = (int)((long long)j * Nc / Np);
        const double d = udepth(rng), pc[3] = {uxy(rng) * d, uxy(rng) * d, d + 1.0};           // pc - t, t = (0,0,-1)
        double X[3]; ssfm::mat3_tvec(Rgt[a].data(), pc, X);
        const double s = 1.0 + 0.01 * n01(rng);
        int p = sfm.AddPoint(Point(X[0] * s, X[1] * s, X[2] * s));
        for (int k = 0; k < K; k++) {
            const int c = ((a + stride * (k - K / 2)) % Nc + Nc) % Nc;
            double q[3]; ssfm::mat3_vec(Rgt[c].data(), X, q); q[2] -= 1.0;
            double x = focal * q[0] / q[2] + 0.5 * n01(rng), y = focal * q[1] / q[2] + 0.5 * n01(rng);
            if (outlier_frac > 0 && u01(rng) < outlier_frac) { x += upx(rng); y += upx(rng); }  // a wrong match: Retriangulate has something to reject
            sfm.AddObservation(c, p, Observation(x, y)); M++;
        }
    }
    sfm.SetFocalFixed(!focal_free);
    FILE* dump = argc > 2 && argv[2][0] && std::string(argv[2]) != "-" ? std::fopen(argv[2], "wb") : nullptr;
    auto dump_state = [&]() {
        if (!dump) return;
        for (int i = 0; i < Nc; i++) { Pose q = sfm.GetPose(i); std::fwrite(q.t.v, 8, 3, dump); std::fwrite(q.r.v, 8, 3, dump); }
        for (int j = 0; j < Np; j++) { Point X = sfm.GetPoint(j); std::fwrite(X.v, 8, 3, dump); }
        double f = sfm.GetFocal(); std::fwrite(&f, 8, 1, dump);
    };
    double stage_ms[12]; int nstage = 0;
    auto stage = [&](int kind, int iterations, bool ok, double cost, double ms) {
        stage_ms[nstage++] = ms;
        if (!dump) return;
        int hdr[4] = {kind, iterations, ok ? 1 : 0, 0}; std::fwrite(hdr, 4, 4, dump); std::fwrite(&cost, 8, 1, dump);
        dump_state();
    };
    if (dump) {
        int hdr[4] = {Nc, Np, (int)M, 9}; std::fwrite(hdr, 4, 4, dump);
        for (int j = 0; j < Np; j++) for (int k = 0; k < K; k++) {
            const int a = (int)((long long)j * Nc / Np), c = ((a + stride * (k - K / 2)) % Nc + Nc) % Nc;
            Observation o; if (sfm.GetObservation(c, j, o)) { int ids[2] = {c, j}; std::fwrite(ids, 4, 2, dump); double xy[2] = {o.x, o.y}; std::fwrite(xy, 8, 2, dump); }
        }
    }
    dump_state();
    sfm.GetContext();                                                                         // library / HIP start-up is not a stage of the pipeline
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    auto optimize = [&]() { const double t0 = now(); const bool ok = sfm.Optimize(); stage(0, sfm.LastSummary().iterations, ok, sfm.LastSummary().final_cost, now() - t0); return ok; };
    auto retriangulate = [&]() { const double t0 = now(); sfm.Retriangulate(); stage(1, 0, true, 0.0, now() - t0); };
    auto normalize = [&]() { const double t0 = now(); sfm.Normalize(false); stage(2, 0, true, 0.0, now() - t0); };
    auto count_zero = [&]() { int z = 0; for (int j = 0; j < Np; j++) { Point X = sfm.GetPoint(j); if (X.v[0] == 0 && X.v[1] == 0 && X.v[2] == 0) z++; } return z; };
    const bool ok1 = optimize();                                                              // spherical BA
    const double f1 = sfm.GetFocal(), c1 = sfm.LastSummary().final_cost; const int it1 = sfm.LastSummary().iterations;
    retriangulate();                                                                          // run_spherical_sfm.cpp:93-95
    const int zero1 = count_zero();
    const bool ok1b = optimize();
    const double c1b = sfm.LastSummary().final_cost;
    for (int i = 1; i < sfm.GetNumCameras(); i++) sfm.SetTranslationFixed(i, false);          // general BA (run_spherical_sfm.cpp:101-112)
    stage(3, 0, true, 0.0, 0.0);
    const bool ok2 = optimize();
    const int it2 = sfm.LastSummary().iterations; const double c2 = sfm.LastSummary().final_cost, f2 = sfm.GetFocal();
    normalize();
    retriangulate();
    const int zero2 = count_zero();
    const bool ok3 = optimize();
    normalize();
    if (dump) std::fclose(dump);
    double mean_radius = 0; for (int i = 0; i < Nc; i++) mean_radius += sfm.GetPose(i).getCenter().norm();
    std::printf("SHIM_RESULT ok1=%d ok1b=%d ok2=%d ok3=%d it1=%d it2=%d focal1=%.9f focal2=%.9f focal3=%.9f cost1=%.9e cost1b=%.9e cost2=%.9e cost3=%.9e dof2=%d "
                "zero1=%d zero2=%d mean_radius=%.12f\n", ok1, ok1b, ok2, ok3, it1, it2, f1, f2, sfm.GetFocal(), c1, c1b, c2, sfm.LastSummary().final_cost,
                sfm.LastSummary().camera_dof, zero1, zero2, mean_radius / Nc);
    std::printf("STAGE_MS");
    for (int s = 0; s < nstage; s++) std::printf(" %.3f", stage_ms[s]);
    std::printf("\n");
    return (ok1 && ok1b && ok2 && ok3) ? 0 : 1;
}
