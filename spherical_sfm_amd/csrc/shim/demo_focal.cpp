// find_best_focal_length_random through the C++ wrapper (shim/tools.cpp) on a closed camera ring whose relative rotations
// are consistent with the guessed focal: the search + refinement must come back to (about) the guess.
// Prints a machine-readable line consumed by tests/test_cpp_shim_gpu.py.
#include <cmath>
#include <cstdio>
#include "tools.h"
#include "../ssfm_math.h"
using namespace sphericalsfm;

int main() {
    const int Nc = 48; const double focal_guess = 1000.0;
    std::vector<std::array<double, 9>> Rgt(Nc);                       // row-major
    for (int i = 0; i < Nc; i++) { double a = 2 * M_PI * i / Nc; if (a > M_PI) a -= 2 * M_PI; const double r[3] = {0, a, 0}; ssfm::so3exp(r, Rgt[i].data()); }
    std::vector<ImageMatch> matches;
    for (int i = 0; i < Nc; i++) for (int d = 1; d <= 3; d++) {
        const int j = i + d; const int jj = j % Nc;
        if (j >= Nc && !(d == 1 || jj < 3)) continue;
        double Rrel[9]; ssfm::mat3_mul_bt(Rgt[jj].data(), Rgt[i].data(), Rrel);          // R_j R_i^T, row-major
        Mat3 cm; for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) cm[r + 3 * c] = Rrel[3 * r + c];
        matches.push_back(ImageMatch(i, jj, Matches(), cm));
    }
    ssfm_ctx* ctx = nullptr;
    if (ssfm_ctx_create(-1, nullptr, &ctx) != SSFM_OK) { std::printf("error: %s\n", ssfm_last_error(nullptr)); return 1; }
    std::vector<Mat3> rotations; double best_focal = 0;
    const bool ok = find_best_focal_length_random(ctx, Nc, matches, false, true, focal_guess, focal_guess / 4, focal_guess * 2, 256, rotations, best_focal, 5, nullptr);
    // closure of the recovered ring: R_{Nc-1} should be one step short of the identity
    double err = 0;
    if (ok) for (int i = 0; i < Nc; i++) { double rm[9]; for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) rm[3 * r + c] = rotations[i][r + 3 * c];
                                           double d[9], w[3]; ssfm::mat3_mul_bt(rm, Rgt[i].data(), d); ssfm::so3ln(d, w); err = std::fmax(err, std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2])); }
    std::printf("FOCAL_RESULT ok=%d focal=%.9f max_rot_err=%.6e n=%zu\n", ok ? 1 : 0, best_focal, err, rotations.size());
    ssfm_ctx_destroy(ctx);
    return ok ? 0 : 1;
}
