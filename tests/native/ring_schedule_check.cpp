// CPU check of the cyclic-reduction schedule of long camera rings (spherical_sfm_amd/csrc/ring_schedule.h) and of the algebra the kernels of band_ring.h follow:
// random SPD block systems whose block graph is a set of cycles (M[s][s] = D_s, M[s][s-1] = E_s, indices mod m) are solved by replaying the records exactly as
// k_ring_cr_elim / k_ring_cr_back read them -- pending updates gathered, couplings from "Z" or as products of stored F blocks, tall Cholesky, back substitution in
// reverse step order -- with plain dense loops, and compared with a dense Cholesky solve of the whole system.  Built and run by tests/test_sanitizers_cpu.py
// (-fsanitize=address,undefined).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "../../spherical_sfm_amd/csrc/ring_schedule.h"
using namespace ssfm;

typedef std::vector<double> Mat;      // row-major Q x Q unless stated

static void chol_lower(Mat& A, int n) {      // in place, lower; aborts on a non-positive pivot
    for (int j = 0; j < n; j++) {
        double d = A[j * n + j]; for (int k = 0; k < j; k++) d -= A[j * n + k] * A[j * n + k];
        if (!(d > 0)) { std::printf("ring_schedule_check: pivot %d not positive (%g)\n", j, d); std::abort(); }
        d = std::sqrt(d); A[j * n + j] = d;
        for (int i = j + 1; i < n; i++) { double s = A[i * n + j]; for (int k = 0; k < j; k++) s -= A[i * n + k] * A[j * n + k]; A[i * n + j] = s / d; }
        for (int i = 0; i < j; i++) A[i * n + j] = 0.0;
    }
}

static double run_case(const std::vector<int>& ms, int Q, unsigned seed) {
    const int NR = 2, DC = 1;                                   // the "band rows" of this test are single scalars: Z[c][p0 + i] with p0 = Q * s
    std::mt19937_64 g(seed); std::normal_distribution<double> N01(0.0, 1.0);
    int nsep = 0; std::vector<std::pair<int, int>> ring_seps;
    for (int m : ms) { ring_seps.push_back({nsep, m}); nsep += m; }
    const int n = nsep * Q;
    std::vector<int> sep_lo(nsep), sep_copy(nsep, -1);
    for (int s = 0; s < nsep; s++) sep_lo[s] = s * Q;
    // E_s (rows s, cols s-1 mod m) lives in Z: Z[c][sep_lo[s] + i] = E_s(i, c); D_s lower in Dd
    Mat Z((size_t)Q * n, 0.0), Dd((size_t)nsep * Q * Q, 0.0), tt((size_t)nsep * NR * Q);
    Mat M((size_t)n * n, 0.0);
    for (const auto& rs : ring_seps) {
        const int s0 = rs.first, m = rs.second;
        for (int k = 0; k < m; k++) {
            const int s = s0 + k, sp = s0 + (k + m - 1) % m;
            for (int i = 0; i < Q; i++) for (int c = 0; c < Q; c++) {
                const double e = 0.3 * N01(g);
                Z[(size_t)c * n + sep_lo[s] + i] = e;
                M[(size_t)(s * Q + i) * n + sp * Q + c] += e; M[(size_t)(sp * Q + c) * n + s * Q + i] += e;      // (m == 2: both couplings of the pair add up)
            }
        }
    }
    for (int s = 0; s < nsep; s++) {                           // diagonally dominant diagonal blocks
        Mat R((size_t)Q * Q); for (double& x : R) x = 0.2 * N01(g);
        for (int i = 0; i < Q; i++) for (int c = 0; c <= i; c++) {
            double v = (i == c) ? 4.0 * Q : 0.0; for (int k = 0; k < Q; k++) v += R[i * Q + k] * R[c * Q + k];
            Dd[((size_t)s * Q + i) * Q + c] = v;
            M[(size_t)(s * Q + i) * n + s * Q + c] = v; M[(size_t)(s * Q + c) * n + s * Q + i] = v;
        }
    }
    for (double& x : tt) x = N01(g);
    // ---- the schedule
    std::vector<int> rec, step_ptr, tail_ptr;
    ring_schedule(ring_seps, sep_lo, sep_copy, rec, step_ptr, tail_ptr);
    const int nsteps = (int)step_ptr.size() - 1;
    if ((int)rec.size() / RING_REC != nsep || tail_ptr.back() != nsep || tail_ptr.front() != step_ptr.back()) { std::printf("ring_schedule_check: %zu records for %d separators\n", rec.size() / RING_REC, nsep); std::abort(); }
    const size_t QQ = (size_t)Q * Q;
    const double poison = std::nan("");                           // a buffer read before its first (storing) write shows up as NaN in the answer
    Mat crL((size_t)nsep * QQ, 0.0), crF((size_t)nsep * 2 * QQ, 0.0), crW((size_t)nsep * NR * Q, 0.0), crP((size_t)nsep * 2 * QQ, poison), crT((size_t)nsep * 2 * NR * Q, poison),
        crE((size_t)nsep * QQ, poison), Y((size_t)NR * n, 0.0);
    std::vector<char> done(nsep, 0);
    auto eliminate = [&](int q) {
        const int* r = rec.data() + (size_t)q * RING_REC;
        const int v = r[0], nn = r[1];
        if (done[v]) { std::printf("ring_schedule_check: separator %d eliminated twice\n", v); std::abort(); }
        Mat A(QQ, 0.0), B[2] = {Mat(QQ, 0.0), Mat(QQ, 0.0)}; std::vector<double> t((size_t)NR * Q);
        for (int i = 0; i < Q; i++) for (int c = 0; c <= i; c++) { double val = Dd[((size_t)v * Q + i) * Q + c]; if (r[2]) val -= crP[((size_t)v * 2) * QQ + i * Q + c]; if (r[3]) val -= crP[((size_t)v * 2 + 1) * QQ + i * Q + c]; A[i * Q + c] = val; }
        for (int e = 0; e < NR * Q; e++) { double val = tt[(size_t)v * NR * Q + e]; if (r[2]) val -= crT[((size_t)v * 2) * NR * Q + e]; if (r[3]) val -= crT[((size_t)v * 2 + 1) * NR * Q + e]; t[e] = val; }
        for (int j = 0; j < nn; j++) {
            const int* qq = r + 8 + 16 * j;
            if (done[qq[0]]) { std::printf("ring_schedule_check: neighbour %d of %d already eliminated\n", qq[0], v); std::abort(); }
            for (int tix = 0; tix < qq[2]; tix++) {
                const int* z = qq + 4 + 5 * tix;
                for (int i = 0; i < Q; i++) for (int c = 0; c < Q; c++) {
                    if (z[0] == 0) B[j][i * Q + c] += z[2] ? Z[(size_t)i * n + (size_t)z[1] * DC + c] : Z[(size_t)c * n + (size_t)z[1] * DC + i];
                    else B[j][i * Q + c] += z[2] ? crE[(size_t)z[1] * QQ + c * Q + i] : crE[(size_t)z[1] * QQ + i * Q + c];
                }
            }
        }
        chol_lower(A, Q);
        for (int j = 0; j < nn; j++)                          // F_j = B_j L^-T, row by row: L f = b
            for (int i = 0; i < Q; i++) for (int c = 0; c < Q; c++) { double s = B[j][i * Q + c]; for (int k = 0; k < c; k++) s -= A[c * Q + k] * crF[((size_t)v * 2 + j) * QQ + i * Q + k]; crF[((size_t)v * 2 + j) * QQ + i * Q + c] = s / A[c * Q + c]; }
        for (int rr = 0; rr < NR; rr++) for (int c = 0; c < Q; c++) { double s = t[rr * Q + c]; for (int k = 0; k < c; k++) s -= A[c * Q + k] * crW[(size_t)v * NR * Q + rr * Q + k]; crW[(size_t)v * NR * Q + rr * Q + c] = s / A[c * Q + c]; }
        for (size_t e = 0; e < QQ; e++) crL[(size_t)v * QQ + e] = A[e];
        for (int j = 0; j < nn; j++) {                        // what the neighbours will need
            const int* qq = r + 8 + 16 * j;
            const double* F = crF.data() + ((size_t)v * 2 + j) * QQ;
            double* P = crP.data() + ((size_t)qq[0] * 2 + qq[3]) * QQ; double* tp = crT.data() + ((size_t)qq[0] * 2 + qq[3]) * NR * Q;
            for (int i = 0; i < Q; i++) for (int c = 0; c <= i; c++) { double s = 0; for (int k = 0; k < Q; k++) s += F[i * Q + k] * F[c * Q + k]; P[i * Q + c] = qq[14] ? P[i * Q + c] + s : s; }
            for (int rr = 0; rr < NR; rr++) for (int i = 0; i < Q; i++) { double s = 0; for (int k = 0; k < Q; k++) s += F[i * Q + k] * crW[(size_t)v * NR * Q + rr * Q + k]; tp[rr * Q + i] = qq[14] ? tp[rr * Q + i] + s : s; }
        }
        if (r[6]) for (int i = 0; i < Q; i++) for (int c = 0; c < Q; c++) { double s = 0; for (int k = 0; k < Q; k++) s += crF[((size_t)v * 2 + 1) * QQ + i * Q + k] * crF[((size_t)v * 2) * QQ + c * Q + k]; crE[(size_t)v * QQ + i * Q + c] = -s; }
    };
    auto back = [&](int q) {
        const int* r = rec.data() + (size_t)q * RING_REC;
        const int v = r[0], nn = r[1], p0 = r[5];
        for (int rr = 0; rr < NR; rr++) {
            std::vector<double> vv(Q);
            for (int k = 0; k < Q; k++) {
                double s = crW[(size_t)v * NR * Q + rr * Q + k];
                for (int j = 0; j < nn; j++) { const int pj = r[8 + 16 * j + 1]; for (int i = 0; i < Q; i++) s -= crF[((size_t)v * 2 + j) * QQ + i * Q + k] * Y[(size_t)rr * n + pj + i]; }
                vv[k] = s;
            }
            for (int k = Q - 1; k >= 0; k--) { double s = vv[k]; for (int i = k + 1; i < Q; i++) s -= crL[(size_t)v * QQ + i * Q + k] * Y[(size_t)rr * n + p0 + i]; Y[(size_t)rr * n + p0 + k] = s / crL[(size_t)v * QQ + k * Q + k]; }
        }
    };
    // parallel steps: the eliminations of one step see the buffers as they were BEFORE the step (they are independent launches of workgroups): check that by
    // running every step twice in opposite orders on copies?  The cheap way: no record of a step may read what another record of the same step writes.
    for (int st = 0; st < nsteps; st++) {
        std::vector<int> writesP, readsP, writesE, readsE;
        for (int q = step_ptr[st]; q < step_ptr[st + 1]; q++) {
            const int* r = rec.data() + (size_t)q * RING_REC;
            if (r[2]) readsP.push_back(2 * r[0]); if (r[3]) readsP.push_back(2 * r[0] + 1);
            for (int j = 0; j < r[1]; j++) { const int* qq = r + 8 + 16 * j; writesP.push_back(2 * qq[0] + qq[3]); for (int t = 0; t < qq[2]; t++) if (qq[4 + 5 * t] == 1) readsE.push_back(qq[4 + 5 * t + 1]); }
            if (r[6]) writesE.push_back(r[0]);
        }
        std::sort(writesP.begin(), writesP.end());
        if (std::adjacent_find(writesP.begin(), writesP.end()) != writesP.end()) { std::printf("ring_schedule_check: two writers of one update buffer in step %d\n", st); std::abort(); }
        for (int x : readsP) if (std::binary_search(writesP.begin(), writesP.end(), x)) { std::printf("ring_schedule_check: step %d reads an update buffer it writes\n", st); std::abort(); }
        for (int x : readsE) if (std::find(writesE.begin(), writesE.end(), x) != writesE.end()) { std::printf("ring_schedule_check: step %d reads a fill block it writes\n", st); std::abort(); }
        for (int q = step_ptr[st]; q < step_ptr[st + 1]; q++) eliminate(q);
        for (int q = step_ptr[st]; q < step_ptr[st + 1]; q++) done[rec[(size_t)q * RING_REC]] = 1;      // eliminations of one step are simultaneous
    }
    for (size_t g = 0; g + 1 < tail_ptr.size(); g++) {            // the tails: one after the other per ring, then their back substitutions
        for (int q = tail_ptr[g]; q < tail_ptr[g + 1]; q++) { eliminate(q); done[rec[(size_t)q * RING_REC]] = 1; }
        for (int q = tail_ptr[g + 1] - 1; q >= tail_ptr[g]; q--) back(q);
    }
    for (int st = nsteps - 1; st >= 0; st--) for (int q = step_ptr[st]; q < step_ptr[st + 1]; q++) back(q);
    // ---- dense reference
    Mat Lm = M; chol_lower(Lm, n);
    double worst = 0.0;
    for (int rr = 0; rr < NR; rr++) {
        std::vector<double> y(n), x(n);
        for (int s = 0; s < nsep; s++) for (int i = 0; i < Q; i++) y[s * Q + i] = tt[(size_t)s * NR * Q + rr * Q + i];
        for (int i = 0; i < n; i++) { double s = y[i]; for (int k = 0; k < i; k++) s -= Lm[(size_t)i * n + k] * y[k]; y[i] = s / Lm[(size_t)i * n + i]; }
        for (int i = n - 1; i >= 0; i--) { double s = y[i]; for (int k = i + 1; k < n; k++) s -= Lm[(size_t)k * n + i] * x[k]; x[i] = s / Lm[(size_t)i * n + i]; }
        double nx = 0; for (int i = 0; i < n; i++) nx = std::fmax(nx, std::fabs(x[i]));
        for (int i = 0; i < n; i++) worst = std::fmax(worst, std::fabs(x[i] - Y[(size_t)rr * n + i]) / nx);
    }
    return worst;
}

int main() {
    std::setvbuf(stdout, nullptr, _IONBF, 0);
    double worst = 0.0; int cases = 0;
    const std::vector<std::vector<int>> shapes = {{2}, {3}, {4}, {5}, {6}, {7}, {8}, {9}, {13}, {16}, {17}, {31}, {40}, {2, 3}, {5, 8, 2}, {20, 20}, {64}};
    for (const auto& ms : shapes)
        for (int Q : {1, 3, 6}) { const double e = run_case(ms, Q, 1000u + (unsigned)cases); worst = std::fmax(worst, e); cases++;
                                  if (e > 1e-10) { std::printf("ring_schedule_check: m0 = %d Q = %d: error %.3e\n", ms[0], Q, e); return 1; } }
    std::printf("RING_SCHEDULE_OK %d cases, worst relative error %.2e\n", cases, worst);
    return 0;
}
