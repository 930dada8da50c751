// Host check of the fixed-point splits behind SSFM_DETERMINISTIC (csrc/det_acc.h): every split is exact (or drops only what lies under the stated quantum), the limb
// sums do not depend on the order of the addends, and the decoded value is the correctly ordered sum to the accuracy a double can show.
#include "../../spherical_sfm_amd/csrc/det_acc.h"
#include <algorithm>
#include <cstdio>
#include <random>
#include <vector>
using namespace ssfm;
static long double limb_value(const long long* L) { long double s = 0; for (int j = LA_NL - 1; j >= 0; j--) s += ldexpl((long double)L[j], LA_E0 + LA_W * j); return s; }
int main() {
    std::mt19937_64 g(7); std::uniform_real_distribution<double> u(-1.0, 1.0); std::uniform_int_distribution<int> ex(-170, 95);
    int bad = 0;
    // 1. long accumulator: exact split over the whole exponent range, bounds of the coefficients
    for (int it = 0; it < 200000; it++) {
        const double v = ldexp(u(g), ex(g));
        int j0; long long c[3];
        if (!lacc_split(v, j0, c)) { bad++; continue; }
        long double s = 0; for (int d = 0; d < 3; d++) if (j0 + d < LA_NL) s += ldexpl((long double)c[d], LA_E0 + LA_W * (j0 + d));
        const bool exact = (ilogb(v) - 52 >= LA_E0) ? (s == (long double)v) : (fabsl(s - (long double)v) < ldexpl(1.0L, LA_E0));
        if (!exact || llabs(c[2]) >= (1LL << 13) || llabs(c[1]) >= (1LL << 40) || llabs(c[0]) >= (1LL << 40)) { if (bad < 5) std::printf("lacc_split(%a): j0 %d c %lld %lld %lld\n", v, j0, c[0], c[1], c[2]); bad++; }
    }
    int j0; long long c[3];
    if (lacc_split(INFINITY, j0, c) || lacc_split(NAN, j0, c) || lacc_split(0x1p100, j0, c) || !lacc_split(0x1p99, j0, c) || !lacc_split(0.0, j0, c) || !lacc_split(0x1p-200, j0, c)) { std::printf("range handling\n"); bad++; }
    // 2. order independence + accuracy: a cost-like sum of 50 000 positive terms over 12 decades, shuffled
    {
        std::vector<double> v(50000); for (double& x : v) x = fabs(ldexp(u(g), (int)(g() % 40) - 20));
        long long ref[LA_STRIDE] = {0}; long double exact = 0;
        for (double x : v) { lacc_split(x, j0, c); for (int d = 0; d < 3; d++) if (j0 + d < LA_NL) ref[j0 + d] += c[d]; exact += x; }
        for (int rep = 0; rep < 5; rep++) {
            std::shuffle(v.begin(), v.end(), g);
            long long L[LA_STRIDE] = {0};
            for (double x : v) { lacc_split(x, j0, c); for (int d = 0; d < 3; d++) if (j0 + d < LA_NL) L[j0 + d] += c[d]; }
            for (int j = 0; j < LA_NL; j++) if (L[j] != ref[j]) { bad++; break; }
        }
        const double dec = lacc_value(ref, 0);
        if (fabsl((long double)dec - limb_value(ref)) > 4e-16L * fabsl(limb_value(ref)) || fabsl(limb_value(ref) - exact) > 1e-15L * exact) { std::printf("decoded %.17g, limbs %.20Lg, exact %.20Lg\n", dec, limb_value(ref), exact); bad++; }
        if (!std::isnan(lacc_value(ref, 1))) bad++;
    }
    // 3. matrix accumulators: split exact to 2^-74, bounds, order independence, cancellation
    {
        std::uniform_int_distribution<int> e2(-60, 39);
        for (int it = 0; it < 200000; it++) {
            const double v = ldexp(u(g), e2(g)); long long hi, lo;
            if (!zsplit(v, hi, lo)) { bad++; continue; }
            const long double s = ldexpl((long double)hi, -ZA_H) + ldexpl((long double)lo, -ZA_L);
            if (fabsl(s - (long double)v) > ldexpl(0.5L, -ZA_L) || llabs(lo) > (1LL << 51)) { if (bad < 5) std::printf("zsplit(%a): %lld %lld\n", v, hi, lo); bad++; }
        }
        long long hi, lo;
        if (zsplit(0x1p40, hi, lo) || zsplit(NAN, hi, lo) || zsplit(-INFINITY, hi, lo) || !zsplit(0x1p39, hi, lo) || !zsplit(-0x1p29, hi, lo)) bad++;
        for (double x : {0.5, -0.5, 1.5, -2.5, 0x1p51, -0x1p51, 123456789.0, -0.49999999999999994}) if (det_rint51(x) != (long long)rint(x)) { std::printf("det_rint51(%a)\n", x); bad++; }
        std::vector<double> v(400); for (double& x : v) x = ldexp(u(g), (int)(g() % 12) - 10);
        long long H = 0, Lo = 0; long double exact = 0; for (double x : v) { zsplit(x, hi, lo); H += hi; Lo += lo; exact += x; }
        for (int rep = 0; rep < 5; rep++) { std::shuffle(v.begin(), v.end(), g); long long h2 = 0, l2 = 0; for (double x : v) { zsplit(x, hi, lo); h2 += hi; l2 += lo; } if (h2 != H || l2 != Lo) bad++; }
        if (fabsl((long double)zdecode(H, Lo) - exact) > 400 * ldexpl(0.5L, -ZA_L) + 2.3e-16L * fabsl(exact)) { std::printf("zdecode %.17g exact %.20Lg\n", zdecode(H, Lo), exact); bad++; }
    }
    std::printf(bad ? "DET_ACC_FAILED %d\n" : "DET_ACC_OK\n", bad);
    return bad ? 1 : 0;
}
