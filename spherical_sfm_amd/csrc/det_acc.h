// spherical_sfm_amd -- order-independent accumulation (round 5, SSFM_DETERMINISTIC=1).
//
// The BA assembly adds into the reduced camera system (block-CSR S, J^T r, diag U, the focal border) and into the scalar block with fp64 atomics from thousands of
// waves: the ORDER of the additions differs from run to run, and floating-point addition is not associative -- repeated solves agree to ~1e-12, not bit for bit.
// Here every addend is converted to FIXED POINT first and added with 64-bit INTEGER atomics, which are exact and therefore commute:
//
//   * matrix / vector accumulators (zadd): two limbs per entry, value = hi 2^-22 + lo 2^-74.  |addend| < 2^40; up to 2^11 addends per entry without overflow of
//     the low limb (|lo| <= 2^51 each); the high limb holds |addend| 2^22 < 2^62 per addend, so the SUM of an entry (not only each addend) has to stay below 2^41
//     (a 64-bit integer of units 2^-22) -- orders of magnitude above anything the Jacobi-scaled system holds, and checked by tests/native/det_acc_check.cpp.  The split of an addend is exact down to 2^-74 = 5e-23 ABSOLUTE -- the Jacobi-scaled system has
//     entries of order 1, so a sum carries ~20 bits more than a double accumulator would.  An addend that is not finite or not below 2^40 bumps a poison word and
//     every entry of THAT assembly decodes to NaN (what the floating-point sum would have propagated: the factorisation then fails and the LM loop rejects the step);
//     the word is cleared behind the decode launch (ba_solver.hip), so one poisoned assembly does not condemn the rest of the solve.  Without Jacobi scaling and
//     with a focal length of thousands of pixels an addend can exceed 2^40 where the fp64 path would still succeed: the mode is meant for the default scaling.
//   * scalars (lacc_add): costs and norms span hundreds of binary orders of magnitude over a solve, so they get a LONG accumulator: seven limbs of 40 bits,
//     limb j in units of 2^(-180 + 40 j); an addend touches the <= 3 limbs its 53 bits overlap.  Exact for 2^-180 <= |v| < 2^100, 2^23 addends per limb.
//
// k_det_decode turns the limbs back into the doubles every consumer reads (and clears them for the next assembly): one extra launch per assembly and one per
// hand-over.  Everything downstream (factorisation, substitutions, candidate) was deterministic already.
//
// Round 6 -- no limbs at all when EVERY point sits in a signature group (the circle workloads; DetZone::part below): k_schur_gram stores each task's blocks and
// vectors with plain stores into its own stretch of a partial buffer and k_finalize_gather folds, per row of S, the partial blocks of every slot in task order from
// a fixed-stride table (ba_flatten.h: GRAM_FOLD_*).  The scalar block keeps the long accumulators; k_finalize_gather and k_publish fold them themselves, so the mode
// has no extra launch.  At BASELINE config 2: k_schur_gram 41.0 -> 34.4 us, k_finalize_gather 5.3 -> 11.5, k_publish 4.7 -> 8.7: +3 % per solve against +19 % with
// the limbs (SSFM_GRAM_FOLD=0 keeps them; ragged tracks, which also add through the pair kernels, always do).
#pragma once
#ifdef __HIPCC__
#include <hip/hip_runtime.h>
#define SSFM_HD __host__ __device__ __forceinline__
#else
#define SSFM_HD inline
#endif
#include <cmath>
#include <cstddef>

namespace ssfm {

constexpr int ZA_H = 22, ZA_L = 74;                 // matrix accumulators: hi in units of 2^-22, lo in units of 2^-74
constexpr int LA_W = 40, LA_E0 = -180, LA_NL = 7, LA_STRIDE = 8;      // long accumulator: LA_NL limbs + one poison word

// where the limbs of a double accumulator live: entry p of the zone [base, ...) has its two limbs at limb[2 (p - base)], the poison word at limb[-1]
// round 6: part != nullptr -> k_schur_gram writes task t's blocks and vectors with plain stores at part + part_off[t] (ba_flatten.h: gram_part_len) and
// k_finalize_gather folds them in a fixed order: no atomics at all for these accumulators (limb is then unused)
struct DetZone { const double* base = nullptr; long long* limb = nullptr; double* part = nullptr; const int* part_off = nullptr; };

// ---- the arithmetic, host-compilable (tests/native/det_acc_check.cpp) ------------------------------------------------------------------
// false: v cannot be represented (not finite, or too large) -> the caller bumps the poison word
// (round-to-nearest-even integer of x, |x| <= 2^51, in two instructions: the low 52 bits of x + 1.5 2^52 are that integer + 2^51)
SSFM_HD long long det_rint51(double x) {
    const double t = x + 0x1.8p52; long long b;
    __builtin_memcpy(&b, &t, sizeof(b));
    return b - 0x4338000000000000LL;
}
SSFM_HD bool zsplit(double v, long long& hi, long long& lo) {
    if (fabs(v) < 0x1p29) {                                                          // the usual case: both conversions through the magic-number addition
        const double t = v * 0x1p22 + 0x1.8p52;                                      // (the rounded value as a double comes from the same addition: no integer -> double conversion)
        __builtin_memcpy(&hi, &t, sizeof(hi)); hi -= 0x4338000000000000LL;
        const double r = v - (t - 0x1.8p52) * 0x1p-22;                               // exact: |r| <= 2^-23
        lo = det_rint51(r * 0x1p74);                                                 // |r 2^74| <= 2^51
        return true;
    }
    if (!(fabs(v) < 0x1p40)) return false;                                           // NaN fails the comparison too
    const double h = rint(v * 0x1p22);
    const double r = v - h * 0x1p-22;
    hi = (long long)h; lo = (long long)rint(r * 0x1p74);
    return true;
}
SSFM_HD double zdecode(long long hi, long long lo) { return (double)hi * 0x1p-22 + (double)lo * 0x1p-74; }
// v = sum_d c[d] 2^(LA_E0 + LA_W (j0 + d)), d = 0..2, with |c[2]| < 2^13 and |c[0]|, |c[1]| < 2^40; what lies under 2^LA_E0 is dropped.  false: see zsplit.
SSFM_HD bool lacc_split(double v, int& j0, long long c[3]) {
    c[0] = c[1] = c[2] = 0; j0 = 0;
    if (v == 0.0) return true;
    if (!(fabs(v) < 0x1p100)) return false;
    const int e = ilogb(v);                                                          // 2^e <= |v| < 2^(e+1): bits e-52 .. e
    const int num = e - 52 - LA_E0;
    j0 = num > 0 ? num / LA_W : 0;                                                   // lowest limb with a bit of v (values reaching under 2^LA_E0 start at limb 0)
    double t = v;
    for (int d = 2; d >= 0; d--) {
        const int j = j0 + d;
        if (j >= LA_NL) continue;                                                    // cannot hold a bit of v: |v| < 2^100 = the top of limb LA_NL - 1
        const double q = trunc(ldexp(t, -(LA_E0 + LA_W * j)));                       // exact: a power-of-two scaling, then the integer part
        c[d] = (long long)q; t -= ldexp(q, LA_E0 + LA_W * j);
    }
    return true;
}
// limbs[0 .. LA_NL): already summed over whatever replicas there are; poison != 0 -> NaN
SSFM_HD double lacc_value(const long long* limbs, long long poison) {
    if (poison != 0) return NAN;
    double s = 0.0;
    for (int j = LA_NL - 1; j >= 0; j--) s += ldexp((double)limbs[j], LA_E0 + LA_W * j);
    return s;
}

#ifdef __HIPCC__
// ---- the atomics ---------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void det_atomic_add(long long* p, long long v) { atomicAdd(reinterpret_cast<unsigned long long*>(p), static_cast<unsigned long long>(v)); }
// the drop-in for unsafeAtomicAdd(p, v) on an accumulator of the zone
__device__ __forceinline__ void zadd(const DetZone& dz, double* p, double v) {
    if (!dz.limb) { unsafeAtomicAdd(p, v); return; }
    long long* q = dz.limb + 2 * (p - dz.base); long long hi, lo;
    if (!zsplit(v, hi, lo)) { det_atomic_add(dz.limb - 1, 1); return; }
    det_atomic_add(q, hi); det_atomic_add(q + 1, lo);
}
// acc: LA_STRIDE words (LA_NL limbs, then the poison word)
__device__ __forceinline__ void lacc_add(long long* acc, double v) {
    int j0; long long c[3];
    if (!lacc_split(v, j0, c)) { det_atomic_add(acc + LA_NL, 1); return; }
#pragma unroll
    for (int d = 0; d < 3; d++) if (c[d] != 0 && j0 + d < LA_NL) det_atomic_add(acc + j0 + d, c[d]);
}
#endif

}  // namespace ssfm
