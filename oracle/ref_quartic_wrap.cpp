// ORACLE (test infrastructure only) -- C entry points around the REFERENCE'S OWN quartic solver.
//
// src/spherical_solvers.cpp:14-98 of the reference (SolveQuartic, SolveQuarticReals x2: Ferrari's method "from Theia") uses
// nothing but <complex> and <cmath>; the rest of that file needs Eigen.  oracle/Makefile (target `ref`, build container
// only) cuts exactly those lines out of /root/reference into a temporary file, compiles THIS wrapper around them into
// oracle/_ref/libssfm_ref.so and deletes the temporary: no text of the reference is committed or travels.
// tests/golden/make_reference_fixtures.py drives it to produce tests/golden/ref_quartic.npz.
#include <cmath>
#include <complex>
#include <cstdint>

namespace sphericalsfm {
#include SSFM_REF_QUARTIC_BODY      // = the cut of src/spherical_solvers.cpp:14-98, given by the Makefile
}

extern "C" int32_t ref_solve_quartic(double a, double b, double c, double d, double e, double re_im[8]) {
    std::complex<double> roots[4];
    const int n = sphericalsfm::SolveQuartic(a, b, c, d, e, roots);
    for (int i = 0; i < 4; i++) { re_im[2 * i] = roots[i].real(); re_im[2 * i + 1] = roots[i].imag(); }
    return n;
}
extern "C" int32_t ref_solve_quartic_reals(double a, double b, double c, double d, double e, double roots[4]) {
    return sphericalsfm::SolveQuarticReals(a, b, c, d, e, roots);
}
extern "C" int32_t ref_solve_quartic_reals_tol(double a, double b, double c, double d, double e, double tolerance, double roots[4]) {
    return sphericalsfm::SolveQuarticReals(a, b, c, d, e, tolerance, roots);
}
