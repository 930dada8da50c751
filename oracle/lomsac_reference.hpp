// ORACLE (test infrastructure only) -- the REFERENCE'S OWN LO-MSAC driving the oracle's estimators.
//
// include/RansacLib/{ransac,sampling,utils}.h of the reference depend on the C++ standard library only, so they compile
// here as they stand (the rest of the reference needs Eigen / Ceres and does not).  This adapter gives
// ransac_lib::LocallyOptimizedMSAC<Model, std::vector<Model>, Solver> the interface of oracle::LoMsac (lomsac.hpp), so that
// ransac_oracle.cpp and triangulation_oracle.cpp can be compiled a second time (oracle/Makefile, target `ref`,
// -DSSFM_ORACLE_REAL_RANSACLIB -I/root/reference/include) into oracle/_ref/libssfm_ref.so with the reference's control
// flow, sampler and iteration-count rule in place of the restatement.  Same estimator code, same compiler flags: the two
// builds must then agree bit for bit on statistics, inlier sets and models -- tests/test_reference_pins_cpu.py -- which
// pins rows a13 (EstimateModel / LocalOptimization / LeastSquaresFit / UniformSampling / NumRequiredIterations /
// RandomShuffleAndResize, include/RansacLib/ransac.h:128-420, sampling.h:46-135, utils.h:48-140) and the control flow of
// N1 (SfM::Retriangulate's per-point LocallyOptimizedMSAC, src/sfm.cpp:175-183) to the reference itself.
// Only the build container has /root/reference; nothing of it is copied into the repository.
#pragma once
#include <RansacLib/ransac.h>      // the reference's header, found through -I/root/reference/include
#include "lomsac.hpp"              // MSACOptions / MSACStats (plain structs)

namespace oracle {

template <class Solver, class Model>
struct LoMsacReference {
    const Solver& S; MSACOptions o;
    LoMsacReference(const Solver& s, const MSACOptions& op) : S(s), o(op) {}
    int estimate(Model* best_model, MSACStats* st) const {
        ransac_lib::LORansacOptions ro;
        ro.min_num_iterations_ = o.min_it; ro.max_num_iterations_ = o.max_it; ro.success_probability_ = o.prob;
        ro.squared_inlier_threshold_ = o.sq_thresh; ro.random_seed_ = o.seed;
        ro.num_lo_steps_ = o.num_lo_steps; ro.threshold_multiplier_ = o.thresh_mult; ro.num_lsq_iterations_ = o.num_lsq_it;
        ro.min_sample_multiplicator_ = o.min_sample_mult; ro.non_min_sample_multiplier_ = o.non_min_mult;
        ro.lo_starting_iterations_ = o.lo_start; ro.final_least_squares_ = o.final_lsq;
        ransac_lib::LocallyOptimizedMSAC<Model, std::vector<Model>, Solver> driver;
        ransac_lib::RansacStatistics rs;
        const int n = driver.EstimateModel(ro, S, best_model, &rs);
        *st = MSACStats();
        st->iterations = rs.num_iterations; st->best_num_inliers = rs.best_num_inliers; st->best_score = rs.best_model_score;
        st->lo_count = rs.number_lo_iterations; st->inliers = rs.inlier_indices;
        return n;
    }
};

}  // namespace oracle
