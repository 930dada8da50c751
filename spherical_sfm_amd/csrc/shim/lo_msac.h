// spherical_sfm_amd -- host-side LO-MSAC driver with RansacLib's interface (include/RansacLib/ransac.h:47-128: RansacOptions,
// LORansacOptions, RansacStatistics, LocallyOptimizedMSAC<Model, ModelVector, Solver>::EstimateModel), so that code written against
// `ransac_lib::LocallyOptimizedMSAC<Eigen::Matrix3d, std::vector<Eigen::Matrix3d>, SphericalEstimator>` (estimate_pairwise,
// examples/spherical_sfm_tools.cpp:380-384) compiles against this build.  RansacLib itself is a third-party header the reference vendors;
// this is a restatement of its documented algorithm (Lebeda, Matas, Chum: "Fixing the Locally Optimized RANSAC", BMVC 2012, as RansacLib
// implements it), organised around one state object per run:
//   * samples: std::mt19937(random_seed_) + uniform_int_distribution -- distinct draws when n / (n - k) < e, else a Fisher-Yates shuffle
//     of 0..n-1 cut to k (sampling.h:46-135);
//   * every minimal sample: all models scored with the truncated (MSAC) sum, the best of the sample compared with the best minimal model
//     so far; a new best minimal model (and iteration lo_starting_iterations_ itself) triggers the local optimisation once
//     lo_starting_iterations_ have passed; after every update max_num_iterations follows the inlier ratio (utils.h:110-140);
//   * local optimisation: least squares on <= min_sample_multiplicator_ * k shuffled inliers of the sqrt(2)-relaxed threshold, then
//     num_lo_steps_ rounds of [non-minimal solve on a shuffled subset, least squares, num_lsq_iterations_ fits with a shrinking threshold];
//   * a run that ended before lo_starting_iterations_ optimises once at the end; final_least_squares_ refits on the final inliers.
// The single-pair path: the GPU is reached through the Solver's virtuals (spherical_estimator.h); batches of pairs run the same control
// flow entirely on the device (ssfm_ransac_batch, csrc/lomsac.hip).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <numeric>
#include <random>
#include <vector>

namespace ransac_lib {

class RansacOptions {
public:
    RansacOptions() : min_num_iterations_(100u), max_num_iterations_(10000u), success_probability_(0.9999), squared_inlier_threshold_(1.0), random_seed_(0u) {}
    uint32_t min_num_iterations_;
    uint32_t max_num_iterations_;
    double success_probability_;
    double squared_inlier_threshold_;
    unsigned int random_seed_;
};

class LORansacOptions : public RansacOptions {
public:
    LORansacOptions() : num_lo_steps_(10), threshold_multiplier_(std::sqrt(2.0)), num_lsq_iterations_(4), min_sample_multiplicator_(7), non_min_sample_multiplier_(3),
                        lo_starting_iterations_(50u), final_least_squares_(false) {}
    int num_lo_steps_;
    double threshold_multiplier_;
    int num_lsq_iterations_;
    int min_sample_multiplicator_;
    int non_min_sample_multiplier_;
    uint32_t lo_starting_iterations_;
    bool final_least_squares_;
};

struct RansacStatistics {
    uint32_t num_iterations;
    int best_num_inliers;
    double best_model_score;
    double inlier_ratio;
    std::vector<int> inlier_indices;
    int number_lo_iterations;
};

template <class Model, class ModelVector, class Solver>
class LocallyOptimizedMSAC {
    struct Run {
        const LORansacOptions& opt; const Solver& solver; const int n, k;
        std::mt19937 sampler, lo_rng;
        Run(const LORansacOptions& o, const Solver& s) : opt(o), solver(s), n(s.num_data()), k(s.min_sample_size()) { sampler.seed(o.random_seed_); lo_rng.seed(o.random_seed_); }

        static constexpr double worst() { return std::numeric_limits<double>::max(); }
        double msac(const Model& m) const { double s = 0; for (int i = 0; i < n; i++) s += std::min(solver.EvaluateModelOnPoint(m, i), opt.squared_inlier_threshold_); return s; }
        int within(const Model& m, double thresh, std::vector<int>* out) const {
            out->clear(); for (int i = 0; i < n; i++) if (solver.EvaluateModelOnPoint(m, i) < thresh) out->push_back(i); return (int)out->size(); }
        static void keep_better(double score, const Model& m, double* best_score, Model* best) { if (score < *best_score) { *best_score = score; *best = m; } }
        void shuffle_cut(std::vector<int>* v, int keep) {
            const int m = (int)v->size();
            for (int i = 0; i + 1 < m; i++) { std::uniform_int_distribution<int> d(i, m - 1); std::swap((*v)[i], (*v)[d(lo_rng)]); }
            v->resize(keep);
        }
        void draw(std::vector<int>* s) {
            if ((double)n / (double)(n - k) < M_E) {
                std::uniform_int_distribution<int> d(0, n - 1);
                s->resize(k);
                for (int i = 0; i < k; i++) { bool again = true; while (again) { (*s)[i] = d(sampler); again = std::find(s->begin(), s->begin() + i, (*s)[i]) != s->begin() + i; } }
            } else {
                s->resize(n); std::iota(s->begin(), s->end(), 0);
                if (k != n) { for (int i = 0; i + 1 < n; i++) { std::uniform_int_distribution<int> d(i, n - 1); std::swap((*s)[i], (*s)[d(sampler)]); } s->resize(k); }
            }
        }
        void refit(double thresh, Model* m) {
            std::vector<int> in; const int c = within(*m, thresh, &in);
            if (c < k) return;
            shuffle_cut(&in, std::min(opt.min_sample_multiplicator_ * k, c));
            solver.LeastSquares(in, m);
        }
        void optimise(Model* best, double* best_score) {
            const int knm = solver.non_minimal_sample_size();
            if (knm > n) return;
            const double thr = opt.squared_inlier_threshold_, wide = thr * opt.threshold_multiplier_;
            Model start = *best; refit(wide, &start);
            keep_better(msac(start), start, best_score, best);
            std::vector<int> base; within(start, wide, &base);
            const int size = std::max(knm, std::min(k * opt.non_min_sample_multiplier_, (int)base.size() / 2));
            for (int r = 0; r < opt.num_lo_steps_; r++) {
                std::vector<int> subset = base; shuffle_cut(&subset, size);
                Model m;
                if (!solver.NonMinimalSolver(subset, &m)) continue;
                keep_better(msac(m), m, best_score, best);
                refit(thr, &m);
                double th = wide; const double shrink = (opt.threshold_multiplier_ - 1.0) * thr / static_cast<int>(opt.num_lsq_iterations_ - 1);
                for (int i = 0; i < opt.num_lsq_iterations_; i++) { refit(th, &m); keep_better(msac(m), m, best_score, best); th -= shrink; }
            }
        }
        uint32_t needed(double ratio) const {
            if (ratio <= 0.0) return opt.max_num_iterations_;
            if (ratio >= 1.0) return opt.min_num_iterations_;
            const double miss = 1.0 - std::pow(ratio, (double)k);
            if (miss >= 0.99999999999999) return opt.max_num_iterations_;
            const double it = std::ceil(std::log(1.0 - opt.success_probability_) / std::log(miss) + 0.5);
            return std::max(opt.min_num_iterations_, std::min(static_cast<uint32_t>(it), opt.max_num_iterations_));
        }
    };

public:
    int EstimateModel(const LORansacOptions& options, const Solver& solver, Model* best_model, RansacStatistics* statistics) const {
        RansacStatistics& st = *statistics;
        st.best_num_inliers = 0; st.best_model_score = Run::worst(); st.num_iterations = 0u; st.inlier_ratio = 0.0; st.inlier_indices.clear(); st.number_lo_iterations = 0;
        Run run(options, solver);
        if (run.k > run.n || run.k <= 0) return 0;
        uint32_t limit = std::max(options.max_num_iterations_, options.min_num_iterations_);
        const double thr = options.squared_inlier_threshold_;
        Model best_minimal{}; double best_minimal_score = Run::worst();
        std::vector<int> sample; ModelVector models;
        auto refresh = [&]() {
            st.best_num_inliers = run.within(*best_model, thr, &st.inlier_indices);
            st.inlier_ratio = (double)st.best_num_inliers / (double)run.n;
        };
        for (st.num_iterations = 0u; st.num_iterations < limit; ++st.num_iterations) {
            const bool at_start = st.num_iterations == options.lo_starting_iterations_;
            if (at_start && best_minimal_score < Run::worst()) {
                ++st.number_lo_iterations; run.optimise(best_model, &st.best_model_score);
                refresh(); limit = run.needed(st.inlier_ratio);
            }
            run.draw(&sample);
            const int found = solver.MinimalSolver(sample, &models);
            if (found <= 0) continue;
            double local = Run::worst(); int local_id = 0;
            for (int m = 0; m < found; m++) { const double s = run.msac(models[m]); if (s < local) { local = s; local_id = m; } }
            if (!(local < best_minimal_score || at_start)) continue;
            const bool improved = local < best_minimal_score;
            if (improved) { best_minimal_score = local; best_minimal = models[local_id]; Run::keep_better(best_minimal_score, best_minimal, &st.best_model_score, best_model); }
            const bool past_start = st.num_iterations >= options.lo_starting_iterations_ && best_minimal_score < Run::worst();
            if (!improved && !past_start) continue;
            if (past_start) {
                ++st.number_lo_iterations;
                double s = best_minimal_score; run.optimise(&best_minimal, &s);
                Run::keep_better(s, best_minimal, &st.best_model_score, best_model);
            }
            refresh(); limit = run.needed(st.inlier_ratio);
        }
        if (st.num_iterations <= options.lo_starting_iterations_ && st.best_model_score < Run::worst()) {
            ++st.number_lo_iterations; run.optimise(best_model, &st.best_model_score); refresh();
        }
        if (options.final_least_squares_) {
            Model refined = *best_model; solver.LeastSquares(st.inlier_indices, &refined);
            const double s = run.msac(refined);
            if (s < st.best_model_score) { st.best_model_score = s; *best_model = refined; refresh(); }
        }
        return st.best_num_inliers;
    }
};

}  // namespace ransac_lib
