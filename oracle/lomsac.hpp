// ORACLE (test infrastructure only) -- RansacLib's LO-MSAC, restated once for every estimator of the reference.
//
//   LocallyOptimizedMSAC::EstimateModel / LocalOptimization / LeastSquaresFit / GetInliers / ScoreModel
//                                                                   include/RansacLib/ransac.h:128-420
//   UniformSampling (std::mt19937 + std::uniform_int_distribution)  include/RansacLib/sampling.h:46-135
//   NumRequiredIterations, RandomShuffleAndResize                    include/RansacLib/utils.h:48-140
// The random streams are libstdc++'s, exactly as a build of the reference would draw them.
// PINNED (round 5): lomsac_reference.hpp runs the reference's own include/RansacLib over the same estimators; tests/test_reference_pins_cpu.py requires the two to
// agree bit for bit on statistics, inlier sets, scores and models (committed fixtures + live where oracle/_ref exists).
// Solver concept (include/sphericalsfm/estimator.h:7-23): min_sample_size, non_minimal_sample_size, num_data,
// MinimalSolver(sample, vector<Model>*), NonMinimalSolver(sample, Model*), EvaluateModelOnPoint(model, i), LeastSquares(sample, Model*).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <numeric>
#include <random>
#include <vector>

namespace oracle {

struct MSACOptions {     // RansacOptions + LORansacOptions defaults, ransac.h:47-101
    uint32_t min_it = 100, max_it = 10000; double prob = 0.9999, sq_thresh = 1.0; unsigned seed = 0;
    int num_lo_steps = 10; double thresh_mult = std::sqrt(2.0); int num_lsq_it = 4, min_sample_mult = 7, non_min_mult = 3;
    uint32_t lo_start = 50; bool final_lsq = false;
};
struct MSACStats { uint32_t iterations = 0; int best_num_inliers = 0; double best_score = std::numeric_limits<double>::max(); int lo_count = 0; std::vector<int> inliers; };

inline uint32_t num_required_iterations(double ratio, double pmiss, int ssize, uint32_t mn, uint32_t mx) {     // utils.h:110-140
    if (ratio <= 0.0) return mx;
    if (ratio >= 1.0) return mn;
    const double pn = 1.0 - std::pow(ratio, (double)ssize);
    if (pn >= 0.99999999999999) return mx;
    const double it = std::ceil(std::log(pmiss) / std::log(pn) + 0.5);
    return std::max(mn, std::min((uint32_t)it, mx));
}
inline void shuffle_resize(int target, std::mt19937* rng, std::vector<int>* s) {                                // utils.h:48-73
    const int n = (int)s->size();
    for (int i = 0; i < n - 1; i++) { std::uniform_int_distribution<int> dist(i, n - 1); std::swap((*s)[i], (*s)[dist(*rng)]); }
    s->resize(target);
}

template <class Solver, class Model>
struct LoMsac {
    const Solver& S; MSACOptions o;
    LoMsac(const Solver& s, const MSACOptions& op) : S(s), o(op) {}
    double score(const Model& m) const { double s = 0; const int n = S.num_data(); for (int i = 0; i < n; i++) s += std::min(S.EvaluateModelOnPoint(m, i), o.sq_thresh); return s; }
    int inliers(const Model& m, double th, std::vector<int>* out) const { out->clear(); const int n = S.num_data(); for (int i = 0; i < n; i++) if (S.EvaluateModelOnPoint(m, i) < th) out->push_back(i); return (int)out->size(); }
    static void update(double sc, const Model& m, double* best_sc, Model* best) { if (sc < *best_sc) { *best_sc = sc; *best = m; } }
    void lsq_fit(double thresh, std::mt19937* rng, Model* model) const {                                        // ransac.h:409-420
        std::vector<int> inl; const int n = inliers(*model, thresh, &inl);
        if (n < S.min_sample_size()) return;
        shuffle_resize(std::min(o.min_sample_mult * S.min_sample_size(), n), rng, &inl);
        S.LeastSquares(inl, model);
    }
    void local_optimization(std::mt19937* rng, Model* best_min, double* score_best) const {                     // ransac.h:341-407
        const int kMinNonMin = S.non_minimal_sample_size();
        if (kMinNonMin > S.num_data()) return;
        Model m_init = *best_min;
        lsq_fit(o.sq_thresh * o.thresh_mult, rng, &m_init);
        update(score(m_init), m_init, score_best, best_min);
        std::vector<int> base; inliers(m_init, o.sq_thresh * o.thresh_mult, &base);
        const int nonmin = std::max(kMinNonMin, std::min(S.min_sample_size() * o.non_min_mult, (int)base.size() / 2));
        for (int r = 0; r < o.num_lo_steps; r++) {
            std::vector<int> sample = base; shuffle_resize(nonmin, rng, &sample);
            Model m;
            if (!S.NonMinimalSolver(sample, &m)) continue;
            update(score(m), m, score_best, best_min);
            lsq_fit(o.sq_thresh, rng, &m);
            double th = o.thresh_mult * o.sq_thresh; const double upd = (o.thresh_mult - 1.0) * o.sq_thresh / static_cast<int>(o.num_lsq_it - 1);
            for (int i = 0; i < o.num_lsq_it; i++) { lsq_fit(th, rng, &m); update(score(m), m, score_best, best_min); th -= upd; }
        }
    }
    int estimate(Model* best_model, MSACStats* st) const {                                                       // ransac.h:128-275
        *st = MSACStats();
        const int kMin = S.min_sample_size(), n = S.num_data();
        if (kMin > n || kMin <= 0) return 0;
        std::mt19937 srng; srng.seed(o.seed); std::uniform_int_distribution<int> udist(0, n - 1);
        const bool draw = ((double)n / (double)(n - kMin)) < M_E;                                                 // sampling.h:66-75
        std::mt19937 rng; rng.seed(o.seed);
        uint32_t max_it = std::max(o.max_it, o.min_it);
        Model best_min{}; double best_min_score = std::numeric_limits<double>::max();
        std::vector<int> sample(kMin); std::vector<Model> models;
        uint32_t it = 0;
        for (it = 0; it < max_it; ++it) {
            if (it == o.lo_start && best_min_score < std::numeric_limits<double>::max()) {
                ++st->lo_count; local_optimization(&rng, best_model, &st->best_score);
                st->best_num_inliers = inliers(*best_model, o.sq_thresh, &st->inliers);
                max_it = num_required_iterations((double)st->best_num_inliers / n, 1.0 - o.prob, kMin, o.min_it, o.max_it);
            }
            if (draw) { sample.resize(kMin); for (int i = 0; i < kMin; i++) { bool found = true; while (found) { found = false; sample[i] = udist(srng); for (int j = 0; j < i; j++) if (sample[j] == sample[i]) { found = true; break; } } } }
            else { sample.resize(n); std::iota(sample.begin(), sample.end(), 0); if (kMin != n) { for (int i = 0; i < n - 1; i++) { std::uniform_int_distribution<int> d(i, n - 1); std::swap(sample[i], sample[d(srng)]); } sample.resize(kMin); } }
            const int nm = S.MinimalSolver(sample, &models);
            if (nm <= 0) continue;
            double bl = std::numeric_limits<double>::max(); int bid = 0;
            for (int m = 0; m < nm; m++) { const double sc = score(models[m]); if (sc < bl) { bl = sc; bid = m; } }
            if (bl < best_min_score || it == o.lo_start) {
                const bool best_min_model = bl < best_min_score;
                if (best_min_model) { best_min_score = bl; best_min = models[bid]; update(best_min_score, best_min, &st->best_score, best_model); }
                const bool run_lo = (it >= o.lo_start && best_min_score < std::numeric_limits<double>::max());
                if (!best_min_model && !run_lo) continue;
                if (run_lo) { ++st->lo_count; double sc = best_min_score; local_optimization(&rng, &best_min, &sc); update(sc, best_min, &st->best_score, best_model); }
                st->best_num_inliers = inliers(*best_model, o.sq_thresh, &st->inliers);
                max_it = num_required_iterations((double)st->best_num_inliers / n, 1.0 - o.prob, kMin, o.min_it, o.max_it);
            }
        }
        if (it <= o.lo_start && st->best_score < std::numeric_limits<double>::max()) {
            ++st->lo_count; local_optimization(&rng, best_model, &st->best_score);
            st->best_num_inliers = inliers(*best_model, o.sq_thresh, &st->inliers);
        }
        if (o.final_lsq) {
            Model refined = *best_model;
            S.LeastSquares(st->inliers, &refined);
            const double sc = score(refined);
            if (sc < st->best_score) { st->best_score = sc; *best_model = refined; st->best_num_inliers = inliers(*best_model, o.sq_thresh, &st->inliers); }
        }
        st->iterations = it;
        return st->best_num_inliers;
    }
};

}  // namespace oracle
