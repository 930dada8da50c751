// Sanitizer harness for the host planner (spherical_sfm_amd/csrc/ba_flatten.h) and the track builder (tracks.cpp): random and
// degenerate problems under -fsanitize=address,undefined (and thread).  Built and run by tests/test_sanitizers_cpu.py.
#include <cstdio>
#include <random>
#include <string>
#include <vector>
#include "../../spherical_sfm_amd/csrc/ba_flatten.h"

extern "C" int ssfm_build_tracks(int32_t, const int32_t*, const double*, int32_t, const int32_t*, const int32_t*, const int32_t*, const int32_t*, const int32_t*,
                                 double, double, int32_t, int32_t*, int32_t*, uint8_t*, int64_t*, int32_t*, int32_t*, double*);

static long check(const ssfm::BAFlat& F, int Nc) {
    long sum = 0;
    if (F.nothing_to_do) return 0;
    // every index the kernels will dereference stays in range
    for (size_t j = 0; j < F.obs_cam.size(); j++) { if (F.obs_cam[j] < 0 || F.obs_cam[j] >= Nc || F.obs_pt[j] < 0 || F.obs_pt[j] >= F.nP) { std::printf("bad obs index\n"); std::abort(); } }
    for (size_t e = 0; e < F.pair_j.size(); e++) {
        const int j = F.pair_j[e], j2 = F.pair_j2[e], p = F.pair_p[e];
        if (j < 0) { if (j2 >= 0 || p >= 0) { std::printf("bad padding\n"); std::abort(); } continue; }
        if (j >= F.M || j2 < 0 || j2 >= F.M || p != F.obs_pt[j] || p != F.obs_pt[j2]) { std::printf("bad pair\n"); std::abort(); }
        sum += j + j2;
    }
    for (size_t t = 0; t < F.chunk_cam.size(); t++) {
        if (F.chunk_b0[t] >= F.chunk_b1[t] || F.chunk_b1[t] * 64 > (long)F.pair_j.size()) { std::printf("bad task\n"); std::abort(); }
        for (int b = F.chunk_b0[t]; b < F.chunk_b1[t]; b++) { const int c = F.chunk_cam[t]; if (F.batch_slot[b] < 0 || F.batch_slot[b] >= F.row_ptr[c + 1] - F.row_ptr[c]) { std::printf("bad slot\n"); std::abort(); } }
    }
    for (int c = 0; c < Nc; c++) for (int e = F.row_ptr[c]; e < F.row_ptr[c + 1]; e++) if (F.col_idx[e] < 0 || F.col_idx[e] >= Nc) { std::printf("bad col\n"); std::abort(); }
    // signature groups: every record of k_schur_gram / k_gram_backsub stays inside the arrays, covers exactly the flagged points, and is sorted by K
    {
        const size_t nnzb = F.col_idx.size(); int lastK = 0; long grouped = 0;
        std::vector<char> seen((size_t)F.nP, 0);
        for (size_t t = 0; t * ssfm::GRAM_REC < F.gr_rec.size(); t++) {
            const int* r = &F.gr_rec[t * ssfm::GRAM_REC];
            const int p0 = r[0], cnt = r[1], K = r[2], j0 = r[3];
            if (p0 < 0 || cnt <= 0 || p0 + cnt > F.nP || K < 2 || K > ssfm::GRAM_KMAX || K < lastK || j0 != F.pt_start[p0]) { std::printf("bad group head\n"); std::abort(); }
            lastK = K;
            for (int q = p0; q < p0 + cnt; q++) {
                if (!F.pt_grouped[q] || seen[q] || F.pt_start[q + 1] - F.pt_start[q] != K) { std::printf("bad group point\n"); std::abort(); }
                seen[q] = 1; grouped++;
                for (int k = 0; k < K; k++) if (F.obs_cam[F.pt_start[q] + k] != r[4 + k]) { std::printf("bad group camera\n"); std::abort(); }
            }
            for (int k = 0; k < ssfm::GRAM_KMAX; k++) { if (r[4 + k] < 0 || r[4 + k] >= Nc || r[40 + k] < 0 || (size_t)r[40 + k] >= nnzb) { std::printf("bad group camera / diagonal slot\n"); std::abort(); } }
            for (int a = 1; a < K; a++) for (int b = 0; b < a; b++) {
                const int sl = r[12 + a * (a - 1) / 2 + b] & 0x3fffffff; const bool tr = (r[12 + a * (a - 1) / 2 + b] >> 30) & 1;
                const int row = tr ? r[4 + b] : r[4 + a], col = tr ? r[4 + a] : r[4 + b];
                if ((size_t)sl >= nnzb || sl < F.row_ptr[row] || sl >= F.row_ptr[row + 1] || F.col_idx[sl] != col) { std::printf("bad group slot\n"); std::abort(); }
            }
            sum += p0 + cnt;
        }
        if (grouped != F.gram_points) { std::printf("group count\n"); std::abort(); }
        for (int q = 0; q < F.nP; q++) if (!F.pt_grouped.empty() && (bool)F.pt_grouped[q] != (bool)seen[q]) { std::printf("flag without group\n"); std::abort(); }
        for (size_t e = 0; e < F.pair_j.size(); e++) if (F.pair_p[e] >= 0 && F.pt_grouped[F.pair_p[e]]) { std::printf("grouped point in the pair lists\n"); std::abort(); }
    }
    // round 6, fold tables of the atomics-free Gram emission (one fixed-stride table per camera row): every partial block / vector stretch of every task appears
    // exactly once, in the table of the row that owns its destination, under the right extended slot, in task order
    if (!F.gpart_off.empty()) {
        const int ng = (int)(F.gr_rec.size() / ssfm::GRAM_REC), DC = F.DC, BB = DC * DC;
        if ((int)F.gpart_off.size() != ng + 1 || F.gram_points != (int64_t)F.nP || F.fold_slot_src.size() != (size_t)Nc * ssfm::GRAM_FOLD_STRIDE) { std::printf("fold: sizes\n"); std::abort(); }
        std::vector<int> owner((size_t)F.gpart_off[ng], -1);            // partial offset -> extended slot it must be folded into
        for (int t = 0; t < ng; t++) {
            const int* r = &F.gr_rec[(size_t)t * ssfm::GRAM_REC]; const int K = r[2], base = F.gpart_off[t];
            if (F.gpart_off[t + 1] - base != ssfm::gram_part_len(K, DC)) { std::printf("fold: stretch length\n"); std::abort(); }
            auto row_of = [&](int sl) { int c = 0; while (F.row_ptr[c + 1] <= sl) c++; return c; };
            for (int a = 0; a < K; a++) {
                owner[base + (a * (a + 1) / 2 + a) * BB] = r[40 + a] + row_of(r[40 + a]);
                owner[base + ssfm::gram_part_blocks(K) * BB + a * 5 * DC] = F.row_ptr[r[4 + a] + 1] + r[4 + a];
                for (int b = 0; b < a; b++) { const int sl = r[12 + a * (a - 1) / 2 + b] & 0x3fffffff; owner[base + (a * (a + 1) / 2 + b) * BB] = sl + row_of(sl); }
            }
        }
        size_t listed = 0;
        for (int c = 0; c < Nc; c++) {
            const int* tab = &F.fold_slot_src[(size_t)c * ssfm::GRAM_FOLD_STRIDE];
            const int nq = tab[0], nx = F.row_ptr[c + 1] - F.row_ptr[c] + 1, x0 = F.row_ptr[c] + c;
            if (nq < 0 || nq > ssfm::GRAM_FOLD_SRCS || nx + 1 > ssfm::GRAM_FOLD_PTRS || tab[1] != 0 || tab[1 + nx] != nq) { std::printf("fold: table head\n"); std::abort(); }
            for (int j = 0; j < nx; j++) {
                if (tab[1 + j] > tab[2 + j]) { std::printf("fold: pointers\n"); std::abort(); }
                int last = -1;
                for (int q = tab[1 + j]; q < tab[2 + j]; q++) {
                    const int src = tab[ssfm::GRAM_FOLD_HEAD + q];
                    if (src < 0 || src >= F.gpart_off[ng] || owner[src] != x0 + j || src <= last) { std::printf("fold: source %d of row %d\n", src, c); std::abort(); }
                    last = src; owner[src] = -2; listed++;
                }
            }
        }
        for (int v : owner) if (v >= 0) { std::printf("fold: a partial block is not listed\n"); std::abort(); }
        sum += (long)listed;
    }
    return sum;
}

int main() {
    std::mt19937 rng(42);
    long acc = 0;
    for (int trial = 0; trial < 60; trial++) {
        const int Nc = 1 + rng() % 40, Np = rng() % 300, K = 1 + rng() % 7;
        std::vector<double> cams((size_t)Nc * 6, 0.1), pts((size_t)Np * 3), xy; std::vector<int32_t> oc, op;
        std::vector<uint8_t> rf(Nc, 0), tf(Nc, trial % 3 == 0), pf(Np, 0);
        for (auto& v : pts) v = (rng() % 100) / 10.0 + 0.5;
        for (int j = 0; j < Np; j++) {
            if (rng() % 11 == 0) { pts[3 * j] = pts[3 * j + 1] = pts[3 * j + 2] = 0.0; }          // zero point: leaves the problem
            const int k = rng() % (K + 1);
            for (int q = 0; q < k; q++) {
                int c = (int)(rng() % (Nc + 2)) - 1;                                               // includes -1 and Nc (out of range ids)
                int pid = (rng() % 37 == 0) ? Np + 3 : j;                                          // and out-of-range point ids
                oc.push_back(c); op.push_back(pid); xy.push_back(1.0); xy.push_back(2.0);
                if (rng() % 13 == 0) { oc.push_back(c); op.push_back(pid); xy.push_back(3.0); xy.push_back(4.0); }   // duplicate key
            }
        }
        if (trial % 2) {                                                                           // unsorted input
            for (size_t i = oc.size(); i > 1; i--) { const size_t k = rng() % i; std::swap(oc[i - 1], oc[k]); std::swap(op[i - 1], op[k]); std::swap(xy[2 * i - 2], xy[2 * k]); std::swap(xy[2 * i - 1], xy[2 * k + 1]); }
        }
        double focal = 800.0;
        ssfm_ba_problem P;
        P.num_cameras = Nc; P.num_points = Np; P.num_observations = (int64_t)oc.size(); P.cameras = cams.data(); P.points = pts.data(); P.focal = &focal;
        P.obs_xy = xy.data(); P.obs_cam = oc.data(); P.obs_pt = op.data(); P.rot_fixed = rf.data(); P.trans_fixed = tf.data(); P.pt_fixed = pf.data(); P.focal_fixed = 1;
        for (int nr = 1; nr <= 3; nr++) for (int r = 0; r < nr; r++) { ssfm::BAFlat F; ssfm::ba_flatten(P, nr, r, F); acc += check(F, Nc); }
    }
    // large sorted problems: the threaded sections of the planner (segment pass cut at point boundaries, bit-matrix S structure at <= 1024 cameras,
    // camera-major sweep above) must give the plan of the single-threaded run
    for (int Nc : {300, 1500}) {
        const int Np = 45000; std::vector<double> cams((size_t)Nc * 6, 0.1), pts((size_t)Np * 3), xy; std::vector<int32_t> oc, op;
        std::vector<uint8_t> rf(Nc, 0), tf(Nc, 0), pf(Np, 0);
        for (auto& v : pts) v = (rng() % 100) / 10.0 + 0.5;
        for (int j = 0; j < Np; j++) {
            const int first = (int)(rng() % Nc), len = (j % 997 == 0) ? 80 : 3 + (int)(rng() % 6);   // a few tracks longer than 64 cameras
            std::vector<int> cs; for (int q = 0; q < len; q++) cs.push_back((first + q) % Nc);
            std::sort(cs.begin(), cs.end());
            for (int c : cs) { oc.push_back(c); op.push_back(j); xy.push_back(1.0); xy.push_back(2.0); }
        }
        double focal = 800.0;
        ssfm_ba_problem P;
        P.num_cameras = Nc; P.num_points = Np; P.num_observations = (int64_t)oc.size(); P.cameras = cams.data(); P.points = pts.data(); P.focal = &focal;
        P.obs_xy = xy.data(); P.obs_cam = oc.data(); P.obs_pt = op.data(); P.rot_fixed = rf.data(); P.trans_fixed = tf.data(); P.pt_fixed = pf.data(); P.focal_fixed = 1;
        const char* keep = std::getenv("SSFM_PLAN_THREADS"); const std::string saved = keep ? keep : "";
        setenv("SSFM_PLAN_THREADS", "1", 1); ssfm::BAFlat F1; ssfm::ba_flatten(P, 1, 0, F1);
        setenv("SSFM_PLAN_THREADS", "4", 1); ssfm::BAFlat F4; ssfm::ba_flatten(P, 1, 0, F4);
        if (keep) setenv("SSFM_PLAN_THREADS", saved.c_str(), 1); else unsetenv("SSFM_PLAN_THREADS");
        if (P.num_observations < 200000 || F1.row_ptr != F4.row_ptr || F1.col_idx != F4.col_idx || F1.diag_slot != F4.diag_slot || F1.obs_cam != F4.obs_cam || F1.obs_pt != F4.obs_pt ||
            F1.nP_global != F4.nP_global || F1.M_global != F4.M_global || F1.max_row_blocks != F4.max_row_blocks) { std::printf("threaded plan differs at Nc=%d\n", Nc); std::abort(); }
        acc += check(F4, Nc);
    }
    // signature groups: blocks of consecutive points that share their camera list (runs of 20..90 points, lists of 2..10 cameras, every now and then a loose point
    // inside a run), host-built pair lists, 1..3 ranks, with and without the K >= 4 rule
    setenv("SSFM_GRAM_MODEL", "0", 1);                   // keep every group that qualifies: the planner's cost model (round 4) would leave these small mixed problems to the pair lists
    for (int trial = 0; trial < 6; trial++) {
        const int Nc = 40 + (int)(rng() % 200), Np = 6000; std::vector<double> cams((size_t)Nc * 6, 0.1), pts((size_t)Np * 3), xy; std::vector<int32_t> oc, op;
        std::vector<uint8_t> rf(Nc, 0), tf(Nc, trial % 2), pf(Np, 0);
        for (auto& v : pts) v = (rng() % 100) / 10.0 + 0.5;
        int left = 0, first = 0, len = 3;
        for (int j = 0; j < Np; j++) {
            if (left == 0) { left = 20 + (int)(rng() % 71); first = (int)(rng() % Nc); len = 2 + (int)(rng() % 9); }
            left--;
            int f2 = first, l2 = len; if (rng() % 41 == 0) { f2 = (int)(rng() % Nc); l2 = 3 + (int)(rng() % 4); }
            std::vector<int> cs; for (int q = 0; q < l2; q++) cs.push_back((f2 + q) % Nc);
            std::sort(cs.begin(), cs.end());
            for (int c : cs) { oc.push_back(c); op.push_back(j); xy.push_back(1.0); xy.push_back(2.0); }
        }
        double focal = 800.0;
        ssfm_ba_problem P;
        P.num_cameras = Nc; P.num_points = Np; P.num_observations = (int64_t)oc.size(); P.cameras = cams.data(); P.points = pts.data(); P.focal = &focal;
        P.obs_xy = xy.data(); P.obs_cam = oc.data(); P.obs_pt = op.data(); P.rot_fixed = rf.data(); P.trans_fixed = tf.data(); P.pt_fixed = pf.data(); P.focal_fixed = 1;
        if (trial % 3 == 2) setenv("SSFM_GRAM_KMIN", "2", 1); else unsetenv("SSFM_GRAM_KMIN");
        if (trial == 4) setenv("SSFM_GRAM_PTS", "24", 1); else unsetenv("SSFM_GRAM_PTS");
        long groups = 0;
        for (int nr = 1; nr <= 3; nr++) for (int r = 0; r < nr; r++) { ssfm::BAFlat F; ssfm::ba_flatten(P, nr, r, F); acc += check(F, Nc); groups += (long)F.gr_rec.size(); }
        if (groups == 0) { std::printf("no signature group found\n"); std::abort(); }
    }
    unsetenv("SSFM_GRAM_KMIN"); unsetenv("SSFM_GRAM_PTS");
    // round 6: fully grouped problems (a strided circle: every point of an anchor camera sees the same K cameras) must come with the fold tables of the atomics-free
    // emission (check() verifies them entry by entry); K = 3 / 6 / 8, 6- and 3-dof, one and two ranks
    {
        long folds = 0;
        for (int trial = 0; trial < 4; trial++) {
            const int Nc = trial == 3 ? 96 : 120, K = trial == 0 ? 3 : (trial == 1 ? 6 : 8), per = 90, Np = Nc * per, stride = trial == 2 ? 3 : 1;
            std::vector<double> cams((size_t)Nc * 6, 0.1), pts((size_t)Np * 3), xy; std::vector<int32_t> oc, op;
            std::vector<uint8_t> rf(Nc, 0), tf(Nc, trial == 3), pf(Np, 0);
            for (auto& v : pts) v = (rng() % 100) / 10.0 + 0.5;
            for (int j = 0; j < Np; j++) {
                std::vector<int> cs; for (int q = 0; q < K; q++) cs.push_back((j / per + stride * q) % Nc);
                std::sort(cs.begin(), cs.end());
                for (int c : cs) { oc.push_back(c); op.push_back(j); xy.push_back(1.0); xy.push_back(2.0); }
            }
            double focal = 800.0;
            ssfm_ba_problem P;
            P.num_cameras = Nc; P.num_points = Np; P.num_observations = (int64_t)oc.size(); P.cameras = cams.data(); P.points = pts.data(); P.focal = &focal;
            P.obs_xy = xy.data(); P.obs_cam = oc.data(); P.obs_pt = op.data(); P.rot_fixed = rf.data(); P.trans_fixed = tf.data(); P.pt_fixed = pf.data(); P.focal_fixed = trial & 1;
            for (int nr = 1; nr <= 2; nr++) for (int r = 0; r < nr; r++) { ssfm::BAFlat F; ssfm::ba_flatten(P, nr, r, F); acc += check(F, Nc); folds += F.gpart_off.empty() ? 0 : 1; }
        }
        if (folds < 8) { std::printf("fully grouped problems without fold tables (%ld of 12)\n", folds); std::abort(); }
    }
    // tracks: random match sets incl. merges
    for (int trial = 0; trial < 20; trial++) {
        const int nk = 2 + rng() % 6; std::vector<int32_t> fp(nk + 1, 0); for (int k = 0; k < nk; k++) fp[k + 1] = fp[k] + 5 + rng() % 20;
        std::vector<double> fxy((size_t)fp[nk] * 2, 1.0);
        std::vector<int32_t> i0, i1, mp(1, 0), f0, f1;
        for (int a = 0; a < nk; a++) for (int b = a + 1; b < nk; b++) {
            if (rng() % 3 == 0) continue;
            i0.push_back(a); i1.push_back(b);
            const int na = fp[a + 1] - fp[a], nb = fp[b + 1] - fp[b];
            for (int x = 0; x < na; x++) if (rng() % 2) { f0.push_back(x); f1.push_back(rng() % nb); }
            mp.push_back((int)f0.size());
        }
        const int nm = (int)f0.size();
        std::vector<int32_t> tracks(fp[nk]), ocam(2 * nm + 1), opt(2 * nm + 1); std::vector<uint8_t> alive(nm + 1); std::vector<double> oxy(4 * nm + 2);
        int32_t npts = 0; int64_t nobs = 0;
        ssfm_build_tracks(nk, fp.data(), fxy.data(), (int)i0.size(), i0.data(), i1.data(), mp.data(), f0.data(), f1.data(), 0.0, 0.0, 1, tracks.data(), &npts, alive.data(), &nobs,
                          ocam.data(), opt.data(), oxy.data());
        acc += npts + nobs;
    }
    // signature sort (round 4): camera lists interleaved point by point -- no run in the caller's order, runs of 300 after the planner's re-ordering; 1..3 ranks, both
    // with the cost model off (groups kept) and on
    for (int trial = 0; trial < 4; trial++) {
        const int Nc = 30 + (int)(rng() % 50), Np = 1200, NS = 4; std::vector<double> cams((size_t)Nc * 6, 0.1), pts((size_t)Np * 3), xy; std::vector<int32_t> oc, op;
        std::vector<uint8_t> rf(Nc, 0), tf(Nc, 0), pf(Np, 0);
        for (auto& v : pts) v = (rng() % 100) / 10.0 + 0.5;
        int first[NS], len[NS]; for (int q = 0; q < NS; q++) { first[q] = (int)(rng() % Nc); len[q] = 3 + (int)(rng() % 6); }
        for (int j = 0; j < Np; j++) {
            const int q = j % NS; std::vector<int> cs; for (int k = 0; k < len[q]; k++) cs.push_back((first[q] + k) % Nc);
            std::sort(cs.begin(), cs.end());
            for (int c : cs) { oc.push_back(c); op.push_back(j); xy.push_back(1.0); xy.push_back(2.0); }
        }
        double focal = 800.0;
        ssfm_ba_problem P;
        P.num_cameras = Nc; P.num_points = Np; P.num_observations = (int64_t)oc.size(); P.cameras = cams.data(); P.points = pts.data(); P.focal = &focal;
        P.obs_xy = xy.data(); P.obs_cam = oc.data(); P.obs_pt = op.data(); P.rot_fixed = rf.data(); P.trans_fixed = tf.data(); P.pt_fixed = pf.data(); P.focal_fixed = 1;
        if (trial % 2) unsetenv("SSFM_GRAM_MODEL"); else setenv("SSFM_GRAM_MODEL", "0", 1);
        long grouped = 0;
        for (int nr = 1; nr <= 3; nr++) for (int r = 0; r < nr; r++) {
            ssfm::BAFlat F; ssfm::ba_flatten(P, nr, r, F); acc += check(F, Nc); grouped += F.gram_points;
            if (!F.gram_sorted) { std::printf("signature sort did not run\n"); std::abort(); }
            std::vector<char> seen(Np, 0); for (int q = 0; q < F.nP; q++) { if (F.pt_ids[q] < 0 || F.pt_ids[q] >= Np || seen[F.pt_ids[q]]) { std::printf("sorted ids are not a permutation\n"); std::abort(); } seen[F.pt_ids[q]] = 1; }
        }
        if (trial % 2 == 0 && grouped < 3 * (Np - NS * 32 * 3)) { std::printf("sorted problem not grouped: %ld\n", grouped); std::abort(); }
    }
    unsetenv("SSFM_GRAM_MODEL");
    std::printf("SANITIZE_OK %ld\n", acc);
    return 0;
}
