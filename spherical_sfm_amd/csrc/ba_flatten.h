// spherical_sfm_amd -- host-side flattening of a bundle-adjustment problem.
//
// Reproduces what the build loop of SfM::Optimize (reference src/sfm.cpp:240-263) decides:
//   * a point enters iff it exists, |X| != 0 and it has >= 3 observations over existing cameras;
//   * then ALL its observations enter, point-major, cameras ascending (std::map iteration order);
//   * a repeated (camera, point) key keeps the last value (map assignment, src/sfm.cpp:140);
//   * constant blocks (src/sfm.cpp:222-225) leave the program; cameras without observations are not in it.
// The reference does this with O(Np*Nc) nested std::map lookups per Optimize() call; here it is one sort
// (skipped when the input is already point-major) plus linear passes.
//
// It also builds what the device kernels need once per problem: camera-major observation lists, the block
// structure of the reduced camera system S (cameras sharing a point), and the point range owned by this
// rank when the problem is sharded over GPUs (contiguous ranges of used points balanced by observations;
// every rank keeps all cameras -- SURVEY.md 8e).
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <numeric>
#include <vector>
#include "../../include/ssfm.h"

namespace ssfm {

struct BAFlat {
    int Nc = 0;
    int nP = 0;               // used points owned by this rank
    int nP_global = 0;
    int64_t M = 0, M_global = 0;
    int DC = 6;               // 3: every translation fixed (spherical BA) -> only rotations vary
    bool focal_free = false;
    std::vector<int> pt_ids;            // [nP] original point id
    std::vector<int> pt_start;          // [nP+1]
    std::vector<int> obs_cam;           // [M]
    std::vector<int> obs_pt;            // [M] compact local point
    std::vector<int64_t> obs_orig;      // [M] index into the caller's arrays
    std::vector<double> obs_xy;         // [2M]
    std::vector<int> cam_start, cam_obs;   // camera-major lists over local observations
    std::vector<int> cam_obs_pt, cs_task_cam, cs_task_q0, cs_task_q1;   // point of each camera-major entry; k_cam_sums2 wave tasks (<= 256 entries)
    std::vector<int> row_ptr, col_idx, diag_slot;   // block-CSR structure of S (global), sorted columns
    std::vector<double> mask_cam;       // [Nc*6] 1 = free parameter that is in the problem
    std::vector<double> mask_pt;        // [nP*3]
    std::vector<double> pts0;           // [nP*3]
    int max_row_blocks = 0;
    bool nothing_to_do = false;
    // elimination order of the camera blocks for the banded Cholesky preconditioner
    std::vector<int> cam_pos;           // [Nc] camera -> position
    int band = 0;                       // block half-bandwidth in that order
    std::vector<int> band_pairs;        // (i_rel | k_rel << 16), i_rel-major, 1 <= k_rel <= i_rel <= band
    std::vector<int> comp_ptr;          // connected components of the camera graph = contiguous position ranges
    // The stored structure (row_ptr/col_idx) is the LOWER triangle in elimination order: row c holds block (c, c2) iff
    // cam_pos[c2] <= cam_pos[c].  trans_* lists, for every camera c, the stored blocks of OTHER rows whose column is c
    // (the upper triangle by symmetry) for the symmetric mat-vec.
    bool sym_lower = false;
    std::vector<int> trans_ptr, trans_blk, trans_row;
    // Schur pair lists (this rank's observations): for row c the entries (j, j2) = (observation of c, observation of the
    // same point by a camera c2 of row c), grouped by slot and padded with -1 to whole 64-lane batches.
    std::vector<int> pair_j, pair_j2, pair_p, batch_slot, cam_batch_ptr;
    // work chunks for the pair kernel: <= 16 consecutive batches of ONE camera each (balances rows of very different size)
    std::vector<int> chunk_cam, chunk_b0, chunk_b1;
};

// Cuthill-McKee order of the camera graph given as block-CSR structure (folds a ring into a band of twice
// its reach; any connected "video-like" graph becomes a narrow band).  Returns the half-bandwidth.
inline int cuthill_mckee(int n, const std::vector<int>& row_ptr, const std::vector<int>& col_idx, std::vector<int>& pos,
                         std::vector<int>* comp_ptr = nullptr) {
    pos.assign(n, -1);
    if (comp_ptr) comp_ptr->assign(1, 0);
    std::vector<int> order; order.reserve(n);
    std::vector<int> queue, nb;
    auto degree = [&](int v) { return row_ptr[v + 1] - row_ptr[v]; };
    auto far_node = [&](int start) {            // last node of a BFS restricted to unplaced nodes
        std::vector<char> seen(n, 0); std::vector<int> q{start}; seen[start] = 1; size_t h = 0;
        while (h < q.size()) { int u = q[h++]; for (int e = row_ptr[u]; e < row_ptr[u + 1]; e++) { int v = col_idx[e]; if (!seen[v] && pos[v] < 0) { seen[v] = 1; q.push_back(v); } } }
        return q.back();
    };
    for (int s0 = 0; s0 < n; s0++) {
        if (pos[s0] >= 0) continue;
        int start = far_node(far_node(s0));
        queue.assign(1, start); pos[start] = (int)order.size(); order.push_back(start);
        size_t head = 0;
        while (head < queue.size()) {
            const int u = queue[head++];
            nb.clear();
            for (int e = row_ptr[u]; e < row_ptr[u + 1]; e++) { const int v = col_idx[e]; if (pos[v] < 0) { pos[v] = -2; nb.push_back(v); } }
            std::sort(nb.begin(), nb.end(), [&](int a, int b) { return degree(a) != degree(b) ? degree(a) < degree(b) : a < b; });
            for (int v : nb) { pos[v] = (int)order.size(); order.push_back(v); queue.push_back(v); }
        }
        if (comp_ptr) comp_ptr->push_back((int)order.size());
    }
    int band = 0;
    for (int u = 0; u < n; u++) for (int e = row_ptr[u]; e < row_ptr[u + 1]; e++) band = std::max(band, std::abs(pos[u] - pos[col_idx[e]]));
    return band;
}

inline void ba_flatten(const ssfm_ba_problem& P, int nranks, int rank, BAFlat& F) {
    const int Nc = P.num_cameras, Np = P.num_points;
    const int64_t M = P.num_observations;
    F = BAFlat();
    F.Nc = Nc; F.focal_free = !P.focal_fixed;
    if (Nc == 0 || Np == 0 || M == 0) { F.nothing_to_do = true; return; }
    // ---- order observations point-major / camera-ascending (skip the sort if they already are)
    bool sorted = true;
    for (int64_t i = 1; i < M && sorted; i++)
        if (P.obs_pt[i] < P.obs_pt[i - 1] || (P.obs_pt[i] == P.obs_pt[i - 1] && P.obs_cam[i] <= P.obs_cam[i - 1])) sorted = false;
    std::vector<int64_t> order;
    if (!sorted) {
        order.resize(M); std::iota(order.begin(), order.end(), (int64_t)0);
        std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) {
            if (P.obs_pt[a] != P.obs_pt[b]) return P.obs_pt[a] < P.obs_pt[b];
            return P.obs_cam[a] < P.obs_cam[b]; });
    }
    auto at = [&](int64_t i) { return sorted ? i : order[i]; };
    // ---- pass 1: which points are used (global), with their observation counts
    struct Seg { int pt; int64_t begin, end; int nobs; };
    std::vector<Seg> segs; segs.reserve(Np);
    for (int64_t i = 0; i < M;) {
        const int p = P.obs_pt[at(i)];
        int64_t e = i; int nobs = 0; int last_cam = -1;
        while (e < M && P.obs_pt[at(e)] == p) { const int c = P.obs_cam[at(e)]; if (c != last_cam && c >= 0 && c < Nc) { nobs++; last_cam = c; } e++; }
        bool valid = p >= 0 && p < Np && nobs >= 3;
        if (valid) { const double* X = &P.points[(size_t)p * 3]; valid = (X[0] * X[0] + X[1] * X[1] + X[2] * X[2]) != 0.0; }
        if (valid) segs.push_back({p, i, e, nobs});
        i = e;
    }
    F.nP_global = (int)segs.size();
    for (const Seg& s : segs) F.M_global += s.nobs;
    if (segs.empty()) { F.nothing_to_do = true; return; }
    // ---- structure of S and camera activity come from ALL used points (identical on every rank)
    std::vector<char> cam_in(Nc, 0);
    {
        std::vector<std::vector<int>> nb(Nc);
        std::vector<int> cams;
        for (const Seg& s : segs) {
            cams.clear(); int last = -1;
            for (int64_t i = s.begin; i < s.end; i++) { const int c = P.obs_cam[at(i)]; if (c != last && c >= 0 && c < Nc) { cams.push_back(c); last = c; } }
            for (int a : cams) { cam_in[a] = 1; for (int b : cams) if (nb[a].empty() || nb[a].back() != b) nb[a].push_back(b); }
        }
        F.row_ptr.assign(Nc + 1, 0); F.diag_slot.assign(Nc, -1);
        for (int c = 0; c < Nc; c++) {
            auto& v = nb[c]; std::sort(v.begin(), v.end()); v.erase(std::unique(v.begin(), v.end()), v.end());
            if (v.empty()) v.push_back(c);          // isolated camera: identity row
            F.row_ptr[c + 1] = F.row_ptr[c] + (int)v.size();
            F.max_row_blocks = std::max(F.max_row_blocks, (int)v.size());
        }
        F.col_idx.resize(F.row_ptr[Nc]);
        for (int c = 0; c < Nc; c++) {
            std::copy(nb[c].begin(), nb[c].end(), F.col_idx.begin() + F.row_ptr[c]);
            F.diag_slot[c] = (int)(std::lower_bound(nb[c].begin(), nb[c].end(), c) - nb[c].begin());
        }
    }
    F.band = cuthill_mckee(Nc, F.row_ptr, F.col_idx, F.cam_pos, &F.comp_ptr);
    for (int ir = 1; ir <= F.band; ir++) for (int kr = 1; kr <= ir; kr++) F.band_pairs.push_back(ir | (kr << 16));
    {   // keep the lower triangle (in elimination order) only
        std::vector<int> rp(Nc + 1, 0), ci; ci.reserve(F.col_idx.size() / 2 + Nc);
        F.max_row_blocks = 0;
        for (int c = 0; c < Nc; c++) {
            for (int e = F.row_ptr[c]; e < F.row_ptr[c + 1]; e++) { const int c2 = F.col_idx[e]; if (F.cam_pos[c2] <= F.cam_pos[c]) ci.push_back(c2); }
            rp[c + 1] = (int)ci.size();
            F.max_row_blocks = std::max(F.max_row_blocks, rp[c + 1] - rp[c]);
            F.diag_slot[c] = (int)(std::lower_bound(ci.begin() + rp[c], ci.end(), c) - (ci.begin() + rp[c]));
        }
        F.row_ptr.swap(rp); F.col_idx.swap(ci); F.sym_lower = true;
        std::vector<int> cnt(Nc + 1, 0);
        for (int c = 0; c < Nc; c++) for (int e = F.row_ptr[c]; e < F.row_ptr[c + 1]; e++) if (F.col_idx[e] != c) cnt[F.col_idx[e] + 1]++;
        for (int c = 0; c < Nc; c++) cnt[c + 1] += cnt[c];
        F.trans_ptr = cnt; F.trans_blk.resize(cnt[Nc]); F.trans_row.resize(cnt[Nc]);
        std::vector<int> fill(cnt.begin(), cnt.end() - 1);
        for (int c = 0; c < Nc; c++) for (int e = F.row_ptr[c]; e < F.row_ptr[c + 1]; e++) if (F.col_idx[e] != c) { const int k = fill[F.col_idx[e]]++; F.trans_blk[k] = e; F.trans_row[k] = c; }
    }
    F.mask_cam.assign((size_t)Nc * 6, 0.0);
    bool all_t_fixed = true;
    for (int c = 0; c < Nc; c++) {
        if (!cam_in[c]) continue;
        const bool tf = P.trans_fixed && P.trans_fixed[c], rf = P.rot_fixed && P.rot_fixed[c];
        if (!tf) { all_t_fixed = false; for (int k = 0; k < 3; k++) F.mask_cam[c * 6 + k] = 1.0; }
        if (!rf) for (int k = 0; k < 3; k++) F.mask_cam[c * 6 + 3 + k] = 1.0;
    }
    F.DC = all_t_fixed ? 3 : 6;
    // ---- this rank's contiguous share of the used points, balanced by observation count
    size_t s0 = 0, s1 = segs.size();
    if (nranks > 1) {
        const int64_t lo = F.M_global * rank / nranks, hi = F.M_global * (rank + 1) / nranks;
        int64_t acc = 0; s0 = s1 = segs.size(); bool have0 = false;
        for (size_t k = 0; k < segs.size(); k++) {
            if (!have0 && acc >= lo) { s0 = k; have0 = true; }
            if (acc >= hi) { s1 = k; break; }
            acc += segs[k].nobs;
        }
        if (!have0) s0 = segs.size();
        if (rank == nranks - 1) s1 = segs.size();
    }
    // ---- pass 2: emit local observations
    F.nP = (int)(s1 - s0);
    F.pt_ids.reserve(F.nP); F.pt_start.assign(1, 0); F.pts0.reserve((size_t)F.nP * 3); F.mask_pt.reserve((size_t)F.nP * 3);
    for (size_t k = s0; k < s1; k++) {
        const Seg& s = segs[k];
        for (int64_t i = s.begin; i < s.end; i++) {
            const int64_t o = at(i); const int c = P.obs_cam[o];
            if (c < 0 || c >= Nc) continue;
            if (i + 1 < s.end && P.obs_cam[at(i + 1)] == c) continue;     // keep the last duplicate
            F.obs_cam.push_back(c); F.obs_pt.push_back((int)F.pt_ids.size()); F.obs_orig.push_back(o);
            F.obs_xy.push_back(P.obs_xy[2 * o]); F.obs_xy.push_back(P.obs_xy[2 * o + 1]);
        }
        const double m = (P.pt_fixed && P.pt_fixed[s.pt]) ? 0.0 : 1.0;
        for (int d = 0; d < 3; d++) { F.pts0.push_back(P.points[(size_t)s.pt * 3 + d]); F.mask_pt.push_back(m); }
        F.pt_ids.push_back(s.pt); F.pt_start.push_back((int)F.obs_cam.size());
    }
    F.M = (int64_t)F.obs_cam.size();
    F.cam_start.assign(Nc + 1, 0);
    for (int64_t j = 0; j < F.M; j++) F.cam_start[F.obs_cam[j] + 1]++;
    for (int c = 0; c < Nc; c++) F.cam_start[c + 1] += F.cam_start[c];
    F.cam_obs.resize(F.M);
    { std::vector<int> fill(F.cam_start.begin(), F.cam_start.end() - 1);
      for (int64_t j = 0; j < F.M; j++) F.cam_obs[fill[F.obs_cam[j]]++] = (int)j; }
    F.cam_obs_pt.resize(F.M);
    for (int64_t q = 0; q < F.M; q++) F.cam_obs_pt[q] = F.obs_pt[F.cam_obs[q]];
    for (int c = 0; c < Nc; c++)
        for (int q = F.cam_start[c]; q < F.cam_start[c + 1]; q += 256) { F.cs_task_cam.push_back(c); F.cs_task_q0.push_back(q); F.cs_task_q1.push_back(std::min(q + 256, F.cam_start[c + 1])); }
    // ---- Schur pair lists, grouped by (row camera, slot), padded to 64-entry batches
    int task_batches = 16;                                   // batches per wave task of k_schur_pairs2 (tuning knob: SSFM_TASK_BATCHES)
    if (const char* e = std::getenv("SSFM_TASK_BATCHES")) task_batches = std::max(1, std::atoi(e));
    F.cam_batch_ptr.assign(Nc + 1, 0);
    std::vector<std::vector<int>> by_slot;      // reused per camera: entries (j, j2) interleaved
    for (int c = 0; c < Nc; c++) {
        const int rb = F.row_ptr[c], nnb = F.row_ptr[c + 1] - rb;
        by_slot.assign(nnb, std::vector<int>());
        for (int q = F.cam_start[c]; q < F.cam_start[c + 1]; q++) {
            const int j = F.cam_obs[q], p = F.obs_pt[j];
            for (int j2 = F.pt_start[p]; j2 < F.pt_start[p + 1]; j2++) {
                const int c2 = F.obs_cam[j2];
                if (F.cam_pos[c2] >= F.cam_pos[c]) continue;          // diagonal blocks are built by k_cam_sums
                const int slot = (int)(std::lower_bound(F.col_idx.begin() + rb, F.col_idx.begin() + rb + nnb, c2) - (F.col_idx.begin() + rb));
                by_slot[slot].push_back(j); by_slot[slot].push_back(j2);
            }
        }
        int nbatch = 0;
        for (int s2 = 0; s2 < nnb; s2++) {
            const int ne = (int)by_slot[s2].size() / 2; if (ne == 0) continue;
            const int nb = (ne + 63) / 64;
            for (int b = 0; b < nb; b++) F.batch_slot.push_back(s2);
            for (int e = 0; e < nb * 64; e++) {
                F.pair_j.push_back(e < ne ? by_slot[s2][2 * e] : -1); F.pair_j2.push_back(e < ne ? by_slot[s2][2 * e + 1] : -1);
                F.pair_p.push_back(e < ne ? F.obs_pt[by_slot[s2][2 * e]] : -1);
            }
            nbatch += nb;
        }
        F.cam_batch_ptr[c + 1] = F.cam_batch_ptr[c] + nbatch;
        for (int b = F.cam_batch_ptr[c]; b < F.cam_batch_ptr[c + 1]; b += task_batches) {
            F.chunk_cam.push_back(c); F.chunk_b0.push_back(b); F.chunk_b1.push_back(std::min(b + task_batches, F.cam_batch_ptr[c + 1]));
        }
    }
}

}  // namespace ssfm
