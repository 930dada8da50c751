// ORACLE (test infrastructure only) -- direct solve of the sparse SPD systems.
//
// The reference solves its normal equations with Ceres' direct sparse Cholesky
// (SPARSE_SCHUR at src/sfm.cpp:273, SPARSE_NORMAL_CHOLESKY at src/rotation_averaging.cpp:77 and
// src/uncalibrated_pose_graph.cpp:188; SuiteSparse is pinned in docker/Dockerfile:23 and is not in
// /root/reference).  Any exact factorisation gives the same step up to rounding, so the oracle
// uses the simplest one that still exploits sparsity: reverse Cuthill-McKee on the block graph,
// then a row-envelope ("skyline") Cholesky.  Fill stays inside the envelope.
#pragma once
#include <algorithm>
#include <cmath>
#include <queue>
#include <vector>

namespace oracle {

// Cuthill-McKee order (reversed) of an undirected graph given as adjacency lists.
// Returns order[k] = node placed at position k.  Handles disconnected graphs.
inline std::vector<int> rcm_order(const std::vector<std::vector<int>>& adj) {
    const int n = (int)adj.size();
    std::vector<int> order; order.reserve(n);
    std::vector<char> seen(n, 0);
    auto bfs_last = [&](int start, std::vector<int>* levels_out) {
        // plain BFS returning the last node visited (pseudo-peripheral search)
        std::vector<int> dist(n, -1); std::queue<int> q; q.push(start); dist[start] = 0; int last = start;
        while (!q.empty()) { int u = q.front(); q.pop(); last = u;
            for (int v : adj[u]) if (dist[v] < 0 && !seen[v]) { dist[v] = dist[u] + 1; q.push(v); } }
        if (levels_out) *levels_out = dist;
        return last;
    };
    for (int s0 = 0; s0 < n; s0++) {
        if (seen[s0]) continue;
        int start = s0;
        for (int it = 0; it < 2; it++) start = bfs_last(start, nullptr);   // walk to a far end
        std::queue<int> q; q.push(start); seen[start] = 1;
        while (!q.empty()) {
            int u = q.front(); q.pop(); order.push_back(u);
            std::vector<int> nb;
            for (int v : adj[u]) if (!seen[v]) { seen[v] = 1; nb.push_back(v); }
            std::sort(nb.begin(), nb.end(), [&](int a, int b) {
                if (adj[a].size() != adj[b].size()) return adj[a].size() < adj[b].size(); return a < b; });
            for (int v : nb) q.push(v);
        }
    }
    std::reverse(order.begin(), order.end());
    return order;
}

// Symmetric matrix, lower triangle, row i holds columns [first[i], i].
struct Skyline {
    int n = 0;
    std::vector<int> first;
    std::vector<long long> ptr;   // ptr[i] + (j - first[i])
    std::vector<double> val;
    void init(const std::vector<int>& first_col) {
        n = (int)first_col.size(); first = first_col; ptr.assign(n + 1, 0);
        for (int i = 0; i < n; i++) ptr[i + 1] = ptr[i] + (i - first[i] + 1);
        val.assign((size_t)ptr[n], 0.0);
    }
    void zero() { std::fill(val.begin(), val.end(), 0.0); }
    inline double& at(int i, int j) { return val[(size_t)(ptr[i] + (j - first[i]))]; }   // requires first[i] <= j <= i
    // in-place Cholesky A = L L^T; false if not positive definite
    bool factor() {
        for (int i = 0; i < n; i++) {
            double* Li = &val[(size_t)ptr[i]] - first[i];
            for (int j = first[i]; j < i; j++) {
                const double* Lj = &val[(size_t)ptr[j]] - first[j];
                const int k0 = std::max(first[i], first[j]);
                double s = Li[j];
                for (int k = k0; k < j; k++) s -= Li[k] * Lj[k];
                Li[j] = s / Lj[j];
            }
            double s = Li[i];
            for (int k = first[i]; k < i; k++) s -= Li[k] * Li[k];
            if (!(s > 0.0) || !std::isfinite(s)) return false;
            Li[i] = std::sqrt(s);
        }
        return true;
    }
    // solve L L^T x = b in place
    void solve(double* b) const {
        for (int i = 0; i < n; i++) {
            const double* Li = &val[(size_t)ptr[i]] - first[i];
            double s = b[i];
            for (int k = first[i]; k < i; k++) s -= Li[k] * b[k];
            b[i] = s / Li[i];
        }
        for (int i = n - 1; i >= 0; i--) {
            const double* Li = &val[(size_t)ptr[i]] - first[i];
            b[i] /= Li[i];
            const double xi = b[i];
            for (int k = first[i]; k < i; k++) b[k] -= Li[k] * xi;
        }
    }
};

}  // namespace oracle
