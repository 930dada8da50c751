// spherical_sfm_amd -- host-side schedule of the cyclic reduction over the separator cycles of long camera rings (band_ring.h has the kernels and the algebra).
// Host-only, no HIP: tests/native/ring_schedule_check.cpp replays the records with plain dense arithmetic against a dense solve.
#pragma once
#include <algorithm>
#include <utility>
#include <vector>

namespace ssfm {

// record of one elimination (ints):
//   [0] separator v   [1] neighbours nn (0..2)   [2] pend_lo   [3] pend_hi (pairs (x, slot) in ring_pend)   [4] first band row of v's copy slot or -1   [5] first band row of v
//   neighbour j at 8 + 16 j:  [0] separator u_j   [1] first band row of u_j   [2] terms (1..2)   term t at + 4 + 6 t:  [0] kind  [1] a  [2] b  [3] c
//     kind 0  original coupling held in Z at the rows of the separator whose first band row is a:  b = 0: B(i, c) = Z[c][a DC + i],  b = 1: B(i, c) = Z[i][a DC + c]
//     kind 1  fill through the eliminated separator a:  B -= F[a][slot b] F[a][slot c]^T
constexpr int RING_REC = 40;

struct RingTermH { int kind, a, b, c; };

// Schedule of all rings of a plan.  ring_seps: (first separator id, m) per ring; separators of a ring are consecutive ids in cyclic order, separator s0 + k lies behind arc k.
inline void ring_schedule(const std::vector<std::pair<int, int>>& ring_seps, const std::vector<int>& sep_lo, const std::vector<int>& sep_copy,
                          std::vector<int>& rec, std::vector<int>& step_ptr, std::vector<int>& pend_out) {
    rec.clear(); pend_out.clear(); step_ptr.assign(1, 0);
    struct Node { int v, nn, nbr[2]; std::vector<RingTermH> terms[2]; std::vector<std::pair<int, int>> pend; };
    std::vector<std::vector<Node>> steps;
    auto transposed = [](const RingTermH& t) { RingTermH r = t; if (t.kind == 0) r.b = 1 - t.b; else { r.b = t.c; r.c = t.b; } return r; };
    for (const auto& rs : ring_seps) {
        const int s0 = rs.first, m = rs.second;
        std::vector<int> act(m); for (int k = 0; k < m; k++) act[k] = s0 + k;
        // adj[i]: the terms of M[act[i+1]][act[i]] (rows = the later node of the pair in cyclic order)
        std::vector<std::vector<RingTermH>> adj(m);
        for (int k = 0; k < m; k++) adj[k].push_back(RingTermH{0, sep_lo[act[(k + 1) % m]], 0, 0});
        std::vector<std::vector<std::pair<int, int>>> pend(m);            // by separator id - s0
        size_t round = 0;
        for (;;) {
            if (steps.size() <= round) steps.emplace_back();
            const int p = (int)act.size();
            if (p == 1) { Node nd; nd.v = act[0]; nd.nn = 0; nd.pend = pend[act[0] - s0]; steps[round].push_back(nd); break; }
            if (p == 2) {
                Node nd; nd.v = act[0]; nd.nn = 1; nd.nbr[0] = act[1]; nd.pend = pend[act[0] - s0];
                for (const auto& t : adj[0]) nd.terms[0].push_back(t);                     // M[c][a] as stored
                for (const auto& t : adj[1]) nd.terms[0].push_back(transposed(t));         // M[a][c] transposed
                pend[act[1] - s0].push_back({act[0], 0});
                steps[round].push_back(nd);
                act.erase(act.begin()); adj.clear(); adj.resize(1);
                round++; continue;
            }
            std::vector<int> act2; std::vector<std::vector<RingTermH>> adj2;
            for (int i = 0; i < p; i += 2) {
                act2.push_back(act[i]);
                if (i + 1 < p) {                                                            // eliminate act[i+1] between act[i] and act[i+2 mod p]
                    const int u = act[i], v = act[i + 1], w = act[(i + 2) % p];
                    Node nd; nd.v = v; nd.nn = 2; nd.nbr[0] = u; nd.nbr[1] = w; nd.pend = pend[v - s0];
                    for (const auto& t : adj[i]) nd.terms[0].push_back(transposed(t));     // M[u][v] = (M[v][u])^T
                    for (const auto& t : adj[i + 1]) nd.terms[1].push_back(t);             // M[w][v]
                    pend[u - s0].push_back({v, 0}); pend[w - s0].push_back({v, 1});
                    steps[round].push_back(nd);
                    adj2.push_back({RingTermH{1, v, 1, 0}});                                // M[w][u] -= F_{v,w} F_{v,u}^T
                } else adj2.push_back(adj[i]);                                              // odd p: the pair (act[p-1], act[0]) keeps its coupling
            }
            act.swap(act2); adj.swap(adj2);
            round++;
        }
    }
    for (const auto& st : steps) {
        for (const Node& nd : st) {
            std::vector<int> r(RING_REC, 0);
            r[0] = nd.v; r[1] = nd.nn; r[2] = (int)pend_out.size() / 2;
            for (const auto& pr : nd.pend) { pend_out.push_back(pr.first); pend_out.push_back(pr.second); }
            r[3] = (int)pend_out.size() / 2; r[4] = sep_copy[nd.v]; r[5] = sep_lo[nd.v];
            for (int j = 0; j < nd.nn; j++) {
                int* q = r.data() + 8 + 16 * j;
                q[0] = nd.nbr[j]; q[1] = sep_lo[nd.nbr[j]]; q[2] = (int)nd.terms[j].size();
                for (size_t t = 0; t < nd.terms[j].size() && t < 2; t++) { int* z = q + 4 + 6 * (int)t; z[0] = nd.terms[j][t].kind; z[1] = nd.terms[j][t].a; z[2] = nd.terms[j][t].b; z[3] = nd.terms[j][t].c; }
            }
            rec.insert(rec.end(), r.begin(), r.end());
        }
        step_ptr.push_back((int)rec.size() / RING_REC);
    }
}

}  // namespace ssfm
