// spherical_sfm_amd -- host-side schedule of the cyclic reduction over the separator cycles of long camera rings (band_ring.h has the kernels and the algebra).
// Host-only, no HIP: tests/native/ring_schedule_check.cpp replays the records with plain dense arithmetic against a dense solve.
#pragma once
#include <algorithm>
#include <utility>
#include <vector>

namespace ssfm {

// record of one elimination (ints):
//   [0] separator v   [1] neighbours nn (0..2)   [2] hasL  [3] hasR: earlier eliminations left Schur updates of v in PL[v] / PR[v] (and tL / tR)
//   [4] first band row of v's copy slot or -1   [5] first band row of v   [6] 1 = store the fill -F_1 F_0^T between the two neighbours in ringE[v]
//   neighbour j at 8 + 16 j:  [0] separator u_j   [1] first band row of u_j   [2] terms (1..2)   [3] side of u_j that takes this elimination's update (0: PL / tL, 1: PR / tR)
//                             [14] 0 = first update of that side (store), 1 = add to what is there      term t at + 4 + 5 t:  [0] kind  [1] a  [2] transposed
//     kind 0  original coupling, held in Z at the rows of the separator whose first band row is a (E(i, c) = Z[c][a DC + i]):  B = E, or E^T when transposed
//     kind 1  fill left by the elimination of separator a in ringE[a]:  B = ringE[a], or its transpose
// Every separator has one record; the eliminations of a step are independent of each other.  The last steps of a ring (from RING_TAIL_P active separators on) go to
// the ring's TAIL list instead of the step lists: one workgroup runs them one after the other in one launch (band_ring.h: k_ring_cr_tail).
constexpr int RING_REC = 40;
constexpr int RING_TAIL_P = 2;

struct RingTermH { int kind, a, tr; };

// Schedule of all rings of a plan.  ring_seps: (first separator id, m) per ring; separators of a ring are consecutive ids in cyclic order, separator s0 + k lies behind arc k.
// out: rec (all records: the steps' first, then the tails'), step_ptr (records of parallel step s: [step_ptr[s], step_ptr[s+1])), tail_ptr (records of ring g's tail:
// [tail_ptr[g], tail_ptr[g+1]), in elimination order)
inline void ring_schedule(const std::vector<std::pair<int, int>>& ring_seps, const std::vector<int>& sep_lo, const std::vector<int>& sep_copy,
                          std::vector<int>& rec, std::vector<int>& step_ptr, std::vector<int>& tail_ptr) {
    rec.clear(); step_ptr.assign(1, 0); tail_ptr.clear();
    struct Node { int v, nn, nbr[2], side[2], accum[2], hasL, hasR, writeE; std::vector<RingTermH> terms[2]; };
    std::vector<std::vector<Node>> steps, tails(ring_seps.size());
    auto transposed = [](const RingTermH& t) { RingTermH r = t; r.tr = 1 - t.tr; return r; };
    int nsep_total = 0; for (const auto& rs : ring_seps) nsep_total = std::max(nsep_total, rs.first + rs.second);
    std::vector<char> wroteL(nsep_total, 0), wroteR(nsep_total, 0);
    for (size_t g = 0; g < ring_seps.size(); g++) {
        const int s0 = ring_seps[g].first, m = ring_seps[g].second;
        std::vector<int> act(m); for (int k = 0; k < m; k++) act[k] = s0 + k;
        // adj[i]: the terms of M[act[i+1]][act[i]] (rows = the later node of the pair in cyclic order)
        std::vector<std::vector<RingTermH>> adj(m);
        for (int k = 0; k < m; k++) adj[k].push_back(RingTermH{0, sep_lo[act[(k + 1) % m]], 0});
        size_t round = 0;
        auto emit = [&](const Node& nd, int p) { if (p <= RING_TAIL_P) tails[g].push_back(nd); else { if (steps.size() <= round) steps.resize(round + 1); steps[round].push_back(nd); } };
        auto target = [&](Node& nd, int j, int node, int side) { nd.side[j] = side; char& w = side ? wroteR[node] : wroteL[node]; nd.accum[j] = w; w = 1; };
        for (;;) {
            const int p = (int)act.size();
            Node nd; nd.nn = 0; nd.writeE = 0; nd.nbr[0] = nd.nbr[1] = 0; nd.side[0] = nd.side[1] = 0; nd.accum[0] = nd.accum[1] = 0;
            if (p == 1) { nd.v = act[0]; nd.hasL = wroteL[nd.v]; nd.hasR = wroteR[nd.v]; emit(nd, p); break; }
            if (p == 2) {
                nd.v = act[0]; nd.nn = 1; nd.nbr[0] = act[1]; nd.hasL = wroteL[nd.v]; nd.hasR = wroteR[nd.v];
                for (const auto& t : adj[0]) nd.terms[0].push_back(t);                     // M[c][a] as stored
                for (const auto& t : adj[1]) nd.terms[0].push_back(transposed(t));         // M[a][c] transposed
                target(nd, 0, act[1], 0);
                emit(nd, p);
                act.erase(act.begin()); adj.clear(); adj.resize(1);
                round++; continue;
            }
            std::vector<int> act2; std::vector<std::vector<RingTermH>> adj2;
            for (int i = 0; i < p; i += 2) {
                act2.push_back(act[i]);
                if (i + 1 < p) {                                                            // eliminate act[i+1] between act[i] and act[i+2 mod p]
                    const int u = act[i], v = act[i + 1], w = act[(i + 2) % p];
                    Node e = nd; e.v = v; e.nn = 2; e.nbr[0] = u; e.nbr[1] = w; e.hasL = wroteL[v]; e.hasR = wroteR[v]; e.writeE = 1;
                    for (const auto& t : adj[i]) e.terms[0].push_back(transposed(t));      // M[u][v] = (M[v][u])^T
                    for (const auto& t : adj[i + 1]) e.terms[1].push_back(t);              // M[w][v]
                    target(e, 0, u, 1); target(e, 1, w, 0);                                // v is u's right neighbour and w's left one
                    emit(e, p);
                    adj2.push_back({RingTermH{1, v, 0}});                                   // M[w][u] -= F_w F_u^T, kept in ringE[v] (rows w)
                } else adj2.push_back(adj[i]);                                              // odd p: the pair (act[p-1], act[0]) keeps its coupling
            }
            act.swap(act2); adj.swap(adj2);
            round++;
        }
    }
    auto put = [&](const Node& nd) {
        std::vector<int> r(RING_REC, 0);
        r[0] = nd.v; r[1] = nd.nn; r[2] = nd.hasL; r[3] = nd.hasR; r[4] = sep_copy[nd.v]; r[5] = sep_lo[nd.v]; r[6] = nd.writeE;
        for (int j = 0; j < nd.nn; j++) {
            int* q = r.data() + 8 + 16 * j;
            q[0] = nd.nbr[j]; q[1] = sep_lo[nd.nbr[j]]; q[2] = (int)nd.terms[j].size(); q[3] = nd.side[j]; q[14] = nd.accum[j];
            for (size_t t = 0; t < nd.terms[j].size() && t < 2; t++) { int* z = q + 4 + 5 * (int)t; z[0] = nd.terms[j][t].kind; z[1] = nd.terms[j][t].a; z[2] = nd.terms[j][t].tr; }
        }
        rec.insert(rec.end(), r.begin(), r.end());
    };
    for (const auto& st : steps) { for (const Node& nd : st) put(nd); step_ptr.push_back((int)rec.size() / RING_REC); }
    tail_ptr.push_back((int)rec.size() / RING_REC);
    for (const auto& tl : tails) { for (const Node& nd : tl) put(nd); tail_ptr.push_back((int)rec.size() / RING_REC); }
}

}  // namespace ssfm
