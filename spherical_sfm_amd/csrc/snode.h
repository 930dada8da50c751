// spherical_sfm_amd -- supernodal solver of the reduced camera system for rings and chains of cameras (round 6).
//
// What it replaces: the direct solve of the reduced system that Ceres' SPARSE_SCHUR does with a sparse Cholesky (reference src/sfm.cpp:205,273), on the
// structures the reference's drivers produce for a closed camera loop (examples/spherical_sfm_tools.cpp:887-950: a track couples the cameras it spans, so
// the reduced matrix of a loop is a PERIODIC block band of half-width r = longest track - 1).
//
// The idea.  The windowed block-band factorisation (band_kernels2.h) is a chain of one dependent step per CAMERA -- 1.27 us each at BASELINE config 2,
// 42 + 15 + 5 + 16 us per LM iteration in four launches, 45 % of the iteration on <= 12 of 256 compute units.  The arithmetic is nothing (6.5 Mflop); what costs is the
// number of dependent steps and what one step waits for.  Here a step eliminates a SUPERNODE of s = 30 / DC cameras (30 scalar columns): the tall panel
//      [ diagonal block (30 rows) ; coupling to the next supernode (30 rows) ]
// is factored by ONE wave with one lane per row, right-looking (sn_panel: the pivot entry and the two rows that become pivots next take the column through
// v_readlane, every other row reads L(c2, c) back from LDS at a wave-uniform address one column later).  A ring in its own circular order with reach r <= s is block
// tridiagonal in supernodes plus one separator T that closes it; it is eliminated from both sides of T towards a middle supernode M by two workgroups (halves A and B,
// 7 steps each at config 2 instead of 33 + 10), which exchange their contributions to [M, T] ONCE through global memory, both solve the 60...90-row remainder
// redundantly (a + b == b + a: identical bits) and back-substitute their own half.  A second wave takes the rows of T and every right-hand side from the finished
// panel one stage behind (sn_trows), two more gather the next supernode's blocks from S and send factor blocks to global memory, the Schur updates run on the matrix
// cores (v_mfma_f64_16x16x4), the back substitution is spread over the four waves.  One launch does the factorisation, both substitutions and the scatter of the
// solution into the layout k_arrow_update reads.  No atomics anywhere: the result does not depend on scheduling.
//
// MEASURED SLOWER than what it replaces (112 us against 86 us at config 2; phase stamps and the five builds in profiles/r06_notes.md): a panel is ~2300 instructions
// on one wave -- 3.8 us per five cameras, the same issue-bound regime as the windowed kernel -- and the stage around it (barriers, tiles, the slower helper waves,
// the exchange) eats the rest.  Hence opt-in (snode_enabled below).
//
// Applicability (snode_plan below): every connected component of the camera graph is a chain or a ring with reach <= s cameras in some linear / circular
// order (camera ids, or a greedy walk), rings have >= 4 s cameras, halves <= SN_MAXSTEPS steps, and all workgroups are resident at once.  Everything else
// keeps the band kernels (ragged tracks with reach > 30 / DC, very long rings, wide pose graphs).
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <hip/hip_runtime.h>
#include "band_kernels2.h"

namespace ssfm {

constexpr int SNQ = 30;            // scalar rows of a supernode (5 cameras x 6 or 10 cameras x 3)
constexpr int SN_CS = 66;          // stride of one column of the panel in LDS: [column][lane], lanes 0..29 the pivot's rows, 32..61 the next supernode's
constexpr int SN_LD = 31;          // leading dimension of the Q-column LDS / workspace matrices (odd: rows of consecutive lanes fall into different banks)
constexpr int SN_QTMAX = 60;       // rows of T: s + (n mod s) cameras <= 2 s - 1
constexpr int SN_MAXSTEPS = 16;    // pivots of one half
constexpr int SN_THREADS = 256;
constexpr int SN_HREC = 16, SN_SREC = 8;

// Opt-in (SSFM_SNODE=1): measured at BASELINE config 2 the launch takes 112 us against 86 us for the four launches of the band kernels it replaces (phase
// stamps in profiles/r06_notes.md), so the LM loop keeps the band kernels by default; the probes below run it regardless.
inline bool snode_enabled() { const char* e = std::getenv("SSFM_SNODE"); return e && std::atoi(e) != 0; }

// ---------------------------------------------------------------------------------------------------------------------------------------------
// Host plan
// ---------------------------------------------------------------------------------------------------------------------------------------------
struct SnodePlan {
    bool enabled = false;
    int DC = 6, S = 5, CAPT = 10;      // cameras per supernode; capacity of a node's camera list (T: up to 2 S - 1)
    int nhalf = 0;                      // workgroups
    int qtm = 0;                        // rows reserved for T in LDS / workspace (0: chains only), a multiple of 2
    int nflags = 0;
    size_t work_doubles = 0, xchg_doubles = 0;
    std::vector<int> half_rec;          // [nhalf][SN_HREC]
    std::vector<int> step_rec;          // [steps][SN_SREC]
    std::vector<int> node_cam;          // [nodes][CAPT] camera ids, -1 = padding
    std::vector<int> tab;               // block tables: entry = slot of S | 1 << 30 when the stored block is the transpose; -1 zero block; -2 identity (padding diagonal)
    size_t work_step() const { return (size_t)2 * SNQ * SN_LD + (size_t)qtm * SN_LD + 2 * SNQ; }
    size_t xchg_half() const { return (size_t)SNQ * SN_LD + 2 * SNQ + (size_t)qtm * SN_LD + (size_t)qtm * (qtm + 1) + 2 * qtm; }
    size_t lds_bytes() const {      // k_snode_solve's carve-up: D, C (two each), E (two), the panel columns (two), LTP, T x T, right-hand sides, small vectors
        const size_t d = (size_t)4 * SNQ * SN_LD + (size_t)2 * qtm * SN_LD + (size_t)2 * SNQ * SN_CS + (size_t)qtm * SN_LD + (size_t)qtm * (qtm + 1)
                       + (size_t)(SN_MAXSTEPS + 2) * 2 * SNQ + 6 * SNQ + (size_t)6 * qtm + 64;
        return d * sizeof(double);
    }
};
// half_rec: [0] steps [1] first step record [2] partner half (-1: none) [3] 1 = this half owns the original data of M and T (and writes their solution)
//           [4] node of T (-1: chain) [5] rows of T [6] workspace offset (doubles) [7] exchange offset of this half (doubles) [8] flag of this half
//           [9] table of D(P_0) [10] table of E(T, P_0) (-1: zero) [11] table of (T, T) (-1: zero / not the owner) [12] node M [13] node P_0 [14] exchange offset of the partner [15] flag of the partner
// step_rec: [0] node P [1] node N [2] table of C(N, P) [3] table of D(N) (-1: zero) [4] table of E(T, N) (-1: zero) [5] 1 = the right-hand side of N is loaded (0: zero)

// the order of a component's cameras in which it is a narrow (periodic) band, and its reach there; returns 0 = none, 1 = chain, 2 = ring
inline int snode_order(const std::vector<int>& comp, const std::vector<std::vector<int>>& adj, int S, std::vector<int>& order) {
    const int n = (int)comp.size();
    std::vector<int> ids(comp); std::sort(ids.begin(), ids.end());
    // local index of a camera
    std::vector<std::pair<int, int>> where_v; where_v.reserve(n);
    auto reach_of = [&](const std::vector<int>& ord, int& lin, int& circ) {
        std::vector<std::pair<int, int>> w(n); for (int i = 0; i < n; i++) w[i] = {ord[i], i};
        std::sort(w.begin(), w.end());
        auto idx = [&](int c) { return std::lower_bound(w.begin(), w.end(), std::make_pair(c, -1))->second; };
        lin = 0; circ = 0;
        for (int i = 0; i < n; i++) for (int v : adj[ord[i]]) { const int d = std::abs(i - idx(v)); lin = std::max(lin, d); circ = std::max(circ, std::min(d, n - d)); }
    };
    // greedy walk: from the current camera to the unvisited neighbour that shares the most neighbours with it; start at a far end (two sweeps of breadth-first search)
    std::vector<int> walk; walk.reserve(n);
    {
        std::vector<std::pair<int, int>> w(n); for (int i = 0; i < n; i++) w[i] = {ids[i], i};
        auto idx = [&](int c) { return std::lower_bound(w.begin(), w.end(), std::make_pair(c, -1))->second; };
        auto far = [&](int s0) { std::vector<int> dist(n, -1), q(1, s0); dist[s0] = 0; size_t hd = 0; int last = s0;
                                 while (hd < q.size()) { const int u = q[hd++]; last = u; for (int v : adj[ids[u]]) { const int vi = idx(v); if (dist[vi] < 0) { dist[vi] = dist[u] + 1; q.push_back(vi); } } }
                                 return last; };
        int cur = far(far(0));
        std::vector<char> vis(n, 0), mine(n, 0);
        vis[cur] = 1; walk.push_back(ids[cur]); int scan = 0;
        for (int step = 1; step < n; step++) {
            for (int v : adj[ids[cur]]) mine[idx(v)] = 1;
            int best = -1, best_cnt = -1;
            int best_open = 0;                                              // ties: the candidate with the fewest unvisited neighbours (the one next to what has been laid out), then the smaller id
            for (int v : adj[ids[cur]]) {
                const int vi = idx(v); if (vis[vi]) continue;
                int cnt = 0, open = 0; for (int v2 : adj[v]) { const int wi = idx(v2); cnt += mine[wi]; open += vis[wi] ? 0 : 1; }
                if (cnt > best_cnt || (cnt == best_cnt && (open < best_open || (open == best_open && vi < best)))) { best_cnt = cnt; best = vi; best_open = open; }
            }
            for (int v : adj[ids[cur]]) mine[idx(v)] = 0;
            if (best < 0) { while (scan < n && vis[scan]) scan++; best = scan; }
            vis[best] = 1; walk.push_back(ids[best]); cur = best;
        }
    }
    int la, ca, lb, cb; reach_of(ids, la, ca); reach_of(walk, lb, cb);
    if (std::min(la, lb) <= S) { order = (la <= lb) ? ids : walk; return 1; }
    if (std::min(ca, cb) <= S && n >= 4 * S) { order = (ca <= cb) ? ids : walk; return 2; }
    return 0;
}

// row_ptr / col_idx: the stored blocks of S (each coupled pair of cameras once, in either orientation, plus the diagonal blocks)
inline bool snode_plan(int Nc, int DC, const std::vector<int>& row_ptr, const std::vector<int>& col_idx, int num_cus, SnodePlan& P) {
    P = SnodePlan();
    if ((DC != 3 && DC != 6) || Nc < 1) return false;
    const int S = SNQ / DC, CAPT = 2 * S;
    P.DC = DC; P.S = S; P.CAPT = CAPT;
    std::vector<std::vector<int>> adj(Nc);
    for (int c = 0; c < Nc; c++) for (int e = row_ptr[c]; e < row_ptr[c + 1]; e++) { const int c2 = col_idx[e]; if (c2 != c) { adj[c].push_back(c2); adj[c2].push_back(c); } }
    for (auto& a : adj) { std::sort(a.begin(), a.end()); a.erase(std::unique(a.begin(), a.end()), a.end()); }
    auto lookup = [&](int r, int c) -> int {          // block (r, c) of S
        for (int e = row_ptr[r]; e < row_ptr[r + 1]; e++) if (col_idx[e] == c) return e;
        for (int e = row_ptr[c]; e < row_ptr[c + 1]; e++) if (col_idx[e] == r) return e | (1 << 30);
        return -1;
    };
    auto add_node = [&](const int* cams, int n) { const int id = (int)(P.node_cam.size() / CAPT); for (int a = 0; a < CAPT; a++) P.node_cam.push_back(a < n ? cams[a] : -1); return id; };
    // table of the matrix (rows: node R, columns: node C), row-major [rows of R in cameras][ncb]; pad_identity: R == C, padding cameras get an identity diagonal; returns -1 when every block is zero
    auto add_table = [&](int R, int C, int ncb, bool pad_identity) {
        const int off = (int)P.tab.size(); bool any = false;
        for (int a = 0; a < CAPT; a++) for (int b = 0; b < ncb; b++) {
            const int ca = P.node_cam[(size_t)R * CAPT + a], cb = (b < CAPT) ? P.node_cam[(size_t)C * CAPT + b] : -1;
            int v = -1;
            if (ca >= 0 && cb >= 0) v = lookup(ca, cb);
            else if (pad_identity && a == b && ca < 0 && cb < 0 && a < ncb) v = -2;
            if (v != -1) any = true;
            P.tab.push_back(v);
        }
        if (!any) { P.tab.resize(off); return -1; }
        return off;
    };
    std::vector<char> seen(Nc, 0);
    int qt_max = 0; size_t work = 0;
    struct Half { std::vector<int> piv; int M, T, qt; bool owner; int partner; };
    std::vector<Half> halves;
    for (int c0 = 0; c0 < Nc; c0++) {
        if (seen[c0]) continue;
        std::vector<int> comp(1, c0); seen[c0] = 1;
        for (size_t hd = 0; hd < comp.size(); hd++) for (int v : adj[comp[hd]]) if (!seen[v]) { seen[v] = 1; comp.push_back(v); }
        std::vector<int> order;
        const int kind = snode_order(comp, adj, S, order);
        if (kind == 0) return false;
        const int n = (int)order.size();
        std::vector<int> nodes; int T = -1, qt = 0;
        if (kind == 1) {
            const int m = (n + S - 1) / S, first = n - (m - 1) * S;
            int at = 0;
            for (int k = 0; k < m; k++) { const int len = k == 0 ? first : S; nodes.push_back(add_node(order.data() + at, len)); at += len; }
        } else {
            const int tq = S + n % S, m = (n - tq) / S;
            if (m < 3) return false;
            T = add_node(order.data(), tq); qt = tq * DC;
            for (int k = 0; k < m; k++) nodes.push_back(add_node(order.data() + tq + k * S, S));
        }
        const int m = (int)nodes.size();
        Half A, B; A.T = B.T = T; A.qt = B.qt = qt; A.owner = true; B.owner = false;
        if (m >= 3) {
            const int mi = m / 2;
            for (int k = 0; k < mi; k++) A.piv.push_back(nodes[k]);
            for (int k = m - 1; k > mi; k--) B.piv.push_back(nodes[k]);
            A.M = B.M = nodes[mi];
            A.partner = (int)halves.size() + 1; B.partner = (int)halves.size();
            halves.push_back(A); halves.push_back(B);
        } else {
            for (int k = 0; k + 1 < m; k++) A.piv.push_back(nodes[k]);
            A.M = nodes[m - 1]; A.partner = -1;
            halves.push_back(A);
        }
        qt_max = std::max(qt_max, qt);
    }
    P.nhalf = (int)halves.size();
    if (P.nhalf > num_cus) return false;                                   // partners wait for each other: every workgroup must be resident
    P.qtm = (qt_max + 1) & ~1;
    if (P.qtm > SN_QTMAX) return false;
    for (const Half& h : halves) if ((int)h.piv.size() > SN_MAXSTEPS) return false;
    if (P.lds_bytes() > 160 * 1024) return false;
    P.half_rec.assign((size_t)P.nhalf * SN_HREC, -1);
    for (int hi = 0; hi < P.nhalf; hi++) {
        const Half& h = halves[hi];
        int* r = &P.half_rec[(size_t)hi * SN_HREC];
        const int ns = (int)h.piv.size();
        r[0] = ns; r[1] = (int)(P.step_rec.size() / SN_SREC); r[2] = h.partner; r[3] = h.owner ? 1 : 0; r[4] = h.T; r[5] = h.qt;
        r[6] = (int)work; work += (size_t)std::max(ns, 1) * P.work_step();
        r[7] = (int)((size_t)hi * P.xchg_half()); r[8] = hi;
        r[14] = h.partner >= 0 ? (int)((size_t)h.partner * P.xchg_half()) : -1; r[15] = h.partner;
        const int P0 = ns > 0 ? h.piv[0] : h.M;
        r[13] = P0; r[12] = h.M;
        // D(P_0): the first pivot, or M itself when there is none; a half that does not own M and has no pivot cannot exist (partners have >= 1 step each)
        r[9] = add_table(P0, P0, S, true);
        r[10] = h.T >= 0 ? add_table(h.T, P0, S, false) : -1;
        if (ns == 0 && !h.owner) return false;
        r[11] = (h.T >= 0 && h.owner) ? add_table(h.T, h.T, CAPT, false) : -1;
        for (int k = 0; k < ns; k++) {
            const int Pn = h.piv[k], Nn = (k + 1 < ns) ? h.piv[k + 1] : h.M;
            const bool orig = (k + 1 < ns) || h.owner;                     // the original blocks of M are the owner's
            int s8[SN_SREC] = {Pn, Nn, add_table(Nn, Pn, S, false), orig ? add_table(Nn, Nn, S, true) : -1, (h.T >= 0 && orig) ? add_table(h.T, Nn, S, false) : -1, orig ? 1 : 0, 0, 0};
            if (s8[4] >= 0) return false;                                  // a supernode behind the first pivot that touches T: cannot happen with >= 3 supernodes per ring; the kernel relies on it
            for (int q = 0; q < SN_SREC; q++) P.step_rec.push_back(s8[q]);
        }
    }
    P.work_doubles = work; P.xchg_doubles = (size_t)P.nhalf * P.xchg_half(); P.nflags = P.nhalf;
    P.enabled = true;
    return true;
}

// ---------------------------------------------------------------------------------------------------------------------------------------------
// Device
// ---------------------------------------------------------------------------------------------------------------------------------------------
typedef double sn_v4d __attribute__((ext_vector_type(4)));

// dst[i * ld + j] (i < nrows, j < ncols) <- the matrix a block table describes (tab < 0: zeros); nt cooperating threads, this one is t; batches of B entries per
// thread: the table entries first, then the values, then the LDS stores -- two memory round trips per batch
template <int DC, int B>
__device__ __forceinline__ void sn_gather_b(double* dst, int ld, int nrows, int ncols, const double* __restrict__ S_val, const int* __restrict__ tabs, int tab, int ncb, int t, int nt) {
    constexpr int BB = DC * DC;
    const int total = nrows * ncols;
    if (tab < 0) { for (int idx = t; idx < total; idx += nt) { const int i = idx / ncols, j = idx - i * ncols; dst[i * ld + j] = 0.0; } return; }
    const int* __restrict__ tb = tabs + tab;
    for (int base = 0; base < total; base += B * nt) {
        int e[B], o[B];
#pragma unroll
        for (int u = 0; u < B; u++) {
            const int idx = min(base + u * nt + t, total - 1), i = idx / ncols, j = idx - i * ncols, a = i / DC, uu = i - a * DC, b = j / DC, v = j - b * DC;
            e[u] = tb[a * ncb + b];
            o[u] = (uu * DC + v) | ((v * DC + uu) << 8) | ((uu == v ? 1 : 0) << 16);
        }
        double val[B];
#pragma unroll
        for (int u = 0; u < B; u++) { const int slot = e[u] >= 0 ? (e[u] & 0x3fffffff) : 0; val[u] = S_val[(size_t)slot * BB + ((e[u] & 0x40000000) ? ((o[u] >> 8) & 255) : (o[u] & 255))]; }
#pragma unroll
        for (int u = 0; u < B; u++) {
            const int idx = base + u * nt + t;
            if (idx < total) { const int i = idx / ncols, j = idx - i * ncols; dst[i * ld + j] = e[u] == -1 ? 0.0 : (e[u] == -2 ? ((o[u] >> 16) ? 1.0 : 0.0) : val[u]); }
        }
    }
}
// right-hand-side rows of a node: dst[i * 2 + t] = (rhs | Sfc)[camera(i) * DC + component(i)], zero for padding cameras or when the node's data are not this half's
template <int DC, int NR>
__device__ __forceinline__ void sn_gather_rhs(double* dst, const int* __restrict__ cams, int nrows, bool load, const double* __restrict__ rhs, const double* __restrict__ Sfc, int t, int nt) {
    for (int idx = t; idx < nrows * NR; idx += nt) {
        const int i = idx / NR, r = idx - i * NR, a = i / DC, u = i - a * DC;
        const int cam = cams[a];
        dst[i * 2 + r] = (load && cam >= 0) ? (r == 0 ? rhs : Sfc)[(size_t)cam * DC + u] : 0.0;
    }
}

// The tall panel of one supernode on ONE wave.  Lane l < 30 enters with row l of the pivot's diagonal block in row[], lane 32 + l with row l of the coupling block
// (next supernode x pivot); the other four lanes with zeros.  Column c of the factor leaves through LDS: Lc[c * SN_CS + lane] = L(lane's row, c) (the diagonal entry
// is L_cc itself), which is at once the panel's only output and its own broadcast channel: the pivot entry and the two rows that become pivots next take column c
// through v_readlane (they sit on the dependent chain), every other row reads L(c2, c) back from LDS with a wave-uniform address -- a broadcast that costs the
// vector pipe nothing -- and applies it one column LATER, when the read has long returned.  Right-hand sides and the rows of T are not here: the second wave
// takes them from Lc one step behind (sn_trows).  Returns false when a pivot was not positive (or not a number).
__device__ __forceinline__ bool sn_panel(double (&row)[SNQ], double* __restrict__ Lc, const int lane) {
    double dacc = 1.0, lp = 0.0, llast = 0.0;
    double2 sb[2][SNQ / 2];                                                  // column c's entries c2 = 2 q, 2 q + 1 as read back from LDS (128-bit broadcasts)
#pragma unroll
    for (int c = 0; c < SNQ; c++) {
        const double d = lane_bcast(row[c], c);
        dacc = fmin(dacc, d);
        const double rs = fast_rsqrt3(d);
        const double l = row[c] * rs;
        Lc[c * SN_CS + lane] = l;
        if (c + 1 < SNQ) row[c + 1] = fma(-l, lane_bcast(l, c + 1), row[c + 1]);
        if (c + 2 < SNQ) row[c + 2] = fma(-l, lane_bcast(l, c + 2), row[c + 2]);
        const double2* __restrict__ col2 = reinterpret_cast<const double2*>(Lc + c * SN_CS);
#pragma unroll
        for (int q = (c + 3) / 2; q < SNQ / 2; q++) sb[c & 1][q] = col2[q];                         // wave-uniform address: broadcast
        if (c >= 1) {
#pragma unroll
            for (int c2 = c + 2; c2 < SNQ; c2++) { const double2 v = sb[(c - 1) & 1][c2 / 2]; row[c2] = fma(-lp, (c2 & 1) ? v.y : v.x, row[c2]); }   // the previous column's share of the rows behind the look-ahead
        }
        lp = l;
        if (c == SNQ - 1) llast = lane_bcast(l, SNQ - 1);
        __builtin_amdgcn_sched_barrier(0);                                                          // (a column is a scheduling region: see the note in DESIGN.md 4c on what the scheduler does otherwise)
    }
    return dacc > 0.0 && llast == llast;
}

// The rows of T and every right-hand side of one elimination step, on the second wave, from the finished panel in LDS (one step behind the first wave):
//   rowT[j]  in: row `lane` of the coupling block (T x pivot)            out: L(T row, j)
//   gr[t]    in: right-hand side t of the panel row this lane stands for (lanes 0..29 the pivot's, 32..61 the next supernode's)
//            out: lanes 0..29 UNSCALED y (the caller multiplies by 1 / L_cc), lanes 32..61 the next supernode's updated right-hand side
//   gT[t]    in / out: right-hand side t of T's row `lane`
// L(c2, c) for the T rows comes from LDS at wave-uniform addresses (broadcasts); only y_c crosses lanes (one v_readlane pair per column and right-hand side).
template <int NR, bool TR>
__device__ __forceinline__ void sn_trows(double (&rowT)[SNQ], double (&gT)[NR], double (&gr)[NR], const double* __restrict__ Lc, const int lane_in) {
    int lane = lane_in;
    double2 sb[2][SNQ / 2];
    double lTp = 0.0;
    // software-pipelined by one column, like the panel: region c issues the 128-bit broadcast reads of column c and applies column c - 1 (whose reads returned long ago);
    // with read and use in one region the compiler waits for every read on the spot (ISA of the first build: 45 cycles per entry)
#pragma unroll
    for (int c = 0; c < SNQ; c++) {
        asm volatile("" : "+v"(lane));                                                              // the (lane > c) masks are compared on the spot, not hoisted and spilled
        const double2* __restrict__ col2 = reinterpret_cast<const double2*>(Lc + c * SN_CS);
        if (TR) {
#pragma unroll
            for (int q = (c + 1) / 2; q < SNQ / 2; q++) sb[c & 1][q] = col2[q];
        }
        const double lcc = Lc[c * SN_CS + c];
        const double lown = Lc[c * SN_CS + lane_in];
        if (TR && c >= 1) {
#pragma unroll
            for (int c2 = c; c2 < SNQ; c2++) { const double2 v = sb[(c - 1) & 1][c2 / 2]; rowT[c2] = fma(-lTp, (c2 & 1) ? v.y : v.x, rowT[c2]); }
        }
        const double rs = fast_rcp(lcc);
        const double lm = (lane > c) ? lown : 0.0;
        double lT = 0.0;
        if (TR) { lT = rowT[c] * rs; rowT[c] = lT; }
#pragma unroll
        for (int t = 0; t < NR; t++) {
            const double yc = lane_bcast(gr[t], c) * rs;
            gr[t] = fma(-lm, yc, gr[t]);
            if (TR) gT[t] = fma(-lT, yc, gT[t]);
        }
        lTp = lT;
        __builtin_amdgcn_sched_barrier(0);
    }
}

// One 16x16 tile on the matrix cores:  C(i, j) (-)= sum_k X(i, k) Y(j, k),  k < SNQ; X(i, k) = X[i * xi + k * xk], rows below nx; Y likewise.  ZERO: C = -sum (no read of C).
template <bool ZERO>
__device__ __forceinline__ void sn_tile(double* C, int ldc, const double* X, int xi, int xk, int nx, const double* Y, int yi, int yk, int ny, int ti, int tj, int lane) {
    const int li = lane & 15, lk = lane >> 4;
    const int i = 16 * ti + li, j = 16 * tj + li;
    const bool vi = i < nx, vj = j < ny;
    const double* Xi = X + (size_t)(vi ? i : 0) * xi; const double* Yj = Y + (size_t)(vj ? j : 0) * yi;
    double a[8], b[8];
#pragma unroll
    for (int ks = 0; ks < 8; ks++) { const int k = min(4 * ks + lk, SNQ - 1); a[ks] = Xi[k * xk]; b[ks] = Yj[k * yk]; }
    sn_v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < 8; ks++) {
        const bool vk = 4 * ks + lk < SNQ;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64((vi && vk) ? a[ks] : 0.0, (vj && vk) ? b[ks] : 0.0, acc, 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 4; q++) { const int r = 16 * ti + lk + 4 * q, c = 16 * tj + li; if (r < nx && c < ny) { if (ZERO) C[(size_t)r * ldc + c] = -acc[q]; else C[(size_t)r * ldc + c] -= acc[q]; } }
}
// lower triangle (tiles) of C -= X X^T, rows below n; the tiles t0, t0 + 1, ... are dealt to `nw` waves, this one is w; returns the next free tile number
__device__ __forceinline__ int sn_syrk(double* C, int ldc, const double* X, int xi, int xk, int n, int w, int nw, int t0, int lane) {
    const int nt = (n + 15) >> 4;
    int t = t0;
    for (int ti = 0; ti < nt; ti++) for (int tj = 0; tj <= ti; tj++, t++) if (t % nw == w) sn_tile<false>(C, ldc, X, xi, xk, n, X, xi, xk, n, ti, tj, lane);
    return t;
}
// all of C = - X Y^T (X: nx rows, Y: ny rows)
__device__ __forceinline__ int sn_gemm_neg(double* C, int ldc, const double* X, int xi, int xk, int nx, const double* Y, int yi, int yk, int ny, int w, int nw, int t0, int lane) {
    const int ntx = (nx + 15) >> 4, nty = (ny + 15) >> 4;
    int t = t0;
    for (int ti = 0; ti < ntx; ti++) for (int tj = 0; tj < nty; tj++, t++) if (t % nw == w) sn_tile<true>(C, ldc, X, xi, xk, nx, Y, yi, yk, ny, ti, tj, lane);
    return t;
}

// Back substitution of one supernode on ONE wave:  x = Ldd^-T ( y - A^T xa - B^T xb ).  Ldd(r, j) = Ldd[r * lr + j * lj] lower with L_jj on the diagonal; y [ny][2] already
// scaled (L^-1 g); A(r, j) = A[r * ar + j * aj], na rows, with xa[na][2]; B likewise (either may be empty).  Lanes 0..31 take A's sum, lanes 32..63 B's.  Exit: x[t] on lanes j < SNQ.
template <int NR>
__device__ __forceinline__ void sn_back(const double* Ldd, int lr, int lj, const double* y, int ny, const double* A, int ar, int aj, int na, const double* xa,
                                        const double* B, int br, int bj, int nb, const double* xb, double (&x)[NR], const int lane) {
    const int half = lane >> 5, jj = min(lane & 31, SNQ - 1);
    double col[SNQ];
#pragma unroll
    for (int r = 0; r < SNQ; r++) col[r] = Ldd[(size_t)r * lr + (size_t)jj * lj];
    const double rsl = fast_rcp(Ldd[(size_t)jj * lr + (size_t)jj * lj]);
    const double* Mt = half ? B : A; const int mr = half ? br : ar, mj = half ? bj : aj, nm = half ? nb : na; const double* xv = half ? xb : xa;
    double acc[NR];
#pragma unroll
    for (int t = 0; t < NR; t++) acc[t] = 0.0;
    const int nmax = max(na, nb);
    for (int r0 = 0; r0 < nmax; r0 += 6) {
        double m[6], v[6][NR];
#pragma unroll
        for (int u = 0; u < 6; u++) {
            const int r = min(r0 + u, max(nm - 1, 0));
            m[u] = nm > 0 ? Mt[(size_t)r * mr + (size_t)jj * mj] : 0.0;
#pragma unroll
            for (int t = 0; t < NR; t++) v[u][t] = nm > 0 ? xv[r * 2 + t] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 6; u++) if (r0 + u < nm) {
#pragma unroll
            for (int t = 0; t < NR; t++) acc[t] = fma(m[u], v[u][t], acc[t]);
        }
    }
    double w[NR];
#pragma unroll
    for (int t = 0; t < NR; t++) { const double other = __shfl_xor(acc[t], 32, 64); const double yv = y[min(jj, ny - 1) * 2 + t]; w[t] = ((lane & 31) < ny ? yv : 0.0) - (acc[t] + other); }
    double rsv[SNQ];
#pragma unroll
    for (int r = 0; r < SNQ; r++) rsv[r] = lane_bcast(rsl, r);
#pragma unroll
    for (int r = SNQ - 1; r >= 0; r--) {
#pragma unroll
        for (int t = 0; t < NR; t++) {
            const double xr = lane_bcast(w[t], r) * rsv[r];
            w[t] = (lane == r) ? xr : ((lane < r) ? fma(-col[r], xr, w[t]) : w[t]);
        }
    }
#pragma unroll
    for (int t = 0; t < NR; t++) x[t] = w[t];
}


// Back substitution of the pivots, spread over the four waves (a wave takes every fourth step): the factor blocks come back from global memory (L2), and one step is far
// shorter than that round trip -- so a wave loads the blocks of its NEXT step right after finishing one and has three steps of the other waves to wait for them
// (lds_barrier does not wait for vector memory).  Lane j < 32: column j of Ldd and of Lsd (and of L(T, .) rows 30.. when T has more than 30); lane 32 + j: column j of
// L(T, .) rows 0..29.
struct SnBackRegs { double col[SNQ], m[SNQ], m2[SNQ], diag, y[2]; };
__device__ __forceinline__ void sn_back_load(SnBackRegs& R, const double* __restrict__ wp, int qtm, int QT, int lane) {
    constexpr int Q = SNQ, LD = SN_LD;
    const int half = lane >> 5, jj = min(lane & 31, Q - 1);
    const double* Ldd = wp; const double* Lsd = wp + Q * LD; const double* LTP = wp + 2 * Q * LD; const double* yv = LTP + (size_t)qtm * LD;
    const double* M = half ? LTP : Lsd; const int nm = half ? min(QT, Q) : Q;
#pragma unroll
    for (int r = 0; r < Q; r++) { R.col[r] = Ldd[r * LD + jj]; R.m[r] = M[min(r, max(nm - 1, 0)) * LD + jj]; }
    if (QT > Q) {
#pragma unroll
        for (int r = 0; r < Q; r++) R.m2[r] = LTP[min(Q + r, QT - 1) * LD + jj];
    }
    R.diag = Ldd[jj * LD + jj]; R.y[0] = yv[jj * 2]; R.y[1] = yv[jj * 2 + 1];
}
template <int NR>
__device__ __forceinline__ void sn_back_step(const SnBackRegs& R, int QT, const double* __restrict__ xN, const double* __restrict__ xT, double (&x)[NR], int lane) {
    constexpr int Q = SNQ;
    const int half = lane >> 5;
    const double* xv = half ? xT : xN; const int nm = half ? min(QT, Q) : Q;
    double acc[NR];
#pragma unroll
    for (int t = 0; t < NR; t++) acc[t] = 0.0;
#pragma unroll
    for (int r = 0; r < Q; r++) {
#pragma unroll
        for (int t = 0; t < NR; t++) { const double v = xv[r * 2 + t]; acc[t] = fma(r < nm ? R.m[r] : 0.0, v, acc[t]); }
    }
    if (QT > Q) {
#pragma unroll
        for (int r = 0; r < Q; r++) {
#pragma unroll
            for (int t = 0; t < NR; t++) { const double v = xT[min(Q + r, QT - 1) * 2 + t]; acc[t] = fma((!half && Q + r < QT) ? R.m2[r] : 0.0, v, acc[t]); }
        }
    }
    const double rsl = fast_rcp(R.diag);
    double w[NR];
#pragma unroll
    for (int t = 0; t < NR; t++) { const double other = __shfl_xor(acc[t], 32, 64); w[t] = R.y[t] - (acc[t] + other); }
    double rsv[SNQ];
#pragma unroll
    for (int r = 0; r < Q; r++) rsv[r] = lane_bcast(rsl, r);
#pragma unroll
    for (int r = Q - 1; r >= 0; r--) {
#pragma unroll
        for (int t = 0; t < NR; t++) {
            const double xr = lane_bcast(w[t], r) * rsv[r];
            w[t] = (lane == r) ? xr : ((lane < r) ? fma(-R.col[r], xr, w[t]) : w[t]);
        }
    }
#pragma unroll
    for (int t = 0; t < NR; t++) x[t] = w[t];
}

// solution rows of a node -> Y (the layout k_arrow_update reads: Y[t * ystride + pos[camera] * DC + component])
template <int DC, int NR>
__device__ __forceinline__ void sn_store_y(double* __restrict__ Y, size_t ystride, const int* __restrict__ cams, const int* __restrict__ pos, int row0, int nrows, const double (&x)[NR], int lane) {
    if (lane < nrows) {
        const int i = row0 + lane, a = i / DC, u = i - a * DC, cam = cams[a];
        if (cam >= 0) {
#pragma unroll
            for (int t = 0; t < NR; t++) Y[(size_t)t * ystride + (size_t)pos[cam] * DC + u] = x[t];
        }
    }
}

// One workgroup per half (four waves), one loop over STAGES so that the panel and the T-row code exist once (the first build had four copies of each: 170 KB of
// code, and the copy behind the exchange spilled 570 registers to scratch):
//   stages 0 .. ns-1   pivot k: wave 0 factors its panel from D / C in LDS into Lc[k & 1] WHILE wave 1 does the rows of T and every right-hand side of pivot k - 1
//                      from Lc[(k - 1) & 1] and waves 2, 3 bring the next supernode's original blocks from S into LDS and send the previous steps' factor blocks to
//                      global memory (the back substitution reads them again); then all four, behind a barrier: D(next) -= Lsd Lsd^T (pivot k),
//                      E(T, P_k) = - L(T, P_k-1) Lsd_k-1^T and T x T -= L(T, P_k-1) L(T, P_k-1)^T (pivot k - 1) on the matrix cores
//   stage ns           drains wave 1 (pivot ns - 1); then the exchange with the partner half
//   stage ns+1         M:   panel, barrier, T rows + right-hand side of the SAME stage, T x T -= L(T, M) L(T, M)^T
//   stage ns+2 (rings) T_1 = the first 30 rows of T, T_2 (<= 30 more) rides as the coupling block; T_2 x T_2 -= L_21 L_21^T
//   stage ns+3 (T_2)   the rest of T
template <int DC, int NR, bool RING>
__global__ void __launch_bounds__(SN_THREADS)
k_snode_solve(const double* __restrict__ S_val, const double* __restrict__ rhs, const double* __restrict__ Sfc,
              const int* __restrict__ half_rec, const int* __restrict__ step_rec, const int* __restrict__ node_cam, const int* __restrict__ tabs,
              const int* __restrict__ pos, double* __restrict__ work, double* __restrict__ xchg, int* __restrict__ flags, int seq, int qtm,
              double* __restrict__ Y, size_t ystride, int* __restrict__ fail_flag, long long* __restrict__ stamps = nullptr) {
    constexpr int S = SNQ / DC, CAPT = 2 * S, Q = SNQ, LD = SN_LD, CS = SN_CS;
#define SN_STAMP(i_) do { if (stamps && tid == 0) stamps[(size_t)blockIdx.x * 64 + (i_)] = (long long)wall_clock64(); } while (0)
#define SN_WSTAMP(i_) do { if (stamps && lane == 0 && st == 2) stamps[(size_t)blockIdx.x * 64 + 8 + wave * 8 + (i_)] = (long long)wall_clock64(); } while (0)
#define SN_TSTAMP(i_) do { if (stamps && lane == 0 && wave <= 1 && st == ns + 1) stamps[(size_t)blockIdx.x * 64 + 40 + wave * 8 + (i_)] = (long long)wall_clock64(); } while (0)
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int LDT = qtm + 1;
    double* Dbuf = lds;                                   // [2][Q][LD]   diagonal block of the pivot (cur) / of the next supernode (next: original, then updated)
    double* Cbuf = Dbuf + 2 * Q * LD;                     // [2][Q][LD]   coupling (next x pivot)
    double* Ebuf = Cbuf + 2 * Q * LD;                     // [2][qtm][LD] coupling (T x pivot of stage k) in Ebuf[k & 1]
    double* Lcb = Ebuf + 2 * qtm * LD;                    // [2][Q][CS]   the panel's columns
    double* LTP = Lcb + 2 * Q * CS;                       // [qtm][LD]    L(T, pivot) of the stage wave 1 finished last
    double* ATT = LTP + qtm * LD;                         // [qtm][LDT]   T x T, accumulated over the half
    double* gAll = ATT + qtm * LDT;                       // [MAXSTEPS + 2][Q][2]  right-hand sides of the pivots and of M (updated in place as the elimination passes)
    double* yP = gAll + (SN_MAXSTEPS + 2) * 2 * Q;        // [Q][2]       y of the stage wave 1 finished last
    double* xN = yP + 2 * Q;                              // [Q][2]       solution of the supernode behind (back substitution)
    double* yM = xN + 2 * Q;                              // [Q][2]
    double* gTb = yM + 2 * Q;                             // [qtm][2]     right-hand side of T
    double* yT = gTb + 2 * qtm;                           // [qtm][2]
    double* xT = yT + 2 * qtm;                            // [qtm][2]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int* hr = half_rec + (size_t)blockIdx.x * SN_HREC;
    const int ns = hr[0], partner = hr[2], owner = hr[3], tnode = RING ? hr[4] : -1, QT = RING ? hr[5] : 0;
    const int* sr0 = step_rec + (size_t)hr[1] * SN_SREC;
    double* wk = work + (size_t)hr[6];
    const size_t wstep = (size_t)2 * Q * LD + (size_t)qtm * LD + 2 * Q;
    const int* tcams = node_cam + (size_t)max(tnode, 0) * CAPT;
    const int sl = lane - 32;
    const bool dl = lane < Q;
    const int QT2 = RING ? max(QT - Q, 0) : 0;
    const bool hasT = RING && QT > 0;
    SN_STAMP(0);
    // ---- prologue: the first pivot (or M when the half has none), every right-hand side, T's own block
    {
        // one matrix per wave, all four at once: two memory round trips for the lot when a matrix fits one batch of 16 per lane (the 30 x 30 ones do)
        if (wave == 0) sn_gather_b<DC, 16>(Dbuf, LD, Q, Q, S_val, tabs, hr[9], S, lane, 64);
        else if (wave == 1) { if (ns > 0) sn_gather_b<DC, 16>(Cbuf, LD, Q, Q, S_val, tabs, sr0[2], S, lane, 64); }
        else if (wave == 2) { if (RING) sn_gather_b<DC, 16>(Ebuf, LD, QT, Q, S_val, tabs, hr[10], S, lane, 64); }
        else if (RING) sn_gather_b<DC, 16>(ATT, LDT, QT, QT, S_val, tabs, hr[11], CAPT, lane, 64);
    }
    sn_gather_rhs<DC, 2>(gAll, node_cam + (size_t)hr[13] * CAPT, Q, ns > 0 || owner, rhs, Sfc, tid, SN_THREADS);
    for (int k = 0; k < ns; k++) sn_gather_rhs<DC, 2>(gAll + (size_t)(k + 1) * 2 * Q, node_cam + (size_t)sr0[k * SN_SREC + 1] * CAPT, Q, sr0[k * SN_SREC + 5] != 0, rhs, Sfc, tid, SN_THREADS);
    if (RING) sn_gather_rhs<DC, 2>(gTb, tcams, QT, owner != 0, rhs, Sfc, tid, SN_THREADS);
    for (int e = tid; e < 2 * qtm; e += SN_THREADS) { xT[e] = 0.0; yT[e] = 0.0; }      // (read under a mask by halves without a T: must be numbers)
    for (int e = tid; e < 2 * Q; e += SN_THREADS) { xN[e] = 0.0; yM[e] = 0.0; yP[e] = 0.0; }
    __syncthreads();
    SN_STAMP(1);
    int cur = 0;
    bool ok = true;
    double row[Q], gT[NR], gr[NR];                                         // row: the panel's row on wave 0, T's row on wave 1 (one array: a wave is one or the other)
#pragma unroll
    for (int t = 0; t < NR; t++) { gT[t] = (wave == 1 && hasT && lane < QT) ? gTb[lane * 2 + t] : 0.0; gr[t] = 0.0; }
    double* LcT2 = Dbuf;                                                   // the panel of T_2: D's two buffers and part of C's (Q x CS doubles), free by then
    const int nstages = ns + 2 + (hasT ? 1 : 0) + (QT2 > 0 ? 1 : 0);
    for (int st = (ns > 0 ? 0 : 1); st < nstages; st++) {
        const int kind = st < ns ? 0 : st - ns + 1;                        // 0 pivot, 1 drain, 2 M, 3 T_1, 4 T_2
        const int nxt = cur ^ 1;
        double* Dc = Dbuf + cur * Q * LD; double* Dn = Dbuf + nxt * Q * LD;
        double* Cc = Cbuf + cur * Q * LD; double* Cn = Cbuf + nxt * Q * LD;
        double* Lk = kind == 4 ? LcT2 : Lcb + (st & 1) * Q * CS;           // this stage's panel columns
        double* Lp = kind <= 1 ? Lcb + ((st + 1) & 1) * Q * CS : Lk;       // the panel wave 1 works on: the previous pivot's while the pivots run, this stage's own behind the exchange
        double* EM = Ebuf + (ns & 1) * qtm * LD;                           // E(T, M), then L(T, M) in place
        double* gM = gAll + (size_t)ns * 2 * Q;
        // ---- the exchange with the partner half sits in front of M: both end up with the same sums
        if (kind == 2) {
            SN_STAMP(2);
            if (ns > 0) {                                                    // what is still in LDS goes to global memory
                double* wp = wk + (size_t)(ns - 1) * wstep;
                if (RING) for (int e = tid; e < QT * LD; e += SN_THREADS) wp[2 * Q * LD + e] = LTP[e];
                for (int e = tid; e < 2 * Q; e += SN_THREADS) wp[2 * Q * LD + (size_t)qtm * LD + e] = yP[e];
            }
            if (hasT && wave == 1 && lane < QT) {                           // T's right-hand side leaves wave 1's registers
#pragma unroll
                for (int t = 0; t < NR; t++) gTb[lane * 2 + t] = gT[t];
                if (NR == 1) gTb[lane * 2 + 1] = 0.0;
            }
            __syncthreads();
            if (partner >= 0) {
                double* mine = xchg + (size_t)hr[7]; const double* other = xchg + (size_t)hr[14];
                const int nD = Q * LD, nE = qtm * LD, nTT = qtm * LDT;
                for (int e = tid; e < nD; e += SN_THREADS) mine[e] = Dc[e];
                for (int e = tid; e < 2 * Q; e += SN_THREADS) mine[nD + e] = gM[e];
                if (RING) {
                    for (int e = tid; e < nE; e += SN_THREADS) mine[nD + 2 * Q + e] = EM[e];
                    for (int e = tid; e < nTT; e += SN_THREADS) mine[nD + 2 * Q + nE + e] = ATT[e];
                    for (int e = tid; e < 2 * qtm; e += SN_THREADS) mine[nD + 2 * Q + nE + nTT + e] = gTb[e];
                }
                __threadfence(); __syncthreads();
                if (tid == 0) {
                    __hip_atomic_store(flags + hr[8], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                    while (__hip_atomic_load(flags + hr[15], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != seq) __builtin_amdgcn_s_sleep(2);
                }
                __syncthreads(); __threadfence();
                for (int e = tid; e < nD; e += SN_THREADS) Dc[e] += other[e];
                for (int e = tid; e < 2 * Q; e += SN_THREADS) gM[e] += other[nD + e];
                if (RING) {
                    for (int e = tid; e < nE; e += SN_THREADS) EM[e] += other[nD + 2 * Q + e];
                    for (int e = tid; e < nTT; e += SN_THREADS) ATT[e] += other[nD + 2 * Q + nE + e];
                    for (int e = tid; e < 2 * qtm; e += SN_THREADS) gTb[e] += other[nD + 2 * Q + nE + nTT + e];
                }
                __syncthreads();
            }
            if (wave == 1 && hasT && lane < QT) {
#pragma unroll
                for (int t = 0; t < NR; t++) gT[t] = gTb[lane * 2 + t];
            }
            SN_STAMP(3);
        }
        if (kind == 4) {                                                     // T_2's diagonal block, identity-padded to 30 rows, into C's place (LcT2 takes D's and the front of C's second half only later)
            double* P2 = Cbuf + Q * LD;
            for (int e = tid; e < Q * Q; e += SN_THREADS) { const int i = e / Q, j = e - i * Q; P2[i * LD + j] = (i < QT2 && j < QT2) ? ATT[(size_t)(Q + i) * LDT + Q + j] : (i == j ? 1.0 : 0.0); }
            __syncthreads();
        }
        SN_WSTAMP(0); SN_TSTAMP(0);
        // ---- wave 0: the panel of this stage
        if (wave == 0 && kind != 1) {
            const double* pD = kind == 0 || kind == 2 ? Dc : (kind == 3 ? ATT : Cbuf + Q * LD); const int ldD = kind == 3 ? LDT : LD;
            const double* pC = kind == 0 ? Cc : ATT + (size_t)Q * LDT; const int ldC = kind == 0 ? LD : LDT, nsub = kind == 0 ? Q : (kind == 3 ? QT2 : 0);
            const bool sub = sl >= 0 && sl < nsub;
            const double* src = dl ? pD + (size_t)lane * ldD : pC + (size_t)(sub ? sl : 0) * ldC;
#pragma unroll
            for (int j = 0; j < Q; j++) { const double v = src[j]; row[j] = (dl || sub) ? v : 0.0; }
            ok = sn_panel(row, Lk, lane) && ok;
        }
        SN_TSTAMP(1);
        if (kind >= 2) __syncthreads();                                      // behind the exchange wave 1 works on THIS stage's panel
        SN_TSTAMP(2);
        // ---- wave 1: rows of T and right-hand sides (of the previous pivot while the pivots run)
        const bool w1 = wave == 1 && (kind >= 2 || st >= 1);
        const bool tr = hasT && kind <= 2;
        if (w1) {
            const double* pE = kind <= 1 ? Ebuf + ((st - 1) & 1) * qtm * LD : EM;
            if (tr) {
                const double* srcT = pE + (size_t)(lane < QT ? lane : 0) * LD;
#pragma unroll
                for (int j = 0; j < Q; j++) { const double v = srcT[j]; row[j] = lane < QT ? v : 0.0; }
            }
            // right-hand sides: lanes 0..29 the stage's pivot rows, lanes 32.. the rows that ride below them
            const double* gP = kind <= 1 ? gAll + (size_t)(st - 1) * 2 * Q : (kind == 2 ? gM : (kind == 3 ? gTb : gTb + 2 * Q));
            const double* gN = kind <= 1 ? gAll + (size_t)st * 2 * Q : gTb + 2 * Q;
            const int nd = kind == 4 ? QT2 : Q, nsubg = kind <= 1 ? Q : (kind == 3 ? QT2 : 0);
            const bool dg = lane < nd, sg = sl >= 0 && sl < nsubg;
            const double* gsrc = dg ? gP + lane * 2 : gN + (sg ? sl : 0) * 2;
#pragma unroll
            for (int t = 0; t < NR; t++) { const double v = gsrc[t]; gr[t] = (dg || sg) ? v : 0.0; }
            if (RING && tr) sn_trows<NR, RING>(row, gT, gr, Lp, lane); else sn_trows<NR, false>(row, gT, gr, Lp, lane);      // (a run-time flag inside the column loop cost 45 %)
        }
        SN_TSTAMP(3);
        // ---- waves 2, 3 while the pivots run: the next supernode's original blocks; the previous stages' factor blocks to global memory
        if (wave >= 2 && kind <= 1) {
            const int ht = tid - 128, hn = SN_THREADS - 128;
            // order matters: vector memory returns in order, so the table loads go first, the stores to global memory fill the time of the two round trips
            constexpr int PER = 16, NE = Q * Q, BB = DC * DC;
            int ge[PER], go[PER]; double gv[PER];
            const int* sr = sr0 + (size_t)min(st, max(ns - 1, 0)) * SN_SREC;
            const int tabA = kind == 0 ? sr[3] : -3, tabB = (kind == 0 && st + 1 < ns) ? sr[SN_SREC + 2] : -3;      // D(N_st), C(N_st+1, N_st)
            if (kind == 0) {
#pragma unroll
                for (int u = 0; u < PER; u++) {
                    const int g = min(u * hn + ht, 2 * NE - 1), which = g >= NE ? 1 : 0, idx = g - which * NE;
                    const int i = idx / Q, j = idx - i * Q, a = i / DC, uu = i - a * DC, b = j / DC, v = j - b * DC;
                    const int tab = which ? tabB : tabA;
                    ge[u] = tab >= 0 ? tabs[tab + a * S + b] : -1;
                    go[u] = (uu * DC + v) | ((v * DC + uu) << 8) | ((uu == v ? 1 : 0) << 16);
                }
            }
            if (st >= 2) {                                                   // L(T, P_st-2) and y_st-2
                double* wp = wk + (size_t)(st - 2) * wstep;
                if (RING) for (int e = ht; e < QT * LD; e += hn) wp[2 * Q * LD + e] = LTP[e];
                for (int e = ht; e < 2 * Q; e += hn) wp[2 * Q * LD + (size_t)qtm * LD + e] = yP[e];
            }
            if (kind == 0) {
#pragma unroll
                for (int u = 0; u < PER; u++) { const int slot = ge[u] >= 0 ? (ge[u] & 0x3fffffff) : 0; gv[u] = S_val[(size_t)slot * BB + ((ge[u] & 0x40000000) ? ((go[u] >> 8) & 255) : (go[u] & 255))]; }
            }
            if (st >= 1) {                                                   // the previous panel, row-major: Ldd | Lsd
                double* wp = wk + (size_t)(st - 1) * wstep;
                for (int e = ht; e < Q * Q; e += hn) { const int r = e / Q, j = e - r * Q; wp[r * LD + j] = Lp[j * CS + r]; wp[Q * LD + r * LD + j] = Lp[j * CS + 32 + r]; }
            }
            if (kind == 0) {
#pragma unroll
                for (int u = 0; u < PER; u++) {
                    const int g = u * hn + ht;
                    if (g < 2 * NE) {
                        const int which = g >= NE ? 1 : 0, idx = g - which * NE, i = idx / Q, j = idx - i * Q;
                        if ((which ? tabB : tabA) != -3) (which ? Cn : Dn)[i * LD + j] = ge[u] == -1 ? 0.0 : (ge[u] == -2 ? ((go[u] >> 16) ? 1.0 : 0.0) : gv[u]);
                    }
                }
            }
        }
        SN_WSTAMP(1);
        lds_barrier();                                                       // B0: LTP / yP / the right-hand sides may be overwritten
        if (w1) {
            double* pLT = kind <= 1 ? LTP : EM;
            if (tr && lane < QT) {
#pragma unroll
                for (int j = 0; j < Q; j++) pLT[lane * LD + j] = row[j];
            }
            double* py = kind <= 1 ? yP : (kind == 2 ? yM : (kind == 3 ? yT : yT + 2 * Q));
            double* pgN = kind <= 1 ? gAll + (size_t)st * 2 * Q : gTb + 2 * Q;
            const int nd = kind == 4 ? QT2 : Q, nsubg = kind <= 1 ? Q : (kind == 3 ? QT2 : 0);
            if (lane < nd) {
                const double rsl = fast_rcp(Lp[lane * CS + lane]);
#pragma unroll
                for (int t = 0; t < NR; t++) py[lane * 2 + t] = gr[t] * rsl;
            } else if (sl >= 0 && sl < nsubg) {
#pragma unroll
                for (int t = 0; t < NR; t++) pgN[sl * 2 + t] = gr[t];
            }
            if (kind == 2 && hasT && lane < QT) {                           // T's right-hand side behind M: the T stages read it from LDS
#pragma unroll
                for (int t = 0; t < NR; t++) gTb[lane * 2 + t] = gT[t];
            }
        }
        lds_barrier();                                                       // B1
        SN_WSTAMP(2); SN_TSTAMP(4);
        // ---- Schur updates on the matrix cores
        int t0 = 0;
        if (kind == 0) t0 = sn_syrk(Dn, LD, Lk + 32, 1, CS, Q, wave, 4, t0, lane);                                    // D(N_k) -= Lsd_k Lsd_k^T
        if (hasT && kind <= 1 && st >= 1) {
            t0 = sn_gemm_neg(Ebuf + (st & 1) * qtm * LD, LD, LTP, LD, 1, QT, Lp + 32, 1, CS, Q, wave, 4, t0, lane);   // E(T, P_st) = - L(T, P_st-1) Lsd_st-1^T
            t0 = sn_syrk(ATT, LDT, LTP, LD, 1, QT, wave, 4, t0, lane);                                                 // T x T -= L(T, P_st-1) L(T, P_st-1)^T
        }
        if (hasT && kind == 2) t0 = sn_syrk(ATT, LDT, EM, LD, 1, QT, wave, 4, t0, lane);                               // T x T -= L(T, M) L(T, M)^T
        if (kind == 3 && QT2 > 0) t0 = sn_syrk(ATT + (size_t)Q * LDT + Q, LDT, Lk + 32, 1, CS, QT2, wave, 4, t0, lane);   // T_2 x T_2 -= L_21 L_21^T
        SN_WSTAMP(3);
        lds_barrier();                                                       // B2
        SN_WSTAMP(4); SN_TSTAMP(5);
        if (kind == 0) cur = nxt;
    }
    if (wave == 0 && lane == 0 && !ok) *fail_flag = 1;
    __syncthreads();
    SN_STAMP(4);
    double* LcM = Lcb + ((ns + 1) & 1) * Q * CS; double* LcT1 = Lcb + (ns & 1) * Q * CS;
    double* EM = Ebuf + (ns & 1) * qtm * LD;
    // ---- back substitution: T_2, T_1, M on wave 0 (from LDS); then this half's pivots from the last to the first, a wave per step in turn (sn_back_load / sn_back_step)
    const int bpos = (wave + 3) & 3;                        // wave 1 takes the first pivot step, 2 the second, 3 the third, 0 the fourth, ...
    SnBackRegs BR;
    if (wave != 0 && bpos < ns) sn_back_load(BR, wk + (size_t)(ns - 1 - bpos) * wstep, qtm, QT, lane);
    if (wave == 0) {
        double x[NR];
        const int* mcams = node_cam + (size_t)hr[12] * CAPT;
        if (hasT) {
            if (QT2 > 0) {
                sn_back<NR>(LcT2, 1, CS, yT + 2 * Q, QT2, nullptr, 0, 0, 0, nullptr, nullptr, 0, 0, 0, nullptr, x, lane);
                if (lane < QT2) {
#pragma unroll
                    for (int t = 0; t < NR; t++) xT[(Q + lane) * 2 + t] = x[t];
                }
                if (owner) sn_store_y<DC, NR>(Y, ystride, tcams, pos, Q, QT2, x, lane);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
            sn_back<NR>(LcT1, 1, CS, yT, Q, LcT1 + 32, 1, CS, QT2, xT + 2 * Q, nullptr, 0, 0, 0, nullptr, x, lane);
            if (lane < Q) {
#pragma unroll
                for (int t = 0; t < NR; t++) xT[lane * 2 + t] = x[t];
            }
            if (owner) sn_store_y<DC, NR>(Y, ystride, tcams, pos, 0, Q, x, lane);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        sn_back<NR>(LcM, 1, CS, yM, Q, nullptr, 0, 0, 0, nullptr, EM, LD, 1, QT, xT, x, lane);
        if (lane < Q) {
#pragma unroll
            for (int t = 0; t < NR; t++) xN[lane * 2 + t] = x[t];
        }
        if (owner) sn_store_y<DC, NR>(Y, ystride, mcams, pos, 0, Q, x, lane);
        if (bpos < ns) sn_back_load(BR, wk + (size_t)(ns - 1 - bpos) * wstep, qtm, QT, lane);
    }
    lds_barrier();                                                           // x_M and x_T are in LDS
    SN_STAMP(6);
    {
        int done = 0;
        for (int i = bpos; i < ns; i += 4) {
            while (done < i) { lds_barrier(); done++; }                      // the other waves' steps
            double x[NR];
            sn_back_step<NR>(BR, QT, xN, xT, x, lane);
            if (lane < Q) {
#pragma unroll
                for (int t = 0; t < NR; t++) xN[lane * 2 + t] = x[t];
            }
            sn_store_y<DC, NR>(Y, ystride, node_cam + (size_t)sr0[(ns - 1 - i) * SN_SREC] * CAPT, pos, 0, Q, x, lane);
            sn_back_load(BR, wk + (size_t)max(ns - 1 - (i + 4), 0) * wstep, qtm, QT, lane);      // this wave's next step (a harmless reload of step 0 when there is none)
            lds_barrier(); done++;
        }
        while (done < ns) { lds_barrier(); done++; }
    }
    SN_STAMP(5);
#undef SN_STAMP
#undef SN_WSTAMP
#undef SN_TSTAMP
}

}  // namespace ssfm
