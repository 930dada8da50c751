// spherical_sfm_amd -- supernodal solver of the reduced camera system for rings and chains of cameras (round 6).
//
// What it replaces: the direct solve of the reduced system that Ceres' SPARSE_SCHUR does with a sparse Cholesky (reference src/sfm.cpp:205,273), on the
// structures the reference's drivers produce for a closed camera loop (examples/spherical_sfm_tools.cpp:887-950: a track couples the cameras it spans, so
// the reduced matrix of a loop is a PERIODIC block band of half-width r = longest track - 1).
//
// Why a new kernel.  The windowed block-band factorisation (band_kernels2.h) is a chain of one dependent step per CAMERA -- 1.27 us each at
// BASELINE config 2, 42 + 15 + 5 + 16 us per LM iteration in four launches, 45 % of the iteration on <= 12 of 256 compute units.  The arithmetic is nothing
// (6.5 Mflop); what costs is the number of dependent steps and what one step has to wait for (two barriers, an LDS round trip of the panel, the 6x6
// factor-and-invert on one wave).  Here a step eliminates a SUPERNODE of s = 30 / DC cameras (30 scalar columns) and the whole tall panel
//      [ diagonal block (30 rows) ; coupling to the next supernode (30 rows) ; coupling to the ring's closing separator T (<= 60 rows) ]
// is factored by ONE wave with one lane per row, right-looking, the pivot row broadcast by v_readlane: no LDS, no barrier inside the 30 columns, and the
// right-hand sides ride along as two more columns (the forward substitution costs two instructions per column).  The Schur updates of the next supernode
// and of T run on the matrix cores (v_mfma_f64_16x16x4) from LDS copies of the panel.  A ring in its own circular order with reach r <= s is block
// tridiagonal in supernodes plus one separator T that closes it; it is eliminated from both sides of T towards a middle supernode M by two workgroups
// (halves A and B, 7 steps each at config 2 instead of 33 + 10), which exchange their contributions to [M, T] ONCE through global memory, both solve the
// 60...90-row remainder redundantly (a + b == b + a: identical bits) and back-substitute their own half.  One launch does the factorisation, both substitutions
// and the scatter of the solution into the layout k_arrow_update reads.  No atomics anywhere: the result does not depend on scheduling.
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
constexpr int SN_LD = 31;          // leading dimension of the Q-column LDS / workspace matrices (odd: rows of consecutive lanes fall into different banks)
constexpr int SN_QTMAX = 60;       // rows of T: s + (n mod s) cameras <= 2 s - 1
constexpr int SN_MAXSTEPS = 16;    // pivots of one half
constexpr int SN_THREADS = 256;
constexpr int SN_HREC = 16, SN_SREC = 8;

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
    size_t lds_bytes() const {
        const size_t d = (size_t)4 * SNQ * SN_LD + (size_t)2 * qtm * SN_LD + (size_t)2 * SNQ * SN_LD + (size_t)qtm * SN_LD + (size_t)qtm * (qtm + 1)
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
    if (const char* e = std::getenv("SSFM_SNODE")) if (std::atoi(e) == 0) return false;
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

// dst[i * ld + j] (i < nrows, j < ncols) <- the matrix a block table describes (tab < 0: zeros); nt cooperating threads, this one is t
template <int DC>
__device__ __forceinline__ void sn_gather(double* dst, int ld, int nrows, int ncols, const double* __restrict__ S_val, const int* __restrict__ tabs, int tab, int ncb, int t, int nt) {
    constexpr int BB = DC * DC;
    const int total = nrows * ncols;
    if (tab < 0) { for (int idx = t; idx < total; idx += nt) { const int i = idx / ncols, j = idx - i * ncols; dst[i * ld + j] = 0.0; } return; }
    const int* __restrict__ tb = tabs + tab;
    for (int base = 0; base < total; base += 4 * nt) {
        int e[4], o[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int idx = min(base + u * nt + t, total - 1), i = idx / ncols, j = idx - i * ncols, a = i / DC, uu = i - a * DC, b = j / DC, v = j - b * DC;
            e[u] = tb[a * ncb + b];
            o[u] = (uu * DC + v) | ((v * DC + uu) << 8) | ((uu == v ? 1 : 0) << 16);
        }
        double val[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const int slot = e[u] >= 0 ? (e[u] & 0x3fffffff) : 0; val[u] = S_val[(size_t)slot * BB + ((e[u] & 0x40000000) ? ((o[u] >> 8) & 255) : (o[u] & 255))]; }
#pragma unroll
        for (int u = 0; u < 4; u++) {
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

// The tall panel.  Lane l < 30: row l of the pivot's diagonal block; lane 32 + l: row l of the coupling block (next supernode x pivot); rowT (TR): row `lane` of the
// coupling block (T x pivot).  row[SNQ + t] / rowT[SNQ + t]: right-hand side t of that row.  Exit: row[j] = L(row, j) (the diagonal entry holds 1 / L_jj),
// row[SNQ + t] = y (pivot rows) or the updated right-hand side (other rows).  Returns false when a pivot is not positive.
template <int NR, bool TR>
__device__ __forceinline__ bool sn_panel(double (&row)[SNQ + NR], double (&rowT)[SNQ + NR], const int lane_in) {
    bool ok = true;
    // Every column is its own scheduling region (sched_barrier): left alone, the compiler turns the fully unrolled right-looking loop into a LEFT-looking one to save
    // registers -- column c then starts with a chain of c dependent multiply-adds on the pivot entry (ISA of the first build: 3.5k cycles of pure latency per panel) and
    // every broadcast value is parked in a VGPR lane for later.  The lane index is made opaque per column so that the 60 (lane == c) / (lane > c) masks are compared on
    // the spot (one instruction) instead of being hoisted, spilled to VGPR lanes and read back (two v_readlane + wait states each).
    int lane = lane_in;
#pragma unroll
    for (int c = 0; c < SNQ; c++) {
        asm volatile("" : "+v"(lane));
        const double d = lane_bcast(row[c], c);
        ok = ok && (d > 0.0);
        const double rs = fast_rsqrt(d);
        const double l = row[c] * rs;
        double lT = 0.0;
        if (TR) { lT = rowT[c] * rs; rowT[c] = lT; }
        row[c] = (lane == c) ? rs : l;
#pragma unroll
        for (int c2 = c + 1; c2 < SNQ; c2++) {
            const double s = lane_bcast(l, c2);
            row[c2] = fma(-l, s, row[c2]);
            if (TR) rowT[c2] = fma(-lT, s, rowT[c2]);
        }
#pragma unroll
        for (int t = 0; t < NR; t++) {
            const double yc = lane_bcast(row[SNQ + t], c) * rs;
            const double lm = (lane > c) ? l : 0.0;
            row[SNQ + t] = (lane == c) ? yc : fma(-lm, yc, row[SNQ + t]);
            if (TR) rowT[SNQ + t] = fma(-lT, yc, rowT[SNQ + t]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    return ok;
}

// One 16x16 tile on the matrix cores:  C(i, j) -= sum_k X(i, k) Y(j, k),  k < SNQ; rows of X below nx, rows of Y below ny.
__device__ __forceinline__ void sn_tile(double* C, int ldc, const double* X, int ldx, int nx, const double* Y, int ldy, int ny, int ti, int tj, int lane) {
    const int li = lane & 15, lk = lane >> 4;
    const int i = 16 * ti + li, j = 16 * tj + li;
    const bool vi = i < nx, vj = j < ny;
    const double* Xi = X + (size_t)(vi ? i : 0) * ldx; const double* Yj = Y + (size_t)(vj ? j : 0) * ldy;
    double a[8], b[8];
#pragma unroll
    for (int ks = 0; ks < 8; ks++) { const int k = min(4 * ks + lk, SNQ - 1); a[ks] = Xi[k]; b[ks] = Yj[k]; }
    sn_v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < 8; ks++) {
        const bool vk = 4 * ks + lk < SNQ;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64((vi && vk) ? a[ks] : 0.0, (vj && vk) ? b[ks] : 0.0, acc, 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 4; q++) { const int r = 16 * ti + lk + 4 * q, c = 16 * tj + li; if (r < nx && c < ny) C[(size_t)r * ldc + c] -= acc[q]; }
}
// lower triangle (tiles) of C -= X X^T, rows below n; tiles dealt to `nw` waves, this one is w
__device__ __forceinline__ void sn_syrk(double* C, int ldc, const double* X, int ldx, int n, int w, int nw, int lane) {
    const int nt = (n + 15) >> 4;
    int t = 0;
    for (int ti = 0; ti < nt; ti++) for (int tj = 0; tj <= ti; tj++, t++) if (t % nw == w) sn_tile(C, ldc, X, ldx, n, X, ldx, n, ti, tj, lane);
}
// all of C -= X Y^T (X: nx rows, Y: ny rows)
__device__ __forceinline__ void sn_gemm(double* C, int ldc, const double* X, int ldx, int nx, const double* Y, int ldy, int ny, int w, int nw, int t0, int lane) {
    const int ntx = (nx + 15) >> 4, nty = (ny + 15) >> 4;
    int t = t0;
    for (int ti = 0; ti < ntx; ti++) for (int tj = 0; tj < nty; tj++, t++) if (t % nw == w) sn_tile(C, ldc, X, ldx, nx, Y, ldy, ny, ti, tj, lane);
}

// Back substitution of one supernode on ONE wave:  x = Ldd^-T ( y - A^T xa - B^T xb ),  Ldd: SNQ x SNQ lower with 1 / L_jj on the diagonal (leading dimension ldd),
// A: na x SNQ (lda) with xa[na][2], B: nb x SNQ (ldb) with xb[nb][2] (either may be empty).  Lanes 0..31 take A's sum, lanes 32..63 B's.  Exit: x[t] on lanes j < SNQ.
template <int NR>
__device__ __forceinline__ void sn_back(const double* Ldd, int ldd, const double* y /* [ny][2] */, int ny, const double* A, int lda, int na, const double* xa,
                                        const double* B, int ldb, int nb, const double* xb, double (&x)[NR], const int lane) {
    const int half = lane >> 5, jj = min(lane & 31, SNQ - 1);
    double col[SNQ];
#pragma unroll
    for (int r = 0; r < SNQ; r++) col[r] = Ldd[(size_t)r * ldd + jj];
    const double* Mt = half ? B : A; const int ldm = half ? ldb : lda, nm = half ? nb : na; const double* xv = half ? xb : xa;
    double acc[NR];
#pragma unroll
    for (int t = 0; t < NR; t++) acc[t] = 0.0;
    const int nmax = max(na, nb);
    for (int r0 = 0; r0 < nmax; r0 += 6) {
        double m[6], v[6][NR];
#pragma unroll
        for (int u = 0; u < 6; u++) {
            const int r = min(r0 + u, max(nm - 1, 0));
            m[u] = nm > 0 ? Mt[(size_t)r * ldm + jj] : 0.0;
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
    for (int r = 0; r < SNQ; r++) rsv[r] = lane_bcast(col[r], r);
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

// solution rows of a node -> Y (the layout k_arrow_update reads: Y[t * ystride + pos[camera] * DC + component])
template <int DC, int NR>
__device__ __forceinline__ void sn_store_y(double* __restrict__ Y, size_t ystride, const int* __restrict__ cams, const int* __restrict__ pos, int nrows, const double (&x)[NR], int lane) {
    if (lane < nrows) {
        const int a = lane / DC, u = lane - a * DC, cam = cams[a];
        if (cam >= 0) {
#pragma unroll
            for (int t = 0; t < NR; t++) Y[(size_t)t * ystride + (size_t)pos[cam] * DC + u] = x[t];
        }
    }
}

template <int DC, int NR, bool RING>
__global__ void __launch_bounds__(SN_THREADS)
k_snode_solve(const double* __restrict__ S_val, const double* __restrict__ rhs, const double* __restrict__ Sfc,
              const int* __restrict__ half_rec, const int* __restrict__ step_rec, const int* __restrict__ node_cam, const int* __restrict__ tabs,
              const int* __restrict__ pos, double* __restrict__ work, double* __restrict__ xchg, int* __restrict__ flags, int seq, int qtm,
              double* __restrict__ Y, size_t ystride, int* __restrict__ fail_flag) {
    constexpr int S = SNQ / DC, CAPT = 2 * S, Q = SNQ, LD = SN_LD;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int LDT = qtm + 1;
    double* Dbuf = lds;                                   // [2][Q][LD]   diagonal block of the pivot (cur) / of the next supernode (next: original, then updated)
    double* Cbuf = Dbuf + 2 * Q * LD;                     // [2][Q][LD]   coupling (next x pivot)
    double* Ebuf = Cbuf + 2 * Q * LD;                     // [2][qtm][LD] coupling (T x pivot)
    double* Ldd = Ebuf + 2 * qtm * LD;                    // [Q][LD]      panel outputs of the step
    double* Lsd = Ldd + Q * LD;                           // [Q][LD]
    double* LTP = Lsd + Q * LD;                           // [qtm][LD]
    double* ATT = LTP + qtm * LD;                         // [qtm][LDT]   T x T, accumulated over the half
    double* gAll = ATT + qtm * LDT;                       // [MAXSTEPS + 2][Q][2]  right-hand sides of the pivots and of M (updated in place as the elimination passes)
    double* yP = gAll + (SN_MAXSTEPS + 2) * 2 * Q;        // [Q][2]       y of the step
    double* xN = yP + 2 * Q;                              // [Q][2]       solution of the supernode behind (back substitution)
    double* yM = xN + 2 * Q;                              // [Q][2]
    double* gT = yM + 2 * Q;                              // [qtm][2]     right-hand side of T
    double* yT = gT + 2 * qtm;                            // [qtm][2]
    double* xT = yT + 2 * qtm;                            // [qtm][2]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int* hr = half_rec + (size_t)blockIdx.x * SN_HREC;
    const int ns = hr[0], partner = hr[2], owner = hr[3], tnode = RING ? hr[4] : -1, QT = RING ? hr[5] : 0;
    const int* sr0 = step_rec + (size_t)hr[1] * SN_SREC;
    double* wk = work + (size_t)hr[6];
    const size_t wstep = (size_t)2 * Q * LD + (size_t)qtm * LD + 2 * Q;
    const int* tcams = node_cam + (size_t)max(tnode, 0) * CAPT;
    // ---- prologue: the first pivot (or M when the half has none), every right-hand side, T's own block
    sn_gather<DC>(Dbuf, LD, Q, Q, S_val, tabs, hr[9], S, tid, SN_THREADS);
    if (RING) sn_gather<DC>(Ebuf, LD, QT, Q, S_val, tabs, hr[10], S, tid, SN_THREADS);
    if (RING) sn_gather<DC>(ATT, LDT, QT, QT, S_val, tabs, hr[11], CAPT, tid, SN_THREADS);
    if (ns > 0) sn_gather<DC>(Cbuf, LD, Q, Q, S_val, tabs, sr0[2], S, tid, SN_THREADS);
    sn_gather_rhs<DC, 2>(gAll, node_cam + (size_t)hr[13] * CAPT, Q, ns > 0 || owner, rhs, Sfc, tid, SN_THREADS);
    for (int k = 0; k < ns; k++) sn_gather_rhs<DC, 2>(gAll + (size_t)(k + 1) * 2 * Q, node_cam + (size_t)sr0[k * SN_SREC + 1] * CAPT, Q, sr0[k * SN_SREC + 5] != 0, rhs, Sfc, tid, SN_THREADS);
    if (RING) sn_gather_rhs<DC, 2>(gT, tcams, QT, owner != 0, rhs, Sfc, tid, SN_THREADS);
    __syncthreads();
    int cur = 0;
    bool ok = true;
    double row[Q + NR], rowT[Q + NR];
    if (wave == 0) {
#pragma unroll
        for (int t = 0; t < NR; t++) rowT[Q + t] = (RING && lane < QT) ? gT[lane * 2 + t] : 0.0;
    }
    // ---- elimination of the half's pivots
    for (int k = 0; k < ns; k++) {
        const int* sr = sr0 + (size_t)k * SN_SREC;
        const int nxt = cur ^ 1;
        double* Dc = Dbuf + cur * Q * LD; double* Dn = Dbuf + nxt * Q * LD;
        double* Cc = Cbuf + cur * Q * LD; double* Cn = Cbuf + nxt * Q * LD;
        double* Ec = Ebuf + cur * qtm * LD; double* En = Ebuf + nxt * qtm * LD;
        double* gP = gAll + (size_t)k * 2 * Q; double* gN = gAll + (size_t)(k + 1) * 2 * Q;
        if (wave == 0) {
            const int sl = lane - 32;
            const bool dl = lane < Q, sub = sl >= 0 && sl < Q;
            const double* src = dl ? Dc + lane * LD : Cc + (sub ? sl : 0) * LD;
#pragma unroll
            for (int j = 0; j < Q; j++) { const double v = src[j]; row[j] = (dl || sub) ? v : 0.0; }
#pragma unroll
            for (int t = 0; t < NR; t++) { const double v = dl ? gP[lane * 2 + t] : gN[(sub ? sl : 0) * 2 + t]; row[Q + t] = (dl || sub) ? v : 0.0; }
            if (RING) {
                const double* srcT = Ec + (lane < QT ? lane : 0) * LD;
#pragma unroll
                for (int j = 0; j < Q; j++) { const double v = srcT[j]; rowT[j] = lane < QT ? v : 0.0; }
            }
            ok = sn_panel<NR, RING>(row, rowT, lane) && ok;
            lds_barrier();                                                   // B0: the helpers are done with the previous step's outputs
            double* dst = dl ? Ldd + lane * LD : Lsd + (sub ? sl : 0) * LD;
            if (dl || sub) {
#pragma unroll
                for (int j = 0; j < Q; j++) dst[j] = row[j];
            }
#pragma unroll
            for (int t = 0; t < NR; t++) { if (dl) yP[lane * 2 + t] = row[Q + t]; else if (sub) gN[sl * 2 + t] = row[Q + t]; }
            if (RING && lane < QT) {
#pragma unroll
                for (int j = 0; j < Q; j++) LTP[lane * LD + j] = rowT[j];
            }
            lds_barrier();                                                   // B1: the panel is in LDS
        } else {
            const int ht = tid - 64, hn = SN_THREADS - 64, hw = wave - 1;
            if (k > 0) {
                if (RING) sn_syrk(ATT, LDT, LTP, LD, QT, hw, 3, lane);       // T x T takes the previous step's share
                double* wp = wk + (size_t)(k - 1) * wstep;                   // the previous step's factor blocks and y -> global memory (read again by the back substitution)
                for (int e = ht; e < Q * LD; e += hn) { wp[e] = Ldd[e]; wp[Q * LD + e] = Lsd[e]; }
                if (RING) for (int e = ht; e < QT * LD; e += hn) wp[2 * Q * LD + e] = LTP[e];
                for (int e = ht; e < 2 * Q; e += hn) wp[2 * Q * LD + (size_t)qtm * LD + e] = yP[e];
            }
            sn_gather<DC>(Dn, LD, Q, Q, S_val, tabs, sr[3], S, ht, hn);      // the next supernode's original blocks
            if (RING) sn_gather<DC>(En, LD, QT, Q, S_val, tabs, sr[4], S, ht, hn);
            if (k + 1 < ns) sn_gather<DC>(Cn, LD, Q, Q, S_val, tabs, sr[SN_SREC + 2], S, ht, hn);
            lds_barrier();                                                   // B0
            lds_barrier();                                                   // B1
        }
        // ---- Schur updates on the matrix cores: D(N) -= Lsd Lsd^T (lower tiles), E(T, N) -= LTP Lsd^T
        sn_syrk(Dn, LD, Lsd, LD, Q, wave, 4, lane);
        if (RING) sn_gemm(En, LD, LTP, LD, QT, Lsd, LD, Q, wave, 4, 3, lane);
        lds_barrier();                                                       // B2
        cur = nxt;
    }
    // ---- the last step's outputs: T x T share, factor blocks to global memory
    if (ns > 0) {
        if (RING) sn_syrk(ATT, LDT, LTP, LD, QT, wave, 4, lane);
        double* wp = wk + (size_t)(ns - 1) * wstep;
        for (int e = tid; e < Q * LD; e += SN_THREADS) { wp[e] = Ldd[e]; wp[Q * LD + e] = Lsd[e]; }
        if (RING) for (int e = tid; e < QT * LD; e += SN_THREADS) wp[2 * Q * LD + e] = LTP[e];
        for (int e = tid; e < 2 * Q; e += SN_THREADS) wp[2 * Q * LD + (size_t)qtm * LD + e] = yP[e];
    }
    if (RING && wave == 0 && lane < QT) {
#pragma unroll
        for (int t = 0; t < NR; t++) gT[lane * 2 + t] = rowT[Q + t];
        if (NR == 1) gT[lane * 2 + 1] = 0.0;
    }
    __syncthreads();
    double* DM = Dbuf + cur * Q * LD;                     // M's diagonal block, E(T, M), M's right-hand side: this half's share
    double* EM = Ebuf + cur * qtm * LD;
    double* gM = gAll + (size_t)ns * 2 * Q;
    // ---- exchange with the partner half: both end up with the same sums
    if (partner >= 0) {
        double* mine = xchg + (size_t)hr[7]; const double* other = xchg + (size_t)hr[14];
        const int nD = Q * LD, nE = qtm * LD, nTT = qtm * LDT;
        for (int e = tid; e < nD; e += SN_THREADS) mine[e] = DM[e];
        for (int e = tid; e < 2 * Q; e += SN_THREADS) mine[nD + e] = gM[e];
        if (RING) {
            for (int e = tid; e < nE; e += SN_THREADS) mine[nD + 2 * Q + e] = EM[e];
            for (int e = tid; e < nTT; e += SN_THREADS) mine[nD + 2 * Q + nE + e] = ATT[e];
            for (int e = tid; e < 2 * qtm; e += SN_THREADS) mine[nD + 2 * Q + nE + nTT + e] = gT[e];
        }
        __threadfence(); __syncthreads();
        if (tid == 0) {
            __hip_atomic_store(flags + hr[8], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(flags + hr[15], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != seq) __builtin_amdgcn_s_sleep(2);
        }
        __syncthreads(); __threadfence();
        for (int e = tid; e < nD; e += SN_THREADS) DM[e] += other[e];
        for (int e = tid; e < 2 * Q; e += SN_THREADS) gM[e] += other[nD + e];
        if (RING) {
            for (int e = tid; e < nE; e += SN_THREADS) EM[e] += other[nD + 2 * Q + e];
            for (int e = tid; e < nTT; e += SN_THREADS) ATT[e] += other[nD + 2 * Q + nE + e];
            for (int e = tid; e < 2 * qtm; e += SN_THREADS) gT[e] += other[nD + 2 * Q + nE + nTT + e];
        }
        __syncthreads();
    }
    // ---- M, then T (both halves of a ring do this redundantly, on identical data)
    const int QT2 = RING ? max(QT - Q, 0) : 0;
    const bool hasT = RING && QT > 0;
    if (wave == 0) {
        const bool dl = lane < Q;
#pragma unroll
        for (int j = 0; j < Q; j++) { const double v = DM[(dl ? lane : 0) * LD + j]; row[j] = dl ? v : 0.0; }
#pragma unroll
        for (int t = 0; t < NR; t++) { const double v = gM[(dl ? lane : 0) * 2 + t]; row[Q + t] = dl ? v : 0.0; }
        if (RING) {
#pragma unroll
            for (int j = 0; j < Q; j++) { const double v = EM[(lane < QT ? lane : 0) * LD + j]; rowT[j] = lane < QT ? v : 0.0; }
#pragma unroll
            for (int t = 0; t < NR; t++) rowT[Q + t] = lane < QT ? gT[lane * 2 + t] : 0.0;
        }
        ok = sn_panel<NR, RING>(row, rowT, lane) && ok;
        if (dl) {
#pragma unroll
            for (int j = 0; j < Q; j++) DM[lane * LD + j] = row[j];
#pragma unroll
            for (int t = 0; t < NR; t++) yM[lane * 2 + t] = row[Q + t];
        }
        if (RING && lane < QT) {
#pragma unroll
            for (int j = 0; j < Q; j++) EM[lane * LD + j] = rowT[j];
#pragma unroll
            for (int t = 0; t < NR; t++) gT[lane * 2 + t] = rowT[Q + t];
        }
    }
    if (hasT) {
        __syncthreads();
        sn_syrk(ATT, LDT, EM, LD, QT, wave, 4, lane);                        // T x T -= L(T, M) L(T, M)^T
        __syncthreads();
        if (wave == 0) {                                                     // T_1 = the first 30 rows of T, T_2 (<= 30 rows) rides as the coupling block
            const int sl = lane - 32;
            const bool dl = lane < Q, sub = sl >= 0 && sl < QT2;
            const int r = dl ? lane : (sub ? Q + sl : 0);
#pragma unroll
            for (int j = 0; j < Q; j++) { const double v = ATT[(size_t)r * LDT + j]; row[j] = (dl || sub) ? v : 0.0; }
#pragma unroll
            for (int t = 0; t < NR; t++) { const double v = gT[r * 2 + t]; row[Q + t] = (dl || sub) ? v : 0.0; }
            double none[Q + NR];
            ok = sn_panel<NR, false>(row, none, lane) && ok;
            if (dl || sub) {
#pragma unroll
                for (int j = 0; j < Q; j++) ATT[(size_t)r * LDT + j] = row[j];
#pragma unroll
                for (int t = 0; t < NR; t++) { if (dl) yT[r * 2 + t] = row[Q + t]; else gT[r * 2 + t] = row[Q + t]; }
            }
        }
        if (QT2 > 0) {
            __syncthreads();
            sn_syrk(ATT + (size_t)Q * LDT + Q, LDT, ATT + (size_t)Q * LDT, LDT, QT2, wave, 4, lane);      // T_2 x T_2 -= L_21 L_21^T
            __syncthreads();
            if (wave == 0) {
                const bool dl = lane < Q, real = lane < QT2;
#pragma unroll
                for (int j = 0; j < Q; j++) { const double v = ATT[(size_t)(Q + (real ? lane : 0)) * LDT + Q + min(j, max(QT2 - 1, 0))]; row[j] = (real && j < QT2) ? v : ((dl && j == lane) ? 1.0 : 0.0); }
#pragma unroll
                for (int t = 0; t < NR; t++) { const double v = gT[(Q + (real ? lane : 0)) * 2 + t]; row[Q + t] = real ? v : 0.0; }
                double none[Q + NR];
                ok = sn_panel<NR, false>(row, none, lane) && ok;
                if (dl) {                                                    // L_22 (identity padded) to a buffer of its own: Ldd is free now
#pragma unroll
                    for (int j = 0; j < Q; j++) Ldd[lane * LD + j] = row[j];
#pragma unroll
                    for (int t = 0; t < NR; t++) if (real) yT[(Q + lane) * 2 + t] = row[Q + t];
                }
            }
        }
    }
    if (wave == 0 && lane == 0 && !ok) *fail_flag = 1;
    __syncthreads();
    // ---- back substitution: T_2, T_1, M, then this half's pivots from the last to the first; one wave
    if (wave == 0) {
        double x[NR];
        const int* mcams = node_cam + (size_t)hr[12] * CAPT;
        if (hasT) {
            if (QT2 > 0) {
                sn_back<NR>(Ldd, LD, yT + 2 * Q, QT2, nullptr, 0, 0, nullptr, nullptr, 0, 0, nullptr, x, lane);
                if (lane < QT2) {
#pragma unroll
                    for (int t = 0; t < NR; t++) xT[(Q + lane) * 2 + t] = x[t];
                }
                if (owner && lane < QT2) {
                    const int i = Q + lane, a = i / DC, u = i - a * DC, cam = tcams[a];
                    if (cam >= 0) {
#pragma unroll
                        for (int t = 0; t < NR; t++) Y[(size_t)t * ystride + (size_t)pos[cam] * DC + u] = x[t];
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
            sn_back<NR>(ATT, LDT, yT, Q, ATT + (size_t)Q * LDT, LDT, QT2, xT + 2 * Q, nullptr, 0, 0, nullptr, x, lane);
            if (lane < Q) {
#pragma unroll
                for (int t = 0; t < NR; t++) xT[lane * 2 + t] = x[t];
            }
            if (owner) sn_store_y<DC, NR>(Y, ystride, tcams, pos, Q, x, lane);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        sn_back<NR>(DM, LD, yM, Q, nullptr, 0, 0, nullptr, EM, LD, QT, xT, x, lane);
        if (lane < Q) {
#pragma unroll
            for (int t = 0; t < NR; t++) xN[lane * 2 + t] = x[t];
        }
        if (owner) sn_store_y<DC, NR>(Y, ystride, mcams, pos, Q, x, lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int k = ns - 1; k >= 0; k--) {
            const double* wp = wk + (size_t)k * wstep;
            sn_back<NR>(wp, LD, wp + 2 * Q * LD + (size_t)qtm * LD, Q, wp + Q * LD, LD, Q, xN, wp + 2 * Q * LD, LD, QT, xT, x, lane);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (lane < Q) {
#pragma unroll
                for (int t = 0; t < NR; t++) xN[lane * 2 + t] = x[t];
            }
            sn_store_y<DC, NR>(Y, ystride, node_cam + (size_t)sr0[k * SN_SREC] * CAPT, pos, Q, x, lane);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
}

}  // namespace ssfm
