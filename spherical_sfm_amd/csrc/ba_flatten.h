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
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <unistd.h>
#include <vector>
#include "../../include/ssfm.h"
#include "ring_comp.h"
#include "knobs.h"

namespace ssfm {

// std::vector whose resize() leaves new elements uninitialised: the planner's per-observation arrays (19 MB at 600k observations) are written
// once, in parallel, right after they are sized -- value-initialising them first was a serial 1 ms memset of the cold plan.
template <class T>
struct NoInitAlloc : std::allocator<T> {
    template <class U> struct rebind { using other = NoInitAlloc<U>; };
    NoInitAlloc() = default;
    template <class U> NoInitAlloc(const NoInitAlloc<U>&) {}
    template <class U, class... A> void construct(U* p, A&&... a) {
        if constexpr (sizeof...(A) == 0) ::new ((void*)p) U; else ::new ((void*)p) U(std::forward<A>(a)...);
    }
};
template <class T> using raw_vector = std::vector<T, NoInitAlloc<T>>;

struct BAFlat {
    int Nc = 0;
    int nP = 0;               // used points owned by this rank
    int nP_global = 0;
    int64_t M = 0, M_global = 0;
    int DC = 6;               // 3: every translation fixed (spherical BA) -> only rotations vary
    bool focal_free = false;
    raw_vector<int> pt_ids;             // [nP] original point id
    std::vector<int> pt_start;          // [nP+1]
    raw_vector<int> obs_cam;            // [M]
    raw_vector<int> obs_pt;             // [M] compact local point
    raw_vector<int64_t> obs_orig;       // [M] index into the caller's arrays
    raw_vector<double> obs_xy;          // [2M]
    std::vector<int> cam_start;         // camera-major lists over local observations
    raw_vector<int> cam_obs, cam_obs_pt;   // observation / point of each camera-major entry
    std::vector<int> cs_task_cam, cs_task_q0, cs_task_q1;   // point of each camera-major entry; k_cam_sums2 wave tasks (<= 256 entries)
    std::vector<int> row_ptr, col_idx, diag_slot;   // block-CSR structure of S (global), sorted columns
    std::vector<double> mask_cam;       // [Nc*6] 1 = free parameter that is in the problem
    raw_vector<double> mask_pt;         // [nP*3]
    raw_vector<double> pts0;            // [nP*3]
    int max_row_blocks = 0;
    bool nothing_to_do = false;
    // elimination order of the camera blocks for the banded Cholesky preconditioner
    std::vector<int> cam_pos;           // [Nc] camera -> position
    int band = 0;                       // block half-bandwidth in that order
    std::vector<int> band_pairs;        // (i_rel | k_rel << 16), i_rel-major, 1 <= k_rel <= i_rel <= band
    std::vector<int> comp_ptr;          // connected components of the camera graph = contiguous ranges of BAND ROWS
    // Band rows.  Normally band row = elimination position.  A "twisted" component (band_twist_plan) is eliminated from both
    // ends towards a separator of `band` rows in the middle; its band rows are  seg_0 | sep | seg_1 reversed | second copy of sep
    // (the copy receives the Schur update from seg_1), so the band has band_rows >= Nc rows.
    std::vector<int> band_row, band_row2;   // [Nc] camera -> band row; second row of a separator camera of a twisted component, else -1
    std::vector<char> comp_twist;       // [components] 1 = twisted
    int band_rows = 0;                  // rows of the band in units of its blocks (band_block x band_block)
    // Block merging (band_plan): with 3-dof camera blocks the factorisation is pure per-step latency, so two consecutive cameras of the
    // elimination order share one 6x6 block row of the band: half the dependent steps.  band_row / band_row2 stay in CAMERA rows
    // (camera c sits in block row band_row[c] / 2, upper or lower half by parity); band, comp_ptr, band_rows are in block rows.
    int band_block = 0;                 // size of the band's blocks: DC, or 6 when DC = 3 and pairs are merged
    std::vector<unsigned char> pair_dummy;   // [Nc] 1 = this (even) camera's partner slot is empty (odd-sized component)
    int y_rows(int dc) const { return band_rows * (band_block > 0 ? band_block : dc) / dc; }   // rows of the right-hand sides in camera units
    // Round 5: long camera rings in their own circular order (band_plan: RingComp); wrap_* = the coupling blocks between a ring's first arc and its last separator,
    // which the gather kernels place (transposed) relative to the separator's copy slot in front of the arc (ring_wrap_table); empty without rings
    std::vector<RingComp> rings;
    std::vector<int> wrap_ptr, wrap_blk, wrap_row2;
    // The stored structure (row_ptr/col_idx) is the LOWER triangle in elimination order: row c holds block (c, c2) iff
    // cam_pos[c2] <= cam_pos[c].  trans_* lists, for every camera c, the stored blocks of OTHER rows whose column is c
    // (the upper triangle by symmetry) for the symmetric mat-vec.
    bool sym_lower = false;
    std::vector<int> trans_ptr, trans_blk, trans_row;
    // Schur pair lists (this rank's observations): for row c the entries (j, j2) = (observation of c, observation of the
    // same point by a camera c2 of row c), grouped by slot and padded with -1 to whole 64-lane batches.
    std::vector<int> pair_j, pair_j2, pair_p, batch_slot, cam_batch_ptr;
    int task_batches = 8;               // batches per wave task of k_schur_pairs2 (pair_layout)
    int64_t pair_batches = 0;           // 64-entry batches of the pair lists (the lists themselves may live on the device only)
    // work chunks for the pair kernel: <= 16 consecutive batches of ONE camera each (balances rows of very different size)
    std::vector<int> chunk_cam, chunk_b0, chunk_b1;
    // Signature groups (round 3, k_schur_gram / k_gram_backsub): runs of >= GRAM_MIN_RUN consecutive points observed by exactly the same K cameras,
    // GRAM_KMIN <= K <= GRAM_KMAX (3..8).  Their Schur blocks (off-diagonal and diagonal) and camera-side sums are formed as ONE Gram product per wave task on the
    // matrix cores, each observation linearised once, instead of lane-per-pair from the pair lists and k_cam_sums2 (which then skip these points: pt_grouped).
    // Tasks are sorted by K (k_schur_gram is launched once per number of 16-row tiles in use).  One record of GRAM_REC ints per task:
    //   [0] first point  [1] points (<= the task length chosen below)  [2] K  [3] first observation (the K of every point follow each other)
    //   [4..11] the cameras, ascending  [12..39] slot[a (a - 1) / 2 + b] (a > b) = block index of (camera a, camera b) in S, bit 30 set when
    //   the stored block is (b, a), i.e. the transpose  [40..47] the diagonal block of every camera
    // k_schur_gram also produces the camera-side sums (k_cam_sums2's) of these points: the cs_task lists only cover cameras that have other points
    std::vector<int> gr_rec;
    raw_vector<unsigned char> pt_grouped;   // [nP] 1 = handled by a signature group
    int64_t gram_points = 0, gram_obs = 0;
    bool gram_sorted = false;               // the local points were re-ordered by camera-list signature (ba_flatten: signature sort)
    // Round 6 -- atomics-free emission of k_schur_gram when EVERY point sits in a signature group (the circle workloads): task t writes its K (K + 1) / 2 blocks and its
    // cameras' vectors with plain stores into its own stretch of a partial buffer (gpart_off[t]; layout: gram_part_* below) and k_finalize_gather folds, per row of S,
    // the partial blocks of every slot in a FIXED order (fold_slot_*).  No atomics, no clears, the same bits every run.
    std::vector<int> gpart_off;             // [tasks + 1] offset of every task's stretch in doubles
    std::vector<int> fold_slot_ptr, fold_slot_src;      // fold_slot_src: [Nc][GRAM_FOLD_STRIDE] one table per camera row (sources, first source of every extended slot, the sources' offsets); fold_slot_ptr: unused
    bool gram_any = true;                   // SSFM_GRAM_ANY as read when the plan was made: every tile class in one k_schur_gram_any launch (the cost model below assumes what the launch then does)
};
// a task's partial stretch: [K (K + 1) / 2 blocks of DC x DC, block (a, b), b <= a, at a (a + 1) / 2 + b, off-diagonal ones in the orientation S stores][K x (diag U | rhs 1 | rhs 2 | Jc^T r | S_fc) of DC each]
inline int gram_part_blocks(int K) { return K * (K + 1) / 2; }
inline int gram_part_len(int K, int DC) { return gram_part_blocks(K) * DC * DC + K * 5 * DC; }
constexpr int GRAM_FOLD_PTRS = 31, GRAM_FOLD_HEAD = 32, GRAM_FOLD_SRCS = 128, GRAM_FOLD_STRIDE = GRAM_FOLD_HEAD + GRAM_FOLD_SRCS;      // the fold table of one camera row: <= 29 blocks + the vectors, <= 128 sources
constexpr int GRAM_KMAX = 8, GRAM_NPAIR = 28, GRAM_REC = 48, GRAM_MIN_RUN = 32, GRAM_KMIN = 3, GRAM_SUB_PTS = 8;      // GRAM_SUB_PTS = GRAM_SUB of ba_kernels.h (points per sub-chunk)

// fork-join over [0, n) in T contiguous chunks; f(thread index, begin, end).  The planner's loops over cameras / points are independent
// once the prefix sums are known.  The T - 1 helpers are persistent (a pool parked on a condition variable): spawning seven std::threads per
// section cost more than some of the sections they ran (0.1 ms each on the MI355X hosts; a plan has six sections).  SSFM_PLAN_POOL=0 goes
// back to one std::thread per chunk.
class PlanPool {
    std::vector<std::thread> workers;
    std::mutex m; std::condition_variable cv_go, cv_done;
    std::function<void(int)> job; int generation = 0, pending = 0, active = 0; bool stop = false;
    void loop(int id) {
        int seen = 0;
        for (;;) {
            std::function<void(int)> f;
            { std::unique_lock<std::mutex> lk(m);
              cv_go.wait(lk, [&] { return stop || (generation != seen && id < active); });
              if (stop) return;
              seen = generation; f = job; }
            f(id + 1);
            { std::lock_guard<std::mutex> lk(m); if (--pending == 0) cv_done.notify_one(); }
        }
    }
public:
    ~PlanPool() { { std::lock_guard<std::mutex> lk(m); stop = true; } cv_go.notify_all(); for (auto& t : workers) t.join(); }
    // runs f(0) here and f(1) .. f(T-1) on the helpers; returns when all are done.  One caller at a time (the planner is single-entry per context;
    // callers from different threads serialise on run_mutex).
    std::mutex run_mutex;
    void run(int T, const std::function<void(int)>& f) {
        std::lock_guard<std::mutex> one(run_mutex);
        while ((int)workers.size() < T - 1) { const int id = (int)workers.size(); workers.emplace_back([this, id] { loop(id); }); }
        { std::lock_guard<std::mutex> lk(m); job = f; active = T - 1; pending = T - 1; generation++; }
        cv_go.notify_all();
        f(0);
        std::unique_lock<std::mutex> lk(m);
        cv_done.wait(lk, [&] { return pending == 0; });
        active = 0;
    }
};
// One pool per process, joined at exit.  A fork()ed child inherits the object but none of its threads: it abandons the inherited pool (whose
// std::thread handles must not be destroyed) and starts a fresh one on first use.
struct PlanPoolHolder {
    PlanPool* pool = nullptr; long owner_pid = 0;
    ~PlanPoolHolder() { if (pool && owner_pid == (long)getpid()) delete pool; }
};
inline PlanPool& plan_pool() {
    static PlanPoolHolder h;
    static std::mutex guard;
    std::lock_guard<std::mutex> lk(guard);
    const long pid = (long)getpid();
    if (!h.pool || h.owner_pid != pid) { h.pool = new PlanPool(); h.owner_pid = pid; }      // (the inherited pool of a forked child is leaked on purpose)
    return *h.pool;
}
template <class Fn>
inline void parallel_chunks(int64_t n, int T, Fn f) {
    if (T <= 1 || n < 4 * (int64_t)T) { f(0, (int64_t)0, n); return; }
    static const bool use_pool = SSFM_LAB_KNOB("SSFM_PLAN_POOL", 1) != 0;
    if (use_pool) { plan_pool().run(T, [&](int t) { f(t, n * t / T, n * (t + 1) / T); }); return; }
    std::vector<std::thread> th; th.reserve(T);
    for (int t = 0; t < T; t++) th.emplace_back([=]() { f(t, n * t / T, n * (t + 1) / T); });
    for (auto& x : th) x.join();
}
inline int planner_threads() {
    if (const char* e = std::getenv("SSFM_PLAN_THREADS")) return std::max(1, std::atoi(e));
    // 16 = the thread count the reference gives Ceres (src/sfm.cpp:209) and the CPU baseline uses.  Measured on the MI355X host (256 cores), 600k observations,
    // first call on a structure: plan 3.1 / 2.0 / 1.35 / 1.15 ms with 4 / 8 / 16 / 32 threads; beyond 16 the observation upload on its own thread loses its overlap
    return (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
}

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
            if (u == start && nb.size() > 2) {
                // The children of the root all have the same parent, so their order is arbitrary to plain Cuthill-McKee -- and it decides how the
                // later levels interleave.  On a ring (root in the middle of its neighbours) the id order mixes the two sides unevenly: half-width
                // 12 instead of 2 x reach = 10 at BASELINE config 2.  Closest first: by the number of neighbours a child shares with the root
                // (s+1, s-1, s+2, s-2, ... on a ring -- the fold; nearest first on a path).
                std::vector<char> is_nb(n, 0);
                for (int e = row_ptr[u]; e < row_ptr[u + 1]; e++) is_nb[col_idx[e]] = 1;
                std::vector<int> shared(nb.size(), 0);
                for (size_t k = 0; k < nb.size(); k++) for (int e = row_ptr[nb[k]]; e < row_ptr[nb[k] + 1]; e++) shared[k] += is_nb[col_idx[e]];
                std::vector<int> idx(nb.size()); for (size_t k = 0; k < nb.size(); k++) idx[k] = (int)k;
                std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return shared[a] > shared[b]; });
                std::vector<int> q2(nb.size()); for (size_t k = 0; k < nb.size(); k++) q2[k] = nb[idx[k]];
                nb.swap(q2);
            }
            for (int v : nb) { pos[v] = (int)order.size(); order.push_back(v); queue.push_back(v); }
        }
        if (comp_ptr) comp_ptr->push_back((int)order.size());
    }
    int band = 0;
    for (int u = 0; u < n; u++) for (int e = row_ptr[u]; e < row_ptr[u + 1]; e++) band = std::max(band, std::abs(pos[u] - pos[col_idx[e]]));
    return band;
}

// ---- band layout of the reduced camera system -------------------------------------------------------------------------------------------
// Twisted (two-sided) elimination of medium-sized components.  A component of n rows in Cuthill-McKee order with half-width b is
// split as seg_0 (m0 rows) | sep (b rows) | seg_1 (m1 rows).  seg_0 is eliminated front to back and seg_1 BACK TO FRONT, so both
// have the separator behind them: no fill, and the dependent chain of the factorisation is (n - b) / 2 + b steps instead of n.
// Elimination order seg_0 | seg_1 reversed | sep; band rows seg_0 | sep | seg_1 reversed | copy of sep.
// A component is twisted when both segments are longer than the band (n >= 3 b + 2) and it is not long enough to be cut into
// several segments (band_sub.h, SUB_MIN_ROWS); the kernels involved need the LDS-resident factorisation (b <= 20).
// Half-widths above 20 (6x6 blocks) need the packed-window factorisation (band_kernels2p.h, ba_handle.h: band_wide_packed): with it on (default) components are
// twisted up to half-width 30, with SSFM_BAND_PACKED=0 up to 20 as before (the global-memory kernels that then serve wide bands know no twisted layout).
inline bool band_packed_enabled() { const char* e = std::getenv("SSFM_BAND_PACKED"); return !(e && std::atoi(e) == 0); }
inline int band_wide_max() { return band_packed_enabled() ? 30 : 20; }
constexpr int BAND_CUT_MIN_ROWS = 512;            // below: twisted (two workgroups, no spike); from here on: cut into a chain of segments (band_sub.h)

// Round 5 -- ring-native layout of long camera rings (band_ring.h).  Cuthill-McKee FOLDS a ring: the two sides of the ring are interleaved in one band of
// twice the ring's reach r, although they do not touch each other except at the two turning points.  Everything downstream pays for the doubled width: the
// factorisation window (b + 1)^2, the b DC spike columns, the dense separator blocks (b DC)^2.5.  A ring in its own circular order is a PERIODIC band of half-width r:
//     [copy of S_{m-1}] | A_0 | S_0 | A_1 | S_1 | ... | A_{m-1} | S_{m-1}            (A = arcs, S = separators of r block rows, m >= 2 cuts)
// Every arc has the separator behind it adjacent (the factorisation's window continues into it) and the separator in front of it adjacent too -- for A_0 that is
// S_{m-1}, which gets a second set of r band rows in front of A_0 that only carries the coupling blocks (A_0, S_{m-1}) (the gather kernels place them there,
// transposed: ring_wrap_table).  The separators form a CYCLE of m dense blocks of r DC unknowns -- half the size of the folded ones --, solved by cyclic reduction
// (band_ring.h).  A component is laid out like this when it is long (or too wide for the LDS window when folded), periodic in camera-id order with a reach well
// under the folded half-width, and its separator blocks fit the cyclic-reduction kernel (r DC <= RING_QMAX).  SSFM_RING=0 turns it off; SSFM_RING_CUTS=m forces m.
constexpr int RING_QMAX = 78;                     // (3 Q + 2) (Q | 1) doubles of LDS in k_ring_cr_elim: 149 KB at Q = 78, i.e. a 160 KB compute unit
// LDS a workgroup may ask for on the device the plans are made for (ctx.hip sets it from hipDeviceAttributeMaxSharedMemoryPerBlock; 160 KB on gfx950, the only target):
// a ring whose cyclic-reduction blocks would not fit stays on the fold instead of failing in hipFuncSetAttribute at solve time
inline bool ring_blocks_fit(int Q) { return (size_t)(3 * Q + 2) * (size_t)(Q | 1) * sizeof(double) <= plan_lds_limit(); }
inline int ring_choose_cuts(int rows, int b) {
    if (const char* e = std::getenv("SSFM_RING_CUTS")) { const int m = std::atoi(e); if (m >= 2) return std::min(m, std::max(2, rows / (2 * b + 1))); }
    // A power of two: the cyclic reduction halves the cycle per step down to 2 separators, which one workgroup per ring finishes (band_ring.h), so m = 2 * 2^k costs k
    // parallel steps whatever lies between two powers.  The largest one that leaves arcs of at least max(2 b - 2, 16) rows: an arc costs its length in dependent
    // steps three times over (factorisation, spike, back substitution: ~2.3 us per row), a doubling of the cuts one more step down and one more step back up (~35 us).
    const int min_arc = std::max(2 * b - 2, 16);      // (r05ae: 300 rows at b = 13: 8 cuts = arcs of 24.5 rows 503 us per iteration, 4 cuts 513; b = 10: 4 / 8 / 16 cuts 408 / 390 / 422; b = 7: 8 / 16 cuts 332 / 346)
    int m = 2;
    while (2 * m <= 256 && (rows - 2 * m * b) / (2 * m) >= min_arc) m *= 2;
    m = std::max(2, std::min(m, rows / (2 * b + 1)));
    return m;
}

// Short rings (under BAND_CUT_MIN_ROWS block rows, folded half-width <= 20): ring or fold by a cost model in microseconds per factorisation + solve, fitted to per-kernel
// times measured on MI355X (profiles/r05_notes.md r05n; 6x6 blocks).
//   fold, twisted (two launches of the windowed factorisation + two of the back substitution): 37 + s(bk) per PAIR of rows, s = 1.27 at half-width 10 ... 2.82 at 20.
//   ring: arcs of L rows 35.5 + 2.1 L (factorisation 0.95, spike 0.85, back substitution 0.3 per row; assembly, left apply); a cyclic-reduction step on separators of Q
//   unknowns 34 + 1.2 (Q - 30) + 0.004 (Q - 30)^2 (elimination + back substitution), the closing launch 47 + 1.5 (Q - 30) + 0.0135 (Q - 30)^2.
// Measured fold -> ring per LM iteration: one ring of 300 cameras with reach 5 / 7 / 8 / 9 / 10: 331 -> 307, 427 -> 361, 476 -> 397, 526 -> 423, 596 -> 459 (model
// differences 9 / 56 / 77 / - / 117 against 24 / 66 / 79 / 103 / 137 measured: the ring's extra launches overlap when nothing synchronises between them, hence the
// 5 us in favour of the ring below); rings of 75 ... 80 (config 2): 175 -> 232, 138 -> 191, 155 -> 212 (stay folded); one ring of 200 / 250 with reach 5:
// 217 -> 224 / 253 -> 239; 150 merged rows with reach 7: 321 -> 347.
constexpr int RING_MIN_ROWS = 64;
inline double fold_model_us(int rows, int bk) { return 37.0 + 0.5 * rows * std::max(0.9, 1.27 + 0.155 * (bk - 10)); }
inline double ring_model_us(int rows, int reach, int Q) {
    const int m = ring_choose_cuts(rows, reach);
    const double L = double(rows - m * reach) / m, dq = Q - 30.0;
    int steps = 0; for (int p = m; p > 2; p = (p + 1) / 2) steps++;
    return 35.5 + 2.1 * L + steps * (34.0 + 1.2 * dq + 0.004 * dq * dq) + 47.0 + 1.5 * dq + 0.0135 * dq * dq;
}

// Elimination order + band layout of the reduced camera system: Cuthill-McKee per component, then per component one of
//   ring (above) | twisted | plain,   with (3-dof blocks) pairs of consecutive cameras merged into 6x6 block rows.
//   out: pos (elimination order, a permutation), band (half-width in blocks), comp_ptr / band_rows (block rows), band_row / band_row2
//        (camera rows), comp_twist, band_block, pair_dummy, rings (optional: nullptr = never lay out rings)
inline void band_plan(int Nc, int dc, const std::vector<int>& row_ptr, const std::vector<int>& col_idx, std::vector<int>& pos, int& band,
                      std::vector<int>& comp_ptr, std::vector<int>& band_row, std::vector<int>& band_row2, std::vector<char>& comp_twist,
                      int& band_rows, int& band_block, std::vector<unsigned char>& pair_dummy, std::vector<RingComp>* rings = nullptr) {
    cuthill_mckee(Nc, row_ptr, col_idx, pos, &comp_ptr);
    pair_dummy.assign(Nc, 0);
    if (rings) rings->clear();
    const char* env_m = std::getenv("SSFM_BAND_MERGE");
    const bool merge = dc == 3 && !(env_m && env_m[0] == '0');
    const int W = merge ? 2 : 1;                                           // cameras per block row
    band_block = merge ? 6 : dc;
    const int ncomp = (int)comp_ptr.size() - 1;
    std::vector<int> inv(Nc); for (int c = 0; c < Nc; c++) inv[pos[c]] = c;  // Cuthill-McKee position -> camera
    // ---- per component: the sequence of its cameras (Cuthill-McKee order, or id order for a ring) and its half-width in block rows
    std::vector<std::vector<int>> seq(ncomp);
    std::vector<int> cb(ncomp, 0), crows(ncomp, 0); std::vector<char> is_ring(ncomp, 0);
    std::vector<std::vector<int>> seq_cm(ncomp); std::vector<int> cb_cm(ncomp, 0);      // rings: what Cuthill-McKee gave, in case the ring layout is withdrawn
    std::vector<int> where(Nc, 0);                                         // index of a camera inside its component's sequence
    const char* env_r = std::getenv("SSFM_RING");
    const bool ring_on = rings && !(env_r && env_r[0] == '0');
    const int wide_b = (dc == 3 && !merge) ? 40 : band_wide_max();        // widest band that is twisted
    for (int k = 0; k < ncomp; k++) {
        const int c0 = comp_ptr[k], n = comp_ptr[k + 1] - c0;
        seq[k].assign(inv.begin() + c0, inv.begin() + c0 + n);
        for (int i = 0; i < n; i++) where[seq[k][i]] = i;
        int bk = 0;
        for (int i = 0; i < n; i++) { const int u = seq[k][i]; for (int e = row_ptr[u]; e < row_ptr[u + 1]; e++) bk = std::max(bk, std::abs(i / W - where[col_idx[e]] / W)); }
        cb[k] = bk; crows[k] = (n + W - 1) / W;
        if (!ring_on) continue;
        // A circular order of the component in which it is a periodic band?  Two candidates: (a) the camera ids (video frames in temporal order with a loop closure),
        // (b) a walk round the ring from neighbour to nearest unvisited neighbour (rings whose ids are strided, e.g. SURVEY 8d's circle).  Whatever order comes out is only
        // USED when the component really is a narrow periodic band in it (the reach below is measured, not assumed).
        // reach = the largest circular distance between two coupled cameras, in block rows.
        if (crows[k] < RING_MIN_ROWS) continue;
        {   // a ring with a reach of <= 20 block rows couples a camera to <= 2 * 20 * W + W - 1 others: a denser component (an all-pairs pose graph) is none, and the walk
            // below costs n deg^2 -- not worth 30 ms of planning to find that out
            int64_t deg = 0; for (int i = 0; i < n; i++) deg += row_ptr[seq[k][i] + 1] - row_ptr[seq[k][i]];
            if (deg > (int64_t)n * (2 * 20 * W + W)) continue;
        }
        const int R = crows[k];
        auto circular_reach = [&](const std::vector<int>& ord) {
            for (int i = 0; i < n; i++) where[ord[i]] = i;
            int reach = 0;
            for (int i = 0; i < n; i++) { const int u = ord[i]; for (int e = row_ptr[u]; e < row_ptr[u + 1]; e++) { const int d = std::abs(i / W - where[col_idx[e]] / W); reach = std::max(reach, std::min(d, R - d)); } }
            return reach;
        };
        std::vector<int> ids(seq[k]); std::sort(ids.begin(), ids.end());
        std::vector<int> unf; unf.reserve(n);
        {   // greedy walk: from the current node to the unvisited neighbour that shares the most neighbours with it (on a ring: the next one in the same direction --
            // two nodes d <= reach apart share 2 reach - 1 - d neighbours), all the way round
            const std::vector<int>& cm = seq[k];
            for (int i = 0; i < n; i++) where[cm[i]] = i;
            std::vector<char> visited(n, 0), mine(n, 0);
            int cur = 0; visited[0] = 1; unf.push_back(cm[0]);
            int scan = 1;                                                  // first index that may still be unvisited (fallback when the walk is stuck)
            for (int step = 1; step < n; step++) {
                const int u = cm[cur];
                for (int e = row_ptr[u]; e < row_ptr[u + 1]; e++) mine[where[col_idx[e]]] = 1;
                int best = -1, best_cnt = -1;
                for (int e = row_ptr[u]; e < row_ptr[u + 1]; e++) {
                    const int vi = where[col_idx[e]];
                    if (visited[vi]) continue;
                    const int v2 = cm[vi]; int cnt = 0;
                    for (int e2 = row_ptr[v2]; e2 < row_ptr[v2 + 1]; e2++) cnt += mine[where[col_idx[e2]]];
                    if (cnt > best_cnt) { best_cnt = cnt; best = vi; }
                }
                for (int e = row_ptr[u]; e < row_ptr[u + 1]; e++) mine[where[col_idx[e]]] = 0;
                if (best < 0) { while (scan < n && visited[scan]) scan++; best = scan; }
                visited[best] = 1; unf.push_back(cm[best]); cur = best;
            }
        }
        const int ra = circular_reach(ids), rb = circular_reach(unf);
        std::vector<int>& best = (ra <= rb) ? ids : unf;
        const int reach = std::min(ra, rb);
        if (std::getenv("SSFM_RING_DEBUG")) std::fprintf(stderr, "[ring] component %d: %d cameras, %d block rows, fold half-width %d, circular reach by id %d, unfolded %d\n", k, n, R, bk, ra, rb);
        bool ok = reach >= 1 && reach * band_block <= RING_QMAX && ring_blocks_fit(reach * band_block) && 5 * reach <= 3 * bk && R >= 2 * (2 * reach + 1) && reach <= 20;
        // short components with a band that the square LDS window holds: only when the measured cost model says so (long ones and wide ones: always)
        if (ok && R < BAND_CUT_MIN_ROWS && bk <= 20 && !(env_r && env_r[0] == '2') && !(ring_model_us(R, reach, reach * band_block) < fold_model_us(R, bk) + 5.0)) ok = false;   // SSFM_RING=2: whenever possible
        if (ok) { is_ring[k] = 1; seq_cm[k] = seq[k]; cb_cm[k] = bk; seq[k] = best; cb[k] = reach; }
        for (int i = 0; i < n; i++) where[seq[k][i]] = i;
    }
    // one half-width for the whole band: a ring whose separators would not fit the cyclic-reduction kernel at THAT width (another component is wider) goes back to the fold
    for (;;) {
        band = merge ? 1 : 0;                                              // (6-dof: a graph without couplings keeps half-width 0, as before)
        for (int k = 0; k < ncomp; k++) band = std::max(band, cb[k]);
        int demote = -1;
        for (int k = 0; k < ncomp; k++) if (is_ring[k] && (band * band_block > RING_QMAX || !ring_blocks_fit(band * band_block) || crows[k] < 2 * (2 * band + 1))) { demote = k; break; }
        if (demote < 0) break;
        is_ring[demote] = 0; seq[demote].swap(seq_cm[demote]); cb[demote] = cb_cm[demote];
    }
    const char* env_t = std::getenv("SSFM_BAND_TWIST");
    const bool twist_ok = !(env_t && env_t[0] == '0') && band >= 1 && band <= wide_b;
    // ---- layout: block row i of component k -> elimination index, band row, second band row
    band_row.assign(Nc, -1); band_row2.assign(Nc, -1); comp_twist.assign(ncomp, 0);
    std::vector<long long> key(Nc);
    std::vector<int> new_ptr(1, 0);
    int base = 0, ebase = 0;                                               // band rows / elimination indices handed out so far (block rows)
    const int b = band;
    for (int k = 0; k < ncomp; k++) {
        const int n = (int)seq[k].size(), R = crows[k];
        std::vector<int> elim(R), brow(R), brow2(R, -1);
        if (is_ring[k]) {
            RingComp rc; rc.comp = k; rc.b = b; rc.rows = R;
            const int m = ring_choose_cuts(R, b), total = R - m * b;
            int row = base + b;                                            // [base, base + b): the copy slot of S_{m-1}
            int i = 0;
            for (int a = 0; a < m; a++) {
                const int len = total / m + (a < total % m ? 1 : 0);
                rc.arc_len.push_back(len);
                for (int t = 0; t < len + b; t++, i++) { elim[i] = i; brow[i] = row++; }
            }
            for (int t = 0; t < b; t++) brow2[R - b + t] = base + t;       // S_{m-1} = the last b block rows of the ring
            rings->push_back(rc);
            base += R + b;
        } else if (twist_ok && R >= 3 * b + 2 && R < BAND_CUT_MIN_ROWS) {
            comp_twist[k] = 1;
            const int m0 = (R - b) / 2, m1 = R - b - m0;
            for (int i = 0; i < m0; i++) { elim[i] = i; brow[i] = base + i; }
            for (int t = 0; t < b; t++) { elim[m0 + t] = m0 + m1 + t; brow[m0 + t] = base + m0 + t; brow2[m0 + t] = base + m0 + b + m1 + (b - 1 - t); }
            for (int u = 0; u < m1; u++) { elim[m0 + b + u] = m0 + (m1 - 1 - u); brow[m0 + b + u] = base + m0 + b + (m1 - 1 - u); }
            base += R + b;
        } else {
            for (int i = 0; i < R; i++) { elim[i] = i; brow[i] = base + i; }
            base += R;
        }
        for (int i = 0; i < n; i++) {
            const int c = seq[k][i], r = i / W, par = i - r * W;
            key[c] = (long long)(ebase + elim[r]) * W + par;
            band_row[c] = brow[r] * W + par;
            if (brow2[r] >= 0) band_row2[c] = brow2[r] * W + par;
        }
        if (merge && (n & 1)) pair_dummy[seq[k][n - 1]] = 1;
        ebase += R;
        new_ptr.push_back(base);
    }
    comp_ptr.swap(new_ptr); band_rows = base;
    // cameras: elimination keys compressed to a permutation
    std::vector<int> order(Nc); for (int c = 0; c < Nc; c++) order[c] = c;
    std::sort(order.begin(), order.end(), [&](int a, int b2) { return key[a] < key[b2]; });
    for (int r = 0; r < Nc; r++) pos[order[r]] = r;
}

// Coupling blocks between A_0 and S_{m-1} of a ring: S stores them in the row of the separator camera (it is eliminated later), the band wants them in the row of
// the A_0 camera, relative to the copy slot in front of it, TRANSPOSED.  Per camera c2 the list of (stored block of S, band row of the copy of its row camera).
// Needs the lower-triangle structure (row_ptr / col_idx by elimination order) and band_row / band_row2.
// (a camera of S_{m-1} is recognised by its second band row lying IN FRONT of its first: the copies of twisted separators lie behind)
inline void ring_wrap_table(int Nc, const std::vector<int>& row_ptr, const std::vector<int>& col_idx, const std::vector<int>& band_row, const std::vector<int>& band_row2,
                            int band, int W, std::vector<int>& wrap_ptr, std::vector<int>& wrap_blk, std::vector<int>& wrap_row2) {
    wrap_ptr.assign(Nc + 1, 0); wrap_blk.clear(); wrap_row2.clear();
    std::vector<std::vector<std::pair<int, int>>> per(Nc);
    for (int c = 0; c < Nc; c++) {
        const int i2 = band_row2[c];
        if (i2 < 0 || i2 >= band_row[c]) continue;
        for (int e = row_ptr[c]; e < row_ptr[c + 1]; e++) {
            const int c2 = col_idx[e], k = band_row[c2];
            if (c2 != c && k / W > i2 / W && k / W - i2 / W <= band && band_row[c] / W - k / W > band) per[c2].push_back({e, i2});
        }
    }
    for (int c = 0; c < Nc; c++) { for (auto& pr : per[c]) { wrap_blk.push_back(pr.first); wrap_row2.push_back(pr.second); } wrap_ptr[c + 1] = (int)wrap_blk.size(); }
}

// ---- Schur pair lists ---------------------------------------------------------------------------------------------------------
// For row camera c the entries (j, j2, p) = (observation of c, observation of the same point p by a camera c2 that precedes c in the
// elimination order), grouped by the slot of c2 in row c of S and padded with -1 to whole 64-entry batches.  Three steps: count per
// slot, layout (serial, tiny), fill.  ssfm_ba_create runs count and fill on the GPU (ba_kernels.h: k_pair_count / k_pair_fill) and only
// the layout here; the host versions serve the host-only callers (sanitizer test) and SSFM_HOST_PAIRS=1.
inline void pair_counts_host(const BAFlat& F, int NT, std::vector<int>& slot_cnt) {
    const int Nc = F.Nc;
    slot_cnt.assign(F.col_idx.size(), 0);
    parallel_chunks(Nc, NT, [&](int, int64_t c0, int64_t c1) {
        std::vector<int> slot_of(Nc, -1);
        for (int c = (int)c0; c < (int)c1; c++) {
            const int rb = F.row_ptr[c], nnb = F.row_ptr[c + 1] - rb, pc = F.cam_pos[c];
            for (int e = 0; e < nnb; e++) slot_of[F.col_idx[rb + e]] = e;
            for (int q = F.cam_start[c]; q < F.cam_start[c + 1]; q++) {
                const int p = F.cam_obs_pt[q];
                if (!F.pt_grouped.empty() && F.pt_grouped[p]) continue;                  // its blocks come from k_schur_gram
                for (int j2 = F.pt_start[p]; j2 < F.pt_start[p + 1]; j2++) { const int c2 = F.obs_cam[j2]; if (F.cam_pos[c2] < pc) slot_cnt[rb + slot_of[c2]]++; }   // diagonal blocks: k_cam_sums2
            }
            for (int e = 0; e < nnb; e++) slot_of[F.col_idx[rb + e]] = -1;
        }
    });
}
// batches of every slot, wave tasks of <= task_batches batches of ONE camera; slot_off = first pair entry of every slot; returns the
// number of 64-entry batches
inline int64_t pair_layout(BAFlat& F, const std::vector<int>& slot_cnt, std::vector<int64_t>& slot_off, int num_cus = 256) {
    const int Nc = F.Nc;
    F.cam_batch_ptr.assign(Nc + 1, 0); F.batch_slot.clear(); F.chunk_cam.clear(); F.chunk_b0.clear(); F.chunk_b1.clear();
    slot_off.assign(F.col_idx.size(), 0);
    int64_t nbatch_total = 0;
    for (int c = 0; c < Nc; c++) {
        int nbatch = 0;
        for (int e = F.row_ptr[c]; e < F.row_ptr[c + 1]; e++) {
            if (slot_cnt[e] == 0) continue;
            const int nb = (slot_cnt[e] + 63) / 64;
            slot_off[e] = (nbatch_total + nbatch) * 64;
            for (int b2 = 0; b2 < nb; b2++) F.batch_slot.push_back(e - F.row_ptr[c]);
            nbatch += nb;
        }
        F.cam_batch_ptr[c + 1] = F.cam_batch_ptr[c] + nbatch; nbatch_total += nbatch;
    }
    // batches per wave task of k_schur_pairs2.  A workgroup = 4 tasks, the CUs take whole workgroups in rounds, and a task costs its
    // batches + about one batch for the folds: 6..12 batches, whichever needs the least rounds x (batches + 1).  At config 2: 8 batches =
    // 808 workgroups = 4 rounds on 256 CUs (36), 9 batches = 720 workgroups = 3 rounds (30): 3.31 -> 3.27 ms per solve, and the measured
    // order of 6..12 follows the model.  SSFM_TASK_BATCHES overrides.
    int task_batches = 8;
    const char* e_tb = std::getenv("SSFM_TASK_BATCHES");
    if (e_tb && std::atoi(e_tb) > 0) task_batches = std::atoi(e_tb);
    else {
        int64_t best = -1;
        for (int tb = 6; tb <= 12; tb++) {
            int64_t tasks = 0;
            for (int c = 0; c < Nc; c++) tasks += (F.cam_batch_ptr[c + 1] - F.cam_batch_ptr[c] + tb - 1) / tb;
            const int64_t wgs = (tasks + 3) / 4, rounds = (wgs + num_cus - 1) / std::max(1, num_cus), cost = rounds * (tb + 1);
            if (best < 0 || cost < best) { best = cost; task_batches = tb; }
        }
    }
    for (int c = 0; c < Nc; c++)
        for (int b2 = F.cam_batch_ptr[c]; b2 < F.cam_batch_ptr[c + 1]; b2 += task_batches) {
            F.chunk_cam.push_back(c); F.chunk_b0.push_back(b2); F.chunk_b1.push_back(std::min(b2 + task_batches, F.cam_batch_ptr[c + 1]));
        }
    F.pair_batches = nbatch_total; F.task_batches = task_batches;
    return nbatch_total;
}
inline void pair_fill_host(BAFlat& F, int NT, std::vector<int64_t>& slot_off) {
    const int Nc = F.Nc;
    F.pair_j.assign((size_t)F.pair_batches * 64, -1); F.pair_j2.assign((size_t)F.pair_batches * 64, -1); F.pair_p.assign((size_t)F.pair_batches * 64, -1);
    parallel_chunks(Nc, NT, [&](int, int64_t c0, int64_t c1) {
        std::vector<int> slot_of(Nc, -1);
        for (int c = (int)c0; c < (int)c1; c++) {
            const int rb = F.row_ptr[c], nnb = F.row_ptr[c + 1] - rb, pc = F.cam_pos[c];
            for (int e = 0; e < nnb; e++) slot_of[F.col_idx[rb + e]] = e;
            for (int q = F.cam_start[c]; q < F.cam_start[c + 1]; q++) {
                const int j = F.cam_obs[q], p = F.cam_obs_pt[q];
                if (!F.pt_grouped.empty() && F.pt_grouped[p]) continue;
                for (int j2 = F.pt_start[p]; j2 < F.pt_start[p + 1]; j2++) {
                    const int c2 = F.obs_cam[j2]; if (!(F.cam_pos[c2] < pc)) continue;
                    const int64_t w = slot_off[rb + slot_of[c2]]++;
                    F.pair_j[w] = j; F.pair_j2[w] = j2; F.pair_p[w] = p;
                }
            }
            for (int e = 0; e < nnb; e++) slot_of[F.col_idx[rb + e]] = -1;
        }
    });
}

// Host-side recycling of the per-observation arrays of a plan.  A first call on a structure allocates ~35 MB of vectors; fresh allocations of
// that size are mmap'ed, and touching them for the first time costs a page fault per 4 KB (~1 ms of the "emit observations" stage at 600k
// observations).  ssfm_ba_destroy hands the arrays of the dying handle to this stash and the next ba_flatten takes them back, already mapped.
struct HostStash {
    std::mutex m;
    raw_vector<int> obs_cam, obs_pt, pt_ids; raw_vector<int64_t> obs_orig; raw_vector<double> obs_xy, pts0, mask_pt;
};
inline HostStash& host_stash() { static HostStash s; return s; }
template <class V> inline void stash_swap_if_larger(V& mine, V& stashed) { if (mine.capacity() > stashed.capacity()) mine.swap(stashed); }
inline void stash_take(BAFlat& F) {
    HostStash& S = host_stash(); std::lock_guard<std::mutex> g(S.m);
    F.obs_cam.swap(S.obs_cam); F.obs_pt.swap(S.obs_pt); F.pt_ids.swap(S.pt_ids); F.obs_orig.swap(S.obs_orig); F.obs_xy.swap(S.obs_xy); F.pts0.swap(S.pts0); F.mask_pt.swap(S.mask_pt);
}
// Bounded: a plan whose per-observation arrays exceed HOST_STASH_MAX_BYTES (the 12 M-observation problems: ~700 MB) is not kept, and whatever is kept is
// released by ssfm_ctx_destroy (host_stash_clear) -- the stash exists for the few-ms first calls of config-2-sized problems, not for the big ones.
constexpr size_t HOST_STASH_MAX_BYTES = (size_t)96 << 20;
inline void host_stash_clear() {
    HostStash& S = host_stash(); std::lock_guard<std::mutex> g(S.m);
    raw_vector<int>().swap(S.obs_cam); raw_vector<int>().swap(S.obs_pt); raw_vector<int>().swap(S.pt_ids); raw_vector<int64_t>().swap(S.obs_orig);
    raw_vector<double>().swap(S.obs_xy); raw_vector<double>().swap(S.pts0); raw_vector<double>().swap(S.mask_pt);
}
inline void stash_give(BAFlat& F) {
    const size_t bytes = (F.obs_cam.capacity() + F.obs_pt.capacity() + F.pt_ids.capacity()) * sizeof(int) + F.obs_orig.capacity() * sizeof(int64_t)
                       + (F.obs_xy.capacity() + F.pts0.capacity() + F.mask_pt.capacity()) * sizeof(double);
    if (bytes > HOST_STASH_MAX_BYTES) return;
    HostStash& S = host_stash(); std::lock_guard<std::mutex> g(S.m);
    stash_swap_if_larger(F.obs_cam, S.obs_cam); stash_swap_if_larger(F.obs_pt, S.obs_pt); stash_swap_if_larger(F.pt_ids, S.pt_ids); stash_swap_if_larger(F.obs_orig, S.obs_orig);
    stash_swap_if_larger(F.obs_xy, S.obs_xy); stash_swap_if_larger(F.pts0, S.pts0); stash_swap_if_larger(F.mask_pt, S.mask_pt);
}

// after_emit (optional): called as soon as this rank's observations and points are laid out (F.obs_*, F.pt_start, F.pts0, F.mask_pt, F.pt_ids are final),
// before the structure of S, the ordering and the task tables are planned -- ba_create_impl starts uploading them on a second host thread there.
inline void ba_flatten(const ssfm_ba_problem& P, int nranks, int rank, BAFlat& F, bool host_pairs = true, int num_cus = 256,
                       const std::function<void(BAFlat&)>* after_emit = nullptr) {
    const bool timing = std::getenv("SSFM_PLAN_TIMING") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) { if (!timing) return; const auto t = std::chrono::steady_clock::now();
                                       std::fprintf(stderr, "[plan] %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t - t_last).count()); t_last = t; };
    const int Nc = P.num_cameras, Np = P.num_points;
    const int64_t M = P.num_observations;
    const int NT = planner_threads();
    F = BAFlat();
    stash_take(F);
    F.Nc = Nc; F.focal_free = !P.focal_fixed;
    if (Nc == 0 || Np == 0 || M == 0) { F.nothing_to_do = true; return; }
    // ---- order observations point-major / camera-ascending (skip the sort if they already are), and
    // ---- pass 1: which points are used (global), with their observation counts.  Chunks of the observation array, cut at point boundaries.
    // The order check rides along with the first attempt at pass 1 (one sweep over the index arrays instead of two: 0.25 ms at 600k observations);
    // an unsorted input is sorted and the pass repeated.
    bool sorted = true;
    std::vector<int64_t> order;
    auto at = [&](int64_t i) { return sorted ? i : order[i]; };
    struct Seg { int pt; int64_t begin, end; int nobs; int first_cam; uint64_t sig; };      // sig: hash of the (deduplicated) camera list, for the signature sort below
    std::vector<Seg> segs;
    auto build_segments = [&](bool check_order) -> bool {
        const int TS = (M >= 200000) ? NT : 1;
        std::vector<std::vector<Seg>> part(TS);
        std::vector<char> bad(TS, 0);
        parallel_chunks(M, TS, [&](int t, int64_t i0, int64_t i1) {
            const int tt = (TS == 1) ? 0 : t;
            if (check_order) {                                            // this chunk's entries against their predecessors (the first one against the previous chunk's last)
                for (int64_t i = std::max<int64_t>(i0, 1); i < i1; i++)
                    if (P.obs_pt[i] < P.obs_pt[i - 1] || (P.obs_pt[i] == P.obs_pt[i - 1] && P.obs_cam[i] <= P.obs_cam[i - 1])) { bad[tt] = 1; return; }
            }
            // a chunk owns the points that START inside it
            while (i0 > 0 && i0 < M && P.obs_pt[at(i0)] == P.obs_pt[at(i0 - 1)]) i0++;
            std::vector<Seg>& out = part[tt]; out.reserve((size_t)((i1 - i0) / 4 + 16));
            for (int64_t i = i0; i < i1;) {
                const int p = P.obs_pt[at(i)];
                int64_t e = i; int nobs = 0; int last_cam = -1, first_cam = -1; uint64_t sig = 0x9E3779B97F4A7C15ull;
                while (e < M && P.obs_pt[at(e)] == p) { const int c = P.obs_cam[at(e)];
                    if (c != last_cam && c >= 0 && c < Nc) { nobs++; last_cam = c; if (first_cam < 0) first_cam = c; sig = (sig ^ (uint64_t)(c + 1)) * 0xFF51AFD7ED558CCDull; sig ^= sig >> 29; }
                    e++; }
                bool valid = p >= 0 && p < Np && nobs >= 3;
                if (valid) { const double* X = &P.points[(size_t)p * 3]; valid = (X[0] * X[0] + X[1] * X[1] + X[2] * X[2]) != 0.0; }
                if (valid) out.push_back({p, i, e, nobs, first_cam, sig});
                i = e;
            }
        });
        for (char c : bad) if (c) return false;
        size_t tot = 0; for (auto& v : part) tot += v.size();
        segs.clear(); segs.reserve(tot);
        for (auto& v : part) segs.insert(segs.end(), v.begin(), v.end());
        return true;
    };
    if (!build_segments(true)) {
        sorted = false;
        order.resize(M); std::iota(order.begin(), order.end(), (int64_t)0);
        std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) {
            if (P.obs_pt[a] != P.obs_pt[b]) return P.obs_pt[a] < P.obs_pt[b];
            return P.obs_cam[a] < P.obs_cam[b]; });
        (void)build_segments(false);
    }
    lap("sorted check + segments");
    F.nP_global = (int)segs.size();
    for (const Seg& s : segs) F.M_global += s.nobs;
    if (segs.empty()) { F.nothing_to_do = true; return; }
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
    // ---- signature sort (round 4).  k_schur_gram takes runs of CONSECUTIVE local points with the same camera list; the order of the local points is the planner's
    // own (pt_ids maps back), so when the caller's order leaves points of equal signature apart -- build_sfm issues point ids in match order
    // (examples/spherical_sfm_tools.cpp:862-955): tracks that start in the same frame get neighbouring ids whatever their length -- this rank's points are
    // re-ordered by (first camera, track length, camera-list hash), stable in the caller's order: equal lists become adjacent and neighbours in the order still
    // share cameras (the XCD-contiguous task order relies on that).  Counting sort by first camera, then every bucket on its own planner thread: ~0.3 ms at 100k
    // points.  Skipped when the caller's order already groups >= 90 % of the points that could be grouped (the synthetic circles: 100 %).  SSFM_GRAM_SORT=0 / 1 forces it.
    {
        const char* e_gs = std::getenv("SSFM_GRAM_SORT");
        const int force = e_gs ? std::atoi(e_gs) : -1;
        bool do_sort = force == 1;
        if (force < 0 && s1 - s0 >= 2 * (size_t)GRAM_MIN_RUN) {
            // (a heuristic: every planner thread looks at its own chunk, a run that crosses a chunk border counts as two)
            std::vector<int64_t> could_t(NT, 0), adj_t(NT, 0);
            parallel_chunks((int64_t)(s1 - s0), NT, [&](int t, int64_t q0, int64_t q1) {
                int64_t could = 0, adjacent = 0;
                for (size_t k = s0 + (size_t)q0, end = s0 + (size_t)q1; k < end;) {
                    size_t e = k + 1; while (e < end && segs[e].sig == segs[k].sig && segs[e].nobs == segs[k].nobs) e++;
                    if (segs[k].nobs <= GRAM_KMAX) { could += (int64_t)(e - k); if (e - k >= (size_t)GRAM_MIN_RUN) adjacent += (int64_t)(e - k); }
                    k = e;
                }
                could_t[t] += could; adj_t[t] += adjacent;
            });
            int64_t could = 0, adjacent = 0; for (int t = 0; t < NT; t++) { could += could_t[t]; adjacent += adj_t[t]; }
            do_sort = could > 0 && adjacent * 10 < could * 9;
        }
        if (do_sort) {
            std::vector<int> bucket_ptr(Nc + 1, 0);
            for (size_t k = s0; k < s1; k++) bucket_ptr[segs[k].first_cam + 1]++;
            for (int c = 0; c < Nc; c++) bucket_ptr[c + 1] += bucket_ptr[c];
            std::vector<Seg> tmp(s1 - s0);
            { std::vector<int> fill(bucket_ptr.begin(), bucket_ptr.end() - 1); for (size_t k = s0; k < s1; k++) tmp[fill[segs[k].first_cam]++] = segs[k]; }
            parallel_chunks(Nc, NT, [&](int, int64_t c0, int64_t c1) {
                for (int c = (int)c0; c < (int)c1; c++)
                    std::stable_sort(tmp.begin() + bucket_ptr[c], tmp.begin() + bucket_ptr[c + 1],
                                     [](const Seg& a, const Seg& b2) { return a.nobs != b2.nobs ? a.nobs < b2.nobs : a.sig < b2.sig; });
            });
            // adopt the new order only if it does produce groups: at least half of the points that could be grouped now sit in runs of GRAM_MIN_RUN (signatures
            // with fewer members than that gain nothing from being adjacent, and the caller's order is kept)
            int64_t could = 0, adjacent = 0;
            for (size_t k = 0; k < tmp.size();) {
                size_t e = k + 1; while (e < tmp.size() && tmp[e].sig == tmp[k].sig && tmp[e].nobs == tmp[k].nobs) e++;
                if (tmp[k].nobs <= GRAM_KMAX) { could += (int64_t)(e - k); if (e - k >= (size_t)GRAM_MIN_RUN) adjacent += (int64_t)(e - k); }
                k = e;
            }
            if (force == 1 || adjacent * 2 >= could) { std::copy(tmp.begin(), tmp.end(), segs.begin() + s0); F.gram_sorted = true; }
        }
        lap("signature sort");
    }
    // ---- pass 2: emit local observations
    F.nP = (int)(s1 - s0);
    {
        int64_t mloc = 0; for (size_t k = s0; k < s1; k++) mloc += segs[k].nobs;
        F.pt_ids.resize(F.nP); F.pt_start.resize((size_t)F.nP + 1); F.pts0.resize((size_t)F.nP * 3); F.mask_pt.resize((size_t)F.nP * 3);
        F.obs_cam.resize(mloc); F.obs_pt.resize(mloc); F.obs_orig.resize(mloc); F.obs_xy.resize((size_t)mloc * 2);
        F.pt_start[0] = 0;
        for (size_t k = s0; k < s1; k++) F.pt_start[k - s0 + 1] = F.pt_start[k - s0] + segs[k].nobs;
        parallel_chunks((int64_t)(s1 - s0), NT, [&](int, int64_t q0, int64_t q1) {
            for (int64_t q = q0; q < q1; q++) {
                const Seg& sg = segs[s0 + (size_t)q];
                int64_t w = F.pt_start[q];
                for (int64_t i = sg.begin; i < sg.end; i++) {
                    const int64_t o = at(i); const int c = P.obs_cam[o];
                    if (c < 0 || c >= Nc) continue;
                    if (i + 1 < sg.end && P.obs_cam[at(i + 1)] == c) continue;     // keep the last duplicate
                    F.obs_cam[w] = c; F.obs_pt[w] = (int)q; F.obs_orig[w] = o; F.obs_xy[2 * w] = P.obs_xy[2 * o]; F.obs_xy[2 * w + 1] = P.obs_xy[2 * o + 1]; w++;
                }
                const double m = (P.pt_fixed && P.pt_fixed[sg.pt]) ? 0.0 : 1.0;
                for (int d = 0; d < 3; d++) { F.pts0[(size_t)q * 3 + d] = P.points[(size_t)sg.pt * 3 + d]; F.mask_pt[(size_t)q * 3 + d] = m; }
                F.pt_ids[q] = sg.pt;
            }
        });
    }
    lap("emit observations");
    if (after_emit) (*after_emit)(F);
    // ---- structure of S and camera activity come from ALL used points (identical on every rank)
    std::vector<char> cam_in(Nc, 0);
    if (Nc <= 1024) {
        // few cameras: the block structure is an Nc x Nc bit matrix (11 KB at 300 cameras).  Every planner thread marks the camera pairs of its
        // share of the points in a private copy; the copies are OR-ed and the rows read off in ascending order.
        const int WPR = (Nc + 63) / 64;                                   // 64-bit words per row
        const int TB = (segs.size() >= 20000) ? NT : 1;
        std::vector<std::vector<uint64_t>> bits(TB, std::vector<uint64_t>((size_t)Nc * WPR, 0));
        parallel_chunks((int64_t)segs.size(), TB, [&](int t, int64_t k0, int64_t k1) {
            std::vector<uint64_t>& B = bits[TB == 1 ? 0 : t];
            int cams[64];
            for (int64_t k = k0; k < k1; k++) {
                const Seg& sg = segs[(size_t)k]; int n = 0, last = -1; bool overflow = false;
                for (int64_t i = sg.begin; i < sg.end; i++) { const int c = P.obs_cam[at(i)]; if (c != last && c >= 0 && c < Nc) { if (n < 64) cams[n++] = c; else overflow = true; last = c; } }
                if (overflow) {                                           // a track longer than 64 cameras: mark pair by pair from the array itself
                    int la = -1;
                    for (int64_t i = sg.begin; i < sg.end; i++) { const int a = P.obs_cam[at(i)]; if (a == la || a < 0 || a >= Nc) continue; la = a; int lb = -1;
                        for (int64_t i2 = sg.begin; i2 < sg.end; i2++) { const int b2 = P.obs_cam[at(i2)]; if (b2 == lb || b2 < 0 || b2 >= Nc) continue; lb = b2; B[(size_t)a * WPR + (b2 >> 6)] |= 1ull << (b2 & 63); } }
                    continue;
                }
                for (int a = 0; a < n; a++) { uint64_t* row = &B[(size_t)cams[a] * WPR]; for (int b2 = 0; b2 < n; b2++) row[cams[b2] >> 6] |= 1ull << (cams[b2] & 63); }
            }
        });
        std::vector<uint64_t>& B0 = bits[0];
        for (int t = 1; t < TB; t++) for (size_t w = 0; w < B0.size(); w++) B0[w] |= bits[t][w];
        F.row_ptr.assign(Nc + 1, 0); F.diag_slot.assign(Nc, -1); F.col_idx.clear(); F.col_idx.reserve((size_t)Nc * 16);
        for (int c = 0; c < Nc; c++) {
            const size_t first = F.col_idx.size();
            for (int wd = 0; wd < WPR; wd++) { uint64_t x = B0[(size_t)c * WPR + wd]; while (x) { const int bit = __builtin_ctzll(x); x &= x - 1; F.col_idx.push_back(64 * wd + bit); } }
            if (F.col_idx.size() == first) F.col_idx.push_back(c);         // isolated camera: identity row
            else cam_in[c] = 1;
            F.diag_slot[c] = (int)(std::lower_bound(F.col_idx.begin() + first, F.col_idx.end(), c) - (F.col_idx.begin() + first));
            F.row_ptr[c + 1] = (int)F.col_idx.size();
            F.max_row_blocks = std::max(F.max_row_blocks, (int)(F.col_idx.size() - first));
        }
    } else {
        // camera-major view of the used observations (global), then per camera one sweep over its points' camera lists with a
        // "last row that saw this column" mark: O(M K) without sorting M K entries
        std::vector<int> gstart(Nc + 1, 0);
        std::vector<int64_t> seg_first(segs.size() + 1, 0);           // compacted (deduplicated) camera lists per used point
        std::vector<int> seg_cams; seg_cams.reserve((size_t)F.M_global);
        for (size_t k = 0; k < segs.size(); k++) {
            const Seg& sg = segs[k]; int last = -1;
            for (int64_t i = sg.begin; i < sg.end; i++) { const int c = P.obs_cam[at(i)]; if (c != last && c >= 0 && c < Nc) { seg_cams.push_back(c); gstart[c + 1]++; cam_in[c] = 1; last = c; } }
            seg_first[k + 1] = (int64_t)seg_cams.size();
        }
        for (int c = 0; c < Nc; c++) gstart[c + 1] += gstart[c];
        std::vector<int> gseg(seg_cams.size());                       // for every camera: the used points (segment ids) it observes
        { std::vector<int> fill(gstart.begin(), gstart.end() - 1);
          for (size_t k = 0; k < segs.size(); k++) for (int64_t i = seg_first[k]; i < seg_first[k + 1]; i++) gseg[fill[seg_cams[i]]++] = (int)k; }
        std::vector<std::vector<int>> rows(Nc);
        parallel_chunks(Nc, NT, [&](int, int64_t c0, int64_t c1) {
            std::vector<int> mark(Nc, -1);
            for (int c = (int)c0; c < (int)c1; c++) {
                std::vector<int>& row = rows[c];
                for (int q = gstart[c]; q < gstart[c + 1]; q++) {
                    const int k = gseg[q];
                    for (int64_t i = seg_first[k]; i < seg_first[k + 1]; i++) { const int b2 = seg_cams[i]; if (mark[b2] != c) { mark[b2] = c; row.push_back(b2); } }
                }
                if (row.empty()) row.push_back(c);          // isolated camera: identity row
                std::sort(row.begin(), row.end());
            }
        });
        F.row_ptr.assign(Nc + 1, 0); F.diag_slot.assign(Nc, -1); F.col_idx.clear(); F.col_idx.reserve((size_t)Nc * 16);
        for (int c = 0; c < Nc; c++) {
            const std::vector<int>& row = rows[c];
            F.diag_slot[c] = (int)(std::lower_bound(row.begin(), row.end(), c) - row.begin());
            F.col_idx.insert(F.col_idx.end(), row.begin(), row.end());
            F.row_ptr[c + 1] = (int)F.col_idx.size();
            F.max_row_blocks = std::max(F.max_row_blocks, (int)row.size());
        }
    }
    lap("S structure");
    {   // camera block size first: it decides whether pairs of cameras are merged in the band
        bool all_t_fixed = true;
        for (int c = 0; c < Nc; c++) if (cam_in[c] && !(P.trans_fixed && P.trans_fixed[c])) { all_t_fixed = false; break; }
        F.DC = all_t_fixed ? 3 : 6;
    }
    band_plan(Nc, F.DC, F.row_ptr, F.col_idx, F.cam_pos, F.band, F.comp_ptr, F.band_row, F.band_row2, F.comp_twist, F.band_rows, F.band_block, F.pair_dummy, &F.rings);
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
        if (!F.rings.empty()) ring_wrap_table(Nc, F.row_ptr, F.col_idx, F.band_row, F.band_row2, F.band, F.band_block == 6 && F.DC == 3 ? 2 : 1, F.wrap_ptr, F.wrap_blk, F.wrap_row2);
    }
    lap("ordering + lower triangle");
    F.mask_cam.assign((size_t)Nc * 6, 0.0);
    bool all_t_fixed = true;
    for (int c = 0; c < Nc; c++) {
        if (!cam_in[c]) continue;
        const bool tf = P.trans_fixed && P.trans_fixed[c], rf = P.rot_fixed && P.rot_fixed[c];
        if (!tf) { all_t_fixed = false; for (int k = 0; k < 3; k++) F.mask_cam[c * 6 + k] = 1.0; }
        if (!rf) for (int k = 0; k < 3; k++) F.mask_cam[c * 6 + 3 + k] = 1.0;
    }
    F.DC = all_t_fixed ? 3 : 6;
    F.M = (int64_t)F.obs_cam.size();
    F.cam_start.assign(Nc + 1, 0);
    for (int64_t j = 0; j < F.M; j++) F.cam_start[F.obs_cam[j] + 1]++;
    for (int c = 0; c < Nc; c++) F.cam_start[c + 1] += F.cam_start[c];
    if (host_pairs) {                                                  // otherwise filled on the device (k_cam_lists), like the pair lists
        F.cam_obs.resize(F.M);
        { std::vector<int> fill(F.cam_start.begin(), F.cam_start.end() - 1);
          for (int64_t j = 0; j < F.M; j++) F.cam_obs[fill[F.obs_cam[j]]++] = (int)j; }
        F.cam_obs_pt.resize(F.M);
        for (int64_t q = 0; q < F.M; q++) F.cam_obs_pt[q] = F.obs_pt[F.cam_obs[q]];
    }
    // entries per wave task of k_cam_sums2: 192..512, whichever needs the least rounds of workgroups (4 tasks each) over the CUs x
    // (batches + 2 for the fold and its atomics) -- the same reasoning as pair_layout; 384 at config 2 (450 workgroups = 2 rounds
    // instead of 600 = 3 with 256).  SSFM_CS_TASK_OBS overrides.
    int cs_run = 256;
    const char* e_cs = std::getenv("SSFM_CS_TASK_OBS");
    if (e_cs && std::atoi(e_cs) > 0) cs_run = std::max(64, std::atoi(e_cs) / 64 * 64);
    else {
        int64_t best = -1;
        for (int run = 192; run <= 512; run += 64) {
            int64_t tasks = 0;
            for (int c = 0; c < Nc; c++) tasks += (F.cam_start[c + 1] - F.cam_start[c] + run - 1) / run;
            const int64_t wgs = (tasks + 3) / 4, rounds = (wgs + num_cus - 1) / std::max(1, num_cus), cost = rounds * (run / 64 + 2);
            if (best < 0 || cost < best) { best = cost; cs_run = run; }
        }
    }
    lap("camera-major lists");
    // ---- signature groups for k_schur_gram / k_gram_backsub (see BAFlat::gr_rec): runs of consecutive points with identical camera lists
    {
        const bool gram_on = !(std::getenv("SSFM_GRAM") && std::atoi(std::getenv("SSFM_GRAM")) == 0);                               // read per plan: tests switch it
        const int kmin = std::getenv("SSFM_GRAM_KMIN") ? std::max(2, std::atoi(std::getenv("SSFM_GRAM_KMIN"))) : GRAM_KMIN;
        const int min_run = std::getenv("SSFM_GRAM_MIN_RUN") ? std::max(8, std::atoi(std::getenv("SSFM_GRAM_MIN_RUN"))) : GRAM_MIN_RUN;       // shortest run that becomes a wave task
        F.gram_any = knob_env_int("SSFM_GRAM_ANY", 1) != 0;                 // read once per plan: tests switch it between solves; the launch uses this value too
        const int gram_pts_env = std::getenv("SSFM_GRAM_PTS") ? std::max(8, std::atoi(std::getenv("SSFM_GRAM_PTS"))) : 0;           // points per wave task (0: by size)
        F.pt_grouped.resize((size_t)F.nP);
        if (F.nP > 0) std::memset(F.pt_grouped.data(), 0, (size_t)F.nP);
        if (gram_on && F.sym_lower) {
            auto same = [&](int q0, int q1) {
                const int k0 = F.pt_start[q0 + 1] - F.pt_start[q0]; if (F.pt_start[q1 + 1] - F.pt_start[q1] != k0) return false;
                for (int k = 0; k < k0; k++) if (F.obs_cam[F.pt_start[q0] + k] != F.obs_cam[F.pt_start[q1] + k]) return false;
                return true;
            };
            std::vector<int> runs;                                   // (first point, end) of every run that qualifies
            // every planner thread owns the runs that START in its chunk of the points (it walks back to no one: a chunk begins at the first point that differs from
            // its predecessor) and follows its last run across the chunk's end; the per-thread lists are concatenated in order
            {
                const int TR = (F.nP >= 20000) ? NT : 1;
                std::vector<std::vector<int>> runs_t(TR);
                parallel_chunks(F.nP, TR, [&](int t, int64_t q0, int64_t q1) {
                    std::vector<int>& out = runs_t[TR == 1 ? 0 : t];
                    int q = (int)q0;
                    while (q > 0 && q < (int)q1 && same(q - 1, q)) q++;        // the run that reaches into this chunk belongs to the previous one
                    while (q < (int)q1) {
                        int e = q + 1; while (e < F.nP && same(q, e)) e++;
                        const int K = F.pt_start[q + 1] - F.pt_start[q];
                        // (every point the flatten rules keep has >= 3 observations.  K = 3 went through the pair lists until its 18 Gram rows ran as one 16-row tile + a 4x4x4
                        //  tail: 29.7 | 230 us against 31.7 | 246 us at 100k | 1.5 M points, scripts/prof_gram_k.py; SSFM_GRAM_KMIN raises the bound)
                        if (e - q >= min_run && K >= kmin && K <= GRAM_KMAX) { out.push_back(q); out.push_back(e); }
                        q = e;
                    }
                });
                for (auto& v : runs_t) runs.insert(runs.end(), v.begin(), v.end());
                for (size_t r = 0; r < runs.size(); r += 2) { const int K = F.pt_start[runs[r] + 1] - F.pt_start[runs[r]]; F.gram_points += runs[r + 1] - runs[r]; F.gram_obs += (int64_t)(runs[r + 1] - runs[r]) * K; }
            }
            // Round 4: do the groups pay?  k_schur_gram is launched once per tile class (rows = DC K: {3}, {4, 5}, {6}, {7, 8} at 6 dof), and a launch of a few hundred
            // wave tasks costs its ~24 us latency floor whatever it holds: 300 cameras / 600k observations with tracks of 3..8 frames, sorted into 1800 signatures
            // of ~60 points, took 4 x 25 us through the groups against 44 + 21 us through the pair lists (scripts/prof_irregular.py, profiles/r04_notes.md).
            // Estimated launch times from the measurements of rounds 3-4 (us): a class max(24, 0.45e-3 points K / 6) (0.27e-3 beyond 300k points); pair lists max(36, 22e-6 pairs) + camera sums
            // max(20, 34.5e-6 observations).  The groups are kept when their estimate (+ the pair path for what stays loose) is below the pair path for everything.
            // SSFM_GRAM_MODEL=0 keeps every group that qualifies.
            if (!(std::getenv("SSFM_GRAM_MODEL") && std::atoi(std::getenv("SSFM_GRAM_MODEL")) == 0) && !runs.empty()) {
                double cls_pts[4 * 2 * 8] = {0}; double all_pairs = 0, loose_pairs = 0;
                { std::vector<double> part(NT, 0.0);
                  parallel_chunks(F.nP, NT, [&](int t, int64_t q0, int64_t q1) { double a = 0; for (int64_t q = q0; q < q1; q++) { const double k = F.pt_start[q + 1] - F.pt_start[q]; a += 0.5 * k * (k - 1); } part[t] += a; });
                  for (double v : part) all_pairs += v; }
                double grouped_pairs = 0;
                for (size_t r = 0; r < runs.size(); r += 2) {
                    const int K = F.pt_start[runs[r] + 1] - F.pt_start[runs[r]], rows = F.DC * K, nt_full = rows / 16, tail = rows - 16 * nt_full;
                    const int cls = (tail > 0 && tail <= 4 && nt_full >= 1) ? 8 + nt_full : (rows + 15) / 16;      // tile class = launch (ba_solver.hip)
                    cls_pts[cls] += (double)(runs[r + 1] - runs[r]) * K / 6.0;
                    grouped_pairs += 0.5 * K * (K - 1) * (runs[r + 1] - runs[r]);
                }
                loose_pairs = all_pairs - grouped_pairs;
                // per point: 0.45 ns in the latency-bound regime of one round of tasks (config 2: 100k points, 45 us), 0.27 ns once a class fills the chip for several
                // rounds (configs[4] size: 1.5 M points of 8 cameras, 391 us)
                // round 5: the classes share ONE launch (k_schur_gram_any) unless SSFM_GRAM_ANY=0 -- the latency floor is paid once
                const bool any_launch = F.gram_any;
                double est_gram = 0; int n_cls = 0;
                for (double v : cls_pts) if (v > 0) { est_gram += std::max(24.0, (v > 3e5 ? 0.27e-3 : 0.45e-3) * v); n_cls++; }
                if (any_launch && n_cls > 1) est_gram *= 0.75;      // (only when several classes really share the launch: single-class problems run the specialised kernel)     // measured: tracks 3 ... 8, 600k observations, 2336 tasks of ~47 points: 70 us in one launch against 4 x 23.6 (pair lists: 43 + 20)
                const double obs_all = (double)F.pt_start[F.nP];
                const double est_loose = loose_pairs > 0 ? std::max(36.0, 22e-6 * loose_pairs) + std::max(20.0, 34.5e-6 * obs_all) : 0.0;
                const double est_pairs_only = std::max(36.0, 22e-6 * all_pairs) + std::max(20.0, 34.5e-6 * obs_all);
                if (timing) std::fprintf(stderr, "[plan] group model: groups %.0f us + loose %.0f us against pair lists only %.0f us\n", est_gram, est_loose, est_pairs_only);
                if (est_gram + est_loose > est_pairs_only) { runs.clear(); F.gram_points = 0; F.gram_obs = 0; }
            }
            // points per wave task: a task pays ~5 us of start-up (index loads, camera records, the atomics of its blocks at the end) whatever its length, and
            // the chip holds ~2300 of these waves at once: two rounds' worth of tasks when the problem is small, 192 points when it is large
            // (measured: 100 000 points 64 -> 39.5 us, 128 -> 41, 192 -> 54; 1.5 M points 64 -> 560 us, 128 -> 453, 192 -> 434)
            int gram_pts = gram_pts_env;
            if (gram_pts <= 0) {
                gram_pts = (int)std::min<int64_t>(192, std::max<int64_t>(64, (F.gram_points / 4608 + 15) / 16 * 16));
                // round 4: a small problem runs fastest with ONE wave per SIMD -- the shortest task length (64..192) that leaves at most 4 x CUs tasks, runs cut in equal
                // parts.  Config 2 (300 runs of 333 points): 3 x 112 = 900 tasks 43.9 us; 6 x 56 = 1800 tasks (two waves on three of four SIMDs) 48.2; the
                // 5 x 64 + 13 of round 3 46.1; 2 x 168 = 600 tasks 49.6 (scripts/prof_gram_ld.py with SSFM_GRAM_PTS, profiles/r04_notes.md)
                for (int g = 64; g <= 192; g += 8) {
                    int64_t tasks = 0; for (size_t r = 0; r < runs.size(); r += 2) tasks += (runs[r + 1] - runs[r] + g - 1) / g;
                    if (tasks <= 4 * (int64_t)num_cus) { gram_pts = g; break; }
                }
            }
            // tasks sorted by K (stable: point order within a K): k_schur_gram is launched once per tile count, over a contiguous range of tasks
            std::vector<size_t> run_order(runs.size() / 2);
            for (size_t r = 0; r < run_order.size(); r++) run_order[r] = 2 * r;
            auto run_K = [&](size_t r) { return F.pt_start[runs[r] + 1] - F.pt_start[runs[r]]; };
            std::stable_sort(run_order.begin(), run_order.end(), [&](size_t a, size_t b) { return run_K(a) < run_K(b); });
            for (size_t r : run_order) {
                const int q = runs[r], e = runs[r + 1], K = F.pt_start[q + 1] - F.pt_start[q];
                const int* cams = &F.obs_cam[F.pt_start[q]];
                int slots[GRAM_NPAIR]; for (int i = 0; i < GRAM_NPAIR; i++) slots[i] = 0;
                for (int a = 1; a < K; a++) for (int b = 0; b < a; b++) {
                    const int ca = cams[a], cb = cams[b];
                    const bool row_a = F.cam_pos[ca] > F.cam_pos[cb];           // the stored block sits in the row of the camera eliminated later
                    const int row = row_a ? ca : cb, col = row_a ? cb : ca;
                    const int sl = (int)(std::lower_bound(F.col_idx.begin() + F.row_ptr[row], F.col_idx.begin() + F.row_ptr[row + 1], col) - F.col_idx.begin());
                    slots[a * (a - 1) / 2 + b] = sl | (row_a ? 0 : (1 << 30));
                }
                // a run is cut into EQUAL tasks (whole sub-chunks of 8 points) instead of full tasks + a short remainder: config 2's runs of 333 points were
                // 5 x 64 + 13 -- a sixth task that pays the full start-up and emission for two sub-chunks; now 5 x 56 + 53
                const int parts = (e - q + gram_pts - 1) / gram_pts, per = (((e - q + parts - 1) / parts) + GRAM_SUB_PTS - 1) / GRAM_SUB_PTS * GRAM_SUB_PTS;
                for (int t0 = q; t0 < e; t0 += per) {
                    const int head[4] = {t0, std::min(per, e - t0), K, F.pt_start[t0]};
                    F.gr_rec.insert(F.gr_rec.end(), head, head + 4);
                    for (int k = 0; k < GRAM_KMAX; k++) F.gr_rec.push_back(k < K ? cams[k] : cams[0]);
                    F.gr_rec.insert(F.gr_rec.end(), slots, slots + GRAM_NPAIR);
                    for (int k = 0; k < GRAM_KMAX; k++) { const int c = k < K ? cams[k] : cams[0]; F.gr_rec.push_back(F.row_ptr[c] + F.diag_slot[c]); }
                }
                std::memset(F.pt_grouped.data() + q, 1, (size_t)(e - q));
            }
        }
    }
    // wave tasks of k_cam_sums2: cameras all of whose points sit in signature groups have none (k_schur_gram has their sums); a camera with other points
    // keeps its whole list, and the kernel gives the grouped points among them a zero weight
    {
        std::vector<int> loose(Nc, F.gram_points == 0 ? 1 : 0);
        if (F.gram_points > 0 && F.gram_points < F.nP)
            for (int q = 0; q < F.nP; q++) if (!F.pt_grouped[q]) for (int j = F.pt_start[q]; j < F.pt_start[q + 1]; j++) loose[F.obs_cam[j]] = 1;
        for (int c = 0; c < Nc; c++)
            if (loose[c])
                for (int q = F.cam_start[c]; q < F.cam_start[c + 1]; q += cs_run) { F.cs_task_cam.push_back(c); F.cs_task_q0.push_back(q); F.cs_task_q1.push_back(std::min(q + cs_run, F.cam_start[c + 1])); }
    }
    if (timing) std::fprintf(stderr, "[plan] signature groups: %lld of %d points in %zu tasks\n", (long long)F.gram_points, F.nP, F.gr_rec.size() / GRAM_REC);
    lap("signature groups");
    // ---- fold lists of the atomics-free Gram emission: only when the signature groups hold every point (otherwise the pair kernels add into the same accumulators).
    // One list per "extended slot": the slots of row c keep their order and the row gets one more entry behind them for the camera's vectors -- extended id of slot s
    // of row c = s + c, of the vectors of camera c = row_ptr[c + 1] + c -- so that everything k_finalize_gather's workgroup c folds is ONE contiguous stretch of sources.
    F.gpart_off.clear(); F.fold_slot_ptr.clear(); F.fold_slot_src.clear();
    if (F.nP > 0 && F.gram_points == (int64_t)F.nP && F.cs_task_cam.empty()) {
        const int ng = (int)(F.gr_rec.size() / GRAM_REC), DC = F.DC, BB = DC * DC;
        const size_t nnzb = F.col_idx.size(), next = nnzb + Nc;
        std::vector<int> row_of(nnzb); for (int c = 0; c < Nc; c++) for (int e = F.row_ptr[c]; e < F.row_ptr[c + 1]; e++) row_of[e] = c;
        auto ext_slot = [&](int sl) { return sl + row_of[sl]; };
        auto ext_cam = [&](int c) { return F.row_ptr[c + 1] + c; };
        F.gpart_off.assign(ng + 1, 0);
        std::vector<int> cnt(next + 1, 0);
        int64_t tot = 0;
        for (int t = 0; t < ng; t++) {
            const int* rec = &F.gr_rec[(size_t)t * GRAM_REC]; const int K = rec[2];
            F.gpart_off[t] = (int)tot; tot += gram_part_len(K, DC);
            for (int a = 0; a < K; a++) { cnt[ext_slot(rec[40 + a]) + 1]++; cnt[ext_cam(rec[4 + a]) + 1]++; for (int b2 = 0; b2 < a; b2++) cnt[ext_slot(rec[12 + a * (a - 1) / 2 + b2] & 0x3fffffff) + 1]++; }
        }
        F.gpart_off[ng] = (int)tot;
        bool fits = tot < ((int64_t)1 << 30);
        for (size_t e = 0; e < next; e++) cnt[e + 1] += cnt[e];
        for (int c = 0; c < Nc && fits; c++) {
            const int x0 = F.row_ptr[c] + c, nx = F.row_ptr[c + 1] - F.row_ptr[c] + 1;
            if (nx + 1 > GRAM_FOLD_PTRS || cnt[x0 + nx] - cnt[x0] > GRAM_FOLD_SRCS) fits = false;      // a row with more blocks / more sources than the fixed table holds
        }
        if (fits) {
            // one fixed-stride table per camera row, so that k_finalize_gather's workgroup c needs ONE round of index loads (no row_ptr -> list pointer -> list chain):
            // [0] sources  [1 .. 1 + GRAM_FOLD_PTRS) first source of every extended slot of the row, relative to the row's first  [GRAM_FOLD_HEAD ..) the sources' offsets
            F.fold_slot_ptr.clear();
            F.fold_slot_src.assign((size_t)Nc * GRAM_FOLD_STRIDE, 0);
            std::vector<int> flat(cnt[next], 0), cur(cnt.begin(), cnt.end() - 1);
            for (int t = 0; t < ng; t++) {                                    // tasks in order: the fold adds in task order
                const int* rec = &F.gr_rec[(size_t)t * GRAM_REC]; const int K = rec[2], base = F.gpart_off[t], vb = base + gram_part_blocks(K) * BB;
                for (int a = 0; a < K; a++) {
                    flat[cur[ext_slot(rec[40 + a])]++] = base + (a * (a + 1) / 2 + a) * BB;
                    flat[cur[ext_cam(rec[4 + a])]++] = vb + a * 5 * DC;
                    for (int b2 = 0; b2 < a; b2++) flat[cur[ext_slot(rec[12 + a * (a - 1) / 2 + b2] & 0x3fffffff)]++] = base + (a * (a + 1) / 2 + b2) * BB;
                }
            }
            for (int c = 0; c < Nc; c++) {
                int* row = &F.fold_slot_src[(size_t)c * GRAM_FOLD_STRIDE];
                const int x0 = F.row_ptr[c] + c, nx = F.row_ptr[c + 1] - F.row_ptr[c] + 1, q0 = cnt[x0];
                row[0] = cnt[x0 + nx] - q0;
                for (int j = 0; j <= nx; j++) row[1 + j] = cnt[x0 + j] - q0;
                for (int j = nx + 1; j < GRAM_FOLD_PTRS; j++) row[1 + j] = row[1 + nx];
                for (int q = 0; q < row[0]; q++) row[GRAM_FOLD_HEAD + q] = flat[q0 + q];
            }
        } else F.gpart_off.clear();
    }
    lap("fold lists");
    // ---- Schur pair lists, grouped by (row camera, slot), padded to 64-entry batches
    if (host_pairs) {
        std::vector<int> slot_cnt; pair_counts_host(F, NT, slot_cnt);
        std::vector<int64_t> slot_off; pair_layout(F, slot_cnt, slot_off);
        pair_fill_host(F, NT, slot_off);
    }
    lap("pair lists");
}

}  // namespace ssfm
