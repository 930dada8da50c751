// spherical_sfm_amd -- state shared by the device solvers: the BA handle (also used by the pose-graph solver as the
// container of its reduced system) and the reduced-system solve (banded Cholesky + PCG refinement, or block-Jacobi PCG).
#pragma once
#include <atomic>
#include <hip/hip_runtime.h>
#include <chrono>
#include <cmath>
#include <cstring>
#include <limits>
#include <string>
#include <vector>
#include "ba_flatten.h"
#include "ba_kernels.h"
#include "band_kernels2.h"
#include "band_kernels2p.h"
#include "band_sub.h"
#include "band_ring.h"
#include "snode.h"
#include "ssfm_ctx.h"
#include "knobs.h"

namespace ssfm {

static double wall_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

enum KernelId { KID_CAM_ROT, KID_POINT_LIN, KID_SCHUR_ROWS, KID_FINALIZE, KID_PCG_INIT, KID_PCG_MATVEC, KID_PCG_VECOPS,
                KID_CAM_UPDATE, KID_BACKSUB, KID_COST, KID_ALLREDUCE, KID_BAND_GATHER, KID_BAND_CHOL, KID_BAND_FWD, KID_BAND_BACK,
                KID_BAND_COMBINE, KID_REF_VEC, KID_CAM_SUMS, KID_SUB_SPIKE, KID_SUB_ASM, KID_SUB_CHAIN, KID_SUB_APPLY, KID_SCHUR_GRAM, KID_GRAM_BACKSUB, KID_RING_ELIM, KID_RING_BACK, KID_RING_TAIL, KID_DET_DECODE, KID_SNODE, KID_COUNT };
// names as rocprofv3 prints them (template arguments dropped)
static const char* kKernelNames[KID_COUNT] = {"k_cam_rot", "k_point_lin", "k_schur_pairs2", "k_finalize_gather", "k_pcg_init",
                                              "k_arrow_update", "k_pcg_vecops", "k_cam_update", "k_point_backsub",
                                              "k_point_cost", "rccl_allreduce", "k_band_gather", "k_band_chol_v2", "k_band_fwd_lds",
                                              "k_band_back_v2", "k_arrow_phi", "k_ref_vecops", "k_cam_sums2", "k_sub_spike_fwd",
                                              "k_sub_sep_assemble", "k_sub_sep_chain", "k_sub_apply_left", "k_schur_gram", "k_gram_backsub", "k_ring_cr_elim", "k_ring_cr_back", "k_ring_cr_tail", "k_det_decode", "k_snode_solve"};

// SSFM_PLAN_TIMING: time spent in hipMalloc (atomic: the observation arrays are allocated by the upload thread of ba_create_impl while the main thread plans)
static bool g_alloc_timing = false; static std::atomic<long long> g_alloc_ns{0}; static std::atomic<int> g_alloc_n{0};
template <typename T>
struct DevBuf {
    T* p = nullptr; size_t n = 0; size_t cap_bytes = 0; int dev = 0;
    hipError_t alloc(size_t count) {
        n = count;
        const size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
        const double t0 = g_alloc_timing ? wall_s() : 0.0;
        (void)hipGetDevice(&dev);
        hipError_t e = hipSuccess;
        p = static_cast<T*>(g_dev_pool.take(bytes, dev, &cap_bytes));
        if (!p) {
            e = hipMalloc((void**)&p, bytes); cap_bytes = bytes;
            if (e != hipSuccess) {                    // the pool may be sitting on the memory: give it back to the driver and try once more
                (void)hipGetLastError(); g_dev_pool.drain(dev);
                e = hipMalloc((void**)&p, bytes);
                if (e != hipSuccess) { p = nullptr; cap_bytes = 0; }
            }
        }
        if (g_alloc_timing) { g_alloc_ns += (long long)(1e9 * (wall_s() - t0)); g_alloc_n++; }
        return e;
    }
    void free() {
        if (p && cap_bytes > 0 && g_dev_pool.give(p, cap_bytes, dev)) { p = nullptr; n = 0; cap_bytes = 0; return; }   // recycled (ssfm_ctx.h: DevPool)
        if (p) (void)hipFree(p);
        p = nullptr; n = 0; cap_bytes = 0;
    }
};

}  // namespace ssfm

using namespace ssfm;

struct ssfm_ba_handle {
    ssfm_ctx* ctx = nullptr;
    int device = 0;              // cached at creation: destroy must not read through ctx (Python may finalise the context first)
    ssfm_ba_options opt;
    BAFlat F;
    int n_red = 0;                       // length of the all-reduced assembly buffer
    // device state
    DevBuf<double> cam_x, cam_c, cam_init, pts_x, pts_c, pts_init, focal3;   // focal3: [x, cand, init]
    DevBuf<double> rot_x, rot_c, scale_cam, scale_pt, scale_f, mask_cam, mask_pt, mask_f, diag_cam, diag_pt, diag_f;
    DevBuf<double> obs_xy; DevBuf<int> obs_cam, obs_pt, pt_start, cam_start, cam_obs, row_ptr, col_idx, diag_slot;
    // BA: scal, pcg and redbuf are views into one of two zones; an LM iteration works in one while the memset of the other (for the next
    // iteration) is already queued behind it, off the host's critical path
    DevBuf<double> zone; bool zone_views = false; size_t zone_len = 0, zone_nnz = 0, zone_n = 0;
    // SSFM_DETERMINISTIC=1 (det_acc.h): the assembly adds fixed-point limbs with integer atomics instead of doubles; k_det_decode turns them into the zone's doubles
    bool det = false; DevBuf<long long> det_limb, det_lacc; DevBuf<double> det_dfpart; size_t det_nacc = 0;
    void set_zone(int which) {
        scal.p = zone.p + (size_t)which * zone_len; pcg.p = scal.p + scal.n; redbuf.p = pcg.p + pcg.n;
        S_val = redbuf.p; rhs = S_val + zone_nnz; Udiag = rhs + (zone_n + 1); Sfc = Udiag + zone_n; gcraw = Sfc + zone_n; red_scal = gcraw + zone_n;
    }
    DevBuf<double> lmdev;                 // [go, radius] of the device-side step decision (k_publish -> speculative k_point_lin)
    DevBuf<double> Vs, gp, redbuf, Minv, Sff, px, pr, pz, pp, pq, pqpart, scal, pcg;
    DevBuf<double> band, Linv, Yb, Yr; DevBuf<int> cam_pos, band_pairs, band_fail, comp_ptr;
    // substructured factorisation of long components (band_sub.h); disabled => segments == components
    BandSub sub; DevBuf<int> sub_seg_lo, sub_seg_hi, sub_seg_wend, sub_left, sub_sep_lo, sub_sep_rseg, sub_chain_ptr, sub_tw_lo, sub_tw_hi, sub_tw_copy, sub_seg_given, sub_seg_mode;
    DevBuf<unsigned char> pair_dummy;    // merged 3-dof pairs: 1 = the partner slot of this camera is empty
    DevBuf<int> cam_pos2;                // second band row of the separator cameras of twisted components (-1 elsewhere); cam_pos holds BAND ROWS
    DevBuf<double> subZ, subD, subT, subF, subL, subW, subC, subTc; DevBuf<int> sub_flags, sub_fz_lo, sub_fz_hi, sub_fz_wend, sub_fz_merge, sub_fz_await, sub_fz_signal, sub_fz_flags; int sub_seq = 0, sub_fz_seq = 0;      // subC / subTc / flags: the two-sided chain's hand-over (band_sub.h 4b)
    DevBuf<int> col_pos, trans_pos;      // band row of the column camera of every stored / transposed block (k_arrow_update)
    // rings (round 5, band_ring.h): wrap table of the gather kernels, cyclic-reduction schedule, per-separator factor / coupling / right-hand-side blocks
    DevBuf<int> wrap_ptr, wrap_blk, wrap_row2, ring_rec, ring_tail; DevBuf<double> crL, crF, crW, crP, crT, crE;
    const int* wrap_ptr_p() const { return wrap_ptr.n ? wrap_ptr.p : nullptr; }
    // supernodal ring / chain solver of the reduced system (snode.h): plan, tables, factor workspace, the halves' exchange buffers and flags
    SnodePlan sn; DevBuf<int> sn_half, sn_step, sn_node, sn_tab, sn_flags; DevBuf<double> sn_work, sn_xchg; int sn_seq = 0;
    DevBuf<int> trans_ptr, trans_blk, trans_row, pair_j, pair_j2, pair_p, batch_slot, cam_batch_ptr, chunk_cam, chunk_b0, chunk_b1, cam_obs_pt, cs_task_cam, cs_task_q0, cs_task_q1;
    double *S_val = nullptr, *rhs = nullptr, *Udiag = nullptr, *Sfc = nullptr, *gcraw = nullptr, *red_scal = nullptr;
    double focal_host = 0, t_flatten_s = 0;
    double* host_sp = nullptr;           // pinned read-back buffer of the LM loop
    double* host_pub = nullptr;          // = ctx->host_pub once publish_alloc ran (k_publish target, coherent pinned memory)
    bool scale_ready = false;
    bool band_filled = false;            // set by k_finalize_gather for the next solve_reduced call
    // set by the LM loop for the next solve_reduced call: the candidate cameras are produced by the arrow kernel (k_arrow_update)
    struct { bool on = false, residual_later = false; const double *cam = nullptr, *focal = nullptr; double *cam_c = nullptr, *focal_c = nullptr, *rot_c = nullptr; } tail;
    DevBuf<int> gr_rec; DevBuf<unsigned char> pt_grouped;     // signature groups of k_schur_gram (ba_flatten.h)
    // atomics-free Gram emission (round 6): the tasks' partial blocks / vectors and the fold lists of k_finalize_gather; gram_fold = the path is in use
    DevBuf<double> gram_part; DevBuf<int> gpart_off, fold_slot_ptr, fold_slot_src; bool gram_fold = false;
    DevBuf<int> pub_ticket;              // arrival counter of the fused hand-over in k_point_backsub (zero between launches)
    int pcg_prev_iters = 16;
    bool external_tail = false;          // the caller runs its own focal arrow / residual check after the direct solve (rotavg_solver.hip: k_rot_step, k_rot_eval)
    // profiling
    bool profile = false;
    std::vector<hipEvent_t> ev_pool; size_t ev_used = 0;
    struct Span { int kid; hipEvent_t a, b; };
    std::vector<Span> spans;
    int64_t k_launches[KID_COUNT]; double k_ms[KID_COUNT];
    hipEvent_t phase_ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    hipEvent_t x0_ev = nullptr;          // behind the copy of |x_0|^2 at a solve's start: the host reads it when the first tail is enqueued, not before the first kernels

    hipEvent_t get_event() {
        if (ev_used == ev_pool.size()) { hipEvent_t e; (void)hipEventCreate(&e); ev_pool.push_back(e); }
        return ev_pool[ev_used++];
    }
    void span_begin(int kid) { if (!profile) return; Span s{kid, get_event(), get_event()}; (void)hipEventRecord(s.a, ctx->stream); spans.push_back(s); }
    void span_end() { if (!profile) return; (void)hipEventRecord(spans.back().b, ctx->stream); }
    void resolve_spans() {
        for (auto& s : spans) { float ms = 0; if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) { k_ms[s.kid] += ms; k_launches[s.kid]++; } }
        spans.clear(); ev_used = 0;
    }
    void free_all() {
        cam_x.free(); cam_c.free(); cam_init.free(); pts_x.free(); pts_c.free(); pts_init.free(); focal3.free();
        rot_x.free(); rot_c.free(); scale_cam.free(); scale_pt.free(); scale_f.free(); mask_cam.free(); mask_pt.free(); mask_f.free();
        diag_cam.free(); diag_pt.free(); diag_f.free(); obs_xy.free(); obs_cam.free(); obs_pt.free(); pt_start.free();
        cam_start.free(); cam_obs.free(); row_ptr.free(); col_idx.free(); diag_slot.free(); Vs.free(); gp.free();
        lmdev.free(); band.free(); Linv.free(); Yb.free(); Yr.free(); cam_pos.free(); band_pairs.free(); band_fail.free(); comp_ptr.free();
        sub_seg_lo.free(); sub_seg_hi.free(); sub_seg_wend.free(); sub_left.free(); sub_sep_lo.free(); sub_sep_rseg.free(); sub_chain_ptr.free(); sub_tw_lo.free(); sub_tw_hi.free(); sub_tw_copy.free(); sub_seg_given.free(); sub_seg_mode.free(); cam_pos2.free(); pair_dummy.free();
        subZ.free(); subD.free(); subT.free(); subF.free(); subL.free(); subW.free(); subC.free(); subTc.free(); sub_flags.free(); sub_fz_lo.free(); sub_fz_hi.free(); sub_fz_wend.free(); sub_fz_merge.free(); sub_fz_await.free(); sub_fz_signal.free(); sub_fz_flags.free();
        col_pos.free(); trans_pos.free(); trans_ptr.free(); trans_blk.free(); trans_row.free(); pair_j.free(); pair_j2.free(); pair_p.free(); batch_slot.free(); cam_batch_ptr.free(); chunk_cam.free(); chunk_b0.free(); chunk_b1.free(); cam_obs_pt.free(); cs_task_cam.free(); cs_task_q0.free(); cs_task_q1.free();
        if (zone_views) { scal.p = nullptr; pcg.p = nullptr; redbuf.p = nullptr; zone_views = false; }
        pub_ticket.free(); gr_rec.free(); pt_grouped.free();
        gram_part.free(); gpart_off.free(); fold_slot_ptr.free(); fold_slot_src.free(); gram_fold = false;
        sn_half.free(); sn_step.free(); sn_node.free(); sn_tab.free(); sn_flags.free(); sn_work.free(); sn_xchg.free(); sn.enabled = false;
        wrap_ptr.free(); wrap_blk.free(); wrap_row2.free(); ring_rec.free(); ring_tail.free(); crL.free(); crF.free(); crW.free(); crP.free(); crT.free(); crE.free();
        zone.free(); redbuf.free(); Minv.free(); Sff.free(); px.free(); pr.free(); pz.free(); pp.free(); pq.free(); pqpart.free(); scal.free(); pcg.free();
        if (host_sp) { (void)hipHostFree(host_sp); host_sp = nullptr; }
        host_pub = nullptr;
        for (auto e : ev_pool) (void)hipEventDestroy(e);
        ev_pool.clear();
        for (auto& e : phase_ev) if (e) { (void)hipEventDestroy(e); e = nullptr; }
        if (x0_ev) { (void)hipEventDestroy(x0_ev); x0_ev = nullptr; }
    }
};

namespace ssfm {

#define MATVEC(h, DCV, vec)                                                                                                   \
    do {                                                                                                                   \
        if ((h)->F.sym_lower)                                                                                              \
            LAUNCH(h, KID_PCG_MATVEC, k_sym_matvec<DCV>, ((h)->F.Nc + 3) / 4, 256, 0, (h)->row_ptr.p, (h)->col_idx.p, (h)->trans_ptr.p, \
                   (h)->trans_blk.p, (h)->trans_row.p, (h)->S_val, (h)->Sfc, vec, (h)->F.Nc, (h)->pcg.p, (h)->pq.p, (h)->pqpart.p);       \
        else                                                                                                               \
            LAUNCH(h, KID_PCG_MATVEC, k_pcg_matvec<DCV>, ((h)->F.Nc + 3) / 4, 256, 0, (h)->row_ptr.p, (h)->col_idx.p, (h)->S_val,       \
                   (h)->Sfc, vec, (h)->F.Nc, (h)->pcg.p, (h)->pq.p, (h)->pqpart.p);                                         \
    } while (0)

// back substitution: one, two or three task sets per lane (b * DC <= 64 / 128 / 192)
#define BACK_V2_LAUNCH(grid, ...)                                                                                          \
    do { if (b * DC > 128) hipLaunchKernelGGL((k_band_back_v2<DC, 3>), grid, dim3(64), 0, st, __VA_ARGS__);               \
         else if (b * DC > 64) hipLaunchKernelGGL((k_band_back_v2<DC, 2>), grid, dim3(64), 0, st, __VA_ARGS__);            \
         else hipLaunchKernelGGL((k_band_back_v2<DC, 1>), grid, dim3(64), 0, st, __VA_ARGS__); } while (0)
#define LAUNCH(h, kid, kernel, grid, block, shmem, ...)                                   \
    do {                                                                                  \
        (h)->span_begin(kid);                                                             \
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), shmem, (h)->ctx->stream, __VA_ARGS__); \
        (h)->span_end();                                                                  \
    } while (0)

template <typename T, typename A>
static hipError_t upload(DevBuf<T>& b, const std::vector<T, A>& v, hipStream_t s) {
    hipError_t e = b.alloc(v.size()); if (e != hipSuccess) return e;
    if (v.empty()) return hipSuccess;
    return hipMemcpyAsync(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s);
}

// Small host arrays through the context's pinned staging buffer (round 6): a pageable hipMemcpyAsync stages and waits inside the call -- the ~25 arrays of a pose-graph
// solve's set-up were 25 copies of 2-5 us with the host's 4-20 us between them (rocprofv3 trace: 210 us); from pinned memory the calls return at once and the copies
// run back to back.  reserve() before the first up(): the buffer is grow-only and must not move while copies are in flight; an array that does not fit takes upload().
struct StagedUploads {
    ssfm_ctx* ctx; hipStream_t st; size_t off = 0;
    StagedUploads(ssfm_ctx* c, hipStream_t s) : ctx(c), st(s) {}
    hipError_t reserve(size_t bytes) {
        const size_t want = (bytes + 7) / 8 + 64;
        if (ctx->dl_stage_n >= want) return hipSuccess;
        if (ctx->dl_stage) (void)hipHostFree(ctx->dl_stage);
        ctx->dl_stage = nullptr; ctx->dl_stage_n = 0;
        const size_t n = want + want / 2;
        hipError_t e = hipHostMalloc((void**)&ctx->dl_stage, n * sizeof(double), hipHostMallocDefault);
        if (e == hipSuccess) ctx->dl_stage_n = n; else { ctx->dl_stage = nullptr; (void)hipGetLastError(); }
        return hipSuccess;                                             // (no pinned memory: every up() falls back to upload())
    }
    template <typename T, typename A>
    hipError_t up(DevBuf<T>& b, const std::vector<T, A>& v) {
        const size_t bytes = v.size() * sizeof(T), need = (bytes + 63) / 64 * 64;
        if (!ctx->dl_stage || off + need > ctx->dl_stage_n * sizeof(double)) return upload(b, v, st);
        hipError_t e = b.alloc(v.size()); if (e != hipSuccess) return e;
        if (v.empty()) return hipSuccess;
        char* dst = reinterpret_cast<char*>(ctx->dl_stage) + off; off += need;
        std::memcpy(dst, v.data(), bytes);
        return hipMemcpyAsync(b.p, dst, bytes, hipMemcpyHostToDevice, st);
    }
};

static int allreduce(ssfm_ba_handle* h, double* buf, size_t n, ncclRedOp_t op) {
    if (!h->ctx->collective) return SSFM_OK;
    // TIMING EXPERIMENT ONLY (bench.py --gpus N, `timing_without_collective`; switched on by ssfm_debug_timing_skip_collectives, which warns on stderr -- no
    // environment variable can do this): every rank solves its own shard without the reductions -- the results are meaningless, the kernels and their sizes are
    // those of the sharded solve, so the difference to the real run prices the collectives
    if (h->ctx->timing_skip_collectives) return SSFM_OK;
    h->span_begin(KID_ALLREDUCE);
    const int rc = ctx_allreduce(h->ctx, buf, n, op);
    h->span_end();
    return rc;
}

// ---- end-of-iteration hand-over to the host without a copy or a stream synchronisation (k_publish, ba_kernels.h) ----
static bool lm_poll() {                                             // read per solve (tests switch it within one process)
    const char* e = std::getenv("SSFM_LM_POLL");
    return !(e && std::atoi(e) == 0);
}
// false: no coherent pinned block to be had -- the caller falls back to the copy + stream synchronisation hand-over
static bool publish_alloc(ssfm_ba_handle* h) {
    ssfm_ctx* ctx = h->ctx;
    if (!ctx->host_pub) {
        if (hipHostMalloc((void**)&ctx->host_pub, 32 * sizeof(double), hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) {
            (void)hipGetLastError(); ctx->host_pub = nullptr; return false;
        }
        std::memset(ctx->host_pub, 0, 32 * sizeof(double));
    }
    h->host_pub = ctx->host_pub;
    return true;
}
static void publish(ssfm_ba_handle* h, const LmGate* gate = nullptr, double* spec = nullptr, unsigned det_kmask = 0) {
    LmGate g; std::memset(&g, 0, sizeof(g)); if (gate) g = *gate;
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(SC_TOTAL * 64), 0, h->ctx->stream, h->scal.p, h->pcg.p, h->host_pub, ++h->ctx->pub_seq, g, spec,
                       (h->det && det_kmask) ? h->det_lacc.p : (long long*)nullptr, det_kmask);
}
static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#else
    std::atomic_signal_fence(std::memory_order_seq_cst);
#endif
}
// spin until the sequence number of the last publish() shows up; now and then ask the stream whether it is still alive
static int wait_published(ssfm_ba_handle* h) {
    ssfm_ctx* ctx = h->ctx;
    volatile unsigned long long* flag = reinterpret_cast<volatile unsigned long long*>(h->host_pub + SC_TOTAL + PCG_TOTAL + 1);
    const unsigned long long want = ctx->pub_seq;
    unsigned spins = 0;
    while (*flag != want) {
        cpu_relax();
        if ((++spins & 0xFFFFu) == 0) {                          // every millisecond or so
            const hipError_t q = hipStreamQuery(ctx->stream);
            if (q == hipSuccess) { if (*flag == want) break; return fail(ctx, SSFM_ERR_HIP, "an LM iteration ended without publishing its scalars"); }
            if (q != hipErrorNotReady) return fail(ctx, SSFM_ERR_HIP, hipGetErrorString(q));
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    return SSFM_OK;
}

// segment / separator tables of the substructured factorisation (band_sub.h) and its work buffers
static int sub_upload(ssfm_ba_handle* h, int DC) {
    ssfm_ctx* ctx = h->ctx; hipStream_t st = ctx->stream; const BAFlat& F = h->F;
    if (F.band_block > 0) DC = F.band_block;                     // block size of the band (merged 3-dof pairs: 6)
    sub_build(F.comp_ptr, F.comp_twist, F.band, DC, h->sub, &F.rings);
    if (!F.wrap_ptr.empty()) { SSFM_HIP_CHECK(ctx, upload(h->wrap_ptr, F.wrap_ptr, st)); SSFM_HIP_CHECK(ctx, upload(h->wrap_blk, F.wrap_blk, st)); SSFM_HIP_CHECK(ctx, upload(h->wrap_row2, F.wrap_row2, st)); }
    if (!h->sub.enabled) return SSFM_OK;
    const BandSub& B = h->sub; const size_t Q = (size_t)F.band * DC, n = (size_t)(F.band_rows > 0 ? F.band_rows : F.Nc) * DC;
    SSFM_HIP_CHECK(ctx, upload(h->sub_tw_lo, B.tw_lo, st)); SSFM_HIP_CHECK(ctx, upload(h->sub_tw_hi, B.tw_hi, st)); SSFM_HIP_CHECK(ctx, upload(h->sub_tw_copy, B.tw_copy, st)); SSFM_HIP_CHECK(ctx, upload(h->sub_seg_given, B.seg_given, st));
    {   // back substitution mode of a segment: 0, or 1 / 2 for the two halves of a twisted component (band_kernels2.h: k_band_back_v2 solves the separator itself)
        std::vector<int> mode(B.seg_twist.size());
        for (size_t i = 0; i < mode.size(); i++) mode[i] = B.seg_twist[i] >= 0 ? 1 + (B.seg_twist[i] & 1) : 0;
        SSFM_HIP_CHECK(ctx, upload(h->sub_seg_mode, mode, st));
    }
    SSFM_HIP_CHECK(ctx, upload(h->sub_seg_lo, B.seg_lo, st)); SSFM_HIP_CHECK(ctx, upload(h->sub_seg_hi, B.seg_hi, st));
    SSFM_HIP_CHECK(ctx, upload(h->sub_seg_wend, B.seg_wend, st)); SSFM_HIP_CHECK(ctx, upload(h->sub_left, B.left_segs, st));
    SSFM_HIP_CHECK(ctx, upload(h->sub_sep_lo, B.sep_lo, st)); SSFM_HIP_CHECK(ctx, upload(h->sub_sep_rseg, B.sep_rseg, st));
    SSFM_HIP_CHECK(ctx, upload(h->sub_chain_ptr, B.chain_ptr, st));
    if (B.ntwist > 0) {
        SSFM_HIP_CHECK(ctx, upload(h->sub_fz_lo, B.fz_lo, st)); SSFM_HIP_CHECK(ctx, upload(h->sub_fz_hi, B.fz_hi, st)); SSFM_HIP_CHECK(ctx, upload(h->sub_fz_wend, B.fz_wend, st));
        SSFM_HIP_CHECK(ctx, upload(h->sub_fz_merge, B.fz_merge, st)); SSFM_HIP_CHECK(ctx, upload(h->sub_fz_await, B.fz_await, st)); SSFM_HIP_CHECK(ctx, upload(h->sub_fz_signal, B.fz_signal, st));
        SSFM_HIP_CHECK(ctx, h->sub_fz_flags.alloc((size_t)4 * B.ntwist)); SSFM_HIP_CHECK(ctx, hipMemsetAsync(h->sub_fz_flags.p, 0, (size_t)4 * B.ntwist * sizeof(int), st)); h->sub_fz_seq = 0;
    }
    if (B.nsep == 0) return SSFM_OK;                             // twisted components only: no spikes, no chain
    SSFM_HIP_CHECK(ctx, h->subZ.alloc(Q * n)); SSFM_HIP_CHECK(ctx, h->subD.alloc((size_t)B.nsep * Q * Q)); SSFM_HIP_CHECK(ctx, h->subT.alloc((size_t)B.nsep * 2 * Q));
    SSFM_HIP_CHECK(ctx, h->subF.alloc((size_t)(B.nsep + B.nchain) * Q * Q)); SSFM_HIP_CHECK(ctx, h->subL.alloc((size_t)B.nsep * Q * (Q + 1) / 2)); SSFM_HIP_CHECK(ctx, h->subW.alloc((size_t)B.nsep * 2 * Q));
    SSFM_HIP_CHECK(ctx, h->subC.alloc((size_t)B.nchain * Q * (Q + 1) / 2)); SSFM_HIP_CHECK(ctx, h->subTc.alloc((size_t)B.nchain * 2 * Q)); SSFM_HIP_CHECK(ctx, h->sub_flags.alloc((size_t)2 * B.nchain));
    SSFM_HIP_CHECK(ctx, hipMemsetAsync(h->sub_flags.p, 0, (size_t)2 * B.nchain * sizeof(int), st)); h->sub_seq = 0;
    if (B.nring > 0) {
        SSFM_HIP_CHECK(ctx, upload(h->ring_rec, B.ring_rec, st)); SSFM_HIP_CHECK(ctx, upload(h->ring_tail, B.ring_tail_ptr, st));
        SSFM_HIP_CHECK(ctx, h->crL.alloc((size_t)B.nsep * Q * Q)); SSFM_HIP_CHECK(ctx, h->crF.alloc((size_t)B.nsep * 2 * Q * Q)); SSFM_HIP_CHECK(ctx, h->crW.alloc((size_t)B.nsep * 2 * Q));
        SSFM_HIP_CHECK(ctx, h->crP.alloc((size_t)B.nsep * 2 * Q * Q)); SSFM_HIP_CHECK(ctx, h->crT.alloc((size_t)B.nsep * 4 * Q)); SSFM_HIP_CHECK(ctx, h->crE.alloc((size_t)B.nsep * Q * Q));
    }
    return SSFM_OK;
}

// Round 4: bands too wide for the square window ring (6x6 blocks, half-width 21..30) keep an LDS-resident factorisation through the PACKED window of
// band_kernels2p.h (the live triangle only: 109 KB at half-width 26) and the single-wave back substitution with three task sets per lane; twisted components
// included (the planner twists up to half-width 30 when this path is on, ba_flatten.h: band_wide_max).  One connected ring of 300 cameras at half-width 26: 4.0 us
// per block row against 6.2 for the global-memory kernel (scripts/lab/chol_lab3.hip, profiles/r04_notes.md).  SSFM_BAND_PACKED=0 keeps the global-memory kernels.
static bool band_wide_packed(int DC, int b, bool use_lds) {
    if (use_lds || DC != 6 || b < 1 || !band_packed_enabled()) return false;
    const int tasks2p = (b * (b + 1) / 2) * 4 - 4 + b * DC;
    return chol2p_lds_bytes(b, 2) <= 160 * 1024 && tasks2p <= 3 * 12 * 64 && b * 36 <= 2 * 1024 && (b + 1) * 36 + 2 * DC <= 9 * 128 && b * DC <= 192;
}


// ---- supernodal solver (snode.h): tables and buffers of a plan; the one launch that factors, substitutes and scatters
static int snode_upload(ssfm_ba_handle* h) {
    ssfm_ctx* ctx = h->ctx; hipStream_t st = ctx->stream;
    const SnodePlan& P = h->sn;
    if (!P.enabled) return SSFM_OK;
    SSFM_HIP_CHECK(ctx, upload(h->sn_half, P.half_rec, st)); SSFM_HIP_CHECK(ctx, upload(h->sn_step, P.step_rec.empty() ? std::vector<int>(SN_SREC, 0) : P.step_rec, st));
    SSFM_HIP_CHECK(ctx, upload(h->sn_node, P.node_cam, st)); SSFM_HIP_CHECK(ctx, upload(h->sn_tab, P.tab.empty() ? std::vector<int>(1, -1) : P.tab, st));
    SSFM_HIP_CHECK(ctx, h->sn_work.alloc(std::max<size_t>(P.work_doubles, 1))); SSFM_HIP_CHECK(ctx, h->sn_xchg.alloc(std::max<size_t>(P.xchg_doubles, 1)));
    SSFM_HIP_CHECK(ctx, h->sn_flags.alloc((size_t)std::max(P.nflags, 1)));
    SSFM_HIP_CHECK(ctx, hipMemsetAsync(h->sn_flags.p, 0, (size_t)std::max(P.nflags, 1) * sizeof(int), st));
    h->sn_seq = 0;
    return SSFM_OK;
}
template <int DC, int NR, bool RING>
static int snode_launch(ssfm_ba_handle* h, double* Y, size_t ystride, long long* stamps = nullptr) {
    ssfm_ctx* ctx = h->ctx;
    const SnodePlan& P = h->sn;
    const size_t lds = P.lds_bytes();
    if (lds > 48 * 1024) SSFM_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_snode_solve<DC, NR, RING>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    h->sn_seq++;
    LAUNCH(h, KID_SNODE, (k_snode_solve<DC, NR, RING>), P.nhalf, SN_THREADS, lds, h->S_val, h->rhs, h->Sfc, h->sn_half.p, h->sn_step.p, h->sn_node.p, h->sn_tab.p, h->cam_pos.p,
           h->sn_work.p, h->sn_xchg.p, h->sn_flags.p, h->sn_seq, P.qtm, Y, ystride, reinterpret_cast<int*>(h->pcg.p + PCG_TOTAL), stamps);
    return SSFM_OK;
}
// S y = [rhs | S_fc] by the supernodal solver; Y in the band-row layout of the right-hand sides (the second column only with a free focal length: it stays as
// k_finalize_gather left it otherwise, zero)
template <int DC>
static int snode_direct(ssfm_ba_handle* h, double* Y, size_t ystride, long long* stamps = nullptr) {
    const bool ring = h->sn.qtm > 0;
    if (h->F.focal_free) return ring ? snode_launch<DC, 2, true>(h, Y, ystride, stamps) : snode_launch<DC, 2, false>(h, Y, ystride, stamps);
    return ring ? snode_launch<DC, 1, true>(h, Y, ystride, stamps) : snode_launch<DC, 1, false>(h, Y, ystride, stamps);
}

// Factor the band in h->band (block-band Cholesky in Cuthill-McKee order) and solve for the two right-hand-side columns of Y
// (band order), in place.  Long components go through the substructured path (band_sub.h) when the plan holds one.
template <int DC>
static int band_direct(ssfm_ba_handle* h, double* Y) {
    ssfm_ctx* ctx = h->ctx; hipStream_t st = ctx->stream;
    const BAFlat& F = h->F;
    const int Nc = F.band_rows > 0 ? F.band_rows : F.Nc, b = F.band;      // Nc here = rows of the band (>= cameras: twisted components)
    constexpr int BB = DC * DC;
    const int ncomp = (int)F.comp_ptr.size() - 1;
    // LDS-resident factorisation (band_kernels2.h): window ring + panel + right-hand-side rows + scratch + pair table
    const size_t lds_win = ((size_t)(b + 1) * (b + 1) * BB + (size_t)b * BB + (size_t)(b + 1) * 2 * DC + 2 * DC + 2 * BB) * sizeof(double) + ((size_t)b * (b + 1) / 2 + 2) * sizeof(int);
    const bool use_lds = lds_win <= 140 * 1024 && b >= 1;
    const bool wide2p = band_wide_packed(DC, b, use_lds);
    const bool back_v2 = (use_lds && b * DC <= 128) || wide2p;          // single-wave back substitution: up to two task sets per lane with the square window, three with the packed one
    const size_t lds2p = wide2p ? chol2p_lds_bytes(b, 2) : 0;
    // wave roles of k_band_chol_v2: 1 look-ahead + trailing-update waves (one block task per lane) + loaders + 1 writer
    // (block tasks and right-hand-side tasks on waves of their own when seven waves allow it: see the trailing role of the kernel)
    const int tpb_ = (DC % 3 == 0) ? (DC / 3) * (DC / 3) : DC * DC, blk_tasks = (b * (b + 1) / 2) * tpb_ - tpb_, rhs_tasks = b * DC;
    const int tr_waves_split = (std::max(blk_tasks, 1) + 63) / 64 + (rhs_tasks + 63) / 64, tr_waves_packed = (blk_tasks + rhs_tasks + 63) / 64;
    const int chol_ntw = std::min(std::max(tr_waves_split <= 7 ? tr_waves_split : tr_waves_packed, 1), 7);
    const int chol_threads = 64 * (2 + CHOL2_LOADERS + chol_ntw);
    static const bool chol_remap = SSFM_LAB_KNOB("SSFM_CHOL_WAVE_MAP", 1) != 0;
    const CholWaveMap chol_map = chol_remap ? chol_wave_map(chol_threads / 64, chol_ntw, tr_waves_split <= 7 ? (std::max(blk_tasks, 1) + 63) / 64 : chol_ntw)
                                            : chol_wave_map(0, 0, 0);
    const size_t lds_chol = (size_t)(2 * BB + 2 * DC + (size_t)b * BB) * sizeof(double);
    const size_t lds_sub2 = (size_t)(2 * (size_t)b * 2 * DC + 2 * DC) * sizeof(double);
    // matrix-core panel + trailing update (band_kernels2.h, MF): 6x6 blocks only; SSFM_BAND_MFMA=0 keeps the VALU version
    static const int band_mfma_env = SSFM_LAB_KNOB("SSFM_BAND_MFMA", 0);      // bit 0: panel, bit 1: trailing update (experiment)
    constexpr int MFP = (DC == 6) ? 1 : 0, MFT = (DC == 6) ? 2 : 0, MFB = (DC == 6) ? 3 : 0; (void)MFP; (void)MFT; (void)MFB;
    const int mf = (DC == 6 && b * DC <= 127) ? (band_mfma_env & 3) : 0;
#define SSFM_LAUNCH_CHOL2_V(V_, grid_, ...)                                                                                                 \
    do { if (lds_win > 48 * 1024) SSFM_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_band_chol_v2<DC, 2, V_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_win)); \
         LAUNCH(h, KID_BAND_CHOL, (k_band_chol_v2<DC, 2, V_>), grid_, chol_threads, lds_win, __VA_ARGS__); } while (0)
    // the packed-window kernel takes the same tables (no wave map: its roles sit in wave order)
#define SSFM_LAUNCH_2P(NPB_, PRE_, grid_, lo_, hi_, wend_, merge_)                                                                                            \
    do { if constexpr (DC == 6) {                                                                                                                            \
         SSFM_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_band_chol_v2p<2, 3, NPB_, PRE_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2p)); \
         LAUNCH(h, KID_BAND_CHOL, (k_band_chol_v2p<2, 3, NPB_, PRE_>), grid_, 1024, lds2p, h->band.p, h->Linv.p, Y, h->band_pairs.p, lo_, hi_, wend_, merge_, Nc, b,        \
                reinterpret_cast<int*>(h->pcg.p + PCG_TOTAL)); } } while (0)
    // up to half-width 26 the 6x6-tile variant (one block per lane, twelve waves, 168 VGPRs): 3.45 against 4.02 us per block row at 26, 2.72 against 3.31 at 22
    const bool t6 = wide2p && (b * (b + 1) / 2 - 1 + b * DC) <= 8 * 64 && b * BB <= 2 * 768 && (b + 1) * BB + 2 * DC <= 8 * 128 && chol2p_lds_bytes(b, 2, true) <= 160 * 1024;
#define SSFM_LAUNCH_2P6(grid_, lo_, hi_, wend_, merge_)                                                                                                      \
    do { if constexpr (DC == 6) { const size_t l6 = chol2p_lds_bytes(b, 2, true);                                                                            \
         SSFM_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_band_chol_v2p<2, 1, 2, 8, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l6)); \
         LAUNCH(h, KID_BAND_CHOL, (k_band_chol_v2p<2, 1, 2, 8, true>), grid_, 768, l6, h->band.p, h->Linv.p, Y, h->band_pairs.p, lo_, hi_, wend_, merge_, Nc, b,            \
                reinterpret_cast<int*>(h->pcg.p + PCG_TOTAL)); } } while (0)
#define SSFM_LAUNCH_CHOL2P(grid_, lo_, hi_, wend_, merge_)                                                                                                   \
    do { if (t6) SSFM_LAUNCH_2P6(grid_, lo_, hi_, wend_, merge_); else if (b * BB <= 1024 && (b + 1) * BB + 2 * DC <= 8 * 128) SSFM_LAUNCH_2P(1, 8, grid_, lo_, hi_, wend_, merge_); else SSFM_LAUNCH_2P(2, 9, grid_, lo_, hi_, wend_, merge_); } while (0)
    // early look-ahead (band_kernels2.h, MF bit 2; experiment, off): the look-ahead wave computes X_1 and the update of the next diagonal block before barrier A.  Measured
    // in the lab (scripts/lab/chol_lab3.hip, same run): 89.8 against 79.9 us for 4 x 75 rows at half-width 10, 106.2 / 98.3 at 12, 124.9 / 123.2 at 14 -- once the pivot test
    // left the factorisation's chain the step is bound by the trailing waves, and the longer stretch in front of barrier A only delays them.  SSFM_BAND_EARLY=1 selects it.
    static const bool early_on = SSFM_LAB_KNOB("SSFM_BAND_EARLY", 0) != 0;
    const bool early = early_on && mf == 0 && b * BB <= BB + chol_threads - 64;
#ifdef SSFM_LAB
#define SSFM_LAUNCH_CHOL2(grid_, ...)                                                                                                       \
    do { if (early) SSFM_LAUNCH_CHOL2_V(4, grid_, __VA_ARGS__); else if (mf == 3) SSFM_LAUNCH_CHOL2_V(MFB, grid_, __VA_ARGS__); else if (mf == 2) SSFM_LAUNCH_CHOL2_V(MFT, grid_, __VA_ARGS__);          \
         else if (mf == 1) SSFM_LAUNCH_CHOL2_V(MFP, grid_, __VA_ARGS__); else SSFM_LAUNCH_CHOL2_V(0, grid_, __VA_ARGS__); } while (0)
#else       // the shipped library holds the VALU factorisation only: the matrix-core panel / trailing update and the early look-ahead were measured slower
#define SSFM_LAUNCH_CHOL2(grid_, ...) do { (void)early; (void)mf; SSFM_LAUNCH_CHOL2_V(0, grid_, __VA_ARGS__); } while (0)
#endif
        if (h->sub.enabled && (use_lds || wide2p) && back_v2) {
            // substructured: segments in parallel, spikes, separator chain, back substitution (band_sub.h)
            const BandSub& B = h->sub;
            const int Q = b * DC;
            // a second packed triangle when it fits (Q <= 96): the chain kernel then brings the next separator's D in while it factors this one (band_sub.h: pingpong)
            static const bool chain_pp_on = SSFM_LAB_KNOB("SSFM_CHAIN_PINGPONG", 1) != 0;
            const size_t lds_chain1 = ((size_t)Q * (Q + 1) / 2 + (size_t)Q * Q + (size_t)(2 * 2) * Q) * sizeof(double);
            const int chain_pp = (chain_pp_on && lds_chain1 + (size_t)Q * (Q + 1) / 2 * sizeof(double) <= 160 * 1024) ? 1 : 0;
            const size_t lds_chain = lds_chain1 + (chain_pp ? (size_t)Q * (Q + 1) / 2 * sizeof(double) : 0);
#ifdef SSFM_LAB
            if (B.nsep > 0 && lds_chain > 48 * 1024) SSFM_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_sub_sep_chain<DC, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_chain));
#endif
            if (B.nsep > 0 && lds_chain > 48 * 1024) SSFM_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_sub_sep_chain_mfma<DC, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_chain));
#ifdef SSFM_LAB
            if (B.nsep > 0 && lds_chain > 48 * 1024) SSFM_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_sub_sep_chain_mfma<DC, 2, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_chain));
#endif
#ifdef SSFM_LAB
            if (B.nsep > 0 && lds_chain > 48 * 1024) SSFM_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_sub_sep_chain_mfma<DC, 2, true, true, 512>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_chain));
#endif
#ifdef SSFM_LAB
            if (B.nsep > 0 && lds_chain > 48 * 1024) SSFM_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_sub_sep_chain_mfma<DC, 2, true, true, 1024, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_chain));
#endif
            // the 16x16 diagonal blocks of the chain are factored and inverted on the matrix cores (band_sub.h: wave_ldl_inverse16_mfma); SSFM_CHAIN_DIAG_MFMA=0: lane per row (round 2)
            static const bool chain_diag_mfma = SSFM_LAB_KNOB("SSFM_CHAIN_DIAG_MFMA", 1) != 0;
            static const bool chain_mfma = SSFM_LAB_KNOB("SSFM_CHAIN_MFMA", 1) != 0;     // matrix-core separator chain (band_sub.h 4b)
            int* failp = reinterpret_cast<int*>(h->pcg.p + PCG_TOTAL);
            // SSFM_CHOL_FUSE=1 (experiment, off): one launch for the segments AND the separators of twisted components, which then wait for their two halves
            // through flags in global memory.  Measured at config 2: 70.7 us for the fused launch against 2 x 35.2, 2.958 vs 2.919 ms per solve -- the fence + flag
            // hand-over costs what the launch boundary did (profiles/r02_notes.md)
            static const bool chol_fuse = SSFM_LAB_KNOB("SSFM_CHOL_FUSE", 0) != 0;
            const bool fused = chol_fuse && !wide2p && B.ntwist > 0 && B.nseg + B.ntwist <= ctx->num_cus;      // waiting workgroups must all be resident (one per compute unit)
            if (fused) { h->sub_fz_seq++;
                SSFM_LAUNCH_CHOL2(B.nseg + B.ntwist, h->band.p, h->Linv.p, Y, h->band_pairs.p, h->sub_fz_lo.p, h->sub_fz_hi.p, h->sub_fz_wend.p, h->sub_fz_merge.p, Nc, b, failp, chol_map,
                                  h->sub_fz_await.p, h->sub_fz_signal.p, h->sub_fz_flags.p, h->sub_fz_seq); }
            else if (wide2p) SSFM_LAUNCH_CHOL2P(B.nseg, h->sub_seg_lo.p, h->sub_seg_hi.p, h->sub_seg_wend.p, (const int*)nullptr);       // (no cut components at these widths: twisted halves only)
            else SSFM_LAUNCH_CHOL2(B.nseg, h->band.p, h->Linv.p, Y, h->band_pairs.p, h->sub_seg_lo.p, h->sub_seg_hi.p, h->sub_seg_wend.p, (const int*)nullptr, Nc, b, failp, chol_map);
            int max_rows = 0; for (int sg : B.left_segs) max_rows = std::max(max_rows, (B.seg_hi[sg] - B.seg_lo[sg]) * DC);
            if (B.nsep > 0) {
                h->span_begin(KID_SUB_SPIKE);
                // columns per wave (band_sub.h 2): one while a step is bound by what a wave can issue (half-widths >= 10), more when the band is narrow and the launch is
                // bound by the factor rows every column streams again
                static const int spike_nc_env = SSFM_LAB_KNOB("SSFM_SPIKE_NC", 0);
                // r05ac: what decides is how many one-wave workgroups the launch has, not the half-width -- one column per wave until the chip is full (~3000 waves),
                // then two, then three (us per launch at NC = 1 / 2 / 3: 4 arcs b = 13: 35 / 45 / 58; 8 arcs b = 7: 18.5 / 24.8 / 32.0; 32 arcs b = 5: 19 / 23 / 29;
                // 128 arcs b = 7: 70 / 52.6 / 58.7)
                const long long spike_waves = (long long)B.nleft * Q;
                const int spike_nc = spike_nc_env > 0 ? spike_nc_env : (spike_waves <= 3000 ? 1 : spike_waves <= 6000 ? 2 : 3);
                if (spike_nc >= 4) hipLaunchKernelGGL((k_sub_spike_fwd<DC, 4>), dim3(B.nleft, (Q + 3) / 4), dim3(64), 0, st, h->band.p, h->Linv.p, h->subZ.p, h->sub_seg_lo.p, h->sub_seg_hi.p, h->sub_seg_wend.p, h->sub_left.p, Nc, b);
                else if (spike_nc == 3) hipLaunchKernelGGL((k_sub_spike_fwd<DC, 3>), dim3(B.nleft, (Q + 2) / 3), dim3(64), 0, st, h->band.p, h->Linv.p, h->subZ.p, h->sub_seg_lo.p, h->sub_seg_hi.p, h->sub_seg_wend.p, h->sub_left.p, Nc, b);
                else if (spike_nc == 2) hipLaunchKernelGGL((k_sub_spike_fwd<DC, 2>), dim3(B.nleft, (Q + 1) / 2), dim3(64), 0, st, h->band.p, h->Linv.p, h->subZ.p, h->sub_seg_lo.p, h->sub_seg_hi.p, h->sub_seg_wend.p, h->sub_left.p, Nc, b);
                else hipLaunchKernelGGL((k_sub_spike_fwd<DC, 1>), dim3(B.nleft, Q), dim3(64), 0, st, h->band.p, h->Linv.p, h->subZ.p, h->sub_seg_lo.p, h->sub_seg_hi.p, h->sub_seg_wend.p, h->sub_left.p, Nc, b);
                h->span_end();
                const int ntl = (Q + SUB_TS - 1) / SUB_TS;
                const int nz = std::max(1, std::min(64, (max_rows + 511) / 512)); (void)ntl; (void)nz;
                // separator blocks on the matrix cores, operands straight from global memory, no atomics and no clears (band_sub.h 3b); SSFM_ASM_MFMA=0: the VALU kernel of round 1
                static const bool asm_mfma = SSFM_LAB_KNOB("SSFM_ASM_MFMA", 1) != 0;
                if (asm_mfma) {
                    const int tq = (Q + 15) / 16;
                    h->span_begin(KID_SUB_ASM);
                    hipLaunchKernelGGL((k_sub_sep_assemble_mfma<DC, 2>), dim3(B.nsep * (tq * (tq + 1) / 2 + tq)), dim3(64 * ASM_NW), 0, st, h->band.p, h->subZ.p, Y, h->sub_sep_lo.p, h->sub_sep_rseg.p, h->sub_seg_lo.p, h->sub_seg_hi.p, Nc, b, h->subD.p, h->subT.p);
                    h->span_end();
                }
#ifdef SSFM_LAB
                else {
                SSFM_HIP_CHECK(ctx, hipMemsetAsync(h->subD.p, 0, h->subD.n * sizeof(double), st));
                SSFM_HIP_CHECK(ctx, hipMemsetAsync(h->subT.p, 0, h->subT.n * sizeof(double), st));
                h->span_begin(KID_SUB_ASM);
                hipLaunchKernelGGL((k_sub_sep_assemble<DC, 2>), dim3(B.nsep, ntl * (ntl + 1) / 2, nz), dim3(256), 0, st, h->band.p, h->subZ.p, Y, h->sub_sep_lo.p, h->sub_sep_rseg.p, h->sub_seg_lo.p, h->sub_seg_hi.p, Nc, b, h->subD.p, h->subT.p);
                h->span_end();
                }
#endif
                if (B.nchain > 0) {
                if (chain_mfma) {
                    // SSFM_CHAIN_STAMPS=1: s_memtime stamps of the phases of one separator, printed once (profiles/*_notes.md)
                    static long long* d_stamps = nullptr; static int stamp_state = SSFM_LAB_KNOB("SSFM_CHAIN_STAMPS", 0) ? 1 : 0;
                    if (stamp_state == 1) { (void)hipMalloc((void**)&d_stamps, 16 * sizeof(long long)); (void)hipMemsetAsync(d_stamps, 0, 16 * sizeof(long long), st); stamp_state = 2; }
                    // chains of three or more separators are eliminated from both ends by two workgroups each (SSFM_CHAIN_TWIST=0: one workgroup, front to back)
                    static const bool chain_twist = SSFM_LAB_KNOB("SSFM_CHAIN_TWIST", 1) != 0;
                    // The two workgroups of a chain wait for each other through flags in global memory: both must be RESIDENT.  A chain workgroup (1024 threads,
                    // > 100 KB of LDS) owns a compute unit, so the two-sided form is only used while all 2 x nchain workgroups fit the device at once;
                    // beyond that the one-sided kernel runs (nothing waits on anything in it).
                    const int tw = (chain_twist && 2 * B.nchain <= ctx->num_cus) ? 1 : 0;
                    h->sub_seq++;
                    static const int chain_threads = SSFM_LAB_KNOB("SSFM_CHAIN_THREADS", 1024); (void)chain_threads;    // 512: eight waves with 256 registers each (band_sub.h)
#ifdef SSFM_LAB
                    if (chain_diag_mfma && chain_threads == 512)
                    LAUNCH(h, KID_SUB_CHAIN, (k_sub_sep_chain_mfma<DC, 2, true, true, 512>), B.nchain * (tw ? 2 : 1), 512, lds_chain, h->subZ.p, h->subD.p, h->subT.p, h->sub_chain_ptr.p, h->sub_sep_lo.p, Nc, b, h->subF.p, h->subL.p, h->subW.p, Y, failp, d_stamps,
                           tw, B.nsep, h->subC.p, h->subTc.p, h->sub_flags.p, h->sub_seq, chain_pp);
                    else if (chain_diag_mfma && d_stamps)
                    LAUNCH(h, KID_SUB_CHAIN, (k_sub_sep_chain_mfma<DC, 2, true, true, 1024, true>), B.nchain * (tw ? 2 : 1), 1024, lds_chain, h->subZ.p, h->subD.p, h->subT.p, h->sub_chain_ptr.p, h->sub_sep_lo.p, Nc, b, h->subF.p, h->subL.p, h->subW.p, Y, failp, d_stamps,
                           tw, B.nsep, h->subC.p, h->subTc.p, h->sub_flags.p, h->sub_seq, chain_pp);
                    else
#endif
                    if (chain_diag_mfma)
                    LAUNCH(h, KID_SUB_CHAIN, (k_sub_sep_chain_mfma<DC, 2>), B.nchain * (tw ? 2 : 1), 1024, lds_chain, h->subZ.p, h->subD.p, h->subT.p, h->sub_chain_ptr.p, h->sub_sep_lo.p, Nc, b, h->subF.p, h->subL.p, h->subW.p, Y, failp, d_stamps,
                           tw, B.nsep, h->subC.p, h->subTc.p, h->sub_flags.p, h->sub_seq, chain_pp);
#ifdef SSFM_LAB
                    else
                    LAUNCH(h, KID_SUB_CHAIN, (k_sub_sep_chain_mfma<DC, 2, false, false>), B.nchain * (tw ? 2 : 1), 1024, lds_chain, h->subZ.p, h->subD.p, h->subT.p, h->sub_chain_ptr.p, h->sub_sep_lo.p, Nc, b, h->subF.p, h->subL.p, h->subW.p, Y, failp, d_stamps,
                           tw, B.nsep, h->subC.p, h->subTc.p, h->sub_flags.p, h->sub_seq);
#endif
                    if (stamp_state == 2) {
                        long long hs[16]; (void)hipMemcpyAsync(hs, d_stamps, sizeof(hs), hipMemcpyDeviceToHost, st); (void)hipStreamSynchronize(st);
                        std::fprintf(stderr, "[chain stamps, cycles] load E %lld | F solve %lld | t update + F store %lld | load D %lld | syrk %lld | chol J=0: diag %lld panel %lld trailing %lld | chol total %lld | stores %lld\n",
                                     hs[1] - hs[0], hs[2] - hs[1], hs[3] - hs[2], hs[4] - hs[3], hs[5] - hs[4], hs[6] - hs[5], hs[7] - hs[6], hs[9] - hs[7], hs[8] - hs[5], hs[10] - hs[8]);
                        stamp_state = 3;
                    }
                }
#ifdef SSFM_LAB
                else LAUNCH(h, KID_SUB_CHAIN, (k_sub_sep_chain<DC, 2>), B.nchain, 1024, lds_chain, h->subZ.p, h->subD.p, h->subT.p, h->sub_chain_ptr.p, h->sub_sep_lo.p, Nc, b, h->subF.p, h->subL.p, h->subW.p, Y, failp);
#endif
                }
                if (B.nring > 0) {
                    // rings (band_ring.h): the separator cycles by cyclic reduction -- one launch per parallel step down, ONE for the last few separators of every ring
                    // (their eliminations, the roots, their back substitutions), one per parallel step back up
                    const size_t le = ring_elim_lds_bytes(Q, 2), lb = ring_back_lds_bytes(Q, 2), lt = std::max(le, lb);
                    // workgroups per elimination (band_ring.h, phase 3): three while the step leaves compute units idle, fewer when it fills the chip by itself
                    static const int roles_env = SSFM_LAB_KNOB("SSFM_RING_ROLES", 0);      // lab: force the number (2 / 3 / 4 / 6 at Q = 78: 49 / 47.3 / 45.9 / 45.5 us)
                    auto elim_roles = [&](int nodes) { if (roles_env > 0) return roles_env; return (Q < 40) ? 1 : (3 * nodes <= ctx->num_cus ? 3 : (2 * nodes <= ctx->num_cus ? 2 : 1)); };
                    const int back_threads = Q < 40 ? 256 : std::min(1024, std::max(256, ((RING_BACK_P * 2 * Q + 255) / 256) * 256));     // RING_BACK_P threads per entry of F^T x
                    if (le > 48 * 1024) SSFM_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_ring_cr_elim<DC, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)le));
                    if (lb > 48 * 1024) SSFM_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_ring_cr_back<DC, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb));
                    if (lt > 48 * 1024) SSFM_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_ring_cr_tail<DC, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lt));
                    const int nsteps = (int)B.ring_step_ptr.size() - 1;
                    const int cr_threads = Q > 48 ? 1024 : 512;            // (rows of the tall factorisation: 3 Q + 2 <= 256 either way; fewer waves make cheaper barriers)
                    // SSFM_RING_STAMPS=1 (timing study): phase stamps of every elimination of the first solve, printed once: [loads | factorisation | stores + products]
                    static int ring_stamp_state = SSFM_LAB_KNOB("SSFM_RING_STAMPS", 0) ? 1 : 0; static long long* ring_stamps = nullptr;
                    if (ring_stamp_state == 1) { (void)hipMalloc((void**)&ring_stamps, (size_t)4 * B.nsep * sizeof(long long)); (void)hipMemsetAsync(ring_stamps, 0, (size_t)4 * B.nsep * sizeof(long long), st); ring_stamp_state = 2; }
                    for (int sidx = 0; sidx < nsteps; sidx++)
                        LAUNCH(h, KID_RING_ELIM, (k_ring_cr_elim<DC, 2>), dim3(B.ring_step_ptr[sidx + 1] - B.ring_step_ptr[sidx], elim_roles(B.ring_step_ptr[sidx + 1] - B.ring_step_ptr[sidx])), cr_threads, le, h->ring_rec.p, B.ring_step_ptr[sidx], h->subZ.p, h->subD.p, h->subT.p,
                               h->crL.p, h->crF.p, h->crW.p, h->crP.p, h->crT.p, h->crE.p, Nc, b, failp, ring_stamp_state == 2 ? ring_stamps : (long long*)nullptr);
                    if (ring_stamp_state == 2) {
                        std::vector<long long> hs((size_t)4 * B.nsep); (void)hipStreamSynchronize(st); (void)hipMemcpy(hs.data(), ring_stamps, hs.size() * sizeof(long long), hipMemcpyDeviceToHost);
                        double a = 0, bb = 0, c = 0; int cnt = 0;
                        for (int q = 0; q < B.nsep; q++) if (hs[4 * q + 3] > 0) { a += hs[4 * q + 1] - hs[4 * q]; bb += hs[4 * q + 2] - hs[4 * q + 1]; c += hs[4 * q + 3] - hs[4 * q + 2]; cnt++; }
                        if (cnt) std::fprintf(stderr, "[ring stamps, 100 MHz ticks] Q = %d, %d eliminations (parallel steps): loads %.0f | factorisation %.0f | stores + neighbour products %.0f\n", Q, cnt, a / cnt, bb / cnt, c / cnt);
                        (void)hipFree(ring_stamps); ring_stamps = nullptr; ring_stamp_state = 3;
                    }
                    LAUNCH(h, KID_RING_TAIL, (k_ring_cr_tail<DC, 2>), B.nring, cr_threads, lt, h->ring_rec.p, h->ring_tail.p, h->subZ.p, h->subD.p, h->subT.p,
                           h->crL.p, h->crF.p, h->crW.p, h->crP.p, h->crT.p, h->crE.p, Y, Nc, b, failp);
                    for (int sidx = nsteps - 1; sidx >= 0; sidx--)
                        LAUNCH(h, KID_RING_BACK, (k_ring_cr_back<DC, 2>), B.ring_step_ptr[sidx + 1] - B.ring_step_ptr[sidx], back_threads, lb, h->ring_rec.p, B.ring_step_ptr[sidx], h->crL.p, h->crF.p, h->crW.p, Y, Nc, b);
                }
            }
            // EXPERIMENT, off (SSFM_BACK_FUSE=1): the separator of a twisted component back-substituted by the two segment waves themselves (k_band_back_v2's modes 1 / 2) instead
            // of by a launch of its own.  Measured at config 2 (scripts/lab/ab_backfuse.sh): one launch of 23.2-23.8 us against two of 12.1-12.6 (event brackets), 2.545-2.555 against
            // 2.530-2.546 ms per solve -- the second wave's redundant separator sweep and a second pipeline fill cost what the launch gap did
            static const bool back_fuse = SSFM_LAB_KNOB("SSFM_BACK_FUSE", 0) != 0;
            if (B.ntwist > 0) {
                // twisted components: both segments left their Schur updates in the separator and in its copy; the factorisation kernel merges
                // them while loading its window and solves the separator as a component of b rows; the reversed segment's back substitution
                // reads that solution through seg_given
                if (wide2p) SSFM_LAUNCH_CHOL2P(B.ntwist, h->sub_tw_lo.p, h->sub_tw_hi.p, h->sub_tw_hi.p, h->sub_tw_copy.p);
                else if (!fused) SSFM_LAUNCH_CHOL2(B.ntwist, h->band.p, h->Linv.p, Y, h->band_pairs.p, h->sub_tw_lo.p, h->sub_tw_hi.p, h->sub_tw_hi.p, h->sub_tw_copy.p, Nc, b, failp, chol_map);
                if (!back_fuse) {
                h->span_begin(KID_BAND_BACK);
                BACK_V2_LAUNCH(dim3(B.ntwist, 2), h->band.p, h->Linv.p, Y, h->sub_tw_lo.p, h->sub_tw_hi.p, h->sub_tw_hi.p, (const int*)nullptr, Nc, b);
                h->span_end();
                }
            }
            if (B.nleft > 0) {
                h->span_begin(KID_SUB_APPLY);
                hipLaunchKernelGGL((k_sub_apply_left<DC, 2>), dim3(B.nleft, (max_rows + APPLY_ROWS - 1) / APPLY_ROWS), dim3(APPLY_ROWS * APPLY_SLICES), (size_t)(2 * Q + APPLY_SLICES * APPLY_ROWS * 2) * sizeof(double), st, h->subZ.p, Y, h->sub_seg_lo.p, h->sub_seg_hi.p, h->sub_left.p, Nc, b);
                h->span_end();
            }
            h->span_begin(KID_BAND_BACK);
            BACK_V2_LAUNCH(dim3(B.nseg, 2), h->band.p, h->Linv.p, Y, h->sub_seg_lo.p, h->sub_seg_hi.p, h->sub_seg_wend.p, h->sub_seg_given.p, Nc, b, (const int*)(back_fuse ? h->sub_seg_mode.p : nullptr));
            h->span_end();
            return SSFM_OK;
        }
        // LDS-resident path: the (b+1)^2-block window and the substitution rings fit the CU; one workgroup per component
        if (wide2p) {
            SSFM_LAUNCH_CHOL2P(ncomp, h->comp_ptr.p, h->comp_ptr.p + 1, h->comp_ptr.p + 1, (const int*)nullptr);
            h->span_begin(KID_BAND_BACK);
            BACK_V2_LAUNCH(dim3(ncomp, 2), h->band.p, h->Linv.p, Y, h->comp_ptr.p, h->comp_ptr.p + 1, h->comp_ptr.p + 1, (const int*)nullptr, Nc, b);
            h->span_end();
        } else if (use_lds) {
            SSFM_LAUNCH_CHOL2(ncomp, h->band.p, h->Linv.p, Y, h->band_pairs.p, h->comp_ptr.p, h->comp_ptr.p + 1, h->comp_ptr.p + 1, (const int*)nullptr, Nc, b, reinterpret_cast<int*>(h->pcg.p + PCG_TOTAL), chol_map);
            if (back_v2) {
                h->span_begin(KID_BAND_BACK);
                BACK_V2_LAUNCH(dim3(ncomp, 2), h->band.p, h->Linv.p, Y, h->comp_ptr.p, h->comp_ptr.p + 1, h->comp_ptr.p + 1, (const int*)nullptr, Nc, b);
                h->span_end();
            } else {
                LAUNCH(h, KID_BAND_BACK, (k_band_back_lds<DC, 2>), ncomp, 256, lds_sub2, h->band.p, h->Linv.p, Y, h->comp_ptr.p, Nc, b);
            }
        } else {
            LAUNCH(h, KID_BAND_CHOL, (k_band_chol<DC, 2>), 1, 1024, lds_chol, h->band.p, h->Linv.p, Y, h->band_pairs.p, Nc, b, reinterpret_cast<int*>(h->pcg.p + PCG_TOTAL));
            LAUNCH(h, KID_BAND_BACK, (k_band_back<DC, 2>), 1, 256, lds_sub2, h->band.p, h->Linv.p, Y, Nc, b);
        }
        return SSFM_OK;
#undef SSFM_LAUNCH_CHOL2
#undef SSFM_LAUNCH_CHOL2_V
#undef SSFM_LAUNCH_CHOL2P
#undef SSFM_LAUNCH_2P
#undef SSFM_LAUNCH_2P6
    }

// Second solve with the factor of band_direct (PCG refinement): forward + back substitution of the first column of Y with the stored
// factor.  needs_refactor is set when the plan is substructured (no stand-alone substitution kernels: the caller rebuilds the band
// and calls band_direct again).
template <int DC>
static int band_resolve(ssfm_ba_handle* h, double* Y, bool* needs_refactor) {
    hipStream_t st = h->ctx->stream;
    const BAFlat& F = h->F;
    const int Nb = F.band_rows > 0 ? F.band_rows : F.Nc, b = F.band;
    constexpr int BB = DC * DC;
    const int ncomp = (int)F.comp_ptr.size() - 1;
    const size_t lds_win = ((size_t)(b + 1) * (b + 1) * BB + (size_t)b * BB + (size_t)(b + 1) * 2 * DC + 2 * DC + 2 * BB) * sizeof(double) + ((size_t)b * (b + 1) / 2 + 2) * sizeof(int);
    const bool use_lds = lds_win <= 140 * 1024 && b >= 1;
    const bool wide2p = band_wide_packed(DC, b, use_lds);
    const bool back_v2 = (use_lds && b * DC <= 128) || wide2p;
    const size_t lds_sub1 = (size_t)(2 * (size_t)b * DC + DC) * sizeof(double);
    *needs_refactor = h->sub.enabled && (use_lds || wide2p) && back_v2;
    if (*needs_refactor) return SSFM_OK;
    if (use_lds) {
        LAUNCH(h, KID_BAND_FWD, (k_band_fwd_lds<DC, 1>), ncomp, 256, lds_sub1, h->band.p, h->Linv.p, Y, h->comp_ptr.p, Nb, b);
        if (back_v2) {
            h->span_begin(KID_BAND_BACK);
            BACK_V2_LAUNCH(dim3(ncomp, 1), h->band.p, h->Linv.p, Y, h->comp_ptr.p, h->comp_ptr.p + 1, h->comp_ptr.p + 1, (const int*)nullptr, Nb, b);
            h->span_end();
        } else {
            LAUNCH(h, KID_BAND_BACK, (k_band_back_lds<DC, 1>), ncomp, 256, lds_sub1, h->band.p, h->Linv.p, Y, h->comp_ptr.p, Nb, b);
        }
    } else {
        LAUNCH(h, KID_BAND_FWD, (k_band_fwd<DC, 1>), 1, 256, lds_sub1, h->band.p, h->Linv.p, Y, Nb, b);
        LAUNCH(h, KID_BAND_BACK, (k_band_back<DC, 1>), 1, 256, lds_sub1, h->band.p, h->Linv.p, Y, Nb, b);
    }
    return SSFM_OK;
}

// Solve S y = rhs (block-CSR S with dense focal border) into h->px.
//   preconditioner 0: exact block-banded Cholesky in Cuthill-McKee order, then PCG refinement on the residual
//   preconditioner 1: block-Jacobi PCG (kept for comparison; needs ~10^3 iterations on a camera ring)
template <int DC>
static int solve_reduced(ssfm_ba_handle* h, double* host_pcg, int* iters_out, bool* ok_out, int stage) {
    // stage 0: enqueue the direct solve + residual check, no host sync (flags are read with the iteration scalars)
    // stage 1: flags are in host_pcg; run PCG refinement if the residual test failed
    // (block-Jacobi PCG, preconditioner 1, does everything in stage 0 with its own syncs)
    ssfm_ctx* ctx = h->ctx; hipStream_t st = ctx->stream;
    const BAFlat& F = h->F; const ssfm_ba_options& O = h->opt;
    const int Nc = F.Nc, n = Nc * DC, b = F.band;
    const int Nb = F.band_rows > 0 ? F.y_rows(DC) : Nc, nb = Nb * DC;      // rows (camera units) / stride of the right-hand-side columns in band order
    const double tol2 = O.pcg_tolerance * O.pcg_tolerance;
    if (O.preconditioner == 1) {
        if (stage == 1) return SSFM_OK;
        LAUNCH(h, KID_PCG_INIT, k_pcg_init<DC>, 1, 1024, 0, h->rhs, h->Minv.p, h->Sff.p, Nc, h->px.p, h->pr.p, h->pz.p, h->pp.p, h->pcg.p);
        int launched = 0; bool done = false;
        int chunk = std::max(8, h->pcg_prev_iters + 2);
        while (!done && launched < O.pcg_max_iterations) {
            const int todo = std::min(chunk, O.pcg_max_iterations - launched);
            for (int k = 0; k < todo; k++) {
                MATVEC(h, DC, h->pp.p);
                LAUNCH(h, KID_PCG_VECOPS, k_pcg_vecops<DC>, 1, 1024, 0, h->Minv.p, h->Sfc, h->Sff.p, Nc, tol2, h->px.p, h->pr.p, h->pz.p, h->pp.p, h->pq.p, h->pqpart.p, h->pcg.p);
            }
            launched += todo;
            SSFM_HIP_CHECK(ctx, hipMemcpyAsync(host_pcg, h->pcg.p, PCG_TOTAL * sizeof(double), hipMemcpyDeviceToHost, st));
            SSFM_HIP_CHECK(ctx, hipStreamSynchronize(st));
            done = host_pcg[PCG_DONE] != 0.0;
            chunk = 8;
        }
        *iters_out = (int)host_pcg[PCG_ITERS];
        *ok_out = done && host_pcg[PCG_BREAKDOWN] == 0.0;
        return SSFM_OK;
    }
    // the band may use bigger blocks than S (two 3-dof cameras per 6x6 block row, ba_flatten.h: band_plan)
    const bool merged = F.band_block > 0 && F.band_block != DC;
    auto gather = [&]() {
        if (merged) LAUNCH(h, KID_BAND_GATHER, (k_band_gather<DC, true>), Nc, 256, 0, h->row_ptr.p, h->col_idx.p, h->S_val, h->cam_pos.p, h->cam_pos2.p, h->pair_dummy.p, Nc, b, h->band.p, h->wrap_ptr_p(), h->wrap_blk.p, h->wrap_row2.p);
        else LAUNCH(h, KID_BAND_GATHER, (k_band_gather<DC, false>), Nc, 256, 0, h->row_ptr.p, h->col_idx.p, h->S_val, h->cam_pos.p, h->cam_pos2.p, (const unsigned char*)nullptr, Nc, b, h->band.p, h->wrap_ptr_p(), h->wrap_blk.p, h->wrap_row2.p);
    };
    auto direct = [&](double* Y) -> int { return merged ? band_direct<6>(h, Y) : band_direct<DC>(h, Y); };
    if (stage == 0) {
    if (!h->zone_views) SSFM_HIP_CHECK(ctx, hipMemsetAsync(h->pcg.p, 0, (PCG_TOTAL + 1) * sizeof(double), st));      // flags + the factorisation fail word behind them
    if (!h->band_filled) {                                       // the BA path fills the band in its fused finalize kernel
        gather();
        hipLaunchKernelGGL(k_band_permute_rhs<DC>, dim3((n + 255) / 256), dim3(256), 0, st, h->rhs, h->Sfc, h->cam_pos.p, h->cam_pos2.p, h->pair_dummy.p, Nc, Nb, h->Yb.p);
    }
    h->band_filled = false;
    { const int rc = h->sn.enabled ? snode_direct<DC>(h, h->Yb.p, (size_t)nb) : direct(h->Yb.p); if (rc) return rc; }
    if (h->external_tail) { *iters_out = 0; *ok_out = true; return SSFM_OK; }
    // ---- focal arrow, then the residual check r = rhs - S x (PCG refinement with the factor as preconditioner while it is too large)
    if (F.sym_lower && h->tail.on) {
        const int phi_parts = !F.focal_free ? -1 : (Nc > 1024 ? std::min(64, (n + 2047) / 2048) : 0);      // -1: focal fixed, the arrow is empty (pqpart: Nc doubles, only used by the PCG mat-vec)
        if (phi_parts > 0) LAUNCH(h, KID_BAND_COMBINE, k_arrow_phi<DC>, phi_parts, 256, 0, h->Yb.p, h->Yb.p + nb, h->Sfc, h->cam_pos.p, Nc, h->pqpart.p);
        LAUNCH(h, KID_PCG_MATVEC, k_arrow_update<DC>, (Nc + 3) / 4, 512, 0, h->Yb.p, h->Yb.p + nb, h->Sfc, h->Sff.p, h->rhs + n, h->cam_pos.p, h->row_ptr.p, h->col_idx.p,
               h->trans_ptr.p, h->trans_blk.p, h->trans_row.p, h->S_val, Nc, h->px.p, h->pq.p, h->tail.cam, h->tail.focal, h->scale_cam.p, h->scale_f.p,
               h->tail.cam_c, h->tail.focal_c, h->tail.rot_c, h->scal.p, h->pqpart.p, phi_parts, h->col_pos.p, h->trans_pos.p, h->det ? h->det_lacc.p : (long long*)nullptr);
    } else if (F.sym_lower) {
        LAUNCH(h, KID_PCG_MATVEC, k_arrow_matvec<DC>, (Nc + 3) / 4, 256, 0, h->Yb.p, h->Yb.p + nb, h->Sfc, h->Sff.p, h->rhs + n, h->cam_pos.p, h->row_ptr.p, h->col_idx.p,
               h->trans_ptr.p, h->trans_blk.p, h->trans_row.p, h->S_val, Nc, h->px.p, h->pq.p);
    } else {
        LAUNCH(h, KID_BAND_COMBINE, k_band_combine<DC>, 1, 1024, 0, h->Yb.p, h->Yb.p + nb, h->Sfc, h->Sff.p, h->rhs + n, h->cam_pos.p, Nc, h->px.p);
        MATVEC(h, DC, h->px.p);
    }
    if (!h->tail.residual_later)     // (else: an extra workgroup of k_point_backsub does it)
        LAUNCH(h, KID_REF_VEC, k_ref_residual<DC>, 1, 1024, 0, h->rhs, h->pq.p, h->px.p, h->Sfc, h->Sff.p, Nc, tol2, h->pr.p, h->pcg.p);
    *iters_out = 0; *ok_out = true;
    return SSFM_OK;
    }
    int it = 0;
    { int fail_flag; std::memcpy(&fail_flag, &host_pcg[PCG_TOTAL], sizeof(int));
      if (fail_flag) { *iters_out = 0; *ok_out = false; return SSFM_OK; } }   // S not positive definite: invalid step
    while (host_pcg[PCG_DONE] == 0.0 && it < O.pcg_max_iterations) {
        hipLaunchKernelGGL(k_band_permute_rhs<DC>, dim3((n + 255) / 256), dim3(256), 0, st, h->pr.p, h->Sfc, h->cam_pos.p, h->cam_pos2.p, h->pair_dummy.p, Nc, Nb, h->Yr.p);
        bool refactor = h->sn.enabled;      // (the supernodal factor has no stand-alone substitution: the refinement path rebuilds the band and takes the band kernels)
        if (!refactor) { const int rc = merged ? band_resolve<6>(h, h->Yr.p, &refactor) : band_resolve<DC>(h, h->Yr.p, &refactor); if (rc) return rc; }
        if (refactor) {
            // the substructured factor has no stand-alone substitution kernels: rebuild the band from S and solve again (rare path)
            gather();
            const int rc = direct(h->Yr.p); if (rc) return rc;
        }
        LAUNCH(h, KID_BAND_COMBINE, k_band_combine<DC>, 1, 1024, 0, h->Yr.p, h->Yb.p + nb, h->Sfc, h->Sff.p, h->pr.p + n, h->cam_pos.p, Nc, h->pz.p);
        LAUNCH(h, KID_REF_VEC, k_ref_direction, 1, 1024, 0, h->pr.p, h->pz.p, n + 1, it == 0 ? 1 : 0, h->pp.p, h->pcg.p);
        MATVEC(h, DC, h->pp.p);
        LAUNCH(h, KID_REF_VEC, k_ref_step<DC>, 1, 1024, 0, h->Sfc, h->Sff.p, Nc, tol2, h->pp.p, h->pq.p, h->pqpart.p, h->px.p, h->pr.p, h->pcg.p);
        SSFM_HIP_CHECK(ctx, hipMemcpyAsync(host_pcg, h->pcg.p, PCG_TOTAL * sizeof(double), hipMemcpyDeviceToHost, st));
        SSFM_HIP_CHECK(ctx, hipStreamSynchronize(st));
        it++;
    }
    *iters_out = it;
    *ok_out = host_pcg[PCG_DONE] != 0.0 && host_pcg[PCG_BREAKDOWN] == 0.0;
    return SSFM_OK;
}


}  // namespace ssfm
