// spherical_sfm_amd -- device-resident Levenberg-Marquardt driver for bundle adjustment.
//
// Replaces what ceres::Solve does for SfM::Optimize (reference src/sfm.cpp:273-289): the trust-region loop
// of Ceres 2.2.0 (TrustRegionMinimizer + LevenbergMarquardtStrategy, restated -- Ceres is not vendored in the
// reference), with the SPARSE_SCHUR direct solve done by an explicit block-sparse Schur complement and an exact block-banded
// Cholesky of the reduced camera system on the GPU (band_kernels2.h, band_sub.h; a block-Jacobi PCG is kept for comparison).
// One hand-over to the host per LM iteration: the last kernel publishes the 16 folded scalars + solver flags into coherent
// pinned memory, the host spins on a sequence number and decides accept/reject, radius update and the termination tests.
//
// Multi-GPU: points are sharded, cameras replicated.  Per LM iteration one RCCL all-reduce of
// the partial reduced system [S | rhs | diag U | S_fc | J_c^T r | scalars] and one of the step scalars.
#include <atomic>
#include <hip/hip_runtime.h>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <vector>
#include "ba_flatten.h"
#include "ba_kernels.h"
#include "band_kernels.h"
#include "knobs.h"
#include "ssfm_ctx.h"

#include "ba_handle.h"

namespace ssfm {

template <int DC>
static int lm_loop(ssfm_ba_handle* h, ssfm_ba_summary* S) {
    ssfm_ctx* ctx = h->ctx; hipStream_t st = ctx->stream;
    const BAFlat& F = h->F; const ssfm_ba_options& O = h->opt;
    const int Nc = F.Nc, nP = F.nP;
    const int loss = O.loss_type; const double la = O.loss_scale;
    const int gp_pts = (nP + 255) / 256, gp_cam = (Nc + 63) / 64;
    // lane-per-point kernels of the LM loop run as single-wave workgroups: 4x more workgroups spread evenly over the CUs
    static const int PTB = SSFM_LAB_KNOB("SSFM_PT_BLOCK", 256);
    const int gp_pts_lm = (nP + PTB - 1) / PTB;
    static const int PLB = SSFM_LAB_KNOB("SSFM_PL_BLOCK", 64);      // k_point_lin has no workgroup-level step: one wave per workgroup spreads best
    const double2* oxy = reinterpret_cast<const double2*>(h->obs_xy.p);
    double* fx = h->focal3.p; double* fc = h->focal3.p + 1;
    double* cam_x = h->cam_x.p; double* cam_c = h->cam_c.p; double* pts_x = h->pts_x.p; double* pts_c = h->pts_c.p;
    double* rot_x = h->rot_x.p; double* rot_c = h->rot_c.p;
    // the only host round trip of an iteration = its scalars [sums | gradient max | solver flags].
    // default: the last kernel of an iteration (k_publish) writes the folded scalars + flags into coherent pinned memory and the host
    // spins on its sequence number (ba_handle.h lm_poll / publish_alloc / wait_published); SSFM_LM_POLL=0: copy of all replicas +
    // stream synchronisation + host fold
    const bool poll = lm_poll() && publish_alloc(h);
    if (!poll && !h->host_sp) SSFM_HIP_CHECK(ctx, hipHostMalloc((void**)&h->host_sp, (SC_NSLOT * SC_TOTAL + PCG_TOTAL + 1) * sizeof(double), hipHostMallocDefault));
    double* host_sp = h->host_sp;
    double* host_scal = poll ? h->host_pub : host_sp; double* host_pcg1 = poll ? h->host_pub + SC_TOTAL : host_sp + SC_NSLOT * SC_TOTAL;
    auto wait_iteration = [&]() -> int {
        if (!poll || h->profile || (h->opt.verbose != 0)) SSFM_HIP_CHECK(ctx, hipStreamSynchronize(st));
        return poll ? wait_published(h) : SSFM_OK;
    };
    auto fold_host_scal = [&]() {                                    // replicas -> replica 0 (sums; the gradient max by max)
        for (int k = 0; k < SC_TOTAL; k++) {
            double a = host_sp[k];
            for (int s2 = 1; s2 < SC_NSLOT; s2++) { const double v = host_sp[(size_t)s2 * SC_TOTAL + k]; a = (k == SC_GMAX) ? std::max(a, v) : a + v; }
            host_sp[k] = a;
        }
    };
    // ---- iteration 0: rotation tables, Jacobi scaling from the initial Jacobian, |x|
    h->set_zone(0);
    if (h->det) {   // a solve starts on clean limbs whatever the previous one on this handle left (normally nothing: every decode clears what it read)
        if (!h->gram_fold) SSFM_HIP_CHECK(ctx, hipMemsetAsync(h->det_limb.p, 0, (2 * h->det_nacc + 2) * sizeof(long long), st));      // (atomics-free emission: no limbs in use)
        SSFM_HIP_CHECK(ctx, hipMemsetAsync(h->det_lacc.p, 0, (size_t)SC_NSLOT * SC_TOTAL * LA_STRIDE * sizeof(long long), st));
    }
    // four launches: [rotation tables + clears] [point column norms + scales] [camera column norms + scales] [|x|^2 + focal scale]
    const bool make_scale = !h->scale_ready;
    LAUNCH(h, KID_CAM_ROT, k_cam_rot0, gp_cam, 64, 0, cam_x, rot_x, Nc, h->scal.p, (int)SC_TOTAL, h->diag_f.p, make_scale ? 1 : 0);
    if (make_scale) {
        if (nP > 0) hipLaunchKernelGGL(k_colnorm_both, dim3(gp_pts + Nc), dim3(256), 0, st, gp_pts, cam_x, rot_x, pts_x, fx, oxy, h->obs_cam.p, h->pt_start.p, nP,
                                       loss, la, h->diag_pt.p, h->diag_f.p, h->mask_pt.p, h->scale_pt.p, O.jacobi_scaling, h->det ? h->det_dfpart.p : (double*)nullptr,
                                       h->obs_pt.p, h->cam_start.p, h->cam_obs.p, h->diag_cam.p, h->mask_cam.p, ctx->collective ? (double*)nullptr : h->scale_cam.p);
        else hipLaunchKernelGGL(k_colnorm_cam, dim3(Nc), dim3(256), 0, st, cam_x, rot_x, pts_x, fx, oxy, h->obs_pt.p, h->cam_start.p, h->cam_obs.p,
                           loss, la, h->diag_cam.p, h->mask_cam.p, ctx->collective ? (double*)nullptr : h->scale_cam.p, O.jacobi_scaling);
        if (ctx->collective) {   // camera / focal column norms are sums over every rank's observations
            int rc = allreduce(h, h->diag_cam.p, (size_t)Nc * 6, ncclSum); if (rc) return rc;
            rc = allreduce(h, h->diag_f.p, 1, ncclSum); if (rc) return rc;
        }
        h->scale_ready = true;
    }
    {
        // (at most 256 + 16 workgroups: each ends in one atomic on the same two scalars, ~12 ns apiece -- 17.6k of them at 1.5 M points took 217 us, 1032 at config 2 15 us)
        const int gpt = nP > 0 ? std::min(256, (nP * 3 + 255) / 256) : 0, gcm = std::min(16, (Nc * 6 + 255) / 256);
        hipLaunchKernelGGL(k_startup_tail, dim3(gpt + gcm), dim3(256), 0, st, pts_x, h->mask_pt.p, nP * 3, gpt, cam_x, h->mask_cam.p, Nc * 6, fx, h->mask_f.p,
                           h->diag_cam.p, (make_scale && ctx->collective) ? h->scale_cam.p : (double*)nullptr, h->diag_f.p, make_scale ? h->scale_f.p : (double*)nullptr,
                           O.jacobi_scaling, h->scal.p + SC_X0N2_PT, h->scal.p + SC_X0N2_CAM, (make_scale && h->det && nP > 0) ? h->det_dfpart.p : (const double*)nullptr, gp_pts,
                           (const double*)h->scal.p, h->det ? h->det_lacc.p : (long long*)nullptr);
        if (h->det) LAUNCH(h, KID_DET_DECODE, k_det_decode, 1, 256, 0, h->S_val, h->det_limb.p + 1, h->det_nacc, 0, h->scal.p, h->det_lacc.p, (1u << SC_X0N2_PT) | (1u << SC_X0N2_CAM));
    }
    { int rc = allreduce(h, h->scal.p + SC_X0N2_PT, 1, ncclSum); if (rc) return rc; }
    SSFM_HIP_CHECK(ctx, hipMemcpyAsync(host_scal, h->scal.p, SC_TOTAL * sizeof(double), hipMemcpyDeviceToHost, st));
    // |x_0| is first needed when iteration 0's tail is enqueued (the device-side gate takes it as an argument): the host waits for the copy THERE, behind the kernels of
    // iteration 0's assembly and solve -- a stream synchronisation here left the GPU idle for the host's wake-up and first launches (~20 us per solve)
    if (!h->x0_ev) SSFM_HIP_CHECK(ctx, hipEventCreateWithFlags(&h->x0_ev, hipEventDisableTiming));
    SSFM_HIP_CHECK(ctx, hipEventRecord(h->x0_ev, st));
    double x_norm = 0.0; bool x0_pending = true;
    auto need_x0 = [&]() -> int {
        if (!x0_pending) return SSFM_OK;
        SSFM_HIP_CHECK(ctx, hipEventSynchronize(h->x0_ev));
        x_norm = std::sqrt(host_scal[SC_X0N2_PT] + host_scal[SC_X0N2_CAM]); x0_pending = false;
        return SSFM_OK;
    };

    SSFM_HIP_CHECK(ctx, hipMemsetAsync(h->zone.p + h->zone_len, 0, h->zone_len * sizeof(double), st));   // zone of iteration 1
    double radius = O.initial_trust_region_radius, decrease_factor = 2.0;
    double x_cost = 0.0, minimum_cost = std::numeric_limits<double>::max();
    int iteration = 0, num_invalid = 0;
    bool last_successful = true;
    S->num_successful_steps = 1; S->num_unsuccessful_steps = 0; S->num_linearizations = 0; S->pcg_iterations_total = 0;
    S->termination = SSFM_NO_CONVERGENCE;
    float ms_lin = 0, ms_schur = 0, ms_pcg = 0, ms_upd = 0;
    // Speculation: k_publish also decides "accepted, go on" + the next radius on the device (LmGate), and the next iteration's
    // k_point_lin is queued right behind it, reading both from device memory -- it runs while the host wakes up, decides and
    // launches the rest.  The host's own decision stays authoritative: a speculative launch that should not have run only touched
    // scratch arrays and the next zone, which is then cleared again.  SSFM_LM_SPECULATE=0 turns it off.
    const char* e_spec = std::getenv("SSFM_LM_SPECULATE");
    const bool spec_on = poll && nP > 0 && !(e_spec && std::atoi(e_spec) == 0);
    bool lin_done = false, spec_launched = false;
    const int gram_bs_env = std::getenv("SSFM_GRAM_BACKSUB") ? std::atoi(std::getenv("SSFM_GRAM_BACKSUB")) : -1;      // read once per solve (tests switch it between solves)
    // Round 5 EXPERIMENT, off (SSFM_GRAM_FUSE=1): when every point sits in a signature group, k_schur_gram does the point pass itself (ba_kernels.h: FUSE) and k_point_lin
    // does not run; the speculative launch behind k_publish is then the Gram kernel of the NEXT iteration, accumulating into the next zone.  Correct (parity 3e-11, same
    // iterations) and SLOWER: 97.6 us against 43.7 + 18.3 at config 2, 1357 against 409 + 179 at the configs[4] size -- the Gram kernel sits at its register limit
    // (256 at two waves per SIMD: 39 camera sums, the tile accumulators, one observation's linearisation) and the point pass's fold spills 43-80 of them to scratch
    // (profiles/r05_notes.md).  One launch less is not worth a kernel that leaves its registers.
    static const bool gram_fuse_env = SSFM_LAB_KNOB("SSFM_GRAM_FUSE", 0) != 0;
    const bool fuse_lin = gram_fuse_env && nP > 0 && !F.gr_rec.empty() && F.gram_points == (int64_t)nP && F.chunk_cam.empty() && F.cs_task_cam.empty();
    struct ZonePtrs { double *scal, *S_val, *rhs, *Udiag, *Sfc, *gcraw; };
    auto zone_ptrs = [&](int which) { ZonePtrs z; z.scal = h->zone.p + (size_t)which * h->zone_len; double* red = z.scal + h->scal.n + h->pcg.n;
                                      z.S_val = red; z.rhs = z.S_val + h->zone_nnz; z.Udiag = z.rhs + (h->zone_n + 1); z.Sfc = z.Udiag + h->zone_n; z.gcraw = z.Sfc + h->zone_n; return z; };
    // SSFM_DETERMINISTIC=1 (det_acc.h): where the limbs of a zone's accumulators live; the scalar block's long accumulators; the decode launches
    long long* const lacc = h->det ? h->det_lacc.p : (long long*)nullptr;
    auto det_zone = [&](const ZonePtrs& z) { DetZone dz; if (h->det) { dz.base = z.S_val; dz.limb = h->det_limb.p + 1; } if (h->gram_fold) { dz.part = h->gram_part.p; dz.part_off = h->gpart_off.p; } return dz; };
    constexpr unsigned DET_K_ASSEMBLY = (1u << SC_COST) | (1u << SC_FJJ) | (1u << SC_FJR) | (1u << SC_FWW) | (1u << SC_FWG);
    constexpr unsigned DET_K_TAIL = (1u << SC_MODEL) | (1u << SC_STEP2_PT) | (1u << SC_XN2_PT) | (1u << SC_CAND_COST) | (1u << SC_STEP2_CAM) | (1u << SC_XN2_CAM);
    auto det_decode = [&](const ZonePtrs& z, bool matrices, unsigned kmask) {
        const int gz = matrices ? (int)std::min<size_t>(512, (h->det_nacc + 255) / 256) : 0;
        LAUNCH(h, KID_DET_DECODE, k_det_decode, gz + 1, 256, 0, z.S_val, h->det_limb.p + 1, h->det_nacc, gz, z.scal, h->det_lacc.p, kmask);
    };
    // EXPERIMENT, off (SSFM_PUBLISH_FUSED=1): the end-of-iteration hand-over in the last workgroup of k_point_backsub (arrival ticket) instead of a k_publish
    // launch.  Measured: k_point_backsub 22.8 -> 55.9 us at config 2 and 380 -> 1290 us at the configs[4] size -- every workgroup needs an agent-scope release
    // fence (an L2 write-back on this multi-XCD part) + a same-address atomic before it may leave, which costs far more than the 4.7 us launch it saves.
    static const bool fused_publish = SSFM_LAB_KNOB("SSFM_PUBLISH_FUSED", 0) != 0;
    if (fused_publish && poll && !h->pub_ticket.p) {
        SSFM_HIP_CHECK(ctx, h->pub_ticket.alloc(1)); SSFM_HIP_CHECK(ctx, hipMemsetAsync(h->pub_ticket.p, 0, sizeof(int), st));
    }
    static const bool zone_clear_fused = SSFM_LAB_KNOB("SSFM_ZONE_CLEAR_FUSED", 1) != 0;
    // per-phase device times (summary.t_kernel_*_ms) cost five event records and four queries per iteration, the queries on the
    // host's critical path between two iterations: only with profiling on (ssfm_ba_set_profiling) or options.verbose
    const bool phases = h->profile || O.verbose;
    if (phases) for (auto& e : h->phase_ev) if (!e) SSFM_HIP_CHECK(ctx, hipEventCreate(&e));

    // SSFM_PAIRS_Y_PROBE=1: the half-product variant of the pair pass runs next to the real kernel, on scratch data, for rocprofv3
    const bool y_probe = SSFM_LAB_KNOB("SSFM_PAIRS_Y_PROBE", 0) != 0;
    DevBuf<double> probe_Y, probe_S;
    if (y_probe) {
        SSFM_HIP_CHECK(ctx, probe_Y.alloc((size_t)F.M * DC * 3)); SSFM_HIP_CHECK(ctx, probe_S.alloc(h->zone_nnz));
        SSFM_HIP_CHECK(ctx, hipMemsetAsync(probe_Y.p, 0x3c, (size_t)F.M * DC * 3 * sizeof(double), st)); SSFM_HIP_CHECK(ctx, hipMemsetAsync(probe_S.p, 0, h->zone_nnz * sizeof(double), st));
    }
    while (true) {
        if (iteration >= O.max_num_iterations) { S->termination = SSFM_NO_CONVERGENCE; break; }
        if (radius <= O.min_trust_region_radius) { S->termination = SSFM_CONVERGENCE; break; }
        iteration++;
        // ================= assemble at x with the current radius =================
        // this iteration's zone (scalars, solver flags, [S | rhs | diag U | S_fc | Jc^T r | sums]) was zeroed behind the previous iteration
        h->set_zone(iteration & 1);
        if (phases) SSFM_HIP_CHECK(ctx, hipEventRecord(h->phase_ev[0], st));
        // per-kernel profiling: the first launch after the host's hand-over would carry the queue's wake-up inside its event bracket (k_point_lin read 25.7 us
        // against 14.7 us under rocprofv3); an empty launch takes that, the bracket of the real kernel then starts behind it like every other one
        if (h->profile && nP > 0 && !lin_done) hipLaunchKernelGGL(k_profile_pad, dim3(1), dim3(64), 0, st);
        if (nP > 0 && !lin_done && !fuse_lin)             // (lin_done: it ran speculatively behind the previous iteration, with this radius)
            LAUNCH(h, KID_POINT_LIN, k_point_lin<3>, (nP + PLB - 1) / PLB, PLB, 0, cam_x, rot_x, pts_x, fx, oxy, h->obs_cam.p, h->pt_start.p, nP, h->scale_pt.p,
                   h->scale_f.p, loss, la, radius, O.min_lm_diagonal, O.max_lm_diagonal, h->Vs.p, h->gp.p, h->scal.p, (const double*)nullptr, lacc);
        if (phases) SSFM_HIP_CHECK(ctx, hipEventRecord(h->phase_ev[1], st));
        if (!F.cs_task_cam.empty()) {
            const int ntasks = (int)F.cs_task_cam.size();
            LAUNCH(h, KID_CAM_SUMS, k_cam_sums2<DC>, (ntasks + 3) / 4, 256, 0, cam_x, rot_x, pts_x, fx, oxy, h->cam_obs.p, h->cam_obs_pt.p, h->cs_task_cam.p,
                   h->cs_task_q0.p, h->cs_task_q1.p, ntasks, h->row_ptr.p, h->diag_slot.p, h->scale_cam.p, h->scale_f.p, h->Vs.p, loss, la, h->S_val, h->rhs,
                   h->Udiag, h->Sfc, h->gcraw, (const unsigned char*)(F.gram_points > 0 ? h->pt_grouped.p : nullptr), det_zone(zone_ptrs(iteration & 1)));
        }
        if (!F.chunk_cam.empty()) {
            const int ntasks = (int)F.chunk_cam.size();
#ifdef SSFM_LAB
            static const bool occ3 = SSFM_LAB_KNOB("SSFM_PAIRS_OCC3", 0) != 0;      // experiment (ba_kernels.h): three waves per SIMD, slower
            if (occ3 && DC == 6)
                LAUNCH(h, KID_SCHUR_ROWS, (k_schur_pairs2<DC, 3>), (ntasks + 3) / 4, 256, 0, cam_x, rot_x, pts_x, fx, oxy, h->row_ptr.p, h->col_idx.p, h->chunk_cam.p,
                       h->chunk_b0.p, h->chunk_b1.p, ntasks, h->batch_slot.p, h->pair_j.p, h->pair_j2.p, h->pair_p.p, h->scale_cam.p, h->Vs.p, loss, la, h->S_val);
            else
#endif
            LAUNCH(h, KID_SCHUR_ROWS, k_schur_pairs2<DC>, (ntasks + 3) / 4, 256, 0, cam_x, rot_x, pts_x, fx, oxy, h->row_ptr.p, h->col_idx.p, h->chunk_cam.p,
                   h->chunk_b0.p, h->chunk_b1.p, ntasks, h->batch_slot.p, h->pair_j.p, h->pair_j2.p, h->pair_p.p, h->scale_cam.p, h->Vs.p, loss, la, h->S_val,
                   det_zone(zone_ptrs(iteration & 1)));
        }
        // signature groups: Gram products on the matrix cores (ba_kernels.h: k_schur_gram); one launch per tile class.  xc / xr / xp / xf = the state it linearises at,
        // z = the zone it accumulates into, spec = device-side [go, radius] of a speculative launch (fused point pass only)
        auto launch_gram = [&](const double* xc, const double* xr, const double* xp, const double* xf, const ZonePtrs& z, double rad, const double* spec) -> int {
            const int ng = (int)(F.gr_rec.size() / GRAM_REC);
            static bool gram_stamps_done = SSFM_LAB_KNOB("SSFM_GRAM_STAMPS", 0) == 0;      // timing study: per-task phase stamps of the first launch
            long long* gram_dbg = nullptr;
            if (!gram_stamps_done) (void)hipMalloc((void**)&gram_dbg, (size_t)4 * ng * sizeof(long long));
            static const bool gram_t4 = SSFM_LAB_KNOB("SSFM_GRAM_T4", 1) != 0;
            static const int gram_waves = std::min(4, std::max(1, SSFM_LAB_KNOB("SSFM_GRAM_WAVES", 1)));   // waves (tasks) per workgroup: 1 measured best (2: +14 %, 4: +13 % at the configs[4] size)
            // one launch per tile class (the tasks are sorted by K, i.e. by rows = DC K): rows <= 16 -> 1 row tile of 16; 17..20 -> 1 tile + a tail of <= 4 rows through
            // the 4x4x4 instruction; <= 32 -> 2 tiles; 33..36 -> 2 tiles + tail; else 3 tiles
            auto tile_class = [&](int K) { const int rows = DC * K; return rows <= 16 ? 0 : (rows <= 20 && gram_t4) ? 1 : rows <= 32 ? 2 : (rows <= 36 && gram_t4) ? 3 : 4; };
            int cls_end[5] = {0, 0, 0, 0, 0};
            for (int t = 0; t < ng; t++) { const int c0 = tile_class(F.gr_rec[(size_t)t * GRAM_REC + 2]); for (int c = c0; c < 5; c++) cls_end[c] = t + 1; }
            GramFuse fz; fz.scale_pt = h->scale_pt.p; fz.radius = rad; fz.min_diag = O.min_lm_diagonal; fz.max_diag = O.max_lm_diagonal; fz.PS_out = h->Vs.p; fz.gp_out = h->gp.p; fz.scal = z.scal; fz.emit_skip = SSFM_LAB_KNOB("SSFM_GRAM_EMIT_SKIP", 0); fz.spec = spec;
#define SSFM_GRAM_LAUNCH_(CLS_, NT_, TI_, FUSE_)                                                                                                       \
            do {                                                                                                                                       \
                const int t0 = (CLS_ == 0) ? 0 : cls_end[(CLS_ >= 1) ? CLS_ - 1 : 0], t1 = cls_end[CLS_];                                             \
                if (t1 > t0) {                                                                                                                         \
                    const int rows_alloc = DC * F.gr_rec[(size_t)(t1 - 1) * GRAM_REC + 2];              /* the largest K of the class: its last task */ \
                    const size_t gram_lds = (size_t)gram_waves * ((size_t)rows_alloc * GRAM_LD + GRAM_TAIL) * sizeof(double);                          \
                    if (gram_lds > 48 * 1024)                                                                                                          \
                        SSFM_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_schur_gram<DC, NT_, TI_, FUSE_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)gram_lds)); \
                    LAUNCH(h, KID_SCHUR_GRAM, (k_schur_gram<DC, NT_, TI_, FUSE_>), (t1 - t0 + gram_waves - 1) / gram_waves, 64 * gram_waves, gram_lds, xc, xr, xp, xf, oxy, t1, h->gr_rec.p,  \
                           h->scale_cam.p, h->scale_f.p, h->Vs.p, loss, la, rows_alloc, F.focal_free ? 1 : 0, t0, z.S_val, z.rhs, z.Udiag, z.Sfc, z.gcraw, gram_dbg, fz, det_zone(z));  \
                }                                                                                                                                      \
            } while (0)
#ifdef SSFM_LAB
#define SSFM_GRAM_LAUNCH(CLS_, NT_, TI_) do { if (fuse_lin) SSFM_GRAM_LAUNCH_(CLS_, NT_, TI_, true); else SSFM_GRAM_LAUNCH_(CLS_, NT_, TI_, false); } while (0)
#else
#define SSFM_GRAM_LAUNCH(CLS_, NT_, TI_) SSFM_GRAM_LAUNCH_(CLS_, NT_, TI_, false)
#endif
            int n_cls = 0; for (int c = 0; c < 5; c++) if (cls_end[c] > (c ? cls_end[c - 1] : 0)) n_cls++;
            const bool gram_any = F.gram_any;                                 // SSFM_GRAM_ANY as the PLAN read it (one value for the cost model and the launch; 0: one launch per tile class)
            bool any_ok = gram_any && n_cls > 1 && gram_waves == 1 && !gram_dbg;
#ifdef SSFM_LAB
            any_ok = any_ok && !fuse_lin;
#endif
            if (any_ok) {
                // tracks of mixed length: every tile class in ONE launch (ba_kernels.h: k_schur_gram_any), LDS for the largest class
                const int rows_alloc = DC * F.gr_rec[(size_t)(ng - 1) * GRAM_REC + 2];
                const size_t gram_lds = ((size_t)rows_alloc * GRAM_LD + GRAM_TAIL) * sizeof(double);
                if (gram_lds > 48 * 1024) SSFM_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_schur_gram_any<DC>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)gram_lds));
                LAUNCH(h, KID_SCHUR_GRAM, (k_schur_gram_any<DC>), ng, 64, gram_lds, xc, xr, xp, xf, oxy, ng, h->gr_rec.p, h->scale_cam.p, h->scale_f.p, h->Vs.p, loss, la, rows_alloc,
                       F.focal_free ? 1 : 0, gram_t4 ? 1 : 0, z.S_val, z.rhs, z.Udiag, z.Sfc, z.gcraw, det_zone(z));
            } else {
                SSFM_GRAM_LAUNCH(0, 1, 0); SSFM_GRAM_LAUNCH(1, 1, 2); SSFM_GRAM_LAUNCH(2, 2, 0);
                if (DC == 6) { SSFM_GRAM_LAUNCH(3, 2, 3); SSFM_GRAM_LAUNCH(4, 3, 0); }
            }
#undef SSFM_GRAM_LAUNCH
#undef SSFM_GRAM_LAUNCH_
            if (gram_dbg) {                                        // print the phase times of this launch (100 MHz clock) and stop stamping
                std::vector<long long> st2((size_t)4 * ng); (void)hipStreamSynchronize(h->ctx->stream); (void)hipMemcpy(st2.data(), gram_dbg, st2.size() * sizeof(long long), hipMemcpyDeviceToHost);
                long long tmin = st2[0], tmax = 0; double a = 0, b = 0, c = 0;
                for (int t = 0; t < ng; t++) { tmin = std::min(tmin, st2[4 * t]); tmax = std::max(tmax, st2[4 * t + 3]); a += st2[4 * t + 1] - st2[4 * t]; b += st2[4 * t + 2] - st2[4 * t + 1]; c += st2[4 * t + 3] - st2[4 * t + 2]; }
                std::fprintf(stderr, "[gram] %d tasks, kernel span %.1f us; mean per task: start + first linearisation %.2f us, tiles + other sub-chunks %.2f us, emission %.2f us\n", ng, (tmax - tmin) * 0.01,
                             a / ng * 0.01, b / ng * 0.01, c / ng * 0.01);
                { double s_mean = 0, s_max = 0, e_mean = 0, a_max = 0, b_max = 0, c_max = 0;
                  for (int t = 0; t < ng; t++) { const double so = (st2[4 * t] - tmin) * 0.01; s_mean += so; s_max = std::max(s_max, so); e_mean += (st2[4 * t + 3] - tmin) * 0.01;
                                                 a_max = std::max(a_max, (st2[4 * t + 1] - st2[4 * t]) * 0.01); b_max = std::max(b_max, (st2[4 * t + 2] - st2[4 * t + 1]) * 0.01); c_max = std::max(c_max, (st2[4 * t + 3] - st2[4 * t + 2]) * 0.01); }
                  std::fprintf(stderr, "[gram] task start after the first: mean %.2f max %.2f us; task end mean %.2f us; longest phases %.2f | %.2f | %.2f us\n", s_mean / ng, s_max, e_mean / ng, a_max, b_max, c_max); }
                (void)hipFree(gram_dbg); gram_dbg = nullptr; gram_stamps_done = true;
            }
            return SSFM_OK;
        };
        if (!F.gr_rec.empty() && !(fuse_lin && lin_done)) {          // (fused + lin_done: the whole assembly of this iteration ran speculatively behind the previous one)
            const int rc = launch_gram(cam_x, rot_x, pts_x, fx, zone_ptrs(iteration & 1), radius, nullptr); if (rc) return rc;
        }
        lin_done = false;
#ifdef SSFM_LAB
        if (y_probe && !F.chunk_cam.empty()) {                     // experiment only (ba_kernels.h: k_pairs_y_probe)
            const int ntasks = (int)F.chunk_cam.size();
            hipLaunchKernelGGL(k_pairs_y_probe<DC>, dim3((ntasks + 3) / 4), dim3(256), 0, st, h->row_ptr.p, h->col_idx.p, h->chunk_cam.p, h->chunk_b0.p, h->chunk_b1.p, ntasks,
                               h->batch_slot.p, h->pair_j.p, h->pair_j2.p, h->scale_cam.p, probe_Y.p, probe_S.p);
        }
#endif
        if (h->det && !h->gram_fold) {
            det_decode(zone_ptrs(iteration & 1), true, DET_K_ASSEMBLY);      // limbs -> the zone's doubles (and cleared); the point pass's sums -> replica 0 of the scalar block
            SSFM_HIP_CHECK(ctx, hipMemsetAsync(h->det_limb.p, 0, sizeof(long long), st));      // the poison word: read by every decode workgroup, cleared behind the launch
        }
        // (atomics-free emission, gram_fold: nothing went through the limbs; the focal sums are folded from the long accumulators by k_finalize_gather, the cost by k_publish:
        //  no decode launch at all)
        if (ctx->collective) {
            // ONE sum all-reduce per assembly: [S | rhs | diag U | S_fc | Jc^T r | scalar sums | one gradient-max slot per rank]
            hipLaunchKernelGGL(k_scal_fold, dim3(1), dim3(SC_TOTAL * 64), 0, st, h->scal.p, h->red_scal, ctx->rank);
            int rc = allreduce(h, h->redbuf.p, (size_t)h->n_red, ncclSum); if (rc) return rc;
            hipLaunchKernelGGL(k_scal_unpack, dim3(1), dim3(64), 0, st, h->scal.p, h->red_scal, ctx->nranks);
        }
        // the next iteration's zone is cleared by the finalize kernel (a slice per workgroup) when its length allows 16-byte stores
        double* next_zone = h->zone.p + (size_t)((iteration + 1) & 1) * h->zone_len;
        const bool clear_in_finalize = O.preconditioner == 0 && zone_clear_fused && (h->zone_len % 2 == 0) && ((reinterpret_cast<uintptr_t>(next_zone) & 15) == 0);
        double2* clear_next = clear_in_finalize ? reinterpret_cast<double2*>(next_zone) : (double2*)nullptr;
        const size_t clear_len2 = clear_in_finalize ? h->zone_len / 2 : 0;
        const double* fold_part = h->gram_fold ? h->gram_part.p : (const double*)nullptr;      // atomics-free Gram emission: k_finalize_gather folds the tasks' partial blocks first
        if (O.preconditioner == 0) {                               // finalize + band gather + rhs permutation in one launch
            if (F.band_block != DC)      // 3-dof cameras merged in pairs into 6x6 block rows of the band
                { if (h->gram_fold) LAUNCH(h, KID_FINALIZE, (k_finalize_gather<DC, true, true>), Nc, 256, 0, h->row_ptr.p, h->col_idx.p, h->diag_slot.p, h->scale_cam.p, h->scale_f.p, h->Udiag, h->gcraw,
                       radius, O.min_lm_diagonal, O.max_lm_diagonal, Nc, h->cam_pos.p, h->cam_pos2.p, h->pair_dummy.p, F.y_rows(DC), F.band, h->S_val, h->rhs, h->Sfc, h->Sff.p, h->band.p, h->Yb.p, h->scal.p, clear_next, clear_len2, h->col_pos.p, h->wrap_ptr_p(), h->wrap_blk.p, h->wrap_row2.p,
                       fold_part, (const int*)nullptr, h->fold_slot_src.p, h->Udiag, h->gcraw, h->Sfc, h->gram_fold ? lacc : (long long*)nullptr); else LAUNCH(h, KID_FINALIZE, (k_finalize_gather<DC, true>), Nc, 256, 0, h->row_ptr.p, h->col_idx.p, h->diag_slot.p, h->scale_cam.p, h->scale_f.p, h->Udiag, h->gcraw,
                       radius, O.min_lm_diagonal, O.max_lm_diagonal, Nc, h->cam_pos.p, h->cam_pos2.p, h->pair_dummy.p, F.y_rows(DC), F.band, h->S_val, h->rhs, h->Sfc, h->Sff.p, h->band.p, h->Yb.p, h->scal.p, clear_next, clear_len2, h->col_pos.p, h->wrap_ptr_p(), h->wrap_blk.p, h->wrap_row2.p,
                       fold_part, (const int*)nullptr, h->fold_slot_src.p, h->Udiag, h->gcraw, h->Sfc, h->gram_fold ? lacc : (long long*)nullptr); }
            else
                { if (h->gram_fold) LAUNCH(h, KID_FINALIZE, (k_finalize_gather<DC, false, true>), Nc, 256, 0, h->row_ptr.p, h->col_idx.p, h->diag_slot.p, h->scale_cam.p, h->scale_f.p, h->Udiag, h->gcraw,
                       radius, O.min_lm_diagonal, O.max_lm_diagonal, Nc, h->cam_pos.p, h->cam_pos2.p, h->pair_dummy.p, F.y_rows(DC), F.band, h->S_val, h->rhs, h->Sfc, h->Sff.p, h->band.p, h->Yb.p, h->scal.p, clear_next, clear_len2, h->col_pos.p, h->wrap_ptr_p(), h->wrap_blk.p, h->wrap_row2.p,
                       fold_part, (const int*)nullptr, h->fold_slot_src.p, h->Udiag, h->gcraw, h->Sfc, h->gram_fold ? lacc : (long long*)nullptr); else LAUNCH(h, KID_FINALIZE, (k_finalize_gather<DC, false>), Nc, 256, 0, h->row_ptr.p, h->col_idx.p, h->diag_slot.p, h->scale_cam.p, h->scale_f.p, h->Udiag, h->gcraw,
                       radius, O.min_lm_diagonal, O.max_lm_diagonal, Nc, h->cam_pos.p, h->cam_pos2.p, h->pair_dummy.p, F.y_rows(DC), F.band, h->S_val, h->rhs, h->Sfc, h->Sff.p, h->band.p, h->Yb.p, h->scal.p, clear_next, clear_len2, h->col_pos.p, h->wrap_ptr_p(), h->wrap_blk.p, h->wrap_row2.p,
                       fold_part, (const int*)nullptr, h->fold_slot_src.p, h->Udiag, h->gcraw, h->Sfc, h->gram_fold ? lacc : (long long*)nullptr); }
            h->band_filled = true;
        } else {
            LAUNCH(h, KID_FINALIZE, k_finalize_S<DC>, gp_cam, 64, 0, h->row_ptr.p, h->diag_slot.p, h->scale_cam.p, h->scale_f.p, h->Udiag, h->gcraw,
                   radius, O.min_lm_diagonal, O.max_lm_diagonal, Nc, h->S_val, h->Minv.p, h->rhs, h->Sff.p, h->scal.p);
        }
        if (phases) SSFM_HIP_CHECK(ctx, hipEventRecord(h->phase_ev[2], st));
        // ================= solve the reduced system, then step / candidate / model cost / candidate cost =================
        // The direct solve and the tail are enqueued back to back; the solver's residual flags come back with the
        // iteration scalars in ONE host synchronisation.  Only if the residual test failed (rare) does PCG refinement
        // run and the tail get redone.
        int pcg_iters = 0; bool pcg_ok = false;
        // with the banded factor and the lower-triangle storage the arrow kernel of the solve also writes the candidate cameras
        const bool fused_cams = O.preconditioner == 0 && F.sym_lower;
        const bool residual_later = fused_cams && nP > 0;
        h->tail.on = fused_cams; h->tail.residual_later = residual_later; h->tail.cam = cam_x; h->tail.focal = fx; h->tail.cam_c = cam_c; h->tail.focal_c = fc; h->tail.rot_c = rot_c;
        { int rc = solve_reduced<DC>(h, host_pcg1, &pcg_iters, &pcg_ok, 0); h->tail.on = false; h->tail.residual_later = false; if (rc) return rc; }
        if (phases) SSFM_HIP_CHECK(ctx, hipEventRecord(h->phase_ev[3], st));
        auto enqueue_tail = [&](bool with_cams) -> int {
            { const int rc0 = need_x0(); if (rc0) return rc0; }
            // candidate cameras + their rotation tables; then per point: back-substitution, candidate, model cost change, candidate cost
            if (with_cams) {
                LAUNCH(h, KID_CAM_UPDATE, k_cam_update<DC>, 1, 1024, 0, cam_x, fx, h->scale_cam.p, h->scale_f.p, h->px.p, Nc, cam_c, fc, rot_c, h->scal.p, Nc <= 1024 ? 1 : 0);
                if (Nc > 1024) LAUNCH(h, KID_CAM_ROT, k_cam_rot, gp_cam, 64, 0, cam_c, rot_c, Nc);
            }
            bool published = false;
            if (nP > 0) {
                const bool res = !with_cams && residual_later;      // first pass of the iteration: the residual check rides along
                // hand-over fused into this launch (its last workgroup publishes): single rank, polling hand-over, first pass of the iteration
                const bool fuse_pub = fused_publish && poll && !ctx->collective && res && h->pub_ticket.p;
                if (fuse_pub) {
                    LmGate g; std::memset(&g, 0, sizeof(g)); double* specp = nullptr;
                    spec_launched = false;
                    if (spec_on && !(h->profile || O.verbose) && iteration < O.max_num_iterations) {
                        g.enabled = 1; g.last_successful = last_successful ? 1 : 0; g.radius = radius; g.x_norm = x_norm;
                        g.function_tolerance = O.function_tolerance; g.gradient_tolerance = O.gradient_tolerance; g.parameter_tolerance = O.parameter_tolerance;
                        g.min_relative_decrease = O.min_relative_decrease; g.max_radius = O.max_trust_region_radius; g.min_radius = O.min_trust_region_radius;
                        specp = h->lmdev.p; spec_launched = true;
                    }
                    LAUNCH(h, KID_BACKSUB, k_point_backsub<DC>, gp_pts_lm + 1, PTB, 0, cam_x, rot_x, pts_x, fx, oxy, h->obs_cam.p, h->pt_start.p, nP, h->scale_cam.p,
                           h->scale_pt.p, h->scale_f.p, h->Vs.p, h->gp.p, h->px.p, Nc, loss, la, cam_c, rot_c, fc, pts_c, h->scal.p,
                           h->rhs, h->pq.p, h->Sfc, h->Sff.p, O.pcg_tolerance * O.pcg_tolerance, h->pr.p, h->pcg.p, (const unsigned char*)nullptr,
                           h->pub_ticket.p, h->host_pub, ++ctx->pub_seq, g, specp);
                    published = true;
                } else {
                    // the points of signature groups take k_gram_backsub (lane = observation, camera records in LDS); k_point_backsub keeps the others and
                    // the residual check's extra workgroup
                    // Its time follows the POINTS (8 lanes per point whatever K), k_point_backsub's the observations: measured (rounds 3-5, k_gram_backsub) 27.3 / 28.1 us
                    // at K = 6 / 8 against 23.4 / 31.7 (100k points) and 271 / 278 against 284 / 379 (1.5 M points).  Round 6, k_gram_backsub2: 195 us at 1.5 M points (K = 8),
                    // 21.1 against 23.0 us at config 2 (K = 6, four waves per task): on from 6 observations per point on average.
                    // SSFM_GRAM_BACKSUB=0 / 1 forces it off / on.
                    const int gram_bs = gram_bs_env;
                    const bool grouped = !fused_publish && !F.gr_rec.empty() && (gram_bs < 0 ? F.gram_obs >= 6 * F.gram_points : gram_bs != 0);
                    const bool all_grouped = grouped && F.gram_points == F.nP;               // then the residual check's workgroup rides with k_gram_backsub
                    if (grouped) {
                        const int ng = (int)(F.gr_rec.size() / GRAM_REC);
                        const bool res_here = res && all_grouped;
                        // waves per task: a small problem (config 2: 900 tasks on 1024 SIMDs) runs with several waves per SIMD, each with a share of its task's sub-chunks
                        // (config 2, hipEvent: 1 wave per task 26.3 us, 2: 24, 4: 21.1, 6 / 8: 24; k_point_backsub 23.0)
                        const int gbs_split_env = std::getenv("SSFM_GBS_SPLIT") ? std::atoi(std::getenv("SSFM_GBS_SPLIT")) : 0;      // (read per launch: tests switch it)
                        const int gbs_wave_target = 14 * ctx->num_cus;      // ~3.5 waves per SIMD in flight (config 2: 900 tasks x 4; 1200 tasks x 1 took 31 us, x 3: see profiles/r06_notes.md r06i)
                        const int gbs_split = gbs_split_env > 0 ? std::min(8, gbs_split_env) : std::max(1, std::min(4, (gbs_wave_target + ng - 1) / std::max(1, ng)));
                        LAUNCH(h, KID_GRAM_BACKSUB, k_gram_backsub2<DC>, (ng * gbs_split + GBS_WAVES - 1) / GBS_WAVES + (res_here ? 1 : 0), 64 * GBS_WAVES, GBS_WAVES * GBS2_TAIL * sizeof(double), cam_x, rot_x, pts_x, fx, oxy, ng, h->gr_rec.p,
                               h->scale_cam.p, h->scale_f.p, h->Vs.p, h->px.p, Nc, loss, la, cam_c, rot_c, fc, pts_c, h->scal.p,
                               h->rhs, h->pq.p, h->Sfc, h->Sff.p, O.pcg_tolerance * O.pcg_tolerance, res_here ? h->pr.p : (double*)nullptr, h->pcg.p, lacc, gbs_split);
                    }
                    // EXPERIMENT, off (SSFM_BACKSUB_LPP=2): two lanes per point -- half the dependent camera gathers per lane, twice the waves.  Measured at config 2
                    // (scripts/lab/ab_lpp.sh, hipEvent averages): 25.1-25.3 us against 23.0-23.5 with one lane per point; 2.557 against 2.538-2.552 ms per solve
                    // r05ai: on LONG tracks it pays -- 300 cameras, tracks of 3 ... 14 (8.5 observations per point, 70 000 points = one wave per SIMD): 40.1 -> 33.0 us;
                    // tracks 3 ... 8 (5.5 per point): 26.7 -> 28.7.  Two lanes per point from 8.25 observations per point on.
                    static const int bs_lpp_env = SSFM_LAB_KNOB("SSFM_BACKSUB_LPP", 0);
                    const int bs_lpp = bs_lpp_env > 0 ? bs_lpp_env : ((int64_t)4 * F.M > (int64_t)33 * F.nP ? 2 : 1);
                    if (!all_grouped && bs_lpp == 2)
                        LAUNCH(h, KID_BACKSUB, (k_point_backsub<DC, 2>), (2 * nP + PTB - 1) / PTB + (res ? 1 : 0), PTB, 0, cam_x, rot_x, pts_x, fx, oxy, h->obs_cam.p, h->pt_start.p, nP, h->scale_cam.p,
                               h->scale_pt.p, h->scale_f.p, h->Vs.p, h->gp.p, h->px.p, Nc, loss, la, cam_c, rot_c, fc, pts_c, h->scal.p,
                               h->rhs, h->pq.p, h->Sfc, h->Sff.p, O.pcg_tolerance * O.pcg_tolerance, res ? h->pr.p : (double*)nullptr, h->pcg.p,
                               (const unsigned char*)(grouped ? h->pt_grouped.p : nullptr), (int*)nullptr, (double*)nullptr, 0ull, LmGate(), (double*)nullptr, lacc);
                    else if (!all_grouped)
                        LAUNCH(h, KID_BACKSUB, k_point_backsub<DC>, gp_pts_lm + (res ? 1 : 0), PTB, 0, cam_x, rot_x, pts_x, fx, oxy, h->obs_cam.p, h->pt_start.p, nP, h->scale_cam.p,
                               h->scale_pt.p, h->scale_f.p, h->Vs.p, h->gp.p, h->px.p, Nc, loss, la, cam_c, rot_c, fc, pts_c, h->scal.p,
                               h->rhs, h->pq.p, h->Sfc, h->Sff.p, O.pcg_tolerance * O.pcg_tolerance, res ? h->pr.p : (double*)nullptr, h->pcg.p,
                               (const unsigned char*)(grouped ? h->pt_grouped.p : nullptr), (int*)nullptr, (double*)nullptr, 0ull, LmGate(), (double*)nullptr, lacc);
                }
            }
            if (published) return SSFM_OK;
            // deterministic mode: the tail's sums (model change, step and candidate norms, candidate cost) sit in the long accumulators; k_publish folds them itself,
            // the copying hand-over needs them as doubles in replica 0 first
            // (with_cams: k_cam_update, one workgroup, has STORED the two camera norms in replica 0 -- they are not in the long accumulators)
            const unsigned det_kt = (with_cams ? (DET_K_TAIL & ~((1u << SC_STEP2_CAM) | (1u << SC_XN2_CAM))) : DET_K_TAIL) | (h->gram_fold ? (1u << SC_COST) : 0u);   // (gram_fold: the cost was not decoded behind the assembly)
            if (h->det && !poll) det_decode(zone_ptrs(iteration & 1), false, det_kt);
            if (ctx->collective) hipLaunchKernelGGL(k_scal_fold, dim3(1), dim3(SC_TOTAL * 64), 0, st, h->scal.p, (double*)nullptr, 0);
            int rc = allreduce(h, h->scal.p + SC_MODEL, 4, ncclSum); if (rc) return rc;   // MODEL, STEP2_PT, XN2_PT, CAND_COST
            if (poll) {
                spec_launched = false;
                if (spec_on && !with_cams && !(h->profile || O.verbose) && iteration < O.max_num_iterations) {
                    LmGate g; g.enabled = 1; g.last_successful = last_successful ? 1 : 0; g.radius = radius; g.x_norm = x_norm;
                    g.function_tolerance = O.function_tolerance; g.gradient_tolerance = O.gradient_tolerance; g.parameter_tolerance = O.parameter_tolerance;
                    g.min_relative_decrease = O.min_relative_decrease; g.max_radius = O.max_trust_region_radius; g.min_radius = O.min_trust_region_radius;
                    publish(h, &g, h->lmdev.p, det_kt); spec_launched = true;
                } else publish(h, nullptr, nullptr, det_kt);
                return SSFM_OK;
            }
            hipError_t e = hipMemcpyAsync(host_sp, h->scal.p, (SC_NSLOT * SC_TOTAL + PCG_TOTAL + 1) * sizeof(double), hipMemcpyDeviceToHost, st);   // scalars + solver flags
            if (e != hipSuccess) return fail(ctx, SSFM_ERR_HIP, hipGetErrorString(e));
            return SSFM_OK;
        };
        { int rc = enqueue_tail(!fused_cams); if (rc) return rc; }
        if (phases) SSFM_HIP_CHECK(ctx, hipEventRecord(h->phase_ev[4], st));
        // (otherwise) the next iteration's zone is cleared while the host wakes up and decides
        if (!clear_in_finalize) SSFM_HIP_CHECK(ctx, hipMemsetAsync(next_zone, 0, h->zone_len * sizeof(double), st));
        if (spec_launched && fuse_lin) { const int rc = launch_gram(cam_c, rot_c, pts_c, fc, zone_ptrs((iteration + 1) & 1), radius, (const double*)h->lmdev.p); if (rc) return rc; }
        else if (spec_launched)              // x = this iteration's candidate, scalars into the next zone (scal is its first block)
            LAUNCH(h, KID_POINT_LIN, k_point_lin<3>, (nP + PLB - 1) / PLB, PLB, 0, cam_c, rot_c, pts_c, fc, oxy, h->obs_cam.p, h->pt_start.p, nP, h->scale_pt.p,
                   h->scale_f.p, loss, la, radius, O.min_lm_diagonal, O.max_lm_diagonal, h->Vs.p, h->gp.p, next_zone, (const double*)h->lmdev.p, lacc);
        { int rc = wait_iteration(); if (rc) return rc; }
        // what the device decided for the speculative k_point_lin (it ran iff dev_go)
        const bool dev_go = spec_launched && h->host_pub[SC_TOTAL + PCG_TOTAL + 2] == 1.0;
        const double dev_radius = spec_launched ? h->host_pub[SC_TOTAL + PCG_TOTAL + 3] : 0.0;
        auto undo_speculation = [&]() -> int {                       // the host goes another way: the next zone must be clean again
            if (dev_go) SSFM_HIP_CHECK(ctx, hipMemsetAsync(next_zone, 0, h->zone_len * sizeof(double), st));
            if (dev_go && h->det) SSFM_HIP_CHECK(ctx, hipMemsetAsync(h->det_lacc.p, 0, (size_t)SC_NSLOT * SC_TOTAL * LA_STRIDE * sizeof(long long), st));   // the speculative point pass's sums
            return SSFM_OK;
        };
        SSFM_HIP_CHECK(ctx, hipGetLastError());                      // a launch that was refused (bad configuration) must not pass silently
        if (!poll) fold_host_scal();
        if (O.preconditioner == 0) {
            int fail_flag; std::memcpy(&fail_flag, &host_pcg1[PCG_TOTAL], sizeof(int));
            if (fail_flag) pcg_ok = false;
            else if (host_pcg1[PCG_DONE] == 0.0) {
                int rc = solve_reduced<DC>(h, host_pcg1, &pcg_iters, &pcg_ok, 1); if (rc) return rc;
                SSFM_HIP_CHECK(ctx, hipMemset2DAsync(h->scal.p + SC_MODEL, SC_TOTAL * sizeof(double), 0, 4 * sizeof(double), SC_NSLOT, st));   // MODEL..CAND_COST of every replica
                SSFM_HIP_CHECK(ctx, hipMemset2DAsync(h->scal.p + SC_STEP2_CAM, SC_TOTAL * sizeof(double), 0, 2 * sizeof(double), SC_NSLOT, st));   // and the camera norms
                rc = enqueue_tail(true); if (rc) return rc;
                rc = wait_iteration(); if (rc) return rc;
                if (!poll) fold_host_scal();
            }
        }
        h->pcg_prev_iters = pcg_iters; S->pcg_iterations_total += pcg_iters;
        if (phases) { float ms;
          if (hipEventElapsedTime(&ms, h->phase_ev[0], h->phase_ev[1]) == hipSuccess) ms_lin += ms;
          if (hipEventElapsedTime(&ms, h->phase_ev[1], h->phase_ev[2]) == hipSuccess) ms_schur += ms;
          if (hipEventElapsedTime(&ms, h->phase_ev[2], h->phase_ev[3]) == hipSuccess) ms_pcg += ms;
          if (hipEventElapsedTime(&ms, h->phase_ev[3], h->phase_ev[4]) == hipSuccess) ms_upd += ms; }
        if (h->profile) h->resolve_spans();
        S->num_linearizations++;
        // ================= host decisions (Ceres TrustRegionMinimizer, restated) =================
        x_cost = host_scal[SC_COST];
        double gmax; { unsigned long long bits; std::memcpy(&bits, &host_scal[SC_GMAX], 8); std::memcpy(&gmax, &bits, 8); }
        if (iteration == 1) { S->initial_cost = x_cost; minimum_cost = x_cost; }
        if (!std::isfinite(x_cost)) { S->termination = SSFM_FAILURE; break; }
        if (last_successful && gmax <= O.gradient_tolerance) { S->termination = SSFM_CONVERGENCE; iteration--; break; }
        const double model_cost_change = -host_scal[SC_MODEL];
        const bool valid = pcg_ok && std::isfinite(model_cost_change) && model_cost_change > 0.0;
        if (!valid) {
            if (++num_invalid >= O.max_num_consecutive_invalid_steps) { S->termination = SSFM_FAILURE; break; }
            radius /= decrease_factor; decrease_factor *= 2.0; last_successful = false; S->num_unsuccessful_steps++;
            { int rc = undo_speculation(); if (rc) return rc; }
            if (O.verbose) std::printf("[ssfm ba] iter %4d invalid step (pcg %d its, model %.3e), radius %.3e\n", iteration, pcg_iters, model_cost_change, radius);
            continue;
        }
        num_invalid = 0;
        double cand_cost = host_scal[SC_CAND_COST];
        if (!std::isfinite(cand_cost)) cand_cost = std::numeric_limits<double>::max();
        const double step_norm = std::sqrt(host_scal[SC_STEP2_PT] + host_scal[SC_STEP2_CAM]);
        if (step_norm <= O.parameter_tolerance * (x_norm + O.parameter_tolerance)) { S->termination = SSFM_CONVERGENCE; break; }
        const double cost_change = x_cost - cand_cost;
        if (std::fabs(cost_change) <= O.function_tolerance * x_cost) { S->termination = SSFM_CONVERGENCE; break; }
        const double rel = (cand_cost >= std::numeric_limits<double>::max()) ? std::numeric_limits<double>::lowest() : cost_change / model_cost_change;
        if (rel > O.min_relative_decrease) {
            std::swap(cam_x, cam_c); std::swap(pts_x, pts_c); std::swap(rot_x, rot_c); std::swap(fx, fc);
            x_norm = std::sqrt(host_scal[SC_XN2_PT] + host_scal[SC_XN2_CAM]);
            if (dev_go) { radius = dev_radius; lin_done = true; }     // the linearisation at the new x is already running with that radius
            else {
                { const double t3 = 2.0 * rel - 1.0; radius = radius / std::fmax(1.0 / 3.0, 1.0 - t3 * t3 * t3); }
                radius = std::fmin(O.max_trust_region_radius, radius);
            }
            decrease_factor = 2.0; last_successful = true; S->num_successful_steps++;
            x_cost = cand_cost; if (x_cost < minimum_cost) minimum_cost = x_cost;
        } else {
            radius /= decrease_factor; decrease_factor *= 2.0; last_successful = false; S->num_unsuccessful_steps++;
            { int rc = undo_speculation(); if (rc) return rc; }
        }
        if (O.verbose)
            std::printf("[ssfm ba] iter %4d cost %.12e change %.3e |g|inf %.3e |step| %.3e rho %.3e radius %.3e pcg %d %s\n", iteration,
                        x_cost, cost_change, gmax, step_norm, rel, radius, pcg_iters, last_successful ? "" : "(rejected)");
    }
    // make the x buffers of the handle hold the final state
    if (cam_x != h->cam_x.p || fx != h->focal3.p) {
        const int nc = cam_x != h->cam_x.p ? Nc * 6 : 0, np = cam_x != h->cam_x.p ? nP * 3 : 0;
        hipLaunchKernelGGL(k_copy_state, dim3((nc + np + 255) / 256 + 1), dim3(256), 0, st, h->cam_x.p, cam_x, nc, h->pts_x.p, pts_x, np,
                           fx != h->focal3.p ? h->focal3.p : (double*)nullptr, fx);
    }
    SSFM_HIP_CHECK(ctx, hipStreamSynchronize(st));
    probe_Y.free(); probe_S.free();
    S->iterations = iteration;
    S->final_cost = (minimum_cost == std::numeric_limits<double>::max()) ? x_cost : minimum_cost;
    S->t_kernel_linearize_ms = ms_lin; S->t_kernel_schur_ms = ms_schur; S->t_kernel_pcg_ms = ms_pcg; S->t_kernel_update_ms = ms_upd;
    return SSFM_OK;
}

}  // namespace ssfm

// =====================================================================================================
extern "C" void ssfm_ba_default_options(ssfm_ba_options* o) {
    o->max_num_iterations = 2000;                    // src/sfm.cpp:205
    o->max_num_consecutive_invalid_steps = 100;      // src/sfm.cpp:206
    o->function_tolerance = 1e-6; o->gradient_tolerance = 1e-10; o->parameter_tolerance = 1e-8;
    o->initial_trust_region_radius = 1e4; o->max_trust_region_radius = 1e16; o->min_trust_region_radius = 1e-32;
    o->min_lm_diagonal = 1e-6; o->max_lm_diagonal = 1e32; o->min_relative_decrease = 1e-3;
    o->loss_type = 1; o->loss_scale = 1.0;           // CauchyLoss(1.0), src/sfm.cpp:196
    o->jacobi_scaling = 1;
    o->pcg_max_iterations = 1000; o->pcg_tolerance = 1e-10; o->preconditioner = 0;
    o->verbose = 0;
}

// Test probe: the reduced-system factor + solve on a caller-supplied block band (band order, lower storage [N][b+1][dc*dc], block d
// of row i = (i, i-d)) and two right-hand-side columns Y [2][N*dc] (in/out).  Optional dumps of the substructured intermediates.
extern "C" int ssfm_band_solve_probe(ssfm_ctx* ctx, int32_t dc, int32_t N, int32_t b, int32_t ncomp, const int32_t* comp_ptr, const double* band,
                                     double* Y, int32_t* segs_seps_fail, double* Zdump, double* Ddump, double* Tdump) {
    if (!ctx || (dc != 3 && dc != 6) || N <= 0 || b < 1 || ncomp < 1 || !comp_ptr || !band || !Y) return fail(ctx, SSFM_ERR_INVALID, "ssfm_band_solve_probe: bad arguments");
    SSFM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    ssfm_ba_handle hh; ssfm_ba_handle* h = &hh;
    h->ctx = ctx; ssfm_ba_default_options(&h->opt);
    std::memset(h->k_launches, 0, sizeof(h->k_launches)); std::memset(h->k_ms, 0, sizeof(h->k_ms));
    BAFlat& F = h->F; F.Nc = N; F.band_rows = N; F.band_block = dc; F.band = b; F.DC = dc; F.comp_ptr.assign(comp_ptr, comp_ptr + ncomp + 1);
    for (int ir = 1; ir <= b; ir++) for (int kr = 1; kr <= ir; kr++) F.band_pairs.push_back(ir | (kr << 16));
    const size_t nb = (size_t)N * (b + 1) * dc * dc, n = (size_t)N * dc;
    int rc = SSFM_OK;
    auto run = [&]() -> int {
        SSFM_HIP_CHECK(ctx, h->band.alloc(nb)); SSFM_HIP_CHECK(ctx, h->Linv.alloc((size_t)N * dc * dc)); SSFM_HIP_CHECK(ctx, h->Yb.alloc(2 * n));
        SSFM_HIP_CHECK(ctx, h->pcg.alloc(PCG_TOTAL + 1));
        SSFM_HIP_CHECK(ctx, hipMemsetAsync(h->pcg.p, 0, (PCG_TOTAL + 1) * sizeof(double), st));
        SSFM_HIP_CHECK(ctx, hipMemsetAsync(h->Linv.p, 0, (size_t)N * dc * dc * sizeof(double), st));
        SSFM_HIP_CHECK(ctx, hipMemcpyAsync(h->band.p, band, nb * sizeof(double), hipMemcpyHostToDevice, st));
        SSFM_HIP_CHECK(ctx, hipMemcpyAsync(h->Yb.p, Y, 2 * n * sizeof(double), hipMemcpyHostToDevice, st));
        SSFM_HIP_CHECK(ctx, upload(h->band_pairs, F.band_pairs, st)); SSFM_HIP_CHECK(ctx, upload(h->comp_ptr, F.comp_ptr, st));
        { const int r = sub_upload(h, dc); if (r) return r; }
        if (h->sub.enabled && h->subZ.p) SSFM_HIP_CHECK(ctx, hipMemsetAsync(h->subZ.p, 0, h->subZ.n * sizeof(double), st));
        { const int r = (dc == 3) ? band_direct<3>(h, h->Yb.p) : band_direct<6>(h, h->Yb.p); if (r) return r; }
        SSFM_HIP_CHECK(ctx, hipMemcpyAsync(Y, h->Yb.p, 2 * n * sizeof(double), hipMemcpyDeviceToHost, st));
        int flag = 0;
        SSFM_HIP_CHECK(ctx, hipMemcpyAsync(&flag, h->pcg.p + PCG_TOTAL, sizeof(int), hipMemcpyDeviceToHost, st));
        if (h->sub.enabled) {
            if (Zdump) SSFM_HIP_CHECK(ctx, hipMemcpyAsync(Zdump, h->subZ.p, h->subZ.n * sizeof(double), hipMemcpyDeviceToHost, st));
            if (Ddump) SSFM_HIP_CHECK(ctx, hipMemcpyAsync(Ddump, h->subD.p, h->subD.n * sizeof(double), hipMemcpyDeviceToHost, st));
            if (Tdump) SSFM_HIP_CHECK(ctx, hipMemcpyAsync(Tdump, h->subT.p, h->subT.n * sizeof(double), hipMemcpyDeviceToHost, st));
        }
        SSFM_HIP_CHECK(ctx, hipStreamSynchronize(st));
        if (segs_seps_fail) { segs_seps_fail[0] = h->sub.enabled ? h->sub.nseg : ncomp; segs_seps_fail[1] = h->sub.nsep; segs_seps_fail[2] = flag; }
        return SSFM_OK;
    };
    rc = run();
    h->free_all();
    return rc;
}


// Test probe: the supernodal solver alone on a caller-supplied block-sparse symmetric positive definite matrix (block-CSR, every coupled pair of cameras stored
// once in either orientation + the diagonal blocks, row-major dc x dc blocks) and two right-hand-side columns; Y [2][Nc * dc] out (camera order).
// info: [0] 1 = the plan applies (else nothing ran) [1] workgroups [2] rows of T [3] factorisation failure flag
extern "C" int ssfm_snode_solve_probe(ssfm_ctx* ctx, int32_t dc, int32_t Nc, const int32_t* row_ptr, const int32_t* col_idx, const double* S_val, const double* rhs2,
                                      int32_t nr, double* Y, int32_t* info) {
    if (!ctx || (dc != 3 && dc != 6) || Nc <= 0 || !row_ptr || !col_idx || !S_val || !rhs2 || !Y || !info || (nr != 1 && nr != 2)) return fail(ctx, SSFM_ERR_INVALID, "ssfm_snode_solve_probe: bad arguments");
    SSFM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    ssfm_ba_handle hh; ssfm_ba_handle* h = &hh;
    h->ctx = ctx; ssfm_ba_default_options(&h->opt);
    std::memset(h->k_launches, 0, sizeof(h->k_launches)); std::memset(h->k_ms, 0, sizeof(h->k_ms));
    std::vector<int> rp(row_ptr, row_ptr + Nc + 1), ci(col_idx, col_idx + row_ptr[Nc]);
    snode_plan(Nc, dc, rp, ci, ctx->num_cus, h->sn);
    info[0] = h->sn.enabled ? 1 : 0; info[1] = h->sn.nhalf; info[2] = h->sn.qtm; info[3] = 0;
    if (!h->sn.enabled) return SSFM_OK;
    const size_t n = (size_t)Nc * dc, nnz = (size_t)row_ptr[Nc] * dc * dc;
    DevBuf<double> dS, dR; std::vector<int> ident(Nc); for (int c = 0; c < Nc; c++) ident[c] = c;
    auto run = [&]() -> int {
        SSFM_HIP_CHECK(ctx, dS.alloc(nnz)); SSFM_HIP_CHECK(ctx, dR.alloc(2 * n)); SSFM_HIP_CHECK(ctx, h->Yb.alloc(2 * n)); SSFM_HIP_CHECK(ctx, h->pcg.alloc(PCG_TOTAL + 1));
        SSFM_HIP_CHECK(ctx, hipMemsetAsync(h->pcg.p, 0, (PCG_TOTAL + 1) * sizeof(double), st)); SSFM_HIP_CHECK(ctx, hipMemsetAsync(h->Yb.p, 0, 2 * n * sizeof(double), st));
        SSFM_HIP_CHECK(ctx, hipMemcpyAsync(dS.p, S_val, nnz * sizeof(double), hipMemcpyHostToDevice, st)); SSFM_HIP_CHECK(ctx, hipMemcpyAsync(dR.p, rhs2, 2 * n * sizeof(double), hipMemcpyHostToDevice, st));
        SSFM_HIP_CHECK(ctx, upload(h->cam_pos, ident, st));
        { const int r = snode_upload(h); if (r) return r; }
        h->S_val = dS.p; h->rhs = dR.p; h->Sfc = dR.p + n; h->F.focal_free = nr == 2;
        // SSFM_SNODE_STAMPS=1 (timing study): phase stamps of every workgroup of the last launch on stderr, 100 MHz ticks
        const bool want_stamps = std::getenv("SSFM_SNODE_STAMPS") != nullptr; DevBuf<long long> dst;
        if (want_stamps) { SSFM_HIP_CHECK(ctx, dst.alloc((size_t)h->sn.nhalf * 64)); SSFM_HIP_CHECK(ctx, hipMemsetAsync(dst.p, 0, (size_t)h->sn.nhalf * 64 * sizeof(long long), st)); }
        const int reps = want_stamps ? 6 : 2;
        for (int rep = 0; rep < reps; rep++) { const int r = (dc == 3) ? snode_direct<3>(h, h->Yb.p, n, dst.p) : snode_direct<6>(h, h->Yb.p, n, dst.p); if (r) return r; }      // twice: the flags' sequence numbers
        if (want_stamps) {
            std::vector<long long> hs((size_t)h->sn.nhalf * 64); SSFM_HIP_CHECK(ctx, hipMemcpyAsync(hs.data(), dst.p, hs.size() * sizeof(long long), hipMemcpyDeviceToHost, st)); SSFM_HIP_CHECK(ctx, hipStreamSynchronize(st));
            long long t00 = hs[0]; for (int b = 0; b < h->sn.nhalf; b++) t00 = std::min(t00, hs[(size_t)b * 64]);
            for (int b = 0; b < h->sn.nhalf; b++) { const long long* q = &hs[(size_t)b * 64];
                std::fprintf(stderr, "[snode] wg %d (%d steps): start +%lld | prologue %lld | elimination %lld | exchange %lld | M,T %lld | back %lld (T, M %lld) | total %lld ticks\n", b, h->sn.half_rec[(size_t)b * SN_HREC],
                             q[0] - t00, q[1] - q[0], q[2] - q[1], q[3] - q[2], q[4] - q[3], q[5] - q[4], q[6] - q[4], q[5] - q[0]);
                if (b < 2) { const long long* u0 = q + 40; const long long* u1 = q + 48;
                    std::fprintf(stderr, "        stage M: panel (wave 0) %lld | barrier %lld | T rows + rhs (wave 1) %lld | outputs + barriers %lld | tiles + barrier %lld | the T stages %lld\n",
                                 u0[1] - u0[0], u0[2] - u0[1], u1[3] - u1[2], u0[4] - u1[3], u0[5] - u0[4], q[4] - u0[5]); }
                if (b < 2) for (int w = 0; w < 4; w++) { const long long* u = q + 8 + w * 8;
                    std::fprintf(stderr, "        slot 2, wave %d: own work %lld | wait B0 + outputs + B1 %lld | tiles %lld | wait B2 %lld\n", w, u[1] - u[0], u[2] - u[1], u[3] - u[2], u[4] - u[3]); } }
            dst.free();
        }
        SSFM_HIP_CHECK(ctx, hipMemcpyAsync(Y, h->Yb.p, 2 * n * sizeof(double), hipMemcpyDeviceToHost, st));
        int flag = 0;
        SSFM_HIP_CHECK(ctx, hipMemcpyAsync(&flag, h->pcg.p + PCG_TOTAL, sizeof(int), hipMemcpyDeviceToHost, st));
        SSFM_HIP_CHECK(ctx, hipStreamSynchronize(st));
        info[3] = flag;
        return SSFM_OK;
    };
    const int rc = run();
    h->S_val = nullptr; h->rhs = nullptr; h->Sfc = nullptr;
    dS.free(); dR.free(); h->free_all();
    return rc;
}


// Host-only (no GPU): the supernodal plan of a block structure, for the CPU tests and for anyone who wants to see what the solver will do.
// sizes: [8] = {applies, workgroups, rows reserved for T, cameras per supernode, node capacity, ints of half_rec, of step_rec, of node_cam}; the three tables and the block
// tables are copied out when the pointers are non-null and the capacities (in ints) suffice; tab_len in/out.
extern "C" int ssfm_snode_plan_probe(int32_t dc, int32_t Nc, const int32_t* row_ptr, const int32_t* col_idx, int32_t num_cus, int32_t* sizes, int32_t* half_rec, int32_t* step_rec,
                                     int32_t* node_cam, int32_t* tab, int32_t* tab_len) {
    if ((dc != 3 && dc != 6) || Nc <= 0 || !row_ptr || !col_idx || !sizes) return SSFM_ERR_INVALID;
    std::vector<int> rp(row_ptr, row_ptr + Nc + 1), ci(col_idx, col_idx + row_ptr[Nc]);
    SnodePlan P;
    snode_plan(Nc, dc, rp, ci, num_cus, P);
    sizes[0] = P.enabled ? 1 : 0; sizes[1] = P.nhalf; sizes[2] = P.qtm; sizes[3] = P.S; sizes[4] = P.CAPT; sizes[5] = (int)P.half_rec.size(); sizes[6] = (int)P.step_rec.size(); sizes[7] = (int)P.node_cam.size();
    if (!P.enabled) return SSFM_OK;
    if (half_rec) std::copy(P.half_rec.begin(), P.half_rec.end(), half_rec);
    if (step_rec) std::copy(P.step_rec.begin(), P.step_rec.end(), step_rec);
    if (node_cam) std::copy(P.node_cam.begin(), P.node_cam.end(), node_cam);
    if (tab_len) { if (tab && *tab_len >= (int)P.tab.size()) std::copy(P.tab.begin(), P.tab.end(), tab); *tab_len = (int)P.tab.size(); }
    return SSFM_OK;
}

// ssfm_ctx_destroy releases the recycled host arrays of the planner (ba_flatten.h: HostStash)
void ssfm_host_stash_clear() { host_stash_clear(); }

extern "C" int ssfm_ba_plan(const ssfm_ba_problem* p, int32_t nranks, int32_t rank, ssfm_ba_plan_info* info, int32_t* point_ids,
                            uint8_t* obs_used, int32_t* cam_pos) {
    if (!p || !info || nranks < 1 || rank < 0 || rank >= nranks) return fail(nullptr, SSFM_ERR_INVALID, "ssfm_ba_plan: bad arguments");
    BAFlat F; ba_flatten(*p, nranks, rank, F, false);     // the pair lists are not part of what the plan reports
    struct GiveBack { BAFlat& F; ~GiveBack() { stash_give(F); } } give_back{F};      // ba_flatten took the recycled host arrays: hand them on
    std::memset(info, 0, sizeof(*info));
    if (obs_used) std::memset(obs_used, 0, (size_t)p->num_observations);
    if (F.nothing_to_do) return SSFM_OK;
    info->camera_dof = F.DC; info->num_points_used = F.nP; info->num_points_used_global = F.nP_global;
    info->reduced_blocks = F.row_ptr[F.Nc]; info->band_half_width = F.band; info->max_row_blocks = F.max_row_blocks;
    info->num_observations_used = F.M; info->num_observations_used_global = F.M_global;
    { BandSub B; sub_build(F.comp_ptr, F.comp_twist, F.band, F.band_block, B, &F.rings);
      info->band_segments = B.enabled ? B.nseg : (int)F.comp_ptr.size() - 1; info->band_separators = B.nsep + B.ntwist; }
    info->num_points_grouped = F.gram_points; info->num_observations_grouped = F.gram_obs; info->group_tasks = (int32_t)(F.gr_rec.size() / GRAM_REC);
    if (point_ids) for (int i = 0; i < F.nP; i++) point_ids[i] = F.pt_ids[i];
    if (obs_used) for (int64_t j = 0; j < F.M; j++) obs_used[F.obs_orig[j]] = 1;
    if (cam_pos) for (int c = 0; c < F.Nc; c++) cam_pos[c] = F.cam_pos[c];
    return SSFM_OK;
}

static int ba_create_impl(ssfm_ctx* ctx, const ssfm_ba_problem* p, const ssfm_ba_options* o, ssfm_ba_handle** out);
// A failed create leaves nothing behind: the half-built handle is destroyed here and *out is null.
extern "C" int ssfm_ba_create(ssfm_ctx* ctx, const ssfm_ba_problem* p, const ssfm_ba_options* o, ssfm_ba_handle** out) {
    if (!ctx || !p || !out) return fail(ctx, SSFM_ERR_INVALID, "ssfm_ba_create: null argument");
    *out = nullptr;
    const int rc = ba_create_impl(ctx, p, o, out);
    if (rc != SSFM_OK && *out) { const std::string msg = ctx->err; ssfm_ba_destroy(*out); *out = nullptr; ctx->err = msg; }
    return rc;
}
static int ba_create_impl(ssfm_ctx* ctx, const ssfm_ba_problem* p, const ssfm_ba_options* o, ssfm_ba_handle** out) {
    SSFM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    ssfm_ba_handle* h = new ssfm_ba_handle();
    h->ctx = ctx; h->device = ctx->device;
    if (o) h->opt = *o; else ssfm_ba_default_options(&h->opt);
    std::memset(h->k_launches, 0, sizeof(h->k_launches)); std::memset(h->k_ms, 0, sizeof(h->k_ms));
    h->det = std::getenv("SSFM_DETERMINISTIC") && std::atoi(std::getenv("SSFM_DETERMINISTIC")) != 0;
    if (h->det && ctx->collective) { delete h; return fail(ctx, SSFM_ERR_INVALID, "SSFM_DETERMINISTIC=1 is implemented for the single-GPU solve only (det_acc.h)"); }
    // default: the pair lists are counted and filled on the GPU (through atomic cursors: the order inside a slot differs from handle to handle, so the deterministic mode
    // takes the host's lists, whose order is a function of the problem alone)
    const bool host_pairs = std::getenv("SSFM_HOST_PAIRS") != nullptr || h->det;
    g_alloc_timing = std::getenv("SSFM_PLAN_TIMING") != nullptr; g_alloc_ns = 0; g_alloc_n = 0;
    const double t_create0 = wall_s();
    // The per-observation arrays are final long before the plan is (the structure of S, the ordering, the task tables follow): a second host thread
    // uploads them on the context's stream while this one goes on planning -- the main thread does not touch the stream until it has joined the upload.
    hipStream_t st = ctx->stream;
    std::thread up_thread; int up_rc = SSFM_OK; double up_s = 0.0;
    struct JoinGuard { std::thread& t; ~JoinGuard() { if (t.joinable()) t.join(); } } up_guard{up_thread};      // ba_flatten may throw (std::bad_alloc) after the thread started
    static const bool overlap_upload = !(std::getenv("SSFM_PLAN_OVERLAP") && std::atoi(std::getenv("SSFM_PLAN_OVERLAP")) == 0);
    auto upload_obs = [&](BAFlat& Fm) -> int {
        const double tu = wall_s();
        SSFM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
        SSFM_HIP_CHECK(ctx, upload(h->obs_xy, Fm.obs_xy, st)); SSFM_HIP_CHECK(ctx, upload(h->obs_cam, Fm.obs_cam, st)); SSFM_HIP_CHECK(ctx, upload(h->obs_pt, Fm.obs_pt, st));
        SSFM_HIP_CHECK(ctx, upload(h->pt_start, Fm.pt_start, st)); SSFM_HIP_CHECK(ctx, upload(h->pts_x, Fm.pts0, st)); SSFM_HIP_CHECK(ctx, upload(h->pts_init, Fm.pts0, st));
        SSFM_HIP_CHECK(ctx, upload(h->mask_pt, Fm.mask_pt, st));
        up_s = wall_s() - tu;
        return SSFM_OK;
    };
    const std::function<void(BAFlat&)> after_emit = [&](BAFlat& Fm) { if (overlap_upload) up_thread = std::thread([&]() { up_rc = upload_obs(Fm); }); };
    try { const double tf = wall_s(); ba_flatten(*p, ctx->nranks, ctx->rank, h->F, host_pairs, ctx->num_cus, &after_emit); h->t_flatten_s = wall_s() - tf; }
    catch (const std::exception& e) { if (up_thread.joinable()) up_thread.join(); *out = h; /* the caller tears it down */ return fail(ctx, SSFM_ERR_INVALID, (std::string("ssfm_ba_create: planning failed: ") + e.what()).c_str()); }
    *out = h;
    if (up_thread.joinable()) up_thread.join();
    const BAFlat& F = h->F;
    if (F.nothing_to_do) return SSFM_OK;
    if (up_rc != SSFM_OK) return up_rc;
    if (!overlap_upload) { const int rc = upload_obs(h->F); if (rc) return rc; }
    if (g_alloc_timing) std::fprintf(stderr, "[create] observation upload %.2f ms (%s)\n", 1e3 * up_s, overlap_upload ? "on a second thread, under the rest of the plan" : "after the plan");
    const int Nc = F.Nc, nP = F.nP, DC = F.DC;
    std::vector<double> cams(p->cameras, p->cameras + (size_t)Nc * 6);
    h->focal_host = *p->focal;
    std::vector<double> f3 = {*p->focal, *p->focal, *p->focal};
    std::vector<double> maskf = {F.focal_free ? 1.0 : 0.0};
#define UP(buf, vec) SSFM_HIP_CHECK(ctx, upload(h->buf, vec, st))
    UP(cam_x, cams); UP(cam_init, cams); UP(focal3, f3);
    UP(mask_cam, F.mask_cam); UP(mask_f, maskf);
    UP(cam_start, F.cam_start);
    if (host_pairs) { UP(cam_obs, F.cam_obs); UP(cam_obs_pt, F.cam_obs_pt); }
    else if (F.M > 0) {                                               // camera-major lists on the device
        DevBuf<unsigned int> cur;
        SSFM_HIP_CHECK(ctx, h->cam_obs.alloc((size_t)F.M)); SSFM_HIP_CHECK(ctx, h->cam_obs_pt.alloc((size_t)F.M)); SSFM_HIP_CHECK(ctx, cur.alloc((size_t)Nc));
        hipError_t e1 = hipMemcpyAsync(cur.p, h->cam_start.p, (size_t)Nc * sizeof(int), hipMemcpyDeviceToDevice, st);
        if (Nc <= CAM_LISTS_MAX_CAMS)
            hipLaunchKernelGGL(k_cam_lists_agg, dim3((unsigned)((F.M + CAM_LISTS_CHUNK - 1) / CAM_LISTS_CHUNK)), dim3(256), 0, st, (int)F.M, Nc, h->obs_cam.p, h->obs_pt.p, cur.p,
                               h->cam_obs.p, h->cam_obs_pt.p);
        else
            hipLaunchKernelGGL(k_cam_lists, dim3((unsigned)((F.M + 255) / 256)), dim3(256), 0, st, (int)F.M, h->obs_cam.p, h->obs_pt.p, cur.p, h->cam_obs.p, h->cam_obs_pt.p);
        hipLaunchKernelGGL(k_cam_lists_sort, dim3(Nc), dim3(256), 0, st, h->cam_start.p, h->obs_pt.p, h->cam_obs.p, h->cam_obs_pt.p);
        hipError_t e2 = hipStreamSynchronize(st);
        cur.free();
        SSFM_HIP_CHECK(ctx, e1); SSFM_HIP_CHECK(ctx, e2);
    }
    UP(cs_task_cam, F.cs_task_cam); UP(cs_task_q0, F.cs_task_q0); UP(cs_task_q1, F.cs_task_q1); UP(row_ptr, F.row_ptr); UP(col_idx, F.col_idx); UP(diag_slot, F.diag_slot);
    if (!F.gr_rec.empty()) UP(gr_rec, F.gr_rec);
    // atomics-free Gram emission + fold (round 6): single rank, direct solver, every point grouped; on with SSFM_DETERMINISTIC=1 (h->det), see ba_flatten.h
    h->gram_fold = h->det && !ctx->collective && h->opt.preconditioner == 0 && !F.gpart_off.empty() && F.max_row_blocks <= 27 && F.max_row_blocks + 2 <= GRAM_FOLD_PTRS && !(std::getenv("SSFM_GRAM_FOLD") && std::atoi(std::getenv("SSFM_GRAM_FOLD")) == 0);
    if (h->gram_fold) {
        UP(gpart_off, F.gpart_off); UP(fold_slot_src, F.fold_slot_src);
        SSFM_HIP_CHECK(ctx, h->gram_part.alloc((size_t)F.gpart_off.back() + 64));      // (+ slack: the fold reads a fixed number of doubles per source)
        SSFM_HIP_CHECK(ctx, hipMemsetAsync(h->gram_part.p, 0, ((size_t)F.gpart_off.back() + 64) * sizeof(double), st));      // entries a task never writes (S_fc with a fixed focal) stay zero
    }
    UP(pt_grouped, F.pt_grouped);
#undef UP
#define AL(buf, count) SSFM_HIP_CHECK(ctx, h->buf.alloc(count))
    AL(cam_c, (size_t)Nc * 6); AL(pts_c, (size_t)nP * 3); AL(rot_x, (size_t)Nc * 27); AL(rot_c, (size_t)Nc * 27);
    AL(scale_cam, (size_t)Nc * 6); AL(scale_pt, (size_t)nP * 3); AL(scale_f, 1);
    AL(diag_cam, (size_t)Nc * 6); AL(diag_pt, (size_t)nP * 3); AL(diag_f, 1);
    AL(lmdev, 2);
    AL(Vs, (size_t)nP * 12); AL(gp, (size_t)nP * 3);
    const size_t nnzb = (size_t)F.row_ptr[Nc], n = (size_t)Nc * DC;
    const size_t n_red = nnzb * DC * DC + (n + 1) + 3 * n + SC_NSUM + (size_t)ctx->nranks;   // ... | scalar sums | gradient-max slot per rank
    h->n_red = (int)n_red;
    // [scal | pcg flags + factorisation fail word | redbuf] share one allocation: one memset per LM iteration zeroes them all,
    // and one copy brings both scalar groups back
    // (two of them, each a multiple of 512 bytes so that the memset is a single fill: ba_handle.h set_zone)
    h->zone_len = (SC_NSLOT * SC_TOTAL + PCG_TOTAL + 1 + n_red + 63) / 64 * 64; h->zone_nnz = nnzb * DC * DC; h->zone_n = n;
    AL(zone, 2 * h->zone_len);
    h->scal.n = SC_NSLOT * SC_TOTAL; h->pcg.n = PCG_TOTAL + 1; h->redbuf.n = n_red; h->zone_views = true;
    h->set_zone(0);
    if (h->det) {
        h->det_nacc = h->zone_nnz + 4 * n + 1;                      // [S | rhs (+ focal row) | diag U | S_fc | Jc^T r]: contiguous in the zone from S_val on
        AL(det_limb, 2 * h->det_nacc + 2); AL(det_lacc, (size_t)SC_NSLOT * SC_TOTAL * LA_STRIDE); AL(det_dfpart, (size_t)std::max(1, (nP + 255) / 256));
        SSFM_HIP_CHECK(ctx, hipMemsetAsync(h->det_limb.p, 0, (2 * h->det_nacc + 2) * sizeof(long long), st));
        SSFM_HIP_CHECK(ctx, hipMemsetAsync(h->det_lacc.p, 0, (size_t)SC_NSLOT * SC_TOTAL * LA_STRIDE * sizeof(long long), st));
    }
    AL(Minv, (size_t)Nc * DC * DC); AL(Sff, 1);
    AL(px, n + 1); AL(pr, n + 1); AL(pz, n + 1); AL(pp, n + 1); AL(pq, n + 1); AL(pqpart, (size_t)Nc);
    // band: F.band_rows block rows of F.band_block x F.band_block blocks (>= cameras: twisted components carry a second copy of their
    // separator; 3-dof cameras are merged in pairs); right-hand sides in camera rows
    const size_t Nb = (size_t)F.band_rows, DCB = (size_t)F.band_block, nbr = (size_t)F.y_rows(DC) * DC;
    AL(band, Nb * (F.band + 1) * DCB * DCB); AL(Linv, Nb * DCB * DCB); AL(Yb, 2 * nbr); AL(Yr, 2 * nbr); AL(band_fail, 1);
#undef AL
    SSFM_HIP_CHECK(ctx, upload(h->cam_pos, F.band_row, st)); SSFM_HIP_CHECK(ctx, upload(h->cam_pos2, F.band_row2, st));   // the device only needs band rows
    SSFM_HIP_CHECK(ctx, upload(h->pair_dummy, F.pair_dummy, st));
    SSFM_HIP_CHECK(ctx, upload(h->band_pairs, F.band_pairs, st));
    SSFM_HIP_CHECK(ctx, upload(h->comp_ptr, F.comp_ptr, st));
    { const int rc = sub_upload(h, DC); if (rc) return rc; }     // long components: segments + separators (band_sub.h)
    // rings / chains of cameras with a reach of <= 30 / DC cameras: the supernodal solver (snode.h) takes the direct solve; the band stays planned for the refinement path
    if (h->opt.preconditioner == 0 && F.sym_lower && snode_enabled()) {
        const double ts = wall_s();
        snode_plan(Nc, DC, F.row_ptr, F.col_idx, ctx->num_cus, h->sn);
        if (g_alloc_timing) std::fprintf(stderr, "[create] supernodal plan %.2f ms: %s, %d workgroups, T rows %d\n", 1e3 * (wall_s() - ts), h->sn.enabled ? "on" : "not applicable", h->sn.nhalf, h->sn.qtm);
        const int rc = snode_upload(h); if (rc) return rc;
    }
    SSFM_HIP_CHECK(ctx, upload(h->trans_ptr, F.trans_ptr, st)); SSFM_HIP_CHECK(ctx, upload(h->trans_blk, F.trans_blk, st));
    SSFM_HIP_CHECK(ctx, upload(h->trans_row, F.trans_row, st));
    {   // band row of every stored block's column camera (and of every transposed block's): k_arrow_update reads it instead of col_idx -> pos, one dependent gather less
        std::vector<int> cp(F.col_idx.size()), tp(F.trans_row.size());
        for (size_t e = 0; e < cp.size(); e++) cp[e] = F.band_row[F.col_idx[e]];
        for (size_t e = 0; e < tp.size(); e++) tp[e] = F.band_row[F.trans_row[e]];
        if (cp.empty()) cp.push_back(0);
        if (tp.empty()) tp.push_back(0);
        SSFM_HIP_CHECK(ctx, upload(h->col_pos, cp, st)); SSFM_HIP_CHECK(ctx, upload(h->trans_pos, tp, st));
    }
    if (host_pairs) {
        SSFM_HIP_CHECK(ctx, upload(h->pair_j, F.pair_j, st)); SSFM_HIP_CHECK(ctx, upload(h->pair_j2, F.pair_j2, st)); SSFM_HIP_CHECK(ctx, upload(h->pair_p, F.pair_p, st));
    } else if (F.M > 0) {
        // pair lists on the device: count per slot, lay the batches out on the host (tiny), fill through atomic cursors
        BAFlat& Fm = h->F;
        DevBuf<int> elim; DevBuf<unsigned int> ctr;
        const size_t nnzb_s = Fm.col_idx.size();
        int rc2 = SSFM_OK;
        auto build = [&]() -> int {
            SSFM_HIP_CHECK(ctx, upload(elim, Fm.cam_pos, st)); SSFM_HIP_CHECK(ctx, ctr.alloc(nnzb_s));
            SSFM_HIP_CHECK(ctx, hipMemsetAsync(ctr.p, 0, nnzb_s * sizeof(unsigned int), st));
            const int gq = (int)((Fm.M + 255) / 256);
            hipLaunchKernelGGL((k_pair_lists<false>), dim3(gq), dim3(256), 0, st, (int)Fm.M, h->cam_obs.p, h->cam_obs_pt.p, h->obs_cam.p, h->pt_start.p, elim.p, h->row_ptr.p,
                               h->col_idx.p, ctr.p, (int*)nullptr, (int*)nullptr, (int*)nullptr, (const unsigned char*)h->pt_grouped.p);
            std::vector<int> slot_cnt(nnzb_s);
            SSFM_HIP_CHECK(ctx, hipMemcpyAsync(slot_cnt.data(), ctr.p, nnzb_s * sizeof(int), hipMemcpyDeviceToHost, st));
            SSFM_HIP_CHECK(ctx, hipStreamSynchronize(st));
            std::vector<int64_t> slot_off; const int64_t nbat = pair_layout(Fm, slot_cnt, slot_off, ctx->num_cus);
            if (nbat * 64 >= (int64_t)1 << 31) return fail(ctx, SSFM_ERR_INVALID, "ssfm_ba_create: more than 2^31 Schur pairs on one rank");
            std::vector<unsigned int> start(nnzb_s); for (size_t e = 0; e < nnzb_s; e++) start[e] = (unsigned int)slot_off[e];
            SSFM_HIP_CHECK(ctx, hipMemcpyAsync(ctr.p, start.data(), nnzb_s * sizeof(unsigned int), hipMemcpyHostToDevice, st));
            const size_t npair = (size_t)nbat * 64;
            SSFM_HIP_CHECK(ctx, h->pair_j.alloc(npair)); SSFM_HIP_CHECK(ctx, h->pair_j2.alloc(npair)); SSFM_HIP_CHECK(ctx, h->pair_p.alloc(npair));
            SSFM_HIP_CHECK(ctx, hipMemsetAsync(h->pair_j.p, 0xFF, npair * sizeof(int), st)); SSFM_HIP_CHECK(ctx, hipMemsetAsync(h->pair_j2.p, 0xFF, npair * sizeof(int), st));
            SSFM_HIP_CHECK(ctx, hipMemsetAsync(h->pair_p.p, 0xFF, npair * sizeof(int), st));
            hipLaunchKernelGGL((k_pair_lists<true>), dim3(gq), dim3(256), 0, st, (int)Fm.M, h->cam_obs.p, h->cam_obs_pt.p, h->obs_cam.p, h->pt_start.p, elim.p, h->row_ptr.p,
                               h->col_idx.p, ctr.p, h->pair_j.p, h->pair_j2.p, h->pair_p.p, (const unsigned char*)h->pt_grouped.p);
            SSFM_HIP_CHECK(ctx, hipStreamSynchronize(st));           // `start` and the temporaries go out of scope
            return SSFM_OK;
        };
        rc2 = build();
        elim.free(); ctr.free();
        if (rc2) return rc2;
    }
    SSFM_HIP_CHECK(ctx, upload(h->batch_slot, F.batch_slot, st));
    SSFM_HIP_CHECK(ctx, upload(h->cam_batch_ptr, F.cam_batch_ptr, st)); SSFM_HIP_CHECK(ctx, upload(h->chunk_cam, F.chunk_cam, st));
    SSFM_HIP_CHECK(ctx, upload(h->chunk_b0, F.chunk_b0, st)); SSFM_HIP_CHECK(ctx, upload(h->chunk_b1, F.chunk_b1, st));
    SSFM_HIP_CHECK(ctx, hipMemsetAsync(h->zone.p, 0, h->zone.n * sizeof(double), st));
    SSFM_HIP_CHECK(ctx, hipStreamSynchronize(st));
    if (g_alloc_timing) std::fprintf(stderr, "[create] total %.2f ms: host plan %.2f, %d hipMalloc %.2f ms, uploads + device lists %.2f ms\n", 1e3 * (wall_s() - t_create0),
                                     1e3 * h->t_flatten_s, g_alloc_n.load(), 1e-6 * g_alloc_ns.load(), 1e3 * (wall_s() - t_create0 - h->t_flatten_s - 1e-9 * g_alloc_ns.load()));
    return SSFM_OK;
}

extern "C" int ssfm_ba_reset(ssfm_ba_handle* h) {
    if (!h) return SSFM_ERR_INVALID;
    if (h->F.nothing_to_do) return SSFM_OK;
    ssfm_ctx* ctx = h->ctx; hipStream_t st = ctx->stream;
    const int nc = (int)h->cam_init.n, np = h->F.nP > 0 ? (int)h->pts_init.n : 0;
    hipLaunchKernelGGL(k_copy_state, dim3((nc + np + 255) / 256 + 1), dim3(256), 0, st, h->cam_x.p, h->cam_init.p, nc, h->pts_x.p, h->pts_init.p, np,
                       h->focal3.p, h->focal3.p + 2);
    SSFM_HIP_CHECK(ctx, hipGetLastError());
    h->scale_ready = false;
    return SSFM_OK;
}

extern "C" int ssfm_ba_run(ssfm_ba_handle* h, ssfm_ba_summary* s) {
    if (!h || !s) return SSFM_ERR_INVALID;
    std::memset(s, 0, sizeof(*s));
    const BAFlat& F = h->F;
    const int32_t nblk = F.nothing_to_do ? 0 : F.row_ptr[F.Nc];
    s->num_residual_blocks = F.M; s->num_residual_blocks_global = F.M_global; s->num_points_used = F.nP; s->camera_dof = F.DC;
    s->reduced_blocks = F.nothing_to_do ? 0 : F.row_ptr[F.Nc]; s->band_half_width = F.band;
    const int32_t nsegs = h->sub.enabled ? h->sub.nseg : (int32_t)F.comp_ptr.size() - 1, nseps = h->sub.nsep + h->sub.ntwist;
    s->band_segments = F.nothing_to_do ? 0 : nsegs; s->band_separators = nseps;
    if (F.nothing_to_do) { s->termination = SSFM_NOTHING_TO_DO; return SSFM_OK; }
    SSFM_HIP_CHECK(h->ctx, hipSetDevice(h->ctx->device));
    std::memset(h->k_launches, 0, sizeof(h->k_launches)); std::memset(h->k_ms, 0, sizeof(h->k_ms));
    const double t0 = wall_s();
    int rc = (F.DC == 3) ? lm_loop<3>(h, s) : lm_loop<6>(h, s);
    s->t_solve_s = wall_s() - t0; s->reduced_blocks = nblk; s->band_half_width = F.band; s->band_segments = nsegs; s->band_separators = nseps;
    return rc;
}

extern "C" int ssfm_ba_download(ssfm_ba_handle* h, ssfm_ba_problem* p) {
    if (!h || !p) return SSFM_ERR_INVALID;
    const BAFlat& F = h->F;
    if (F.nothing_to_do) return SSFM_OK;
    ssfm_ctx* ctx = h->ctx; hipStream_t st = ctx->stream;
    // through the context's pinned staging buffer -- a fresh 2.4 MB std::vector is page faults + a staged pageable copy
    const size_t n_cam = (size_t)F.Nc * 6, n_pt = (size_t)F.nP * 3;
    std::vector<double> tmp;
    double* stage = nullptr;
    {                                                               // the context's pinned staging buffer (grow-only, outlives the handles; shared with upload_state)
        if (ctx->dl_stage_n < n_cam + n_pt + 1) {
            if (ctx->dl_stage) (void)hipHostFree(ctx->dl_stage);
            ctx->dl_stage = nullptr; ctx->dl_stage_n = 0;
            const size_t want = (n_cam + n_pt + 1) + (n_cam + n_pt + 1) / 4;
            if (hipHostMalloc((void**)&ctx->dl_stage, want * sizeof(double), hipHostMallocDefault) == hipSuccess) ctx->dl_stage_n = want; else { ctx->dl_stage = nullptr; (void)hipGetLastError(); }
        }
        stage = ctx->dl_stage;
    }
    if (!stage) { tmp.resize(n_cam + n_pt + 1); stage = tmp.data(); }
    double* cams = stage; double* pts = stage + n_cam; double& f = stage[n_cam + n_pt];
    SSFM_HIP_CHECK(ctx, hipMemcpyAsync(cams, h->cam_x.p, n_cam * sizeof(double), hipMemcpyDeviceToHost, st));
    if (F.nP > 0) SSFM_HIP_CHECK(ctx, hipMemcpyAsync(pts, h->pts_x.p, n_pt * sizeof(double), hipMemcpyDeviceToHost, st));
    SSFM_HIP_CHECK(ctx, hipMemcpyAsync(&f, h->focal3.p, sizeof(double), hipMemcpyDeviceToHost, st));
    SSFM_HIP_CHECK(ctx, hipStreamSynchronize(st));
    // only parameters that were free are written back (constant blocks are bit-identical anyway)
    for (size_t i = 0; i < n_cam; i++) if (F.mask_cam[i] > 0.0) p->cameras[i] = cams[i];
    for (int q = 0; q < F.nP; q++) if (F.mask_pt[(size_t)q * 3] > 0.0) for (int d = 0; d < 3; d++) p->points[(size_t)F.pt_ids[q] * 3 + d] = pts[(size_t)q * 3 + d];
    if (F.focal_free) *p->focal = f;
    if (ctx->collective && ctx->nranks > 1) {
        // every rank leaves with every point: [x y z owned] per global point id, summed over ranks (each point has one owner)
        const size_t Np = (size_t)p->num_points;
        std::vector<double> all(Np * 4, 0.0);
        for (int q = 0; q < F.nP; q++) if (F.mask_pt[(size_t)q * 3] > 0.0) {
            const size_t g = (size_t)F.pt_ids[q];
            for (int d = 0; d < 3; d++) all[g * 4 + d] = pts[(size_t)q * 3 + d];
            all[g * 4 + 3] = 1.0;
        }
        DevBuf<double> dall; SSFM_HIP_CHECK(ctx, upload(dall, all, st));
        int rc = allreduce(h, dall.p, all.size(), ncclSum);
        if (rc == SSFM_OK) {
            SSFM_HIP_CHECK(ctx, hipMemcpyAsync(all.data(), dall.p, all.size() * sizeof(double), hipMemcpyDeviceToHost, st));
            SSFM_HIP_CHECK(ctx, hipStreamSynchronize(st));
            for (size_t g = 0; g < Np; g++) if (all[g * 4 + 3] > 0.5) for (int d = 0; d < 3; d++) p->points[g * 3 + d] = all[g * 4 + d];
        }
        dall.free();
        return rc;
    }
    return SSFM_OK;
}

extern "C" void ssfm_ba_destroy(ssfm_ba_handle* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    // nothing of this handle may still be in flight when its buffers go back to the pool (asynchronous copies of ssfm_ba_reset, a
    // speculative launch); the device-wide wait does not touch the context, which the caller may already have destroyed
    (void)hipDeviceSynchronize();
    stash_give(h->F);                  // the per-observation host arrays go to the next plan, already mapped (ba_flatten.h: HostStash)
    h->free_all();
    delete h;
}

extern "C" int ssfm_ba_set_profiling(ssfm_ba_handle* h, int32_t on) { if (!h) return SSFM_ERR_INVALID; h->profile = on != 0; return SSFM_OK; }

extern "C" int ssfm_ba_kernel_times(ssfm_ba_handle* h, int32_t max_entries, char names[][32], int64_t* launches, double* total_ms) {
    if (!h) return 0;
    int k = 0;
    for (int i = 0; i < KID_COUNT && k < max_entries; i++) {
        if (h->k_launches[i] == 0) continue;
        std::strncpy(names[k], kKernelNames[i], 31); names[k][31] = 0; launches[k] = h->k_launches[i]; total_ms[k] = h->k_ms[i]; k++;
    }
    return k;
}

extern "C" int ssfm_ba_evaluate(ssfm_ba_handle* h, double* cost, double* residuals, double* jacobians) {
    if (!h) return SSFM_ERR_INVALID;
    const BAFlat& F = h->F; ssfm_ctx* ctx = h->ctx; hipStream_t st = ctx->stream;
    if (F.nothing_to_do) { if (cost) *cost = 0; return SSFM_OK; }
    const int Nc = F.Nc, nP = F.nP; const int64_t M = F.M;
    const double2* oxy = reinterpret_cast<const double2*>(h->obs_xy.p);
    DevBuf<double> dres, djac, dcost;
    SSFM_HIP_CHECK(ctx, dres.alloc((size_t)M * 2)); SSFM_HIP_CHECK(ctx, djac.alloc((size_t)M * 20)); SSFM_HIP_CHECK(ctx, dcost.alloc(1));
    SSFM_HIP_CHECK(ctx, hipMemsetAsync(dcost.p, 0, sizeof(double), st));
    hipLaunchKernelGGL(k_cam_rot, dim3((Nc + 63) / 64), dim3(64), 0, st, h->cam_x.p, h->rot_x.p, Nc);
    if (M > 0) {
        hipLaunchKernelGGL(k_eval_dump, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, st, h->cam_x.p, h->rot_x.p, h->pts_x.p, h->focal3.p, oxy,
                           h->obs_cam.p, h->obs_pt.p, (int)M, h->opt.loss_type, h->opt.loss_scale, dres.p, djac.p);
        hipLaunchKernelGGL(k_point_cost, dim3((nP + 255) / 256), dim3(256), 0, st, h->cam_x.p, h->rot_x.p, h->pts_x.p, h->focal3.p, oxy, h->obs_cam.p,
                           h->pt_start.p, nP, h->opt.loss_type, h->opt.loss_scale, dcost.p, 0);
    }
    std::vector<double> res((size_t)M * 2), jac((size_t)M * 20); double c = 0;
    if (M > 0) {
        SSFM_HIP_CHECK(ctx, hipMemcpyAsync(res.data(), dres.p, res.size() * sizeof(double), hipMemcpyDeviceToHost, st));
        SSFM_HIP_CHECK(ctx, hipMemcpyAsync(jac.data(), djac.p, jac.size() * sizeof(double), hipMemcpyDeviceToHost, st));
    }
    SSFM_HIP_CHECK(ctx, hipMemcpyAsync(&c, dcost.p, sizeof(double), hipMemcpyDeviceToHost, st));
    SSFM_HIP_CHECK(ctx, hipStreamSynchronize(st));
    dres.free(); djac.free(); dcost.free();
    if (ctx->collective) { /* cost of this rank's shard only; callers sum */ }
    if (cost) *cost = c;
    for (int64_t j = 0; j < M; j++) {
        const int64_t o = F.obs_orig[j];
        if (residuals) { residuals[2 * o] = res[2 * j]; residuals[2 * o + 1] = res[2 * j + 1]; }
        if (jacobians) std::memcpy(&jacobians[20 * o], &jac[20 * j], 20 * sizeof(double));
    }
    return SSFM_OK;
}

// ---- plan cache of ssfm_ba_solve ---------------------------------------------------------------------------------
// The drivers call Optimize() several times on the same structure (run_spherical_sfm.cpp:93-112: BA, Retriangulate, BA, ...).
// Planning + allocation + index uploads cost more than the solve itself at config 2, so ssfm_ba_solve keeps the handle of the
// last structure it saw: a 128-bit hash over everything the plan depends on (observation ids in order, the fixed masks, which
// points are zero, the sizes, the rank layout) decides between "upload the new parameters and run" and a fresh plan.
namespace {
struct PlanCache { ssfm_ba_handle* h = nullptr; int Nc = 0, Np = 0; int64_t M = 0; int nranks = 0, rank = 0, focal_fixed = 0; uint64_t h1 = 0, h2 = 0; };
void plan_cache_free(void* c) { PlanCache* pc = static_cast<PlanCache*>(c); if (pc->h) ssfm_ba_destroy(pc->h); delete pc; }
inline void mix(uint64_t& a, uint64_t& b, uint64_t w) {
    a = (a ^ w) * 0x9E3779B97F4A7C15ull; a ^= a >> 32;
    b = (b + w + 0x632BE59BD9B4E019ull) * 0xD6E8FEB86659FD93ull; b ^= b >> 29;
}
void hash_chunk(const unsigned char* p, size_t n, uint64_t& a, uint64_t& b) {
    // four interleaved lanes: the multiply chain of mix() is serial, four independent chains keep the multiplier busy
    uint64_t la[4] = {a, a ^ 0x1111111111111111ull, a ^ 0x2222222222222222ull, a ^ 0x3333333333333333ull}, lb[4] = {b, b + 1, b + 2, b + 3};
    size_t i = 0;
    for (; i + 32 <= n; i += 32) { uint64_t w[4]; std::memcpy(w, p + i, 32); for (int k = 0; k < 4; k++) mix(la[k], lb[k], w[k]); }
    for (; i + 8 <= n; i += 8) { uint64_t w; std::memcpy(&w, p + i, 8); mix(la[0], lb[0], w); }
    uint64_t w = 0; if (i < n) std::memcpy(&w, p + i, n - i);
    mix(la[0], lb[0], w ^ ((uint64_t)n << 56));
    for (int k = 1; k < 4; k++) { mix(la[0], lb[0], la[k]); mix(la[0], lb[0], lb[k]); }
    a = la[0]; b = lb[0];
}
// Hash of everything the plan depends on.  The long arrays (observation ids, the points' zero test) are cut into fixed pieces hashed on
// the planner's threads in ONE fork-join; the piece hashes are folded in order, so the result does not depend on the thread count.
void structure_hash(const ssfm_ba_problem* p, uint64_t& a, uint64_t& b) {
    a = 0x243F6A8885A308D3ull; b = 0x13198A2E03707344ull;
    constexpr size_t PIECE = 64 * 1024;
    const size_t nb_ids = (size_t)p->num_observations * sizeof(int32_t);
    const size_t np_ids = (nb_ids + PIECE - 1) / PIECE, PPTS = 16384, np_pts = ((size_t)p->num_points + PPTS - 1) / PPTS;
    const size_t np = 2 * np_ids + np_pts;
    std::vector<uint64_t> ha(np), hb(np);
    parallel_chunks((int64_t)np, np > 8 ? planner_threads() : 1, [&](int, int64_t lo, int64_t hi) {
        for (int64_t k = lo; k < hi; k++) {
            uint64_t x = 0x452821E638D01377ull + (uint64_t)k, y = 0xBE5466CF34E90C6Cull;
            if ((size_t)k < 2 * np_ids) {
                const unsigned char* base = reinterpret_cast<const unsigned char*>((size_t)k < np_ids ? p->obs_cam : p->obs_pt);
                const size_t kk = (size_t)k % np_ids, off = kk * PIECE;
                hash_chunk(base + off, std::min(PIECE, nb_ids - off), x, y);
            } else {                                             // which points are (0,0,0): they leave the problem (src/sfm.cpp:243)
                const size_t j0 = ((size_t)k - 2 * np_ids) * PPTS, j1 = std::min((size_t)p->num_points, j0 + PPTS);
                uint64_t bits = 0; int nb = 0;
                for (size_t j = j0; j < j1; j++) {
                    const double* X = p->points + 3 * j;
                    bits = (bits << 1) | ((X[0] * X[0] + X[1] * X[1] + X[2] * X[2]) == 0.0 ? 1u : 0u);
                    if (++nb == 64) { mix(x, y, bits); bits = 0; nb = 0; }
                }
                mix(x, y, bits ^ ((uint64_t)nb << 57));
            }
            ha[k] = x; hb[k] = y;
        }
    });
    for (size_t k = 0; k < np; k++) { mix(a, b, ha[k]); mix(a, b, hb[k]); }
    auto small = [&](const void* d, size_t n) { hash_chunk(static_cast<const unsigned char*>(d), n, a, b); };
    if (p->rot_fixed) small(p->rot_fixed, (size_t)p->num_cameras); else mix(a, b, 1);
    if (p->trans_fixed) small(p->trans_fixed, (size_t)p->num_cameras); else mix(a, b, 2);
    if (p->pt_fixed) small(p->pt_fixed, (size_t)p->num_points); else mix(a, b, 3);
}
// new parameter values into a resident handle whose structure matches p
int upload_state(ssfm_ba_handle* h, const ssfm_ba_problem* p) {
    ssfm_ctx* ctx = h->ctx; hipStream_t st = ctx->stream; BAFlat& F = h->F;
    if (F.nothing_to_do) return SSFM_OK;
    // gather points / pixels in the plan's order straight into a pinned staging buffer, on the planner's threads; then three asynchronous copies
    const double t_up0 = wall_s();
    const size_t n_pts = (size_t)F.nP * 3, n_xy = (size_t)F.M * 2;
    // the staging buffer belongs to the CONTEXT (grow-only; one stream, solves one after the other): a hipHostMalloc per handle on its first reuse cost 5 ms at 1 M
    // observations (26 MB) -- two of the four Optimize stages of the drivers' sequence paid it
    if (ctx->dl_stage_n < n_pts + n_xy + 1) {
        if (ctx->dl_stage) (void)hipHostFree(ctx->dl_stage);
        ctx->dl_stage = nullptr; ctx->dl_stage_n = 0;
        const size_t want = (n_pts + n_xy + 1) + (n_pts + n_xy + 1) / 4;
        SSFM_HIP_CHECK(ctx, hipHostMalloc((void**)&ctx->dl_stage, want * sizeof(double), hipHostMallocDefault));
        ctx->dl_stage_n = want;
    }
    double* sp = ctx->dl_stage; double* sx = ctx->dl_stage + n_pts;
    const int NT = planner_threads();
    parallel_chunks(F.M, NT, [&](int t, int64_t lo, int64_t hi) {      // one fork-join: every thread takes its share of the pixels and of the points
        for (int64_t j = lo; j < hi; j++) { const int64_t o = F.obs_orig[j]; sx[2 * j] = p->obs_xy[2 * o]; sx[2 * j + 1] = p->obs_xy[2 * o + 1]; }
        const int64_t T = (F.M < 4 * (int64_t)NT || NT <= 1) ? 1 : NT, q0 = (int64_t)F.nP * t / T, q1 = (int64_t)F.nP * (t + 1) / T;
        for (int64_t q = q0; q < q1; q++) for (int d = 0; d < 3; d++) sp[(size_t)q * 3 + d] = p->points[(size_t)F.pt_ids[q] * 3 + d]; });
    const double f3[3] = {*p->focal, *p->focal, *p->focal};
    h->focal_host = *p->focal;
    SSFM_HIP_CHECK(ctx, hipMemcpyAsync(h->cam_init.p, p->cameras, (size_t)F.Nc * 6 * sizeof(double), hipMemcpyHostToDevice, st));
    if (F.nP > 0) SSFM_HIP_CHECK(ctx, hipMemcpyAsync(h->pts_init.p, sp, n_pts * sizeof(double), hipMemcpyHostToDevice, st));
    if (F.M > 0) SSFM_HIP_CHECK(ctx, hipMemcpyAsync(h->obs_xy.p, sx, n_xy * sizeof(double), hipMemcpyHostToDevice, st));
    SSFM_HIP_CHECK(ctx, hipMemcpyAsync(h->focal3.p, f3, sizeof(f3), hipMemcpyHostToDevice, st));
    const double tg = wall_s();
    SSFM_HIP_CHECK(ctx, hipStreamSynchronize(st));                   // the host sources above are reused by the caller
    if (std::getenv("SSFM_PLAN_TIMING")) std::fprintf(stderr, "[solve] upload: gather+enqueue %.3f ms, copies %.3f ms\n", 1e3 * (tg - t_up0), 1e3 * (wall_s() - tg));
    return ssfm_ba_reset(h);
}
}  // namespace

extern "C" int ssfm_ba_solve(ssfm_ctx* ctx, ssfm_ba_problem* p, const ssfm_ba_options* o, ssfm_ba_summary* s) {
    if (!ctx || !p || !s) return fail(ctx, SSFM_ERR_INVALID, "ssfm_ba_solve: null argument");
    const bool no_cache = std::getenv("SSFM_NO_PLAN_CACHE") != nullptr;       // read per call: tests switch it
    const double t0 = wall_s();
    PlanCache key; key.Nc = p->num_cameras; key.Np = p->num_points; key.M = p->num_observations; key.nranks = ctx->nranks; key.rank = ctx->rank;
    key.focal_fixed = p->focal_fixed ? 1 : 0;
    PlanCache* pc = static_cast<PlanCache*>(ctx->plan_cache);
    ssfm_ba_handle* h = nullptr; bool reused = false; int rc = SSFM_OK;
    const bool timing = std::getenv("SSFM_PLAN_TIMING") != nullptr;
    if (!no_cache) {
        structure_hash(p, key.h1, key.h2);
        if (timing) std::fprintf(stderr, "[solve] structure hash %.3f ms\n", 1e3 * (wall_s() - t0));
        if (pc && pc->h && pc->Nc == key.Nc && pc->Np == key.Np && pc->M == key.M && pc->nranks == key.nranks && pc->rank == key.rank &&
            pc->focal_fixed == key.focal_fixed && pc->h1 == key.h1 && pc->h2 == key.h2 &&
            pc->h->det == (std::getenv("SSFM_DETERMINISTIC") && std::atoi(std::getenv("SSFM_DETERMINISTIC")) != 0)) {     // (the accumulation mode is a property of the handle)
            h = pc->h; reused = true;
            if (o) h->opt = *o; else ssfm_ba_default_options(&h->opt);
            h->t_flatten_s = 0.0;
            const double tu = wall_s();
            rc = upload_state(h, p);
            if (timing) std::fprintf(stderr, "[solve] upload_state %.3f ms\n", 1e3 * (wall_s() - tu));
        }
    }
    if (!reused) {
        if (pc) { if (pc->h) ssfm_ba_destroy(pc->h); pc->h = nullptr; }
        rc = ssfm_ba_create(ctx, p, o, &h);
        if (rc != SSFM_OK) { if (h) ssfm_ba_destroy(h); return rc; }
    }
    const double t1 = wall_s();
    if (rc == SSFM_OK) rc = ssfm_ba_run(h, s);
    const double t2 = wall_s();
    if (rc == SSFM_OK) rc = ssfm_ba_download(h, p);
    const double t3 = wall_s();
    s->t_flatten_s = h->t_flatten_s; s->t_upload_s = (t1 - t0) - h->t_flatten_s; s->t_download_s = t3 - t2;   // host planning | hash / allocation + H2D | D2H + scatter
    if (no_cache || rc != SSFM_OK) {
        if (reused && pc) pc->h = nullptr;
        ssfm_ba_destroy(h);
    } else {
        if (!pc) { pc = new PlanCache(); ctx->plan_cache = pc; ctx->plan_cache_free = plan_cache_free; }
        key.h = h; *pc = key;
    }
    return rc;
}
