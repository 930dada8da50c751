// spherical_sfm_amd -- context shared by the C-ABI entry points (include/ssfm.h).
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <cstdio>
#include <mutex>
#include <string>
#include <vector>
#include "../../include/ssfm.h"

struct ssfm_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;
    ncclComm_t comm = nullptr;   // RCCL communicator (one rank per GPU), null for single-GPU
    int nranks = 1, rank = 0;
    bool timing_skip_collectives = false;      // ssfm_debug_timing_skip_collectives (bench.py's timing probe): the BA reductions return at once -- results are NOT a solution
    bool collective = false;     // reductions go through RCCL or the host hook (nranks > 1, or a forced 1-rank communicator)
    ssfm_host_allreduce_fn host_allreduce = nullptr; void* host_allreduce_user = nullptr;   // alternative to RCCL
    double* host_stage = nullptr; size_t host_stage_n = 0;                                  // pinned staging for the hook
    double* dl_stage = nullptr; size_t dl_stage_n = 0;                                      // pinned staging of upload_state / ssfm_ba_download (grow-only: outlives the handles)
    int num_cus = 256;
    // one-entry plan cache of ssfm_ba_solve (ba_solver.hip): the resident handle of the last problem structure
    void* plan_cache = nullptr; void (*plan_cache_free)(void*) = nullptr;
    // coherent pinned block the LM loops publish their end-of-iteration scalars into (k_publish) + its sequence number; owned by the
    // context (one stream, solves run one after the other): a hipHostMalloc per handle cost 0.2 ms of a 3 ms solve
    double* host_pub = nullptr; unsigned long long pub_seq = 0;
    double ransac_kernel_ms = 0.0;      // device time of the kernels of the last ssfm_ransac_batch* call (hipEvent brackets per slab), ssfm_ransac_last_kernel_ms
};

namespace ssfm {

// Device-memory recycling.  A handle owns ~80 buffers; hipFree synchronises the device, and tearing a handle down on a structure
// change cost 3 ms of the 16 ms of a first call.  Freed buffers go to this pool and the next handle's allocations are served from it
// (best fit, at most twice the requested size); ssfm_ctx_destroy drains it, and so does an allocation that the driver refuses.
// RULE (the pool is keyed by device, not by stream): a buffer is only given to the pool after the stream that used it has been
// synchronised -- ssfm_ba_destroy / free_all and every entry point that frees its temporaries do so -- which is what hipFree's implicit
// synchronisation used to guarantee.
struct DevPool {
    struct Blk { void* p; size_t bytes; int dev; };
    std::vector<Blk> blocks; std::mutex m; size_t total = 0;
    void* take(size_t bytes, int dev, size_t* got) {
        std::lock_guard<std::mutex> g(m);
        int best = -1;
        for (int i = 0; i < (int)blocks.size(); i++)
            if (blocks[i].dev == dev && blocks[i].bytes >= bytes && blocks[i].bytes <= 2 * bytes + 4096 && (best < 0 || blocks[i].bytes < blocks[best].bytes)) best = i;
        if (best < 0) return nullptr;
        void* p = blocks[best].p; *got = blocks[best].bytes; total -= blocks[best].bytes;
        blocks[best] = blocks.back(); blocks.pop_back();
        return p;
    }
    bool give(void* p, size_t bytes, int dev) {
        std::lock_guard<std::mutex> g(m);
        if (blocks.size() >= 1024 || total + bytes > ((size_t)16 << 30)) return false;
        blocks.push_back({p, bytes, dev}); total += bytes;
        return true;
    }
    void drain(int dev) {
        std::lock_guard<std::mutex> g(m);
        for (size_t i = 0; i < blocks.size();) {
            if (blocks[i].dev == dev) { (void)hipFree(blocks[i].p); total -= blocks[i].bytes; blocks[i] = blocks.back(); blocks.pop_back(); } else i++;
        }
    }
};
inline DevPool g_dev_pool;

extern std::string g_last_error;   // for failures before a context exists

inline int fail(ssfm_ctx* ctx, int code, const std::string& msg) {
    if (ctx) ctx->err = msg; else g_last_error = msg;
    return code;
}

#define SSFM_HIP_CHECK(ctx, expr)                                                                     \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess)                                                                         \
            return ssfm::fail(ctx, SSFM_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

#define SSFM_NCCL_CHECK(ctx, expr)                                                                     \
    do {                                                                                               \
        ncclResult_t r_ = (expr);                                                                      \
        if (r_ != ncclSuccess)                                                                         \
            return ssfm::fail(ctx, SSFM_ERR_COMM, std::string(#expr) + ": " + ncclGetErrorString(r_)); \
    } while (0)

// Sum / max all-reduce of a device buffer of doubles over the ranks attached to the context (RCCL on the context's stream, or
// the caller-supplied host collective staged through pinned memory).  No-op without a communicator.
inline int ctx_allreduce(ssfm_ctx* ctx, double* buf, size_t n, ncclRedOp_t op) {
    if (!ctx->collective) return SSFM_OK;
    if (ctx->host_allreduce) {
        if (ctx->host_stage_n < n) {
            if (ctx->host_stage) (void)hipHostFree(ctx->host_stage);
            ctx->host_stage = nullptr; ctx->host_stage_n = 0;
            SSFM_HIP_CHECK(ctx, hipHostMalloc((void**)&ctx->host_stage, n * sizeof(double), hipHostMallocDefault));
            ctx->host_stage_n = n;
        }
        SSFM_HIP_CHECK(ctx, hipMemcpyAsync(ctx->host_stage, buf, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        SSFM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->host_allreduce(ctx->host_allreduce_user, ctx->host_stage, (uint64_t)n, op == ncclMax ? SSFM_REDUCE_MAX : SSFM_REDUCE_SUM) != 0)
            return fail(ctx, SSFM_ERR_COMM, "host all-reduce hook failed");
        SSFM_HIP_CHECK(ctx, hipMemcpyAsync(buf, ctx->host_stage, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        return SSFM_OK;
    }
    ncclResult_t r = ncclAllReduce(buf, buf, n, ncclDouble, op, ctx->comm, ctx->stream);
    if (r != ncclSuccess) return fail(ctx, SSFM_ERR_COMM, std::string("ncclAllReduce: ") + ncclGetErrorString(r));
    return SSFM_OK;
}

}  // namespace ssfm
