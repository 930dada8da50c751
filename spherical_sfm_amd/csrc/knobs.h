// spherical_sfm_amd -- environment knobs, in two classes.
//
// PRODUCT switches select a SUPPORTED alternative path (a plan shape, a fall-back, a hand-over mode) that the tests exercise and a maintainer may need:
//   SSFM_DETERMINISTIC (det_acc.h: order-independent accumulation), SSFM_GRAM_ANY, SSFM_RING, SSFM_RING_CUTS, SSFM_BAND_TWIST, SSFM_BAND_MERGE, SSFM_BAND_SEGMENTS, SSFM_BAND_PACKED, SSFM_GRAM, SSFM_GRAM_KMIN, SSFM_GRAM_PTS, SSFM_GRAM_MIN_RUN,
//   SSFM_GRAM_SORT, SSFM_GRAM_MODEL, SSFM_GRAM_BACKSUB, SSFM_NO_PLAN_CACHE, SSFM_HOST_PAIRS, SSFM_LM_POLL, SSFM_LM_SPECULATE, SSFM_ROT_NODE_MAJOR, SSFM_RETRI_ENUMERATE,
//   SSFM_RETRI_WAVES, SSFM_RETRI_WORDS, SSFM_RANSAC_SLAB_*, SSFM_RANSAC_STAGE_THREADS, SSFM_PLAN_THREADS, SSFM_PLAN_TIMING, SSFM_PLAN_OVERLAP, SSFM_TASK_BATCHES,
//   SSFM_CS_TASK_OBS, SSFM_COMM_SINGLE_RANK.  They are read where they apply, most of them once per process.
//
// LAB knobs select a variant that was MEASURED AND REJECTED, a kernel-shape sweep or a timing study (profiles/r0*_notes.md say which): the shipped library compiles
// their defaults in and does not contain the rejected kernels; `make lab` builds libssfm_hip_lab.so with -DSSFM_LAB, where every one of them is live again
// (Python: SSFM_LIB_PATH=spherical_sfm_amd/libssfm_hip_lab.so).  SSFM_LAB_KNOB(name, default) is the value, read once per call site.
#pragma once
#include <cstddef>
#include <cstdlib>

namespace ssfm {
// LDS bytes a workgroup may ask for on the device the plans are made for: 160 KB on gfx950 (the only target); ssfm_ctx_create overwrites it with the device's attribute
inline size_t& plan_lds_limit() { static size_t v = 160 * 1024; return v; }
inline int knob_env_int(const char* name, int dflt) { const char* e = std::getenv(name); return e ? std::atoi(e) : dflt; }
}  // namespace ssfm

#ifdef SSFM_LAB
#define SSFM_LAB_KNOB(name, dflt) ([] { static const int v_ = ::ssfm::knob_env_int(name, dflt); return v_; }())
#else
#define SSFM_LAB_KNOB(name, dflt) (dflt)
#endif
