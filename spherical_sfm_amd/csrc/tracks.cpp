// spherical_sfm_amd -- feature-track assignment of build_sfm (reference examples/spherical_sfm_tools.cpp:862-950).
//
// Integer bookkeeping on the host ("host, integer; not a GPU target"), bit-exact with the reference:
// image matches are consumed in order, each match list in ascending first-feature order (Matches = std::map<size_t,size_t>),
// with the reference's four cases
//   (set, unset)   -> the unset feature joins the track, AddObservation(index1, track0)        :913-917
//   (unset, set)   -> symmetric                                                             :918-922
//   (unset, unset) -> new point = AddPoint() (ids count up from 0), both observations added  :923-931
//   (set, set, different) -> merge: MergePoint(track0, track1) + retag every feature of track1  :932-945
//                            no merge: both observations added to their own tracks              :946-949
// AddObservation overwrites an existing (camera, point) entry (src/sfm.cpp:143-146); MergePoint copies the removed point's
// observations onto the kept one camera by camera, then removes the point (src/sfm.cpp:129-141, 435-444).
// The reference rescans all features of all keyframes per merge; here every track keeps its member list (same result).
#include <cstdint>
#include <cstring>
#include <map>
#include <vector>
#include "../../include/ssfm.h"

extern "C" int ssfm_build_tracks(int32_t num_keyframes, const int32_t* feat_ptr, const double* feat_xy, int32_t num_match_sets,
                                 const int32_t* ms_index0, const int32_t* ms_index1, const int32_t* ms_ptr, const int32_t* m_f0,
                                 const int32_t* m_f1, double centerx, double centery, int32_t merge, int32_t* tracks,
                                 int32_t* num_points, uint8_t* point_alive, int64_t* num_observations, int32_t* obs_cam, int32_t* obs_pt,
                                 double* obs_xy) {
    if (num_keyframes <= 0 || !feat_ptr || !tracks || !num_points || !num_observations) return SSFM_ERR_INVALID;
    const int total = feat_ptr[num_keyframes];
    for (int i = 0; i < total; i++) tracks[i] = -1;                                   // :865-871
    struct Obs { double x, y; };
    std::vector<std::map<int, Obs>> obs(num_keyframes);                               // observations[camera][point]
    std::vector<std::vector<int>> members;                                            // track -> global feature ids
    std::vector<uint8_t> alive;
    auto add_point = [&]() { members.emplace_back(); alive.push_back(1); return (int)members.size() - 1; };
    for (int s = 0; s < num_match_sets; s++) {
        const int i0 = ms_index0[s], i1 = ms_index1[s];
        if (i0 < 0 || i0 >= num_keyframes || i1 < 0 || i1 >= num_keyframes) return SSFM_ERR_INVALID;
        for (int m = ms_ptr[s]; m < ms_ptr[s + 1]; m++) {
            const int g0 = feat_ptr[i0] + m_f0[m], g1 = feat_ptr[i1] + m_f1[m];
            if (g0 >= feat_ptr[i0 + 1] || g1 >= feat_ptr[i1 + 1] || m_f0[m] < 0 || m_f1[m] < 0) return SSFM_ERR_INVALID;
            const Obs o0{feat_xy[2 * g0] - centerx, feat_xy[2 * g0 + 1] - centery}, o1{feat_xy[2 * g1] - centerx, feat_xy[2 * g1 + 1] - centery};   // :907-908
            int& t0 = tracks[g0]; int& t1 = tracks[g1];
            if (t0 != -1 && t1 == -1) { t1 = t0; members[t0].push_back(g1); obs[i1][t0] = o1; }
            else if (t0 == -1 && t1 != -1) { t0 = t1; members[t1].push_back(g0); obs[i0][t1] = o0; }
            else if (t0 == -1 && t1 == -1) {
                const int p = add_point(); t0 = t1 = p; members[p].push_back(g0); members[p].push_back(g1);
                obs[i0][p] = o0; obs[i1][p] = o1;
            } else if (t0 != t1) {
                if (merge) {
                    const int keep = t0, gone = t1;
                    for (int c = 0; c < num_keyframes; c++) {                         // MergePoint: camera by camera
                        auto it = obs[c].find(gone);
                        if (it != obs[c].end()) { obs[c][keep] = it->second; obs[c].erase(gone); }
                    }
                    alive[gone] = 0;
                    for (int g : members[gone]) { tracks[g] = keep; members[keep].push_back(g); }
                    members[gone].clear();
                } else { obs[i0][t0] = o0; obs[i1][t1] = o1; }
            }
        }
    }
    *num_points = (int32_t)members.size();
    if (point_alive && !alive.empty()) std::memcpy(point_alive, alive.data(), alive.size());
    int64_t n = 0;
    for (int c = 0; c < num_keyframes; c++)
        for (auto& kv : obs[c]) { if (obs_cam) { obs_cam[n] = c; obs_pt[n] = kv.first; obs_xy[2 * n] = kv.second.x; obs_xy[2 * n + 1] = kv.second.y; } n++; }
    *num_observations = n;
    return SSFM_OK;
}
