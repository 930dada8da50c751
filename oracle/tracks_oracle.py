"""ORACLE (test infrastructure only) -- literal pure-Python restatement of the track loop of build_sfm
(reference examples/spherical_sfm_tools.cpp:862-950) on top of SfM's map semantics (src/sfm.cpp:113-146, 435-444):
per-keyframe track arrays, dict-of-dict observations, full rescans on merge exactly as the reference does them.
Small inputs only.  PARITY UNPINNED for floats; the integer outputs are what the tests compare bit-exactly."""


def build_tracks(features, image_matches, centerx=0.0, centery=0.0, merge=True):
    K = len(features)
    tracks = [[-1] * len(f) for f in features]                      # :865-871
    observations = {}                                               # observations[camera][point] = (x, y)
    points = {}                                                     # live points
    next_point = 0
    num_cameras = K
    for index0, index1, matches in image_matches:                   # :887
        for f0, f1 in sorted(dict(matches).items()):                # Matches = std::map<size_t,size_t>
            pt0 = features[index0][f0]; pt1 = features[index1][f1]
            obs0 = (pt0[0] - centerx, pt0[1] - centery); obs1 = (pt1[0] - centerx, pt1[1] - centery)
            t0 = tracks[index0][f0]; t1 = tracks[index1][f1]
            if t0 != -1 and t1 == -1:
                tracks[index1][f1] = t0
                observations.setdefault(index1, {})[t0] = obs1
            elif t0 == -1 and t1 != -1:
                tracks[index0][f0] = t1
                observations.setdefault(index0, {})[t1] = obs0
            elif t0 == -1 and t1 == -1:
                p = next_point; next_point += 1; points[p] = True    # AddPoint, src/sfm.cpp:113-127
                tracks[index0][f0] = tracks[index1][f1] = p
                observations.setdefault(index0, {})[p] = obs0
                observations.setdefault(index1, {})[p] = obs1
            elif t0 != t1:
                if merge:
                    for i in range(num_cameras):                    # MergePoint, src/sfm.cpp:129-141
                        if i in observations and t1 in observations[i]:
                            observations[i][t0] = observations[i][t1]
                    for i in range(num_cameras):                    # RemovePoint, src/sfm.cpp:435-444
                        if i in observations:
                            observations[i].pop(t1, None)
                    points.pop(t1, None)
                    for i in range(K):                              # :937-943
                        for j in range(len(tracks[i])):
                            if tracks[i][j] == t1:
                                tracks[i][j] = t0
                else:
                    observations.setdefault(index0, {})[t0] = obs0
                    observations.setdefault(index1, {})[t1] = obs1
    obs = [(c, p, xy) for c in sorted(observations) for p, xy in sorted(observations[c].items())]
    return dict(tracks=tracks, num_points=next_point, alive=[p in points for p in range(next_point)], obs=obs)
