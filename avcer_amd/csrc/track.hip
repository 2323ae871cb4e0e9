// HOST code of libavcer_hip.so: the face tracker between the detector and the tile kernel.
//
//   ref: data/face_detection/ibug/face_detection/utils/simple_face_tracker.py:10-90 (SimpleFaceTracker: IoU distances, Hungarian
//        assignment through scipy.optimize.linear_sum_assignment, tracklet bookkeeping), data/get_face_images.py:38-63
//        (VideoPredictor.process: one tracker call per frame, crop rectangle of every detection, one file per (track, frame)).
//
// The tracker is sequential in time and acts on a handful of boxes per frame, so it stays on the host as in the reference --
// but as ONE native call per video instead of 750 Python iterations of small numpy operations + scipy (25-30 ms per 30 s video on
// the GPU box's host cores, during which the visual branch's stream sat idle: profiles/r06_run_inference_trace_before.txt).
// Arithmetic follows avcer_amd/face_tiles.py SimpleFaceTracker operation for operation (float32 IoU arithmetic without
// contraction, float64 distance matrix); the assignment is the shortest-augmenting-path algorithm scipy implements (Crouse, "On
// implementing 2D rectangular assignment algorithms", 2016; scipy/optimize/rectangular_lsap), with its tie-breaking, so that
// equal-cost assignments resolve the same way (tests/test_host_logic.py checks it against scipy on tie-heavy matrices).
#include "common.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <limits>
#include <numeric>

namespace {

// scipy.optimize.linear_sum_assignment for a dense nr x nc cost matrix (row-major), nr, nc >= 0: fills `rows` / `cols` with the
// min(nr, nc) assigned pairs sorted by row.  Returns false if the problem is infeasible (cannot happen for finite costs).
bool lsap(int nr, int nc, const double* cost_in, std::vector<int>& rows, std::vector<int>& cols) {
    rows.clear();
    cols.clear();
    if (nr == 0 || nc == 0) return true;
    const bool transpose = nc < nr;
    std::vector<double> tmp;
    const double* cost = cost_in;
    if (transpose) {
        tmp.resize((size_t)nr * nc);
        for (int i = 0; i < nr; ++i)
            for (int j = 0; j < nc; ++j) tmp[(size_t)j * nr + i] = cost_in[(size_t)i * nc + j];
        std::swap(nr, nc);
        cost = tmp.data();
    }
    const double INF = std::numeric_limits<double>::infinity();
    std::vector<double> u(nr, 0.0), v(nc, 0.0), spc(nc);
    std::vector<int> path(nc, -1), col4row(nr, -1), row4col(nc, -1), remaining(nc);
    std::vector<char> SR(nr), SC(nc);
    for (int cur = 0; cur < nr; ++cur) {
        // augmenting path from row `cur`
        double min_val = 0.0;
        int num_remaining = nc;
        for (int it = 0; it < nc; ++it) remaining[it] = nc - it - 1;  // reverse fill: a constant matrix solves to the identity
        std::fill(SR.begin(), SR.end(), 0);
        std::fill(SC.begin(), SC.end(), 0);
        std::fill(spc.begin(), spc.end(), INF);
        int sink = -1, i = cur;
        while (sink == -1) {
            int index = -1;
            double lowest = INF;
            SR[i] = 1;
            for (int it = 0; it < num_remaining; ++it) {
                const int j = remaining[it];
                const double r = min_val + cost[(size_t)i * nc + j] - u[i] - v[j];
                if (r < spc[j]) {
                    path[j] = i;
                    spc[j] = r;
                }
                // among equal minima prefer a column that is a new sink
                if (spc[j] < lowest || (spc[j] == lowest && row4col[j] == -1)) {
                    lowest = spc[j];
                    index = it;
                }
            }
            min_val = lowest;
            if (min_val == INF) return false;
            const int j = remaining[index];
            if (row4col[j] == -1) sink = j;
            else i = row4col[j];
            SC[j] = 1;
            remaining[index] = remaining[--num_remaining];
        }
        // dual update
        u[cur] += min_val;
        for (int r = 0; r < nr; ++r)
            if (SR[r] && r != cur) u[r] += min_val - spc[col4row[r]];
        for (int j = 0; j < nc; ++j)
            if (SC[j]) v[j] -= min_val - spc[j];
        // augment
        int j = sink;
        while (true) {
            const int r = path[j];
            row4col[j] = r;
            std::swap(col4row[r], j);
            if (r == cur) break;
        }
    }
    if (transpose) {
        std::vector<int> idx(nr);
        std::iota(idx.begin(), idx.end(), 0);
        std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return col4row[a] < col4row[b]; });
        for (int k : idx) {
            rows.push_back(col4row[k]);
            cols.push_back(k);
        }
    } else {
        for (int r = 0; r < nr; ++r) {
            rows.push_back(r);
            cols.push_back(col4row[r]);
        }
    }
    return true;
}

// numpy's `float32 -> int64` cast (truncate toward zero; non-finite and out-of-range values become INT64_MIN)
inline long long f2i(float x) {
    if (!(x > -9.2e18f && x < 9.2e18f)) return std::numeric_limits<long long>::min();
    return (long long)x;
}

// `slice(max(0, s), min(size - 1, e)).indices(size)` of avcer_amd/face_tiles.py crop_rects: the half-open range numpy's
// `fr[s:e]` selects after the reference's clamping (a negative stop counts from the end, like any Python slice)
inline void slice_range(long long s, long long e, long long size, long long& lo, long long& hi) {
    long long start = std::max<long long>(0, s);
    if (start > size) start = size;
    long long stop = std::min<long long>(size - 1, e);
    if (stop < 0) {
        stop += size;
        if (stop < 0) stop = 0;
    }
    lo = start;
    hi = std::max(start, stop);
}

struct Tracker {
    double iou_threshold;
    double minimum_face_size;
    std::vector<float> boxes;  // [m][4]
    std::vector<float> areas;  // [m]
    std::vector<int> ids;      // [m]
    int counter = 0;

    // face_boxes [n][ld] (x0, y0, x1, y1, ...) -> ids[n] (0 = None)
    void step(const float* fb, int ld, int n, std::vector<int>& out) {
#pragma clang fp contract(off)
        out.assign(n, 0);
        if (n <= 0) {  // a frame without faces drops every tracklet, the counter runs on
            boxes.clear();
            areas.clear();
            ids.clear();
            return;
        }
        const int m = (int)ids.size();
        std::vector<float> ar(n);
        std::vector<char> big(n);
        const double thr = std::min(1.0, std::max(0.0, 1.0 - iou_threshold));
        // `areas >= max(minimum_face_size ** 2, np.finfo(float).eps)`: a float32 array against a Python float is a float32 comparison
        const float floor_area = (float)std::max(minimum_face_size * minimum_face_size, DBL_EPSILON);
        for (int r = 0; r < n; ++r) {
            const float* b = fb + (size_t)r * ld;
            const float dx = b[2] - b[0], dy = b[3] - b[1];
            ar[r] = std::fabs(dx * dy);
            big[r] = ar[r] >= floor_area;
        }
        std::vector<double> dist((size_t)n * m, 2.0 * std::min(n, m));
        for (int r = 0; r < n && m; ++r) {
            const float* b = fb + (size_t)r * ld;
            const float bx0 = std::min(b[0], b[2]), bx1 = std::max(b[0], b[2]), by0 = std::min(b[1], b[3]), by1 = std::max(b[1], b[3]);
            for (int c = 0; c < m; ++c) {
                const float* t = boxes.data() + (size_t)c * 4;
                const float xl = std::max(bx0, std::min(t[0], t[2])), yt = std::max(by0, std::min(t[1], t[3]));
                const float xr = std::min(bx1, std::max(t[0], t[2])), yb = std::min(by1, std::max(t[1], t[3]));
                const float inter = (xr - xl) * (yb - yt);
                const float uni = (ar[r] + areas[c]) - inter;
                double d = (double)(1.0f - inter / uni);
                if (xr <= xl || yb <= yt) d = 1.0;
                if (d <= thr && big[r]) dist[(size_t)r * m + c] = d;
            }
        }
        std::vector<int> rr, cc;
        lsap(n, m, dist.data(), rr, cc);
        std::vector<float> nb = boxes, na = areas;
        std::vector<char> tracked(m, 0);
        for (size_t k = 0; k < rr.size(); ++k) {
            const int r = rr[k], c = cc[k];
            if (dist[(size_t)r * m + c] <= thr) {
                out[r] = ids[c];
                std::copy(fb + (size_t)r * ld, fb + (size_t)r * ld + 4, nb.begin() + (size_t)c * 4);
                na[c] = ar[r];
                tracked[c] = 1;
            }
        }
        std::vector<float> kb, ka;
        std::vector<int> ki;
        for (int c = 0; c < m; ++c)
            if (tracked[c]) {
                kb.insert(kb.end(), nb.begin() + (size_t)c * 4, nb.begin() + (size_t)c * 4 + 4);
                ka.push_back(na[c]);
                ki.push_back(ids[c]);
            }
        for (int r = 0; r < n; ++r)
            if (big[r] && out[r] == 0) {
                out[r] = ++counter;
                kb.insert(kb.end(), fb + (size_t)r * ld, fb + (size_t)r * ld + 4);
                ka.push_back(ar[r]);
                ki.push_back(out[r]);
            }
        boxes.swap(kb);
        areas.swap(ka);
        ids.swap(ki);
    }
};

}  // namespace

extern "C" int avcer_lsap(int nr, int nc, const double* cost, int32_t* rows, int32_t* cols) {
    if (nr < 0 || nc < 0 || (nr && nc && !cost) || !rows || !cols) return AVCER_EINVAL;
    std::vector<int> r, c;
    if (!lsap(nr, nc, cost, r, c)) return AVCER_EINVAL;
    for (size_t k = 0; k < r.size(); ++k) {
        rows[k] = r[k];
        cols[k] = c[k];
    }
    return AVCER_OK;
}

extern "C" int avcer_track_faces(avcer_ctx* ctx, const float* dets, int ld, const int32_t* counts, int n_frames, int frame_w,
                                 int frame_h, double iou_threshold, double minimum_face_size, int64_t* records, int64_t* n_records) {
    // ctx may be NULL (host-only call, no device needed): errors then come back as the code alone
    if (!counts || !records || !n_records || n_frames < 0 || ld < 4 || frame_w <= 0 || frame_h <= 0)
        return set_err(ctx, AVCER_EINVAL, "track_faces: bad arguments");
    Tracker tr;
    tr.iou_threshold = iou_threshold;
    tr.minimum_face_size = minimum_face_size;
    std::vector<int> tids;
    size_t row = 0;
    int64_t n_out = 0;
    for (int t = 0; t < n_frames; ++t) {
        const int n = counts[t];
        if (n < 0 || (n > 0 && !dets)) return set_err(ctx, AVCER_EINVAL, "track_faces: frame %d has %d detections", t, n);
        const float* fb = dets ? dets + row * (size_t)ld : nullptr;
        tr.step(fb, ld, n, tids);
        for (int r = 0; r < n; ++r) {
            const float* b = fb + (size_t)r * ld;
            long long x0, x1, y0, y1;
            slice_range(f2i(b[0]), f2i(b[2]), frame_w, x0, x1);
            slice_range(f2i(b[1]), f2i(b[3]), frame_h, y0, y1);
            if (tids[r] == 0)
                return set_err(ctx, AVCER_EINVAL, "frame %d: a zero-area detection has no track id (TypeError in the reference)", t);
            if (x1 <= x0 || y1 <= y0)
                return set_err(ctx, AVCER_EINVAL, "frame %d: empty crop (%lld, %lld, %lld, %lld) (cv2.imwrite fails in the reference)", t,
                               x0, y0, x1, y1);
            int64_t* o = records + 6 * n_out++;
            o[0] = t; o[1] = tids[r] - 1; o[2] = x0; o[3] = y0; o[4] = x1; o[5] = y1;
        }
        row += (size_t)n;
    }
    *n_records = n_out;
    return AVCER_OK;
}
