// Host-side (CPU) part of one frame, native: packed decode rows -> image-space detections
// (inverse affine) -> score cut -> greedy displacement association -> track list, and the
// prior-heat-map parameters of the NEXT frame.  Restates, with identical float32 /
// float64 promotion points so that track IDs are bit-identical:
//   generic_post_process      src/lib/utils/post_process.py:21-91 (2D fields + the ddd centre)
//   Detector.merge_outputs    src/lib/detector.py:371-377
//   Tracker.init_track / step src/lib/utils/tracker.py:13-127 (greedy_assignment :129-138, the --hungarian branch
//                             :52-55,63-73 and the --public_det branch :83-101)
//   Detector._get_additional_inputs / _trans_bbox / gaussian_radius
//                             src/lib/detector.py:242-290, src/lib/utils/image.py:105-126
// --hungarian: the reference calls sklearn's removed linear_assignment_; every modern environment (and the golden
// vectors) substitutes scipy.optimize.linear_sum_assignment, whose rectangular solver (shortest augmenting paths,
// D. F. Crouse 2016) is restated in ct_linear_assignment below, including its tie-breaking, because which of several
// equally gated-out pairs gets "matched" decides the order in which new track ids are handed out.
// Compiled with -ffp-contract=off: every float32 product/sum is rounded separately, like
// numpy's element-wise arithmetic.
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include <xmmintrin.h>

#include "centertrack_hip.h"

void ct_set_error(const char *fmt, ...);

namespace {

struct Track {
    ct_track t;
};

struct Tracker {
    float new_thresh;
    int max_age;
    int id_count;
    int hungarian = 0, public_det = 0;
    std::vector<ct_track> tracks;
    // per-step scratch (kept to avoid heap traffic on the frame loop)
    std::vector<ct_track> scratch_dets, scratch_ret;
    std::vector<float> scratch_t;
    std::vector<int> scratch_dm, scratch_tm;
    std::vector<double> scratch_cost;
};

inline void xform(const float *m, float x, float y, float *ox, float *oy)
{
    // trans[2,3] (float32) applied to (x, y, 1)
    *ox = (m[0] * x + m[1] * y) + m[2];
    *oy = (m[3] * x + m[4] * y) + m[5];
}

inline float area(const float *b) { return (b[2] - b[0]) * (b[3] - b[1]); }

double gaussian_radius(long height, long width)
{
    const double mo = 0.7;
    const double hw = (double)(width * height);
    const long b1 = height + width;
    const double c1 = hw * (1 - mo) / (1 + mo);
    const double r1 = ((double)b1 + sqrt((double)(b1 * b1) - 4.0 * c1)) / 2;
    const long b2 = 2 * (height + width);
    const double c2 = (1 - mo) * (double)width * (double)height;
    const double r2 = ((double)b2 + sqrt((double)(b2 * b2) - 16.0 * c2)) / 2;
    const double a3 = 4 * mo;
    const double b3 = -2 * mo * (double)(height + width);
    const double c3 = (mo - 1) * (double)width * (double)height;
    const double r3 = (b3 + sqrt(b3 * b3 - 4 * a3 * c3)) / 2;
    double r = r1 < r2 ? r1 : r2;
    return r < r3 ? r : r3;
}

// One shortest augmenting path from free row `i` (Crouse's algorithm as scipy's rectangular solver runs it): columns
// are scanned in the order of `remaining` (filled back to front), a tie on the smallest reduced cost prefers a column
// that is still free.  Returns the sink column, or -1 for an infeasible matrix.
long augmenting_path(long nc, const double *cost, std::vector<double> &u, std::vector<double> &v, std::vector<long> &path,
                     std::vector<long> &row4col, std::vector<double> &shortest, long i, std::vector<char> &SR,
                     std::vector<char> &SC, std::vector<long> &remaining, double *p_min)
{
    double min_val = 0;
    long num_remaining = nc;
    for (long it = 0; it < nc; ++it) remaining[it] = nc - it - 1;
    std::fill(SR.begin(), SR.end(), 0);
    std::fill(SC.begin(), SC.end(), 0);
    std::fill(shortest.begin(), shortest.end(), INFINITY);
    long sink = -1;
    while (sink == -1) {
        long index = -1;
        double lowest = INFINITY;
        SR[i] = 1;
        for (long it = 0; it < num_remaining; ++it) {
            const long j = remaining[it];
            const double r = min_val + cost[i * nc + j] - u[i] - v[j];
            if (r < shortest[j]) {
                path[j] = i;
                shortest[j] = r;
            }
            if (shortest[j] < lowest || (shortest[j] == lowest && row4col[j] == -1)) {
                lowest = shortest[j];
                index = it;
            }
        }
        min_val = lowest;
        if (min_val == INFINITY) return -1;
        const long j = remaining[index];
        if (row4col[j] == -1) sink = j;
        else i = row4col[j];
        SC[j] = 1;
        remaining[index] = remaining[--num_remaining];
    }
    *p_min = min_val;
    return sink;
}

// rows[k], cols[k], k < min(nr, nc): minimum-cost assignment of the nr x nc matrix, pairs sorted by row
int lsap(long nr, long nc, const double *cost_in, long *rows, long *cols)
{
    if (nr == 0 || nc == 0) return 0;
    const bool transpose = nc < nr;                 // a tall matrix is solved transposed
    std::vector<double> temp;
    const double *cost = cost_in;
    if (transpose) {
        temp.resize((size_t)nr * nc);
        for (long i = 0; i < nr; ++i)
            for (long j = 0; j < nc; ++j) temp[(size_t)j * nr + i] = cost_in[(size_t)i * nc + j];
        std::swap(nr, nc);
        cost = temp.data();
    }
    for (size_t k = 0; k < (size_t)nr * nc; ++k)
        if (cost[k] != cost[k] || cost[k] == -INFINITY) return -1;
    std::vector<double> u(nr, 0.0), v(nc, 0.0), shortest(nc);
    std::vector<long> path(nc, -1), col4row(nr, -1), row4col(nc, -1), remaining(nc);
    std::vector<char> SR(nr), SC(nc);
    for (long cur = 0; cur < nr; ++cur) {
        double min_val;
        const long sink = augmenting_path(nc, cost, u, v, path, row4col, shortest, cur, SR, SC, remaining, &min_val);
        if (sink < 0) return -1;
        u[cur] += min_val;
        for (long i = 0; i < nr; ++i)
            if (SR[i] && i != cur) u[i] += min_val - shortest[col4row[i]];
        for (long j = 0; j < nc; ++j)
            if (SC[j]) v[j] -= min_val - shortest[j];
        long j = sink;
        while (true) {
            const long i = path[j];
            row4col[j] = i;
            std::swap(col4row[i], j);
            if (i == cur) break;
        }
    }
    if (transpose) {
        // (rows of the transposed problem are the original columns: report pairs sorted by original row)
        std::vector<long> order(nr);
        for (long i = 0; i < nr; ++i) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](long a, long b) { return col4row[a] < col4row[b]; });
        for (long k = 0; k < nr; ++k) {
            rows[k] = col4row[order[k]];
            cols[k] = order[k];
        }
    } else {
        for (long i = 0; i < nr; ++i) {
            rows[i] = i;
            cols[i] = col4row[i];
        }
    }
    return (int)nr;
}

}  // namespace

extern "C" int ct_linear_assignment(const double *cost, int nr, int nc, int *rows, int *cols)
{
    if ((!cost && nr > 0 && nc > 0) || nr < 0 || nc < 0 || !rows || !cols) {
        ct_set_error("ct_linear_assignment: bad argument");
        return -1;
    }
    const int n = nr < nc ? nr : nc;
    std::vector<long> r(n > 0 ? n : 1), c(n > 0 ? n : 1);
    const int got = lsap(nr, nc, cost, r.data(), c.data());
    if (got < 0) {
        ct_set_error("ct_linear_assignment: infeasible or invalid cost matrix");
        return -1;
    }
    for (int k = 0; k < got; ++k) {
        rows[k] = (int)r[k];
        cols[k] = (int)c[k];
    }
    return got;
}

extern "C" void *ct_tracker_create(float new_thresh, int max_age)
{
    Tracker *t = new Tracker();
    t->new_thresh = new_thresh;
    t->max_age = max_age;
    t->id_count = 0;
    return t;
}

extern "C" void ct_tracker_destroy(void *h) { delete static_cast<Tracker *>(h); }

extern "C" void ct_tracker_reset(void *h)
{
    Tracker *t = static_cast<Tracker *>(h);
    t->id_count = 0;
    t->tracks.clear();
}

extern "C" int ct_tracker_num_tracks(void *h) { return (int)static_cast<Tracker *>(h)->tracks.size(); }
extern "C" int ct_tracker_id_count(void *h) { return static_cast<Tracker *>(h)->id_count; }

extern "C" int ct_tracker_get_tracks(void *h, ct_track *out, int cap)
{
    Tracker *t = static_cast<Tracker *>(h);
    const int n = (int)t->tracks.size() < cap ? (int)t->tracks.size() : cap;
    if (n > 0) memcpy(out, t->tracks.data(), sizeof(ct_track) * n);
    return (int)t->tracks.size();
}

namespace {

// packed decode rows -> image-space detections (generic_post_process + merge_outputs): stop at the first
// score < out_thresh, keep score > out_thresh
void rows_to_dets(const float *rows, int K, int F, const ct_row_layout *lay, float out_thresh, const float *trans_inv,
                  std::vector<ct_track> &dets)
{
    dets.clear();
    dets.reserve(K);
    for (int j = 0; j < K; ++j) {
        const float *r = rows + (size_t)j * F;
        const float score = r[lay->score];
        if (score < out_thresh) break;
        ct_track d;
        memset(&d, 0, sizeof(d));
        d.score = score;
        d.cls = (int)r[lay->cls] + 1;
        d.row = j;
        const float cx = r[lay->cts], cy = r[lay->cts + 1];
        xform(trans_inv, cx, cy, &d.ct[0], &d.ct[1]);
        float tx, ty;
        xform(trans_inv, r[lay->tracking] + cx, r[lay->tracking + 1] + cy, &tx, &ty);
        d.tracking[0] = tx - d.ct[0];
        d.tracking[1] = ty - d.ct[1];
        xform(trans_inv, r[lay->bbox], r[lay->bbox + 1], &d.bbox[0], &d.bbox[1]);
        xform(trans_inv, r[lay->bbox + 2], r[lay->bbox + 3], &d.bbox[2], &d.bbox[3]);
        if (lay->amodel_offset >= 0) {
            // ddd: the tracked centre is the projected amodal centre (post_process.py:66-75)
            const float mx = (r[lay->bbox] + r[lay->bbox + 2]) / 2.0f + r[lay->amodel_offset];
            const float my = (r[lay->bbox + 1] + r[lay->bbox + 3]) / 2.0f + r[lay->amodel_offset + 1];
            xform(trans_inv, mx, my, &d.ct[0], &d.ct[1]);
        }
        if (score > out_thresh) dets.push_back(d);
    }
}

// greedy_assignment (tracker.py:129-138) over the implicit distance matrix: det_match / trk_match
void match_greedy(Tracker *tr, const std::vector<ct_track> &dets, std::vector<int> &det_match, std::vector<int> &trk_match)
{
    const int N = (int)dets.size(), M = (int)tr->tracks.size();
    // dist[i][m] = float32 squared distance, + 1e18 (promoting to float64) when gated out; greedy_assignment
    // walks the detections in score order, takes the FIRST minimum of its row and, if it is < 1e16, overwrites
    // that track's column with 1e18.  The matrix is not materialised: a row is evaluated when its detection is
    // visited and an assigned track simply reads as 1e18 (identical values, identical first-minimum choice).
    // (structure-of-arrays copy of the tracks so that the distance row vectorises; a key of +inf stands for the
    //  reference's ">= 1e18" entries: gated-out pairs, already assigned tracks and -- never reached in practice --
    //  valid pairs with d2 >= 1e16, which greedy_assignment rejects too)
    std::vector<float> &soa = tr->scratch_t;
    const int MP = (M + 3) & ~3;                       // padded to the SSE width (pads are never available)
    soa.assign((size_t)6 * (MP > 0 ? MP : 4), 0.0f);
    float *tcx = soa.data(), *tcy = tcx + MP, *tsz = tcy + MP, *tcl = tsz + MP, *key = tcl + MP, *avail = key + MP;
    for (int m = 0; m < M; ++m) {
        tcx[m] = tr->tracks[m].ct[0];
        tcy[m] = tr->tracks[m].ct[1];
        tsz[m] = area(tr->tracks[m].bbox);
        tcl[m] = (float)tr->tracks[m].cls;
        avail[m] = 1.0f;
    }
    const float INF = __builtin_inff();
    const __m128 vinf = _mm_set1_ps(INF), vbig = _mm_set1_ps(1e16f), vone = _mm_set1_ps(1.0f);
    for (int i = 0; i < N && M > 0; ++i) {
        const __m128 isz = _mm_set1_ps(area(dets[i].bbox));
        const __m128 px = _mm_set1_ps(dets[i].ct[0] + dets[i].tracking[0]);
        const __m128 py = _mm_set1_ps(dets[i].ct[1] + dets[i].tracking[1]);
        const __m128 cls = _mm_set1_ps((float)dets[i].cls);
        __m128 vmin = vinf;
        for (int m = 0; m < MP; m += 4) {
            const __m128 dx = _mm_sub_ps(_mm_loadu_ps(tcx + m), px), dy = _mm_sub_ps(_mm_loadu_ps(tcy + m), py);
            const __m128 d2 = _mm_add_ps(_mm_mul_ps(dx, dx), _mm_mul_ps(dy, dy));      // (no fused multiply-add)
            __m128 ok = _mm_and_ps(_mm_cmpngt_ps(d2, _mm_loadu_ps(tsz + m)), _mm_cmpngt_ps(d2, isz));
            ok = _mm_and_ps(ok, _mm_cmpeq_ps(cls, _mm_loadu_ps(tcl + m)));
            ok = _mm_and_ps(ok, _mm_cmpeq_ps(_mm_loadu_ps(avail + m), vone));
            ok = _mm_and_ps(ok, _mm_cmplt_ps(d2, vbig));
            const __m128 k = _mm_or_ps(_mm_and_ps(ok, d2), _mm_andnot_ps(ok, vinf));
            _mm_storeu_ps(key + m, k);
            vmin = _mm_min_ps(vmin, k);
        }
        float lanes[4];
        _mm_storeu_ps(lanes, vmin);
        const float bv = fminf(fminf(lanes[0], lanes[1]), fminf(lanes[2], lanes[3]));
        if (bv < INF) {
            int best = 0;
            while (key[best] != bv) ++best;                                            // first minimum, like numpy argmin
            det_match[i] = best;
            trk_match[best] = i;
            avail[best] = 0.0f;
        }
    }
}

// Tracker.step (tracker.py:28-127) on image-space detections; returns the number of tracks or -1
int associate(Tracker *tr, std::vector<ct_track> &dets, const float *public_cts, int n_public, ct_track *out, int cap)
{
    const int N = (int)dets.size(), M = (int)tr->tracks.size();
    std::vector<int> &det_match = tr->scratch_dm, &trk_match = tr->scratch_tm;
    det_match.assign(N, -1);
    trk_match.assign(M, -1);
    // births / carried-over tracks follow the ORDER of the reference's unmatched lists: never-assigned entries in
    // index order, then (Hungarian only) the pairs the solver assigned although they are gated out, in row order
    std::vector<int> late_dets, late_tracks;
    if (!tr->hungarian) {
        match_greedy(tr, dets, det_match, trk_match);
    } else if (N > 0 && M > 0) {
        // float64 matrix exactly as numpy builds it: float32 d2, + 1e18 where gated out, clamped to 1e18 (:44-54)
        std::vector<double> &cost = tr->scratch_cost;
        cost.resize((size_t)N * M);
        for (int i = 0; i < N; ++i) {
            const float px = dets[i].ct[0] + dets[i].tracking[0], py = dets[i].ct[1] + dets[i].tracking[1];
            const float isz = area(dets[i].bbox);
            for (int m = 0; m < M; ++m) {
                const ct_track &t = tr->tracks[m];
                const float dx = t.ct[0] - px, dy = t.ct[1] - py;
                const float d2 = dx * dx + dy * dy;
                const bool invalid = d2 > area(t.bbox) || d2 > isz || dets[i].cls != t.cls;
                double v = (double)d2 + (invalid ? 1e18 : 0.0);
                if (v > 1e18) v = 1e18;
                cost[(size_t)i * M + m] = v;
            }
        }
        const int n = N < M ? N : M;
        std::vector<long> r(n), c(n);
        if (lsap(N, M, cost.data(), r.data(), c.data()) < 0) {
            ct_set_error("ct_tracker_step: the assignment problem is infeasible (non-finite distances)");
            return -1;
        }
        std::vector<char> det_seen(N, 0), trk_seen(M, 0);
        for (int k = 0; k < n; ++k) {
            det_seen[r[k]] = 1;
            trk_seen[c[k]] = 1;
            if (cost[(size_t)r[k] * M + c[k]] > 1e16) {
                late_dets.push_back((int)r[k]);
                late_tracks.push_back((int)c[k]);
            } else {
                det_match[r[k]] = (int)c[k];
                trk_match[c[k]] = (int)r[k];
            }
        }
    }
    std::vector<int> unmatched_dets, unmatched_tracks;
    {
        std::vector<char> late_d(N, 0), late_t(M, 0);
        for (int i : late_dets) late_d[i] = 1;
        for (int m : late_tracks) late_t[m] = 1;
        for (int i = 0; i < N; ++i)
            if (det_match[i] < 0 && !late_d[i]) unmatched_dets.push_back(i);
        for (int m = 0; m < M; ++m)
            if (trk_match[m] < 0 && !late_t[m]) unmatched_tracks.push_back(m);
        unmatched_dets.insert(unmatched_dets.end(), late_dets.begin(), late_dets.end());
        unmatched_tracks.insert(unmatched_tracks.end(), late_tracks.begin(), late_tracks.end());
    }
    std::vector<ct_track> &ret = tr->scratch_ret;
    ret.clear();
    ret.reserve(N + M);
    for (int i = 0; i < N; ++i)
        if (det_match[i] >= 0) {
            ct_track d = dets[i];
            d.tracking_id = tr->tracks[det_match[i]].tracking_id;
            d.age = 1;
            d.active = tr->tracks[det_match[i]].active + 1;
            ret.push_back(d);
        }
    auto birth = [&](int i) {
        if (dets[i].score > tr->new_thresh) {
            ct_track d = dets[i];
            d.tracking_id = ++tr->id_count;
            d.age = 1;
            d.active = 1;
            ret.push_back(d);
        }
    };
    if (tr->public_det && !unmatched_dets.empty()) {
        // tracker.py:83-101: new tracks only where a provided detection is closest to an unmatched detection
        // and within its box size; float32 arithmetic like numpy's
        if (n_public > 0 && !public_cts) {
            ct_set_error("ct_tracker_step: public-detection mode needs the frame's public detections");
            return -1;
        }
        std::vector<float> d3((size_t)N * (n_public > 0 ? n_public : 1));
        std::vector<char> is_unmatched(N, 0);
        for (int i : unmatched_dets) is_unmatched[i] = 1;
        for (int i = 0; i < N; ++i) {
            const float px = dets[i].ct[0] + dets[i].tracking[0], py = dets[i].ct[1] + dets[i].tracking[1];
            for (int j = 0; j < n_public; ++j) {
                const float dx = px - public_cts[2 * j], dy = py - public_cts[2 * j + 1];
                d3[(size_t)i * n_public + j] = is_unmatched[i] ? dx * dx + dy * dy : 1e18f;
            }
        }
        for (int j = 0; j < n_public; ++j) {
            int bi = 0;
            for (int i = 1; i < N; ++i)
                if (d3[(size_t)i * n_public + j] < d3[(size_t)bi * n_public + j]) bi = i;     // first minimum
            if (d3[(size_t)bi * n_public + j] < area(dets[bi].bbox)) {
                for (int q = 0; q < n_public; ++q) d3[(size_t)bi * n_public + q] = 1e18f;
                birth(bi);
            }
        }
    } else {
        for (int i : unmatched_dets) birth(i);
    }
    for (int m : unmatched_tracks)
        if (tr->tracks[m].age < tr->max_age) {
            ct_track t = tr->tracks[m];
            t.age += 1;
            t.active = 0;
            t.row = -1;
            ret.push_back(t);
        }
    tr->tracks.swap(ret);
    const int n = (int)tr->tracks.size();
    // the reference's track list is unbounded (max_age > 0 keeps unmatched tracks): the step always completes;
    // when n > cap only the first cap tracks are copied and the caller fetches the rest with ct_tracker_get_tracks
    const int ncopy = n < cap ? n : cap;
    if (ncopy > 0 && out) memcpy(out, tr->tracks.data(), sizeof(ct_track) * ncopy);
    return n;
}

}  // namespace

extern "C" int ct_transform_points(const float *trans, const float *xy, int n, float *out)
{
    if (!trans || (n > 0 && (!xy || !out)) || n < 0) {
        ct_set_error("ct_transform_points: bad argument");
        return -1;
    }
    for (int i = 0; i < n; ++i) xform(trans, xy[2 * i], xy[2 * i + 1], &out[2 * i], &out[2 * i + 1]);
    return n;
}

extern "C" int ct_tracker_set_mode(void *h, int hungarian, int public_det)
{
    Tracker *tr = static_cast<Tracker *>(h);
    if (!tr) {
        ct_set_error("ct_tracker_set_mode: null tracker");
        return -1;
    }
    tr->hungarian = hungarian ? 1 : 0;
    tr->public_det = public_det ? 1 : 0;
    return 0;
}

extern "C" int ct_tracker_init_tracks(void *h, const ct_track *items, int n)
{
    Tracker *tr = static_cast<Tracker *>(h);
    if (!tr || (n > 0 && !items) || n < 0) {
        ct_set_error("ct_tracker_init_tracks: bad argument");
        return -1;
    }
    for (int i = 0; i < n; ++i)                    // tracker.py:13-26 (ct is given by the caller: the box centre if absent)
        if (items[i].score > tr->new_thresh) {
            ct_track t = items[i];
            t.tracking_id = ++tr->id_count;
            t.age = 1;
            t.active = 1;
            t.row = -1;
            tr->tracks.push_back(t);
        }
    return (int)tr->tracks.size();
}

extern "C" int ct_tracker_step_public(void *h, const float *rows, int K, int F, const ct_row_layout *lay, float out_thresh,
                                      const float *trans_inv, const float *public_cts, int n_public, ct_track *out,
                                      int cap)
{
    Tracker *tr = static_cast<Tracker *>(h);
    if (!tr || !rows || !lay || !trans_inv || !out) {
        ct_set_error("ct_tracker_step: null pointer");
        return -1;
    }
    if (lay->cts < 0 || lay->tracking < 0 || lay->bbox < 0) {
        ct_set_error("ct_tracker_step: rows need cts, tracking and bbox fields");
        return -1;
    }
    rows_to_dets(rows, K, F, lay, out_thresh, trans_inv, tr->scratch_dets);
    return associate(tr, tr->scratch_dets, public_cts, n_public, out, cap);
}

extern "C" int ct_tracker_step(void *h, const float *rows, int K, int F, const ct_row_layout *lay, float out_thresh,
                               const float *trans_inv, ct_track *out, int cap)
{
    return ct_tracker_step_public(h, rows, K, F, lay, out_thresh, trans_inv, nullptr, 0, out, cap);
}

extern "C" int ct_tracker_step_dets(void *h, const ct_track *dets, int n, const float *public_cts, int n_public,
                                    ct_track *out, int cap)
{
    Tracker *tr = static_cast<Tracker *>(h);
    if (!tr || (n > 0 && !dets) || n < 0 || !out) {
        ct_set_error("ct_tracker_step_dets: bad argument");
        return -1;
    }
    tr->scratch_dets.assign(dets, dets + n);
    return associate(tr, tr->scratch_dets, public_cts, n_public, out, cap);
}

extern "C" int ct_tracker_prehm_params(void *h, float pre_thresh, const double *trans_input, int inp_w, int inp_h,
                                       int *params, int cap)
{
    Tracker *tr = static_cast<Tracker *>(h);
    if (!tr || !trans_input || !params) {
        ct_set_error("ct_tracker_prehm_params: null pointer");
        return -1;
    }
    int n = 0;
    for (const ct_track &t : tr->tracks) {
        if (t.score < pre_thresh || t.active == 0) continue;
        float b[4];
        for (int c = 0; c < 2; ++c) {
            // affine_transform: float64 [2,3] @ float32 (x, y, 1) -> float64, stored as float32
            const double x = (double)t.bbox[2 * c], y = (double)t.bbox[2 * c + 1];
            b[2 * c] = (float)((trans_input[0] * x + trans_input[1] * y) + trans_input[2]);
            b[2 * c + 1] = (float)((trans_input[3] * x + trans_input[4] * y) + trans_input[5]);
        }
        const float wmax = (float)(inp_w - 1), hmax = (float)(inp_h - 1);
        b[0] = fminf(fmaxf(b[0], 0.f), wmax);
        b[2] = fminf(fmaxf(b[2], 0.f), wmax);
        b[1] = fminf(fmaxf(b[1], 0.f), hmax);
        b[3] = fminf(fmaxf(b[3], 0.f), hmax);
        const float hh = b[3] - b[1], ww = b[2] - b[0];
        if (!(hh > 0 && ww > 0)) continue;
        double r = gaussian_radius((long)ceil((double)hh), (long)ceil((double)ww));
        int radius = (int)r;
        if (radius < 0) radius = 0;
        const float cx = (b[0] + b[2]) / 2.0f, cy = (b[1] + b[3]) / 2.0f;
        if (n >= cap) {
            ct_set_error("ct_tracker_prehm_params: more than %d prior blobs", cap);
            return -1;
        }
        params[3 * n] = (int)cx;
        params[3 * n + 1] = (int)cy;
        params[3 * n + 2] = radius;
        ++n;
    }
    return n;
}
