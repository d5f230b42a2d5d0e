// Host-side (CPU) part of one frame, native: packed decode rows -> image-space detections
// (inverse affine) -> score cut -> greedy displacement association -> track list, and the
// prior-heat-map parameters of the NEXT frame.  Restates, with identical float32 /
// float64 promotion points so that track IDs are bit-identical:
//   generic_post_process      src/lib/utils/post_process.py:21-91 (2D fields + the ddd centre)
//   Detector.merge_outputs    src/lib/detector.py:371-377
//   Tracker.step / greedy     src/lib/utils/tracker.py:28-138 (private detections, greedy)
//   Detector._get_additional_inputs / _trans_bbox / gaussian_radius
//                             src/lib/detector.py:242-290, src/lib/utils/image.py:105-126
// The Hungarian and public-detection branches stay on the Python path (tracker.py there).
// Compiled with -ffp-contract=off: every float32 product/sum is rounded separately, like
// numpy's element-wise arithmetic.
#include <math.h>
#include <stdint.h>
#include <string.h>

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
    std::vector<ct_track> tracks;
    // per-step scratch (kept to avoid heap traffic on the frame loop)
    std::vector<ct_track> scratch_dets, scratch_ret;
    std::vector<float> scratch_t;
    std::vector<int> scratch_dm, scratch_tm;
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

}  // namespace

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

extern "C" int ct_tracker_step(void *h, const float *rows, int K, int F, const ct_row_layout *lay, float out_thresh,
                               const float *trans_inv, ct_track *out, int cap)
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
    // ---- post-process: stop at the first score < out_thresh, keep score > out_thresh ----
    std::vector<ct_track> &dets = tr->scratch_dets;
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
    const int N = (int)dets.size(), M = (int)tr->tracks.size();
    // ---- association (tracker.py:28-57, 129-138) ----
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
    std::vector<int> &det_match = tr->scratch_dm, &trk_match = tr->scratch_tm;
    det_match.assign(N, -1);
    trk_match.assign(M, -1);
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
    for (int i = 0; i < N; ++i)
        if (det_match[i] < 0 && dets[i].score > tr->new_thresh) {
            ct_track d = dets[i];
            d.tracking_id = ++tr->id_count;
            d.age = 1;
            d.active = 1;
            ret.push_back(d);
        }
    for (int m = 0; m < M; ++m)
        if (trk_match[m] < 0 && tr->tracks[m].age < tr->max_age) {
            ct_track t = tr->tracks[m];
            t.age += 1;
            t.active = 0;
            t.row = -1;
            ret.push_back(t);
        }
    tr->tracks.swap(ret);
    const int n = (int)tr->tracks.size();
    if (n > cap) {
        ct_set_error("ct_tracker_step: %d results exceed the output capacity %d", n, cap);
        return -1;
    }
    if (n > 0) memcpy(out, tr->tracks.data(), sizeof(ct_track) * n);
    return n;
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
