// Pose branch of generic_decode (src/lib/model/decode.py:161-171 with _update_kps_with_hm :11-81 and
// utils._topk_channel, src/lib/model/utils.py:60-69) -- SURVEY.md section 8f rank 3.
//
//   1. peaks of the joint heat-maps: hm_hp [B,J,h,w] is decoded as B*J single-class images by the SAME
//      two decode kernels as the centre heat-map (3x3 NMS + exact top-K; csrc/decode.hip) -> per (image,
//      joint) K peaks (score, x0, y0, flat index), ties: lower pixel first;
//   2. pose_match_kernel, one workgroup per (image, joint): the K peaks (+ hp_offset / reg sub-pixel
//      offset, peaks <= 0.2 dropped) sit in LDS; thread k regresses joint j of detection k
//      (hps[b, 2j..2j+1, ind_k] + centre), scans the peaks for the nearest one (fp32 sqrt(dx^2+dy^2),
//      contraction off, first minimum like torch.min) and keeps the regressed joint when that peak is weak
//      or outside the detection's box -- the wh / ltrb box from the packed row, or rebuilt from the heads when the
//      row carries the ltrb_amodal box instead, or (no box head: decode.py:60-71) the extent of the detection's
//      regressed joints widened by 25 %;
//   3. pose_score_kernel: kps_score[b,k] = score * mean_j(peak score or, where not snapped, score)
//      (joints summed in order, then / J, like torch.mean over a strided dim).
// Integer / compare work with B*J*K*K distance evaluations (1.7e5 per image): latency-bound, no MFMA.
#include <stdint.h>

#include "ct_common.h"

namespace {

constexpr float POSE_THRESH = 0.2f;      // decode.py:16

struct PoseArgs {
    const float *rows;        // [B,K,F] packed detections
    const long long *inds;    // [B,K]
    const float *hps;         // [B,2J,h,w]
    const float *off;         // [B,2,h,w] or nullptr
    const float *jrows;       // [B*J,K,4]: score, cls, x0, y0 of the joint peaks
    const long long *jinds;   // [B*J,K]
    float *sc;                // [B,J,K] per-joint score term
    float *out;               // [B,K,2J+1]
    size_t hps_bs, off_bs;
    int B, K, J, F, HW, box_col, OW;
    const float *bwh, *breg, *bltrb;   // box_col < 0: gate box from the heads (all null: box-less variant)
    size_t bwh_bs, breg_bs, bltrb_bs;
};

__global__ __launch_bounds__(128) void pose_match_kernel(PoseArgs a)
{
#pragma clang fp contract(off)
    extern __shared__ float pk[];                 // [3][K]: x, y, score of the joint's peaks
    float *px = pk, *py = pk + a.K, *ps = pk + 2 * a.K;
    const int bj = blockIdx.x, b = bj / a.J, j = bj - b * a.J;
    for (int c = threadIdx.x; c < a.K; c += blockDim.x) {
        const float *jr = a.jrows + ((size_t)bj * a.K + c) * 4;
        const float s = jr[0];
        float x = jr[2], y = jr[3];
        if (a.off) {
            const long long ind = a.jinds[(size_t)bj * a.K + c];
            x = x + a.off[(size_t)b * a.off_bs + ind];
            y = y + a.off[(size_t)b * a.off_bs + a.HW + ind];
        } else {
            x = x + 0.5f; y = y + 0.5f;
        }
        const bool strong = s > POSE_THRESH;
        ps[c] = strong ? s : -1.0f;
        px[c] = strong ? x : -10000.0f;
        py[c] = strong ? y : -10000.0f;
    }
    __syncthreads();
    const int OW = a.OW;
    for (int k = threadIdx.x; k < a.K; k += blockDim.x) {
        const float *row = a.rows + ((size_t)b * a.K + k) * a.F;
        const long long ind = a.inds[(size_t)b * a.K + k];
        const float rx = a.hps[(size_t)b * a.hps_bs + (size_t)(2 * j) * a.HW + ind] + row[2];
        const float ry = a.hps[(size_t)b * a.hps_bs + (size_t)(2 * j + 1) * a.HW + ind] + row[3];
        float best = __builtin_inff();
        int bi = 0;
        for (int c = 0; c < a.K; ++c) {
            const float dx = rx - px[c], dy = ry - py[c];
            const float d = __fsqrt_rn(dx * dx + dy * dy);
            if (d < best) { best = d; bi = c; }
        }
        const float hs = ps[bi], hx = px[bi], hy = py[bi];
        float bl, bt, br, bb;
        if (a.box_col >= 0) {
            bl = row[a.box_col]; bt = row[a.box_col + 1]; br = row[a.box_col + 2]; bb = row[a.box_col + 3];
        } else if (a.bltrb) {                                         // decode.py:131-139
            const float *q = a.bltrb + (size_t)b * a.bltrb_bs + ind;
            bl = row[2] + q[0]; bt = row[3] + q[(size_t)a.HW]; br = row[2] + q[(size_t)2 * a.HW]; bb = row[3] + q[(size_t)3 * a.HW];
        } else if (a.bwh) {                                           // decode.py:102-128
            float xs = row[2] + 0.5f, ys = row[3] + 0.5f;
            if (a.breg) {
                xs = row[2] + a.breg[(size_t)b * a.breg_bs + ind];
                ys = row[3] + a.breg[(size_t)b * a.breg_bs + a.HW + ind];
            }
            float ww = a.bwh[(size_t)b * a.bwh_bs + ind], hh = a.bwh[(size_t)b * a.bwh_bs + a.HW + ind];
            if (ww < 0.f) ww = 0.f;
            if (hh < 0.f) hh = 0.f;
            bl = xs - ww / 2; bt = ys - hh / 2; br = xs + ww / 2; bb = ys + hh / 2;
        } else {                                                      // decode.py:60-71: extent of the regressed joints
            bl = bt = __builtin_inff();
            br = bb = -__builtin_inff();
            for (int q = 0; q < a.J; ++q) {
                const float qx = a.hps[(size_t)b * a.hps_bs + (size_t)(2 * q) * a.HW + ind] + row[2];
                const float qy = a.hps[(size_t)b * a.hps_bs + (size_t)(2 * q + 1) * a.HW + ind] + row[3];
                bl = fminf(bl, qx); br = fmaxf(br, qx);
                bt = fminf(bt, qy); bb = fmaxf(bb, qy);
            }
            const float margin = 0.25f;
            bl = bl - (br - bl) * margin;
            br = br + (br - bl) * margin;                             // (from the widened l, like the reference)
            bt = bt - (bb - bt) * margin;
            bb = bb + (bb - bt) * margin;
        }
        const bool keep_reg = hs < POSE_THRESH || hx < bl || hx > br || hy < bt || hy > bb;
        float *o = a.out + ((size_t)b * a.K + k) * OW;
        o[2 * j] = keep_reg ? rx : hx;
        o[2 * j + 1] = keep_reg ? ry : hy;
        a.sc[((size_t)b * a.J + j) * a.K + k] = keep_reg ? row[0] : hs;
    }
}

__global__ __launch_bounds__(128) void pose_score_kernel(PoseArgs a)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.B * a.K) return;
    const int b = i / a.K, k = i - b * a.K;
    float sum = 0.0f;
    for (int j = 0; j < a.J; ++j) sum += a.sc[((size_t)b * a.J + j) * a.K + k];
    a.out[(size_t)i * a.OW + 2 * a.J] = a.rows[(size_t)i * a.F] * (sum / (float)a.J);
}

// the joint heat-maps as B*J single-class images for ct_decode
void joint_desc(const ct_pose_desc *d, ct_decode_desc *jd)
{
    *jd = ct_decode_desc();
    jd->hm = d->hm_hp; jd->B = d->B * d->num_joints; jd->C = 1; jd->h = d->h; jd->w = d->w; jd->K = d->K;
}

size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

int check(const ct_pose_desc *d, const char *who)
{
    if (!d || !d->rows || !d->inds || !d->hps || !d->hm_hp) CT_FAIL_ARG("%s: null pointer", who);
    if (d->B <= 0 || d->h <= 0 || d->w <= 0 || d->K <= 0 || d->num_joints <= 0) CT_FAIL_ARG("%s: bad shape", who);
    if (d->box_col != -1 && (d->box_col < 4 || d->box_col + 4 > d->row_floats))
        CT_FAIL_ARG("%s: box_col=%d is neither -1 (box from the heads / box-less) nor a box column of a %d-float row", who,
                    d->box_col, d->row_floats);
    if (d->box_col == -1 && d->box_reg && !d->box_wh) CT_FAIL_ARG("%s: box_reg given without box_wh", who);
    if (d->hm_hp_batch_stride && d->hm_hp_batch_stride != (size_t)d->num_joints * d->h * d->w)
        CT_FAIL_ARG("%s: hm_hp must be densely packed [B,J,h,w]", who);
    return CT_OK;
}

}  // namespace

extern "C" size_t ct_decode_pose_workspace_bytes(const ct_pose_desc *d)
{
    if (check(d, "ct_decode_pose_workspace_bytes") != CT_OK) return 0;
    ct_decode_desc jd;
    joint_desc(d, &jd);
    const size_t inner = ct_decode_workspace_bytes(&jd);
    if (!inner) return 0;
    const size_t BJK = (size_t)d->B * d->num_joints * d->K;
    return align256(BJK * 4 * sizeof(float)) + align256(BJK * sizeof(int64_t)) + align256(BJK * sizeof(float)) +
           align256(inner);
}

extern "C" int ct_decode_pose(const ct_pose_desc *d, void *stream)
{
    int rc = check(d, "ct_decode_pose");
    if (rc != CT_OK) return rc;
    if (!d->out) CT_FAIL_ARG("ct_decode_pose: null output");
    const size_t need = ct_decode_pose_workspace_bytes(d);
    if (!need) return CT_ERR_ARG;
    if (!d->workspace || d->workspace_bytes < need) {
        ct_set_error("ct_decode_pose: needs %zu workspace bytes, got %zu", need, d->workspace_bytes);
        return CT_ERR_WORKSPACE;
    }
    const size_t BJK = (size_t)d->B * d->num_joints * d->K;
    char *ws = (char *)d->workspace;
    float *jrows = (float *)ws;            ws += align256(BJK * 4 * sizeof(float));
    int64_t *jinds = (int64_t *)ws;        ws += align256(BJK * sizeof(int64_t));
    float *sc = (float *)ws;               ws += align256(BJK * sizeof(float));
    ct_decode_desc jd;
    joint_desc(d, &jd);
    jd.out = jrows; jd.inds = jinds;
    jd.workspace = ws; jd.workspace_bytes = d->workspace_bytes - (size_t)(ws - (char *)d->workspace);
    rc = ct_decode(&jd, stream);           // 1. K peaks of every joint heat-map
    if (rc != CT_OK) return rc;
    PoseArgs a;
    const int HW = d->h * d->w;
    a.rows = d->rows; a.inds = (const long long *)d->inds; a.hps = d->hps; a.off = d->hp_offset;
    a.jrows = jrows; a.jinds = (const long long *)jinds; a.sc = sc; a.out = d->out;
    a.hps_bs = d->hps_batch_stride ? d->hps_batch_stride : (size_t)2 * d->num_joints * HW;
    a.off_bs = d->hp_offset_batch_stride ? d->hp_offset_batch_stride : (size_t)2 * HW;
    a.B = d->B; a.K = d->K; a.J = d->num_joints; a.F = d->row_floats; a.HW = HW; a.box_col = d->box_col;
    a.bwh = d->box_col < 0 ? d->box_wh : nullptr;
    a.breg = d->box_col < 0 ? d->box_reg : nullptr;
    a.bltrb = d->box_col < 0 ? d->box_ltrb : nullptr;
    a.bwh_bs = d->box_wh_batch_stride ? d->box_wh_batch_stride : (size_t)2 * HW;
    a.breg_bs = d->box_reg_batch_stride ? d->box_reg_batch_stride : (size_t)2 * HW;
    a.bltrb_bs = d->box_ltrb_batch_stride ? d->box_ltrb_batch_stride : (size_t)4 * HW;
    a.OW = d->out_stride ? d->out_stride : 2 * d->num_joints + 1;
    if (a.OW < 2 * d->num_joints + 1) CT_FAIL_ARG("ct_decode_pose: out_stride %d too small", d->out_stride);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(pose_match_kernel, dim3((unsigned)(d->B * d->num_joints)), dim3(128),
                       3 * (size_t)d->K * sizeof(float), s, a);
    CT_CHECK_LAUNCH("ct_decode_pose(match)");
    hipLaunchKernelGGL(pose_score_kernel, dim3((unsigned)ct_cdiv(d->B * d->K, 128)), dim3(128), 0, s, a);
    CT_CHECK_LAUNCH("ct_decode_pose(score)");
    return CT_OK;
}
