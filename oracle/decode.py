"""Oracle: heat-map decode on CPU (torch).  TEST INFRASTRUCTURE ONLY.

Restates ``src/lib/model/utils.py`` (_nms :52-58, _topk :71-87, _gather_feat
:16-20, _tranpose_and_gather_feat :22-26) and ``src/lib/model/decode.py``
(generic_decode :83-182 incl. the pose branch :161-171 with _update_kps_with_hm
:11-81 and utils._topk_channel :60-69).  Pinned against the imported reference by
tests/golden/make_golden.py (decode.npz; the pose case is 'pose').
"""
import torch
import torch.nn.functional as F

REGRESSION_HEADS = ['tracking', 'dep', 'rot', 'dim', 'amodel_offset',
                    'nuscenes_att', 'velocity']          # decode.py:142-143


def nms(heat, kernel=3):
    """utils.py:52-58 : keep = (maxpool3x3(heat) == heat)"""
    hmax = F.max_pool2d(heat, (kernel, kernel), stride=1, padding=(kernel - 1) // 2)
    return heat * (hmax == heat).float()


def gather_feat(feat, ind):
    """utils.py:16-20 : feat [B,N,F], ind [B,K] -> [B,K,F]"""
    ind = ind.unsqueeze(2).expand(ind.size(0), ind.size(1), feat.size(2))
    return feat.gather(1, ind)


def transpose_and_gather_feat(feat, ind):
    """utils.py:22-26 : NCHW -> [B,HW,C] -> gather rows"""
    feat = feat.permute(0, 2, 3, 1).contiguous()
    feat = feat.view(feat.size(0), -1, feat.size(3))
    return gather_feat(feat, ind)


def topk(scores, K=100):
    """utils.py:71-87 : per-class top-K over H*W, then top-K over C*K."""
    batch, cat, height, width = scores.size()
    topk_scores, topk_inds = torch.topk(scores.view(batch, cat, -1), K)
    topk_inds = topk_inds % (height * width)
    topk_ys = (topk_inds / width).int().float()       # true division + truncation (:77)
    topk_xs = (topk_inds % width).int().float()
    topk_score, topk_ind = torch.topk(topk_scores.view(batch, -1), K)
    topk_clses = (topk_ind / K).int()
    topk_inds = gather_feat(topk_inds.view(batch, -1, 1), topk_ind).view(batch, K)
    topk_ys = gather_feat(topk_ys.view(batch, -1, 1), topk_ind).view(batch, K)
    topk_xs = gather_feat(topk_xs.view(batch, -1, 1), topk_ind).view(batch, K)
    return topk_score, topk_inds, topk_clses, topk_ys, topk_xs


def topk_channel(scores, K=100):
    """utils.py:60-69 : top-K of every channel separately -> scores, flat inds, ys, xs, each [B,C,K]"""
    batch, cat, height, width = scores.size()
    top_scores, top_inds = torch.topk(scores.view(batch, cat, -1), K)
    top_inds = top_inds % (height * width)
    return top_scores, top_inds, (top_inds / width).int().float(), (top_inds % width).int().float()


POSE_THRESH = 0.2          # decode.py:16


def refine_keypoints(kps, output, K, bboxes, scores):
    """decode.py:11-81 (_update_kps_with_hm): every regressed joint
    (kps [B,K,2J], centre + hps offset) snaps to the nearest peak of its joint heat-map ``hm_hp`` (top-K
    peaks per joint, + hp_offset / reg sub-pixel offset, peaks <= 0.2 discarded) unless that peak is weak or
    falls outside the detection's box; kps_score = score * mean_j(peak score or, where not snapped, score).
    ``bboxes`` None (no wh / ltrb head, decode.py:60-71): the box is the extent of the detection's regressed joints
    widened by 25 % per side -- r and b are widened from the ALREADY widened l and t, as the reference's in-place
    sequence does.  Written per (image, joint) instead of the reference's 5-D broadcast; same fp32 operations."""
    batch, J = kps.shape[0], kps.shape[2] // 2
    heat = nms(output['hm_hp'])
    p_score, p_inds, p_ys, p_xs = topk_channel(heat, K=K)                      # [B,J,K]
    off = output.get('hp_offset', output.get('reg'))
    if off is not None:
        o = transpose_and_gather_feat(off, p_inds.view(batch, -1)).view(batch, J, K, 2)
        p_xs, p_ys = p_xs + o[..., 0], p_ys + o[..., 1]
    else:
        p_xs, p_ys = p_xs + 0.5, p_ys + 0.5
    strong = p_score > POSE_THRESH
    p_score = torch.where(strong, p_score, torch.full_like(p_score, -1.0))
    p_xs = torch.where(strong, p_xs, torch.full_like(p_xs, -10000.0))
    p_ys = torch.where(strong, p_ys, torch.full_like(p_ys, -10000.0))
    new_kps = kps.clone()
    joint_score = torch.empty((batch, J, K), dtype=kps.dtype)
    for b in range(batch):
        if bboxes is not None:
            l, t, r, bt = bboxes[b, :, 0], bboxes[b, :, 1], bboxes[b, :, 2], bboxes[b, :, 3]
        else:                                                                    # decode.py:60-71
            xs_, ys_ = kps[b, :, 0::2], kps[b, :, 1::2]                          # [K, J]
            l, r = xs_.min(dim=1)[0], xs_.max(dim=1)[0]
            t, bt = ys_.min(dim=1)[0], ys_.max(dim=1)[0]
            margin = 0.25
            l = l - (r - l) * margin
            r = r + (r - l) * margin
            t = t - (bt - t) * margin
            bt = bt + (bt - t) * margin
        for j in range(J):
            rx, ry = kps[b, :, 2 * j], kps[b, :, 2 * j + 1]                      # regressed joint of every detection
            dx = rx[:, None] - p_xs[b, j][None, :]
            dy = ry[:, None] - p_ys[b, j][None, :]
            dist = (dx ** 2 + dy ** 2) ** 0.5                                    # [det, peak]
            near = dist.argmin(dim=1)
            hs, hx, hy = p_score[b, j][near], p_xs[b, j][near], p_ys[b, j][near]
            keep_reg = (hs < POSE_THRESH) | (hx < l) | (hx > r) | (hy < t) | (hy > bt)
            new_kps[b, :, 2 * j] = torch.where(keep_reg, rx, hx)
            new_kps[b, :, 2 * j + 1] = torch.where(keep_reg, ry, hy)
            joint_score[b, j] = torch.where(keep_reg, scores[b], hs)
    return new_kps, scores * joint_score.mean(dim=1)


def generic_decode(output, K=100, zero_tracking=False, return_inds=False):
    """decode.py:83-182.  ``output`` maps head name ->
    [B,c,h,w]; 'hm' must already be sigmoid-ed (detector.py:300-308)."""
    if 'hm' not in output:
        return {}
    if zero_tracking:
        output['tracking'] *= 0
    heat = output['hm']
    batch, cat, height, width = heat.size()
    heat = nms(heat)
    scores, inds, clses, ys0, xs0 = topk(heat, K=K)
    clses = clses.view(batch, K)
    scores = scores.view(batch, K)
    cts = torch.cat([xs0.unsqueeze(2), ys0.unsqueeze(2)], dim=2)
    ret = {'scores': scores, 'clses': clses.float(), 'xs': xs0, 'ys': ys0, 'cts': cts}
    bboxes = None                       # (the box the pose branch gates on: wh / ltrb, not ltrb_amodal; decode.py:97,123,137)
    if 'reg' in output:
        reg = transpose_and_gather_feat(output['reg'], inds).view(batch, K, 2)
        xs = xs0.view(batch, K, 1) + reg[:, :, 0:1]
        ys = ys0.view(batch, K, 1) + reg[:, :, 1:2]
    else:
        xs = xs0.view(batch, K, 1) + 0.5
        ys = ys0.view(batch, K, 1) + 0.5
    if 'wh' in output:
        wh = transpose_and_gather_feat(output['wh'], inds).view(batch, K, 2)
        wh[wh < 0] = 0                                                     # decode.py:117
        bboxes = torch.cat([xs - wh[..., 0:1] / 2, ys - wh[..., 1:2] / 2,
                            xs + wh[..., 0:1] / 2, ys + wh[..., 1:2] / 2], dim=2)
        ret['bboxes'] = bboxes
    if 'ltrb' in output:
        ltrb = transpose_and_gather_feat(output['ltrb'], inds).view(batch, K, 4)
        bboxes = torch.cat([xs0.view(batch, K, 1) + ltrb[..., 0:1],
                            ys0.view(batch, K, 1) + ltrb[..., 1:2],
                            xs0.view(batch, K, 1) + ltrb[..., 2:3],
                            ys0.view(batch, K, 1) + ltrb[..., 3:4]], dim=2)
        ret['bboxes'] = bboxes
    for head in REGRESSION_HEADS:
        if head in output:
            ret[head] = transpose_and_gather_feat(output[head], inds).view(batch, K, -1)
    if 'ltrb_amodal' in output:
        la = transpose_and_gather_feat(output['ltrb_amodal'], inds).view(batch, K, 4)
        amodal = torch.cat([xs0.view(batch, K, 1) + la[..., 0:1],
                            ys0.view(batch, K, 1) + la[..., 1:2],
                            xs0.view(batch, K, 1) + la[..., 2:3],
                            ys0.view(batch, K, 1) + la[..., 3:4]], dim=2)
        ret['bboxes_amodal'] = amodal
        ret['bboxes'] = amodal                                             # decode.py:159
    if 'hps' in output:                                                        # decode.py:161-171
        J = output['hps'].shape[1] // 2
        kps = transpose_and_gather_feat(output['hps'], inds).view(batch, K, 2 * J).clone()
        kps[..., 0::2] += xs0.view(batch, K, 1)
        kps[..., 1::2] += ys0.view(batch, K, 1)
        if 'hm_hp' in output:
            ret['hps'], ret['kps_score'] = refine_keypoints(kps, output, K, bboxes, scores)
        else:
            ret['hps'], ret['kps_score'] = kps, kps                           # decode.py:80-81
    if output.get('pre_inds', None) is not None:
        pre_inds = output['pre_inds']
        pre_ys = (pre_inds / width).int().float()
        pre_xs = (pre_inds % width).int().float()
        ret['pre_cts'] = torch.cat([pre_xs.unsqueeze(2), pre_ys.unsqueeze(2)], dim=2)
    if return_inds:
        ret['inds'] = inds
    return ret
