"""Drop-in for ``generic_decode`` (SURVEY.md boundary B3).

``generic_decode(output, K=100, opt=None) -> dict`` follows src/lib/model/decode.py:83-182
(incl. the pose branch :161-171 = ``_update_kps_with_hm`` + ``_topk_channel``, through
``ct_decode_pose``) with its helpers ``_nms`` / ``_topk`` / ``_tranpose_and_gather_feat``
(src/lib/model/utils.py:16-87): same keys, shapes and dtypes (``clses`` is float32,
decode.py:100), ``output['tracking'] *= 0`` in place under ``opt.zero_tracking``, ``{}``
when there is no ``'hm'``.  Underneath it is ONE ``ct_decode`` call (3x3 max NMS + exact
top-K over classes x pixels + all head gathers + box assembly into a packed [B,K,F]
buffer); the dict entries are views of that buffer.

Exact-tie order (unspecified by ``torch.topk``): lower class first, then lower pixel index.
CUDA fp32 NCHW tensors only; no CPU fallback.
"""
import torch

from . import _lib, ops

_POSE_HEADS = ('hps', 'hm_hp', 'hp_offset')        # multi_pose branch, decode.py:161-171


def generic_decode(output, K=100, opt=None):
    if 'hm' not in output:
        return {}
    if opt is not None and getattr(opt, 'zero_tracking', False):
        output['tracking'] *= 0
    hm = output['hm']
    if not hm.is_cuda or hm.dtype != torch.float32:
        raise _lib.CTError('generic_decode runs on an MI355X (cuda fp32 tensors); no CPU fallback')
    def planes(t):     # ct_decode wants every image's [c,h,w] block contiguous (images may be strided)
        ok = t.stride(3) == 1 and t.stride(2) == t.shape[3] and (t.shape[1] == 1 or t.stride(1) == t.shape[2] * t.shape[3])
        return t if ok else t.contiguous()
    heads = {k: planes(v) for k, v in output.items() if k in _lib.HEAD_INDEX}
    if 'hps' in output:
        heads.update({k: (output[k].contiguous() if k == 'hm_hp' else planes(output[k])) for k in _POSE_HEADS
                      if k in output})
    dec = ops.Decoder(planes(hm), heads, K)
    packed = dec.run()
    ret = dec.unpack(packed)
    # (decode.py:159: with an ltrb_amodal head the kernel already writes the amodal box into 'bboxes')
    if output.get('pre_inds', None) is not None:       # decode.py:173-180
        width = hm.shape[3]
        pre_inds = output['pre_inds']
        pre_ys = (pre_inds / width).int().float()
        pre_xs = (pre_inds % width).int().float()
        ret['pre_cts'] = torch.cat([pre_xs.unsqueeze(2), pre_ys.unsqueeze(2)], dim=2)
    return ret
