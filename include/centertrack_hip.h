/*
 * centertrack_hip.h -- C ABI of libcentertrack_hip.so (gfx950 / MI355X).
 *
 * Drop-in boundary B5 of SURVEY.md section 8(b): these entry points are what the reference's
 * Python hot path binds instead of its PyTorch/cuDNN ops and the un-vendored DCNv2
 * CUDA extension.  Each function cites the reference interface it replaces
 * (paths relative to xingyizhou/CenterTrack).  Conventions:
 *   - every pointer is a DEVICE pointer owned by the caller (torch's allocator); the
 *     library never allocates, frees or synchronises; work is enqueued on `stream`
 *     (a hipStream_t passed as void*; NULL = the default stream); graph-capture safe.
 *   - return 0 on success, non-zero (CT_ERR_*) on error; ct_last_error() describes it.
 *   - activations inside the pipeline are fp32 NHWC "views": base pointer already
 *     offset to the first channel, `ld` = channel pitch in floats (>= C), so a
 *     layer can read/write a channel slice of a wider concat buffer in place
 *     (replaces torch.cat in Root.forward, dla.py:166).  Model inputs/outputs at the
 *     boundary are the reference's NCHW fp32 tensors.
 */
#ifndef CENTERTRACK_HIP_H
#define CENTERTRACK_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CT_OK 0
#define CT_ERR_ARG 1      /* bad shape / alignment / unsupported configuration */
#define CT_ERR_LAUNCH 2   /* hipLaunch failure (message holds hipGetErrorString) */
#define CT_ERR_WORKSPACE 3

/* epilogue flags */
#define CT_RELU 1
#define CT_OUT_NCHW 2      /* write y as NCHW [N,Cout,Ho,Wo] instead of an NHWC view */

/* ABI version of THIS header: bumped whenever a descriptor struct changes layout or a tuning key changes meaning.  A
 * binding compares it with ct_version() of the loaded library before the first descriptor call (centertrack_amd/_lib.py
 * does).  100 = rounds 1-3; 101 = ct_conv_desc.proj_* (fused Tree.project), key "stem_rows", box probes;
 * 102 = ct_decode_desc.sparse (sparse heads); 103 = the ct_calib_* box probes left this header and the library (they are
 * diagnostics of the measuring box: tools/micro/probes.hip -> tools/micro/libct_probes.so; ct_dcn_desc.w_off_winograd).
 * Added under 103 without a layout change: ct_dcn_desc.algo 53264 / 532128 / 63264 / 632128 (persistent DCN launch), the
 * tuning keys "dcn_slots" and "dcn_xcd". */
#define CT_ABI_VERSION 103

const char *ct_last_error(void);
int ct_version(void);                      /* CT_ABI_VERSION the library was built with */
/* Experiment knobs of the launch heuristics (process-global; not part of the reference's
 * interface): "conv_cfg" (-1 auto, 0..5 force a tile shape), "conv_pipe" (0/1 pinned B prefetch),
 * "conv_small_tiles" (below this many workgroups pick the smaller tile), "splitk_target"
 * (workgroups aimed at when splitting K), "dcn_bn" (0 auto, 64, 128), "conv_ks" (-1 auto, -2 never,
 * 0..4 force a K-split-in-workgroup tile), "conv_ks_below" / "conv_ks_waves" (K-split kernel is
 * used below this many 64x64 tiles / sized to reach this many waves), "xcd_remap" (0/1 XCD-aware workgroup order of
 * the wide layers), "heads_order" (0 = tile-major, 1 = head-major, 2 = head-major and
 * XCD-affine [default]: workgroup order of ct_heads_fused), "stem_rows" (0 auto,
 * 8 / 16: rows of 32 pixels per workgroup of ct_stem_forward), "dcn_slots" (resident workgroups of the persistent
 * DCN launches, algo 5xxxx / 6xxxx: 1024 = four per CU), "dcn_xcd" (0/1 [default]: XCD-aware workgroup order of the DCN
 * MAIN launches -- every XCD's L2 sees its own K splits and a band of consecutive pixel tiles). */
int ct_set_tuning(const char *key, int value);

/* ---- weight packing ---------------------------------------------------------------
 * MFMA-ready layout [tap][Cin/16][CoutPad/16][4][16][4] (CoutPad = Cout rounded up to
 * 16, zero filled): for one (tap, 16-channel slab, 16-cout tile) the 64 lanes of a wave
 * read one contiguous 1 KiB block, lane l holding cout (l&15), channels 4*(l>>4)..+3.
 * Source: the reference's OIHW conv weight (nn.Conv2d.weight / DCN.weight,
 * dla.py:38-66,154-172,506-518; base_model.py:24-40).  Cin must be a multiple of 16. */
size_t ct_packed_weight_elems(int Cout, int Cin, int ks);
int ct_pack_conv_weight(const float *w_oihw, float *packed, int Cout, int Cin, int ks, void *stream);
/* Winograd F(2x2,3x3) form of a 3x3 weight: U = G g G^T per (cout, cin), 16 positions in the same fragment
 * order [pos][Cin/16][CoutPad/16][4][16][4] (used by ct_conv2d algo 201 / 202). */
size_t ct_packed_winograd_elems(int Cout, int Cin);
int ct_pack_winograd_weight(const float *w_oihw, float *packed, int Cout, int Cin, void *stream);

/* ---- dense convolution (implicit GEMM on fp32 MFMA) -------------------------------
 * Replaces nn.Conv2d + eval-mode nn.BatchNorm2d (+ residual add) (+ ReLU) of BasicBlock /
 * Root / Tree.project / _make_conv_level (dla.py:38-66,154-172,206-213,293-303), the
 * DCN offset/mask conv (upstream dcn_v2.py DCN.conv_offset_mask) and the head convs
 * (base_model.py:24-65,86-90; sigmoid / depth transform of detector.py:300-308 fused).
 *   y = act( conv(x, w) * scale[c] + shift[c] + res )   ks in {1,3}, stride in {1,2},
 *   pad = ks/2; scale/shift NULL => 1/0.  Channels [sig_lo,sig_hi) get a sigmoid,
 *   channels [dep_lo,dep_hi) get 1/(sigmoid(v)+1e-6)-1 times depth_scale.  */
typedef struct ct_conv_desc {
    const float *x; int N, H, W, Cin, ldx;
    const float *w_packed; int Cout, ks, stride;
    const float *scale; const float *shift;
    const float *res; int ldr;
    float *y; int ldy;
    int flags;
    int sig_lo, sig_hi;
    int dep_lo, dep_hi; float depth_scale;
    float *workspace; size_t workspace_bytes;   /* split-K partials; may be NULL (no split) */
    int split_k;                                /* 0 = choose automatically */
    int algo;                                   /* 0 = heuristic; 1..8 = tile shape 0..7 of the row-tiled
                                                   kernel; 101..105 = K-split-in-workgroup shape 0..4
                                                   (CT_ERR_ARG if the shape cannot run this layer); 201..211 =
                                                   Winograd F(2x2,3x3) with 64px x 64 / 64 x 32 / 128 x 32 /
                                                   128 x 16 (pixels x couts) per workgroup, 205..207: 64 x 32
                                                   / 64 x 32 / 64 x 16 with K split over 2 / 4 / 4 wave groups,
                                                   208..211: 64 x 32 walking 2 / 4 / 5 / 8 cout blocks per workgroup
                                                   on one input transform (Cin == 64)
                                                   (3x3 stride 1, Cin % 64 == 0, NHWC output, needs w_winograd);
                                                   picked per layer by the host-side autotuner */
    const float *w_winograd;                    /* ct_pack_winograd_weight() of the same OIHW weight, or NULL */
    /* optional side output of a 3x3 stride-2 conv (round 3): the 2x2 / stride-2 max-pool of its INPUT, NHWC [N, H/2, W/2,
     * Cin] (pitch pool_ld) -- Tree.downsample (dla.py:207, nn.MaxPool2d(2, 2)) of the tensor tree1.conv1 reads; every
     * input pixel passes through the workgroups' LDS patches anyway, so the pool costs no launch and no extra read.
     * H and W must be even; NULL = off. */
    float *pool_y; int pool_ld;
    /* optional second output of a 3x3 stride-2 conv (round 4): Tree.project of the pooled input -- nn.Conv2d(Cin, Cout, 1,
     * bias=False) + eval-mode BatchNorm, no ReLU, applied to Tree.downsample(x) = max_pool2d(x, 2, 2) (dla.py:196-203,
     * 207, 217-218: the residual tree1 adds) -- computed by the workgroups of this launch from the input patches they
     * stage anyway (the 2x2 window of an output pixel is four taps of its 3x3 stride-2 window): one launch and one read
     * of the input less per DLA level.  proj_w_packed: ct_pack_conv_weight of the [Cout, Cin, 1, 1] weight (the SAME
     * Cout as this conv); proj_y: NHWC [N, H/2, W/2, Cout] view (pitch proj_ldy); proj_scale / proj_shift NULL => 1 / 0.
     * H and W must be even, no split-K.  NULL = off. */
    const float *proj_w_packed; const float *proj_scale; const float *proj_shift;
    float *proj_y; int proj_ldy;
} ct_conv_desc;
int ct_conv2d(const ct_conv_desc *d, void *stream);
size_t ct_conv2d_workspace_bytes(const ct_conv_desc *d);

/* ---- the heads of the network, fused ------------------------------------------------------------------
 * Replaces the per-head nn.Sequential(conv3x3 64 -> 256 + bias, ReLU, conv1x1 256 -> c + bias) of BaseModel
 * (src/lib/model/networks/base_model.py:24-65,86-90) for every listed head, with Detector._sigmoid_output
 * (src/lib/detector.py:300-308) on the channels [sig_lo,sig_hi) / [dep_lo,dep_hi) of `out`, in ONE launch: the
 * 256-channel hidden maps (16.8 MB per head and frame at 512x512) stay inside the workgroups.  x: NHWC feature map
 * (64 channels, pitch ldx); w0_winograd: ct_pack_winograd_weight() of the heads' first-layer weights concatenated
 * along Cout ([nheads*256, 64, 3, 3]); b0 [nheads*256]; w2 [nheads][8][256] = the heads' 1x1 weights, rows >= cout[i]
 * zero; b2 [nheads][8]; head i writes channels coff[i] .. coff[i]+cout[i]-1 of out (NCHW [N, ctot, H, W]).  Heads
 * with more than 8 output channels (hm of an 80-class model, hps, hm_hp) go through ct_conv2d. */
#define CT_MAX_FUSED_HEADS 8
typedef struct ct_heads_desc {
    const float *x; int N, H, W, Cin, ldx;
    const float *w0_winograd; const float *b0;
    int nheads;
    const float *w2; const float *b2;
    int cout[CT_MAX_FUSED_HEADS], coff[CT_MAX_FUSED_HEADS];
    float *out; int ctot;
    int sig_lo, sig_hi, dep_lo, dep_hi; float depth_scale;
} ct_heads_desc;
int ct_heads_fused(const ct_heads_desc *d, void *stream);

/* ---- modulated deformable convolution v2 (3x3, stride 1, pad 1, dil 1, dg 1) -------
 * Replaces DCNv2's _ext.dcn_v2_forward (modulated_deformable_im2col + SGEMM; upstream
 * src/cuda/dcn_v2_cuda.cu, dcn_v2_im2col_cuda.cu) as called from DeformConv.forward,
 * dla.py:513-518, with the following BatchNorm + ReLU fused:
 *   y[p,co] = act( (sum_{k,ci} W[co,ci,k] * m_k(p) * bilinear(x[.,ci], p + tap_k + d_k(p)))
 *                  * scale[co] + shift[co] )      (DCN bias folded into shift by the caller)
 * `om` is the NHWC offset/mask map [N,H,W,>=27] (ld = ldom): channels 2k,2k+1 = (dy,dx)
 * of tap k, channel 18+k = mask AFTER sigmoid (produced by ct_conv2d with sig_lo=18). */
typedef struct ct_dcn_desc {
    const float *x; int N, H, W, Cin, ldx;
    const float *om; int ldom;
    const float *w_packed; int Cout;
    const float *scale; const float *shift;
    float *y; int ldy;
    int flags;                                  /* CT_RELU */
    float *workspace; size_t workspace_bytes;
    int split_k;
    int algo;                                   /* 0 = heuristic; 64 / 128 = 64-pixel tile x 64 / 128 couts per
                                                   workgroup; 3264 / 32128 = 32-pixel tile x 64 / 128 couts;
                                                   43264 / 432128 = the same stepping through 64 instead of 32
                                                   channels per barrier (Cin % 64 == 0);
                                                   53264 / 532128 / 63264 / 632128 = 3264 / 32128 / 43264 / 432128 as a
                                                   PERSISTENT launch: "dcn_slots" resident workgroups, each striding over
                                                   the pixel tiles of one (cout block, K split) column with the gathers,
                                                   weight loads and the next tile's sampling table running across tile
                                                   boundaries; needs fuse_offset != 1 and an even number of step units
                                                   per split; bit-identical to the non-persistent code */
    int fuse_offset;                            /* 1: compute DCN.conv_offset_mask (+ mask sigmoid) inside this launch
                                                   from w_off_packed [27,Cin,3,3 packed] / b_off [27]; `om` is then
                                                   unused (may be NULL); needs Cin % 64 == 0 and a 32-pixel tile.
                                                   2: the same conv K-split over Cin / 64 chunks by the CT_DCN_OFFSETS
                                                   launch (one workgroup per 32-pixel tile and chunk, whatever Cin) into
                                                   om_partial; the main launch sums the chunks (+ bias, mask sigmoid)
                                                   while it builds its sampling table; `om` unused.
                                                   3: om_partial holds ONE map of raw sums (no bias, no sigmoid; channel
                                                   pitch 32) written by any earlier launch -- e.g. ct_conv2d's Winograd
                                                   shapes, which have no sigmoid epilogue; the main launch adds b_off and
                                                   the mask sigmoid (w_off_packed is not read) */
    const float *w_off_packed; const float *b_off;
    /* optional fused IDAUp step (dla.py:543-545) for a `proj` DCN: when up_w != NULL the layer's result goes
     * through ct_upsample_add(result, up_w, up_f, up_skip) into up_y; with split-K the reduction kernel does
     * it directly from the partials (`y` is then never written), otherwise `y` holds the DCN output. */
    const float *up_w; int up_f; const float *up_skip; int up_lds; float *up_y; int up_ldy;
    float *om_partial; size_t om_partial_bytes; /* fuse_offset == 2: [Cin/64][N,H,W,32] floats, 3: [N,H,W,32] = ct_dcn_v2_offsets_bytes(d) */
    const float *w_off_winograd;                /* (ABI 103) fuse_offset == 2 only, optional: conv_offset_mask.weight packed by
                                                   ct_pack_winograd_weight.  The CT_DCN_OFFSETS launch then runs the K-split
                                                   offset/mask convs of ALL such layers of the group as ONE Winograd F(2x2,3x3)
                                                   launch (one workgroup per 64-pixel block and 64-channel chunk, 2.25x fewer
                                                   MFMAs than the direct form); same partial maps, same consumer */
} ct_dcn_desc;
int ct_dcn_v2(const ct_dcn_desc *d, void *stream);
size_t ct_dcn_v2_workspace_bytes(const ct_dcn_desc *d);
size_t ct_dcn_v2_offsets_bytes(const ct_dcn_desc *d);   /* 0 unless fuse_offset == 2 or 3 */
/* Up to 4 INDEPENDENT DeformConv layers in one launch (+ one reduce launch that finishes all of them): the IDAUp /
 * DLAUp tree (dla.py:539-574) has several layers ready at the same time -- every `proj_i` only needs a finished
 * level, `node_i` of different IDAUp stages do not depend on each other -- and at one stream each of them alone
 * cannot fill 256 CUs.  All layers run on the same 32-pixel tile shape (algo 0 / 3264 / 32128 / 43264 / 432128);
 * split_k == 0 gives every workgroup two 32-channel chunks (64 input channels), so the layers of a group finish
 * together whatever their Cin; fuse_offset is per layer; a fused IDAUp step (up_w) always goes through the
 * workspace (y is then not written).  Each layer needs its OWN workspace of ct_dcn_v2_group_workspace_bytes(d)
 * bytes (the layers run concurrently).  `phases`: CT_DCN_MAIN | CT_DCN_FINISH runs both launches back to back;
 * the two may also be issued separately, with different groupings -- the IDAUp step of a `proj` needs its skip
 * tensor (the previous node's output) only in the finishing launch, so the contraction can start before that
 * tensor exists.  Results are bit-identical to ct_dcn_v2 with the same split_k. */
enum { CT_DCN_MAIN = 1,      /* the gather + contraction launch (results or split-K partials) */
       CT_DCN_FINISH = 2,    /* the launch that finishes the layers holding partials: reduction + BN + ReLU (+ IDAUp step) */
       CT_DCN_OFFSETS = 4    /* the K-split offset/mask convs of the layers with fuse_offset == 2 (no-op for the others);
                                must precede CT_DCN_MAIN of the same layers */ };
int ct_dcn_v2_group(const ct_dcn_desc *descs, int n, int phases, void *stream);
size_t ct_dcn_v2_group_workspace_bytes(const ct_dcn_desc *d);
/* The same query with an error code: ct_dcn_v2_group_workspace_bytes returns 0 both for "no workspace needed" and for
 * a descriptor the launch would reject; this one validates the descriptor as a member of a group launch (workspace /
 * output pointers may still be NULL) and returns CT_OK with *workspace_bytes (0 = the layer finishes in its MAIN
 * launch) and *splits (K splits the launch will use), or the launch's error code (see ct_last_error). */
int ct_dcn_v2_group_plan(const ct_dcn_desc *d, size_t *workspace_bytes, int *splits);

/* ---- the three 7x7 stems, fused --------------------------------------------------
 * Replaces DLA.forward's base_layer / pre_img_layer / pre_hm_layer and their sum
 * (dla.py:238-267,305-311): y = sum_s relu(bn_s(conv7x7_s(in_s))), inputs NCHW
 * ([N,3,H,W], [N,3,H,W] or NULL, [N,1,H,W] or NULL), output NHWC [N,H,W,16] (ld = ldy).
 * w_s: OIHW [16,Cin_s,7,7]; scale_s/shift_s: folded BN [16]. */
int ct_stem_forward(const float *x, const float *pre_img, const float *pre_hm, int N, int H, int W,
                    const float *w_x, const float *w_img, const float *w_hm,
                    const float *scale3, const float *shift3,   /* [3][16] each */
                    float *y, int ldy, void *stream);
/* The same with any subset of the three stems (a NULL input skips its stem), accumulated on top of `add` (NHWC, 16
 * channels, pitch ldadd; NULL = 0).  The terms are added in the order x, pre_img, pre_hm, so a launch of {x, pre_img}
 * into a partial map followed by a launch of {pre_hm} with add = that map is bit-identical to ct_stem_forward: the
 * detector runs the first for frame t+1 -- those two terms do not depend on the tracker (dla.py:305-311) -- while
 * the host still associates frame t. */
int ct_stem_forward_parts(const float *x, const float *pre_img, const float *pre_hm, const float *add, int ldadd,
                          int N, int H, int W, const float *w_x, const float *w_img, const float *w_hm,
                          const float *scale3, const float *shift3, float *y, int ldy, void *stream);

/* ---- glue ops ---------------------------------------------------------------------- */
/* nn.MaxPool2d(2,2) of Tree.downsample (dla.py:207-208,216), NHWC views */
int ct_maxpool2x2(const float *x, int N, int H, int W, int C, int ldx, float *y, int ldy, void *stream);
/* IDAUp step `up(proj) + skip` (dla.py:529-532,543-545): depth-wise ConvTranspose2d
 * (kernel 2f, stride f, padding f/2, groups=C, no bias) of x [N,H,W,C] plus skip [N,fH,fW,C]
 * -> y [N,fH,fW,C].  w = the module's weight [C,1,2f,2f] transposed once to [2f,2f,C]. */
int ct_upsample_add(const float *x, int N, int H, int W, int C, int ldx, const float *w, int f,
                    const float *skip, int lds, float *y, int ldy, void *stream);
/* layout converters for the NCHW drop-in ops */
int ct_nchw_to_nhwc(const float *x, int N, int C, int H, int W, float *y, int ldy, void *stream);
int ct_nhwc_to_nchw(const float *x, int N, int C, int H, int W, int ldx, float *y, void *stream);

/* ---- heat-map decode ----------------------------------------------------------------
 * Replaces generic_decode (src/lib/model/decode.py:83-182, non-pose heads) with its
 * helpers _nms / _topk / _tranpose_and_gather_feat (src/lib/model/utils.py:16-87):
 * 3x3 max-pool pseudo-NMS, exact top-K over all classes/pixels (ties: lower class, then
 * lower pixel index first), gathers of every regression head, box assembly, into ONE
 * packed buffer (replaces the 8-14 D2H copies of detector.py:349-350).
 * hm: NCHW [B,C,h,w] post-sigmoid.  heads[i]: NCHW [B,head_ch[i],h,w] or NULL.
 * Head order / packed layout: see ct_decode_layout(). */
enum { CT_HEAD_REG = 0, CT_HEAD_WH, CT_HEAD_TRACKING, CT_HEAD_LTRB, CT_HEAD_LTRB_AMODAL,
       CT_HEAD_DEP, CT_HEAD_ROT, CT_HEAD_DIM, CT_HEAD_AMODEL_OFFSET, CT_HEAD_NUSCENES_ATT,
       CT_HEAD_VELOCITY, CT_NUM_HEADS };
/* Sparse heads (round 5, opt-in; no reference equivalent -- the reference computes every head as a dense map,
 * base_model.py:86-90, and then reads it at K pixels, decode.py:99-180): the regression heads listed here are evaluated
 * at the K winners of every image only, after the selection, and the packed rows are assembled from those values; their
 * entries in ct_decode_desc.heads must be NULL.  feat: DEVICE NHWC view of the 64-channel feature map the heads read
 * (dla.py:631-640), image 0, channel pitch ldf (16-byte aligned, >= 64), images h * w * ldf floats apart.  Per head:
 * head[i] = CT_HEAD_* id; w1 = conv3x3 64 -> 256 weights in ct_pack_conv_weight layout, b1 [256]; w2 = conv1x1 weights
 * [c][256] row-major, b2 [c].  The dep head gets the dense epilogue's transform (1 / (sigmoid(v) + 1e-6) - 1) *
 * depth_scale (detector.py:305-307); zero_tracking != 0 zeroes the tracking output (decode.py:95-96).  Two launches behind
 * the selection: the heads' partial products at the winners, then the rows (+ host copy + end-of-frame flag). */
typedef struct ct_sparse_heads_desc {
    const float *feat; int ldf;
    int nheads;
    int head[CT_NUM_HEADS];
    const float *w1[CT_NUM_HEADS], *b1[CT_NUM_HEADS], *w2[CT_NUM_HEADS], *b2[CT_NUM_HEADS];
    float depth_scale;
    int zero_tracking;
    /* flip_test (detector.py:311-332): flip_B = B when feat holds 2 * B images (image B + b = the mirrored frame of stream
     * b), else 0; flip_mode[i]: 0 = the head is read from image b alone (reg, tracking, ltrb*, rot, ...), 1 = averaged
     * with the mirrored image's value at the mirrored pixel (wh, dep, dim), 2 = the same with the even (x) channels of the
     * mirrored value negated (amodel_offset) */
    int flip_B;
    int flip_mode[CT_NUM_HEADS];
} ct_sparse_heads_desc;
typedef struct ct_decode_desc {
    const float *hm; int B, C, h, w, K;
    const float *heads[CT_NUM_HEADS];
    float *out;                /* [B,K,F] floats, F = ct_decode_row_floats(heads present) */
    int64_t *inds;             /* [B,K] flat pixel indices (may be NULL) */
    void *workspace; size_t workspace_bytes;
    /* floats between consecutive images of hm / of each head map; 0 = densely packed ([B,c,h,w] contiguous).
     * Lets the head maps be channel slices of one wider NCHW tensor (all heads written by one conv launch). */
    size_t hm_batch_stride; size_t head_batch_stride[CT_NUM_HEADS];
    int out_stride;            /* floats between consecutive rows of out; 0 = F (lets a wider row carry the pose fields) */
    /* optional (round 3): the same rows ALSO stored straight into pinned HOST memory (same pitch), and *done_flag (pinned
     * HOST int) set to 1 with system-scope release once every image's rows are there -- the frame graph then needs
     * neither a D2H copy node nor a separate flag kernel behind the decode.  done_counter: DEVICE unsigned, zero before
     * the first launch (the last workgroup resets it).  All three NULL: device rows only. */
    float *host_out; int *done_flag; unsigned *done_counter;
    const ct_sparse_heads_desc *sparse;    /* HOST pointer, NULL = every head is a dense map (ABI 102) */
} ct_decode_desc;
/* row layout: score, cls, xs0, ys0, then for each present field in this order:
 * bbox[4] (if wh|ltrb|ltrb_amodal), bbox_amodal[4] (if ltrb_amodal), tracking[2], dep[1],
 * rot[8], dim[3], amodel_offset[2], nuscenes_att[8], velocity[3] */
int ct_decode_row_floats(const ct_decode_desc *d);
size_t ct_decode_workspace_bytes(const ct_decode_desc *d);
int ct_decode(const ct_decode_desc *d, void *stream);

/* ---- pose branch of the decode ------------------------------------
 * Replaces the `'hps' in output` branch of generic_decode (src/lib/model/decode.py:161-171) with
 * _update_kps_with_hm (decode.py:11-81) and _topk_channel (src/lib/model/utils.py:60-69).  Runs after ct_decode on
 * the same frame: rows / inds are ct_decode's outputs (row pitch row_floats).  The box a snapped joint must lie in
 * (decode.py:45-57) is generic_decode's LOCAL `bboxes` -- the wh box, overridden by the ltrb box (decode.py:123,137),
 * never the ltrb_amodal box (that one only replaces ret['bboxes'], decode.py:159):
 *   box_col >= 4  column of that box in a packed row (rows of a model without an ltrb_amodal head);
 *   box_col == -1 the rows do not hold it (their box columns carry the amodal box, or there is none): it is rebuilt
 *                 from the heads box_ltrb [B,4,h,w], else box_wh [B,2,h,w] (+ box_reg [B,2,h,w], NULL = +0.5), with
 *                 ct_decode's own expressions; all three NULL = the box-less variant (decode.py:60-71): the box is the
 *                 extent of the detection's regressed joints widened by 25 % per side (r and b grow from the already
 *                 widened l and t, as in the reference).
 * hps: NCHW [B,2J,h,w]; hm_hp: NCHW
 * [B,J,h,w] post-sigmoid, densely packed; hp_offset: NCHW [B,2,h,w] (the hp_offset head, else the reg
 * head, else NULL = +0.5).  out: [B,K,2J+1] = refined key points (x0,y0,x1,y1,...) then kps_score
 * (with out_stride it may point into the packed rows themselves: one buffer, one D2H).
 * Peaks of hm_hp with exactly equal scores rank lower pixel first (torch.topk: unspecified). */
typedef struct ct_pose_desc {
    const float *rows; int row_floats, box_col;
    const int64_t *inds;
    int B, h, w, K, num_joints;
    const float *hps, *hm_hp, *hp_offset;
    size_t hps_batch_stride, hm_hp_batch_stride, hp_offset_batch_stride;   /* floats between images; 0 = dense */
    float *out; int out_stride;            /* floats between consecutive rows of out; 0 = 2J+1 */
    void *workspace; size_t workspace_bytes;
    const float *box_wh, *box_reg, *box_ltrb;                              /* box_col == -1, see above */
    size_t box_wh_batch_stride, box_reg_batch_stride, box_ltrb_batch_stride;   /* floats between images; 0 = dense */
} ct_pose_desc;
size_t ct_decode_pose_workspace_bytes(const ct_pose_desc *d);
int ct_decode_pose(const ct_pose_desc *d, void *stream);

/* ---- prior heat-map rendering on device ------------------------------------------------
 * Replaces the numpy Gaussian splatting + full-map H2D of Detector._get_additional_inputs
 * (src/lib/detector.py:254-290; draw_umich_gaussian / gaussian2D, src/lib/utils/image.py:
 * 130-154): out[b,0,y,x] = max over the stream's blobs (cx, cy, r) with |x-cx|<=r, |y-cy|<=r
 * of float32(exp(-(dx^2+dy^2) / (2*((2r+1)/6)^2))) (evaluated in float64 like numpy).
 * params: device int32 [B][cap][3]; counts: device int32 [B]; out: [B(*2),1,H,W] fp32; with
 * also_flipped the W-mirrored map of stream b is written to out[B+b] (flip_test). */
int ct_render_pre_hm(const int *params, const int *counts, int cap, int B, int H, int W, float *out,
                     int also_flipped, void *stream);

/* ---- flip_test on the device ------------------------------------------------------------
 * Replaces Detector._flip_output (src/lib/detector.py:311-332) with flip_tensor / flip_lr / flip_lr_off
 * (src/lib/model/utils.py:28-50) and the `np.concatenate((images, images[:, :, :, ::-1]))` of
 * Detector.pre_process (detector.py:224-226) for B streams: the network runs on [B originals ; B mirrored images].
 * ct_flip_merge: for every listed head  dst[b,c,y,x] = (src[b,c,y,x] + sgn(c) * src[B+b, perm(c), y, w-1-x]) / 2
 * (bit-identical to the reference's fp32 add + divide).  Heads the reference takes from image 0 alone (reg,
 * tracking, ltrb, ltrb_amodal, rot, nuscenes_att, velocity, hp_offset) are not listed: read the first B images.
 * src: NCHW maps of the 2B images, each image's [C,h,w] block dense, src_batch_stride floats between images;
 * dst: dense [B,C,h,w].  flip_idx: HOST int32 [npairs][2] left/right joint pairs (dataset.flip_idx; NULL / 0 when
 * no pose head is listed).  At most 8 heads per call. */
enum { CT_FLIP_AVG = 0,            /* hm, wh, dep, dim */
       CT_FLIP_NEG_EVEN = 1,       /* amodel_offset: mirrored x offsets change sign */
       CT_FLIP_JOINTS = 2,         /* hm_hp [J]: partner joint's map (flip_lr) */
       CT_FLIP_JOINT_OFFSETS = 3   /* hps [2J]: partner joint's (x, y), x negated (flip_lr_off) */ };
typedef struct ct_flip_head {
    const float *src; float *dst;
    size_t src_batch_stride;
    int C, mode;
} ct_flip_head;
int ct_flip_merge(const ct_flip_head *heads, int nheads, const int *flip_idx, int npairs, int B, int h, int w,
                  void *stream);
/* dst[r, x] = src[r, W-1-x] for `rows` rows of W floats (the mirrored input batch: rows = B*3*H) */
int ct_flip_images(const float *src, float *dst, size_t rows, int W, void *stream);

/* ---- host side of a frame (CPU, no device work) ----------------------------------------
 * Replaces generic_post_process (src/lib/utils/post_process.py:21-91), Detector.merge_outputs
 * (src/lib/detector.py:371-377) and Tracker.step with greedy_assignment
 * (src/lib/utils/tracker.py:28-138, private detections) for the packed rows of ct_decode, and
 * the per-track part of Detector._get_additional_inputs (detector.py:254-290). */
typedef struct ct_row_layout {      /* float offsets inside one packed row; -1 = field absent */
    int score, cls, cts, tracking, bbox, amodel_offset;
} ct_row_layout;
typedef struct ct_track {
    float score; int cls;           /* class is 1-based like the reference's item['class'] */
    float ct[2], tracking[2], bbox[4];
    int tracking_id, age, active;
    int row;                        /* source row in the packed decode buffer (-1: carried-over track) */
} ct_track;
void *ct_tracker_create(float new_thresh, int max_age);
void ct_tracker_destroy(void *tracker);
void ct_tracker_reset(void *tracker);
int ct_tracker_num_tracks(void *tracker);
int ct_tracker_id_count(void *tracker);
int ct_tracker_get_tracks(void *tracker, ct_track *out, int cap);
/* rows: HOST pointer [K,F]; trans_inv: float32 [2,3] output-grid -> image affine; returns the number of tracks
 * after the step (the reference's list is unbounded: with max_age > 0 unmatched tracks accumulate) or -1; at most
 * cap of them are written to out -- a caller that gets n > cap grows its buffer and reads ct_tracker_get_tracks */
int ct_tracker_step(void *tracker, const float *rows, int K, int F, const ct_row_layout *lay, float out_thresh,
                    const float *trans_inv, ct_track *out, int cap);
/* Remaining branches of the reference's Tracker (src/lib/utils/tracker.py): hungarian != 0 = optimal assignment
 * instead of greedy (:52-55,63-73, through ct_linear_assignment), public_det != 0 = new tracks only where one of
 * the frame's provided detections supports them (:83-101); ct_tracker_init_tracks = init_track (:13-26: items with
 * score > new_thresh start tracks; score, cls, ct, bbox are read); ct_tracker_step_public = ct_tracker_step with the
 * frame's public detection centres (float32 [n_public,2], image coordinates; NULL / 0 otherwise);
 * ct_tracker_step_dets = Tracker.step on detections that are already in image space (score, cls, ct, tracking, bbox). */
int ct_tracker_set_mode(void *tracker, int hungarian, int public_det);
int ct_tracker_init_tracks(void *tracker, const ct_track *items, int n);
int ct_tracker_step_public(void *tracker, const float *rows, int K, int F, const ct_row_layout *lay, float out_thresh,
                           const float *trans_inv, const float *public_cts, int n_public, ct_track *out, int cap);
int ct_tracker_step_dets(void *tracker, const ct_track *dets, int n, const float *public_cts, int n_public,
                         ct_track *out, int cap);
/* Key points of the pose task (`hps`, src/lib/utils/post_process.py:51-54): n (x, y) pairs on the output grid ->
 * image coordinates with the same float32 [2,3] affine and operation order as the boxes / centres of
 * ct_tracker_step (transform_preds_with_trans, src/lib/utils/image.py:20-26); returns n or -1. */
int ct_transform_points(const float *trans_inv, const float *xy, int n, float *out);
/* (cx, cy, radius) int32 triples of the next frame's prior heat-map; trans_input float64 [2,3] */
int ct_tracker_prehm_params(void *tracker, float pre_thresh, const double *trans_input, int inp_w, int inp_h,
                            int *params, int cap);

/* Minimum-cost assignment of an nr x nc float64 cost matrix (row-major): what the reference's --hungarian branch
 * gets from linear_assignment (src/lib/utils/tracker.py:2,55 -- sklearn's removed module, scipy's
 * linear_sum_assignment in every current environment; same pairs, same tie-breaking).  rows / cols: min(nr, nc)
 * pairs sorted by row; returns their number or -1. */
int ct_linear_assignment(const double *cost, int nr, int nc, int *rows, int *cols);

/* ---- runtime helpers of the per-frame host loop (no reference equivalent: they replace the torch
 * dispatcher on the four runtime calls a frame needs).  ct_graph_begin/end capture everything enqueued on
 * `stream` in between (the launches of one frame) into an executable HIP graph; ct_graph_launch replays it. */
int ct_graph_begin(void *stream);
void *ct_graph_end(void *stream);               /* executable graph handle, NULL on error */
int ct_graph_launch(void *graph_exec, void *stream);
void ct_graph_destroy(void *graph_exec);
int ct_memcpy_async(void *dst, const void *src, size_t bytes, int kind /*0 D2D, 1 H2D, 2 D2H*/, void *stream);
int ct_memset_async(void *dst, int value, size_t bytes, void *stream);      /* DEVICE memory (zero_tracking) */
int ct_stream_synchronize(void *stream);
/* One-thread kernel that stores `value` to *flag (pinned HOST memory, system-scope release): appended to a frame graph
 * behind its last copy node, it lets the host see the end of the frame by polling a cache line instead of waiting in
 * the runtime (ct_frame_loop_wait) -- and independently of work that was enqueued behind the graph for the next frame. */
int ct_signal_host(int *flag, int value, void *stream);
/* ---- the host loop of one frame of B streams, natively (round 3) ----------------------------------------------
 * Replaces, for the steady state of the tracking path, the per-frame host work of Detector.run
 * (src/lib/detector.py:139-165: process -> post_process -> merge_outputs -> tracker.step -> pre_images = images, and
 * the per-track part of _get_additional_inputs, detector.py:254-290, of the NEXT run): no interpreter between the end
 * of frame t and the launch of frame t+1.  The caller (centertrack_amd/detector.py) describes the loop once:
 * B trackers, the packed-row layout, the pinned host blocks the frame graph's copy nodes use (host_rows: destination
 * of its last node; blob_params / blob_counts: source of its first node, NULL without a prior heat-map), one
 * executable graph (ct_graph_end) and one DEVICE frame buffer per rotation slot -- frame t lives in slot t % nslots
 * and is read as pre_img from there by the graph of slot (t+1) % nslots -- and the result buffers.
 *   ct_frame_loop_submit   prior-heat-map blobs from the trackers, the frame into slot `slot` (frame_kind: copy from
 *                          DEVICE / pinned HOST memory, already IN_PLACE, or UPLOADED = wait on the device for the upload
 *                          the previous submit started), graph launch, upload of `next_frame` (pinned HOST) into the
 *                          next slot on a copy stream;
 *   ct_frame_loop_wait     block until the frame in flight finished (rows are in host_rows);
 *   ct_frame_loop_finish   wait + post-process / association of every stream: counts[b] tracks in
 *                          results[b * results_cap ..] (n > results_cap: grow and read ct_tracker_get_tracks);
 *                          CT_ERR_ARG when no frame is in flight (a second association of stale rows would age the
 *                          tracks); a failure on a helper thread is reported with its own message and stream index;
 *   (a submit that fails AFTER its graph launch -- upload / pre-stage of the next frame -- leaves the frame in
 *    flight: drain it with ct_frame_loop_wait or _finish before the next submit)
 *   ct_frame_loop_finish_submit   both, back to back (frame t+1 = the frame uploaded ahead);
 *   ct_frame_loop_upload   the upload alone (when the frame in flight was launched by a finish_submit and the caller
 *                          only now learns the frame after it).
 * trans_input: float64 [B][2][3] network-input affine of each stream's frame (prior heat-map); trans_inv: float32
 * [B][2][3] output-grid -> image affine (post-process).  One frame in flight at a time. */
enum { CT_FRAME_DEVICE = 0, CT_FRAME_HOST = 1, CT_FRAME_IN_PLACE = 2, CT_FRAME_UPLOADED = 3 };
typedef struct ct_prestage_desc {          /* the part of a frame that does not depend on the tracker (round 3): the mirrored
                                              half of a flip_test batch and the x / pre_img terms of the stem
                                              (ct_stem_forward_parts into a partial map the frame graph completes).  The loop
                                              runs it for frame t+1 right behind the graph of frame t -- while the host
                                              still associates frame t -- whenever frame t+1 has been uploaded by then. */
    int enabled;
    int N, H, W;                           /* images of a frame batch (mirrored half included), input size */
    const float *w_x, *w_img, *scale3, *shift3;
    float *partial[3]; int ldp;            /* NHWC [N,H,W,16] partial stem maps, one per rotation slot */
    int flip_B;                            /* > 0: images [0, flip_B) of the slot are mirrored into [flip_B, 2 flip_B) first */
} ct_prestage_desc;
typedef struct ct_frame_loop_desc {
    int B, K, F;
    void *const *trackers;                 /* [B] ct_tracker_create handles */
    ct_row_layout layout;
    float out_thresh, pre_thresh;
    int inp_w, inp_h;
    const float *host_rows;                /* pinned HOST [B,K,F] */
    float *rows_keep;                      /* HOST [B,K,F] or NULL: finish copies the rows here first and works on the copy,
                                              so that results / the decode dict of frame t stay valid while the graph of
                                              frame t+1 (launched by the same call) overwrites host_rows */
    int *blob_params, *blob_counts;        /* pinned HOST [B][blob_cap][3], [B]; NULL = no prior heat-map */
    int blob_cap;
    int nslots;                            /* 1 .. 3 */
    void *graphs[3];
    float *frames[3];
    size_t frame_bytes;                    /* bytes of one frame batch as the caller hands it over ([B,3,H,W] fp32) */
    void *stream;
    ct_track *results; int results_cap;    /* HOST [B][results_cap] */
    int *done_flag;                        /* pinned HOST int the frame graphs set to 1 with their last node (ct_signal_host),
                                              or NULL: wait through the runtime */
    ct_prestage_desc pre;
} ct_frame_loop_desc;
typedef struct ct_frame_step_args {
    int slot, frame_kind;
    const float *frame;                    /* DEVICE / HOST source (frame_kind 0 / 1), else ignored */
    const float *next_frame;               /* pinned HOST frame to upload for the next call, or NULL */
    const double *trans_input;
    const float *trans_inv;
} ct_frame_step_args;
void *ct_frame_loop_create(const ct_frame_loop_desc *d);      /* NULL on error */
void ct_frame_loop_destroy(void *loop);
int ct_frame_loop_submit(void *loop, const ct_frame_step_args *a);
int ct_frame_loop_wait(void *loop);
int ct_frame_loop_finish(void *loop, const ct_frame_step_args *a, int *counts);
int ct_frame_loop_finish_submit(void *loop, const ct_frame_step_args *cur, int *counts, const ct_frame_step_args *next);
int ct_frame_loop_upload(void *loop, int slot, const float *frame);   /* upload a pinned HOST frame into `slot` on the copy stream
                                                                         (the frame a later submit names with CT_FRAME_UPLOADED) */
int ct_frame_loop_prestage(void *loop, int slot); /* run the pre-stage of `slot` now unless it already ran (callers that launch a
                                                     frame graph themselves) */
int ct_frame_loop_pending_slot(void *loop);      /* slot an upload is pending for, or -1 */
int ct_frame_loop_in_flight(void *loop);         /* slot of the frame in flight, or -1 */
void ct_frame_loop_forget_upload(void *loop);    /* drop a pending upload (waits for the copy) */

/* ---- image pre-processing (CPU, like the reference's: it runs in DataLoader worker processes) -----
 * Replaces the cv2.warpAffine + normalise + HWC->CHW (+ flipped copy) of Detector.pre_process
 * (src/lib/detector.py:207-239).  img: HOST u8 [h, w, channels] (row pitch `stride` bytes); trans: float64
 * [2,3] trans_input; out: HOST fp32 [channels (*2 if flip_copy), dst_h, dst_w]. */
int ct_preprocess_image(const uint8_t *img, int h, int w, int stride, int channels, const double *trans,
                        int dst_w, int dst_h, const float *mean, const float *stdv, float *out, int flip_copy);

/* ---- the same pre-processing on the device (no reference equivalent: there the warp
 * is cv2 on the CPU and the fp32 result is uploaded, detector.py:218-226,338).  A raw u8 frame is uploaded
 * instead and warped / normalised / transposed (+ mirrored) by one kernel; results are bit-identical to
 * ct_preprocess_image.  img: DEVICE u8 [h, w, channels] (row pitch `stride` bytes); trans: HOST float64 [2,3]
 * trans_input; lut: DEVICE fp32 [channels][256], the normalisation table ct_preprocess_lut fills on the HOST
 * (lut[c][v] = float(((double)v / 255. - mean[c]) / std[c])); out: DEVICE fp32 [channels, dst_h, dst_w];
 * out_flip: optional DEVICE plane set of the same shape that receives the left-right mirrored copy
 * (flip_test, detector.py:225-226) or NULL. */
int ct_preprocess_lut(const float *mean, const float *stdv, int channels, float *lut_host);
int ct_preprocess_device(const uint8_t *img, int h, int w, int stride, int channels, const double *trans,
                         int dst_w, int dst_h, const float *lut, float *out, float *out_flip, void *stream);

#ifdef __cplusplus
}
#endif
#endif
