"""ctypes binding of libcentertrack_hip.so (the C ABI of include/centertrack_hip.h).

No torch types cross the boundary: tensors are passed as raw device pointers
(``tensor.data_ptr()``) plus sizes, and the HIP stream as ``void*``.  There is NO CPU
fallback: if the shared library is missing or a call fails this raises."""
import ctypes
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('CENTERTRACK_LIB') or os.path.join(_PKG, 'libcentertrack_hip.so')   # (override: debugging builds)

c_float_p = ctypes.c_void_p      # device pointers are opaque to the host
NUM_HEADS = 11
(HEAD_REG, HEAD_WH, HEAD_TRACKING, HEAD_LTRB, HEAD_LTRB_AMODAL, HEAD_DEP, HEAD_ROT, HEAD_DIM,
 HEAD_AMODEL_OFFSET, HEAD_NUSCENES_ATT, HEAD_VELOCITY) = range(NUM_HEADS)
HEAD_INDEX = {'reg': HEAD_REG, 'wh': HEAD_WH, 'tracking': HEAD_TRACKING, 'ltrb': HEAD_LTRB,
              'ltrb_amodal': HEAD_LTRB_AMODAL, 'dep': HEAD_DEP, 'rot': HEAD_ROT, 'dim': HEAD_DIM,
              'amodel_offset': HEAD_AMODEL_OFFSET, 'nuscenes_att': HEAD_NUSCENES_ATT,
              'velocity': HEAD_VELOCITY}
HEAD_CH = {'reg': 2, 'wh': 2, 'tracking': 2, 'ltrb': 4, 'ltrb_amodal': 4, 'dep': 1, 'rot': 8, 'dim': 3,
           'amodel_offset': 2, 'nuscenes_att': 8, 'velocity': 3}

CT_RELU = 1
CT_OUT_NCHW = 2
CT_DCN_MAIN, CT_DCN_FINISH, CT_DCN_OFFSETS = 1, 2, 4
CT_OK, CT_ERR_ARG, CT_ERR_LAUNCH, CT_ERR_WORKSPACE = 0, 1, 2, 3


class ConvDesc(ctypes.Structure):
    _fields_ = [('x', ctypes.c_void_p), ('N', ctypes.c_int), ('H', ctypes.c_int), ('W', ctypes.c_int),
                ('Cin', ctypes.c_int), ('ldx', ctypes.c_int),
                ('w_packed', ctypes.c_void_p), ('Cout', ctypes.c_int), ('ks', ctypes.c_int),
                ('stride', ctypes.c_int),
                ('scale', ctypes.c_void_p), ('shift', ctypes.c_void_p),
                ('res', ctypes.c_void_p), ('ldr', ctypes.c_int),
                ('y', ctypes.c_void_p), ('ldy', ctypes.c_int),
                ('flags', ctypes.c_int),
                ('sig_lo', ctypes.c_int), ('sig_hi', ctypes.c_int),
                ('dep_lo', ctypes.c_int), ('dep_hi', ctypes.c_int), ('depth_scale', ctypes.c_float),
                ('workspace', ctypes.c_void_p), ('workspace_bytes', ctypes.c_size_t),
                ('split_k', ctypes.c_int), ('algo', ctypes.c_int), ('w_winograd', ctypes.c_void_p),
                ('pool_y', ctypes.c_void_p), ('pool_ld', ctypes.c_int),
                ('proj_w_packed', ctypes.c_void_p), ('proj_scale', ctypes.c_void_p), ('proj_shift', ctypes.c_void_p),
                ('proj_y', ctypes.c_void_p), ('proj_ldy', ctypes.c_int)]


class DcnDesc(ctypes.Structure):
    _fields_ = [('x', ctypes.c_void_p), ('N', ctypes.c_int), ('H', ctypes.c_int), ('W', ctypes.c_int),
                ('Cin', ctypes.c_int), ('ldx', ctypes.c_int),
                ('om', ctypes.c_void_p), ('ldom', ctypes.c_int),
                ('w_packed', ctypes.c_void_p), ('Cout', ctypes.c_int),
                ('scale', ctypes.c_void_p), ('shift', ctypes.c_void_p),
                ('y', ctypes.c_void_p), ('ldy', ctypes.c_int),
                ('flags', ctypes.c_int),
                ('workspace', ctypes.c_void_p), ('workspace_bytes', ctypes.c_size_t),
                ('split_k', ctypes.c_int), ('algo', ctypes.c_int), ('fuse_offset', ctypes.c_int),
                ('w_off_packed', ctypes.c_void_p), ('b_off', ctypes.c_void_p),
                ('up_w', ctypes.c_void_p), ('up_f', ctypes.c_int), ('up_skip', ctypes.c_void_p), ('up_lds', ctypes.c_int),
                ('up_y', ctypes.c_void_p), ('up_ldy', ctypes.c_int),
                ('om_partial', ctypes.c_void_p), ('om_partial_bytes', ctypes.c_size_t),
                ('w_off_winograd', ctypes.c_void_p)]


CT_MAX_FUSED_HEADS = 8


class HeadsDesc(ctypes.Structure):
    _fields_ = [('x', ctypes.c_void_p), ('N', ctypes.c_int), ('H', ctypes.c_int), ('W', ctypes.c_int),
                ('Cin', ctypes.c_int), ('ldx', ctypes.c_int),
                ('w0_winograd', ctypes.c_void_p), ('b0', ctypes.c_void_p), ('nheads', ctypes.c_int),
                ('w2', ctypes.c_void_p), ('b2', ctypes.c_void_p),
                ('cout', ctypes.c_int * CT_MAX_FUSED_HEADS), ('coff', ctypes.c_int * CT_MAX_FUSED_HEADS),
                ('out', ctypes.c_void_p), ('ctot', ctypes.c_int),
                ('sig_lo', ctypes.c_int), ('sig_hi', ctypes.c_int), ('dep_lo', ctypes.c_int), ('dep_hi', ctypes.c_int),
                ('depth_scale', ctypes.c_float)]


class SparseHeadsDesc(ctypes.Structure):
    _fields_ = [('feat', ctypes.c_void_p), ('ldf', ctypes.c_int), ('nheads', ctypes.c_int),
                ('head', ctypes.c_int * NUM_HEADS),
                ('w1', ctypes.c_void_p * NUM_HEADS), ('b1', ctypes.c_void_p * NUM_HEADS),
                ('w2', ctypes.c_void_p * NUM_HEADS), ('b2', ctypes.c_void_p * NUM_HEADS),
                ('depth_scale', ctypes.c_float), ('zero_tracking', ctypes.c_int),
                ('flip_B', ctypes.c_int), ('flip_mode', ctypes.c_int * NUM_HEADS)]


class DecodeDesc(ctypes.Structure):
    _fields_ = [('hm', ctypes.c_void_p), ('B', ctypes.c_int), ('C', ctypes.c_int), ('h', ctypes.c_int),
                ('w', ctypes.c_int), ('K', ctypes.c_int),
                ('heads', ctypes.c_void_p * NUM_HEADS),
                ('out', ctypes.c_void_p), ('inds', ctypes.c_void_p),
                ('workspace', ctypes.c_void_p), ('workspace_bytes', ctypes.c_size_t),
                ('hm_batch_stride', ctypes.c_size_t), ('head_batch_stride', ctypes.c_size_t * NUM_HEADS),
                ('out_stride', ctypes.c_int),
                ('host_out', ctypes.c_void_p), ('done_flag', ctypes.c_void_p), ('done_counter', ctypes.c_void_p),
                ('sparse', ctypes.POINTER(SparseHeadsDesc))]


class PoseDesc(ctypes.Structure):
    _fields_ = [('rows', ctypes.c_void_p), ('row_floats', ctypes.c_int), ('box_col', ctypes.c_int),
                ('inds', ctypes.c_void_p),
                ('B', ctypes.c_int), ('h', ctypes.c_int), ('w', ctypes.c_int), ('K', ctypes.c_int),
                ('num_joints', ctypes.c_int),
                ('hps', ctypes.c_void_p), ('hm_hp', ctypes.c_void_p), ('hp_offset', ctypes.c_void_p),
                ('hps_batch_stride', ctypes.c_size_t), ('hm_hp_batch_stride', ctypes.c_size_t),
                ('hp_offset_batch_stride', ctypes.c_size_t),
                ('out', ctypes.c_void_p), ('out_stride', ctypes.c_int),
                ('workspace', ctypes.c_void_p), ('workspace_bytes', ctypes.c_size_t),
                ('box_wh', ctypes.c_void_p), ('box_reg', ctypes.c_void_p), ('box_ltrb', ctypes.c_void_p),
                ('box_wh_batch_stride', ctypes.c_size_t), ('box_reg_batch_stride', ctypes.c_size_t),
                ('box_ltrb_batch_stride', ctypes.c_size_t)]


class FlipHead(ctypes.Structure):
    _fields_ = [('src', ctypes.c_void_p), ('dst', ctypes.c_void_p), ('src_batch_stride', ctypes.c_size_t),
                ('C', ctypes.c_int), ('mode', ctypes.c_int)]


CT_FLIP_AVG, CT_FLIP_NEG_EVEN, CT_FLIP_JOINTS, CT_FLIP_JOINT_OFFSETS = range(4)


class RowLayout(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in ('score', 'cls', 'cts', 'tracking', 'bbox', 'amodel_offset')]


class Track(ctypes.Structure):
    _fields_ = [('score', ctypes.c_float), ('cls', ctypes.c_int), ('ct', ctypes.c_float * 2),
                ('tracking', ctypes.c_float * 2), ('bbox', ctypes.c_float * 4),
                ('tracking_id', ctypes.c_int), ('age', ctypes.c_int), ('active', ctypes.c_int),
                ('row', ctypes.c_int)]


class PrestageDesc(ctypes.Structure):
    _fields_ = [('enabled', ctypes.c_int), ('N', ctypes.c_int), ('H', ctypes.c_int), ('W', ctypes.c_int),
                ('w_x', ctypes.c_void_p), ('w_img', ctypes.c_void_p), ('scale3', ctypes.c_void_p), ('shift3', ctypes.c_void_p),
                ('partial', ctypes.c_void_p * 3), ('ldp', ctypes.c_int), ('flip_B', ctypes.c_int)]


class FrameLoopDesc(ctypes.Structure):
    _fields_ = [('B', ctypes.c_int), ('K', ctypes.c_int), ('F', ctypes.c_int),
                ('trackers', ctypes.POINTER(ctypes.c_void_p)),
                ('layout', RowLayout),
                ('out_thresh', ctypes.c_float), ('pre_thresh', ctypes.c_float),
                ('inp_w', ctypes.c_int), ('inp_h', ctypes.c_int),
                ('host_rows', ctypes.c_void_p), ('rows_keep', ctypes.c_void_p),
                ('blob_params', ctypes.c_void_p), ('blob_counts', ctypes.c_void_p), ('blob_cap', ctypes.c_int),
                ('nslots', ctypes.c_int),
                ('graphs', ctypes.c_void_p * 3), ('frames', ctypes.c_void_p * 3),
                ('frame_bytes', ctypes.c_size_t),
                ('stream', ctypes.c_void_p),
                ('results', ctypes.c_void_p), ('results_cap', ctypes.c_int), ('done_flag', ctypes.c_void_p),
                ('pre', PrestageDesc)]


class FrameStepArgs(ctypes.Structure):
    _fields_ = [('slot', ctypes.c_int), ('frame_kind', ctypes.c_int),
                ('frame', ctypes.c_void_p), ('next_frame', ctypes.c_void_p),
                ('trans_input', ctypes.c_void_p), ('trans_inv', ctypes.c_void_p)]


CT_FRAME_DEVICE, CT_FRAME_HOST, CT_FRAME_IN_PLACE, CT_FRAME_UPLOADED = range(4)


ABI_VERSION = 103       # CT_ABI_VERSION of include/centertrack_hip.h

EXPORTS = ['ct_last_error', 'ct_version', 'ct_set_tuning', 'ct_packed_weight_elems', 'ct_pack_conv_weight',
           'ct_packed_winograd_elems', 'ct_pack_winograd_weight', 'ct_conv2d',
           'ct_conv2d_workspace_bytes', 'ct_heads_fused', 'ct_dcn_v2', 'ct_dcn_v2_workspace_bytes', 'ct_dcn_v2_offsets_bytes', 'ct_dcn_v2_group', 'ct_dcn_v2_group_workspace_bytes', 'ct_dcn_v2_group_plan', 'ct_stem_forward',
           'ct_maxpool2x2', 'ct_upsample_add', 'ct_nchw_to_nhwc', 'ct_nhwc_to_nchw',
           'ct_decode_row_floats', 'ct_decode_workspace_bytes', 'ct_decode', 'ct_decode_pose_workspace_bytes',
           'ct_decode_pose', 'ct_render_pre_hm',
           'ct_tracker_create', 'ct_tracker_destroy', 'ct_tracker_reset', 'ct_tracker_num_tracks',
           'ct_tracker_id_count', 'ct_tracker_get_tracks', 'ct_tracker_step', 'ct_tracker_prehm_params', 'ct_linear_assignment', 'ct_tracker_set_mode', 'ct_tracker_init_tracks',
           'ct_tracker_step_public', 'ct_tracker_step_dets', 'ct_transform_points',
           'ct_preprocess_image', 'ct_preprocess_lut', 'ct_preprocess_device', 'ct_graph_begin', 'ct_graph_end', 'ct_graph_launch', 'ct_graph_destroy',
           'ct_memcpy_async', 'ct_memset_async', 'ct_stream_synchronize', 'ct_flip_merge', 'ct_flip_images',
           'ct_frame_loop_create', 'ct_frame_loop_destroy', 'ct_frame_loop_submit', 'ct_frame_loop_wait', 'ct_frame_loop_finish',
           'ct_frame_loop_finish_submit', 'ct_frame_loop_upload', 'ct_frame_loop_pending_slot', 'ct_frame_loop_in_flight',
           'ct_frame_loop_forget_upload', 'ct_frame_loop_prestage', 'ct_stem_forward_parts', 'ct_signal_host']

_lib = None


class CTError(RuntimeError):
    pass


def load():
    """Load the HIP library (once).  Raises if it has not been built: the product path
    never falls back to a CPU implementation."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CTError('%s not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                      '(hipcc --offload-arch=gfx950); there is no CPU fallback' % LIB_PATH)
    # torch first: its wheel bundles its own libamdhip64 / HSA runtime, and the library must bind to THAT copy (same
    # soname as /opt/rocm's).  Loaded the other way round (this library, then torch) the process ends up with the
    # system runtime under torch's allocator and every launch fails with "no ROCm-capable device is detected".
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    i, p, sz = ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t
    lib.ct_last_error.restype = ctypes.c_char_p
    lib.ct_version.restype = i
    if lib.ct_version() != ABI_VERSION:      # the ctypes structs below mirror ONE header version (tests/test_cabi.py)
        raise CTError('%s is ABI version %d, this binding was written against %d: rebuild it (python -m '
                      'centertrack_amd.build)' % (LIB_PATH, lib.ct_version(), ABI_VERSION))
    lib.ct_set_tuning.argtypes = [ctypes.c_char_p, i]
    lib.ct_packed_weight_elems.restype = sz
    lib.ct_packed_weight_elems.argtypes = [i, i, i]
    lib.ct_pack_conv_weight.argtypes = [p, p, i, i, i, p]
    lib.ct_packed_winograd_elems.restype = sz
    lib.ct_packed_winograd_elems.argtypes = [i, i]
    lib.ct_pack_winograd_weight.argtypes = [p, p, i, i, p]
    lib.ct_conv2d.argtypes = [ctypes.POINTER(ConvDesc), p]
    lib.ct_conv2d_workspace_bytes.restype = sz
    lib.ct_conv2d_workspace_bytes.argtypes = [ctypes.POINTER(ConvDesc)]
    lib.ct_heads_fused.argtypes = [ctypes.POINTER(HeadsDesc), p]
    lib.ct_dcn_v2.argtypes = [ctypes.POINTER(DcnDesc), p]
    lib.ct_dcn_v2_workspace_bytes.restype = sz
    lib.ct_dcn_v2_workspace_bytes.argtypes = [ctypes.POINTER(DcnDesc)]
    lib.ct_dcn_v2_offsets_bytes.restype = sz
    lib.ct_dcn_v2_offsets_bytes.argtypes = [ctypes.POINTER(DcnDesc)]
    lib.ct_dcn_v2_group.argtypes = [ctypes.POINTER(DcnDesc), i, i, p]
    lib.ct_dcn_v2_group_workspace_bytes.restype = sz
    lib.ct_dcn_v2_group_workspace_bytes.argtypes = [ctypes.POINTER(DcnDesc)]
    lib.ct_dcn_v2_group_plan.argtypes = [ctypes.POINTER(DcnDesc), ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_int)]
    lib.ct_stem_forward.argtypes = [p, p, p, i, i, i, p, p, p, p, p, p, i, p]
    lib.ct_stem_forward_parts.argtypes = [p, p, p, p, i, i, i, i, p, p, p, p, p, p, i, p]
    lib.ct_maxpool2x2.argtypes = [p, i, i, i, i, i, p, i, p]
    lib.ct_upsample_add.argtypes = [p, i, i, i, i, i, p, i, p, i, p, i, p]
    lib.ct_nchw_to_nhwc.argtypes = [p, i, i, i, i, p, i, p]
    lib.ct_nhwc_to_nchw.argtypes = [p, i, i, i, i, i, p, p]
    lib.ct_decode_row_floats.argtypes = [ctypes.POINTER(DecodeDesc)]
    lib.ct_decode_workspace_bytes.restype = sz
    lib.ct_decode_workspace_bytes.argtypes = [ctypes.POINTER(DecodeDesc)]
    lib.ct_decode.argtypes = [ctypes.POINTER(DecodeDesc), p]
    lib.ct_decode_pose_workspace_bytes.restype = sz
    lib.ct_decode_pose_workspace_bytes.argtypes = [ctypes.POINTER(PoseDesc)]
    lib.ct_decode_pose.argtypes = [ctypes.POINTER(PoseDesc), p]
    lib.ct_render_pre_hm.argtypes = [p, p, i, i, i, i, p, i, p]
    lib.ct_tracker_create.restype = p
    lib.ct_tracker_create.argtypes = [ctypes.c_float, i]
    lib.ct_tracker_destroy.restype = None
    lib.ct_tracker_destroy.argtypes = [p]
    lib.ct_tracker_reset.restype = None
    lib.ct_tracker_reset.argtypes = [p]
    lib.ct_tracker_num_tracks.argtypes = [p]
    lib.ct_tracker_id_count.argtypes = [p]
    lib.ct_tracker_get_tracks.argtypes = [p, p, i]
    lib.ct_tracker_step.argtypes = [p, p, i, i, ctypes.POINTER(RowLayout), ctypes.c_float, p, p, i]
    lib.ct_tracker_prehm_params.argtypes = [p, ctypes.c_float, p, i, i, p, i]
    lib.ct_linear_assignment.argtypes = [p, i, i, p, p]
    lib.ct_tracker_set_mode.argtypes = [p, i, i]
    lib.ct_tracker_init_tracks.argtypes = [p, p, i]
    lib.ct_tracker_step_public.argtypes = [p, p, i, i, ctypes.POINTER(RowLayout), ctypes.c_float, p, p, i, p, i]
    lib.ct_tracker_step_dets.argtypes = [p, p, i, p, i, p, i]
    lib.ct_transform_points.argtypes = [p, p, i, p]
    lib.ct_preprocess_image.argtypes = [p, i, i, i, i, p, i, i, p, p, p, i]
    lib.ct_preprocess_lut.argtypes = [p, p, i, p]
    lib.ct_preprocess_device.argtypes = [p, i, i, i, i, p, i, i, p, p, p, p]
    lib.ct_graph_begin.argtypes = [p]
    lib.ct_graph_end.restype = p
    lib.ct_graph_end.argtypes = [p]
    lib.ct_graph_launch.argtypes = [p, p]
    lib.ct_graph_destroy.restype = None
    lib.ct_graph_destroy.argtypes = [p]
    lib.ct_memcpy_async.argtypes = [p, p, sz, i, p]
    lib.ct_stream_synchronize.argtypes = [p]
    lib.ct_memset_async.argtypes = [p, i, sz, p]
    lib.ct_signal_host.argtypes = [p, i, p]
    lib.ct_flip_merge.argtypes = [ctypes.POINTER(FlipHead), i, p, i, i, i, i, p]
    lib.ct_flip_images.argtypes = [p, p, sz, i, p]
    lib.ct_frame_loop_create.restype = p
    lib.ct_frame_loop_create.argtypes = [ctypes.POINTER(FrameLoopDesc)]
    lib.ct_frame_loop_destroy.restype = None
    lib.ct_frame_loop_destroy.argtypes = [p]
    lib.ct_frame_loop_submit.argtypes = [p, ctypes.POINTER(FrameStepArgs)]
    lib.ct_frame_loop_wait.argtypes = [p]
    lib.ct_frame_loop_finish.argtypes = [p, ctypes.POINTER(FrameStepArgs), p]
    lib.ct_frame_loop_finish_submit.argtypes = [p, ctypes.POINTER(FrameStepArgs), p, ctypes.POINTER(FrameStepArgs)]
    lib.ct_frame_loop_upload.argtypes = [p, i, p]
    lib.ct_frame_loop_pending_slot.argtypes = [p]
    lib.ct_frame_loop_prestage.argtypes = [p, i]
    lib.ct_frame_loop_in_flight.argtypes = [p]
    lib.ct_frame_loop_forget_upload.restype = None
    lib.ct_frame_loop_forget_upload.argtypes = [p]
    # CENTERTRACK_TUNE="key=value,key=value": launch-heuristic knobs of ct_set_tuning (A/B runs)
    for kv in filter(None, os.environ.get('CENTERTRACK_TUNE', '').split(',')):
        k, _, v = kv.partition('=')
        if lib.ct_set_tuning(k.strip().encode(), int(v)) != 0:
            raise CTError('CENTERTRACK_TUNE: %s' % lib.ct_last_error().decode())
    _lib = lib
    return lib


def check(rc, what=''):
    if rc != 0:
        raise CTError('%s failed (code %d): %s' % (what or 'libcentertrack_hip call', rc,
                                                  load().ct_last_error().decode()))


def stream_ptr():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


import contextlib as _contextlib


@_contextlib.contextmanager
def capture_guard(collect=True):
    """No run of the cyclic garbage collector while a stream is capturing.  A detector that is only reachable through reference
    cycles (its context holds closures over itself) is destroyed whenever the collector happens to run; its ``__del__`` waits for
    the device and destroys HIP objects, and such a call in the middle of a capture invalidates the capture ("operation failed due
    to a previous error during capture") -- in whichever test or request happens to be capturing at that moment (round 6: 13
    cascading failures in some runs of the GPU suite, none in others).  Garbage is collected BEFORE the capture instead, as
    ``torch.cuda.graph`` does on entry."""
    import gc
    if os.environ.get('CT_NO_CAPTURE_GUARD') == '1':        # (tests only: shows what the guard prevents)
        yield
        return
    if collect:             # (``torch.cuda.graph`` collects on entry itself: its call sites pass False)
        gc.collect()
    was = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()

