"""ctypes binding of libyunet_hip.so (the C ABI declared in include/yunet_hip.h).

There is deliberately NO fallback: if the shared library is missing the product path
raises.  Build it with `python __graft_entry__.py` or `make -C libfacedetection.train_amd/csrc`.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# YUNET_HIP_LIB selects another build of the same library (kernel A/B measurements)
LIB_PATH = os.environ.get('YUNET_HIP_LIB') or os.path.join(_HERE, 'libyunet_hip.so')

c_f32p = C.c_void_p
MAX_LEVELS = 5

EINVAL, EOPCODE = -1, -2            # YUNET_EINVAL / YUNET_EOPCODE (include/yunet_hip.h)
T_IDENTITY, T_BNRELU = 0, 1
F32, BF16 = 0, 1
BOX_EIOU, BOX_DIOU, BOX_IOU_LINEAR, BOX_IOU_SQUARE, BOX_IOU_LOG, BOX_GIOU, BOX_CIOU = 0, 1, 2, 3, 4, 5, 6
(OP_STEM_FWD, OP_STEM_BWD, OP_DP_FWD, OP_DP_BWD, OP_POOL_FWD, OP_POOL_BWD, OP_UPADD_FWD,
 OP_UPADD_BWD, OP_BN_RUNNING, OP_BN_PARAM_GRAD, OP_REDUCE_PARTIALS, OP_ASSIGN, OP_LOSS_NORM,
 OP_LOSS, OP_LOSS_FINALIZE, OP_SGD, OP_MEMSET, OP_BN_BATCH, OP_REDUCE_BATCH, OP_FORK, OP_JOIN, OP_ADD) = range(1, 23)
OP_LANE, MAX_LANES = 10, 2          # YunetOp.i[OP_LANE]: side stream of the op (0 = the caller's stream)
OP_GROUP, DP_GROUP_MAX = 9, 3       # YunetOp.i[OP_GROUP] = g: this DP_FWD op and the g - 1 after it are independent (ABI 10)


class YunetBN(C.Structure):
    _fields_ = [('stats', C.c_void_p), ('bstats', C.c_void_p), ('gamma', C.c_void_p),
                ('beta', C.c_void_p), ('count', C.c_int32), ('eps', C.c_float),
                ('slots', C.c_int32), ('reserved_', C.c_int32)]


class YunetDP(C.Structure):
    _fields_ = [('N', C.c_int32), ('H', C.c_int32), ('W', C.c_int32), ('cin', C.c_int32),
                ('cout', C.c_int32), ('in_transform', C.c_int32), ('out_has_bn', C.c_int32),
                ('accumulate_dx', C.c_int32), ('x_img_stride', C.c_int64),
                ('z_img_stride', C.c_int64), ('x', C.c_void_p), ('in_bn', YunetBN),
                ('w_pw', C.c_void_p), ('b_pw', C.c_void_p), ('w_dw', C.c_void_p),
                ('b_dw', C.c_void_p), ('z', C.c_void_p), ('out_bn', YunetBN),
                ('dy', C.c_void_p), ('dy_scale', C.c_void_p), ('dx', C.c_void_p),
                ('wgrad_partials', C.c_void_p), ('wgrad_blocks', C.c_int32),
                ('prof', C.c_void_p), ('x_dtype', C.c_int32), ('z_dtype', C.c_int32),
                ('pool_out', C.c_void_p), ('pool_idx', C.c_void_p)]


class YunetLevels(C.Structure):
    _fields_ = [('num_levels', C.c_int32), ('h', C.c_int32 * MAX_LEVELS),
                ('w', C.c_int32 * MAX_LEVELS), ('stride', C.c_int32 * MAX_LEVELS)]


class YunetLossCfg(C.Structure):
    _fields_ = [('box_loss', C.c_int32), ('w_cls', C.c_float), ('w_box', C.c_float),
                ('w_obj', C.c_float), ('w_kps', C.c_float), ('box_eps', C.c_float),
                ('smooth_point', C.c_float), ('kps_beta', C.c_float), ('defer_num_total', C.c_int32)]


class YunetAssignCfg(C.Structure):
    _fields_ = [('center_radius', C.c_float), ('candidate_topk', C.c_int32), ('iou_weight', C.c_float),
                ('cls_weight', C.c_float)]


class YunetAugCfg(C.Structure):
    _fields_ = [('out_size', C.c_int32), ('n_choice', C.c_int32), ('crop_choice', C.c_double * 8),
                ('flip_ratio', C.c_double), ('pad_value', C.c_float), ('seed', C.c_uint32),
                ('max_attempts', C.c_int32), ('max_retries', C.c_int32), ('gmax', C.c_int32)]


MAX_RANKS, IPC_HANDLE_BYTES, COMM_HEADER_BYTES = 8, 64, 20480


class YunetComm(C.Structure):
    _fields_ = [('rank', C.c_int32), ('world', C.c_int32), ('seq', C.c_uint32), ('reserved_', C.c_int32),
                ('slot_bytes', C.c_uint64), ('inbox', C.c_void_p * MAX_RANKS), ('status', C.c_void_p)]


class YunetOp(C.Structure):
    _fields_ = [('opcode', C.c_int32), ('i', C.c_int32 * 12), ('f', C.c_float * 8),
                ('p', C.c_void_p * 12), ('bn', YunetBN * 2), ('dp', YunetDP),
                ('lv', YunetLevels), ('loss', YunetLossCfg)]


_SIGNATURES = {
    'yunet_abi_version': (C.c_int, []),
    'yunet_conv_blocks': (C.c_int, []),
    'yunet_loss_blocks': (C.c_int, [C.c_int, C.c_int]),
    'yunet_stem_fwd': (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 4 + [C.c_void_p]),
    'yunet_stem_bwd': (C.c_int, [C.c_void_p] * 3 + [C.POINTER(YunetBN), C.c_void_p] +
                       [C.c_int] * 5 + [C.c_void_p]),
    'yunet_stem_bwd_rz': (C.c_int, [C.c_void_p] * 4 + [C.POINTER(YunetBN), C.c_void_p] + [C.c_int] * 5 + [C.c_void_p]),
    'yunet_dp_fwd': (C.c_int, [C.POINTER(YunetDP), C.c_void_p]),
    'yunet_dp_bwd': (C.c_int, [C.POINTER(YunetDP), C.c_void_p]),
    'yunet_dp_fwd_group': (C.c_int, [C.POINTER(C.POINTER(YunetDP)), C.c_int, C.c_void_p]),
    'yunet_dp_bwd_blocks': (C.c_int, [C.c_int] * 5),
    'yunet_stem_bwd_blocks': (C.c_int, [C.c_int] * 3),
    'yunet_pool_fwd': (C.c_int, [C.c_void_p, C.POINTER(YunetBN), C.c_void_p] + [C.c_int] * 4 +
                       [C.c_void_p]),
    'yunet_dp_pool_fusion_ok': (C.c_int, [C.c_int] * 5),
    'yunet_pool_bwd': (C.c_int, [C.c_void_p, C.POINTER(YunetBN), C.c_void_p, C.c_void_p] +
                       [C.c_int] * 5 + [C.c_void_p]),
    'yunet_pool_bwd_add': (C.c_int, [C.c_void_p, C.POINTER(YunetBN), C.c_void_p, C.c_void_p, C.c_void_p] +
                           [C.c_int] * 5 + [C.c_void_p]),
    'yunet_upadd_fwd': (C.c_int, [C.c_void_p, C.POINTER(YunetBN), C.c_void_p, C.POINTER(YunetBN),
                                  C.c_void_p] + [C.c_int] * 4 + [C.c_void_p]),
    'yunet_upadd_bwd': (C.c_int, [C.c_void_p, C.POINTER(YunetBN), C.c_void_p, C.POINTER(YunetBN),
                                  C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int] +
                        [C.c_int] * 4 + [C.c_void_p]),
    'yunet_bn_update_running': (C.c_int, [C.c_void_p] * 3 + [C.c_int, C.c_int, C.c_float,
                                                             C.c_void_p]),
    'yunet_bn_param_grad': (C.c_int, [C.c_void_p] * 3 + [C.c_int, C.c_int, C.c_void_p]),
    'yunet_bn_batch': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float,
                                 C.c_void_p, C.c_int, C.c_void_p]),
    'yunet_reduce_partials_batch': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    'yunet_detect': (C.c_int, [C.c_void_p, C.POINTER(YunetLevels), C.c_int, C.c_int, C.c_float, C.c_float,
                              C.c_int] + [C.c_void_p] * 5),
    'yunet_detect_scratch_bytes': (C.c_size_t, [C.c_int, C.c_int]),
    'yunet_nms': (C.c_int, [C.c_void_p] * 3 + [C.c_int, C.c_int, C.c_float, C.c_float, C.c_int] + [C.c_void_p] * 5),
    'yunet_aug_decide': (C.c_int, [C.c_void_p] * 4 + [C.POINTER(YunetAugCfg), C.c_uint32, C.c_int] +
                         [C.c_void_p] * 5),
    'yunet_aug_pixels': (C.c_int, [C.c_void_p] * 4 + [C.POINTER(YunetAugCfg), C.c_int, C.c_void_p, C.c_void_p]),
    'yunet_reduce_partials': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                        C.c_void_p]),
    'yunet_assign': (C.c_int, [C.c_void_p] * 5 + [C.POINTER(YunetLevels)] + [C.c_int] * 3 +
                     [C.c_float] + [C.c_void_p] * 6),
    'yunet_assign_ex': (C.c_int, [C.c_void_p] * 7 + [C.POINTER(YunetLevels)] + [C.c_int] * 3 +
                        [C.c_float] + [C.c_void_p] * 6),
    'yunet_assign_cfg': (C.c_int, [C.c_void_p] * 7 + [C.POINTER(YunetLevels)] + [C.c_int] * 3 +
                         [C.POINTER(YunetAssignCfg)] + [C.c_void_p] * 6),
    'yunet_loss_norm': (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    'yunet_loss': (C.c_int, [C.c_void_p] * 5 + [C.POINTER(YunetLevels), C.POINTER(YunetLossCfg),
                                                C.c_void_p] + [C.c_int] * 3 +
                   [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    'yunet_loss_finalize': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    'yunet_loss_finalize_ex': (C.c_int, [C.c_void_p, C.c_int] + [C.c_void_p] * 5),
    'yunet_sgd_step_ex': (C.c_int, [C.c_void_p] * 3 + [C.c_int64, C.c_void_p, C.c_float, C.c_float, C.c_int, C.c_float,
                                    C.c_float, C.c_int, C.c_void_p]),
    'yunet_add': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    'yunet_sgd_step': (C.c_int, [C.c_void_p] * 3 + [C.c_int64, C.c_void_p, C.c_float, C.c_float,
                                                    C.c_float, C.c_int, C.c_void_p]),
    'yunet_exec': (C.c_int, [C.POINTER(YunetOp), C.c_int, C.c_void_p]),
    'yunet_exec_lanes': (C.c_int, [C.c_int]),
    'yunet_set_option': (C.c_int, [C.c_char_p, C.c_int]),
    'yunet_comm_inbox_bytes': (C.c_size_t, [C.c_int, C.c_size_t]),
    'yunet_comm_alloc': (C.c_int, [C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    'yunet_comm_free': (C.c_int, [C.c_void_p, C.c_void_p]),
    'yunet_comm_export': (C.c_int, [C.c_void_p, C.c_char_p]),
    'yunet_comm_open': (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    'yunet_comm_close': (C.c_int, [C.c_void_p]),
    'yunet_allreduce': (C.c_int, [C.POINTER(YunetComm), C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]),
    'yunet_comm_status': (C.c_int, [C.POINTER(YunetComm)]),
}
for _n in ('yunet_stem_fwd', 'yunet_stem_bwd', 'yunet_dp_fwd', 'yunet_dp_bwd', 'yunet_dp_fwd_group', 'yunet_pool_fwd', 'yunet_pool_bwd',
           'yunet_pool_bwd_add', 'yunet_upadd_fwd', 'yunet_upadd_bwd'):
    _SIGNATURES[_n + '_bf16'] = _SIGNATURES[_n]      # same arguments, bf16 activation storage

EXPORTED = sorted(_SIGNATURES)

_lib = None


class YunetHipError(RuntimeError):
    pass


def load():
    """dlopen the kernel library (no GPU needed to load it); raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise YunetHipError(
            f'{LIB_PATH} not found: the HIP extension is not built.  Run '
            '`python __graft_entry__.py` (or `make -C libfacedetection.train_amd/csrc`). '
            'There is no CPU fallback.')
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.yunet_abi_version() != 11:
        raise YunetHipError('libyunet_hip.so ABI version mismatch; rebuild it')
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise YunetHipError(f'{what} failed with status {rc}')


def set_option(name, value):
    """yunet_set_option(): a measurement switch of the C dispatchers (include/yunet_hip.h); returns the previous value."""
    rc = load().yunet_set_option(name.encode(), int(value))
    if rc < 0:
        raise YunetHipError(f'yunet_set_option({name!r}, {value}) rejected')
    return rc
