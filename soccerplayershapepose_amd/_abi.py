"""ctypes binding of include/spsp.h -- the ONLY way the Python layer reaches the HIP kernels.

There is no CPU fallback: if `csrc/libspsp.so` is missing `load()` raises, and every op of the
package raises with it.  Structure layouts below mirror include/spsp.h field for field.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, 'csrc', 'libspsp.so')

NUM_CONVS = 20
RECORD_FLOATS = 229

FIT_GLOBAL_ORIENT = 1 << 0
FIT_CAM = 1 << 24
FIT_BETAS = 1 << 25
FIT_MASK_CAMERA = FIT_GLOBAL_ORIENT | FIT_CAM
FIT_MASK_POSE_SHAPE = FIT_GLOBAL_ORIENT | FIT_CAM | FIT_BETAS | 0x003FFE7E
FIT_SELECT_LAST, FIT_SELECT_BEST_L2E = 0, 1
OK, EINVAL, ENOMEM, EHIP, ENCCL = 0, -1, -2, -3, -4           # include/spsp.h status codes
PROXY_NCHW_F32, PROXY_NATIVE = 0, 1
PRECISION_BF16, PRECISION_F16 = 0, 1
SIL_U8, SIL_F32 = 0, 1
DEBUG_IEF_STALL, DEBUG_PLAN_S2, DEBUG_PLAN_LAYER4_CHANNELS, DEBUG_PLAN_PATCH, DEBUG_PLAN_PATCH_PRODUCER, DEBUG_PLAN_S2RS, DEBUG_PLAN_P4, DEBUG_PLAN_WRES_CUS, DEBUG_PLAN_STEM_BAND, DEBUG_PLAN_P4_CUS, DEBUG_PLAN_S2RS_CUS, DEBUG_IEF_BLOCK, DEBUG_PLAN_FB, DEBUG_PLAN_S2Q, DEBUG_PLAN_STEM32, DEBUG_PLAN_RS, DEBUG_IEF_POLL = 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17
DEBUG_FIT_SCHEDULE, DEBUG_SKIN_PLAYERS = 1, 2          # spsp_smpl_debug
REGRESSOR_MAX_BATCH, REGRESSOR_MAX_CONCURRENT = 256, 8
PLAN_THROUGHPUT, PLAN_LATENCY = 0, 1

_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int32)


class ConvBN(C.Structure):
    _fields_ = [('weight', _fp), ('bn_weight', _fp), ('bn_bias', _fp), ('bn_mean', _fp), ('bn_var', _fp)]


class RegressorWeights(C.Structure):
    _fields_ = [('conv', ConvBN * NUM_CONVS), ('fc_weight', _fp * 3), ('fc_bias', _fp * 3),
                ('init_params', _fp), ('ief_iters', C.c_int)]


NUM_RELATIONS = 27


class RelateUnit(C.Structure):
    _fields_ = [(f'{w}_{k}', _fp) for w in ('wg', 'wk', 'wq', 'wv') for k in ('weight', 'bias')]


class RelateWeights(C.Structure):
    _fields_ = [('unit', RelateUnit * NUM_RELATIONS)]


class SmplArrays(C.Structure):
    _fields_ = [('v_template', _fp), ('shapedirs', _fp), ('posedirs', _fp), ('J_regressor', _fp),
                ('lbs_weights', _fp), ('parents', _ip), ('vertex_joint_ids', _ip),
                ('J_regressor_extra', _fp), ('J_regressor_cocoplus', _fp), ('J_regressor_h36m', _fp)]


class FitConfig(C.Structure):
    _fields_ = [('iters', C.c_int), ('lr', C.c_double), ('beta1', C.c_double), ('beta2', C.c_double),
                ('eps', C.c_double), ('loss_weight', C.c_double), ('param_mask', C.c_uint32),
                ('img_wh', C.c_float), ('loss_wh', C.c_float), ('select_mode', C.c_int)]


class FitMultiviewConfig(C.Structure):
    _fields_ = [('iters', C.c_int), ('rounds', C.c_int), ('lr', C.c_double), ('beta1', C.c_double), ('beta2', C.c_double),
                ('eps', C.c_double), ('loss_weight', C.c_double), ('img_wh', C.c_float), ('loss_wh', C.c_float),
                ('optimise_body_pose', C.c_int)]


class FitSilhouetteConfig(C.Structure):
    _fields_ = [('iters', C.c_int), ('lr', C.c_double), ('beta1', C.c_double), ('beta2', C.c_double), ('eps', C.c_double),
                ('joints2d_weight', C.c_double), ('silhouette_weight', C.c_double), ('param_mask', C.c_uint32),
                ('img_wh', C.c_float), ('loss_wh', C.c_float), ('focal', C.c_float)]


class PipelineOut(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ('cam', 'pose6d', 'betas', 'rotmats', 'vertices', 'joints', 'translation',
                                          'record', 'relate_tmp')]


def f32(a):
    """C-contiguous float32 numpy view/copy + its ctypes pointer (keep the array alive!)."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_fp)


def i32(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_ip)


def smpl_arrays(model):
    """(SmplArrays, keepalive list) from a dict with the smplx array names (see synthetic.smpl_model)."""
    keep, s = [], SmplArrays()
    for name in ('v_template', 'shapedirs', 'posedirs', 'J_regressor', 'lbs_weights',
                 'J_regressor_extra', 'J_regressor_cocoplus', 'J_regressor_h36m'):
        a, p = f32(model[name])
        keep.append(a)
        setattr(s, name, p)
    for name in ('parents', 'vertex_joint_ids'):
        a, p = i32(model[name])
        keep.append(a)
        setattr(s, name, p)
    expect = {'v_template': 6890 * 3, 'shapedirs': 6890 * 30, 'posedirs': 207 * 20670, 'J_regressor': 24 * 6890,
              'lbs_weights': 6890 * 24, 'J_regressor_extra': 9 * 6890, 'J_regressor_cocoplus': 19 * 6890,
              'J_regressor_h36m': 17 * 6890, 'parents': 24, 'vertex_joint_ids': 21}
    for (name, n), a in zip(expect.items(), keep):
        if a.size != n:
            raise ValueError(f'SMPL array {name} has {a.size} elements, expected {n}')
    return s, keep


_vp = C.c_void_p
_PROTOS = {
    'spsp_init': (C.c_int, [C.c_int, C.POINTER(_vp)]),
    'spsp_destroy': (None, [_vp]),
    'spsp_last_error': (C.c_char_p, []),
    'spsp_version': (C.c_char_p, []),
    'spsp_profile_enable': (C.c_int, [_vp, C.c_int]),
    'spsp_profile_select': (C.c_int, [_vp, C.c_int]),
    'spsp_profile_num_kernels': (C.c_int, []),
    'spsp_profile_kernel_name': (C.c_char_p, [C.c_int]),
    'spsp_profile_kernel_flop_per_crop': (C.c_double, [C.c_int]),
    'spsp_profile_kernel_in_plan': (C.c_int, [C.c_int, C.c_int]),
    'spsp_profile_timeline': (C.c_int, [_vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_ulonglong), C.POINTER(C.c_double),
                                       C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    'spsp_profile_read': (C.c_int, [_vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    'spsp_profile_read_grids': (C.c_int, [_vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_double)]),
    'spsp_conv_name': (C.c_char_p, [C.c_int]),
    'spsp_conv_bn_name': (C.c_char_p, [C.c_int]),
    'spsp_regressor_load': (C.c_int, [_vp, C.POINTER(RegressorWeights), C.c_int, C.c_int, C.POINTER(_vp)]),
    'spsp_regressor_free': (None, [_vp]),
    'spsp_proxy_native_bytes': (C.c_size_t, []),
    'spsp_proxy_pack': (C.c_int, [_vp, _vp, C.c_int, C.c_int, _vp, _vp]),
    'spsp_comm_unique_id': (C.c_int, [_vp]),
    'spsp_comm_init_rank': (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.POINTER(_vp)]),
    'spsp_comm_destroy': (None, [_vp]),
    'spsp_allgather_records': (C.c_int, [_vp, _vp, _vp, C.c_int, _vp, _vp]),
    'spsp_pipeline': (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int, _vp, _vp, C.POINTER(FitConfig), C.c_int, C.c_float,
                               C.c_float, C.POINTER(PipelineOut), _vp]),
    'spsp_pipeline_group': (C.c_int, [_vp, _vp, _vp, C.POINTER(_vp), C.c_int, C.c_int, _vp, C.POINTER(FitConfig), C.c_int, C.c_float,
                                     C.c_float, C.POINTER(PipelineOut), _vp]),
    'spsp_relate_load': (C.c_int, [_vp, C.POINTER(RelateWeights), C.c_int, C.POINTER(_vp)]),
    'spsp_relate_free': (None, [_vp]),
    'spsp_relate_forward': (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, _vp, _vp]),
    'spsp_proxy_build': (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    'spsp_crop_players': (C.c_int, [_vp, _vp, C.c_int, C.c_int, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    'spsp_regressor_forward': (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp]),
    'spsp_regressor_backbone': (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    'spsp_regressor_ief': (C.c_int, [_vp, _vp, C.c_int, _vp, _vp, _vp, _vp, _vp]),
    'spsp_regressor_read_stage': (C.c_int, [_vp, _vp, C.c_int, C.c_int, _vp, _vp]),
    'spsp_regressor_status': (C.c_int, [_vp]),
    'spsp_regressor_set_plan': (C.c_int, [_vp, C.c_int]),
    'spsp_regressor_debug': (C.c_int, [_vp, C.c_int, C.c_int]),
    'spsp_rot6d_to_rotmat': (C.c_int, [_vp, _vp, C.c_int, _vp, _vp]),
    'spsp_project': (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_float, _vp, _vp]),
    'spsp_wp_to_translation': (C.c_int, [_vp, _vp, C.c_int, C.c_float, C.c_float, _vp, _vp]),
    'spsp_pack_record': (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, _vp, _vp]),
    'spsp_unpack_record': (C.c_int, [_vp, _vp, C.c_int, _vp, _vp, _vp, _vp]),
    'spsp_smpl_load': (C.c_int, [_vp, C.POINTER(SmplArrays), C.c_int, C.POINTER(_vp)]),
    'spsp_smpl_free': (None, [_vp]),
    'spsp_smpl_debug': (C.c_int, [_vp, C.c_int, C.c_int]),
    'spsp_smpl_forward': (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, _vp, _vp, _vp]),
    'spsp_fit_config_default': (None, [C.POINTER(FitConfig), C.c_int]),
    'spsp_fit': (C.c_int, [_vp, _vp, C.POINTER(FitConfig), _vp, _vp, _vp, _vp, C.c_int, _vp, _vp, _vp]),
    'spsp_fit_trajectory': (C.c_int, [_vp, _vp, C.POINTER(FitConfig), _vp, _vp, _vp, _vp, C.c_int, _vp, _vp, _vp, _vp]),
    'spsp_rasterizer_load': (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.POINTER(_vp)]),
    'spsp_rasterizer_free': (None, [_vp]),
    'spsp_render_silhouettes': (C.c_int, [_vp, _vp, _vp, _vp, C.c_float, C.c_int, _vp, C.c_int, _vp, _vp, _vp]),
    'spsp_silhouette_loss_grad': (C.c_int, [_vp, _vp, _vp, _vp, C.c_float, C.c_int, _vp, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp]),
    'spsp_smpl_vertices_vjp': (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, _vp, _vp, _vp, _vp]),
    'spsp_fit_silhouette_config_default': (None, [C.POINTER(FitSilhouetteConfig)]),
    'spsp_fit_silhouette': (C.c_int, [_vp, _vp, _vp, C.POINTER(FitSilhouetteConfig), _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, _vp, _vp,
                                      _vp, _vp, _vp, _vp]),
    'spsp_fit_select': (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp]),
    'spsp_fit_multiview_config_default': (None, [C.POINTER(FitMultiviewConfig)]),
    'spsp_fit_multiview': (C.c_int, [_vp, _vp, C.POINTER(FitMultiviewConfig), C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'spsp_fit_multiview_ordered': (C.c_int, [_vp, _vp, C.POINTER(FitMultiviewConfig), C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'spsp_fit_multiview_block': (C.c_int, [_vp, _vp, C.POINTER(FitMultiviewConfig), C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                           _vp, _vp, _vp]),
}
EXPORTED_SYMBOLS = tuple(_PROTOS)

_lib = None


def load():
    """dlopen csrc/libspsp.so and attach prototypes.  Raises if the library was not built."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get('SPSP_LIB_PATH', LIB_PATH)      # diagnostic twins (csrc/libspsp_stamps.so) for tools/ only
    if not os.path.exists(path):
        raise RuntimeError(
            f'{path} not found: the HIP library is not built. Run `python -c "import __graft_entry__ as g; '
            f'g.build()"` (or `make -C soccerplayershapepose_amd/csrc`). There is no CPU fallback.')
    lib = C.CDLL(path)
    for name, (res, args) in _PROTOS.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().spsp_last_error().decode('utf-8', 'replace')
        if rc == EINVAL:
            raise ValueError(f'{what}: {msg}')
        raise RuntimeError(f'{what} failed (status {rc}): {msg}')
