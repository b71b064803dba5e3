"""ctypes binding of libepos_hip.so (the C ABI declared in include/epos_hip.h).

There is no fallback: if the HIP library is missing or does not load, importing a
compute entry point raises. Nothing in here (or anywhere in epos_amd) touches the
CPU oracle.
"""
import ctypes
import os

from epos_amd import build as _build

c_f32p = ctypes.POINTER(ctypes.c_float)
c_f64p = ctypes.POINTER(ctypes.c_double)
c_i32p = ctypes.POINTER(ctypes.c_int32)
c_i64p = ctypes.POINTER(ctypes.c_int64)
c_u64p = ctypes.POINTER(ctypes.c_uint64)
vp = ctypes.c_void_p


class EposError(RuntimeError):
  pass


class PointwiseArgs(ctypes.Structure):
  _fields_ = [
      ('A', vp), ('lda', ctypes.c_int64),
      ('Wp', vp), ('bias', vp),
      ('R', vp), ('ldr', ctypes.c_int64),
      ('C', vp), ('ldc', ctypes.c_int64),
      ('M', ctypes.c_int32), ('N', ctypes.c_int32), ('K', ctypes.c_int32),
      ('relu', ctypes.c_int32), ('relu_in', ctypes.c_int32),
      ('sub', ctypes.c_int32),
      ('Ho', ctypes.c_int32), ('Wo', ctypes.c_int32),
      ('Hi', ctypes.c_int32), ('Wi', ctypes.c_int32),
      ('Ws', vp),
      # ABI 5: fp16-pair GEMM (include/epos_hip.h)
      ('Wh', vp), ('a_amax', vp), ('a_amax2', vp),
      ('a_gain', ctypes.c_float), ('a_bias', ctypes.c_float),
      ('a_presplit', ctypes.c_int32),
      ('c_amax', vp),
      # ABI 6: 32-row block sums, streaming stores (ABI 7: the fused softmax flag is reserved)
      ('col_sums', vp), ('col_ld', ctypes.c_int64),
      ('c_stream', ctypes.c_int32), ('reserved0', ctypes.c_int32),
  ]


class DepthwiseArgs(ctypes.Structure):
  _fields_ = [
      ('X', vp), ('ldx', ctypes.c_int64),
      ('w9c', vp), ('bias', vp),
      ('Y', vp), ('ldy', ctypes.c_int64),
      ('B', ctypes.c_int32), ('Hi', ctypes.c_int32), ('Wi', ctypes.c_int32),
      ('Ho', ctypes.c_int32), ('Wo', ctypes.c_int32), ('C', ctypes.c_int32),
      ('stride', ctypes.c_int32), ('rate', ctypes.c_int32),
      ('relu_in', ctypes.c_int32), ('relu_out', ctypes.c_int32),
      # ABI 5: fp16-pair output (include/epos_hip.h)
      ('y_h2', ctypes.c_int32), ('x_amax', vp), ('x_amax2', vp),
      ('gain', ctypes.c_float), ('bias0', ctypes.c_float),
  ]


class SepConvArgs(ctypes.Structure):
  _fields_ = [('dw', DepthwiseArgs), ('pw', PointwiseArgs)]


class Conv3x3Args(ctypes.Structure):
  _fields_ = [
      ('X', vp), ('ldx', ctypes.c_int64),
      ('Wp', vp), ('bias', vp),
      ('Y', vp), ('ldy', ctypes.c_int64),
      ('B', ctypes.c_int32), ('H', ctypes.c_int32), ('W', ctypes.c_int32),
      ('Cin', ctypes.c_int32), ('Cout', ctypes.c_int32),
      ('stride', ctypes.c_int32), ('rate', ctypes.c_int32),
      ('relu', ctypes.c_int32),
      ('Ws', vp),
      ('Wh', vp), ('x_amax', vp), ('y_amax', vp),
  ]


class Im2colArgs(ctypes.Structure):
  _fields_ = [
      ('X', vp), ('ldx', ctypes.c_int64),
      ('col', vp), ('ldcol', ctypes.c_int64),
      ('B', ctypes.c_int32), ('Hi', ctypes.c_int32), ('Wi', ctypes.c_int32),
      ('Ho', ctypes.c_int32), ('Wo', ctypes.c_int32), ('C', ctypes.c_int32),
      ('stride', ctypes.c_int32), ('rate', ctypes.c_int32),
      ('pad', ctypes.c_int32), ('preprocess', ctypes.c_int32),
      ('amax_clear', vp), ('amax_words', ctypes.c_int64),        # ABI 6
  ]


class CorrSlot(ctypes.Structure):
  _fields_ = [('image', ctypes.c_int32), ('obj_id', ctypes.c_int32)]


class CorrOut(ctypes.Structure):
  _fields_ = [('px_id', vp), ('frag_id', vp), ('coord_2d', vp),
              ('coord_3d', vp), ('conf', vp), ('conf_obj', vp),
              ('conf_frag', vp)]


class FitParams(ctypes.Structure):
  _fields_ = [
      ('threshold', ctypes.c_double),
      ('neighborhood_ball_radius', ctypes.c_double),
      ('spatial_coherence_weight', ctypes.c_double),
      ('scaling_from_millimeters', ctypes.c_double),
      ('max_tanimoto_similarity', ctypes.c_double),
      ('conf', ctypes.c_double),
      ('proposal_engine_conf', ctypes.c_double),
      ('min_coverage', ctypes.c_double),
      ('min_triangle_area', ctypes.c_double),
      ('max_iters', ctypes.c_int32),
      ('min_point_number', ctypes.c_int32),
      ('max_model_number', ctypes.c_int32),
      ('max_model_number_for_optimization', ctypes.c_int32),
      ('use_prosac', ctypes.c_int32),
      ('lo_iters', ctypes.c_int32),
      ('gc_sweeps', ctypes.c_int32),
      ('pearl_iters', ctypes.c_int32),
  ]


class PnpRansacParams(ctypes.Structure):
  _fields_ = [
      ('iterations_count', ctypes.c_int32),
      ('min_point_number', ctypes.c_int32),
      ('reprojection_error', ctypes.c_double),
      ('confidence', ctypes.c_double),
  ]


# Every symbol include/epos_hip.h declares: name -> (restype, argtypes or None).
SYMBOLS = {
    'epos_abi_version': (ctypes.c_int, []),
    'epos_last_error': (ctypes.c_char_p, []),
    'epos_device_count': (ctypes.c_int, []),
    'epos_clock_probe': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    'epos_pack_pointwise_weights': (ctypes.c_int64,
                                    [vp, ctypes.c_int, ctypes.c_int, vp]),
    'epos_pack_pointwise_weights_split': (ctypes.c_int64,
                                          [vp, ctypes.c_int, ctypes.c_int, vp]),
    'epos_pack_pointwise_weights_h2': (ctypes.c_int64,
                                       [vp, ctypes.c_int, ctypes.c_int, vp]),
    'epos_absmax_f32': (ctypes.c_int, [vp, ctypes.c_int64, ctypes.c_int64,
                                       ctypes.c_int64, vp, vp]),
    'epos_amax_clear': (ctypes.c_int, [vp, ctypes.c_int64, vp]),
    'epos_pointwise_conv_f32': (ctypes.c_int,
                                [ctypes.POINTER(PointwiseArgs), vp]),
    'epos_pointwise_conv_grouped_f32': (ctypes.c_int, [
        ctypes.POINTER(PointwiseArgs), ctypes.c_int, vp]),
    'epos_conv3x3_f32': (ctypes.c_int, [ctypes.POINTER(Conv3x3Args), ctypes.c_void_p]),
    'epos_depthwise3x3_f32': (ctypes.c_int,
                              [ctypes.POINTER(DepthwiseArgs), vp]),
    'epos_separable_conv_f32': (ctypes.c_int, [ctypes.POINTER(SepConvArgs), vp]),
    'epos_set_h2_narrow_tile_limit': (ctypes.c_int, [ctypes.c_int]),
    'epos_im2col3x3_f32': (ctypes.c_int, [ctypes.POINTER(Im2colArgs), vp]),
    'epos_global_avg_pool_partial_f32': (ctypes.c_int, [
        vp, ctypes.c_int64, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp]),
    'epos_global_avg_pool_f32': (ctypes.c_int, [
        vp, ctypes.c_int64, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp]),
    'epos_resize_bilinear_f32': (ctypes.c_int, [
        vp, ctypes.c_int64, vp, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
        ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp]),
    'epos_maxpool3x3_s2_f32': (ctypes.c_int, [
        vp, ctypes.c_int64, vp, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
        ctypes.c_int, ctypes.c_int, vp]),
    'epos_subsample_f32': (ctypes.c_int, [
        vp, ctypes.c_int64, vp, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
        ctypes.c_int, ctypes.c_int, ctypes.c_int, vp]),
    'epos_add_relu_f32': (ctypes.c_int, [vp, vp, vp, ctypes.c_int64, vp]),
    'epos_softmax_groups_f32': (ctypes.c_int,
                                [vp, ctypes.c_int64, ctypes.c_int, vp]),
    'epos_u8_to_f32': (ctypes.c_int, [vp, vp, ctypes.c_int64, vp]),
    'epos_scatter_blocks_f32': (ctypes.c_int, [vp, vp, vp, ctypes.c_int64, ctypes.c_int, vp]),
    'epos_argmax_i64': (ctypes.c_int, [
        vp, ctypes.c_int64, vp, ctypes.c_int64, ctypes.c_int, vp]),
    'epos_softmax_slots_f32': (ctypes.c_int, [
        vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp]),
    'epos_corr_count': (ctypes.c_int, [
        vp, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
        ctypes.c_int, ctypes.c_float, ctypes.c_float, vp, vp, vp, vp, vp]),
    'epos_corr_fill': (ctypes.c_int, [
        vp, vp, vp, vp, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
        ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, vp, vp, vp,
        vp, ctypes.c_int64, ctypes.POINTER(CorrOut), vp, vp]),
    'epos_corr_slot_bases': (ctypes.c_int, [vp, ctypes.c_int, vp, vp]),
    'epos_project_to_mesh_f64': (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64,
        ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
        ctypes.c_void_p]),
    'epos_fragmentation_fps': (ctypes.c_int, [
        vp, ctypes.c_int64, ctypes.c_int, vp, vp, vp, vp, vp]),
    'epos_fit_params_default': (None, [ctypes.POINTER(FitParams)]),
    'epos_find6d_poses': (ctypes.c_int, [
        vp, vp, ctypes.c_int64, vp, ctypes.POINTER(FitParams), ctypes.c_uint64,
        vp, vp, vp, ctypes.c_int32]),
    'epos_fit_workspace_bytes': (ctypes.c_int64, [
        ctypes.c_int, ctypes.c_int64, ctypes.POINTER(FitParams),
        ctypes.c_int32]),
    'epos_find6d_poses_device': (ctypes.c_int, [
        vp, vp, vp, ctypes.c_int, ctypes.c_int64, vp, vp, vp,
        ctypes.POINTER(FitParams), ctypes.c_int32, vp, vp, vp, vp, vp, vp]),
    'epos_pnp_ransac_params_default': (None, [ctypes.POINTER(PnpRansacParams)]),
    'epos_solve_pnp_ransac': (ctypes.c_int, [
        vp, vp, ctypes.c_int64, vp, ctypes.POINTER(PnpRansacParams), vp, vp, vp]),
    'epos_pnp_ransac_workspace_bytes': (ctypes.c_int64, [
        ctypes.c_int, ctypes.c_int64, ctypes.POINTER(PnpRansacParams)]),
    'epos_solve_pnp_ransac_device': (ctypes.c_int, [
        vp, vp, vp, ctypes.c_int, ctypes.c_int64, vp, ctypes.POINTER(PnpRansacParams),
        vp, vp, vp, vp, vp, vp]),
}

_lib = None


def lib_path():
  # EPOS_HIP_LIB: another build of the same library (A/B runs of kernel variants on one
  # box, tools/); the default is the in-tree build
  return os.environ.get('EPOS_HIP_LIB') or _build.LIB_PATH


def load_ref():
  """The TEST build of the library: the product library + the fp32-MFMA reference GEMM kernels
  (epos_amd/build.py: build_ref). A second, independent handle -- the accuracy tests run the
  product kernels and the reference kernels through it side by side. Not used by the product
  path (a process that wants its whole plan on the reference kernels sets
  EPOS_HIP_LIB=<libepos_hip_ref.so> and EPOS_GEMM_SPLIT=0 instead)."""
  global _ref
  if _ref is None:
    _ref = _open(_build.REF_LIB_PATH)
  return _ref


_ref = None


def load():
  """Loads libepos_hip.so (never builds implicitly on a GPU box: the library must
  have been built by __graft_entry__.build() / python -m epos_amd.build)."""
  global _lib
  if _lib is not None:
    return _lib
  _lib = _open(lib_path())
  return _lib


def _open(path):
  # PyTorch's HIP runtime has to be the first one mapped into the process: loading
  # this library (linked against /opt/rocm's libamdhip64) BEFORE torch initialises
  # leaves the process with a runtime that reports "no ROCm-capable device" to
  # whichever side came second (seen with build() and smoke() in one interpreter).
  import torch  # noqa: F401
  if not os.path.exists(path):
    raise EposError(
        'libepos_hip.so not found at %s -- build it with '
        '`python -m epos_amd.build` (there is no CPU fallback).' % path)
  lib = ctypes.CDLL(path)
  for name, (restype, argtypes) in SYMBOLS.items():
    fn = getattr(lib, name)           # AttributeError if a symbol is missing
    fn.restype = restype
    if argtypes is not None:
      fn.argtypes = argtypes
  if lib.epos_abi_version() != 7:
    raise EposError('libepos_hip.so ABI version mismatch')
  return lib


AMAX_WORDS = 64          # EPOS_AMAX_WORDS: uint32 words per absmax slot


def check(rc, what='', lib=None):
  if rc < 0:
    msg = (lib or load()).epos_last_error().decode('utf-8', 'replace')
    raise EposError('%s failed (%d): %s' % (what or 'epos call', rc, msg))
  return rc
