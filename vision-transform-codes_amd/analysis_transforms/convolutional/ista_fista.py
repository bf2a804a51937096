"""
Convolutional ISTA / FISTA sparse inference on MI355X.

Drop-in for vision_transform_codes/analysis_transforms/convolutional/
ista_fista.py:18-197.  Synthesis (transposed convolution of the code maps),
masked residual, analysis (correlation with the kernels), shrinkage and the
FISTA extrapolation are two HIP kernels per iteration; no im2col buffer and no
mask tensor are materialised.
"""
import ctypes

import torch

import vtc_hip
from utils import convolutions


def run(images_padded, dictionary, kernel_stride, padding_dims,
        sparsity_weight, num_iters, variant='fista', initial_codes=None,
        early_stopping_epsilon=None, nonnegative_only=False,
        hard_threshold=False, stepsize=None, precision=None):
  """
  images_padded (b, c, h, w), dictionary (s, c, kh, kw): float32 on a HIP
  device.  kernel_stride (sv, sh); padding_dims ((lead_v, trail_v),
  (lead_h, trail_h)) or None.  Returns codes (b, s, code_h, code_w).
  Extensions: `stepsize` (skip the eigen-solve) and `precision` in {None,
  'auto', 'f32', 'f16x3', 'bf16x3'}.  The split modes run both convolutions as
  hi/lo split contractions on the matrix cores (stride 1, square kernels of
  5/8/11/16, one to four image channels): 'f16x3' in power-of-two scaled units with 11 + 11
  significand bits (float32-level results: 4e-6 from the reference's codes at
  T = 100, profiles/r03_precision_conv.txt), 'bf16x3' with 8 + 8 (1.8e-5).
  'auto' picks f16x3 where it applies and the direct f32 kernels elsewhere.
  """
  assert variant in ['ista', 'fista']
  lib = vtc_hip.load_library()
  images_padded = vtc_hip.require_device_tensor(
      images_padded, 'images_padded').contiguous()
  dictionary = vtc_hip.require_device_tensor(
      dictionary, 'dictionary').contiguous()
  device = images_padded.device
  geom = convolutions.geometry(images_padded, dictionary, kernel_stride,
                               padding_dims)
  code_h = convolutions.code_dim_from_padded_img_dim(geom.h, geom.kh,
                                                     geom.stride_v)
  code_w = convolutions.code_dim_from_padded_img_dim(geom.w, geom.kw,
                                                     geom.stride_h)
  if initial_codes is not None:
    initial_codes = vtc_hip.require_device_tensor(
        initial_codes, 'initial_codes').contiguous()
    assert initial_codes.shape[0] == images_padded.shape[0]
    assert initial_codes.shape[1] == dictionary.shape[0]
    assert initial_codes.shape[2] == code_h
    assert initial_codes.shape[3] == code_w
  if num_iters < 1:
    raise UnboundLocalError(
        "local variable 'codes' referenced before assignment")
  if stepsize is None:
    # largest eigenvalue of the (s, s) Gram matrix of the flattened kernels
    flat = dictionary.reshape(dictionary.shape[0], -1)
    stepsize = vtc_hip.stepsize_from_gram(
        vtc_hip.gram(flat, transpose_a=False), dictionary)

  codes = torch.empty((geom.b, geom.s, code_h, code_w), dtype=torch.float32,
                      device=device)
  ws = vtc_hip.workspace(
      lib.vtc_conv_ista_fista_workspace_bytes(ctypes.byref(geom)), device)
  iters_run = ctypes.c_int(0)
  eps = -1.0 if early_stopping_epsilon is None else float(
      early_stopping_epsilon)
  name = precision if precision is not None else (
      vtc_hip.get_default_precision())
  if name == 'auto':
    # hard thresholds stay on the exact-f32 kernels: a discontinuous threshold
    # turns last-bit differences of the products into flips of the cutoff's
    # size, which the split paths are only pinned for over a single iteration
    name = 'f16x3' if (geom.s >= 32 and not hard_threshold and
                        lib.vtc_conv_x3_supported(ctypes.byref(geom))) else (
                            'f32')
  if name == 'bf16':
    raise NotImplementedError('convolutional inference has no bf16 fast mode')
  vtc_hip.check(lib.vtc_conv_ista_fista(
      vtc_hip.ptr(images_padded), vtc_hip.ptr(dictionary),
      vtc_hip.ptr(initial_codes), vtc_hip.ptr(codes), ctypes.byref(geom),
      float(stepsize), float(sparsity_weight), int(num_iters),
      vtc_hip.variant_code(variant),
      vtc_hip.threshold_mode(nonnegative_only, hard_threshold), eps,
      vtc_hip.PRECISIONS[name], vtc_hip.ptr(ws), ws.numel(), ctypes.byref(iters_run),
      vtc_hip.current_stream(device)), 'vtc_conv_ista_fista')
  run.last_iters = iters_run.value
  return codes
