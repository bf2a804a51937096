"""Shared driver of the convolutional dictionary-update plugins."""
import ctypes

import torch

import vtc_hip
from vtc_hip import parallel
from utils import convolutions


def descend(images_padded, dictionary, codes, kernel_stride, padding_dims,
            stepsize, num_iters, normalize_dictionary, hessian_diagonal=None,
            lowest_code_val=0.001):
  """num_iters x { G = sum_b,p,q codes * masked residual; [all-reduce G];
  optional Hessian divide; global rescale to ||D||_F; D -= step; per-kernel
  normalise } -- dict_update_rules/convolutional/sc_steepest_descent.py:54-72.
  In a data-parallel run the Frobenius rescale is taken AFTER the all-reduce.
  """
  lib = vtc_hip.load_library()
  images_padded = vtc_hip.require_device_tensor(
      images_padded, 'images_padded').contiguous()
  codes = vtc_hip.require_device_tensor(codes, 'codes').contiguous()
  vtc_hip.require_device_tensor(dictionary, 'dictionary')
  if not dictionary.is_contiguous():
    raise ValueError('dictionary must be contiguous: it is updated in place')
  if hessian_diagonal is not None:
    hessian_diagonal = vtc_hip.require_device_tensor(
        hessian_diagonal, 'hessian_diagonal').contiguous()
  device = images_padded.device
  stream = vtc_hip.current_stream(device)
  geom = convolutions.geometry(images_padded, dictionary, kernel_stride,
                               padding_dims)
  s = geom.s
  kernel_elems = geom.c * geom.kh * geom.kw
  ws = vtc_hip.workspace(
      lib.vtc_conv_dict_gradient_workspace_bytes(ctypes.byref(geom)), device)
  grad_sum = parallel.take(tuple(dictionary.shape), device)
  scratch = torch.empty_like(dictionary)
  total_batch = parallel.global_batch(geom.b, device)
  # precision policy of the process (vtc_hip.set_default_precision): 'auto'
  # takes the matrix-core route where the inference plugin does (stride 1,
  # square kernels 5/8/11/16, at least 32 kernels).  The gradient is
  # a single product -- nothing iterates on its rounding -- and runs on the
  # bf16 split whichever split mode is named (5e-6 on the updated kernels)
  name = vtc_hip.get_default_precision()
  if name == 'auto':
    name = 'bf16x3' if (geom.s >= 32 and lib.vtc_conv_x3_supported(
        ctypes.byref(geom))) else 'f32'
  elif name != 'f32':
    name = 'bf16x3' if lib.vtc_conv_x3_supported(ctypes.byref(geom)) else (
        'f32')
  for _ in range(num_iters):
    vtc_hip.check(lib.vtc_conv_dict_gradient(
        vtc_hip.ptr(images_padded), vtc_hip.ptr(dictionary),
        vtc_hip.ptr(codes), vtc_hip.ptr(grad_sum), ctypes.byref(geom),
        vtc_hip.PRECISIONS[name], vtc_hip.ptr(ws), ws.numel(), stream),
        'vtc_conv_dict_gradient')
    parallel.all_reduce_sum_(grad_sum)
    vtc_hip.check(lib.vtc_conv_dict_apply(
        vtc_hip.ptr(dictionary), vtc_hip.ptr(grad_sum),
        vtc_hip.ptr(hessian_diagonal), total_batch, float(stepsize),
        float(lowest_code_val), 1 if normalize_dictionary else 0, s,
        kernel_elems, vtc_hip.ptr(scratch), stream), 'vtc_conv_dict_apply')
