"""Shared driver of the fully-connected dictionary-update plugins."""
import torch

import vtc_hip
from vtc_hip import parallel


def descend(images, dictionary, codes, stepsize, num_iters,
            normalize_dictionary, hessian_diagonal=None, lowest_code_val=0.001,
            penalty=None):
  """num_iters x { G = C^T(C D - X); [all-reduce G]; D -= step; normalise }.

  penalty: None or (alignment_penalty, callable returning the (s, n) alignment
  gradient of the current dictionary).
  Mutates `dictionary` in place and returns None, like the reference
  (dict_update_rules/fully_connected/sc_steepest_descent.py:37-41).
  """
  lib = vtc_hip.load_library()
  images = vtc_hip.require_device_tensor(images, 'images').contiguous()
  codes = vtc_hip.require_device_tensor(codes, 'codes').contiguous()
  vtc_hip.require_device_tensor(dictionary, 'dictionary')
  if not dictionary.is_contiguous():
    raise ValueError('dictionary must be contiguous: it is updated in place')
  b, n = images.shape
  s = dictionary.shape[0]
  assert tuple(codes.shape) == (b, s) and dictionary.shape[1] == n
  if hessian_diagonal is not None:
    hessian_diagonal = vtc_hip.require_device_tensor(
        hessian_diagonal, 'hessian_diagonal').contiguous()
  device = images.device
  stream = vtc_hip.current_stream(device)
  ws = vtc_hip.workspace(lib.vtc_fc_dict_gradient_workspace_bytes(b, n, s),
                         device)
  grad_sum = parallel.take((s, n), device)
  total_batch = parallel.global_batch(b, device)
  for _ in range(num_iters):
    vtc_hip.check(lib.vtc_fc_dict_gradient(
        vtc_hip.ptr(images), vtc_hip.ptr(dictionary), vtc_hip.ptr(codes),
        vtc_hip.ptr(grad_sum), b, n, s, vtc_hip.ptr(ws), ws.numel(), stream),
        'vtc_fc_dict_gradient')
    parallel.all_reduce_sum_(grad_sum)
    pen_weight, pen_grad = 0.0, None
    if penalty is not None:
      pen_weight, pen_fn = penalty
      pen_grad = pen_fn()
    vtc_hip.check(lib.vtc_fc_dict_apply(
        vtc_hip.ptr(dictionary), vtc_hip.ptr(grad_sum),
        vtc_hip.ptr(hessian_diagonal), vtc_hip.ptr(pen_grad),
        float(pen_weight), total_batch, float(stepsize),
        float(lowest_code_val), 1 if normalize_dictionary else 0, s, n,
        stream), 'vtc_fc_dict_apply')
