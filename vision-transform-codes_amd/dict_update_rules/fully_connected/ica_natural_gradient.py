"""
ICA natural-gradient dictionary update on MI355X.

Drop-in for vision_transform_codes/dict_update_rules/fully_connected/
ica_natural_gradient.py:6-35.  In a data-parallel run (vtc_hip.parallel.enable)
the (s, s) moment codes^T sign(codes) is all-reduced over ranks and divided by
the global batch before it is applied.
"""
import torch

import vtc_hip
from vtc_hip import parallel


def run(dictionary, codes, stepsize=0.001, num_iters=1):
  """
  D <- D + stepsize * ((C^T sign(C)) / b - I) D, num_iters times (the
  reference ascends the gradient; no images are needed).

  dictionary (s, n) [updated IN PLACE], codes (b, s): float32 tensors on a HIP
  device.  Returns None.
  """
  lib = vtc_hip.load_library()
  dictionary = vtc_hip.require_device_tensor(dictionary, 'dictionary')
  assert dictionary.is_contiguous(), 'dictionary is updated in place'
  codes = vtc_hip.require_device_tensor(codes, 'codes').contiguous()
  b, s = codes.shape
  assert dictionary.shape[0] == s
  n = dictionary.shape[1]
  device = dictionary.device
  stream = vtc_hip.current_stream(device)
  moment = torch.empty((s, s), dtype=torch.float32, device=device)
  ws = vtc_hip.workspace(max(lib.vtc_ica_moment_workspace_bytes(b, s),
                             lib.vtc_ica_apply_workspace_bytes(s, n)), device)
  # the moment does not depend on the dictionary: once for all iterations
  vtc_hip.check(lib.vtc_ica_moment(
      vtc_hip.ptr(codes), vtc_hip.ptr(moment), b, s, vtc_hip.ptr(ws),
      ws.numel(), stream), 'vtc_ica_moment')
  parallel.all_reduce_sum_(moment)
  total = parallel.global_batch(b, device)
  for _ in range(num_iters):
    vtc_hip.check(lib.vtc_ica_apply(
        vtc_hip.ptr(dictionary), vtc_hip.ptr(moment), total, s, n,
        float(stepsize), vtc_hip.ptr(ws), ws.numel(), stream),
        'vtc_ica_apply')
