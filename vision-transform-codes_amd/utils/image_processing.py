"""
Image whitening on MI355X: the step before the sparse-coding path.

Device counterpart of whiten_center_surround in vision_transform_codes/utils/
image_processing.py:267-308 (filter_fd :63-92, get_low_pass_filter :173-231,
get_whitening_ramp_filter :234-264): rolled-off ramp times an order-8
exponential low-pass, applied in the frequency domain in float64 (hipFFT) and
returned as float32, as the reference does with numpy.
"""
import torch

import vtc_hip


def whiten_center_surround(image, cutoffs, return_filter=False,
                           norm_and_threshold=True):
  """
  image : float32 tensor on a HIP device, (h, w, c) like the reference, or a
      stack (count, h, w, c) of equally sized images (an extension: one
      batched transform).
  cutoffs : {'low': ..., 'high': ...} as in the reference.
  norm_and_threshold : as in the reference (default True: the transfer
      function is divided by its maximum and floored at 1e-3; the dataset
      pipeline, dataset_generation.py:231-238, passes False).
  Returns the filtered image(s), same shape.  return_filter=True is host-side
  debugging output and not implemented on the device.
  """
  if return_filter:
    raise NotImplementedError('return_filter is host-side debugging output')
  lib = vtc_hip.load_library()
  image = vtc_hip.require_device_tensor(image, 'image').contiguous()
  assert image.dim() in (3, 4), 'expected (h, w, c) or (count, h, w, c)'
  stacked = image if image.dim() == 4 else image[None]
  count, h, w, c = stacked.shape
  out = torch.empty_like(stacked)
  ws = vtc_hip.workspace(
      lib.vtc_whiten_center_surround_workspace_bytes(count, h, w, c),
      image.device)
  vtc_hip.check(lib.vtc_whiten_center_surround(
      vtc_hip.ptr(stacked), vtc_hip.ptr(out), count, h, w, c,
      float(cutoffs['low']), float(cutoffs['high']),
      1 if norm_and_threshold else 0, vtc_hip.ptr(ws), ws.numel(),
      vtc_hip.current_stream(image.device)), 'vtc_whiten_center_surround')
  return out if image.dim() == 4 else out[0]
