"""
Geometry helpers of the convolutional plugins (host side, pure Python).

Same three entry points as vision_transform_codes/utils/convolutions.py:7-24.
create_mask is provided for callers of the reference API; the HIP kernels
evaluate the mask analytically from padding_dims and never materialise it.
"""
import ctypes
import math

import torch

import vtc_hip


def get_padding_amt(image_dim, kernel_dim, dim_stride):
  """(leading, trailing) zero padding so that strided kernels tile the axis."""
  lead = trail = kernel_dim - dim_stride
  remainder = image_dim % dim_stride
  if remainder != 0:
    trail += dim_stride - remainder
  return lead, trail


def code_dim_from_padded_img_dim(padded_image_dim, kernel_dim, dim_stride):
  return 1 + int(math.ceil((padded_image_dim - kernel_dim) / dim_stride))


def create_mask(images_with_padding, padding):
  """Ones over the un-padded image, zeros over the padding frame."""
  mask = torch.ones_like(images_with_padding)
  if padding is not None:
    (lead_v, trail_v), (lead_h, trail_h) = padding
    mask[:, :, 0:lead_v, :] = 0.0
    mask[:, :, -trail_v:, :] = 0.0
    mask[:, :, :, 0:lead_h] = 0.0
    mask[:, :, :, -trail_h:] = 0.0
  return mask


def geometry(images_padded, dictionary, kernel_stride, padding_dims):
  """Fill the vtc_conv_geometry struct of include/vtc_hip.h."""
  g = vtc_hip.ConvGeometry()
  g.b, g.c, g.h, g.w = [int(v) for v in images_padded.shape]
  g.s, c2, g.kh, g.kw = [int(v) for v in dictionary.shape]
  assert c2 == g.c, 'dictionary and images disagree on channel count'
  g.stride_v, g.stride_h = int(kernel_stride[0]), int(kernel_stride[1])
  if padding_dims is None:
    g.has_padding = 0
  else:
    g.has_padding = 1
    g.pad_lead_v, g.pad_trail_v = [int(v) for v in padding_dims[0]]
    g.pad_lead_h, g.pad_trail_h = [int(v) for v in padding_dims[1]]
  return g
