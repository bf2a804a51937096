"""
Training-patch extraction on MI355X.

Device counterpart of the 'patch' operation in vision_transform_codes/utils/
dataset_generation.py:184-222: patches are cut out of (already whitened) images
that stay in HBM; the positions are drawn on the host with the caller's numpy
generator in the reference's order (image index, vertical, horizontal per
patch), so a seeded run selects the same patches.
"""
import numpy as np
import torch

import vtc_hip


def draw_patch_positions(num_samples, image_shape, patch_dimensions,
                         edge_buffer, num_images, rng=np.random):
  """The three randint calls per patch of dataset_generation.py:205-214.
  image_shape: one (h, w) for equally sized images, or a sequence of
  num_images shapes -- the reference keeps per-image position ranges
  (:185-198).  Returns int32 arrays (img_idx, vert_pos, horz_pos)."""
  shapes = [tuple(image_shape)[:2]] * num_images if np.isscalar(
      image_shape[0]) else [tuple(x)[:2] for x in image_shape]
  assert len(shapes) == num_images
  max_vert = [x[0] - patch_dimensions[0] - edge_buffer for x in shapes]
  max_horz = [x[1] - patch_dimensions[1] - edge_buffer for x in shapes]
  img_idx = np.empty(num_samples, np.int32)
  vert = np.empty(num_samples, np.int32)
  horz = np.empty(num_samples, np.int32)
  for p_idx in range(num_samples):
    img_idx[p_idx] = rng.randint(low=0, high=num_images)
    vert[p_idx] = rng.randint(low=edge_buffer, high=max_vert[img_idx[p_idx]])
    horz[p_idx] = rng.randint(low=edge_buffer, high=max_horz[img_idx[p_idx]])
  return img_idx, vert, horz


def extract_patches(images, img_idx, vert_pos, horz_pos, patch_dimensions,
                    flatten=True):
  """
  images : (count, h, w, c) float32 on a HIP device.
  img_idx, vert_pos, horz_pos : integer arrays (numpy or tensors), one entry
      per patch.
  Returns (num, ph*pw*c) if flatten else (num, ph, pw, c): patch p is
  images[img_idx[p], vert:vert+ph, horz:horz+pw, :].
  """
  lib = vtc_hip.load_library()
  images = vtc_hip.require_device_tensor(images, 'images').contiguous()
  assert images.dim() == 4, 'expected (count, h, w, c)'
  count, h, w, c = images.shape
  ph, pw = int(patch_dimensions[0]), int(patch_dimensions[1])
  device = images.device

  def as_i32(a):
    a = np.asarray(a.cpu() if torch.is_tensor(a) else a)
    return torch.from_numpy(a.astype(np.int32)).to(device)

  idx_np = np.asarray(img_idx.cpu() if torch.is_tensor(img_idx) else img_idx)
  v_np = np.asarray(vert_pos.cpu() if torch.is_tensor(vert_pos) else vert_pos)
  h_np = np.asarray(horz_pos.cpu() if torch.is_tensor(horz_pos) else horz_pos)
  num = int(idx_np.shape[0])
  if num:
    if (idx_np.min() < 0 or idx_np.max() >= count or v_np.min() < 0 or
        v_np.max() + ph > h or h_np.min() < 0 or h_np.max() + pw > w):
      raise IndexError('patch position outside the image stack')
  patches = torch.empty((num, ph * pw * c), dtype=torch.float32, device=device)
  # keep the index tensors alive until the launch is enqueued
  idx_d, vert_d, horz_d = as_i32(idx_np), as_i32(v_np), as_i32(h_np)
  vtc_hip.check(lib.vtc_extract_patches(
      vtc_hip.ptr(images), vtc_hip.ptr(idx_d), vtc_hip.ptr(vert_d),
      vtc_hip.ptr(horz_d), vtc_hip.ptr(patches), num, h, w, c, ph, pw,
      vtc_hip.current_stream(device)), 'vtc_extract_patches')
  return patches if flatten else patches.reshape(num, ph, pw, c)
