"""
The patch pipeline of utils/dataset_generation.py on MI355X: range
standardisation (:169-183), patch positions (:205-214) and extraction
(:184-222).  Images stay in HBM; the positions come from the caller's numpy
generator in the reference's order (image index, vertical, horizontal per
patch), so a seeded run selects the same patches.
"""
import ctypes

import numpy as np
import torch

import vtc_hip


def standardize_data_range(images):
  """(images - min) / (max - min) over the whole stack, float32 like numpy's
  (dataset_generation.py:169-183; asserts max > min as the reference does).
  images: float32 tensor of any shape on a HIP device; returns a new tensor."""
  lib = vtc_hip.load_library()
  images = vtc_hip.require_device_tensor(images, 'images').contiguous()
  out = torch.empty_like(images)
  min_max = torch.empty(2, dtype=torch.float32, device=images.device)
  ws = vtc_hip.workspace(lib.vtc_window_minmax_workspace_bytes(),
                         images.device)
  vtc_hip.check(lib.vtc_standardize_data_range(
      vtc_hip.ptr(images), vtc_hip.ptr(out), images.numel(),
      vtc_hip.ptr(min_max), vtc_hip.ptr(ws), ws.numel(),
      vtc_hip.current_stream(images.device)), 'vtc_standardize_data_range')
  lo, hi = min_max.tolist()
  assert hi > lo
  return out


def _position_limits(image_shape, patch_dimensions, edge_buffer, num_images):
  shapes = [tuple(image_shape)[:2]] * num_images if np.isscalar(
      image_shape[0]) else [tuple(x)[:2] for x in image_shape]
  assert len(shapes) == num_images
  max_vert = [x[0] - patch_dimensions[0] - edge_buffer for x in shapes]
  max_horz = [x[1] - patch_dimensions[1] - edge_buffer for x in shapes]
  return max_vert, max_horz


def draw_patch_positions_loop(num_samples, image_shape, patch_dimensions,
                              edge_buffer, num_images, rng=np.random):
  """The reference's loop as it stands: three randint calls per patch
  (dataset_generation.py:205-214).  Kept as the statement of what
  draw_patch_positions must reproduce (tests/test_host_logic.py) and for
  generators that are not numpy's legacy RandomState."""
  max_vert, max_horz = _position_limits(image_shape, patch_dimensions,
                                        edge_buffer, num_images)
  img_idx = np.empty(num_samples, np.int32)
  vert = np.empty(num_samples, np.int32)
  horz = np.empty(num_samples, np.int32)
  for p_idx in range(num_samples):
    img_idx[p_idx] = rng.randint(low=0, high=num_images)
    vert[p_idx] = rng.randint(low=edge_buffer, high=max_vert[img_idx[p_idx]])
    horz[p_idx] = rng.randint(low=edge_buffer, high=max_horz[img_idx[p_idx]])
  return img_idx, vert, horz


def draw_patch_positions(num_samples, image_shape, patch_dimensions,
                         edge_buffer, num_images, rng=np.random):
  """The three randint values per patch of dataset_generation.py:205-214, for
  all patches in one native call (vtc_draw_patch_positions: MT19937 + randint's
  masked rejection sampling on the generator's own state, which is advanced
  exactly as the loop would advance it): 131 072 positions in ~2 ms instead of
  0.3-0.4 s of interpreter time.  image_shape: one (h, w) for equally sized
  images, or a sequence of num_images shapes -- the reference keeps per-image
  position ranges (:185-198).  rng: numpy.random (the global legacy generator)
  or a numpy.random.RandomState; anything else takes the loop.  Returns int32
  arrays (img_idx, vert_pos, horz_pos)."""
  state_owner = rng.mtrand._rand if rng is np.random else rng
  if not isinstance(state_owner, np.random.RandomState):
    return draw_patch_positions_loop(num_samples, image_shape,
                                     patch_dimensions, edge_buffer,
                                     num_images, rng)
  max_vert, max_horz = _position_limits(image_shape, patch_dimensions,
                                        edge_buffer, num_images)
  kind, key, pos, has_gauss, cached = state_owner.get_state()
  assert kind == 'MT19937'
  key = np.ascontiguousarray(key, dtype=np.uint32).copy()
  pos_c = ctypes.c_int32(int(pos))
  mv = np.asarray(max_vert, dtype=np.int32)
  mh = np.asarray(max_horz, dtype=np.int32)
  img_idx = np.empty(num_samples, np.int32)
  vert = np.empty(num_samples, np.int32)
  horz = np.empty(num_samples, np.int32)

  def p(a):
    return ctypes.c_void_p(a.ctypes.data)
  vtc_hip.check(vtc_hip.load_library().vtc_draw_patch_positions(
      p(key), ctypes.cast(ctypes.byref(pos_c), ctypes.c_void_p),
      int(num_samples), int(num_images), int(edge_buffer), p(mv), p(mh),
      p(img_idx), p(vert), p(horz)), 'vtc_draw_patch_positions')
  state_owner.set_state((kind, key, pos_c.value, has_gauss, cached))
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
