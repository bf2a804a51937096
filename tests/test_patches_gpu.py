"""Device whitening + patch extraction against
the patches the reference's whiten_center_surround produced
(tests/golden/whitened.npz, oracle/make_golden.py make_whitened)."""
import numpy as np
import pytest
import torch

import helpers

pytestmark = pytest.mark.gpu


def _synthetic_images_and_positions():
  """Same generator state walk as make_whitened: two 1/f images, then 64
  (vertical, horizontal) positions per image."""
  rs = np.random.RandomState(50)
  size = 128
  fy = np.fft.fftfreq(size)[:, None]
  fx = np.fft.fftfreq(size)[None, :]
  amp = 1. / np.maximum(np.sqrt(fy**2 + fx**2), 1. / size)
  imgs = []
  for _ in range(2):
    spec = amp * np.exp(2j * np.pi * rs.rand(size, size))
    img = np.real(np.fft.ifft2(spec))
    img = (img - img.min()) / (img.max() - img.min())
    imgs.append(img.astype(np.float32))
  idx, vert, horz = [], [], []
  for i in range(2):
    for _ in range(64):
      y, x = rs.randint(5, size - 21, size=2)
      idx.append(i), vert.append(y), horz.append(x)
  return np.stack(imgs)[:, :, :, None], idx, vert, horz


def test_whitened_patches_match_reference(device):
  from utils import image_processing, dataset_generation
  g = helpers.load('whitened')
  imgs, idx, vert, horz = _synthetic_images_and_positions()
  dev_imgs = helpers.to_dev(imgs, device)
  white = image_processing.whiten_center_surround(
      dev_imgs, {'low': 1e-3, 'high': 0.9}, norm_and_threshold=False)
  single = image_processing.whiten_center_surround(
      dev_imgs[1], {'low': 1e-3, 'high': 0.9}, norm_and_threshold=False)
  assert torch.equal(single, white[1])
  patches = dataset_generation.extract_patches(white, idx, vert, horz,
                                               (16, 16))
  # float64 transform then a cast to float32 on both sides: 1e-6 relative
  assert patches.shape == g['images'].shape
  assert helpers.rel_err(patches.cpu().numpy(), g['images']) < 1e-6
  # the reference function's default, norm_and_threshold=True: transfer
  # function divided by its maximum over the grid, floored at 1e-3
  white_nat = image_processing.whiten_center_surround(
      dev_imgs, {'low': 1e-3, 'high': 0.9})
  patches = dataset_generation.extract_patches(white_nat, idx, vert, horz,
                                               (16, 16))
  assert helpers.rel_err(patches.cpu().numpy(),
                         g['images_norm_and_threshold']) < 1e-6
  with pytest.raises(NotImplementedError):
    image_processing.whiten_center_surround(dev_imgs, {'low': 1e-3,
                                                       'high': 0.9},
                                            return_filter=True)
  with pytest.raises(IndexError):
    dataset_generation.extract_patches(white, [0], [120], [0], (16, 16))


def test_patch_extraction_layout_and_channels(device):
  from utils import dataset_generation
  rs = np.random.RandomState(3)
  imgs = rs.randn(3, 20, 31, 2).astype(np.float32)
  idx, vert, horz = dataset_generation.draw_patch_positions(
      40, (20, 31), (5, 7), 2, 3, rng=np.random.RandomState(9))
  got = dataset_generation.extract_patches(helpers.to_dev(imgs, device), idx,
                                           vert, horz, (5, 7), flatten=False)
  want = np.stack([imgs[i, y:y + 5, x:x + 7] for i, y, x in
                   zip(idx, vert, horz)])
  assert np.array_equal(got.cpu().numpy(), want)
  # the reference's draw order (image, vertical, horizontal per patch)
  rng = np.random.RandomState(9)
  for p in range(40):
    assert idx[p] == rng.randint(low=0, high=3)
    assert vert[p] == rng.randint(low=2, high=20 - 5 - 2)
    assert horz[p] == rng.randint(low=2, high=31 - 7 - 2)


def test_standardize_data_range(device):
  """dataset_generation.py:169-183: (x - min) / (max - min) over the whole
  stack in float32, bit for bit numpy's; a constant stack trips the
  reference's assert."""
  from utils import dataset_generation
  rs = np.random.RandomState(4)
  imgs = (255.0 * rs.rand(3, 37, 53, 2)).astype(np.float32) - 17.0
  lo, hi = np.min(imgs), np.max(imgs)
  want = (imgs - lo) / (hi - lo)
  got = dataset_generation.standardize_data_range(helpers.to_dev(imgs, device))
  assert got.shape == imgs.shape and np.array_equal(got.cpu().numpy(), want)
  assert float(got.min()) == 0.0 and float(got.max()) == 1.0
  with pytest.raises(AssertionError):
    dataset_generation.standardize_data_range(
        torch.full((4, 4), 2.5, device=device))
