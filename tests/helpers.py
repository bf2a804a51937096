"""Shared helpers of the test-suite."""
import pathlib

import numpy as np
import torch

GOLDEN = pathlib.Path(__file__).resolve().parent / 'golden'

# Tolerances.  north_star asks for codes within 1e-5 relative error of the
# reference with an identical support.  Measured on MI355X against the
# reference's own codes after 200 FISTA iterations
# (profiles/r02_precision_fc.txt): exact-f32 path 2.5e-6, f16x3 (the default
# of the fused kernel) 2.5e-6, both with 0 support flips; bf16x3 1.75e-5.
REL_TOL_F32 = 1e-5      # exact-f32 MFMA path and f16x3, 200 iterations
REL_TOL_BF16X3 = 3e-5   # bf16 hi/lo split (2^-17 per product), 200 iterations
REL_TOL_SHORT = 5e-6    # <= 50 iterations
REL_TOL_DICT = 2e-6     # one dictionary update
NEAR_THRESHOLD = 2e-6   # support flips are only tolerated this close to it


def load(name):
  return np.load(GOLDEN / (name + '.npz'))


def gaussian_patches(seed, b, n, scale=0.1):
  return (scale * np.random.RandomState(seed).randn(b, n)).astype(np.float32)


def unit_rows(seed, s, n):
  d = np.random.RandomState(seed).randn(s, n).astype(np.float32)
  return d / np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)


def rel_err(ours, ref):
  ours = np.asarray(ours, dtype=np.float64)
  ref = np.asarray(ref, dtype=np.float64)
  return float(np.linalg.norm(ours - ref) / max(np.linalg.norm(ref), 1e-30))


def support_mismatch(ours, ref):
  return int(((np.asarray(ours) != 0) != (np.asarray(ref) != 0)).sum())


def assert_codes_match(ours, ref, rel_tol, what, max_flip_mag=NEAR_THRESHOLD):
  """Relative l2 error below rel_tol and identical sparse support, except for
  entries whose magnitude (in whichever result is non-zero) is below
  max_flip_mag: those sat within rounding distance of the threshold."""
  ours = np.asarray(ours)
  ref = np.asarray(ref)
  assert ours.shape == ref.shape, what
  assert np.isfinite(ours).all(), what + ': non-finite values'
  err = rel_err(ours, ref)
  flips = (ours != 0) != (ref != 0)
  flip_mag = np.maximum(np.abs(ours), np.abs(ref))[flips]
  worst = float(flip_mag.max()) if flip_mag.size else 0.0
  assert err <= rel_tol, '%s: rel err %.3e > %.1e' % (what, err, rel_tol)
  assert worst <= max_flip_mag, (
      '%s: %d support flips, largest magnitude %.3e' % (
          what, int(flips.sum()), worst))
  return err, int(flips.sum())


def to_dev(a, device):
  return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def metrics_cases(g):
  """The three training runs behind tests/golden/metrics.npz
  (oracle/make_golden.py make_metrics): tag -> (params, train, validation,
  initial dictionary, batch size)."""
  fc = {
      'mode': 'fully-connected', 'num_epochs': 1,
      'code_inference_algorithm': 'fista',
      'inference_param_schedule': {
          0: {'sparsity_weight': 0.02, 'num_iters': 15}},
      'dictionary_update_algorithm': 'sc_steepest_descent',
      'dict_update_param_schedule': {0: {'stepsize': 0.1, 'num_iters': 1}},
      'training_visualization_schedule': set([0, 2])}
  sub = dict(fc)
  sub.update({'code_inference_algorithm': 'subspace_fista',
              'dictionary_update_algorithm':
                  'subspace_sc_cheap_quadratic_descent',
              'group_assignments': [list(range(4 * k, 4 * k + 4))
                                    for k in range(32)],
              'subspace_alignment_penalty': 2e-4})
  pad = tuple(tuple(int(v) for v in row) for row in g['conv_padding'])
  conv = {
      'mode': 'convolutional', 'num_epochs': 1,
      'code_inference_algorithm': 'ista', 'strides': (4, 4), 'padding': pad,
      'inference_param_schedule': {
          0: {'sparsity_weight': 0.05, 'num_iters': 8}},
      'dictionary_update_algorithm': 'sc_steepest_descent',
      'dict_update_param_schedule': {0: {'stepsize': 0.005, 'num_iters': 1}},
      'training_visualization_schedule': set([0, 2])}
  return {
      'fc': (fc, g['fc_images'], g['fc_validation'], g['fc_dictionary0'], 32),
      'sub': (sub, g['fc_images'], g['fc_validation'], g['fc_dictionary0'],
              32),
      'conv': (conv, g['conv_images_padded'], g['conv_validation'],
               g['conv_dictionary0'], 2)}
