"""The production pipeline end to end at the headline shape (16x16 patches,
1024 atoms) against a trajectory recorded from the reference's own
train_dictionary (tests/golden/trainer_c2.npz, oracle/make_golden.py
make_trainer_c2): Gram -> Lanczos eta -> fused f16x3 FISTA -> code energy ->
Hessian EMA -> dictionary gradient -> apply, with the DEFAULT precision policy
and NO injected stepsize."""
import numpy as np
import pytest
import torch

import helpers

pytestmark = pytest.mark.gpu


def _params():
  return {
      'mode': 'fully-connected', 'num_epochs': 1,
      'code_inference_algorithm': 'fista',
      'inference_param_schedule': {
          0: {'sparsity_weight': 0.008, 'num_iters': 50}},
      'dictionary_update_algorithm': 'sc_cheap_quadratic_descent',
      'dict_update_param_schedule': {0: {'stepsize': 0.1, 'num_iters': 1}}}


def _inputs(g):
  X = helpers.gaussian_patches(int(g['patch_seed']), 192, 256)
  D0 = helpers.unit_rows(int(g['dict_seed']), 1024, 256)
  return X, D0


def test_default_policy_is_the_fused_kernel_at_the_headline_shape(device):
  import vtc_hip
  from analysis_transforms.fully_connected import ista_fista
  assert vtc_hip.get_default_precision() == 'auto'
  assert ista_fista._resolve_precision(None, 64, 256, 1024, None) == (
      vtc_hip.F16X3)


def test_own_eta_matches_the_reference(device):
  """eta from vtc_gram + vtc_lambda_max against the eigvalsh-based eta the
  reference computed for the same dictionaries."""
  import vtc_hip
  g = helpers.load('trainer_c2')
  _, D0 = _inputs(g)
  for D, want in ((D0, g['eta'][0]), (g['dict_after_step1'], g['eta'][1])):
    Dd = helpers.to_dev(D, device)
    eta = vtc_hip.stepsize_from_gram(vtc_hip.gram(Dd, transpose_a=True), Dd)
    assert abs(float(eta) - float(want)) <= 2e-6 * float(want)


def test_first_step_codes(device):
  from analysis_transforms.fully_connected import ista_fista
  g = helpers.load('trainer_c2')
  X, D0 = _inputs(g)
  codes = ista_fista.run(helpers.to_dev(X[:64], device),
                         helpers.to_dev(D0, device), 0.008, 50,
                         variant='fista')
  helpers.assert_codes_match(codes.cpu().numpy(), g['codes_step1'],
                             helpers.REL_TOL_SHORT, 'pipeline codes step 1')


def test_three_step_trajectory(device):
  """Dictionary after steps 1 and 3 (full arrays), step 2 (row / column sums)
  and the Hessian diagonal, default precision, own eta."""
  from training import sparse_coding
  g = helpers.load('trainer_c2')
  X, D0 = _inputs(g)
  Xd = helpers.to_dev(X, device)
  seen = {}
  for steps in (1, 2, 3):
    D = helpers.to_dev(D0.copy(), device)
    batches = [Xd[64 * i: 64 * i + 64] for i in range(steps)]
    state = sparse_coding.train_dictionary(batches, batches, D, _params())
    seen[steps] = D.cpu().numpy()
  # one update moves a unit-norm dictionary by ~1e-3 relative; 2e-6 on the
  # whole dictionary = 2e-3 of the update itself
  assert helpers.rel_err(seen[1], g['dict_after_step1']) < helpers.REL_TOL_DICT
  assert helpers.rel_err(seen[3], g['dict_after_step3']) < 3 * helpers.REL_TOL_DICT
  rows = seen[2].astype(np.float64).sum(axis=1)
  cols = seen[2].astype(np.float64).sum(axis=0)
  assert np.abs(rows - g['dict_after_step2_rowsum']).max() < 2e-5
  assert np.abs(cols - g['dict_after_step2_colsum']).max() < 2e-5
  assert helpers.rel_err(state.hessian_diag.cpu().numpy(),
                         g['hessian_after_step3']) < 2e-5
  # the update really moved the dictionary (the tolerance above is not vacuous)
  assert helpers.rel_err(seen[1], D0) > 1e-4
