"""The per-batch training step on the GPU against trajectories recorded from
the reference's own train_dictionary."""
import pathlib
import pickle

import numpy as np
import pytest
import torch

import helpers

pytestmark = pytest.mark.gpu


def _fc_params():
  return {
      'mode': 'fully-connected', 'num_epochs': 1,
      'code_inference_algorithm': 'fista',
      'inference_param_schedule': {
          0: {'sparsity_weight': 0.02, 'num_iters': 15},
          2: {'sparsity_weight': 0.01, 'num_iters': 30}},
      'dictionary_update_algorithm': 'sc_cheap_quadratic_descent',
      'dict_update_param_schedule': {
          0: {'stepsize': 0.1, 'num_iters': 1},
          2: {'stepsize': 0.05, 'num_iters': 2}}}


def test_fc_trajectory(device, tmp_path):
  from training import sparse_coding
  g = helpers.load('trainer')
  X = helpers.to_dev(g['fc_images'], device)
  for steps in (1, 2, 3):
    D = helpers.to_dev(g['fc_dictionary0'].copy(), device)
    batches = [X[32 * i: 32 * i + 32] for i in range(steps)]
    params = _fc_params()
    if steps == 3:
      params['checkpoint_schedule'] = {0, 2}
      params['logging_folder_fullpath'] = pathlib.Path(tmp_path)
    state = sparse_coding.train_dictionary(batches, batches, D, params)
    assert helpers.rel_err(D.cpu().numpy(),
                           g['fc_dict_after_step%d' % steps]) < 2e-5
  assert helpers.rel_err(state.hessian_diag.cpu().numpy(),
                         g['fc_hessian_after_step3']) < 2e-5
  # checkpoints are plain pickled numpy arrays, the reference's format
  with open(tmp_path / 'checkpoint_dictionary_iter_0', 'rb') as f:
    first = pickle.load(f)
  assert np.array_equal(first, g['fc_dictionary0'])
  assert (tmp_path / 'checkpoint_dictionary_iter_2').exists()


def test_conv_trajectory(device):
  from training import sparse_coding
  g = helpers.load('trainer')
  imgs = helpers.to_dev(g['conv_images_padded'], device)
  pad = tuple(tuple(int(v) for v in row) for row in g['conv_padding'])
  params = {
      'mode': 'convolutional', 'num_epochs': 1,
      'code_inference_algorithm': 'ista', 'strides': (4, 4), 'padding': pad,
      'inference_param_schedule': {0: {'sparsity_weight': 0.05,
                                       'num_iters': 8}},
      'dictionary_update_algorithm': 'sc_cheap_quadratic_descent',
      'dict_update_param_schedule': {0: {'stepsize': 0.005, 'num_iters': 1}}}
  for steps in (1, 3):
    K = helpers.to_dev(g['conv_dictionary0'].copy(), device)
    batches = [imgs[2 * i: 2 * i + 2] for i in range(steps)]
    sparse_coding.train_dictionary(batches, batches, K, params)
    assert helpers.rel_err(K.cpu().numpy(),
                           g['conv_dict_after_step%d' % steps]) < 2e-5


@pytest.mark.parametrize('tag', ['fc', 'sub', 'conv'])
def test_validation_metrics(device, tag):
  """training_visualization_schedule: the validation metrics (device
  reductions of vtc_hip) against the scalars the reference's train_dictionary
  sent to TensorBoard.  Tolerance 2e-5 relative (float32 sums in a different
  order; pSNR is a log of them)."""
  from training import sparse_coding
  g = helpers.load('metrics')
  params, train, val, D0, batch = helpers.metrics_cases(g)[tag]
  D = helpers.to_dev(D0.copy(), device)
  X, V = helpers.to_dev(train, device), helpers.to_dev(val, device)
  tb = [X[batch * i: batch * i + batch] for i in range(len(train) // batch)]
  vb = [V[batch * i: batch * i + batch] for i in range(len(val) // batch)]
  state = sparse_coding.train_dictionary(tb, vb, D, params)
  names = [str(x) for x in g[tag + '_names']]
  assert [it for it, _ in state.metrics_log] == [int(x)
                                                 for x in g[tag + '_steps']]
  for si, (_, got) in enumerate(state.metrics_log):
    assert sorted(got) == names
    for ni, name in enumerate(names):
      want = g[tag + '_values'][si, ni]
      assert abs(float(got[name]) - want) <= 2e-5 * max(abs(want), 1e-12), (
          tag, si, name, float(got[name]), want)
  assert helpers.rel_err(D.cpu().numpy(), g[tag + '_dictionary_final']) < 2e-5


def test_conv_unit_stride_training_step_on_the_matrix_core_path(device):
  """Stride-1 single-channel convolutional training: the default precision
  policy sends inference to the bf16x3 MFMA kernels (s >= 32); two training
  steps against the oracle's train_steps."""
  import sc_oracle
  from training import sparse_coding
  rs = np.random.RandomState(321)
  k, s_k, pad = 11, 32, 10
  imgs = np.zeros((4, 1, 40 + 2 * pad, 44 + 2 * pad), np.float32)
  imgs[:, :, pad:-pad, pad:-pad] = (0.5 * rs.randn(4, 1, 40, 44)).astype(
      np.float32)
  K0 = rs.randn(s_k, 1, k, k).astype(np.float32)
  K0 /= np.sqrt((K0.astype(np.float64) ** 2).sum(axis=(1, 2, 3)))[
      :, None, None, None].astype(np.float32)
  params = {
      'mode': 'convolutional', 'num_epochs': 1,
      'code_inference_algorithm': 'fista', 'strides': (1, 1),
      'padding': ((pad, pad), (pad, pad)),
      'inference_param_schedule': {0: {'sparsity_weight': 0.05,
                                       'num_iters': 4}},
      'dictionary_update_algorithm': 'sc_steepest_descent',
      'dict_update_param_schedule': {0: {'stepsize': 0.005, 'num_iters': 1}}}
  Kref = torch.from_numpy(K0.copy())
  sc_oracle.train_steps([torch.from_numpy(imgs[2 * i: 2 * i + 2])
                         for i in range(2)], Kref, params)
  K = helpers.to_dev(K0.copy(), device)
  X = helpers.to_dev(imgs, device)
  sparse_coding.train_dictionary([X[0:2], X[2:4]], [X[0:2]], K, params)
  assert helpers.rel_err(K.cpu().numpy(), Kref.numpy()) < 2e-5


def test_subspace_step_runs_and_keeps_unit_norm(device):
  from training import sparse_coding
  X = helpers.to_dev(helpers.gaussian_patches(80, 64, 64), device)
  D = helpers.to_dev(helpers.unit_rows(81, 32, 64), device)
  params = {
      'mode': 'fully-connected', 'num_epochs': 2,
      'code_inference_algorithm': 'subspace_fista',
      'group_assignments': [list(range(4 * i, 4 * i + 4)) for i in range(8)],
      'subspace_alignment_penalty': 2e-4,
      'inference_param_schedule': {0: {'sparsity_weight': 0.02,
                                       'num_iters': 10}},
      'dictionary_update_algorithm': 'subspace_sc_cheap_quadratic_descent',
      'dict_update_param_schedule': {0: {'stepsize': 0.05, 'num_iters': 1}}}
  before = D.clone()
  sparse_coding.train_dictionary([X[:32], X[32:]], [], D, params)
  assert not torch.allclose(D, before)
  assert np.allclose(D.norm(dim=1).cpu().numpy(), 1.0, atol=1e-5)


def test_training_recovers_a_planted_dictionary(device):
  """End to end on the fused bf16x3 path: patches synthesised from a planted
  dictionary with sparse codes; a few epochs of FISTA + cheap-quadratic updates
  must drive the reconstruction error down and move the learned atoms
  towards the planted ones."""
  from training import sparse_coding
  from analysis_transforms.fully_connected import ista_fista
  rs = np.random.RandomState(90)
  s, n, b = 256, 256, 4096
  planted = helpers.unit_rows(91, s, n)
  codes = (rs.rand(b, s) < 0.03) * rs.randn(b, s)
  X = (codes @ planted + 0.01 * rs.randn(b, n)).astype(np.float32)
  X = helpers.to_dev(X, device)
  D = helpers.to_dev(helpers.unit_rows(92, s, n), device)
  params = {
      'mode': 'fully-connected', 'num_epochs': 30,
      'code_inference_algorithm': 'fista',
      'inference_param_schedule': {0: {'sparsity_weight': 0.05,
                                       'num_iters': 40}},
      'dictionary_update_algorithm': 'sc_cheap_quadratic_descent',
      'dict_update_param_schedule': {0: {'stepsize': 0.05, 'num_iters': 1}}}

  def recon_error(dictionary):
    c = ista_fista.run(X[:1024], dictionary, 0.05, 60)
    return float(((c @ dictionary - X[:1024]) ** 2).mean())

  planted_dev = helpers.to_dev(planted, device)

  def alignment(dictionary):
    return float((planted_dev @ dictionary.t()).abs().max(dim=1).values.mean())

  before = recon_error(D)
  align_before = alignment(D)
  batches = [X[i: i + 1024] for i in range(0, b, 1024)]
  sparse_coding.train_dictionary(batches, [], D, params)
  after = recon_error(D)
  assert after < 0.5 * before, (before, after)
  # the learned atoms move towards the planted ones (mean best |cosine|)
  align_after = alignment(D)
  print('recon %.4g -> %.4g, alignment %.3f -> %.3f' % (
      before, after, align_before, align_after))
  assert align_after > align_before + 0.05, (align_before, align_after)
