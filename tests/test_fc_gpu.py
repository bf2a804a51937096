"""Fully-connected inference + dictionary updates on the GPU, through the
C-ABI, against the golden vectors (reference outputs) and the CPU oracle."""
import numpy as np
import pytest
import torch

import helpers
import sc_oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def plugins():
  from analysis_transforms.fully_connected import ista_fista
  from dict_update_rules.fully_connected import sc_steepest_descent
  from dict_update_rules.fully_connected import sc_cheap_quadratic_descent
  return ista_fista, sc_steepest_descent, sc_cheap_quadratic_descent


def test_c1_all_threshold_modes_f32(device, plugins):
  ista_fista = plugins[0]
  g = helpers.load('fc_c1')
  X, D = helpers.to_dev(g['images'], device), helpers.to_dev(
      g['dictionary'], device)
  lam, eta = float(g['sparsity_weight']), float(g['stepsize'])
  modes = {'soft': (False, False), 'soft_nonneg': (True, False),
           'hard': (False, True), 'hard_nonneg': (True, True)}
  for name, (nonneg, hard) in modes.items():
    codes = ista_fista.run(X, D, lam, 20, variant='ista',
                           nonnegative_only=nonneg, hard_threshold=hard,
                           precision='f32', stepsize=eta)
    helpers.assert_codes_match(codes.cpu().numpy(), g['codes_ista_' + name],
                               helpers.REL_TOL_SHORT, 'ista ' + name)
  codes = ista_fista.run(X, D, lam, 20, precision='f32', stepsize=eta)
  helpers.assert_codes_match(codes.cpu().numpy(), g['codes_fista_soft'],
                             helpers.REL_TOL_SHORT, 'fista')


def test_c1_own_lipschitz_step(device, plugins):
  """Without an injected stepsize the plugin runs its own Gram + eigen-solve
  (ista_fista.py:72-80)."""
  ista_fista = plugins[0]
  g = helpers.load('fc_c1')
  X, D = helpers.to_dev(g['images'], device), helpers.to_dev(
      g['dictionary'], device)
  import vtc_hip
  gram = vtc_hip.gram(D, transpose_a=True).cpu().numpy()
  ref = g['dictionary'].astype(np.float64)
  assert helpers.rel_err(gram, ref.T @ ref) < 1e-6
  eta = float(vtc_hip.stepsize_from_gram(vtc_hip.gram(D, True), D))
  assert abs(eta - float(g['stepsize'])) / float(g['stepsize']) < 1e-5
  codes = ista_fista.run(X, D, float(g['sparsity_weight']), 20,
                         precision='f32')
  helpers.assert_codes_match(codes.cpu().numpy(), g['codes_fista_soft'],
                             2e-5, 'fista, own eta', max_flip_mag=1e-5)


def test_c1_early_stopping(device, plugins):
  ista_fista = plugins[0]
  g = helpers.load('fc_c1')
  X, D = helpers.to_dev(g['images'], device), helpers.to_dev(
      g['dictionary'], device)
  lam, eta = float(g['sparsity_weight']), float(g['stepsize'])
  for variant in ('ista', 'fista'):
    codes = ista_fista.run(X, D, lam, 500, variant=variant,
                           early_stopping_epsilon=1e-2, precision='f32',
                           stepsize=eta)
    assert 1 < ista_fista.run.last_iters < 500
    helpers.assert_codes_match(codes.cpu().numpy(),
                               g['codes_%s_earlystop' % variant],
                               helpers.REL_TOL_SHORT, variant + ' early stop')


def test_inputs_are_not_mutated_and_warm_start_moves(device, plugins):
  """The only assertions of the reference's own tests
  (vision_transform_codes/tests/ista_fista_1.py:45-54)."""
  ista_fista = plugins[0]
  g = helpers.load('fc_c1')
  X, D = helpers.to_dev(g['images'], device), helpers.to_dev(
      g['dictionary'], device)
  X0, D0 = X.clone(), D.clone()
  first = ista_fista.run(X, D, 0.05, 10, variant='ista',
                         nonnegative_only=True, hard_threshold=True,
                         precision='f32')
  keep = first.clone()
  again = ista_fista.run(X, D, 0.05, 100, variant='ista', initial_codes=first,
                         nonnegative_only=True, hard_threshold=True,
                         precision='f32')
  assert torch.equal(X, X0) and torch.equal(D, D0)
  assert torch.equal(first, keep)
  assert not torch.allclose(again, keep)
  assert again.data_ptr() != first.data_ptr()


def test_c2_mini_trace_f32(device, plugins):
  ista_fista = plugins[0]
  g = helpers.load('fc_c2_mini')
  Xn = helpers.gaussian_patches(int(g['seed_images']), 64, 256)
  Dn = helpers.unit_rows(int(g['seed_dictionary']), 1024, 256)
  X, D = helpers.to_dev(Xn, device), helpers.to_dev(Dn, device)
  lam, eta = float(g['sparsity_weight']), float(g['stepsize'])
  for k, tol in ((1, helpers.REL_TOL_SHORT), (2, helpers.REL_TOL_SHORT),
                 (20, helpers.REL_TOL_SHORT), (200, helpers.REL_TOL_F32)):
    codes = ista_fista.run(X, D, lam, k, precision='f32', stepsize=eta)
    err, flips = helpers.assert_codes_match(
        codes.cpu().numpy(), g['codes_fista_T%d' % k], tol, 'T=%d' % k)
    print('fc_c2 f32 T=%d rel %.2e flips %d' % (k, err, flips))
  codes = ista_fista.run(X, D, lam, 50, variant='ista', precision='f32',
                         stepsize=eta)
  helpers.assert_codes_match(codes.cpu().numpy(), g['codes_ista_T50'],
                             helpers.REL_TOL_SHORT, 'ista T=50')
  warm = ista_fista.run(X, D, lam, 20, precision='f32', stepsize=eta,
                        initial_codes=helpers.to_dev(g['codes_fista_T20'],
                                                     device))
  helpers.assert_codes_match(warm.cpu().numpy(), g['codes_fista_warm20'],
                             helpers.REL_TOL_SHORT, 'warm start')


def test_whitened_patches_f32(device, plugins):
  ista_fista = plugins[0]
  g = helpers.load('whitened')
  X = helpers.to_dev(g['images'], device)
  D = helpers.to_dev(helpers.unit_rows(int(g['seed_dictionary']), 512, 256),
                     device)
  codes = ista_fista.run(X, D, float(g['sparsity_weight']), 100,
                         precision='f32', stepsize=float(g['stepsize']))
  helpers.assert_codes_match(codes.cpu().numpy(), g['codes_fista_T100'],
                             helpers.REL_TOL_F32, 'whitened patches')


@pytest.mark.parametrize('b,n,s', [(1, 16, 6), (37, 20, 50), (130, 64, 129),
                                   (257, 100, 200)])
def test_ragged_shapes_against_oracle(device, plugins, b, n, s):
  """Sizes that are not multiples of any tile dimension."""
  ista_fista = plugins[0]
  Xn = helpers.gaussian_patches(100 + b, b, n)
  Dn = helpers.unit_rows(200 + s, s, n)
  eta = sc_oracle.fc_stepsize(torch.from_numpy(Dn))
  ref = sc_oracle.fc_ista_fista(torch.from_numpy(Xn), torch.from_numpy(Dn),
                                0.03, 25, stepsize=eta)
  codes = ista_fista.run(helpers.to_dev(Xn, device),
                         helpers.to_dev(Dn, device), 0.03, 25,
                         precision='f32', stepsize=float(eta))
  helpers.assert_codes_match(codes.cpu().numpy(), ref.numpy(),
                             helpers.REL_TOL_SHORT, 'ragged %s' % ((b, n, s),))


def test_empty_batch_and_bad_arguments(device, plugins):
  ista_fista = plugins[0]
  D = helpers.to_dev(helpers.unit_rows(3, 8, 4), device)
  out = ista_fista.run(torch.zeros(0, 4, device=device), D, 0.1, 3,
                       precision='f32', stepsize=0.5)
  assert tuple(out.shape) == (0, 8)
  X = torch.zeros(2, 4, device=device)
  with pytest.raises(AssertionError):
    ista_fista.run(X, D, 0.1, 3, variant='lista')
  with pytest.raises(UnboundLocalError):
    ista_fista.run(X, D, 0.1, 0)


def test_dictionary_updates_c1(device, plugins):
  _, steepest, cheapquad = plugins
  g = helpers.load('fc_c1')
  X = helpers.to_dev(g['images'], device)
  C = helpers.to_dev(g['codes_fista_soft'], device)
  X0, C0 = X.clone(), C.clone()
  D = helpers.to_dev(g['dictionary'].copy(), device)
  alias = D
  assert steepest.run(X, D, C, stepsize=0.1) is None
  assert alias.data_ptr() == D.data_ptr()
  assert helpers.rel_err(D.cpu().numpy(),
                         g['dict_after_steepest']) < helpers.REL_TOL_DICT
  assert np.allclose(D.norm(dim=1).cpu().numpy(), 1.0, atol=1e-6)
  D = helpers.to_dev(g['dictionary'].copy(), device)
  steepest.run(X, D, C, stepsize=0.1, num_iters=3, normalize_dictionary=False)
  assert helpers.rel_err(
      D.cpu().numpy(), g['dict_after_steepest_3it_nonorm']
  ) < helpers.REL_TOL_DICT
  D = helpers.to_dev(g['dictionary'].copy(), device)
  h = helpers.to_dev(g['hessian_diagonal'], device)
  h0 = h.clone()
  cheapquad.run(X, D, C, h, stepsize=0.1, num_iters=2)
  assert helpers.rel_err(D.cpu().numpy(),
                         g['dict_after_cheapquad_2it']) < helpers.REL_TOL_DICT
  assert torch.equal(X, X0) and torch.equal(C, C0) and torch.equal(h, h0)


def test_dictionary_updates_c2_mini(device, plugins):
  _, steepest, cheapquad = plugins
  g = helpers.load('fc_c2_mini')
  X = helpers.to_dev(helpers.gaussian_patches(0, 64, 256), device)
  Dn = helpers.unit_rows(1, 1024, 256)
  C = helpers.to_dev(g['codes_fista_T200'], device)
  D = helpers.to_dev(Dn.copy(), device)
  steepest.run(X, D, C, stepsize=0.1)
  assert helpers.rel_err(D.cpu().numpy(),
                         g['dict_after_steepest']) < helpers.REL_TOL_DICT
  D = helpers.to_dev(Dn.copy(), device)
  cheapquad.run(X, D, C, helpers.to_dev(g['hessian_diagonal'], device),
                stepsize=0.1)
  assert helpers.rel_err(D.cpu().numpy(),
                         g['dict_after_cheapquad']) < helpers.REL_TOL_DICT


def test_gradient_is_reproducible_and_linear(device, plugins):
  """Size-independent properties at a batch the oracle would take long on:
  the split-K gradient is bitwise reproducible, and the gradient of a
  concatenated batch is the sum of the halves' gradients (the property the
  multi-GPU all-reduce relies on)."""
  import vtc_hip
  lib = vtc_hip.load_library()
  b, n, s = 8192, 256, 1024
  X = helpers.to_dev(helpers.gaussian_patches(5, b, n), device)
  D = helpers.to_dev(helpers.unit_rows(6, s, n), device)
  C = torch.relu(helpers.to_dev(helpers.gaussian_patches(7, b, s), device)
                 - 0.1)

  def grad(Xp, Cp):
    bb = Xp.shape[0]
    ws = vtc_hip.workspace(lib.vtc_fc_dict_gradient_workspace_bytes(bb, n, s),
                           device)
    out = torch.empty(s, n, device=device)
    vtc_hip.check(lib.vtc_fc_dict_gradient(
        vtc_hip.ptr(Xp), vtc_hip.ptr(D), vtc_hip.ptr(Cp), vtc_hip.ptr(out),
        bb, n, s, vtc_hip.ptr(ws), ws.numel(),
        vtc_hip.current_stream(device)), 'grad')
    return out
  full, again = grad(X, C), grad(X, C)
  assert torch.equal(full, again)
  halves = grad(X[:b // 2].contiguous(), C[:b // 2].contiguous()) + grad(
      X[b // 2:].contiguous(), C[b // 2:].contiguous())
  assert helpers.rel_err(halves.cpu().numpy(), full.cpu().numpy()) < 1e-6
  ref = (C.double().t() @ (C.double() @ D.double() - X.double()))
  assert helpers.rel_err(full.cpu().numpy(), ref.cpu().numpy()) < 1e-6


def test_ica_natural_gradient_update(device):
  """The sibling update rule of the plugin API:
  D += stepsize ((C^T sign(C)) / b - I) D, against the reference's output
  (exact-f32 MFMA contractions, 2e-6 relative on the dictionary)."""
  from dict_update_rules.fully_connected import ica_natural_gradient
  g = helpers.load('ica')
  for tag in ('square', 'wide'):
    C = helpers.to_dev(g[tag + '_codes'], device)
    C0 = C.clone()
    for iters in (1, 3):
      D = helpers.to_dev(g[tag + '_dictionary0'].copy(), device)
      assert ica_natural_gradient.run(D, C, stepsize=0.01,
                                      num_iters=iters) is None
      assert helpers.rel_err(
          D.cpu().numpy(),
          g['%s_dictionary_after_%d' % (tag, iters)]) < helpers.REL_TOL_DICT
    assert torch.equal(C, C0)


def test_large_patches_use_the_device_eigen_solver(device, plugins):
  """20x20 patches (n = 400 > 256): the Lipschitz step runs the workspace
  variant of the device Lanczos kernel, not a library eigen-solver."""
  import vtc_hip
  ista_fista = plugins[0]
  Xn = helpers.gaussian_patches(77, 64, 400)
  Dn = helpers.unit_rows(78, 500, 400)
  ref = sc_oracle.fc_ista_fista(torch.from_numpy(Xn), torch.from_numpy(Dn),
                                0.02, 30)
  assert vtc_hip.LANCZOS_MAX_N >= 400
  out = ista_fista.run(helpers.to_dev(Xn, device), helpers.to_dev(Dn, device),
                       0.02, 30, precision='f32')
  helpers.assert_codes_match(out.cpu().numpy(), ref.numpy(),
                             helpers.REL_TOL_SHORT, 'n = 400')


@pytest.mark.parametrize('mode,tol,flip', [
    ('f16x3', helpers.REL_TOL_F32, helpers.NEAR_THRESHOLD),
    ('bf16x3', helpers.REL_TOL_BF16X3, 1e-5)])
def test_tiled_split_contractions(device, plugins, mode, tol, flip):
  """Shapes outside the fused kernels (20x20 patches, 500 atoms) on the tiled
  hi/lo split contractions (gemm_x3.h): f16x3 -- operands in power-of-two
  scaled units, maxima handed from launch to launch on the device -- at
  north_star's tolerance over 200 FISTA iterations, bf16x3 at its 3e-5; warm
  start (the first operand's scale comes from the initial codes), ISTA, and
  data far from unit scale (the scales are exact powers of two)."""
  ista_fista = plugins[0]
  Xn = helpers.gaussian_patches(81, 96, 400)
  Dn = helpers.unit_rows(82, 500, 400)
  Xc, Dc = torch.from_numpy(Xn), torch.from_numpy(Dn)
  eta = sc_oracle.fc_stepsize(Dc)
  X, D = helpers.to_dev(Xn, device), helpers.to_dev(Dn, device)
  for iters in (30, 200):
    ref = sc_oracle.fc_ista_fista(Xc, Dc, 0.02, iters, stepsize=eta)
    out = ista_fista.run(X, D, 0.02, iters, precision=mode,
                         stepsize=float(eta))
    helpers.assert_codes_match(out.cpu().numpy(), ref.numpy(), tol,
                               '%s T=%d' % (mode, iters), max_flip_mag=flip)
  warm = sc_oracle.fc_ista_fista(Xc, Dc, 0.02, 5, stepsize=eta)
  ref = sc_oracle.fc_ista_fista(Xc, Dc, 0.02, 20, stepsize=eta,
                                initial_codes=warm, variant='ista')
  out = ista_fista.run(X, D, 0.02, 20, precision=mode, stepsize=float(eta),
                       initial_codes=helpers.to_dev(warm.numpy(), device),
                       variant='ista')
  helpers.assert_codes_match(out.cpu().numpy(), ref.numpy(), tol,
                             mode + ' warm ista', max_flip_mag=flip)
  if mode == 'f16x3':
    # 2^30 times larger patches and threshold: codes 2^30 times larger,
    # bit for bit (every scale in the path is a power of two)
    base = ista_fista.run(X, D, 0.02, 25, precision=mode, stepsize=float(eta))
    big = ista_fista.run(X * 2.0 ** 30, D, 0.02 * 2.0 ** 30, 25,
                         precision=mode, stepsize=float(eta))
    assert torch.equal(big, base * 2.0 ** 30)
