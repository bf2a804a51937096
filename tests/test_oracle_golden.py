"""The CPU oracle against the golden vectors produced by the reference.

Runs without a GPU.  In the container that generated the vectors the oracle is
bit-identical to the reference (same torch CPU ops in the same order); on
another host CPU the BLAS kernels may sum in a different order, hence the
small tolerances."""
import numpy as np
import pytest
import torch

import helpers
import sc_oracle

T = torch.from_numpy


def test_fc_c1_all_threshold_modes():
  g = helpers.load('fc_c1')
  X, D = T(g['images']), T(g['dictionary'])
  lam = float(g['sparsity_weight'])
  assert abs(float(sc_oracle.fc_stepsize(D)) - float(g['stepsize'])) < 1e-6
  modes = {'soft': (False, False), 'soft_nonneg': (True, False),
           'hard': (False, True), 'hard_nonneg': (True, True)}
  for name, (nonneg, hard) in modes.items():
    codes = sc_oracle.fc_ista_fista(X, D, lam, 20, variant='ista',
                                    nonnegative_only=nonneg,
                                    hard_threshold=hard)
    helpers.assert_codes_match(codes.numpy(), g['codes_ista_' + name],
                               helpers.REL_TOL_SHORT, 'ista ' + name)
  codes = sc_oracle.fc_ista_fista(X, D, lam, 20)
  helpers.assert_codes_match(codes.numpy(), g['codes_fista_soft'],
                             helpers.REL_TOL_SHORT, 'fista')


def test_fc_c1_early_stopping():
  g = helpers.load('fc_c1')
  X, D = T(g['images']), T(g['dictionary'])
  lam = float(g['sparsity_weight'])
  for variant in ('ista', 'fista'):
    codes = sc_oracle.fc_ista_fista(X, D, lam, 500, variant=variant,
                                    early_stopping_epsilon=1e-2)
    helpers.assert_codes_match(codes.numpy(),
                               g['codes_%s_earlystop' % variant],
                               helpers.REL_TOL_SHORT, variant + ' early stop')
    full = sc_oracle.fc_ista_fista(X, D, lam, 500, variant=variant)
    assert helpers.rel_err(full.numpy(), codes.numpy()) > 1e-6, (
        'early stopping did not stop early')


def test_fc_c1_dictionary_updates():
  g = helpers.load('fc_c1')
  X, C = T(g['images']), T(g['codes_fista_soft'])
  D = T(g['dictionary'].copy())
  sc_oracle.fc_steepest_descent(X, D, C, stepsize=0.1)
  assert helpers.rel_err(D.numpy(), g['dict_after_steepest']) < 1e-6
  assert np.allclose(np.linalg.norm(D.numpy(), axis=1), 1.0, atol=1e-6)
  D = T(g['dictionary'].copy())
  sc_oracle.fc_steepest_descent(X, D, C, stepsize=0.1, num_iters=3,
                                normalize_dictionary=False)
  assert helpers.rel_err(D.numpy(), g['dict_after_steepest_3it_nonorm']) < 1e-6
  D = T(g['dictionary'].copy())
  sc_oracle.fc_cheap_quadratic_descent(X, D, C, T(g['hessian_diagonal']),
                                       stepsize=0.1, num_iters=2)
  assert helpers.rel_err(D.numpy(), g['dict_after_cheapquad_2it']) < 1e-6


def test_fc_c2_mini_trace_and_updates():
  g = helpers.load('fc_c2_mini')
  X = helpers.gaussian_patches(int(g['seed_images']), 64, 256)
  Dn = helpers.unit_rows(int(g['seed_dictionary']), 1024, 256)
  # the regenerated inputs are the ones the vectors were made from
  assert abs(X.astype(np.float64).sum() - float(g['images_sum'])) < 1e-9
  assert abs(Dn.astype(np.float64).sum() - float(g['dictionary_sum'])) < 1e-9
  X, D = T(X), T(Dn)
  lam = float(g['sparsity_weight'])
  codes, trace = sc_oracle.fc_ista_fista(X, D, lam, 200,
                                         trace_at=[1, 2, 20, 200])
  for k in (1, 2, 20):
    helpers.assert_codes_match(trace[k].numpy(), g['codes_fista_T%d' % k],
                               helpers.REL_TOL_SHORT, 'T=%d' % k)
  helpers.assert_codes_match(trace[200].numpy(), g['codes_fista_T200'],
                             helpers.REL_TOL_F32, 'T=200')
  # the float64 run of the same algorithm: the reference's own noise floor
  truth = sc_oracle.fc_ista_fista(X.double(), D.double(), lam, 200)
  assert helpers.rel_err(truth.numpy(), g['codes_fista_T200_fp64']) < 1e-6
  floor = helpers.rel_err(g['codes_fista_T200'], truth.numpy())
  assert 1e-6 < floor < 5e-5
  warm = sc_oracle.fc_ista_fista(X, D, lam, 20,
                                 initial_codes=T(g['codes_fista_T20']))
  helpers.assert_codes_match(warm.numpy(), g['codes_fista_warm20'],
                             helpers.REL_TOL_SHORT, 'warm start')
  C = T(g['codes_fista_T200'])
  D1 = T(Dn.copy())
  sc_oracle.fc_steepest_descent(X, D1, C, stepsize=0.1)
  assert helpers.rel_err(D1.numpy(), g['dict_after_steepest']) < 1e-6
  h = sc_oracle.hessian_diag_ema_(torch.zeros(1024), C)
  assert helpers.rel_err(h.numpy(), g['hessian_diagonal']) < 1e-6
  D2 = T(Dn.copy())
  sc_oracle.fc_cheap_quadratic_descent(X, D2, C, T(g['hessian_diagonal']),
                                       stepsize=0.1)
  assert helpers.rel_err(D2.numpy(), g['dict_after_cheapquad']) < 1e-6


def test_subspace_golden():
  g = helpers.load('subspace')
  groups = [[0, 2, 5], [1], [2, 3, 4, 5]]
  X, D = T(g['ro_images']), T(g['ro_dictionary'])
  for variant in ('ista', 'fista'):
    codes = sc_oracle.subspace_ista_fista(X, D, groups, 0.02, 30,
                                          variant=variant)
    helpers.assert_codes_match(codes.numpy(), g['ro_codes_' + variant],
                               helpers.REL_TOL_SHORT, 'ragged ' + variant)
  warm = sc_oracle.subspace_ista_fista(X, D, groups, 0.02, 10,
                                       initial_codes=T(g['ro_codes_ista']))
  helpers.assert_codes_match(warm.numpy(), g['ro_codes_warm'],
                             helpers.REL_TOL_SHORT, 'ragged warm')
  groups4 = [list(range(4 * i, 4 * i + 4)) for i in range(16)]
  X, D = T(g['g4_images']), T(g['g4_dictionary'])
  codes = sc_oracle.subspace_ista_fista(X, D, groups4, 0.02, 40)
  helpers.assert_codes_match(codes.numpy(), g['g4_codes_fista'],
                             helpers.REL_TOL_SHORT, 'groups of 4')
  h = T(g['g4_hessian'])
  for name, pen in (('pen0', 0.), ('pen2e-4', 2e-4), ('pen0.05', 0.05)):
    Dn = T(g['g4_dictionary'].copy())
    sc_oracle.subspace_cheap_quadratic_descent(X, Dn, T(g['g4_codes_fista']),
                                               groups4, h, pen, stepsize=0.1)
    assert helpers.rel_err(Dn.numpy(), g['g4_dict_after_' + name]) < 1e-6
  Dn = T((g['g4_dictionary'] * 1.5).copy())
  sc_oracle.subspace_cheap_quadratic_descent(
      X, Dn, T(g['g4_codes_fista']), groups4, h, 0.05, stepsize=0.1,
      normalize_dictionary=False)
  assert helpers.rel_err(Dn.numpy(), g['g4_dict_after_pen0.05_nonorm']) < 1e-6
  with pytest.raises(NotImplementedError):
    sc_oracle.subspace_ista_fista(X, D, groups4, 0.02, 5, hard_threshold=True)


def test_conv_golden_and_index_conventions():
  g = helpers.load('conv')
  for name in ('k16s8', 'k11s1', 'k8s4_ragged'):
    imgs, D = T(g[name + '_images_padded']), T(g[name + '_dictionary'])
    stride = tuple(int(v) for v in g[name + '_stride'])
    pad = tuple(tuple(int(v) for v in row) for row in g[name + '_padding'])
    for variant in ('ista', 'fista'):
      codes = sc_oracle.conv_ista_fista(imgs, D, stride, pad, 0.05, 10,
                                        variant=variant)
      helpers.assert_codes_match(codes.numpy(),
                                 g['%s_codes_%s' % (name, variant)],
                                 helpers.REL_TOL_SHORT, name + ' ' + variant)
    C = T(g[name + '_codes_fista'])
    # the loop forms pin the flip / stride conventions of the torch calls
    assert helpers.rel_err(
        sc_oracle.conv_synthesis_naive(C, D, stride).numpy(),
        sc_oracle.conv_synthesis(C, D, stride).numpy()) < 1e-6
    assert helpers.rel_err(
        sc_oracle.conv_analysis_naive(imgs, D, stride).numpy(),
        sc_oracle.conv_analysis(imgs, D, stride).numpy()) < 1e-6
    assert helpers.rel_err(
        sc_oracle.conv_gradient_naive(imgs, D, C, stride, pad).numpy(),
        sc_oracle.conv_gradient(imgs, D, C, stride, pad).numpy()) < 1e-5
    Dn = T(g[name + '_dictionary'].copy())
    sc_oracle.conv_steepest_descent(imgs, Dn, C, stride, pad, stepsize=0.005)
    assert helpers.rel_err(Dn.numpy(), g[name + '_dict_after_steepest']) < 1e-6
    Dn = T(g[name + '_dictionary'].copy())
    sc_oracle.conv_cheap_quadratic_descent(imgs, Dn, C, T(g[name + '_hessian']),
                                           stride, pad, stepsize=0.005)
    assert helpers.rel_err(Dn.numpy(), g[name + '_dict_after_cheapquad']) < 1e-6


def test_conv_long_horizon_golden():
  """The oracle follows the reference through 100 FISTA iterations on both
  long-horizon cases (bit-identical when the fixture was written), and its
  step size is the reference's."""
  g = helpers.load('conv_long')
  lam = float(g['sparsity_weight'])
  for name in ('ex_k16s8', 'nd_k11s1'):
    imgs, D = T(g[name + '_images_padded']), T(g[name + '_dictionary'])
    stride = tuple(int(v) for v in g[name + '_stride'])
    pad = tuple(tuple(int(v) for v in row) for row in g[name + '_padding'])
    assert abs(float(sc_oracle.conv_stepsize(D)) -
               float(g[name + '_stepsize'])) < 1e-7
    for iters in (10, 100):
      codes = sc_oracle.conv_ista_fista(imgs, D, stride, pad, lam, iters,
                                        variant='fista')
      assert helpers.rel_err(
          codes.numpy(), g['%s_codes_fista_T%d' % (name, iters)]) < 1e-6


def test_conv_geometry_matches_config5():
  # 256 px, 11x11 kernels, stride 1 -> pad (10,10),
  # padded 276, code map 266
  assert sc_oracle.conv_padding_amount(256, 11, 1) == (10, 10)
  assert sc_oracle.conv_code_dim(276, 11, 1) == 266
  assert sc_oracle.conv_padding_amount(30, 8, 4) == (4, 6)


def test_trainer_trajectories():
  g = helpers.load('trainer')
  params = {
      'mode': 'fully-connected', 'code_inference_algorithm': 'fista',
      'inference_param_schedule': {
          0: {'sparsity_weight': 0.02, 'num_iters': 15},
          2: {'sparsity_weight': 0.01, 'num_iters': 30}},
      'dictionary_update_algorithm': 'sc_cheap_quadratic_descent',
      'dict_update_param_schedule': {
          0: {'stepsize': 0.1, 'num_iters': 1},
          2: {'stepsize': 0.05, 'num_iters': 2}}}
  X = g['fc_images']
  D = T(g['fc_dictionary0'].copy())
  hist = sc_oracle.train_steps([T(X[32 * i: 32 * i + 32]) for i in range(3)],
                               D, params)
  for i in range(3):
    assert helpers.rel_err(hist[i]['dictionary'].numpy(),
                           g['fc_dict_after_step%d' % (i + 1)]) < 2e-6
  assert helpers.rel_err(hist[2]['hessian'].numpy(),
                         g['fc_hessian_after_step3']) < 2e-6
  pad = tuple(tuple(int(v) for v in row) for row in g['conv_padding'])
  cparams = {
      'mode': 'convolutional', 'code_inference_algorithm': 'ista',
      'strides': (4, 4), 'padding': pad,
      'inference_param_schedule': {0: {'sparsity_weight': 0.05,
                                       'num_iters': 8}},
      'dictionary_update_algorithm': 'sc_cheap_quadratic_descent',
      'dict_update_param_schedule': {0: {'stepsize': 0.005, 'num_iters': 1}}}
  imgs = g['conv_images_padded']
  K = T(g['conv_dictionary0'].copy())
  hist = sc_oracle.train_steps([T(imgs[2 * i: 2 * i + 2]) for i in range(3)],
                               K, cparams)
  for i in range(3):
    assert helpers.rel_err(hist[i]['dictionary'].numpy(),
                           g['conv_dict_after_step%d' % (i + 1)]) < 2e-6


def test_validation_metrics():
  """compute_metrics restatement against the scalars the reference's own
  train_dictionary sent to its SummaryWriter (sparse_coding.py:177-229,
  :497-505): fully-connected, subspace and convolutional."""
  g = helpers.load('metrics')
  for tag, (params, train, val, D0, batch) in helpers.metrics_cases(g).items():
    D = T(D0.copy())
    hist = sc_oracle.train_steps(
        [T(train[batch * i: batch * i + batch])
         for i in range(len(train) // batch)], D, params,
        validation_batches=[T(val[batch * i: batch * i + batch])
                            for i in range(len(val) // batch)])
    names = [str(x) for x in g[tag + '_names']]
    for si, step in enumerate(g[tag + '_steps']):
      got = hist[int(step)]['validation']
      for ni, name in enumerate(names):
        want = g[tag + '_values'][si, ni]
        assert abs(float(got[name]) - want) <= 1e-6 * max(abs(want), 1e-12), (
            tag, int(step), name, float(got[name]), want)
    assert hist[1]['validation'] is None
    assert helpers.rel_err(D.numpy(), g[tag + '_dictionary_final']) < 2e-6


def test_ica_natural_gradient():
  g = helpers.load('ica')
  for tag in ('square', 'wide'):
    for iters in (1, 3):
      D = T(g[tag + '_dictionary0'].copy())
      sc_oracle.ica_natural_gradient(D, T(g[tag + '_codes']), stepsize=0.01,
                                     num_iters=iters)
      assert helpers.rel_err(
          D.numpy(), g['%s_dictionary_after_%d' % (tag, iters)]) < 1e-7


def test_whitened_patches():
  g = helpers.load('whitened')
  X = T(g['images'])
  D = T(helpers.unit_rows(int(g['seed_dictionary']), 512, 256))
  codes = sc_oracle.fc_ista_fista(X, D, float(g['sparsity_weight']), 100)
  helpers.assert_codes_match(codes.numpy(), g['codes_fista_T100'],
                             helpers.REL_TOL_F32, 'whitened')


def test_momentum_schedule():
  betas = sc_oracle.fista_betas(5)
  assert betas[0] == 0.0
  t1 = (1 + 5 ** 0.5) / 2
  t2 = (1 + (1 + 4 * t1 * t1) ** 0.5) / 2
  assert abs(betas[1] - (t1 - 1) / t2) < 1e-15
  assert all(0 <= b < 1 for b in betas)


def test_reset_prune_oracle_matches_reference():
  """f4: every non-interactive reset / prune mode of the reference's
  reset_or_prune_dict_elements, seeded as in oracle/make_golden.py."""
  import make_golden
  g = helpers.load('reset_prune')
  D0, groups0, C = make_golden.reset_prune_inputs()
  assert np.array_equal(D0, g['dictionary'])
  for tag, f_type, f_params, action, np_seed, torch_seed in (
      make_golden.reset_prune_cases()):
    groups = [list(x) for x in groups0]
    params = dict(f_params)
    params.update({'group_assignments': groups,
                   'coding_mode': 'fully-connected'})
    np.random.seed(np_seed)
    torch.manual_seed(torch_seed)
    Dn, rows = sc_oracle.reset_or_prune(torch.from_numpy(D0.copy()),
                                        torch.from_numpy(C), f_type, params,
                                        action)
    assert np.array_equal(np.asarray(rows), g[tag + '_affected']), tag
    assert np.array_equal(Dn.numpy(), g[tag + '_dictionary']), tag
    assert [len(x) for x in groups] == g[tag + '_group_sizes'].tolist(), tag
