"""Subspace (group-LASSO) inference and update on the GPU vs golden vectors."""
import numpy as np
import pytest
import torch

import helpers
import sc_oracle

pytestmark = pytest.mark.gpu

RAGGED = [[0, 2, 5], [1], [2, 3, 4, 5]]
GROUPS4 = [list(range(4 * i, 4 * i + 4)) for i in range(16)]


@pytest.fixture(scope='module')
def plugins():
  from analysis_transforms.fully_connected import subspace_ista_fista
  from dict_update_rules.fully_connected import (
      subspace_sc_cheap_quadratic_descent)
  return subspace_ista_fista, subspace_sc_cheap_quadratic_descent


def test_ragged_overlapping_groups(device, plugins):
  sub = plugins[0]
  g = helpers.load('subspace')
  X = helpers.to_dev(g['ro_images'], device)
  D = helpers.to_dev(g['ro_dictionary'], device)
  for variant in ('ista', 'fista'):
    codes = sub.run(X, D, RAGGED, 0.02, 30, variant=variant)
    helpers.assert_codes_match(codes.cpu().numpy(), g['ro_codes_' + variant],
                               2e-5, 'ragged ' + variant, max_flip_mag=1e-5)
  init = helpers.to_dev(g['ro_codes_ista'], device)
  keep = init.clone()
  warm = sub.run(X, D, RAGGED, 0.02, 10, initial_codes=init)
  assert torch.equal(init, keep)
  helpers.assert_codes_match(warm.cpu().numpy(), g['ro_codes_warm'], 2e-5,
                             'ragged warm', max_flip_mag=1e-5)


def test_groups_of_four_and_mini_c4(device, plugins):
  sub = plugins[0]
  g = helpers.load('subspace')
  X = helpers.to_dev(g['g4_images'], device)
  D = helpers.to_dev(g['g4_dictionary'], device)
  codes = sub.run(X, D, GROUPS4, 0.02, 40)
  helpers.assert_codes_match(codes.cpu().numpy(), g['g4_codes_fista'], 2e-5,
                             'groups of 4', max_flip_mag=1e-5)
  X = helpers.to_dev(helpers.gaussian_patches(24, 32, 256), device)
  D = helpers.to_dev(helpers.unit_rows(25, 512, 256), device)
  groups = [list(map(int, x)) for x in np.array_split(np.arange(512), 64)]
  codes = sub.run(X, D, groups, 0.008, 50)
  helpers.assert_codes_match(codes.cpu().numpy(), g['c4_codes_fista'], 2e-5,
                             'mini config 4', max_flip_mag=1e-5)


def test_early_stopping_and_unsupported_options(device, plugins):
  sub = plugins[0]
  g = helpers.load('subspace')
  X = helpers.to_dev(g['g4_images'], device)
  D = helpers.to_dev(g['g4_dictionary'], device)
  Xc, Dc = torch.from_numpy(g['g4_images']), torch.from_numpy(
      g['g4_dictionary'])
  ref = sc_oracle.subspace_ista_fista(Xc, Dc, GROUPS4, 0.02, 400,
                                      variant='ista',
                                      early_stopping_epsilon=5e-3)
  codes = sub.run(X, D, GROUPS4, 0.02, 400, variant='ista',
                  early_stopping_epsilon=5e-3)
  assert 1 < sub.run.last_iters < 400
  helpers.assert_codes_match(codes.cpu().numpy(), ref.numpy(), 1e-4,
                             'early stop', max_flip_mag=1e-4)
  with pytest.raises(NotImplementedError):
    sub.run(X, D, GROUPS4, 0.02, 5, hard_threshold=True)
  with pytest.raises(NotImplementedError):
    sub.run(X, D, GROUPS4, 0.02, 5, ret_summed_gduplicates=False)


def test_subspace_cheap_quadratic_update(device, plugins):
  upd = plugins[1]
  g = helpers.load('subspace')
  X = helpers.to_dev(g['g4_images'], device)
  C = helpers.to_dev(g['g4_codes_fista'], device)
  h = helpers.to_dev(g['g4_hessian'], device)
  for name, pen in (('pen0', 0.), ('pen2e-4', 2e-4), ('pen0.05', 0.05)):
    D = helpers.to_dev(g['g4_dictionary'].copy(), device)
    assert upd.run(X, D, C, GROUPS4, h, pen, stepsize=0.1) is None
    assert helpers.rel_err(D.cpu().numpy(),
                           g['g4_dict_after_' + name]) < helpers.REL_TOL_DICT
  D = helpers.to_dev((g['g4_dictionary'] * 1.5).copy(), device)
  upd.run(X, D, C, GROUPS4, h, 0.05, stepsize=0.1, normalize_dictionary=False)
  assert helpers.rel_err(
      D.cpu().numpy(), g['g4_dict_after_pen0.05_nonorm']
  ) < helpers.REL_TOL_DICT


def test_alignment_gradient_with_shared_atoms(device, plugins):
  """Overlapping groups: an atom's penalty gradient is the sum over its
  groups (subspace_sc_cheap_quadratic_descent.py:65-69)."""
  upd = plugins[1]
  rs = np.random.RandomState(3)
  Xn = helpers.gaussian_patches(60, 16, 16)
  Dn = helpers.unit_rows(61, 6, 16)
  Cn = (0.1 * rs.randn(16, 6)).astype(np.float32)
  hn = np.abs(rs.randn(6)).astype(np.float32) * 0.01
  ref = torch.from_numpy(Dn.copy())
  sc_oracle.subspace_cheap_quadratic_descent(
      torch.from_numpy(Xn), ref, torch.from_numpy(Cn), RAGGED,
      torch.from_numpy(hn), 0.05, stepsize=0.05)
  D = helpers.to_dev(Dn.copy(), device)
  upd.run(helpers.to_dev(Xn, device), D, helpers.to_dev(Cn, device), RAGGED,
          helpers.to_dev(hn, device), 0.05, stepsize=0.05)
  assert helpers.rel_err(D.cpu().numpy(), ref.numpy()) < helpers.REL_TOL_DICT


def test_bf16x3_early_stopping(device, plugins):
  """Early stopping with the proximal step fused into the product epilogue:
  same iteration count and codes as the oracle."""
  sub = plugins[0]
  m, num_groups, n = 4, 24, 64
  groups = [list(range(g * m, g * m + m)) for g in range(num_groups)]
  Xn = helpers.gaussian_patches(610, 70, n)
  Dn = helpers.unit_rows(611, num_groups * m, n)
  ref = sc_oracle.subspace_ista_fista(torch.from_numpy(Xn),
                                      torch.from_numpy(Dn), groups, 0.03, 200,
                                      early_stopping_epsilon=5e-3)
  out = sub.run(helpers.to_dev(Xn, device), helpers.to_dev(Dn, device), groups,
                0.03, 200, early_stopping_epsilon=5e-3, precision='bf16x3')
  assert 1 < sub.run.last_iters < 200
  helpers.assert_codes_match(out.cpu().numpy(), ref.numpy(), 2e-5,
                             'bf16x3 early stop', max_flip_mag=1e-4)


def test_bf16x3_contraction_matches_reference(device, plugins):
  """The tiled bf16 hi/lo split contraction on the subspace path."""
  sub = plugins[0]
  g = helpers.load('subspace')
  X = helpers.to_dev(g['g4_images'], device)
  D = helpers.to_dev(g['g4_dictionary'], device)
  codes = sub.run(X, D, GROUPS4, 0.02, 40, precision='bf16x3')
  helpers.assert_codes_match(codes.cpu().numpy(), g['g4_codes_fista'], 2e-5,
                             'groups of 4, bf16x3', max_flip_mag=1e-5)
  # the f16 split of the same tiles: north_star's tolerance
  codes = sub.run(X, D, GROUPS4, 0.02, 40, precision='f16x3')
  helpers.assert_codes_match(codes.cpu().numpy(), g['g4_codes_fista'],
                             helpers.REL_TOL_SHORT, 'groups of 4, f16x3')
  X = helpers.to_dev(helpers.gaussian_patches(24, 32, 256), device)
  D = helpers.to_dev(helpers.unit_rows(25, 512, 256), device)
  groups = [list(map(int, x)) for x in np.array_split(np.arange(512), 64)]
  codes = sub.run(X, D, groups, 0.008, 50, precision='bf16x3')
  helpers.assert_codes_match(codes.cpu().numpy(), g['c4_codes_fista'], 2e-5,
                             'mini config 4, bf16x3', max_flip_mag=1e-5)
  with pytest.raises(NotImplementedError):
    sub.run(X, D, groups, 0.008, 5, precision='bf16')


@pytest.mark.parametrize('m,precision', [
    (3, 'f32'), (5, 'f32'), (2, 'f32'), (1, 'f32'), (16, 'bf16x3'),
    (1, 'bf16x3'), (2, 'bf16x3'), (4, 'bf16x3'), (8, 'bf16x3'),
    (32, 'bf16x3'), (12, 'bf16x3'), (64, 'bf16x3'),
    (1, 'f16x3'), (2, 'f16x3'), (4, 'f16x3'), (8, 'f16x3'), (16, 'f16x3'),
    (32, 'f16x3'), (12, 'f16x3')])
def test_group_sizes_against_oracle(device, plugins, m, precision):
  """f32: power-of-two group sizes take the coalesced shuffle kernel, the
  others the thread-per-group kernel.  Split modes on the tiled contractions:
  powers of two up to 32 run the proximal step in the epilogue of the gradient
  product (f16x3: the f16 split in scaled units, north_star's tolerance), the
  others the separate kernels on the bf16 split.  All against the oracle."""
  sub = plugins[0]
  num_groups = 12
  s_atoms, n = num_groups * m, 64
  if precision != 'f32':
    num_groups = 20 if m <= 16 else 5   # slots not a multiple of the 128 tile
    s_atoms = num_groups * m
  groups = [list(range(g * m, g * m + m)) for g in range(num_groups)]
  Xn = helpers.gaussian_patches(500 + m, 40, n)
  Dn = helpers.unit_rows(501 + m, s_atoms, n)
  ref = sc_oracle.subspace_ista_fista(torch.from_numpy(Xn),
                                      torch.from_numpy(Dn), groups, 0.03, 25)
  out = sub.run(helpers.to_dev(Xn, device), helpers.to_dev(Dn, device), groups,
                0.03, 25, precision=precision)
  tight = precision == 'f32' or (precision == 'f16x3' and m != 12)
  helpers.assert_codes_match(
      out.cpu().numpy(), ref.numpy(),
      helpers.REL_TOL_SHORT if tight else 2e-5, 'groups of %d' % m,
      max_flip_mag=helpers.NEAR_THRESHOLD if tight else 1e-5)


def test_full_size_properties(device, plugins):
  """BASELINE configs[3] (4096 atoms in 512 groups of 8, b = 8192; too big
  for the oracle): rows are independent (a slice of the batch reproduces the
  same rows), the run is reproducible, and the bf16x3 path (prox
  fused into the product epilogue, split-K residual) agrees with the exact-f32
  kernels on a slice."""
  sub = plugins[0]
  b, s_atoms, n = 8192, 4096, 256
  X = helpers.to_dev(helpers.gaussian_patches(31, b, n), device)
  D = helpers.to_dev(helpers.unit_rows(32, s_atoms, n), device)
  groups = [list(map(int, x)) for x in np.array_split(np.arange(s_atoms), 512)]
  eta = 0.05
  full = sub.run(X, D, groups, 0.008, 30, precision='bf16x3', stepsize=eta)
  again = sub.run(X, D, groups, 0.008, 30, precision='bf16x3', stepsize=eta)
  assert torch.equal(full, again)
  part = sub.run(X[1024:1024 + 256].contiguous(), D, groups, 0.008, 30,
                 precision='bf16x3', stepsize=eta)
  # not bitwise: the residual product splits its K axis by the batch size,
  # so the f32 summation order differs between the two runs
  helpers.assert_codes_match(part.cpu().numpy(),
                             full[1024:1024 + 256].cpu().numpy(), 5e-6,
                             'slice vs full batch', max_flip_mag=1e-6)
  exact = sub.run(X[:512].contiguous(), D, groups, 0.008, 30, precision='f32',
                  stepsize=eta)
  helpers.assert_codes_match(full[:512].cpu().numpy(), exact.cpu().numpy(),
                             5e-5, 'subspace bf16x3 vs f32 path',
                             max_flip_mag=5e-6)


@pytest.mark.parametrize('precision', ['f32', 'bf16x3', 'f16x3'])
def test_saturated_memory_pipeline_is_reproducible(device, plugins, precision):
  """20 000 patches x 300 slots (rows that are not multiples of 128 bytes,
  enough blocks to keep the memory pipeline busy): two runs are bit-identical
  and agree with the exact-f32 path.  This is the size at which a 16-byte
  buffer store with a register scalar offset lost its first element in a few
  lanes (a wait state hipcc leaves out on gfx950; csrc/epi_prox.h) -- small
  problems never showed it."""
  sub = plugins[0]
  rs = np.random.RandomState(0)
  m, num_groups, n, b = 4, 75, 144, 20000
  X = helpers.to_dev((0.1 * rs.randn(b, n)).astype(np.float32), device)
  D = helpers.to_dev(helpers.unit_rows(103, num_groups * m, n), device)
  groups = [list(range(g * m, g * m + m)) for g in range(num_groups)]
  runs = [sub.run(X, D, groups, 0.01, 3, stepsize=0.05, precision=precision)
          for _ in range(4)]
  for other in runs[1:]:
    assert torch.equal(runs[0], other)
  exact = runs[0] if precision == 'f32' else sub.run(
      X, D, groups, 0.01, 3, stepsize=0.05, precision='f32')
  helpers.assert_codes_match(
      runs[0].cpu().numpy(), exact.cpu().numpy(),
      2e-5 if precision == 'bf16x3' else helpers.REL_TOL_SHORT,
      'large batch ' + precision,
      max_flip_mag=1e-5 if precision == 'bf16x3' else helpers.NEAR_THRESHOLD)


def test_empty_batch(device, plugins):
  sub = plugins[0]
  D = helpers.to_dev(helpers.unit_rows(3, 16, 32), device)
  groups = [list(range(4 * g, 4 * g + 4)) for g in range(4)]
  out = sub.run(torch.zeros(0, 32, device=device), D, groups, 0.05, 3)
  assert tuple(out.shape) == (0, 16)
