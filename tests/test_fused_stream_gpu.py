"""The fused persistent kernel with streamed state (csrc/fused_stream.hip):
16x16 patches against dictionaries beyond the on-chip state of the fused FC
kernel -- the subspace plugin's padded dictionaries (group sizes 2 / 4 / 8) and
fully-connected dictionaries above 1024 atoms -- against the reference's golden
vectors, the oracle and the exact-f32 HIP kernels."""
import numpy as np
import pytest
import torch

import helpers
import sc_oracle

pytestmark = pytest.mark.gpu

TOL = {'f16x3': (5e-6, 2e-6), 'bf16x3': (2e-5, 1e-5)}   # rel l2, flip size


@pytest.fixture(scope='module')
def sub():
  from analysis_transforms.fully_connected import subspace_ista_fista
  return subspace_ista_fista


@pytest.fixture(scope='module')
def fc():
  from analysis_transforms.fully_connected import ista_fista
  return ista_fista


def _routes_to_the_streamed_kernel(b, slots, m):
  import vtc_hip
  lib = vtc_hip.load_library()
  # the streamed kernel asks for two fragment-order iterates of the padded
  # codes on top of what the tiled path needs
  return lib.vtc_subspace_ista_fista_workspace_bytes(b, 256, slots // m, m) >= (
      2 * ((b + 31) // 32) * 32 * slots * 4)


@pytest.mark.parametrize('prec', ['f16x3', 'bf16x3'])
def test_mini_config4_against_the_reference(device, sub, prec):
  """64 groups of 8 over a 512-atom dictionary: the reference's own codes."""
  g = helpers.load('subspace')
  X = helpers.to_dev(helpers.gaussian_patches(24, 32, 256), device)
  D = helpers.to_dev(helpers.unit_rows(25, 512, 256), device)
  groups = [list(map(int, x)) for x in np.array_split(np.arange(512), 64)]
  assert _routes_to_the_streamed_kernel(32, 512, 8)
  codes = sub.run(X, D, groups, 0.008, 50, precision=prec)
  helpers.assert_codes_match(codes.cpu().numpy(), g['c4_codes_fista'],
                             TOL[prec][0], 'mini config 4 ' + prec,
                             max_flip_mag=TOL[prec][1])
  # the default policy picks the streamed f16x3 kernel for this shape
  auto = sub.run(X, D, groups, 0.008, 50)
  f16 = sub.run(X, D, groups, 0.008, 50, precision='f16x3')
  assert torch.equal(auto, f16)


@pytest.mark.parametrize('m', [2, 4, 8])
@pytest.mark.parametrize('variant', ['ista', 'fista'])
def test_group_sizes_against_oracle(device, sub, m, variant):
  """768 slots (6 phases) in groups of m, ragged batch, warm start."""
  s_atoms, b = 768, 45
  groups = [list(range(g * m, g * m + m)) for g in range(s_atoms // m)]
  Xn = helpers.gaussian_patches(700 + m, b, 256)
  Dn = helpers.unit_rows(701 + m, s_atoms, 256)
  eta = float(sc_oracle.fc_stepsize(torch.from_numpy(Dn)))
  ref = sc_oracle.subspace_ista_fista(torch.from_numpy(Xn),
                                      torch.from_numpy(Dn), groups, 0.02, 30,
                                      variant=variant)
  X, D = helpers.to_dev(Xn, device), helpers.to_dev(Dn, device)
  out = sub.run(X, D, groups, 0.02, 30, variant=variant, precision='f16x3')
  helpers.assert_codes_match(out.cpu().numpy(), ref.numpy(), 5e-6,
                             'groups of %d %s' % (m, variant),
                             max_flip_mag=2e-6)
  init = ref.clone()
  warm_ref = sc_oracle.subspace_ista_fista(
      torch.from_numpy(Xn), torch.from_numpy(Dn), groups, 0.02, 7,
      variant=variant, initial_codes=init)
  init_dev = helpers.to_dev(ref.numpy(), device)
  keep = init_dev.clone()
  warm = sub.run(X, D, groups, 0.02, 7, variant=variant, precision='f16x3',
                 initial_codes=init_dev, stepsize=eta)
  assert torch.equal(init_dev, keep)
  helpers.assert_codes_match(warm.cpu().numpy(), warm_ref.numpy(), 5e-6,
                             'warm start, groups of %d' % m, max_flip_mag=2e-6)


@pytest.mark.parametrize('s_atoms', [1280, 2048])
@pytest.mark.parametrize('nonneg,hard', [(False, False), (True, False),
                                         (False, True), (True, True)])
def test_fully_connected_beyond_1024_atoms(device, fc, s_atoms, nonneg, hard):
  """The fully-connected plugin on 16x16 patches with more atoms than the
  on-chip state holds: all four thresholds against the oracle."""
  b = 37
  Xn = helpers.gaussian_patches(800, b, 256)
  Dn = helpers.unit_rows(801, s_atoms, 256)
  eta = sc_oracle.fc_stepsize(torch.from_numpy(Dn))
  ref = sc_oracle.fc_ista_fista(torch.from_numpy(Xn), torch.from_numpy(Dn),
                                0.02, 25, stepsize=eta, nonnegative_only=nonneg,
                                hard_threshold=hard)
  X, D = helpers.to_dev(Xn, device), helpers.to_dev(Dn, device)
  out = fc.run(X, D, 0.02, 25, nonnegative_only=nonneg, hard_threshold=hard)
  import vtc_hip
  assert fc._resolve_precision(None, b, 256, s_atoms, None) == vtc_hip.F16X3
  if hard:
    flips = helpers.support_mismatch(out.cpu().numpy(), ref.numpy())
    assert flips <= 4
    same = (out.cpu() != 0) == (ref != 0)
    assert helpers.rel_err((out.cpu() * same).numpy(),
                           (ref * same).numpy()) < 3e-5
  else:
    helpers.assert_codes_match(out.cpu().numpy(), ref.numpy(), 5e-6,
                               'fc %d atoms' % s_atoms, max_flip_mag=2e-6)


def test_scale_invariance_and_reproducibility(device, sub):
  s_atoms, m, b = 512, 8, 64
  groups = [list(range(g * m, g * m + m)) for g in range(s_atoms // m)]
  X = helpers.to_dev(helpers.gaussian_patches(900, b, 256), device)
  D = helpers.to_dev(helpers.unit_rows(901, s_atoms, 256), device)
  base = sub.run(X, D, groups, 0.01, 20, precision='f16x3', stepsize=0.2)
  again = sub.run(X, D, groups, 0.01, 20, precision='f16x3', stepsize=0.2)
  assert torch.equal(base, again)
  for k in (-30, 12):
    f = float(2.0 ** k)
    out = sub.run(X * f, D, groups, 0.01 * f, 20, precision='f16x3',
                  stepsize=0.2)
    assert torch.equal(out, base * f), k
  # a slice of the batch reproduces the same rows bit for bit
  part = sub.run(X[32:64].contiguous(), D, groups, 0.01, 20,
                 precision='f16x3', stepsize=0.2)
  assert torch.equal(part, base[32:64])


def test_full_size_against_the_exact_kernels(device, sub):
  """configs[3] (4096 atoms in 512 groups of 8), 2048 patches: the streamed
  kernel against the exact-f32 HIP path on a slice, both precisions."""
  b, s_atoms = 2048, 4096
  X = helpers.to_dev(helpers.gaussian_patches(31, b, 256), device)
  D = helpers.to_dev(helpers.unit_rows(32, s_atoms, 256), device)
  groups = [list(map(int, x)) for x in np.array_split(np.arange(s_atoms), 512)]
  eta = 0.05
  exact = sub.run(X[:256].contiguous(), D, groups, 0.008, 30, precision='f32',
                  stepsize=eta)
  for prec, tol in (('f16x3', 5e-6), ('bf16x3', 5e-5)):
    full = sub.run(X, D, groups, 0.008, 30, precision=prec, stepsize=eta)
    helpers.assert_codes_match(full[:256].cpu().numpy(), exact.cpu().numpy(),
                               tol, 'streamed %s vs f32 path' % prec,
                               max_flip_mag=5e-6)
