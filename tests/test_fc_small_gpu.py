"""The on-chip kernels for small patches -- 8x8 (csrc/fc_small.hip: n = 64,
64 / 128 / 192 atoms) and 12x12 (csrc/fc_chip16.hip: n = 144, 288 / 576
atoms), exact f32 on the f32 matrix pipe -- against the CPU oracle: ISTA and
FISTA, the four thresholds, warm start, batches that are not multiples of a
wave's 32 patches, the engine's own step size from device memory.  Tolerance
of the exact-f32 path: 1e-5 relative on the codes, support identical above
2e-6."""
import numpy as np
import pytest
import torch

import helpers
import sc_oracle

pytestmark = pytest.mark.gpu


def _case(seed, b, s, n=64):
  rs = np.random.RandomState(seed)
  X = (0.1 * rs.randn(b, n)).astype(np.float32)
  D = rs.randn(s, n).astype(np.float32)
  D /= np.linalg.norm(D.astype(np.float64), axis=1, keepdims=True).astype(
      np.float32)
  return X, D


def test_policy_routes_8x8_patches_to_the_on_chip_kernel(device):
  import vtc_hip
  from analysis_transforms.fully_connected import ista_fista
  for s in (64, 128, 192):
    assert ista_fista._resolve_precision(None, 1 << 17, 64, s, None) == (
        vtc_hip.F32)
  for s in (288, 576):
    assert ista_fista._resolve_precision(None, 1 << 15, 144, s, None) == (
        vtc_hip.F32)
  lib = vtc_hip.load_library()
  # no workspace beyond the tiled path's is asked for; the shape is accepted
  assert lib.vtc_fc_ista_fista_workspace_bytes(1000, 64, 64, vtc_hip.F32) > 0


@pytest.mark.parametrize('s', [64, 128, 192])
@pytest.mark.parametrize('b', [1, 31, 32, 33, 1000])
def test_against_the_oracle(device, s, b):
  from analysis_transforms.fully_connected import ista_fista
  X, D = _case(100 * s + b, b, s)
  Xd, Dd = helpers.to_dev(X, device), helpers.to_dev(D, device)
  eta = float(sc_oracle.fc_stepsize(torch.from_numpy(D)))
  for kw in ({'variant': 'fista'}, {'variant': 'ista'},
             {'variant': 'fista', 'nonnegative_only': True},
             {'variant': 'ista', 'hard_threshold': True},
             {'variant': 'fista', 'hard_threshold': True,
              'nonnegative_only': True}):
    # a hard threshold is discontinuous: few iterations, flips at the cutoff
    iters = 3 if kw.get('hard_threshold') else 40
    ref = sc_oracle.fc_ista_fista(torch.from_numpy(X), torch.from_numpy(D),
                                  0.02, iters, stepsize=eta, **kw)
    out = ista_fista.run(Xd, Dd, 0.02, iters, stepsize=eta, precision='f32',
                         **kw)
    flip = 0.02 * eta * 1.01 if kw.get('hard_threshold') else 2e-6
    helpers.assert_codes_match(out.cpu().numpy(), ref.numpy(), 1e-5,
                               's=%d b=%d %r' % (s, b, kw), max_flip_mag=flip)


def test_warm_start_own_stepsize_and_reproducibility(device):
  from analysis_transforms.fully_connected import ista_fista
  X, D = _case(5, 777, 128)
  Xd, Dd = helpers.to_dev(X, device), helpers.to_dev(D, device)
  warm = sc_oracle.fc_ista_fista(torch.from_numpy(X), torch.from_numpy(D),
                                 0.02, 5)
  ref = sc_oracle.fc_ista_fista(torch.from_numpy(X), torch.from_numpy(D),
                                0.02, 30, initial_codes=warm)
  # default precision, default (device-side) step size
  a = ista_fista.run(Xd, Dd, 0.02, 30,
                     initial_codes=helpers.to_dev(warm.numpy(), device))
  b = ista_fista.run(Xd, Dd, 0.02, 30,
                     initial_codes=helpers.to_dev(warm.numpy(), device))
  helpers.assert_codes_match(a.cpu().numpy(), ref.numpy(), 1e-5,
                             'warm start, own eta', max_flip_mag=2e-6)
  assert torch.equal(a, b)


@pytest.mark.parametrize('n,s', [(144, 288), (144, 576), (64, 256), (64, 512)])
@pytest.mark.parametrize('b', [5, 16, 17, 300])
def test_register_resident_kernel_against_the_oracle(device, n, s, b):
  """csrc/fc_chip16.hip: 16 patches per wave, dictionary streamed from L2."""
  from analysis_transforms.fully_connected import ista_fista
  X, D = _case(7 * s + b, b, s, n=n)
  Xd, Dd = helpers.to_dev(X, device), helpers.to_dev(D, device)
  eta = float(sc_oracle.fc_stepsize(torch.from_numpy(D)))
  for kw in ({'variant': 'fista'}, {'variant': 'ista'},
             {'variant': 'fista', 'nonnegative_only': True},
             {'variant': 'fista', 'hard_threshold': True}):
    iters = 3 if kw.get('hard_threshold') else 30
    ref = sc_oracle.fc_ista_fista(torch.from_numpy(X), torch.from_numpy(D),
                                  0.02, iters, stepsize=eta, **kw)
    out = ista_fista.run(Xd, Dd, 0.02, iters, stepsize=eta, precision='f32',
                         **kw)
    flip = 0.02 * eta * 1.01 if kw.get('hard_threshold') else 2e-6
    helpers.assert_codes_match(out.cpu().numpy(), ref.numpy(), 1e-5,
                               '12x12 s=%d b=%d %r' % (s, b, kw),
                               max_flip_mag=flip)
  warm = sc_oracle.fc_ista_fista(torch.from_numpy(X), torch.from_numpy(D),
                                 0.02, 4, stepsize=eta)
  ref = sc_oracle.fc_ista_fista(torch.from_numpy(X), torch.from_numpy(D),
                                0.02, 20, stepsize=eta, initial_codes=warm)
  a = ista_fista.run(Xd, Dd, 0.02, 20, stepsize=eta,
                     initial_codes=helpers.to_dev(warm.numpy(), device))
  b2 = ista_fista.run(Xd, Dd, 0.02, 20, stepsize=eta,
                      initial_codes=helpers.to_dev(warm.numpy(), device))
  helpers.assert_codes_match(a.cpu().numpy(), ref.numpy(), 1e-5,
                             '12x12 warm start', max_flip_mag=2e-6)
  assert torch.equal(a, b2)


def test_misaligned_views_fall_back_to_the_tiled_path(device):
  """The register-resident kernels move patches and codes as float4; a view
  that starts 4 bytes into its storage takes the tiled path instead (and gives
  the same codes within the f32 tolerance)."""
  from analysis_transforms.fully_connected import ista_fista
  X, D = _case(11, 100, 64)
  eta = float(sc_oracle.fc_stepsize(torch.from_numpy(D)))
  ref = sc_oracle.fc_ista_fista(torch.from_numpy(X), torch.from_numpy(D),
                                0.02, 20, stepsize=eta)
  flat = torch.zeros(X.size + 1, dtype=torch.float32, device=device)
  flat[1:] = helpers.to_dev(X, device).reshape(-1)
  Xview = flat[1:].view(100, 64)
  assert Xview.data_ptr() % 16 == 4 and Xview.is_contiguous()
  out = ista_fista.run(Xview, helpers.to_dev(D, device), 0.02, 20,
                       stepsize=eta, precision='f32')
  helpers.assert_codes_match(out.cpu().numpy(), ref.numpy(), 1e-5,
                             'misaligned patches', max_flip_mag=2e-6)
