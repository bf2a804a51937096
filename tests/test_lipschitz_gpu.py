"""vtc_lambda_max (device Lanczos) against LAPACK's symmetric eigen-solver."""
import numpy as np
import pytest
import torch

import helpers

pytestmark = pytest.mark.gpu


def _lambda_max(mat, device):
  import vtc_hip
  lib = vtc_hip.load_library()
  g = helpers.to_dev(mat.astype(np.float32), device)
  out = torch.empty(2, dtype=torch.float32, device=device)
  ws = vtc_hip.workspace(lib.vtc_lambda_max_workspace_bytes(g.shape[0]),
                         device)
  vtc_hip.check(lib.vtc_lambda_max(vtc_hip.ptr(g), g.shape[0], vtc_hip.ptr(out),
                                   vtc_hip.ptr(ws), ws.numel(),
                                   vtc_hip.current_stream(device)), 'lanczos')
  lam, inv = out.tolist()
  return lam, inv


@pytest.mark.parametrize('s,n,seed', [(1024, 256, 1), (512, 256, 51),
                                      (64, 64, 11), (6, 16, 21), (256, 256, 3),
                                      (4096, 256, 5), (300, 200, 9),
                                      (800, 400, 13), (2048, 1024, 17),
                                      (300, 576, 19)])
def test_dictionary_grams(device, s, n, seed):
  """n <= 256: Krylov basis in LDS; 256 < n <= 1024 (20x20, 24x24, 32x32
  patches; the last case is rank deficient): basis in a global workspace."""
  D = helpers.unit_rows(seed, s, n).astype(np.float64)
  gram = D.T @ D
  ref = np.linalg.eigvalsh(gram)[-1]
  lam, inv = _lambda_max(gram, device)
  assert abs(lam - ref) / ref < 2e-6, (lam, ref)
  assert abs(inv - 1.0 / ref) * ref < 2e-6


def test_rank_deficient_and_degenerate(device):
  rs = np.random.RandomState(0)
  F = rs.randn(128, 121)                    # conv: s = 128 kernels of 11x11
  F /= np.linalg.norm(F, axis=1, keepdims=True)
  gram = F @ F.T                            # rank 121 < 128
  ref = np.linalg.eigvalsh(gram)[-1]
  lam, _ = _lambda_max(gram, device)
  assert abs(lam - ref) / ref < 2e-6
  lam, inv = _lambda_max(np.eye(7) * 3.0, device)   # one-step Krylov space
  assert abs(lam - 3.0) < 1e-6 and abs(inv - 1 / 3.0) < 1e-6
  lam, _ = _lambda_max(np.array([[2.0]]), device)
  assert abs(lam - 2.0) < 1e-6
  diag = np.diag(np.linspace(1.0, 2.0, 256))        # clustered top end
  lam, _ = _lambda_max(diag, device)
  assert abs(lam - 2.0) < 1e-5
  # rank 64 inside a 256 x 256 Gram (subspace test geometry: 64 atoms of 256
  # pixels): the Krylov space is exhausted after ~64 steps
  g = helpers.load('subspace')
  D = g['g4_dictionary'].astype(np.float64)
  ref = np.linalg.eigvalsh(D.T @ D)[-1]
  lam, _ = _lambda_max(D.T @ D, device)
  assert abs(lam - ref) / ref < 2e-6
  D = g['ro_dictionary'].astype(np.float64)           # rank 6 in 16 x 16
  ref = np.linalg.eigvalsh(D.T @ D)[-1]
  lam, _ = _lambda_max(D.T @ D, device)
  assert abs(lam - ref) / ref < 2e-6


def test_plugin_step_size_agrees_with_library_solver(device):
  import vtc_hip
  D = helpers.to_dev(helpers.unit_rows(1, 1024, 256), device)
  gram = vtc_hip.gram(D, transpose_a=True)
  ours = vtc_hip.stepsize_from_gram(gram, D)
  lib = float(1. / torch.linalg.eigvalsh(gram, UPLO='U')[-1])
  assert abs(ours - lib) / lib < 3e-6
  bad = D.clone()
  bad[3, 5] = float('nan')
  with pytest.raises(RuntimeError):
    vtc_hip.stepsize_from_gram(vtc_hip.gram(bad, transpose_a=True), bad)
