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
  out = torch.full((3,), -1.0, dtype=torch.float32, device=device)
  ws = vtc_hip.workspace(lib.vtc_lambda_max_workspace_bytes(g.shape[0]),
                         device)
  vtc_hip.check(lib.vtc_lambda_max(vtc_hip.ptr(g), g.shape[0], vtc_hip.ptr(out),
                                   vtc_hip.ptr(ws), ws.numel(),
                                   vtc_hip.current_stream(device)), 'lanczos')
  lam, inv, converged = out.tolist()
  assert converged == 1.0, 'the solve reported that it had not converged'
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


def test_clustered_top_and_the_convergence_flag(device, monkeypatch):
  """Dictionary learning produces near-duplicate atoms: the two top
  eigenvalues of the Gram matrix then sit close together, the case in which a
  fixed number of Lanczos steps could stop short -- from below, i.e. with a
  step size above 1 / L.  The kernel compares the top Ritz value of all steps
  with that of the steps up to 8 earlier and keeps going while they differ by
  more than 1e-6; the third output says whether they agreed."""
  import vtc_hip
  lib = vtc_hip.load_library()
  rs = np.random.RandomState(3)
  for gap in (1e-2, 1e-3, 1e-4, 1e-6):
    # spectrum: a flat bulk, and a top pair separated by `gap` relative
    n = 256
    evals = np.concatenate([np.linspace(0.5, 3.0, n - 2),
                            [4.0 * (1 - gap), 4.0]])
    Q, _ = np.linalg.qr(rs.randn(n, n))
    gram = (Q * evals) @ Q.T
    lam, _ = _lambda_max(gram, device)       # asserts the flag
    # a value inside the top pair is all the flag promises for a tiny gap
    assert lam <= 4.0 * (1 + 2e-6)
    assert lam >= 4.0 * (1 - max(gap, 1e-6) - 2e-6), (gap, lam)
  # near-duplicate atoms in a real dictionary
  D = helpers.unit_rows(7, 1024, 256).astype(np.float64)
  D[1] = D[0] + 1e-3 * rs.randn(256)
  D[2] = D[0] + 1e-3 * rs.randn(256)
  D[1] /= np.linalg.norm(D[1])
  D[2] /= np.linalg.norm(D[2])
  gram = D.T @ D
  ref = np.linalg.eigvalsh(gram)[-1]
  lam, _ = _lambda_max(gram, device)
  assert abs(lam - ref) / ref < 2e-6


def test_unconverged_solve_is_reported(device):
  """With the step limit forced down to 6 (test hook of the library) the top
  Ritz value is still moving: flag 0, and the plugin-side helpers raise like
  the reference does on a failed eigen-solve (ista_fista.py:75-79)."""
  import os
  import subprocess
  import sys
  code = (
      "import sys, numpy as np, torch\n"
      "sys.path.insert(0, %r)\n"
      "import vtc_hip\n"
      "rs = np.random.RandomState(1)\n"
      "D = rs.randn(1024, 256).astype(np.float32)\n"
      "D /= np.linalg.norm(D, axis=1, keepdims=True)\n"
      "Dg = torch.from_numpy(D).cuda()\n"
      "g = vtc_hip.gram(Dg, transpose_a=True)\n"
      "out = vtc_hip.lambda_max_device(g).tolist()\n"
      "assert out[2] == 0.0, out\n"
      "ref = float(torch.linalg.eigvalsh(g.double())[-1])\n"
      "assert out[0] < ref * (1 - 1e-6), (out, ref)\n"
      "try:\n"
      "  vtc_hip.stepsize_from_gram(g, Dg)\n"
      "except RuntimeError as e:\n"
      "  print('raised', e)\n"
      "else:\n"
      "  raise SystemExit('no exception')\n"
      "vtc_hip.stepsize_on_device(g, Dg)\n"
      "try:\n"
      "  vtc_hip.poll_spectrum_checks(block=True)\n"
      "except RuntimeError as e:\n"
      "  print('raised late', e)\n"
      "else:\n"
      "  raise SystemExit('no deferred exception')\n"
  ) % os.path.dirname(os.path.dirname(os.path.abspath(
      __import__('vtc_hip').__file__)))
  env = dict(os.environ, VTC_LANCZOS_MAX_STEPS='6')
  done = subprocess.run([sys.executable, '-c', code], env=env,
                        capture_output=True, text=True, timeout=300)
  assert done.returncode == 0, done.stdout + done.stderr
  assert 'raised late' in done.stdout
