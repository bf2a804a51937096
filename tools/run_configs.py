"""Times the secondary configurations of BASELINE.json (configs[3], configs[4])
at full size.  Parity of these paths is covered by tests/; this prints speed.

  python3 tools/run_configs.py [subspace] [conv] [conv_geometries]
"""
import pathlib
import sys
import time

import numpy as np
import torch

REPO = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / 'vision-transform-codes_amd'))


def timed(fn, reps=2):
  best = float('inf')
  for _ in range(reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    best = min(best, time.perf_counter() - t0)
  return best, out


def subspace(dev):
  from analysis_transforms.fully_connected import subspace_ista_fista
  from dict_update_rules.fully_connected import (
      subspace_sc_cheap_quadratic_descent as upd)
  b, s, n, iters = 8192, 4096, 256, 200
  X = torch.from_numpy((0.1 * np.random.RandomState(0).randn(b, n)).astype(
      np.float32)).to(dev)
  D = np.random.RandomState(1).randn(s, n).astype(np.float32)
  D /= np.linalg.norm(D, axis=1, keepdims=True)
  D = torch.from_numpy(D).to(dev)
  groups = [list(map(int, g)) for g in np.array_split(np.arange(s), 512)]
  dt, codes = timed(lambda: subspace_ista_fista.run(X, D, groups, 0.008,
                                                    iters))
  flops = 4.0 * s * n * iters * b
  print('config 4 subspace: b=%d  %d-iter group FISTA  %.1f ms  %.0f patches/s'
        '  %.1f TFLOP/s  nnz %.3f' % (b, iters, dt * 1e3, b / dt,
                                      flops / dt / 1e12,
                                      float((codes != 0).float().mean())))
  h = torch.full((s,), 0.01, device=dev)
  dt, _ = timed(lambda: upd.run(X, D, codes, groups, h, 2e-4, stepsize=0.05))
  print('          update with alignment penalty: %.2f ms' % (dt * 1e3))


def conv(dev):
  from analysis_transforms.convolutional import ista_fista
  from dict_update_rules.convolutional import sc_steepest_descent
  b, s, k, img, iters = 8, 128, 11, 256, 20
  pad = k - 1
  rs = np.random.RandomState(0)
  X = np.zeros((b, 1, img + 2 * pad, img + 2 * pad), np.float32)
  X[:, :, pad:pad + img, pad:pad + img] = 0.1 * rs.randn(b, 1, img, img)
  D = rs.randn(s, 1, k, k).astype(np.float32)
  D /= np.sqrt((D ** 2).sum(axis=(1, 2, 3)))[:, None, None, None]
  X, D = torch.from_numpy(X).to(dev), torch.from_numpy(D).to(dev)
  padding = ((pad, pad), (pad, pad))
  flop_iter = 4.0 * s * k * k * 266 * 266
  for prec in ('f32', 'bf16x3', 'f16x3'):
    dt, codes = timed(lambda: ista_fista.run(X, D, (1, 1), padding, 0.02,
                                             iters, precision=prec))
    print('config 5 conv [%s]: b=%d  %d-iter conv FISTA  %.1f ms  = %.3f '
          'ms/image-iter  %.2f TFLOP/s  nnz %.3f' % (
              prec, b, iters, dt * 1e3, dt * 1e3 / (b * iters),
              flop_iter * b * iters / dt / 1e12,
              float((codes != 0).float().mean())))
  dt, _ = timed(lambda: sc_steepest_descent.run(X, D, codes, (1, 1), padding,
                                                stepsize=0.005))
  print('          conv dictionary update: %.2f ms' % (dt * 1e3))


def conv_geometries(dev):
  """Stride-1 geometries beside configs[4] on the matrix-core routes: the
  reference's own kernel size (16x16, vtc/tests/ista_fista_2.py:16-24) and
  colour images -- time per image-iteration and the distance of the split
  result from the exact-f32 kernels (convergent step, 20 iterations)."""
  from analysis_transforms.convolutional import ista_fista
  from dict_update_rules.convolutional import sc_steepest_descent
  iters, b, img = 20, 8, 256
  print('%-34s %-8s %10s %14s %12s' % ('geometry (stride 1, 256x256, b=8)',
                                       'mode', 'ms/img-it', 'TFLOP/s',
                                       'vs f32'))
  for k, c, s in ((16, 1, 64), (16, 1, 128), (11, 3, 128), (16, 3, 64),
                  (8, 3, 96), (5, 1, 128)):
    pad = k - 1
    rs = np.random.RandomState(k + c + s)
    X = np.zeros((b, c, img + 2 * pad, img + 2 * pad), np.float32)
    X[:, :, pad:pad + img, pad:pad + img] = 0.1 * rs.randn(b, c, img, img)
    D = rs.randn(s, c, k, k).astype(np.float32)
    D /= np.sqrt((D ** 2).sum(axis=(1, 2, 3)))[:, None, None, None]
    X, D = torch.from_numpy(X).to(dev), torch.from_numpy(D).to(dev)
    padding = ((pad, pad), (pad, pad))
    side = img + pad
    flop_iter = 4.0 * s * c * k * k * side * side
    exact = None
    for prec in ('f32', 'f16x3', 'bf16x3'):
      dt, codes = timed(lambda: ista_fista.run(
          X, D, (1, 1), padding, 0.02, iters, stepsize=0.9 / s,
          precision=prec))
      if exact is None:
        exact = codes
      diff = float(torch.linalg.norm((codes - exact).double()) /
                   torch.linalg.norm(exact.double()))
      print('%-34s %-8s %10.3f %14.1f %12.2e' % (
          '%d kernels %dx%d, %d channel(s)' % (s, k, k, c), prec,
          dt * 1e3 / (b * iters), flop_iter * b * iters / dt / 1e12, diff))
    dt, _ = timed(lambda: sc_steepest_descent.run(X, D.clone(), exact, (1, 1),
                                                  padding, stepsize=0.005))
    print('%-34s %-8s %10.2f ms per dictionary update' % ('', 'auto',
                                                          dt * 1e3))


if __name__ == '__main__':
  which = sys.argv[1:] or ['subspace', 'conv']
  dev = torch.device('cuda:0')
  if 'subspace' in which:
    subspace(dev)
  if 'conv' in which:
    conv(dev)
  if 'conv_geometries' in which:
    conv_geometries(dev)
