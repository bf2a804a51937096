"""bf16x3 against the exact-f32 HIP kernels after the full 200 iterations, on
sizes too large for the CPU oracle: relative l2 difference of the codes and
support flips (entries that are zero in one result only), subspace (configs[3]
geometry, 2048 patches) and convolutional (configs[4] geometry, one image,
convergent step 0.9/128).

  python3 tools/compare_precisions.py
"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'vision-transform-codes_amd'))
import numpy as np, torch
from analysis_transforms.fully_connected import subspace_ista_fista
from analysis_transforms.convolutional import ista_fista as conv

dev = torch.device('cuda:0')


def report(tag, a, b):
  diff = float((a - b).norm() / b.norm())
  flips = (a != 0) != (b != 0)
  mag = float(torch.maximum(a.abs(), b.abs())[flips].max()) if flips.any() else 0.0
  print('%-34s rel %.3e   support flips %d of %d (largest magnitude %.2e, largest code %.2e)   nnz %.3f' % (
      tag, diff, int(flips.sum()), a.numel(), mag, float(b.abs().max()), float((b != 0).float().mean())))


rs = np.random.RandomState(5)
b, n, s = 2048, 256, 4096
X = torch.from_numpy((0.1 * rs.randn(b, n)).astype(np.float32)).to(dev)
D = rs.randn(s, n).astype(np.float32)
D /= np.sqrt((D ** 2).sum(1))[:, None]
D = torch.from_numpy(D).to(dev)
groups = [list(map(int, x)) for x in np.array_split(np.arange(s), 512)]
for iters in (50, 200):
  f32 = subspace_ista_fista.run(X, D, groups, 0.008, iters, precision='f32')
  x3 = subspace_ista_fista.run(X, D, groups, 0.008, iters, precision='bf16x3')
  report('subspace T=%d bf16x3 vs f32' % iters, x3, f32)

k, s_k, img, pad = 11, 128, 256, 10
imgs = np.zeros((1, 1, img + 2 * pad, img + 2 * pad), np.float32)
imgs[:, :, pad:-pad, pad:-pad] = 0.1 * rs.randn(1, 1, img, img)
K = rs.randn(s_k, 1, k, k).astype(np.float32)
K /= np.sqrt((K ** 2).sum(axis=(1, 2, 3)))[:, None, None, None]
Xc, Kc = torch.from_numpy(imgs).to(dev), torch.from_numpy(K).to(dev)
padding = ((pad, pad), (pad, pad))
for iters in (50, 200):
  f32 = conv.run(Xc, Kc, (1, 1), padding, 0.02, iters, stepsize=0.9 / 128, precision='f32')
  x3 = conv.run(Xc, Kc, (1, 1), padding, 0.02, iters, stepsize=0.9 / 128, precision='bf16x3')
  report('conv T=%d bf16x3 vs f32' % iters, x3, f32)
