"""Time the tiled bf16x3 path of the fully-connected plugin on a shape the
fused kernel does not cover (12x12 patches, 576 atoms).

  python3 tools/time_fc_generic.py
"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'vision-transform-codes_amd'))
import numpy as np, torch
from analysis_transforms.fully_connected import ista_fista
dev = torch.device('cuda:0')
b, n, s, iters = 32768, 144, 576, 50
rs = np.random.RandomState(0)
X = torch.from_numpy((0.1 * rs.randn(b, n)).astype(np.float32)).to(dev)
D = rs.randn(s, n).astype(np.float32)
D /= np.sqrt((D ** 2).sum(1))[:, None]
D = torch.from_numpy(D).to(dev)
for prec in ('f32', 'bf16x3'):
  for _ in range(3):
    torch.cuda.synchronize(); t = time.time()
    codes = ista_fista.run(X, D, 0.008, iters, precision=prec, stepsize=0.2)
    torch.cuda.synchronize(); dt = time.time() - t
  print('%s: b=%d n=%d s=%d T=%d  %.2f ms  %.0f patches/s  %.1f TFLOP/s  nnz %.3f' % (
      prec, b, n, s, iters, dt * 1e3, b / dt, 4.0 * s * n * b * iters / dt / 1e12,
      float((codes != 0).float().mean())))
