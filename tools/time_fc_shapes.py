"""Fully-connected ISTA/FISTA on shapes outside the fused kernel's n = 256,
s in {256, 512, 1024}: which path `precision='auto'` takes, time, algorithmic
TFLOP/s (4 s n flops per patch-iteration).

  python3 tools/time_fc_shapes.py > profiles/r02_fc_other_shapes.txt
"""
import pathlib
import sys
import time

import numpy as np
import torch

REPO = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / 'vision-transform-codes_amd'))
import vtc_hip
from analysis_transforms.fully_connected import ista_fista

dev = torch.device('cuda:0')
NAMES = {vtc_hip.F32: 'f32 tiles', vtc_hip.BF16X3: 'bf16x3', vtc_hip.F16X3: 'f16x3',
         vtc_hip.BF16: 'bf16'}
print('%-34s %-10s %-34s %9s %9s' % ('shape', 'variant', 'path taken by precision=auto',
                                       'ms', 'TFLOP/s'))
for n, s, b, iters, variant in ((64, 64, 131072, 20, 'ista'),
                                (64, 128, 131072, 50, 'fista'),
                                (64, 192, 131072, 50, 'fista'),
                                (64, 256, 131072, 50, 'fista'),
                                (64, 512, 65536, 50, 'fista'),
                                (144, 576, 32768, 50, 'fista'),
                                (400, 1600, 16384, 50, 'fista'),
                                (256, 1024, 131072, 200, 'fista'),
                                (256, 1280, 32768, 50, 'fista'),
                                (256, 2048, 32768, 50, 'fista'),
                                (256, 4096, 8192, 50, 'fista')):
  rs = np.random.RandomState(n + s)
  X = torch.from_numpy((0.1 * rs.randn(b, n)).astype(np.float32)).to(dev)
  D = rs.randn(s, n).astype(np.float32)
  D /= np.linalg.norm(D, axis=1, keepdims=True)
  D = torch.from_numpy(D).to(dev)
  prec = ista_fista._resolve_precision(None, b, n, s, None)
  if n == 256 and s in (256, 512, 1024):
    path = 'fused kernel, ' + NAMES[prec]
  elif n == 256 and s > 1024 and s % 256 == 0 and prec != vtc_hip.F32:
    path = 'fused, streamed state, ' + NAMES[prec]
  elif ((n == 144 and s in (288, 576)) or (n == 64 and s in (256, 512))) and (
      prec == vtc_hip.F32):
    path = 'registers + L2 stream, exact f32'
  elif n == 64 and s in (64, 128, 192) and prec == vtc_hip.F32:
    path = 'on-chip 8x8 kernel, exact f32'
  else:
    path = 'tiled contractions, ' + NAMES[prec]
  best = 1e9
  for _ in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ista_fista.run(X, D, 0.008, iters, variant=variant, stepsize=0.05)
    torch.cuda.synchronize()
    best = min(best, time.perf_counter() - t0)
  print('%-34s %-10s %-34s %9.2f %9.1f' % (
      'n=%d s=%d b=%d T=%d' % (n, s, b, iters), variant, path, best * 1e3,
      4.0 * s * n * iters * b / best / 1e12))
