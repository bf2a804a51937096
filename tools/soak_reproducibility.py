"""Bitwise reproducibility soak of every inference / update route at shapes
whose rows are NOT multiples of 128 bytes and which are large enough to spread
over all XCDs: each route is run repeatedly on the same inputs and every run is
compared bit for bit with the first.  (The hazard it looks for: blocks on
different XCDs read-modify-writing parts of one cache line -- DESIGN.md 4.4.)

  python3 tools/soak_reproducibility.py [runs]
"""
import pathlib
import sys

import numpy as np
import torch

REPO = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / 'vision-transform-codes_amd'))
from analysis_transforms.fully_connected import ista_fista, subspace_ista_fista
from analysis_transforms.convolutional import ista_fista as conv_ista_fista
from dict_update_rules.fully_connected import sc_cheap_quadratic_descent as fc_upd
from dict_update_rules.convolutional import sc_steepest_descent as conv_upd

dev = torch.device('cuda:0')
RUNS = int(sys.argv[1]) if len(sys.argv) > 1 else 20


def rows(seed, s, n):
  D = np.random.RandomState(seed).randn(s, n).astype(np.float32)
  D /= np.linalg.norm(D, axis=1, keepdims=True)
  return torch.from_numpy(D).to(dev)


def soak(name, fn):
  first = fn()
  bad = 0
  for _ in range(RUNS - 1):
    out = fn()
    if not torch.equal(first, out):
      bad += 1
  print('%-64s %d runs, %d differ%s' % (name, RUNS, bad, '' if bad == 0 else '   <-- NOT reproducible'))
  return bad


total = 0
rs = np.random.RandomState(0)
# fully-connected, tiled paths (s % 32 != 0: rows of codes share cache lines)
for n, s, b, prec in ((144, 100, 20000, 'f32'), (144, 580, 20000, 'bf16x3'),
                      (144, 580, 20000, 'f16x3'), (64, 36, 60000, 'f32')):
  X = torch.from_numpy((0.1 * rs.randn(b, n)).astype(np.float32)).to(dev)
  D = rows(n + s, s, n)
  for variant in ('fista', 'ista'):
    total += soak('FC tiled %s n=%d s=%d b=%d %s' % (prec, n, s, b, variant),
                  lambda: ista_fista.run(X, D, 0.01, 12, variant=variant,
                                         stepsize=0.05, precision=prec))
# subspace: fused epilogue (groups of 4) and separate prox (groups of 3)
for m, G, prec in ((4, 75, 'bf16x3'), (4, 75, 'f16x3'), (3, 100, 'f32'),
                   (3, 100, 'bf16x3')):
  n, b = 144, 20000
  s = G * m
  X = torch.from_numpy((0.1 * rs.randn(b, n)).astype(np.float32)).to(dev)
  D = rows(7 * m + G, s, n)
  groups = [list(range(g * m, g * m + m)) for g in range(G)]
  try:
    total += soak('subspace %s m=%d groups=%d b=%d' % (prec, m, G, b),
                  lambda: subspace_ista_fista.run(X, D, groups, 0.01, 12,
                                                  stepsize=0.05, precision=prec))
  except Exception as e:   # a precision not offered for the shape
    print('subspace %s m=%d: skipped (%s)' % (prec, m, type(e).__name__))
# convolution: f32 unit-stride kernels, bf16x3 two-kernel route (16x16
# kernels), fused route (11x11), strided patch route
# (colour images and chunked synthesis planes: the two-kernel route of round 3)
for k, s, stride, prec, label, c in (
    (11, 24, 1, 'f32', 'f32 unit-stride', 1),
    (16, 40, 1, 'bf16x3', 'bf16x3 two kernels', 1),
    (11, 128, 1, 'bf16x3', 'bf16x3 fused', 1),
    (16, 40, 1, 'f16x3', 'f16x3 two kernels', 1),
    (11, 128, 1, 'f16x3', 'f16x3 fused', 1),
    (11, 96, 1, 'f16x3', 'f16x3 two kernels, 3 channels', 3),
    (16, 112, 1, 'f16x3', 'f16x3 two kernels, chunked planes', 1),
    (16, 32, 8, 'f32', 'f32 patch route', 1)):
  img, b = 200, 4
  pad = k - 1 if stride == 1 else 8
  X = np.zeros((b, c, img + 2 * pad, img + 2 * pad), np.float32)
  X[:, :, pad:pad + img, pad:pad + img] = 0.1 * rs.randn(b, c, img, img)
  D = rs.randn(s, c, k, k).astype(np.float32)
  D /= np.sqrt((D ** 2).sum(axis=(1, 2, 3)))[:, None, None, None]
  Xd, Dd = torch.from_numpy(X).to(dev), torch.from_numpy(D).to(dev)
  padding = ((pad, pad), (pad, pad))
  for variant in ('fista', 'ista'):
    total += soak('conv %s k=%d s=%d stride %d %s' % (label, k, s, stride, variant),
                  lambda: conv_ista_fista.run(Xd, Dd, (stride, stride), padding,
                                              0.02, 6, variant=variant,
                                              stepsize=0.5 / s, precision=prec))
# dictionary updates (in place on a copy)
n, s, b = 144, 100, 20000
X = torch.from_numpy((0.1 * rs.randn(b, n)).astype(np.float32)).to(dev)
D0 = rows(5, s, n)
C = ista_fista.run(X, D0, 0.01, 12, stepsize=0.05, precision='f32')
h = torch.full((s,), 0.01, device=dev)


def upd():
  D = D0.clone()
  fc_upd.run(X, D, C, h, stepsize=0.05)
  return D


total += soak('FC cheap-quadratic update n=%d s=%d' % (n, s), upd)
# odd pixel count: rows of 121 floats never end on a 128-byte line, the apply
# kernel's blocks own 32 rows (the smallest footprint that does)
n, s, b = 121, 1000, 20000
X = torch.from_numpy((0.1 * rs.randn(b, n)).astype(np.float32)).to(dev)
D0 = rows(6, s, n)
C = ista_fista.run(X, D0, 0.01, 12, stepsize=0.05, precision='f32')
h = torch.full((s,), 0.01, device=dev)
total += soak('FC cheap-quadratic update n=%d s=%d' % (n, s), upd)
print('TOTAL differing runs:', total)
sys.exit(1 if total else 0)
