"""Measured distance of every precision of the FC FISTA path from the
REFERENCE's own codes (tests/golden/fc_c2_mini.npz: 64 patches, 1024 atoms,
lambda 0.008, FISTA), with the fixture's eta and with the engine's own
Gram + Lanczos eta.  Writes a table to stdout; commit it under profiles/.

  python3 tools/precision_report.py > profiles/r02_precision_fc.txt
"""
import os
import sys

REPO = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(REPO, 'vision-transform-codes_amd'))
sys.path.insert(0, os.path.join(REPO, 'tests'))
import numpy as np
import torch

import helpers
from analysis_transforms.fully_connected import ista_fista

dev = torch.device('cuda:0')
g = helpers.load('fc_c2_mini')
X = helpers.to_dev(helpers.gaussian_patches(0, 64, 256), dev)
D = helpers.to_dev(helpers.unit_rows(1, 1024, 256), dev)
lam, eta = float(g['sparsity_weight']), float(g['stepsize'])
truth = g['codes_fista_T200_fp64']


def row(tag, codes, ref):
  codes = codes.cpu().numpy()
  flips = (codes != 0) != (ref != 0)
  mag = float(np.maximum(np.abs(codes), np.abs(ref))[flips].max()) if (
      flips.any()) else 0.0
  print('%-40s rel l2 %.3e   support flips %3d of %d   largest flipped '
        'magnitude %.1e' % (tag, helpers.rel_err(codes, ref),
                            int(flips.sum()), ref.size, mag))


print('FC FISTA, fixture fc_c2_mini (64 x 256 patches, 1024 atoms, lambda '
      '0.008), distance from the reference codes')
print("reference float32 vs float64 run of the same algorithm, T=200: "
      'rel l2 %.3e, %d flips' % (
          helpers.rel_err(g['codes_fista_T200'], truth),
          helpers.support_mismatch(g['codes_fista_T200'], truth)))
for prec in ('f32', 'f16x3', 'bf16x3', 'bf16'):
  for T in (1, 2, 20, 200):
    codes = ista_fista.run(X, D, lam, T, precision=prec, stepsize=eta)
    row('%-6s T=%-3d fixture eta' % (prec, T), codes,
        g['codes_fista_T%d' % T])
  codes = ista_fista.run(X, D, lam, 200, precision=prec)
  row('%-6s T=200 own Gram+Lanczos eta' % prec, codes, g['codes_fista_T200'])
  print('%-40s rel l2 %.3e' % (
      '%-6s T=200 vs float64 truth' % prec,
      helpers.rel_err(ista_fista.run(X, D, lam, 200, precision=prec,
                                     stepsize=eta).cpu().numpy(), truth)))
for prec in ('f16x3', 'bf16x3'):
  codes = ista_fista.run(X, D, lam, 50, variant='ista', precision=prec,
                         stepsize=eta)
  row('%-6s ISTA T=50' % prec, codes, g['codes_ista_T50'])
codes = ista_fista.run(X, D, lam, 50, variant='ista', precision='f32',
                       stepsize=eta)
row('f32    ISTA T=50', codes, g['codes_ista_T50'])

# the tiled contractions (shapes outside the fused kernels): 20x20 patches, 500
# atoms, against the CPU oracle (bit-identical to the reference on every
# fixture; there is no reference fixture at this shape)
sys.path.insert(0, os.path.join(REPO, 'oracle'))
import sc_oracle
Xn = helpers.gaussian_patches(81, 96, 400)
Dn = helpers.unit_rows(82, 500, 400)
eta_n = sc_oracle.fc_stepsize(torch.from_numpy(Dn))
print('tiled contractions, 96 x 400 patches, 500 atoms, lambda 0.02, distance '
      'from the oracle')
for T in (30, 200):
  ref = sc_oracle.fc_ista_fista(torch.from_numpy(Xn), torch.from_numpy(Dn),
                                0.02, T, stepsize=eta_n).numpy()
  for prec in ('f32', 'f16x3', 'bf16x3'):
    codes = ista_fista.run(helpers.to_dev(Xn, dev), helpers.to_dev(Dn, dev),
                           0.02, T, precision=prec, stepsize=float(eta_n))
    row('%-6s T=%-3d tiled n=400' % (prec, T), codes, ref)
