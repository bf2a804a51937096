"""Distance of the convolutional inference paths from the REFERENCE's own codes
(tests/golden/conv_long.npz, written by oracle/make_golden.py from the
reference run in the development container) at T = 10 / 50 / 100, per
precision mode.  Output goes to profiles/r03_precision_conv.txt.

  python3 tools/precision_conv_report.py
"""
import pathlib
import sys

import numpy as np
import torch

REPO = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / 'vision-transform-codes_amd'))
sys.path.insert(0, str(REPO / 'tests'))
import helpers  # noqa: E402


def main():
  from analysis_transforms.convolutional import ista_fista as conv
  dev = torch.device('cuda:0')
  g = helpers.load('conv_long')
  lam = float(g['sparsity_weight'])
  print('conv FISTA vs the reference\'s codes (conv_long.npz), lambda = %g' % lam)
  print('%-10s %-8s %-10s %5s %12s %8s %14s %12s' % (
      'case', 'mode', 'step from', 'T', 'rel-err', 'flips', 'largest flip',
      'max |code|'))
  for name in ('ex_k16s8', 'nd_k11s1'):
    imgs = torch.from_numpy(g[name + '_images_padded']).to(dev)
    D = torch.from_numpy(g[name + '_dictionary'].copy()).to(dev)
    stride = tuple(int(v) for v in g[name + '_stride'])
    pad = tuple(tuple(int(v) for v in row) for row in g[name + '_padding'])
    # the step size: the reference's own (fixture) and the engine's (Gram +
    # Lanczos on the device) -- they differ in the last bits, and at these
    # horizons the codes are ~100x as sensitive to the step as to anything else
    import vtc_hip
    flat = D.reshape(D.shape[0], -1)
    own = vtc_hip.stepsize_from_gram(vtc_hip.gram(flat, transpose_a=False), D)
    fixture = float(g[name + '_stepsize'])
    print('%-10s stepsize: reference %.9g, engine %.9g (relative difference '
          '%.2e)' % (name, fixture, own, abs(own - fixture) / fixture))
    for mode in ('f32', 'f16x3', 'bf16x3', 'auto'):
      for eta_name, eta in (('reference', fixture), ('engine', None)):
        for iters in (10, 50, 100):
          ref = g['%s_codes_fista_T%d' % (name, iters)]
          try:
            codes = conv.run(imgs, D, stride, pad, lam, iters, variant='fista',
                             precision=mode, stepsize=eta).cpu().numpy()
          except (NotImplementedError, ValueError) as e:
            print('%-10s %-8s %5d   unsupported (%s)' % (
                name, mode, iters, str(e)[:50]))
            continue
          err = helpers.rel_err(codes, ref)
          flips = (codes != 0) != (ref != 0)
          mag = np.maximum(np.abs(codes), np.abs(ref))[flips]
          print('%-10s %-8s %-10s %5d %12.3e %8d %14.3e %12.3e' % (
              name, mode, eta_name, iters, err, int(flips.sum()),
              float(mag.max()) if mag.size else 0.0,
              float(np.abs(ref).max())))


if __name__ == '__main__':
  main()
