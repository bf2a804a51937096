import os, sys, pathlib
import numpy as np, torch
REPO = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(REPO / 'vision-transform-codes_amd'))
sys.path.insert(0, str(REPO / 'tests'))
from analysis_transforms.convolutional import ista_fista
dev = torch.device('cuda:0')
b, s, k, img = 2, 128, 11, 256
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 2
pad = k - 1
rs = np.random.RandomState(0)
X = np.zeros((b, 1, img + 2 * pad, img + 2 * pad), np.float32)
X[:, :, pad:pad + img, pad:pad + img] = 0.1 * rs.randn(b, 1, img, img)
D = rs.randn(s, 1, k, k).astype(np.float32)
D /= np.sqrt((D ** 2).sum(axis=(1, 2, 3)))[:, None, None, None]
X, D = torch.from_numpy(X).to(dev), torch.from_numpy(D).to(dev)
padding = ((pad, pad), (pad, pad))
step = 0.9 / 128
outs = []
for rep in range(3):
  outs.append(ista_fista.run(X, D, (1, 1), padding, 0.05, iters, stepsize=step, precision='bf16x3').cpu().numpy())
os.environ['VTC_CONV_NO_FUSED'] = '1'
for a in range(1, 3):
  d = outs[a] != outs[0]
  print('run', a, 'vs 0: mismatches', int(d.sum()), 'of', d.size)
  if d.any():
    idx = np.argwhere(d)
    print(' images', np.unique(idx[:, 0]), 'atoms', np.unique(idx[:, 1])[:20], len(np.unique(idx[:, 1])))
    print(' rows', np.unique(idx[:, 2])[:40], len(np.unique(idx[:, 2])))
    print(' cols', np.unique(idx[:, 3])[:40], len(np.unique(idx[:, 3])))
    print(' max abs diff', float(np.abs(outs[a] - outs[0])[d].max()))
