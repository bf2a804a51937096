import os, sys, pathlib
import numpy as np, torch
REPO = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(REPO / 'vision-transform-codes_amd'))
from analysis_transforms.convolutional import ista_fista
dev = torch.device('cuda:0')
b, s, k, img = 2, 128, 11, 256
pad = k - 1
rs = np.random.RandomState(0)
X = np.zeros((b, 1, img + 2 * pad, img + 2 * pad), np.float32)
X[:, :, pad:pad + img, pad:pad + img] = 0.1 * rs.randn(b, 1, img, img)
D = rs.randn(s, 1, k, k).astype(np.float32)
D /= np.sqrt((D ** 2).sum(axis=(1, 2, 3)))[:, None, None, None]
X, D = torch.from_numpy(X).to(dev), torch.from_numpy(D).to(dev)
padding = ((pad, pad), (pad, pad))
step = 0.9 / 128
def run(n):
  return ista_fista.run(X, D, (1, 1), padding, 0.05, n, stepsize=step, precision='bf16x3')
found = 0
for n in (2, 3, 4):
  ref = run(n)
  for rep in range(25):
    out = run(n)
    d = (out != ref)
    if bool(d.any()):
      idx = torch.nonzero(d).cpu().numpy()
      print('iters', n, 'rep', rep, 'mismatches', len(idx))
      for im in np.unique(idx[:, 0]):
        for ch in (0, 1):
          sel = idx[(idx[:, 0] == im) & (idx[:, 1] // 64 == ch)]
          if len(sel):
            print('  img', im, 'chunk', ch, 'n', len(sel), 'atoms', len(np.unique(sel[:, 1])),
                  'rows', sel[:, 2].min(), sel[:, 2].max(), 'cols', sel[:, 3].min(), sel[:, 3].max(),
                  'maxdiff', float((out - ref).abs()[d].max()))
      found += 1
      if found >= 6: sys.exit(0)
print('done, found', found)
