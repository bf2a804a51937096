"""Time BASELINE configs[4] (128 kernels 11x11, 256x256 images, b=8).

  python3 tools/time_conv.py [f32] [bf16x3]      (IMG=<side> to change the image)
"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'vision-transform-codes_amd'))
import numpy as np, torch
from analysis_transforms.convolutional import ista_fista
dev = torch.device('cuda:0')
b, s, k, img, iters = 8, 128, 11, int(os.environ.get('IMG', '256')), 20
pad = k - 1
rs = np.random.RandomState(0)
X = np.zeros((b, 1, img + 2 * pad, img + 2 * pad), np.float32)
X[:, :, pad:pad + img, pad:pad + img] = 0.1 * rs.randn(b, 1, img, img)
D = rs.randn(s, 1, k, k).astype(np.float32)
D /= np.sqrt((D ** 2).sum(axis=(1, 2, 3)))[:, None, None, None]
X, D = torch.from_numpy(X).to(dev), torch.from_numpy(D).to(dev)
padding = ((pad, pad), (pad, pad))
for prec in (sys.argv[1:] or ['bf16x3']):
  for _ in range(2):
    torch.cuda.synchronize(); t = time.time()
    ista_fista.run(X, D, (1, 1), padding, 0.02, iters, precision=prec, stepsize=0.005)
    torch.cuda.synchronize(); dt = time.time() - t
  print(prec, 'img', img, '%.2f ms  %.3f ms/image-iter  %.3f ns/position-iter' % (dt * 1e3, dt * 1e3 / (b * iters), dt * 1e9 / (b * iters * (img + k - 1) ** 2)))
