"""Time the reference's own convolutional example geometry
(examples/train_convolutional_sparse_coding.py:25-39: 64 kernels of 16x16,
stride 8, 256x256 images, batch 5) through the strided direct-f32 kernels.

  python3 tools/time_conv_example.py [iters]
"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'vision-transform-codes_amd'))
import numpy as np, torch
from analysis_transforms.convolutional import ista_fista
from dict_update_rules.convolutional import sc_cheap_quadratic_descent
from utils import convolutions
dev = torch.device('cuda:0')
b, s, k, stride, img = 5, 64, 16, 8, 256
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
lead, trail = convolutions.get_padding_amt(img, k, stride)
rs = np.random.RandomState(0)
X = np.zeros((b, 1, img + lead + trail, img + lead + trail), np.float32)
X[:, :, lead:lead + img, lead:lead + img] = 0.1 * rs.randn(b, 1, img, img)
D = rs.randn(s, 1, k, k).astype(np.float32)
D /= np.sqrt((D ** 2).sum(axis=(1, 2, 3)))[:, None, None, None]
X, D = torch.from_numpy(X).to(dev), torch.from_numpy(D).to(dev)
padding = ((lead, trail), (lead, trail))
for _ in range(3):
  torch.cuda.synchronize(); t = time.time()
  codes = ista_fista.run(X, D, (stride, stride), padding, 0.05, iters)
  torch.cuda.synchronize(); dt = time.time() - t
print('inference: b=%d %d iters  %.2f ms  = %.1f us per iteration (2 launches)  codes %s nnz %.3f' % (
    b, iters, dt * 1e3, dt * 1e6 / iters, tuple(codes.shape), float((codes != 0).float().mean())))
h = torch.full((s,), 0.01, device=dev)
for _ in range(3):
  torch.cuda.synchronize(); t = time.time()
  sc_cheap_quadratic_descent.run(X, D, codes, h, (stride, stride), padding, stepsize=0.005)
  torch.cuda.synchronize(); dt = time.time() - t
print('dictionary update: %.3f ms' % (dt * 1e3))
