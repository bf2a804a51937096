"""Time one training step at the reference's fully-connected example size
(examples/train_sparse_coding.py:20-65: 16x16 patches, 256 atoms, batch 250,
FISTA 25/50/100 iterations, cheap-quadratic update).

  python3 tools/time_fc_example.py
"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'vision-transform-codes_amd'))
import numpy as np, torch
from training import sparse_coding
dev = torch.device('cuda:0')
rs = np.random.RandomState(0)
X = torch.from_numpy((0.1 * rs.randn(250, 256)).astype(np.float32)).to(dev)
D = rs.randn(256, 256).astype(np.float32)
D /= np.sqrt((D ** 2).sum(1))[:, None]
D = torch.from_numpy(D).to(dev)
for iters in (25, 100):
  params = {'mode': 'fully-connected', 'num_epochs': 1,
            'code_inference_algorithm': 'fista',
            'inference_param_schedule': {0: {'sparsity_weight': 0.008, 'num_iters': iters}},
            'dictionary_update_algorithm': 'sc_cheap_quadratic_descent',
            'dict_update_param_schedule': {0: {'stepsize': 0.1, 'num_iters': 1}}}
  step = sparse_coding.TrainingStep(D, params)
  step.sparsity_weight, step.inf_num_iters = 0.008, iters
  step.upd_stepsize, step.upd_num_iters = 0.1, 1
  for _ in range(5):
    step(X)
  torch.cuda.synchronize(); t = time.time()
  for _ in range(50):
    step(X)
  torch.cuda.synchronize(); dt = (time.time() - t) / 50
  print('T=%d: %.3f ms per training step (b=250) = %.0f patches/s' % (iters, dt * 1e3, 250 / dt))
