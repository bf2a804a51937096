"""Soak run: many training steps of the three coding modes through
train_dictionary (with validation metrics and checkpoints), watching device
memory and the finiteness / unit norm of the dictionary.

  python3 tools/soak.py [steps]
"""
import os, sys, tempfile, pathlib, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'vision-transform-codes_amd'))
import numpy as np, torch
from training import sparse_coding
from utils import convolutions

dev = torch.device('cuda:0')
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rs = np.random.RandomState(0)


def unit_rows(s, *shape):
  D = rs.randn(s, *shape).astype(np.float32)
  D /= np.sqrt((D.reshape(s, -1) ** 2).sum(1)).reshape((s,) + (1,) * len(shape))
  return torch.from_numpy(D).to(dev)


def run(tag, params, batches, val, D):
  torch.cuda.synchronize(); torch.cuda.reset_peak_memory_stats()
  base = torch.cuda.memory_allocated()
  t = time.time()
  with tempfile.TemporaryDirectory() as tmp:
    p = dict(params)
    p['logging_folder_fullpath'] = pathlib.Path(tmp)
    p['checkpoint_schedule'] = {0, steps // 2}
    p['training_visualization_schedule'] = set(range(0, steps, max(1, steps // 5)))
    state = sparse_coding.train_dictionary(batches, val, D, p)
  torch.cuda.synchronize()
  flat = D.reshape(D.shape[0], -1)
  norms = flat.norm(dim=1)
  print('%-12s %d steps in %.2f s  leak %d B  peak %.1f MiB  dict finite %s  |norm-1| max %.1e  metrics logged %d  last loss %.4g' % (
      tag, len(batches), time.time() - t, torch.cuda.memory_allocated() - base,
      torch.cuda.max_memory_allocated() / 2**20, bool(torch.isfinite(D).all()),
      float((norms - 1).abs().max()), len(state.metrics_log),
      float(state.metrics_log[-1][1]['Average LASSO Loss'])))


X = torch.from_numpy((0.1 * rs.randn(steps * 250, 256)).astype(np.float32)).to(dev)
fc = {'mode': 'fully-connected', 'num_epochs': 1, 'code_inference_algorithm': 'fista',
      'inference_param_schedule': {0: {'sparsity_weight': 0.008, 'num_iters': 25},
                                   steps // 2: {'sparsity_weight': 0.008, 'num_iters': 50}},
      'dictionary_update_algorithm': 'sc_cheap_quadratic_descent',
      'dict_update_param_schedule': {0: {'stepsize': 0.1, 'num_iters': 1}}}
run('fc', fc, [X[250 * i: 250 * i + 250] for i in range(steps)], [X[:250], X[250:500]], unit_rows(256, 256))
sub = dict(fc)
sub.update({'code_inference_algorithm': 'subspace_fista',
            'dictionary_update_algorithm': 'subspace_sc_cheap_quadratic_descent',
            'group_assignments': [list(range(4 * g, 4 * g + 4)) for g in range(64)],
            'subspace_alignment_penalty': 2e-4})
run('subspace', sub, [X[250 * i: 250 * i + 250] for i in range(steps)], [X[:250]], unit_rows(256, 256))
lead, trail = convolutions.get_padding_amt(64, 16, 8)
imgs = np.zeros((steps * 2, 1, 64 + lead + trail, 64 + lead + trail), np.float32)
imgs[:, :, lead:lead + 64, lead:lead + 64] = 0.1 * rs.randn(steps * 2, 1, 64, 64)
imgs = torch.from_numpy(imgs).to(dev)
cv = {'mode': 'convolutional', 'num_epochs': 1, 'code_inference_algorithm': 'fista',
      'strides': (8, 8), 'padding': ((lead, trail), (lead, trail)),
      'inference_param_schedule': {0: {'sparsity_weight': 0.02, 'num_iters': 20}},
      'dictionary_update_algorithm': 'sc_cheap_quadratic_descent',
      'dict_update_param_schedule': {0: {'stepsize': 0.005, 'num_iters': 1}}}
run('conv k16s8', cv, [imgs[2 * i: 2 * i + 2] for i in range(steps)], [imgs[:2]], unit_rows(32, 1, 16, 16))
