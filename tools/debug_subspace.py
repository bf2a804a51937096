import sys, pathlib
import numpy as np, torch
REPO = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / 'vision-transform-codes_amd')); sys.path.insert(0, str(REPO / 'tests')); sys.path.insert(0, str(REPO / 'oracle'))
import helpers, vtc_hip, sc_oracle
from analysis_transforms.fully_connected import subspace_ista_fista as sub
dev = torch.device('cuda:0')
g = helpers.load('subspace')
GROUPS4 = [list(range(4 * i, 4 * i + 4)) for i in range(16)]
X = helpers.to_dev(g['g4_images'], dev); D = helpers.to_dev(g['g4_dictionary'], dev)
gram = vtc_hip.gram(D, transpose_a=True)
print('lanczos eta', vtc_hip.stepsize_from_gram(gram, D), 'lib eta', float(1./torch.linalg.eigvalsh(gram)[-1]))
for prec in ('f32', 'bf16x3'):
  codes = sub.run(X, D, GROUPS4, 0.02, 40, precision=prec)
  print(prec, 'rel', helpers.rel_err(codes.cpu().numpy(), g['g4_codes_fista']), 'nnz', float((codes != 0).float().mean()), 'absmax', float(codes.abs().max()))
  codes = sub.run(X, D, GROUPS4, 0.02, 40, precision=prec, stepsize=float(sc_oracle.fc_stepsize(torch.from_numpy(g['g4_dictionary']))))
  print(prec, 'given eta: rel', helpers.rel_err(codes.cpu().numpy(), g['g4_codes_fista']))
