import sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent / 'vision-transform-codes_amd'))
import numpy as np, torch
from analysis_transforms.fully_connected import subspace_ista_fista
dev = torch.device('cuda:0')
b, s, n, iters = 8192, 4096, 256, 50
X = torch.from_numpy((0.1 * np.random.RandomState(0).randn(b, n)).astype(np.float32)).to(dev)
D = np.random.RandomState(1).randn(s, n).astype(np.float32)
D /= np.linalg.norm(D, axis=1, keepdims=True)
D = torch.from_numpy(D).to(dev)
groups = [list(map(int, g)) for g in np.array_split(np.arange(s), 512)]
for _ in range(2):
  subspace_ista_fista.run(X, D, groups, 0.008, iters, stepsize=0.05)
torch.cuda.synchronize()
