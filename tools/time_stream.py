"""Times the streamed fused kernel: element-wise (FC, m = 1) and group prox.
  python3 tools/time_stream.py [atoms] [batch] [iters]"""
import sys, pathlib, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent / 'vision-transform-codes_amd'))
import numpy as np, torch
from analysis_transforms.fully_connected import ista_fista, subspace_ista_fista
dev = torch.device('cuda:0')
s = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
b = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 100
X = torch.from_numpy((0.1 * np.random.RandomState(0).randn(b, 256)).astype(np.float32)).to(dev)
D = np.random.RandomState(1).randn(s, 256).astype(np.float32)
D /= np.linalg.norm(D, axis=1, keepdims=True)
D = torch.from_numpy(D).to(dev)
def timed(fn):
  fn(); torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize()
  return (time.perf_counter() - t) * 1e3
for prec in ('f16x3', 'bf16x3'):
  ms = timed(lambda: ista_fista.run(X, D, 0.008, iters, precision=prec, stepsize=0.05))
  print('fc   s=%d b=%d %s: %.2f ms = %.1f us/iter' % (s, b, prec, ms, ms * 1e3 / iters))
  for m in (2, 8):
    groups = [list(range(g * m, g * m + m)) for g in range(s // m)]
    ms = timed(lambda: subspace_ista_fista.run(X, D, groups, 0.008, iters, precision=prec, stepsize=0.05))
    print('sub  s=%d b=%d m=%d %s: %.2f ms = %.1f us/iter' % (s, b, m, prec, ms, ms * 1e3 / iters))
