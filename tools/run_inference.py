"""Runs the FC FISTA inference alone a few times (for rocprofv3 passes).

  python3 tools/run_inference.py --precision bf16 --batch 131072 --iters 200
"""
import argparse
import pathlib
import sys
import time

import numpy as np
import torch

REPO = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / 'vision-transform-codes_amd'))


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--precision', default='bf16')
  ap.add_argument('--batch', type=int, default=131072)
  ap.add_argument('--iters', type=int, default=200)
  ap.add_argument('--atoms', type=int, default=1024)
  ap.add_argument('--reps', type=int, default=3)
  args = ap.parse_args()
  from analysis_transforms.fully_connected import ista_fista
  dev = torch.device('cuda:0')
  X = torch.from_numpy((0.1 * np.random.RandomState(0).randn(
      args.batch, 256)).astype(np.float32)).to(dev)
  D = np.random.RandomState(1).randn(args.atoms, 256).astype(np.float32)
  D /= np.linalg.norm(D, axis=1, keepdims=True)
  D = torch.from_numpy(D).to(dev)
  eta = float(1.0 / torch.linalg.eigvalsh(D.t() @ D)[-1])
  for rep in range(args.reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    codes = ista_fista.run(X, D, 0.008, args.iters, precision=args.precision,
                           stepsize=eta)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    flops = 4.0 * args.atoms * 256 * args.iters * args.batch
    print('rep %d: %.2f ms  %.1f TFLOP/s algorithmic  nnz %.3f' % (
        rep, dt * 1e3, flops / dt / 1e12,
        float((codes != 0).float().mean())), flush=True)


if __name__ == '__main__':
  main()
