"""Randomised parity sweep: random small shapes through every inference path
(fully-connected f32 / tiled bf16x3 / fused, subspace, convolutional: strided
patch contractions, stride-1 f32 and bf16x3 kernels) and the cheap-quadratic
update, each case against the CPU oracle.  Tolerance: 1e-5 relative on the
codes with support flips only within 2e-6 of the threshold for the exact-f32
and f16x3 paths, 2e-5 / 1e-5 for bf16x3; dictionary 5e-6."""
import traceback

import numpy as np
import pytest
import torch

import sc_oracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('seed', [1, 2, 3])
def test_random_shapes_against_the_oracle(device, seed):
  from analysis_transforms.fully_connected import ista_fista
  from analysis_transforms.fully_connected import subspace_ista_fista
  from analysis_transforms.convolutional import ista_fista as conv
  from dict_update_rules.fully_connected import (
      sc_cheap_quadratic_descent as fc_cq)
  dev = device
  rs = np.random.RandomState(seed)
  cases = 40
  failures = []

  def report(msg):
    failures.append(msg)

  def unit(s, *shape):
    D = rs.randn(s, *shape).astype(np.float32)
    D /= np.sqrt((D.reshape(s, -1).astype(np.float64) ** 2).sum(1)).reshape(
        (s,) + (1,) * len(shape)).astype(np.float32)
    return D


  def check(tag, ours, ref, tol, flip=1e-5):
    ours, ref = ours.cpu().numpy(), ref.numpy()
    err = np.linalg.norm(ours - ref) / max(np.linalg.norm(ref), 1e-30)
    flips = (ours != 0) != (ref != 0)
    worst = float(np.maximum(np.abs(ours), np.abs(ref))[flips].max()) if flips.any() else 0.0
    ok = np.isfinite(ours).all() and err <= tol and worst <= flip
    if not ok:
      report('%s: rel %.2e flips %d worst %.2e' % (tag, err, int(flips.sum()), worst))
    return ok


  for c in range(cases):
    kind = rs.choice(['fc', 'fc', 'sub', 'conv', 'upd'])
    try:
      if kind in ('fc', 'upd'):
        b = int(rs.choice([1, 7, 32, 33, 100, 257]))
        n = int(rs.choice([16, 36, 64, 100, 144, 256]))
        s = int(rs.choice([8, 20, 64, 128, 200, 256, 512]))
        X = (0.1 * rs.randn(b, n)).astype(np.float32); D = unit(s, n)
        lam = float(rs.choice([0.005, 0.02, 0.05])); T = int(rs.choice([1, 3, 12, 30]))
        kw = {'variant': str(rs.choice(['ista', 'fista'])),
              'nonnegative_only': bool(rs.rand() < 0.3)}
        prec = str(rs.choice(['f32', 'bf16x3', 'f16x3'])) if (n % 4 == 0 and s % 4 == 0) else 'f32'
        eta = sc_oracle.fc_stepsize(torch.from_numpy(D))
        ref = sc_oracle.fc_ista_fista(torch.from_numpy(X), torch.from_numpy(D), lam, T, stepsize=eta, **kw)
        out = ista_fista.run(torch.from_numpy(X).to(dev), torch.from_numpy(D).to(dev), lam, T,
                             precision=prec, stepsize=float(eta), **kw)
        ok = check('fc b=%d n=%d s=%d T=%d %s %r' % (b, n, s, T, prec, kw), out, ref,
                   *((2e-5, 1e-5) if prec == 'bf16x3' else (1e-5, 2e-6)))
        if kind == 'upd' and ok:
          h = (0.01 + rs.rand(s)).astype(np.float32)
          refD = torch.from_numpy(D.copy())
          sc_oracle.fc_cheap_quadratic_descent(torch.from_numpy(X), refD, ref, torch.from_numpy(h), 0.1, 2)
          Dd = torch.from_numpy(D.copy()).to(dev)
          fc_cq.run(torch.from_numpy(X).to(dev), Dd, torch.from_numpy(ref.numpy()).to(dev),
                    torch.from_numpy(h).to(dev), stepsize=0.1, num_iters=2)
          ok = check('fc update b=%d n=%d s=%d' % (b, n, s), Dd, refD, 5e-6, flip=1.0)
      elif kind == 'sub':
        m = int(rs.choice([1, 2, 3, 4, 8, 16])); G = int(rs.choice([4, 9, 32, 65]))
        n = int(rs.choice([32, 64, 100])); b = int(rs.choice([1, 40, 130]))
        s = G * m
        X = (0.1 * rs.randn(b, n)).astype(np.float32); D = unit(s, n)
        groups = [list(range(g * m, g * m + m)) for g in range(G)]
        T = int(rs.choice([2, 10, 25]))
        prec = str(rs.choice(['f32', 'bf16x3', 'f16x3'])) if s % 4 == 0 and n % 4 == 0 else 'f32'
        ref = sc_oracle.subspace_ista_fista(torch.from_numpy(X), torch.from_numpy(D), groups, 0.03, T)
        out = subspace_ista_fista.run(torch.from_numpy(X).to(dev), torch.from_numpy(D).to(dev), groups,
                                      0.03, T, precision=prec)
        ok = check('sub b=%d n=%d G=%d m=%d T=%d %s' % (b, n, G, m, T, prec), out, ref,
                   *((1e-5, 2e-6) if prec == 'f32' or (prec == 'f16x3' and m != 3) else (2e-5, 1e-5)))
      else:
        k = int(rs.choice([4, 5, 8, 11])); st = int(rs.choice([1, 1, 2, 4]))
        if k % st: st = 1
        s = int(rs.choice([3, 8, 32, 40])); b = int(rs.choice([1, 2, 3]))
        h, w = int(rs.randint(20, 60)), int(rs.randint(20, 70))
        cch = int(rs.choice([1, 1, 2, 3]))          # image channels
        lead = k - st
        H = ((h + 2 * lead - k + st - 1) // st) * st + k; W = ((w + 2 * lead - k + st - 1) // st) * st + k
        imgs = np.zeros((b, cch, H, W), np.float32)
        imgs[:, :, lead:lead + h, lead:lead + w] = 0.3 * rs.randn(b, cch, h, w)
        pad = ((lead, H - lead - h), (lead, W - lead - w))
        D = unit(s, cch, k, k)
        T = int(rs.choice([1, 4, 9])); step = 0.5 / s
        prec = 'auto'
        ref = sc_oracle.conv_ista_fista(torch.from_numpy(imgs), torch.from_numpy(D), (st, st), pad, 0.05, T, stepsize=step)
        out = conv.run(torch.from_numpy(imgs).to(dev), torch.from_numpy(D).to(dev), (st, st), pad, 0.05, T,
                       stepsize=step, precision=prec)
        ok = check('conv b=%d c=%d s=%d k=%d st=%d %dx%d T=%d' % (b, cch, s, k, st, H, W, T), out, ref, 1e-5, 2e-6)
    except Exception:
      report('case %d (%s): %s' % (c, kind, traceback.format_exc()))
  assert not failures, '\n'.join(failures)
