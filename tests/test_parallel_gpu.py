"""Data-parallel path on ONE GPU:

* a 1-rank "nccl" (= RCCL) process group with vtc_hip.parallel enabled through
  the update plugins and the training step: the collective is really issued
  and the results are bit-equal to the non-parallel run;
* shard emulation: the per-shard outputs of the C-ABI gradient / code-energy
  entry points are added (what the all-reduce does) and applied with the
  GLOBAL batch size, against the oracle's full-batch update.  The
  convolutional case proves that the Frobenius rescale is taken after the sum.
"""
import ctypes
import os
import socket

import numpy as np
import pytest
import torch

import helpers

pytestmark = pytest.mark.gpu


def _free_port():
  with socket.socket() as s:
    s.bind(('127.0.0.1', 0))
    return s.getsockname()[1]


@pytest.fixture(scope='module')
def one_rank_group(device):
  import torch.distributed as dist
  from vtc_hip import parallel
  if dist.is_initialized():
    pytest.skip('a process group already exists in this process')
  os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
  dist.init_process_group(
      'nccl', init_method='tcp://127.0.0.1:%d' % _free_port(), rank=0,
      world_size=1, device_id=device)
  yield parallel
  parallel.disable()
  dist.destroy_process_group()


def _fc_case(device, b=96, n=256, s=256):
  X = helpers.to_dev(helpers.gaussian_patches(70, b, n), device)
  D = helpers.to_dev(helpers.unit_rows(71, s, n), device)
  return X, D


def _conv_case(device):
  import sc_oracle
  lead, trail = sc_oracle.conv_padding_amount(24, 8, 4)
  side = 24 + lead + trail
  rs = np.random.RandomState(72)
  imgs = np.zeros((4, 1, side, side), np.float32)
  imgs[:, :, lead:lead + 24, lead:lead + 24] = (
      0.5 * rs.randn(4, 1, 24, 24)).astype(np.float32)
  K = rs.randn(12, 1, 8, 8).astype(np.float32)
  K /= np.sqrt((K.astype(np.float64) ** 2).sum(axis=(1, 2, 3)))[
      :, None, None, None].astype(np.float32)
  pad = ((lead, trail), (lead, trail))
  return imgs, K, (4, 4), pad


def test_fc_plugins_with_a_live_process_group(device, one_rank_group):
  from analysis_transforms.fully_connected import ista_fista
  from dict_update_rules.fully_connected import (sc_cheap_quadratic_descent,
                                                 sc_steepest_descent)
  parallel = one_rank_group
  X, D = _fc_case(device)
  codes = ista_fista.run(X, D, 0.02, 20)
  h = torch.full((D.shape[0],), 0.01, device=device)
  results = {}
  for on in (False, True):
    if on:
      parallel.enable()
    else:
      parallel.disable()
    before = parallel.collectives_issued
    D1, D2 = D.clone(), D.clone()
    sc_steepest_descent.run(X, D1, codes, stepsize=0.1, num_iters=2)
    sc_cheap_quadratic_descent.run(X, D2, codes, h, stepsize=0.1)
    results[on] = (D1, D2, parallel.collectives_issued - before)
  parallel.disable()
  assert results[False][2] == 0
  assert results[True][2] == 3      # one all-reduce per update iteration
  assert torch.equal(results[True][0], results[False][0])
  assert torch.equal(results[True][1], results[False][1])


def test_conv_plugin_with_a_live_process_group(device, one_rank_group):
  from analysis_transforms.convolutional import ista_fista
  from dict_update_rules.convolutional import sc_steepest_descent
  parallel = one_rank_group
  imgs, K, stride, pad = _conv_case(device)
  Xd = helpers.to_dev(imgs, device)
  codes = ista_fista.run(Xd, helpers.to_dev(K, device), stride, pad, 0.05, 6,
                         variant='ista')
  out = {}
  for on in (False, True):
    parallel.enable() if on else parallel.disable()
    Kd = helpers.to_dev(K.copy(), device)
    sc_steepest_descent.run(Xd, Kd, codes, stride, pad, stepsize=0.005)
    out[on] = Kd
  parallel.disable()
  assert torch.equal(out[True], out[False])


def test_training_step_with_a_live_process_group(device, one_rank_group):
  """Cheap-quadratic training steps: with data parallelism on, the code
  energy rides on the gradient's all-reduce -- ONE collective per step -- and
  the dictionary and the Hessian diagonal stay bit-equal."""
  from training import sparse_coding
  parallel = one_rank_group
  X, D0 = _fc_case(device, b=128)
  params = {
      'mode': 'fully-connected', 'num_epochs': 1,
      'code_inference_algorithm': 'fista',
      'inference_param_schedule': {
          0: {'sparsity_weight': 0.02, 'num_iters': 15}},
      'dictionary_update_algorithm': 'sc_cheap_quadratic_descent',
      'dict_update_param_schedule': {0: {'stepsize': 0.1, 'num_iters': 1}}}
  batches = [X[32 * i: 32 * i + 32] for i in range(4)]
  out = {}
  for on in (False, True):
    parallel.enable() if on else parallel.disable()
    before = parallel.collectives_issued
    D = D0.clone()
    state = sparse_coding.train_dictionary(batches, batches, D, dict(params))
    out[on] = (D, state.hessian_diag.clone(),
               parallel.collectives_issued - before)
  parallel.disable()
  assert out[True][2] == 4 and out[False][2] == 0
  assert torch.equal(out[True][0], out[False][0])
  assert torch.equal(out[True][1], out[False][1])


# --------------------------------------------------------------------------
# shard emulation through the C ABI
# --------------------------------------------------------------------------
def _fc_gradient_sum(lib, X, D, C):
  import vtc_hip
  b, n = X.shape
  s = D.shape[0]
  ws = vtc_hip.workspace(lib.vtc_fc_dict_gradient_workspace_bytes(b, n, s),
                         X.device)
  g = torch.empty((s, n), dtype=torch.float32, device=X.device)
  vtc_hip.check(lib.vtc_fc_dict_gradient(
      vtc_hip.ptr(X), vtc_hip.ptr(D), vtc_hip.ptr(C), vtc_hip.ptr(g), b, n, s,
      vtc_hip.ptr(ws), ws.numel(), vtc_hip.current_stream(X.device)),
      'vtc_fc_dict_gradient')
  return g


def _energy_sum(lib, C):
  import vtc_hip
  b, s = C.shape[0], C.shape[1]
  positions = int(np.prod(C.shape[2:])) if C.dim() > 2 else 1
  ws = vtc_hip.workspace(lib.vtc_code_energy_workspace_bytes(b, s, positions),
                         C.device)
  e = torch.empty(s, dtype=torch.float32, device=C.device)
  vtc_hip.check(lib.vtc_code_energy(
      vtc_hip.ptr(C), b, s, positions, vtc_hip.ptr(e), vtc_hip.ptr(ws),
      ws.numel(), vtc_hip.current_stream(C.device)), 'vtc_code_energy')
  return e


@pytest.mark.parametrize('shards', [2, 8])
def test_fc_sharded_update_equals_full_batch(device, shards):
  """sum_r grad_r, sum_r energy_r, then EMA and apply with the global batch
  = the oracle's full-batch cheap-quadratic step."""
  import sc_oracle
  import vtc_hip
  from analysis_transforms.fully_connected import ista_fista
  lib = vtc_hip.load_library()
  b = 64 * shards
  X = helpers.to_dev(helpers.gaussian_patches(80, b, 256), device)
  D = helpers.to_dev(helpers.unit_rows(81, 512, 256), device)
  C = ista_fista.run(X, D, 0.02, 25)
  per = b // shards
  grad = torch.zeros_like(D)
  energy = torch.zeros(D.shape[0], device=device)
  for r in range(shards):
    rows = slice(per * r, per * (r + 1))
    grad += _fc_gradient_sum(lib, X[rows].contiguous(), D,
                             C[rows].contiguous())
    energy += _energy_sum(lib, C[rows].contiguous())
  h = torch.full((D.shape[0],), 0.004, device=device)
  stream = vtc_hip.current_stream(device)
  vtc_hip.check(lib.vtc_hessian_ema(vtc_hip.ptr(h), vtc_hip.ptr(energy), b,
                                    D.shape[0], stream), 'vtc_hessian_ema')
  Dn = D.clone()
  vtc_hip.check(lib.vtc_fc_dict_apply(
      vtc_hip.ptr(Dn), vtc_hip.ptr(grad), vtc_hip.ptr(h), vtc_hip.ptr(None),
      0.0, b, 0.1, 0.001, 1, D.shape[0], 256, stream), 'vtc_fc_dict_apply')
  # oracle, full batch, from the same codes
  Xc, Cc = X.cpu(), C.cpu()
  href = torch.full((D.shape[0],), 0.004)
  sc_oracle.hessian_diag_ema_(href, Cc)
  Dref = D.cpu().clone()
  sc_oracle.fc_cheap_quadratic_descent(Xc, Dref, Cc, href, stepsize=0.1)
  assert helpers.rel_err(h.cpu().numpy(), href.numpy()) < 2e-6
  assert helpers.rel_err(Dn.cpu().numpy(), Dref.numpy()) < helpers.REL_TOL_DICT
  assert helpers.rel_err(Dn.cpu().numpy(), D.cpu().numpy()) > 1e-4


def test_conv_sharded_update_rescales_after_the_sum(device):
  """Two image shards: gradients summed, THEN the Frobenius rescale inside
  vtc_conv_dict_apply -- equals the oracle's full-batch step; rescaling each
  shard first would not."""
  import sc_oracle
  import vtc_hip
  from analysis_transforms.convolutional import ista_fista
  from utils import convolutions
  lib = vtc_hip.load_library()
  imgs, K, stride, pad = _conv_case(device)
  Xd, Kd = helpers.to_dev(imgs, device), helpers.to_dev(K, device)
  codes = ista_fista.run(Xd, Kd, stride, pad, 0.05, 6, variant='ista')
  grad = torch.zeros_like(Kd)
  stream = vtc_hip.current_stream(device)
  for r in range(2):
    xs = Xd[2 * r: 2 * r + 2].contiguous()
    cs = codes[2 * r: 2 * r + 2].contiguous()
    geom = convolutions.geometry(xs, Kd, stride, pad)
    ws = vtc_hip.workspace(
        lib.vtc_conv_dict_gradient_workspace_bytes(ctypes.byref(geom)), device)
    g = torch.empty_like(Kd)
    vtc_hip.check(lib.vtc_conv_dict_gradient(
        vtc_hip.ptr(xs), vtc_hip.ptr(Kd), vtc_hip.ptr(cs), vtc_hip.ptr(g),
        ctypes.byref(geom), vtc_hip.F32, vtc_hip.ptr(ws), ws.numel(), stream),
        'vtc_conv_dict_gradient')
    grad += g
  Kn = Kd.clone()
  scratch = torch.empty_like(Kd)
  vtc_hip.check(lib.vtc_conv_dict_apply(
      vtc_hip.ptr(Kn), vtc_hip.ptr(grad), vtc_hip.ptr(None), 4, 0.005, 0.001,
      1, Kd.shape[0], 64, vtc_hip.ptr(scratch), stream),
      'vtc_conv_dict_apply')
  Kref = torch.from_numpy(K.copy())
  sc_oracle.conv_steepest_descent(torch.from_numpy(imgs), Kref, codes.cpu(),
                                  stride, pad, stepsize=0.005)
  assert helpers.rel_err(Kn.cpu().numpy(), Kref.numpy()) < 5e-6
  assert helpers.rel_err(Kn.cpu().numpy(), K) > 1e-5
