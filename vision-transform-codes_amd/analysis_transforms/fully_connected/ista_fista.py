"""
ISTA / FISTA sparse inference for fully-connected dictionaries on MI355X.

Drop-in for the reference plugin of the same import path
(vision_transform_codes/analysis_transforms/fully_connected/ista_fista.py:14-148):
same `run` signature, keyword names, return value and non-mutation guarantees.
The arithmetic runs in libvtc_hip.so (vtc_fc_ista_fista): per iteration the two
contractions Y D - X and (.) D^T on the matrix cores with the shrinkage and the
FISTA extrapolation fused into the second contraction's epilogue.
"""
import ctypes

import torch

import vtc_hip


def run(images, dictionary, sparsity_weight, num_iters, variant='fista',
        initial_codes=None, early_stopping_epsilon=None,
        nonnegative_only=False, hard_threshold=False, precision=None,
        stepsize=None):
  """
  Infers sparse codes for a batch of flattened image patches.

  Parameters
  ----------
  images : torch.Tensor(float32, size=(b, n)) on a HIP device
  dictionary : torch.Tensor(float32, size=(s, n)), rows are basis functions
  sparsity_weight : float or 0-d tensor, the lambda of the LASSO objective
  num_iters : int, number of ISTA/FISTA steps (>= 1)
  variant : 'ista' or 'fista'
  initial_codes : torch.Tensor(float32, size=(b, s)), optional warm start;
      never written
  early_stopping_epsilon : float, optional.  Stop once the mean absolute code
      change per component, divided by the stepsize, drops below this (checked
      after every iteration but the first; costs a host sync per iteration).
  nonnegative_only, hard_threshold : bool, choose among the four thresholding
      functions of the reference (ista_fista.py:107-120)
  precision : None | 'auto' | 'f32' | 'f16x3' | 'bf16x3' | 'bf16' -- extension,
      see vtc_hip.set_default_precision.  None uses the process-wide default.
      'f16x3' / 'bf16x3' run the fused persistent kernel when the shape allows
      (n == 256, s in {256, 512, 1024}, no early stopping) and a tiled bf16
      hi/lo split contraction otherwise; 'bf16' exists only fused.
      'auto' = 'f16x3' for the fused kernel's shapes, 'bf16x3' for other large
      problems (b*s >= 2^22, n and s multiples of 4), 'f32' otherwise.
  stepsize : float, optional -- extension: skip the Lipschitz eigen-solve and
      use this eta (tests inject the eta of a golden vector this way).
      Without it (and without early stopping) eta stays on the device, as the
      reference's 0-d `stepsize` tensor does: the call only enqueues.

  Returns
  -------
  codes : torch.Tensor(float32, size=(b, s)), freshly allocated
  """
  assert variant in ['ista', 'fista']
  lib = vtc_hip.load_library()
  images = vtc_hip.require_device_tensor(images, 'images').contiguous()
  dictionary = vtc_hip.require_device_tensor(
      dictionary, 'dictionary').contiguous()
  b, n = images.shape
  s = dictionary.shape[0]
  assert dictionary.shape[1] == n
  vtc_hip.prepare_device(images.device)
  if initial_codes is not None:
    initial_codes = vtc_hip.require_device_tensor(
        initial_codes, 'initial_codes').contiguous()
    assert tuple(initial_codes.shape) == (b, s)
  if num_iters < 1:
    # the reference leaves `codes` unbound in this case
    raise UnboundLocalError(
        "local variable 'codes' referenced before assignment")

  mode = vtc_hip.threshold_mode(nonnegative_only, hard_threshold)
  prec = _resolve_precision(precision, b, n, s, early_stopping_epsilon)
  lam = float(sparsity_weight)
  eps = -1.0 if early_stopping_epsilon is None else float(
      early_stopping_epsilon)
  ws_bytes = lib.vtc_fc_ista_fista_workspace_bytes(b, n, s, prec)
  ws = vtc_hip.workspace(ws_bytes, images.device)
  codes = torch.empty((b, s), dtype=torch.float32, device=images.device)
  iters_run = ctypes.c_int(0)
  common = (int(num_iters), vtc_hip.variant_code(variant), mode, eps, prec,
            vtc_hip.ptr(ws), ws.numel(), ctypes.byref(iters_run),
            vtc_hip.current_stream(images.device))
  if (stepsize is None and early_stopping_epsilon is None and
      vtc_hip.device_stepsize_available(n)):
    # largest eigenvalue of D^T D (n x n), as ista_fista.py:72-80, kept on the
    # device like the reference's 0-d `stepsize` tensor: nothing in this call
    # waits for the GPU
    eta_dev = vtc_hip.stepsize_on_device(
        vtc_hip.gram(dictionary, transpose_a=True), dictionary)
    with vtc_hip.timed_call(images.device):
      status = lib.vtc_fc_ista_fista_dev(
          vtc_hip.ptr(images), vtc_hip.ptr(dictionary),
          vtc_hip.ptr(initial_codes), vtc_hip.ptr(codes), b, n, s,
          vtc_hip.ptr(eta_dev), lam, *common)
    vtc_hip.check(status, 'vtc_fc_ista_fista_dev')
  else:
    if stepsize is None:
      stepsize = vtc_hip.stepsize_from_gram(
          vtc_hip.gram(dictionary, transpose_a=True), dictionary)
    with vtc_hip.timed_call(images.device):
      status = lib.vtc_fc_ista_fista(
          vtc_hip.ptr(images), vtc_hip.ptr(dictionary),
          vtc_hip.ptr(initial_codes), vtc_hip.ptr(codes), b, n, s,
          float(stepsize), lam, *common)
    vtc_hip.check(status, 'vtc_fc_ista_fista')
  run.last_iters = iters_run.value
  return codes


def _resolve_precision(precision, b, n, s, early_stopping_epsilon):
  name = precision if precision is not None else (
      vtc_hip.get_default_precision())
  if name == 'auto':
    # the default policy: the f16 hi/lo split (three products on the 16-bit
    # matrix pipe, float32-level results: 2.5e-6 from the reference at T = 200)
    # wherever it pays -- the fused kernels' shapes, and large problems on the
    # tiled contraction; the exact f32 kernels for small or oddly sized ones.
    # (bf16x3, 1.75e-5, stays available by name and is never the default.)
    fused_ok = (early_stopping_epsilon is None and n == 256 and
                s in (256, 512, 1024) and fused_available())
    # (cut-over measured on one MI355X, T = 50: below ~2.4e8 multiply-adds per
    # product the 32x32-tile exact-f32 kernels win -- 1536 x 144 x 640: 1.6 vs
    # 1.9 ms, 600 x 400 x 1000: 2.1 vs 2.5 --, above it the split tiles --
    # 4096 x 400 x 1000: 3.7 vs 7.0, 3072 x 144 x 640: 2.1 vs 2.3)
    tiled_ok = (n % 4 == 0 and s % 4 == 0 and b * s * n >= 240000000 and
                fused_available())
    streamed_ok = (early_stopping_epsilon is None and n == 256 and
                   s > 1024 and s % 256 == 0 and fused_available())
    if fused_ok or streamed_ok:
      return vtc_hip.F16X3
    # 8x8 patches against 64 / 128 / 192 atoms: the exact-f32 on-chip kernel
    # (csrc/fc_small.hip) beats the split-bf16 tiles, which are HBM-bound there
    small_ok = (early_stopping_epsilon is None and n == 64 and
                s in (64, 128, 192))
    if small_ok:
      return vtc_hip.F32
    # 12x12 patches against 288 / 576 atoms, 8x8 against 256 / 512: exact-f32
    # kernel with the state in registers (csrc/fc_chip16.hip)
    if early_stopping_epsilon is None and (
        (n == 144 and s in (288, 576)) or (n == 64 and s in (256, 512))):
      return vtc_hip.F32
    return vtc_hip.F16X3 if tiled_ok else vtc_hip.F32
  return vtc_hip.PRECISIONS[name]


def fused_available():
  """True when libvtc_hip was built with the fused bf16 FISTA kernel."""
  lib = vtc_hip.load_library()
  return lib.vtc_fc_ista_fista_workspace_bytes(32, 256, 1024,
                                               vtc_hip.BF16X3) > 256
