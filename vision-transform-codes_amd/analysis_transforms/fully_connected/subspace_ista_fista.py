"""
Subspace (group-LASSO) ISTA / FISTA for fully-connected dictionaries on MI355X.

Drop-in for vision_transform_codes/analysis_transforms/fully_connected/
subspace_ista_fista.py:23-192.  Thresholding acts on the l2 norm of each group
of coefficients; groups may be ragged and may overlap, exactly as in the
reference, by working on the zero-padded (b, G, m) layout with a duplicated
(G*m, n) dictionary -- built here by HIP gather kernels instead of Python loops.
"""
import ctypes

import torch

import vtc_hip
from vtc_hip import groups as group_tables


def run(images, dictionary, group_assignments, sparsity_weight,
        num_iters, variant='fista', ret_summed_gduplicates=True,
        initial_codes=None, early_stopping_epsilon=None, hard_threshold=False,
        stepsize=None, precision=None):
  """
  images (b, n), dictionary (s, n): float32 tensors on a HIP device.
  group_assignments : list of index lists, e.g. [[0, 2, 5], [1], [2, 3, 4, 5]]
  Returns codes (b, s); an atom that belongs to several groups gets the sum of
  its per-group coefficients (ret_summed_gduplicates=True, the only mode the
  reference implements).  Extensions: `stepsize` (skip the eigen-solve) and
  `precision` in {None, 'auto', 'f32', 'f16x3', 'bf16x3'}: 'auto' (the default
  policy) uses the f16 hi/lo split (float32-level results) on the fused
  streamed kernel (16x16 patches) and on the tiled contractions for large
  problems (>= 1024 slots, groups of a power of two), the exact-f32 kernels
  otherwise.
  """
  assert variant in ['ista', 'fista']
  if hard_threshold:
    raise NotImplementedError('TODO')
  if not ret_summed_gduplicates:
    raise NotImplementedError('TODO')
  lib = vtc_hip.load_library()
  images = vtc_hip.require_device_tensor(images, 'images').contiguous()
  dictionary = vtc_hip.require_device_tensor(
      dictionary, 'dictionary').contiguous()
  b, n = images.shape
  vtc_hip.prepare_device(images.device)
  s = dictionary.shape[0]
  device = images.device
  stream = vtc_hip.current_stream(device)
  tables = group_tables.tables_for(group_assignments, s, device)
  slots, num_groups, m = tables.slots, tables.num_groups, tables.m
  if num_iters < 1:
    raise UnboundLocalError(
        "local variable 'grouped_codes_tensor' referenced before assignment")

  grouped_dictionary = torch.empty((slots, n), dtype=torch.float32,
                                   device=device)
  vtc_hip.check(lib.vtc_group_gather_rows(
      vtc_hip.ptr(dictionary), vtc_hip.ptr(tables.index),
      vtc_hip.ptr(tables.valid), vtc_hip.ptr(grouped_dictionary), slots, n,
      stream), 'vtc_group_gather_rows')
  if stepsize is None:
    # Lipschitz bound from the grouped dictionary (:115-123)
    stepsize = vtc_hip.stepsize_from_gram(
        vtc_hip.gram(grouped_dictionary, transpose_a=True), dictionary)

  initial_grouped = None
  if initial_codes is not None:
    initial_codes = vtc_hip.require_device_tensor(
        initial_codes, 'initial_codes').contiguous()
    assert tuple(initial_codes.shape) == (b, s)
    initial_grouped = torch.empty((b, slots), dtype=torch.float32,
                                  device=device)
    vtc_hip.check(lib.vtc_group_gather_cols(
        vtc_hip.ptr(initial_codes), vtc_hip.ptr(tables.index),
        vtc_hip.ptr(tables.valid), vtc_hip.ptr(initial_grouped), b, s, slots,
        stream), 'vtc_group_gather_cols')

  grouped_codes = torch.empty((b, slots), dtype=torch.float32, device=device)
  ws = vtc_hip.workspace(
      lib.vtc_subspace_ista_fista_workspace_bytes(b, n, num_groups, m), device)
  iters_run = ctypes.c_int(0)
  eps = -1.0 if early_stopping_epsilon is None else float(
      early_stopping_epsilon)
  name = precision if precision is not None else (
      vtc_hip.get_default_precision())
  if name == 'auto':
    # 16x16 patches, groups of 1/2/4/8 slots: the fused persistent kernel with
    # streamed state; other large problems whose groups are powers of two
    # (the proximal step rides in the gradient product's epilogue): the tiled
    # contractions -- both on the f16 hi/lo split, float32-level results;
    # everything else: exact f32 (the bf16 split, 1.75e-5 at T = 200, is not a
    # default any more)
    if (n == 256 and slots % 256 == 0 and m in (1, 2, 4, 8) and
        early_stopping_epsilon is None):
      name = 'f16x3'
    elif (slots >= 1024 and slots % 4 == 0 and n % 4 == 0 and
          m in (1, 2, 4, 8, 16, 32)):
      name = 'f16x3'
    else:
      name = 'f32'
  if name == 'bf16':
    raise NotImplementedError('subspace inference has no bf16 fast mode')
  vtc_hip.check(lib.vtc_subspace_ista_fista(
      vtc_hip.ptr(images), vtc_hip.ptr(grouped_dictionary),
      vtc_hip.ptr(initial_grouped), vtc_hip.ptr(grouped_codes), b, n,
      num_groups, m, float(stepsize), float(sparsity_weight), int(num_iters),
      vtc_hip.variant_code(variant), eps, vtc_hip.PRECISIONS[name],
      vtc_hip.ptr(ws), ws.numel(), ctypes.byref(iters_run), stream),
      'vtc_subspace_ista_fista')
  run.last_iters = iters_run.value

  codes = torch.empty((b, s), dtype=torch.float32, device=device)
  vtc_hip.check(lib.vtc_group_scatter_add(
      vtc_hip.ptr(grouped_codes), vtc_hip.ptr(tables.atom_ptr),
      vtc_hip.ptr(tables.atom_slots), vtc_hip.ptr(codes), b, s, slots,
      stream), 'vtc_group_scatter_add')
  return codes
