"""
Dictionary training loop for sparse coding on MI355X: the per-batch step.

Mirrors the calling convention of vision_transform_codes/training/
sparse_coding.py (train_dictionary at :9-10, parameter dictionary at :52-117)
for the part of it that is on the hot path: schedule lookup, code inference,
Hessian-diagonal EMA, dictionary update (:124-168, :444-517), plus dictionary
checkpoints (:170-175) and the validation metrics (:177-229, :497-505) as
device reductions.  TensorBoard image summaries and the interactive reset/prune
machinery are host-side bookkeeping outside this engine's scope: the former are
skipped (the metrics are kept in TrainingStep.metrics_log instead of a
SummaryWriter), the latter raises NotImplementedError if requested.

Data parallelism (not in the reference): when vtc_hip.parallel is enabled each
rank feeds its own shard of every batch; inference is local, the dictionary
gradient and the code energy are summed over ranks with ONE RCCL all-reduce
per update iteration (the energy rides in the gradient's buffer), and all
ranks apply the same update.
"""
import ctypes
import pickle
import time

import numpy as np
import torch

import vtc_hip
from vtc_hip import parallel

_INFERENCE_ALGS = ['ista', 'fista', 'subspace_ista', 'subspace_fista']
_UPDATE_ALGS = ['sc_steepest_descent', 'sc_cheap_quadratic_descent',
                'subspace_sc_steepest_descent',
                'subspace_sc_cheap_quadratic_descent']


def _load_plugins(coding_mode, code_inf_alg, dict_update_alg):
  """String -> plugin module, the same table as sparse_coding.py:389-439."""
  if code_inf_alg in ('ista', 'fista'):
    if coding_mode == 'fully-connected':
      from analysis_transforms.fully_connected import (
          ista_fista as inference_alg)
    else:
      from analysis_transforms.convolutional import (
          ista_fista as inference_alg)
  elif code_inf_alg in ('subspace_ista', 'subspace_fista'):
    if coding_mode != 'fully-connected':
      raise KeyError('Havent implemented subspace ISTA for convolutional yet')
    from analysis_transforms.fully_connected import (
        subspace_ista_fista as inference_alg)
  else:
    raise KeyError('Unrecognized code inference algorithm: ' + code_inf_alg)

  if dict_update_alg == 'sc_steepest_descent':
    if coding_mode == 'fully-connected':
      from dict_update_rules.fully_connected import (
          sc_steepest_descent as dict_update)
    else:
      from dict_update_rules.convolutional import (
          sc_steepest_descent as dict_update)
  elif dict_update_alg == 'sc_cheap_quadratic_descent':
    if coding_mode == 'fully-connected':
      from dict_update_rules.fully_connected import (
          sc_cheap_quadratic_descent as dict_update)
    else:
      from dict_update_rules.convolutional import (
          sc_cheap_quadratic_descent as dict_update)
  elif dict_update_alg == 'subspace_sc_steepest_descent':
    if coding_mode != 'fully-connected':
      raise KeyError('Not implemented for convolutional')
    # the reference imports a module that does not exist in its tree
    # (sparse_coding.py:423-424); keep the failure mode
    from dict_update_rules.fully_connected import (
        subspace_sc_steepest_descent as dict_update)
  elif dict_update_alg == 'subspace_sc_cheap_quadratic_descent':
    if coding_mode != 'fully-connected':
      raise KeyError('Not implemented for convolutional')
    from dict_update_rules.fully_connected import (
        subspace_sc_cheap_quadratic_descent as dict_update)
  else:
    raise KeyError('Unrecognized dict update algorithm: ' + dict_update_alg)
  return inference_alg, dict_update


class TrainingStep(object):
  """Holds what one batch step needs: plugins, schedules' current values, the
  aliased dictionary and the Hessian-diagonal estimate."""

  def __init__(self, dictionary, all_params):
    self.dictionary = dictionary   # alias of the caller's tensor (:444)
    self.mode = all_params['mode']
    self.inf_name = all_params['code_inference_algorithm']
    self.upd_name = all_params['dictionary_update_algorithm']
    assert self.mode in ['fully-connected', 'convolutional']
    assert self.inf_name in _INFERENCE_ALGS
    assert self.upd_name in _UPDATE_ALGS
    self.nonneg_only = all_params.get('nonnegative_only', False)
    self.hard_threshold = all_params.get('hard_threshold', False)
    self.groups = None
    if 'group_assignments' in all_params:
      groups = all_params['group_assignments']
      assert all([len(set(x)) == len(x) for x in groups])
      if type(groups[0]) != list:
        groups = [x.tolist() for x in groups]
      self.groups = groups
    if self.mode == 'convolutional':
      self.strides = all_params['strides']
      self.padding = all_params['padding']
      assert self.padding != ((0, 0), (0, 0)), 'Please use None instead'
    self.inference_alg, self.dict_update = _load_plugins(
        self.mode, self.inf_name, self.upd_name)
    if self.inf_name.startswith('subspace_'):
      assert self.groups is not None
    self.uses_hessian = self.upd_name in (
        'sc_cheap_quadratic_descent', 'subspace_sc_cheap_quadratic_descent')
    self.hessian_diag = (dictionary.new_zeros(dictionary.shape[0])
                         if self.uses_hessian else None)
    self.alignment_penalty = all_params.get('subspace_alignment_penalty')
    # current schedule values
    self.sparsity_weight = None
    self.inf_num_iters = None
    self.upd_stepsize = None
    self.upd_num_iters = None
    # dictionary before the latest update (:514) and the validation metrics
    # recorded at the iterations of 'training_visualization_schedule'
    self.previous_dictionary = dictionary.clone()
    self.metrics_log = []

  def infer_codes(self, batch_images):
    """Keyword call into the inference plugin (sparse_coding.py:124-140)."""
    kwargs = {'dictionary': self.dictionary,
              'sparsity_weight': self.sparsity_weight,
              'num_iters': self.inf_num_iters, 'variant': self.inf_name,
              'nonnegative_only': self.nonneg_only,
              'hard_threshold': self.hard_threshold}
    if self.mode == 'fully-connected':
      kwargs['images'] = batch_images
    else:
      kwargs.update({'images_padded': batch_images,
                     'kernel_stride': self.strides,
                     'padding_dims': self.padding})
    if self.inf_name in ('subspace_ista', 'subspace_fista'):
      kwargs['group_assignments'] = self.groups
      kwargs.pop('nonnegative_only')
      kwargs['variant'] = self.inf_name[len('subspace_'):]
    return self.inference_alg.run(**kwargs)

  def _update_hessian_diag(self, codes):
    """h <- 0.99 h + mean_b(sum_pos codes^2) / 100 (sparse_coding.py:154,
    :160-161), with the sum over the global batch when data-parallel."""
    lib = vtc_hip.load_library()
    codes = codes.contiguous()
    b, s = codes.shape[0], codes.shape[1]
    positions = 1
    for extent in codes.shape[2:]:
      positions *= int(extent)
    device = codes.device
    ws = vtc_hip.workspace(
        lib.vtc_code_energy_workspace_bytes(b, s, positions), device)
    energy = torch.empty(s, dtype=torch.float32, device=device)
    stream = vtc_hip.current_stream(device)
    vtc_hip.check(lib.vtc_code_energy(
        vtc_hip.ptr(codes), b, s, positions, vtc_hip.ptr(energy),
        vtc_hip.ptr(ws), ws.numel(), stream), 'vtc_code_energy')
    total = parallel.global_batch(b, device)
    hessian = self.hessian_diag

    def ema():
      vtc_hip.check(lib.vtc_hessian_ema(
          vtc_hip.ptr(hessian), vtc_hip.ptr(energy), total, s, stream),
          'vtc_hessian_ema')
    # data parallel: the s floats ride on the gradient's all-reduce (first
    # update iteration of the plugin) and the EMA runs right behind it
    parallel.defer(energy, ema)

  def update_dictionary(self, batch_images, batch_codes):
    """Keyword call into the update plugin (sparse_coding.py:142-168)."""
    kwargs = {'dictionary': self.dictionary, 'codes': batch_codes,
              'stepsize': self.upd_stepsize, 'num_iters': self.upd_num_iters}
    if self.mode == 'fully-connected':
      kwargs['images'] = batch_images
    else:
      kwargs.update({'images_padded': batch_images,
                     'kernel_stride': self.strides,
                     'padding_dims': self.padding})
    if self.uses_hessian:
      if (self.mode != 'fully-connected' and
          self.upd_name == 'subspace_sc_cheap_quadratic_descent'):
        raise NotImplementedError('TODO for convolutional')
      self._update_hessian_diag(batch_codes)
      kwargs['hessian_diagonal'] = self.hessian_diag
    if self.upd_name in ('subspace_sc_steepest_descent',
                         'subspace_sc_cheap_quadratic_descent'):
      kwargs.update({'group_assignments': self.groups,
                     'alignment_penalty': self.alignment_penalty})
    self.dict_update.run(**kwargs)
    parallel.flush_deferred()

  def __call__(self, batch_images):
    codes = self.infer_codes(batch_images)
    self.update_dictionary(batch_images, codes)
    return codes

  def compute_metrics(self, batch_images, batch_codes):
    """The validation metrics of sparse_coding.py:177-229, same keys.  The
    residual, its per-sample energy, the l1 / l0 / group norms of the codes,
    the signal range and the dictionary change are HIP reductions; only
    b + s floats come back to the host, where the means and the pSNR
    logarithm are taken in numpy as in the reference."""
    lib = vtc_hip.load_library()
    images = vtc_hip.require_device_tensor(
        batch_images, 'batch_images').contiguous()
    codes = vtc_hip.require_device_tensor(
        batch_codes, 'batch_codes').contiguous()
    D = self.dictionary.contiguous()
    device = images.device
    stream = vtc_hip.current_stream(device)
    b = images.shape[0]
    residual = torch.empty_like(images)
    minmax = torch.empty(2, dtype=torch.float32, device=device)
    mm_ws = vtc_hip.workspace(lib.vtc_window_minmax_workspace_bytes(), device)
    if self.mode == 'fully-connected':
      n, s = images.shape[1], D.shape[0]
      vtc_hip.check(lib.vtc_fc_residual(
          vtc_hip.ptr(images), vtc_hip.ptr(D), vtc_hip.ptr(codes),
          vtc_hip.ptr(residual), b, n, s, stream), 'vtc_fc_residual')
      pixels = n
      vtc_hip.check(lib.vtc_window_minmax(
          vtc_hip.ptr(images), 1, b, n, 0, n, vtc_hip.ptr(minmax),
          vtc_hip.ptr(mm_ws), mm_ws.numel(), stream), 'vtc_window_minmax')
    else:
      from utils import convolutions
      geom = convolutions.geometry(images, D, self.strides, self.padding)
      vtc_hip.check(lib.vtc_conv_residual(
          vtc_hip.ptr(images), vtc_hip.ptr(D), vtc_hip.ptr(codes),
          vtc_hip.ptr(residual), ctypes.byref(geom), stream),
          'vtc_conv_residual')
      c, h, w = images.shape[1:]
      (lv, tv), (lh, th) = self.padding if self.padding is not None else (
          (0, 0), (0, 0))
      # the reference crops `lead:-trail` (:188-194); the mask of the residual
      # zeroes the same frame
      ih, iw = h - lv - tv, w - lh - th
      pixels = c * ih * iw
      first = images.reshape(-1)[lv * w + lh:]
      vtc_hip.check(lib.vtc_window_minmax(
          vtc_hip.ptr(first), b * c, ih, iw, h * w, w, vtc_hip.ptr(minmax),
          vtc_hip.ptr(mm_ws), mm_ws.numel(), stream), 'vtc_window_minmax')
    per_sample = residual.reshape(b, -1).shape[1]
    code_len = codes.reshape(b, -1).shape[1]
    sq_err = torch.empty(b, dtype=torch.float32, device=device)
    l1 = torch.empty(b, dtype=torch.float32, device=device)
    l0 = torch.empty(b, dtype=torch.float32, device=device)
    vtc_hip.check(lib.vtc_row_stats(
        vtc_hip.ptr(residual), b, per_sample, vtc_hip.ptr(sq_err),
        vtc_hip.ptr(None), vtc_hip.ptr(None),
        stream), 'vtc_row_stats')
    vtc_hip.check(lib.vtc_row_stats(
        vtc_hip.ptr(codes), b, code_len, vtc_hip.ptr(None), vtc_hip.ptr(l1),
        vtc_hip.ptr(l0), stream), 'vtc_row_stats')
    if self.inf_name in ('subspace_ista', 'subspace_fista'):
      from vtc_hip import groups as group_tables
      tables = group_tables.tables_for(self.groups, D.shape[0], device)
      lagrange_rows = torch.empty(b, dtype=torch.float32, device=device)
      vtc_hip.check(lib.vtc_group_norm_sum(
          vtc_hip.ptr(codes), vtc_hip.ptr(tables.index),
          vtc_hip.ptr(tables.valid), vtc_hip.ptr(lagrange_rows), b,
          D.shape[0], tables.num_groups, tables.m, stream),
          'vtc_group_norm_sum')
    else:
      lagrange_rows = l1
    flatD = D.reshape(D.shape[0], -1)
    change = torch.empty(D.shape[0], dtype=torch.float32, device=device)
    vtc_hip.check(lib.vtc_rows_mean_abs_diff(
        vtc_hip.ptr(flatD), vtc_hip.ptr(self.previous_dictionary.contiguous()),
        flatD.shape[0], flatD.shape[1], vtc_hip.ptr(change), stream),
        'vtc_rows_mean_abs_diff')

    sq_err = sq_err.cpu().numpy()
    lo, hi = minmax.cpu().numpy()
    metrics = {}
    metrics['Average LASSO L2 component'] = np.mean(0.5 * sq_err)
    metrics['Average LASSO lagrange component'] = np.mean(
        np.float32(self.sparsity_weight) * lagrange_rows.cpu().numpy())
    metrics['Average LASSO Loss'] = (
        metrics['Average LASSO L2 component'] +
        metrics['Average LASSO lagrange component'])
    metrics['Average Normalized L0'] = float(
        np.mean(l0.cpu().numpy() / np.float32(code_len)))
    sig_mag = hi - lo
    mse = sq_err / np.float32(pixels)
    nonzero = mse != 0
    metrics['Average pSNR of reconstructions'] = np.mean(
        10. * np.log10((sig_mag ** 2) / mse[nonzero]))
    metrics['Average change in dictionary kernels'] = change.cpu().numpy()
    return metrics


def train_dictionary(training_image_dataset, validation_image_dataset,
                     init_dictionary, all_params):
  """
  Train a sparse coding dictionary; `init_dictionary` is updated IN PLACE.

  training_image_dataset / validation_image_dataset: iterables of batches
  ((b, n) patches or (b, c, h, w) padded images), e.g. torch DataLoaders.
  all_params: the reference's parameter dictionary.  Mandatory keys: 'mode',
  'num_epochs', 'code_inference_algorithm', 'inference_param_schedule',
  'dictionary_update_algorithm', 'dict_update_param_schedule' ('strides' and
  'padding' too when convolutional).  Optional: 'nonnegative_only',
  'hard_threshold', 'group_assignments', 'subspace_alignment_penalty',
  'renormalize_dictionary', 'checkpoint_schedule' + 'logging_folder_fullpath',
  'stdout_print_interval'.  Schedules map the global iteration index at which
  a value takes effect to {'sparsity_weight', 'num_iters'} resp.
  {'stepsize', 'num_iters'}; index 0 must be present.
  """
  assert 0 in all_params['inference_param_schedule']
  assert 0 in all_params['dict_update_param_schedule']
  if 'dict_element_rp_schedule' in all_params:
    raise NotImplementedError(
        'dict_element_rp_schedule is host-side bookkeeping outside the scope '
        'of the MI355X engine (SURVEY.md section 8f)')
  vis_schedule = all_params.get('training_visualization_schedule')
  inf_schedule = all_params['inference_param_schedule']
  upd_schedule = all_params['dict_update_param_schedule']
  vtc_hip.require_device_tensor(init_dictionary, 'init_dictionary')
  if all_params.get('renormalize_dictionary', True):
    flat = init_dictionary.reshape(init_dictionary.shape[0], -1)
    assert torch.allclose(
        flat.norm(p=2, dim=1),
        torch.ones(flat.shape[0], device=flat.device)), (
            'Please ensure the initial dictionary is already normalized')
  ckpt_schedule = all_params.get('checkpoint_schedule')
  logging_path = all_params.get('logging_folder_fullpath')
  if ckpt_schedule is not None:
    assert logging_path is not None and type(logging_path) != str, (
        'should be pathlib.Path')
    logging_path.mkdir(parents=True, exist_ok=True)
  print_interval = all_params.get('stdout_print_interval', 1000)

  step = TrainingStep(init_dictionary, all_params)
  previous_dictionary = step.previous_dictionary

  start = time.time()
  total_iter_idx = 0
  for epoch_idx in range(all_params['num_epochs']):
    for batch_images in training_image_dataset:
      if total_iter_idx % print_interval == 0 and total_iter_idx != 0:
        print(total_iter_idx, 'iterations complete')
        print('Time elapsed:', '{:.1f}'.format(time.time() - start), 'seconds')
        print('-----')
      if total_iter_idx in inf_schedule:
        step.sparsity_weight = inf_schedule[total_iter_idx]['sparsity_weight']
        step.inf_num_iters = inf_schedule[total_iter_idx]['num_iters']
      if total_iter_idx in upd_schedule:
        step.upd_stepsize = upd_schedule[total_iter_idx]['stepsize']
        step.upd_num_iters = upd_schedule[total_iter_idx]['num_iters']
      if (ckpt_schedule is not None and total_iter_idx in ckpt_schedule and
          parallel.rank() == 0):
        # plain pickle of the numpy array, the reference's on-disk format
        with open(logging_path / ('checkpoint_dictionary_iter_' +
                                  str(total_iter_idx)), 'wb') as f:
          pickle.dump(init_dictionary.cpu().numpy(), f)
      if vis_schedule is not None and total_iter_idx in vis_schedule:
        # validation pass (:497-505): mean of every metric over the batches of
        # the validation set; kept in step.metrics_log (iteration, dict)
        # instead of a TensorBoard writer, dictionary images are not drawn
        val_metrics = []
        for v_batch_images in validation_image_dataset:
          if init_dictionary.device != v_batch_images.device:
            v_batch_images = v_batch_images.to(init_dictionary.device)
          v_codes = step.infer_codes(v_batch_images)
          val_metrics.append(step.compute_metrics(v_batch_images, v_codes))
        step.metrics_log.append((total_iter_idx, {
            x: np.mean([val_metrics[y][x] for y in range(len(val_metrics))])
            for x in val_metrics[0]}))
      if init_dictionary.device != batch_images.device:
        batch_images = batch_images.to(init_dictionary.device)
      previous_dictionary.copy_(init_dictionary)
      step(batch_images)
      total_iter_idx += 1
    print("Epoch", epoch_idx + 1, "finished")
  # the sync-free inference path reports a failed eigen-solve late: collect
  # whatever is still pending before handing the dictionary back
  vtc_hip.poll_spectrum_checks(block=True)
  return step
