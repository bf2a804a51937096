"""
Dictionary training loop for sparse coding on MI355X: the per-batch step.

Mirrors the calling convention of vision_transform_codes/training/
sparse_coding.py (train_dictionary at :9-10, parameter dictionary at :52-117)
for the part of it that is on the hot path: schedule lookup, code inference,
Hessian-diagonal EMA, dictionary update (:124-168, :444-517), plus dictionary
checkpoints (:170-175), the parameter dump (:364-379), the validation metrics
(:177-229, :497-505) as device reductions and the non-interactive modes of the
dictionary reset / prune schedule (:466-492, :522-764).  TensorBoard image
summaries are skipped (the metrics are kept in TrainingStep.metrics_log
instead of a SummaryWriter); asking the reset/prune filter to cue the user for
a threshold raises NotImplementedError.

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

  def replace_dictionary(self, dictionary, affected, action):
    """After reset_or_prune_dict_elements: a pruned dictionary is a new,
    smaller tensor -- the copy of the previous dictionary and the Hessian
    diagonal shrink with it (sparse_coding.py:483-491)."""
    if action == 'prune' and len(affected) > 0:
      self.dictionary = dictionary
      self.previous_dictionary = dictionary.clone()
      if self.uses_hessian:
        keep = torch.ones(self.hessian_diag.shape[0], dtype=torch.bool)
        keep[torch.as_tensor(np.asarray(affected, dtype=np.int64))] = False
        self.hessian_diag = self.hessian_diag[
            keep.to(self.hessian_diag.device)].contiguous()

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
    energy = parallel.take((s,), device)
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
    try:
      self.dict_update.run(**kwargs)
      parallel.flush_deferred()
    except BaseException:
      # a plugin that failed before its all-reduce must not leave the code
      # energy queued on this rank only (the next collective would then carry
      # a different element count than on the other ranks)
      parallel.drop_deferred()
      raise

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


def _save_training_params(all_params, logging_path):
  """training_params.yaml (every parameter but the schedules of when to write
  files, group assignments as plain lists) and, when the caller passed its own
  source as 'str_entire_calling_script', called_script.py -- the files the
  reference leaves next to its logs (sparse_coding.py:364-379)."""
  import yaml
  saved = {k: all_params[k] for k in all_params if k not in (
      'checkpoint_schedule', 'training_visualization_schedule',
      'group_assignments')}
  groups = all_params.get('group_assignments')
  if groups is not None and type(groups[0]) != list:
    groups = [x.tolist() for x in groups]
  saved['group_assignments'] = groups
  with open(logging_path / 'training_params.yaml', 'w') as f:
    yaml.dump(saved, f, default_flow_style=None)
  if 'str_entire_calling_script' in all_params:
    with open(logging_path / 'called_script.py', 'w') as f:
      f.write(all_params['str_entire_calling_script'])


def load_newest_dictionary_checkpoint(checkpoint_dir):
  """The dictionary of the highest checkpoint iteration in `checkpoint_dir`
  (the convention of the reference's utils/misc.py:8-20: files named
  checkpoint_dictionary_iter_<i>, a pickled numpy array each)."""
  prefix = 'checkpoint_dictionary_iter_'
  iters = [int(p.name[len(prefix):]) for p in checkpoint_dir.iterdir()
           if p.is_file() and p.name.startswith(prefix) and
           p.name[len(prefix):].isdigit()]
  print('checkpoint idx: ', max(iters))
  with open(checkpoint_dir / (prefix + str(max(iters))), 'rb') as f:
    return _ArrayUnpickler(f).load()


class _ArrayUnpickler(pickle.Unpickler):
  """Reads a pickled numpy array and nothing else: a checkpoint directory may
  come from anywhere, and a plain pickle.load would run whatever callable the
  file names."""
  _ALLOWED = {('numpy.core.multiarray', '_reconstruct'),
              ('numpy._core.multiarray', '_reconstruct'),
              ('numpy', 'ndarray'), ('numpy', 'dtype')}

  def find_class(self, module, name):
    if (module, name) not in self._ALLOWED:
      raise pickle.UnpicklingError(
          'checkpoint refers to %s.%s: only plain numpy arrays are read' % (
              module, name))
    return super().find_class(module, name)


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
  rp_schedule = all_params.get('dict_element_rp_schedule')
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
  if ckpt_schedule is not None or (vis_schedule is not None and
                                   logging_path is not None):
    assert logging_path is not None and type(logging_path) != str, (
        'should be pathlib.Path')
    logging_path.mkdir(parents=True, exist_ok=True)
    if parallel.rank() == 0:
      _save_training_params(all_params, logging_path)
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
      if rp_schedule is not None and total_iter_idx in rp_schedule:
        # reset or prune dictionary elements (sparse_coding.py:466-492); the
        # code distribution some filters need comes from the validation set
        entry = rp_schedule[total_iter_idx]
        f_params = entry['filter_params']
        f_params.update({'group_assignments': step.groups,
                         'coding_mode': step.mode})
        v_codes = torch.cat([
            step.infer_codes(v.to(init_dictionary.device))
            for v in validation_image_dataset])
        # data parallel: the selection and the replacement atoms come from
        # each process's own numpy / torch CPU generators, so rank 0 decides
        # and every rank adopts its dictionary (and its pruned group lists)
        new_dictionary, affected = step.dictionary, []
        if parallel.rank() == 0:
          new_dictionary, affected = reset_or_prune_dict_elements(
              step.dictionary, v_codes, entry['filter_type'], f_params,
              entry['action'])
        new_dictionary, affected, _ = parallel.broadcast_reset_or_prune(
            new_dictionary, affected, step.groups)
        step.replace_dictionary(new_dictionary, affected,
                                action=entry['action'])
        init_dictionary = step.dictionary
        previous_dictionary = step.previous_dictionary
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


# ---------------------------------------------------------------------------
# dictionary reset / prune (sparse_coding.py:466-492, :522-764 of the reference)
# ---------------------------------------------------------------------------
def _row_gram(dictionary):
  """D D^T of the (s, n) dictionary on the device (vtc_gram, exact-f32 MFMA)
  as a host array; its diagonal holds the squared row norms."""
  flat = dictionary.reshape(dictionary.shape[0], -1).contiguous()
  return vtc_hip.gram(flat, transpose_a=False).cpu().numpy()


def _cosine_similarities(gram, rows=None):
  if rows is not None:
    gram = gram[np.ix_(rows, rows)]
  norms = np.sqrt(np.diag(gram)).astype(np.float32)
  return gram / (norms[:, None] * norms[None, :])


def _one_of_each_problem_pair(pairs):
  """Walk the offending (i, j) pairs in row-major order and flag one member
  (numpy's global generator picks which) of every pair not yet touched."""
  chosen = []
  for pair in pairs:
    if pair[0] not in chosen and pair[1] not in chosen:
      chosen.append(pair[np.random.choice([0, 1])])
  return chosen


def _noise_rows(count, width, average_norm, device):
  """Fresh random atoms with the given norm.  Drawn from torch's CPU generator
  (so that a seeded run is reproducible on any device), then moved."""
  noise = torch.randn((count, width))
  noise.mul_(float(average_norm) / noise.norm(p=2, dim=1)[:, None])
  return noise.to(device)


def _write_rows(dictionary, rows, values):
  """dictionary[rows] = values with the CPU's semantics for repeated indices
  (numpy's choice() draws with replacement): the last occurrence wins.  An
  indexed store with duplicates has no defined winner on the device."""
  rows = np.asarray(rows, dtype=np.int64).ravel()
  last = {}
  for position, row in enumerate(rows.tolist()):
    last[row] = position
  unique = np.array(sorted(last), dtype=np.int64)
  source = np.array([last[row] for row in unique.tolist()], dtype=np.int64)
  dictionary[torch.as_tensor(unique).to(dictionary.device)] = values[
      torch.as_tensor(source).to(values.device)]


def _drop_rows(dictionary, groups, rows):
  keep = torch.ones(dictionary.shape[0], dtype=torch.bool)
  keep[torch.as_tensor(np.asarray(rows, dtype=np.int64))] = False
  if groups is not None:
    dropped = set(int(x) for x in np.asarray(rows).ravel())
    for g_idx in range(len(groups)):
      groups[g_idx] = [a for a in groups[g_idx] if a not in dropped]
  return dictionary[keep.to(dictionary.device)]


def reset_or_prune_dict_elements(dictionary, codes, filter_type,
                                 filter_params, action):
  """
  Reset (to random atoms of average norm) or prune dictionary elements during
  training -- the non-interactive modes of the reference's function of the
  same name (training/sparse_coding.py:522-764).

  filter_type: 'random' (filter_params['num_to_modify'] atoms drawn with
  numpy's global generator), 'cosine_sim_threshold' (one atom of every pair
  whose cosine similarity exceeds filter_params['threshold']; within groups
  only -- and on |cos| -- when filter_params['only_sim_within_group']), or
  'nonuniformity_within_group' (groups whose code phases fill the sphere
  unevenly, estimated from `codes` over filter_params['num_gc_in_average']
  random great circles).  filter_params also carries 'coding_mode' and
  'group_assignments' (pruning edits the group lists in place, as the
  reference does).  action: 'reset' or 'prune'.

  Returns (dictionary, affected_atoms): the same tensor, modified in place,
  for 'reset'; a new, smaller tensor for 'prune'.  The similarity matrix is a
  device contraction (vtc_gram); the selection logic runs on the host like the
  reference's (it goes through .cpu().numpy() there as well).
  """
  groups = filter_params['group_assignments']
  coding_mode = filter_params['coding_mode']
  if coding_mode == 'convolutional':
    raise NotImplementedError('Not yet implemented for convolutional dict')
  if coding_mode != 'fully-connected':
    raise KeyError('Unrecognized coding mode')
  assert action in ('reset', 'prune')
  width, device = dictionary.shape[1], dictionary.device

  def average_norm(gram, rows=None):
    diag = np.diag(gram) if rows is None else np.diag(gram)[rows]
    return np.mean(np.sqrt(diag).astype(np.float32))

  if filter_type == 'random':
    modify_these = np.random.choice(np.arange(dictionary.shape[0]),
                                    filter_params['num_to_modify'])
    if action == 'reset':
      gram = _row_gram(dictionary)
      _write_rows(dictionary, modify_these, _noise_rows(
          len(modify_these), width, average_norm(gram), device))
    else:
      dictionary = _drop_rows(dictionary, groups, modify_these)
    return dictionary, modify_these

  if filter_type == 'cosine_sim_threshold':
    if filter_params['cue_user']:
      raise NotImplementedError(
          'interactive threshold selection (a matplotlib window and input()) '
          'is not part of this engine: pass filter_params["threshold"]')
    threshold = filter_params['threshold']
    gram = _row_gram(dictionary)
    if filter_params['only_sim_within_group']:
      assert groups is not None
      flagged = []
      for g_idx in range(len(groups)):
        members = np.array(groups[g_idx])
        sims = _cosine_similarities(gram, members)
        pairs = np.argwhere(np.abs(np.triu(sims, k=1)) > threshold)
        local = _one_of_each_problem_pair(pairs)
        if len(local) > 0:
          print('Action ', action, 'applied to ', local, 'in group', g_idx)
          if action == 'reset':
            _write_rows(dictionary, members[local], _noise_rows(
                len(local), width, average_norm(gram, members), device))
            # the reference takes each group's similarities and norms from the
            # dictionary as it stands (:632-655): with overlapping groups a
            # later group sees the rows just replaced
            gram = _row_gram(dictionary)
          flagged.append(members[local])
      modify_these = np.array(flagged).flatten()
      if action == 'prune' and len(modify_these) > 0:
        dictionary = _drop_rows(dictionary, groups, modify_these)
      return dictionary, modify_these
    sims = _cosine_similarities(gram)
    pairs = np.argwhere(np.triu(sims, k=1) > threshold)
    modify_these = np.array(_one_of_each_problem_pair(pairs))
    if len(modify_these) > 0:
      if action == 'reset':
        _write_rows(dictionary, modify_these, _noise_rows(
            len(modify_these), width, average_norm(gram), device))
      else:
        dictionary = _drop_rows(dictionary, groups, modify_these)
    return dictionary, modify_these

  if filter_type == 'nonuniformity_within_group':
    num_great_circles = filter_params['num_gc_in_average']
    host_codes = codes.cpu().numpy()
    spread = []
    for g_idx in range(len(groups)):
      members = np.array(groups[g_idx])
      block = host_codes[:, members]
      block = block[np.sum(block != 0, axis=1) != 0]
      renormed = block / np.linalg.norm(block, axis=1, keepdims=True)
      variances = []
      for _ in range(num_great_circles):
        first = np.random.randn(len(members))
        first /= np.linalg.norm(first)
        second = np.random.randn(len(members))
        second /= np.linalg.norm(second)
        plane, _ = np.linalg.qr(np.c_[first, second])
        proj = np.dot(renormed, plane)
        angle = np.angle(proj[:, 0] + 1j * proj[:, 1])
        counts, _ = np.histogram(angle, np.linspace(-np.pi, np.pi, 21))
        variances.append(np.var(counts / np.sum(counts)))
      spread.append(np.mean(variances))
    spread = np.array(spread)
    outliers = np.nonzero(np.logical_and(
        np.abs(spread - np.mean(spread)) > np.std(spread),
        np.abs(spread) > 0.002))[0]
    modify_these = np.array([groups[x] for x in outliers]).flatten()
    if len(modify_these) > 0:
      if action == 'reset':
        gram = _row_gram(dictionary)
        _write_rows(dictionary, modify_these, _noise_rows(
            len(modify_these), width, average_norm(gram), device))
      else:
        dictionary = _drop_rows(dictionary, groups, modify_these)
    return dictionary, modify_these

  raise KeyError('Unrecognized reset type')
