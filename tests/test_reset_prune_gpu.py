"""f4 / f2: dictionary reset / prune during training, the parameter dump and
the checkpoint files, against fixtures recorded from the reference's own
reset_or_prune_dict_elements (tests/golden/reset_prune.npz)."""
import os
import pathlib
import pickle

import numpy as np
import pytest
import torch
import yaml

import helpers
import make_golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('case', make_golden.reset_prune_cases(),
                         ids=lambda c: c[0])
def test_modes_against_the_reference(device, case):
  from training import sparse_coding
  tag, f_type, f_params, action, np_seed, torch_seed = case
  g = helpers.load('reset_prune')
  D0, groups0, C = make_golden.reset_prune_inputs()
  groups = [list(x) for x in groups0]
  params = dict(f_params)
  params.update({'group_assignments': groups, 'coding_mode': 'fully-connected'})
  np.random.seed(np_seed)
  torch.manual_seed(torch_seed)
  D = helpers.to_dev(D0.copy(), device)
  Dn, rows = sparse_coding.reset_or_prune_dict_elements(
      D, helpers.to_dev(C, device), f_type, params, action)
  assert np.array_equal(np.asarray(rows), g[tag + '_affected'])
  assert [len(x) for x in groups] == g[tag + '_group_sizes'].tolist()
  assert [a for x in groups for a in x] == g[tag + '_groups_flat'].tolist()
  want = g[tag + '_dictionary']
  assert tuple(Dn.shape) == want.shape
  # untouched / surviving atoms bit for bit; fresh atoms: same CPU draw,
  # scaled by an average norm taken from the device Gram (1e-6)
  assert helpers.rel_err(Dn.cpu().numpy(), want) < 1e-6
  if action == 'reset':
    assert Dn is D
    untouched = np.setdiff1d(np.arange(32), np.asarray(rows))
    assert np.array_equal(Dn.cpu().numpy()[untouched], want[untouched])
  else:
    assert np.array_equal(Dn.cpu().numpy(), want)


def test_interactive_and_convolutional_requests_raise(device):
  from training import sparse_coding
  D = helpers.to_dev(helpers.unit_rows(1, 8, 16), device)
  with pytest.raises(NotImplementedError):
    sparse_coding.reset_or_prune_dict_elements(
        D, None, 'cosine_sim_threshold',
        {'cue_user': True, 'only_sim_within_group': False,
         'group_assignments': None, 'coding_mode': 'fully-connected'}, 'reset')
  with pytest.raises(NotImplementedError):
    sparse_coding.reset_or_prune_dict_elements(
        D, None, 'random', {'num_to_modify': 1, 'group_assignments': None,
                            'coding_mode': 'convolutional'}, 'reset')
  with pytest.raises(KeyError):
    sparse_coding.reset_or_prune_dict_elements(
        D, None, 'by_magic', {'group_assignments': None,
                              'coding_mode': 'fully-connected'}, 'reset')


def test_training_with_a_prune_schedule_and_the_files_it_leaves(device,
                                                               tmp_path):
  """dict_element_rp_schedule inside train_dictionary (prune at step 2: the
  dictionary, its previous copy and the Hessian diagonal shrink and training
  goes on), training_params.yaml + called_script.py, and checkpoints that the
  reference's newest-checkpoint convention (utils/misc.py:8-20) finds."""
  from training import sparse_coding
  X = helpers.to_dev(helpers.gaussian_patches(90, 160, 64), device)
  D = helpers.to_dev(helpers.unit_rows(91, 48, 64), device)
  logdir = pathlib.Path(tmp_path) / 'run'
  params = {
      'mode': 'fully-connected', 'num_epochs': 1,
      'code_inference_algorithm': 'fista',
      'inference_param_schedule': {
          0: {'sparsity_weight': 0.02, 'num_iters': 10}},
      'dictionary_update_algorithm': 'sc_cheap_quadratic_descent',
      'dict_update_param_schedule': {0: {'stepsize': 0.1, 'num_iters': 1}},
      'dict_element_rp_schedule': {
          2: {'filter_type': 'random', 'filter_params': {'num_to_modify': 6},
              'action': 'prune'},
          3: {'filter_type': 'random', 'filter_params': {'num_to_modify': 3},
              'action': 'reset'}},
      'checkpoint_schedule': {0, 2, 4},
      'logging_folder_fullpath': logdir,
      'str_entire_calling_script': '# the calling script\nprint(1)\n'}
  batches = [X[32 * i: 32 * i + 32] for i in range(5)]
  np.random.seed(0)
  torch.manual_seed(0)
  state = sparse_coding.train_dictionary(batches, batches[:1], D, params)
  pruned = len(set(np.random.RandomState(0).choice(np.arange(48), 6).tolist()))
  assert state.dictionary.shape[0] == 48 - pruned
  assert state.hessian_diag.shape[0] == 48 - pruned
  assert state.previous_dictionary.shape == state.dictionary.shape
  norms = state.dictionary.norm(dim=1).cpu().numpy()
  assert np.allclose(norms, 1.0, atol=1e-5)       # updated after the reset too
  # parameter dump: everything but the file schedules
  saved = yaml.load(open(logdir / 'training_params.yaml'), Loader=yaml.Loader)
  assert saved['code_inference_algorithm'] == 'fista'
  assert 'checkpoint_schedule' not in saved
  assert saved['group_assignments'] is None
  assert (logdir / 'called_script.py').read_text().startswith('# the calling')
  # the reference's loader convention: highest iteration among files named
  # checkpoint_dictionary_iter_<i>, each a pickled numpy array
  iters = []
  for _, _, names in os.walk(logdir):
    iters = [int(n[27:]) for n in names
             if n[:27] == 'checkpoint_dictionary_iter_']
    break
  assert sorted(iters) == [0, 2, 4]
  newest = pickle.load(open(logdir / ('checkpoint_dictionary_iter_%d' % max(iters)), 'rb'))
  assert isinstance(newest, np.ndarray) and newest.dtype == np.float32
  assert newest.shape == (48 - pruned, 64)
  again = sparse_coding.load_newest_dictionary_checkpoint(logdir)
  assert np.array_equal(again, newest)
  first = pickle.load(open(logdir / 'checkpoint_dictionary_iter_0', 'rb'))
  assert first.shape == (48, 64)
