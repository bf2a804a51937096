"""Host-side logic that needs no GPU: option mapping, group index tables,
convolution geometry, the data-parallel helpers (gloo, world_size 2)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import helpers
import sc_oracle


def test_threshold_and_variant_codes():
  import vtc_hip
  assert vtc_hip.threshold_mode(False, False) == vtc_hip.SOFT
  assert vtc_hip.threshold_mode(True, False) == vtc_hip.SOFT_NONNEG
  assert vtc_hip.threshold_mode(False, True) == vtc_hip.HARD
  assert vtc_hip.threshold_mode(True, True) == vtc_hip.HARD_NONNEG
  assert vtc_hip.variant_code('ista') == vtc_hip.ISTA
  assert vtc_hip.variant_code('fista') == vtc_hip.FISTA
  with pytest.raises(AssertionError):
    vtc_hip.variant_code('lista')


def test_group_tables_match_oracle_layout():
  from vtc_hip import groups as group_tables
  groups = [[0, 2, 5], [1], [2, 3, 4, 5]]
  t = group_tables.GroupTables(groups, 6, torch.device('cpu'))
  gather_index, valid, num_groups, m = sc_oracle.group_layout(groups, 6)
  assert (t.num_groups, t.m, t.slots) == (num_groups, m, num_groups * m)
  assert np.array_equal(t.valid.numpy().astype(bool), valid.numpy())
  assert np.array_equal(t.index.numpy()[valid.numpy()],
                        gather_index.numpy()[valid.numpy()])
  # CSR inverse: every valid slot appears once, under its atom, ascending
  ptr, slots = t.atom_ptr.numpy(), t.atom_slots.numpy()
  seen = []
  for atom in range(6):
    mine = slots[ptr[atom]: ptr[atom + 1]]
    assert list(mine) == sorted(mine)
    for slot in mine:
      assert t.index.numpy()[slot] == atom and t.valid.numpy()[slot]
      seen.append(slot)
  assert sorted(seen) == list(np.nonzero(valid.numpy())[0])
  with pytest.raises(IndexError):
    group_tables.GroupTables([[0, 9]], 6, torch.device('cpu'))


def test_conv_geometry_helpers():
  from utils import convolutions
  for img, k, st in ((256, 11, 1), (30, 8, 4), (32, 16, 8), (16, 8, 4)):
    assert convolutions.get_padding_amt(img, k, st) == (
        sc_oracle.conv_padding_amount(img, k, st))
    lead, trail = convolutions.get_padding_amt(img, k, st)
    padded = img + lead + trail
    assert convolutions.code_dim_from_padded_img_dim(padded, k, st) == (
        sc_oracle.conv_code_dim(padded, k, st))
  imgs = torch.zeros(2, 1, 20, 24)
  pad = ((3, 4), (2, 5))
  assert torch.equal(convolutions.create_mask(imgs, pad),
                     sc_oracle.conv_mask(imgs, pad))
  assert torch.equal(convolutions.create_mask(imgs, None),
                     torch.ones_like(imgs))
  g = convolutions.geometry(imgs, torch.zeros(5, 1, 4, 4), (2, 2), pad)
  assert (g.b, g.c, g.h, g.w, g.s, g.kh, g.kw) == (2, 1, 20, 24, 5, 4, 4)
  assert (g.pad_lead_v, g.pad_trail_v, g.pad_lead_h, g.pad_trail_h) == (
      3, 4, 2, 5)


def test_trainer_rejects_out_of_scope_features_and_bad_names():
  import vtc_hip
  from training import sparse_coding
  D = torch.eye(4)
  base = {'mode': 'fully-connected', 'num_epochs': 1,
          'code_inference_algorithm': 'fista',
          'inference_param_schedule': {0: {'sparsity_weight': 0.1,
                                           'num_iters': 2}},
          'dictionary_update_algorithm': 'sc_steepest_descent',
          'dict_update_param_schedule': {0: {'stepsize': 0.1,
                                             'num_iters': 1}}}
  with pytest.raises(AssertionError):
    sparse_coding.train_dictionary(
        [], [], D, dict(base, inference_param_schedule={1: {}}))
  with pytest.raises(vtc_hip.VtcHipError):   # CPU dictionary: no fallback
    sparse_coding.train_dictionary([], [], D, base)
  with pytest.raises(KeyError):
    sparse_coding._load_plugins('convolutional', 'subspace_fista',
                                'sc_steepest_descent')
  with pytest.raises(ImportError):
    # the reference's tree lacks this module too (sparse_coding.py:423-424)
    sparse_coding._load_plugins('fully-connected', 'fista',
                                'subspace_sc_steepest_descent')


# ---------------------------------------------------------------- 2 ranks
def _free_port():
  with socket.socket() as s:
    s.bind(('127.0.0.1', 0))
    return s.getsockname()[1]


def _dp_worker(rank, world, port, out_dir):
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  import sys
  sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..',
                                  'vision-transform-codes_amd'))
  sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'oracle'))
  from vtc_hip import parallel
  import sc_oracle as oracle
  torch.set_num_threads(2)
  parallel.enable()
  assert parallel.world_size() == world and parallel.rank() == rank
  X = torch.from_numpy(helpers.gaussian_patches(7, 64, 32))
  D = torch.from_numpy(helpers.unit_rows(8, 48, 32))
  C = oracle.fc_ista_fista(X, D, 0.05, 10)
  shard = slice(rank * 32, rank * 32 + 32)
  Xs, Cs = X[shard], C[shard]
  # local un-normalised partials, exactly what the HIP gradient entry returns
  grad_sum = torch.mm(Cs.t(), torch.mm(Cs, D) - Xs)
  energy = (Cs * Cs).sum(0)
  # the trainer's protocol: the code energy is deferred, rides on the
  # gradient's all-reduce, and its callback (the Hessian EMA) runs after the
  # sum and before the caller applies the gradient
  order = []
  seen_in_callback = {}

  def ema_stand_in():
    order.append('ema')
    seen_in_callback['energy'] = energy.clone()
  before = parallel.collectives_issued
  parallel.defer(energy, ema_stand_in)
  assert order == []                              # not yet: waits for the reduce
  parallel.all_reduce_sum_(grad_sum)              # packed: ONE collective
  order.append('apply')
  assert order == ['ema', 'apply']
  assert parallel.collectives_issued - before == 1
  assert torch.equal(seen_in_callback['energy'], energy)
  parallel.flush_deferred()                       # nothing left: no collective
  assert parallel.collectives_issued - before == 1
  total = parallel.global_batch(Xs.shape[0], X.device)
  assert total == 64
  parallel.enable(equal_shards=False)
  assert parallel.global_batch(Xs.shape[0] - rank, X.device) == 63
  Dn = D.clone()
  Dn.sub_(0.1 * (grad_sum / total))
  Dn.div_(Dn.norm(p=2, dim=1)[:, None])
  torch.save({'dict': Dn, 'energy': energy},
             os.path.join(out_dir, 'rank%d.pt' % rank))
  dist.barrier()
  dist.destroy_process_group()


def test_deferred_reduce_without_a_process_group_runs_at_once():
  from vtc_hip import parallel
  parallel.disable()
  ran = []
  parallel.defer(torch.ones(3), lambda: ran.append(1))
  assert ran == [1]
  parallel.flush_deferred()


def test_bench_refuses_a_mismatched_launcher_and_spawns_without_one():
  """bench.py --gpus N: under a launcher WORLD_SIZE must equal N; without one
  it starts its own N rank processes before any GPU call (checked here only
  up to the environment each child would get)."""
  import importlib.util
  import subprocess
  import sys
  repo = os.path.join(os.path.dirname(__file__), '..')
  env = dict(os.environ, WORLD_SIZE='2', RANK='0', LOCAL_RANK='0')
  proc = subprocess.run([sys.executable, os.path.join(repo, 'bench.py'),
                         '--gpus', '4'], env=env, capture_output=True,
                        text=True)
  assert proc.returncode != 0 and 'WORLD_SIZE=2' in proc.stderr
  spec = importlib.util.spec_from_file_location(
      'bench_module', os.path.join(repo, 'bench.py'))
  bench = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(bench)
  launched = []

  class FakeChild(object):
    def __init__(self, argv, env):
      launched.append(env)

    def wait(self):
      return 0
  real = subprocess.Popen
  subprocess.Popen = FakeChild
  try:
    args = type('A', (), {'gpus': 4})()
    assert bench.spawn_ranks(args) == 0
  finally:
    subprocess.Popen = real
  assert [e['RANK'] for e in launched] == ['0', '1', '2', '3']
  assert all(e['WORLD_SIZE'] == '4' and e['MASTER_ADDR'] == '127.0.0.1'
             for e in launched)
  assert len(set(e['MASTER_PORT'] for e in launched)) == 1


def test_data_parallel_update_matches_single_process(tmp_path):
  """Shard a batch over 2 gloo ranks: summed shard gradients / global batch
  reproduce the full-batch update, and both ranks end bit-identical."""
  world = 2
  mp.spawn(_dp_worker, args=(world, _free_port(), str(tmp_path)),
           nprocs=world, join=True)
  r0 = torch.load(tmp_path / 'rank0.pt', weights_only=True)
  r1 = torch.load(tmp_path / 'rank1.pt', weights_only=True)
  assert torch.equal(r0['dict'], r1['dict'])
  assert torch.equal(r0['energy'], r1['energy'])
  X = torch.from_numpy(helpers.gaussian_patches(7, 64, 32))
  D = torch.from_numpy(helpers.unit_rows(8, 48, 32))
  C = sc_oracle.fc_ista_fista(X, D, 0.05, 10)
  full = D.clone()
  sc_oracle.fc_steepest_descent(X, full, C, stepsize=0.1)
  assert helpers.rel_err(r0['dict'].numpy(), full.numpy()) < 1e-6
  assert helpers.rel_err(r0['energy'].numpy(), (C * C).sum(0).numpy()) < 1e-6


def _rp_worker(rank, world, port, out_dir):
  """dict_element_rp_schedule under data parallelism: every rank makes its OWN
  random choice (different generators), rank 0's wins everywhere; then one
  update step through the persistent flat buffer (energy + gradient slices,
  summed in place by ONE collective without packing)."""
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  import sys
  sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..',
                                  'vision-transform-codes_amd'))
  from vtc_hip import parallel
  torch.set_num_threads(2)
  parallel.enable()
  np.random.seed(100 + rank)                 # the ranks' generators differ
  torch.manual_seed(200 + rank)
  D = torch.from_numpy(helpers.unit_rows(8, 12, 16))
  groups = [[0, 1, 2], [3, 4, 5], [6, 7, 8], [9, 10, 11]]
  hessian = torch.arange(12, dtype=torch.float32)
  results = {}
  for action in ('reset', 'prune'):
    mine = D.clone()
    my_groups = [list(g) for g in groups]
    affected = np.random.choice(np.arange(12), 3, replace=False)
    new = mine
    if rank == 0:                            # what rank 0's filter returns
      if action == 'reset':
        mine[torch.as_tensor(affected)] = torch.randn(3, 16)
      else:
        keep = torch.ones(12, dtype=torch.bool)
        keep[torch.as_tensor(affected)] = False
        new = mine[keep]
        dropped = set(int(a) for a in affected)
        for g in range(len(my_groups)):
          my_groups[g] = [a for a in my_groups[g] if a not in dropped]
    else:
      affected = []                          # a non-zero rank decides nothing
    new, affected, _ = parallel.broadcast_reset_or_prune(new, affected,
                                                         my_groups)
    if action == 'reset':
      assert new is mine                     # in place: callers hold aliases
    keep = torch.ones(12, dtype=torch.bool)
    if action == 'prune':
      keep[torch.as_tensor(np.asarray(affected, dtype=np.int64))] = False
    results[action] = {'dict': new.clone(), 'affected': [int(a) for a in affected],
                       'groups': my_groups, 'hessian': hessian[keep].clone()}
  # one update step through the flat buffer
  before_c, before_p = parallel.collectives_issued, parallel.packed_copies
  for step in range(2):                      # the second step re-uses the buffer
    energy = parallel.take((12,), 'cpu')
    energy.copy_(torch.full((12,), float(rank + 1)))
    ran = []
    parallel.defer(energy, lambda: ran.append(energy.clone()))
    grad = parallel.take((12, 16), 'cpu')
    grad.copy_(torch.full((12, 16), 10.0 * (rank + 1)))
    parallel.all_reduce_sum_(grad)
    assert torch.equal(ran[0], torch.full((12,), 3.0))
    assert torch.equal(grad, torch.full((12, 16), 30.0))
  assert parallel.collectives_issued - before_c == 2
  # the first step may pack (the buffer is created by its first take; both
  # slices come from the same buffer here), the second must not
  assert parallel.packed_copies - before_p == 0
  # a plugin that dies between defer() and its reduce leaves nothing behind
  parallel.defer(parallel.take((12,), 'cpu'), lambda: None)
  parallel.drop_deferred()
  parallel.flush_deferred()
  assert parallel.collectives_issued - before_c == 2
  torch.save(results, os.path.join(out_dir, 'rp_rank%d.pt' % rank))
  dist.barrier()
  dist.destroy_process_group()


def test_reset_and_prune_are_rank0s_decision_everywhere(tmp_path):
  world = 2
  mp.spawn(_rp_worker, args=(world, _free_port(), str(tmp_path)),
           nprocs=world, join=True)
  r0 = torch.load(tmp_path / 'rp_rank0.pt', weights_only=True)
  r1 = torch.load(tmp_path / 'rp_rank1.pt', weights_only=True)
  for action in ('reset', 'prune'):
    assert torch.equal(r0[action]['dict'], r1[action]['dict'])
    assert r0[action]['affected'] == r1[action]['affected']
    assert r0[action]['groups'] == r1[action]['groups']
    assert r0[action]['hessian'].shape == r1[action]['hessian'].shape
  assert r0['prune']['dict'].shape[0] == 9 == r0['prune']['hessian'].shape[0]
  assert sum(len(g) for g in r0['prune']['groups']) == 9


def test_checkpoint_loader_reads_arrays_only(tmp_path):
  """load_newest_dictionary_checkpoint: newest iteration by number, stray
  suffixes ignored, and a pickle that names anything but a numpy array is
  refused instead of executed."""
  import pickle
  from training import sparse_coding as trainer
  arr = np.arange(12, dtype=np.float32).reshape(3, 4)
  for it in (5, 40):
    with open(tmp_path / ('checkpoint_dictionary_iter_%d' % it), 'wb') as f:
      pickle.dump(arr + it, f)
  (tmp_path / 'checkpoint_dictionary_iter_90.bak').write_bytes(b'junk')
  got = trainer.load_newest_dictionary_checkpoint(tmp_path)
  assert np.array_equal(got, arr + 40)

  class Evil(object):
    def __reduce__(self):
      return (os.system, ('true',))
  with open(tmp_path / 'checkpoint_dictionary_iter_99', 'wb') as f:
    pickle.dump(Evil(), f)
  with pytest.raises(pickle.UnpicklingError):
    trainer.load_newest_dictionary_checkpoint(tmp_path)


def test_native_position_draw_equals_the_reference_loop():
  """draw_patch_positions (one native call on numpy's MT19937 state) gives the
  numbers of the reference's three-randint-per-patch loop
  (dataset_generation.py:205-214) and leaves the generator where the loop
  leaves it: equal image sizes, one image (randint over a single value draws
  nothing), ragged image sizes, ranges with heavy rejection, the global
  generator; and it is fast enough not to stall a training step."""
  import time
  from utils import dataset_generation as dg
  cases = [((512, 512), 10, (16, 16), 5), ((512, 512), 1, (16, 16), 5),
           ([(100, 120), (64, 300), (257, 90)], 3, (8, 8), 0),
           ((16 + 2 * 4 + 129, 16 + 2 * 4 + 2), 33, (16, 16), 4)]
  for shapes, count, patch, edge in cases:
    a, b = np.random.RandomState(5), np.random.RandomState(5)
    fast = dg.draw_patch_positions(5000, shapes, patch, edge, count, rng=a)
    slow = dg.draw_patch_positions_loop(5000, shapes, patch, edge, count,
                                        rng=b)
    for x, y in zip(fast, slow):
      assert x.dtype == np.int32 and np.array_equal(x, y)
    assert a.randint(0, 1 << 30) == b.randint(0, 1 << 30)
    assert a.standard_normal() == b.standard_normal()
  np.random.seed(11)
  fast = dg.draw_patch_positions(3000, (64, 64), (16, 16), 2, 7)
  after_fast = np.random.randint(0, 1 << 30)
  np.random.seed(11)
  slow = dg.draw_patch_positions_loop(3000, (64, 64), (16, 16), 2, 7)
  assert np.random.randint(0, 1 << 30) == after_fast
  assert all(np.array_equal(x, y) for x, y in zip(fast, slow))
  rng = np.random.RandomState(1)
  dg.draw_patch_positions(1000, (512, 512), (16, 16), 5, 40, rng=rng)  # warm
  t0 = time.perf_counter()
  dg.draw_patch_positions(131072, (512, 512), (16, 16), 5, 40, rng=rng)
  assert time.perf_counter() - t0 < 0.05    # ~2-3 ms; the loop takes 0.3-0.4 s
  with pytest.raises(ValueError):             # an image smaller than a patch
    dg.draw_patch_positions(4, (20, 20), (16, 16), 5, 2, rng=rng)


def test_wide_buffer_stores_keep_a_constant_scalar_offset():
  """Source guard for a gfx950 / ROCm 7.2 code-generation hazard found in
  round 3 (csrc/epi_prox.h): a 12- or 16-byte buffer store whose scalar-offset
  operand is a register gets no wait state before a VALU instruction
  overwrites its data registers, and the store can then write the NEW value
  in a few lanes.  Every such store in the library passes its offsets through
  the vector offset and the constant 0 as the scalar offset."""
  import pathlib
  import re
  csrc = pathlib.Path(__file__).resolve().parent.parent / (
      'vision-transform-codes_amd') / 'csrc'
  call = re.compile(
      r'__builtin_amdgcn_raw(?:_ptr)?_buffer_store_b(?:96|128)\s*\(')
  found = 0
  for path in sorted(csrc.glob('*.h')) + sorted(csrc.glob('*.hip')):
    text = path.read_text()
    for m in call.finditer(text):
      depth, i, args, cur = 1, m.end(), [], ''
      while depth:
        ch = text[i]
        if ch == '(':
          depth += 1
        elif ch == ')':
          depth -= 1
          if depth == 0:
            break
        if ch == ',' and depth == 1:
          args.append(cur.strip())
          cur = ''
        else:
          cur += ch
        i += 1
      args.append(cur.strip())
      found += 1
      assert len(args) == 5 and args[3] == '0', (
          '%s: wide buffer store with scalar offset %r' % (path.name, args[3]))
  assert found >= 3
