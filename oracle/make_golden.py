"""
Generates tests/golden/*.npz by running the REFERENCE's own plugin functions.

Development-container only: it imports /root/reference (absent on the GPU box)
with the three container-only shims SURVEY.md section 8(c) lists, feeds it
seeded numpy.random.RandomState inputs and stores inputs + outputs as small
.npz vectors.  It also runs oracle/sc_oracle.py on the same inputs and prints
how far the restatement is from the reference (expected: 0, both run the same
torch CPU ops in the same order).

  python oracle/make_golden.py            # rewrite every fixture
  python oracle/make_golden.py fc_c1 ...  # only the named ones

Nothing here is imported by the product path.
"""
import os
import sys
import types
import pathlib

os.environ.setdefault('MPLBACKEND', 'Agg')
sys.dont_write_bytecode = True

import numpy as np
import torch

REPO = pathlib.Path(__file__).resolve().parent.parent
GOLDEN = REPO / 'tests' / 'golden'
REFERENCE = pathlib.Path('/root/reference/vision_transform_codes')

sys.path.insert(0, str(REPO / 'oracle'))
import sc_oracle  # noqa: E402


def import_reference():
  """Shims: torch.symeig (removed in torch>=2), skimage, h5py (not installed).
  The reference tree itself is never modified."""
  torch.symeig = lambda A, eigenvectors=False, upper=True: (
      torch.linalg.eigvalsh(A, UPLO='U' if upper else 'L'), A.new_empty(0))
  skimage = types.ModuleType('skimage')
  measure = types.ModuleType('skimage.measure')
  measure.compare_ssim = lambda *a, **k: 0.0
  skimage.measure = measure
  sys.modules.setdefault('skimage', skimage)
  sys.modules.setdefault('skimage.measure', measure)
  sys.modules.setdefault('h5py', types.ModuleType('h5py'))
  sys.path.insert(0, str(REFERENCE))
  import importlib
  ref = types.SimpleNamespace()
  ref.fc_inf = importlib.import_module(
      'analysis_transforms.fully_connected.ista_fista')
  ref.sub_inf = importlib.import_module(
      'analysis_transforms.fully_connected.subspace_ista_fista')
  ref.conv_inf = importlib.import_module(
      'analysis_transforms.convolutional.ista_fista')
  ref.fc_sd = importlib.import_module(
      'dict_update_rules.fully_connected.sc_steepest_descent')
  ref.fc_cq = importlib.import_module(
      'dict_update_rules.fully_connected.sc_cheap_quadratic_descent')
  ref.sub_cq = importlib.import_module(
      'dict_update_rules.fully_connected.subspace_sc_cheap_quadratic_descent')
  ref.conv_sd = importlib.import_module(
      'dict_update_rules.convolutional.sc_steepest_descent')
  ref.conv_cq = importlib.import_module(
      'dict_update_rules.convolutional.sc_cheap_quadratic_descent')
  ref.ica = importlib.import_module(
      'dict_update_rules.fully_connected.ica_natural_gradient')
  ref.conv_utils = importlib.import_module('utils.convolutions')
  ref.trainer = importlib.import_module('training.sparse_coding')
  ref.image_processing = importlib.import_module('utils.image_processing')
  return ref


# ---------------------------------------------------------------- inputs
def gaussian_patches(seed, b, n, scale=0.1):
  return (scale * np.random.RandomState(seed).randn(b, n)).astype(np.float32)


def unit_rows(seed, s, n):
  d = np.random.RandomState(seed).randn(s, n).astype(np.float32)
  return d / np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)


def unit_kernels(seed, s, c, kh, kw):
  d = np.random.RandomState(seed).randn(s, c, kh, kw).astype(np.float32)
  nrm = np.sqrt((d.astype(np.float64)**2).sum(axis=(1, 2, 3))).astype(
      np.float32)
  return d / nrm[:, None, None, None]


def T(a):
  return torch.from_numpy(np.ascontiguousarray(a))


def report(name, ours, theirs):
  ours = ours.detach().numpy().astype(np.float64)
  theirs = theirs.detach().numpy().astype(np.float64)
  denom = max(np.linalg.norm(theirs), 1e-30)
  print('   oracle vs reference  %-34s rel %.2e  max|d| %.2e  support-diff %d'
        % (name, np.linalg.norm(ours - theirs) / denom,
           np.abs(ours - theirs).max(),
           int(((ours != 0) != (theirs != 0)).sum())))


def ref_eta_fc(D):
  return float(1. / torch.linalg.eigvalsh(torch.mm(D.t(), D), UPLO='U')[-1])


THRESH_MODES = {'soft': dict(nonnegative_only=False, hard_threshold=False),
                'soft_nonneg': dict(nonnegative_only=True,
                                    hard_threshold=False),
                'hard': dict(nonnegative_only=False, hard_threshold=True),
                'hard_nonneg': dict(nonnegative_only=True,
                                    hard_threshold=True)}


def trace_reference(run_fn, iters_list, **kw):
  """The reference has no trace hook: run it once per checkpoint count."""
  return {k: run_fn(num_iters=k, **kw) for k in iters_list}


# ---------------------------------------------------------------- fixtures
def make_fc_c1(ref):
  """F1: config-1 shape.  X(256,64), D(64,64), ISTA T=20, four threshold
  modes; FISTA T=20 soft; both FC dictionary updates."""
  X = gaussian_patches(10, 256, 64)
  D = unit_rows(11, 64, 64)
  lam = 0.05
  out = {'images': X, 'dictionary': D, 'sparsity_weight': np.float32(lam),
         'stepsize': np.float32(ref_eta_fc(T(D)))}
  for mode, flags in THRESH_MODES.items():
    codes = ref.fc_inf.run(T(X), T(D), lam, 20, variant='ista', **flags)
    mine = sc_oracle.fc_ista_fista(T(X), T(D), lam, 20, variant='ista',
                                   **flags)
    report('fc_c1 ista ' + mode, mine, codes)
    out['codes_ista_' + mode] = codes.numpy()
  codes = ref.fc_inf.run(T(X), T(D), lam, 20, variant='fista')
  report('fc_c1 fista', sc_oracle.fc_ista_fista(T(X), T(D), lam, 20), codes)
  out['codes_fista_soft'] = codes.numpy()
  # early stopping: number of iterations actually taken is not returned by
  # the reference, only the codes
  codes_es = ref.fc_inf.run(T(X), T(D), lam, 500, variant='ista',
                            early_stopping_epsilon=1e-2)
  report('fc_c1 ista early-stop',
         sc_oracle.fc_ista_fista(T(X), T(D), lam, 500, variant='ista',
                                 early_stopping_epsilon=1e-2), codes_es)
  out['codes_ista_earlystop'] = codes_es.numpy()
  codes_es = ref.fc_inf.run(T(X), T(D), lam, 500, variant='fista',
                            early_stopping_epsilon=1e-2)
  report('fc_c1 fista early-stop',
         sc_oracle.fc_ista_fista(T(X), T(D), lam, 500, variant='fista',
                                 early_stopping_epsilon=1e-2), codes_es)
  out['codes_fista_earlystop'] = codes_es.numpy()
  # dictionary updates from the FISTA codes
  C = codes
  for name, fn, kw in (
      ('steepest', ref.fc_sd.run, {}),
      ('steepest_3it_nonorm', ref.fc_sd.run,
       dict(num_iters=3, normalize_dictionary=False))):
    Dn = T(D.copy())
    fn(T(X), Dn, C, stepsize=0.1, **kw)
    Dm = T(D.copy())
    sc_oracle.fc_steepest_descent(T(X), Dm, C, stepsize=0.1, **kw)
    report('fc_c1 ' + name, Dm, Dn)
    out['dict_after_' + name] = Dn.numpy()
  h = T(np.abs(np.random.RandomState(12).randn(64)).astype(np.float32) * 0.01)
  out['hessian_diagonal'] = h.numpy()
  Dn = T(D.copy())
  ref.fc_cq.run(T(X), Dn, C, h, stepsize=0.1, num_iters=2)
  Dm = T(D.copy())
  sc_oracle.fc_cheap_quadratic_descent(T(X), Dm, C, h, stepsize=0.1,
                                       num_iters=2)
  report('fc_c1 cheap-quad', Dm, Dn)
  out['dict_after_cheapquad_2it'] = Dn.numpy()
  np.savez_compressed(GOLDEN / 'fc_c1.npz', **out)


def make_fc_c2_mini(ref):
  """F2: mini config-2.  X(64,256), D(1024,256), lambda 0.008, FISTA with
  checkpoints after 1, 2, 20, 200 iterations; warm start; both updates.
  Inputs are regenerated from the seeds by the tests (not stored)."""
  X = gaussian_patches(0, 64, 256)
  D = unit_rows(1, 1024, 256)
  lam = 0.008
  out = {'seed_images': 0, 'seed_dictionary': 1,
         'sparsity_weight': np.float32(lam),
         'stepsize': np.float32(ref_eta_fc(T(D))),
         'images_sum': np.float64(X.astype(np.float64).sum()),
         'dictionary_sum': np.float64(D.astype(np.float64).sum())}
  traced = trace_reference(ref.fc_inf.run, [1, 2, 20, 200], images=T(X),
                           dictionary=T(D), sparsity_weight=lam,
                           variant='fista')
  mine, mine_trace = sc_oracle.fc_ista_fista(T(X), T(D), lam, 200,
                                             trace_at=[1, 2, 20, 200])
  for k, codes in traced.items():
    report('fc_c2 fista T=%d' % k, mine_trace[k], codes)
    out['codes_fista_T%d' % k] = codes.numpy()
  # float64 truth for scoring (oracle, not reference)
  truth = sc_oracle.fc_ista_fista(T(X).double(), T(D).double(), lam, 200)
  report('fc_c2 fp64-oracle vs fp32-reference', truth.float(), traced[200])
  out['codes_fista_T200_fp64'] = truth.numpy()
  # ISTA 50 iterations
  codes = ref.fc_inf.run(T(X), T(D), lam, 50, variant='ista')
  report('fc_c2 ista T=50',
         sc_oracle.fc_ista_fista(T(X), T(D), lam, 50, variant='ista'), codes)
  out['codes_ista_T50'] = codes.numpy()
  # warm start from the T=20 codes, 20 more FISTA iterations
  warm = ref.fc_inf.run(T(X), T(D), lam, 20, variant='fista',
                        initial_codes=traced[20])
  report('fc_c2 warm start',
         sc_oracle.fc_ista_fista(T(X), T(D), lam, 20, initial_codes=traced[20]),
         warm)
  out['codes_fista_warm20'] = warm.numpy()
  C = traced[200]
  Dn = T(D.copy())
  ref.fc_sd.run(T(X), Dn, C, stepsize=0.1)
  Dm = T(D.copy())
  sc_oracle.fc_steepest_descent(T(X), Dm, C, stepsize=0.1)
  report('fc_c2 steepest', Dm, Dn)
  out['dict_after_steepest'] = Dn.numpy()
  h = torch.zeros(1024)
  h.mul_(0.99).add_(torch.pow(C, 2).mean(0) / 100)   # trainer's EMA, step 0
  report('fc_c2 hessian ema', sc_oracle.hessian_diag_ema_(torch.zeros(1024), C),
         h)
  out['hessian_diagonal'] = h.numpy()
  Dn = T(D.copy())
  ref.fc_cq.run(T(X), Dn, C, h, stepsize=0.1)
  Dm = T(D.copy())
  sc_oracle.fc_cheap_quadratic_descent(T(X), Dm, C, h, stepsize=0.1)
  report('fc_c2 cheap-quad', Dm, Dn)
  out['dict_after_cheapquad'] = Dn.numpy()
  np.savez_compressed(GOLDEN / 'fc_c2_mini.npz', **out)


def subspace_groupings():
  return {
      'ragged_overlap': [[0, 2, 5], [1], [2, 3, 4, 5]],
      'groups_of_4': [list(range(4 * g, 4 * g + 4)) for g in range(16)],
  }


def make_subspace(ref):
  """F3: subspace inference + subspace cheap-quadratic update."""
  out = {}
  lam = 0.02
  # (i) docstring grouping on a 6-atom dictionary
  X = gaussian_patches(20, 32, 16)
  D = unit_rows(21, 6, 16)
  groups = subspace_groupings()['ragged_overlap']
  out['ro_images'], out['ro_dictionary'] = X, D
  for variant in ('ista', 'fista'):
    codes = ref.sub_inf.run(T(X), T(D), groups, lam, 30, variant=variant)
    report('subspace ragged ' + variant,
           sc_oracle.subspace_ista_fista(T(X), T(D), groups, lam, 30,
                                         variant=variant), codes)
    out['ro_codes_' + variant] = codes.numpy()
  warm = ref.sub_inf.run(T(X), T(D), groups, lam, 10, variant='fista',
                         initial_codes=T(out['ro_codes_ista']))
  report('subspace ragged warm',
         sc_oracle.subspace_ista_fista(
             T(X), T(D), groups, lam, 10,
             initial_codes=T(out['ro_codes_ista'])), warm)
  out['ro_codes_warm'] = warm.numpy()
  # (ii) reference-test geometry: 64 atoms, 16 groups of 4, 16x16 patches
  X = gaussian_patches(22, 48, 256)
  D = unit_rows(23, 64, 256)
  groups = subspace_groupings()['groups_of_4']
  out['g4_images'], out['g4_dictionary'] = X, D
  codes = ref.sub_inf.run(T(X), T(D), groups, lam, 40, variant='fista')
  report('subspace groups-of-4 fista',
         sc_oracle.subspace_ista_fista(T(X), T(D), groups, lam, 40), codes)
  out['g4_codes_fista'] = codes.numpy()
  h = torch.zeros(64)
  h.mul_(0.99).add_(torch.pow(codes, 2).mean(0) / 100)
  out['g4_hessian'] = h.numpy()
  for pen_name, pen in (('pen0', 0.), ('pen2e-4', 2e-4), ('pen0.05', 0.05)):
    Dn = T(D.copy())
    ref.sub_cq.run(T(X), Dn, codes, groups, h, pen, stepsize=0.1)
    Dm = T(D.copy())
    sc_oracle.subspace_cheap_quadratic_descent(T(X), Dm, codes, groups, h,
                                               pen, stepsize=0.1)
    report('subspace cheap-quad ' + pen_name, Dm, Dn)
    out['g4_dict_after_' + pen_name] = Dn.numpy()
  # un-normalised alignment gradient branch
  Dn = T((D * 1.5).copy())
  ref.sub_cq.run(T(X), Dn, codes, groups, h, 0.05, stepsize=0.1,
                 normalize_dictionary=False)
  Dm = T((D * 1.5).copy())
  sc_oracle.subspace_cheap_quadratic_descent(
      T(X), Dm, codes, groups, h, 0.05, stepsize=0.1,
      normalize_dictionary=False)
  report('subspace cheap-quad un-normalised', Dm, Dn)
  out['g4_dict_after_pen0.05_nonorm'] = Dn.numpy()
  # (iii) mini config-4: 512 atoms, 64 groups of 8
  X = gaussian_patches(24, 32, 256)
  D = unit_rows(25, 512, 256)
  groups = [list(map(int, g)) for g in np.array_split(np.arange(512), 64)]
  out['c4_seed_images'], out['c4_seed_dictionary'] = 24, 25
  codes = ref.sub_inf.run(T(X), T(D), groups, 0.008, 50, variant='fista')
  report('subspace mini-c4 fista',
         sc_oracle.subspace_ista_fista(T(X), T(D), groups, 0.008, 50), codes)
  out['c4_codes_fista'] = codes.numpy()
  np.savez_compressed(GOLDEN / 'subspace.npz', **out)


def make_conv(ref):
  """F4: convolutional inference + both conv updates on two geometries."""
  out = {}
  lam = 0.05
  geoms = {
      # reference test geometry (vtc/tests/ista_fista_2.py:16-24) on 32x32
      'k16s8': dict(img=32, k=16, stride=8, s=16, b=3, seed=30),
      # stride-1 11x11 (config-5 geometry) on 32x32
      'k11s1': dict(img=32, k=11, stride=1, s=8, b=2, seed=32),
      # image size not a multiple of the stride: ragged trailing pad (the
      # reference's geometry only closes when the kernel is a multiple of the
      # stride, so 8/4 here)
      'k8s4_ragged': dict(img=30, k=8, stride=4, s=5, b=2, seed=34),
  }
  for name, g in geoms.items():
    lead, trail = ref.conv_utils.get_padding_amt(g['img'], g['k'], g['stride'])
    assert (lead, trail) == sc_oracle.conv_padding_amount(
        g['img'], g['k'], g['stride'])
    padded = g['img'] + lead + trail
    rs = np.random.RandomState(g['seed'])
    imgs = np.zeros((g['b'], 1, padded, padded), np.float32)
    imgs[:, :, lead:lead + g['img'], lead:lead + g['img']] = (
        0.5 * rs.randn(g['b'], 1, g['img'], g['img'])).astype(np.float32)
    D = unit_kernels(g['seed'] + 1, g['s'], 1, g['k'], g['k'])
    stride = (g['stride'], g['stride'])
    pad = ((lead, trail), (lead, trail))
    out[name + '_images_padded'] = imgs
    out[name + '_dictionary'] = D
    out[name + '_stride'] = np.array(stride)
    out[name + '_padding'] = np.array(pad)
    for variant in ('ista', 'fista'):
      codes = ref.conv_inf.run(T(imgs), T(D), stride, pad, lam, 10,
                               variant=variant)
      report('conv %s %s' % (name, variant),
             sc_oracle.conv_ista_fista(T(imgs), T(D), stride, pad, lam, 10,
                                       variant=variant), codes)
      out['%s_codes_%s' % (name, variant)] = codes.numpy()
    codes_nn = ref.conv_inf.run(T(imgs), T(D), stride, pad, lam, 10,
                                variant='ista', nonnegative_only=True,
                                hard_threshold=True)
    report('conv %s ista hard nonneg' % name,
           sc_oracle.conv_ista_fista(T(imgs), T(D), stride, pad, lam, 10,
                                     variant='ista', nonnegative_only=True,
                                     hard_threshold=True), codes_nn)
    out[name + '_codes_ista_hard_nonneg'] = codes_nn.numpy()
    Dn = T(D.copy())
    ref.conv_sd.run(T(imgs), Dn, codes, stride, pad, stepsize=0.005)
    Dm = T(D.copy())
    sc_oracle.conv_steepest_descent(T(imgs), Dm, codes, stride, pad,
                                    stepsize=0.005)
    report('conv %s steepest' % name, Dm, Dn)
    out[name + '_dict_after_steepest'] = Dn.numpy()
    h = torch.zeros(g['s'])
    h.mul_(0.99).add_(torch.mean(torch.sum(codes**2, dim=(2, 3)), dim=0) / 100)
    report('conv %s hessian ema' % name,
           sc_oracle.hessian_diag_ema_(torch.zeros(g['s']), codes), h)
    out[name + '_hessian'] = h.numpy()
    Dn = T(D.copy())
    ref.conv_cq.run(T(imgs), Dn, codes, h, stride, pad, stepsize=0.005)
    Dm = T(D.copy())
    sc_oracle.conv_cheap_quadratic_descent(T(imgs), Dm, codes, h, stride, pad,
                                           stepsize=0.005)
    report('conv %s cheap-quad' % name, Dm, Dn)
    out[name + '_dict_after_cheapquad'] = Dn.numpy()
  np.savez_compressed(GOLDEN / 'conv.npz', **out)


def near_delta_kernels(seed, s, k, eps):
  """Unit-norm kernels that are a centred delta plus eps * N(0,1): nearly
  parallel as vectors (lambda_max(F F^T) ~ s) and nearly white as filters, so
  that the reference's own step 1 / lambda_max(F F^T)
  (convolutional/ista_fista.py:104-113) is a valid step for a STRIDE-1
  convolution and the iteration converges.  (With random kernels that estimate
  is far below the operator norm and the iterates grow geometrically.)"""
  rs = np.random.RandomState(seed)
  d = (eps * rs.randn(s, 1, k, k)).astype(np.float32)
  d[:, 0, k // 2, k // 2] += 1.0
  nrm = np.sqrt((d.astype(np.float64)**2).sum(axis=(1, 2, 3))).astype(
      np.float32)
  return (d / nrm[:, None, None, None]).astype(np.float32)


def make_conv_long(ref):
  """Long-horizon convolutional traces (T = 10, 50, 100, FISTA, the
  reference's own step size): the reference's example geometry -- 64 kernels
  of 16x16, stride 8 (vtc/examples/train_convolutional_sparse_coding.py:39,
  vtc/tests/ista_fista_2.py:16-24), whose iterates GROW with random kernels
  (4.8e27 at T = 100, still finite in float32) -- and a convergent stride-1
  11x11 case with 64 near-delta kernels (the geometry class of BASELINE
  configs[4], on the fused matrix-core kernel)."""
  out = {}
  lam = 0.05
  cases = {
      'ex_k16s8': dict(img=64, k=16, stride=8, s=64, b=2, seed=40,
                       kernels=lambda: unit_kernels(41, 64, 1, 16, 16)),
      'nd_k11s1': dict(img=40, k=11, stride=1, s=64, b=2, seed=44,
                       kernels=lambda: near_delta_kernels(45, 64, 11, 0.02)),
  }
  for name, g in cases.items():
    lead, trail = ref.conv_utils.get_padding_amt(g['img'], g['k'], g['stride'])
    padded = g['img'] + lead + trail
    rs = np.random.RandomState(g['seed'])
    imgs = np.zeros((g['b'], 1, padded, padded), np.float32)
    imgs[:, :, lead:lead + g['img'], lead:lead + g['img']] = (
        0.5 * rs.randn(g['b'], 1, g['img'], g['img'])).astype(np.float32)
    D = g['kernels']()
    stride = (g['stride'], g['stride'])
    pad = ((lead, trail), (lead, trail))
    out[name + '_images_padded'] = imgs
    out[name + '_dictionary'] = D
    out[name + '_stride'] = np.array(stride)
    out[name + '_padding'] = np.array(pad)
    F = T(D.reshape(D.shape[0], -1))
    out[name + '_stepsize'] = np.float32(
        1. / torch.linalg.eigvalsh(torch.mm(F, F.t()), UPLO='U')[-1])
    for iters in (10, 50, 100):
      codes = ref.conv_inf.run(T(imgs), T(D), stride, pad, lam, iters,
                               variant='fista')
      assert bool(torch.isfinite(codes).all())
      report('conv_long %s fista T=%d' % (name, iters),
             sc_oracle.conv_ista_fista(T(imgs), T(D), stride, pad, lam, iters,
                                       variant='fista'), codes)
      out['%s_codes_fista_T%d' % (name, iters)] = codes.numpy()
  out['sparsity_weight'] = np.float32(lam)
  np.savez_compressed(GOLDEN / 'conv_long.npz', **out)


class _ListDataset(torch.utils.data.Dataset):
  def __init__(self, tensor):
    self.tensor = tensor

  def __len__(self):
    return self.tensor.shape[0]

  def __getitem__(self, idx):
    return self.tensor[idx]


def make_trainer(ref):
  """F5: three-step trainer trajectories through the reference's own
  train_dictionary (shuffle off), FC fista + cheap-quad and conv ista +
  steepest, with a schedule change at step 2."""
  out = {}
  X = gaussian_patches(40, 96, 64)
  D0 = unit_rows(41, 128, 64)
  params = {
      'mode': 'fully-connected', 'num_epochs': 1,
      'code_inference_algorithm': 'fista',
      'inference_param_schedule': {
          0: {'sparsity_weight': 0.02, 'num_iters': 15},
          2: {'sparsity_weight': 0.01, 'num_iters': 30}},
      'dictionary_update_algorithm': 'sc_cheap_quadratic_descent',
      'dict_update_param_schedule': {
          0: {'stepsize': 0.1, 'num_iters': 1},
          2: {'stepsize': 0.05, 'num_iters': 2}}}
  loader = torch.utils.data.DataLoader(_ListDataset(T(X)), batch_size=32,
                                       shuffle=False)
  Dref = T(D0.copy())
  # one run per prefix length so that every intermediate dictionary is seen
  for steps in (1, 2, 3):
    Dref = T(D0.copy())
    sub = torch.utils.data.DataLoader(_ListDataset(T(X[:32 * steps])),
                                      batch_size=32, shuffle=False)
    ref.trainer.train_dictionary(sub, sub, Dref, dict(params))
    out['fc_dict_after_step%d' % steps] = Dref.numpy().copy()
  Dm = T(D0.copy())
  hist = sc_oracle.train_steps([T(X[32 * i: 32 * i + 32]) for i in range(3)],
                               Dm, params)
  for i in range(3):
    report('trainer fc step %d' % (i + 1), hist[i]['dictionary'],
           T(out['fc_dict_after_step%d' % (i + 1)]))
  out['fc_images'], out['fc_dictionary0'] = X, D0
  out['fc_hessian_after_step3'] = hist[2]['hessian'].numpy()
  # convolutional: ista + steepest
  lead, trail = sc_oracle.conv_padding_amount(16, 8, 4)
  padded = 16 + lead + trail
  rs = np.random.RandomState(42)
  imgs = np.zeros((6, 1, padded, padded), np.float32)
  imgs[:, :, lead:lead + 16, lead:lead + 16] = (
      0.5 * rs.randn(6, 1, 16, 16)).astype(np.float32)
  K0 = unit_kernels(43, 6, 1, 8, 8)
  cparams = {
      'mode': 'convolutional', 'num_epochs': 1,
      'code_inference_algorithm': 'ista',
      'strides': (4, 4), 'padding': ((lead, trail), (lead, trail)),
      'inference_param_schedule': {
          0: {'sparsity_weight': 0.05, 'num_iters': 8}},
      'dictionary_update_algorithm': 'sc_cheap_quadratic_descent',
      'dict_update_param_schedule': {
          0: {'stepsize': 0.005, 'num_iters': 1}}}
  for steps in (1, 2, 3):
    Kref = T(K0.copy())
    sub = torch.utils.data.DataLoader(_ListDataset(T(imgs[:2 * steps])),
                                      batch_size=2, shuffle=False)
    ref.trainer.train_dictionary(sub, sub, Kref, dict(cparams))
    out['conv_dict_after_step%d' % steps] = Kref.numpy().copy()
  Km = T(K0.copy())
  hist = sc_oracle.train_steps([T(imgs[2 * i: 2 * i + 2]) for i in range(3)],
                               Km, cparams)
  for i in range(3):
    report('trainer conv step %d' % (i + 1), hist[i]['dictionary'],
           T(out['conv_dict_after_step%d' % (i + 1)]))
  out['conv_images_padded'], out['conv_dictionary0'] = imgs, K0
  out['conv_padding'] = np.array(cparams['padding'])
  np.savez_compressed(GOLDEN / 'trainer.npz', **out)


def trainer_c2_params():
  """The parameter dictionary behind tests/golden/trainer_c2.npz (also used by
  tests/test_pipeline_gpu.py)."""
  return {
      'mode': 'fully-connected', 'num_epochs': 1,
      'code_inference_algorithm': 'fista',
      'inference_param_schedule': {
          0: {'sparsity_weight': 0.008, 'num_iters': 50}},
      'dictionary_update_algorithm': 'sc_cheap_quadratic_descent',
      'dict_update_param_schedule': {0: {'stepsize': 0.1, 'num_iters': 1}}}


def make_trainer_c2(ref):
  """Headline shape end to end: three steps of the reference's own
  train_dictionary at n = 256, s = 1024 (b = 64, FISTA T = 50, cheap-quadratic
  update).  Inputs are regenerated from their seeds by the test (patches seed
  60, dictionary seed 61); stored: the dictionary after steps 1 and 3 (float32),
  the step-2 dictionary as a checksum vector (row sums in float64), the
  Hessian diagonal after step 3, the reference's stepsize eta at each step and
  the codes of the first step."""
  out = {}
  X = gaussian_patches(60, 192, 256)
  D0 = unit_rows(61, 1024, 256)
  params = trainer_c2_params()
  dicts = []
  for steps in (1, 2, 3):
    Dref = T(D0.copy())
    sub = torch.utils.data.DataLoader(_ListDataset(T(X[:64 * steps])),
                                      batch_size=64, shuffle=False)
    ref.trainer.train_dictionary(sub, sub, Dref, dict(params))
    dicts.append(Dref.numpy().copy())
  out['dict_after_step1'] = dicts[0]
  out['dict_after_step3'] = dicts[2]
  out['dict_after_step2_rowsum'] = dicts[1].astype(np.float64).sum(axis=1)
  out['dict_after_step2_colsum'] = dicts[1].astype(np.float64).sum(axis=0)
  # eta the reference's inference used at each step (D0, D1, D2)
  out['eta'] = np.array([ref_eta_fc(T(D0)), ref_eta_fc(T(dicts[0])),
                         ref_eta_fc(T(dicts[1]))], np.float32)
  out['codes_step1'] = ref.fc_inf.run(T(X[:64]), T(D0), 0.008, 50,
                                      variant='fista').numpy()
  Dm = T(D0.copy())
  hist = sc_oracle.train_steps([T(X[64 * i: 64 * i + 64]) for i in range(3)],
                               Dm, params)
  for i in range(3):
    report('trainer_c2 step %d' % (i + 1), hist[i]['dictionary'],
           T(dicts[i]))
  report('trainer_c2 codes step 1', hist[0]['codes'], T(out['codes_step1']))
  out['hessian_after_step3'] = hist[2]['hessian'].numpy()
  out['patch_seed'], out['dict_seed'] = np.int64(60), np.int64(61)
  np.savez_compressed(GOLDEN / 'trainer_c2.npz', **out)


def reset_prune_cases():
  """(tag, filter_type, filter_params without groups, action, numpy seed,
  torch seed) behind tests/golden/reset_prune.npz."""
  return [
      ('random_reset', 'random', {'num_to_modify': 5}, 'reset', 3, 4),
      ('random_prune', 'random', {'num_to_modify': 4}, 'prune', 5, 6),
      ('cos_reset', 'cosine_sim_threshold',
       {'cue_user': False, 'only_sim_within_group': False, 'threshold': 0.9},
       'reset', 7, 8),
      ('cos_prune', 'cosine_sim_threshold',
       {'cue_user': False, 'only_sim_within_group': False, 'threshold': 0.9},
       'prune', 9, 10),
      ('cos_group_prune', 'cosine_sim_threshold',
       {'cue_user': False, 'only_sim_within_group': True, 'threshold': 0.9},
       'prune', 11, 12),
      ('cos_group_reset', 'cosine_sim_threshold',
       {'cue_user': False, 'only_sim_within_group': True, 'threshold': 0.9},
       'reset', 13, 14),
      ('nonuniform_reset', 'nonuniformity_within_group',
       {'num_gc_in_average': 6}, 'reset', 15, 16)]


def reset_prune_inputs():
  """A 32-atom dictionary in 8 groups of 4 with three near-duplicate pairs
  (two of them inside a group), and codes whose phase inside two groups is
  concentrated on one direction."""
  rs = np.random.RandomState(70)
  D = rs.randn(32, 16).astype(np.float32)
  D[5] = D[4] + 0.05 * rs.randn(16).astype(np.float32)      # within group 1
  D[14] = -D[13] + 0.05 * rs.randn(16).astype(np.float32)   # within group 3
  D[30] = D[2] + 0.05 * rs.randn(16).astype(np.float32)     # across groups
  D /= np.linalg.norm(D, axis=1, keepdims=True)
  groups = [list(range(4 * g, 4 * g + 4)) for g in range(8)]
  C = np.zeros((600, 32), np.float32)
  for g in range(8):
    rows = rs.rand(600) < 0.5
    C[np.ix_(rows, groups[g])] = rs.randn(int(rows.sum()), 4)
  for g in (2, 6):                                # lopsided phases
    C[:, groups[g][1:]] *= 0.02
  return D, groups, C


def make_reset_prune(ref):
  """f4: the reference's own reset_or_prune_dict_elements on CPU tensors, every
  non-interactive mode; numpy and torch seeded per case."""
  D0, groups0, C = reset_prune_inputs()
  out = {'dictionary': D0, 'codes': C}
  for tag, f_type, f_params, action, np_seed, torch_seed in reset_prune_cases():
    results = []
    for fn in (ref.trainer.reset_or_prune_dict_elements,
               sc_oracle.reset_or_prune):
      groups = [list(g) for g in groups0]
      params = dict(f_params)
      params.update({'group_assignments': groups,
                     'coding_mode': 'fully-connected'})
      np.random.seed(np_seed)
      torch.manual_seed(torch_seed)
      Dn, rows = fn(T(D0.copy()), T(C), f_type, params, action)
      results.append((Dn, np.asarray(rows), groups))
    (Dr, rows_r, groups_r), (Dm, rows_m, groups_m) = results
    assert np.array_equal(rows_r, rows_m) and groups_r == groups_m, tag
    report('reset_prune ' + tag, Dm, Dr)
    print('      affected', rows_r.tolist())
    assert len(rows_r) > 0, tag
    out[tag + '_dictionary'] = Dr.numpy()
    out[tag + '_affected'] = rows_r.astype(np.int64)
    out[tag + '_group_sizes'] = np.array([len(g) for g in groups_r], np.int64)
    out[tag + '_groups_flat'] = np.array(
        [a for g in groups_r for a in g], np.int64)
  np.savez_compressed(GOLDEN / 'reset_prune.npz', **out)


def make_ica(ref):
  """F8: the ICA natural-gradient update rule (f4 sibling of the dictionary
  update plugins) on sparse codes, one and three iterations, square and
  overcomplete dictionaries."""
  out = {}
  X = gaussian_patches(70, 200, 64)
  for tag, s_atoms in (('square', 64), ('wide', 96)):
    D0 = unit_rows(71 + s_atoms, s_atoms, 64)
    C = ref.fc_inf.run(T(X), T(D0), 0.02, 10, variant='fista')
    out[tag + '_dictionary0'] = D0
    out[tag + '_codes'] = C.numpy()
    for iters in (1, 3):
      Dref = T(D0.copy())
      ref.ica.run(Dref, C, stepsize=0.01, num_iters=iters)
      Dm = T(D0.copy())
      sc_oracle.ica_natural_gradient(Dm, C, stepsize=0.01, num_iters=iters)
      report('ica %s %d iters' % (tag, iters), Dm, Dref)
      out['%s_dictionary_after_%d' % (tag, iters)] = Dref.numpy().copy()
  np.savez_compressed(GOLDEN / 'ica.npz', **out)


class _RecordingWriter(object):
  """Stand-in for torch.utils.tensorboard.SummaryWriter (tensorboard is not
  installed): keeps the scalars the reference sends, ignores the images."""
  records = []

  def __init__(self, *args, **kwargs):
    pass

  def add_scalar(self, tag, value, step):
    _RecordingWriter.records.append((tag, float(value), int(step)))

  def add_image(self, *args, **kwargs):
    pass


def make_metrics(ref):
  """F7: the validation metrics the reference's own train_dictionary sends to
  its SummaryWriter (compute_metrics closure, sparse_coding.py:177-229, via
  :497-505) at iterations 0 and 2 -- fully-connected fista, subspace fista and
  convolutional ista with padding."""
  import shutil
  import tempfile
  tb = types.ModuleType('torch.utils.tensorboard')
  tb.SummaryWriter = _RecordingWriter
  sys.modules['torch.utils.tensorboard'] = tb
  out = {}
  tmp = pathlib.Path(tempfile.mkdtemp(dir=str(REPO / 'oracle')))

  def run(tag, params, train, val, D0, batch):
    _RecordingWriter.records = []
    D = T(D0.copy())
    tl = torch.utils.data.DataLoader(_ListDataset(T(train)), batch_size=batch,
                                     shuffle=False)
    vl = torch.utils.data.DataLoader(_ListDataset(T(val)), batch_size=batch,
                                     shuffle=False)
    p = dict(params)
    p['logging_folder_fullpath'] = tmp / tag
    ref.trainer.train_dictionary(tl, vl, D, p)
    names = sorted(set(r[0] for r in _RecordingWriter.records))
    steps = sorted(set(r[2] for r in _RecordingWriter.records))
    table = np.zeros((len(steps), len(names)))
    for name, value, step in _RecordingWriter.records:
      table[steps.index(step), names.index(name)] = value
    out[tag + '_names'] = np.array(names)
    out[tag + '_steps'] = np.array(steps)
    out[tag + '_values'] = table
    # the restatement, same loop
    Dm = T(D0.copy())
    q = dict(params)
    hist = sc_oracle.train_steps(
        [T(train[batch * i: batch * i + batch])
         for i in range(len(train) // batch)], Dm, q,
        validation_batches=[T(val[batch * i: batch * i + batch])
                            for i in range(len(val) // batch)])
    for si, step in enumerate(steps):
      for ni, name in enumerate(names):
        ours = float(hist[step]['validation'][name])
        theirs = table[si, ni]
        print('  %-22s step %d %-40s ref %.9g  oracle %.9g  rel %.2e' % (
            tag, step, name, theirs, ours,
            abs(ours - theirs) / max(abs(theirs), 1e-30)))
    report('metrics %s final dictionary' % tag, Dm, D)
    return D.numpy().copy()

  X = gaussian_patches(60, 96, 64)
  V = gaussian_patches(61, 64, 64)
  D0 = unit_rows(62, 128, 64)
  fc = {
      'mode': 'fully-connected', 'num_epochs': 1,
      'code_inference_algorithm': 'fista',
      'inference_param_schedule': {
          0: {'sparsity_weight': 0.02, 'num_iters': 15}},
      'dictionary_update_algorithm': 'sc_steepest_descent',
      'dict_update_param_schedule': {0: {'stepsize': 0.1, 'num_iters': 1}},
      'training_visualization_schedule': set([0, 2]),
      'reshaped_kernel_size': (8, 8)}
  out['fc_images'], out['fc_validation'], out['fc_dictionary0'] = X, V, D0
  out['fc_dictionary_final'] = run('fc', fc, X, V, D0, 32)
  groups = [list(range(4 * g, 4 * g + 4)) for g in range(32)]
  sub = dict(fc)
  sub.update({'code_inference_algorithm': 'subspace_fista',
              'dictionary_update_algorithm':
                  'subspace_sc_cheap_quadratic_descent',
              'group_assignments': groups,
              'subspace_alignment_penalty': 2e-4})
  out['sub_dictionary_final'] = run('sub', sub, X, V, D0, 32)
  lead, trail = sc_oracle.conv_padding_amount(16, 8, 4)
  padded = 16 + lead + trail
  rs = np.random.RandomState(63)
  imgs = np.zeros((10, 1, padded, padded), np.float32)
  imgs[:, :, lead:lead + 16, lead:lead + 16] = (
      0.5 * rs.randn(10, 1, 16, 16)).astype(np.float32)
  K0 = unit_kernels(64, 6, 1, 8, 8)
  conv = {
      'mode': 'convolutional', 'num_epochs': 1,
      'code_inference_algorithm': 'ista',
      'strides': (4, 4), 'padding': ((lead, trail), (lead, trail)),
      'inference_param_schedule': {
          0: {'sparsity_weight': 0.05, 'num_iters': 8}},
      'dictionary_update_algorithm': 'sc_steepest_descent',
      'dict_update_param_schedule': {0: {'stepsize': 0.005, 'num_iters': 1}},
      'training_visualization_schedule': set([0, 2])}
  out['conv_images_padded'], out['conv_validation'] = imgs[:6], imgs[6:]
  out['conv_dictionary0'] = K0
  out['conv_padding'] = np.array(conv['padding'])
  out['conv_dictionary_final'] = run('conv', conv, imgs[:6], imgs[6:], K0, 2)
  shutil.rmtree(str(tmp), ignore_errors=True)
  np.savez_compressed(GOLDEN / 'metrics.npz', **out)


def make_whitened(ref):
  """F6: 'realistic' patches.  Synthetic 1/f images -> the reference's own
  whiten_center_surround (vtc/utils/image_processing.py:267-308, parameters
  of vtc/utils/dataset_generation.py:112-120) -> 16x16 patches; FISTA T=100
  on a 512-atom dictionary."""
  rs = np.random.RandomState(50)
  size = 128
  fy = np.fft.fftfreq(size)[:, None]
  fx = np.fft.fftfreq(size)[None, :]
  amp = 1. / np.maximum(np.sqrt(fy**2 + fx**2), 1. / size)
  imgs = []
  for _ in range(2):
    spec = amp * np.exp(2j * np.pi * rs.rand(size, size))
    img = np.real(np.fft.ifft2(spec))
    img = (img - img.min()) / (img.max() - img.min())   # data range [0, 1]
    imgs.append(img.astype(np.float32))
  white = [ref.image_processing.whiten_center_surround(
      im[:, :, None], cutoffs={'low': 1e-3, 'high': 0.9},
      norm_and_threshold=False)[:, :, 0] for im in imgs]
  # the reference function's own default, norm_and_threshold=True (transfer
  # function divided by its maximum, floored at 1e-3): same positions
  white_nat = [ref.image_processing.whiten_center_surround(
      im[:, :, None], cutoffs={'low': 1e-3, 'high': 0.9})[:, :, 0]
               for im in imgs]
  patches, patches_nat = [], []
  for w, wn in zip(white, white_nat):
    for _ in range(64):
      y, x = rs.randint(5, size - 21, size=2)
      patches.append(np.asarray(w)[y:y + 16, x:x + 16].reshape(-1))
      patches_nat.append(np.asarray(wn)[y:y + 16, x:x + 16].reshape(-1))
  X = np.stack(patches).astype(np.float32)
  X_nat = np.stack(patches_nat).astype(np.float32)
  D = unit_rows(51, 512, 256)
  lam = 0.008
  codes = ref.fc_inf.run(T(X), T(D), lam, 100, variant='fista')
  report('whitened fista T=100', sc_oracle.fc_ista_fista(T(X), T(D), lam, 100),
         codes)
  print('   whitened patches: std %.4f, non-zero fraction %.3f'
        % (X.std(), float((codes != 0).float().mean())))
  np.savez_compressed(
      GOLDEN / 'whitened.npz', images=X, images_norm_and_threshold=X_nat,
      seed_dictionary=51,
      sparsity_weight=np.float32(lam),
      stepsize=np.float32(ref_eta_fc(T(D))), codes_fista_T100=codes.numpy())


MAKERS = {'fc_c1': make_fc_c1, 'fc_c2_mini': make_fc_c2_mini,
          'subspace': make_subspace, 'conv': make_conv,
          'conv_long': make_conv_long,
          'trainer': make_trainer, 'trainer_c2': make_trainer_c2, 'reset_prune': make_reset_prune,
          'whitened': make_whitened,
          'metrics': make_metrics, 'ica': make_ica}


def main(argv):
  torch.set_num_threads(8)
  GOLDEN.mkdir(parents=True, exist_ok=True)
  ref = import_reference()
  for name in (argv or list(MAKERS)):
    print('== ' + name)
    MAKERS[name](ref)
  for f in sorted(GOLDEN.glob('*.npz')):
    print('%-24s %8.1f KiB' % (f.name, f.stat().st_size / 1024))


if __name__ == '__main__':
  main(sys.argv[1:])
