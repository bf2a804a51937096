"""
CPU oracle for the sparse-coding hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, on PyTorch *CPU* tensors, what the reference computes on the
path SURVEY.md section 8 names (rows a1..a10).  It exists so that the HIP
kernels can be checked against an independent implementation on the GPU box,
where /root/reference is absent.  Only tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py may import it; the product path (the package under
vision-transform-codes_amd/) never does and fails loudly without its HIP library.

Pinning: every public function here is compared, in the development container,
against the reference's own functions run on the same seeded inputs
(oracle/make_golden.py) and the resulting vectors are committed under
tests/golden/.  tests/test_oracle_golden.py re-checks this file against those
vectors on every run, so the oracle is pinned ("parity pinned by reference
outputs generated here"; the reference ships no numeric known-answers itself,
SURVEY.md section 4).

Paths below are relative to /root/reference/vision_transform_codes/ ("vtc/").

All functions take and return torch CPU tensors.  `dtype` is float32 to follow
the reference bit for bit (same op order, same scalar rounding) or float64 to
get a higher-precision "truth" against which both the reference and the HIP
kernels can be scored.
"""
import math

import numpy as np

import torch


# --------------------------------------------------------------------------
# scalars shared by every ISTA/FISTA flavour
# --------------------------------------------------------------------------
def largest_eigenvalue(sym_matrix):
  """Largest eigenvalue of a symmetric matrix as a 0-d tensor.

  The reference asks torch.symeig for the spectrum and takes the last entry
  (vtc/analysis_transforms/fully_connected/ista_fista.py:72-80).  symeig was
  removed from torch >= 2; eigvalsh is the same LAPACK driver family and
  returns ascending eigenvalues too.
  """
  return torch.linalg.eigvalsh(sym_matrix, UPLO='U')[-1]


def fc_stepsize(dictionary):
  """eta = 1 / lambda_max(D^T D), D is (s, n).
  vtc/analysis_transforms/fully_connected/ista_fista.py:72-80."""
  return 1. / largest_eigenvalue(torch.mm(dictionary.t(), dictionary))


def conv_stepsize(dictionary):
  """eta = 1 / lambda_max(F F^T), F = kernels flattened to (s, c*kh*kw).
  vtc/analysis_transforms/convolutional/ista_fista.py:104-113."""
  flat = dictionary.reshape(dictionary.shape[0], -1)
  return 1. / largest_eigenvalue(torch.mm(flat, flat.t()))


def fista_betas(num_iters):
  """Momentum weights beta_1..beta_T as Python float64 numbers.

  t_1 = 1, t_{k+1} = (1 + sqrt(1 + 4 t_k^2)) / 2, beta_k = (t_k - 1)/t_{k+1}
  (vtc/analysis_transforms/fully_connected/ista_fista.py:123-125).  The first
  weight is exactly 0.
  """
  betas = []
  t_now = 1.
  for _ in range(num_iters):
    t_next = (1 + (1 + (4 * t_now**2))**0.5) / 2
    betas.append((t_now - 1) / t_next)
    t_now = t_next
  return betas


def shrink_(codes, cutoff, nonnegative_only=False, hard_threshold=False):
  """In-place thresholding with the four flavours of
  vtc/analysis_transforms/fully_connected/ista_fista.py:107-120.

  cutoff is the 0-d tensor lambda*eta.
  """
  if hard_threshold:
    if nonnegative_only:
      codes[codes < cutoff] = 0
    else:
      codes[codes.abs() < cutoff] = 0
  elif nonnegative_only:
    codes.sub_(cutoff).clamp_(min=0.)
  else:
    sgn = torch.sign(codes)
    codes.abs_().sub_(cutoff).clamp_(min=0.).mul_(sgn)
  return codes


def _early_stop_reached(delta, stepsize, epsilon, iter_idx):
  """mean(|delta| / eta) < eps and not the first iteration
  (vtc/analysis_transforms/fully_connected/ista_fista.py:135-144)."""
  return bool(torch.mean(delta.abs() / stepsize) < epsilon) and iter_idx > 0


def _momentum_loop(gradient_step, prox, start_point, num_iters, variant,
                   stepsize, early_stopping_epsilon, trace_at):
  """The iteration skeleton shared by the three inference plugins.

  gradient_step(y) -> y - eta * grad f(y);  prox(c) thresholds c in place.
  Returns (codes, {k: codes after k iterations for k in trace_at}).
  """
  if variant not in ('ista', 'fista'):
    raise AssertionError('variant must be ista or fista')
  if num_iters < 1:
    # the reference would hit an unbound local
    raise UnboundLocalError('num_iters must be >= 1')
  y = start_point
  need_previous = (variant == 'fista') or (early_stopping_epsilon is not None)
  previous = start_point.clone() if need_previous else None
  betas = fista_betas(num_iters) if variant == 'fista' else None
  trace = {}
  codes = None
  for k in range(num_iters):
    codes = prox(gradient_step(y))
    if variant == 'fista':
      delta = codes - previous
      y = codes + betas[k] * delta
      previous.copy_(codes)
    else:
      y = codes
      if early_stopping_epsilon is not None:
        delta = codes - previous
        previous.copy_(codes)
    if trace_at and (k + 1) in trace_at:
      trace[k + 1] = codes.clone()
    if (early_stopping_epsilon is not None and
        _early_stop_reached(delta, stepsize, early_stopping_epsilon, k)):
      break
  return codes, trace


# --------------------------------------------------------------------------
# a1: fully-connected ISTA / FISTA
# --------------------------------------------------------------------------
def fc_ista_fista(images, dictionary, sparsity_weight, num_iters,
                  variant='fista', initial_codes=None,
                  early_stopping_epsilon=None, nonnegative_only=False,
                  hard_threshold=False, stepsize=None, trace_at=None):
  """Restates vtc/analysis_transforms/fully_connected/ista_fista.py:14-148.

  images (b, n), dictionary (s, n) -> codes (b, s).  `stepsize` lets a test
  inject the eta of a golden vector; None computes it as the reference does.
  With trace_at (a collection of iteration counts) returns (codes, trace).
  """
  eta = fc_stepsize(dictionary) if stepsize is None else stepsize
  if not torch.is_tensor(eta):
    eta = torch.tensor(eta, dtype=images.dtype)
  cutoff = sparsity_weight * eta
  atoms_t = dictionary.t()

  def gradient_step(y):
    return y - eta * torch.mm(torch.mm(y, dictionary) - images, atoms_t)

  def prox(c):
    return shrink_(c, cutoff, nonnegative_only, hard_threshold)

  if initial_codes is None:
    start = images.new_zeros(images.shape[0], dictionary.shape[0])
  else:
    start = initial_codes
  codes, trace = _momentum_loop(gradient_step, prox, start, num_iters,
                                variant, eta, early_stopping_epsilon, trace_at)
  return (codes, trace) if trace_at else codes


# --------------------------------------------------------------------------
# a3: subspace (group-LASSO) ISTA / FISTA
# --------------------------------------------------------------------------
def group_layout(group_assignments, num_atoms):
  """Padded (G, m) index layout of a ragged, possibly overlapping grouping.

  Returns (gather_index, valid) with gather_index (G*m,) int64 holding, for
  slot g*m+j, the dictionary row of the j-th member of group g (0 for padding)
  and valid (G*m,) bool marking real members.  This is the structure the
  reference builds with Python loops at
  vtc/analysis_transforms/fully_connected/subspace_ista_fista.py:94-111.
  """
  sizes = [len(g) for g in group_assignments]
  m = max(sizes)
  num_groups = len(group_assignments)
  gather_index = torch.zeros(num_groups * m, dtype=torch.int64)
  valid = torch.zeros(num_groups * m, dtype=torch.bool)
  for g, members in enumerate(group_assignments):
    members = torch.as_tensor([int(a) for a in members], dtype=torch.int64)
    assert int(members.max()) < num_atoms
    gather_index[g * m: g * m + len(members)] = members
    valid[g * m: g * m + len(members)] = True
  return gather_index, valid, num_groups, m


def grouped_dictionary(dictionary, group_assignments):
  """(G*m, n) dictionary with duplicated rows for shared atoms and zero rows
  for padding (vtc/.../subspace_ista_fista.py:106-111)."""
  gather_index, valid, _, _ = group_layout(group_assignments,
                                           dictionary.shape[0])
  return dictionary[gather_index] * valid[:, None].to(dictionary.dtype)


def subspace_ista_fista(images, dictionary, group_assignments, sparsity_weight,
                        num_iters, variant='fista',
                        ret_summed_gduplicates=True, initial_codes=None,
                        early_stopping_epsilon=None, hard_threshold=False,
                        stepsize=None, trace_at=None):
  """Restates vtc/analysis_transforms/fully_connected/subspace_ista_fista.py
  :23-192: proximal step shrinks each group's l2 norm by lambda*eta."""
  if hard_threshold:
    raise NotImplementedError('TODO')  # vtc/.../subspace_ista_fista.py:152-153
  if not ret_summed_gduplicates:
    raise NotImplementedError('TODO')  # vtc/.../subspace_ista_fista.py:191-192
  num_atoms = dictionary.shape[0]
  gather_index, valid, num_groups, m = group_layout(group_assignments,
                                                    num_atoms)
  dict_g = dictionary[gather_index] * valid[:, None].to(dictionary.dtype)
  # Lipschitz bound from the *grouped* dictionary (:115-123)
  if stepsize is None:
    eta = 1. / largest_eigenvalue(torch.mm(dict_g.t(), dict_g))
  else:
    eta = stepsize if torch.is_tensor(stepsize) else torch.tensor(
        stepsize, dtype=images.dtype)
  batch = images.shape[0]
  dict_g_t = dict_g.t()

  def gradient_step(y):
    flat = y.reshape(batch, -1)
    g = torch.mm(torch.mm(flat, dict_g) - images, dict_g_t)
    return y - eta * g.reshape(y.shape)

  def prox(c):
    norms = torch.norm(c, p=2, dim=2, keepdim=True)
    norms[norms == 0] = 1.0
    return c.mul_(torch.clamp(1 - (sparsity_weight * eta / norms), min=0.))

  start = images.new_zeros(batch, num_groups, m)
  if initial_codes is not None:
    warm = initial_codes[:, gather_index] * valid[None, :].to(images.dtype)
    start = warm.reshape(batch, num_groups, m).clone()
  grouped, trace_g = _momentum_loop(gradient_step, prox, start, num_iters,
                                    variant, eta, early_stopping_epsilon,
                                    trace_at)

  def scatter_sum(gc):
    # atoms living in several groups receive the sum of their copies
    # (:184-190).  index_add_ visits slots in increasing order, the same
    # order in which the reference's group loop accumulates them.
    out = images.new_zeros(batch, num_atoms)
    flat = gc.reshape(batch, -1)
    keep = valid.nonzero().squeeze(1)
    out.index_add_(1, gather_index[keep], flat[:, keep])
    return out

  codes = scatter_sum(grouped)
  if trace_at:
    return codes, {k: scatter_sum(v) for k, v in trace_g.items()}
  return codes


# --------------------------------------------------------------------------
# a4: convolutional geometry + ISTA / FISTA
# --------------------------------------------------------------------------
def conv_padding_amount(image_dim, kernel_dim, dim_stride):
  """vtc/utils/convolutions.py:7-12."""
  lead = kernel_dim - dim_stride
  trail = kernel_dim - dim_stride
  if image_dim % dim_stride != 0:
    trail += dim_stride - (image_dim % dim_stride)
  return lead, trail


def conv_code_dim(padded_image_dim, kernel_dim, dim_stride):
  """vtc/utils/convolutions.py:14-15."""
  return 1 + int(math.ceil((padded_image_dim - kernel_dim) / dim_stride))


def conv_mask(images_padded, padding_dims):
  """1 inside the un-padded image, 0 on the padding frame
  (vtc/utils/convolutions.py:17-24).  A trailing pad of 0 blanks the whole
  axis exactly like the reference's `-0:` slice does."""
  mask = torch.ones_like(images_padded)
  if padding_dims is None:
    return mask
  (lead_v, trail_v), (lead_h, trail_h) = padding_dims
  height, width = mask.shape[2], mask.shape[3]
  mask[:, :, :lead_v, :] = 0.
  mask[:, :, (height - trail_v) if trail_v != 0 else 0:, :] = 0.
  mask[:, :, :, :lead_h] = 0.
  mask[:, :, :, (width - trail_h) if trail_h != 0 else 0:] = 0.
  return mask


def conv_synthesis(codes, dictionary, kernel_stride):
  """recon[b,c,y,x] = sum_{s,p,q,dy,dx : p*sv+dy=y, q*sh+dx=x}
  codes[b,s,p,q] * D[s,c,dy,dx]   (conv_transpose2d, no kernel flip)."""
  return torch.nn.functional.conv_transpose2d(codes, dictionary,
                                              stride=kernel_stride)


def conv_analysis(residual, dictionary, kernel_stride):
  """g[b,s,p,q] = sum_{c,dy,dx} residual[b,c,p*sv+dy,q*sh+dx] * D[s,c,dy,dx]
  (conv2d = cross-correlation)."""
  return torch.nn.functional.conv2d(residual, dictionary,
                                    stride=kernel_stride)


def conv_synthesis_naive(codes, dictionary, kernel_stride):
  """Loop form of conv_synthesis, used by the tests to pin the index
  convention of the torch call above on tiny shapes."""
  b, s, ch, cw = codes.shape
  _, c, kh, kw = dictionary.shape
  sv, sh = kernel_stride
  out = codes.new_zeros(b, c, (ch - 1) * sv + kh, (cw - 1) * sh + kw)
  for p in range(ch):
    for q in range(cw):
      patch = torch.einsum('bs,scyx->bcyx', codes[:, :, p, q], dictionary)
      out[:, :, p * sv: p * sv + kh, q * sh: q * sh + kw] += patch
  return out


def conv_analysis_naive(residual, dictionary, kernel_stride):
  """Loop form of conv_analysis (tiny shapes only)."""
  b, c, height, width = residual.shape
  s, _, kh, kw = dictionary.shape
  sv, sh = kernel_stride
  ch = (height - kh) // sv + 1
  cw = (width - kw) // sh + 1
  out = residual.new_zeros(b, s, ch, cw)
  for p in range(ch):
    for q in range(cw):
      window = residual[:, :, p * sv: p * sv + kh, q * sh: q * sh + kw]
      out[:, :, p, q] = torch.einsum('bcyx,scyx->bs', window, dictionary)
  return out


def conv_ista_fista(images_padded, dictionary, kernel_stride, padding_dims,
                    sparsity_weight, num_iters, variant='fista',
                    initial_codes=None, early_stopping_epsilon=None,
                    nonnegative_only=False, hard_threshold=False,
                    stepsize=None, trace_at=None):
  """Restates vtc/analysis_transforms/convolutional/ista_fista.py:18-197."""
  eta = conv_stepsize(dictionary) if stepsize is None else stepsize
  if not torch.is_tensor(eta):
    eta = torch.tensor(eta, dtype=images_padded.dtype)
  cutoff = sparsity_weight * eta
  code_h = conv_code_dim(images_padded.shape[2], dictionary.shape[2],
                         kernel_stride[0])
  code_w = conv_code_dim(images_padded.shape[3], dictionary.shape[3],
                         kernel_stride[1])
  shape = (images_padded.shape[0], dictionary.shape[0], code_h, code_w)
  if initial_codes is None:
    start = images_padded.new_zeros(shape)
  else:
    assert tuple(initial_codes.shape) == shape
    start = initial_codes
  mask = conv_mask(images_padded, padding_dims)

  def gradient_step(y):
    residual = mask * (conv_synthesis(y, dictionary, kernel_stride) -
                       images_padded)
    return y - eta * conv_analysis(residual, dictionary, kernel_stride)

  def prox(c):
    return shrink_(c, cutoff, nonnegative_only, hard_threshold)

  codes, trace = _momentum_loop(gradient_step, prox, start, num_iters,
                                variant, eta, early_stopping_epsilon, trace_at)
  return (codes, trace) if trace_at else codes


# --------------------------------------------------------------------------
# a5 / a6 / a7: fully-connected dictionary updates (in place, return None)
# --------------------------------------------------------------------------
def _normalize_rows_(dictionary):
  dictionary.div_(dictionary.norm(p=2, dim=1)[:, None])


def fc_gradient(images, dictionary, codes):
  """C^T (C D - X) / b  (vtc/dict_update_rules/fully_connected/
  sc_steepest_descent.py:38-39): divide by the batch first."""
  return torch.mm(codes.t(), torch.mm(codes, dictionary) - images) / (
      codes.shape[0])


def fc_steepest_descent(images, dictionary, codes, stepsize=0.001,
                        num_iters=1, normalize_dictionary=True):
  """vtc/dict_update_rules/fully_connected/sc_steepest_descent.py:9-41."""
  for _ in range(num_iters):
    dictionary.sub_(stepsize * fc_gradient(images, dictionary, codes))
    if normalize_dictionary:
      _normalize_rows_(dictionary)


def fc_cheap_quadratic_descent(images, dictionary, codes, hessian_diagonal,
                               stepsize=0.001, num_iters=1,
                               lowest_code_val=0.001,
                               normalize_dictionary=True):
  """vtc/dict_update_rules/fully_connected/sc_cheap_quadratic_descent.py
  :11-48: the scaled gradient of every atom is divided by (h + 0.001)."""
  for _ in range(num_iters):
    step = stepsize * fc_gradient(images, dictionary, codes)
    step.div_(hessian_diagonal[:, None] + lowest_code_val)
    dictionary.sub_(step)
    if normalize_dictionary:
      _normalize_rows_(dictionary)


def alignment_gradients(group_atoms, dict_is_normalized):
  """Gradient of sum_{i,j} |cos(d_i, d_j)| inside one group, rows = atoms.
  Restates regularization_gradients,
  vtc/dict_update_rules/fully_connected/subspace_sc_cheap_quadratic_descent.py
  :91-127:  grad_i = sum_j sign(cos_ij) (a1_ij - a0_ij)."""
  m = group_atoms.shape[0]
  own = group_atoms[:, None, :].expand(m, m, -1)    # d_i along axis 1
  other = group_atoms[None, :, :].expand(m, m, -1)  # d_j along axis 0
  if dict_is_normalized:
    cos = torch.mm(group_atoms, group_atoms.t())[:, :, None]
    toward_self = cos * own
    toward_other = other
  else:
    norms = torch.norm(group_atoms, p=2, dim=1, keepdim=True)
    outer = torch.mm(norms, norms.t())
    cos = (torch.mm(group_atoms, group_atoms.t()) / outer)[:, :, None]
    toward_self = (cos / (norms**2)[:, None]) * own
    toward_other = other / outer[:, :, None]
  return torch.sum(torch.sign(cos) * (toward_other - toward_self), dim=1)


def subspace_cheap_quadratic_descent(images, dictionary, codes,
                                     group_assignments, hessian_diagonal,
                                     alignment_penalty, stepsize=0.001,
                                     num_iters=1, lowest_code_val=0.001,
                                     normalize_dictionary=True):
  """vtc/dict_update_rules/fully_connected/
  subspace_sc_cheap_quadratic_descent.py:13-88."""
  if alignment_penalty == 0:
    return fc_cheap_quadratic_descent(
        images, dictionary, codes, hessian_diagonal, stepsize, num_iters,
        lowest_code_val, normalize_dictionary)
  for _ in range(num_iters):
    penalty_grad = torch.zeros_like(dictionary)
    for members in group_assignments:
      members = [int(a) for a in members]
      penalty_grad[members] = penalty_grad[members] + alignment_gradients(
          dictionary[members], normalize_dictionary)
    step = stepsize * (fc_gradient(images, dictionary, codes) +
                       alignment_penalty * penalty_grad)
    step.div_(hessian_diagonal[:, None] + lowest_code_val)
    dictionary.sub_(step)
    if normalize_dictionary:
      _normalize_rows_(dictionary)


# --------------------------------------------------------------------------
# a8 / a9: convolutional dictionary updates
# --------------------------------------------------------------------------
def conv_gradient(images_padded, dictionary, codes, kernel_stride,
                  padding_dims):
  """dD[s,c,dy,dx] = sum_{b,p,q} codes[b,s,p,q] r[b,c,p*sv+dy,q*sh+dx] / b
  with r = mask * (synthesis - images)
  (vtc/dict_update_rules/convolutional/sc_steepest_descent.py:54-65: a conv2d
  with batch and channel axes swapped and dilation = stride)."""
  mask = conv_mask(images_padded, padding_dims)
  residual = mask * (conv_synthesis(codes, dictionary, kernel_stride) -
                     images_padded)
  grad = torch.nn.functional.conv2d(residual.transpose(0, 1),
                                    codes.transpose(0, 1),
                                    dilation=kernel_stride)
  return (grad / images_padded.shape[0]).transpose(0, 1)


def conv_gradient_naive(images_padded, dictionary, codes, kernel_stride,
                        padding_dims):
  """Loop form of conv_gradient for tiny shapes (pins the index convention)."""
  mask = conv_mask(images_padded, padding_dims)
  residual = mask * (conv_synthesis_naive(codes, dictionary, kernel_stride) -
                     images_padded)
  s, c, kh, kw = dictionary.shape
  sv, sh = kernel_stride
  _, _, ch, cw = codes.shape
  grad = torch.zeros_like(dictionary)
  for dy in range(kh):
    for dx in range(kw):
      window = residual[:, :, dy: dy + (ch - 1) * sv + 1: sv,
                        dx: dx + (cw - 1) * sh + 1: sh]
      grad[:, :, dy, dx] = torch.einsum('bspq,bcpq->sc', codes, window)
  return grad / images_padded.shape[0]


def _normalize_kernels_(dictionary):
  dictionary.div_(torch.squeeze(dictionary.norm(p=2, dim=(1, 2, 3)))[
      :, None, None, None])


def conv_steepest_descent(images_padded, dictionary, codes, kernel_stride,
                          padding_dims, stepsize=0.001, num_iters=1,
                          normalize_dictionary=True):
  """vtc/dict_update_rules/convolutional/sc_steepest_descent.py:12-72."""
  for _ in range(num_iters):
    grad = conv_gradient(images_padded, dictionary, codes, kernel_stride,
                         padding_dims).contiguous()
    grad.mul_(dictionary.norm(p=2) / grad.norm(p=2))   # global rescale (:68)
    dictionary.sub_(stepsize * grad)
    if normalize_dictionary:
      _normalize_kernels_(dictionary)


def conv_cheap_quadratic_descent(images_padded, dictionary, codes,
                                 hessian_diagonal, kernel_stride, padding_dims,
                                 stepsize=0.001, num_iters=1,
                                 lowest_code_val=0.001,
                                 normalize_dictionary=True):
  """vtc/dict_update_rules/convolutional/sc_cheap_quadratic_descent.py:14-79:
  the Hessian divide comes before the global rescale (:72,75)."""
  for _ in range(num_iters):
    grad = conv_gradient(images_padded, dictionary, codes, kernel_stride,
                         padding_dims).contiguous()
    grad.div_(hessian_diagonal[:, None, None, None] + lowest_code_val)
    grad.mul_(dictionary.norm(p=2) / grad.norm(p=2))
    dictionary.sub_(stepsize * grad)
    if normalize_dictionary:
      _normalize_kernels_(dictionary)


# --------------------------------------------------------------------------
# a6 / a9 / a10: the trainer's per-batch step
# --------------------------------------------------------------------------
def hessian_diag_ema_(hessian_diagonal, codes):
  """h <- 0.99 h + mean_b(sum_positions codes^2) / 100
  (vtc/training/sparse_coding.py:154 fully-connected, :160-161 conv)."""
  if codes.dim() == 2:
    hessian_diagonal.mul_(0.99).add_(torch.pow(codes, 2).mean(0) / 100)
  else:
    hessian_diagonal.mul_(0.99).add_(
        torch.mean(torch.sum(codes**2, dim=(2, 3)), dim=0) / 100)
  return hessian_diagonal


def ica_natural_gradient(dictionary, codes, stepsize=0.001, num_iters=1):
  """Restates vtc/dict_update_rules/fully_connected/ica_natural_gradient.py
  :26-35.  Updates `dictionary` in place (gradient ASCENT)."""
  eye_mat = torch.eye(codes.size(1), dtype=codes.dtype)
  for _ in range(num_iters):
    dict_update = stepsize * torch.mm(
        (torch.mm(codes.t(), torch.sign(codes)) / codes.size(0)) - eye_mat,
        dictionary)
    dictionary.add_(dict_update)
  return dictionary


def compute_metrics(batch_images, batch_codes, dictionary,
                    previous_dictionary, sparsity_weight, params):
  """Restates the `compute_metrics` closure of
  vtc/training/sparse_coding.py:177-229 (numpy on the host, as there)."""
  mode = params['mode']
  inf_alg = params['code_inference_algorithm']
  metrics = {}
  images_np = batch_images.numpy()
  if mode == 'fully-connected':
    recons = torch.mm(batch_codes, dictionary).numpy()
    axes = 1
  else:
    recons = torch.nn.functional.conv_transpose2d(
        batch_codes, dictionary, stride=params['strides']).numpy()
    pad = params['padding']
    if pad is not None:                                        # :188-194
      recons = recons[:, :, pad[0][0]:-pad[0][1], pad[1][0]:-pad[1][1]]
      images_np = images_np[:, :, pad[0][0]:-pad[0][1], pad[1][0]:-pad[1][1]]
    axes = (1, 2, 3)
  metrics['Average LASSO L2 component'] = np.mean(
      0.5 * np.sum(np.square(recons - images_np), axis=axes))
  if inf_alg in ('subspace_ista', 'subspace_fista'):           # :199-206
    groups = params['group_assignments']
    sum_of_group_norms = np.zeros((len(batch_codes),))
    for g in groups:
      sum_of_group_norms += torch.norm(batch_codes[:, g], p=2, dim=1).numpy()
    metrics['Average LASSO lagrange component'] = np.mean(
        sparsity_weight * sum_of_group_norms)
  else:
    metrics['Average LASSO lagrange component'] = np.mean(
        sparsity_weight * torch.norm(batch_codes, p=1, dim=axes).numpy())
  metrics['Average LASSO Loss'] = (
      metrics['Average LASSO L2 component'] +
      metrics['Average LASSO lagrange component'])
  metrics['Average Normalized L0'] = float(torch.mean(
      torch.norm(batch_codes, p=0, dim=axes) /
      np.prod(batch_codes.shape[1:])).numpy())
  sig_mag = np.max(images_np) - np.min(images_np)              # :218
  psnrs = []
  for i in range(recons.shape[0]):                             # plotting.py:35-39
    mse = np.mean(np.square(images_np[i] - recons[i]))
    if mse != 0:
      psnrs.append(10. * np.log10((sig_mag ** 2) / mse))
  metrics['Average pSNR of reconstructions'] = np.mean(psnrs)
  metrics['Average change in dictionary kernels'] = torch.mean(
      torch.abs(dictionary - previous_dictionary), dim=axes).numpy()
  return metrics


def train_steps(batches, dictionary, params, validation_batches=None):
  """Runs the per-batch step of vtc/training/sparse_coding.py:444-517 over a
  list of batches, mutating `dictionary` in place.  Returns a list with one
  dict per step: {'codes', 'dictionary' (copy after the update), 'hessian',
  'validation'}.  'validation' is the mean of compute_metrics over
  `validation_batches` taken BEFORE the step, at the iterations listed in
  params['training_visualization_schedule'] (:497-505), else None.

  params: mode, code_inference_algorithm, inference_param_schedule,
  dictionary_update_algorithm, dict_update_param_schedule and the optional
  nonnegative_only / hard_threshold / group_assignments /
  subspace_alignment_penalty / strides / padding keys of the reference's
  all_params dict (vtc/training/sparse_coding.py:52-117).
  """
  mode = params['mode']
  inf_alg = params['code_inference_algorithm']
  upd_alg = params['dictionary_update_algorithm']
  inf_sched = params['inference_param_schedule']
  upd_sched = params['dict_update_param_schedule']
  groups = params.get('group_assignments')
  uses_hessian = upd_alg in ('sc_cheap_quadratic_descent',
                             'subspace_sc_cheap_quadratic_descent')
  hessian = dictionary.new_zeros(dictionary.shape[0]) if uses_hessian else None
  history = []
  vis_sched = params.get('training_visualization_schedule')
  previous_dictionary = dictionary.clone()

  def infer(batch):
    # ---- inference (vtc/training/sparse_coding.py:124-140)
    if inf_alg in ('subspace_ista', 'subspace_fista'):
      return subspace_ista_fista(
          batch, dictionary, groups, lam, inf_iters, variant=inf_alg[9:],
          hard_threshold=params.get('hard_threshold', False))
    if mode == 'fully-connected':
      return fc_ista_fista(
          batch, dictionary, lam, inf_iters, variant=inf_alg,
          nonnegative_only=params.get('nonnegative_only', False),
          hard_threshold=params.get('hard_threshold', False))
    return conv_ista_fista(
        batch, dictionary, params['strides'], params['padding'], lam,
        inf_iters, variant=inf_alg,
        nonnegative_only=params.get('nonnegative_only', False),
        hard_threshold=params.get('hard_threshold', False))

  for step_idx, batch in enumerate(batches):
    if step_idx in inf_sched:
      lam = inf_sched[step_idx]['sparsity_weight']
      inf_iters = inf_sched[step_idx]['num_iters']
    if step_idx in upd_sched:
      upd_step = upd_sched[step_idx]['stepsize']
      upd_iters = upd_sched[step_idx]['num_iters']
    validation = None
    if vis_sched is not None and step_idx in vis_sched:
      per_batch = [compute_metrics(v, infer(v), dictionary,
                                   previous_dictionary, lam, params)
                   for v in validation_batches]
      validation = {x: np.mean([m[x] for m in per_batch])
                    for x in per_batch[0]}
    codes = infer(batch)
    previous_dictionary.copy_(dictionary)                      # :514
    # ---- dictionary update (vtc/training/sparse_coding.py:142-168)
    if uses_hessian:
      hessian_diag_ema_(hessian, codes)
    if mode == 'fully-connected':
      if upd_alg == 'sc_steepest_descent':
        fc_steepest_descent(batch, dictionary, codes, upd_step, upd_iters)
      elif upd_alg == 'sc_cheap_quadratic_descent':
        fc_cheap_quadratic_descent(batch, dictionary, codes, hessian,
                                   upd_step, upd_iters)
      elif upd_alg == 'subspace_sc_cheap_quadratic_descent':
        subspace_cheap_quadratic_descent(
            batch, dictionary, codes, groups, hessian,
            params['subspace_alignment_penalty'], upd_step, upd_iters)
      else:
        raise KeyError('Unrecognized dict update algorithm: ' + upd_alg)
    else:
      if upd_alg == 'sc_steepest_descent':
        conv_steepest_descent(batch, dictionary, codes, params['strides'],
                              params['padding'], upd_step, upd_iters)
      elif upd_alg == 'sc_cheap_quadratic_descent':
        conv_cheap_quadratic_descent(batch, dictionary, codes, hessian,
                                     params['strides'], params['padding'],
                                     upd_step, upd_iters)
      else:
        raise KeyError('Unrecognized dict update algorithm: ' + upd_alg)
    history.append({'codes': codes, 'dictionary': dictionary.clone(),
                    'hessian': None if hessian is None else hessian.clone(),
                    'validation': validation})
  return history


# --------------------------------------------------------------------------
# f4: dictionary reset / prune, non-interactive modes
# --------------------------------------------------------------------------
def reset_or_prune(dictionary, codes, filter_type, filter_params, action):
  """vtc/training/sparse_coding.py:522-764 on CPU tensors (cue_user excluded).
  Randomness as in the reference: numpy's global generator for every choice,
  torch's CPU generator for the replacement atoms.  Returns (dictionary,
  affected atoms); pruning edits filter_params['group_assignments'] in place."""
  groups = filter_params['group_assignments']
  assert filter_params['coding_mode'] == 'fully-connected'

  def fresh(count, like_rows):
    target = torch.mean(like_rows.norm(p=2, dim=1))
    noise = torch.randn((count, dictionary.shape[1]))
    return noise * (target / noise.norm(p=2, dim=1)[:, None])

  def cosines(rows):
    nrm = torch.norm(rows, p=2, dim=1, keepdim=True)
    return (torch.mm(rows, rows.t()) / torch.mm(nrm, nrm.t())).numpy()

  def pick(pairs):
    out = []
    for pr in pairs:
      if pr[0] not in out and pr[1] not in out:
        out.append(pr[np.random.choice([0, 1])])
    return out

  def prune(rows):
    keep = torch.ones(dictionary.shape[0], dtype=torch.bool)
    keep[torch.as_tensor(np.asarray(rows, dtype=np.int64))] = False
    if groups is not None:
      for g in range(len(groups)):
        groups[g] = [a for a in groups[g] if a not in rows]
    return dictionary[keep]

  if filter_type == 'random':
    rows = np.random.choice(np.arange(dictionary.shape[0]),
                            filter_params['num_to_modify'])
    if action == 'reset':
      dictionary[rows] = fresh(len(rows), dictionary)
      return dictionary, rows
    return prune(rows), rows
  if filter_type == 'cosine_sim_threshold':
    assert not filter_params['cue_user']
    thr = filter_params['threshold']
    if filter_params['only_sim_within_group']:
      hit = []
      for g in range(len(groups)):
        members = np.array(groups[g])
        sims = cosines(dictionary[groups[g]])
        local = pick(np.argwhere(np.abs(np.triu(sims, k=1)) > thr))
        if len(local) > 0:
          if action == 'reset':
            dictionary[members[local]] = fresh(len(local),
                                               dictionary[groups[g]])
          hit.append(members[local])
      rows = np.array(hit).flatten()
      if action == 'prune' and len(rows) > 0:
        return prune(rows), rows
      return dictionary, rows
    rows = np.array(pick(np.argwhere(np.triu(cosines(dictionary), k=1) > thr)))
    if len(rows) > 0:
      if action == 'reset':
        dictionary[rows] = fresh(len(rows), dictionary)
      else:
        return prune(rows), rows
    return dictionary, rows
  if filter_type == 'nonuniformity_within_group':
    spread = []
    for g in range(len(groups)):
      block = codes[:, groups[g]]
      block = block[torch.sum(block != 0, dim=1) != 0]
      unit = (block / torch.norm(block, p=2, dim=1, keepdim=True)).numpy()
      var = []
      for _ in range(filter_params['num_gc_in_average']):
        u = np.random.randn(len(groups[g]))
        u /= np.linalg.norm(u)
        v = np.random.randn(len(groups[g]))
        v /= np.linalg.norm(v)
        basis, _ = np.linalg.qr(np.c_[u, v])
        pr = np.dot(unit, basis)
        ang = np.angle(pr[:, 0] + 1j * pr[:, 1])
        cnt, _ = np.histogram(ang, np.linspace(-np.pi, np.pi, 21))
        var.append(np.var(cnt / np.sum(cnt)))
      spread.append(np.mean(var))
    spread = np.array(spread)
    odd = np.nonzero(np.logical_and(
        np.abs(spread - np.mean(spread)) > np.std(spread),
        np.abs(spread) > 0.002))[0]
    rows = np.array([groups[x] for x in odd]).flatten()
    if len(rows) > 0:
      if action == 'reset':
        dictionary[rows] = fresh(len(rows), dictionary)
      else:
        return prune(rows), rows
    return dictionary, rows
  raise KeyError('Unrecognized reset type')
