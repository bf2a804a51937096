"""
Host-side binding of libvtc_hip.so (the C ABI declared in include/vtc_hip.h).

PyTorch is used here for what it is good at on ROCm -- device memory, the
current HIP stream, torch.distributed -- and nothing else: every arithmetic
step of the plugins goes through the C entry points below, on raw device
pointers.  There is deliberately no CPU or eager-PyTorch fallback: if the
library is missing or a tensor is not on the GPU the call raises.  (One
exception, stated where it lives: Gram matrices beyond 1024 x 1024 -- none of
the path's configurations -- take torch.linalg.eigvalsh for the Lipschitz
constant, see stepsize_from_gram.)
"""
import ctypes
import os
import pathlib

import torch

_PKG_ROOT = pathlib.Path(__file__).resolve().parent.parent
LIBRARY_PATH = _PKG_ROOT / 'libvtc_hip.so'

OK, ERR_INVALID_ARGUMENT, ERR_UNSUPPORTED, ERR_WORKSPACE, ERR_HIP = range(5)
ISTA, FISTA = 0, 1
SOFT, SOFT_NONNEG, HARD, HARD_NONNEG = range(4)
F32, BF16X3, BF16, F16X3 = range(4)
PRECISIONS = {'f32': F32, 'bf16x3': BF16X3, 'bf16': BF16, 'f16x3': F16X3}
ABI_VERSION = 4   # VTC_ABI_VERSION of include/vtc_hip.h this binding matches

_lib = None

_c_f32p = ctypes.c_void_p
_i64 = ctypes.c_int64
_i32 = ctypes.c_int
_f32 = ctypes.c_float
_vp = ctypes.c_void_p
_sz = ctypes.c_size_t


class ConvGeometry(ctypes.Structure):
  """struct vtc_conv_geometry of include/vtc_hip.h."""
  _fields_ = [('b', ctypes.c_int64),
              ('c', ctypes.c_int32), ('h', ctypes.c_int32),
              ('w', ctypes.c_int32),
              ('s', ctypes.c_int32), ('kh', ctypes.c_int32),
              ('kw', ctypes.c_int32),
              ('stride_v', ctypes.c_int32), ('stride_h', ctypes.c_int32),
              ('has_padding', ctypes.c_int32),
              ('pad_lead_v', ctypes.c_int32), ('pad_trail_v', ctypes.c_int32),
              ('pad_lead_h', ctypes.c_int32), ('pad_trail_h', ctypes.c_int32)]


_GEOM_P = ctypes.POINTER(ConvGeometry)

# name -> (restype, argtypes); kept in one table so that the CPU-only test can
# check that the library exports every symbol the header declares.
SIGNATURES = {
    'vtc_version': (ctypes.c_char_p, []),
    'vtc_last_error': (ctypes.c_char_p, []),
    'vtc_abi_version': (_i32, []),
    'vtc_init': (_i32, []),
    'vtc_gram': (_i32, [_vp, _i64, _i64, _i32, _vp, _vp]),
    'vtc_lambda_max_workspace_bytes': (_sz, [_i64]),
    'vtc_lambda_max': (_i32, [_vp, _i64, _vp, _vp, _sz, _vp]),
    'vtc_lambda_max_mirrored': (_i32, [_vp, _i64, _vp, _vp, _vp, _sz, _vp]),
    'vtc_fc_ista_fista_workspace_bytes': (_sz, [_i64, _i64, _i64, _i32]),
    'vtc_fc_ista_fista': (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _f32,
                                 _f32, _i32, _i32, _i32, _f32, _i32, _vp, _sz,
                                 ctypes.POINTER(_i32), _vp]),
    'vtc_fc_ista_fista_dev': (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp,
                                     _f32, _i32, _i32, _i32, _f32, _i32, _vp,
                                     _sz, ctypes.POINTER(_i32), _vp]),
    'vtc_group_gather_rows': (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _vp]),
    'vtc_group_gather_cols': (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _i64,
                                     _vp]),
    'vtc_group_scatter_add': (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _i64,
                                     _vp]),
    'vtc_subspace_ista_fista_workspace_bytes': (_sz, [_i64, _i64, _i64, _i64]),
    'vtc_subspace_ista_fista': (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _i64,
                                       _i64, _f32, _f32, _i32, _i32, _f32,
                                       _i32, _vp, _sz, ctypes.POINTER(_i32),
                                       _vp]),
    'vtc_conv_code_dims': (_i32, [_GEOM_P, ctypes.POINTER(ctypes.c_int32),
                                  ctypes.POINTER(ctypes.c_int32)]),
    'vtc_conv_ista_fista_workspace_bytes': (_sz, [_GEOM_P]),
    'vtc_conv_x3_supported': (_i32, [_GEOM_P]),
    'vtc_fc_residual': (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp]),
    'vtc_conv_residual': (_i32, [_vp, _vp, _vp, _vp, _GEOM_P, _vp]),
    'vtc_row_stats': (_i32, [_vp, _i64, _i64, _vp, _vp, _vp, _vp]),
    'vtc_group_norm_sum': (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64,
                                  _vp]),
    'vtc_window_minmax_workspace_bytes': (_sz, []),
    'vtc_window_minmax': (_i32, [_vp, _i64, _i64, _i64, _i64, _i64, _vp, _vp,
                                 _sz, _vp]),
    'vtc_rows_mean_abs_diff': (_i32, [_vp, _vp, _i64, _i64, _vp, _vp]),
    'vtc_whiten_center_surround_workspace_bytes': (_sz, [_i64, _i32, _i32,
                                                         _i32]),
    'vtc_whiten_center_surround': (_i32, [_vp, _vp, _i64, _i32, _i32, _i32,
                                          _f32, _f32, _i32, _vp, _sz, _vp]),
    'vtc_extract_patches': (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32,
                                   _i32, _i32, _i32, _vp]),
    'vtc_standardize_data_range': (_i32, [_vp, _vp, _i64, _vp, _vp, _sz, _vp]),
    'vtc_draw_patch_positions': (_i32, [_vp, _vp, _i64, _i32, _i32, _vp, _vp,
                                        _vp, _vp, _vp]),
    'vtc_ica_moment_workspace_bytes': (_sz, [_i64, _i64]),
    'vtc_ica_moment': (_i32, [_vp, _vp, _i64, _i64, _vp, _sz, _vp]),
    'vtc_ica_apply_workspace_bytes': (_sz, [_i64, _i64]),
    'vtc_ica_apply': (_i32, [_vp, _vp, _i64, _i64, _i64, _f32, _vp, _sz,
                             _vp]),
    'vtc_conv_ista_fista': (_i32, [_vp, _vp, _vp, _vp, _GEOM_P, _f32, _f32,
                                   _i32, _i32, _i32, _f32, _i32, _vp, _sz,
                                   ctypes.POINTER(_i32), _vp]),
    'vtc_fc_dict_gradient_workspace_bytes': (_sz, [_i64, _i64, _i64]),
    'vtc_fc_dict_gradient': (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp,
                                    _sz, _vp]),
    'vtc_subspace_alignment_gradient_workspace_bytes': (_sz, [_i64, _i64]),
    'vtc_subspace_alignment_gradient': (_i32, [_vp, _vp, _vp, _vp, _vp, _vp,
                                               _i64, _i64, _i64, _i64, _i32,
                                               _vp, _sz, _vp]),
    'vtc_fc_dict_apply': (_i32, [_vp, _vp, _vp, _vp, _f32, _i64, _f32, _f32,
                                 _i32, _i64, _i64, _vp]),
    'vtc_conv_dict_gradient_workspace_bytes': (_sz, [_GEOM_P]),
    'vtc_conv_dict_gradient': (_i32, [_vp, _vp, _vp, _vp, _GEOM_P, _i32, _vp,
                                      _sz, _vp]),
    'vtc_conv_dict_apply': (_i32, [_vp, _vp, _vp, _i64, _f32, _f32, _i32, _i64,
                                   _i64, _vp, _vp]),
    'vtc_code_energy_workspace_bytes': (_sz, [_i64, _i64, _i64]),
    'vtc_code_energy': (_i32, [_vp, _i64, _i64, _i64, _vp, _vp, _sz, _vp]),
    'vtc_hessian_ema': (_i32, [_vp, _vp, _i64, _i64, _vp]),
}


class VtcHipError(RuntimeError):
  pass


def load_library():
  """dlopen libvtc_hip.so (built in-tree by __graft_entry__.build() / make)."""
  global _lib
  if _lib is not None:
    return _lib
  if not LIBRARY_PATH.exists():
    raise ImportError(
        'libvtc_hip.so not found at %s -- build it with `make -C %s` '
        '(hipcc --offload-arch=gfx950).  There is no CPU fallback.'
        % (LIBRARY_PATH, _PKG_ROOT / 'csrc'))
  lib = ctypes.CDLL(str(LIBRARY_PATH), mode=os.RTLD_NOW)
  for name, (restype, argtypes) in SIGNATURES.items():
    fn = getattr(lib, name)   # AttributeError if the export is missing
    fn.restype = restype
    fn.argtypes = argtypes
  if lib.vtc_abi_version() != ABI_VERSION:
    raise ImportError('libvtc_hip.so ABI version mismatch')
  _lib = lib
  return lib


_prepared_devices = set()


def prepare_device(device):
  """vtc_init() on `device`, once per process: the library's per-device
  constants (the FISTA momentum table) are placed before the first inference
  call is enqueued there."""
  index = torch.device(device).index
  if index is None:
    index = torch.cuda.current_device()
  if index in _prepared_devices:
    return
  with torch.cuda.device(index):
    check(load_library().vtc_init(), 'vtc_init')
  _prepared_devices.add(index)


def check(status, what):
  """Map a vtc_status to the exception type the reference would raise."""
  if status == OK:
    return
  msg = load_library().vtc_last_error().decode('utf-8', 'replace')
  text = '%s: %s' % (what, msg)
  if status == ERR_UNSUPPORTED:
    raise NotImplementedError(text)
  if status == ERR_INVALID_ARGUMENT:
    raise ValueError(text)
  raise VtcHipError(text)


def require_device_tensor(t, name, dtype=torch.float32):
  if not torch.is_tensor(t):
    raise TypeError('%s must be a torch.Tensor' % name)
  if not t.is_cuda:
    raise VtcHipError(
        '%s lives on %s: the MI355X engine only runs on HIP device tensors '
        '(no CPU path is provided on purpose)' % (name, t.device))
  if t.dtype != dtype:
    raise TypeError('%s must be %s, got %s' % (name, dtype, t.dtype))
  return t


def ptr(t):
  return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def current_stream(device):
  return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def workspace(nbytes, device):
  """Scratch handed to the C side; freed back to the caching allocator when
  the returned tensor dies (stream-ordered, so safe for enqueued work)."""
  return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


def threshold_mode(nonnegative_only, hard_threshold):
  if hard_threshold:
    return HARD_NONNEG if nonnegative_only else HARD
  return SOFT_NONNEG if nonnegative_only else SOFT


def variant_code(variant):
  assert variant in ['ista', 'fista']
  return FISTA if variant == 'fista' else ISTA


# ---------------------------------------------------------------------------
# precision policy of the inference plugins (not part of the reference API:
# the reference has one precision, float32)
# ---------------------------------------------------------------------------
_default_precision = os.environ.get('VTC_PRECISION', 'auto')


def set_default_precision(name):
  """'auto' | 'f32' | 'f16x3' | 'bf16x3' | 'bf16'.  'auto' picks f16x3 (an
  f16 hi/lo split, three products on the 16-bit matrix pipe: float32-level
  accuracy) when the fused kernel supports the shape, bf16x3 tiles for other
  large problems and exact-f32 MFMA otherwise."""
  global _default_precision
  assert name in ('auto',) + tuple(PRECISIONS)
  _default_precision = name


def get_default_precision():
  return _default_precision


# ---------------------------------------------------------------------------
# optional device-side timing of the inference entry point (bench.py uses it to
# time the FISTA kernel itself, without the Lipschitz eigen-solve around it)
# ---------------------------------------------------------------------------
kernel_timing = None   # None, or a list that receives (start, stop) events


class timed_call(object):
  """with timed_call(device): lib.vtc_...()  -- records a pair of events on
  the current stream when kernel_timing is a list."""

  def __init__(self, device):
    self.device = device

  def __enter__(self):
    if kernel_timing is not None:
      self.start = torch.cuda.Event(enable_timing=True)
      self.stop = torch.cuda.Event(enable_timing=True)
      self.start.record(torch.cuda.current_stream(self.device))
    return self

  def __exit__(self, *exc):
    if kernel_timing is not None:
      self.stop.record(torch.cuda.current_stream(self.device))
      kernel_timing.append((self.start, self.stop))
    return False


# ---------------------------------------------------------------------------
# Lipschitz constant
# ---------------------------------------------------------------------------
def gram(matrix, transpose_a):
  """A^T A (transpose_a) or A A^T of a 2-d device tensor through vtc_gram."""
  lib = load_library()
  rows, cols = matrix.shape
  side = cols if transpose_a else rows
  out = torch.empty((side, side), dtype=torch.float32, device=matrix.device)
  check(lib.vtc_gram(ptr(matrix), rows, cols, 1 if transpose_a else 0,
                     ptr(out), current_stream(matrix.device)), 'vtc_gram')
  return out


LANCZOS_MAX_N = 1024
_use_device_eigensolver = os.environ.get('VTC_EIGEN', 'lanczos') != 'library'
# Deferred error channel of the sync-free path: the eigen-solve kernel also
# stores [lambda_max, eta] through a pointer into pinned host memory (no copy,
# no wait), which is looked at once the event recorded behind the kernel has
# completed -- at the next plugin call.  A non-finite spectrum then raises the
# reference's RuntimeError (ista_fista.py:75-79) one call late instead of
# stalling every step behind a device-to-host read.  False: no mirror at all.
spectrum_check = os.environ.get('VTC_SPECTRUM_CHECK', '1') != '0'
_pending_spectra = []


def _report_bad_spectrum(dictionary_for_message):
  print('the eigen-solve threw an exception. Likely due to one of the',
        'dictionary elements overflowing. The norm of each dictionary',
        'element is')
  flat = dictionary_for_message.reshape(dictionary_for_message.shape[0], -1)
  print(torch.norm(flat, dim=1, p=2))
  raise RuntimeError()


def poll_spectrum_checks(block=False):
  """Look at the eigen-solves whose results have reached the host (all of
  them when block=True); raise like the reference if one was not finite."""
  while _pending_spectra:
    event, pinned, dictionary = _pending_spectra[0]
    if block:
      event.synchronize()
    elif not event.query():
      return
    _pending_spectra.pop(0)
    lam = float(pinned[0])
    if not (lam == lam) or lam in (float('inf'), float('-inf')):
      del _pending_spectra[:]
      _report_bad_spectrum(dictionary)
    if float(pinned[2]) != 1.0:
      del _pending_spectra[:]
      _report_unconverged(lam)


def _report_unconverged(lam):
  """The reference's symeig is exact; a Lanczos run that was still moving when
  it hit its step limit may sit BELOW lambda_max, i.e. give a step size above
  1 / L.  Treated like a failed eigen-solve (vtc_lambda_max's third output)."""
  raise RuntimeError(
      'the Lanczos eigen-solve had not converged at its step limit (top Ritz '
      'value %r still moving by more than 1e-6 per 8 steps)' % lam)


def lambda_max_device(gram_matrix, host_mirror=None):
  """[lambda_max, 1 / lambda_max] of a symmetric (n, n) device matrix,
  n <= 1024, and the convergence flag of the solve, as a 3-element device
  tensor (vtc_lambda_max: one small HIP kernel, Lanczos + Sturm counts).
  host_mirror: optional pinned CPU tensor of 3 floats the kernel writes the
  same values through."""
  lib = load_library()
  n = gram_matrix.shape[0]
  out = torch.empty(3, dtype=torch.float32, device=gram_matrix.device)
  ws = workspace(lib.vtc_lambda_max_workspace_bytes(n), gram_matrix.device)
  check(lib.vtc_lambda_max_mirrored(
      ptr(gram_matrix), n, ptr(out),
      ctypes.c_void_p(host_mirror.data_ptr() if host_mirror is not None else 0),
      ptr(ws), ws.numel(), current_stream(gram_matrix.device)),
        'vtc_lambda_max')
  return out


def device_stepsize_available(n):
  return _use_device_eigensolver and n <= LANCZOS_MAX_N


def stepsize_on_device(gram_matrix, dictionary_for_message):
  """eta = 1 / lambda_max(gram) as a 1-element DEVICE tensor, the way the
  reference keeps it (ista_fista.py:80): no host synchronisation.  Failure
  of the eigen-solve surfaces through poll_spectrum_checks()."""
  poll_spectrum_checks()
  pinned = None
  if spectrum_check:
    pinned = torch.empty(3, dtype=torch.float32, pin_memory=True)
  out = lambda_max_device(gram_matrix, pinned)
  if spectrum_check:
    event = torch.cuda.Event()
    event.record(torch.cuda.current_stream(gram_matrix.device))
    _pending_spectra.append((event, pinned, dictionary_for_message))
  return out[1:2]


def stepsize_from_gram(gram_matrix, dictionary_for_message):
  """eta = 1 / lambda_max(gram) as a Python float (one host sync).

  n <= 1024: vtc_lambda_max.  Otherwise torch.linalg.eigvalsh, the successor
  of the torch.symeig the reference calls (removed in torch >= 2) -- the one
  place a library routine stands in for a kernel of this engine; no
  configuration of the path produces such a matrix (patches beyond 32 x 32 or
  more than 1024 convolution kernels).  Mirrors the reference's error path:
  on failure print the kernel norms and raise a bare RuntimeError
  (ista_fista.py:75-79)."""
  n = gram_matrix.shape[0]
  unconverged = None
  try:
    if device_stepsize_available(n):
      out = lambda_max_device(gram_matrix)
      lipschitz_constant, stepsize, converged = [
          float(v) for v in out.tolist()]
      if not (lipschitz_constant == lipschitz_constant) or (
          lipschitz_constant in (float('inf'), float('-inf'))):
        raise RuntimeError('non-finite spectrum')
      if converged != 1.0:
        unconverged = lipschitz_constant
      else:
        return stepsize
    if unconverged is None:
      lipschitz_constant = torch.linalg.eigvalsh(gram_matrix, UPLO='U')[-1]
  except RuntimeError:
    _report_bad_spectrum(dictionary_for_message)
  if unconverged is not None:
    _report_unconverged(unconverged)
  return float(1. / lipschitz_constant)
