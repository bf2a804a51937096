"""The C-ABI library loads and exports every symbol include/vtc_hip.h declares.
No GPU needed: nothing here launches a kernel."""
import ctypes
import pathlib
import re

import pytest

REPO = pathlib.Path(__file__).resolve().parent.parent
HEADER = REPO / 'include' / 'vtc_hip.h'


def declared_functions():
  text = HEADER.read_text()
  text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
  names = re.findall(r'\b(vtc_[a-z0-9_]+)\s*\(', text)
  return sorted(set(names))


def test_header_declares_the_expected_surface():
  names = declared_functions()
  for required in ('vtc_fc_ista_fista', 'vtc_subspace_ista_fista',
                   'vtc_conv_ista_fista', 'vtc_fc_dict_gradient',
                   'vtc_fc_dict_apply', 'vtc_conv_dict_gradient',
                   'vtc_conv_dict_apply', 'vtc_gram', 'vtc_code_energy',
                   'vtc_hessian_ema'):
    assert required in names


def test_library_exports_every_declared_symbol():
  import vtc_hip
  assert vtc_hip.LIBRARY_PATH.exists(), (
      'libvtc_hip.so missing: run __graft_entry__.build()')
  raw = ctypes.CDLL(str(vtc_hip.LIBRARY_PATH))
  for name in declared_functions():
    assert hasattr(raw, name), 'library does not export ' + name
  # and the Python binding table covers exactly the header
  assert sorted(vtc_hip.SIGNATURES) == declared_functions()


def test_binding_loads_and_reports_version():
  import vtc_hip
  lib = vtc_hip.load_library()
  assert lib.vtc_abi_version() == vtc_hip.ABI_VERSION == 4
  assert b'gfx950' in lib.vtc_version()


def test_workspace_queries_are_host_only():
  import vtc_hip
  lib = vtc_hip.load_library()
  b, n, s = 4096, 256, 1024
  need = lib.vtc_fc_ista_fista_workspace_bytes(b, n, s, vtc_hip.F32)
  assert need >= 4 * (b * s + b * n)
  assert lib.vtc_fc_dict_gradient_workspace_bytes(b, n, s) >= 4 * b * n
  assert lib.vtc_code_energy_workspace_bytes(b, s, 1) >= 4 * s
  geom = vtc_hip.ConvGeometry(b=2, c=1, h=52, w=52, s=8, kh=11, kw=11,
                              stride_v=1, stride_h=1, has_padding=1,
                              pad_lead_v=10, pad_trail_v=10, pad_lead_h=10,
                              pad_trail_h=10)
  ch, cw = ctypes.c_int32(0), ctypes.c_int32(0)
  assert lib.vtc_conv_code_dims(ctypes.byref(geom), ctypes.byref(ch),
                                ctypes.byref(cw)) == 0
  assert (ch.value, cw.value) == (42, 42)
  assert lib.vtc_conv_ista_fista_workspace_bytes(ctypes.byref(geom)) >= (
      4 * 2 * 8 * 42 * 42)


def test_argument_errors_do_not_touch_the_gpu():
  import vtc_hip
  lib = vtc_hip.load_library()
  # null pointers are rejected before any HIP call
  rc = lib.vtc_fc_ista_fista(None, None, None, None, 1, 1, 1, 1.0, 0.1, 1, 1,
                             0, -1.0, 0, None, 0, None, None)
  assert rc == vtc_hip.ERR_INVALID_ARGUMENT
  assert b'null' in lib.vtc_last_error()
  with pytest.raises(ValueError):
    vtc_hip.check(rc, 'vtc_fc_ista_fista')


def test_cpu_tensors_are_refused_loudly():
  """No silent CPU fallback: a CPU tensor is an error, not a slow path."""
  import torch
  import vtc_hip
  from analysis_transforms.fully_connected import ista_fista
  from dict_update_rules.fully_connected import sc_steepest_descent
  X, D = torch.zeros(4, 8), torch.eye(8)
  with pytest.raises(vtc_hip.VtcHipError):
    ista_fista.run(X, D, 0.1, 3)
  with pytest.raises(vtc_hip.VtcHipError):
    sc_steepest_descent.run(X, D, torch.zeros(4, 8))
