"""
Hessian-diagonal-scaled dictionary update with a within-group alignment
penalty, for subspace sparse coding on MI355X.

Drop-in for vision_transform_codes/dict_update_rules/fully_connected/
subspace_sc_cheap_quadratic_descent.py:13-127.  The penalty gradient (sum over
the other members j of a group of sign(cos_ij) (d_j - cos_ij d_i), or its
norm-aware form for un-normalised dictionaries) is computed by one HIP block
per group instead of a Python loop over groups.
"""
import torch

import vtc_hip
from vtc_hip import groups as group_tables
from dict_update_rules.fully_connected import _common


def run(images, dictionary, codes, group_assignments, hessian_diagonal,
        alignment_penalty, stepsize=0.001, num_iters=1, lowest_code_val=0.001,
        normalize_dictionary=True):
  """
  images (b, n), dictionary (s, n) [updated IN PLACE], codes (b, s),
  hessian_diagonal (s,), group_assignments as in subspace_ista_fista.
  Returns None.
  """
  penalty = None
  if alignment_penalty != 0:
    lib = vtc_hip.load_library()
    vtc_hip.require_device_tensor(dictionary, 'dictionary')
    s, n = dictionary.shape
    device = dictionary.device
    tables = group_tables.tables_for(group_assignments, s, device)
    ws = vtc_hip.workspace(
        lib.vtc_subspace_alignment_gradient_workspace_bytes(tables.slots, n),
        device)
    penalty_grad = torch.empty((s, n), dtype=torch.float32, device=device)

    def compute_penalty_gradient():
      vtc_hip.check(lib.vtc_subspace_alignment_gradient(
          vtc_hip.ptr(dictionary), vtc_hip.ptr(tables.index),
          vtc_hip.ptr(tables.valid), vtc_hip.ptr(tables.atom_ptr),
          vtc_hip.ptr(tables.atom_slots), vtc_hip.ptr(penalty_grad), s, n,
          tables.num_groups, tables.m, 1 if normalize_dictionary else 0,
          vtc_hip.ptr(ws), ws.numel(), vtc_hip.current_stream(device)),
          'vtc_subspace_alignment_gradient')
      return penalty_grad

    penalty = (float(alignment_penalty), compute_penalty_gradient)
  _common.descend(images, dictionary, codes, stepsize, num_iters,
                  normalize_dictionary, hessian_diagonal=hessian_diagonal,
                  lowest_code_val=lowest_code_val, penalty=penalty)
