"""Padded-group index tables shared by the subspace plugins.

The reference builds a zero-padded (b, G, m) code tensor and a (G*m, n)
duplicated dictionary with Python loops over the groups
(analysis_transforms/fully_connected/subspace_ista_fista.py:94-111) on every
call.  Here the group structure is turned once into four small index arrays on
the device -- slot -> atom (`index`, `valid`) and its inverse in CSR form
(`atom_ptr`, `atom_slots`) -- and the HIP gather/scatter kernels do the rest.
"""
import numpy as np
import torch

_cache = {}


class GroupTables(object):
  def __init__(self, group_assignments, num_atoms, device):
    groups = [[int(a) for a in g] for g in group_assignments]
    self.num_groups = len(groups)
    self.m = max(len(g) for g in groups)
    self.num_atoms = int(num_atoms)
    slots = self.num_groups * self.m
    index = np.zeros(slots, dtype=np.int32)
    valid = np.zeros(slots, dtype=np.uint8)
    per_atom = [[] for _ in range(self.num_atoms)]
    for g, members in enumerate(groups):
      for j, atom in enumerate(members):
        if not 0 <= atom < self.num_atoms:
          raise IndexError('group %d refers to atom %d of %d'
                           % (g, atom, self.num_atoms))
        index[g * self.m + j] = atom
        valid[g * self.m + j] = 1
        per_atom[atom].append(g * self.m + j)   # increasing slot order
    atom_ptr = np.zeros(self.num_atoms + 1, dtype=np.int32)
    atom_ptr[1:] = np.cumsum([len(x) for x in per_atom])
    atom_slots = np.array([t for x in per_atom for t in x] or [0],
                          dtype=np.int32)
    self.slots = slots
    self.index = torch.from_numpy(index).to(device)
    self.valid = torch.from_numpy(valid).to(device)
    self.atom_ptr = torch.from_numpy(atom_ptr).to(device)
    self.atom_slots = torch.from_numpy(atom_slots).to(device)


def tables_for(group_assignments, num_atoms, device):
  """Memoised on the identity-free content of the grouping."""
  key = (tuple(tuple(int(a) for a in g) for g in group_assignments),
         int(num_atoms), str(device))
  hit = _cache.get(key)
  if hit is None:
    if len(_cache) > 8:
      _cache.clear()
    hit = _cache[key] = GroupTables(group_assignments, num_atoms, device)
  return hit
