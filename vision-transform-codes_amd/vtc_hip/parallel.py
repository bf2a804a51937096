"""
Data-parallel context of the dictionary-update plugins.

The reference is single-device.  Here the patch batch is sharded over the GPUs
of a node (one process per GPU); inference needs no communication, and the
dictionary update needs exactly one exchange: the un-normalised gradient sum
C^T (C D - X) -- plus, for the cheap-quadratic rules, the per-atom code energy
behind the Hessian-diagonal EMA, packed into the same buffer -- is summed over
ranks with ONE RCCL all-reduce per update iteration.  Every rank then divides by the GLOBAL batch
size and applies the identical update, so dictionaries stay bit-identical.

torch.distributed's "nccl" backend is RCCL on ROCm; CPU tests use "gloo".
"""
import torch
import torch.distributed as dist

_group = None
_enabled = False
_equal_shards = True
_deferred = []      # [(tensor, callback)] riding on the next all_reduce_sum_
collectives_issued = 0   # all-reduce calls so far (tests count them)
packed_copies = 0        # all-reduces that had to pack their tensors first
# One persistent flat float32 buffer per device: `take` hands out consecutive
# slices of it (the code energy, then the gradient), so that the all-reduce of
# an update step sums one contiguous range in place -- no torch.cat, no copy
# back, no allocation per step.
_flat = {}          # device -> [flat tensor, cursor]
_ALIGN = 64         # elements (256 bytes) between slices


def enable(group=None, equal_shards=True):
  """Turn on gradient all-reduce inside dict_update_rules.*.run.

  group: a torch.distributed process group (None = the default group).
  equal_shards: every rank feeds the same local batch size, so the global
  batch is local*world_size and no size exchange is needed.
  """
  global _group, _enabled, _equal_shards
  if not dist.is_initialized():
    raise RuntimeError('torch.distributed is not initialised')
  _group, _enabled, _equal_shards = group, True, equal_shards


def disable():
  global _group, _enabled
  _group, _enabled = None, False
  del _deferred[:]
  _flat.clear()


def take(shape, device):
  """A float32 tensor of `shape` for a quantity that will be summed over
  ranks.  Data parallel: the next slice of the device's persistent flat
  buffer (released again by the all-reduce that consumes it); otherwise a
  plain allocation."""
  device = torch.device(device)
  count = 1
  for extent in shape:
    count *= int(extent)
  if not is_enabled():
    return torch.empty(tuple(shape), dtype=torch.float32, device=device)
  entry = _flat.get(device)
  padded = (count + _ALIGN - 1) // _ALIGN * _ALIGN
  if entry is None or entry[1] + padded > entry[0].numel():
    # first use, or outgrown: slices handed out earlier stay valid (they keep
    # the old storage alive) and that one step packs with a copy
    size = max(2 * padded + (entry[1] if entry else 0), 1 << 18)
    entry = _flat[device] = [
        torch.zeros(size, dtype=torch.float32, device=device), 0]
  flat, cursor = entry
  entry[1] = cursor + padded
  return flat[cursor: cursor + count].view(tuple(shape))


def _contiguous_range(tensors):
  """(flat, start, end) when every tensor is a slice of one persistent flat
  buffer and together (with their alignment gaps) they cover [start, end);
  None otherwise."""
  if not tensors:
    return None
  entry = _flat.get(tensors[0].device)
  if entry is None:
    return None
  flat = entry[0]
  base = flat.data_ptr()
  spans = []
  for t in tensors:
    if (t.dtype != torch.float32 or not t.is_contiguous() or
        t.untyped_storage().data_ptr() != flat.untyped_storage().data_ptr()):
      return None
    first = (t.data_ptr() - base) // 4
    spans.append((first, first + t.numel()))
  spans.sort()
  for (_, end), (start, _) in zip(spans, spans[1:]):
    if start < end or start - end >= _ALIGN:
      return None
  return flat, spans[0][0], spans[-1][1]


def drop_deferred():
  """Forget what `defer` queued and release the flat buffer's slices: called
  when an update plugin raised between defer() and its all-reduce, so that the
  next step does not reduce a different element count than the other ranks."""
  del _deferred[:]
  for entry in _flat.values():
    entry[1] = 0


def is_enabled():
  return _enabled and dist.is_initialized()


def world_size():
  return dist.get_world_size(_group) if is_enabled() else 1


def rank():
  return dist.get_rank(_group) if is_enabled() else 0


def global_batch(local_batch, device):
  """Number of samples over all ranks."""
  if not is_enabled():
    return int(local_batch)
  if _equal_shards:
    return int(local_batch) * world_size()
  global collectives_issued
  collectives_issued += 1
  count = torch.tensor([int(local_batch)], dtype=torch.int64, device=device)
  dist.all_reduce(count, op=dist.ReduceOp.SUM, group=_group)
  return int(count.item())


def defer(tensor, callback):
  """Sum `tensor` over ranks as part of the NEXT all_reduce_sum_ call (same
  flat buffer, same collective) and run `callback()` right after it, before
  that call returns.  The trainer uses it for the code energy of the
  Hessian-diagonal EMA: the EMA kernel is enqueued after the reduce and before
  the update plugin applies the gradient, so the cheap-quadratic step costs
  one collective, not two.  Without data parallelism the callback runs at
  once."""
  if not is_enabled():
    callback()
    return
  _deferred.append((tensor, callback))


def flush_deferred():
  """Reduce whatever is still deferred (a plugin that never reduced)."""
  if _deferred:
    all_reduce_sum_()


def all_reduce_sum_(*tensors):
  """In-place sum over ranks.  Several tensors (and whatever `defer` queued)
  are packed into one flat buffer so that one collective (one launch latency)
  covers them: at 1 MiB the exchange is latency-bound on xGMI, not
  bandwidth-bound."""
  global collectives_issued
  if not is_enabled():
    return
  pending = list(_deferred)
  del _deferred[:]
  everything = list(tensors) + [t for t, _ in pending]
  if not everything:
    return
  collectives_issued += 1
  in_place = _contiguous_range(everything)
  for entry in _flat.values():
    entry[1] = 0           # the slices handed out so far are consumed
  if in_place is not None:
    flat, start, end = in_place
    dist.all_reduce(flat[start:end], op=dist.ReduceOp.SUM, group=_group)
  elif len(everything) == 1:
    dist.all_reduce(everything[0], op=dist.ReduceOp.SUM, group=_group)
  else:
    global packed_copies
    packed_copies += 1
    flat = torch.cat([t.reshape(-1) for t in everything])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=_group)
    offset = 0
    for t in everything:
      t.copy_(flat[offset: offset + t.numel()].view_as(t))
      offset += t.numel()
  for _, callback in pending:
    callback()


def broadcast_reset_or_prune(dictionary, affected, groups):
  """Make every rank adopt rank 0's reset / prune decision
  (training/sparse_coding.py: dict_element_rp_schedule under data parallelism:
  the selection draws from each process's own numpy / torch CPU generators, so
  only rank 0's draw counts).

  dictionary: rank 0: the tensor reset_or_prune_dict_elements returned; other
  ranks: their current dictionary.  affected / groups: rank 0's results
  (ignored elsewhere).  Returns (dictionary, affected, groups) identical on
  all ranks: after a reset the caller's tensor, overwritten in place; after a
  prune a tensor of rank 0's new shape."""
  if not is_enabled():
    return dictionary, affected, groups
  header = [None]
  if rank() == 0:
    header = [(tuple(dictionary.shape),
               [int(a) for a in torch.as_tensor(affected).reshape(-1).tolist()]
               if len(affected) else [], groups)]
  dist.broadcast_object_list(header, src=0, group=_group)
  shape, affected_list, groups0 = header[0]
  if rank() != 0:
    import numpy as np
    affected = np.asarray(affected_list, dtype=np.int64)
    if groups is not None and groups0 is not None:
      groups[:] = groups0           # in place: the trainer aliases this list
    if tuple(dictionary.shape) != tuple(shape):
      dictionary = torch.empty(shape, dtype=dictionary.dtype,
                               device=dictionary.device)
  if not dictionary.is_contiguous():
    dictionary = dictionary.contiguous()
  dist.broadcast(dictionary, src=0, group=_group)
  return dictionary, affected, groups
