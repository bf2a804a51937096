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
  if len(everything) == 1:
    dist.all_reduce(everything[0], op=dist.ReduceOp.SUM, group=_group)
  else:
    flat = torch.cat([t.reshape(-1) for t in everything])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=_group)
    offset = 0
    for t in everything:
      t.copy_(flat[offset: offset + t.numel()].view_as(t))
      offset += t.numel()
  for _, callback in pending:
    callback()
