"""(batch, kv-head) sharding of the attention forward over the GPUs of one node.

The reference has no multi-GPU attention path at all (no collective call site; its
only multi-GPU code is a Ray task farm for autotuning, src/ffpa_attn/ray/).  The forward is
embarrassingly parallel over ``(b, hkv)``: output rows of query head ``hq`` depend only on
``Q[b, hq]`` and ``K/V[b, hq // group]`` (csrc/cuffpa/native/sm_80/split_d.cuh:135-142).  So the
MI355X plan is: one process per GPU, each owns a contiguous block of ``B*Hkv`` *units* (a KV head
together with its ``group`` query heads, so K/V are never duplicated), runs the kernel locally, and
— only if the caller wants the full tensor on every rank — one ``all_gather`` of O over RCCL/xGMI.
There is no reduction and no exchange inside the attention itself.
"""

from __future__ import annotations

from typing import Callable

import torch
import torch.distributed as dist


def partition_units(n_units: int, world_size: int, rank: int) -> tuple[int, int]:
  """Contiguous, balanced ``[start, stop)`` block of ``n_units`` for ``rank`` (first ranks get the
  remainder)."""
  if world_size <= 0 or not (0 <= rank < world_size):
    raise ValueError(f"bad rank {rank} / world_size {world_size}")
  base, rem = divmod(n_units, world_size)
  start = rank * base + min(rank, rem)
  return start, start + base + (1 if rank < rem else 0)


def to_units(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor):
  """``[B,Hq,Nq,D]`` / ``[B,Hkv,Nkv,D]`` -> unit-major views ``[B*Hkv, group, Nq, D]`` /
  ``[B*Hkv, 1, Nkv, D]`` (no copy for contiguous inputs)."""
  B, Hq, Nq, D = q.shape
  _, Hkv, Nkv, _ = k.shape
  if Hq % Hkv != 0:
    raise ValueError(f"num_heads: Hq={Hq} is not a multiple of Hkv={Hkv}")
  g = Hq // Hkv
  qu = q.reshape(B * Hkv, g, Nq, D)
  ku = k.reshape(B * Hkv, 1, Nkv, D)
  vu = v.reshape(B * Hkv, 1, Nkv, D)
  return qu, ku, vu


def shard_units(q, k, v, world_size: int, rank: int):
  """This rank's block of units (views)."""
  qu, ku, vu = to_units(q, k, v)
  s, e = partition_units(qu.size(0), world_size, rank)
  return qu[s:e], ku[s:e], vu[s:e]


def sharded_attention(
  q: torch.Tensor,
  k: torch.Tensor,
  v: torch.Tensor,
  attn_fn: Callable[[torch.Tensor, torch.Tensor, torch.Tensor], torch.Tensor],
  *,
  group: "dist.ProcessGroup | None" = None,
  gather: bool = True,
) -> torch.Tensor:
  """Every rank holds the full ``q, k, v``; each computes its block of units with ``attn_fn`` (which
  sees ``[U, group, Nq, D]`` / ``[U, 1, Nkv, D]`` and must treat dim 1 as GQA heads) and, with
  ``gather=True``, the blocks are all-gathered into the full ``[B, Hq, Nq, D]`` output.
  """
  world = dist.get_world_size(group)
  rank = dist.get_rank(group)
  B, Hq, Nq, D = q.shape
  Hkv = k.size(1)
  g = Hq // Hkv
  ql, kl, vl = shard_units(q, k, v, world, rank)
  o_local = attn_fn(ql, kl, vl) if ql.size(0) > 0 else ql.new_empty((0, g, Nq, D))
  if not gather:
    return o_local
  n_units = B * Hkv
  per = -(-n_units // world)  # pad every block to the largest so one all_gather_into_tensor suffices
  buf = o_local.new_zeros((per, g, Nq, D))
  buf[: o_local.size(0)] = o_local
  out = o_local.new_empty((world * per, g, Nq, D))
  dist.all_gather_into_tensor(out, buf, group=group)
  pieces = []
  for r in range(world):
    s, e = partition_units(n_units, world, r)
    pieces.append(out[r * per : r * per + (e - s)])
  return torch.cat(pieces, dim=0).reshape(B, Hq, Nq, D)
