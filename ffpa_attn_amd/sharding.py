"""(batch, kv-head) sharding of the attention forward over the GPUs of one node.

The reference has no multi-GPU attention path at all (no collective call site; its
only multi-GPU code is a Ray task farm for autotuning, src/ffpa_attn/ray/).  The forward is
embarrassingly parallel over ``(b, hkv)``: output rows of query head ``hq`` depend only on
``Q[b, hq]`` and ``K/V[b, hq // group]`` (csrc/cuffpa/native/sm_80/split_d.cuh:135-142).  So the
MI355X plan is: one process per GPU, each owns a contiguous block of ``B*Hkv`` *units* (a KV head
together with its ``group`` query heads, so K/V are never duplicated), runs the HIP kernel on its
block, and — only if the caller wants the full tensor on every rank — ONE ``all_gather`` of O over
RCCL/xGMI, straight into the output tensor.  There is no reduction and no exchange inside the
attention itself.

Two entries:

* born-sharded (the data-parallel case, what ``bench.py`` measures): every rank holds only its block in
  unit-major layout ``q [U, group, Nq, D]``, ``k / v [U, 1, Nkv, D]`` — ``local_units`` /
  ``synthetic_unit_block`` build it, ``attend_units`` runs the kernel, ``gather_units`` is the optional
  collective, ``attend_and_gather_units`` overlaps the two (the block in pieces, each piece sent point-to-point into its final place on every other rank
  while the next computes);
* replicated inputs (``sharded_attention``): every rank holds the full tensors and takes views of its block.
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


def local_units(n_units: int, group: "dist.ProcessGroup | None" = None) -> tuple[int, int]:
  """``[start, stop)`` of the calling rank's units (the whole range without an initialised process group)."""
  if dist.is_available() and dist.is_initialized():
    return partition_units(n_units, dist.get_world_size(group), dist.get_rank(group))
  return 0, n_units


def synthetic_unit_block(start: int, stop: int, group_size: int, nq: int, nkv: int, head_dim: int, *, dtype=torch.bfloat16,
                         device="cuda", seed: int = 0):
  """Born-sharded synthetic inputs: units ``[start, stop)`` of a global problem, each unit drawn from its own
  generator (seed, unit index) — q, then k, then v, N(0,1) like the reference's bench inputs
  (src/ffpa_attn/cli/_runner_fwd.py:344-347).  The values of a unit do not depend on how the units are spread over
  ranks, so any partition of the same global problem computes the same numbers; nothing is replicated."""
  n = stop - start
  q = torch.empty((n, group_size, nq, head_dim), dtype=dtype, device=device)
  k = torch.empty((n, 1, nkv, head_dim), dtype=dtype, device=device)
  v = torch.empty((n, 1, nkv, head_dim), dtype=dtype, device=device)
  gen = torch.Generator(device=device)
  for i in range(n):
    gen.manual_seed((int(seed) << 32) + start + i)
    q[i].normal_(generator=gen)
    k[i].normal_(generator=gen)
    v[i].normal_(generator=gen)
  return q, k, v


def attend_units(qu: torch.Tensor, ku: torch.Tensor, vu: torch.Tensor, **kwargs) -> torch.Tensor:
  """The local step: this rank's block through ``ffpa_attn_func`` (HIP kernel; dim 0 = units as the batch axis,
  dim 1 = the unit's query heads against its single KV head)."""
  from .interface import ffpa_attn_func

  if qu.size(0) == 0:
    return qu.new_empty(qu.shape)
  return ffpa_attn_func(qu, ku, vu, enable_gqa=qu.size(1) != ku.size(1), **kwargs)


def _check_out(out: "torch.Tensor | None", shape: tuple, like: torch.Tensor) -> None:
  """A caller-supplied result buffer is written by slices and by RCCL: a wrong shape / dtype / device / layout must fail here, with
  words, not as a partial fill or an opaque collective error."""
  if out is None:
    return
  if tuple(out.shape) != tuple(shape) or out.dtype != like.dtype or out.device != like.device or not out.is_contiguous():
    raise ValueError(f"out must be a contiguous {like.dtype} tensor of shape {tuple(shape)} on {like.device}, got {out.dtype} "
                     f"{tuple(out.shape)} on {out.device} (contiguous: {out.is_contiguous()})")


def gather_units(o_local: torch.Tensor, n_units: int, group: "dist.ProcessGroup | None" = None,
                 out: torch.Tensor | None = None) -> torch.Tensor:
  """All ranks' blocks -> ``[n_units, group, Nq, D]`` on every rank with ONE ``all_gather_into_tensor``.
  When the units divide evenly (config 5: 256 units over 1 / 2 / 4 / 8 GPUs) the collective writes straight
  into the result — no padding, no concatenation; otherwise blocks are padded to the largest and compacted
  in place afterwards (the only copy, of the tail blocks)."""
  world = dist.get_world_size(group)
  g, nq, d = o_local.shape[1:]
  _check_out(out, (n_units, g, nq, d), o_local)
  per = -(-n_units // world)
  if n_units == 0:  # nothing to gather (no collective over empty buffers)
    return out if out is not None else o_local.new_empty((0, g, nq, d))
  if n_units % world == 0:
    if out is None:
      out = o_local.new_empty((n_units, g, nq, d))
    dist.all_gather_into_tensor(out, o_local.contiguous(), group=group)
    return out
  buf = o_local.new_empty((per, g, nq, d))
  buf[: o_local.size(0)] = o_local
  padded = o_local.new_empty((world * per, g, nq, d))
  dist.all_gather_into_tensor(padded, buf, group=group)
  for r in range(1, world):  # block r sits at r * per, belongs at start(r) <= r * per: move down, in rank order
    s, e = partition_units(n_units, world, r)
    if s != r * per:
      padded[s:e] = padded[r * per : r * per + (e - s)].clone()
  if out is None:
    return padded[:n_units]
  out.copy_(padded[:n_units])  # (uneven split with a caller's buffer: the collective needs equal shards, so this one copy stays)
  return out


def _piece_plan_chunks(qu: torch.Tensor, ku: torch.Tensor, chunks: int, kwargs: dict) -> int:
  """``chunks`` lowered until every piece of this rank's block runs the launch plan of the whole block.  The library picks tile and KV-split count from
  the size of a launch (ffpa_capi.hip make_plan: under-filled / ragged-round KV splits — fp32 partials + an LSE merge —, the wide-row tile of D = 320, the
  short-query split rule): the same values to rounding, not to the bit.  Asked of the library itself (``hip.launch_plan``), never re-derived here, for the
  launch ``hip.forward`` would make of it: GQA heads packed into the row axis for Nq <= 7 (``[B, Hkv, group Nq]``), ``FFPA_HIP_PREFILL_SPLITS=0`` honoured.
  What the question does NOT carry: the mask's strides (a key-bias layout is assumed) and precomputed mask ranges — for masked calls the plan of a piece
  can still differ from the whole's in the rare shapes where those decide a rule; the result then agrees to rounding instead of to the bit.  Calls
  that do not reach the HIP kernel (an SDPA backend, head dims / sequence lengths the dispatch sends to SDPA) and boxes without the library keep the
  requested count: there is no launch plan to preserve."""
  per = qu.size(0)
  if chunks <= 1 or not qu.is_cuda:
    return max(1, chunks)
  try:
    import os

    from . import hip
    from .functional import FFPAAttnMeta

    kw = dict(kwargs)
    mask = kw.pop("attn_mask", None)
    dropout_p = float(kw.pop("dropout_p", 0.0))
    causal = bool(kw.pop("is_causal", False))
    kw.pop("scale", None)
    meta = FFPAAttnMeta.from_kwargs(**kw)
    if meta.fallback(qu, ku, mask, dropout_p, is_causal=causal):
      return chunks
    g, nq, d = qu.shape[1:]
    # the launch hip.forward makes of a piece: packed GQA heads for short queries, the prefill-split opt-out of the environment
    packed = g > 1 and nq <= 7 and g * nq <= 32 and mask is None and dropout_p == 0.0
    heads_q, rows = (1, g * nq) if packed else (g, nq)
    no_prefill_splits = os.environ.get("FFPA_HIP_PREFILL_SPLITS", "1").lower() in ("0", "off", "false", "no")
    num_splits = 1 if (rows > 32 and no_prefill_splits) else 0

    def plan(units: int) -> tuple:
      pl = hip.launch_plan(units, heads_q, 1, rows, ku.size(2), d, dtype=qu.dtype, causal=causal, bias_dtype=None if mask is None else mask.dtype,
                           dropout_p=dropout_p, device=qu.device, num_splits=num_splits)
      return pl["variant"], pl["block_rows"], pl["block_keys"], pl["splits"]

    whole = plan(per)
    while chunks > 1:
      sizes = {per * (c + 1) // chunks - per * c // chunks for c in range(chunks)}
      if all(plan(n) == whole for n in sizes):
        break
      chunks -= 1
    return chunks
  except (RuntimeError, OSError, ValueError, TypeError, NotImplementedError, LookupError):
    return chunks  # (library missing / shape or dtype it refuses: the call below raises or falls back exactly as the plain call would)


# ---- what the ranks of a group have agreed on, once per process: how pieces travel, and into how many pieces a block of a given shape class is cut.
# Both MUST be the same on every rank (the receives one rank posts have to match the sends of the others): each is settled by a MIN all-reduce the
# first time it is needed and remembered, so the steady state pays no extra collective and no host synchronisation.
_TRANSPORT: "dict[object, str]" = {}
_AGREED_CHUNKS: "dict[tuple, int]" = {}


def _group_key(group) -> object:
  return id(group) if group is not None else None


def _agree_min(value: int, like: torch.Tensor, group) -> int:
  t = torch.tensor([int(value)], dtype=torch.int64, device=like.device)
  dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
  return int(t.item())


def gather_transport(like: torch.Tensor, group: "dist.ProcessGroup | None" = None) -> str:
  """How ``attend_and_gather_units`` moves a finished piece: ``"p2p"`` — ``dist.batch_isend_irecv`` straight into the final slices — or
  ``"all_gather"`` — ``all_gather_into_tensor`` into a ``world x piece`` temporary + one copy per peer.  Decided ONCE per group: a two-element
  point-to-point ring exchange is tried (RCCL P2P needs peer access between the GPUs of the node; where it is unavailable the call raises), the ranks take
  the MIN of their outcomes, and everybody uses the same form from then on.  ``FFPA_SHARDING_TRANSPORT=p2p|all_gather`` pins it (tests; a node whose
  P2P hangs instead of raising)."""
  import os

  key = _group_key(group)
  hit = _TRANSPORT.get(key)
  if hit is not None:
    return hit
  forced = os.environ.get("FFPA_SHARDING_TRANSPORT", "").lower()
  if forced in ("p2p", "all_gather"):
    _TRANSPORT[key] = forced
    return forced
  world, rank = dist.get_world_size(group), dist.get_rank(group)
  ok = 1
  if world > 1:
    try:
      send = torch.full((2,), float(rank), dtype=torch.float32, device=like.device)
      recv = torch.empty_like(send)
      dst, src = (rank + 1) % world, (rank - 1) % world
      ops = [dist.P2POp(dist.isend, send, dst if group is None else dist.get_global_rank(group, dst), group),
             dist.P2POp(dist.irecv, recv, src if group is None else dist.get_global_rank(group, src), group)]
      for r in dist.batch_isend_irecv(ops):
        r.wait()
      if like.is_cuda:
        torch.cuda.synchronize(like.device)
      ok = 1 if float(recv[0].item()) == float(src) else 0
    except RuntimeError:
      ok = 0
    ok = _agree_min(ok, like, group)
  _TRANSPORT[key] = "p2p" if ok else "all_gather"
  return _TRANSPORT[key]


def attend_and_gather_units(qu: torch.Tensor, ku: torch.Tensor, vu: torch.Tensor, n_units: int, *, chunks: int = 4,
                            group: "dist.ProcessGroup | None" = None, out: torch.Tensor | None = None, stats: dict | None = None,
                            **kwargs) -> torch.Tensor:
  """The local step and the optional collective, overlapped: this rank's block is attended in ``chunks`` pieces (whole units) and every finished
  piece travels to the other ranks while the next piece computes — the gather of a 256 MiB shard costs about as much as the kernel (7 x 153 GB/s xGMI
  links), so hiding it under compute is the difference between ~1x and ~2x the step time.  Needs an even split (``n_units % world == 0``); returns
  ``[n_units, group, Nq, D]`` on every rank.

  Every piece goes STRAIGHT into its final place: piece c of rank r belongs at ``out[r * per + c0 : r * per + c1]`` — contiguous, but the world's pieces of
  one chunk are ``per`` units apart, which no single all-gather can write (the list form of ``dist.all_gather`` accepts such views and then gathers into a
  temporary of ``world x piece`` and copies out: a hidden extra pass over the data on the stream the overlap is meant to keep free).  So a chunk is one
  batch of point-to-point operations (``dist.batch_isend_irecv``: world - 1 sends of the piece, world - 1 receives into the final slices; RCCL runs the batch
  as one group on its own stream, ordered behind the kernel that produced the piece) — on xGMI's full mesh of point-to-point links that is also the
  native pattern: every peer is reached over its own link.  The rank's own piece is one device-to-device copy.

  Where RCCL point-to-point is unavailable (``gather_transport``: probed once per group, agreed by all ranks) a chunk travels as one asynchronous
  ``all_gather_into_tensor`` into a ``world x piece`` temporary and is copied to its slices on arrival — the same result, one more pass over the data.
  ``stats``, if given, receives ``transport`` (``"p2p"`` / ``"all_gather"`` / ``"local"``), ``chunks`` and ``world``.

  ``chunks`` is a request: it is lowered until every piece runs the launch plan of the whole block (``_piece_plan_chunks``) and then to the smallest count
  any rank arrived at (agreed once per shape class), so for unmasked calls that reach the HIP kernel the result does not depend on it, not even at the
  rounding level (masked calls: see ``_piece_plan_chunks``)."""
  world = dist.get_world_size(group)
  rank = dist.get_rank(group)
  if n_units % world != 0:
    return gather_units(attend_units(qu, ku, vu, **kwargs), n_units, group, out)
  per = n_units // world
  if qu.size(0) != per:
    raise ValueError(f"this rank holds {qu.size(0)} units, expected {per} (= {n_units} / {world})")
  g, nq, d = qu.shape[1:]
  _check_out(out, (n_units, g, nq, d), qu)
  if out is None:
    out = qu.new_empty((n_units, g, nq, d))
  if per == 0:
    return out
  want = max(1, min(chunks, per))
  ckey = (_group_key(group), want, tuple(qu.shape), tuple(ku.shape), qu.dtype, str(qu.device.type),
          tuple(sorted((k_, (tuple(v_.shape), v_.dtype) if isinstance(v_, torch.Tensor) else repr(v_)) for k_, v_ in kwargs.items())))
  chunks = _AGREED_CHUNKS.get(ckey)
  if chunks is None:
    # the lowered count is derived from local device state (CU count, environment): ranks that disagreed would post mismatched receives — a hang or
    # a corrupted gather.  The smallest count every rank can honour is one every rank's plan rule accepts (fewer, larger pieces only move a piece
    # TOWARDS the whole block's plan).
    chunks = _piece_plan_chunks(qu, ku, want, kwargs)
    if world > 1:
      chunks = max(1, _agree_min(chunks, qu, group))
    if len(_AGREED_CHUNKS) >= 256:
      _AGREED_CHUNKS.clear()
    _AGREED_CHUNKS[ckey] = chunks
  transport = gather_transport(qu, group) if world > 1 else "local"
  if stats is not None:
    stats.update(transport=transport, chunks=chunks, world=world)
  bounds = [per * c // chunks for c in range(chunks + 1)]
  works = []
  for c0, c1 in zip(bounds, bounds[1:]):
    o_c = attend_units(qu[c0:c1], ku[c0:c1], vu[c0:c1], **kwargs).contiguous()
    if world == 1 or transport == "p2p":
      out[rank * per + c0 : rank * per + c1].copy_(o_c)
    if world > 1 and transport == "p2p":
      ops = []
      for step in range(1, world):  # (peers in rotated order: at every step of the batch each rank sends to and receives from a different peer)
        dst, src = (rank + step) % world, (rank - step) % world
        ops.append(dist.P2POp(dist.isend, o_c, dst if group is None else dist.get_global_rank(group, dst), group))
        ops.append(dist.P2POp(dist.irecv, out[src * per + c0 : src * per + c1], src if group is None else dist.get_global_rank(group, src), group))
      works.append((dist.batch_isend_irecv(ops), o_c, None))  # (the piece stays referenced until its sends have completed)
    elif world > 1:
      # no point-to-point on this node: the world's pieces of this chunk into a temporary (asynchronously, RCCL's stream), copied to their slices — ``per``
      # units apart in the result — once they have arrived
      tmp = o_c.new_empty((world, c1 - c0, g, nq, d))
      works.append(([dist.all_gather_into_tensor(tmp.view(world * (c1 - c0), g, nq, d), o_c, group=group, async_op=True)], o_c, (tmp, c0, c1)))
  for reqs, _, landed in works:
    for r in reqs:
      r.wait()
    if landed is not None:
      tmp, c0, c1 = landed
      out.view(world, per, g, nq, d)[:, c0:c1].copy_(tmp)
  return out


def sharded_attention(
  q: torch.Tensor,
  k: torch.Tensor,
  v: torch.Tensor,
  attn_fn: Callable[[torch.Tensor, torch.Tensor, torch.Tensor], torch.Tensor] | None = None,
  *,
  group: "dist.ProcessGroup | None" = None,
  gather: bool = True,
) -> torch.Tensor:
  """Replicated-input entry: every rank holds the full ``q, k, v``; each computes its block of units with
  ``attn_fn`` (default: ``attend_units``, the HIP kernel; it sees ``[U, group, Nq, D]`` / ``[U, 1, Nkv, D]``
  and must treat dim 1 as GQA heads) and, with ``gather=True``, the blocks are all-gathered into the full
  ``[B, Hq, Nq, D]`` output.  ``gather=False`` returns the local block."""
  world = dist.get_world_size(group)
  rank = dist.get_rank(group)
  B, Hq, Nq, D = q.shape
  Hkv = k.size(1)
  g = Hq // Hkv
  ql, kl, vl = shard_units(q, k, v, world, rank)
  fn = attn_fn or attend_units
  o_local = fn(ql, kl, vl) if ql.size(0) > 0 else ql.new_empty((0, g, Nq, D))
  if not gather:
    return o_local
  return gather_units(o_local, B * Hkv, group).reshape(B, Hq, Nq, D)
