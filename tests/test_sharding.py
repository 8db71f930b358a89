"""(batch, kv-head) sharding across ranks: partition arithmetic + a real world_size=2 gloo run on CPU
(the N > 1 path of bench.py / sharding.sharded_attention with an injected local attention)."""

import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ffpa_attn_amd.sharding import partition_units, shard_units, to_units


def test_partition_is_a_contiguous_cover():
  for n in (1, 2, 7, 16, 256):
    for w in (1, 2, 3, 4, 8):
      spans = [partition_units(n, w, r) for r in range(w)]
      assert spans[0][0] == 0 and spans[-1][1] == n
      for a, b in zip(spans, spans[1:]):
        assert a[1] == b[0]
      sizes = [e - s for s, e in spans]
      assert max(sizes) - min(sizes) <= 1


def test_units_keep_kv_heads_with_their_query_group():
  q = torch.arange(2 * 8 * 3 * 4, dtype=torch.float32).reshape(2, 8, 3, 4)
  k = torch.arange(2 * 2 * 5 * 4, dtype=torch.float32).reshape(2, 2, 5, 4)
  qu, ku, vu = to_units(q, k, k)
  assert qu.shape == (4, 4, 3, 4) and ku.shape == (4, 1, 5, 4)
  # unit 3 = batch 1, kv head 1 -> query heads 4..7
  assert torch.equal(qu[3], q[1, 4:8]) and torch.equal(ku[3, 0], k[1, 1])
  ql, kl, _ = shard_units(q, k, k, world_size=2, rank=1)
  assert torch.equal(ql, qu[2:]) and torch.equal(kl, ku[2:])


def _free_port():
  with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    return s.getsockname()[1]


def _worker(rank, world, port, B, Hq, Hkv, ret):
  os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
  dist.init_process_group("gloo", rank=rank, world_size=world)
  from ffpa_attn_amd.sharding import sharded_attention

  torch.manual_seed(0)
  q = torch.randn(B, Hq, 40, 64)
  k = torch.randn(B, Hkv, 56, 64)
  v = torch.randn(B, Hkv, 56, 64)

  def local(qu, ku, vu):  # stand-in for the HIP kernel: same contract (dim 1 = GQA heads)
    return torch.nn.functional.scaled_dot_product_attention(qu, ku, vu, enable_gqa=True)

  out = sharded_attention(q, k, v, local, gather=True)
  ref = torch.nn.functional.scaled_dot_product_attention(q, k, v, enable_gqa=True)
  ok = torch.allclose(out, ref, atol=1e-5)
  part = sharded_attention(q, k, v, local, gather=False)
  ret[rank] = (bool(ok), tuple(part.shape))
  dist.barrier()
  dist.destroy_process_group()


def _born_worker(rank, world, port, n_units, g, ret):
  """The born-sharded path bench.py measures: per-unit seeded blocks, local attention, one all_gather."""
  os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
  os.environ.pop("FFPA_SHARDING_TRANSPORT", None)
  dist.init_process_group("gloo", rank=rank, world_size=world)
  from ffpa_attn_amd import sharding as sh

  s, e = sh.local_units(n_units)
  q, k, v = sh.synthetic_unit_block(s, e, g, 24, 40, 64, dtype=torch.float32, device="cpu", seed=3)
  o_local = torch.nn.functional.scaled_dot_product_attention(q, k, v, enable_gqa=True)  # stand-in for the HIP kernel
  full = sh.gather_units(o_local, n_units)
  # the same global problem generated in one piece: a unit's values do not depend on the partition
  qa, ka, va = sh.synthetic_unit_block(0, n_units, g, 24, 40, 64, dtype=torch.float32, device="cpu", seed=3)
  ref = torch.nn.functional.scaled_dot_product_attention(qa, ka, va, enable_gqa=True)
  ok = bool(torch.equal(full, ref))
  if n_units % world == 0:  # the overlapped form: pieces of the block gathered while the next piece computes
    import ffpa_attn_amd.sharding as shm
    real = shm.attend_units
    shm.attend_units = lambda a, b, c, **kw: torch.nn.functional.scaled_dot_product_attention(a, b, c, enable_gqa=True)
    # every piece goes point-to-point into its final slice of the result: no collective that would gather into a world x piece temporary and copy out
    # (the list form of dist.all_gather does exactly that for non-contiguous output views) — and a caller's buffer is filled in place
    real_ag, real_agt = dist.all_gather, dist.all_gather_into_tensor

    def forbidden(*a, **kw):
      raise AssertionError("attend_and_gather_units must not call a gather collective")

    dist.all_gather = dist.all_gather_into_tensor = forbidden
    try:
      st = {}
      for chunks in (1, 2, 3):
        ok = ok and bool(torch.equal(sh.attend_and_gather_units(q, k, v, n_units, chunks=chunks, stats=st), ref))
        ok = ok and st == {"transport": "p2p", "chunks": min(chunks, n_units // world), "world": world}  # (gloo has point-to-point: the probe says so on every rank)
      mine = torch.full_like(ref, float("nan"))
      got = sh.attend_and_gather_units(q, k, v, n_units, chunks=2, out=mine)
      ok = ok and got.data_ptr() == mine.data_ptr() and bool(torch.equal(mine, ref))
    finally:
      shm.attend_units = real
      dist.all_gather, dist.all_gather_into_tensor = real_ag, real_agt
    # the other transport (a node without RCCL point-to-point: gather_transport's probe fails on some rank -> every rank agrees on all_gather): the same
    # result through all_gather_into_tensor + the copy to the final slices, and NO point-to-point operation after the probe
    shm.attend_units = lambda a, b, c, **kw: torch.nn.functional.scaled_dot_product_attention(a, b, c, enable_gqa=True)
    shm._TRANSPORT.clear()
    real_batch = dist.batch_isend_irecv
    broke = {"n": 0}

    def failing(ops):
      broke["n"] += 1
      # (a node property: every rank's point-to-point raises; a rank that raised while the others posted theirs would leave them waiting — the probe
      # cannot cure that, FFPA_SHARDING_TRANSPORT=all_gather pins the fallback for such a node)
      raise RuntimeError("peer access is not available (stand-in)")

    dist.batch_isend_irecv = failing
    try:
      st = {}
      for chunks in (1, 3):
        ok = ok and bool(torch.equal(sh.attend_and_gather_units(q, k, v, n_units, chunks=chunks, stats=st), ref)) and st["transport"] == "all_gather"
      ok = ok and broke["n"] == 1  # (probed once per group, remembered)
      mine = torch.full_like(ref, float("nan"))
      ok = ok and bool(torch.equal(sh.attend_and_gather_units(q, k, v, n_units, chunks=2, out=mine), ref))
    finally:
      dist.batch_isend_irecv = real_batch
      shm.attend_units = real
      shm._TRANSPORT.clear()
  ret[rank] = (ok, (s, e), bool(torch.equal(q, qa[s:e])))
  dist.barrier()
  dist.destroy_process_group()


def test_two_rank_gloo_born_sharded_blocks_and_gather():
  for n_units, g in ((8, 2), (5, 1)):  # even: the collective writes straight into the result; uneven: padded + compacted
    port = _free_port()
    ret = mp.Manager().dict()
    mp.spawn(_born_worker, args=(2, port, n_units, g, ret), nprocs=2, join=True)
    assert ret[0][0] and ret[1][0] and ret[0][2] and ret[1][2]
    assert ret[0][1][0] == 0 and ret[0][1][1] == ret[1][1][0] and ret[1][1][1] == n_units


def test_four_rank_gloo_born_sharded_blocks_and_gather():
  """World size 4 (the scaling curve goes to 8): contiguous unit blocks, the overlapped gather in 1 / 2 / 3 pieces, an uneven split (6 units over 4 ranks)."""
  for n_units, g in ((8, 2), (6, 1)):
    port = _free_port()
    ret = mp.Manager().dict()
    mp.spawn(_born_worker, args=(4, port, n_units, g, ret), nprocs=4, join=True)
    assert all(ret[r][0] and ret[r][2] for r in range(4)), dict(ret)
    assert ret[0][1][0] == 0 and ret[3][1][1] == n_units and all(ret[r][1][1] == ret[r + 1][1][0] for r in range(3))


def test_two_rank_gloo_sharded_attention_matches_unsharded():
  for (B, Hq, Hkv) in ((2, 8, 2), (1, 6, 3)):  # even and uneven (3 units over 2 ranks) splits
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, B, Hq, Hkv, ret), nprocs=2, join=True)
    assert ret[0][0] and ret[1][0]
    n_units = B * Hkv
    assert ret[0][1][0] + ret[1][1][0] == n_units
