"""DecodeStep — the short-query (decode) step without the Python launch tax.

The reference's launcher issues a decode step's two kernels (split-KV stage 1 + LSE merge) from C++ in one call
(``csrc/cuffpa/native/launch.cuh:306-340``); here the step is launched from Python through ctypes, and on MI355X the host side of that
(~ 30 us of interpreter per call) is longer than the gap the GPU leaves between two 85 us steps: a decode loop that calls
``ffpa_attn_func`` per layer is host-bound (105 us per step against 83 + 4.5 us of kernels, ``profiles/r05_bench_decode*.json``).
What removes the host from the step is a HIP graph: the C-ABI allocates nothing, never synchronises and launches on the caller's stream,
so a call captures as it is.  ``DecodeStep`` ships that as product: the first call with a given set of tensors captures
``ffpa_attn_func(q, k, v, ...)`` — every kernel it launches: the mask scan, the split kernel, the merge — into a HIP graph; later calls
replay it (one ``hipGraphLaunch``).

What a replay reads and writes is fixed at capture time, so an entry is keyed by everything that fixes it: device, stream, and the
data pointer, shape, strides and dtype of every tensor argument.  Any change — another KV length, a reallocated cache, another stream —
is another key: it is captured on first use and kept (least recently used of ``max_graphs`` entries is dropped).  Tensors are NOT kept
alive by an entry: a graph reads addresses, and whichever tensor lives at a captured address with the captured shape is what the call
that matched the key passed in.  The returned tensor is the graph's own output buffer: the next replay of the SAME entry overwrites it
(``clone()`` it to keep it across steps — a decode loop consumes it at once).

Use it where shapes repeat: a static KV cache with the valid length expressed by a mask that lives on the device (the mask's CONTENT may
change between replays, its address may not), speculative-decoding verification, benchmarks.  A cache that grows by one key per token
is a new shape per token: every step would be a capture (~ 1 ms) — call ``ffpa_attn_func`` there, or keep the cache at its CAPACITY and the
length on the device::

    mask   = torch.zeros(1, 1, 1, capacity, dtype=torch.bool, device=dev)     # True = key holds a token
    bounds = torch.zeros(1, 1, 1, 4, dtype=torch.int32, device=dev)           # [first, end, free_lo, free_hi) of the mask (hip.mask_kv_bounds' layout)
    step   = DecodeStep(forward_backend=HIPBackend(forward=True, kv_bounds=bounds))
    ...
    mask[..., n] = True; bounds.copy_(torch.tensor([0, n + 1, 0, n + 1], ...))   # per token, in place, on the stream
    o = step(q, k_cache, v_cache, mask)                                          # ONE graph for every length

The split kernel's workgroups past ``end`` leave at once and keys inside ``[free_lo, free_hi)`` skip the mask read, so the bytes a step streams follow the
valid length, not the capacity (tests/test_host_path_gpu.py::test_decode_step_with_a_device_side_kv_length: bit-equal to the plain call on the same
arguments, equal to the call on the sliced cache to rounding, a step at 700 of 8192 keys several times cheaper than at 8192).
"""

from __future__ import annotations

import torch

from .interface import ffpa_attn_func

__all__ = ["DecodeStep"]


def _sig(t):
  return None if t is None else (t.data_ptr(), tuple(t.shape), tuple(t.stride()), t.dtype)


class _Entry:
  __slots__ = ("graph", "out", "replays")

  def __init__(self, graph, out):
    self.graph, self.out, self.replays = graph, out, 0


class DecodeStep:
  """``step = DecodeStep(is_causal=..., scale=..., enable_gqa=..., **backend_kw)``; ``o = step(q, k, v, attn_mask=None)`` — the same
  arguments, results (bit for bit) and errors as ``ffpa_attn_func``; inference only (no autograd graph is recorded).  See the module docstring."""

  def __init__(self, *, dropout_p: float = 0.0, is_causal: bool = False, scale: float | None = None, enable_gqa: bool = False,
               max_graphs: int = 16, **backend_kw):
    if dropout_p != 0.0:
      # (a replayed graph would replay the captured Philox offset: the same mask every step)
      raise ValueError("DecodeStep: dropout is not supported (a captured step would replay one dropout mask)")
    if max_graphs < 1:
      raise ValueError("DecodeStep: max_graphs must be >= 1")
    self._kw = dict(dropout_p=0.0, is_causal=is_causal, scale=scale, enable_gqa=enable_gqa, **backend_kw)
    self._max = int(max_graphs)
    self._entries: "dict[tuple, _Entry]" = {}
    self.captures = 0  # how many graphs this object has captured so far (a serving loop that sees this grow per token is holding it wrong)

  def _key(self, q, k, v, attn_mask):
    return (q.device.index, torch.cuda.current_stream(q.device).cuda_stream, _sig(q), _sig(k), _sig(v), _sig(attn_mask))

  def _capture(self, q, k, v, attn_mask) -> _Entry:
    cur = torch.cuda.current_stream(q.device)
    side = torch.cuda.Stream(device=q.device)
    side.wait_stream(cur)
    with torch.cuda.stream(side):
      # one eager call first: argument errors surface here as ffpa_attn_func raises them (not as a broken capture), the library is loaded, the
      # plan / scratch tables are filled — nothing inside the capture below touches the host allocator's non-graph pools
      ffpa_attn_func(q, k, v, attn_mask=attn_mask, **self._kw)
    cur.wait_stream(side)
    torch.cuda.synchronize(q.device)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
      out = ffpa_attn_func(q, k, v, attn_mask=attn_mask, **self._kw)
    self.captures += 1
    return _Entry(graph, out)

  @torch.no_grad()
  def __call__(self, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, attn_mask: torch.Tensor | None = None) -> torch.Tensor:
    if not q.is_cuda or torch.cuda.is_current_stream_capturing():
      # CPU tensors (the reference's CPU path is SDPA), or the caller is capturing a graph of its own: the plain call is what belongs there
      return ffpa_attn_func(q, k, v, attn_mask=attn_mask, **self._kw)
    key = self._key(q, k, v, attn_mask)
    e = self._entries.pop(key, None)
    if e is None:
      e = self._capture(q, k, v, attn_mask)
      while len(self._entries) >= self._max:
        self._entries.pop(next(iter(self._entries)))  # least recently used (dicts keep insertion order; hits are re-inserted below)
    self._entries[key] = e
    e.graph.replay()
    e.replays += 1
    return e.out

  def clear(self) -> None:
    """Drop every captured graph (their output buffers and scratch go back to the allocator)."""
    self._entries.clear()

  def __len__(self) -> int:
    return len(self._entries)
