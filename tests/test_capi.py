"""The C-ABI library loads on a CPU-only box and exports every symbol include/ffpa_attn.h declares;
argument validation returns the documented status codes BEFORE any device work (no GPU needed)."""

import ctypes
import os
import re

import pytest

from conftest import ROOT
from ffpa_attn_amd import hip


@pytest.fixture(scope="module")
def lib():
  if not hip.library_available():
    from ffpa_attn_amd import build

    build.build()
  return hip.load_library()


def _declared_functions():
  text = open(os.path.join(ROOT, "include", "ffpa_attn.h")).read()
  text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
  return sorted(set(re.findall(r"\b(ffpa_attn_\w+)\s*\(", text)))


def test_every_declared_symbol_is_exported(lib):
  names = _declared_functions()
  assert set(names) == set(hip.EXPORTS)
  for n in names:
    assert getattr(lib, n) is not None


def test_struct_layout_matches_header(lib):
  # the library checks struct_size itself; a mismatch would surface as FFPA_ERR_BAD_ABI (10)
  p = hip.FfpaFwdParams()
  p.struct_size = ctypes.sizeof(hip.FfpaFwdParams) - 8
  p.abi_version = hip.ABI_VERSION
  assert lib.ffpa_attn_fwd(ctypes.byref(p), None) == 10
  assert b"ABI mismatch" in lib.ffpa_attn_last_error()
  assert ctypes.sizeof(hip.FfpaFwdParams) == 312  # (ABI 4: + split_tickets)


def test_ctypes_mirror_matches_the_c_header(tmp_path):
  """Compile the header with gcc (plain C) and compare size / field offsets with the ctypes mirror."""
  import subprocess

  fields = [f[0] for f in hip.FfpaFwdParams._fields_]
  src = tmp_path / "layout.c"
  body = "".join(f'printf("{f} %zu\\n", offsetof(ffpa_fwd_params, {f}));\n' for f in fields)
  src.write_text(
    '#include <stdio.h>\n#include <stddef.h>\n#include "ffpa_attn.h"\nint main(void){\n'
    'printf("sizeof %zu\\n", sizeof(ffpa_fwd_params));\n' + body + "return 0;}\n"
  )
  exe = tmp_path / "layout"
  subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
  out = dict(line.split() for line in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines())
  assert int(out["sizeof"]) == ctypes.sizeof(hip.FfpaFwdParams)
  for f in fields:
    assert int(out[f]) == getattr(hip.FfpaFwdParams, f).offset, f


def test_queries(lib):
  assert lib.ffpa_attn_query(0) == hip.ABI_VERSION
  assert lib.ffpa_attn_query(1) == 1
  assert (lib.ffpa_attn_query(2), lib.ffpa_attn_query(3), lib.ffpa_attn_query(4)) == (8, 1024, 8)  # min / max head dim, multiple
  assert lib.ffpa_attn_query(5) == 1 and lib.ffpa_attn_query(6) == 1
  assert lib.ffpa_attn_query(99) == -1
  assert lib.ffpa_attn_version().startswith(b"ffpa-attn-amd")
  assert lib.ffpa_attn_fwd_workspace_bytes(None) == 0


def test_short_query_plan_and_workspace(lib):
  """<= 32 query rows: short-query tiles + KV splits sized by the caller's scratch (no GPU needed for the plan)."""
  plan = (ctypes.c_int * 4)()
  p = _params(seqlen_q=1, seqlen_kv=8192, heads_q=4, heads_kv=4)
  assert lib.ffpa_attn_fwd_plan(ctypes.byref(p), plan) == 0
  assert list(plan) == [1, 32, 64, 1]  # no workspace -> no split (D = 512: 64-key short-query tiles)
  need = lib.ffpa_attn_fwd_workspace_bytes(ctypes.byref(p))
  assert need > 0 and need % (4 * 1 * 4 * 1 * (512 + 1)) == 0
  p.workspace, p.workspace_bytes = 16, need
  assert lib.ffpa_attn_fwd_plan(ctypes.byref(p), plan) == 0
  splits = plan[3]
  assert splits > 1 and need == splits * 4 * 1 * 4 * 1 * (512 + 1)
  assert -(-8192 // 64) // splits >= 4  # at least 4 KV tiles per split
  p.num_splits = 1
  assert lib.ffpa_attn_fwd_plan(ctypes.byref(p), plan) == 0 and plan[3] == 1
  p.num_splits = 3
  assert lib.ffpa_attn_fwd_plan(ctypes.byref(p), plan) == 0 and 1 < plan[3] <= 3
  big = _params(seqlen_q=4096, seqlen_kv=4096, heads_q=32, heads_kv=8)   # 1024 workgroups: fills the chip
  assert lib.ffpa_attn_fwd_plan(ctypes.byref(big), plan) == 0 and list(plan) == [0, 128, 64, 1]
  assert lib.ffpa_attn_fwd_workspace_bytes(ctypes.byref(big)) == 0
  d1024 = _params(seqlen_q=4096, seqlen_kv=4096, head_dim=1024, heads_q=32, heads_kv=8)
  assert lib.ffpa_attn_fwd_plan(ctypes.byref(d1024), plan) == 0 and list(plan) == [0, 64, 32, 1]


@pytest.mark.parametrize("over, want", [
  (dict(batch=1, heads_q=32, heads_kv=32, seqlen_kv=8192, head_dim=512), (64, 8)),     # 32 row tiles: 256 CUs / 32 = 8 splits (one workgroup per CU), 64-key tiles
  (dict(batch=1, heads_q=32, heads_kv=32, seqlen_kv=8192, head_dim=1024), (32, 8)),    # above D = 512 only one workgroup fits a CU: 32-key tiles, 8 splits
  (dict(batch=3, heads_q=20, heads_kv=20, seqlen_kv=10000, head_dim=512), (64, 4)),    # 60 row tiles: rounded DOWN (5 splits = 300 workgroups would take two rounds)
  (dict(batch=8, heads_q=8, heads_kv=8, seqlen_kv=8192, head_dim=512), (64, 4)),       # 64 row tiles
  (dict(batch=4, heads_q=32, heads_kv=32, seqlen_kv=8192, head_dim=128), (64, 4)),     # small head dims: two workgroups per CU (128 row tiles x 4 = 512), 64-key tiles
  (dict(batch=1, heads_q=64, heads_kv=64, seqlen_kv=8192, head_dim=256), (32, 8)),     # D = 256 keeps 32-key tiles (two workgroups per CU)
      (dict(batch=1, heads_q=32, heads_kv=32, seqlen_kv=8192, head_dim=320), (64, 8)),     # two-wave split (D % 128 != 0): 64-key tiles since round 4 (D = 320 / 448), one workgroup per CU
      (dict(batch=1, heads_q=32, heads_kv=32, seqlen_kv=8192, head_dim=576), (32, 8)),     # ... 32-key tiles where the LDS has no room for 64 (D >= 576)
  (dict(batch=16, heads_q=32, heads_kv=32, seqlen_kv=4096, head_dim=512), (64, 1)),    # 512 row tiles fill the chip: no split
])
def test_short_query_split_rule(lib, over, want):
  """The KV-split count of short-query launches on a 256-CU device (the plan needs no GPU: the CU count falls back to 256): one workgroup per CU
  for head dims >= 320 — rounded down —, two below; tile sizes by head dim (profiles/r03_decode_splits.txt)."""
  plan = (ctypes.c_int * 4)()
  p = _params(seqlen_q=1, **over)
  p.workspace, p.workspace_bytes = 16, 1 << 40
  assert lib.ffpa_attn_fwd_plan(ctypes.byref(p), plan) == 0
  rows = 32 if over["head_dim"] % 128 == 0 else 64  # (head dims that are not multiples of 128: two row blocks x two D-halves)
  assert (plan[0], plan[1]) == (1, rows) and (plan[2], plan[3]) == want, list(plan)


def test_underfilled_prefill_plan_splits_the_kv_axis(lib):
  """Prefill tiles whose launch would leave more than half of the CUs idle (chunked prefill against a long
  context with few heads) split the KV axis too: >= 8 KV tiles per split, one workgroup per CU in total
  (256 CUs are assumed when no device is present)."""
  plan = (ctypes.c_int * 4)()
  p = _params(seqlen_q=512, seqlen_kv=65536, heads_q=8, heads_kv=8)        # 8 heads x 4 row tiles = 32 workgroups
  assert lib.ffpa_attn_fwd_plan(ctypes.byref(p), plan) == 0 and list(plan) == [0, 128, 64, 1]   # no scratch
  need = lib.ffpa_attn_fwd_workspace_bytes(ctypes.byref(p))
  assert need == 8 * (1 * 8 * 512 * (512 + 1) * 4)
  p.workspace, p.workspace_bytes = 16, need
  assert lib.ffpa_attn_fwd_plan(ctypes.byref(p), plan) == 0 and list(plan) == [0, 128, 64, 8]
  p.workspace_bytes = need // 2                                             # less scratch -> fewer splits
  assert lib.ffpa_attn_fwd_plan(ctypes.byref(p), plan) == 0 and plan[3] == 4
  short_ctx = _params(seqlen_q=512, seqlen_kv=1024, heads_q=8, heads_kv=8)  # 16 KV tiles: at most 2 splits
  short_ctx.workspace, short_ctx.workspace_bytes = 16, 1 << 40
  assert lib.ffpa_attn_fwd_plan(ctypes.byref(short_ctx), plan) == 0 and plan[3] == 2
  half = _params(seqlen_q=4096, seqlen_kv=4096, heads_q=4, heads_kv=4)      # 128 workgroups = half the chip
  half.workspace, half.workspace_bytes = 16, 1 << 40
  assert lib.ffpa_attn_fwd_plan(ctypes.byref(half), plan) == 0 and plan[3] == 2
  full = _params(seqlen_q=4096, seqlen_kv=4096, heads_q=8, heads_kv=8)      # 256 workgroups: no split
  full.workspace, full.workspace_bytes = 16, 1 << 40
  assert lib.ffpa_attn_fwd_plan(ctypes.byref(full), plan) == 0 and plan[3] == 1


@pytest.mark.parametrize("over, want", [
  (dict(heads_q=9, heads_kv=9, seqlen_q=4096, seqlen_kv=8192), 3),      # 288 workgroups = 1.125 rounds of 256 CUs -> 864 = 3.4 rounds (measured + 18 %)
  (dict(heads_q=10, heads_kv=10, seqlen_q=4096, seqlen_kv=8192), 3),    # 320 -> + 15 %
  (dict(heads_q=11, heads_kv=11, seqlen_q=4096, seqlen_kv=8192), 2),    # 352 -> 704 = 2.75 rounds (+ 10 %; three splits measured slower)
  (dict(heads_q=12, heads_kv=12, seqlen_q=4096, seqlen_kv=8192), 2),    # 384 -> three whole rounds (+ 5 %)
  (dict(heads_q=40, heads_kv=40, seqlen_q=1024, seqlen_kv=8192), 3),    # 320 workgroups of a short-query-axis launch (+ 14 %)
  (dict(heads_q=5, heads_kv=5, seqlen_q=4096, seqlen_kv=8192, head_dim=1024), 3),  # D = 1024: 64-row tiles, 320 workgroups (+ 19 %)
  (dict(heads_q=12, heads_kv=12, seqlen_q=4096, seqlen_kv=2048), 1),    # short context: the partials and their merge would cost more than the ragged round (measured - 18 %)
  (dict(heads_q=12, heads_kv=12, seqlen_q=4096, seqlen_kv=4096, causal=1), 1),  # causal: half the keys per row tile (measured - 37 %)
  (dict(heads_q=17, heads_kv=17, seqlen_q=4096, seqlen_kv=8192), 1),    # 2.1 rounds: a predicted 4 % is under the rule's 10 %
  (dict(heads_q=20, heads_kv=20, seqlen_q=4096, seqlen_kv=8192), 1),    # 2.5 rounds
  (dict(heads_q=32, heads_kv=32, seqlen_q=1024, seqlen_kv=8192), 1),    # exactly one round (`cross`): splits measured - 14 %
  (dict(heads_q=32, heads_kv=32, seqlen_q=8192, seqlen_kv=8192), 1),    # the headline shape: 8 whole rounds
  # part of one round (CUs / 2 < workgroups < CUs)
  (dict(heads_q=5, heads_kv=5, seqlen_q=4096, seqlen_kv=8192), 3),      # 160 workgroups -> 480 = two rounds of a third (measured + 10 %)
  (dict(heads_q=5, heads_kv=5, seqlen_q=4096, seqlen_kv=16384), 3),     # ... + 19 % on 16384 keys
  (dict(heads_q=20, heads_kv=20, seqlen_q=1024, seqlen_kv=8192), 3),    # 160 workgroups of a short query axis (+ 8 %)
  (dict(heads_q=5, heads_kv=5, seqlen_q=4096, seqlen_kv=8192, head_dim=320), 3),  # + 7 %
  (dict(heads_q=6, heads_kv=6, seqlen_q=4096, seqlen_kv=8192), 1),      # 192: no split count fills whole rounds (best measured arm - 9 %)
  (dict(heads_q=7, heads_kv=7, seqlen_q=4096, seqlen_kv=8192), 1),      # 224
  (dict(heads_q=3, heads_kv=3, seqlen_q=4096, seqlen_kv=8192, head_dim=1024), 1),  # 192 workgroups of 64 rows
  (dict(heads_q=5, heads_kv=5, seqlen_q=4096, seqlen_kv=2048), 1),      # short context (measured - 20 %)
  (dict(heads_q=5, heads_kv=5, seqlen_q=4096, seqlen_kv=4096, causal=1, causal_offset=0), 2),  # (uniform ranges lose under the causal flag; PER-ROW-TILE ranges: + 33 % — test_causal_tile_range_rule)
  # under-filled (workgroups <= CUs / 2): CUs / workgroups splits fill one round; a count that makes two rounds of shorter workgroups is taken at 5 % predicted
  (dict(heads_q=3, heads_kv=3, seqlen_q=4096, seqlen_kv=8192), 2),      # 96 workgroups: 2 x 96 = 192 of 256 CUs; five ranges measured + 2 % — inside the margin
  (dict(heads_q=3, heads_kv=3, seqlen_q=4096, seqlen_kv=16384), 5),     # ... on 16384 keys 5 x 96 = 480 = two rounds of a fifth: + 6 ... 8 %
  (dict(heads_q=12, heads_kv=12, seqlen_q=1024, seqlen_kv=16384), 5),   # + 11 %
  (dict(heads_q=3, heads_kv=3, seqlen_q=2048, seqlen_kv=8192, head_dim=1024), 5),  # + 7 %
  (dict(heads_q=5, heads_kv=5, seqlen_q=2048, seqlen_kv=8192), 3),      # 80 x 3 = 240: the one-round count is the fastest measured arm
  (dict(heads_q=7, heads_kv=7, seqlen_q=1024, seqlen_kv=8192), 4),      # 56 x 4 = 224: likewise
  (dict(heads_q=4, heads_kv=4, seqlen_q=4096, seqlen_kv=8192), 2),      # 128 x 2 = 256: exact
])
def test_ragged_round_split_rule(lib, over, want):
  """Prefill launches of a little over one round of workgroups (1 < workgroups / CUs <= 1.5) or of part of one (0.5 < ... < 1) split the KV axis in 2 or 3 when the cost model of
  ffpa_capi.hip predicts >= 10 % (profiles/r04_launch_side.txt); nothing else that fills the chip splits.  256-CU fallback: no GPU needed."""
  plan = (ctypes.c_int * 4)()
  p = _params(**over)
  p.workspace, p.workspace_bytes = 16, 1 << 40
  assert lib.ffpa_attn_fwd_plan(ctypes.byref(p), plan) == 0
  assert plan[0] == 0 and plan[3] == want, list(plan)
  if want > 1:
    p.workspace, p.workspace_bytes = None, 0  # a caller without scratch: the plain launch
    assert lib.ffpa_attn_fwd_plan(ctypes.byref(p), plan) == 0 and plan[3] == 1
    p.workspace, p.workspace_bytes = 16, 1 << 40
    p.num_splits = 1                           # ... or one that forbids splits
    assert lib.ffpa_attn_fwd_plan(ctypes.byref(p), plan) == 0 and plan[3] == 1


@pytest.mark.parametrize("over, rows", [
  (dict(batch=2, heads_q=32, heads_kv=8, seqlen_q=8192, seqlen_kv=2048, head_dim=320), 192),             # config 4 without its mask: 11 rounds of 192 rows < 16 of 128
  (dict(batch=2, heads_q=32, heads_kv=8, seqlen_q=8192, seqlen_kv=2048, head_dim=320, causal=1), 192),   # config 4 (causal flag): 10.75 workgroups per CU
  (dict(batch=2, heads_q=32, heads_kv=8, seqlen_q=8192, seqlen_kv=2048, head_dim=296, causal=1), 192),   # a head dim served by the D = 320 object
  (dict(batch=1, heads_q=32, heads_kv=32, seqlen_q=8192, seqlen_kv=8192, head_dim=320), 128),            # 1376 workgroups = 5.4 -> six rounds of 192 > 8 of 128
  (dict(batch=1, heads_q=32, heads_kv=32, seqlen_q=8192, seqlen_kv=8192, head_dim=320, causal=1), 128),  # ragged, but only 5.4 workgroups per CU
  (dict(batch=2, heads_q=32, heads_kv=8, seqlen_q=8192, seqlen_kv=2048, head_dim=320, dropout_p=0.1), 128),  # no dropout build
  (dict(batch=2, heads_q=32, heads_kv=8, seqlen_q=8192, seqlen_kv=2048, head_dim=512), 128),             # no wide tile at this head dim
  (dict(batch=2, heads_q=32, heads_kv=8, seqlen_q=8192, seqlen_kv=2048, head_dim=256), 128),
])
def test_wide_tile_rule(lib, over, rows):
  """Which prefill launches take the wide-row tile (ffpa_fwd_m16w_kernel: D = 320, 192 rows per workgroup, 64-key tiles): the pricing of
  ffpa_capi.hip::pick_wide_tile (profiles/r05_wide_tile.txt); FFPA_FLAG_WIDE_TILE / _NO_WIDE_TILE override it where a build exists.  256-CU fallback: no GPU needed."""
  plan = (ctypes.c_int * 4)()
  name = ctypes.create_string_buffer(160)
  p = _params(**over)
  assert lib.ffpa_attn_fwd_plan(ctypes.byref(p), plan) == 0
  assert plan[1] == rows and plan[2] == (64 if rows == 192 else plan[2]), list(plan)
  assert lib.ffpa_attn_fwd_kernel(ctypes.byref(p), name, len(name)) == 0
  assert name.value.decode().startswith("ffpa_fwd_m16w_kernel<bf16, 320, RH=3, MK=0>" if rows == 192 else "ffpa_fwd_m16_kernel<"), name.value
  p.flags = hip.FLAG_NO_WIDE_TILE
  assert lib.ffpa_attn_fwd_plan(ctypes.byref(p), plan) == 0 and plan[1] == 128
  p.flags = hip.FLAG_WIDE_TILE
  has_build = (over["head_dim"] + 63) // 64 * 64 == 320 and not over.get("dropout_p")
  assert lib.ffpa_attn_fwd_plan(ctypes.byref(p), plan) == 0 and plan[1] == (192 if has_build else 128)


def test_causal_pricing_uses_the_keys_rows_can_see(lib):
  """The split pricing walks the KV tiles an average row tile can SEE under the causal flag (causal_offset + Nq / 2 keys, clamped), and no split range may lie
  entirely behind every row's diagonal.  Top-left causal against a long context (rows see at most Nq keys): a ragged round that used to be cut into three
  KV ranges — two of them workgroups that did nothing, plus 240 MiB of partials — is one plain launch; tail-aligned causal on the same shape sees the whole
  context and still splits."""
  plan = (ctypes.c_int * 4)()
  p = _params(heads_q=10, heads_kv=10, seqlen_q=4096, seqlen_kv=32768, causal=1, causal_offset=0)
  p.workspace, p.workspace_bytes = 16, 1 << 40
  assert lib.ffpa_attn_fwd_plan(ctypes.byref(p), plan) == 0 and plan[3] == 1, list(plan)
  p.causal_offset = 32768 - 4096
  assert lib.ffpa_attn_fwd_plan(ctypes.byref(p), plan) == 0 and plan[3] > 1, list(plan)
  # an under-filled launch (96 workgroups) under a top-left mask that hides all but the first eighth of the context: one range at most per visible eighth
  p = _params(heads_q=3, heads_kv=3, seqlen_q=4096, seqlen_kv=32768, causal=1, causal_offset=0)
  p.workspace, p.workspace_bytes = 16, 1 << 40
  assert lib.ffpa_attn_fwd_plan(ctypes.byref(p), plan) == 0
  tiles = (32768 + plan[2] - 1) // plan[2]
  per = (tiles + plan[3] - 1) // plan[3]
  assert (plan[3] - 1) * per * plan[2] < 4096, list(plan)  # the last range starts inside the visible keys


def test_plan_prices_the_device_it_runs_on(lib, monkeypatch):
  """CU count, engine clock and HBM bandwidth come from the device (MI355X figures without one); FFPA_HIP_FAKE_CUS (test-only) swaps the CU count in.
  The same launch is a ragged round on 256 CUs, whole rounds on 128 / 320 and under-filled on 1024."""
  assert lib.ffpa_attn_query(8) == 256 and lib.ffpa_attn_query(9) == 2400 and lib.ffpa_attn_query(10) == 8000  # (no GPU here: the fallbacks)
  plan = (ctypes.c_int * 4)()
  p = _params(heads_q=10, heads_kv=10, seqlen_q=4096, seqlen_kv=8192)  # 320 workgroups
  p.workspace, p.workspace_bytes = 16, 1 << 40
  got = {}
  for cus in (128, 256, 320, 1024):
    monkeypatch.setenv("FFPA_HIP_FAKE_CUS", str(cus))
    assert lib.ffpa_attn_query(8) == cus
    assert lib.ffpa_attn_fwd_plan(ctypes.byref(p), plan) == 0
    got[cus] = plan[3]
  assert got[256] == 3 and got[128] == 1 and got[320] == 1 and got[1024] == 3, got  # (1024: 320 workgroups x 3 = 960, one round)


def test_tile_configs():
  for d in range(64, 1025, 64):
    c = hip.tile_config(d)
    if d <= 512:
      # 128-key tiles at D = 64 (32x32x16 kernel) and at D = 256 / 320 (16x16x32 kernel: worth their 64 score registers per lane only there;
      # above 320 K + V of a 128-key tile would not fit the 160 KiB of LDS), 64-key tiles elsewhere
      bc = 128 if d in (64, 256, 320) else 64
      assert (c["block_rows"], c["block_keys"]) == (128, bc) and c["lds_bytes"] == 2 * bc * d * 2
    else:
      # split-D tiles: K + V images + the partial-S exchange (6 KiB per wave since round 4: key block 0 of the next tile is published one
      # step early into a double-buffered half — the softmax pipeline) + 5 KiB since round 5 (the two waves of a row block share the softmax by rows
      # and trade P^T fragments and per-row scalars)
      assert (c["block_rows"], c["block_keys"]) == (64, 32) and c["lds_bytes"] == 2 * 32 * d * 2 + 4 * 6144 + 4 * 1024 + 4 * 256
    assert c["lds_bytes"] <= 160 * 1024  # one CU's LDS
  with pytest.raises(RuntimeError, match="headdim not support"):
    hip.tile_config(100)


def _params(**over):
  buf = (ctypes.c_char * 4096)()
  base = (ctypes.addressof(buf) + 15) & ~15
  p = hip.FfpaFwdParams()
  p.struct_size = ctypes.sizeof(hip.FfpaFwdParams)
  p.abi_version = hip.ABI_VERSION
  p.q = p.k = p.v = p.o = base
  p.batch, p.heads_q, p.heads_kv, p.seqlen_q, p.seqlen_kv, p.head_dim = 1, 4, 2, 128, 128, 512
  for n in ("q_stride", "k_stride", "v_stride", "o_stride"):
    getattr(p, n)[:] = [4 * 128 * 512, 128 * 512, 512]
  p.softmax_scale = 0.04
  p.rescale_threshold = -1.0
  for k, v in over.items():
    if isinstance(v, (list, tuple)):
      getattr(p, k)[:] = v
    else:
      setattr(p, k, v)
  p._keepalive = buf
  return p


@pytest.mark.parametrize(
  "over, status, text",
  [
    ({"q": None}, 1, b"non-NULL"),
    ({"dtype": 7}, 2, b"dtype"),
    ({"head_dim": 100}, 3, b"headdim not support"),
    ({"heads_q": 5}, 4, b"num_heads"),
    ({"seqlen_kv": 0}, 4, b"non-positive"),
    ({"k_stride": [4 * 128 * 512, 128 * 512, 516]}, 5, b"multiple of 8"),
    ({"v_stride": [4 * 128 * 512, 128 * 512, 256]}, 5, b"rows must not overlap"),
    ({"k_stride": [4 * 128 * 512, 128 * 512, 1 << 24]}, 5, b"2^24 elements"),
    ({"bias_dtype": 2}, 2, b"bias pointer and bias_dtype disagree"),
    ({"dropout_p": 1.0}, 4, b"dropout_p"),
    ({"dropout_p": 0.1, "causal_row_mod": 2}, 7, b"packed query heads"),
    ({"kv_bounds": 64, "causal_row_mod": 2}, 7, b"kv_bounds with packed"),
    ({"softmax_scale": float("nan")}, 4, b"not finite"),
  ],
)
def test_validation_status_codes(lib, over, status, text):
  p = _params(**over)
  assert lib.ffpa_attn_fwd(ctypes.byref(p), None) == status
  assert text in lib.ffpa_attn_last_error()


def test_misaligned_pointer(lib):
  p = _params()
  p.k = p.k + 2
  assert lib.ffpa_attn_fwd(ctypes.byref(p), None) == 6


def test_no_gfx950_device_is_a_status_not_a_launch(lib):
  """Valid parameters on a box without a gfx950 (this CPU container): FFPA_ERR_NO_DEVICE, nothing is launched.  (On an
  MI355X the same call would launch on the host buffer's address, so the test only runs where there is no GPU.)"""
  import torch

  if torch.cuda.is_available():
    pytest.skip("a GPU is present: the call would launch")
  p = _params()
  assert lib.ffpa_attn_fwd(ctypes.byref(p), None) == 9
  assert b"device" in lib.ffpa_attn_last_error()
  # boolean masks (FFPA_BIAS_BOOL8 = 4) are a known bias dtype; 5 is not
  buf = p._keepalive
  q = _params(bias=ctypes.addressof(buf), bias_dtype=4)
  assert lib.ffpa_attn_fwd(ctypes.byref(q), None) == 9
  q = _params(bias=ctypes.addressof(buf), bias_dtype=5)
  assert lib.ffpa_attn_fwd(ctypes.byref(q), None) == 2


def test_integration_md_binding_mirrors_the_header():
  """INTEGRATION.md shows the ctypes stub a maintainer of the reference would add: its struct must be the header's (same fields, same order, same size
  as the ctypes mirror the tests validate against gcc above), and it must pass the ABI version the header defines."""
  text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
  m = re.search(r"class _Params\(ctypes\.Structure\):.*?\n(\s+_fields_ = \[.*?\])\s*(#[^\n]*)?\n\n", text, flags=re.S)
  assert m, "the _Params struct of the binding was not found in INTEGRATION.md"
  ns = {"ctypes": ctypes}
  exec("class _Params(ctypes.Structure):\n" + m.group(1), ns)
  doc = ns["_Params"]
  assert [f[0] for f in doc._fields_] == [f[0] for f in hip.FfpaFwdParams._fields_]
  assert ctypes.sizeof(doc) == ctypes.sizeof(hip.FfpaFwdParams)
  assert f"abi_version={hip.ABI_VERSION}," in text


def test_plan_invariants_on_random_launches(lib, monkeypatch):
  """3000 seeded random launches x three CU counts (the plan needs no GPU): whatever make_plan picks, the answers of the three query entry points
  agree with each other and with the documented contract — the scratch is exactly splits x B x Hq x Nq x (built D + 1) fp32 (0 without splits) and stays under
  the 1 GiB cap for automatic prefill splits, `num_splits = 1` and a missing scratch mean one range, a requested count is a ceiling, every split keeps
  whole KV tiles, tiles come from the built set, the wide-row tile only where it exists (D in (256, 320], no additive bias, no dropout) and never
  against FFPA_FLAG_NO_WIDE_TILE, and the kernel name says what the plan says."""
  import random

  rng = random.Random(20250927)
  plan = (ctypes.c_int * 4)()
  name = ctypes.create_string_buffer(160)
  dims = [64 * i for i in range(1, 17)] + [72, 264, 300, 520, 1000]
  for cus in ("", "128", "304"):
    if cus:
      monkeypatch.setenv("FFPA_HIP_FAKE_CUS", cus)
    else:
      monkeypatch.delenv("FFPA_HIP_FAKE_CUS", raising=False)
    for _ in range(1000):
      D = rng.choice(dims)
      group = rng.choice([1, 1, 2, 4, 8])
      hkv = rng.choice([1, 2, 3, 8, 9, 32])
      B = rng.choice([1, 1, 2, 3, 8])
      nq = rng.choice([1, 2, 7, 8, 31, 32, 33, 100, 512, 1000, 2048, 4096, 8192])
      nkv = rng.choice([1, 63, 64, 500, 1024, 2048, 8192, 10000, 32768])
      causal = rng.random() < 0.3
      bias = rng.choice([0, 0, 0, 1, 3, 4])  # none / fp16 / fp32 / bool
      drop = 0.1 if rng.random() < 0.15 else 0.0
      flags = rng.choice([0, 0, 0, hip.FLAG_WIDE_TILE, hip.FLAG_NO_WIDE_TILE])
      req = rng.choice([0, 0, 0, 1, 2, 5])
      Dp = (D + 7) // 8 * 8
      p = _params(batch=B, heads_q=hkv * group, heads_kv=hkv, seqlen_q=nq, seqlen_kv=nkv, head_dim=Dp, causal=int(causal), causal_offset=max(0, nkv - nq),
                  dropout_p=drop, flags=flags, num_splits=req)
      for n, rows in (("q_stride", nq), ("o_stride", nq), ("k_stride", nkv), ("v_stride", nkv)):
        heads = hkv * group if n[0] in "qo" else hkv
        getattr(p, n)[:] = [heads * rows * Dp, rows * Dp, Dp]
      if bias:
        p.bias, p.bias_dtype = p.q, bias
        p.bias_stride[:] = [0, 0, rng.choice([0, nkv]), 1]
      what = dict(cus=cus, D=D, B=B, hq=hkv * group, hkv=hkv, nq=nq, nkv=nkv, causal=causal, bias=bias, drop=drop, flags=hex(flags), req=req)
      # without scratch: one range
      assert lib.ffpa_attn_fwd_plan(ctypes.byref(p), plan) == 0, (what, lib.ffpa_attn_last_error())
      assert plan[3] == 1, (what, list(plan))
      need = lib.ffpa_attn_fwd_workspace_bytes(ctypes.byref(p))
      p.workspace, p.workspace_bytes = 16, 1 << 42
      assert lib.ffpa_attn_fwd_plan(ctypes.byref(p), plan) == 0, what
      variant, br, bc, splits = plan
      assert variant in (0, 1) and br in (32, 64, 128, 192) and bc in (32, 64, 128) and splits >= 1, (what, list(plan))
      assert (variant == 1) == (nq <= 32), (what, list(plan))
      per_split = 4 * B * hkv * group * nq * ((Dp + 63) // 64 * 64 + 1)  # (partials in the BUILT head dim: the next multiple of 64)
      assert need == (splits * per_split if splits > 1 else 0), (what, list(plan), need)
      if req == 1:
        assert splits == 1, (what, list(plan))
      elif req > 1:
        assert splits <= req, (what, list(plan))
      if splits > 1:
        assert -(-nkv // bc) >= splits, (what, list(plan))  # every range holds at least one KV tile
        if variant == 0 and req == 0:
          assert need <= 1 << 30, (what, need)
      assert lib.ffpa_attn_fwd_kernel(ctypes.byref(p), name, len(name)) == 0, what
      kernel = name.value.decode()
      wide = kernel.startswith("ffpa_fwd_m16w_kernel")
      assert wide == (br == 192), (what, kernel, list(plan))
      if wide:
        assert 256 < Dp <= 320 and bias in (0, 4) and drop == 0.0 and not (flags & hip.FLAG_NO_WIDE_TILE), (what, kernel)
      if variant == 1:
        assert kernel.startswith("ffpa_fwd_split_d_kernel"), (what, kernel)
      assert f" {(Dp + 63) // 64 * 64}," in kernel or f" {(Dp + 63) // 64 * 64}>" in kernel, (what, kernel)


def test_deterministic_flag_pins_the_plan_of_a_slice(lib):
  """FFPA_FLAG_DETERMINISTIC (Python: FFPA_HIP_DETERMINISTIC=1): the plan of a (batch, head) slice does not depend on how many slices share the launch —
  prefill launches never split the KV axis and never take the wide-row tile; short-query launches split by the KV length alone."""
  plan = (ctypes.c_int * 4)()

  def ask(flags=0, **over):
    p = _params(**over)
    p.workspace, p.workspace_bytes = 16, 1 << 40
    p.flags = flags
    assert lib.ffpa_attn_fwd_plan(ctypes.byref(p), plan) == 0
    return list(plan)

  det = hip.FLAG_DETERMINISTIC
  # prefill: the launch-size rules (under-filled: H2 -> 4 ranges; ragged round: H9 x Nq 4096 -> 3; wide tile: config 4) all depend on the head / batch count
  assert ask(batch=1, heads_q=2, heads_kv=2, seqlen_q=4096, seqlen_kv=16384)[3] > 1 and ask(batch=1, heads_q=9, heads_kv=9, seqlen_q=4096, seqlen_kv=8192)[3] > 1
  assert ask(batch=2, heads_q=32, heads_kv=8, seqlen_q=8192, seqlen_kv=2048, head_dim=320, causal=1)[1] == 192
  for heads in (1, 2, 5, 9, 32):
    assert ask(det, batch=1, heads_q=heads, heads_kv=heads, seqlen_q=4096, seqlen_kv=16384) == [0, 128, 64, 1]
    assert ask(det, batch=2, heads_q=4 * heads, heads_kv=heads, seqlen_q=8192, seqlen_kv=2048, head_dim=320, causal=1) == [0, 128, 128, 1]
  # short query: the default rule aims at a workgroup count (so the split count moves with batch x heads); pinned, it is a function of the KV length
  assert ask(batch=1, heads_q=32, heads_kv=32, seqlen_q=1, seqlen_kv=8192)[3] != ask(batch=8, heads_q=8, heads_kv=8, seqlen_q=1, seqlen_kv=8192)[3]
  for nkv, want in ((8192, 8), (4096, 4), (1000, 1), (140000, 137)):  # 16 tiles of 64 keys per range
    got = {tuple(ask(det, batch=b, heads_q=h, heads_kv=h, seqlen_q=1, seqlen_kv=nkv)) for b, h in ((1, 1), (1, 32), (8, 8), (16, 32))}
    assert got == {(1, 32, 64, want)}, (nkv, got)
  assert ask(det, batch=1, heads_q=32, heads_kv=32, seqlen_q=1, seqlen_kv=8192, num_splits=1)[3] == 1  # (an explicit "never split" still wins)
  assert hip.FLAG_DETERMINISTIC == 0x4000


@pytest.mark.parametrize("over, paired", [
  (dict(batch=1, heads_q=32, heads_kv=32, seqlen_q=4096, seqlen_kv=4096, causal=1), True),      # 32 row tiles per head
  (dict(batch=4, heads_q=32, heads_kv=32, seqlen_q=2048, seqlen_kv=2048, causal=1), True),      # 16
  (dict(batch=1, heads_q=32, heads_kv=32, seqlen_q=8192, seqlen_kv=8192, causal=1), False),     # 64 row tiles: measured neutral (- 1 ... + 2.4 %): not taken
  (dict(batch=1, heads_q=32, heads_kv=32, seqlen_q=12288, seqlen_kv=12288, causal=1), False),   # 96 row tiles: two rounds of long workgroups lose
  (dict(batch=1, heads_q=32, heads_kv=32, seqlen_q=8192, seqlen_kv=8192, head_dim=1024, causal=1), False),  # 64-row tiles: 128 of them
  (dict(batch=2, heads_q=32, heads_kv=32, seqlen_q=2048, seqlen_kv=2048, head_dim=1024, causal=1), True),   # 64-row tiles: 32 of them
  (dict(batch=1, heads_q=32, heads_kv=32, seqlen_q=4096, seqlen_kv=16384, causal=1, causal_offset=12288), False),  # tail-aligned against a long context: tiles are nearly equal
  (dict(batch=1, heads_q=32, heads_kv=32, seqlen_q=4096, seqlen_kv=16384, causal=1, causal_offset=0), True),       # top-left: the diagonal decides
  (dict(batch=1, heads_q=32, heads_kv=32, seqlen_q=4096, seqlen_kv=4096), False),                # not causal
  (dict(batch=1, heads_q=2, heads_kv=2, seqlen_q=4096, seqlen_kv=16384, causal=1, causal_offset=12288), False),    # (an under-filled launch splits the KV axis: no pairing)
  (dict(batch=2, heads_q=32, heads_kv=8, seqlen_q=8192, seqlen_kv=2048, head_dim=320, causal=1), False),  # config 4 with the causal flag takes the wide-row tile
])
def test_paired_row_tile_rule(lib, over, paired):
  """Which causal prefill launches pair row tiles i and n - 1 - i in one workgroup (ffpa_capi.hip::pick_pair_tiles, profiles/r06_pair_tiles.txt): up to 32 row
  tiles per head, where the diagonal makes the tiles unequal; the kernel name says so.  FFPA_FLAG_PAIR_TILES / _NO_PAIR_TILES force it where the build exists."""
  name = ctypes.create_string_buffer(200)
  p = _params(**over)
  p.workspace, p.workspace_bytes = 16, 1 << 40
  assert lib.ffpa_attn_fwd_kernel(ctypes.byref(p), name, len(name)) == 0
  assert (", PAIR>" in name.value.decode()) == paired, name.value
  p.flags = hip.FLAG_NO_PAIR_TILES
  assert lib.ffpa_attn_fwd_kernel(ctypes.byref(p), name, len(name)) == 0 and ", PAIR>" not in name.value.decode()
  p.flags = hip.FLAG_PAIR_TILES
  can = bool(over.get("causal")) and over["heads_q"] > 2 and over.get("head_dim", 512) != 320
  assert lib.ffpa_attn_fwd_kernel(ctypes.byref(p), name, len(name)) == 0 and (", PAIR>" in name.value.decode()) == can, name.value


@pytest.mark.parametrize("over, chunk", [
  (dict(batch=1, heads_q=32, heads_kv=8, seqlen_q=8192, seqlen_kv=8192, causal=1, causal_offset=0), 4),      # a KV group per chunk, eight chunks: one per XCD
  (dict(batch=4, heads_q=32, heads_kv=8, seqlen_q=2048, seqlen_kv=2048, causal=1, causal_offset=0), 4),      # (16 row tiles per head: the chunk order, not the paired tiles)
  (dict(batch=1, heads_q=64, heads_kv=8, seqlen_q=4096, seqlen_kv=4096, causal=1, causal_offset=0), 8),
  (dict(batch=1, heads_q=32, heads_kv=4, seqlen_q=8192, seqlen_kv=8192, causal=1, causal_offset=0), 4),      # groups of 8: two chunks of 4 per group (eight chunks in all)
  (dict(batch=1, heads_q=32, heads_kv=8, seqlen_q=8192, seqlen_kv=8192, head_dim=1024, causal=1, causal_offset=0), 4),
  (dict(batch=1, heads_q=32, heads_kv=8, seqlen_q=4096, seqlen_kv=16384, causal=1, causal_offset=12288), 4),  # tail-aligned against a longer context
  (dict(batch=1, heads_q=32, heads_kv=32, seqlen_q=8192, seqlen_kv=8192, causal=1, causal_offset=0), 1),     # MHA: heads share nothing (measured - 3.7 % with chunks)
  (dict(batch=1, heads_q=32, heads_kv=8, seqlen_q=8192, seqlen_kv=8192), 1),                                 # not causal: measured +- 0
  (dict(batch=1, heads_q=12, heads_kv=4, seqlen_q=8192, seqlen_kv=8192, causal=1, causal_offset=0), 1),      # 12 heads do not make eight chunks
  (dict(batch=1, heads_q=32, heads_kv=8, seqlen_q=8192, seqlen_kv=4096, causal=1, causal_offset=-4096), 1),  # rows without a visible key: the dense kernel's NaN contract
  (dict(batch=1, heads_q=32, heads_kv=8, seqlen_q=8192, seqlen_kv=8192, causal=1, causal_offset=0, dropout_p=0.1), 1),
  (dict(batch=2, heads_q=32, heads_kv=8, seqlen_q=8192, seqlen_kv=2048, head_dim=320, causal=1, causal_offset=0), 1),  # config 4 with the causal flag: the wide-row tile
  (dict(batch=1, heads_q=32, heads_kv=8, seqlen_q=16, seqlen_kv=8192, causal=1, causal_offset=8176), 1),     # short-query launch
])
def test_head_chunk_order_rule(lib, over, chunk):
  """Which dense launches run in the head-chunk workgroup order (ffpa_capi.hip::pick_dense_head_chunk; profiles/r06_head_chunks.txt): causal + GQA, the build
  without bias / mask ranges / dropout, every row sees a key, and the KV groups divide into a multiple of eight chunks.  The kernel name says so;
  FFPA_FLAG_NO_HEAD_CHUNKS keeps the (batch, head, row tile) order.  (Same tile, same bits either way: tests/test_m16_gpu.py.)"""
  name = ctypes.create_string_buffer(200)
  p = _params(**over)
  p.workspace, p.workspace_bytes = 16, 1 << 40
  assert lib.ffpa_attn_fwd_kernel(ctypes.byref(p), name, len(name)) == 0
  got = name.value.decode()
  d = over.get("head_dim", 512)
  if chunk > 1:
    assert got == f"ffpa_fwd_m16_varlen_kernel<bf16, {d}> (dense launch, head chunks of {chunk})", got
  else:
    assert "varlen" not in got, got
  p.flags = hip.FLAG_NO_HEAD_CHUNKS
  assert lib.ffpa_attn_fwd_kernel(ctypes.byref(p), name, len(name)) == 0 and "varlen" not in name.value.decode()


@pytest.mark.parametrize("over, ranges", [
  (dict(heads_q=8, heads_kv=8, seqlen_q=4096, seqlen_kv=4096), 2),                     # 256 row tiles = one round: 212 -> 175 us (+ 21 %)
  (dict(heads_q=8, heads_kv=2, seqlen_q=4096, seqlen_kv=4096), 2),                     # GQA: + 18 %
  (dict(heads_q=6, heads_kv=6, seqlen_q=4096, seqlen_kv=4096), 2),                     # 192 workgroups: + 26 %
  (dict(heads_q=5, heads_kv=5, seqlen_q=4096, seqlen_kv=4096), 2),                     # 160: + 33 %
  (dict(heads_q=4, heads_kv=4, seqlen_q=8192, seqlen_kv=8192), 2),                     # + 4 %
  (dict(heads_q=8, heads_kv=8, seqlen_q=4096, seqlen_kv=4096, head_dim=320), 2),       # + 5 %
  (dict(heads_q=8, heads_kv=8, seqlen_q=4096, seqlen_kv=4096, head_dim=128), 3),       # small tiles, two workgroups per CU: three ranges + 25 % (two: + 15 %)
  (dict(heads_q=8, heads_kv=8, seqlen_q=3000, seqlen_kv=3000, head_dim=128), 2),       # ... 500 keys per range of the average row tile are too few for three
  (dict(heads_q=8, heads_kv=8, seqlen_q=3000, seqlen_kv=3000), 2),                     # 750 keys per range: + 22 %
  (dict(heads_q=2, heads_kv=2, seqlen_q=8192, seqlen_kv=8192, head_dim=1024), 2),      # split-D tiles (64 rows): + 8 % at 2048 keys per range
  (dict(heads_q=4, heads_kv=4, seqlen_q=4096, seqlen_kv=4096, head_dim=1024), 1),      # ... - 4 % at 1024
  (dict(heads_q=16, heads_kv=16, seqlen_q=2048, seqlen_kv=2048), 1),                   # 512 keys per range: - 1 %
  (dict(batch=2, heads_q=8, heads_kv=8, seqlen_q=2048, seqlen_kv=2048), 1),
  (dict(batch=4, heads_q=8, heads_kv=8, seqlen_q=1024, seqlen_kv=1024), 1),
  (dict(heads_q=8, heads_kv=8, seqlen_q=4096, seqlen_kv=8192, causal_offset=4096), 1),  # a chunk against a longer context: the longest row tile walks 1.33 x the average one
  (dict(heads_q=12, heads_kv=12, seqlen_q=4096, seqlen_kv=4096), 1),                   # 1.5 rounds: the short row tiles fill the tail by themselves
  (dict(heads_q=32, heads_kv=32, seqlen_q=8192, seqlen_kv=8192), 1),                   # the causal headline shape: eight rounds
  (dict(heads_q=8, heads_kv=8, seqlen_q=4096, seqlen_kv=4096, dropout_p=0.1), 1),      # the dense mode's builds: no dropout, no bias, every row sees a key
  (dict(heads_q=8, heads_kv=8, seqlen_q=4096, seqlen_kv=2048, causal_offset=-2048), 1),
])
def test_causal_tile_range_rule(lib, over, ranges):
  """Which dense causal launches split every row tile's OWN visible KV tiles (ffpa_capi.hip::pick_tile_ranges; profiles/r06_tile_ranges.txt): one round of
  workgroups or less (and more than half of one: fewer take the under-filled rule's uniform ranges), the longest row tile >= 1.5 x the average one, a range of the
  average one >= 640 keys (2048 for the split-D tiles); head dims under 256 take three.  The kernel name says so; no scratch / num_splits = 1 / the deterministic
  flag / FFPA_FLAG_NO_TILE_RANGES keep one range; FFPA_FLAG_TILE_RANGES | FFPA_FLAG_FORCE_SPLITS with num_splits = n force n.  256-CU fallback: no GPU needed."""
  base = dict(batch=1, causal=1, causal_offset=0)
  base.update(over)
  name = ctypes.create_string_buffer(200)
  plan = (ctypes.c_int * 4)()
  p = _params(**base)
  p.workspace, p.workspace_bytes = 16, 1 << 40
  assert lib.ffpa_attn_fwd_plan(ctypes.byref(p), plan) == 0 and lib.ffpa_attn_fwd_kernel(ctypes.byref(p), name, len(name)) == 0
  got = name.value.decode()
  assert ("KV ranges per row tile" in got) == (ranges > 1), (got, list(plan))
  if ranges > 1:
    d = base.get("head_dim", 512)
    dk = (d + 63) // 64 * 64
    assert plan[3] == ranges and got.startswith(f"ffpa_fwd_m16_varlen_kernel<bf16, {dk}> (dense launch, head chunks of ") and got.endswith("+ ffpa_fwd_merge_kernel"), (got, list(plan))
    assert lib.ffpa_attn_fwd_workspace_bytes(ctypes.byref(p)) == ranges * base["batch"] * base["heads_q"] * base["seqlen_q"] * (dk + 1) * 4
    for change in (dict(flags=hip.FLAG_NO_TILE_RANGES), dict(flags=hip.FLAG_DETERMINISTIC), dict(num_splits=1), dict(workspace=None, workspace_bytes=0)):
      q = _params(**base)
      q.workspace, q.workspace_bytes = 16, 1 << 40
      for key, val in change.items():
        setattr(q, key, val)
      assert lib.ffpa_attn_fwd_kernel(ctypes.byref(q), name, len(name)) == 0 and "KV ranges" not in name.value.decode(), (change, name.value)
    p.workspace_bytes = (ranges - 1) * base["batch"] * base["heads_q"] * base["seqlen_q"] * (dk + 1) * 4 + 64  # a smaller scratch: as many ranges as fit
    assert lib.ffpa_attn_fwd_plan(ctypes.byref(p), plan) == 0 and plan[3] == ranges - 1
    p.workspace_bytes = 1 << 40
  if not base.get("dropout_p") and base["causal_offset"] >= 0:
    p.flags, p.num_splits = hip.FLAG_TILE_RANGES | hip.FLAG_FORCE_SPLITS, 5
    assert lib.ffpa_attn_fwd_plan(ctypes.byref(p), plan) == 0 and plan[3] == 5 and lib.ffpa_attn_fwd_kernel(ctypes.byref(p), name, len(name)) == 0
    assert "KV ranges per row tile" in name.value.decode(), name.value
