"""bench.py pieces that need no GPU: the HBM-traffic figure is quoted only from a profile of the very library that runs (round-3 review:
the line read a committed file with no check that it described the running binary)."""

import importlib.util
import json
import os

from conftest import ROOT


def _bench():
  spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
  mod = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mod)
  return mod


def test_traffic_is_only_quoted_from_a_profile_of_the_running_library(tmp_path, monkeypatch):
  bench = _bench()
  prof = tmp_path / "profiles"
  prof.mkdir()
  doc = {"provenance": {"lib_sha16": "aaaabbbbccccdddd"}, "derived": {"hbm_read_bytes_corrected_x2": 1000.0, "hbm_write_bytes": 24.0}}
  (prof / "r04_bench_cfg2_pmc.json").write_text(json.dumps(doc))
  monkeypatch.setattr(bench, "ROOT", str(tmp_path))
  assert bench.measured_traffic("cfg2", "aaaabbbbccccdddd") == (1024, os.path.join("profiles", "r04_bench_cfg2_pmc.json"), False)
  traffic, src, stale = bench.measured_traffic("cfg2", "0000000000000000")
  assert traffic is None and stale is True and src.endswith("r04_bench_cfg2_pmc.json")  # another binary's profile: named, not quoted
  assert bench.measured_traffic("cfg2", None)[0] is None                                # the running library's sha is unknown: not quoted either
  assert bench.measured_traffic("cfg3", "aaaabbbbccccdddd") == (None, None, False)       # no profile committed


def test_committed_r04_profiles_name_their_binary():
  """Every committed round-4 PMC summary carries the library sha / git head / clock it was taken with."""
  import glob

  files = glob.glob(os.path.join(ROOT, "profiles", "r04_bench_*_pmc*.json"))
  assert files
  for f in files:
    prov = json.load(open(f)).get("provenance", {})
    assert len(prov.get("lib_sha16", "")) == 16 and prov.get("git_head") and prov.get("bench_kernel"), f
