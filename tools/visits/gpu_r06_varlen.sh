#!/bin/bash
# round 6, the packed-sequence visits (v14 ... v24), as run through `gpurun -- '<one of the stanzas>'`; outputs under gpurun_out/r06/, digests in profiles/r06_varlen.txt,
# profiles/r06_head_chunks.txt.  Each stanza is what one visit ran; the library of each visit was the tree's at that commit (the bench lines carry its sha).
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
case ${1:-tests} in
tests)        # v14: the packed-sequence GPU tests alone (after every kernel change)
  python -m pytest tests/test_varlen_gpu.py -q -m gpu ;;
bench)        # v15 ... v18: smoke + the packed bench line (sequence-major / head-major / chunk orders: one visit per order, profiles/r06_varlen.txt)
  python -c "import __graft_entry__ as g; g.smoke()"
  python bench.py --workload varlen --steps 20 --warmup 5 --no-cpu-baseline ;;
decode)       # v19 ... v21: packed decode batches before / after the GQA row packing
  python tools/gpu_varlen_decode.py ;;
head_chunks)  # v22: the head-chunk order on dense causal GQA shapes, same-run A/B of the two orders (profiles/r06_head_chunks.txt)
  python -m pytest tests/test_m16_gpu.py -q -m gpu -k head_chunk
  python tools/gpu_dense_vs_packed.py ;;
order)        # v23: does the order of the sequences in the batch matter?
  python tools/gpu_varlen_order.py ;;
fuzz)         # v24: 400 more random packed batches
  FFPA_VARLEN_FUZZ=100:500 python -m pytest tests/test_varlen_gpu.py -m gpu -q -k test_randomized_packed_batches ;;
esac
