"""Host-side simulation of the kernel's LDS addressing (developer tool).

Re-derives, in plain Python, where the LDS-DMA puts every 16-byte slot of a K / V tile and which
bytes each lane's ds_read_b128 / ds_read_b64_tr_b16 touches, and checks that
  * K fragment (lane l, step s, key block kb) = K[kb*32 + pi(l%32)][dh*DW + 16 s + 8 (l/32) .. +8] with
    pi(a) = 16 (a/4 % 2) + a%4 + 4 (a/8): MFMA row a of S^T = K.Q^T is fed key pi(a), so that the C layout
    (row (r&3) + 8 (r>>2) + 4 (l/32) in register r) hands lane half h = l/32 the 16 CONTIGUOUS keys 16 h + r
  * V^T fragment (lane l, column block db, step ks, element jj) =
        V[32 (ks/2) + 16 (l/32) + 8 (ks%2) + jj][dh*DW + 32 db + l%32]   (the same key <-> slot map)
    under the transpose-read rule  result[lane i][e] = data of lane 4e + i/4, element i%4
    (cdna_hip_programming.md §2 "ds_read_b64_tr_b16")
and reports the worst bank conflict per LDS instruction group (MI355X_MICROARCH.md §LDS).
"""
import sys


def k_sw(D, key):
  return (key & 15) if D % 128 == 0 else ((key >> 1) & 7)


def v_sw(D, key):
  return ((key & 3) << 2) if D % 128 == 0 else (((key >> 1) & 1) << 2)


def pi(a):
  return 16 * ((a >> 2) & 1) + (a & 3) + 4 * (a >> 3)


def check(D, ND=None):
  ND = ND or (1 if D <= 512 else 2)
  DW, BC = D // ND, ((128 if D <= 320 else 64) if ND == 1 else 32)
  RB, SPR = D * 2, D // 8
  PIECES = BC * D * 2 // 1024
  PPW = PIECES // 4
  # ---- DMA image: lds[slot index] = (key, source slot)
  for is_v in (False, True):
    lds = {}
    for wave in range(4):
      for i in range(PPW):
        p = wave * PPW + i
        for lane in range(64):
          if (D * 2) % 1024 == 0:
            RPP = D * 2 // 1024
            key = p // RPP
            sw = v_sw(D, key) if is_v else k_sw(D, key)
            src_byte = ((lane ^ sw) << 4) + (p % RPP) * 1024
          else:
            g = p * 64 + lane
            key, slot = divmod(g, SPR)
            src_byte = (slot ^ (v_sw(D, key) if is_v else k_sw(D, key))) << 4
          assert 0 <= src_byte < RB, (D, is_v, src_byte)
          dst = p * 1024 + lane * 16
          assert dst not in lds
          lds[dst] = (key, src_byte)
    assert len(lds) == BC * RB // 16
    img = {}  # lds byte -> (key, source byte) at 2-byte granularity
    for dst, (key, sb) in lds.items():
      for b in range(0, 16, 2):
        img[dst + b] = (key, sb + b)
    if not is_v:
      kimg = img
    else:
      vimg = img
  worst_k = worst_v = 1
  for dh in range(ND):
    # ---- K fragments
    for kb in range(BC // 32):
      for s in range(DW // 16):
        addrs = []
        for lane in range(64):
          l31, h = lane & 31, lane >> 5
          kx = k_sw(D, pi(l31))
          c0 = dh * (DW // 8)
          kaddr = pi(l31) * RB + (((c0 + 2 * (s & 7) + h) ^ kx) << 4)
          a = kaddr + (s >> 3) * 256 + kb * 32 * RB
          addrs.append(a)
          for e in range(8):
            key, sb = kimg[a + 2 * e]
            assert key == kb * 32 + pi(l31) and sb == (dh * DW + 16 * s + 8 * h + e) * 2, (D, lane, s, kb, e, key, sb)
        # ds_read_b128: 4 groups of 16 lanes, bank = (a/4) % 64, 4 dwords each
        for grp in ([0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
                    [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]):
          for half in (0, 32):
            banks = {}
            for l in grp:
              banks.setdefault((addrs[l + half] // 16) % 16, set()).add(addrs[l + half])
            worst_k = max(worst_k, max(len(v) for v in banks.values()))
    # ---- V^T fragments
    for db in range(DW // 32):
      for ks in range(BC // 16):
        for hh in range(2):
          lane_addr = []
          for lane in range(64):
            h, j4 = lane >> 5, (lane & 15) >> 2
            vsw = v_sw(D, j4) * 16
            vcol = dh * DW * 2 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8
            vaddr = (16 * h + j4) * RB + ((vcol + (db & 3) * 64) ^ vsw)
            lane_addr.append(vaddr + (db >> 2) * 256 + (32 * (ks >> 1) + 8 * (ks & 1) + 4 * hh) * RB)
          for lane in range(64):
            l31, h = lane & 31, lane >> 5
            i, base = lane & 15, lane & ~15
            for e in range(4):
              src_lane = base + 4 * e + (i >> 2)
              key, sb = vimg[lane_addr[src_lane] + 2 * (i & 3)]
              jj = 4 * hh + e
              want_key = 32 * (ks >> 1) + 16 * h + 8 * (ks & 1) + jj
              want_col = dh * DW + db * 32 + l31
              assert (key, sb) == (want_key, want_col * 2), (D, lane, db, ks, hh, e, key, sb, want_key, want_col)
          # ds_read_b64_tr_b16: 2 groups of 32 lanes, bank = (a/4) % 64, 2 dwords each
          for half in (0, 32):
            banks = {}
            for l in range(32):
              a = lane_addr[l + half]
              banks.setdefault((a // 8) % 32, set()).add(a)
            worst_v = max(worst_v, max(len(v) for v in banks.values()))
  print(f"D={D:5d} ND={ND} BC={BC}: K/V fragment maps OK; worst bank conflict: ds_read_b128 {worst_k}-way, tr_b16 {worst_v}-way")
  return worst_k, worst_v


def m16_k_sw(D, key):
  return (key & 15) if D % 128 == 0 else ((key >> 1) & 7)


def m16_v_sw(D, key):
  return ((key & 7) << 1) if D % 128 == 0 else (((key >> 1) & 3) << 1)


def check_m16(D):
  """The 16x16x32-MFMA build (csrc/ffpa_fwd_m16_kernel.h); wave (qb, dh) owns columns dh * DW .. + DW (DW = D for D <= 512, D / 2 above):
    K fragment (lane l, step s, 16-key block kb) = K[16 kb + l % 16][dh DW + 32 s + 8 (l / 16) .. + 8]
    V^T fragment (lane l, column block db, key step ks, element e) = V[32 ks + 16 (e / 4) + 4 (l / 16) + e % 4][dh DW + 16 db + l % 16]
  and both instruction groups conflict-free."""
  ND = 1 if D <= 512 else 2
  DW = D // ND
  BC = 32 if ND == 2 else (128 if D <= 320 else 64)
  RB, SPR = D * 2, D // 8
  PPW = BC * D * 2 // 4096
  row_dma = RB % 1024 == 0
  RPP = RB // 1024 if row_dma else 1
  imgs = []
  for is_v in (False, True):
    sw = m16_v_sw if is_v else m16_k_sw
    img = {}
    for wave in range(4):
      for i in range(PPW):
        for lane in range(64):
          if row_dma:
            jk, half = divmod(i, RPP)
            key = 16 * (jk >> 2) + 4 * wave + (jk & 3)
            src = ((lane ^ sw(D, 4 * wave + (jk & 3))) << 4) + half * 1024
            assert sw(D, key) == sw(D, 4 * wave + (jk & 3))
            dst = key * RB + half * 1024 + lane * 16
          else:
            g = (wave * PPW + i) * 64 + lane
            key, slot = divmod(g, SPR)
            src = (slot ^ sw(D, key)) << 4
            dst = (wave * PPW + i) * 1024 + lane * 16
          assert 0 <= src < RB and key < BC, (D, is_v, key, src)
          for b in range(0, 16, 2):
            assert dst + b not in img
            img[dst + b] = (key, src + b)
    assert len(img) == BC * RB // 2
    imgs.append(img)
  kimg, vimg = imgs
  KV, KVB = (4, 256) if D % 128 == 0 else (2, 128)
  VV, VVB = (8, 256) if D % 128 == 0 else (4, 128)
  worst_k = worst_v = 1
  for dh in range(ND):
    c0 = dh * (DW // 8)
    for kb in range(BC // 16):
      for s in range(DW // 32):
        addrs = []
        for lane in range(64):
          n, c = lane & 15, lane >> 4
          base = (64 * (kb // 4) + n) * RB + (((c0 + 4 * (s % KV) + c) ^ m16_k_sw(D, n)) << 4)
          a = base + (s // KV) * KVB + (kb % 4) * 16 * RB
          assert (s // KV) * KVB + (kb % 4) * 16 * RB < 65536
          addrs.append(a)
          for e in range(8):
            assert kimg[a + 2 * e] == (16 * kb + n, (dh * DW + 32 * s + 8 * c + e) * 2), (D, lane, s, kb, e)
        for grp in ([0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]):
          for half in (0, 32):
            banks = {}
            for l in grp:
              banks.setdefault((addrs[l + half] // 16) % 16, set()).add(addrs[l + half])
            worst_k = max(worst_k, max(len(v) for v in banks.values()))
    for db in range(DW // 16):
      for ks in range(BC // 32):
        for second in (0, 1):
          la = []
          for lane in range(64):
            n, c = lane & 15, lane >> 4
            vkey = 4 * c + (n >> 2)
            base = (64 * (ks // 2) + vkey) * RB + (((c0 + 2 * (db % VV) + ((n & 3) >> 1)) ^ m16_v_sw(D, vkey)) << 4) + 8 * (n & 1)
            off = (db // VV) * VVB + ((ks % 2) * 32 + 16 * second) * RB
            assert off < 65536
            la.append(base + off)
          for lane in range(64):
            i, base_lane, c = lane & 15, lane & ~15, lane >> 4
            for e in range(4):
              src_lane = base_lane + 4 * e + (i >> 2)
              got = vimg[la[src_lane] + 2 * (i & 3)]
              want = (32 * ks + 16 * second + 4 * c + e, (dh * DW + 16 * db + i) * 2)
              assert got == want, (D, lane, db, ks, second, e, got, want)
          for half in (0, 32):
            banks = {}
            for l in range(32):
              banks.setdefault((la[l + half] // 8) % 32, set()).add(la[l + half])
            worst_v = max(worst_v, max(len(v) for v in banks.values()))
  print(f"D={D:5d} 16x16x32 build ND={ND} BC={BC}: K/V fragment maps OK; worst bank conflict: ds_read_b128 {worst_k}-way, tr_b16 {worst_v}-way")
  return worst_k, worst_v


def variants(D):
  """(D, ND) pairs the library instantiates: the prefill tiles and the short-query (split over 2 / 4 waves) tiles."""
  return [(D, 1 if D <= 512 else 2), (D, 4 if D % 128 == 0 else 2)]


if __name__ == "__main__":
  for D in ([int(x) for x in sys.argv[1:]] or [64, 128, 192, 256, 320, 384, 448, 512, 576, 640, 704, 768, 832, 896, 960, 1024]):
    for d, nd in dict.fromkeys(variants(D)):
      check(d, nd)
    check_m16(D)
