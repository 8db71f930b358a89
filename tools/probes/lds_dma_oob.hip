// Probe (developer tool): what does `buffer_load_dwordx4 ... lds` do for lanes that are out of range of the
// raw buffer descriptor — write zeros to LDS, or leave LDS untouched?
//   hipcc --offload-arch=gfx950 -O2 tools/probes/lds_dma_oob.hip -o /tmp/lds_dma_oob && /tmp/lds_dma_oob
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
#define LDSAS __attribute__((address_space(3)))

__global__ void probe(const uint32_t* src, uint32_t valid_bytes, uint32_t soff, uint32_t* out) {
  __shared__ __attribute__((aligned(16))) uint32_t buf[256];
  const int lane = threadIdx.x;
  for (int i = lane; i < 256; i += 64) buf[i] = 0x7fc00001u;  // NaN pattern
  __syncthreads();
  const uint64_t a = (uint64_t)src;
  u32x4 rsrc = {(uint32_t)a, (uint32_t)(a >> 32) & 0xffffu, valid_bytes, 0x00020000u};
  const uint32_t voff = lane * 16;
  const uint32_t lds = (uint32_t)(uintptr_t)(LDSAS uint32_t*)buf;
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_waitcnt vmcnt(0)"
               :: "s"(lds), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
  __syncthreads();
  for (int i = lane; i < 256; i += 64) out[i] = buf[i];
}

int main() {
  std::vector<uint32_t> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = 0x1000 + i;
  uint32_t *d, *o;
  (void)hipMalloc(&d, 4096);
  (void)hipMalloc(&o, 1024);
  (void)hipMemcpy(d, h.data(), 4096, hipMemcpyHostToDevice);
  for (uint32_t valid : {1024u, 512u, 520u, 16u}) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, valid, 0u, o);
    std::vector<uint32_t> r(256);
    hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost);
    int ok = 0, zero = 0, stale = 0, other = 0;
    for (int i = 0; i < 256; ++i) {
      if (r[i] == h[i]) ++ok; else if (r[i] == 0) ++zero; else if (r[i] == 0x7fc00001u) ++stale; else ++other;
    }
    printf("PROBE num_records=%4u: %3d dwords copied, %3d zero, %3d stale(NaN pattern), %3d other; first non-copied dword index %d\n",
           valid, ok, zero, stale, other, ok);
  }
  // soffset: is it part of the range check?  lanes read src[soff + 16 lane .. +16) with num_records = 768, soff = 512
  {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, 768u, 512u, o);
    std::vector<uint32_t> r(256);
    (void)hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost);
    int copied = 0, zero = 0;
    for (int i = 0; i < 256; ++i) {
      if (r[i] == h[128 + i]) ++copied; else if (r[i] == 0) ++zero;
    }
    printf("PROBE soffset=512 num_records=768: %d dwords copied from src+512, %d zero  (64 copied => soffset counts; 192 copied => only voffset is checked)\n", copied, zero);
  }
  return 0;
}
