// Does the plain vector-memory path (global_load_dwordx4 -> VGPR) have its own bandwidth next to the LDS-DMA path
// (global_load_lds_dwordx4), or do they share one ~34 B/clk/CU pipe?  8 waves per CU; masks pick DMA waves and
// plain-load waves; every wave streams 4 KiB per step from an L2-resident buffer with 2 steps in flight.
//   hipcc --offload-arch=gfx950 -O3 -o vmem_mix_probe vmem_mix_probe.hip && ./vmem_mix_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512, 2) void probe(const char* src, size_t span, int steps, unsigned dma_mask,
                                                unsigned ld_mask, int to_lds, float* sink) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* my = lds + wave * (2 * 4 * 256);
  size_t off = ((size_t)blockIdx.x * 8 + wave) * 4096;
  const char* lp = src + (size_t)(lane >> 3) * 128 + (lane & 7) * 16;
  if ((dma_mask >> wave) & 1) {
    auto issue = [&](int s) {
      float* base = my + (s & 1) * 4 * 256;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        size_t o = (off + (size_t)q * 1024 + (size_t)s * 65536) % span;
        __builtin_amdgcn_global_load_lds(GLB_PTR(lp + o), LDS_PTR(base + q * 256), 16, 0, 0);
      }
    };
    issue(0);
    for (int s = 0; s < steps; ++s) {
      issue(s + 1);
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else if ((ld_mask >> wave) & 1) {
    f32x4 cur[4], nxt[4];
    f32x4 accv = {0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < 4; ++q) cur[q] = *(const f32x4*)(lp + (off + (size_t)q * 1024) % span);
    for (int s = 0; s < steps; ++s) {
#pragma unroll
      for (int q = 0; q < 4; ++q) nxt[q] = *(const f32x4*)(lp + (off + (size_t)q * 1024 + (size_t)(s + 1) * 65536) % span);
      if (to_lds) {
#pragma unroll
        for (int q = 0; q < 4; ++q) *(f32x4*)(my + (s & 1) * 1024 + q * 256 + lane * 4) = cur[q];
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) accv += cur[q];
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) cur[q] = nxt[q];
    }
    if (accv[0] == 12345.678f) sink[0] = accv[1];
  }
}

static void run(const char* buf, size_t span, unsigned dma_mask, unsigned ld_mask, int to_lds, const char* what, float* sink) {
  const int steps = 3000;
  const size_t lds = 8 * 2 * 4 * 1024;
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipLaunchKernelGGL(probe, dim3(256), dim3(512), lds, 0, buf, span, 20, dma_mask, ld_mask, to_lds, sink);
  hipEventRecord(a);
  hipLaunchKernelGGL(probe, dim3(256), dim3(512), lds, 0, buf, span, steps, dma_mask, ld_mask, to_lds, sink);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const int nw = __builtin_popcount(dma_mask) + __builtin_popcount(ld_mask);
  const double bytes = 256.0 * nw * 4096.0 * (steps + 1);
  printf("%-52s %6.3f ms  %6.2f TB/s  %5.1f B/clk/CU @2.4 GHz\n", what, ms, bytes / ms / 1e9, bytes / (ms * 1e-3) / 256.0 / 2.4e9);
}

int main() {
  const size_t span = (size_t)24 << 20;
  char* buf = nullptr;
  float* sink = nullptr;
  hipMalloc((void**)&buf, span + (4 << 20));
  hipMemset(buf, 1, span + (4 << 20));
  hipMalloc((void**)&sink, 64);
  run(buf, span, 0xFF, 0x00, 0, "LDS-DMA, 8 waves", sink);
  run(buf, span, 0x0F, 0x00, 0, "LDS-DMA, 4 waves", sink);
  run(buf, span, 0x00, 0xFF, 0, "plain loads -> VGPR, 8 waves", sink);
  run(buf, span, 0x00, 0x0F, 0, "plain loads -> VGPR, 4 waves", sink);
  run(buf, span, 0x00, 0xFF, 1, "plain loads -> VGPR -> ds_write_b128, 8 waves", sink);
  run(buf, span, 0x0F, 0xF0, 0, "LDS-DMA 4 waves + plain loads 4 waves", sink);
  run(buf, span, 0x0F, 0xF0, 1, "LDS-DMA 4 waves + plain loads+ds_write 4 waves", sink);
  return 0;
}
