// Do LDS-DMA (global_load_lds_dwordx4) and MFMA issue interfere when they share a SIMD?
// 8 waves per workgroup, 1 workgroup per CU.  `dma_mask` / `mfma_mask` select by wave index which waves stream
// DMA (4 KiB per slab, 2 slabs in flight) and which issue back-to-back v_mfma_f32_32x32x16_bf16 on 4 independent
// accumulators.  Waves w and w+4 share a SIMD.  Reports DMA B/clk per wave and MFMA busy fraction per MFMA wave.
//   hipcc --offload-arch=gfx950 -O3 -o dma_mfma_probe dma_mfma_probe.hip && ./dma_mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int SADDR>
__global__ __launch_bounds__(512, 2) void probe(const char* src, size_t span, int slabs, int n_mfma, unsigned dma_mask,
                                                unsigned mfma_mask, float* sink, long long* cycles) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  long long t0 = __builtin_readcyclecounter();
  if ((dma_mask >> wave) & 1) {
    float* my = lds + wave * (2 * 4 * 256);
    size_t off = ((size_t)blockIdx.x * 8 + wave) * 4096;
    const char* lp = src + (size_t)(lane >> 3) * 128 + (lane & 7) * 16;
    auto issue = [&](int s) {
      float* base = my + (s & 1) * 4 * 256;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        size_t o = (off + (size_t)q * 1024 + (size_t)s * 65536) % span;
        if constexpr (SADDR) {
          // uniform 64-bit base in SGPRs + one 32-bit offset VGPR (saddr form); M0 = LDS destination
          const char* ub = src + o;
          const unsigned lo = (unsigned)((lane >> 3) * 128 + (lane & 7) * 16);
          const unsigned ldsa = (unsigned)(size_t)LDS_PTR(base + q * 256);
          asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lo), "s"(ub), "s"(ldsa) : "memory", "m0");
        } else {
          __builtin_amdgcn_global_load_lds(GLB_PTR(lp + o), LDS_PTR(base + q * 256), 16, 0, 0);
        }
      }
    };
    issue(0);
    for (int s = 0; s < slabs; ++s) {
      issue(s + 1);
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else if ((mfma_mask >> wave) & 1) {
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a)
      for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    bf16x8 x, y;
    for (int e = 0; e < 8; ++e) {
      x[e] = (__bf16)(float)(lane + e);
      y[e] = (__bf16)(float)(lane - e);
    }
    for (int i = 0; i < n_mfma; i += 4) {
#pragma unroll
      for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[a], 0, 0, 0);
    }
    float s = 0;
    for (int a = 0; a < 4; ++a) s += acc[a][lane & 15];
    if (s == 12345.678f) sink[0] = s;
  }
  long long t1 = __builtin_readcyclecounter();
  if (lane == 0 && blockIdx.x == 0) cycles[wave] = t1 - t0;
}

template <int SADDR>
static void run(const char* buf, size_t span, unsigned dma_mask, unsigned mfma_mask, const char* what, float* sink,
                long long* cyc_dev) {
  const int slabs = 3000, n_mfma = 40000;
  const size_t lds = 8 * 2 * 4 * 1024;
  hipFuncSetAttribute((const void*)probe<SADDR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipLaunchKernelGGL(probe<SADDR>, dim3(256), dim3(512), lds, 0, buf, span, 20, 100, dma_mask, mfma_mask, sink, cyc_dev);
  hipEventRecord(a);
  hipLaunchKernelGGL(probe<SADDR>, dim3(256), dim3(512), lds, 0, buf, span, slabs, n_mfma, dma_mask, mfma_mask, sink, cyc_dev);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  long long cyc[8];
  hipMemcpy(cyc, cyc_dev, sizeof(cyc), hipMemcpyDeviceToHost);
  printf("%-48s kernel %6.3f ms |", what, ms);
  for (int w = 0; w < 8; ++w) {
    if ((dma_mask >> w) & 1)
      printf(" w%d dma %4.1f", w, 4096.0 * (slabs + 1) / (double)cyc[w]);       // B per cycle-counter tick, this wave
    else if ((mfma_mask >> w) & 1)
      printf(" w%d mfma %5.0f", w, (double)cyc[w] / n_mfma);                    // cycle-counter ticks per MFMA
  }
  printf("\n");
}

int main() {
  const size_t span = (size_t)24 << 20;
  char* buf = nullptr;
  float* sink = nullptr;
  long long* cyc = nullptr;
  hipMalloc((void**)&buf, span + (4 << 20));
  hipMemset(buf, 1, span + (4 << 20));
  hipMalloc((void**)&sink, 64);
  hipMalloc((void**)&cyc, 64);
  run<1>(buf, span, 0x0F, 0x00, "SADDR dma only: waves 0-3", sink, cyc);
  run<1>(buf, span, 0xFF, 0x00, "SADDR dma only: all 8 waves", sink, cyc);
  run<1>(buf, span, 0x0F, 0xF0, "SADDR dma 0-3 + mfma 4-7 (SHARED SIMDs)", sink, cyc);
  run<0>(buf, span, 0x00, 0xF0, "mfma only: waves 4-7 (one per SIMD)", sink, cyc);
  run<0>(buf, span, 0x00, 0xFF, "mfma only: all 8 waves (two per SIMD)", sink, cyc);
  run<0>(buf, span, 0x0F, 0x00, "dma only: waves 0-3 (one per SIMD)", sink, cyc);
  run<0>(buf, span, 0xFF, 0x00, "dma only: all 8 waves", sink, cyc);
  run<0>(buf, span, 0x0F, 0xF0, "dma 0-3 + mfma 4-7 (SHARED SIMDs)", sink, cyc);
  run<0>(buf, span, 0x33, 0xCC, "dma {0,1,4,5} + mfma {2,3,6,7} (SEPARATE SIMDs)", sink, cyc);
  run<0>(buf, span, 0x03, 0xF0, "dma 0-1 + mfma 4-7", sink, cyc);
  run<0>(buf, span, 0x01, 0xF0, "dma 0 + mfma 4-7", sink, cyc);
  return 0;
}
