// LDS-DMA (global_load_lds_dwordx4) throughput probe for gfx950: how many bytes per clock per CU does the
// global -> LDS path sustain from an L2-resident buffer, as a function of workgroups per CU, waves per workgroup,
// slabs in flight and row stride?   hipcc --offload-arch=gfx950 -O3 -o dma_probe dma_probe.hip && ./dma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// Each wave issues PER instructions per slab (1 KiB each: 8 rows x 128 B, row r of the instruction at
// base + r * row_stride), keeps DEPTH slabs in flight, for `slabs` slabs; no consumer.
template <int PER, int DEPTH>
__global__ void probe(const char* src, size_t span, int row_stride, int slabs, int lds_per_wave) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nw = blockDim.x >> 6;
  float* my = lds + wave * (lds_per_wave / 4);
  size_t off = ((size_t)blockIdx.x * nw + wave) * 8 * (size_t)row_stride * PER;
  const char* lp = src + (size_t)(lane >> 3) * row_stride + (lane & 7) * 16;
  auto issue = [&](int s) {
    float* base = my + (s % DEPTH) * PER * 256;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      size_t o = (off + (size_t)q * 8 * row_stride + (size_t)s * 128) % span;
      __builtin_amdgcn_global_load_lds(GLB_PTR(lp + o), LDS_PTR(base + q * 256), 16, 0, 0);
    }
  };
#pragma unroll
  for (int i = 0; i < DEPTH - 1; ++i) issue(i);
  for (int s = 0; s < slabs; ++s) {
    issue(s + DEPTH - 1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 1) * PER) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int PER, int DEPTH>
static void run(const char* buf, size_t span, int row_stride, int wg_per_cu, int waves, double ghz) {
  const int slabs = 2000;
  const int lds_per_wave = PER * DEPTH * 1024;
  const size_t lds = (size_t)lds_per_wave * waves;
  hipFuncSetAttribute((const void*)probe<PER, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int grid = 256 * wg_per_cu;
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipLaunchKernelGGL((probe<PER, DEPTH>), dim3(grid), dim3(64 * waves), lds, 0, buf, span, row_stride, 50, lds_per_wave);
  hipEventRecord(a);
  hipLaunchKernelGGL((probe<PER, DEPTH>), dim3(grid), dim3(64 * waves), lds, 0, buf, span, row_stride, slabs, lds_per_wave);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double bytes = (double)grid * waves * PER * 1024.0 * (slabs + DEPTH - 1);
  printf("wg/cu %d waves %d per %d depth %d stride %5d in-flight/CU %4zu KB : %7.2f TB/s  %5.1f B/clk/CU @%.2f GHz  (lds %zu KB/wg)\n",
         wg_per_cu, waves, PER, DEPTH, row_stride, (size_t)wg_per_cu * waves * PER * (DEPTH - 1), bytes / ms / 1e9,
         bytes / (ms * 1e-3) / 256.0 / (ghz * 1e9), ghz, lds / 1024);
}

int main() {
  const size_t span = (size_t)24 << 20;       // 24 MiB: inside the 8 x 4 MiB L2s / the 256 MiB MALL
  char* buf = nullptr;
  hipMalloc((void**)&buf, span + (4 << 20));
  hipMemset(buf, 1, span + (4 << 20));
  const double ghz = 2.4;
  for (int stride : {128, 768}) {
    run<6, 2>(buf, span, stride, 2, 4, ghz);
    run<6, 3>(buf, span, stride, 2, 4, ghz);   // the conv kernel's shape: 2 WG x 4 waves x 6 KiB x 2 slabs in flight
    run<6, 4>(buf, span, stride, 1, 4, ghz);
    run<6, 3>(buf, span, stride, 1, 4, ghz);
    run<4, 3>(buf, span, stride, 1, 8, ghz);
    run<4, 4>(buf, span, stride, 1, 8, ghz);
    run<4, 5>(buf, span, stride, 1, 8, ghz);
    run<2, 3>(buf, span, stride, 4, 4, ghz);
    run<2, 5>(buf, span, stride, 4, 4, ghz);
    run<1, 3>(buf, span, stride, 8, 4, ghz);
    run<1, 8>(buf, span, stride, 8, 4, ghz);
  }
  return 0;
}
