// Microbenchmark: rate of L2 -> LDS fills per CU as a function of the contiguous piece a group of lanes reads.
// The row-shared correlation kernel and the halo convolution kernel stage 64-byte pieces (32 channels of one plane of one
// site, sites 512 B apart); this measures what that access shape costs against wider pieces and against plain loads.
//   hipcc --offload-arch=gfx950 -O3 tools/debug/dma_rate.hip -o gpurun_out/dma_rate && gpurun_out/dma_rate
// Output: one line per mode: GB/s per CU, cycles per 1-KiB instruction per CU (at the measured clock of 2.4 GHz).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u32x4 raw_rsrc(const void* p, size_t bytes) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(p);
  u32x4 r;
  r.x = (unsigned)a;
  r.y = (unsigned)(a >> 32) & 0xffffu;
  r.z = (unsigned)bytes;
  r.w = 0x00020000u;
  return r;
}
__device__ __forceinline__ void dma16(int voff, u32x4 r, unsigned d) {
  unsigned keep;
  asm volatile("s_mov_b32 %[keep], m0\n\ts_mov_b32 m0, %[d]\n\ts_nop 0\n\tbuffer_load_dwordx4 %[v], %[r], 0 offen lds\n\ts_mov_b32 m0, %[keep]"
               : [keep] "=&s"(keep)
               : [v] "v"(voff), [r] "s"(r), [d] "s"(d)
               : "memory");
}
__device__ __forceinline__ void dma4(int voff, u32x4 r, unsigned d) {
  unsigned keep;
  asm volatile("s_mov_b32 %[keep], m0\n\ts_mov_b32 m0, %[d]\n\ts_nop 0\n\tbuffer_load_dword %[v], %[r], 0 offen lds\n\ts_mov_b32 m0, %[keep]"
               : [keep] "=&s"(keep)
               : [v] "v"(voff), [r] "s"(r), [d] "s"(d)
               : "memory");
}

constexpr int NI = 16;                                  // instructions per wave per round (one "chunk")
constexpr int REGION = 1 << 20;                         // bytes one workgroup walks over

// MODE 0: DMA, 16 B per lane.  MODE 1: DMA, 4 B per lane.  MODE 2: plain 16-byte loads into registers.
template <int MODE>
__global__ __launch_bounds__(256, 2) void fill_kernel(const char* src, int piece, int stride, int iters, unsigned* sink) {
  extern __shared__ char smem[];
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int per = (MODE == 1 ? 4 : 16);
  const int lp = piece / per;                           // lanes per piece
  const int ppi = 64 / lp;                              // pieces per instruction
  const char* base = src + (size_t)((blockIdx.x >> 3) & 1) * REGION;
  const u32x4 rs = raw_rsrc(base, REGION);
  const int voff0 = (lane / lp) * stride + (lane % lp) * per + wid * piece;   // the four waves read neighbouring pieces of a site
  const unsigned dst0 =
      __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(__attribute__((address_space(3))) const void*)smem + wid * NI * 1024);
  const int span = ppi * stride;                        // bytes of source one instruction walks over
  u32x4 accv = {0, 0, 0, 0};
  int pos = 0;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NI; i++) {
      int vo = voff0 + pos;
      if (MODE == 0) dma16(vo, rs, dst0 + i * 1024);
      if (MODE == 1) dma4(vo, rs, dst0 + i * 256);
      if (MODE == 2) {
        u32x4 v;
        asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(v) : "v"(vo), "s"(rs) : "memory");
        asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
        accv.x ^= 1;                                    // the value is never consumed before the final wait
        if (i == NI - 1) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); accv ^= v; }
      }
      pos += span;
      if (pos + span + 4 * piece > REGION) pos = 0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if (accv.x == 0x12345678u) sink[0] = accv.y;
  if (iters < 0) sink[threadIdx.x] = ((unsigned*)smem)[threadIdx.x];
}

template <int MODE>
static void run(const char* name, const char* src, int piece, int stride, unsigned* sink) {
  const int iters = 400, grid = 512;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  const int smem = 4 * NI * 1024;
  fill_kernel<MODE><<<grid, 256, smem>>>(src, piece, stride, 20, sink);
  hipEventRecord(a);
  fill_kernel<MODE><<<grid, 256, smem>>>(src, piece, stride, iters, sink);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double per = (MODE == 1 ? 256.0 : 1024.0);
  const double bytes = (double)grid * 4 * NI * per * iters;
  const double gbs = bytes / (ms * 1e-3) / 1e9;
  const double instr_per_cu = (double)grid / 256 * 4 * NI * iters;
  const double cyc = ms * 1e-3 * 2.4e9 / instr_per_cu;
  printf("%-34s piece %4d B stride %4d: %8.1f GB/s chip %6.1f GB/s per CU  %6.1f cycles per instruction per CU (%s B)\n", name, piece,
         stride, gbs, gbs / 256, cyc, MODE == 1 ? "256" : "1024");
}

int main() {
  char* src;
  unsigned* sink;
  hipMalloc(&src, 2 * REGION + 65536);
  hipMemset(src, 1, 2 * REGION + 65536);
  hipMalloc(&sink, 4096);
  for (int rep = 0; rep < 2; rep++) {
    run<0>("lds-dma 16 B/lane", src, 64, 512, sink);
    run<0>("lds-dma 16 B/lane", src, 128, 512, sink);
    run<0>("lds-dma 16 B/lane", src, 256, 512, sink);
    run<0>("lds-dma 16 B/lane", src, 512, 512, sink);
    run<0>("lds-dma 16 B/lane", src, 64, 64, sink);
    run<0>("lds-dma 16 B/lane", src, 64, 128, sink);
    run<0>("lds-dma 16 B/lane", src, 64, 256, sink);
    run<0>("lds-dma 16 B/lane", src, 128, 1024, sink);
    run<1>("lds-dma 4 B/lane", src, 64, 512, sink);
    run<1>("lds-dma 4 B/lane", src, 256, 256, sink);
    run<2>("plain 16-byte loads to registers", src, 64, 512, sink);
    run<2>("plain 16-byte loads to registers", src, 128, 512, sink);
    run<2>("plain 16-byte loads to registers", src, 1024, 1024, sink);
  }
  return 0;
}
