// Does v_pk_mul_f32 with its destination pair overlapping the source pair that BOTH halves read through op_sel
//     v_pk_mul_f32 v[d:d+1], v[a:a+1], v[d:d+1] op_sel:[0,1]        (lo = a.lo * d.hi, hi = a.hi * d.hi -> d.hi is overwritten)
// return a wrong LOW half when another kernel shares the SIMD?  Round 2 saw ONE wrong sum in ~10 % of the replays of a two-branch
// hipGraph, always in the low half of exactly this compiler-allocated instruction (unflow_amd/build.py), never
// alone on the chip; the library has been built with -fno-slp-vectorize -fno-vectorize since.  This isolates the instruction:
// kernel `probe` executes it (inline asm, the same register overlap) on changing data and counts results that differ from the
// two scalar products; it runs alone, beside an MFMA-bound kernel and beside a VALU/LDS-bound kernel on a second stream, all
// CUs shared (both grids keep <= half the wave slots).
//   hipcc --offload-arch=gfx950 -O3 -o pk_mul_hazard tools/microbench/pk_mul_hazard.hip && ./pk_mul_hazard
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void probe(unsigned long long* bad, unsigned long long* done, int iters, unsigned seed) {
  unsigned s = seed ^ (blockIdx.x * 2654435761u) ^ (threadIdx.x * 40503u);
  unsigned long long wrong = 0;
  for (int it = 0; it < iters; it++) {
    s = s * 1664525u + 1013904223u;
    const float x = 1.0f + (float)(s & 0xffff) * (1.0f / 65536.0f);
    s = s * 1664525u + 1013904223u;
    const float y = 1.0f + (float)(s & 0xffff) * (1.0f / 65536.0f);
    s = s * 1664525u + 1013904223u;
    const float p = 1.0f + (float)(s & 0xffff) * (1.0f / 65536.0f);
    s = s * 1664525u + 1013904223u;
    const float q = 1.0f + (float)(s & 0xffff) * (1.0f / 65536.0f);
    f32x2 a, d;
    a.x = x; a.y = y; d.x = p; d.y = q;
    asm volatile("v_pk_mul_f32 %0, %1, %0 op_sel:[0,1]" : "+v"(d) : "v"(a));
    const float lo = x * q, hi = y * q;
    if (d.x != lo || d.y != hi) wrong++;
  }
  if (wrong) atomicAdd(bad, wrong);
  if (threadIdx.x == 0) atomicAdd(done, (unsigned long long)iters * 256ull);
}

__global__ __launch_bounds__(256) void mfma_corunner(float* sink, int iters) {
  bf16x8 a, b;
  for (int i = 0; i < 8; i++) { a[i] = (__bf16)(1.0f + threadIdx.x * 0.001f + i); b[i] = (__bf16)(0.5f + i * 0.25f); }
  f32x16 c0 = {}, c1 = {};
  for (int it = 0; it < iters; it++) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, c1, 0, 0, 0);
  }
  if (c0[0] + c1[3] == 12345.f) sink[0] = c0[1];
}

__global__ __launch_bounds__(256) void valu_corunner(float* sink, int iters) {
  __shared__ float lds[1024];
  float v = threadIdx.x * 0.5f, w = 1.0001f;
  for (int it = 0; it < iters; it++) {
    lds[(threadIdx.x * 7 + it) & 1023] = v;
    v = v * w + lds[(threadIdx.x * 13 + it) & 1023];
    w = w * 0.99999f + 1e-6f;
  }
  if (v == 12345.f) sink[0] = v;
}

int main() {
  unsigned long long *bad, *done;
  float* sink;
  hipMalloc(&bad, 8); hipMalloc(&done, 8); hipMalloc(&sink, 64);
  hipStream_t s0, s1;
  hipStreamCreate(&s0); hipStreamCreate(&s1);
  const char* names[3] = {"alone", "beside an MFMA-bound kernel", "beside a VALU/LDS-bound kernel"};
  for (int mode = 0; mode < 3; mode++) {
    unsigned long long tb = 0, td = 0;
    for (int rep = 0; rep < 20; rep++) {
      hipMemsetAsync(bad, 0, 8, s0); hipMemsetAsync(done, 0, 8, s0);
      hipStreamSynchronize(s0);
      if (mode == 1) mfma_corunner<<<1024, 256, 0, s1>>>(sink, 40000);
      if (mode == 2) valu_corunner<<<1024, 256, 0, s1>>>(sink, 60000);
      probe<<<1024, 256, 0, s0>>>(bad, done, 20000, 1234u + rep);
      hipDeviceSynchronize();
      unsigned long long b = 0, d = 0;
      hipMemcpy(&b, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&d, done, 8, hipMemcpyDeviceToHost);
      tb += b; td += d;
    }
    printf("%-34s %llu wrong of %llu packed multiplies\n", names[mode], tb, td);
  }
  return 0;
}
