// Hardware-semantics probes for gfx950 (run on the GPU box; prints what the instructions actually do):
//   A. ds_read_b64_tr_b16 with linear per-lane addresses (lane * 8 bytes)
//   B. ds_read_b64_tr_b16 as the filter-gradient kernels use it: [k][m] bf16 image with row pitch P, a 16-lane group reads
//      4 k-rows x 16 m-columns and lane i should receive column m0 + i, rows k0..k0+3
//   C. buffer_load_dwordx4 ... lds with out-of-range offsets: does the DMA write zeros?
//   D. 16-byte buffer loads from 8-byte-aligned addresses
// build: hipcc --offload-arch=gfx950 -O2 probe_semantics.hip -o probe_semantics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef short s4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define LDSP(p) ((s4 __attribute__((address_space(3)))*)(p))

__global__ void probe_a(short* out) {
  __shared__ __attribute__((aligned(16))) short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
  __syncthreads();
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDSP(lds + threadIdx.x * 4));
  for (int j = 0; j < 4; j++) out[threadIdx.x * 4 + j] = v[j];
}

// image[k][m] = k * 256 + m (k < 16, m < 128), pitch P shorts; lane l: group g = l >> 4 handles m0 = 16 * (g & 1), k0 = 4 * (g >> 1) + kbase
__global__ void probe_b(short* out, int pitch) {
  __shared__ __attribute__((aligned(16))) short lds[16 * 256];
  for (int i = threadIdx.x; i < 16 * 128; i += 64) {
    const int k = i / 128, m = i % 128;
    lds[k * pitch + m] = (short)(k * 256 + m);
  }
  __syncthreads();
  const int l = threadIdx.x, i = l & 15, g = l >> 4;
  const int m0 = 16 * (g & 1), k0 = 4 * (g >> 1);
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDSP(lds + (k0 + (i >> 2)) * pitch + m0 + 4 * (i & 3)));
  for (int j = 0; j < 4; j++) out[l * 4 + j] = v[j];
}

__global__ void probe_c(const float* src, int n_valid_bytes, float* out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  for (int i = threadIdx.x; i < 512; i += 64) lds[i] = -7.f;
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, n_valid_bytes, 0x00020000);
  // lanes 0..31 in range, lanes 32..47 past num_records, lanes 48..63 a huge (marked) offset
  int voff = threadIdx.x * 16;
  if (threadIdx.x >= 48) voff = 0x40000000;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, voff, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += 64) out[i] = lds[i];
}

__global__ void probe_d(const float* src, float* out) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 4096, 0x00020000);
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, 8 + threadIdx.x * 24, 0, 0);   // 8-byte aligned, 24-byte pitch
  out[threadIdx.x * 4 + 0] = __uint_as_float(v.x);
  out[threadIdx.x * 4 + 1] = __uint_as_float(v.y);
  out[threadIdx.x * 4 + 2] = __uint_as_float(v.z);
  out[threadIdx.x * 4 + 3] = __uint_as_float(v.w);
}

int main() {
  short* ds;
  hipMalloc(&ds, 64 * 4 * sizeof(short));
  std::vector<short> h(256);
  // ---- A
  probe_a<<<1, 64>>>(ds);
  hipMemcpy(h.data(), ds, 512, hipMemcpyDeviceToHost);
  printf("A: ds_read_tr16_b64, lane address = lane*8 bytes (LDS holds its own short index)\n");
  int okA = 1;
  for (int l = 0; l < 64; l++) {
    printf("  lane %2d:", l);
    for (int j = 0; j < 4; j++) {
      printf(" %4d", h[l * 4 + j]);
      if (h[l * 4 + j] != (l & 15) + j * 16 + (l >> 4) * 64) okA = 0;
    }
    printf("\n");
  }
  printf("A hypothesis lds[(l&15) + j*16 + (l>>4)*64]: %s\n", okA ? "CONFIRMED" : "REFUTED");
  // ---- B
  for (int pitch : {128, 160}) {
    probe_b<<<1, 64>>>(ds, pitch);
    hipMemcpy(h.data(), ds, 512, hipMemcpyDeviceToHost);
    int ok = 1;
    for (int l = 0; l < 64; l++) {
      const int i = l & 15, g = l >> 4, m0 = 16 * (g & 1), k0 = 4 * (g >> 1);
      for (int j = 0; j < 4; j++)
        if (h[l * 4 + j] != (k0 + j) * 256 + m0 + i) ok = 0;
    }
    printf("B pitch %d: lane i of a 16-lane group gets (k0+j, m0+i), j=0..3: %s\n", pitch, ok ? "CONFIRMED" : "REFUTED");
    if (!ok)
      for (int l = 0; l < 64; l++)
        printf("  lane %2d: k,m = (%d,%d) (%d,%d) (%d,%d) (%d,%d)\n", l, h[l * 4] >> 8, h[l * 4] & 255, h[l * 4 + 1] >> 8,
               h[l * 4 + 1] & 255, h[l * 4 + 2] >> 8, h[l * 4 + 2] & 255, h[l * 4 + 3] >> 8, h[l * 4 + 3] & 255);
  }
  // ---- C
  float *src, *out;
  hipMalloc(&src, 4096);
  hipMalloc(&out, 4096);
  std::vector<float> hs(1024), ho(1024);
  for (int i = 0; i < 1024; i++) hs[i] = 1.f + i;
  hipMemcpy(src, hs.data(), 4096, hipMemcpyHostToDevice);
  probe_c<<<1, 64, 4096>>>(src, 32 * 16, out);
  hipMemcpy(ho.data(), out, 2048, hipMemcpyDeviceToHost);
  int in_ok = 1, oob_zero = 1, mark_zero = 1, untouched = 1;
  for (int i = 0; i < 128; i++) in_ok &= ho[i] == hs[i];
  for (int i = 128; i < 192; i++) oob_zero &= ho[i] == 0.f;
  for (int i = 192; i < 256; i++) mark_zero &= ho[i] == 0.f;
  for (int i = 256; i < 512; i++) untouched &= ho[i] == -7.f;
  printf("C buffer_load_dwordx4 lds: in-range data %s; past num_records -> zeros %s (first %g); marked offset -> zeros %s (first %g); "
         "rest untouched %s\n", in_ok ? "OK" : "WRONG", oob_zero ? "YES" : "NO", ho[128], mark_zero ? "YES" : "NO", ho[192],
         untouched ? "yes" : "NO");
  // ---- D
  probe_d<<<1, 64>>>(src, out);
  hipMemcpy(ho.data(), out, 1024, hipMemcpyDeviceToHost);
  int okD = 1;
  for (int l = 0; l < 64; l++)
    for (int j = 0; j < 4; j++) okD &= ho[l * 4 + j] == hs[2 + l * 6 + j];
  printf("D 16-byte buffer load at 8-byte-aligned addresses: %s\n", okD ? "correct" : "WRONG");
  return 0;
}
