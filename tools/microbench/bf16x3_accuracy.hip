// Accuracy of the 3-way bf16 split (6 product terms, fp32 accumulation on v_mfma_f32_32x32x16_bf16) against the fp32 MFMA
// the shipped kernels use, both measured against an fp64 host reference: C[32x32] = A[32xK] * B[32xK]^T, K = 4608
// (a 3x3 conv over 512 channels), operands ~ N(0,1) and, second case, with a 1e4 dynamic range across K.
//   hipcc --offload-arch=gfx950 -O3 bf16x3_accuracy.hip -o bf16x3_accuracy
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
  unsigned r;
  asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
struct Split { bf16x8 p[3]; };
__device__ __forceinline__ Split split8(const float* x) {
  unsigned w[3][4];
#pragma unroll
  for (int e = 0; e < 4; e++) {
    const float x0 = x[2 * e], x1 = x[2 * e + 1];
    const unsigned h = cvt_pk_bf16(x0, x1);
    const float r0 = x0 - __builtin_bit_cast(float, h << 16), r1 = x1 - __builtin_bit_cast(float, h & 0xffff0000u);
    const unsigned m = cvt_pk_bf16(r0, r1);
    const float s0 = r0 - __builtin_bit_cast(float, m << 16), s1 = r1 - __builtin_bit_cast(float, m & 0xffff0000u);
    w[0][e] = h; w[1][e] = m; w[2][e] = cvt_pk_bf16(s0, s1);
  }
  Split s;
#pragma unroll
  for (int pl = 0; pl < 3; pl++) {
    uint4 v = make_uint4(w[pl][0], w[pl][1], w[pl][2], w[pl][3]);
    s.p[pl] = __builtin_bit_cast(bf16x8, v);
  }
  return s;
}

template <int MODE, int NTERMS>
__global__ void gemm32(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int K) {
  const int lane = threadIdx.x, l31 = lane & 31, lh = lane >> 5;
  f32x16 acc;
  for (int r = 0; r < 16; r++) acc[r] = 0.f;
  if (MODE == 0) {
    for (int k = 0; k < K; k += 2)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[l31 * K + k + lh], B[l31 * K + k + lh], acc, 0, 0, 0);
  } else {
    const int ta[6] = {2, 0, 1, 1, 0, 0}, tb[6] = {0, 2, 1, 0, 1, 0};   // smallest terms first
    for (int k = 0; k < K; k += 16) {
      const Split a = split8(A + l31 * K + k + 8 * lh), b = split8(B + l31 * K + k + 8 * lh);
#pragma unroll
      for (int t = 6 - NTERMS; t < 6; t++) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[ta[t]], b.p[tb[t]], acc, 0, 0, 0);
    }
  }
  for (int r = 0; r < 16; r++) C[((r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + l31] = acc[r];
}

static double nrand() {
  double u = (rand() + 1.0) / (RAND_MAX + 2.0), v = (rand() + 1.0) / (RAND_MAX + 2.0);
  return sqrt(-2 * log(u)) * cos(6.283185307179586 * v);
}

int main() {
  const int K = 4608;
  for (int cs = 0; cs < 2; cs++) {
    std::vector<float> A(32 * K), B(32 * K);
    srand(1 + cs);
    for (int i = 0; i < 32 * K; i++) {
      const double sc = cs ? pow(10.0, 4.0 * (rand() / (double)RAND_MAX) - 2.0) : 1.0;
      A[i] = (float)(nrand() * sc);
      B[i] = (float)(nrand() / sc);
    }
    std::vector<double> ref(1024);
    double scale = 0;   // sum |a||b| of a typical entry: the natural unit of the error
    for (int i = 0; i < 32; i++)
      for (int j = 0; j < 32; j++) {
        double s = 0, sa = 0;
        for (int k = 0; k < K; k++) { s += (double)A[i * K + k] * B[j * K + k]; sa += fabs((double)A[i * K + k] * B[j * K + k]); }
        ref[i * 32 + j] = s; scale += sa / 1024;
      }
    float *dA, *dB, *dC;
    CK(hipMalloc(&dA, 4 * 32 * K)); CK(hipMalloc(&dB, 4 * 32 * K)); CK(hipMalloc(&dC, 4 * 1024));
    CK(hipMemcpy(dA, A.data(), 4 * 32 * K, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, B.data(), 4 * 32 * K, hipMemcpyHostToDevice));
    std::vector<float> C(1024);
    const char* names[4] = {"fp32 MFMA 32x32x2", "bf16x3, 6 terms", "bf16x3, 3 terms (hi*hi, hi*mid, mid*hi)", "bf16, 1 term (plain bf16)"};
    for (int v = 0; v < 4; v++) {
      if (v == 0) gemm32<0, 6><<<1, 64>>>(dA, dB, dC, K);
      if (v == 1) gemm32<1, 6><<<1, 64>>>(dA, dB, dC, K);
      if (v == 2) gemm32<1, 3><<<1, 64>>>(dA, dB, dC, K);
      if (v == 3) gemm32<1, 1><<<1, 64>>>(dA, dB, dC, K);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost));
      double mx = 0, rms = 0;
      for (int i = 0; i < 1024; i++) { const double e = fabs(C[i] - ref[i]) / scale; mx = fmax(mx, e); rms += e * e; }
      printf("case %d  %-42s max |err| / sum|a||b| = %.3e   rms = %.3e\n", cs, names[v], mx, sqrt(rms / 1024));
    }
    CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC));
  }
  return 0;
}
