// Issue interval of the K=16 bf16 MFMA (v_mfma_f32_16x16x16_bf16) vs the K=32 one on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef short s4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf4 __attribute__((ext_vector_type(4)));

template <int K>
__global__ void k(float* out, unsigned long long* cyc, int iters) {
  const int lane = threadIdx.x & 63;
  bf8 a8, b8; bf4 a4, b4;
  for (int e = 0; e < 8; ++e) { a8[e] = (__bf16)(0.001f * (lane + e)); b8[e] = (__bf16)(0.002f * (lane - e)); }
  for (int e = 0; e < 4; ++e) { a4[e] = a8[e]; b4[e] = b8[e]; }
  f4 acc[4];
  for (int c = 0; c < 4; ++c) acc[c] = f4{0.f, 0.f, 0.f, 0.f};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (K == 32) acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[i & 3], 0, 0, 0);
      else acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s4, a4), __builtin_bit_cast(s4, b4), acc[i & 3], 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int c = 0; c < 4; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) *cyc = t1 - t0;
}

template <int K>
void run() {
  float* out; unsigned long long* cyc; unsigned long long h;
  (void)hipMalloc(&out, 4096); (void)hipMalloc(&cyc, 8);
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<K>), dim3(1), dim3(64), 0, 0, out, cyc, 2000);
  (void)hipDeviceSynchronize();
  (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("v_mfma_f32_16x16x%d bf16: %.2f cycles per MFMA\n", K, (double)h / 2000 / 16);
  (void)hipFree(out); (void)hipFree(cyc);
}
int main() { run<32>(); run<16>(); return 0; }
