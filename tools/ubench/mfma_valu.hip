// Microbenchmark: do v_mfma_f32_16x16x32_bf16 and VALU work overlap on gfx950, within a wave and across
// the waves of a SIMD?  Build: hipcc -O3 --offload-arch=gfx950 mfma_valu.hip -o mfma_valu ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

template <int NM, int NV, int CH>   // per iteration: NM MFMAs over CH independent accumulators, NV VALU fmas (4 chains)
__global__ void k(float* out, unsigned long long* cyc, int iters) {
  const int lane = threadIdx.x & 63;
  bf8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (lane + e)); b[e] = (__bf16)(0.002f * (lane - e)); }
  f4 acc[CH];
  for (int c = 0; c < CH; ++c) acc[c] = f4{0.f, 0.f, 0.f, 0.f};
  float v0 = lane * 0.5f, v1 = lane * 0.25f, v2 = 1.f, v3 = 2.f;
  const float m = 1.0001f, d = 0.0001f;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < (NM > NV / 4 ? NM : NV / 4); ++i) {
      if (i < NM) acc[i % CH] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i % CH], 0, 0, 0);
      if (4 * i < NV) {
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v0) : "v"(m), "v"(d));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v1) : "v"(m), "v"(d));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v2) : "v"(m), "v"(d));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v3) : "v"(m), "v"(d));
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = v0 + v1 + v2 + v3;
  for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (lane == 0 && blockIdx.x == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

template <int NM, int NV, int CH>
void run(const char* name, int threads) {
  float* out; unsigned long long* cyc; unsigned long long h[16];
  hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 128);
  const int iters = 2000;
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<NM, NV, CH>), dim3(1), dim3(threads), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  hipMemcpy(h, cyc, 128, hipMemcpyDeviceToHost);
  unsigned long long lo = ~0ull, hi = 0;
  for (int w = 0; w < threads / 64; ++w) { if (h[w] < lo) lo = h[w]; if (h[w] > hi) hi = h[w]; }
  printf("%-44s threads %4d : min %7.1f max %7.1f ticks/iter  (%d MFMA + %d VALU per wave-iter)\n", name, threads,
         (double)lo / iters, (double)hi / iters, NM, NV);
  hipFree(out); hipFree(cyc);
}

int main() {
  // s_memtime tick rate vs shader clock: compare with a known-latency VALU chain
  run<0, 64, 1>("VALU only 64 fma (4 chains)", 64);
  run<0, 64, 1>("VALU only, 2 waves/SIMD", 512);
  run<16, 0, 4>("MFMA only 16 (4 acc chains)", 64);
  run<16, 0, 2>("MFMA only 16 (2 acc chains)", 64);
  run<16, 0, 1>("MFMA only 16 (1 acc chain, dependent)", 64);
  run<16, 0, 4>("MFMA only, 1 wave/SIMD (256 thr)", 256);
  run<16, 0, 4>("MFMA only, 2 waves/SIMD (512 thr)", 512);
  run<16, 64, 4>("MFMA 16 + VALU 64 interleaved, 1 wave", 64);
  run<16, 32, 4>("MFMA 16 + VALU 32 interleaved, 1 wave", 64);
  run<16, 64, 4>("MFMA 16 + VALU 64, 2 waves/SIMD", 512);
  run<16, 64, 2>("MFMA 16 + VALU 64, 2 chains, 2 waves/SIMD", 512);
  run<16, 128, 4>("MFMA 16 + VALU 128, 2 waves/SIMD", 512);
  run<16, 128, 4>("MFMA 16 + VALU 128, 1 wave", 64);
  run<16, 16, 4>("MFMA 16 + VALU 16, 1 wave", 64);
  run<0, 64, 1>("VALU only, 4 waves/SIMD", 1024);
  run<16, 0, 4>("MFMA only, 4 waves/SIMD", 1024);
  return 0;
}
