// Round-2 probes (run on the MI355X box):  hipcc -O3 --offload-arch=gfx950 r2_probe.hip -o r2_probe
//  (1) ds_read_b128 throughput for the lane->address patterns of the conv kernel (pixel-major stride 112 B,
//      as shipped in round 1) against a plane layout (16 B per pixel, the two 8-channel halves 256*k B apart),
//      with the hardware's real ds_read_b128 lane groups {0-3,12-15,20-27},{4-11,16-19,28-31},...
//  (2) do VALU instructions hide under v_mfma_f32_16x16x32_{bf16,f16} / v_mfma_f32_32x32x16_bf16 when they are
//      interleaved finely (MFMA, then n VALU, ...), with one and two waves per SIMD?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// ---------------------------------------------------------------- (1) LDS patterns
template <int PAT>
__global__ __launch_bounds__(512) void k_lds(float* out, unsigned long long* cyc, int iters, int tapoff) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, j = lane & 15, qa = lane >> 4, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 40960; i += 512) ((float*)lds)[i] = (float)i;
  __syncthreads();
  int addr;
  if (PAT == 0) addr = lane * 16;                                               // contiguous
  else if (PAT == 1) addr = j * 112 + (qa & 1) * 16 + (qa >> 1) * tapoff * 112; // round-1 conv layout
  else if (PAT == 2) addr = j * 16 + (qa & 1) * 8192 + (qa >> 1) * tapoff * 16; // planes, halves 8 KiB apart
  else addr = j * 16 + (qa & 1) * 8192 + (qa >> 1) * tapoff * 16 + ((j >= 7) ? 3 * 16 : 0);  // tile row wrap (tw=7 in RW=10)
  addr += wave * 64;
  f4 acc = {0, 0, 0, 0};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const f4 v = *(const f4*)(lds + ((addr + k * 1792) & 0x1FFF0) % 65536);
      acc += v;
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
  if (lane == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}

// ---------------------------------------------------------------- (2) MFMA + VALU interleave
// SHAPE 0: 16x16x32 bf16, 1: 16x16x32 f16, 2: 32x32x16 bf16.  NV VALU fma per MFMA.
template <int SHAPE, int NV>
__global__ void k_mix(float* out, unsigned long long* cyc, int iters) {
  const int lane = threadIdx.x & 63;
  bf8 a, b; h8 ha, hb;
  for (int e = 0; e < 8; ++e) {
    a[e] = (__bf16)(0.001f * (lane + e)); b[e] = (__bf16)(0.002f * (lane - e));
    ha[e] = (_Float16)(0.001f * (lane + e)); hb[e] = (_Float16)(0.002f * (lane - e));
  }
  f4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  f16v big[2] = {{0}, {0}};
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = lane * 0.5f + i;
  const float m = 1.0001f, d = 0.0001f;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (SHAPE == 0) acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i & 3], 0, 0, 0);
      else if (SHAPE == 1) acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[i & 3], 0, 0, 0);
      else big[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, big[i & 1], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < NV; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[k & 7]) : "v"(m), "v"(d));
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += v[i];
  for (int c = 0; c < 4; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  s += big[0][0] + big[1][5];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (lane == 0 && blockIdx.x == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

static float* g_out; static unsigned long long* g_cyc;
template <typename F> void report(const char* name, int threads, int iters, double per, F launch) {
  unsigned long long h[16];
  for (int r = 0; r < 2; ++r) launch();
  hipDeviceSynchronize();
  hipMemcpy(h, g_cyc, 128, hipMemcpyDeviceToHost);
  unsigned long long lo = ~0ull, hi = 0;
  for (int w = 0; w < threads / 64; ++w) { if (h[w] < lo) lo = h[w]; if (h[w] > hi) hi = h[w]; }
  printf("%-64s thr %4d: %8.2f .. %8.2f ticks per %s\n", name, threads, lo / (double)iters / per, hi / (double)iters / per, "unit");
}

template <int PAT> void run_lds(const char* name, int tapoff) {
  const int iters = 2000;
  hipFuncSetAttribute((const void*)k_lds<PAT>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
  report(name, 512, iters, 16.0, [&] { hipLaunchKernelGGL(k_lds<PAT>, dim3(1), dim3(512), 163840, 0, g_out, g_cyc, iters, tapoff); });
}
template <int SHAPE, int NV> void run_mix(const char* name, int threads) {
  const int iters = 2000;
  report(name, threads, iters, 16.0, [&] { hipLaunchKernelGGL((k_mix<SHAPE, NV>), dim3(1), dim3(threads), 0, 0, g_out, g_cyc, iters); });
}

int main() {
  hipMalloc(&g_out, 1 << 20); hipMalloc(&g_cyc, 128);
  printf("== ds_read_b128: ticks per wave-instruction as seen by ONE wave (8 waves issue concurrently; x/8 = CU cycles per instr)\n");
  run_lds<0>("contiguous lane*16", 1);
  run_lds<1>("round-1 conv: px stride 112 B, half +16 B, tap +1 px", 1);
  run_lds<1>("round-1 conv: tap +3 px", 3);
  run_lds<2>("planes: px stride 16 B, half +8 KiB, tap +1 px", 1);
  run_lds<2>("planes: tap +3 px", 3);
  run_lds<2>("planes: tap +8 px", 8);
  run_lds<3>("planes with a row wrap inside the unit (+3 px after lane 7)", 1);
  printf("== MFMA + VALU: ticks per MFMA (16 MFMA per iteration, NV v_fma_f32 behind each)\n");
  run_mix<0, 0>("16x16x32 bf16, no VALU, 1 wave/SIMD", 256);
  run_mix<1, 0>("16x16x32 f16,  no VALU, 1 wave/SIMD", 256);
  run_mix<2, 0>("32x32x16 bf16, no VALU, 1 wave/SIMD", 256);
  run_mix<0, 1>("16x16x32 bf16 + 1 fma, 1 wave/SIMD", 256);
  run_mix<0, 2>("16x16x32 bf16 + 2 fma, 1 wave/SIMD", 256);
  run_mix<0, 4>("16x16x32 bf16 + 4 fma, 1 wave/SIMD", 256);
  run_mix<0, 8>("16x16x32 bf16 + 8 fma, 1 wave/SIMD", 256);
  run_mix<0, 2>("16x16x32 bf16 + 2 fma, 2 waves/SIMD", 512);
  run_mix<0, 4>("16x16x32 bf16 + 4 fma, 2 waves/SIMD", 512);
  run_mix<0, 8>("16x16x32 bf16 + 8 fma, 2 waves/SIMD", 512);
  run_mix<1, 4>("16x16x32 f16 + 4 fma, 2 waves/SIMD", 512);
  run_mix<2, 2>("32x32x16 bf16 + 2 fma, 1 wave/SIMD", 256);
  run_mix<2, 4>("32x32x16 bf16 + 4 fma, 1 wave/SIMD", 256);
  run_mix<2, 6>("32x32x16 bf16 + 6 fma, 1 wave/SIMD", 256);
  run_mix<2, 8>("32x32x16 bf16 + 8 fma, 1 wave/SIMD", 256);
  run_mix<2, 16>("32x32x16 bf16 + 16 fma, 1 wave/SIMD", 256);
  run_mix<2, 8>("32x32x16 bf16 + 8 fma, 2 waves/SIMD", 512);
  run_mix<2, 16>("32x32x16 bf16 + 16 fma, 2 waves/SIMD", 512);
  run_mix<0, 0>("16x16x32 bf16, no VALU, 2 waves/SIMD", 512);
  run_mix<2, 0>("32x32x16 bf16, no VALU, 2 waves/SIMD", 512);
  return 0;
}
