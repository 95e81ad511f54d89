// Issue cost (cycles per wave64 instruction, SIMD saturated by 4 waves) of the VALU ops the kernels lean on.
// Build: hipcc -O3 --offload-arch=gfx950 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));

#define REP8(x) x x x x x x x x
template <int OP>
__global__ void k(float* out, unsigned long long* cyc, int iters) {
  const int lane = threadIdx.x & 63;
  float a0 = lane * 0.5f + 1.f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
  f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
  const float m = 1.0001f, d = 0.0001f;
  const f2 pm = {m, m}, pd = {d, d};
  unsigned u = lane;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    // 8 independent chains x 8 = 64 instructions per iteration
#define EIGHT(fmt, A, B, C) REP8( \
    asm volatile(fmt : "+v"(a0) : "v"(A), "v"(B)); asm volatile(fmt : "+v"(a1) : "v"(A), "v"(B)); \
    asm volatile(fmt : "+v"(a2) : "v"(A), "v"(B)); asm volatile(fmt : "+v"(a3) : "v"(A), "v"(B)); \
    asm volatile(fmt : "+v"(a4) : "v"(A), "v"(B)); asm volatile(fmt : "+v"(a5) : "v"(A), "v"(B)); \
    asm volatile(fmt : "+v"(a6) : "v"(A), "v"(B)); asm volatile(fmt : "+v"(a7) : "v"(A), "v"(B));)
    if (OP == 0) { EIGHT("v_fma_f32 %0, %0, %1, %2", m, d, 0) }
    if (OP == 1) { REP8(
      asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p0) : "v"(pm), "v"(pd)); asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p1) : "v"(pm), "v"(pd));
      asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p2) : "v"(pm), "v"(pd)); asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p3) : "v"(pm), "v"(pd));
      asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p4) : "v"(pm), "v"(pd)); asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p5) : "v"(pm), "v"(pd));
      asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p6) : "v"(pm), "v"(pd)); asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p7) : "v"(pm), "v"(pd));) }
    if (OP == 2) { EIGHT("v_exp_f32 %0, %0 ; %1 %2", m, d, 0) }
    if (OP == 3) { EIGHT("v_rcp_f32 %0, %0 ; %1 %2", m, d, 0) }
    if (OP == 4) { EIGHT("v_max_f32 %0, %0, %1 ; %2", m, d, 0) }
    if (OP == 5) { EIGHT("v_and_b32 %0, %0, %1 ; %2", m, d, 0) }
    if (OP == 6) { EIGHT("v_perm_b32 %0, %0, %1, %2", m, d, 0) }
    if (OP == 7) { EIGHT("v_mul_f32 %0, %0, %1 ; %2", m, d, 0) }
    if (OP == 8) { EIGHT("v_log_f32 %0, %0 ; %1 %2", m, d, 0) }
    if (OP == 9) { EIGHT("v_cndmask_b32 %0, %0, %1, vcc ; %2", m, d, 0) }
    if (OP == 10) { EIGHT("v_sqrt_f32 %0, %0 ; %1 %2", m, d, 0) }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y + u;
  if (lane == 0 && blockIdx.x == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

template <int OP>
void run(const char* name) {
  float* out; unsigned long long* cyc; unsigned long long h[16];
  (void)hipMalloc(&out, 1 << 20); (void)hipMalloc(&cyc, 128);
  const int iters = 1000, threads = 1024;   // 4 waves per SIMD
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<OP>), dim3(1), dim3(threads), 0, 0, out, cyc, iters);
  (void)hipDeviceSynchronize();
  (void)hipMemcpy(h, cyc, 128, hipMemcpyDeviceToHost);
  unsigned long long hi = 0;
  for (int w = 0; w < threads / 64; ++w) if (h[w] > hi) hi = h[w];
  printf("%-16s %6.2f cycles per wave64 instruction (4 waves/SIMD, 64 per wave-iter)\n", name, (double)hi / iters / 64.0 / 4.0);
  (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
  run<0>("v_fma_f32"); run<7>("v_mul_f32"); run<1>("v_pk_fma_f32"); run<4>("v_max_f32"); run<5>("v_and_b32"); run<6>("v_perm_b32");
  run<9>("v_cndmask_b32"); run<2>("v_exp_f32"); run<3>("v_rcp_f32"); run<8>("v_log_f32"); run<10>("v_sqrt_f32");
  return 0;
}
