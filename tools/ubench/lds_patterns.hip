// Which ds_read_b128 address patterns are bank-conflict free on gfx950?  Times 8 waves hammering the LDS
// with one pattern each run.  Build: hipcc -O3 --offload-arch=gfx950 lds_patterns.hip -o lds_patterns
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void k(const int* offs, float* out, unsigned long long* cyc, int iters) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) ((f4*)lds)[i % 4096] = f4{1.f, 2.f, 3.f, 4.f};
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int off = offs[lane];
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const f4 v = *(const f4*)(lds + ((off + r * 1792 + it * 16) & 0xfff0));
      acc += v;
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
  if (lane == 0 && blockIdx.x == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

static void run(const char* name, const int* h_off) {
  int* d_off; float* out; unsigned long long* cyc; unsigned long long h[8];
  (void)hipMalloc(&d_off, 256); (void)hipMalloc(&out, 1 << 16); (void)hipMalloc(&cyc, 64);
  (void)hipMemcpy(d_off, h_off, 256, hipMemcpyHostToDevice);
  const int iters = 2000;
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(k, dim3(1), dim3(512), 65536, 0, d_off, out, cyc, iters);
  (void)hipDeviceSynchronize();
  (void)hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
  unsigned long long hi = 0;
  for (int w = 0; w < 8; ++w) if (h[w] > hi) hi = h[w];
  printf("%-58s %6.2f cycles per ds_read_b128 (8 waves)\n", name, (double)hi / iters / 8.0 / 8.0);
  (void)hipFree(d_off); (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
  int o[64];
  (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  for (int l = 0; l < 64; ++l) o[l] = l * 16;
  run("contiguous lane*16", o);
  for (int s : {96, 112, 128, 144, 176, 80, 48, 16}) {
    char nm[96];
    for (int l = 0; l < 64; ++l) { const int j = l & 15, qa = l >> 4; o[l] = (j + (qa >> 1)) * s + (qa & 1) * 16; }
    snprintf(nm, sizeof nm, "conv pattern: (j + qa>>1)*%d + (qa&1)*16", s);
    run(nm, o);
    for (int l = 0; l < 64; ++l) { const int j = l & 15, qa = l >> 4; o[l] = j * s + qa * 4096; }
    snprintf(nm, sizeof nm, "  16 lanes stride %d, qa groups far apart (+4096)", s);
    run(nm, o);
    for (int l = 0; l < 64; ++l) { const int j = l & 15, qa = l >> 4; o[l] = j * s + (qa & 1) * 16 + (qa >> 1) * 4096; }
    snprintf(nm, sizeof nm, "  stride %d, (qa&1)*16, qa>>1 far apart", s);
    run(nm, o);
  }
  // conv pattern (stride 112) with some lanes (taps outside the grid) redirected to a zero pixel
  for (int nz : {1, 3, 6}) {
    char nm[96];
    for (int l = 0; l < 64; ++l) { const int j = l & 15, qa = l >> 4; const bool z = j < nz; o[l] = z ? 400 * 112 + (qa & 1) * 16 : (j + (qa >> 1)) * 112 + (qa & 1) * 16; }
    snprintf(nm, sizeof nm, "conv pattern, %d of 16 pixels -> one shared zero pixel", nz);
    run(nm, o);
    for (int l = 0; l < 64; ++l) { const int j = l & 15, qa = l >> 4; const int np = j + (qa >> 1); const bool z = j < nz; o[l] = (z ? (400 | (np & 15)) : np) * 112 + (qa & 1) * 16; }
    snprintf(nm, sizeof nm, "  same, zero pixels bank-matched (index & 15)");
    run(nm, o);
    for (int l = 0; l < 64; ++l) { const int j = l & 15, qa = l >> 4; const int np = j + (qa >> 1); const bool z = j < nz; o[l] = (z ? (400 | (np & 7)) : np) * 112 + (qa & 1) * 16; }
    snprintf(nm, sizeof nm, "  same, zero pixels bank-matched (index & 7)");
    run(nm, o);
  }
  // row wrap inside a unit: pixels 0..9 in one region row, 10..15 in the next (region 6 pixels wider than the tile)
  for (int l = 0; l < 64; ++l) { const int j = l & 15, qa = l >> 4; const int np = j + (qa >> 1) + (j >= 10 ? 6 : 0); o[l] = np * 112 + (qa & 1) * 16; }
  run("conv pattern with a row wrap (+6 pixels from lane 10)", o);
  return 0;
}
