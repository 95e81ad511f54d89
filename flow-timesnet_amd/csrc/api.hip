// Error reporting, ABI version and the MFMA lane-map self test.
#include <stdarg.h>
#include <stdio.h>
#include "ftn_common.h"

static thread_local char g_err[512] = "";

void ftn_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* ftn_last_error(void) { return g_err; }
extern "C" int ftn_abi_version(void) { return FTN_ABI_VERSION; }

// D[16][16] = A[16][8] * B[8][16], A[i][k] = i*8+k+1, B[k][j] = (k+1)*100+j, as
// two v_mfma_f32_16x16x4_f32 steps with the operand/result lane maps every
// kernel in this library assumes.  The asymmetric B catches a transposed store.
__global__ void k_selftest(float* __restrict__ out) {
  const int lane = threadIdx.x & 63, j = lane & 15, q = lane >> 4;
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < 2; ++s) {
    const int k = 4 * s + q;
    const float a = (float)(j * 8 + k + 1);       // A[i=j][k]
    const float b = (float)((k + 1) * 100 + j);   // B[k][j]
    acc = mfma16(a, b, acc);
  }
  for (int r = 0; r < 4; ++r) out[(4 * q + r) * 16 + j] = acc[r];
}

extern "C" int ftn_selftest_mfma(float* out_dev, void* stream) {
  FTN_CHECK_ARG(out_dev, "ftn_selftest_mfma: null pointer");
  hipLaunchKernelGGL(k_selftest, dim3(1), dim3(64), 0, (hipStream_t)stream, out_dev);
  FTN_CHECK_LAUNCH();
  return 0;
}

// out[i] = the kernels' GELU(in[i]) (ftn_common.h gelu_erf2), for the accuracy test.
__global__ void k_selftest_gelu(const float* __restrict__ in, float* __restrict__ out, long long n) {
  const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i + 3 < n) {
    const f4 v = act4<0>(*(const f4*)(in + i));
    *(f4*)(out + i) = v;
  } else {
    for (long long k = i; k < n; ++k) out[k] = gelu_erf(in[k]);
  }
}

extern "C" int ftn_selftest_gelu(const float* in_dev, float* out_dev, long long n, void* stream) {
  FTN_CHECK_ARG(in_dev && out_dev && n >= 1 && (((uintptr_t)in_dev | (uintptr_t)out_dev) & 15) == 0,
                "ftn_selftest_gelu: bad arguments");
  const long long nthr = (n + 3) / 4;
  hipLaunchKernelGGL(k_selftest_gelu, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in_dev,
                     out_dev, n);
  FTN_CHECK_LAUNCH();
  return 0;
}
