// mfma16_probe.hip — lane layout of v_mfma_f32_16x16x4_f32 on gfx950 (development probe).
//   hipcc -O2 --offload-arch=gfx950 -o tools/abl/mfma16_probe tools/mfma16_probe.hip && tools/abl/mfma16_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* a, const float* b, float* d) {
  const int l = threadIdx.x;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[l], b[l], acc, 0, 0, 0);
  for (int r = 0; r < 4; r++) d[4 * l + r] = acc[r];
}
int main() {
  float *a, *b, *d;
  (void)hipMalloc(&a, 256); (void)hipMalloc(&b, 256); (void)hipMalloc(&d, 1024);
  int ok = 1;
  for (int kk = 0; kk < 4; kk++) {
    float ha[64], hb[64], hd[256];
    for (int l = 0; l < 64; l++) { ha[l] = l / 16 == kk ? 2.0f * l + 1.0f : 0.f; hb[l] = l / 16 == kk ? ldexpf(1.0f, l - 20) : 0.f; }
    (void)hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); (void)hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, b, d);
    (void)hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l++)
      for (int r = 0; r < 4; r++) {
        // hypothesis: D[lane l][reg r] = A[lane 16 kk + 4 (l / 16) + r] * B[lane 16 kk + l % 16]
        const float want = ha[16 * kk + 4 * (l / 16) + r] * hb[16 * kk + l % 16];
        if (hd[4 * l + r] != want) { ok = 0; if (l < 3) printf("kk %d lane %d reg %d: got %g want %g\n", kk, l, r, hd[4 * l + r], want); }
      }
  }
  printf("hypothesis D[l][r] = sum_k A[16 k + 4 (l/16) + r] * B[16 k + l %% 16]: %s\n", ok ? "holds" : "FAILS");
  return 0;
}
