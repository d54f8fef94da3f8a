// mfma4_probe.hip — lane layout of v_mfma_f32_4x4x1_16b_f32 on gfx950 (development probe).
// 16 independent 4x4 outer products per instruction: D[b][i][j] += A[b][i] * B[b][j].
//   hipcc -O2 --offload-arch=gfx950 -o tools/abl/mfma4_probe tools/mfma4_probe.hip && tools/abl/mfma4_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* a, const float* b, float* d) {
  const int l = threadIdx.x;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, 0, 0, 0);
  for (int r = 0; r < 4; r++) d[4 * l + r] = acc[r];
}
int main() {
  float ha[64], hb[64], hd[256];
  for (int l = 0; l < 64; l++) { ha[l] = 2.0f * l + 1.0f; hb[l] = ldexpf(1.0f, l - 20); }   // odd x power of two: the product identifies both lanes
  float *a, *b, *d;
  (void)hipMalloc(&a, 256); (void)hipMalloc(&b, 256); (void)hipMalloc(&d, 1024);
  (void)hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); (void)hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, b, d);
  (void)hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost);
  int ok = 1;
  for (int l = 0; l < 64; l++)
    for (int r = 0; r < 4; r++) {
      const float v = hd[4 * l + r];
      int la = -1, lb = -1;
      for (int x = 0; x < 64 && la < 0; x++)
        for (int y = 0; y < 64; y++)
          if (v == ha[x] * hb[y]) { la = x; lb = y; break; }
      if (l < 8 || l >= 60) printf("lane %2d reg %d = A[lane %2d] * B[lane %2d]\n", l, r, la, lb);
      // hypothesis: block = l / 4, column j = l % 4 (B lane = 4*block + j), row i = r (A lane = 4*block + r)
      if (la != 4 * (l / 4) + r || lb != l) ok = 0;
    }
  printf("hypothesis D[lane l][reg r] = A[4*(l/4) + r] * B[l]: %s\n", ok ? "holds" : "FAILS");
  return 0;
}
