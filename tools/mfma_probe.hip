// mfma_probe.hip — what limits an fp32-MFMA slab loop on gfx950?  Development probe (not shipped).
// Variants (runtime arg): 0 = MFMA only; 1 = + ds_read_b128 fragments per tile; 2 = + 2 barriers per slab;
// 3 = + LDS staging writes per slab (from registers, no global loads).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int LD = 36;

template <int MODE, int NT, int ORDER>
__global__ __launch_bounds__(256, 2) void probe(float* out, int nslab) {
  __shared__ float lds[(64 + 256) * LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5, wm = wave >> 1, wn = wave & 1;
  for (int i = tid; i < (64 + 256) * LD; i += 256) lds[i] = (float)(i % 7) * 0.01f;
  __syncthreads();
  float* As = lds; float* Bs = lds + 64 * LD;
  f32x16 acc[NT];
  for (int n = 0; n < NT; n++) for (int r = 0; r < 16; r++) acc[n][r] = 0.f;
  float4 st[10];
  for (int i = 0; i < 10; i++) st[i] = make_float4(tid * 0.001f, i, 1.f, 2.f);
  const int cr = tid >> 3, ck = (tid & 7) * 4;
  float4 a4 = make_float4(1.f, 0.5f, 0.25f, 2.f), bb = make_float4(0.1f, 0.2f, 0.3f, 0.4f);
  for (int s = 0; s < nslab; s++) {
    if (MODE >= 2) __syncthreads();
    if (MODE >= 3) {
#pragma unroll
      for (int i = 0; i < 2; i++) *reinterpret_cast<float4*>(As + (cr + 32 * i) * LD + ck) = st[i];
#pragma unroll
      for (int i = 0; i < 8; i++) *reinterpret_cast<float4*>(Bs + (cr + 32 * i) * LD + ck) = st[2 + i];
    }
    if (MODE >= 2) __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
      if (MODE >= 1) a4 = *reinterpret_cast<const float4*>(As + (32 * wm + lr) * LD + ks * 8 + 4 * lh);
      if (ORDER == 0) {
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
          float4 b4 = bb;
          if (MODE >= 1) b4 = *reinterpret_cast<const float4*>(Bs + (32 * (wn * NT + nt) + lr) * LD + ks * 8 + 4 * lh);
          acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc[nt], 0, 0, 0);
          acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc[nt], 0, 0, 0);
          acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc[nt], 0, 0, 0);
          acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc[nt], 0, 0, 0);
        }
      } else {
        float4 b4[NT];
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
          b4[nt] = bb;
          if (MODE >= 1) b4[nt] = *reinterpret_cast<const float4*>(Bs + (32 * (wn * NT + nt) + lr) * LD + ks * 8 + 4 * lh);
        }
#pragma unroll
        for (int nt = 0; nt < NT; nt++) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4[nt].x, acc[nt], 0, 0, 0);
#pragma unroll
        for (int nt = 0; nt < NT; nt++) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4[nt].y, acc[nt], 0, 0, 0);
#pragma unroll
        for (int nt = 0; nt < NT; nt++) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4[nt].z, acc[nt], 0, 0, 0);
#pragma unroll
        for (int nt = 0; nt < NT; nt++) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4[nt].w, acc[nt], 0, 0, 0);
      }
    }
  }
  float s = 0;
  for (int n = 0; n < NT; n++) for (int r = 0; r < 16; r++) s += acc[n][r];
  out[blockIdx.x * 256 + tid] = s;
}

template <int MODE, int ORDER>
void run(const char* name, int blocks, int nslab, float* out) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<MODE, 4, ORDER>), dim3(blocks), dim3(256), 0, 0, out, nslab);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0, 0);
  for (int i = 0; i < 5; i++) hipLaunchKernelGGL((probe<MODE, 4, ORDER>), dim3(blocks), dim3(256), 0, 0, out, nslab);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  const double fl = (double)blocks * 4 /*waves*/ * nslab * 4 /*ks*/ * 4 /*NT*/ * 4 * 4096.0;
  printf("%-40s blocks=%d  %.3f ms  %.1f TFLOP/s (%.1f%%)\n", name, blocks, ms, fl / ms / 1e9, fl / ms / 1e9 / 157.3 * 100);
}

int main(int argc, char** argv) {
  const int per_cu = argc > 1 ? atoi(argv[1]) : 3;
  const int blocks = 256 * per_cu, nslab = 256;
  float* out; (void)hipMalloc(&out, sizeof(float) * blocks * 256);
  run<0, 0>("mfma only, chained x4 per acc", blocks, nslab, out);
  run<0, 1>("mfma only, accumulators interleaved", blocks, nslab, out);
  run<1, 0>("+ds_read, chained", blocks, nslab, out);
  run<1, 1>("+ds_read, interleaved", blocks, nslab, out);
  run<2, 0>("+2 barriers/slab, chained", blocks, nslab, out);
  run<2, 1>("+2 barriers/slab, interleaved", blocks, nslab, out);
  run<3, 0>("+LDS staging writes, chained", blocks, nslab, out);
  run<3, 1>("+LDS staging writes, interleaved", blocks, nslab, out);
  return 0;
}
