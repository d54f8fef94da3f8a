// gemm_bench.hip — standalone timing of the grouped fp32-MFMA GEMM kernel (development tool, not shipped).
// Build variants with -D flags, e.g.  hipcc -O3 --offload-arch=gfx950 -I../lammps-ani_amd/csrc -DVARIANT=... gemm_bench.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../lammps-ani_amd/csrc/ani_kernels_mlp.hip"

using namespace ani;

int main(int argc, char** argv) {
  const int rows = argc > 1 ? atoi(argv[1]) : 66688;
  const int N = argc > 2 ? atoi(argv[2]) : 256;
  const int K = argc > 3 ? atoi(argv[3]) : 1008;
  const int batch = argc > 4 ? atoi(argv[4]) : 1;
  const int reps = 20;
  float *A, *B, *C, *bias;
  hipMalloc(&A, sizeof(float) * (size_t)rows * K);
  hipMalloc(&B, sizeof(float) * (size_t)batch * N * K);
  hipMalloc(&C, sizeof(float) * (size_t)rows * N * batch);
  hipMalloc(&bias, sizeof(float) * N * batch);
  std::vector<float> hA((size_t)rows * K), hB((size_t)batch * N * K), hb(N * batch, 0.1f);
  for (auto& v : hA) v = (float)rand() / RAND_MAX - 0.5f;
  for (auto& v : hB) v = (float)rand() / RAND_MAX - 0.5f;
  hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(bias, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
  GemmArgs g{};
  g.A = A; g.lda = K; g.sA = 0; g.Bt = B; g.ldb = K; g.sB = (long long)N * K; g.C = C; g.ldc = N * batch; g.sC = N;
  g.bias = bias; g.sBias = N; g.rows = rows; g.row0 = 0; g.N = N; g.K = K; g.batch = batch; g.alpha = 0.1f; g.inv_alpha = 10.f;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; i++) launch_gemm(g, EPI_CELU, 0);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int i = 0; i < reps; i++) launch_gemm(g, EPI_CELU, 0);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  const double fl = 2.0 * rows * N * K * batch;
  // spot check a few entries on the host
  std::vector<float> hC((size_t)rows * N * batch);
  hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost);
  double maxerr = 0;
  for (int t = 0; t < 64; t++) {
    const int m = (t * 7919) % rows, n = (t * 104729) % N, b = t % batch;
    double acc = 0.1;
    for (int k = 0; k < K; k++) acc += (double)hA[(size_t)m * K + k] * hB[((size_t)b * N + n) * K + k];
    const double ref = acc > 0 ? acc : 0.1 * (exp(acc / 0.1) - 1);
    const double err = fabs(ref - hC[(size_t)m * N * batch + b * N + n]);
    if (err > maxerr) maxerr = err;
  }
  printf("rows=%d N=%d K=%d batch=%d  %.3f ms  %.1f TFLOP/s (%.1f%% of 157.3)  maxerr=%.2e\n", rows, N, K, batch, ms, fl / ms / 1e9,
         fl / ms / 1e9 / 157.3 * 100, maxerr);
  return 0;
}
