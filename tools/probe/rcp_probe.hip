// accuracy of the gfx950 fp64 reciprocal / rsqrt seeds and of 1 vs 2 Newton steps (development probe)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const double* x, double* out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double v = x[i];
  double r0 = __builtin_amdgcn_rcp(v);
  double r1 = fma(fma(-v, r0, 1.0), r0, r0);
  double r2 = fma(fma(-v, r1, 1.0), r1, r1);
  double q0 = __builtin_amdgcn_rsq(v);
  double q1 = fma(fma(-0.5 * v * q0, q0, 0.5), q0, q0);
  double q2 = fma(fma(-0.5 * v * q1, q1, 0.5), q1, q1);
  out[i * 6 + 0] = r0; out[i * 6 + 1] = r1; out[i * 6 + 2] = r2;
  out[i * 6 + 3] = q0; out[i * 6 + 4] = q1; out[i * 6 + 5] = q2;
}
int main() {
  const int n = 1 << 16;
  double *hx = new double[n], *ho = new double[n * 6];
  for (int i = 0; i < n; ++i) hx[i] = exp(((double)rand() / RAND_MAX) * 60.0 - 30.0) * (1.0 + (double)rand() / RAND_MAX);
  double *dx, *dout;
  hipMalloc(&dx, n * 8); hipMalloc(&dout, n * 48);
  hipMemcpy(dx, hx, n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
  hipMemcpy(ho, dout, n * 48, hipMemcpyDeviceToHost);
  double e[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < n; ++i) {
    const double rr = 1.0 / hx[i], qq = 1.0 / sqrt(hx[i]);
    for (int j = 0; j < 3; ++j) e[j] = fmax(e[j], fabs(ho[i * 6 + j] / rr - 1.0));
    for (int j = 3; j < 6; ++j) e[j] = fmax(e[j], fabs(ho[i * 6 + j] / qq - 1.0));
  }
  printf("rcp: seed %.3e  1 step %.3e  2 steps %.3e\nrsq: seed %.3e  1 step %.3e  2 steps %.3e\n", e[0], e[1], e[2], e[3], e[4], e[5]);
  return 0;
}
