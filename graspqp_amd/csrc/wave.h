// gfx950 wave64 helpers for the low-rank QP kernels: multi-value fp64 sums over the wavefront and Newton-refined fp64
// reciprocals.
#pragma once
#include "common.h"

// 1/x and 1/sqrt(x) in fp64: hardware seed (v_rcp_f64 / v_rsq_f64, measured relative error 4.5e-8 / 5.1e-8 on gfx950,
// tools/probe/rcp_probe.hip) + ONE Newton step -> 2e-15 / 4e-15, far below what the 6x6 systems need.  A full IEEE
// division costs ~12 dependent fp64 instructions; the factorisation runs wave-uniform on a single wave per SIMD, so
// every dependent instruction is paid at full latency.
__device__ __forceinline__ double gq_rcp_d(double x) {
  const double r = __builtin_amdgcn_rcp(x);
  return fma(fma(-x, r, 1.0), r, r);
}
__device__ __forceinline__ double gq_rsq_d(double x) {
  const double r = __builtin_amdgcn_rsq(x);
  return fma(fma(-0.5 * x * r, r, 0.5), r, r);
}

struct GqD2 {
  int lo, hi;
};
__device__ __forceinline__ GqD2 gq_split_d(double v) {
  const long long b = __double_as_longlong(v);
  return GqD2{(int)(b & 0xffffffffll), (int)(b >> 32)};
}
__device__ __forceinline__ double gq_join_d(int lo, int hi) {
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
template <int CTRL>
__device__ __forceinline__ double gq_dpp_all_d(double v) {  // full-row DPP pattern, every lane has a source
  const GqD2 s = gq_split_d(v);
  return gq_join_d(__builtin_amdgcn_mov_dpp(s.lo, CTRL, 0xf, 0xf, true), __builtin_amdgcn_mov_dpp(s.hi, CTRL, 0xf, 0xf, true));
}
// x + y folded across the two wave halves: lanes 0..31 <- x[l] + x[l+32], lanes 32..63 <- y[l-32] + y[l]
// (v_permlane32_swap: swaps lanes 32..63 of the first operand with lanes 0..31 of the second; gfx950)
__device__ __forceinline__ double gq_fold32_d(double x, double y) {
  const GqD2 a = gq_split_d(x), b = gq_split_d(y);
  const auto lo = __builtin_amdgcn_permlane32_swap(a.lo, b.lo, false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap(a.hi, b.hi, false, false);
  return gq_join_d(lo[0], hi[0]) + gq_join_d(lo[1], hi[1]);
}
// same across row pairs: rows 0,2 <- x[row] + x[row+1], rows 1,3 <- y[row-1] + y[row]
// (v_permlane16_swap: swaps the odd rows of the first operand with the even rows of the second)
__device__ __forceinline__ double gq_fold16_d(double x, double y) {
  const GqD2 a = gq_split_d(x), b = gq_split_d(y);
  const auto lo = __builtin_amdgcn_permlane16_swap(a.lo, b.lo, false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap(a.hi, b.hi, false, false);
  return gq_join_d(lo[0], hi[0]) + gq_join_d(lo[1], hi[1]);
}

// v[k] <- sum over the 64 lanes of v[k], for all k at once, result wave-uniform.  Instead of K independent 6-stage
// butterflies (6K exchanges) the first two stages fold PAIRS of values into one register (the halves / row pairs of the
// wave then carry different k), so only ceil(K/4) registers go through the four in-row stages: ~K/2 + K/4 + 4K/4
// exchanges.  The summation tree is fixed, so results are bitwise reproducible.
template <int K>
__device__ __forceinline__ void gq_wave_sums_d(double (&v)[K]) {
  constexpr int K1 = (K + 1) / 2, K2 = (K1 + 1) / 2;
  double s1[K1], s2[K2];
#pragma unroll
  for (int i = 0; i < K1; ++i) s1[i] = gq_fold32_d(v[2 * i], (2 * i + 1 < K) ? v[2 * i + 1] : 0.0);
#pragma unroll
  for (int i = 0; i < K2; ++i) s2[i] = gq_fold16_d(s1[2 * i], (2 * i + 1 < K1) ? s1[2 * i + 1] : 0.0);
#ifdef GQ_SUMS_STAGE_MAJOR  // A/B: the K2 independent chains advance stage by stage (fills DPP hazard slots with other chains)
#pragma unroll
  for (int i = 0; i < K2; ++i) s2[i] += gq_dpp_all_d<0xb1>(s2[i]);
#pragma unroll
  for (int i = 0; i < K2; ++i) s2[i] += gq_dpp_all_d<0x4e>(s2[i]);
#pragma unroll
  for (int i = 0; i < K2; ++i) s2[i] += gq_dpp_all_d<0x141>(s2[i]);
#pragma unroll
  for (int i = 0; i < K2; ++i) s2[i] += gq_dpp_all_d<0x140>(s2[i]);
#else
#pragma unroll
  for (int i = 0; i < K2; ++i) {
    double t = s2[i];
    t += gq_dpp_all_d<0xb1>(t);   // quad_perm [1,0,3,2]
    t += gq_dpp_all_d<0x4e>(t);   // quad_perm [2,3,0,1]
    t += gq_dpp_all_d<0x141>(t);  // row_half_mirror
    t += gq_dpp_all_d<0x140>(t);  // row_mirror
    s2[i] = t;
  }
#endif
  // row r of s2[i] now holds: r=0 -> v[4i], r=1 -> v[4i+2], r=2 -> v[4i+1], r=3 -> v[4i+3]
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int i = k / 4, q = k % 4;
    const int src_lane = (q == 0) ? 0 : (q == 2) ? 16 : (q == 1) ? 32 : 48;
    const GqD2 s = gq_split_d(s2[i]);
    v[k] = gq_join_d(__builtin_amdgcn_readlane(s.lo, src_lane), __builtin_amdgcn_readlane(s.hi, src_lane));
  }
}

// fp32 version of gq_wave_sums_d
__device__ __forceinline__ float gq_fold32_f(float x, float y) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_int(x), __float_as_int(y), false, false);
  return __int_as_float(r[0]) + __int_as_float(r[1]);
}
__device__ __forceinline__ float gq_fold16_f(float x, float y) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_int(x), __float_as_int(y), false, false);
  return __int_as_float(r[0]) + __int_as_float(r[1]);
}
template <int CTRL>
__device__ __forceinline__ float gq_dpp_all_f(float v) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <int K>
__device__ __forceinline__ void gq_wave_sums_f(float (&v)[K]) {
  constexpr int K1 = (K + 1) / 2, K2 = (K1 + 1) / 2;
  float s1[K1], s2[K2];
#pragma unroll
  for (int i = 0; i < K1; ++i) s1[i] = gq_fold32_f(v[2 * i], (2 * i + 1 < K) ? v[2 * i + 1] : 0.0f);
#pragma unroll
  for (int i = 0; i < K2; ++i) s2[i] = gq_fold16_f(s1[2 * i], (2 * i + 1 < K1) ? s1[2 * i + 1] : 0.0f);
#ifdef GQ_SUMS_STAGE_MAJOR
#pragma unroll
  for (int i = 0; i < K2; ++i) s2[i] += gq_dpp_all_f<0xb1>(s2[i]);
#pragma unroll
  for (int i = 0; i < K2; ++i) s2[i] += gq_dpp_all_f<0x4e>(s2[i]);
#pragma unroll
  for (int i = 0; i < K2; ++i) s2[i] += gq_dpp_all_f<0x141>(s2[i]);
#pragma unroll
  for (int i = 0; i < K2; ++i) s2[i] += gq_dpp_all_f<0x140>(s2[i]);
#else
#pragma unroll
  for (int i = 0; i < K2; ++i) {
    float t = s2[i];
    t += gq_dpp_all_f<0xb1>(t);   // quad_perm [1,0,3,2]
    t += gq_dpp_all_f<0x4e>(t);   // quad_perm [2,3,0,1]
    t += gq_dpp_all_f<0x141>(t);  // row_half_mirror
    t += gq_dpp_all_f<0x140>(t);  // row_mirror
    s2[i] = t;
  }
#endif
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int i = k / 4, q = k % 4;
    const int src_lane = (q == 0) ? 0 : (q == 2) ? 16 : (q == 1) ? 32 : 48;
    v[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s2[i]), src_lane));
  }
}

// LDS hand-over between the lanes of ONE wavefront (blocks whose other wavefronts have exited, or single-wave blocks):
// LDS operations of a wavefront execute in order, so only the compiler has to be kept from moving them.
__device__ __forceinline__ void gq_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
