// Shared device/host helpers for the gfx950 kernels.  Wavefront = 64 lanes, hard-coded.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/graspqp_hip.h"  // every definition is checked against the published prototypes

#define GQ_WAVE 64
#define GQ_INF_F __builtin_inff()

// ---- error plumbing across the C ABI (no exceptions; 0 = ok) -------------------------------------
extern "C" void gq_set_error_(const char* msg);
#define GQ_FAIL(code, ...)                          \
  do {                                              \
    char gq_buf_[512];                              \
    snprintf(gq_buf_, sizeof(gq_buf_), __VA_ARGS__); \
    gq_set_error_(gq_buf_);                         \
    return (code);                                  \
  } while (0)
#define GQ_CHECK_HIP(expr)                                                               \
  do {                                                                                   \
    hipError_t gq_e_ = (expr);                                                           \
    if (gq_e_ != hipSuccess) GQ_FAIL(3, "%s failed: %s", #expr, hipGetErrorString(gq_e_)); \
  } while (0)
#define GQ_REQUIRE(cond, ...) \
  do {                        \
    if (!(cond)) GQ_FAIL(2, __VA_ARGS__); \
  } while (0)
#define GQ_LAUNCH_CHECK() GQ_CHECK_HIP(hipGetLastError())

enum { GQ_OK = 0, GQ_ERR_ARG = 2, GQ_ERR_HIP = 3, GQ_ERR_UNSUPPORTED = 4 };

// ---- lane helpers -----------------------------------------------------------------------------------
__device__ __forceinline__ int gq_lane() { return threadIdx.x & (GQ_WAVE - 1); }

// value of `v` in lane `l` (l must be wave-uniform); result is wave-uniform (lives in an SGPR)
__device__ __forceinline__ float gq_readlane(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}
__device__ __forceinline__ int gq_readlane_i(int v, int l) { return __builtin_amdgcn_readlane(v, l); }

__device__ __forceinline__ float gq_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, GQ_WAVE);
  return v;
}
__device__ __forceinline__ double gq_wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, GQ_WAVE);
  return v;
}
// NaN-propagating min / max (torch.min / torch.max semantics)
__device__ __forceinline__ float gq_nanmin(float a, float b) {  // branch-free: v_min + unordered compare + select
  const float m = fminf(a, b);
  return __builtin_isunordered(a, b) ? __builtin_nanf("") : m;
}
__device__ __forceinline__ float gq_nanmax(float a, float b) {
  const float m = fmaxf(a, b);
  return __builtin_isunordered(a, b) ? __builtin_nanf("") : m;
}
__device__ __forceinline__ float gq_wave_nanmin(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = gq_nanmin(v, __shfl_xor(v, o, GQ_WAVE));
  return v;
}

// ---- DPP wave reductions (gfx9 row_shr / row_bcast network; result taken from lane 63, returned wave-uniform) ----
// Same network as rocPRIM's warp_reduce_dpp: quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_shr:4, row_shr:8,
// row_bcast:15 (rows 1,3), row_bcast:31 (rows 2,3).  Lanes that shift in nothing read 0; their partial results
// never reach lane 63, so the network is valid for any associative op.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float gq_dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double gq_dpp_d(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, ROW_MASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK, 0xf, false);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ float gq_dpp_sum(float v) {
  v += gq_dpp<0xb1, 0xf>(v);
  v += gq_dpp<0x4e, 0xf>(v);
  v += gq_dpp<0x114, 0xf>(v);
  v += gq_dpp<0x118, 0xf>(v);
  v += gq_dpp<0x142, 0xa>(v);
  v += gq_dpp<0x143, 0xc>(v);
  return gq_readlane(v, 63);
}
__device__ __forceinline__ double gq_dpp_sum_d(double v) {
  v += gq_dpp_d<0xb1, 0xf>(v);
  v += gq_dpp_d<0x4e, 0xf>(v);
  v += gq_dpp_d<0x114, 0xf>(v);
  v += gq_dpp_d<0x118, 0xf>(v);
  v += gq_dpp_d<0x142, 0xa>(v);
  v += gq_dpp_d<0x143, 0xc>(v);
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), 63);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// plain (NaN-ignoring) min over the wave
__device__ __forceinline__ float gq_dpp_min(float v) {
  v = fminf(v, gq_dpp<0xb1, 0xf>(v));
  v = fminf(v, gq_dpp<0x4e, 0xf>(v));
  v = fminf(v, gq_dpp<0x114, 0xf>(v));
  v = fminf(v, gq_dpp<0x118, 0xf>(v));
  v = fminf(v, gq_dpp<0x142, 0xa>(v));
  v = fminf(v, gq_dpp<0x143, 0xc>(v));
  return gq_readlane(v, 63);
}
// NaN-propagating min over the wave (torch.min semantics)
__device__ __forceinline__ float gq_dpp_nanmin(float v) {
  v = gq_nanmin(v, gq_dpp<0xb1, 0xf>(v));
  v = gq_nanmin(v, gq_dpp<0x4e, 0xf>(v));
  v = gq_nanmin(v, gq_dpp<0x114, 0xf>(v));
  v = gq_nanmin(v, gq_dpp<0x118, 0xf>(v));
  v = gq_nanmin(v, gq_dpp<0x142, 0xa>(v));
  v = gq_nanmin(v, gq_dpp<0x143, 0xc>(v));
  return gq_readlane(v, 63);
}

// ---- small vector math --------------------------------------------------------------------------------
struct gq3 {
  float x, y, z;
};
__device__ __forceinline__ gq3 gq_mk(float x, float y, float z) { return gq3{x, y, z}; }
__device__ __forceinline__ gq3 operator+(gq3 a, gq3 b) { return gq3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ gq3 operator-(gq3 a, gq3 b) { return gq3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ gq3 operator*(float s, gq3 a) { return gq3{s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ float gq_dot(gq3 a, gq3 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }
__device__ __forceinline__ gq3 gq_cross(gq3 a, gq3 b) {
  return gq3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
// y = M x, M row-major 3x3 at m[0..8]
__device__ __forceinline__ gq3 gq_mv(const float* m, gq3 v) {
  return gq3{m[0] * v.x + m[1] * v.y + m[2] * v.z, m[3] * v.x + m[4] * v.y + m[5] * v.z,
             m[6] * v.x + m[7] * v.y + m[8] * v.z};
}
// y = M^T x
__device__ __forceinline__ gq3 gq_mtv(const float* m, gq3 v) {
  return gq3{m[0] * v.x + m[3] * v.y + m[6] * v.z, m[1] * v.x + m[4] * v.y + m[7] * v.z,
             m[2] * v.x + m[5] * v.y + m[8] * v.z};
}
