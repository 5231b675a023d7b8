// Shared host/device helpers for libpita_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <mutex>
#include <unordered_map>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../include/pita_hip.h"

namespace pita {

// ---- error plumbing: thread-local message, integer codes (no exceptions cross the C ABI)
inline char* err_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}
inline int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}
#define PITA_HIP_CHECK(expr)                                                              \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess)                                                                 \
      return ::pita::fail(PITA_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                          __FILE__, __LINE__);                                            \
  } while (0)
#define PITA_LAUNCH_CHECK() PITA_HIP_CHECK(hipGetLastError())
#define PITA_REQUIRE(cond, ...)                                   \
  do {                                                            \
    if (!(cond)) return ::pita::fail(PITA_EINVAL, __VA_ARGS__);   \
  } while (0)

constexpr int kWave = 64;  // CDNA wavefront

// Launch state that HIP keeps PER DEVICE -- the dynamic-LDS opt-in of a kernel (hipFuncSetAttribute), occupancy and CU
// counts -- is cached per device: a process that drives several GPUs (handles carry their device, PitaDeviceGuard) must
// not reuse the first device's answers on the second.
constexpr int kMaxDevices = 32;
inline int current_device_slot() {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0) d = 0;
  return d % kMaxDevices;
}
template <class T>
struct PerDevice {
  T v[kMaxDevices] = {};
  T& get() { return v[current_device_slot()]; }
};

// The dynamic-LDS limit of a kernel (hipFuncAttributeMaxDynamicSharedMemorySize) is state of (device, kernel function),
// not of a handle: two handles that share an instantiation but need different sizes (other n_layers / n_particles) must
// not lower each other's limit, and a cache "this handle configured it" goes stale when another handle configures the
// same kernel smaller.  One table per device: kernel -> largest size configured; the attribute is only ever raised.
inline hipError_t ensure_dynamic_lds(const void* kernel, size_t bytes) {
  static std::mutex mu;
  static std::unordered_map<const void*, size_t> configured[kMaxDevices];
  std::lock_guard<std::mutex> lock(mu);
  auto& m = configured[current_device_slot()];
  const auto it = m.find(kernel);
  if (it != m.end() && it->second >= bytes) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e == hipSuccess) m[kernel] = bytes;
  return e;
}

// ---- device math with explicit accuracy choices
// exp2/rcp map to single v_exp_f32 / v_rcp_f32 (about 1 ulp); used where the reference applies
// sigmoid-family activations (relative error ~1e-7, no cancellation).
__device__ __forceinline__ float fast_sigmoid(float v) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * v));
}
__device__ __forceinline__ float fast_silu(float v) { return v * fast_sigmoid(v); }
// tanh with small-argument series: the coordinate head of a fresh EGNN outputs ~1e-4, where
// 1 - 2/(1+e^{2v}) would lose all relative accuracy.
__device__ __forceinline__ float accurate_tanh(float v) {
  // both branches are evaluated and one is selected: the lanes of a wave nearly always need both anyway, and without
  // the divergent branch the callers' loops stay one basic block (same values as the branching form)
  const float a = fabsf(v);
  const float v2 = v * v;
  float p = 62.0f / 2835.0f;
  p = fmaf(p, v2, -17.0f / 315.0f);
  p = fmaf(p, v2, 2.0f / 15.0f);
  p = fmaf(p, v2, -1.0f / 3.0f);
  p = fmaf(p, v2, 1.0f);
  const float e = __builtin_amdgcn_exp2f(2.88539008177792681f * a);  // e^{2a}
  const float t = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + e);
  return a < 0.25f ? v * p : copysignf(t, v);
}

// ---- Philox4x32-10 counter RNG + Box-Muller (the generator used when the caller passes no noise)
struct Philox {
  static constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
  __host__ __device__ static inline void round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    uint64_t p0 = (uint64_t)M0 * c[0], p1 = (uint64_t)M1 * c[2];
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
  }
  __host__ __device__ static inline void gen(uint32_t (&c)[4], uint64_t seed) {
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      round(c, k0, k1);
      k0 += W0;
      k1 += W1;
    }
  }
};

// Four standard normals for (seed, walker, step, particle).  Counter layout:
// c0,c1 = walker id (64 bit), c2 = step (low 32) , c3 = particle | (step high bits << 20).
__device__ __forceinline__ void philox_normal4(uint64_t seed, uint64_t walker, int64_t step, uint32_t particle,
                                               float (&z)[4]) {
  uint32_t c[4] = {(uint32_t)walker, (uint32_t)(walker >> 32), (uint32_t)step,
                   particle ^ ((uint32_t)((uint64_t)step >> 32) << 20)};
  Philox::gen(c, seed);
  // u in (0,1): 24 random bits + half-ulp offset
  const float s = 1.0f / 16777216.0f;
  float u0 = ((c[0] >> 8) + 0.5f) * s, u1 = ((c[1] >> 8) + 0.5f) * s;
  float u2 = ((c[2] >> 8) + 0.5f) * s, u3 = ((c[3] >> 8) + 0.5f) * s;
  // Box-Muller on the transcendental unit: v_log_f32 is log2, v_sin/v_cos take revolutions (no range reduction
  // needed for u in (0,1)); absolute error of a normal ~1e-6, far below the sampler's fp32 noise floor
  const float kM2Ln2 = -1.38629436111989062f;  // -2 ln 2
  float r0 = __builtin_amdgcn_sqrtf(kM2Ln2 * __builtin_amdgcn_logf(u0));
  float r1 = __builtin_amdgcn_sqrtf(kM2Ln2 * __builtin_amdgcn_logf(u2));
  float s0 = __builtin_amdgcn_sinf(u1), c0 = __builtin_amdgcn_cosf(u1);
  float s1 = __builtin_amdgcn_sinf(u3), c1 = __builtin_amdgcn_cosf(u3);
  z[0] = r0 * c0; z[1] = r0 * s0; z[2] = r1 * c1; z[3] = r1 * s1;
}

// one uniform in (0,1) for (seed, walker, step, tag)
__device__ __forceinline__ float philox_uniform(uint64_t seed, uint64_t walker, int64_t step, uint32_t tag) {
  uint32_t c[4] = {(uint32_t)walker, (uint32_t)(walker >> 32), (uint32_t)step,
                   tag ^ ((uint32_t)((uint64_t)step >> 32) << 20)};
  Philox::gen(c, seed);
  return ((c[0] >> 8) + 0.5f) * (1.0f / 16777216.0f);
}

}  // namespace pita
