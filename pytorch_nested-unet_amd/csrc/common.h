// common.h — shared device/host helpers for libnunet (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <type_traits>

#include "../../include/nunet.h"

typedef __bf16 bf16_t;
typedef _Float16 f16_t;

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

#define WAVE 64

// ---------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------
void nunet_set_error(const char* fmt, ...);
int nunet_check_launch(const char* what);

#define NUNET_REQUIRE(cond, ...)            \
  do {                                      \
    if (!(cond)) {                          \
      nunet_set_error(__VA_ARGS__);         \
      return NUNET_EINVAL;                  \
    }                                       \
  } while (0)

// ---------------------------------------------------------------------------
// optional per-launch timing (bench.py's roofline leg): hipEvents on the launch
// stream around every kernel, aggregated per kernel class. Off by default.
// ---------------------------------------------------------------------------
enum {
  PC_CONV_M256N32 = 0, PC_CONV_M128N64, PC_WGRAD_1x4, PC_WGRAD_2x2, PC_BN_FWD, PC_BN_BWD_REDUCE,
  PC_BN_BWD_APPLY, PC_UP_FWD, PC_UP_BWD, PC_POOL, PC_HEAD, PC_PACK, PC_UNPACK, PC_LOSS, PC_SGD,
  PC_LAYOUT, PC_COUNT
};
extern thread_local bool g_prof_on;
extern thread_local int g_prof_alg_cin;  // >0: algorithmic Cin of the next conv/wgrad (padded first layer)
void nunet_prof_push(int cls, double flops, double bytes, hipStream_t st);
void nunet_prof_pop(hipStream_t st);
struct ProfScope {
  hipStream_t st; bool on;
  ProfScope(int cls, double flops, double bytes, hipStream_t s) : st(s), on(g_prof_on) { if (on) nunet_prof_push(cls, flops, bytes, s); }
  ~ProfScope() { if (on) nunet_prof_pop(st); }
};

// zero-fill by a kernel on the caller's stream. hipMemsetAsync is NOT used anywhere:
// captured into a hipGraph its memset node ran unordered w.r.t. the neighbouring kernel
// nodes on replay (accumulators were cleared late / early), see DESIGN.md.
int nunet_zero_async(void* p, size_t bytes, hipStream_t st);

static inline int dtype_size(int dt) { return dt == NUNET_F32 ? 4 : 2; }
static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t a, size_t b) { return (a + b - 1) / b * b; }

// ---------------------------------------------------------------------------
// storage-type traits: EPV = elements per 16-byte vector
// ---------------------------------------------------------------------------
template <typename T> struct Tr;
template <> struct Tr<float> {
  static constexpr int EPV = 4;
  static constexpr int DT = NUNET_F32;
};
template <> struct Tr<bf16_t> {
  static constexpr int EPV = 8;
  static constexpr int DT = NUNET_BF16;
};
template <> struct Tr<f16_t> {
  static constexpr int EPV = 8;
  static constexpr int DT = NUNET_F16;
};

template <typename T> __device__ __forceinline__ float to_f32(T v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v) { return (T)v; }

// 16-byte vector of T with element access as float
// Replicas of a per-channel sum buffer actually used for C channels (include/nunet.h, NUNET_BN_SUM_REPLICAS):
// contention is a matter of the shallow, wide levels (many workgroups, few channels); deep levels have
// few workgroups and reading 8 x 2 x C values per consumer block would cost more than it saves.
__host__ __device__ __forceinline__ int bn_sum_replicas(int C) {
  const int r = 256 / C;
  return r < 1 ? 1 : (r > NUNET_BN_SUM_REPLICAS ? NUNET_BN_SUM_REPLICAS : r);
}

// Linear index -> (channel group, x, y, n) of an [N][H][W][G] iteration space. The element-wise kernels used three 64-bit
// divisions per element (>100 instructions each on this ISA): with the extents fixed per launch they become multiplies
// by ceil(2^32 / d), exact while index * d < 2^32 (checked on the host; the 64-bit path stays for larger tensors).
struct Dec4 { int G, W, H; unsigned iG, iW, iH; int fast; };
static inline unsigned dec_inv(int d) { return d > 1 ? (unsigned)(((1ull << 32) + (unsigned)d - 1) / (unsigned)d) : 0u; }
static inline Dec4 make_dec4(long long total, int G, int W, int H) {
  Dec4 d; d.G = G; d.W = W; d.H = H; d.iG = dec_inv(G); d.iW = dec_inv(W); d.iH = dec_inv(H);
  const long long mx = G > W ? (G > H ? G : H) : (W > H ? W : H);
  d.fast = total > 0 && total * mx < (1ll << 32) ? 1 : 0;
  return d;
}
__device__ __forceinline__ int dec_div(int n, unsigned inv) { return inv ? (int)__umulhi((unsigned)n, inv) : n; }
__device__ __forceinline__ void dec4(const Dec4& d, long long i, int& cg, long long& o, int& x, int& y, int& n) {
  if (d.fast) {
    const int ii = (int)i;
    const int oo = dec_div(ii, d.iG); cg = ii - oo * d.G;
    const int t = dec_div(oo, d.iW); x = oo - t * d.W;
    n = dec_div(t, d.iH); y = t - n * d.H;
    o = oo;
  } else {
    cg = (int)(i % d.G); o = i / d.G;
    x = (int)(o % d.W);
    const long long t = o / d.W;
    y = (int)(t % d.H); n = (int)(t / d.H);
  }
}

template <typename T> struct Vec16 {
  // zero-initialised: set() of a 16-bit element read-modify-writes its 32-bit word, and doing that on an
  // indeterminate word is undefined (it miscompiled for fp16 on the odd elements of words 0 and 1)
  u32x4 raw = {0u, 0u, 0u, 0u};
  __device__ __forceinline__ float get(int i) const {
    if constexpr (std::is_same<T, float>::value) {
      const uint32_t w = raw[i];  // copy first: bit_cast of a vector-element lvalue reads element 0
      return __builtin_bit_cast(float, w);
    } else {
      uint32_t w = raw[i >> 1];
      uint16_t hv = (i & 1) ? (uint16_t)(w >> 16) : (uint16_t)(w & 0xffff);
      return (float)__builtin_bit_cast(T, hv);
    }
  }
  __device__ __forceinline__ void set(int i, float v) {
    if constexpr (std::is_same<T, float>::value) {
      const uint32_t w = __builtin_bit_cast(uint32_t, v);
      raw[i] = w;
    } else {
      uint16_t hv = __builtin_bit_cast(uint16_t, (T)v);
      uint32_t w = raw[i >> 1];
      w = (i & 1) ? ((w & 0x0000ffffu) | ((uint32_t)hv << 16)) : ((w & 0xffff0000u) | hv);
      raw[i >> 1] = w;
    }
  }
};

template <typename T> __device__ __forceinline__ Vec16<T> ld16(const T* p) {
  Vec16<T> v;
  v.raw = *reinterpret_cast<const u32x4*>(p);
  return v;
}
template <typename T> __device__ __forceinline__ void st16(T* p, const Vec16<T>& v) {
  *reinterpret_cast<u32x4*>(p) = v.raw;
}
template <typename T> __device__ __forceinline__ Vec16<T> zero16() {
  Vec16<T> v;
  v.raw = u32x4{0u, 0u, 0u, 0u};
  return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// dispatch a templated launcher on dtype
#define NUNET_DISPATCH(dt, FN, ...)                                   \
  ((dt) == NUNET_F32    ? FN<float>(__VA_ARGS__)                      \
   : (dt) == NUNET_BF16 ? FN<bf16_t>(__VA_ARGS__)                     \
   : (dt) == NUNET_F16  ? FN<f16_t>(__VA_ARGS__)                      \
                        : (nunet_set_error("bad dtype %d", (int)(dt)), NUNET_EINVAL))
