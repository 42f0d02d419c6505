// common.h — shared device/host helpers for libnunet (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <type_traits>

#include "../../include/nunet.h"
#include "../../include/nunet_diag.h"

typedef __bf16 bf16_t;
typedef _Float16 f16_t;

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

#define WAVE 64
#define CONV_GROUP_MAX 4      // independent 3x3 convolutions one launch may carry (conv3x3.hip ConvGroup, plan.hip Sched::run_wave)

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
  PC_CONV_M256N32 = 0, PC_CONV_M128N64, PC_CONV_M128N32, PC_WGRAD_1x4, PC_WGRAD_2x2, PC_BN_FWD, PC_BN_BWD_REDUCE,
  PC_BN_BWD_APPLY, PC_UP_FWD, PC_UP_BWD, PC_POOL, PC_HEAD, PC_PACK, PC_UNPACK, PC_LOSS, PC_SGD,
  PC_LAYOUT, PC_COUNT
};
extern thread_local bool g_prof_on;
extern thread_local int g_prof_alg_cin;  // >0: algorithmic Cin of the next conv/wgrad (padded first layer)
void nunet_prof_push(int cls, double flops, double bytes);
void nunet_prof_pop();
void nunet_prof_kernel_events(hipEvent_t* e0, hipEvent_t* e1);   // a fresh start/stop pair owned by the innermost open scope (null: none)
struct ProfScope {
  bool on;
  ProfScope(int cls, double flops, double bytes, hipStream_t) : on(g_prof_on) { if (on) nunet_prof_push(cls, flops, bytes); }
  ~ProfScope() { if (on) nunet_prof_pop(); }
};
// Segmented step recording (graph.hip, nunet_seg_*): while a DRY pass runs, launches are skipped - the pass only finds out which
// events are waited on across lanes.
extern thread_local bool g_dry_run;
// Every kernel launch goes through this. Timing off: plain hipLaunchKernelGGL. Timing on (bench.py's roofline leg):
// hipExtLaunchKernelGGL with a start/stop event pair, which carries the DISPATCH's own begin/end timestamps (what
// rocprofv3 --kernel-trace reports) - events recorded around a launch as separate stream markers add ~5 us to a 15 us kernel.
#define NUNET_LAUNCH(kernel, grid, block, shmem, stream, ...)                                                    \
  do {                                                                                                          \
    if (g_dry_run) break;                                                                                       \
    hipEvent_t pe0_ = nullptr, pe1_ = nullptr;                                                                  \
    if (g_prof_on) nunet_prof_kernel_events(&pe0_, &pe1_);                                                      \
    if (pe0_) hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, pe0_, pe1_, 0, __VA_ARGS__);            \
    else hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);                                   \
  } while (0)

// zero-fill by a kernel on the caller's stream. hipMemsetAsync is NOT used anywhere:
// captured into a hipGraph its memset node ran unordered w.r.t. the neighbouring kernel
// nodes on replay (accumulators were cleared late / early), see DESIGN.md.
int nunet_zero_async(void* p, size_t bytes, hipStream_t st);

// ---------------------------------------------------------------------------
// Segmented recording of a multi-lane step (graph.hip). ROCm 7.2 replays a hipGraph with parallel branches by enqueueing
// node after node with a synchronisation of its own (2.6-5 us per node, tools/graph_gap_probe.py), while a single-stream
// graph replays as one batch of pre-built packets (0.7 us per node). So the plan's lanes are NOT captured as branches of one
// graph: every lane keeps a real stream, its ops are captured into single-stream graph SEGMENTS, and the cross-lane
// dependencies become event records / waits between the graph launches. The lane scheduler (plan.hip, Sched) routes its
// stream waits, event records and "about to launch on this stream" through these hooks; they return false when no
// recording is active (the caller then does the real HIP call).
// ---------------------------------------------------------------------------
bool seg_active();
bool seg_wait(hipStream_t st, hipEvent_t ev);      // `st` must wait for `ev` (recorded on another stream)
bool seg_record(hipEvent_t ev, hipStream_t st);    // `ev` marks the current tail of `st`
void seg_touch(hipStream_t st);                    // kernels are about to be launched on `st`

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

// ---------------------------------------------------------------------------
// Order-independent per-channel sums (BatchNorm statistics, BatchNorm-backward sums).
// Hundreds of workgroups add one partial sum each to the same channel. fp32 atomics would make the
// result depend on arrival order (run-to-run noise of ~1e-7 relative that the ill-conditioned small
// cases amplify past the 1e-4 parity bound), so an accumulator is 128-bit FIXED POINT in two int64
// words, value = hi * 2^-20 + lo * 2^-60: integer atomics commute, the total is bit-identical from
// run to run and exact to 2^-60 per partial. |partial| is clamped to 2^30 (non-finite -> 2^30).
// ---------------------------------------------------------------------------
#define NUNET_FX_WORDS 2
__device__ __forceinline__ void fx_add(long long* acc, float v) {
  double s = (v == v) ? (double)v : 1073741824.0;
  s = fmin(fmax(s, -1073741824.0), 1073741824.0) * 1048576.0;   // exact scaling
  const double f = floor(s);
  const long long hi = (long long)f;
  const long long lo = (long long)((s - f) * 1099511627776.0);  // [0, 2^40)
  atomicAdd(reinterpret_cast<unsigned long long*>(acc), (unsigned long long)hi);
  atomicAdd(reinterpret_cast<unsigned long long*>(acc) + 1, (unsigned long long)lo);
}
__device__ __forceinline__ double fx_value(long long hi, long long lo) {
  return (double)hi * (1.0 / 1048576.0) + (double)lo * (1.0 / 1152921504606846976.0);
}
// totals of channel c (both values) over the replicas of a [rep][2][C] accumulator array (integer sums: exact, any
// order). All 2 * NUNET_BN_SUM_REPLICAS 16-byte loads are issued before the first use: ONE memory round trip - this
// runs in the prologue of every consumer workgroup, where a loop of dependent loads would cost ~1 us per replica.
typedef __attribute__((ext_vector_type(2))) long long fx2_t;
__device__ __forceinline__ void fx_totals(const long long* acc, int C, int nrep, int c, double& t0, double& t1) {
  fx2_t q[NUNET_BN_SUM_REPLICAS][2];
#pragma unroll
  for (int r = 0; r < NUNET_BN_SUM_REPLICAS; ++r) {
    const int rr = r < nrep ? r : 0;            // (re-reads replica 0 instead of branching; masked below)
#pragma unroll
    for (int v = 0; v < 2; ++v) q[r][v] = *reinterpret_cast<const fx2_t*>(acc + ((size_t)(rr * 2 + v) * C + c) * NUNET_FX_WORDS);
  }
  long long h0 = 0, l0 = 0, h1 = 0, l1 = 0;
#pragma unroll
  for (int r = 0; r < NUNET_BN_SUM_REPLICAS; ++r) {
    const bool on = r < nrep;
    h0 += on ? q[r][0][0] : 0; l0 += on ? q[r][0][1] : 0;
    h1 += on ? q[r][1][0] : 0; l1 += on ? q[r][1][1] : 0;
  }
  t0 = fx_value(h0, l0); t1 = fx_value(h1, l1);
}

// BatchNorm2d forward coefficients of channel c (nn.BatchNorm2d at reference finished/archs1.py:19,21).
// The 3x3 convs store their output WITHOUT the conv bias b (it is absorbed here): the first layers'
// outputs are bias-dominated (inputs are ~1e-2, reference dataset.py:71), so a 16-bit store of acc+b
// would lose the signal. With y = acc + b:  train: bn(y) = gamma*(acc - mean(acc))*invstd + beta (the
// bias cancels and only shifts running_mean);  eval: bn(y) = gamma*(acc - (rm - b))*invstd + beta.
// `mean` is the value to subtract from the STORED tensor; `mean_full` = E[y]. The batch variance is
// combined in double from the exact fixed-point totals (E[x^2] - E[x]^2 in fp32 cancels when |mean| >> std).
struct BnStatArgs {
  const long long* fx;        // training: [rep][2][C] fixed-point sums of the stored tensor
  const float* conv_bias;     // or NULL
  const float* rm; const float* rv;
  int C, training; float M, eps;
};
__device__ __forceinline__ void bn_stat_coeffs(const BnStatArgs& a, int c, float& mean, float& invstd, float& var, float& mean_full) {
  const float b = a.conv_bias ? a.conv_bias[c] : 0.f;
  if (a.training) {
    const int nrep = bn_sum_replicas(a.C);
    double t1, t2;
    fx_totals(a.fx, a.C, nrep, c, t1, t2);
    const double m = t1 / (double)a.M;
    const double v = t2 / (double)a.M - m * m;
    mean = (float)m;
    var = v > 0.0 ? (float)v : 0.f;
    mean_full = mean + b;
  } else {
    mean_full = a.rm[c];
    mean = mean_full - b;
    var = a.rv[c];
  }
  invstd = 1.0f / sqrtf(var + a.eps);
}

// running_mean / running_var update of nn.BatchNorm2d in training mode (momentum form, unbiased variance): ONE operation order for
// every kernel that performs it (contraction off: whether `a * b + c * d` fuses is otherwise decided per call site, and the
// BatchNorm taken in a conv's loader must leave the same bits as the stand-alone kernel)
__device__ __forceinline__ void bn_running_update(float* rm, float* rv, int c, float momentum, float mean_full, float var, float M) {
#pragma clang fp contract(off)
  const float unb = M > 1.f ? var * (M / (M - 1.f)) : var;
  const float keep = 1.f - momentum;
  const float a = keep * rm[c], b = momentum * mean_full;
  rm[c] = a + b;
  const float cc = keep * rv[c], d = momentum * unb;
  rv[c] = cc + d;
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

// Whole-vector conversion between a 16-byte storage vector and fp32 lanes. With ext_vector types the compiler emits
// one shift / v_cvt per element on the way in, v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32 per PAIR on the way out and packed
// fp32 math (v_pk_fma_f32) in between - the element-wise get()/set() accessors cost ~3x the VALU instructions, which
// matters where a transform sits between a global load and an LDS write on a conv's critical path.
typedef __attribute__((ext_vector_type(8))) float f32x8;
template <typename T> struct FV { typedef f32x8 type; };
template <> struct FV<float> { typedef f32x4 type; };
template <typename T> __device__ __forceinline__ typename FV<T>::type vec_to_f(const Vec16<T>& v) {
  if constexpr (std::is_same<T, float>::value) return __builtin_bit_cast(f32x4, v.raw);
  else if constexpr (std::is_same<T, bf16_t>::value) return __builtin_convertvector(__builtin_bit_cast(bf16x8, v.raw), f32x8);
  else return __builtin_convertvector(__builtin_bit_cast(f16x8, v.raw), f32x8);
}
template <typename T> __device__ __forceinline__ Vec16<T> vec_from_f(const typename FV<T>::type& f) {
  Vec16<T> v;
  if constexpr (std::is_same<T, float>::value) v.raw = __builtin_bit_cast(u32x4, f);
  else if constexpr (std::is_same<T, bf16_t>::value) v.raw = __builtin_bit_cast(u32x4, __builtin_convertvector(f, bf16x8));
  else v.raw = __builtin_bit_cast(u32x4, __builtin_convertvector(f, f16x8));
  return v;
}
// fp32 lanes <- EPV consecutive floats of a table (LDS or global), 16-byte aligned
template <typename T> __device__ __forceinline__ typename FV<T>::type ldf(const float* p) {
  if constexpr (std::is_same<T, float>::value) return *reinterpret_cast<const f32x4*>(p);
  else {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
    return f32x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  }
}
// BatchNorm + ReLU forward / backward-apply on whole vectors; explicit fma so that every kernel that evaluates them
// (stand-alone BN kernels, the convs' input transforms) rounds identically.
//   forward : relu(fma(y, sc, sh))
//   backward: dz = (fma(y, sc, sh) > 0) ? da : 0;  dy = sc * (dz - k1 - xhat * k2) evaluated as fma(sc, dz, -fma(B, y, A))
//             with B = sc * k2 * invstd, A = sc * k1 - B * mean  (bn_bwd_AB)
template <typename V> __device__ __forceinline__ V bn_relu_apply(const V& y, const V& sc, const V& sh) {
  const V z = __builtin_elementwise_fma(y, sc, sh);
  return __builtin_elementwise_max(z, V(0.f));
}
template <typename V> __device__ __forceinline__ V bn_relu_bwd_apply(const V& da, const V& y, const V& sc, const V& sh, const V& A, const V& B) {
  const V act = __builtin_elementwise_fma(y, sc, sh);
  const V dz = act > V(0.f) ? da : V(0.f);
  return __builtin_elementwise_fma(sc, dz, -__builtin_elementwise_fma(B, y, A));
}
__device__ __forceinline__ void bn_bwd_AB(float mean, float istd, float sc, float k1, float k2, float& A, float& B) {
  B = sc * k2 * istd;
  A = __builtin_fmaf(-B, mean, sc * k1);
}

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
