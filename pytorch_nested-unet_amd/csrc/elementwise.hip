// elementwise.hip — the HBM-bound kernels of the UNet++ path (NHWC, 16 B/lane):
//   BatchNorm2d(+ReLU)(+MaxPool2d) forward, BatchNorm/ReLU backward,
//   MaxPool2d(2,2) fwd/bwd, bilinear x2 align_corners upsample fwd/bwd,
//   1x1 heads fwd/bwd, BCEDiceLoss fwd/bwd, IoU counts, SGD, layout helpers.
// Reference arithmetic: finished/archs1.py:17-21,82-83,105-111; losses.py:103-117;
// metrics.py:6-18; trains.py:229-231.
#include <stdlib.h>
#include <string.h>

#include "common.h"

static thread_local char g_err[512] = "";
void nunet_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
int nunet_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    nunet_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return NUNET_ELAUNCH;
  }
  // NUNET_TRACE_LAUNCH=1 (diagnostic, eager mode only): name every launch on stderr and wait for it, so that a
  // faulting kernel is the last name printed
  static int trace = -1;
  if (trace < 0) { const char* t = getenv("NUNET_TRACE_LAUNCH"); trace = t ? atoi(t) : 0; }
  if (trace) {
    fprintf(stderr, "[nunet] %s ... ", what); fflush(stderr);
    e = hipDeviceSynchronize();
    fprintf(stderr, "%s\n", e == hipSuccess ? "ok" : hipGetErrorString(e)); fflush(stderr);
    if (e != hipSuccess) { nunet_set_error("%s: execution failed: %s", what, hipGetErrorString(e)); return NUNET_ELAUNCH; }
  }
  return NUNET_OK;
}
extern "C" const char* nunet_last_error(void) { return g_err; }
extern "C" int nunet_version(void) { return 100; }

static inline int grid_for(int64_t items, int block, int cap = 256 * 16) {
  int64_t g = ceil_div64(items, block);
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

__global__ __launch_bounds__(256) void zero_kernel(u32x4* __restrict__ p16, size_t n16, uint32_t* __restrict__ tail, int ntail) {
  const u32x4 z = {0u, 0u, 0u, 0u};
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p16[i] = z;
  if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = 0u;
}
int nunet_zero_async(void* p, size_t bytes, hipStream_t st) {
  NUNET_REQUIRE(((uintptr_t)p % 16) == 0 && bytes % 4 == 0, "zero: buffer must be 16-byte aligned, size a multiple of 4");
  const size_t n16 = bytes / 16;
  const int ntail = (int)((bytes % 16) / 4);
  NUNET_LAUNCH(zero_kernel, dim3(grid_for((int64_t)n16, 256 * 4, 2048)), dim3(256), 0, st, (u32x4*)p, n16,
                     (uint32_t*)((char*)p + n16 * 16), ntail);
  return nunet_check_launch("zero");
}

// ---------------------------------------------------------------------------
// layout: NCHW fp32 -> NHWC T with zero channel padding
// ---------------------------------------------------------------------------
// One thread per (pixel, 16-byte channel group): consecutive threads write consecutive 16 bytes; blockIdx.y is the image,
// so there is no division in the kernel (the first version spent two 64-bit divisions per 2-byte element).
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ x, int C, int hw, T* __restrict__ y, int cpad, int gshift) {
  constexpr int EPV = Tr<T>::EPV;
  const int n = blockIdx.y;
  const int groups = 1 << gshift;
  const float* xi = x + (size_t)n * C * hw;
  T* yi = y + (size_t)n * hw * cpad;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < (hw << gshift); j += gridDim.x * blockDim.x) {
    const int pix = j >> gshift, c0 = (j & (groups - 1)) * EPV;
    Vec16<T> v;
#pragma unroll
    for (int e = 0; e < EPV; ++e) v.set(e, c0 + e < C ? xi[(size_t)(c0 + e) * hw + pix] : 0.f);
    st16(yi + (size_t)pix * cpad + c0, v);
  }
}
template <typename T> static int launch_nchw_to_nhwc(const float* x, int N, int C, int H, int W, void* y, int cpad, hipStream_t st) {
  constexpr int EPV = Tr<T>::EPV;
  const int groups = cpad / EPV;
  int gshift = 0;
  while ((1 << gshift) < groups) ++gshift;
  NUNET_REQUIRE(cpad % EPV == 0 && (1 << gshift) == groups && N <= 65535, "nchw_to_nhwc: cpad=%d must be a power-of-two number of 16-byte groups", cpad);
  const long long hw = (long long)H * W;
  NUNET_REQUIRE(hw * groups < (1ll << 31), "nchw_to_nhwc: image too large");
  const int64_t total = (int64_t)N * H * W * cpad;
  ProfScope ps(PC_LAYOUT, 0, (double)total * sizeof(T) + (double)N * C * H * W * 4, st);
  NUNET_LAUNCH((nchw_to_nhwc_kernel<T>), dim3(grid_for(hw * groups, 256, 1024), N), dim3(256), 0, st, x, C, (int)hw, (T*)y, cpad, gshift);
  return nunet_check_launch("nchw_to_nhwc");
}
extern "C" int nunet_nchw_to_nhwc(const float* x, int32_t N, int32_t C, int32_t H, int32_t W, int32_t dtype, void* y, int32_t cpad, nunet_stream_t s) {
  NUNET_REQUIRE(x && y && N > 0 && C > 0 && H > 0 && W > 0 && cpad >= C, "nchw_to_nhwc: bad args");
  return NUNET_DISPATCH(dtype, launch_nchw_to_nhwc, x, N, C, H, W, y, cpad, (hipStream_t)s);
}

// ---------------------------------------------------------------------------
// Device-side input pipeline: what reference dataset.py:66-74 does on the host
// after decoding — albumentations Normalize() (trains.py:266), the extra /255 (dataset.py:71),
// HWC -> CHW (dataset.py:72) — plus the geometric augmentations of trains.py:258-259
// (RandomRotate90 / Flip) applied per sample from host-drawn codes. Input stays uint8 over PCIe.
//   aug[n] = rot90 count (bits 0-1, counter-clockwise like np.rot90) | hflip (bit 2) | vflip (bit 3)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void preprocess_u8_kernel(const uint8_t* __restrict__ u, int N, int H, int W, int C,
                                                             const float* __restrict__ mean, const float* __restrict__ stdv,
                                                             const int32_t* __restrict__ aug, float post_scale, float* __restrict__ out) {
  const int64_t total = (int64_t)N * C * H * W;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % W);
    int64_t t = i / W;
    const int y = (int)(t % H); t /= H;
    const int c = (int)(t % C);
    const int n = (int)(t / C);
    // destination (y, x) <- source (sy, sx): undo flips, then the rotation
    int sy = y, sx = x;
    const int a = aug ? aug[n] : 0;
    if (a & 8) sy = H - 1 - sy;
    if (a & 4) sx = W - 1 - sx;
    const int k = a & 3;                       // out = rot90(src, k)  (H == W when k is odd)
    int ry = sy, rx = sx;
    if (k == 1) { ry = sx; rx = W - 1 - sy; }
    else if (k == 2) { ry = H - 1 - sy; rx = W - 1 - sx; }
    else if (k == 3) { ry = H - 1 - sx; rx = sy; }
    const float v = (float)u[(((int64_t)n * H + ry) * W + rx) * C + c];
    const float m = mean ? mean[c] : 0.f, sd = stdv ? stdv[c] : 1.f;
    out[i] = ((v / 255.f - m) / sd) * post_scale;
  }
}
extern "C" int nunet_preprocess_u8(const uint8_t* u8_nhwc, int32_t N, int32_t H, int32_t W, int32_t C, const float* mean, const float* stdv,
                                   const int32_t* aug, float post_scale, float* out_nchw, nunet_stream_t s) {
  NUNET_REQUIRE(u8_nhwc && out_nchw && N > 0 && H > 0 && W > 0 && C > 0, "preprocess_u8: bad args");
  const int64_t total = (int64_t)N * C * H * W;
  hipStream_t st = (hipStream_t)s;
  ProfScope ps(PC_LAYOUT, 0, (double)total * 5, st);
  NUNET_LAUNCH(preprocess_u8_kernel, dim3(grid_for(total, 256 * 4, 2048)), dim3(256), 0, st, u8_nhwc, N, H, W, C, mean, stdv, aug, post_scale, out_nchw);
  return nunet_check_launch("preprocess_u8");
}

// ---------------------------------------------------------------------------
// BatchNorm (+ReLU) (+2x2 max-pool) forward
// ---------------------------------------------------------------------------
__device__ __forceinline__ void up_taps(int o, float scale, int n_in, int& i0, int& i1, float& l1);
__device__ __forceinline__ float up_lerp(float v00, float v01, float v10, float v11, float ly, float lx);
struct BnFwdP {
  void* up; int PU; int nb_main; Dec4 dcu;     // fused x2 bilinear upsample of the activation (second block role), or up == NULL
  const void* y; int PY;
  const float* conv_bias; const long long* stats; const float* gamma; const float* beta;
  float* rm; float* rv; int64_t* nbt; float* save;
  int training; float momentum, eps;
  void* a; int PA; void* pooled; int PP;
  int N, H, W, C;
};

// (the conv bias is absorbed here, see bn_stat_coeffs in common.h)
__device__ __forceinline__ void bn_channel_coeffs(const BnFwdP& p, int c, float M, float& mean, float& invstd, float& var, float& mean_full) {
  BnStatArgs a; a.fx = p.stats; a.conv_bias = p.conv_bias; a.rm = p.rm; a.rv = p.rv; a.C = p.C; a.training = p.training; a.M = M; a.eps = p.eps;
  bn_stat_coeffs(a, c, mean, invstd, var, mean_full);
}

template <typename T, bool POOL>
__global__ __launch_bounds__(256) void bn_relu_fwd_kernel(BnFwdP p) {
  constexpr int EPV = Tr<T>::EPV;
  __shared__ __attribute__((aligned(16))) float s_sc[2048], s_sh[2048];
  const int G = p.C / EPV;  // channel groups; 256 % G == 0
  const int cg = threadIdx.x % G;
  const float M = (float)p.N * p.H * p.W;
  // first batch of loads ahead of the coefficient prologue (latencies overlap)
  constexpr int U = 4;   // 16-byte loads kept in flight per thread
  const int ppb0 = blockDim.x / G, pl0 = threadIdx.x / G;
  const int64_t npix0 = (int64_t)p.N * p.H * p.W;
  const int nbm = p.up ? p.nb_main : (int)gridDim.x;     // blocks of the BatchNorm role (the rest, if any, upsample)
  const bool main_role = (int)blockIdx.x < nbm;
  const int64_t stride0 = (int64_t)nbm * ppb0;
  Vec16<T> v[U];
  auto load_batch = [&](int64_t p0) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t pix = p0 + u * stride0;
      if (pix < npix0) v[u] = ld16((const T*)p.y + pix * p.PY + cg * EPV);
    }
  };
  int64_t pix0 = (int64_t)blockIdx.x * ppb0 + pl0;
  if constexpr (!POOL) { if (main_role && pix0 < npix0) load_batch(pix0); }
  const int H2 = p.H / 2, W2 = p.W / 2;
  const int64_t nq = (int64_t)p.N * H2 * W2;
  Vec16<T> vq[4];
  const bool q32 = nq < (1ll << 31);   // 32-bit index decode (a 64-bit division is >100 instructions here)
  auto quad_base = [&](int64_t q) {
    if (q32) {
      const unsigned t = (unsigned)q / (unsigned)W2, qx = (unsigned)q - t * (unsigned)W2;
      const unsigned n = t / (unsigned)H2, qy = t - n * (unsigned)H2;
      return ((int64_t)n * p.H + 2 * qy) * p.W + 2 * qx;
    }
    const int qx = (int)(q % W2);
    const int64_t t = q / W2;
    const int qy = (int)(t % H2);
    const int n = (int)(t / H2);
    return ((int64_t)n * p.H + 2 * qy) * p.W + 2 * qx;
  };
  int64_t p00_cur = 0;                  // base pixel of the quad whose loads are in flight
  auto qload = [&](int64_t q) {
    const int64_t p00 = p00_cur = quad_base(q);
#pragma unroll
    for (int k = 0; k < 4; ++k) vq[k] = ld16((const T*)p.y + (p00 + (k >> 1) * p.W + (k & 1)) * p.PY + cg * EPV);
  };
  const int64_t q0 = (int64_t)blockIdx.x * ppb0 + pl0;
  if constexpr (POOL) { if (main_role && q0 < nq) qload(q0); }
  // per-channel coefficients once per block (not per thread), block 0 also owns the
  // running-stat update and the saved mean/invstd for backward
  for (int c = threadIdx.x; c < p.C; c += blockDim.x) {
    float mean, invstd, var, mf;
    bn_channel_coeffs(p, c, M, mean, invstd, var, mf);
    const float sc = p.gamma[c] * invstd;
    s_sc[c] = sc;
    s_sh[c] = __builtin_fmaf(-mean, sc, p.beta[c]);
    if (blockIdx.x == 0 && p.training) {
      p.save[c] = mean;
      p.save[p.C + c] = invstd;
      if (p.rm) bn_running_update(p.rm, p.rv, c, p.momentum, mf, var, M);
    }
  }
  if (blockIdx.x == 0 && p.training && threadIdx.x == 0 && p.nbt) *p.nbt += 1;
  __syncthreads();
  typedef typename FV<T>::type V;
  if (p.up && (int)blockIdx.x >= p.nb_main) {
    // ---- second role: nn.Upsample(x2, bilinear, align_corners) of the activation (archs1.py:83,116-131), straight from the raw
    // tensor: every tap is relu(bn(y)) ROUNDED to the storage type - the value the first role stores - so the result equals the
    // stand-alone upsample of the stored activation bit for bit, without waiting for it (one launch less per block on the chain)
    const int HO = 2 * p.H, WO = 2 * p.W;
    const float sy = HO > 1 ? (float)(p.H - 1) / (float)(HO - 1) : 0.f;
    const float sx = WO > 1 ? (float)(p.W - 1) / (float)(WO - 1) : 0.f;
    const int64_t total = (int64_t)p.N * HO * WO * G;
    const int64_t nthr = (int64_t)(gridDim.x - p.nb_main) * blockDim.x;
    for (int64_t i = ((int64_t)blockIdx.x - p.nb_main) * blockDim.x + threadIdx.x; i < total; i += nthr) {
      int cgu, ox, oy, n; long long o;
      dec4(p.dcu, i, cgu, o, ox, oy, n);
      int y0, y1, x0, x1; float ly, lx;
      up_taps(oy, sy, p.H, y0, y1, ly);
      up_taps(ox, sx, p.W, x0, x1, lx);
      const T* b = (const T*)p.y + (int64_t)n * p.H * p.W * p.PY + cgu * EPV;
      const Vec16<T> r00 = ld16(b + ((int64_t)y0 * p.W + x0) * p.PY), r01 = ld16(b + ((int64_t)y0 * p.W + x1) * p.PY);
      const Vec16<T> r10 = ld16(b + ((int64_t)y1 * p.W + x0) * p.PY), r11 = ld16(b + ((int64_t)y1 * p.W + x1) * p.PY);
      const V scu = ldf<T>(&s_sc[cgu * EPV]), shu = ldf<T>(&s_sh[cgu * EPV]);
      const Vec16<T> v00 = vec_from_f<T>(bn_relu_apply<V>(vec_to_f<T>(r00), scu, shu)), v01 = vec_from_f<T>(bn_relu_apply<V>(vec_to_f<T>(r01), scu, shu));
      const Vec16<T> v10 = vec_from_f<T>(bn_relu_apply<V>(vec_to_f<T>(r10), scu, shu)), v11 = vec_from_f<T>(bn_relu_apply<V>(vec_to_f<T>(r11), scu, shu));
      Vec16<T> r;
#pragma unroll
      for (int e = 0; e < EPV; ++e) r.set(e, up_lerp(v00.get(e), v01.get(e), v10.get(e), v11.get(e), ly, lx));
      st16((T*)p.up + o * p.PU + cgu * EPV, r);
    }
    return;
  }
  const V sc = ldf<T>(&s_sc[cg * EPV]), sh = ldf<T>(&s_sh[cg * EPV]);
  const int ppb = blockDim.x / G;  // pixels (or quads) per block iteration
  const int pl = threadIdx.x / G;
  if constexpr (!POOL) {
    const int64_t npix = npix0, stride = stride0;
    while (pix0 < npix) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t pix = pix0 + u * stride;
        if (pix < npix) st16((T*)p.a + pix * p.PA + cg * EPV, vec_from_f<T>(bn_relu_apply<V>(vec_to_f<T>(v[u]), sc, sh)));
      }
      pix0 += U * stride;
      if (pix0 < npix) load_batch(pix0);
    }
  } else {
    // (the quad loads of the first iteration were issued ahead of the prologue, see qload)
    int64_t q = q0;
    while (q < nq) {
      const int64_t p00 = p00_cur;
      // the running maximum is taken over the ROUNDED activations (what a later pool would see): converting the stored
      // vector back is exact, and max is order-independent
      V mx;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int64_t pix = p00 + (k >> 1) * p.W + (k & 1);
        const Vec16<T> o = vec_from_f<T>(bn_relu_apply<V>(vec_to_f<T>(vq[k]), sc, sh));
        const V ar = vec_to_f<T>(o);
        mx = (k == 0) ? ar : __builtin_elementwise_max(mx, ar);
        st16((T*)p.a + pix * p.PA + cg * EPV, o);
      }
      st16((T*)p.pooled + q * p.PP + cg * EPV, vec_from_f<T>(mx));
      q += (int64_t)nbm * ppb;
      if (q < nq) qload(q);
    }
  }
}

template <typename T> static int launch_bn_fwd(const nunet_bn_fwd_desc* d, hipStream_t st) {
  BnFwdP p;
  p.y = d->y; p.PY = d->PY; p.conv_bias = d->conv_bias; p.stats = (const long long*)d->stats; p.gamma = d->gamma; p.beta = d->beta;
  p.rm = d->running_mean; p.rv = d->running_var; p.nbt = d->num_batches_tracked; p.save = d->save_mean_invstd;
  p.training = d->training; p.momentum = d->momentum; p.eps = d->eps;
  p.a = d->a; p.PA = d->PA; p.pooled = d->pooled; p.PP = d->PP;
  p.N = d->N; p.H = d->H; p.W = d->W; p.C = d->C;
  const int G = d->C / Tr<T>::EPV;
  const int ppb = 256 / G;
  // optional second role of the launch: the x2 upsample of the activation (extra blocks behind the BatchNorm role's)
  p.up = d->up; p.PU = d->PU;
  const int64_t utotal = (int64_t)d->N * 4 * d->H * d->W * G;
  // (persistent blocks, ~8 outputs per thread: every block pays the coefficient prologue - two dependent global-memory latencies -
  //  so one block per 256 outputs, the stand-alone upsample's grid, made the fused launch twice as long as the two it replaces)
  const int nb_up = d->up ? grid_for(utotal, 256 * 8, 1024) : 0;
  p.dcu = make_dec4(utotal, G, 2 * d->W, 2 * d->H);
  ProfScope ps(PC_BN_FWD, 0, (double)d->N * d->H * d->W * d->C * sizeof(T) * ((d->pooled ? 2.25 : 2.0) + (d->up ? 5.0 : 0.0)), st);
  if (d->pooled) {
    const int64_t nq = (int64_t)d->N * (d->H / 2) * (d->W / 2);
    p.nb_main = grid_for(nq, ppb * 2, 2048);
    NUNET_LAUNCH((bn_relu_fwd_kernel<T, true>), dim3(p.nb_main + nb_up), dim3(256), 0, st, p);
  } else {
    const int64_t np = (int64_t)d->N * d->H * d->W;
    p.nb_main = grid_for(np, ppb * 4, 2048);
    NUNET_LAUNCH((bn_relu_fwd_kernel<T, false>), dim3(p.nb_main + nb_up), dim3(256), 0, st, p);
  }
  return nunet_check_launch("bn_relu_fwd");
}
static bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
extern "C" int nunet_bn_relu_fwd(const nunet_bn_fwd_desc* d, nunet_stream_t s) {
  NUNET_REQUIRE(d && d->y && d->a && d->gamma && d->beta, "bn_relu_fwd: null pointer");
  const int epv = 16 / dtype_size(d->dtype);
  NUNET_REQUIRE(pow2(d->C) && d->C >= epv && d->C / epv <= 256, "bn_relu_fwd: C=%d must be a power of two in [%d, %d]", d->C, epv, 256 * epv);
  NUNET_REQUIRE(d->PY % epv == 0 && d->PA % epv == 0 && (!d->pooled || d->PP % epv == 0), "bn_relu_fwd: pitch alignment");
  NUNET_REQUIRE(!d->pooled || (d->H % 2 == 0 && d->W % 2 == 0), "bn_relu_fwd: fused pool needs even H, W");
  NUNET_REQUIRE(!d->up || d->PU % epv == 0, "bn_relu_fwd: fused upsample pitch alignment");
  if (d->training) NUNET_REQUIRE(d->stats && d->save_mean_invstd, "bn_relu_fwd: training needs stats and save buffers");
  else NUNET_REQUIRE(d->running_mean && d->running_var, "bn_relu_fwd: eval needs running stats");
  return NUNET_DISPATCH(d->dtype, launch_bn_fwd, d, (hipStream_t)s);
}

// ---------------------------------------------------------------------------
// BatchNorm (+ReLU) backward: reduce pass and apply pass
// ---------------------------------------------------------------------------
struct BnBwdP {
  const void* da; int PDA; const void* y; int PY;
  const float* mi; const float* gamma; const float* beta;
  long long* sums; float* dgamma; float* dbeta; float* dbias;
  void* dy; int PDY;
  int N, H, W, C;
};

template <typename T, bool APPLY>
__global__ __launch_bounds__(256) void bn_relu_bwd_kernel(BnBwdP p) {
  constexpr int EPV = Tr<T>::EPV;
  constexpr int NV = 2;                             // partial sums per channel (reduce pass)
  __shared__ __attribute__((aligned(16))) float s_co[6 * 2048 / 4];   // coefficient tables, C <= 512: [6][C]
  __shared__ float s_part[APPLY ? 1 : 256 * NV * EPV];   // per-thread partials for the block reduction
  const int G = p.C / EPV;
  const int cg = threadIdx.x % G;
  const int ppb = blockDim.x / G, pl = threadIdx.x / G;
  const float M = (float)p.N * p.H * p.W;
  const int C = p.C;
  // the first batch of activation loads is issued BEFORE the per-channel coefficient prologue, so the two
  // global-memory latencies overlap (the small pyramid levels run exactly one iteration per thread)
  const int64_t npix = (int64_t)p.N * p.H * p.W;
  const int64_t stride = (int64_t)gridDim.x * ppb;
  constexpr int U = 4;   // pixels per thread per iteration: 8 sixteen-byte loads in flight
  Vec16<T> vy[U], vd[U];
  auto load_batch = [&](int64_t p0) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t pix = p0 + u * stride;
      if (pix < npix) {
        vy[u] = ld16((const T*)p.y + pix * p.PY + cg * EPV);
        vd[u] = ld16((const T*)p.da + pix * p.PDA + cg * EPV);
      }
    }
  };
  int64_t pix0 = (int64_t)blockIdx.x * ppb + pl;
  if (pix0 < npix) load_batch(pix0);
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const float mean = p.mi[c], istd = p.mi[C + c];
    const float sc = p.gamma[c] * istd;
    s_co[c] = mean; s_co[C + c] = istd; s_co[2 * C + c] = sc; s_co[3 * C + c] = __builtin_fmaf(-mean, sc, p.beta[c]);
    if constexpr (APPLY) {
      const int nrep = bn_sum_replicas(C);
      double t1, t2;
      fx_totals(p.sums, C, nrep, c, t1, t2);
      float A, B;
      bn_bwd_AB(mean, istd, sc, (float)(t1 / (double)M), (float)(t2 / (double)M), A, B);
      s_co[4 * C + c] = A; s_co[5 * C + c] = B;
      if (blockIdx.x == 0) {
        // d beta = sum dz, d gamma = sum dz * xhat; the conv bias in front of the BatchNorm has gradient sum(dy) == 0
        if (p.dbeta) p.dbeta[c] = (float)t1;
        if (p.dgamma) p.dgamma[c] = (float)t2;
        if (p.dbias) p.dbias[c] = 0.f;
      }
    }
  }
  __syncthreads();
  __attribute__((aligned(16))) float sc[EPV], sh[EPV], mean[EPV], istd[EPV], k1[EPV], k2[EPV];   // APPLY: k1, k2 hold A, B (bn_bwd_AB)
#pragma unroll
  for (int e = 0; e < EPV; ++e) {
    const int c = cg * EPV + e;
    mean[e] = s_co[c]; istd[e] = s_co[C + c]; sc[e] = s_co[2 * C + c]; sh[e] = s_co[3 * C + c];
    if constexpr (APPLY) { k1[e] = s_co[4 * C + c]; k2[e] = s_co[5 * C + c]; }
  }
  float a1[APPLY ? 1 : EPV], a2[APPLY ? 1 : EPV];
  if constexpr (!APPLY) {
#pragma unroll
    for (int e = 0; e < EPV; ++e) { a1[e] = 0.f; a2[e] = 0.f; }
  }
  while (pix0 < npix) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t pix = pix0 + u * stride;
      if (pix < npix) {
        if constexpr (APPLY) {
          typedef typename FV<T>::type V;
          st16((T*)p.dy + pix * p.PDY + cg * EPV,
               vec_from_f<T>(bn_relu_bwd_apply<V>(vec_to_f<T>(vd[u]), vec_to_f<T>(vy[u]), ldf<T>(sc), ldf<T>(sh), ldf<T>(k1), ldf<T>(k2))));
        } else {
#pragma unroll
          for (int e = 0; e < EPV; ++e) {
            const float yv = vy[u].get(e);
            const float act = __builtin_fmaf(yv, sc[e], sh[e]);
            const float dz = act > 0.f ? vd[u].get(e) : 0.f;
            const float xh = (yv - mean[e]) * istd[e];
            a1[e] += dz; a2[e] += dz * xh;
          }
        }
      }
    }
    pix0 += U * stride;
    if (pix0 < npix) load_batch(pix0);
  }
  if constexpr (!APPLY) {
    // block reduction over the ppb threads that share a channel group (fixed order), then one fixed-point add per channel
#pragma unroll
    for (int e = 0; e < EPV; ++e) {
      s_part[(threadIdx.x * NV + 0) * EPV + e] = a1[e];
      s_part[(threadIdx.x * NV + 1) * EPV + e] = a2[e];
    }
    __syncthreads();
    for (int t = threadIdx.x; t < NV * C; t += blockDim.x) {
      const int v = t / C, c = t - v * C;
      const int g = c / EPV, e = c - g * EPV;
      float sum = 0.f;
      for (int q = 0; q < ppb; ++q) sum += s_part[((q * G + g) * NV + v) * EPV + e];
      fx_add(p.sums + ((size_t)((blockIdx.x & (bn_sum_replicas(C) - 1)) * 2 + v) * C + c) * NUNET_FX_WORDS, sum);
    }
  }
}

// fat blocks: every block of the reduce pass ends with 2C fixed-point adds (8 pixels per thread, at most 512 / 1024 blocks)
static int bn_bwd_ipb() { return 8; }
static int bn_bwd_cap(bool apply) { return apply ? 1024 : 512; }
template <typename T, bool APPLY> static int launch_bn_bwd_t(const nunet_bn_bwd_desc* d, hipStream_t st) {
  BnBwdP p;
  p.da = d->da; p.PDA = d->PDA; p.y = d->y; p.PY = d->PY; p.mi = d->mean_invstd; p.gamma = d->gamma; p.beta = d->beta;
  p.sums = (long long*)d->sums; p.dgamma = d->dgamma; p.dbeta = d->dbeta; p.dbias = d->dbias; p.dy = d->dy; p.PDY = d->PDY;
  p.N = d->N; p.H = d->H; p.W = d->W; p.C = d->C;
  const int G = d->C / Tr<T>::EPV;
  const int64_t np = (int64_t)d->N * d->H * d->W;
  // fewer, fatter blocks: each block ends with 2C global atomics
  ProfScope ps(APPLY ? PC_BN_BWD_APPLY : PC_BN_BWD_REDUCE, 0, (double)np * d->C * sizeof(T) * (APPLY ? 3.0 : 2.0), st);
  // fat blocks: every block ends with C (2C) same-address global atomics
  NUNET_LAUNCH((bn_relu_bwd_kernel<T, APPLY>), dim3(grid_for(np, (256 / G) * bn_bwd_ipb(), bn_bwd_cap(APPLY))), dim3(256), 0, st, p);
  return nunet_check_launch(APPLY ? "bn_relu_bwd_apply" : "bn_relu_bwd_reduce");
}
template <typename T> static int launch_bn_bwd_reduce(const nunet_bn_bwd_desc* d, hipStream_t st) { return launch_bn_bwd_t<T, false>(d, st); }
template <typename T> static int launch_bn_bwd_apply(const nunet_bn_bwd_desc* d, hipStream_t st) { return launch_bn_bwd_t<T, true>(d, st); }

static int bn_bwd_check(const nunet_bn_bwd_desc* d, bool apply) {
  NUNET_REQUIRE(d && d->da && d->y && d->mean_invstd && d->gamma && d->beta && d->sums, "bn_relu_bwd: null pointer");
  const int epv = 16 / dtype_size(d->dtype);
  NUNET_REQUIRE(pow2(d->C) && d->C >= epv && d->C / epv <= 256 && d->C <= 512, "bn_relu_bwd: C=%d unsupported (power of two <= 512)", d->C);
  NUNET_REQUIRE(d->PY % epv == 0 && d->PDA % epv == 0, "bn_relu_bwd: pitch alignment");
  if (apply) NUNET_REQUIRE(d->dy && d->PDY % epv == 0, "bn_relu_bwd_apply: dy");
  return NUNET_OK;
}
extern "C" int nunet_bn_relu_bwd_reduce(const nunet_bn_bwd_desc* d, nunet_stream_t s) {
  int rc = bn_bwd_check(d, false);
  if (rc) return rc;
  return NUNET_DISPATCH(d->dtype, launch_bn_bwd_reduce, d, (hipStream_t)s);
}
extern "C" int nunet_bn_relu_bwd_apply(const nunet_bn_bwd_desc* d, nunet_stream_t s) {
  int rc = bn_bwd_check(d, true);
  if (rc) return rc;
  return NUNET_DISPATCH(d->dtype, launch_bn_bwd_apply, d, (hipStream_t)s);
}

// ---------------------------------------------------------------------------
// BatchNorm+ReLU backward REDUCE taken by the kernel that COMPLETES a gradient tensor, on the values it has just stored -
// instead of a separate pass (nunet_bn_relu_bwd_reduce) that would read the gradient and the raw conv output again.
// Used by the head backward (the last writer of x0_4's gradient, first kernel of the backward chain). The same fusion in
// the upsample- and pool-backward kernels was measured and dropped: those run thousands of small latency-bound blocks,
// and 2C fixed-point adds per block cost more (fused 21.9 us vs 9.4 + 8.6 us for the plain kernel + the fat-block reduce;
// round 3, again with at most 256 blocks and the list-scheduled executor: fused 17.6-25.9 us against 6.5-15.3 + 9.0-12.4).
// A thread's channel group is fixed (the launch geometry keeps grid stride % G == 0): coefficients and the two partial
// sums of its EPV channels live in registers; the block reduces them in a fixed order, then one fixed-point add each.
// ---------------------------------------------------------------------------
struct BnrP { const void* y; int py; const float* mi; const float* gamma; const float* beta; long long* sums; int C; };
template <typename T> struct BnrAcc {
  static constexpr int EPV = Tr<T>::EPV;
  typedef typename FV<T>::type V;
  V r1, r2;                                   // the only per-thread state: partial sums of this thread's EPV channels
  // LDS: [sc | sh | istd | mean * istd][C] coefficient table (a thread visits one or two elements: per-thread coefficient
  // registers would cost 32 VGPRs and halve the occupancy of a gather kernel that lives on it), then the reduction scratch
  static constexpr int LDS_FLOATS(int) { return 0; }
  __device__ __forceinline__ void init(const BnrP& b, float* s_tab) {
    for (int c = threadIdx.x; c < b.C; c += blockDim.x) {
      const float mean = b.mi[c], istd = b.mi[b.C + c];
      const float sc = b.gamma[c] * istd;
      s_tab[c] = sc; s_tab[b.C + c] = __builtin_fmaf(-mean, sc, b.beta[c]); s_tab[2 * b.C + c] = istd; s_tab[3 * b.C + c] = mean * istd;
    }
    r1 = V(0.f); r2 = V(0.f);
    __syncthreads();
  }
  __device__ __forceinline__ Vec16<T> fetch(const BnrP& b, int cg, long long pix) const { return ld16((const T*)b.y + pix * b.py + cg * EPV); }
  // `stored`: the gradient vector as written (rounded to T); yv: the raw conv output at the same pixel (fetch(), issued early)
  __device__ __forceinline__ void add(const BnrP& b, const float* s_tab, int cg, const Vec16<T>& yv, const Vec16<T>& stored) {
    const V y = vec_to_f<T>(yv), g = vec_to_f<T>(stored);
    const V sc = ldf<T>(&s_tab[cg * EPV]), sh = ldf<T>(&s_tab[b.C + cg * EPV]);
    const V istd = ldf<T>(&s_tab[2 * b.C + cg * EPV]), mis = ldf<T>(&s_tab[3 * b.C + cg * EPV]);
    const V act = __builtin_elementwise_fma(y, sc, sh);
    const V dz = act > V(0.f) ? g : V(0.f);
    r1 += dz;
    r2 += dz * __builtin_elementwise_fma(y, istd, -mis);      // dz * xhat
  }
  // Block reduction in a fixed order, then ONE fixed-point add per channel and sum: lanes of a wave that share a channel
  // group (lane % G when G < 64) are summed with xor-shuffles, the waves meet in LDS (s_part: waves x 2 x C floats).
  // Blocks of 1024 threads keep the number of adds per launch low: every block ends with 2C same-address atomics, and
  // a thousand small blocks hammering 64 cache lines cost more than the kernel itself.
  __device__ __forceinline__ void finish(const BnrP& b, float* s_part, int G) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if (G < 64) {
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) {
        if (off >= G) {
#pragma unroll
          for (int e = 0; e < EPV; ++e) { r1[e] += __shfl_xor(r1[e], off); r2[e] += __shfl_xor(r2[e], off); }
        }
      }
    }
    // after the shuffles lanes 0..min(G,64)-1 hold the wave's sums of channel groups (threadIdx.x % G)
    const int gl = G < 64 ? G : 64;
    if (lane < gl) {
      const int g = threadIdx.x % G;
#pragma unroll
      for (int e = 0; e < EPV; ++e) { s_part[(wv * 2 + 0) * b.C + g * EPV + e] = r1[e]; s_part[(wv * 2 + 1) * b.C + g * EPV + e] = r2[e]; }
    }
    __syncthreads();
    // waves that own channel group g: all of them when G <= 64, else waves wv with (wv * 64) % G == g - (g % 64)
    for (int t = threadIdx.x; t < 2 * b.C; t += blockDim.x) {
      const int v = t / b.C, c = t - v * b.C;
      float sum = 0.f;
      if (G <= 64) { for (int q = 0; q < nw; ++q) sum += s_part[(q * 2 + v) * b.C + c]; }
      else { const int g = c / EPV, per = G / 64; for (int q = (g / 64); q < nw; q += per) sum += s_part[(q * 2 + v) * b.C + c]; }
      fx_add(b.sums + ((size_t)((blockIdx.x & (bn_sum_replicas(b.C) - 1)) * 2 + v) * b.C + c) * NUNET_FX_WORDS, sum);
    }
  }
};
static int bnr_fill(BnrP& b, const nunet_bnr_desc* d, int C, int dtype) {
  b.y = d->y; b.py = d->PY; b.mi = d->mean_invstd; b.gamma = d->gamma; b.beta = d->beta; b.sums = (long long*)d->sums; b.C = C;
  NUNET_REQUIRE(d->y && d->mean_invstd && d->gamma && d->beta && d->sums && d->PY % (16 / dtype_size(dtype)) == 0 && d->PY >= C,
                "fused BN-backward reduce: y, mean/invstd, gamma, beta, sums and an aligned pitch are required");
  return NUNET_OK;
}

// ---------------------------------------------------------------------------
// MaxPool2d(2,2)
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* __restrict__ x, int PX, T* __restrict__ y, int PY, int N, int H, int W, int C, Dec4 dc) {
  constexpr int EPV = Tr<T>::EPV;
  const int G = C / EPV, H2 = H / 2, W2 = W / 2;
  const int64_t total = (int64_t)N * H2 * W2 * G;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int cg, qx, qy, n; long long q;
    dec4(dc, i, cg, q, qx, qy, n);
    const int64_t p00 = ((int64_t)n * H + 2 * qy) * W + 2 * qx;
    float mx[EPV];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const Vec16<T> v = ld16(x + (p00 + (k >> 1) * W + (k & 1)) * PX + cg * EPV);
#pragma unroll
      for (int e = 0; e < EPV; ++e) mx[e] = k == 0 ? v.get(e) : fmaxf(mx[e], v.get(e));
    }
    Vec16<T> o;
#pragma unroll
    for (int e = 0; e < EPV; ++e) o.set(e, mx[e]);
    st16(y + q * PY + cg * EPV, o);
  }
}
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* __restrict__ x, int PX, const T* __restrict__ dy, int PDY, T* __restrict__ dx, int PDX, int accumulate, int N, int H, int W, int C, Dec4 dc) {
  constexpr int EPV = Tr<T>::EPV;
  const int G = C / EPV, H2 = H / 2, W2 = W / 2;
  const int64_t total = (int64_t)N * H2 * W2 * G;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int cg, qx, qy, n; long long q;
    dec4(dc, i, cg, q, qx, qy, n);
    const int64_t p00 = ((int64_t)n * H + 2 * qy) * W + 2 * qx;
    Vec16<T> v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = ld16(x + (p00 + (k >> 1) * W + (k & 1)) * PX + cg * EPV);
    const Vec16<T> g = ld16(dy + q * PDY + cg * EPV);
    int am[EPV];
#pragma unroll
    for (int e = 0; e < EPV; ++e) {
      float m = v[0].get(e);
      int a = 0;
#pragma unroll
      for (int k = 1; k < 4; ++k) {
        const float f = v[k].get(e);
        if (f > m) { m = f; a = k; }  // first maximum wins (PyTorch scan order)
      }
      am[e] = a;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      T* q4 = dx + (p00 + (k >> 1) * W + (k & 1)) * PDX + cg * EPV;
      Vec16<T> o = accumulate ? ld16(q4) : zero16<T>();
#pragma unroll
      for (int e = 0; e < EPV; ++e) {
        const float base = accumulate ? o.get(e) : 0.f;
        o.set(e, am[e] == k ? base + g.get(e) : base);
      }
      st16(q4, o);
    }
  }
}
template <typename T> static int launch_maxpool_fwd(int N, int H, int W, int C, const void* x, int PX, void* y, int PY, hipStream_t st) {
  const int64_t total = (int64_t)N * (H / 2) * (W / 2) * (C / Tr<T>::EPV);
  ProfScope ps(PC_POOL, 0, (double)N * H * W * C * sizeof(T) * 1.25, st);
  NUNET_LAUNCH((maxpool_fwd_kernel<T>), dim3(grid_for(total, 256)), dim3(256), 0, st, (const T*)x, PX, (T*)y, PY, N, H, W, C, make_dec4(total, C / Tr<T>::EPV, W / 2, H / 2));
  return nunet_check_launch("maxpool_fwd");
}
template <typename T> static int launch_maxpool_bwd(int N, int H, int W, int C, const void* x, int PX, const void* dy, int PDY, void* dx, int PDX, int acc, hipStream_t st) {
  const int64_t total = (int64_t)N * (H / 2) * (W / 2) * (C / Tr<T>::EPV);
  ProfScope ps(PC_POOL, 0, (double)N * H * W * C * sizeof(T) * (acc ? 3.25 : 2.25), st);
  NUNET_LAUNCH((maxpool_bwd_kernel<T>), dim3(grid_for(total, 256)), dim3(256), 0, st, (const T*)x, PX, (const T*)dy, PDY, (T*)dx, PDX, acc, N, H, W, C, make_dec4(total, C / Tr<T>::EPV, W / 2, H / 2));
  return nunet_check_launch("maxpool_bwd");
}
static int ew_check(const char* what, int dtype, int N, int H, int W, int C, int p0, int p1) {
  const int epv = 16 / dtype_size(dtype);
  NUNET_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % epv == 0, "%s: bad shape N=%d H=%d W=%d C=%d", what, N, H, W, C);
  NUNET_REQUIRE(p0 % epv == 0 && p1 % epv == 0 && p0 >= C && p1 >= C, "%s: pitch", what);
  return NUNET_OK;
}
extern "C" int nunet_maxpool2x2_fwd(int32_t dtype, int32_t N, int32_t H, int32_t W, int32_t C, const void* x, int32_t PX, void* y, int32_t PY, nunet_stream_t s) {
  int rc = ew_check("maxpool_fwd", dtype, N, H, W, C, PX, PY);
  if (rc) return rc;
  NUNET_REQUIRE(x && y && H % 2 == 0 && W % 2 == 0, "maxpool_fwd: needs even H, W");
  return NUNET_DISPATCH(dtype, launch_maxpool_fwd, N, H, W, C, x, PX, y, PY, (hipStream_t)s);
}
extern "C" int nunet_maxpool2x2_bwd(int32_t dtype, int32_t N, int32_t H, int32_t W, int32_t C, const void* x, int32_t PX, const void* dy, int32_t PDY, void* dx, int32_t PDX, int32_t accumulate, nunet_stream_t s) {
  int rc = ew_check("maxpool_bwd", dtype, N, H, W, C, PX, PDX);
  if (rc) return rc;
  NUNET_REQUIRE(x && dy && dx && H % 2 == 0 && W % 2 == 0 && PDY % (16 / dtype_size(dtype)) == 0, "maxpool_bwd: bad args");
  return NUNET_DISPATCH(dtype, launch_maxpool_bwd, N, H, W, C, x, PX, dy, PDY, dx, PDX, accumulate, (hipStream_t)s);
}

// ---------------------------------------------------------------------------
// Upsample x2, bilinear, align_corners=True
// ---------------------------------------------------------------------------
// (contraction off: whether `src - i0` fuses with the multiply into an fma is otherwise decided per call site, and the forward
//  pass, its backward and the form fused into the BatchNorm launch must agree on the weights to the last bit)
__device__ __forceinline__ void up_taps(int dst, float scale, int n_in, int& i0, int& i1, float& l1) {
#pragma clang fp contract(off)
  const float src = scale * (float)dst;
  i0 = (int)src;
  if (i0 > n_in - 1) i0 = n_in - 1;
  i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
  l1 = src - (float)i0;
}
// the four-tap interpolation, one explicit operation order for every kernel that evaluates it (the stand-alone upsample and
// the form fused into the BatchNorm launch must round identically)
__device__ __forceinline__ float up_lerp(float v00, float v01, float v10, float v11, float ly, float lx) {
#pragma clang fp contract(off)
  const float hy = 1.f - ly, hx = 1.f - lx;
  const float top = __builtin_fmaf(lx, v01, hx * v00), bot = __builtin_fmaf(lx, v11, hx * v10);
  return __builtin_fmaf(ly, bot, hy * top);
}
template <typename T>
__global__ __launch_bounds__(256) void upsample_fwd_kernel(const T* __restrict__ x, int PX, T* __restrict__ y, int PY, int N, int H, int W, int C, Dec4 dc) {
  constexpr int EPV = Tr<T>::EPV;
  const int G = C / EPV, HO = 2 * H, WO = 2 * W;
  const float sy = HO > 1 ? (float)(H - 1) / (float)(HO - 1) : 0.f;
  const float sx = WO > 1 ? (float)(W - 1) / (float)(WO - 1) : 0.f;
  const int64_t total = (int64_t)N * HO * WO * G;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int cg, ox, oy, n; long long o;
    dec4(dc, i, cg, o, ox, oy, n);
    int y0, y1, x0, x1; float ly, lx;
    up_taps(oy, sy, H, y0, y1, ly);
    up_taps(ox, sx, W, x0, x1, lx);
    const T* b = x + (int64_t)n * H * W * PX + cg * EPV;
    const Vec16<T> v00 = ld16(b + ((int64_t)y0 * W + x0) * PX), v01 = ld16(b + ((int64_t)y0 * W + x1) * PX);
    const Vec16<T> v10 = ld16(b + ((int64_t)y1 * W + x0) * PX), v11 = ld16(b + ((int64_t)y1 * W + x1) * PX);
    Vec16<T> r;
#pragma unroll
    for (int e = 0; e < EPV; ++e) r.set(e, up_lerp(v00.get(e), v01.get(e), v10.get(e), v11.get(e), ly, lx));
    st16(y + o * PY + cg * EPV, r);
  }
}
// gather form of the transposed interpolation: one thread per low-res pixel
template <typename T>
__global__ __launch_bounds__(256) void upsample_bwd_kernel(const T* __restrict__ dy, int PDY, T* __restrict__ dx, int PDX, int accumulate, int N, int H, int W, int C, Dec4 dc) {
  constexpr int EPV = Tr<T>::EPV;
  const int G = C / EPV, HO = 2 * H, WO = 2 * W;
  const float sy = HO > 1 ? (float)(H - 1) / (float)(HO - 1) : 0.f;
  const float sx = WO > 1 ? (float)(W - 1) / (float)(WO - 1) : 0.f;
  const int64_t total = (int64_t)N * H * W * G;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int cg, ix, iy, n; long long o;
    dec4(dc, i, cg, o, ix, iy, n);
    const T* b = dy + (int64_t)n * HO * WO * PDY + cg * EPV;
    float acc[EPV];
#pragma unroll
    for (int e = 0; e < EPV; ++e) acc[e] = 0.f;
    if (H >= 4 && W >= 4) {
      // The high-res rows that interpolate from low-res row iy have source coordinate in (iy-1, iy+1): an open interval
      // of 2/sy = 4 + 2/(H-1) < 5 output rows, so it holds at most 5 of them, found among the 6 starting at
      // floor((iy-1)/sy). Weights come from the same up_taps() as the forward pass; the 5x5 window is then read with
      // unconditional (clamped, zero-weighted) loads that the compiler can keep in flight together, in the same
      // (oy, ox) order as the general loop below.
      int yb, xb; float wy[5], wx[5];
      auto window = [&](int ic, float sc, int n_in, int n_out, int& base, float* w5) {
        base = max(0, (int)floorf((float)(ic - 1) / sc));
        float w6[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          int i0, i1; float l;
          up_taps(min(base + j, n_out - 1), sc, n_in, i0, i1, l);
          w6[j] = base + j < n_out ? (i0 == ic ? 1.f - l : 0.f) + (i1 == ic ? l : 0.f) : 0.f;
        }
        const bool shift = w6[0] == 0.f;
#pragma unroll
        for (int j = 0; j < 5; ++j) w5[j] = shift ? w6[j + 1] : w6[j];
        base += shift ? 1 : 0;
      };
      window(iy, sy, H, HO, yb, wy);
      window(ix, sx, W, WO, xb, wx);
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const int oy = min(yb + j, HO - 1);
#pragma unroll
        for (int k = 0; k < 5; ++k) {
          const int ox = min(xb + k, WO - 1);
          const Vec16<T> g = ld16(b + ((int64_t)oy * WO + ox) * PDY);
          const float w = wy[j] * wx[k];
#pragma unroll
          for (int e = 0; e < EPV; ++e) acc[e] += w * g.get(e);
        }
      }
    } else {
    int ylo = 0, yhi = HO - 1, xlo = 0, xhi = WO - 1;
    if (sy > 0.f) { ylo = max(0, (int)floorf((float)(iy - 1) / sy) - 1); yhi = min(HO - 1, (int)ceilf((float)(iy + 1) / sy) + 1); }
    if (sx > 0.f) { xlo = max(0, (int)floorf((float)(ix - 1) / sx) - 1); xhi = min(WO - 1, (int)ceilf((float)(ix + 1) / sx) + 1); }
    for (int oy = ylo; oy <= yhi; ++oy) {
      int y0, y1; float ly;
      up_taps(oy, sy, H, y0, y1, ly);
      const float wy = (y0 == iy ? 1.f - ly : 0.f) + (y1 == iy ? ly : 0.f);
      if (wy == 0.f) continue;
      for (int ox = xlo; ox <= xhi; ++ox) {
        int x0, x1; float lx;
        up_taps(ox, sx, W, x0, x1, lx);
        const float wx = (x0 == ix ? 1.f - lx : 0.f) + (x1 == ix ? lx : 0.f);
        if (wx == 0.f) continue;
        const Vec16<T> g = ld16(b + ((int64_t)oy * WO + ox) * PDY);
        const float w = wy * wx;
#pragma unroll
        for (int e = 0; e < EPV; ++e) acc[e] += w * g.get(e);
      }
    }
    }
    T* q = dx + o * PDX + cg * EPV;
    Vec16<T> r = accumulate ? ld16(q) : zero16<T>();
#pragma unroll
    for (int e = 0; e < EPV; ++e) r.set(e, (accumulate ? r.get(e) : 0.f) + acc[e]);
    st16(q, r);
  }
}
template <typename T> static int launch_up_fwd(int N, int H, int W, int C, const void* x, int PX, void* y, int PY, hipStream_t st) {
  const int64_t total = (int64_t)N * 4 * H * W * (C / Tr<T>::EPV);
  ProfScope ps(PC_UP_FWD, 0, (double)N * H * W * C * sizeof(T) * 5.0, st);
  NUNET_LAUNCH((upsample_fwd_kernel<T>), dim3(grid_for(total, 256)), dim3(256), 0, st, (const T*)x, PX, (T*)y, PY, N, H, W, C, make_dec4(total, C / Tr<T>::EPV, 2 * W, 2 * H));
  return nunet_check_launch("upsample_fwd");
}
template <typename T> static int launch_up_bwd(int N, int H, int W, int C, const void* dy, int PDY, void* dx, int PDX, int acc, hipStream_t st) {
  const int64_t total = (int64_t)N * H * W * (C / Tr<T>::EPV);
  ProfScope ps(PC_UP_BWD, 0, (double)N * H * W * C * sizeof(T) * (acc ? 6.0 : 5.0), st);
  NUNET_LAUNCH((upsample_bwd_kernel<T>), dim3(grid_for(total, 256)), dim3(256), 0, st, (const T*)dy, PDY, (T*)dx, PDX, acc, N, H, W, C, make_dec4(total, C / Tr<T>::EPV, W, H));
  return nunet_check_launch("upsample_bwd");
}
extern "C" int nunet_upsample2x_fwd(int32_t dtype, int32_t N, int32_t H, int32_t W, int32_t C, const void* x, int32_t PX, void* y, int32_t PY, nunet_stream_t s) {
  int rc = ew_check("upsample_fwd", dtype, N, H, W, C, PX, PY);
  if (rc) return rc;
  NUNET_REQUIRE(x && y, "upsample_fwd: null pointer");
  return NUNET_DISPATCH(dtype, launch_up_fwd, N, H, W, C, x, PX, y, PY, (hipStream_t)s);
}
extern "C" int nunet_upsample2x_bwd(int32_t dtype, int32_t N, int32_t H, int32_t W, int32_t C, const void* dy, int32_t PDY, void* dx, int32_t PDX, int32_t accumulate, nunet_stream_t s) {
  int rc = ew_check("upsample_bwd", dtype, N, H, W, C, PDY, PDX);
  if (rc) return rc;
  NUNET_REQUIRE(dy && dx, "upsample_bwd: null pointer");
  return NUNET_DISPATCH(dtype, launch_up_bwd, N, H, W, C, dy, PDY, dx, PDX, accumulate, (hipStream_t)s);
}

// ---------------------------------------------------------------------------
// 1x1 heads. Lane = channel: 32-lane half-waves walk pixels.
// ---------------------------------------------------------------------------
#define HEAD_MAXK 8
template <typename T>
__global__ __launch_bounds__(256) void head_fwd_kernel(const T* __restrict__ x, int PX, const float* __restrict__ w, const float* __restrict__ b, float* __restrict__ logits, int N, int H, int W, int C, int K) {
  // thread per pixel; C (=32) channels read as 16-byte vectors
  constexpr int EPV = Tr<T>::EPV;
  __shared__ float s_w[HEAD_MAXK * 64];
  for (int i = threadIdx.x; i < K * C; i += blockDim.x) s_w[i] = w[i];
  __syncthreads();
  const int64_t hw = (int64_t)H * W, npix = (int64_t)N * hw;
  for (int64_t pix = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; pix < npix; pix += (int64_t)gridDim.x * blockDim.x) {
    float acc[HEAD_MAXK];
#pragma unroll
    for (int k = 0; k < HEAD_MAXK; ++k) acc[k] = 0.f;
    for (int c0 = 0; c0 < C; c0 += EPV) {
      const Vec16<T> v = ld16(x + pix * PX + c0);
#pragma unroll
      for (int e = 0; e < EPV; ++e) {
        const float xv = v.get(e);
#pragma unroll
        for (int k = 0; k < HEAD_MAXK; ++k)
          if (k < K) acc[k] += xv * s_w[k * C + c0 + e];
      }
    }
    const int n = npix < (1ll << 31) ? (int)((unsigned)pix / (unsigned)hw) : (int)(pix / hw);
    const int64_t rem = pix - n * hw;
#pragma unroll
    for (int k = 0; k < HEAD_MAXK; ++k)
      if (k < K) logits[((int64_t)n * K + k) * hw + rem] = acc[k] + b[k];
  }
}
template <typename T, int KT, bool BNR>
__global__ __launch_bounds__(256) void head_bwd_kernel(const T* __restrict__ x, int PX, const float* __restrict__ w, const float* __restrict__ dl, T* __restrict__ dx, int PDX, int accumulate, float* __restrict__ dw, int N, int H, int W, int C, int K, BnrP bn) {
  // C == 32. A thread owns one 16-byte channel group (EPV channels) of a pixel; G = 32/EPV threads
  // cover a pixel. dW/db partials live in registers (KT = compile-time class count, 0 = generic)
  // and meet through LDS once per block.
  constexpr int EPV = Tr<T>::EPV;
  constexpr int G = 32 / EPV;
  constexpr int KM = KT > 0 ? KT : HEAD_MAXK;
  __shared__ float s_dw[4 * (HEAD_MAXK * 32 + HEAD_MAXK)];   // per wave
  const int cg = threadIdx.x % G, pl = threadIdx.x / G, ppb = blockDim.x / G;
  __shared__ __attribute__((aligned(16))) float s_part[BNR ? 4096 : 1];   // waves x 2 x C floats (BnrAcc::finish)
  __shared__ __attribute__((aligned(16))) float s_tab[BNR ? 4 * 32 : 1];
  BnrAcc<T> ba;
  if constexpr (BNR) ba.init(bn, s_tab);
  float wk[KM][EPV], aw[KM][EPV], ab[KM];
#pragma unroll
  for (int k = 0; k < KM; ++k) {
    ab[k] = 0.f;
#pragma unroll
    for (int e = 0; e < EPV; ++e) { wk[k][e] = k < K ? w[k * C + cg * EPV + e] : 0.f; aw[k][e] = 0.f; }
  }
  const int hw = H * W;
  const int64_t npix = (int64_t)N * hw;
  // U pixels per thread and pass, every load of a pass issued before the first use (the kernel opens the backward
  // pass alone on the device, so its latency is the step's)
  constexpr int U = 4;
  const int64_t stride = (int64_t)gridDim.x * ppb;
  for (int64_t pix0 = (int64_t)blockIdx.x * ppb + pl; pix0 < npix; pix0 += U * stride) {
    Vec16<T> xv[U], ov[U], yb[BNR ? U : 1];
    float d[U][KM];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t pix = pix0 + u * stride;
      if (pix < npix) {
        const int n = (int)((unsigned)pix / (unsigned)hw);      // npix < 2^31 (checked on the host)
        const int rem = (int)pix - n * hw;
        xv[u] = ld16(x + pix * PX + cg * EPV);
        if (dx && accumulate) ov[u] = ld16(dx + pix * PDX + cg * EPV);
        if constexpr (BNR) yb[u] = ba.fetch(bn, cg, pix);
#pragma unroll
        for (int k = 0; k < KM; ++k) d[u][k] = (KT > 0 || k < K) ? dl[((int64_t)n * K + k) * hw + rem] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t pix = pix0 + u * stride;
      if (pix < npix) {
        float g[EPV];
#pragma unroll
        for (int e = 0; e < EPV; ++e) g[e] = 0.f;
#pragma unroll
        for (int k = 0; k < KM; ++k) {
          if (KT > 0 || k < K) {
            if (cg == 0) ab[k] += d[u][k];
#pragma unroll
            for (int e = 0; e < EPV; ++e) { g[e] += d[u][k] * wk[k][e]; aw[k][e] += d[u][k] * xv[u].get(e); }
          }
        }
        if (dx) {
          Vec16<T> o;
#pragma unroll
          for (int e = 0; e < EPV; ++e) o.set(e, (accumulate ? ov[u].get(e) : 0.f) + g[e]);
          st16(dx + pix * PDX + cg * EPV, o);
          if constexpr (BNR) ba.add(bn, s_tab, cg, yb[u], o);
        }
      }
    }
  }
  // block reduction in a fixed order (bit-reproducible): lanes that own the same channel group are summed with
  // xor-shuffles, the four waves through LDS
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < KM; ++k) {
    if (k < K) {
#pragma unroll
      for (int off = G; off < 64; off <<= 1) {
#pragma unroll
        for (int e = 0; e < EPV; ++e) aw[k][e] += __shfl_xor(aw[k][e], off);
        ab[k] += __shfl_xor(ab[k], off);
      }
      if (lane < G) {
#pragma unroll
        for (int e = 0; e < EPV; ++e) s_dw[wv * (HEAD_MAXK * 33) + k * 32 + lane * EPV + e] = aw[k][e];
        if (lane == 0) s_dw[wv * (HEAD_MAXK * 33) + HEAD_MAXK * 32 + k] = ab[k];
      }
    }
  }
  __syncthreads();
  // per-block slab [K*32 weights | K biases]: plain stores, summed by the caller (no same-address atomics)
  float* slab = dw + (size_t)blockIdx.x * (K * 33);
  for (int i = threadIdx.x; i < K * 33; i += blockDim.x) {
    const int j = i < K * 32 ? i : HEAD_MAXK * 32 + (i - K * 32);
    slab[i] = (s_dw[j] + s_dw[HEAD_MAXK * 33 + j]) + (s_dw[2 * HEAD_MAXK * 33 + j] + s_dw[3 * HEAD_MAXK * 33 + j]);
  }
  if constexpr (BNR) ba.finish(bn, s_part, G);
}
template <typename T> static int launch_head_fwd(int N, int H, int W, int C, int K, const void* x, int PX, const float* w, const float* b, float* logits, hipStream_t st) {
  ProfScope ps(PC_HEAD, 2.0 * N * H * W * C * K, (double)N * H * W * (C * sizeof(T) + K * 4), st);
  NUNET_LAUNCH((head_fwd_kernel<T>), dim3(grid_for((int64_t)N * H * W, 256)), dim3(256), 0, st, (const T*)x, PX, w, b, logits, N, H, W, C, K);
  return nunet_check_launch("head_fwd");
}
template <typename T> static int launch_head_bwd(int N, int H, int W, int C, int K, const void* x, int PX, const float* w, const float* dl, void* dx, int PDX, int acc, float* dw_slabs, int nslabs, const nunet_bnr_desc* bnr, hipStream_t st) {
  ProfScope ps(PC_HEAD, 4.0 * N * H * W * C * K, (double)N * H * W * (C * sizeof(T) * ((acc ? 3 : 2) + (bnr ? 1 : 0)) + K * 4), st);
  const dim3 grid(nslabs), blk(256);
  BnrP b; memset(&b, 0, sizeof(b));
  if (bnr) { NUNET_REQUIRE(dx, "head_bwd: fused BN reduce needs dx"); int rc = bnr_fill(b, bnr, C, Tr<T>::DT); if (rc) return rc; }
#define NUNET_HB(KT, BN_) NUNET_LAUNCH((head_bwd_kernel<T, KT, BN_>), grid, blk, 0, st, (const T*)x, PX, w, dl, (T*)dx, PDX, acc, dw_slabs, N, H, W, C, K, b)
  if (bnr) { if (K == 1) NUNET_HB(1, true); else if (K == 2) NUNET_HB(2, true); else if (K == 4) NUNET_HB(4, true); else NUNET_HB(0, true); }
  else { if (K == 1) NUNET_HB(1, false); else if (K == 2) NUNET_HB(2, false); else if (K == 4) NUNET_HB(4, false); else NUNET_HB(0, false); }
#undef NUNET_HB
  return nunet_check_launch("head_bwd");
}
extern "C" int nunet_head_fwd(int32_t dtype, int32_t N, int32_t H, int32_t W, int32_t C, int32_t K, const void* x, int32_t PX, const float* w, const float* b, float* logits, nunet_stream_t s) {
  NUNET_REQUIRE(x && w && b && logits, "head_fwd: null pointer");
  NUNET_REQUIRE(C > 0 && C <= 64 && C % (16 / dtype_size(dtype)) == 0 && K >= 1 && K <= HEAD_MAXK, "head_fwd: C=%d K=%d unsupported (C<=64, K<=%d)", C, K, HEAD_MAXK);
  NUNET_REQUIRE(PX % (16 / dtype_size(dtype)) == 0, "head_fwd: pitch");
  return NUNET_DISPATCH(dtype, launch_head_fwd, N, H, W, C, K, x, PX, w, b, logits, (hipStream_t)s);
}
extern "C" int nunet_head_bwd_bnr(int32_t dtype, int32_t N, int32_t H, int32_t W, int32_t C, int32_t K, const void* x, int32_t PX, const float* w, const float* dlogits, void* dx, int32_t PDX, int32_t accumulate, float* dw_slabs, int32_t nslabs, const nunet_bnr_desc* bnr, nunet_stream_t s) {
  NUNET_REQUIRE(x && w && dlogits && dw_slabs && nslabs >= 1 && nslabs <= 4096, "head_bwd: bad args");
  NUNET_REQUIRE(C == 32 && K >= 1 && K <= HEAD_MAXK, "head_bwd: C=%d K=%d unsupported (C==32, K<=%d)", C, K, HEAD_MAXK);
  NUNET_REQUIRE((int64_t)N * H * W < (1LL << 31), "head_bwd: too many pixels");
  return NUNET_DISPATCH(dtype, launch_head_bwd, N, H, W, C, K, x, PX, w, dlogits, dx, PDX, accumulate, dw_slabs, nslabs, bnr, (hipStream_t)s);
}
extern "C" int nunet_head_bwd(int32_t dtype, int32_t N, int32_t H, int32_t W, int32_t C, int32_t K, const void* x, int32_t PX, const float* w, const float* dlogits, void* dx, int32_t PDX, int32_t accumulate, float* dw_slabs, int32_t nslabs, nunet_stream_t s) {
  return nunet_head_bwd_bnr(dtype, N, H, W, C, K, x, PX, w, dlogits, dx, PDX, accumulate, dw_slabs, nslabs, nullptr, s);
}

// ---------------------------------------------------------------------------
// BCEDiceLoss (losses.py:107-117)
// ---------------------------------------------------------------------------
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

constexpr int BCE_GX = 64;      // most blocks per image
__global__ __launch_bounds__(256) void bce_dice_partial_kernel(const float* __restrict__ x, const float* __restrict__ t, int64_t per, float* __restrict__ ws, int N) {
  const int n = blockIdx.y;
  const float* xs = x + (int64_t)n * per;
  const float* ts = t + (int64_t)n * per;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < per; i += (int64_t)gridDim.x * blockDim.x) {
    const float xv = xs[i], tv = ts[i];
    const float pv = sigmoidf_(xv);
    a0 += pv * tv; a1 += pv; a2 += tv;
    a3 += fmaxf(xv, 0.f) - xv * tv + log1pf(expf(-fabsf(xv)));
  }
  a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2); a3 = wave_sum(a3);
  __shared__ float s[4][4];
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { s[wv][0] = a0; s[wv][1] = a1; s[wv][2] = a2; s[wv][3] = a3; }
  __syncthreads();
  // per-block partials to a slab (plain stores; the final kernel sums them in a fixed order: bit-reproducible)
  if (threadIdx.x < 4)
    ws[3 * N + 1 + ((size_t)n * BCE_GX + blockIdx.x) * 4 + threadIdx.x] = s[0][threadIdx.x] + s[1][threadIdx.x] + s[2][threadIdx.x] + s[3][threadIdx.x];
}
__global__ void bce_dice_final_kernel(float* __restrict__ ws, int N, int gx, int64_t per, float* __restrict__ loss) {
  // single wave: lane l holds partial l of image n (gx <= 64)
  const int l = threadIdx.x;
  float bce = 0.f, d = 0.f;
  for (int n = 0; n < N; ++n) {
    const float* q = ws + 3 * N + 1 + ((size_t)n * BCE_GX + l) * 4;
    float v0 = l < gx ? q[0] : 0.f, v1 = l < gx ? q[1] : 0.f, v2 = l < gx ? q[2] : 0.f, v3 = l < gx ? q[3] : 0.f;
    v0 = wave_sum(v0); v1 = wave_sum(v1); v2 = wave_sum(v2); v3 = wave_sum(v3);   // (butterfly: every lane holds the totals)
    if (l == 0) { ws[n * 3] = v0; ws[n * 3 + 1] = v1; ws[n * 3 + 2] = v2; }
    bce += v3;
    d += (2.f * v0 + 1e-5f) / (v1 + v2 + 1e-5f);
  }
  if (l == 0) {
    ws[N * 3] = bce;
    loss[0] = 0.5f * (bce / ((float)N * (float)per)) + (1.f - d / (float)N);
  }
}
__global__ __launch_bounds__(256) void bce_dice_bwd_kernel(const float* __restrict__ x, const float* __restrict__ t, int64_t per, const float* __restrict__ ws, const float* __restrict__ gscale, float* __restrict__ dx, int N) {
  const int n = blockIdx.y;
  const float I = ws[n * 3], D = ws[n * 3 + 1] + ws[n * 3 + 2] + 1e-5f;
  const float g = gscale ? gscale[0] : 1.f;
  const float kb = 0.5f / ((float)N * (float)per);
  const float num = 2.f * I + 1e-5f;
  const float invD2 = 1.f / (D * D);
  const float invN = 1.f / (float)N;
  const float* xs = x + (int64_t)n * per;
  const float* ts = t + (int64_t)n * per;
  float* ds = dx + (int64_t)n * per;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < per; i += (int64_t)gridDim.x * blockDim.x) {
    const float xv = xs[i], tv = ts[i];
    const float pv = sigmoidf_(xv);
    const float ddice = (2.f * tv * D - num) * invD2 * pv * (1.f - pv);
    ds[i] = g * (kb * (pv - tv) - invN * ddice);
  }
}
extern "C" size_t nunet_bce_dice_ws_bytes(int32_t N) { return (size_t)(3 * N + 1 + (size_t)N * BCE_GX * 4) * sizeof(float); }
extern "C" int nunet_bce_dice_fwd(const float* logits, const float* target, int32_t N, int64_t per, float* ws, size_t ws_bytes, float* loss, nunet_stream_t s) {
  NUNET_REQUIRE(logits && target && ws && loss && N > 0 && per > 0, "bce_dice_fwd: bad args");
  NUNET_REQUIRE(ws_bytes >= nunet_bce_dice_ws_bytes(N), "bce_dice_fwd: workspace of %zu bytes, nunet_bce_dice_ws_bytes(%d) = %zu", ws_bytes, (int)N, nunet_bce_dice_ws_bytes(N));
  hipStream_t st = (hipStream_t)s;
  const int gx = grid_for(per, 256 * 4, BCE_GX);
  ProfScope ps(PC_LOSS, 0, (double)N * per * 8, st);
  NUNET_LAUNCH(bce_dice_partial_kernel, dim3(gx, N), dim3(256), 0, st, logits, target, per, ws, N);
  NUNET_LAUNCH(bce_dice_final_kernel, dim3(1), dim3(64), 0, st, ws, N, gx, per, loss);
  return nunet_check_launch("bce_dice_fwd");
}
extern "C" int nunet_bce_dice_bwd(const float* logits, const float* target, int32_t N, int64_t per, const float* ws, size_t ws_bytes, const float* gscale, float* dlogits, nunet_stream_t s) {
  NUNET_REQUIRE(logits && target && ws && dlogits && N > 0 && per > 0, "bce_dice_bwd: bad args");
  NUNET_REQUIRE(ws_bytes >= nunet_bce_dice_ws_bytes(N), "bce_dice_bwd: workspace of %zu bytes, nunet_bce_dice_ws_bytes(%d) = %zu", ws_bytes, (int)N, nunet_bce_dice_ws_bytes(N));
  const int gx = grid_for(per, 256 * 4, 64);
  ProfScope ps(PC_LOSS, 0, (double)N * per * 12, (hipStream_t)s);
  NUNET_LAUNCH(bce_dice_bwd_kernel, dim3(gx, N), dim3(256), 0, (hipStream_t)s, logits, target, per, ws, gscale, dlogits, N);
  return nunet_check_launch("bce_dice_bwd");
}

// ---------------------------------------------------------------------------
// Fused loss step of the training loop (trains.py:118-128,135-136): all heads' BCEDice
// partial sums + IoU counts of the last head in one launch; then gradients of the
// head-averaged loss, the loss values and the running epoch meters in a second one.
// ---------------------------------------------------------------------------
// Per-block partial sums go to a slab [heads][N][LOSS_GX] x {I, P, T, BCE, iou-inter, iou-union} with plain stores (no
// zero launch ahead, no same-address atomics, and bit-reproducible run to run); the second kernel adds the <= 64 slabs
// of its image with one wave.
constexpr int LOSS_GX = 64;     // most blocks per (image, head)
__global__ __launch_bounds__(256) void loss_step_partial_kernel(const float* __restrict__ x, const float* __restrict__ t, int64_t per, float* __restrict__ ws, int N, int heads, float iou_thr) {
  const int n = blockIdx.y, hd = blockIdx.z;
  const float* xs = x + ((int64_t)hd * N + n) * per;
  const float* ts = t + (int64_t)n * per;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  unsigned ci = 0, cu = 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < per; i += (int64_t)gridDim.x * blockDim.x) {
    const float xv = xs[i], tv = ts[i];
    const float pv = sigmoidf_(xv);
    a0 += pv * tv; a1 += pv; a2 += tv;
    a3 += fmaxf(xv, 0.f) - xv * tv + log1pf(expf(-fabsf(xv)));
    const bool a = xv >= iou_thr, b = tv > 0.5f; ci += (a && b) ? 1u : 0u; cu += (a || b) ? 1u : 0u;
  }
  a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2); a3 = wave_sum(a3);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { ci += __shfl_xor(ci, o); cu += __shfl_xor(cu, o); }
  __shared__ float s_f[4][6];
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { s_f[wv][0] = a0; s_f[wv][1] = a1; s_f[wv][2] = a2; s_f[wv][3] = a3; s_f[wv][4] = (float)ci; s_f[wv][5] = (float)cu; }
  __syncthreads();
  if (threadIdx.x < 6)    // (counts <= 256 * trip count per wave: exact in fp32 up to 2^24 per block)
    ws[((((size_t)hd * N + n) * LOSS_GX) + blockIdx.x) * 6 + threadIdx.x] = s_f[0][threadIdx.x] + s_f[1][threadIdx.x] + s_f[2][threadIdx.x] + s_f[3][threadIdx.x];
}
__global__ __launch_bounds__(256) void loss_step_bwd_kernel(const float* __restrict__ x, const float* __restrict__ t, int64_t per, const float* __restrict__ ws, int gx, int N, int heads, float* __restrict__ dx, float* __restrict__ loss_out, double* __restrict__ meters) {
  const int n = blockIdx.y, hd = blockIdx.z;
  const int wv = threadIdx.x >> 6, l = threadIdx.x & 63;
  const bool on = l < gx;
  __shared__ float s_img[3];
  __shared__ float s_acc[4][8][2];     // [wave][head] {sum of per-image dice terms, BCE sum}
  __shared__ double s_cnt[4][2];
  auto slab = [&](int k, int q) { return ws + ((((size_t)k * N + q) * LOSS_GX) + l) * 6; };
  if (wv == 0) {
    const float* w = slab(hd, n);
    const float I = wave_sum(on ? w[0] : 0.f), P = wave_sum(on ? w[1] : 0.f), T = wave_sum(on ? w[2] : 0.f);
    if (l == 0) { s_img[0] = I; s_img[1] = P; s_img[2] = T; }
  }
  // one block also owns the loss per head, their mean (trains.py:120-123) and the IoU of the last head
  // (trains.py:124,128): its four waves take four images each per pass, all loads of a pass in flight together
  const bool fin = blockIdx.x == 0 && n == 0 && hd == 0;
  if (fin) {
    double ci_s = 0.0, cu_s = 0.0;
    for (int k = 0; k < heads; ++k) {
      float d = 0.f, bce = 0.f;
      for (int q0 = wv; q0 < N; q0 += 16) {
        float v[4][6];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int q = q0 + 4 * u;
#pragma unroll
          for (int c = 0; c < 6; ++c) v[u][c] = (on && q < N) ? slab(k, q)[c] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (q0 + 4 * u < N) {
            const float I = wave_sum(v[u][0]), P = wave_sum(v[u][1]), T = wave_sum(v[u][2]);
            d += (2.f * I + 1e-5f) / (P + T + 1e-5f);
            bce += wave_sum(v[u][3]);
            if (k == heads - 1) { ci_s += (double)wave_sum(v[u][4]); cu_s += (double)wave_sum(v[u][5]); }
          }
        }
      }
      if (l == 0) { s_acc[wv][k][0] = d; s_acc[wv][k][1] = bce; }
    }
    if (l == 0) { s_cnt[wv][0] = ci_s; s_cnt[wv][1] = cu_s; }
  }
  __syncthreads();
  if (fin && threadIdx.x == 0) {
    float mean = 0.f;
    for (int k = 0; k < heads; ++k) {
      const float d = s_acc[0][k][0] + s_acc[1][k][0] + s_acc[2][k][0] + s_acc[3][k][0];
      const float bce = s_acc[0][k][1] + s_acc[1][k][1] + s_acc[2][k][1] + s_acc[3][k][1];
      const float lk = 0.5f * bce / ((float)N * (float)per) + (1.f - d / (float)N);
      loss_out[k] = lk;
      mean += lk;
    }
    mean /= (float)heads;
    loss_out[heads] = mean;
    if (meters) {
      const double inter = s_cnt[0][0] + s_cnt[1][0] + s_cnt[2][0] + s_cnt[3][0], uni = s_cnt[0][1] + s_cnt[1][1] + s_cnt[2][1] + s_cnt[3][1];
      meters[0] += (double)mean;
      meters[1] += (inter + 1e-5) / (uni + 1e-5);
      meters[2] = inter; meters[3] = uni;
    }
  }
  const float I = s_img[0], D = s_img[1] + s_img[2] + 1e-5f;
  const float g = 1.f / (float)heads;
  const float kb = 0.5f / ((float)N * (float)per);
  const float num = 2.f * I + 1e-5f;
  const float invD2 = 1.f / (D * D);
  const float invN = 1.f / (float)N;
  const float* xs = x + ((int64_t)hd * N + n) * per;
  const float* ts = t + (int64_t)n * per;
  float* ds = dx + ((int64_t)hd * N + n) * per;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < per; i += (int64_t)gridDim.x * blockDim.x) {
    const float xv = xs[i], tv = ts[i];
    const float pv = sigmoidf_(xv);
    const float ddice = (2.f * tv * D - num) * invD2 * pv * (1.f - pv);
    ds[i] = g * (kb * (pv - tv) - invN * ddice);
  }
}
size_t lovasz_step_ws_bytes(int32_t N, int64_t per, int32_t heads);    // lovasz.hip
int lovasz_loss_step(const float* logits, const float* target, int32_t N, int64_t per, int32_t heads, float* ws, float* dlogits,
                     float* loss_out, double* meters, float iou_thr, hipStream_t st);
extern "C" size_t nunet_loss_step_ws_bytes(int32_t N, int64_t per, int32_t heads, int32_t loss_kind) {
  if (N <= 0 || per <= 0 || heads < 1) return 0;
  if (loss_kind == NUNET_LOSS_LOVASZ_HINGE) return lovasz_step_ws_bytes(N, per, heads);
  return (size_t)heads * N * LOSS_GX * 6 * sizeof(float);
}
extern "C" int nunet_loss_step(const float* logits, const float* target, int32_t N, int64_t per, int32_t heads, int32_t loss_kind, float* ws, size_t ws_bytes,
                               float* dlogits, float* loss_out, double* meters, float iou_logit_threshold, nunet_stream_t s) {
  NUNET_REQUIRE(logits && target && ws && dlogits && loss_out && N > 0 && per > 0 && heads >= 1 && heads <= 8, "loss_step: bad args");
  NUNET_REQUIRE(loss_kind == NUNET_LOSS_BCE_DICE || loss_kind == NUNET_LOSS_LOVASZ_HINGE, "loss_step: loss_kind %d", (int)loss_kind);
  NUNET_REQUIRE(per <= (1ll << 24), "loss_step: image too large");    // per-image IoU counts stay exact in fp32
  NUNET_REQUIRE(ws_bytes >= nunet_loss_step_ws_bytes(N, per, heads, loss_kind), "loss_step: workspace of %zu bytes, nunet_loss_step_ws_bytes = %zu",
                ws_bytes, nunet_loss_step_ws_bytes(N, per, heads, loss_kind));
  hipStream_t st = (hipStream_t)s;
  if (loss_kind == NUNET_LOSS_LOVASZ_HINGE) return lovasz_loss_step(logits, target, N, per, heads, ws, dlogits, loss_out, meters, iou_logit_threshold, st);
  const int gx = grid_for(per, 256, LOSS_GX);     // one element per thread up to 128x128 images: the step waits on this pair of launches
  ProfScope ps(PC_LOSS, 0, (double)N * per * heads * 16, st);
  NUNET_LAUNCH(loss_step_partial_kernel, dim3(gx, N, heads), dim3(256), 0, st, logits, target, per, ws, N, heads, iou_logit_threshold);
  NUNET_LAUNCH(loss_step_bwd_kernel, dim3(gx, N, heads), dim3(256), 0, st, logits, target, per, ws, gx, N, heads, dlogits, loss_out, meters);
  return nunet_check_launch("loss_step");
}

// ---------------------------------------------------------------------------
// iou_score counts (metrics.py:10-14): sigmoid(x) > 0.5  <=>  x >= thr, with thr the smallest fp32 logit whose
// REFERENCE sigmoid (fp32, rounded) exceeds 0.5 - not x > 0: in fp32, sigmoid(x) == 0.5 exactly for 0 < x <~ 6e-8.
// The caller derives thr from the reference's own sigmoid by bisection (metrics.iou_logit_threshold). NaN -> false.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void iou_counts_kernel(const float* __restrict__ x, const float* __restrict__ t, int64_t n, float thr, unsigned long long* __restrict__ counts) {
  unsigned int ci = 0, cu = 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const bool a = x[i] >= thr, b = t[i] > 0.5f;
    ci += (a && b) ? 1u : 0u;
    cu += (a || b) ? 1u : 0u;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { ci += __shfl_xor(ci, o); cu += __shfl_xor(cu, o); }
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&counts[0], (unsigned long long)ci);
    atomicAdd(&counts[1], (unsigned long long)cu);
  }
}
extern "C" int nunet_iou_counts(const float* logits, const float* target, int64_t n, float logit_threshold, unsigned long long* counts, nunet_stream_t s) {
  NUNET_REQUIRE(logits && target && counts && n > 0, "iou_counts: bad args");
  ProfScope ps(PC_LOSS, 0, (double)n * 8, (hipStream_t)s);
  NUNET_LAUNCH(iou_counts_kernel, dim3(grid_for(n, 256 * 4, 256)), dim3(256), 0, (hipStream_t)s, logits, target, n, logit_threshold, counts);
  return nunet_check_launch("iou_counts");
}

// ---------------------------------------------------------------------------
// Mask export of the evaluation driver: uint8(sigmoid(logit) * 255), truncating like numpy's astype('uint8')
// (reference val.py:100-105: `(output[i, c] * 255).astype('uint8')` after torch.sigmoid)
// ---------------------------------------------------------------------------
// byte = number of thresholds <= x: thr[k-1] (k = 1..255) is the smallest fp32 logit whose reference byte is >= k, found
// by the caller with the reference's own sigmoid (bisection over fp32 bit patterns, metrics.sigmoid_u8_thresholds), so
// the byte matches `(torch.sigmoid(x) * 255).astype('uint8')` EXACTLY - a device expf one ulp away from the host's would
// flip bytes whose sigmoid * 255 sits next to an integer. Branch-free 8-step binary search in LDS.
__device__ __forceinline__ uint32_t sigmoid_byte(const float* s_thr, float x) {
  int lo = 0;                                   // invariant: thr[lo-1] <= x (or lo == 0), answer in [lo, lo + span]
#pragma unroll
  for (int span = 128; span > 0; span >>= 1) {
    const int mid = lo + span;                  // candidate count
    if (mid <= 255 && s_thr[mid - 1] <= x) lo = mid;
  }
  return (uint32_t)lo;
}
__global__ __launch_bounds__(256) void sigmoid_u8_kernel(const float* __restrict__ logits, const float* __restrict__ thr, uint8_t* __restrict__ out, int64_t n) {
  __shared__ float s_thr[256];
  if (threadIdx.x < 255) s_thr[threadIdx.x] = thr[threadIdx.x];
  __syncthreads();
  const int64_t n4 = n / 4;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const f32x4 v = reinterpret_cast<const f32x4*>(logits)[i];
    uint32_t w = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) w |= sigmoid_byte(s_thr, v[e]) << (8 * e);
    reinterpret_cast<uint32_t*>(out)[i] = w;
  }
  if (blockIdx.x == 0 && threadIdx.x < (unsigned)(n - n4 * 4)) {
    const int64_t i = n4 * 4 + threadIdx.x;
    out[i] = (uint8_t)sigmoid_byte(s_thr, logits[i]);
  }
}
extern "C" int nunet_sigmoid_u8(const float* logits, const float* thresholds, uint8_t* out, int64_t n, nunet_stream_t s) {
  NUNET_REQUIRE(logits && thresholds && out && n > 0, "sigmoid_u8: bad args");
  NUNET_REQUIRE(((uintptr_t)logits & 15) == 0 && ((uintptr_t)out & 3) == 0, "sigmoid_u8: logits must be 16-byte and out 4-byte aligned");
  NUNET_LAUNCH(sigmoid_u8_kernel, dim3(grid_for(n / 4 + 1, 256, 2048)), dim3(256), 0, (hipStream_t)s, logits, thresholds, out, n);
  return nunet_check_launch("sigmoid_u8");
}

// ---------------------------------------------------------------------------
// SGD with momentum / weight decay / nesterov (torch.optim.SGD semantics)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, int64_t n, const float* __restrict__ lr_dev, float mom, float wd, int nesterov, int first, float gscale) {
  const float lr = lr_dev[0];
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float pv = p[i];
    float gv = g[i] * gscale + wd * pv;
    if (mom != 0.f) {
      const float b = first ? gv : mom * m[i] + gv;
      m[i] = b;
      gv = nesterov ? gv + mom * b : b;
    }
    p[i] = pv - lr * gv;
  }
}
extern "C" int nunet_sgd_step(float* p, const float* g, float* mom, int64_t n, const float* lr_dev, float momentum, float weight_decay, int32_t nesterov, int32_t first, float grad_scale, nunet_stream_t s) {
  NUNET_REQUIRE(p && g && lr_dev && n > 0 && (momentum == 0.f || mom), "sgd_step: bad args");
  ProfScope ps(PC_SGD, 0, (double)n * 20, (hipStream_t)s);
  NUNET_LAUNCH(sgd_kernel, dim3(grid_for(n, 256 * 4, 2048)), dim3(256), 0, (hipStream_t)s, p, g, mom, n, lr_dev, momentum, weight_decay, nesterov, first, grad_scale);
  return nunet_check_launch("sgd_step");
}
