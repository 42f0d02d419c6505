// plan.hip — whole-network schedule of NestedUNet / UNet forward and backward
// (reference finished/archs1.py:35-71 UNet, :74-143 NestedUNet) over the per-op
// kernels, with a static arena layout:
//   * one NHWC buffer per pyramid level holding every x_{i,j} of that level as a
//     channel slot, so torch.cat (archs1.py:116-131) is zero-copy;
//   * packed KRSC weights (forward + flipped/transposed dgrad copies);
//   * native-layout fp32 gradient scratch that a single kernel unpacks into the
//     caller's flat OIHW gradient arena (reference parameters() order).
#include <stdlib.h>
#include <string.h>

#include <initializer_list>
#include <functional>

#include <vector>
#include <string>
#include <stdarg.h>

#include "common.h"

static const int NBF[5] = {32, 64, 128, 256, 512};  // archs1.py:78

// ---------------------------------------------------------------------------
// batched weight pack / gradient unpack
// ---------------------------------------------------------------------------
#define MAXENT 40
#define HEAD_SLABS 256
struct PackEnt { long long src, wf, wd; int cout, cin, cinpad, pad_; };
struct PackTab { int n; int ntiles; PackEnt e[MAXENT]; int tile0[MAXENT + 1]; };
struct UnpackEnt { long long src, dst; int cout, cin, cinpad, taps, nvec, nslab; unsigned inv_taps, inv_cin; int fast; };   // inv_*: dec_inv(), fast: 32-bit decode is exact   // nslab > 1: sum of partial slabs (heads)
struct UnpackTab { int n; int accumulate; UnpackEnt e[MAXENT]; };

// One block per (layer, 32 Cout x 32 Cin tile): the OIHW rows of a tile are contiguous runs of
// 32*9 floats (coalesced loads into LDS); both packed layouts are then written as contiguous
// 32-element rows. PackTab carries a prefix sum of tiles per entry for the block -> tile map.
template <typename T>
__global__ __launch_bounds__(256) void pack_kernel(const float* __restrict__ params, T* __restrict__ arena, PackTab tab) {
  __shared__ float s_t[32][32 * 9 + 1];
  int e = 0;
  while (e + 1 < tab.n && (int)blockIdx.x >= tab.tile0[e + 1]) ++e;
  const PackEnt en = tab.e[e];
  const int t = blockIdx.x - tab.tile0[e];
  const int nci = (en.cinpad + 31) / 32;
  const int co0 = (t / nci) * 32, ci0 = (t % nci) * 32;
  const float* w = params + en.src;
  const int cw = min(32, en.cin - ci0);            // real input channels in this tile (<= 0: pure padding)
  const int rw = min(32, en.cout - co0);
  // full tiles (all but the first layer's): 16-byte loads and 16-byte stores (8 packed elements); the scalar
  // form below keeps the ragged ones
  constexpr int EPV = Tr<T>::EPV;
  const float* wrow = w + ((long long)co0 * en.cin + ci0) * 9;
  const bool vload = rw == 32 && cw == 32 && en.cin % 4 == 0 && ((uintptr_t)wrow & 15) == 0;
  if (vload) {
    for (int i = threadIdx.x; i < 32 * 72; i += blockDim.x) {
      const int ro = i / 72, k4 = i - ro * 72;
      const f32x4 v = *reinterpret_cast<const f32x4*>(wrow + (long long)ro * en.cin * 9 + k4 * 4);
      float* d = &s_t[ro][k4 * 4];
      d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
    }
  } else {
    for (int i = threadIdx.x; i < 32 * 288; i += blockDim.x) {
      const int ro = i / 288, k = i - ro * 288;      // k = ci_local*9 + tap
      float v = 0.f;
      if (ro < rw && k < cw * 9) v = w[((long long)(co0 + ro) * en.cin + ci0) * 9 + k];
      s_t[ro][k] = v;
    }
  }
  __syncthreads();
  T* wf = arena + en.wf;
  T* wd = en.wd >= 0 ? arena + en.wd : nullptr;
  const bool vst = rw == 32 && en.cinpad % 8 == 0 && en.cout % 8 == 0 && ((uintptr_t)wf & 15) == 0 && ((uintptr_t)wd & 15) == 0;
  if (vst) {
    for (int i = threadIdx.x; i < 9 * 32 * 4; i += blockDim.x) {    // wf[tap][co][ci], ci fastest: 8 ci per thread
      const int g = i & 3, ro = (i >> 2) & 31, tap = i >> 7;
      if (ci0 + g * 8 < en.cinpad) {
        T* q = wf + ((long long)tap * en.cout + co0 + ro) * en.cinpad + ci0 + g * 8;
#pragma unroll
        for (int h = 0; h < 8 / EPV; ++h) {
          Vec16<T> o;
#pragma unroll
          for (int e = 0; e < EPV; ++e) o.set(e, s_t[ro][(g * 8 + h * EPV + e) * 9 + tap]);
          st16(q + h * EPV, o);
        }
      }
    }
    if (wd) {
      for (int i = threadIdx.x; i < 9 * 32 * 4; i += blockDim.x) {  // wd[8-tap][ci][co], co fastest: 8 co per thread
        const int g = i & 3, ci = (i >> 2) & 31, tap = i >> 7;
        if (ci < cw) {
          T* q = wd + ((long long)(8 - tap) * en.cin + ci0 + ci) * en.cout + co0 + g * 8;
#pragma unroll
          for (int h = 0; h < 8 / EPV; ++h) {
            Vec16<T> o;
#pragma unroll
            for (int e = 0; e < EPV; ++e) o.set(e, s_t[g * 8 + h * EPV + e][ci * 9 + tap]);
            st16(q + h * EPV, o);
          }
        }
      }
    }
    return;
  }
  for (int i = threadIdx.x; i < 9 * 32 * 32; i += blockDim.x) {   // wf[tap][co][ci], ci fastest
    const int ci = i & 31, ro = (i >> 5) & 31, tap = i >> 10;
    if (ro < rw && ci0 + ci < en.cinpad)
      wf[((long long)tap * en.cout + co0 + ro) * en.cinpad + ci0 + ci] = from_f32<T>(s_t[ro][ci * 9 + tap]);
  }
  if (wd) {
    for (int i = threadIdx.x; i < 9 * 32 * 32; i += blockDim.x) { // wd[8-tap][ci][co], co fastest
      const int ro = i & 31, ci = (i >> 5) & 31, tap = i >> 10;
      if (ro < rw && ci < cw)
        wd[((long long)(8 - tap) * en.cin + ci0 + ci) * en.cout + co0 + ro] = from_f32<T>(s_t[ro][ci * 9 + tap]);
    }
  }
}

__global__ __launch_bounds__(256) void unpack_kernel(const float* __restrict__ scratch, float* __restrict__ grads, UnpackTab tab) {
  const UnpackEnt en = tab.e[blockIdx.y];
  const float* dw = scratch + en.src;
  float* g = grads + en.dst;
  const long long nw = (long long)en.cout * en.cin * en.taps;
  const long long total = nw + (long long)en.nvec * en.cout;
  if (en.nslab > 1) {
    // 1x1 head: [nslab][cout*cin + cout] partial slabs (already OIHW order), summed by one block:
    // thread = slab, wave shuffles + LDS across the 4 waves, a handful of elements in total
    if (blockIdx.x != 0) return;
    __shared__ float s_p[256];
    const int tot = (int)total;                       // <= HEAD classes * 33 = 264
    for (int e0 = 0; e0 < tot; e0 += 256) {
      const int ne = min(256, tot - e0);
      const int parts = 256 / ne;                     // threads per element
      const int e = threadIdx.x % ne, part = threadIdx.x / ne;
      float v = 0.f;
      if (part < parts) {
#pragma unroll 8
        for (int sl = part; sl < en.nslab; sl += parts) v += dw[(long long)sl * total + e0 + e];
      }
      s_p[threadIdx.x] = part < parts ? v : 0.f;
      __syncthreads();
      if ((int)threadIdx.x < ne) {
        float t = 0.f;
        for (int q = 0; q < parts; ++q) t += s_p[q * ne + threadIdx.x];
        g[e0 + threadIdx.x] = tab.accumulate ? g[e0 + threadIdx.x] + t : t;
      }
      __syncthreads();
    }
    return;
  }
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    float v;
    if (i < nw) {
      int tap, ci, co;
      if (en.fast) {   // (two 64-bit divisions per element were most of this kernel's instructions)
        const int ii = (int)i, t = dec_div(ii, en.inv_taps);
        tap = ii - t * en.taps; co = dec_div(t, en.inv_cin); ci = t - co * en.cin;
      } else {
        tap = (int)(i % en.taps);
        const long long t = i / en.taps;
        ci = (int)(t % en.cin); co = (int)(t / en.cin);
      }
      v = dw[((long long)tap * en.cout + co) * en.cinpad + ci];
    } else {
      v = dw[(long long)en.taps * en.cout * en.cinpad + (i - nw)];
    }
    g[i] = tab.accumulate ? g[i] + v : v;
  }
}

// The same gather as unpack_kernel, tiled like pack_kernel: one block per (layer, 32 Cout x 32 Cin tile) moves the
// tile's [tap][co][ci] gradients through LDS with 16-byte loads along ci and writes the OIHW rows as 16-byte runs
// (unpack_kernel's element-wise gather touches nine 128-byte lines per wave load). Blocks past the last tile own one
// table entry each: its bias / BatchNorm vectors, or the slab sum of a 1x1 head.
__global__ __launch_bounds__(256) void unpack_tiled_kernel(const float* __restrict__ scratch, float* __restrict__ grads, PackTab tab, UnpackTab ut) {
  __shared__ float s_t[32][32 * 9 + 1];
  if ((int)blockIdx.x >= tab.ntiles) {
    const UnpackEnt en = ut.e[(int)blockIdx.x - tab.ntiles];
    const float* dw = scratch + en.src;
    float* g = grads + en.dst;
    const int nw = en.cout * en.cin * en.taps;
    if (en.nslab > 1) {                                  // 1x1 head: <= 264 elements, slabs summed in fixed order
      const int tot = nw + en.nvec * en.cout;
      for (int e0 = 0; e0 < tot; e0 += 256) {
        const int ne = min(256, tot - e0);
        const int parts = 256 / ne;
        const int e = threadIdx.x % ne, part = threadIdx.x / ne;
        float v = 0.f;
        if (part < parts) {
#pragma unroll 8
          for (int sl = part; sl < en.nslab; sl += parts) v += dw[(long long)sl * tot + e0 + e];
        }
        s_t[0][threadIdx.x] = part < parts ? v : 0.f;
        __syncthreads();
        if ((int)threadIdx.x < ne) {
          float t = 0.f;
          for (int q = 0; q < parts; ++q) t += s_t[0][q * ne + threadIdx.x];
          g[e0 + threadIdx.x] = ut.accumulate ? g[e0 + threadIdx.x] + t : t;
        }
        __syncthreads();
      }
      return;
    }
    const float* vsrc = dw + (long long)en.taps * en.cout * en.cinpad;
    for (int i = threadIdx.x; i < en.nvec * en.cout; i += blockDim.x) g[nw + i] = ut.accumulate ? g[nw + i] + vsrc[i] : vsrc[i];
    return;
  }
  int e = 0;
  while (e + 1 < tab.n && (int)blockIdx.x >= tab.tile0[e + 1]) ++e;
  const PackEnt en = tab.e[e];
  const UnpackEnt ue = ut.e[e];
  const int t = blockIdx.x - tab.tile0[e];
  const int nci = (en.cinpad + 31) / 32;
  const int co0 = (t / nci) * 32, ci0 = (t % nci) * 32;
  const int cw = min(32, en.cin - ci0), rw = min(32, en.cout - co0);
  if (cw <= 0) return;                                   // pure padding tile
  const float* dw = scratch + ue.src;
  float* grow = grads + ue.dst + ((long long)co0 * en.cin + ci0) * 9;
  const bool full = rw == 32 && cw == 32;
  if (full && en.cinpad % 4 == 0 && ((uintptr_t)dw & 15) == 0) {
    for (int i = threadIdx.x; i < 9 * 32 * 8; i += blockDim.x) {
      const int c4 = i & 7, ro = (i >> 3) & 31, tap = i >> 8;
      const f32x4 v = *reinterpret_cast<const f32x4*>(dw + ((long long)tap * en.cout + co0 + ro) * en.cinpad + ci0 + c4 * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) s_t[ro][(c4 * 4 + j) * 9 + tap] = v[j];
    }
  } else {
    for (int i = threadIdx.x; i < 9 * 32 * 32; i += blockDim.x) {
      const int ci = i & 31, ro = (i >> 5) & 31, tap = i >> 10;
      if (ro < rw && ci < cw) s_t[ro][ci * 9 + tap] = dw[((long long)tap * en.cout + co0 + ro) * en.cinpad + ci0 + ci];
    }
  }
  __syncthreads();
  if (full && en.cin % 4 == 0 && ((uintptr_t)grow & 15) == 0) {
    for (int i = threadIdx.x; i < 32 * 72; i += blockDim.x) {
      const int ro = i / 72, k4 = i - ro * 72;
      f32x4* q = reinterpret_cast<f32x4*>(grow + (long long)ro * en.cin * 9 + k4 * 4);
      const float* sp = &s_t[ro][k4 * 4];
      f32x4 v = {sp[0], sp[1], sp[2], sp[3]};
      if (ut.accumulate) { const f32x4 o = *q; v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3]; }
      *q = v;
    }
  } else {
    for (int i = threadIdx.x; i < 32 * 288; i += blockDim.x) {
      const int ro = i / 288, k = i - ro * 288;
      if (ro < rw && k < cw * 9) {
        float* q = grow + (long long)ro * en.cin * 9 + k;
        *q = ut.accumulate ? *q + s_t[ro][k] : s_t[ro][k];
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------
// Fused optimiser step: native-layout gradient scratch -> SGD(momentum, weight decay, nesterov) on the fp32 master
// parameters -> both packed 16-bit weight layouts of the NEXT forward/backward, one launch (replaces
// unpack_kernel + sgd_kernel + pack_kernel, which ran one after the other with nothing to overlap them).
// Same block -> (layer, 32 Cout x 32 Cin tile) map as pack_kernel; the tile's gradients arrive [tap][co][ci]
// (coalesced along ci) and meet the OIHW parameters / momentum through LDS. The block of input-channel tile 0
// also steps the layer's conv bias and BatchNorm gamma / beta; blocks past the last tile own the 1x1 heads
// (sum of their gradient slabs, then the same step). torch.optim.SGD semantics (reference trains.py:229-231).
// ---------------------------------------------------------------------------------------------------------
struct UpdP { float* params; float* mom; const float* scratch; float* grads; const float* lr; float momc, wd, gscale; int nesterov; int nconv; };

__device__ __forceinline__ float sgd_one(float p, float g, float* m, const UpdP& u, float lr) {
  float gv = g + u.wd * p;
  if (u.momc != 0.f) {
    const float b = u.momc * (*m) + gv;
    *m = b;
    gv = u.nesterov ? gv + u.momc * b : b;
  }
  return p - lr * gv;
}

template <typename T>
__global__ __launch_bounds__(512) void update_kernel(UpdP u, T* __restrict__ arena, PackTab tab, UnpackTab ut) {
  __shared__ float s_t[32][32 * 9 + 1];
  __shared__ float s_g[32][32 * 9 + 1];
  const float lr = u.lr[0];
  if ((int)blockIdx.x >= tab.ntiles) {
    // ---- 1x1 head: sum the gradient slabs (fixed order), then the step; <= 264 elements
    const UnpackEnt en = ut.e[u.nconv + (int)blockIdx.x - tab.ntiles];
    const float* dw = u.scratch + en.src;
    const int tot = en.cout * en.cin * en.taps + en.nvec * en.cout;
    const int B = blockDim.x;     // B/ne threads share an element's slabs (see unpack_sgd_tiled_kernel), fixed order
    for (int e0 = 0; e0 < tot; e0 += B) {
      const int ne = min(B, tot - e0);
      const int parts = B / ne;
      const int el = threadIdx.x % ne, part = threadIdx.x / ne;
      float v = 0.f;
      if (part < parts) {
#pragma unroll 8
        for (int sl = part; sl < en.nslab; sl += parts) v += dw[(long long)sl * tot + e0 + el];
      }
      (&s_t[0][0])[threadIdx.x] = part < parts ? v : 0.f;
      __syncthreads();
      if ((int)threadIdx.x < ne) {
        float g = 0.f;
        for (int q = 0; q < parts; ++q) g += (&s_t[0][0])[q * ne + threadIdx.x];
        g *= u.gscale;
        const long long idx = en.dst + e0 + threadIdx.x;
        if (u.grads) u.grads[idx] = g;
        float m = u.mom[idx];
        const float pn = sgd_one(u.params[idx], g, &m, u, lr);
        u.mom[idx] = m; u.params[idx] = pn;
      }
      __syncthreads();
    }
    return;
  }
  int e = 0;
  while (e + 1 < tab.n && (int)blockIdx.x >= tab.tile0[e + 1]) ++e;
  const PackEnt en = tab.e[e];
  const UnpackEnt ue = ut.e[e];
  const int t = blockIdx.x - tab.tile0[e];
  const int nci = (en.cinpad + 31) / 32;
  const int co0 = (t / nci) * 32, ci0 = (t % nci) * 32;
  const float* w = u.params + en.src;
  const float* dw = u.scratch + ue.src;
  const int cw = min(32, en.cin - ci0);            // real input channels in this tile (<= 0: pure padding)
  const int rw = min(32, en.cout - co0);
#pragma unroll 6
  for (int i = threadIdx.x; i < 32 * 288; i += blockDim.x) {
    const int ro = i / 288, k = i - ro * 288;      // k = ci_local*9 + tap
    float v = 0.f;
    if (ro < rw && k < cw * 9) v = w[((long long)(co0 + ro) * en.cin + ci0) * 9 + k];
    s_t[ro][k] = v;
  }
#pragma unroll 6
  for (int i = threadIdx.x; i < 9 * 32 * 32; i += blockDim.x) {   // scratch dw[tap][co][cinpad], ci fastest
    const int ci = i & 31, ro = (i >> 5) & 31, tap = i >> 10;
    float v = 0.f;
    if (ro < rw && ci < cw) v = dw[((long long)tap * en.cout + co0 + ro) * en.cinpad + ci0 + ci];
    s_g[ro][ci * 9 + tap] = v;
  }
  __syncthreads();
#pragma unroll 6
  for (int i = threadIdx.x; i < 32 * 288; i += blockDim.x) {
    const int ro = i / 288, k = i - ro * 288;
    if (ro < rw && k < cw * 9) {
      const long long idx = en.src + ((long long)(co0 + ro) * en.cin + ci0) * 9 + k;
      const float g = s_g[ro][k] * u.gscale;
      if (u.grads) u.grads[idx] = g;
      float m = u.mom[idx];
      const float pn = sgd_one(s_t[ro][k], g, &m, u, lr);
      u.mom[idx] = m; u.params[idx] = pn;
      s_t[ro][k] = pn;
    }
  }
  if (ci0 == 0) {
    // conv bias, BN gamma, BN beta of the output channels of this tile (they follow the weights in both arenas)
    const long long nw = (long long)en.cout * en.cin * 9;
    for (int i = threadIdx.x; i < ue.nvec * rw; i += blockDim.x) {
      const int v = i / rw, c = co0 + (i - v * rw);
      const long long idx = ue.dst + nw + (long long)v * en.cout + c;
      const float g = dw[9LL * en.cout * en.cinpad + (long long)v * en.cout + c] * u.gscale;
      if (u.grads) u.grads[idx] = g;
      float m = u.mom[idx];
      const float pn = sgd_one(u.params[idx], g, &m, u, lr);
      u.mom[idx] = m; u.params[idx] = pn;
    }
  }
  __syncthreads();
  T* wf = arena + en.wf;
#pragma unroll 6
  for (int i = threadIdx.x; i < 9 * 32 * 32; i += blockDim.x) {   // wf[tap][co][ci], ci fastest
    const int ci = i & 31, ro = (i >> 5) & 31, tap = i >> 10;
    if (ro < rw && ci0 + ci < en.cinpad)
      wf[((long long)tap * en.cout + co0 + ro) * en.cinpad + ci0 + ci] = from_f32<T>(s_t[ro][ci * 9 + tap]);
  }
  if (en.wd >= 0) {
    T* wd = arena + en.wd;
#pragma unroll 6
    for (int i = threadIdx.x; i < 9 * 32 * 32; i += blockDim.x) { // wd[8-tap][ci][co], co fastest
      const int ro = i & 31, ci = (i >> 5) & 31, tap = i >> 10;
      if (ro < rw && ci < cw)
        wd[((long long)(8 - tap) * en.cin + ci0 + ci) * en.cout + co0 + ro] = from_f32<T>(s_t[ro][ci * 9 + tap]);
    }
  }
}

// unpack_kernel with the optimiser step as its epilogue: the gradient is gathered from the native-layout scratch and
// consumed on the spot (no OIHW gradient round trip, one launch less); `u.grads` optionally still receives it.
__global__ __launch_bounds__(256) void unpack_sgd_kernel(UpdP u, UnpackTab tab) {
  const UnpackEnt en = tab.e[blockIdx.y];
  const float* dw = u.scratch + en.src;
  const float lr = u.lr[0];
  const long long nw = (long long)en.cout * en.cin * en.taps;
  const long long total = nw + (long long)en.nvec * en.cout;
  if (en.nslab > 1) {
    if (blockIdx.x != 0) return;
    for (int e = threadIdx.x; e < (int)total; e += blockDim.x) {   // 1x1 head: slabs summed in fixed order
      float g = 0.f;
      for (int sl = 0; sl < en.nslab; ++sl) g += dw[(long long)sl * total + e];
      g *= u.gscale;
      const long long idx = en.dst + e;
      if (u.grads) u.grads[idx] = g;
      float m = u.mom[idx];
      const float pn = sgd_one(u.params[idx], g, &m, u, lr);
      u.mom[idx] = m; u.params[idx] = pn;
    }
    return;
  }
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    float g;
    if (i < nw) {
      int tap, ci, co;
      if (en.fast) {
        const int ii = (int)i, t = dec_div(ii, en.inv_taps);
        tap = ii - t * en.taps; co = dec_div(t, en.inv_cin); ci = t - co * en.cin;
      } else {
        tap = (int)(i % en.taps);
        const long long t = i / en.taps;
        ci = (int)(t % en.cin); co = (int)(t / en.cin);
      }
      g = dw[((long long)tap * en.cout + co) * en.cinpad + ci];
    } else {
      g = dw[(long long)en.taps * en.cout * en.cinpad + (i - nw)];
    }
    g *= u.gscale;
    const long long idx = en.dst + i;
    if (u.grads) u.grads[idx] = g;
    float m = u.mom[idx];
    const float pn = sgd_one(u.params[idx], g, &m, u, lr);
    u.mom[idx] = m; u.params[idx] = pn;
  }
}

template <typename T> static int launch_pack(const float* params, void* arena_t, PackTab& tab, long long maxn, hipStream_t st) {
  (void)maxn;
  int nt = 0;
  for (int i = 0; i < tab.n; ++i) { tab.tile0[i] = nt; nt += ((tab.e[i].cout + 31) / 32) * ((tab.e[i].cinpad + 31) / 32); }
  tab.tile0[tab.n] = nt; tab.ntiles = nt;
  double pb = 0;
  for (int i = 0; i < tab.n; ++i) pb += 9.0 * tab.e[i].cout * tab.e[i].cin * (4 + (tab.e[i].wd >= 0 ? 2 : 1) * sizeof(T));
  ProfScope ps(PC_PACK, 0, pb, st);
  hipLaunchKernelGGL((pack_kernel<T>), dim3(nt), dim3(256), 0, st, params, (T*)arena_t, tab);
  return nunet_check_launch("pack_weights");
}

extern "C" int nunet_pack_weights(const float* w, int32_t cout, int32_t cin, int32_t cin_pad, int32_t dtype, void* wf, void* wd, nunet_stream_t s) {
  NUNET_REQUIRE(w && wf && cout > 0 && cin > 0 && cin_pad >= cin, "pack_weights: bad args");
  NUNET_REQUIRE(dtype >= 0 && dtype <= 2, "pack_weights: bad dtype");
  PackTab tab; memset(&tab, 0, sizeof(tab));
  tab.n = 1;
  const int es = dtype_size(dtype);
  char* base = (wd && (char*)wd < (char*)wf) ? (char*)wd : (char*)wf;
  NUNET_REQUIRE(((char*)wf - base) % es == 0 && (!wd || ((char*)wd - base) % es == 0), "pack_weights: wf/wd misaligned");
  tab.e[0].src = 0;
  tab.e[0].wf = ((char*)wf - base) / es;
  tab.e[0].wd = wd ? ((char*)wd - base) / es : -1;
  tab.e[0].cout = cout; tab.e[0].cin = cin; tab.e[0].cinpad = cin_pad;
  return NUNET_DISPATCH(dtype, launch_pack, w, (void*)base, tab, 9LL * cout * cin_pad, (hipStream_t)s);
}

extern "C" int nunet_unpack_wgrad(const float* dw, int32_t cout, int32_t cin, int32_t cin_pad, float* g, int32_t accumulate, nunet_stream_t s) {
  NUNET_REQUIRE(dw && g && cout > 0 && cin > 0 && cin_pad >= cin, "unpack_wgrad: bad args");
  UnpackTab tab; memset(&tab, 0, sizeof(tab));
  tab.n = 1; tab.accumulate = accumulate;
  tab.e[0].src = 0; tab.e[0].dst = 0; tab.e[0].cout = cout; tab.e[0].cin = cin; tab.e[0].cinpad = cin_pad; tab.e[0].taps = 9; tab.e[0].nvec = 0; tab.e[0].nslab = 1;
  tab.e[0].inv_taps = dec_inv(9); tab.e[0].inv_cin = dec_inv(cin); tab.e[0].fast = 9LL * cout * cin * (cin > 9 ? cin : 9) < (1ll << 32);
  int gx = (int)ceil_div64(9LL * cout * cin, 256 * 4);
  if (gx > 512) gx = 512;
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL(unpack_kernel, dim3(gx, 1), dim3(256), 0, (hipStream_t)s, dw, g, tab);
  return nunet_check_launch("unpack_wgrad");
}

// ---------------------------------------------------------------------------
// plan
// ---------------------------------------------------------------------------
struct ConvL {
  int cin, cinpad, cout;
  long long w_off, b_off, g_off, be_off;  // floats into flat params
  long long rm_off, rv_off; int bn_index; // floats into bnbuf / index into nbt
  long long wf, wd;                       // elements of T into the packed-weight region (wd -1: none)
  long long gs;                           // floats into grad scratch: [dw 9*cout*cinpad][db][dgamma][dbeta]
  long long stats, save, bsum;            // floats into the fp32 small-vector regions
};
struct Node {
  int i, j;
  int in_prefix;   // number of level slots concatenated in front (0: encoder input)
  int out_slot;    // slot of X_i receiving the block output
  int up_slot;     // slot of X_{i+1} that is upsampled into the concat, -1: none
  ConvL c1, c2;
  size_t y1, a1, y2, up, pin;  // arena byte offsets
};
struct Head { long long w_off, b_off, gs; int slot; };

#define NLANES 10
struct PlanRt {  // runtime objects owned by the plan (host side only)
  hipStream_t lanes[NLANES];
  bool lanes_ok;
  std::vector<hipEvent_t> events[2];   // [0] forward, [1] backward: an event is never re-recorded within one capture
  size_t events_used[2];
  int multistream;
  bool lanes_external;
  std::vector<hipStream_t> cap_streams;   // never-reused streams for capture-time lane continuation
  size_t cap_next;
  void* gs_clean_arena;
  void* sk_ready_arena;                    // arena whose K-split arrival counters have been zeroed
  bool bwd_written[5][5]; int bwd_pp[5];   // state carried between backward phases   // arena whose gradient scratch was cleared by the last training forward
  // NUNET_STAMPS=1 diagnostic: a 1-thread kernel after every scheduled op writes the 100 MHz wall clock,
  // so the real timeline of an (unprofiled) hipGraph replay can be read back (tools/stamp_timeline.py)
  unsigned long long* stamps;              // device, [2][STAMP_CAP]
  hipEvent_t b0_event;                     // recorded when the first gradient bucket (phase-1 nodes + heads) is complete
  bool b0_enabled;
  std::vector<std::string> stamp_labels[2];
};
void graph_tag_tail(hipStream_t st, int lane);   // graph.hip: lane bookkeeping of an active nunet_graph capture
int graph_record_external(hipStream_t st, hipEvent_t ev);   // graph.hip: event record node at the tail of a capturing stream
#define STAMP_CAP 512
__global__ void stamp_kernel(unsigned long long* p) { *p = wall_clock64(); }
struct nunet_plan;
static PlanRt* rt_of(nunet_plan* P);
struct nunet_plan {
  nunet_plan_cfg cfg;
  int es;                       // element size of T
  int hl[5], wl[5]; long long px[5];
  int nslots[5], PX[5];
  std::vector<Node> exec;       // execution order
  std::vector<int> reg;         // registration (parameter) order -> index into exec
  std::vector<Head> heads;
  long long nparams, nbnbuf; int nbn;
  int first_phase_nodes; long long gs_bucket0;   // backward phase 1 = heads + this many last nodes; its gradient-scratch prefix
  // arena regions (byte offsets)
  size_t off_stats, stats_floats;
  size_t off_gs, gs_floats;     // grad scratch + bn-bwd sums (zeroed every backward)
  size_t off_save;
  size_t off_wpack; long long wpack_elems;
  size_t off_img;
  size_t X[5], GX[5];
  size_t off_dy[16][2], off_da1[16], off_gup[16], off_gpin[16];   // per-BLOCK backward scratch (dY ping-pong), so blocks of one level can run on different lanes
  size_t off_sk[5]; long long sk_floats[5];   // per-level fp32 K-split slabs (levels 3, 4)   // per-level backward scratch (dY ping-pong)
  struct PlanRt* rt;
  size_t total;
  PackTab ptab; long long pack_maxn;
  PackTab ptab_lvl[5]; long long pack_maxn_lvl[5];   // the same entries grouped by pyramid level (issued per lane)
  UnpackTab utab; long long unpack_maxn;
};

static size_t bump(size_t& cur, size_t bytes) {
  size_t o = align_up(cur, 256);
  cur = o + bytes;
  return o;
}

extern "C" nunet_plan* nunet_plan_create(const nunet_plan_cfg* cfg) {
  if (!cfg) { nunet_set_error("plan_create: null cfg"); return nullptr; }
  if (cfg->N <= 0 || cfg->H <= 0 || cfg->W <= 0 || cfg->H % 16 || cfg->W % 16) {
    nunet_set_error("plan_create: H=%d W=%d must be positive multiples of 16 (four 2x2 pools, archs1.py:114-131)", cfg->H, cfg->W);
    return nullptr;
  }
  if (cfg->input_channels < 1 || cfg->input_channels > 32) { nunet_set_error("plan_create: input_channels=%d unsupported (1..32)", cfg->input_channels); return nullptr; }
  if (cfg->num_classes < 1 || cfg->num_classes > 8) { nunet_set_error("plan_create: num_classes=%d unsupported (1..8)", cfg->num_classes); return nullptr; }
  if (cfg->dtype < 0 || cfg->dtype > 2) { nunet_set_error("plan_create: bad dtype"); return nullptr; }
  nunet_plan* P = new nunet_plan();
  P->cfg = *cfg;
  P->es = dtype_size(cfg->dtype);
  const bool unet = cfg->unet != 0;
  for (int i = 0; i < 5; ++i) {
    P->hl[i] = cfg->H >> i; P->wl[i] = cfg->W >> i;
    P->px[i] = (long long)cfg->N * P->hl[i] * P->wl[i];
    P->nslots[i] = unet ? (i == 4 ? 1 : 2) : 5 - i;
    P->PX[i] = P->nslots[i] * NBF[i];
  }
  // ---- nodes in execution order ------------------------------------------------
  if (!unet) {
    for (int s = 0; s < 5; ++s)
      for (int j = 0; j <= s; ++j) {
        Node n; memset(&n, 0, sizeof(n));
        n.i = s - j; n.j = j; n.in_prefix = j; n.out_slot = j; n.up_slot = j > 0 ? j - 1 : -1;
        P->exec.push_back(n);
      }
  } else {
    for (int i = 0; i < 5; ++i) { Node n; memset(&n, 0, sizeof(n)); n.i = i; n.j = 0; n.in_prefix = 0; n.out_slot = 0; n.up_slot = -1; P->exec.push_back(n); }
    for (int i = 3; i >= 0; --i) {
      Node n; memset(&n, 0, sizeof(n));
      n.i = i; n.j = 4 - i; n.in_prefix = 1; n.out_slot = 1; n.up_slot = (i + 1 == 4) ? 0 : 1;
      P->exec.push_back(n);
    }
  }
  // registration order: by column j, then row i (archs1.py:45-56 / 85-103)
  for (int j = 0; j < 5; ++j)
    for (int i = 0; i < 5; ++i)
      for (size_t k = 0; k < P->exec.size(); ++k)
        if (P->exec[k].i == i && P->exec[k].j == j) P->reg.push_back((int)k);

  // ---- parameter / buffer offsets in registration order -------------------------
  long long po = 0, bo = 0, gs = 0, sv = 0; int bn = 0;
  long long wp = 0;
  auto setup_conv = [&](ConvL& c, int cin, int cout, bool need_wd) {
    c.cin = cin; c.cout = cout; c.cinpad = cin < 32 ? 32 : cin;
    c.w_off = po; po += (long long)cout * cin * 9;
    c.b_off = po; po += cout;
    c.g_off = po; po += cout;
    c.be_off = po; po += cout;
    c.rm_off = bo; bo += cout; c.rv_off = bo; bo += cout; c.bn_index = bn++;
    c.wf = wp; wp += 9LL * cout * c.cinpad; wp = (wp + 127) / 128 * 128;
    if (need_wd) { c.wd = wp; wp += 9LL * cout * cin; wp = (wp + 127) / 128 * 128; } else c.wd = -1;
    c.gs = -1;   // assigned below, in gradient-ready order
    c.stats = sv; c.save = sv; c.bsum = sv; sv += 2LL * cout;
  };
  for (size_t r = 0; r < P->reg.size(); ++r) {
    Node& n = P->exec[P->reg[r]];
    const int f = NBF[n.i];
    int cin;
    if (n.in_prefix == 0) cin = n.i == 0 ? cfg->input_channels : NBF[n.i - 1];
    else cin = n.in_prefix * f + NBF[n.i + 1];
    setup_conv(n.c1, cin, f, !(n.i == 0 && n.j == 0));
    setup_conv(n.c2, f, f, true);
  }
  const int nheads = (cfg->deep_supervision && !unet) ? 4 : 1;
  for (int k = 0; k < nheads; ++k) {
    Head h;
    h.w_off = po; po += (long long)cfg->num_classes * NBF[0];
    h.b_off = po; po += cfg->num_classes;
    h.gs = -1;
    h.slot = unet ? 1 : (nheads == 4 ? k + 1 : 4);
    P->heads.push_back(h);
  }
  // Gradient scratch in the order gradients COMPLETE during backward (heads, then blocks in reverse
  // execution order): the first bucket = heads + the last anti-diagonal (75 % of the bytes,
  // SURVEY.md §3.4) is a contiguous prefix, ready for an early all-reduce.
  for (size_t k = 0; k < P->heads.size(); ++k) {
    P->heads[k].gs = gs; gs += (long long)HEAD_SLABS * ((long long)cfg->num_classes * NBF[0] + cfg->num_classes);
  }
  P->first_phase_nodes = unet ? 4 : 5;
  P->gs_bucket0 = 0;
  for (int k = (int)P->exec.size() - 1; k >= 0; --k) {
    Node& n = P->exec[k];
    for (int cv = 1; cv >= 0; --cv) {
      ConvL& c = cv ? n.c2 : n.c1;
      c.gs = gs; gs += 9LL * c.cout * c.cinpad + 3LL * c.cout;
    }
    if (k == (int)P->exec.size() - P->first_phase_nodes) P->gs_bucket0 = gs;
  }
  P->nparams = po; P->nbnbuf = bo; P->nbn = bn;
  P->wpack_elems = wp;

  // ---- arena ---------------------------------------------------------------------
  size_t cur = 0;
  P->stats_floats = (size_t)sv;
  P->off_stats = bump(cur, P->stats_floats * 4 * NUNET_BN_SUM_REPLICAS);   // replicated per-channel sums
  P->gs_floats = (size_t)gs + (size_t)sv * NUNET_BN_SUM_REPLICAS;  // grad scratch followed by the (replicated) bn-bwd sums
  P->off_gs = bump(cur, P->gs_floats * 4);
  P->off_save = bump(cur, (size_t)sv * 4);
  P->off_wpack = bump(cur, (size_t)wp * P->es);
  P->off_img = bump(cur, (size_t)P->px[0] * 32 * P->es);
  for (int i = 0; i < 5; ++i) {
    P->X[i] = bump(cur, (size_t)P->px[i] * P->PX[i] * P->es);
    P->GX[i] = bump(cur, (size_t)P->px[i] * P->PX[i] * P->es);
  }
  for (size_t k = 0; k < P->exec.size(); ++k) {
    Node& n = P->exec[k];
    const int f = NBF[n.i];
    const size_t plane = (size_t)P->px[n.i] * f * P->es;
    n.y1 = bump(cur, plane); n.a1 = bump(cur, plane); n.y2 = bump(cur, plane);
    if (n.up_slot >= 0) n.up = bump(cur, (size_t)P->px[n.i] * NBF[n.i + 1] * P->es);
    if (n.in_prefix == 0 && n.i > 0) n.pin = bump(cur, (size_t)P->px[n.i] * NBF[n.i - 1] * P->es);
  }
  for (size_t k = 0; k < P->exec.size(); ++k) {
    const int i = P->exec[k].i;
    const size_t plane = (size_t)P->px[i] * NBF[i] * P->es;
    P->off_dy[k][0] = bump(cur, plane); P->off_dy[k][1] = bump(cur, plane);
    P->off_da1[k] = bump(cur, plane);
    P->off_gup[k] = bump(cur, i < 4 ? (size_t)P->px[i] * NBF[i + 1] * P->es : 256);
    P->off_gpin[k] = bump(cur, i > 0 ? (size_t)P->px[i] * NBF[i - 1] * P->es : 256);
  }
  // K-split slabs for the grid-starved levels 3 and 4 (up to 8 slices of the widest output at the
  // level); deterministic (fixed summation order). Measured +1.2 % on the bench; NUNET_SPLITK=0 disables.
  const char* ske = getenv("NUNET_SPLITK");
  const bool sk_on = !ske || atoi(ske) != 0;
  for (int i = 0; i < 5; ++i) {
    const int maxc = (i < 3 || !sk_on) ? 0 : (i < 4 ? (4 - i) * NBF[i] + NBF[i + 1] : NBF[4]);
    P->sk_floats[i] = 8LL * P->px[i] * maxc + NUNET_SPLITK_COUNTER_FLOATS;   // arrival counters + up to 8 slabs
    P->off_sk[i] = bump(cur, (size_t)P->sk_floats[i] * 4 + 16);
  }
  P->total = align_up(cur, 256);

  // ---- pack / unpack tables ------------------------------------------------------
  memset(&P->ptab, 0, sizeof(P->ptab)); memset(&P->utab, 0, sizeof(P->utab));
  P->pack_maxn = 0; P->unpack_maxn = 0;
  auto add_conv = [&](const ConvL& c) {
    PackEnt& pe = P->ptab.e[P->ptab.n++];
    pe.src = c.w_off; pe.wf = c.wf; pe.wd = c.wd; pe.cout = c.cout; pe.cin = c.cin; pe.cinpad = c.cinpad;
    if (9LL * c.cout * c.cinpad > P->pack_maxn) P->pack_maxn = 9LL * c.cout * c.cinpad;
    UnpackEnt& ue = P->utab.e[P->utab.n++];
    ue.src = c.gs; ue.dst = c.w_off; ue.cout = c.cout; ue.cin = c.cin; ue.cinpad = c.cinpad; ue.taps = 9; ue.nvec = 3; ue.nslab = 1;
    ue.inv_taps = dec_inv(9); ue.inv_cin = dec_inv(c.cin); ue.fast = 9LL * c.cout * c.cin * (c.cin > 9 ? c.cin : 9) < (1ll << 32);
    if (9LL * c.cout * c.cin + 3 * c.cout > P->unpack_maxn) P->unpack_maxn = 9LL * c.cout * c.cin + 3 * c.cout;
  };
  for (size_t r = 0; r < P->reg.size(); ++r) { add_conv(P->exec[P->reg[r]].c1); add_conv(P->exec[P->reg[r]].c2); }
  for (int l = 0; l < 5; ++l) { memset(&P->ptab_lvl[l], 0, sizeof(PackTab)); P->pack_maxn_lvl[l] = 0; }
  for (size_t k = 0; k < P->exec.size(); ++k) {
    const Node& n = P->exec[k];
    for (int cv = 0; cv < 2; ++cv) {
      const ConvL& c = cv ? n.c2 : n.c1;
      PackTab& t = P->ptab_lvl[n.i];
      PackEnt& pe = t.e[t.n++];
      pe.src = c.w_off; pe.wf = c.wf; pe.wd = c.wd; pe.cout = c.cout; pe.cin = c.cin; pe.cinpad = c.cinpad;
      if (9LL * c.cout * c.cinpad > P->pack_maxn_lvl[n.i]) P->pack_maxn_lvl[n.i] = 9LL * c.cout * c.cinpad;
    }
  }
  for (size_t k = 0; k < P->heads.size(); ++k) {
    UnpackEnt& ue = P->utab.e[P->utab.n++];
    ue.src = P->heads[k].gs; ue.dst = P->heads[k].w_off; ue.cout = cfg->num_classes; ue.cin = NBF[0]; ue.cinpad = NBF[0]; ue.taps = 1; ue.nvec = 1; ue.nslab = HEAD_SLABS; ue.inv_taps = 0; ue.inv_cin = dec_inv(NBF[0]); ue.fast = 1;
  }
  PlanRt* rt = new PlanRt();
  rt->lanes_ok = true;
  rt->lanes_external = false;
  rt->cap_next = 0;
  rt->gs_clean_arena = nullptr;
  rt->sk_ready_arena = nullptr;
  rt->stamps = nullptr;
  rt->b0_event = nullptr; rt->b0_enabled = false;
  rt->events_used[0] = rt->events_used[1] = 0;
  { const char* e = getenv("NUNET_MULTISTREAM"); rt->multistream = e ? atoi(e) : 1; }
  for (int l = 0; l < NLANES; ++l) {
    rt->lanes[l] = nullptr;
    if (hipStreamCreateWithFlags(&rt->lanes[l], hipStreamNonBlocking) != hipSuccess) { rt->lanes_ok = false; (void)hipGetLastError(); }
  }
  P->rt = rt;
  return P;
}


static PlanRt* rt_of(nunet_plan* P) { return P->rt; }

extern "C" void nunet_plan_destroy(nunet_plan* p) {
  if (!p) return;
  if (p->rt) {
    if (!p->rt->lanes_external)
      for (int l = 0; l < NLANES; ++l) if (p->rt->lanes[l]) (void)hipStreamDestroy(p->rt->lanes[l]);
    for (size_t k = 0; k < p->rt->cap_streams.size(); ++k) (void)hipStreamDestroy(p->rt->cap_streams[k]);
    for (int q = 0; q < 2; ++q)
      for (size_t k = 0; k < p->rt->events[q].size(); ++k) (void)hipEventDestroy(p->rt->events[q][k]);
    if (p->rt->stamps) (void)hipFree(p->rt->stamps);
    delete p->rt;
  }
  delete p;
}
extern "C" size_t nunet_plan_arena_bytes(const nunet_plan* p) { return p ? p->total : 0; }
extern "C" int64_t nunet_plan_param_count(const nunet_plan* p) { return p ? p->nparams : 0; }
extern "C" int64_t nunet_plan_bnbuf_count(const nunet_plan* p) { return p ? p->nbnbuf : 0; }
extern "C" int32_t nunet_plan_bn_layers(const nunet_plan* p) { return p ? p->nbn : 0; }
extern "C" int32_t nunet_plan_num_heads(const nunet_plan* p) { return p ? (int32_t)p->heads.size() : 0; }
extern "C" int64_t nunet_plan_feature(const nunet_plan* p, int32_t i, int32_t j, int32_t* pitch, int32_t* channels) {
  if (!p || i < 0 || i > 4) return -1;
  for (size_t k = 0; k < p->exec.size(); ++k)
    if (p->exec[k].i == i && p->exec[k].j == j) {
      if (pitch) *pitch = p->PX[i];
      if (channels) *channels = NBF[i];
      return (int64_t)(p->X[i] + (size_t)p->exec[k].out_slot * NBF[i] * p->es);
    }
  return -1;
}

#define CK(expr)                \
  do {                          \
    int rc_ = (expr);           \
    if (rc_ != NUNET_OK) return rc_; \
  } while (0)

static inline char* AB(void* arena, size_t off) { return (char*)arena + off; }

// ---------------------------------------------------------------------------
// Lane scheduler. The x_{i,j} grid has natural concurrency: blocks of level i depend only
// on levels i and i+-1, and weight gradients depend on nothing downstream. Every kernel
// here is latency-bound on its own (short contractions, grid-starved deep levels), so the
// plan issues level i on stream "lane i" (wgrad of level i on lane 5+i), forked from and
// joined to the caller's stream with events. Exact RAW / WAW / WAR dependencies come from a
// per-buffer tracker (last-writer event, last-reader event per lane). Captured by the
// caller, the lanes become parallel branches of ONE hipGraph.
// ---------------------------------------------------------------------------
#define NRES 480
enum { R_X = 0, R_GX = 25, R_BLK = 50, R_LVL = 330, R_IMG = 230, R_LOGITS = 231, R_DLOGITS = 232, R_WP = 233, R_GS = 238, R_SK = 470, R_GSW = 240, R_GSV = 280 };
enum { B_Y1 = 0, B_A1, B_Y2, B_UP, B_PIN, B_ST1, B_ST2, B_STRIDE = 8 };
enum { L_DY0 = 0, L_DY1, L_DA1, L_GUP, L_GPIN, L_STRIDE = 8 };

struct Sched {
  hipStream_t main_s;
  hipStream_t lane_s[NLANES];      // stream currently carrying each lane (fixed lanes when not capturing)
  hipEvent_t lane_tail[NLANES];    // event after the last op issued on the lane
  bool multi, capturing, failed;
  bool used[NLANES];
  PlanRt* rt;
  std::vector<hipEvent_t>* pool;
  size_t* pool_used;
  struct Res { hipEvent_t w_ev; hipStream_t w_st; hipEvent_t r_ev[NLANES]; hipStream_t r_st[NLANES]; };
  Res res[NRES];
  int cur_lane; int nreads, nwrites; int reads[16], writes[16];
  hipEvent_t fork_ev;
  hipEvent_t pend[64]; int npend;
  int lane_map[NLANES];
  int pass;
  char cur_name[32];
  void name(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(cur_name, sizeof(cur_name), fmt, ap); va_end(ap);
  }
  void stamp(hipStream_t st, int lane) {
    if (!rt->stamps) return;
    std::vector<std::string>& lab = rt->stamp_labels[pass];
    if ((int)lab.size() >= STAMP_CAP) return;
    hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, st, rt->stamps + (size_t)pass * STAMP_CAP + lab.size());
    char b[48]; snprintf(b, sizeof(b), "L%d %s", lane, cur_name);
    lab.push_back(b);
  }

  hipEvent_t new_event() {
    if (*pool_used == pool->size()) {
      hipEvent_t e;
      (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
      pool->push_back(e);
    }
    return (*pool)[(*pool_used)++];
  }
  void init(nunet_plan* P, hipStream_t s, int pass);
  // One wait per distinct event, never on an event of the stream itself.
  void want(hipEvent_t ev, hipStream_t ev_st, hipStream_t st) {
    if (!ev || ev_st == st) return;
    for (int k = 0; k < npend; ++k) if (pend[k] == ev) return;
    if (npend < 64) pend[npend++] = ev;
  }
  // ROCm 7.2: inside a stream capture, a stream that waits on an event DESCENDING from its
  // own tail node crashes hipStreamEndCapture (tools/capture_patterns3.py "pp1"). The x_{i,j}
  // grid ping-pongs between levels all the time, so while capturing a lane with cross-lane
  // waits continues on a never-used stream that waits on the lane's tail event AND the
  // cross-lane events (no own tail -> plain fork semantics). Eager issue keeps fixed lanes.
  hipStream_t fresh_stream() {
    if (rt->cap_next >= rt->cap_streams.size()) { failed = true; return nullptr; }
    return rt->cap_streams[rt->cap_next++];
  }
  // begin an op on `lane` reading `rd` and writing `wr` resources; returns the stream to launch on
  hipStream_t begin(int lane, std::initializer_list<int> rd, std::initializer_list<int> wr) {
    int r[16], w[16], nr = 0, nw = 0;
    for (int x : rd) if (x >= 0 && nr < 16) r[nr++] = x;
    for (int x : wr) if (x >= 0 && nw < 16) w[nw++] = x;
    return begin_v(lane_map[lane], r, nr, w, nw);
  }
  hipStream_t begin_v(int lane, const int* rd, int nrd, const int* wr, int nwr) {   // `lane` already mapped
    if (!multi) return main_s;
    cur_lane = lane; nreads = 0; nwrites = 0; npend = 0;
    hipStream_t st = lane_s[lane];      // may be null while capturing (lane not started yet)
    for (int q = 0; q < nrd; ++q) {
      const int r = rd[q];
      reads[nreads++] = r;
      Res& R = res[r];
      want(R.w_ev, R.w_st, st);
    }
    for (int q = 0; q < nwr; ++q) {
      const int w = wr[q];
      writes[nwrites++] = w;
      Res& R = res[w];
      want(R.w_ev, R.w_st, st);
      for (int l = 0; l < NLANES; ++l) want(R.r_ev[l], R.r_st[l], st);
    }
    if (capturing) {
      if (!used[lane] || npend > 0) {
        hipStream_t ns = fresh_stream();
        if (!ns) return main_s;   // pool exhausted: flagged, caller gets an error after the join
        if (used[lane] && lane_tail[lane]) (void)hipStreamWaitEvent(ns, lane_tail[lane], 0);
        else (void)hipStreamWaitEvent(ns, fork_ev, 0);
        st = ns; lane_s[lane] = ns; used[lane] = true;
      }
    } else if (!used[lane]) {
      used[lane] = true;
      (void)hipStreamWaitEvent(st, fork_ev, 0);
    }
    for (int k = 0; k < npend; ++k) (void)hipStreamWaitEvent(st, pend[k], 0);
    return st;
  }

  // ---- deferred ops + list scheduling ------------------------------------------------------
  // ROCm resolves a cross-queue dependency of a graph node against what has ALREADY been placed on
  // the producer's queue, so the order in which ops are issued (captured) decides when a side block
  // can start: issuing the x_{i,j} grid block by block left three lanes idle for the first third of
  // the backward pass (profiles/r01_summary.md). Ops are therefore collected first, then issued in
  // the order of a simulated list schedule (earliest start, longest remaining path first). `leaf`
  // ops (weight gradients: nothing but the final unpack consumes them) float to whichever lane is
  // idle. Conflicting ops (RAW/WAW/WAR on a resource) keep their program order, so the events
  // derived at issue time are the ones program order would give.
  struct Op { int lane, leaf; float cost; int nrd, nwr; int rd[12], wr[8]; char name[32]; std::function<int(hipStream_t)> fn; };
  std::vector<Op> ops;
  void add(int lane, int leaf, float cost, std::initializer_list<int> rd, std::initializer_list<int> wr, std::function<int(hipStream_t)> fn) {
    Op o; o.lane = lane_map[lane]; o.leaf = leaf; o.cost = cost; o.nrd = o.nwr = 0;
    for (int x : rd) if (x >= 0 && o.nrd < 12) o.rd[o.nrd++] = x;
    for (int x : wr) if (x >= 0 && o.nwr < 8) o.wr[o.nwr++] = x;
    memcpy(o.name, cur_name, sizeof(o.name)); cur_name[0] = 0;
    o.fn = std::move(fn);
    ops.push_back(std::move(o));
  }
  void add_v(int lane, int leaf, float cost, const int* rd, int nrd, std::function<int(hipStream_t)> fn) {
    Op o; o.lane = lane_map[lane]; o.leaf = leaf; o.cost = cost; o.nrd = o.nwr = 0;
    for (int q = 0; q < nrd; ++q) if (rd[q] >= 0 && o.nrd < 12) o.rd[o.nrd++] = rd[q];
    memcpy(o.name, cur_name, sizeof(o.name)); cur_name[0] = 0;
    o.fn = std::move(fn);
    ops.push_back(std::move(o));
  }
  int run_ops();
  void end() {
    if (!multi) { stamp(main_s, 0); cur_name[0] = 0; return; }
    hipStream_t st = lane_s[cur_lane];
    if (!st) return;
    stamp(st, cur_lane); cur_name[0] = 0;
    if (capturing) graph_tag_tail(st, cur_lane);
    hipEvent_t ev = new_event();
    (void)hipEventRecord(ev, st);
    lane_tail[cur_lane] = ev;
    for (int k = 0; k < nreads; ++k) { res[reads[k]].r_ev[cur_lane] = ev; res[reads[k]].r_st[cur_lane] = st; }
    for (int k = 0; k < nwrites; ++k) {
      Res& R = res[writes[k]];
      R.w_ev = ev; R.w_st = st;
      for (int l = 0; l < NLANES; ++l) { R.r_ev[l] = nullptr; R.r_st[l] = nullptr; }
    }
  }
  void join() {
    if (!multi) return;
    for (int l = 0; l < NLANES; ++l)
      if (used[l] && lane_tail[l]) (void)hipStreamWaitEvent(main_s, lane_tail[l], 0);
  }
};

void Sched::init(nunet_plan* P, hipStream_t s, int pass_) {
  const int pass = pass_;
  this->pass = pass_;
  cur_name[0] = 0;
  rt = rt_of(P);
  main_s = s;
  {
    static int want_stamps = -1;
    if (want_stamps < 0) { const char* e = getenv("NUNET_STAMPS"); want_stamps = e ? atoi(e) : 0; }
    hipStreamCaptureStatus cs0 = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(s, &cs0);
    if (want_stamps && !rt->stamps && cs0 != hipStreamCaptureStatusActive) {
      if (hipMalloc((void**)&rt->stamps, 2 * STAMP_CAP * sizeof(unsigned long long)) != hipSuccess) { (void)hipGetLastError(); rt->stamps = nullptr; }
    }
    rt->stamp_labels[pass].clear();
    if (rt->stamps) { snprintf(cur_name, sizeof(cur_name), "start"); stamp(s, 0); cur_name[0] = 0; }
  }
  multi = rt->multistream != 0 && rt->lanes_ok;
  failed = false;
  capturing = false;
  pool = &rt->events[pass]; pool_used = &rt->events_used[pass];
  *pool_used = 0;
  memset(res, 0, sizeof(res));
  cur_lane = 0; nreads = nwrites = 0; npend = 0;
  fork_ev = nullptr;
  if (multi) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cs) == hipSuccess && cs == hipStreamCaptureStatusActive) capturing = true;
    else (void)hipGetLastError();
    if (!capturing && rt->cap_streams.empty()) {
      // pool for later captures, created outside any capture (warm-up passes run eagerly first)
      for (int k = 0; k < 320; ++k) {
        hipStream_t q;
        if (hipStreamCreateWithFlags(&q, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); break; }
        rt->cap_streams.push_back(q);
      }
    }
    if (capturing && (rt->cap_streams.empty())) multi = false;     // never warmed up eagerly: stay on one stream
    if (capturing && pass == 0) rt->cap_next = 0;                   // forward + backward of one capture share the pool
  }
  {
    // NUNET_LANE_MAP: 10 digits, lane of (level 0..4, wgrad of level 0..4). Default "0123401234":
    // one lane per pyramid level, weight gradients on their level's lane (measured best on MI355X:
    // separate wgrad lanes add cross-queue edges that cost more than the overlap they buy).
    const char* e = getenv("NUNET_LANE_MAP");
    for (int l = 0; l < NLANES; ++l) lane_map[l] = (e && strlen(e) == NLANES && e[l] >= '0' && e[l] <= '9') ? e[l] - '0' : l % 5;
  }
  static int lane_mod = 0;
  if (!lane_mod) { const char* e = getenv("NUNET_LANE_MOD"); lane_mod = e ? atoi(e) : NLANES; if (lane_mod < 1 || lane_mod > NLANES) lane_mod = NLANES; }
  for (int l = 0; l < NLANES; ++l) { lane_s[l] = capturing ? nullptr : rt->lanes[l % lane_mod]; used[l] = false; lane_tail[l] = nullptr; }
  if (multi) {
    fork_ev = new_event();
    (void)hipEventRecord(fork_ev, main_s);
    if (capturing) graph_tag_tail(main_s, -1);   // launches made so far on the caller's stream are not ours to place
  }
}

int Sched::run_ops() {
  const int n = (int)ops.size();
  int rc = NUNET_OK;
  static int listsched = -1;
  if (listsched < 0) { const char* e = getenv("NUNET_LISTSCHED"); listsched = e ? atoi(e) : 0; }
  std::vector<int> order; order.reserve(n);
  std::vector<int> on_lane(n);
  for (int j = 0; j < n; ++j) on_lane[j] = ops[j].lane;
  if (!multi || !listsched || n < 3) {
    for (int j = 0; j < n; ++j) order.push_back(j);
  } else {
    // dependency edges from program order
    std::vector<std::vector<int>> deps(n), succ(n);
    {
      std::vector<int> lastw(NRES, -1);
      std::vector<std::vector<int>> rds(NRES);
      for (int j = 0; j < n; ++j) {
        auto dep = [&](int d) { if (d >= 0 && d != j) { for (int x : deps[j]) if (x == d) return; deps[j].push_back(d); succ[d].push_back(j); } };
        for (int q = 0; q < ops[j].nrd; ++q) dep(lastw[ops[j].rd[q]]);
        for (int q = 0; q < ops[j].nwr; ++q) { const int r = ops[j].wr[q]; dep(lastw[r]); for (int x : rds[r]) dep(x); }
        for (int q = 0; q < ops[j].nrd; ++q) rds[ops[j].rd[q]].push_back(j);
        for (int q = 0; q < ops[j].nwr; ++q) { const int r = ops[j].wr[q]; lastw[r] = j; rds[r].clear(); }
      }
    }
    std::vector<float> bl(n, 0.f);   // longest path to a sink, own cost included
    for (int j = n - 1; j >= 0; --j) { float m = 0.f; for (int x : succ[j]) m = bl[x] > m ? bl[x] : m; bl[j] = ops[j].cost + m; }
    int nl = 4;
    for (int j = 0; j < n; ++j) if (ops[j].lane + 1 > nl) nl = ops[j].lane + 1;
    if (nl > NLANES) nl = NLANES;
    const float XL = 6.f;            // dispatch latency of an edge that crosses hardware queues (us)
    std::vector<float> fin(n, 0.f), lane_free(nl, 0.f);
    std::vector<char> done(n, 0);
    std::vector<int> ndep(n);
    int nonleaf_left = 0;
    for (int j = 0; j < n; ++j) { ndep[j] = (int)deps[j].size(); if (!ops[j].leaf) ++nonleaf_left; }
    auto ready_at = [&](int j, int L) { float r = 0.f; for (int d : deps[j]) { float f = fin[d] + (on_lane[d] != L ? XL : 0.f); r = f > r ? f : r; } return r; };
    float t = 0.f;
    int left = n;
    while (left > 0) {
      bool progress = false;
      for (int L = 0; L < nl; ++L) {
        if (lane_free[L] > t) continue;
        int best = -1; bool best_leaf = true; float best_bl = -1.f;
        for (int j = 0; j < n; ++j) {
          if (done[j] || ndep[j] > 0) continue;
          const bool leaf = ops[j].leaf != 0;
          if (!leaf && ops[j].lane != L) continue;
          if (leaf && L == 0 && nonleaf_left > 0) continue;          // keep the critical-chain lane free
          if (ready_at(j, L) > t) continue;
          if (best < 0 || (best_leaf && !leaf) || (best_leaf == leaf && bl[j] > best_bl)) { best = j; best_leaf = leaf; best_bl = bl[j]; }
        }
        if (best >= 0 && best_leaf) {
          // do not start a long leaf just before an op of this lane becomes ready
          for (int j = 0; j < n && best >= 0; ++j)
            if (!done[j] && ndep[j] == 0 && !ops[j].leaf && ops[j].lane == L && ready_at(j, L) < t + 0.5f * ops[best].cost) best = -1;
        }
        if (best < 0) continue;
        done[best] = 1; --left; if (!ops[best].leaf) --nonleaf_left;
        on_lane[best] = L; fin[best] = t + ops[best].cost; lane_free[L] = fin[best];
        for (int x : succ[best]) --ndep[x];
        order.push_back(best);
        progress = true;
      }
      if (progress) continue;
      float nt = 1e30f;
      for (int L = 0; L < nl; ++L) if (lane_free[L] > t && lane_free[L] < nt) nt = lane_free[L];
      for (int j = 0; j < n; ++j) {
        if (done[j] || ndep[j] > 0) continue;
        for (int L = 0; L < nl; ++L) {
          if (!ops[j].leaf && ops[j].lane != L) continue;
          float r = ready_at(j, L); if (r < lane_free[L]) r = lane_free[L];
          if (r > t && r < nt) nt = r;
        }
      }
      if (nt > 1e29f) nt = t + 1.f;     // cannot happen for a valid program order; keeps the loop finite
      t = nt;
      if (t > 1e7f) break;
    }
    if (left > 0) { order.clear(); for (int j = 0; j < n; ++j) { order.push_back(j); on_lane[j] = ops[j].lane; } }
    if (getenv("NUNET_SCHED_DUMP")) {
      for (int j : order) fprintf(stderr, "sched %3d lane %d leaf %d cost %6.1f start %8.1f bl %7.1f\n", j, on_lane[j], ops[j].leaf, ops[j].cost, fin[j] - ops[j].cost, bl[j]);
    }
  }
  for (int q = 0; q < n && rc == NUNET_OK; ++q) {
    Op& o = ops[order[q]];
    hipStream_t st = begin_v(on_lane[order[q]], o.rd, o.nrd, o.wr, o.nwr);
    rc = o.fn(st);
    memcpy(cur_name, o.name, sizeof(cur_name));
    end();
  }
  ops.clear();
  return rc;
}

// Lane of a block. Crossing HW queues costs 5-10 us of dispatch latency per dependency edge, so the
// assignment decides how many edges of the critical chain (B00>B10>B20>B30>B40>B31>B22>B13>B04 and its
// mirror in backward) cross lanes. NUNET_LANE_MODE: 0 = pyramid level, 1 = anti-diagonal,
// 2 = critical chain on lane 0 and the side blocks on lanes 1-3 by anti-diagonal.
static int lane_of(const nunet_plan* P, const Node& n) {
  static int mode = -1;
  if (mode < 0) { const char* e = getenv("NUNET_LANE_MODE"); mode = e ? atoi(e) : 2; }   // measured best: 2
  if (P->cfg.unet || mode == 0) return n.i;
  if (mode == 1) return (n.i + n.j) % 5;
  if (n.j == 0 || n.i + n.j == 4) return 0;
  if (mode == 2) return n.i + n.j;           // side blocks: diagonals 1..3 -> lanes 1..3
  if (mode == 3) return 1 + n.i;             // side blocks by level 0..2 -> lanes 1..3
  if (mode == 4) return n.j;                 // side blocks by column 1..3 -> lanes 1..3
  if (mode == 5) return 1 + (n.i + n.j) % 2; // two side lanes
  return 1;                                  // mode 6: one side lane
}

static int blk_index(const nunet_plan* P, int i, int in_prefix_zero_only) {
  for (size_t q = 0; q < P->exec.size(); ++q)
    if (P->exec[q].i == i && (!in_prefix_zero_only || P->exec[q].in_prefix == 0)) return (int)q;
  return -1;
}

extern "C" int nunet_plan_forward(nunet_plan* P, const float* params, float* bnbuf, int64_t* nbt, const float* input, void* arena, float* logits, int32_t training_flags, nunet_stream_t s) {
  const int32_t training = training_flags & 1;
  NUNET_REQUIRE(P && params && input && arena && logits, "plan_forward: null pointer");
  NUNET_REQUIRE(bnbuf, "plan_forward: bnbuf (running stats) required");
  hipStream_t st = (hipStream_t)s;
  const nunet_plan_cfg& c = P->cfg;
  const int dt = c.dtype, es = P->es;
  float* stats = (float*)AB(arena, P->off_stats);
  float* save = (float*)AB(arena, P->off_save);
  char* wpack = AB(arena, P->off_wpack);
  // prerequisites of everything on the caller's stream, before the fork
  // NUNET_PREP_LANE=1: statistics zero + input layout run on lane 1 beside the weight pack on lane 0 instead of ahead
  // of the fork (every conv descends from B00.conv1, which reads R_IMG, so the zeroed statistics are ordered too)
  static int prep_lane = -1;
  if (prep_lane < 0) { const char* e = getenv("NUNET_PREP_LANE"); prep_lane = e ? atoi(e) : 0; }
  auto prep = [&](hipStream_t ps) -> int {
    if (training) CK(nunet_zero_async(stats, P->stats_floats * 4 * NUNET_BN_SUM_REPLICAS, ps));
    if (rt_of(P)->sk_ready_arena != arena) {   // K-split arrival counters: zero once, every launch leaves them zero
      for (int l = 0; l < 5; ++l)
        if (P->sk_floats[l] > 0) CK(nunet_zero_async(AB(arena, P->off_sk[l]), NUNET_SPLITK_COUNTER_FLOATS * 4, ps));
      rt_of(P)->sk_ready_arena = arena;
    }
    CK(nunet_nchw_to_nhwc(input, c.N, c.input_channels, c.H, c.W, dt, AB(arena, P->off_img), 32, (nunet_stream_t)ps));
    return NUNET_OK;
  };
  if (!prep_lane) CK(prep(st));

  Sched S; S.init(P, st, 0);
  int rc = NUNET_OK;
  if (prep_lane) {
    S.name("prep");
    hipStream_t ls = S.begin(1, {}, {R_IMG, R_SK + 0, R_SK + 1, R_SK + 2, R_SK + 3, R_SK + 4});
    rc = prep(ls);
    S.end();
  }
  // Measured on MI355X: repacking the weights per level on the lanes (NUNET_PACK_LANES=1) is 9 %
  // SLOWER than one pack launch ahead of the lanes; clearing the gradient scratch on a lane during
  // forward (NUNET_GS_FWD=1) is neutral. Both stay off by default.
  static int pack_lanes = -1, gs_fwd = -1;
  if (pack_lanes < 0) { const char* e = getenv("NUNET_PACK_LANES"); pack_lanes = e ? atoi(e) : 0; e = getenv("NUNET_GS_FWD"); gs_fwd = e ? atoi(e) : 0; }
  const bool skip_pack = (training_flags & 2) != 0;   // the caller vouches that nunet_plan_update / _repack left the packed weights current
  if (!pack_lanes && !skip_pack) {
    S.name("pack");
    hipStream_t ls = S.begin(0, {}, {R_WP + 0, R_WP + 1, R_WP + 2, R_WP + 3, R_WP + 4});
    if (dt == NUNET_F32) rc = launch_pack<float>(params, wpack, P->ptab, P->pack_maxn, ls);
    else if (dt == NUNET_BF16) rc = launch_pack<bf16_t>(params, wpack, P->ptab, P->pack_maxn, ls);
    else rc = launch_pack<f16_t>(params, wpack, P->ptab, P->pack_maxn, ls);
    S.end();
  }
  for (int l = 0; l < 5 && rc == NUNET_OK && pack_lanes && !skip_pack; ++l) {
    if (P->ptab_lvl[l].n == 0) continue;
    // 1: each level on its own lane; 2: level 0 on the chain lane (needed first), the rest one after the other on lane 3
    S.name("pack%d", l);
    hipStream_t ls = S.begin(pack_lanes == 2 ? (l == 0 ? 0 : 3) : l, {}, {R_WP + l});
    if (dt == NUNET_F32) rc = launch_pack<float>(params, wpack, P->ptab_lvl[l], P->pack_maxn_lvl[l], ls);
    else if (dt == NUNET_BF16) rc = launch_pack<bf16_t>(params, wpack, P->ptab_lvl[l], P->pack_maxn_lvl[l], ls);
    else rc = launch_pack<f16_t>(params, wpack, P->ptab_lvl[l], P->pack_maxn_lvl[l], ls);
    S.end();
  }
  if (training && rc == NUNET_OK && gs_fwd) {
    // the gradient scratch of the coming backward is cleared here, on the least loaded lane
    hipStream_t ls = S.begin(4, {}, {R_GS});
    rc = nunet_zero_async(AB(arena, P->off_gs), P->gs_floats * 4, ls);
    S.end();
    rt_of(P)->gs_clean_arena = arena;
  }
  for (size_t k = 0; k < P->exec.size() && rc == NUNET_OK; ++k) {
    const Node& n = P->exec[k];
    const int i = n.i, f = NBF[i], H = P->hl[i], W = P->wl[i];
    const int lane = lane_of(P, n), rb = R_BLK + (int)k * B_STRIDE;
    if (n.up_slot >= 0) {
      S.name("B%d%d.upF", n.i, n.j);
      hipStream_t ls = S.begin(lane, {R_X + (i + 1) * 5 + n.up_slot}, {rb + B_UP});
      rc = nunet_upsample2x_fwd(dt, c.N, P->hl[i + 1], P->wl[i + 1], NBF[i + 1],
                                AB(arena, P->X[i + 1] + (size_t)n.up_slot * NBF[i + 1] * es), P->PX[i + 1],
                                AB(arena, n.up), NBF[i + 1], ls);
      S.end();
      if (rc) break;
    }
    for (int cv = 0; cv < 2 && rc == NUNET_OK; ++cv) {
      const ConvL& L = cv == 0 ? n.c1 : n.c2;
      nunet_conv_desc d; memset(&d, 0, sizeof(d));
      d.dtype = dt; d.N = c.N; d.H = H; d.W = W;
      hipStream_t ls;
      const int rskf = P->sk_floats[i] > 0 ? R_SK + i : -1;
      S.name("B%d%d.conv%d", n.i, n.j, cv + 1);
      if (cv == 1) {
        d.src0 = AB(arena, n.a1); d.C0 = f; d.P0 = f;
        ls = S.begin(lane, {rb + B_A1, R_WP + i}, {rb + B_Y2, rb + B_ST2, rskf});
      } else if (n.in_prefix == 0) {
        if (i == 0) { d.src0 = AB(arena, P->off_img); d.C0 = 32; d.P0 = 32; ls = S.begin(lane, {R_IMG, R_WP + i}, {rb + B_Y1, rb + B_ST1, rskf}); }
        else { d.src0 = AB(arena, n.pin); d.C0 = NBF[i - 1]; d.P0 = NBF[i - 1]; ls = S.begin(lane, {rb + B_PIN, R_WP + i}, {rb + B_Y1, rb + B_ST1, rskf}); }
      } else {
        d.src0 = AB(arena, P->X[i]); d.C0 = n.in_prefix * f; d.P0 = P->PX[i];
        d.src1 = AB(arena, n.up); d.C1 = NBF[i + 1]; d.P1 = NBF[i + 1];
        ls = S.begin(lane, {R_X + i * 5 + 0, n.in_prefix > 1 ? R_X + i * 5 + 1 : -1, n.in_prefix > 2 ? R_X + i * 5 + 2 : -1,
                            n.in_prefix > 3 ? R_X + i * 5 + 3 : -1, rb + B_UP, R_WP + i}, {rb + B_Y1, rb + B_ST1, rskf});
      }
      d.wpack = wpack + (size_t)L.wf * es;
      d.bias = nullptr;  // absorbed by the BatchNorm that follows (see bn_channel_coeffs)
      d.dst0 = AB(arena, cv == 0 ? n.y1 : n.y2); d.D0 = f; d.Q0 = f;
      d.stats = training ? stats + L.stats * NUNET_BN_SUM_REPLICAS : nullptr;
      if (P->sk_floats[i] > 0) { d.splitk_ws = (float*)AB(arena, P->off_sk[i]); d.splitk_ws_floats = P->sk_floats[i]; }
      g_prof_alg_cin = (cv == 0 && i == 0 && n.in_prefix == 0) ? c.input_channels : 0;
      rc = nunet_conv3x3_fwd(&d, ls);
      g_prof_alg_cin = 0;
      S.end();
      if (rc) break;

      nunet_bn_fwd_desc b; memset(&b, 0, sizeof(b));
      b.dtype = dt; b.N = c.N; b.H = H; b.W = W; b.C = f;
      b.y = d.dst0; b.PY = f; b.conv_bias = params + L.b_off; b.stats = stats + L.stats * NUNET_BN_SUM_REPLICAS;
      b.gamma = params + L.g_off; b.beta = params + L.be_off;
      b.running_mean = bnbuf + L.rm_off; b.running_var = bnbuf + L.rv_off;
      b.num_batches_tracked = nbt ? nbt + L.bn_index : nullptr;
      b.save_mean_invstd = save + L.save; b.training = training; b.momentum = 0.1f; b.eps = 1e-5f;
      S.name("B%d%d.bnF%d", n.i, n.j, cv + 1);
      if (cv == 0) {
        b.a = AB(arena, n.a1); b.PA = f;
        ls = S.begin(lane, {rb + B_Y1, rb + B_ST1}, {rb + B_A1});
      } else {
        b.a = AB(arena, P->X[i] + (size_t)n.out_slot * f * es); b.PA = P->PX[i];
        int rpin = -1;
        if (n.in_prefix == 0 && i < 4) {  // encoder column: feed the next level (archs1.py:115,118,122,127)
          const int q = blk_index(P, i + 1, 1);
          if (q >= 0) { b.pooled = AB(arena, P->exec[q].pin); b.PP = f; rpin = R_BLK + q * B_STRIDE + B_PIN; }
        }
        ls = S.begin(lane, {rb + B_Y2, rb + B_ST2}, {R_X + i * 5 + n.out_slot, rpin});
      }
      rc = nunet_bn_relu_fwd(&b, ls);
      S.end();
    }
  }
  if (rc == NUNET_OK) {
    const long long plane = (long long)c.N * c.num_classes * c.H * c.W;
    for (size_t k = 0; k < P->heads.size() && rc == NUNET_OK; ++k) {
      S.name("head%d.F", (int)k);
      hipStream_t ls = S.begin(0, {R_X + P->heads[k].slot}, {R_LOGITS});
      rc = nunet_head_fwd(dt, c.N, c.H, c.W, NBF[0], c.num_classes, AB(arena, P->X[0] + (size_t)P->heads[k].slot * NBF[0] * es), P->PX[0],
                          params + P->heads[k].w_off, params + P->heads[k].b_off, logits + plane * k, ls);
      S.end();
    }
  }
  S.join();  // always rejoin the caller's stream (also on error: a capture must not be left forked)
  if (rc == NUNET_OK && S.failed) { nunet_set_error("plan_forward: capture lane pool exhausted"); rc = NUNET_EINVAL; }
  return rc;
}

extern "C" int nunet_plan_backward_phase(nunet_plan* P, const float* params, const float* dlogits, void* arena, float* grads, int32_t accumulate, int32_t phases, nunet_stream_t s);
extern "C" int nunet_plan_backward(nunet_plan* P, const float* params, const float* dlogits, void* arena, float* grads, int32_t accumulate, nunet_stream_t s) {
  return nunet_plan_backward_phase(P, params, dlogits, arena, grads, accumulate, 7, s);
}

extern "C" int nunet_plan_grad_scratch(const nunet_plan* P, int64_t* byte_offset, int64_t* bucket0_floats, int64_t* total_floats) {
  NUNET_REQUIRE(P && byte_offset && bucket0_floats && total_floats, "plan_grad_scratch: null pointer");
  *byte_offset = (int64_t)P->off_gs;
  *bucket0_floats = P->gs_bucket0;
  *total_floats = (int64_t)(P->gs_floats - P->stats_floats * NUNET_BN_SUM_REPLICAS);
  return NUNET_OK;
}

// Fused optimiser step on the plan's own buffers (update_kernel above): scratch -> SGD -> repacked weights.
// `grads` (flat OIHW arena) is optional: when given it receives the (scaled) gradients as nunet_plan_backward would
// have left them. Afterwards the packed weights in `arena` are current: the next nunet_plan_forward may be called
// with bit 1 of `training` set (skip the repack).
extern "C" int nunet_plan_update(nunet_plan* P, float* params, float* momentum, void* arena, const float* lr_dev, float mom, float wd,
                                 int32_t nesterov, float grad_scale, float* grads, nunet_stream_t s) {
  NUNET_REQUIRE(P && params && momentum && arena && lr_dev, "plan_update: null pointer");
  hipStream_t st = (hipStream_t)s;
  PackTab& tab = P->ptab;
  int nt = 0;
  for (int i = 0; i < tab.n; ++i) { tab.tile0[i] = nt; nt += ((tab.e[i].cout + 31) / 32) * ((tab.e[i].cinpad + 31) / 32); }
  tab.tile0[tab.n] = nt; tab.ntiles = nt;
  UpdP u;
  u.params = params; u.mom = momentum; u.scratch = (const float*)AB(arena, P->off_gs); u.grads = grads; u.lr = lr_dev;
  u.momc = mom; u.wd = wd; u.gscale = grad_scale; u.nesterov = nesterov; u.nconv = tab.n;
  const int nheads = P->utab.n - tab.n;
  ProfScope ps(PC_SGD, 0, (double)P->nparams * (grads ? 28.0 : 24.0), st);
  void* wp = AB(arena, P->off_wpack);
  const dim3 grid(nt + nheads), blk(512);
  if (P->cfg.dtype == NUNET_F32) hipLaunchKernelGGL((update_kernel<float>), grid, blk, 0, st, u, (float*)wp, tab, P->utab);
  else if (P->cfg.dtype == NUNET_BF16) hipLaunchKernelGGL((update_kernel<bf16_t>), grid, blk, 0, st, u, (bf16_t*)wp, tab, P->utab);
  else hipLaunchKernelGGL((update_kernel<f16_t>), grid, blk, 0, st, u, (f16_t*)wp, tab, P->utab);
  return nunet_check_launch("plan_update");
}

// Optimiser step straight from the gradient scratch (unpack_sgd_kernel): replaces nunet_plan_backward_phase bit 2 +
// nunet_sgd_step; the weights are repacked by the next nunet_plan_forward as usual.
// unpack_tiled_kernel with the optimiser step as the epilogue of its store phase: the tile's gradients meet the OIHW
// parameters and momentum as 16-byte runs, the flat gradient arena is written only when the caller wants it.
__global__ __launch_bounds__(256) void unpack_sgd_tiled_kernel(UpdP u, PackTab tab, UnpackTab ut) {
  __shared__ float s_t[32][32 * 9 + 1];
  const float lr = u.lr[0];
  auto step1 = [&](long long idx, float g) {
    g *= u.gscale;
    if (u.grads) u.grads[idx] = g;
    float m = u.mom[idx];
    const float pn = sgd_one(u.params[idx], g, &m, u, lr);
    u.mom[idx] = m; u.params[idx] = pn;
  };
  if ((int)blockIdx.x >= tab.ntiles) {
    const UnpackEnt en = ut.e[(int)blockIdx.x - tab.ntiles];
    const float* dw = u.scratch + en.src;
    const int nw = en.cout * en.cin * en.taps;
    if (en.nslab > 1) {
      // 1x1 head: 256/ne threads share an element's slabs (a single thread walking all 256 slabs is a chain of 256
      // dependent-latency loads: that loop alone made the earlier fused kernels 100 us long), fixed summation order
      const int tot = nw + en.nvec * en.cout;
      for (int e0 = 0; e0 < tot; e0 += 256) {
        const int ne = min(256, tot - e0);
        const int parts = 256 / ne;
        const int e = threadIdx.x % ne, part = threadIdx.x / ne;
        float v = 0.f;
        if (part < parts) {
#pragma unroll 8
          for (int sl = part; sl < en.nslab; sl += parts) v += dw[(long long)sl * tot + e0 + e];
        }
        s_t[0][threadIdx.x] = part < parts ? v : 0.f;
        __syncthreads();
        if ((int)threadIdx.x < ne) {
          float g = 0.f;
          for (int q = 0; q < parts; ++q) g += s_t[0][q * ne + threadIdx.x];
          step1(en.dst + e0 + threadIdx.x, g);
        }
        __syncthreads();
      }
      return;
    }
    const float* vsrc = dw + (long long)en.taps * en.cout * en.cinpad;
    for (int i = threadIdx.x; i < en.nvec * en.cout; i += blockDim.x) step1(en.dst + nw + i, vsrc[i]);
    return;
  }
  int e = 0;
  while (e + 1 < tab.n && (int)blockIdx.x >= tab.tile0[e + 1]) ++e;
  const PackEnt en = tab.e[e];
  const UnpackEnt ue = ut.e[e];
  const int t = blockIdx.x - tab.tile0[e];
  const int nci = (en.cinpad + 31) / 32;
  const int co0 = (t / nci) * 32, ci0 = (t % nci) * 32;
  const int cw = min(32, en.cin - ci0), rw = min(32, en.cout - co0);
  if (cw <= 0) return;
  const float* dw = u.scratch + ue.src;
  const long long row0 = ue.dst + ((long long)co0 * en.cin + ci0) * 9;
  const bool full = rw == 32 && cw == 32;
  if (full && en.cinpad % 4 == 0 && ((uintptr_t)dw & 15) == 0) {
    for (int i = threadIdx.x; i < 9 * 32 * 8; i += blockDim.x) {
      const int c4 = i & 7, ro = (i >> 3) & 31, tap = i >> 8;
      const f32x4 v = *reinterpret_cast<const f32x4*>(dw + ((long long)tap * en.cout + co0 + ro) * en.cinpad + ci0 + c4 * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) s_t[ro][(c4 * 4 + j) * 9 + tap] = v[j];
    }
  } else {
    for (int i = threadIdx.x; i < 9 * 32 * 32; i += blockDim.x) {
      const int ci = i & 31, ro = (i >> 5) & 31, tap = i >> 10;
      if (ro < rw && ci < cw) s_t[ro][ci * 9 + tap] = dw[((long long)tap * en.cout + co0 + ro) * en.cinpad + ci0 + ci];
    }
  }
  __syncthreads();
  const bool al = ((uintptr_t)(u.params + row0) & 15) == 0 && ((uintptr_t)(u.mom + row0) & 15) == 0 && (!u.grads || ((uintptr_t)(u.grads + row0) & 15) == 0);
  if (full && en.cin % 4 == 0 && al) {
    for (int i = threadIdx.x; i < 32 * 72; i += blockDim.x) {
      const int ro = i / 72, k4 = i - ro * 72;
      const long long idx = row0 + (long long)ro * en.cin * 9 + k4 * 4;
      const float* sp = &s_t[ro][k4 * 4];
      f32x4 g = {sp[0] * u.gscale, sp[1] * u.gscale, sp[2] * u.gscale, sp[3] * u.gscale};
      f32x4 pv = *reinterpret_cast<const f32x4*>(u.params + idx), mv = *reinterpret_cast<const f32x4*>(u.mom + idx);
      if (u.grads) *reinterpret_cast<f32x4*>(u.grads + idx) = g;
#pragma unroll
      for (int j = 0; j < 4; ++j) { float m = mv[j]; pv[j] = sgd_one(pv[j], g[j], &m, u, lr); mv[j] = m; }
      *reinterpret_cast<f32x4*>(u.mom + idx) = mv;
      *reinterpret_cast<f32x4*>(u.params + idx) = pv;
    }
  } else {
    for (int i = threadIdx.x; i < 32 * 288; i += blockDim.x) {
      const int ro = i / 288, k = i - ro * 288;
      if (ro < rw && k < cw * 9) step1(row0 + (long long)ro * en.cin * 9 + k, s_t[ro][k]);
    }
  }
}

extern "C" int nunet_plan_sgd(nunet_plan* P, float* params, float* momentum, void* arena, const float* lr_dev, float mom, float wd,
                              int32_t nesterov, float grad_scale, float* grads, nunet_stream_t s) {
  NUNET_REQUIRE(P && params && momentum && arena && lr_dev, "plan_sgd: null pointer");
  hipStream_t st = (hipStream_t)s;
  UpdP u;
  u.params = params; u.mom = momentum; u.scratch = (const float*)AB(arena, P->off_gs); u.grads = grads; u.lr = lr_dev;
  u.momc = mom; u.wd = wd; u.gscale = grad_scale; u.nesterov = nesterov; u.nconv = P->ptab.n;
  int gx = (int)ceil_div64(P->unpack_maxn, 256 * 4);
  if (gx > 512) gx = 512;
  ProfScope ps(PC_SGD, 0, (double)P->nparams * (grads ? 24.0 : 20.0), st);
  static int tiled = -1;
  if (tiled < 0) { const char* e = getenv("NUNET_UNPACK_TILED"); tiled = e ? atoi(e) : 1; }
  if (tiled) {
    PackTab& tab = P->ptab;
    int nt = 0;
    for (int i = 0; i < tab.n; ++i) { tab.tile0[i] = nt; nt += ((tab.e[i].cout + 31) / 32) * ((tab.e[i].cinpad + 31) / 32); }
    tab.tile0[tab.n] = nt; tab.ntiles = nt;
    hipLaunchKernelGGL(unpack_sgd_tiled_kernel, dim3(nt + P->utab.n), dim3(256), 0, st, u, tab, P->utab);
  } else {
    hipLaunchKernelGGL(unpack_sgd_kernel, dim3(gx, P->utab.n), dim3(256), 0, st, u, P->utab);
  }
  return nunet_check_launch("plan_sgd");
}

// Repack the 16-bit weight layouts from the fp32 master parameters (what nunet_plan_forward does first unless told
// that they are current): needed once before a loop that relies on nunet_plan_update, and after the parameters were
// changed by anything else (checkpoint load, a stock optimiser).
extern "C" int nunet_plan_repack(nunet_plan* P, const float* params, void* arena, nunet_stream_t s) {
  NUNET_REQUIRE(P && params && arena, "plan_repack: null pointer");
  char* wpack = AB(arena, P->off_wpack);
  if (P->cfg.dtype == NUNET_F32) return launch_pack<float>(params, wpack, P->ptab, P->pack_maxn, (hipStream_t)s);
  if (P->cfg.dtype == NUNET_BF16) return launch_pack<bf16_t>(params, wpack, P->ptab, P->pack_maxn, (hipStream_t)s);
  return launch_pack<f16_t>(params, wpack, P->ptab, P->pack_maxn, (hipStream_t)s);
}

// phases: 1 = clear scratch, heads and the last anti-diagonal's blocks (75 % of the gradient bytes);
//         2 = the remaining blocks; 4 = unpack into the flat OIHW gradient arena. 7 = everything.
extern "C" int nunet_plan_backward_phase(nunet_plan* P, const float* params, const float* dlogits, void* arena, float* grads, int32_t accumulate, int32_t phases, nunet_stream_t s) {
  NUNET_REQUIRE(P && params && dlogits && arena && grads, "plan_backward: null pointer");
  hipStream_t st = (hipStream_t)s;
  const nunet_plan_cfg& c = P->cfg;
  const int dt = c.dtype, es = P->es;
  float* gsr = (float*)AB(arena, P->off_gs);
  float* bsums = gsr + (P->gs_floats - P->stats_floats * NUNET_BN_SUM_REPLICAS);
  float* save = (float*)AB(arena, P->off_save);
  char* wpack = AB(arena, P->off_wpack);
  bool (&written)[5][5] = rt_of(P)->bwd_written;
  int (&pp)[5] = rt_of(P)->bwd_pp;   // per-level ping-pong of the dY scratch
  if (phases & 1) {
    if (rt_of(P)->gs_clean_arena == arena) rt_of(P)->gs_clean_arena = nullptr;   // cleared by the forward that produced the activations
    else CK(nunet_zero_async(gsr, P->gs_floats * 4, st));
    memset(written, 0, sizeof(written));
    memset(pp, 0, sizeof(pp));
  }
  const int nnodes = (int)P->exec.size();
  const int k_split = nnodes - P->first_phase_nodes;      // phase 1: nodes [k_split, nnodes); phase 2: [0, k_split)
  const int k_hi = (phases & 1) ? nnodes - 1 : k_split - 1;
  const int k_lo = (phases & 2) ? 0 : k_split;

  Sched S; S.init(P, st, 1);
  int rc = NUNET_OK;
  const long long plane = (long long)c.N * c.num_classes * c.H * c.W;
  // cost model (us) of the simulated schedule: launch floor + algorithmic work at the rate these
  // kernels reach on MI355X (profiles/r01_summary.md); only the relative order matters
  const double conv_rate = dt == NUNET_F32 ? 0.09e9 : 0.55e9, wg_rate = dt == NUNET_F32 ? 0.07e9 : 0.40e9;   // FLOP per us
  auto cost_conv = [&](int lvl, double cin, double cout) { return (float)(9.0 + (lvl >= 3 ? 8.0 : 0.0) + 18.0 * cin * cout * (double)P->px[lvl] / conv_rate); };
  auto cost_wg = [&](int lvl, double cin, double cout) { return (float)(12.0 + 18.0 * cin * cout * (double)P->px[lvl] / wg_rate); };
  auto cost_mem = [&](double floor_us, double bytes) { return (float)(floor_us + bytes / 4.0e6); };
  static int bnr_fuse_env = -1;
  // measured: fusing takes 1 % off the summed kernel time but the step gets 0.5 % slower (schedule): off by default
  if (bnr_fuse_env < 0) { const char* e = getenv("NUNET_BNR_FUSE"); bnr_fuse_env = e ? atoi(e) : 0; }
  const bool bnr_fuse = bnr_fuse_env != 0;
  for (size_t k = 0; k < P->heads.size() && rc == NUNET_OK && (phases & 1); ++k) {
    const Head& h = P->heads[k];
    const int acc = written[0][h.slot] ? 1 : 0;
    S.name("head%d.B", (int)k);
    S.add(0, 0, 12.f, {R_X + h.slot, R_DLOGITS}, {R_GX + h.slot, R_GSV + 30 + (int)k}, [=](hipStream_t ls) {
      return nunet_head_bwd(dt, c.N, c.H, c.W, NBF[0], c.num_classes, AB(arena, P->X[0] + (size_t)h.slot * NBF[0] * es), P->PX[0],
                            params + h.w_off, dlogits + plane * k, AB(arena, P->GX[0] + (size_t)h.slot * NBF[0] * es), P->PX[0],
                            acc, gsr + h.gs, HEAD_SLABS, ls);
    });
    written[0][h.slot] = true;
  }

  for (int k = k_hi; k >= k_lo && rc == NUNET_OK; --k) {
    const Node& n = P->exec[k];
    const int i = n.i, f = NBF[i], H = P->hl[i], W = P->wl[i];
    const int lane = lane_of(P, n), wlane = 5 + lane, rb = R_BLK + k * B_STRIDE, rl = R_LVL + k * L_STRIDE;
    const int rsk = P->sk_floats[i] > 0 ? R_SK + i : -1;
    if (!written[i][n.out_slot]) { nunet_set_error("plan_backward: internal: grad of x%d_%d never produced", n.i, n.j); rc = NUNET_EINVAL; break; }
    nunet_wgrad_desc wdesc[2]; int wrdy[2] = {-1, -1};
    for (int cv = 1; cv >= 0 && rc == NUNET_OK; --cv) {
      const ConvL& L = cv == 0 ? n.c1 : n.c2;
      const int cidx = 2 * k + cv;
      const int rdy = rl + (pp[i] ? L_DY1 : L_DY0);
      char* dybuf = AB(arena, P->off_dy[k][pp[i]]);
      pp[i] ^= 1;
      // BN + ReLU backward
      nunet_bn_bwd_desc b; memset(&b, 0, sizeof(b));
      b.dtype = dt; b.N = c.N; b.H = H; b.W = W; b.C = f;
      int rda, ry;
      if (cv == 1) { b.da = AB(arena, P->GX[i] + (size_t)n.out_slot * f * es); b.PDA = P->PX[i]; b.y = AB(arena, n.y2); rda = R_GX + i * 5 + n.out_slot; ry = rb + B_Y2; }
      else { b.da = AB(arena, P->off_da1[k]); b.PDA = f; b.y = AB(arena, n.y1); rda = rl + L_DA1; ry = rb + B_Y1; }
      b.PY = f; b.mean_invstd = save + L.save; b.gamma = params + L.g_off; b.beta = params + L.be_off;
      b.sums = bsums + L.bsum * NUNET_BN_SUM_REPLICAS;
      float* gl = gsr + L.gs + 9LL * L.cout * L.cinpad;
      b.dbias = gl; b.dgamma = gl + L.cout; b.dbeta = gl + 2 * L.cout;
      b.dy = dybuf; b.PDY = f;
      S.name("B%d%d.bnB%d", n.i, n.j, cv + 1);
      // the reduce pass of the FIRST conv's BN is taken in the epilogue of the dgrad that produces its
      // input gradient (conv3x3 BNR kernels): one launch and one read of da1 and y1 less per block
      const bool fused_reduce = bnr_fuse && cv == 0;
      S.add(lane, 0, cost_mem(fused_reduce ? 6.0 : 12.0, (fused_reduce ? 3.0 : 5.0) * (double)P->px[i] * f * es), {rda, ry}, {rdy, R_GSV + cidx}, [=](hipStream_t ls) {
        int r = fused_reduce ? NUNET_OK : nunet_bn_relu_bwd_reduce(&b, ls);
        return r ? r : nunet_bn_relu_bwd_apply(&b, ls);
      });
      // weight gradient: a leaf of the dependency graph (only the final unpack reads it)
      nunet_wgrad_desc& w = wdesc[cv]; memset(&w, 0, sizeof(w));
      w.dtype = dt; w.N = c.N; w.H = H; w.W = W;
      if (cv == 1) { w.src0 = AB(arena, n.a1); w.C0 = f; w.P0 = f; }
      else if (n.in_prefix == 0) {
        if (i == 0) { w.src0 = AB(arena, P->off_img); w.C0 = 32; w.P0 = 32; }
        else { w.src0 = AB(arena, n.pin); w.C0 = NBF[i - 1]; w.P0 = NBF[i - 1]; }
      } else {
        w.src0 = AB(arena, P->X[i]); w.C0 = n.in_prefix * f; w.P0 = P->PX[i];
        w.src1 = AB(arena, n.up); w.C1 = NBF[i + 1]; w.P1 = NBF[i + 1];
      }
      w.dy = b.dy; w.Cout = f; w.PY = f; w.dw = gsr + L.gs;
      wrdy[cv] = rdy;
      // dgrad
      if (cv == 0 && i == 0 && n.in_prefix == 0) continue;  // no gradient into the image
      nunet_conv_desc d; memset(&d, 0, sizeof(d));
      d.dtype = dt; d.N = c.N; d.H = H; d.W = W;
      d.src0 = b.dy; d.C0 = f; d.P0 = f;
      d.wpack = wpack + (size_t)L.wd * es;
      if (P->sk_floats[i] > 0) { d.splitk_ws = (float*)AB(arena, P->off_sk[i]); d.splitk_ws_floats = P->sk_floats[i]; }
      S.name("B%d%d.dgrad%d", n.i, n.j, cv + 1);
      if (cv == 1) {
        d.dst0 = AB(arena, P->off_da1[k]); d.D0 = f; d.Q0 = f;
        if (bnr_fuse) {
          const ConvL& L1 = n.c1;
          d.bn_y = AB(arena, n.y1); d.bn_py = f; d.bn_mean_invstd = save + L1.save;
          d.bn_gamma = params + L1.g_off; d.bn_beta = params + L1.be_off; d.bn_sums = bsums + L1.bsum * NUNET_BN_SUM_REPLICAS;
        }
        S.add(lane, 0, cost_conv(i, f, f), {rdy, bnr_fuse ? rb + B_Y1 : -1}, {rl + L_DA1, rsk, bnr_fuse ? R_GSV + 2 * k : -1}, [=](hipStream_t ls) { return nunet_conv3x3_fwd(&d, ls); });
      } else if (n.in_prefix == 0) {
        d.dst0 = AB(arena, P->off_gpin[k]); d.D0 = NBF[i - 1]; d.Q0 = NBF[i - 1];
        S.add(lane, 0, cost_conv(i, f, NBF[i - 1]) + (i >= 3 ? 12.f : 0.f), {rdy}, {rl + L_GPIN, rsk}, [=](hipStream_t ls) { return nunet_conv3x3_fwd(&d, ls); });
      } else {
        d.dst0 = AB(arena, P->GX[i]); d.D0 = n.in_prefix * f; d.Q0 = P->PX[i]; d.acc_slot_w = f;
        for (int q = 0; q < n.in_prefix; ++q) { if (written[i][q]) d.acc0_mask |= 1u << q; written[i][q] = true; }
        d.dst1 = AB(arena, P->off_gup[k]); d.D1 = NBF[i + 1]; d.Q1 = NBF[i + 1];
        S.add(lane, 0, cost_conv(i, f, n.in_prefix * f + NBF[i + 1]), {rdy},
              {R_GX + i * 5 + 0, n.in_prefix > 1 ? R_GX + i * 5 + 1 : -1, n.in_prefix > 2 ? R_GX + i * 5 + 2 : -1,
               n.in_prefix > 3 ? R_GX + i * 5 + 3 : -1, rl + L_GUP, rsk}, [=](hipStream_t ls) { return nunet_conv3x3_fwd(&d, ls); });
      }
      if (cv == 0) {
        if (n.in_prefix == 0) {
          // through MaxPool2d(2,2) into x_{i-1,0}
          const int acc = written[i - 1][0] ? 1 : 0;
          S.name("B%d%d.poolB", n.i, n.j);
          S.add(lane, 0, cost_mem(5.0, 2.5 * (double)P->px[i - 1] * NBF[i - 1] * es), {rl + L_GPIN, R_X + (i - 1) * 5 + 0}, {R_GX + (i - 1) * 5 + 0}, [=](hipStream_t ls) {
            return nunet_maxpool2x2_bwd(dt, c.N, P->hl[i - 1], P->wl[i - 1], NBF[i - 1], AB(arena, P->X[i - 1]), P->PX[i - 1],
                                        AB(arena, P->off_gpin[k]), NBF[i - 1], AB(arena, P->GX[i - 1]), P->PX[i - 1], acc, ls);
          });
          written[i - 1][0] = true;
        } else {
          // through the bilinear upsample into x_{i+1,up_slot}
          const int acc = written[i + 1][n.up_slot] ? 1 : 0;
          S.name("B%d%d.upB", n.i, n.j);
          S.add(lane, 0, cost_mem(8.0, 1.5 * (double)P->px[i] * NBF[i + 1] * es), {rl + L_GUP}, {R_GX + (i + 1) * 5 + n.up_slot}, [=](hipStream_t ls) {
            return nunet_upsample2x_bwd(dt, c.N, P->hl[i + 1], P->wl[i + 1], NBF[i + 1], AB(arena, P->off_gup[k]), NBF[i + 1],
                                        AB(arena, P->GX[i + 1] + (size_t)n.up_slot * NBF[i + 1] * es), P->PX[i + 1], acc, ls);
          });
          written[i + 1][n.up_slot] = true;
        }
      }
    }
    static int wg_pair = -1;
    if (wg_pair < 0) { const char* e = getenv("NUNET_WGRAD_PAIR"); wg_pair = e ? atoi(e) : 1; }
    if (wg_pair && rc == NUNET_OK) {
      // both weight gradients of the block in one launch (conv1's problem first: it may carry the
      // first layer's algorithmic Cin for the profiler)
      const nunet_wgrad_desc w0 = wdesc[0], w1 = wdesc[1];
      const int alg_cin = (i == 0 && n.in_prefix == 0) ? c.input_channels : 0;
      const float cw = cost_wg(i, (double)w0.C0 + w0.C1, f) + cost_wg(i, (double)w1.C0 + w1.C1, f) - 8.f;
      S.name("B%d%d.wgrad", n.i, n.j);
      int rx[4] = {-1, -1, -1, -1}, r_in = -1, r_up = -1;
      if (n.in_prefix == 0) r_in = (i == 0 ? R_IMG : rb + B_PIN);
      else { for (int q = 0; q < n.in_prefix && q < 4; ++q) rx[q] = R_X + i * 5 + q; r_up = rb + B_UP; }
      S.add(wlane, 1, cw, {rb + B_A1, wrdy[1], wrdy[0], r_in, rx[0], rx[1], rx[2], rx[3], r_up}, {R_GSW + 2 * k, R_GSW + 2 * k + 1},
            [=](hipStream_t ls) { g_prof_alg_cin = alg_cin; int r = nunet_conv3x3_wgrad_pair(&w0, &w1, ls); g_prof_alg_cin = 0; return r; });
    }
    for (int cv = 1; cv >= 0 && rc == NUNET_OK && !wg_pair; --cv) {
      const int cidx = 2 * k + cv;
      const nunet_wgrad_desc w = wdesc[cv];
      const int alg_cin = (cv == 0 && i == 0 && n.in_prefix == 0) ? c.input_channels : 0;
      auto fn = [=](hipStream_t ls) { g_prof_alg_cin = alg_cin; int r = nunet_conv3x3_wgrad(&w, ls); g_prof_alg_cin = 0; return r; };
      const float cw = cost_wg(i, (double)w.C0 + w.C1, f);
      S.name("B%d%d.wgrad%d", n.i, n.j, cv + 1);
      if (cv == 1) S.add(wlane, 1, cw, {rb + B_A1, wrdy[cv]}, {R_GSW + cidx}, fn);
      else if (n.in_prefix == 0) S.add(wlane, 1, cw, {i == 0 ? R_IMG : rb + B_PIN, wrdy[cv]}, {R_GSW + cidx}, fn);
      else S.add(wlane, 1, cw, {R_X + i * 5 + 0, n.in_prefix > 1 ? R_X + i * 5 + 1 : -1, n.in_prefix > 2 ? R_X + i * 5 + 2 : -1,
                                n.in_prefix > 3 ? R_X + i * 5 + 3 : -1, rb + B_UP, wrdy[cv]}, {R_GSW + cidx}, fn);
    }
    // "bucket 0 complete" (data-parallel exchange beside the rest of the backward pass, nunet_plan_bucket0_*): an empty op
    // on the otherwise unused lane 4 that reads every gradient resource of the phase-1 nodes and the heads, then records
    // the plan's event there - as an external event record node when the pass is being captured into a graph
    if (k == k_split && (phases & 3) == 3 && rt_of(P)->b0_enabled && !P->cfg.unet && S.multi) {
      std::vector<int> rs;
      for (int kk = k_split; kk < nnodes; ++kk) { rs.push_back(R_GSW + 2 * kk); rs.push_back(R_GSW + 2 * kk + 1); rs.push_back(R_GSV + 2 * kk); rs.push_back(R_GSV + 2 * kk + 1); }
      for (size_t h = 0; h < P->heads.size(); ++h) rs.push_back(R_GSV + 30 + (int)h);
      hipEvent_t ev = rt_of(P)->b0_event;
      for (size_t q = 0; q < rs.size(); q += 10) {
        const bool last = q + 10 >= rs.size();
        S.name("b0rdy");
        S.add_v(4, 1, 0.f, rs.data() + q, (int)std::min<size_t>(10, rs.size() - q), [=](hipStream_t ls) {
          if (!last) return (int)NUNET_OK;
          hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
          (void)hipStreamIsCapturing(ls, &cs);
          if (cs == hipStreamCaptureStatusActive) return graph_record_external(ls, ev);
          const hipError_t e = hipEventRecord(ev, ls);
          if (e != hipSuccess) { nunet_set_error("plan_backward: bucket-0 event record: %s", hipGetErrorString(e)); return (int)NUNET_ELAUNCH; }
          return (int)NUNET_OK;
        });
      }
    }
  }
  if (rc == NUNET_OK) rc = S.run_ops();
  S.join();
  if (rc == NUNET_OK && S.failed) { nunet_set_error("plan_backward: capture lane pool exhausted"); rc = NUNET_EINVAL; }
  if (rc) return rc;
  if (!(phases & 4)) return NUNET_OK;
  P->utab.accumulate = accumulate;
  int gx = (int)ceil_div64(P->unpack_maxn, 256 * 4);
  if (gx > 512) gx = 512;
  ProfScope ps(PC_UNPACK, 0, (double)P->nparams * (accumulate ? 12 : 8), st);
  static int tiled = -1;
  if (tiled < 0) { const char* e = getenv("NUNET_UNPACK_TILED"); tiled = e ? atoi(e) : 1; }
  if (tiled) {
    PackTab& tab = P->ptab;
    int nt = 0;
    for (int i = 0; i < tab.n; ++i) { tab.tile0[i] = nt; nt += ((tab.e[i].cout + 31) / 32) * ((tab.e[i].cinpad + 31) / 32); }
    tab.tile0[tab.n] = nt; tab.ntiles = nt;
  }
  if (tiled) hipLaunchKernelGGL(unpack_tiled_kernel, dim3(P->ptab.ntiles + P->utab.n), dim3(256), 0, st, gsr, grads, P->ptab, P->utab);
  else hipLaunchKernelGGL(unpack_kernel, dim3(gx, P->utab.n), dim3(256), 0, st, gsr, grads, P->utab);
  return nunet_check_launch("unpack_grads");
}

// Data-parallel exchange beside the backward pass. After nunet_plan_bucket0_enable(P, 1), a backward call that runs phases 1 and 2
// together records the plan's event once every gradient of the first bucket (nunet_plan_grad_scratch) is complete;
// nunet_plan_bucket0_wait makes `s` wait for the most recent such record (call it after launching the pass or the graph holding it).
extern "C" int nunet_plan_bucket0_enable(nunet_plan* P, int32_t on) {
  NUNET_REQUIRE(P, "plan_bucket0_enable: null plan");
  PlanRt* rt = rt_of(P);
  if (on && !rt->b0_event && hipEventCreateWithFlags(&rt->b0_event, hipEventDisableTiming) != hipSuccess) {
    nunet_set_error("plan_bucket0_enable: %s", hipGetErrorString(hipGetLastError())); return NUNET_ELAUNCH;
  }
  rt->b0_enabled = on != 0 && !P->cfg.unet;
  return rt->b0_enabled ? 1 : 0;     // 1: armed; 0: not available for this plan (callers exchange after the pass)
}
extern "C" int nunet_plan_bucket0_wait(nunet_plan* P, nunet_stream_t s) {
  NUNET_REQUIRE(P && rt_of(P)->b0_event && rt_of(P)->b0_enabled, "plan_bucket0_wait: not enabled");
  if (hipStreamWaitEvent((hipStream_t)s, rt_of(P)->b0_event, 0) != hipSuccess) { nunet_set_error("plan_bucket0_wait: %s", hipGetErrorString(hipGetLastError())); return NUNET_ELAUNCH; }
  return NUNET_OK;
}

extern "C" int nunet_plan_stamps_read(nunet_plan* P, int32_t pass, uint64_t* ticks, int32_t cap, int32_t* n_out, char* labels, int32_t label_bytes) {
  NUNET_REQUIRE(P && ticks && n_out && labels && (pass == 0 || pass == 1) && label_bytes > 0, "plan_stamps_read: bad args");
  PlanRt* rt = rt_of(P);
  *n_out = 0; labels[0] = 0;
  if (!rt->stamps) return NUNET_OK;                    // NUNET_STAMPS not set
  const std::vector<std::string>& lab = rt->stamp_labels[pass];
  int n = (int)lab.size(); if (n > cap) n = cap;
  if (hipDeviceSynchronize() != hipSuccess ||
      hipMemcpy(ticks, rt->stamps + (size_t)pass * STAMP_CAP, (size_t)n * 8, hipMemcpyDeviceToHost) != hipSuccess) {
    nunet_set_error("plan_stamps_read: copy failed"); return NUNET_ELAUNCH;
  }
  size_t o = 0;
  for (int k = 0; k < n; ++k) {
    if (o + lab[k].size() + 2 > (size_t)label_bytes) break;
    memcpy(labels + o, lab[k].c_str(), lab[k].size()); o += lab[k].size(); labels[o++] = '\n';
  }
  labels[o] = 0;
  *n_out = n;
  return NUNET_OK;
}

extern "C" int nunet_plan_set_lanes(nunet_plan* P, nunet_stream_t* lanes, int32_t n) {
  NUNET_REQUIRE(P && lanes && n >= 1, "plan_set_lanes: bad args");
  PlanRt* rt = rt_of(P);
  for (int l = 0; l < NLANES; ++l) rt->lanes[l] = (hipStream_t)lanes[l % n];   // caller-owned streams
  rt->lanes_ok = true;
  rt->lanes_external = true;
  return NUNET_OK;
}

extern "C" int nunet_plan_set_multistream(nunet_plan* P, int32_t enable) {
  NUNET_REQUIRE(P, "plan_set_multistream: null plan");
  rt_of(P)->multistream = enable;
  return NUNET_OK;
}
