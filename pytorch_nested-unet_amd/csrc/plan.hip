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
#include <algorithm>

#include <map>
#include <string>
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
struct UpdP { float* params; float* mom; const float* scratch; float* grads; const float* lr; float momc, wd, gscale; int nesterov; int nconv;
              int bid_off; };   // first block's index in the whole-model block numbering (a launch may cover one VGGBlock's tiles, or the heads)

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
  const int bid = (int)blockIdx.x + u.bid_off;
  if (bid >= tab.ntiles) {
    // ---- 1x1 head: sum the gradient slabs (fixed order), then the step; <= 264 elements
    const UnpackEnt en = ut.e[u.nconv + bid - tab.ntiles];
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
  while (e + 1 < tab.n && bid >= tab.tile0[e + 1]) ++e;
  const PackEnt en = tab.e[e];
  const UnpackEnt ue = ut.e[e];
  const int t = bid - tab.tile0[e];
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

template <typename T> static int launch_pack(const float* params, void* arena_t, PackTab& tab, long long maxn, hipStream_t st) {
  (void)maxn;
  int nt = 0;
  for (int i = 0; i < tab.n; ++i) { tab.tile0[i] = nt; nt += ((tab.e[i].cout + 31) / 32) * ((tab.e[i].cinpad + 31) / 32); }
  tab.tile0[tab.n] = nt; tab.ntiles = nt;
  double pb = 0;
  for (int i = 0; i < tab.n; ++i) pb += 9.0 * tab.e[i].cout * tab.e[i].cin * (4 + (tab.e[i].wd >= 0 ? 2 : 1) * sizeof(T));
  ProfScope ps(PC_PACK, 0, pb, st);
  NUNET_LAUNCH((pack_kernel<T>), dim3(nt), dim3(256), 0, st, params, (T*)arena_t, tab);
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
  NUNET_LAUNCH(unpack_kernel, dim3(gx, 1), dim3(256), 0, (hipStream_t)s, dw, g, tab);
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
  long long stats, save, bsum;            // floats into the fp32 small-vector regions (x REPLICAS x 2 int64 words in the fixed-point regions)
  long long slab; int ks, wg_target;      // weight-gradient K-split slabs: floats into the slab region, slices, workgroup target
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
  int wave;                                // 1: single-stream schedule with grouped launches (Sched::run_wave) instead of lanes
  bool lanes_external;
  std::vector<hipStream_t> cap_streams;   // never-reused streams for capture-time lane continuation
  size_t cap_next;
  bool bwd_written[5][5]; int bwd_pp[5];   // state carried between backward phases
  // NUNET_STAMPS=1 diagnostic: a 1-thread kernel after every scheduled op writes the 100 MHz wall clock,
  // so the real timeline of an (unprofiled) hipGraph replay can be read back (tools/stamp_timeline.py)
  unsigned long long* stamps;              // device, [2][STAMP_CAP]
  // nunet_plan_set_inpass_update: the optimiser step of every VGGBlock as an op of the backward pass (params == NULL: off)
  struct { float* params; float* mom; const float* lr; float momc, wd, gscale; int nesterov; float* grads; } upd;
  int lane_low_priority;                   // side lanes of the flag-synchronised program at the lowest stream priority (default 1)
  int calibrating;                         // nunet_plan_calibrate: single lane + stamps, to measure every op's isolated cost
  std::map<std::string, float> op_cost[2]; // measured cost (us) by op name, per pass; empty: the built-in estimates
  hipEvent_t b0_event;                     // recorded when the first gradient bucket (phase-1 nodes + heads) is complete
  bool b0_enabled;
  int seg_lanes_distinct;                  // how many of them were measured to run beside the caller's stream and each other
  hipStream_t seg_lanes[3];                // side-lane streams of the segmented recording (created together: distinct hardware queues)
  std::vector<hipStream_t> seg_owned;      // every stream seg_pick_lanes created and kept (destroyed with the plan)
  struct Sched* open_sched;                // backward pass left open after phase 1 (nunet_plan_backward_phase bit 3): lanes, dependency state
  std::vector<hipEvent_t> b0_events;       // ... and the last-writer events of the first bucket's gradients, for nunet_plan_bucket0_wait
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
  size_t off_fx, stats_floats;  // fixed-point per-channel sums: [BatchNorm statistics | BatchNorm-backward sums], zeroed by every training forward
  size_t off_gs, gs_floats;     // reduced gradient scratch (native layout, gradient-ready order): what a data-parallel job exchanges
  size_t off_slab; long long slab_floats;   // K-split slabs of the weight gradients (summed into the scratch by reduce_kernel)
  size_t off_save;
  size_t off_wpack; long long wpack_elems;
  size_t off_img;
  size_t X[5], GX[5];
  size_t off_dy[16][2], off_da1[16], off_gup[16], off_gpin[16];   // per-BLOCK backward scratch (dY ping-pong), so blocks of one level can run on different lanes
  size_t off_sk[16]; long long sk_floats[16];   // per-BLOCK fp32 K-split slabs (blocks of the grid-starved levels; 0: none)   // per-level backward scratch (dY ping-pong)
  struct PlanRt* rt;
  size_t total;
  PackTab ptab; long long pack_maxn;
  PackTab ptab_lvl[5]; long long pack_maxn_lvl[5];   // the same entries grouped by pyramid level (issued per lane)
  UnpackTab utab; long long unpack_maxn;
};

static size_t fx_region_bytes(const nunet_plan* P) { return P->stats_floats * NUNET_BN_SUM_REPLICAS * NUNET_FX_WORDS * sizeof(long long); }
static long long* fx_of(void* arena, const nunet_plan* P, int region, long long off) {   // region 0: BN statistics, 1: BN-backward sums
  return (long long*)((char*)arena + P->off_fx + (size_t)region * fx_region_bytes(P)) + off * NUNET_BN_SUM_REPLICAS * NUNET_FX_WORDS;
}
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
  long long slab = 0;
  const int ks_max = 0;
  for (int k = (int)P->exec.size() - 1; k >= 0; --k) {
    Node& n = P->exec[k];
    for (int cv = 1; cv >= 0; --cv) {
      ConvL& c = cv ? n.c2 : n.c1;
      c.gs = gs; gs += 9LL * c.cout * c.cinpad + 3LL * c.cout;
      // K-split of the weight gradient: the two problems of a block share one launch, the one with fewer input
      // channels takes half the workgroups (fewer, fatter slices: less slab traffic)
      const ConvL& o = cv ? n.c1 : n.c2;
      // (96x96 bs16, measured: 64/128, 64/256, 128/512, 256/512 are all slower in the step. 256x256 bs32: twice the workgroups are
      //  24 % faster alone - 4.34 -> 3.29 ms over the 15 launches - and +2.3 % on the step: with 7 x the pixels per slice the
      //  slices are long enough to amortise a workgroup's start and the K-split slabs stay small beside the tensors read.
      //  Re-measured under the list-scheduled executor at 96x96 bs16: 64/128, 80/160, 96/192, 128/256, 160/320 = 9242, 9277, 9211-9357,
      //  9146-9247, 9107 images/s: a little fewer workgroups leave the chain's kernels more room: 96/192.)
      const bool big = (long long)cfg->N * cfg->H * cfg->W >= (1LL << 20);
      c.wg_target = big ? (c.cinpad < o.cinpad ? 256 : 512) : (c.cinpad < o.cinpad ? 96 : 192);
      nunet_wgrad_desc wd; memset(&wd, 0, sizeof(wd));
      wd.N = cfg->N; wd.H = P->hl[n.i]; wd.W = P->wl[n.i]; wd.C0 = c.cinpad; wd.Cout = c.cout; wd.target_wgs = c.wg_target; wd.max_slabs = ks_max;
      c.ks = nunet_conv3x3_wgrad_slabs(&wd);
      c.slab = slab; slab += (long long)c.ks * 9LL * c.cout * c.cinpad;
    }
    if (k == (int)P->exec.size() - P->first_phase_nodes) P->gs_bucket0 = gs;
  }
  P->nparams = po; P->nbnbuf = bo; P->nbn = bn;
  P->wpack_elems = wp;

  // ---- arena ---------------------------------------------------------------------
  size_t cur = 0;
  P->stats_floats = (size_t)sv;
  P->off_fx = bump(cur, 2 * fx_region_bytes(P));   // [forward statistics | backward sums], replicated fixed-point accumulators
  P->gs_floats = (size_t)gs;
  P->off_gs = bump(cur, P->gs_floats * 4);
  P->slab_floats = slab;
  P->off_slab = bump(cur, (size_t)slab * 4);
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
  // K-split slabs for the blocks of the grid-starved levels (up to 8 slices of the block's widest conv output, the input
  // gradient of conv1); deterministic (fixed summation order). One buffer per BLOCK: the four convs of a block follow each
  // other anyway, and blocks of one level on different lanes must not serialise on a shared scratch.
#ifndef NUNET_SK_MINLEV
#define NUNET_SK_MINLEV 3
#endif
  for (size_t k = 0; k < P->exec.size() && k < 16; ++k) {
    const Node& n = P->exec[k];
    const int maxc = n.i < NUNET_SK_MINLEV ? 0 : (n.c1.cinpad > n.c1.cout ? n.c1.cinpad : n.c1.cout);
    P->sk_floats[k] = 8LL * P->px[n.i] * maxc;   // up to 8 slabs
    P->off_sk[k] = bump(cur, (size_t)P->sk_floats[k] * 4 + 16);
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
  rt->stamps = nullptr;
  rt->calibrating = 0;
  rt->lane_low_priority = 1;
  memset(&rt->upd, 0, sizeof(rt->upd));
  { PackTab& tab = P->ptab; int nt = 0; for (int i = 0; i < tab.n; ++i) { tab.tile0[i] = nt; nt += ((tab.e[i].cout + 31) / 32) * ((tab.e[i].cinpad + 31) / 32); } tab.tile0[tab.n] = nt; tab.ntiles = nt; }
  rt->b0_event = nullptr; rt->b0_enabled = false; rt->open_sched = nullptr;
  rt->seg_lanes[0] = rt->seg_lanes[1] = rt->seg_lanes[2] = nullptr;
  rt->events_used[0] = rt->events_used[1] = 0;
  { const char* e = getenv("NUNET_MULTISTREAM"); rt->multistream = e ? atoi(e) : 1; }
  rt->wave = 0;
  for (int l = 0; l < NLANES; ++l) {
    rt->lanes[l] = nullptr;
    if (hipStreamCreateWithFlags(&rt->lanes[l], hipStreamNonBlocking) != hipSuccess) { rt->lanes_ok = false; (void)hipGetLastError(); }
  }
  P->rt = rt;
  return P;
}


static PlanRt* rt_of(nunet_plan* P) { return P->rt; }

static void sched_free(struct Sched* s);
extern "C" void nunet_plan_destroy(nunet_plan* p) {
  if (!p) return;
  if (p->rt) {
    if (p->rt->open_sched) { sched_free(p->rt->open_sched); p->rt->open_sched = nullptr; }
    if (!p->rt->lanes_external)
      for (int l = 0; l < NLANES; ++l) if (p->rt->lanes[l]) (void)hipStreamDestroy(p->rt->lanes[l]);
    for (size_t k = 0; k < p->rt->cap_streams.size(); ++k) (void)hipStreamDestroy(p->rt->cap_streams[k]);
    for (size_t k = 0; k < p->rt->seg_owned.size(); ++k) (void)hipStreamDestroy(p->rt->seg_owned[k]);
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
#define NRES 512
enum { R_X = 0, R_GX = 25, R_BLK = 50, R_LVL = 330, R_IMG = 230, R_LOGITS = 231, R_DLOGITS = 232, R_WP = 233, R_GS = 238, R_SK = 470, R_GSW = 240, R_GSV = 280, R_PRM = 485 };
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
    NUNET_LAUNCH(stamp_kernel, dim3(1), dim3(1), 0, st, rt->stamps + (size_t)pass * STAMP_CAP + lab.size());
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
    else failed = true;        // a dropped wait would be a silent race between lanes: reported by the caller after the join
  }
  // ROCm 7.2: inside a stream capture, a stream that waits on an event DESCENDING from its
  // own tail node crashes hipStreamEndCapture (regression test: tests/test_capture_gpu.py). The x_{i,j}
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
    for (int x : rd) if (x >= 0) { if (nr < 16) r[nr++] = x; else failed = true; }
    for (int x : wr) if (x >= 0) { if (nw < 16) w[nw++] = x; else failed = true; }
    return begin_v(lane_map[lane], r, nr, w, nw);
  }
  // stream waits / event records: real HIP calls, or instructions of the segmented program being recorded (graph.hip)
  static void wait_on(hipStream_t st, hipEvent_t ev) { if (!seg_wait(st, ev)) (void)hipStreamWaitEvent(st, ev, 0); }
  static void record_on(hipEvent_t ev, hipStream_t st) { if (!seg_record(ev, st)) (void)hipEventRecord(ev, st); }
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
      if (st != main_s) wait_on(st, fork_ev);
    }
    for (int k = 0; k < npend; ++k) wait_on(st, pend[k]);
    seg_touch(st);                       // (segmented recording: the op's launches open / continue this stream's segment)
    return st;
  }

  // ---- deferred ops: the backward pass collects its ops first (descriptors captured by value), then issues them
  // kind: what the single-stream schedule (run_wave) may group into one launch; the lane schedule runs every op through fn
  enum { K_GEN = 0, K_CONV = 1 };
  struct Op { int lane, leaf; float cost; int nrd, nwr; int rd[12], wr[8]; char name[32]; std::function<int(hipStream_t)> fn;
              int kind; nunet_conv_desc cd; int alg_cin; };
  std::vector<Op> ops;
  // a 3x3 convolution (forward or input gradient): grouped with other ready convolutions of the same kernel variant by run_wave
  void add_conv(int lane, std::initializer_list<int> rd, std::initializer_list<int> wr, const nunet_conv_desc& d, int alg_cin = 0) {
    const double px = (double)d.N * d.H * d.W;
    add(lane, 0, 6.f + (float)(2.0 * 9 * (d.C0 + d.C1) * (d.D0 + d.D1) * px / 4e8), rd, wr, [d, alg_cin](hipStream_t ls) {
      g_prof_alg_cin = alg_cin; const int r = nunet_conv3x3_fwd(&d, ls); g_prof_alg_cin = 0; return r; });
    ops.back().kind = K_CONV; ops.back().cd = d; ops.back().alg_cin = alg_cin;
  }
  void add(int lane, int leaf, float cost, std::initializer_list<int> rd, std::initializer_list<int> wr, std::function<int(hipStream_t)> fn) {
    Op o; o.lane = lane_map[lane]; o.leaf = leaf; o.cost = cost; o.nrd = o.nwr = 0; o.kind = K_GEN; o.alg_cin = 0;
    for (int x : rd) if (x >= 0) { if (o.nrd < 12) o.rd[o.nrd++] = x; else failed = true; }
    for (int x : wr) if (x >= 0) { if (o.nwr < 8) o.wr[o.nwr++] = x; else failed = true; }
    memcpy(o.name, cur_name, sizeof(o.name)); cur_name[0] = 0;
    measured_cost(o);
    o.fn = std::move(fn);
    ops.push_back(std::move(o));
  }
  // the op's isolated cost as nunet_plan_calibrate measured it (by name), instead of the built-in estimate
  void measured_cost(Op& o) {
    const std::map<std::string, float>& m = rt->op_cost[pass & 1];
    if (m.empty() || !o.name[0]) return;
    const auto it = m.find(o.name);
    if (it != m.end()) o.cost = it->second;
  }
  void add_v(int lane, int leaf, float cost, const int* rd, int nrd, std::function<int(hipStream_t)> fn) {
    Op o; o.lane = lane_map[lane]; o.leaf = leaf; o.cost = cost; o.nrd = o.nwr = 0; o.kind = K_GEN; o.alg_cin = 0;
    for (int q = 0; q < nrd; ++q) if (rd[q] >= 0) { if (o.nrd < 12) o.rd[o.nrd++] = rd[q]; else failed = true; }
    memcpy(o.name, cur_name, sizeof(o.name)); cur_name[0] = 0;
    measured_cost(o);
    o.fn = std::move(fn);
    ops.push_back(std::move(o));
  }
  int run_ops();
  int run_wave();
  int run_list();
  bool wave;                           // single-stream schedule: no lanes, no events; run() picks
  bool list;                           // lanes assigned by a list scheduler over the dependency graph (run_list) instead of by block
  int run() { return wave ? run_wave() : (list && multi) ? run_list() : run_ops(); }
  void end() {
    if (!multi) { stamp(main_s, 0); cur_name[0] = 0; return; }
    hipStream_t st = lane_s[cur_lane];
    if (!st) return;
    stamp(st, cur_lane); cur_name[0] = 0;
    if (capturing) graph_tag_tail(st, cur_lane);
    hipEvent_t ev = new_event();
    record_on(ev, st);
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
      if (used[l] && lane_tail[l] && lane_s[l] != main_s) wait_on(main_s, lane_tail[l]);
    seg_touch(main_s);                   // what the caller launches next on its stream opens a new segment
  }
};

static void sched_free(Sched* s) { delete s; }

// Side-lane streams of the segmented program. ROCm maps every stream onto one of 4 hardware queues at creation and offers no way
// to ask which (tools/seg_overlap_probe.py: of 8 fresh streams some pairs share a queue, and two lanes that share a queue run
// strictly one after the other). So the lanes are CHOSEN BY MEASUREMENT: a 100 us single-workgroup spin kernel on two streams at
// once takes 100 us when they sit on different queues and 200 us when not. Three candidates that overlap with the caller's
// stream and with each other become lanes 1-3 (the rest are destroyed). Called outside any capture, once per plan.
int nunet_debug_spin(int32_t us, int32_t tag, nunet_stream_t s);
static void seg_pick_lanes(PlanRt* rt, hipStream_t main_s) {
  const int NC = 12, SPIN_US = 100;
  hipStream_t cand[NC];
  int nc = 0;
  // the side lanes get the LOWEST stream priority: the chain lane's workgroups are dispatched ahead of theirs (+1.4 % on the
  // flag-synchronised step) - unless the caller says otherwise (nunet_plan_set_lane_priority): with RCCL's high-priority stream in
  // the process, lowest-priority lanes are served in time slices (every kernel on one of them took 100-190 us in the
  // data-parallel rehearsal: 3.4-4.4 ms per step against 1.82 at default priority). NUNET_SIDE_PRIO overrides both.
  int side_prio = rt->lane_low_priority;
  { const char* e = getenv("NUNET_SIDE_PRIO"); if (e) side_prio = atoi(e); }
  int pr_least = 0, pr_greatest = 0;
  (void)hipDeviceGetStreamPriorityRange(&pr_least, &pr_greatest);
  // (measured and not kept: side lanes on CU-masked streams - hipExtStreamCreateWithCUMask, 16 to 96 CUs kept free for the chain -
  //  and the chain on a high-priority stream both run the step at 5.3 ms instead of 1.85)
  for (int k = 0; k < NC; ++k) {
    const hipError_t ce = side_prio ? hipStreamCreateWithPriority(&cand[nc], hipStreamNonBlocking, pr_least) : hipStreamCreateWithFlags(&cand[nc], hipStreamNonBlocking);
    if (ce == hipSuccess) ++nc; else (void)hipGetLastError();
  }
  hipEvent_t e0 = nullptr, e1 = nullptr, eb = nullptr;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventCreateWithFlags(&eb, hipEventDisableTiming);
  auto overlap = [&](hipStream_t a, hipStream_t b) {          // true: a and b run side by side
    (void)hipStreamSynchronize(a); (void)hipStreamSynchronize(b);
    (void)hipEventRecord(e0, a);
    (void)nunet_debug_spin(SPIN_US, 1, a);
    (void)nunet_debug_spin(SPIN_US, 1, b);
    (void)hipEventRecord(eb, b);
    (void)hipStreamWaitEvent(a, eb, 0);
    (void)hipEventRecord(e1, a);
    (void)hipStreamSynchronize(a);
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e0, e1) != hipSuccess) { (void)hipGetLastError(); return false; }
    return ms * 1000.f < 1.5f * SPIN_US;
  };
  // A lane must also DISPATCH at the caller's stream's rate: with RCCL initialised in the process, some streams sit on queues that
  // the hardware scheduler serves in time slices - every kernel on such a lane took 100-190 us in the data-parallel rehearsal
  // (4.1 ms per step). Eight back-to-back 2 us kernels, timed on the candidate against the same on the caller's stream.
  auto chain_us = [&](hipStream_t a) {
    (void)hipStreamSynchronize(a);
    (void)nunet_debug_spin(2, 1, a);
    (void)hipEventRecord(e0, a);
    for (int q = 0; q < 8; ++q) (void)nunet_debug_spin(2, 1, a);
    (void)hipEventRecord(e1, a);
    (void)hipStreamSynchronize(a);
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e0, e1) != hipSuccess) { (void)hipGetLastError(); return 1e9f; }
    return ms * 1000.f;
  };
  const float base_us = std::min(chain_us(main_s), chain_us(main_s));
  hipStream_t pick[3]; int np = 0;
  bool usedc[NC] = {false};
  for (int k = 0; k < nc && np < 3; ++k) {
    // (also clear of the process's default stream: the copies of the next batch are usually queued there)
    bool ok = overlap(main_s, cand[k]) && overlap((hipStream_t)nullptr, cand[k]);
    if (ok) { const float c_us = std::min(chain_us(cand[k]), chain_us(cand[k])); ok = c_us < 2.5f * base_us + 20.f; }
    for (int q = 0; q < np && ok; ++q) ok = overlap(pick[q], cand[k]);
    if (ok) { pick[np++] = cand[k]; usedc[k] = true; }
  }
  for (int k = 0; k < nc; ++k) { if (!usedc[k]) (void)hipStreamDestroy(cand[k]); else rt->seg_owned.push_back(cand[k]); }
  // (fewer than three distinct queues found: lanes share a STREAM - never two streams of one queue, which the flag-synchronised
  //  program could not survive: a polling kernel ahead of its signal in the same queue)
  for (int q = 0; q < 3; ++q) rt->seg_lanes[q] = q < np ? pick[q] : (np > 0 ? pick[q % np] : main_s);
  rt->seg_lanes_distinct = np;
  if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); if (eb) (void)hipEventDestroy(eb);
}

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
  multi = rt->multistream != 0 && rt->lanes_ok && !rt->calibrating;
  wave = rt->wave == 1 && !rt->calibrating;
  list = rt->wave == 2 && !rt->calibrating;
  if (wave) multi = false;             // one stream: no lanes to fork, no events
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
    if (seg_active()) capturing = false;   // segmented recording: fixed lanes on real streams (the caller's stream is merely inside a segment)
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
  // lane of (level 0..4, weight gradients of level 0..4): one lane per pyramid level, weight gradients on their
  // block's lane (measured best on MI355X: separate weight-gradient lanes add cross-queue edges that cost more than the
  // overlap they buy; the exception is the deferral of the shallow chain blocks' weight gradients, see the backward pass)
  for (int l = 0; l < NLANES; ++l) { lane_map[l] = l % 5; lane_s[l] = capturing ? nullptr : rt->lanes[l]; used[l] = false; lane_tail[l] = nullptr; }
  if (multi && seg_active()) {
    // four real streams = ROCm's four hardware queues: the critical chain (lane 0) runs on the caller's stream itself - no
    // fork / join edge on the chain -, the side lanes on three streams of the plan's, the deferred weight gradients share lane 3's
    lane_s[0] = main_s; used[0] = true;
    if (!rt->seg_lanes[0]) {
      // (the dry pass of a recording comes first and captures nothing: the only moment the calibration kernels may run on `s`)
      hipStreamCaptureStatus cs2 = hipStreamCaptureStatusNone;
      (void)hipStreamIsCapturing(main_s, &cs2);
      if (cs2 != hipStreamCaptureStatusActive) { const bool keep = g_dry_run; g_dry_run = false; seg_pick_lanes(rt, main_s); g_dry_run = keep; }
      else for (int q = 0; q < 3; ++q) rt->seg_lanes[q] = rt->lanes[q + 1];
    }
    for (int q = 0; q < 3; ++q) lane_s[q + 1] = rt->seg_lanes[q];
    // (the deferred weight gradients of the chain blocks go to the stream of lane 1: its own block, x0_1, is the LAST of the
    //  level-0 row's backward chain B04 -> B03 -> B02 -> B01, so that stream idles for the first 500 us of the backward pass; on
    //  lane 3's stream they sat in front of B03 and held the whole row back by 280 us)
    lane_s[4] = lane_s[1];
  }
  if (multi) {
    fork_ev = new_event();
    record_on(fork_ev, main_s);
    if (capturing) graph_tag_tail(main_s, -1);   // launches made so far on the caller's stream are not ours to place
  }
}

int Sched::run_ops() {
  // ops are issued in program order (a simulated list schedule and a rewrite of the captured graph's edge order were
  // both measured slower than the plain capture order on ROCm 7.2, see DESIGN.md, and are not kept)
  int rc = NUNET_OK;
  for (size_t q = 0; q < ops.size() && rc == NUNET_OK; ++q) {
    Op& o = ops[q];
    hipStream_t st = begin_v(o.lane, o.rd, o.nrd, o.wr, o.nwr);
    rc = o.fn(st);
    memcpy(cur_name, o.name, sizeof(cur_name));
    end();
  }
  ops.clear();
  return rc;
}

// List schedule: the lane of every op is CHOSEN here instead of following its block. The hazards of the program order (read after
// write, write after write - the accumulation order into the shared gradient slots -, write after read) are the edges of a DAG;
// ops are taken by critical-path priority (cost + longest path to the end) and each goes to the lane on which it can start
// earliest in a simulation with the ops' cost estimates (a dependency that crosses lanes costs XSYNC), then issued in simulated
// start order. Any topological order of that DAG keeps every conflicting pair in program order, so the resource tracker sees
// the same hazards and the results are bit-identical to the block-lane schedule (tests assert it). Meant for lanes that are
// real in-order streams (the flag-synchronised program), where the issue order on a lane IS its execution order: the
// block-lane order put leaf work (weight gradients) in front of critical ops of the same lane.
int Sched::run_list() {
  const int n = (int)ops.size();
  constexpr int NLMAX = 4;
  // three lanes: measured on MI355X (96x96 bs16, flag-synchronised lanes) 2 / 3 / 4 lanes = 8640 / 9270 / 8950 images/s - the fourth
  // concurrent stream costs the critical chain more than its overlap buys; limiting the summed chip share of the concurrent ops
  // instead (a capacity model over the launch grids) only lost: 1.5 / 2.0 / 2.5 / 3.0 full-chip ops at once = 6610 / 8480 / 8720 /
  // 9260. A crossing dependency costs one sync kernel on each side (XSYNC; 0 / 3 / 10 us in the model: no difference).
  static int NLs[2] = {0, 0}; static float XSYNC = 3.f;
  if (!NLs[0]) {
    const char* e = getenv("NUNET_LIST_LANES"); NLs[0] = NLs[1] = e ? atoi(e) : 3;
    const char* f = getenv("NUNET_LIST_LANES_FWD"); if (f) NLs[0] = atoi(f);
    for (int q = 0; q < 2; ++q) if (NLs[q] < 1 || NLs[q] > NLMAX) NLs[q] = 3;
  }
  const int NL = NLs[pass & 1];
  std::vector<std::vector<int>> succ(n), pred(n);
  std::vector<int> indeg(n, 0);
  {
    std::vector<int> lastw(NRES, -1);
    std::vector<std::vector<int>> readers(NRES);
    auto edge = [&](int a, int b) { if (a >= 0 && a != b && std::find(succ[a].begin(), succ[a].end(), b) == succ[a].end()) { succ[a].push_back(b); pred[b].push_back(a); ++indeg[b]; } };
    for (int i = 0; i < n; ++i) {
      const Op& o = ops[i];
      for (int q = 0; q < o.nrd; ++q) edge(lastw[o.rd[q]], i);
      for (int q = 0; q < o.nwr; ++q) { edge(lastw[o.wr[q]], i); for (int r : readers[o.wr[q]]) edge(r, i); }
      for (int q = 0; q < o.nrd; ++q) readers[o.rd[q]].push_back(i);
      for (int q = 0; q < o.nwr; ++q) { lastw[o.wr[q]] = i; readers[o.wr[q]].clear(); }
    }
  }
  std::vector<float> prio(n, 0.f), tend(n, 0.f);
  for (int i = n - 1; i >= 0; --i) { float m = 0.f; for (int s : succ[i]) m = std::max(m, prio[s]); prio[i] = ops[i].cost + m; }
  // (measured, no effect or worse: a priority bias for the weight gradients (-100 / 0: the same, +300 us: -3 %), the cost of the
  //  grid-starved levels 2-4 scaled by 1.5 / 2 / 3 as a model of their in-step inflation (-0.5 / -2 / -2 %): the isolated costs
  //  schedule best)
  std::vector<int> ready, lane(n, 0), order;
  float lane_free[NLMAX] = {0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < n; ++i) if (indeg[i] == 0) ready.push_back(i);
  // event-driven: the lane that falls idle first takes the most urgent op that could start on it by then; when nothing could, the
  // lane's clock moves on to the next moment something can (so leaf work fills the holes of a lane instead of queueing at its end)
  auto est = [&](int p, int l) { float t = 0.f; for (int q : pred[p]) t = std::max(t, tend[q] + (lane[q] != l ? XSYNC : 0.f)); return t; };
  int guard = 0;
  while (!ready.empty() && guard++ < 100000) {
    int l = 0;
    for (int k = 1; k < NL; ++k) if (lane_free[k] < lane_free[l]) l = k;
    const float T = lane_free[l];
    int best = -1;
    for (int q = 0; q < (int)ready.size(); ++q) {
      if (est(ready[q], l) > T + 0.01f) continue;
      if (best < 0 || prio[ready[q]] > prio[ready[best]] || (prio[ready[q]] == prio[ready[best]] && ready[q] < ready[best])) best = q;
    }
    if (best < 0) {
      // nothing can start on this lane yet: wait for the earliest of (an op becoming startable here, another lane falling idle)
      float nt = 1e30f;
      for (int q : ready) nt = std::min(nt, est(q, l));
      for (int k = 0; k < NL; ++k) if (k != l && lane_free[k] > T) nt = std::min(nt, lane_free[k]);
      lane_free[l] = nt > T ? nt : T + 0.5f;
      continue;
    }
    const int p = ready[best];
    ready.erase(ready.begin() + best);
    lane[p] = l; tend[p] = T + ops[p].cost; lane_free[l] = tend[p];
    order.push_back(p);
    for (int s2 : succ[p]) if (--indeg[s2] == 0) ready.push_back(s2);
  }
  int rc = NUNET_OK;
  if ((int)order.size() != n) { nunet_set_error("plan: list schedule placed %d of %d ops (dependency cycle)", (int)order.size(), n); rc = NUNET_EINVAL; }
  // issue in simulated start order (stable: a topological order of the hazard DAG)
  std::vector<int> idx(order);
  std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return tend[a] - ops[a].cost < tend[b] - ops[b].cost; });
  // (the sort by start time keeps predecessors first: a successor starts at or after its predecessor's end)
  {
    static int dbg = -1;
    if (dbg < 0) { const char* e = getenv("NUNET_LIST_DEBUG"); dbg = e ? atoi(e) : 0; }
    if (dbg && !g_dry_run) {
      float mk = 0.f; for (int i = 0; i < n; ++i) mk = std::max(mk, tend[i]);
      fprintf(stderr, "list schedule pass %d: %d ops, simulated makespan %.1f us, critical path %.1f us%s\n", pass, n, mk, n ? prio[0] : 0.f, rt->op_cost[pass & 1].empty() ? " (built-in costs)" : " (measured costs)");
      if (dbg > 1) for (int i : idx) fprintf(stderr, "  %8.1f %6.1f  L%d %s\n", tend[i] - ops[i].cost, ops[i].cost, lane[i], ops[i].name);
    }
  }
  for (size_t q = 0; q < idx.size() && rc == NUNET_OK; ++q) {
    Op& o = ops[idx[q]];
    hipStream_t st = begin_v(lane[idx[q]], o.rd, o.nrd, o.wr, o.nwr);
    rc = o.fn(st);
    memcpy(cur_name, o.name, sizeof(cur_name));
    end();
  }
  ops.clear();
  return rc;
}

// Single-stream schedule ("wave" mode). ROCm 7.2 replays a hipGraph with parallel branches node by node from the host (3-6 us of
// host time per node, 2-10 us of extra latency per dependent node around every fork / join, tools/graph_gap_probe*.py) but a
// single-stream graph as one batch of pre-built packets (0.9 us per node, 0.3 us of host time). And two independent
// convolutions running side by side cost 1.3-1.4 x one (tools/conv_concurrency_probe.py). So instead of forking lanes the pass is
// emitted on ONE stream, in dependency order, with the concurrency INSIDE the launches: a list scheduler walks the ops by
// critical-path priority and puts every ready convolution of the same kernel variant into the launch of the one it picked
// (nunet_conv3x3_group: up to CONV_GROUP_MAX problems, workgroups dealt round-robin).
int nunet_conv_group_key(const nunet_conv_desc* d);
int nunet_conv3x3_group(const nunet_conv_desc* const* ds, int n, hipStream_t st);
int Sched::run_wave() {
  const int n = (int)ops.size();
  std::vector<std::vector<int>> succ(n);
  std::vector<int> indeg(n, 0), key(n, -1);
  {
    std::vector<int> lastw(NRES, -1);
    std::vector<std::vector<int>> readers(NRES);
    auto edge = [&](int a, int b) { if (a >= 0 && a != b && std::find(succ[a].begin(), succ[a].end(), b) == succ[a].end()) { succ[a].push_back(b); ++indeg[b]; } };
    for (int i = 0; i < n; ++i) {
      const Op& o = ops[i];
      for (int q = 0; q < o.nrd; ++q) edge(lastw[o.rd[q]], i);
      for (int q = 0; q < o.nwr; ++q) { edge(lastw[o.wr[q]], i); for (int r : readers[o.wr[q]]) edge(r, i); }
      for (int q = 0; q < o.nrd; ++q) readers[o.rd[q]].push_back(i);
      for (int q = 0; q < o.nwr; ++q) { lastw[o.wr[q]] = i; readers[o.wr[q]].clear(); }
      if (o.kind == K_CONV) key[i] = nunet_conv_group_key(&o.cd);
    }
  }
  std::vector<float> prio(n, 0.f);
  for (int i = n - 1; i >= 0; --i) { float m = 0.f; for (int s : succ[i]) m = std::max(m, prio[s]); prio[i] = ops[i].cost + m; }
  std::vector<int> ready;
  std::vector<char> done(n, 0);
  for (int i = 0; i < n; ++i) if (indeg[i] == 0) ready.push_back(i);
  int rc = NUNET_OK, emitted = 0;
  while (!ready.empty() && rc == NUNET_OK) {
    int best = 0;
    for (int q = 1; q < (int)ready.size(); ++q) if (prio[ready[q]] > prio[ready[best]] || (prio[ready[q]] == prio[ready[best]] && ready[q] < ready[best])) best = q;
    const int p = ready[best];
    int grp[CONV_GROUP_MAX]; int ng = 0;
    grp[ng++] = p;
    if (key[p] >= 0) {
      // the other ready convolutions of the same variant, most urgent first
      std::vector<int> cand;
      for (int q : ready) if (q != p && key[q] == key[p]) cand.push_back(q);
      std::sort(cand.begin(), cand.end(), [&](int a, int b) { return prio[a] > prio[b] || (prio[a] == prio[b] && a < b); });
      for (int q : cand) if (ng < CONV_GROUP_MAX) grp[ng++] = q;
    }
    if (ng > 1) {
      const nunet_conv_desc* ds[CONV_GROUP_MAX];
      for (int k = 0; k < ng; ++k) ds[k] = &ops[grp[k]].cd;
      rc = nunet_conv3x3_group(ds, ng, main_s);
      snprintf(cur_name, sizeof(cur_name), "%.12s+%d", ops[p].name, ng - 1);
    } else {
      rc = ops[p].fn(main_s);
      memcpy(cur_name, ops[p].name, sizeof(cur_name));
    }
    stamp(main_s, 0); cur_name[0] = 0;
    for (int k = 0; k < ng; ++k) {
      const int i = grp[k];
      done[i] = 1; ++emitted;
      ready.erase(std::find(ready.begin(), ready.end(), i));
      for (int s : succ[i]) if (--indeg[s] == 0) ready.push_back(s);
    }
  }
  if (rc == NUNET_OK && emitted != n) { nunet_set_error("plan: single-stream schedule emitted %d of %d ops (dependency cycle)", emitted, n); rc = NUNET_EINVAL; }
  ops.clear();
  return rc;
}

// Lane of a block. Crossing hardware queues costs 5-10 us of dispatch latency per dependency edge, so the assignment
// decides how many edges of the critical chain (B00>B10>B20>B30>B40>B31>B22>B13>B04 and its mirror in backward) cross
// lanes: the chain runs on lane 0, the side blocks on lanes 1-3 by anti-diagonal (measured best of seven assignments:
// by level, by anti-diagonal, by column, one or two side lanes).
static int lane_of(const nunet_plan* P, const Node& n) {
  if (P->cfg.unet) return n.i;
  if (n.j == 0 || n.i + n.j == 4) return 0;
  return n.i + n.j;           // side blocks: diagonals 1..3 -> lanes 1..3
}

static int blk_index(const nunet_plan* P, int i, int in_prefix_zero_only) {
  for (size_t q = 0; q < P->exec.size(); ++q)
    if (P->exec[q].i == i && (!in_prefix_zero_only || P->exec[q].in_prefix == 0)) return (int)q;
  return -1;
}

// every plan entry that touches the arena checks what the caller says it owns against the plan's own layout
#define ARENA_CHECK(what) \
  NUNET_REQUIRE(arena_bytes >= P->total, what ": arena of %zu bytes, nunet_plan_arena_bytes() = %zu", (size_t)arena_bytes, P->total); \
  NUNET_REQUIRE(((uintptr_t)arena & 255) == 0, what ": arena must be 256-byte aligned")

extern "C" int nunet_plan_forward(nunet_plan* P, const float* params, float* bnbuf, int64_t* nbt, const float* input, void* arena, size_t arena_bytes, float* logits, int32_t training_flags, nunet_stream_t s) {
  const int32_t training = training_flags & 1;
  const bool staged = (training_flags & 4) != 0;       // the image already sits in the arena (nunet_plan_stage_u8): no layout launch
  NUNET_REQUIRE(P && params && (input || staged) && arena && logits, "plan_forward: null pointer");
  ARENA_CHECK("plan_forward");
  NUNET_REQUIRE(bnbuf, "plan_forward: bnbuf (running stats) required");
  hipStream_t st = (hipStream_t)s;
  const nunet_plan_cfg& c = P->cfg;
  const int dt = c.dtype, es = P->es;
  float* save = (float*)AB(arena, P->off_save);
  char* wpack = AB(arena, P->off_wpack);
  // prerequisites of everything on the caller's stream, before the fork: both fixed-point sum regions (the
  // BatchNorm statistics of this pass and the BatchNorm-backward sums of the pass that may follow) in one launch
  if (training) CK(nunet_zero_async(AB(arena, P->off_fx), 2 * fx_region_bytes(P), st));
  if (!staged) CK(nunet_nchw_to_nhwc(input, c.N, c.input_channels, c.H, c.W, dt, AB(arena, P->off_img), 32, (nunet_stream_t)st));

  Sched S; S.init(P, st, 0);
  int rc = NUNET_OK;
  const bool skip_pack = (training_flags & 2) != 0;   // the caller vouches that nunet_plan_update / _repack left the packed weights current
  if (!skip_pack) {
    S.name("pack");
    S.add(0, 0, 15.f, {}, {R_WP + 0, R_WP + 1, R_WP + 2, R_WP + 3, R_WP + 4}, [=](hipStream_t ls) {
      if (dt == NUNET_F32) return launch_pack<float>(params, wpack, P->ptab, P->pack_maxn, ls);
      if (dt == NUNET_BF16) return launch_pack<bf16_t>(params, wpack, P->ptab, P->pack_maxn, ls);
      return launch_pack<f16_t>(params, wpack, P->ptab, P->pack_maxn, ls);
    });
  }
  // (The x2 upsample of a block output can ride in the producer's BatchNorm launch as a second block role - nunet_bn_fwd_desc.up,
  // bit-identical, ten launches fewer per forward. Measured on MI355X, same box: single-lane step unchanged (2.557 vs 2.554 ms),
  // multi-lane graph step SLOWER (8326 vs 8700 and 8041 vs 8232 img/s). The plan keeps the stand-alone launch.)
  for (size_t k = 0; k < P->exec.size() && rc == NUNET_OK; ++k) {
    const Node& n = P->exec[k];
    const int i = n.i, f = NBF[i], H = P->hl[i], W = P->wl[i];
    const int lane = lane_of(P, n), rb = R_BLK + (int)k * B_STRIDE;
    const int rskf = P->sk_floats[k] > 0 ? R_SK + (int)k : -1;
    if (n.up_slot >= 0) {
      S.name("B%d%d.upF", n.i, n.j);
      S.add(lane, 0, 7.f, {R_X + (i + 1) * 5 + n.up_slot}, {rb + B_UP}, [=](hipStream_t ls) {
        return nunet_upsample2x_fwd(dt, c.N, P->hl[i + 1], P->wl[i + 1], NBF[i + 1],
                                    AB(arena, P->X[i + 1] + (size_t)n.up_slot * NBF[i + 1] * es), P->PX[i + 1], AB(arena, n.up), NBF[i + 1], ls);
      });
    }
    // ---- conv1: raw output y1 + its BatchNorm sums (fixed point) ---------------------------------------------
    {
      const ConvL& L = n.c1;
      nunet_conv_desc d; memset(&d, 0, sizeof(d));
      d.dtype = dt; d.N = c.N; d.H = H; d.W = W;
      d.wpack = wpack + (size_t)L.wf * es;
      d.bias = nullptr;  // absorbed by the BatchNorm that follows (bn_stat_coeffs)
      d.dst0 = AB(arena, n.y1); d.D0 = f; d.Q0 = f;
      d.stats = training ? (int64_t*)fx_of(arena, P, 0, L.stats) : nullptr;
      if (P->sk_floats[k] > 0) { d.splitk_ws = (float*)AB(arena, P->off_sk[k]); d.splitk_ws_floats = P->sk_floats[k]; }
      const int alg_cin = (i == 0 && n.in_prefix == 0) ? c.input_channels : 0;
      S.name("B%d%d.conv1", n.i, n.j);
      if (n.in_prefix == 0) {
        if (i == 0) { d.src0 = AB(arena, P->off_img); d.C0 = 32; d.P0 = 32; S.add_conv(lane, {R_IMG, R_WP + i}, {rb + B_Y1, rb + B_ST1, rskf}, d, alg_cin); }
        else { d.src0 = AB(arena, n.pin); d.C0 = NBF[i - 1]; d.P0 = NBF[i - 1]; S.add_conv(lane, {rb + B_PIN, R_WP + i}, {rb + B_Y1, rb + B_ST1, rskf}, d, alg_cin); }
      } else {
        d.src0 = AB(arena, P->X[i]); d.C0 = n.in_prefix * f; d.P0 = P->PX[i];
        d.src1 = AB(arena, n.up); d.C1 = NBF[i + 1]; d.P1 = NBF[i + 1];
        S.add_conv(lane, {R_X + i * 5 + 0, n.in_prefix > 1 ? R_X + i * 5 + 1 : -1, n.in_prefix > 2 ? R_X + i * 5 + 2 : -1,
                          n.in_prefix > 3 ? R_X + i * 5 + 3 : -1, rb + B_UP, R_WP + i}, {rb + B_Y1, rb + B_ST1, rskf}, d, alg_cin);
      }
    }
    // ---- conv2: BatchNorm1 + ReLU applied to y1 on the way into LDS (archs1.py:23-28); the activation a1 is
    // stored on the side for the weight gradient when a backward pass may follow -----------------------------------
    {
      const ConvL& L = n.c2; const ConvL& L1 = n.c1;
      nunet_conv_desc d; memset(&d, 0, sizeof(d));
      d.dtype = dt; d.N = c.N; d.H = H; d.W = W;
      d.src0 = AB(arena, n.y1); d.C0 = f; d.P0 = f;
      d.in_tf = NUNET_TF_BN_RELU; d.tf_training = training;
      d.tf_fx = training ? (const int64_t*)fx_of(arena, P, 0, L1.stats) : nullptr;
      d.tf_gamma = params + L1.g_off; d.tf_beta = params + L1.be_off; d.tf_conv_bias = params + L1.b_off;
      d.tf_running_mean = bnbuf + L1.rm_off; d.tf_running_var = bnbuf + L1.rv_off; d.tf_nbt = nbt ? nbt + L1.bn_index : nullptr;
      d.tf_mean_invstd = save + L1.save; d.tf_momentum = 0.1f; d.tf_eps = 1e-5f;
      d.tf_store = training ? AB(arena, n.a1) : nullptr; d.tf_ps = f;
      d.wpack = wpack + (size_t)L.wf * es;
      d.dst0 = AB(arena, n.y2); d.D0 = f; d.Q0 = f;
      d.stats = training ? (int64_t*)fx_of(arena, P, 0, L.stats) : nullptr;
      if (P->sk_floats[k] > 0) { d.splitk_ws = (float*)AB(arena, P->off_sk[k]); d.splitk_ws_floats = P->sk_floats[k]; }
      S.name("B%d%d.conv2", n.i, n.j);
      S.add_conv(lane, {rb + B_Y1, rb + B_ST1, R_WP + i}, {rb + B_Y2, rb + B_ST2, rb + B_A1, rskf}, d);
    }
    // ---- BatchNorm2 + ReLU (+ 2x2 max-pool for the encoder column): the block output has many consumers
    // (convs of the same level, the upsample, the pool, a head) and is materialised once in its level-buffer slot ------
    {
      const ConvL& L = n.c2;
      nunet_bn_fwd_desc b; memset(&b, 0, sizeof(b));
      b.dtype = dt; b.N = c.N; b.H = H; b.W = W; b.C = f;
      b.y = AB(arena, n.y2); b.PY = f; b.conv_bias = params + L.b_off; b.stats = (const int64_t*)fx_of(arena, P, 0, L.stats);
      b.gamma = params + L.g_off; b.beta = params + L.be_off;
      b.running_mean = bnbuf + L.rm_off; b.running_var = bnbuf + L.rv_off;
      b.num_batches_tracked = nbt ? nbt + L.bn_index : nullptr;
      b.save_mean_invstd = save + L.save; b.training = training; b.momentum = 0.1f; b.eps = 1e-5f;
      b.a = AB(arena, P->X[i] + (size_t)n.out_slot * f * es); b.PA = P->PX[i];
      int rpin = -1;
      if (n.in_prefix == 0 && i < 4) {  // encoder column: feed the next level (archs1.py:115,118,122,127)
        const int q = blk_index(P, i + 1, 1);
        if (q >= 0) { b.pooled = AB(arena, P->exec[q].pin); b.PP = f; rpin = R_BLK + q * B_STRIDE + B_PIN; }
      }
      S.name("B%d%d.bnF2", n.i, n.j);
      S.add(lane, 0, 6.f, {rb + B_Y2, rb + B_ST2}, {R_X + i * 5 + n.out_slot, rpin}, [=](hipStream_t ls) { return nunet_bn_relu_fwd(&b, ls); });
    }
  }
  if (rc == NUNET_OK) {
    const long long plane = (long long)c.N * c.num_classes * c.H * c.W;
    for (size_t k = 0; k < P->heads.size() && rc == NUNET_OK; ++k) {
      const Head hd = P->heads[k];
      S.name("head%d.F", (int)k);
      S.add(0, 0, 10.f, {R_X + hd.slot}, {R_LOGITS + 0}, [=](hipStream_t ls) {
        return nunet_head_fwd(dt, c.N, c.H, c.W, NBF[0], c.num_classes, AB(arena, P->X[0] + (size_t)hd.slot * NBF[0] * es), P->PX[0],
                              params + hd.w_off, params + hd.b_off, logits + plane * k, ls);
      });
    }
  }
  if (rc == NUNET_OK) rc = S.run();
  S.join();  // always rejoin the caller's stream (also on error: a capture must not be left forked)
  if (rc == NUNET_OK && S.failed) { nunet_set_error("plan_forward: lane scheduler overflow (capture stream pool / dependency lists)"); rc = NUNET_EINVAL; }
  return rc;
}

// ---------------------------------------------------------------------------------------------------------
// Device-side input pipeline straight into the plan's image buffer (reference dataset.py:66-74 + trains.py:258-259,266:
// Normalize -> /255 -> HWC->CHW, RandomRotate90 / Flip): uint8 NHWC batch -> ((u/255 - mean)/std) * post_scale, rounded
// to the storage type, as the padded NHWC tile the first conv reads. One thread per pixel writes the pixel's 32 channels
// (zeros beyond C) as 16-byte stores. The arithmetic is nunet_preprocess_u8's followed by nunet_nchw_to_nhwc's rounding, so
// the staged image is bit-identical to the float path's; only uint8 crosses PCIe and the per-step layout launch goes away.
// ---------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void stage_u8_kernel(const uint8_t* __restrict__ u, int N, int H, int W, int C, const float* __restrict__ mean,
                                                       const float* __restrict__ stdv, const int32_t* __restrict__ aug, float post_scale, T* __restrict__ img) {
  constexpr int EPV = Tr<T>::EPV;
  const long long npix = (long long)N * H * W;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < npix; i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % W);
    long long t = i / W;
    const int y = (int)(t % H);
    const int n = (int)(t / H);
    int sy = y, sx = x;                       // destination (y, x) <- source: undo flips, then the rotation (as preprocess_u8_kernel)
    const int a = aug ? aug[n] : 0;
    if (a & 8) sy = H - 1 - sy;
    if (a & 4) sx = W - 1 - sx;
    const int k = a & 3;
    int ry = sy, rx = sx;
    if (k == 1) { ry = sx; rx = W - 1 - sy; }
    else if (k == 2) { ry = H - 1 - sy; rx = W - 1 - sx; }
    else if (k == 3) { ry = H - 1 - sx; rx = sy; }
    const uint8_t* src = u + (((long long)n * H + ry) * W + rx) * C;
    T* dst = img + i * 32;
#pragma unroll
    for (int v = 0; v < 32 / EPV; ++v) {
      Vec16<T> o;
#pragma unroll
      for (int e = 0; e < EPV; ++e) {
        const int c = v * EPV + e;
        float f = 0.f;
        if (c < C) {
          const float m = mean ? mean[c] : 0.f, sd = stdv ? stdv[c] : 1.f;
          f = (((float)src[c] / 255.f - m) / sd) * post_scale;
        }
        o.set(e, f);
      }
      st16(dst + v * EPV, o);
    }
  }
}
template <typename T> static int launch_stage_u8(const uint8_t* u, int N, int H, int W, int C, const float* mean, const float* stdv, const int32_t* aug,
                                                 float post_scale, void* img, hipStream_t st) {
  const long long npix = (long long)N * H * W;
  long long g = (npix + 255) / 256; if (g > 4096) g = 4096;
  ProfScope ps(PC_LAYOUT, 0, (double)npix * (C + 32.0 * sizeof(T)), st);
  NUNET_LAUNCH((stage_u8_kernel<T>), dim3((unsigned)g), dim3(256), 0, st, u, N, H, W, C, mean, stdv, aug, post_scale, (T*)img);
  return nunet_check_launch("plan_stage_u8");
}
extern "C" int nunet_plan_stage_u8(nunet_plan* P, const uint8_t* u8_nhwc, const float* mean, const float* stdv, const int32_t* aug, float post_scale,
                                   void* arena, size_t arena_bytes, nunet_stream_t s) {
  NUNET_REQUIRE(P && u8_nhwc && arena, "plan_stage_u8: null pointer");
  ARENA_CHECK("plan_stage_u8");
  const nunet_plan_cfg& c = P->cfg;
  return NUNET_DISPATCH(c.dtype, launch_stage_u8, u8_nhwc, c.N, c.H, c.W, c.input_channels, mean, stdv, aug, post_scale, (void*)AB(arena, P->off_img), (hipStream_t)s);
}

extern "C" int nunet_plan_backward_phase(nunet_plan* P, const float* params, const float* dlogits, void* arena, size_t arena_bytes, float* grads, int32_t accumulate, int32_t phases, nunet_stream_t s);
extern "C" int nunet_plan_backward(nunet_plan* P, const float* params, const float* dlogits, void* arena, size_t arena_bytes, float* grads, int32_t accumulate, nunet_stream_t s) {
  return nunet_plan_backward_phase(P, params, dlogits, arena, arena_bytes, grads, accumulate, 7, s);
}

extern "C" int nunet_plan_grad_scratch(const nunet_plan* P, int64_t* byte_offset, int64_t* bucket0_floats, int64_t* total_floats) {
  NUNET_REQUIRE(P && byte_offset && bucket0_floats && total_floats, "plan_grad_scratch: null pointer");
  *byte_offset = (int64_t)P->off_gs;
  *bucket0_floats = P->gs_bucket0;
  *total_floats = (int64_t)P->gs_floats;
  return NUNET_OK;
}

// Fused optimiser step on the plan's own buffers (update_kernel above): scratch -> SGD -> repacked weights.
// `grads` (flat OIHW arena) is optional: when given it receives the (scaled) gradients as nunet_plan_backward would
// have left them. Afterwards the packed weights in `arena` are current: the next nunet_plan_forward may be called
// with bit 1 of `training` set (skip the repack).
extern "C" int nunet_plan_update(nunet_plan* P, float* params, float* momentum, void* arena, size_t arena_bytes, const float* lr_dev, float mom, float wd,
                                 int32_t nesterov, float grad_scale, float* grads, nunet_stream_t s) {
  NUNET_REQUIRE(P && params && momentum && arena && lr_dev, "plan_update: null pointer");
  ARENA_CHECK("plan_update");
  hipStream_t st = (hipStream_t)s;
  PackTab& tab = P->ptab;
  int nt = 0;
  for (int i = 0; i < tab.n; ++i) { tab.tile0[i] = nt; nt += ((tab.e[i].cout + 31) / 32) * ((tab.e[i].cinpad + 31) / 32); }
  tab.tile0[tab.n] = nt; tab.ntiles = nt;
  UpdP u;
  u.params = params; u.mom = momentum; u.scratch = (const float*)AB(arena, P->off_gs); u.grads = grads; u.lr = lr_dev;
  u.momc = mom; u.wd = wd; u.gscale = grad_scale; u.nesterov = nesterov; u.nconv = tab.n; u.bid_off = 0;
  const int nheads = P->utab.n - tab.n;
  ProfScope ps(PC_SGD, 0, (double)P->nparams * (grads ? 28.0 : 24.0), st);
  void* wp = AB(arena, P->off_wpack);
  const dim3 grid(nt + nheads), blk(512);
  if (P->cfg.dtype == NUNET_F32) NUNET_LAUNCH((update_kernel<float>), grid, blk, 0, st, u, (float*)wp, tab, P->utab);
  else if (P->cfg.dtype == NUNET_BF16) NUNET_LAUNCH((update_kernel<bf16_t>), grid, blk, 0, st, u, (bf16_t*)wp, tab, P->utab);
  else NUNET_LAUNCH((update_kernel<f16_t>), grid, blk, 0, st, u, (f16_t*)wp, tab, P->utab);
  return nunet_check_launch("plan_update");
}

// A slice of the same launch: blocks [bid0, bid0 + nblocks) of the whole-model numbering (one VGGBlock's tiles, or the heads)
template <typename U> static int launch_update(nunet_plan* P, void* arena, const U& s, int bid0, int nblocks, hipStream_t st) {
  if (nblocks <= 0) return NUNET_OK;
  UpdP u;
  u.params = s.params; u.mom = s.mom; u.scratch = (const float*)AB(arena, P->off_gs); u.grads = s.grads; u.lr = s.lr;
  u.momc = s.momc; u.wd = s.wd; u.gscale = s.gscale; u.nesterov = s.nesterov; u.nconv = P->ptab.n; u.bid_off = bid0;
  void* wp = AB(arena, P->off_wpack);
  const dim3 grid(nblocks), blk(512);
  ProfScope ps(PC_SGD, 0, 0.0, st);
  if (P->cfg.dtype == NUNET_F32) NUNET_LAUNCH((update_kernel<float>), grid, blk, 0, st, u, (float*)wp, P->ptab, P->utab);
  else if (P->cfg.dtype == NUNET_BF16) NUNET_LAUNCH((update_kernel<bf16_t>), grid, blk, 0, st, u, (bf16_t*)wp, P->ptab, P->utab);
  else NUNET_LAUNCH((update_kernel<f16_t>), grid, blk, 0, st, u, (f16_t*)wp, P->ptab, P->utab);
  return nunet_check_launch("plan update (in pass)");
}

// The optimiser step as part of the backward pass: with parameters set here, every whole pass (nunet_plan_backward, or
// nunet_plan_backward_phase with bits 0 and 1) steps each VGGBlock's parameters - and repacks its 16-bit weights - as an op of
// the pass, scheduled behind that block's weight gradients, and the heads at the end. The caller then calls neither
// nunet_plan_update / nunet_plan_sgd nor lets the next forward repack (training flag bit 1). Single-process training only: a
// data-parallel step exchanges the gradients first. params == NULL switches it off.
extern "C" int nunet_plan_set_inpass_update(nunet_plan* P, float* params, float* momentum, const float* lr_dev, float mom, float wd,
                                            int32_t nesterov, float grad_scale, float* grads) {
  NUNET_REQUIRE(P && (!params || (momentum && lr_dev)), "plan_set_inpass_update: null pointer");
  auto& s = rt_of(P)->upd;
  s.params = params; s.mom = momentum; s.lr = lr_dev; s.momc = mom; s.wd = wd; s.gscale = grad_scale; s.nesterov = nesterov; s.grads = grads;
  return NUNET_OK;
}

// Optimiser step straight from the gradient scratch: replaces nunet_plan_backward_phase bit 2 +
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

extern "C" int nunet_plan_sgd(nunet_plan* P, float* params, float* momentum, void* arena, size_t arena_bytes, const float* lr_dev, float mom, float wd,
                              int32_t nesterov, float grad_scale, float* grads, nunet_stream_t s) {
  NUNET_REQUIRE(P && params && momentum && arena && lr_dev, "plan_sgd: null pointer");
  ARENA_CHECK("plan_sgd");
  hipStream_t st = (hipStream_t)s;
  UpdP u;
  u.params = params; u.mom = momentum; u.scratch = (const float*)AB(arena, P->off_gs); u.grads = grads; u.lr = lr_dev;
  u.momc = mom; u.wd = wd; u.gscale = grad_scale; u.nesterov = nesterov; u.nconv = P->ptab.n; u.bid_off = 0;
  ProfScope ps(PC_SGD, 0, (double)P->nparams * (grads ? 24.0 : 20.0), st);
  PackTab& tab = P->ptab;
  int nt = 0;
  for (int i = 0; i < tab.n; ++i) { tab.tile0[i] = nt; nt += ((tab.e[i].cout + 31) / 32) * ((tab.e[i].cinpad + 31) / 32); }
  tab.tile0[tab.n] = nt; tab.ntiles = nt;
  NUNET_LAUNCH(unpack_sgd_tiled_kernel, dim3(nt + P->utab.n), dim3(256), 0, st, u, tab, P->utab);
  return nunet_check_launch("plan_sgd");
}

// Repack the 16-bit weight layouts from the fp32 master parameters (what nunet_plan_forward does first unless told
// that they are current): needed once before a loop that relies on nunet_plan_update, and after the parameters were
// changed by anything else (checkpoint load, a stock optimiser).
extern "C" int nunet_plan_repack(nunet_plan* P, const float* params, void* arena, size_t arena_bytes, nunet_stream_t s) {
  NUNET_REQUIRE(P && params && arena, "plan_repack: null pointer");
  ARENA_CHECK("plan_repack");
  char* wpack = AB(arena, P->off_wpack);
  if (P->cfg.dtype == NUNET_F32) return launch_pack<float>(params, wpack, P->ptab, P->pack_maxn, (hipStream_t)s);
  if (P->cfg.dtype == NUNET_BF16) return launch_pack<bf16_t>(params, wpack, P->ptab, P->pack_maxn, (hipStream_t)s);
  return launch_pack<f16_t>(params, wpack, P->ptab, P->pack_maxn, (hipStream_t)s);
}

// Sum of the weight gradients' K-split slabs into the native-layout gradient scratch, all layers of a phase in one
// launch. A block owns 64 float4 outputs of one layer; thread (part, e) adds slabs part, part + 4, ... in order, the
// four parts meet in LDS in order: a fixed summation tree, so the gradient is bit-identical from run to run.
#define MAXRED 40
struct RedEnt { long long slab, dst, stride, n4; int ks, blk0; };
struct RedTab { int n, nblocks; RedEnt e[MAXRED]; };
__global__ __launch_bounds__(256) void reduce_kernel(const float* __restrict__ slabs, float* __restrict__ gs, RedTab tab) {
  __shared__ f32x4 s_p[4][64];
  int e = 0;
  while (e + 1 < tab.n && (int)blockIdx.x >= tab.e[e + 1].blk0) ++e;
  const RedEnt en = tab.e[e];
  const int part = threadIdx.x >> 6, el = threadIdx.x & 63;
  const long long i = ((long long)blockIdx.x - en.blk0) * 64 + el;
  f32x4 a = {0.f, 0.f, 0.f, 0.f};
  if (i < en.n4) {
    const float* q = slabs + en.slab + i * 4;
#pragma unroll 4
    for (int sl = part; sl < en.ks; sl += 4) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(q + (size_t)sl * en.stride);
      a[0] += v[0]; a[1] += v[1]; a[2] += v[2]; a[3] += v[3];
    }
  }
  s_p[part][el] = a;
  __syncthreads();
  if (part == 0 && i < en.n4) {
    f32x4 r = s_p[0][el];
#pragma unroll
    for (int q = 1; q < 4; ++q) { const f32x4 v = s_p[q][el]; r[0] += v[0]; r[1] += v[1]; r[2] += v[2]; r[3] += v[3]; }
    *reinterpret_cast<f32x4*>(gs + en.dst + i * 4) = r;
  }
}
// layers of exec nodes [k_lo, k_hi] whose gradient was split (ks > 1; unsplit layers write the scratch directly)
static int launch_reduce(nunet_plan* P, void* arena, int k_lo, int k_hi, hipStream_t st) {
  RedTab tab; tab.n = 0; tab.nblocks = 0;
  double bytes = 0;
  for (int k = k_hi; k >= k_lo; --k)
    for (int cv = 1; cv >= 0; --cv) {
      const ConvL& c = cv ? P->exec[k].c2 : P->exec[k].c1;
      if (c.ks <= 1) continue;
      RedEnt& en = tab.e[tab.n++];
      en.slab = c.slab; en.dst = c.gs; en.stride = 9LL * c.cout * c.cinpad; en.n4 = en.stride / 4; en.ks = c.ks; en.blk0 = tab.nblocks;
      tab.nblocks += (int)((en.n4 + 63) / 64);
      bytes += (double)en.stride * 4 * (c.ks + 1);
    }
  if (tab.n == 0) return NUNET_OK;
  ProfScope ps(PC_UNPACK, 0, bytes, st);
  NUNET_LAUNCH(reduce_kernel, dim3(tab.nblocks), dim3(256), 0, st, (const float*)AB(arena, P->off_slab), (float*)AB(arena, P->off_gs), tab);
  return nunet_check_launch("wgrad slab reduce");
}

// Which op COMPLETES the gradient of block output x_{l,s} (slot s of GX_l)? Writers: a head (level 0), the dgrad of conv1
// of a block of level l whose concat holds slot s, the upsample-backward of the level l-1 block that upsampled it, the
// pool-backward of the level l+1 encoder block (slot 0). Ops are issued heads first, then blocks for k descending, and
// writers of one slot are serialised in that order by the lane scheduler - so the last writer is the block writer with
// the smallest k (or the head when no block writes the slot). A head that completes its slot (the output block's) takes
// that block's BatchNorm-backward reduce in its epilogue (nunet_head_bwd_bnr); every other block runs the stand-alone
// reduce (the same fusion in upsample- / pool-backward was measured slower, see elementwise.hip).
enum { LW_NONE = 0, LW_HEAD, LW_DGRAD, LW_UPB, LW_POOLB };
static void last_writer(const nunet_plan* P, int l, int s, int& kind, int& kk) {
  kind = LW_NONE; kk = 1 << 30;
  for (size_t h = 0; h < P->heads.size(); ++h) if (l == 0 && P->heads[h].slot == s) { kind = LW_HEAD; kk = 1 << 29; }
  for (int k = (int)P->exec.size() - 1; k >= 0; --k) {
    const Node& n = P->exec[k];
    int w = LW_NONE;
    if (n.i == l && n.in_prefix > s && !(n.i == 0 && n.in_prefix == 0)) w = LW_DGRAD;
    if (n.i + 1 == l && n.in_prefix > 0 && n.up_slot == s) w = LW_UPB;
    if (n.i - 1 == l && n.in_prefix == 0 && n.i > 0 && s == 0) w = LW_POOLB;
    if (w != LW_NONE && k < kk) { kind = w; kk = k; }
  }
}
static int block_of_slot(const nunet_plan* P, int l, int s) {
  for (size_t k = 0; k < P->exec.size(); ++k) if (P->exec[k].i == l && P->exec[k].out_slot == s) return (int)k;
  return -1;
}

// phases: 1 = heads and the last anti-diagonal's blocks (75 % of the gradient bytes);
//         2 = the remaining blocks; 4 = unpack into the flat OIHW gradient arena. 7 = everything.
extern "C" int nunet_plan_backward_phase(nunet_plan* P, const float* params, const float* dlogits, void* arena, size_t arena_bytes, float* grads, int32_t accumulate, int32_t phases, nunet_stream_t s) {
  NUNET_REQUIRE(P && params && dlogits && arena && grads, "plan_backward: null pointer");
  ARENA_CHECK("plan_backward");
  hipStream_t st = (hipStream_t)s;
  const nunet_plan_cfg& c = P->cfg;
  const int dt = c.dtype, es = P->es;
  float* gsr = (float*)AB(arena, P->off_gs);
  float* slabs = (float*)AB(arena, P->off_slab);
  float* save = (float*)AB(arena, P->off_save);
  char* wpack = AB(arena, P->off_wpack);
  bool (&written)[5][5] = rt_of(P)->bwd_written;
  if (phases & 1) memset(written, 0, sizeof(written));
  const int nnodes = (int)P->exec.size();
  const int k_split = nnodes - P->first_phase_nodes;      // phase 1: nodes [k_split, nnodes); phase 2: [0, k_split)
  const int k_hi = (phases & 1) ? nnodes - 1 : k_split - 1;
  const int k_lo = (phases & 2) ? 0 : k_split;
  // bit 3: leave the pass OPEN after this call - no join: the lanes, their tails and the dependency tracker stay alive in the plan,
  // the caller's stream is not made to wait for anything, and nunet_plan_bucket0_wait(P, s2) orders another stream behind exactly
  // the gradients of the first bucket; bit 4: CONTINUE that open pass (its lanes, its tracker) instead of forking anew, and join at
  // the end. Together they let a data-parallel caller put the first bucket's exchange BESIDE phase 2 inside one captured graph
  // without the fork / join barrier that cutting the pass into two calls used to cost.
  const bool leave_open = (phases & 8) != 0, cont = (phases & 16) != 0;
  PlanRt* const rt = rt_of(P);
  NUNET_REQUIRE(!cont || rt->open_sched, "plan_backward: phase bit 4 (continue) without an open pass");
  NUNET_REQUIRE(!(leave_open && (phases & 4)), "plan_backward: an open pass cannot unpack (bit 2)");
  if (!cont && rt->open_sched) { sched_free(rt->open_sched); rt->open_sched = nullptr; }     // an abandoned open pass
  Sched* const Sp = cont ? rt->open_sched : new Sched();
  Sched& S = *Sp;
  if (!cont) S.init(P, st, 1);
  else NUNET_REQUIRE(S.main_s == st, "plan_backward: continue on the stream the open pass was started on");
  rt->open_sched = nullptr;
  struct Owner { Sched* s; PlanRt* rt; bool keep; ~Owner() { if (keep) rt->open_sched = s; else sched_free(s); } } owner{Sp, rt, false};
  int rc = NUNET_OK;
  const long long plane = (long long)c.N * c.num_classes * c.H * c.W;
  // reduce pass of block kt's second BatchNorm, for the kernel that completes its output gradient
  auto bnr_of = [&](int kt) {
    const Node& t = P->exec[kt];
    nunet_bnr_desc b; memset(&b, 0, sizeof(b));
    b.y = AB(arena, t.y2); b.PY = NBF[t.i]; b.mean_invstd = save + t.c2.save;
    b.gamma = params + t.c2.g_off; b.beta = params + t.c2.be_off; b.sums = (int64_t*)fx_of(arena, P, 1, t.c2.bsum);
    return b;
  };
  auto completes = [&](int l, int sl, int kind, int k) {      // does op (kind, block k) complete x_{l,sl}'s gradient?
    int lk, lkk; last_writer(P, l, sl, lk, lkk);
    if (lk != kind || (kind != LW_HEAD && lkk != k)) return -1;
    return block_of_slot(P, l, sl);
  };
  for (size_t k = 0; k < P->heads.size() && rc == NUNET_OK && (phases & 1); ++k) {
    const Head& h = P->heads[k];
    const int acc = written[0][h.slot] ? 1 : 0;
    const int kt = completes(0, h.slot, LW_HEAD, -1);
    nunet_bnr_desc bd; memset(&bd, 0, sizeof(bd));
    if (kt >= 0) bd = bnr_of(kt);
    S.name("head%d.B", (int)k);
    S.add(0, 0, 15.f, {R_X + h.slot, R_DLOGITS, kt >= 0 ? R_BLK + kt * B_STRIDE + B_Y2 : -1}, {R_GX + h.slot, R_GSV + 30 + (int)k, kt >= 0 ? R_GSV + 2 * kt + 1 : -1}, [=](hipStream_t ls) {
      return nunet_head_bwd_bnr(dt, c.N, c.H, c.W, NBF[0], c.num_classes, AB(arena, P->X[0] + (size_t)h.slot * NBF[0] * es), P->PX[0],
                                params + h.w_off, dlogits + plane * k, AB(arena, P->GX[0] + (size_t)h.slot * NBF[0] * es), P->PX[0],
                                acc, gsr + h.gs, HEAD_SLABS, kt >= 0 ? &bd : nullptr, ls);
    });
    written[0][h.slot] = true;
  }
  const bool b0_inside = (phases & 3) == 3 && rt_of(P)->b0_enabled && !P->cfg.unet;   // bucket 0 signalled from inside the pass
  // in-pass optimiser step: only a WHOLE pass may carry it (a data-parallel caller exchanges the gradients between the phases)
  const auto upd = rt_of(P)->upd;
  const bool inpass = upd.params != nullptr && (phases & 3) == 3;
  // NUNET_DEBUG_SPIN_US (tests only): a spin kernel of that many microseconds heads phase 2 on the chain lane, so that
  // "bucket 0 is complete well before the pass ends" can be asserted with a margin (tests/test_dist_gpu.py)
  static int spin_us = -1;
  if (spin_us < 0) { const char* e = getenv("NUNET_DEBUG_SPIN_US"); spin_us = e ? atoi(e) : 0; }

  for (int k = k_hi; k >= k_lo && rc == NUNET_OK; --k) {
    const Node& n = P->exec[k];
    const int i = n.i, f = NBF[i], H = P->hl[i], W = P->wl[i];
    const int lane = lane_of(P, n), wlane = 5 + lane, rb = R_BLK + k * B_STRIDE, rl = R_LVL + k * L_STRIDE;
    const int rsk = P->sk_floats[k] > 0 ? R_SK + k : -1;
    if (!written[i][n.out_slot]) { nunet_set_error("plan_backward: internal: grad of x%d_%d never produced", n.i, n.j); rc = NUNET_EINVAL; break; }
    if (spin_us > 0 && k == k_split - 1 && (phases & 2)) {
      const int us = spin_us;
      S.name("spin");
      S.add(lane, 0, 0.f, {R_GX + i * 5 + n.out_slot}, {R_GX + i * 5 + n.out_slot}, [=](hipStream_t ls) { return nunet_debug_spin(us, 1, (nunet_stream_t)ls); });
    }
    const ConvL& L1 = n.c1; const ConvL& L2 = n.c2;
    char* const dy2 = AB(arena, P->off_dy[k][0]); char* const dy1 = AB(arena, P->off_dy[k][1]);
    char* const da1 = AB(arena, P->off_da1[k]);
    const int r_dy2 = rl + L_DY0, r_dy1 = rl + L_DY1, r_da1 = rl + L_DA1;
    const int r_v2 = R_GSV + 2 * k + 1, r_v1 = R_GSV + 2 * k;       // per-conv sums + small-vector gradients
    const int r_gxo = R_GX + i * 5 + n.out_slot;
    float* const gv2 = gsr + L2.gs + 9LL * L2.cout * L2.cinpad;     // [dbias | dgamma | dbeta]
    float* const gv1 = gsr + L1.gs + 9LL * L1.cout * L1.cinpad;
    const bool has_dgrad1 = !(i == 0 && n.in_prefix == 0);           // no gradient into the image

    // ---- BatchNorm2 + ReLU backward, REDUCE pass: sum dz, sum dz * xhat over the block-output gradient (which several
    // consumers accumulated into its level-buffer slot) -> fixed-point sums
    nunet_bn_bwd_desc b2; memset(&b2, 0, sizeof(b2));
    b2.dtype = dt; b2.N = c.N; b2.H = H; b2.W = W; b2.C = f;
    b2.da = AB(arena, P->GX[i] + (size_t)n.out_slot * f * es); b2.PDA = P->PX[i]; b2.y = AB(arena, n.y2); b2.PY = f;
    b2.mean_invstd = save + L2.save; b2.gamma = params + L2.g_off; b2.beta = params + L2.be_off;
    b2.sums = (int64_t*)fx_of(arena, P, 1, L2.bsum);
    {
      int lk, lkk; last_writer(P, i, n.out_slot, lk, lkk);
      const bool fused = lk == LW_HEAD;   // taken by the head backward, which completed the gradient
      if (!fused) {
        S.name("B%d%d.bnR2", n.i, n.j);
        S.add(lane, 0, 9.f, {r_gxo, rb + B_Y2}, {r_v2}, [=](hipStream_t ls) { return nunet_bn_relu_bwd_reduce(&b2, ls); });
      }
    }

    // ---- dgrad of conv2: the APPLY pass of BatchNorm2's backward happens on the way into LDS (dy2 is stored on the
    // side for the weight gradient); the epilogue takes BatchNorm1's reduce pass on the da1 it stores (archs1.py:23-30)
    {
      nunet_conv_desc d; memset(&d, 0, sizeof(d));
      d.dtype = dt; d.N = c.N; d.H = H; d.W = W;
      d.src0 = b2.da; d.C0 = f; d.P0 = P->PX[i];
      d.in_tf = NUNET_TF_BN_RELU_BWD; d.tf_y = AB(arena, n.y2); d.tf_py = f; d.tf_fx = b2.sums;
      d.tf_gamma = b2.gamma; d.tf_beta = b2.beta; d.tf_mean_invstd = save + L2.save;
      d.tf_dbias = gv2; d.tf_dgamma = gv2 + L2.cout; d.tf_dbeta = gv2 + 2 * L2.cout;
      d.tf_store = dy2; d.tf_ps = f;
      d.wpack = wpack + (size_t)L2.wd * es;
      d.dst0 = da1; d.D0 = f; d.Q0 = f;
      d.bn_y = AB(arena, n.y1); d.bn_py = f; d.bn_mean_invstd = save + L1.save;
      d.bn_gamma = params + L1.g_off; d.bn_beta = params + L1.be_off; d.bn_sums = (int64_t*)fx_of(arena, P, 1, L1.bsum);
      if (P->sk_floats[k] > 0) { d.splitk_ws = (float*)AB(arena, P->off_sk[k]); d.splitk_ws_floats = P->sk_floats[k]; }
      S.name("B%d%d.dgrad2", n.i, n.j);
      S.add_conv(lane, {r_gxo, rb + B_Y2, rb + B_Y1, R_WP + i}, {r_dy2, r_da1, r_v2, r_v1, rsk}, d);
    }
    // ---- dgrad of conv1 with BatchNorm1's apply pass on the way in; the first block has no input gradient and
    // runs the stand-alone apply pass for its weight gradient instead
    if (has_dgrad1) {
      nunet_conv_desc d; memset(&d, 0, sizeof(d));
      d.dtype = dt; d.N = c.N; d.H = H; d.W = W;
      d.src0 = da1; d.C0 = f; d.P0 = f;
      d.in_tf = NUNET_TF_BN_RELU_BWD; d.tf_y = AB(arena, n.y1); d.tf_py = f; d.tf_fx = (const int64_t*)fx_of(arena, P, 1, L1.bsum);
      d.tf_gamma = params + L1.g_off; d.tf_beta = params + L1.be_off; d.tf_mean_invstd = save + L1.save;
      d.tf_dbias = gv1; d.tf_dgamma = gv1 + L1.cout; d.tf_dbeta = gv1 + 2 * L1.cout;
      d.tf_store = dy1; d.tf_ps = f;
      d.wpack = wpack + (size_t)L1.wd * es;
      if (P->sk_floats[k] > 0) { d.splitk_ws = (float*)AB(arena, P->off_sk[k]); d.splitk_ws_floats = P->sk_floats[k]; }
      S.name("B%d%d.dgrad1", n.i, n.j);
      if (n.in_prefix == 0) {
        d.dst0 = AB(arena, P->off_gpin[k]); d.D0 = NBF[i - 1]; d.Q0 = NBF[i - 1];
        S.add_conv(lane, {r_da1, rb + B_Y1, r_v1, R_WP + i}, {r_dy1, rl + L_GPIN, rsk}, d);
        // through MaxPool2d(2,2) into x_{i-1,0}
        const int acc = written[i - 1][0] ? 1 : 0;
        S.name("B%d%d.poolB", n.i, n.j);
        S.add(lane, 0, 8.f, {rl + L_GPIN, R_X + (i - 1) * 5 + 0}, {R_GX + (i - 1) * 5 + 0}, [=](hipStream_t ls) {
          return nunet_maxpool2x2_bwd(dt, c.N, P->hl[i - 1], P->wl[i - 1], NBF[i - 1], AB(arena, P->X[i - 1]), P->PX[i - 1],
                                      AB(arena, P->off_gpin[k]), NBF[i - 1], AB(arena, P->GX[i - 1]), P->PX[i - 1], acc, ls);
        });
        written[i - 1][0] = true;
      } else {
        d.dst0 = AB(arena, P->GX[i]); d.D0 = n.in_prefix * f; d.Q0 = P->PX[i]; d.acc_slot_w = f;
        for (int q = 0; q < n.in_prefix; ++q) { if (written[i][q]) d.acc0_mask |= 1u << q; written[i][q] = true; }
        d.dst1 = AB(arena, P->off_gup[k]); d.D1 = NBF[i + 1]; d.Q1 = NBF[i + 1];
        S.add_conv(lane, {r_da1, rb + B_Y1, r_v1, R_WP + i},
                   {r_dy1, R_GX + i * 5 + 0, n.in_prefix > 1 ? R_GX + i * 5 + 1 : -1, n.in_prefix > 2 ? R_GX + i * 5 + 2 : -1,
                    n.in_prefix > 3 ? R_GX + i * 5 + 3 : -1, rl + L_GUP, rsk}, d);
        // through the bilinear upsample into x_{i+1,up_slot}
        const int acc = written[i + 1][n.up_slot] ? 1 : 0;
        S.name("B%d%d.upB", n.i, n.j);
        S.add(lane, 0, 11.f, {rl + L_GUP}, {R_GX + (i + 1) * 5 + n.up_slot}, [=](hipStream_t ls) {
          return nunet_upsample2x_bwd(dt, c.N, P->hl[i + 1], P->wl[i + 1], NBF[i + 1], AB(arena, P->off_gup[k]), NBF[i + 1],
                                      AB(arena, P->GX[i + 1] + (size_t)n.up_slot * NBF[i + 1] * es), P->PX[i + 1], acc, ls);
        });
        written[i + 1][n.up_slot] = true;
      }
    } else {
      nunet_bn_bwd_desc b1; memset(&b1, 0, sizeof(b1));
      b1.dtype = dt; b1.N = c.N; b1.H = H; b1.W = W; b1.C = f;
      b1.da = da1; b1.PDA = f; b1.y = AB(arena, n.y1); b1.PY = f;
      b1.mean_invstd = save + L1.save; b1.gamma = params + L1.g_off; b1.beta = params + L1.be_off;
      b1.sums = (int64_t*)fx_of(arena, P, 1, L1.bsum);
      b1.dbias = gv1; b1.dgamma = gv1 + L1.cout; b1.dbeta = gv1 + 2 * L1.cout;
      b1.dy = dy1; b1.PDY = f;
      S.name("B%d%d.bnA1", n.i, n.j);
      S.add(lane, 0, 9.f, {r_da1, rb + B_Y1, r_v1}, {r_dy1}, [=](hipStream_t ls) { return nunet_bn_relu_bwd_apply(&b1, ls); });
    }
    // ---- both weight gradients of the block in one launch (leaves of the dependency graph: only the slab reduce
    // reads them); conv1's problem first: it may carry the first layer's algorithmic Cin for the profiler
    {
      nunet_wgrad_desc w1, w2; memset(&w1, 0, sizeof(w1)); memset(&w2, 0, sizeof(w2));
      w1.dtype = w2.dtype = dt; w1.N = w2.N = c.N; w1.H = w2.H = H; w1.W = w2.W = W;
      w2.src0 = AB(arena, n.a1); w2.C0 = f; w2.P0 = f;
      if (n.in_prefix == 0) {
        if (i == 0) { w1.src0 = AB(arena, P->off_img); w1.C0 = 32; w1.P0 = 32; }
        else { w1.src0 = AB(arena, n.pin); w1.C0 = NBF[i - 1]; w1.P0 = NBF[i - 1]; }
      } else {
        w1.src0 = AB(arena, P->X[i]); w1.C0 = n.in_prefix * f; w1.P0 = P->PX[i];
        w1.src1 = AB(arena, n.up); w1.C1 = NBF[i + 1]; w1.P1 = NBF[i + 1];
      }
      w1.dy = dy1; w2.dy = dy2; w1.Cout = w2.Cout = f; w1.PY = w2.PY = f;
      w1.dw = L1.ks > 1 ? slabs + L1.slab : gsr + L1.gs; w1.slab_stride = 9LL * L1.cout * L1.cinpad; w1.max_slabs = L1.ks; w1.target_wgs = L1.wg_target;
      w2.dw = L2.ks > 1 ? slabs + L2.slab : gsr + L2.gs; w2.slab_stride = 9LL * L2.cout * L2.cinpad; w2.max_slabs = L2.ks; w2.target_wgs = L2.wg_target;
      w1.dw_floats = (int64_t)L1.ks * w1.slab_stride; w2.dw_floats = (int64_t)L2.ks * w2.slab_stride;
      const int alg_cin = (i == 0 && n.in_prefix == 0) ? c.input_channels : 0;
      S.name("B%d%d.wgrad", n.i, n.j);
      int rx[4] = {-1, -1, -1, -1}, r_in = -1, r_up = -1;
      if (n.in_prefix == 0) r_in = (i == 0 ? R_IMG : rb + B_PIN);
      else { for (int q = 0; q < n.in_prefix && q < 4; ++q) rx[q] = R_X + i * 5 + q; r_up = rb + B_UP; }
      // The weight gradients of the shallow blocks of the critical chain (B04, B13, B22: full-chip launches) are held
      // back until the chain reaches the deep levels (B31 ..., grid-starved kernels that leave most CUs idle): they run
      // on lane 4 behind a dependency on the gradient that B22's upsample-backward hands to B31
      int wl = wlane, r_gate = -1;     // (measured +1.9 % on the step)
      if (!P->cfg.unet && n.j > 0 && n.i + n.j == 4 && n.i <= 2 && (phases & 3) == 3 && !S.wave && !S.list) { wl = 4; r_gate = R_GX + 3 * 5 + 1; }
      const float wcost = 8.f + (float)(2.0 * 9 * ((double)L1.cinpad + L2.cinpad) * f * (double)c.N * H * W / 3.5e8);
      S.add(wl, 1, wcost, {rb + B_A1, r_dy2, r_dy1, r_in, rx[0], rx[1], rx[2], rx[3], r_up, r_gate}, {R_GSW + 2 * k, R_GSW + 2 * k + 1},
            [=](hipStream_t ls) {
              g_prof_alg_cin = alg_cin; int r = nunet_conv3x3_wgrad_pair(&w1, &w2, ls); g_prof_alg_cin = 0;
              // the block's K-split slabs are summed right behind it on the same lane: 15 small launches hidden beside the
              // rest of the pass instead of one 58 us launch over all layers after the join, with nothing beside it
              return r ? r : launch_reduce(P, arena, k, k, ls);
            });
    }
    // ---- in-pass optimiser step (nunet_plan_set_inpass_update): this block's two convolutions are stepped - scratch -> SGD ->
    // both packed 16-bit layouts - as soon as their gradients are complete, beside the rest of the backward pass, instead of in
    // one launch over all layers after it (43 us) plus the next forward's repack (15 us), with nothing to overlap either
    if (inpass) {
      int r0 = -1;
      for (size_t r = 0; r < P->reg.size(); ++r) if (P->reg[r] == k) r0 = (int)r;
      if (r0 >= 0) {
        const PackTab& tab = P->ptab;
        const int t0 = tab.tile0[2 * r0], t1 = tab.tile0[2 * r0 + 2];
        const long long np = 9LL * (L1.cout * L1.cinpad + L2.cout * L2.cinpad);
        S.name("B%d%d.upd", n.i, n.j);
        S.add(wlane, 1, 4.f + (float)(np * 28.0 / 3.0e6), {R_GSW + 2 * k, R_GSW + 2 * k + 1, R_GSV + 2 * k, R_GSV + 2 * k + 1}, {R_PRM + k}, [=](hipStream_t ls) {
          return launch_update(P, arena, upd, t0, t1 - t0, ls);
        });
      }
    }
    // "bucket 0 complete" (data-parallel exchange beside the rest of the backward pass, nunet_plan_bucket0_*): on the otherwise
    // unused lane 4, ops that read every gradient resource of the phase-1 nodes and the heads; the last one sums the phase's
    // weight-gradient slabs and records the plan's event - as an external event record node when the pass is being captured
    if (k == k_split && b0_inside && S.multi) {
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
  if (inpass && (phases & 1) && !P->heads.empty()) {
    // the 1x1 heads: blocks past the last tile of the update kernel's numbering
    int rs[12]; int nr = 0;
    for (size_t h = 0; h < P->heads.size() && nr < 12; ++h) rs[nr++] = R_GSV + 30 + (int)h;
    const int nh = P->utab.n - P->ptab.n, nt = P->ptab.ntiles;
    S.name("heads.upd");
    S.add_v(0, 1, 5.f, rs, nr, [=](hipStream_t ls) { return launch_update(P, arena, upd, nt, nh, ls); });
  }
  if (rc == NUNET_OK) rc = S.run();
  if (leave_open && rc == NUNET_OK && !S.failed) {
    // the producers of the first bucket: last-writer events of every gradient resource of the phase-1 nodes and the heads
    // (single-lane issue: none - everything is in order on the caller's stream, which the caller makes its side stream wait for)
    rt->b0_events.clear();
    auto take = [&](int r) {
      if (!S.multi) return;
      hipEvent_t e = S.res[r].w_ev;
      if (!e) return;
      for (hipEvent_t q : rt->b0_events) if (q == e) return;
      rt->b0_events.push_back(e);
    };
    for (int kk = k_split; kk < nnodes; ++kk) { take(R_GSW + 2 * kk); take(R_GSW + 2 * kk + 1); take(R_GSV + 2 * kk); take(R_GSV + 2 * kk + 1); }
    for (size_t h = 0; h < P->heads.size(); ++h) take(R_GSV + 30 + (int)h);
    owner.keep = true;
    return NUNET_OK;
  }
  S.join();
  if (rc == NUNET_OK && S.failed) { nunet_set_error("plan_backward: lane scheduler overflow (capture stream pool / dependency lists)"); rc = NUNET_EINVAL; }
  if (rc) return rc;
  if (!(phases & 4)) return NUNET_OK;
  P->utab.accumulate = accumulate;
  ProfScope ps(PC_UNPACK, 0, (double)P->nparams * (accumulate ? 12 : 8), st);
  PackTab& tab = P->ptab;
  int nt = 0;
  for (int i = 0; i < tab.n; ++i) { tab.tile0[i] = nt; nt += ((tab.e[i].cout + 31) / 32) * ((tab.e[i].cinpad + 31) / 32); }
  tab.tile0[tab.n] = nt; tab.ntiles = nt;
  NUNET_LAUNCH(unpack_tiled_kernel, dim3(P->ptab.ntiles + P->utab.n), dim3(256), 0, st, gsr, grads, P->ptab, P->utab);
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
  NUNET_REQUIRE(P, "plan_bucket0_wait: null plan");
  if (rt_of(P)->open_sched) {
    // an open pass (backward phase bit 3): `s` waits for the kernels that complete the first bucket, nothing else. While the pass is
    // being captured `s` must be a stream the capture has not used (it joins the capture here, as a lane continuation does).
    for (hipEvent_t e : rt_of(P)->b0_events)
      if (hipStreamWaitEvent((hipStream_t)s, e, 0) != hipSuccess) { nunet_set_error("plan_bucket0_wait: %s", hipGetErrorString(hipGetLastError())); return NUNET_ELAUNCH; }
    return NUNET_OK;
  }
  NUNET_REQUIRE(rt_of(P)->b0_event && rt_of(P)->b0_enabled, "plan_bucket0_wait: neither an open pass nor the bucket-0 event is armed");
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

// Calibration of the list scheduler: between begin = 1 and begin = 0 every pass runs on ONE lane with a device timestamp behind
// every op (capture one step into a graph, replay it a few times, synchronise); begin = 0 turns the stamps into the isolated cost
// of every op (by name: "B04.dgrad1" ...), which later passes use instead of the built-in estimates.
// Forget the side lanes of the flag-synchronised program: the next recording measures and picks new ones (the old streams stay
// alive - and keep their hardware-queue slots - until the plan is destroyed, so the new candidates land elsewhere). For a caller
// that finds its program slow: which queue a stream inherits is ROCm's choice, and 1 process in 5 came up with a program at
// 4.8 ms per step instead of 1.7 when nothing checked. No program recorded on the old lanes may be replayed afterwards... it may:
// the streams live on; it just keeps its lanes.
extern "C" int nunet_plan_reset_lanes(nunet_plan* P) {
  NUNET_REQUIRE(P, "plan_reset_lanes: null plan");
  PlanRt* rt = rt_of(P);
  rt->seg_lanes[0] = rt->seg_lanes[1] = rt->seg_lanes[2] = nullptr;
  rt->seg_lanes_distinct = 0;
  return NUNET_OK;
}

extern "C" int nunet_plan_set_lane_priority(nunet_plan* P, int32_t lowest) {
  NUNET_REQUIRE(P, "plan_set_lane_priority: null plan");
  rt_of(P)->lane_low_priority = lowest ? 1 : 0;
  return NUNET_OK;
}

extern "C" int nunet_plan_calibrate(nunet_plan* P, int32_t begin) {
  NUNET_REQUIRE(P, "plan_calibrate: null plan");
  PlanRt* rt = rt_of(P);
  if (begin) {
    if (!rt->stamps && hipMalloc((void**)&rt->stamps, 2 * STAMP_CAP * sizeof(unsigned long long)) != hipSuccess) {
      (void)hipGetLastError(); rt->stamps = nullptr;
      nunet_set_error("plan_calibrate: cannot allocate the stamp buffer"); return NUNET_ELAUNCH;
    }
    rt->calibrating = 1;
    return NUNET_OK;
  }
  NUNET_REQUIRE(rt->calibrating && rt->stamps, "plan_calibrate: end without begin");
  rt->calibrating = 0;
  int rc = NUNET_OK;
  if (hipDeviceSynchronize() != hipSuccess) { nunet_set_error("plan_calibrate: synchronise failed"); rc = NUNET_ELAUNCH; }
  for (int pass = 0; pass < 2 && rc == NUNET_OK; ++pass) {
    const std::vector<std::string>& lab = rt->stamp_labels[pass];
    const int n = (int)lab.size();
    std::vector<unsigned long long> t(n > 0 ? n : 1);
    if (n > 0 && hipMemcpy(t.data(), rt->stamps + (size_t)pass * STAMP_CAP, (size_t)n * 8, hipMemcpyDeviceToHost) != hipSuccess) {
      nunet_set_error("plan_calibrate: copy failed"); rc = NUNET_ELAUNCH; break;
    }
    rt->op_cost[pass].clear();
    for (int k = 1; k < n; ++k) {
      const size_t sp = lab[k].find(' ');                         // "L0 name"
      if (sp == std::string::npos || t[k] <= t[k - 1]) continue;
      // 100 MHz ticks; the stamp kernel and its node gap (~2 us) belong to the measurement, not to the op
      const float us = (float)(t[k] - t[k - 1]) / 100.f - 2.f;
      rt->op_cost[pass][lab[k].substr(sp + 1)] = us > 1.f ? us : 1.f;
    }
  }
  static int keep = -1;
  if (keep < 0) { const char* e = getenv("NUNET_STAMPS"); keep = e ? atoi(e) : 0; }
  if (!keep) { (void)hipFree(rt->stamps); rt->stamps = nullptr; rt->stamp_labels[0].clear(); rt->stamp_labels[1].clear(); }
  return rc;
}

extern "C" int nunet_debug_stamp(uint64_t* dst, nunet_stream_t s) {
  NUNET_REQUIRE(dst, "debug_stamp: null pointer");
  NUNET_LAUNCH(stamp_kernel, dim3(1), dim3(1), 0, (hipStream_t)s, (unsigned long long*)dst);
  return nunet_check_launch("debug_stamp");
}

extern "C" int nunet_plan_set_lanes(nunet_plan* P, nunet_stream_t* lanes, int32_t n) {
  NUNET_REQUIRE(P && lanes && n >= 1, "plan_set_lanes: bad args");
  PlanRt* rt = rt_of(P);
  for (int l = 0; l < NLANES; ++l) rt->lanes[l] = (hipStream_t)lanes[l % n];   // caller-owned streams
  rt->lanes_ok = true;
  rt->lanes_external = true;
  return NUNET_OK;
}

extern "C" int nunet_plan_set_schedule(nunet_plan* P, int32_t schedule) {
  NUNET_REQUIRE(P && (schedule == NUNET_SCHEDULE_LANES || schedule == NUNET_SCHEDULE_WAVE || schedule == NUNET_SCHEDULE_LIST), "plan_set_schedule: bad args");
  rt_of(P)->wave = schedule;
  return NUNET_OK;
}

extern "C" int nunet_plan_set_multistream(nunet_plan* P, int32_t enable) {
  NUNET_REQUIRE(P, "plan_set_multistream: null plan");
  rt_of(P)->multistream = enable;
  return NUNET_OK;
}
