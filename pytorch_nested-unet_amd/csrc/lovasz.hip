// lovasz.hip — LovaszHingeLoss (reference losses.py:49-96,120-129; per_image=True) on device.
// One 1024-thread workgroup per image: errors e = 1 - x*(2t-1), bitonic sort (descending) of
// 64-bit (sortable key | pixel index | label) words in LDS, inclusive scan of the sorted labels,
// Jaccard increments (lovasz_grad, losses.py:49-61), loss = sum relu(e_k) * g_k and the
// sub-gradient d loss / d x[perm_k] = -(2t-1) * g_k * [e_k > 0] written in the same pass.
// Images of up to 16384 pixels sort in LDS in that one kernel; larger ones (256x256, 512x512: BASELINE cfg4/cfg5
// geometries) go through the global-memory pipeline at the end of this file (chunk sorts in LDS + global bitonic
// merge passes + chunked scan), up to 2^22 pixels per image.
#include "common.h"

#define LOVASZ_NMAX 16384
#define LOVASZ_NT 1024

__device__ __forceinline__ uint32_t f32_sortable(float f) {
  const uint32_t u = __builtin_bit_cast(uint32_t, f);
  return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);   // ascending unsigned order == ascending float order
}
__device__ __forceinline__ float sortable_f32(uint32_t s) {
  const uint32_t u = s ^ ((s >> 31) ? 0x80000000u : 0xFFFFFFFFu);
  return __builtin_bit_cast(float, u);
}

__global__ __launch_bounds__(LOVASZ_NT) void lovasz_hinge_kernel(const float* __restrict__ x, const float* __restrict__ t, int P, int NP2,
                                                                float* __restrict__ dx, float* __restrict__ loss_img, float inv_batch) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long s_k[];   // NP2 words
  __shared__ float s_part[LOVASZ_NT];
  __shared__ float s_red[LOVASZ_NT / 64];
  const int img = blockIdx.x, tid = threadIdx.x;
  const float* xs = x + (size_t)img * P;
  const float* ts = t + (size_t)img * P;
  float* ds = dx + (size_t)img * P;
  // 1) keys: [63:32] sortable error, [15:1] pixel index, [0] label; padding sorts last (key 0 = -max)
  for (int i = tid; i < NP2; i += LOVASZ_NT) {
    unsigned long long w = 0ull;
    if (i < P) {
      const float lab = ts[i];
      const float e = 1.f - xs[i] * (2.f * lab - 1.f);
      w = ((unsigned long long)f32_sortable(e) << 32) | ((unsigned long long)i << 1) | (lab > 0.5f ? 1ull : 0ull);
    }
    s_k[i] = w;
  }
  __syncthreads();
  // 2) bitonic sort, descending
  for (int k = 2; k <= NP2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int q = tid; q < NP2 / 2; q += LOVASZ_NT) {
        const int i = 2 * q - (q & (j - 1));
        const int p2 = i + j;
        const unsigned long long a = s_k[i], b = s_k[p2];
        const bool desc = (i & k) == 0;            // this run is sorted descending
        if (desc ? (a < b) : (a > b)) { s_k[i] = b; s_k[p2] = a; }
      }
      __syncthreads();
    }
  }
  // 3) inclusive scan of the sorted labels: thread owns a contiguous run of NP2/NT elements
  const int per = NP2 / LOVASZ_NT > 0 ? NP2 / LOVASZ_NT : 1;
  const int lo = tid * per;
  float run = 0.f;
  for (int i = lo; i < lo + per && i < NP2; ++i) run += (float)(s_k[i] & 1ull);
  s_part[tid] = run;
  __syncthreads();
  // exclusive prefix of the per-thread sums (Hillis-Steele over 1024 entries)
  for (int off = 1; off < LOVASZ_NT; off <<= 1) {
    const float v = tid >= off ? s_part[tid - off] : 0.f;
    __syncthreads();
    s_part[tid] += v;
    __syncthreads();
  }
  const float gts = s_part[LOVASZ_NT - 1];
  float cum = s_part[tid] - run;                   // labels before this thread's run
  // 4) Jaccard increments, loss and gradient
  float acc = 0.f;
  float jprev;
  {
    // jaccard just before this run (index lo-1); 0 contribution convention for lo == 0 handled below
    const float kprev = (float)lo;                 // number of elements before the run
    const float inter = gts - cum, uni = gts + (kprev - cum);
    jprev = lo > 0 ? 1.f - inter / uni : 0.f;
  }
  for (int i = lo; i < lo + per && i < NP2; ++i) {
    const unsigned long long w = s_k[i];
    const float lab = (float)(w & 1ull);
    cum += lab;
    const float inter = gts - cum, uni = gts + ((float)(i + 1) - cum);
    const float jac = 1.f - inter / uni;
    const float g = i == 0 ? jac : jac - jprev;
    jprev = jac;
    const int idx = (int)((w >> 1) & 0x7FFFull);
    const float e = sortable_f32((uint32_t)(w >> 32));
    if (i < P && w != 0ull) {
      const bool on = e > 0.f;
      if (on) acc += e * g;
      ds[idx] = on ? -(2.f * lab - 1.f) * g * inv_batch : 0.f;
    }
  }
  acc = wave_sum(acc);
  if ((tid & 63) == 0) s_red[tid >> 6] = acc;
  __syncthreads();
  if (tid == 0) {
    float s = 0.f;
    for (int q = 0; q < LOVASZ_NT / 64; ++q) s += s_red[q];
    loss_img[img] = s;
  }
}

__global__ void lovasz_mean_kernel(const float* __restrict__ loss_img, int N, float* __restrict__ loss) {
  float s = 0.f;
  for (int i = threadIdx.x; i < N; i += 64) s += loss_img[i];
  s = wave_sum(s);
  if (threadIdx.x == 0) loss[0] = s / (float)N;     // mean() of losses.py:28-46
}

__global__ __launch_bounds__(256) void scale_kernel(const float* __restrict__ a, const float* __restrict__ g, float* __restrict__ o, int64_t n) {
  const float s = g ? g[0] : 1.f;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) o[i] = a[i] * s;
}

// ---------------------------------------------------------------------------------------------------------
// Large images: keys in global memory. key = [63:32] sortable error | [31:1] pixel index | [0] label
// (padding = 0 sorts last in descending order). Bitonic network over NP2 = 2^m keys per image:
//   stages k <= CHUNK entirely in LDS per 16384-key chunk (lv_local_kernel, first = true),
//   stage k > CHUNK: steps j >= CHUNK as global compare-exchange passes (lv_global_kernel), the remaining
//   steps j < CHUNK again in LDS per chunk (lv_local_kernel, first = false).
// Then labels are summed per chunk (lv_chunksum_kernel) and lv_final_kernel walks each chunk with its prefix.
// ---------------------------------------------------------------------------------------------------------
#define LV_CHUNK 16384

__global__ __launch_bounds__(256) void lv_keys_kernel(const float* __restrict__ x, const float* __restrict__ t, int P, int NP2, unsigned long long* __restrict__ keys) {
  const int img = blockIdx.y;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < NP2; i += gridDim.x * blockDim.x) {
    unsigned long long w = 0ull;
    if (i < P) {
      const float lab = t[(size_t)img * P + i];
      const float e = 1.f - x[(size_t)img * P + i] * (2.f * lab - 1.f);
      w = ((unsigned long long)f32_sortable(e) << 32) | ((unsigned long long)i << 1) | (lab > 0.5f ? 1ull : 0ull);
    }
    keys[(size_t)img * NP2 + i] = w;
  }
}

// one workgroup per (chunk, image); `first`: all stages k = 2..CHUNK; else: the steps j = CHUNK/2..1 of stage k
__global__ __launch_bounds__(LOVASZ_NT) void lv_local_kernel(unsigned long long* __restrict__ keys, int NP2, int first, int kstage) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long s_k[];   // LV_CHUNK words
  const int tid = threadIdx.x;
  const int base = blockIdx.x * LV_CHUNK;                                    // global index of the chunk's first key
  unsigned long long* g = keys + (size_t)blockIdx.y * NP2 + base;
  for (int i = tid; i < LV_CHUNK; i += LOVASZ_NT) s_k[i] = g[i];
  __syncthreads();
  for (int k = first ? 2 : kstage; k <= (first ? LV_CHUNK : kstage); k <<= 1) {
    for (int j = (k >> 1) < LV_CHUNK ? (k >> 1) : (LV_CHUNK >> 1); j > 0; j >>= 1) {
      for (int q = tid; q < LV_CHUNK / 2; q += LOVASZ_NT) {
        const int i = 2 * q - (q & (j - 1));
        const int p2 = i + j;
        const unsigned long long a = s_k[i], b = s_k[p2];
        const bool desc = ((base + i) & k) == 0;
        if (desc ? (a < b) : (a > b)) { s_k[i] = b; s_k[p2] = a; }
      }
      __syncthreads();
    }
  }
  for (int i = tid; i < LV_CHUNK; i += LOVASZ_NT) g[i] = s_k[i];
}

// step (k, j) with j >= LV_CHUNK: partners live in different chunks
__global__ __launch_bounds__(256) void lv_global_kernel(unsigned long long* __restrict__ keys, int NP2, int k, int j) {
  unsigned long long* g = keys + (size_t)blockIdx.y * NP2;
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < NP2 / 2; q += gridDim.x * blockDim.x) {
    const int i = 2 * q - (q & (j - 1));
    const int p2 = i + j;
    const unsigned long long a = g[i], b = g[p2];
    const bool desc = (i & k) == 0;
    if (desc ? (a < b) : (a > b)) { g[i] = b; g[p2] = a; }
  }
}

__global__ __launch_bounds__(LOVASZ_NT) void lv_chunksum_kernel(const unsigned long long* __restrict__ keys, int NP2, float* __restrict__ csum) {
  __shared__ float s_red[LOVASZ_NT / 64];
  const unsigned long long* g = keys + (size_t)blockIdx.y * NP2 + (size_t)blockIdx.x * LV_CHUNK;
  float acc = 0.f;
  for (int i = threadIdx.x; i < LV_CHUNK; i += LOVASZ_NT) acc += (float)(g[i] & 1ull);
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int q = 0; q < LOVASZ_NT / 64; ++q) s += s_red[q];
    csum[blockIdx.y * gridDim.x + blockIdx.x] = s;
  }
}

// Jaccard increments, loss partial and sub-gradient of one chunk (losses.py:49-61,64-96)
__global__ __launch_bounds__(LOVASZ_NT) void lv_final_kernel(const unsigned long long* __restrict__ keys, int P, int NP2, const float* __restrict__ csum,
                                                            float* __restrict__ dx, float* __restrict__ part, float inv_batch) {
  __shared__ float s_part[LOVASZ_NT];
  __shared__ float s_red[LOVASZ_NT / 64];
  const int tid = threadIdx.x, img = blockIdx.y, chunk = blockIdx.x, nch = gridDim.x;
  const unsigned long long* g = keys + (size_t)img * NP2 + (size_t)chunk * LV_CHUNK;
  float before = 0.f, gts = 0.f;
  for (int c = 0; c < nch; ++c) { const float v = csum[img * nch + c]; gts += v; if (c < chunk) before += v; }
  constexpr int per = LV_CHUNK / LOVASZ_NT;
  const int lo = tid * per;
  unsigned long long w[per];
  float run = 0.f;
#pragma unroll
  for (int q = 0; q < per; ++q) { w[q] = g[lo + q]; run += (float)(w[q] & 1ull); }
  s_part[tid] = run;
  __syncthreads();
  for (int off = 1; off < LOVASZ_NT; off <<= 1) {
    const float v = tid >= off ? s_part[tid - off] : 0.f;
    __syncthreads();
    s_part[tid] += v;
    __syncthreads();
  }
  float cum = before + s_part[tid] - run;            // labels before this thread's run (whole image)
  const long long gi0 = (long long)chunk * LV_CHUNK + lo;   // index of the run's first element in the sorted image
  float jprev = 0.f;
  if (gi0 > 0) { const float inter = gts - cum, uni = gts + ((float)gi0 - cum); jprev = 1.f - inter / uni; }
  float acc = 0.f;
  float* ds = dx + (size_t)img * P;
#pragma unroll
  for (int q = 0; q < per; ++q) {
    const long long gi = gi0 + q;
    const float lab = (float)(w[q] & 1ull);
    cum += lab;
    const float inter = gts - cum, uni = gts + ((float)(gi + 1) - cum);
    const float jac = 1.f - inter / uni;
    const float gk = gi == 0 ? jac : jac - jprev;
    jprev = jac;
    if (w[q] != 0ull) {
      const int idx = (int)((w[q] >> 1) & 0x7FFFFFFFull);
      const float e = sortable_f32((uint32_t)(w[q] >> 32));
      const bool on = e > 0.f;
      if (on) acc += e * gk;
      ds[idx] = on ? -(2.f * lab - 1.f) * gk * inv_batch : 0.f;
    }
  }
  acc = wave_sum(acc);
  if ((tid & 63) == 0) s_red[tid >> 6] = acc;
  __syncthreads();
  if (tid == 0) {
    float s = 0.f;
    for (int q = 0; q < LOVASZ_NT / 64; ++q) s += s_red[q];
    part[img * nch + chunk] = s;
  }
}

__global__ void lv_mean_kernel(const float* __restrict__ part, int total, int N, float* __restrict__ loss) {
  float s = 0.f;
  for (int i = threadIdx.x; i < total; i += 64) s += part[i];
  s = wave_sum(s);
  if (threadIdx.x == 0) loss[0] = s / (float)N;
}

static int lovasz_np2(int64_t per_image) {
  int np2 = 1;
  while (np2 < per_image) np2 <<= 1;
  return np2;
}
// workspace: per-image losses; for images above the LDS limit also the keys, chunk sums and chunk partials
extern "C" size_t nunet_lovasz_ws_bytes(int32_t N, int64_t per_image) {
  if (per_image <= LOVASZ_NMAX) return (size_t)N * sizeof(float);
  const size_t np2 = (size_t)lovasz_np2(per_image), nch = np2 / LV_CHUNK;
  return (size_t)N * np2 * 8 + 2 * (size_t)N * nch * sizeof(float) + 256;
}

// loss = mean over images of the Lovasz hinge; dlogits_unit = d loss / d logits (for an upstream gradient of 1)
extern "C" int nunet_lovasz_hinge_fwd(const float* logits, const float* target, int32_t N, int64_t per_image,
                                      float* ws, size_t ws_bytes, float* dlogits_unit, float* loss, nunet_stream_t s) {
  NUNET_REQUIRE(logits && target && ws && dlogits_unit && loss && N > 0 && per_image > 0, "lovasz_hinge_fwd: bad args");
  NUNET_REQUIRE(per_image <= (1LL << 22), "lovasz_hinge_fwd: %lld pixels per image exceed the limit of 2^22", (long long)per_image);
  NUNET_REQUIRE(ws_bytes >= nunet_lovasz_ws_bytes(N, per_image), "lovasz_hinge_fwd: workspace of %zu bytes, nunet_lovasz_ws_bytes = %zu", ws_bytes, nunet_lovasz_ws_bytes(N, per_image));
  NUNET_REQUIRE(((uintptr_t)ws & 7) == 0, "lovasz_hinge_fwd: workspace must be 8-byte aligned");
  hipStream_t st = (hipStream_t)s;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)lovasz_hinge_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LOVASZ_NMAX * 8);
    (void)hipFuncSetAttribute((const void*)lv_local_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LV_CHUNK * 8);
    attr_set = true;
  }
  ProfScope ps(PC_LOSS, 0, (double)N * per_image * 12, st);
  if (per_image <= LOVASZ_NMAX) {
    int np2 = lovasz_np2(per_image);
    if (np2 < 2 * LOVASZ_NT) np2 = 2 * LOVASZ_NT;      // every thread owns at least one compare pair / scan run
    const size_t lds = (size_t)np2 * 8;
    NUNET_LAUNCH(lovasz_hinge_kernel, dim3(N), dim3(LOVASZ_NT), lds, st, logits, target, (int)per_image, np2, dlogits_unit, ws, 1.f / (float)N);
    NUNET_LAUNCH(lovasz_mean_kernel, dim3(1), dim3(64), 0, st, ws, N, loss);
    return nunet_check_launch("lovasz_hinge_fwd");
  }
  const int np2 = lovasz_np2(per_image), nch = np2 / LV_CHUNK;
  unsigned long long* keys = reinterpret_cast<unsigned long long*>(ws);
  float* csum = reinterpret_cast<float*>(keys + (size_t)N * np2);
  float* part = csum + (size_t)N * nch;
  const unsigned gx = (unsigned)((np2 / 2 + 255) / 256 > 1024 ? 1024 : (np2 / 2 + 255) / 256);
  NUNET_LAUNCH(lv_keys_kernel, dim3(gx, N), dim3(256), 0, st, logits, target, (int)per_image, np2, keys);
  NUNET_LAUNCH(lv_local_kernel, dim3(nch, N), dim3(LOVASZ_NT), LV_CHUNK * 8, st, keys, np2, 1, 0);
  for (int k = 2 * LV_CHUNK; k <= np2; k <<= 1) {
    for (int j = k >> 1; j >= LV_CHUNK; j >>= 1)
      NUNET_LAUNCH(lv_global_kernel, dim3(gx, N), dim3(256), 0, st, keys, np2, k, j);
    NUNET_LAUNCH(lv_local_kernel, dim3(nch, N), dim3(LOVASZ_NT), LV_CHUNK * 8, st, keys, np2, 0, k);
  }
  NUNET_LAUNCH(lv_chunksum_kernel, dim3(nch, N), dim3(LOVASZ_NT), 0, st, keys, np2, csum);
  NUNET_LAUNCH(lv_final_kernel, dim3(nch, N), dim3(LOVASZ_NT), 0, st, keys, (int)per_image, np2, csum, dlogits_unit, part, 1.f / (float)N);
  NUNET_LAUNCH(lv_mean_kernel, dim3(1), dim3(64), 0, st, part, N * nch, N, loss);
  return nunet_check_launch("lovasz_hinge_fwd (global sort)");
}
extern "C" int nunet_lovasz_hinge_bwd(const float* dlogits_unit, const float* gscale, int64_t n, float* dlogits, nunet_stream_t s) {
  NUNET_REQUIRE(dlogits_unit && dlogits && n > 0, "lovasz_hinge_bwd: bad args");
  int64_t g = (n + 1023) / 1024;
  if (g > 2048) g = 2048;
  NUNET_LAUNCH(scale_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)s, dlogits_unit, gscale, dlogits, n);
  return nunet_check_launch("lovasz_hinge_bwd");
}


// ---------------------------------------------------------------------------------------------------------
// LovaszHingeLoss inside the fused training step (nunet_loss_step with NUNET_LOSS_LOVASZ_HINGE): the loss of every
// head (reference trains.py:118-123: their mean under deep supervision), d mean / d logits, the IoU counts of the
// last head (trains.py:124,128) and the epoch meters - the same outputs as the BCEDice form, so the step's graph
// does not care which loss it carries. Workspace: per head one nunet_lovasz_ws_bytes() region, then uint64[2] counts.
// ---------------------------------------------------------------------------------------------------------
static size_t lovasz_head_ws(int32_t N, int64_t per) { return (nunet_lovasz_ws_bytes(N, per) + 255) / 256 * 256; }
size_t lovasz_step_ws_bytes(int32_t N, int64_t per, int32_t heads) { return (size_t)heads * lovasz_head_ws(N, per) + 256; }

// dlogits (unit gradients of every head) *= 1 / heads; IoU counts of the last head (integer atomics: order-independent)
__global__ __launch_bounds__(256) void lovasz_step_scale_kernel(float* __restrict__ dx, const float* __restrict__ x_last, const float* __restrict__ t,
                                                                int64_t n_all, int64_t n_img, float inv_heads, float thr, unsigned long long* __restrict__ cnt) {
  unsigned ci = 0, cu = 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n_all; i += (int64_t)gridDim.x * blockDim.x) {
    dx[i] *= inv_heads;
    if (i < n_img) { const bool a = x_last[i] >= thr, b = t[i] > 0.5f; ci += (a && b) ? 1u : 0u; cu += (a || b) ? 1u : 0u; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { ci += __shfl_xor(ci, o); cu += __shfl_xor(cu, o); }
  if ((threadIdx.x & 63) == 0 && (ci | cu)) { atomicAdd(&cnt[0], (unsigned long long)ci); atomicAdd(&cnt[1], (unsigned long long)cu); }
}
__global__ void lovasz_step_final_kernel(float* __restrict__ loss_out, int heads, const unsigned long long* __restrict__ cnt, double* __restrict__ meters) {
  float mean = 0.f;
  for (int k = 0; k < heads; ++k) mean += loss_out[k];
  mean /= (float)heads;
  loss_out[heads] = mean;
  if (meters) {
    const double inter = (double)cnt[0], uni = (double)cnt[1];
    meters[0] += (double)mean;
    meters[1] += (inter + 1e-5) / (uni + 1e-5);
    meters[2] = inter; meters[3] = uni;
  }
}
int lovasz_loss_step(const float* logits, const float* target, int32_t N, int64_t per, int32_t heads, float* ws, float* dlogits,
                     float* loss_out, double* meters, float iou_thr, hipStream_t st) {
  const size_t hw = lovasz_head_ws(N, per);
  unsigned long long* cnt = reinterpret_cast<unsigned long long*>((char*)ws + (size_t)heads * hw);
  int rc = nunet_zero_async(cnt, 16, st);
  for (int k = 0; k < heads && rc == NUNET_OK; ++k)
    rc = nunet_lovasz_hinge_fwd(logits + (size_t)k * N * per, target, N, per, reinterpret_cast<float*>((char*)ws + (size_t)k * hw), hw,
                                dlogits + (size_t)k * N * per, loss_out + k, (nunet_stream_t)st);
  if (rc) return rc;
  const int64_t n_img = (int64_t)N * per, n_all = n_img * heads;
  int64_t g = (n_all + 1023) / 1024;
  if (g > 1024) g = 1024;
  NUNET_LAUNCH(lovasz_step_scale_kernel, dim3((unsigned)g), dim3(256), 0, st, dlogits, logits + (size_t)(heads - 1) * N * per, target, n_all, n_img,
               1.f / (float)heads, iou_thr, cnt);
  NUNET_LAUNCH(lovasz_step_final_kernel, dim3(1), dim3(1), 0, st, loss_out, (int)heads, cnt, meters);
  return nunet_check_launch("loss_step (lovasz hinge)");
}
