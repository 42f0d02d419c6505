// lovasz.hip — LovaszHingeLoss (reference losses.py:49-96,120-129; per_image=True) on device.
// One 1024-thread workgroup per image: errors e = 1 - x*(2t-1), bitonic sort (descending) of
// 64-bit (sortable key | pixel index | label) words in LDS, inclusive scan of the sorted labels,
// Jaccard increments (lovasz_grad, losses.py:49-61), loss = sum relu(e_k) * g_k and the
// sub-gradient d loss / d x[perm_k] = -(2t-1) * g_k * [e_k > 0] written in the same pass.
// Limit this round: pixels per image <= 16384 (128 KiB of LDS); larger images are refused.
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

extern "C" size_t nunet_lovasz_ws_bytes(int32_t N) { return (size_t)N * sizeof(float); }

// loss = mean over images of the Lovasz hinge; dlogits_unit = d loss / d logits (for an upstream gradient of 1)
extern "C" int nunet_lovasz_hinge_fwd(const float* logits, const float* target, int32_t N, int64_t per_image,
                                      float* ws, float* dlogits_unit, float* loss, nunet_stream_t s) {
  NUNET_REQUIRE(logits && target && ws && dlogits_unit && loss && N > 0 && per_image > 0, "lovasz_hinge_fwd: bad args");
  NUNET_REQUIRE(per_image <= LOVASZ_NMAX, "lovasz_hinge_fwd: %lld pixels per image exceed the in-LDS sort limit of %d", (long long)per_image, LOVASZ_NMAX);
  int np2 = 1;
  while (np2 < per_image) np2 <<= 1;
  if (np2 < 2 * LOVASZ_NT) np2 = 2 * LOVASZ_NT;      // every thread owns at least one compare pair / scan run
  hipStream_t st = (hipStream_t)s;
  const size_t lds = (size_t)np2 * 8;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)lovasz_hinge_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LOVASZ_NMAX * 8);
    attr_set = true;
  }
  ProfScope ps(PC_LOSS, 0, (double)N * per_image * 12, st);
  hipLaunchKernelGGL(lovasz_hinge_kernel, dim3(N), dim3(LOVASZ_NT), lds, st, logits, target, (int)per_image, np2, dlogits_unit, ws, 1.f / (float)N);
  hipLaunchKernelGGL(lovasz_mean_kernel, dim3(1), dim3(64), 0, st, ws, N, loss);
  return nunet_check_launch("lovasz_hinge_fwd");
}
extern "C" int nunet_lovasz_hinge_bwd(const float* dlogits_unit, const float* gscale, int64_t n, float* dlogits, nunet_stream_t s) {
  NUNET_REQUIRE(dlogits_unit && dlogits && n > 0, "lovasz_hinge_bwd: bad args");
  int64_t g = (n + 1023) / 1024;
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(scale_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)s, dlogits_unit, gscale, dlogits, n);
  return nunet_check_launch("lovasz_hinge_bwd");
}
