// conv3x3.hip — 3x3 / pad 1 convolution on MFMA, NHWC, im2col-free.
//
// Replaces nn.Conv2d(ci, co, 3, padding=1) forward and its autograd
// (reference finished/archs1.py:18,20; autograd at trains.py:132).
//
//   fwd / dgrad : implicit GEMM  D[pixel][co] = sum_{tap,ci} X[pixel+tap][ci] * W[tap][co][ci]
//                 A operand = halo tile of the NHWC input staged once in LDS and read
//                 at 9 shifted addresses; B operand = KRSC weights staged in LDS.
//   wgrad       : dW[tap][co][ci] = sum_pixel dY[pixel][co] * X[pixel+tap][ci]
//                 contraction over pixels -> both operands are read with the
//                 gfx950 transposing LDS read (ds_read_b64_tr_b16) for 16-bit types.
//
// MFMA shapes: 32x32x16 (bf16 / f16) and 32x32x2 (f32, exact fp32).
#include "common.h"

// ---------------------------------------------------------------------------
// MFMA wrappers: one "k-step" = 16 contraction elements on a 32x32 tile.
// Lane (r = lane&31, h = lane>>5) owns contraction elements [8h, 8h+8) of the
// k-step for row/col r. For f32 the 8 elements are fed to 8 successive
// 32x32x2 MFMAs (any k permutation is valid as long as A and B agree).
// ---------------------------------------------------------------------------
template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  typedef bf16x8 Frag;
  static __device__ __forceinline__ Frag load(const bf16_t* p) { return *reinterpret_cast<const bf16x8*>(p); }
  static __device__ __forceinline__ void mma(f32x16& acc, const Frag& a, const Frag& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  }
};
template <> struct Mma<f16_t> {
  typedef f16x8 Frag;
  static __device__ __forceinline__ Frag load(const f16_t* p) { return *reinterpret_cast<const f16x8*>(p); }
  static __device__ __forceinline__ void mma(f32x16& acc, const Frag& a, const Frag& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  struct Frag { f32x4 lo, hi; };
  static __device__ __forceinline__ Frag load(const float* p) {
    Frag f;
    f.lo = *reinterpret_cast<const f32x4*>(p);
    f.hi = *reinterpret_cast<const f32x4*>(p + 4);
    return f;
  }
  static __device__ __forceinline__ void mma(f32x16& acc, const Frag& a, const Frag& b) {
#pragma unroll
    for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.lo[t], b.lo[t], acc, 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.hi[t], b.hi[t], acc, 0, 0, 0);
  }
};

// row of accumulator register `reg` for lane half h (32x32 C/D layout)
__device__ __forceinline__ int acc_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

// ---------------------------------------------------------------------------
// forward / dgrad kernel
// ---------------------------------------------------------------------------
struct ConvP {
  const void* src0; const void* src1;
  int C0, C1, P0, P1;
  const void* w; const float* bias;
  void* dst0; void* dst1;
  int D0, D1, Q0, Q1;
  int slot_w; unsigned acc0_mask; int acc1;
  float* stats;
  int N, H, W, Cin, Cout;
  int NI, TH, TW, tilesX, tilesY, tilesG, nCoT;
};

template <typename T, int WM_, int WN_, int SM_, int SN_> struct ConvCfg {
  static constexpr int WM = WM_, WN = WN_, SM = SM_, SN = SN_;
  static constexpr int NT = 64 * WM * WN;
  static constexpr int BM = 32 * SM * WM;
  static constexpr int BN = 32 * SN * WN;
  static constexpr int EPV = Tr<T>::EPV;
  static constexpr int KC = 64 / (int)sizeof(T);  // channels per LDS chunk (64 B per pixel)
  static constexpr int PS = KC + EPV;             // padded pixel stride (80 B)
  static constexpr int HPMAX = BM + BM / 2;       // halo pixel capacity
  static constexpr int UH = (HPMAX * 4 + NT - 1) / NT;
  static constexpr int UW = (9 * BN * 4 + NT - 1) / NT;
};

template <typename T, int WM, int WN, int SM, int SN>
__global__ __launch_bounds__(64 * WM * WN) void conv3x3_kernel(ConvP p) {
  typedef ConvCfg<T, WM, WN, SM, SN> C;
  typedef Mma<T> M;
  constexpr int PS = C::PS, EPV = C::EPV, BN = C::BN, BM = C::BM, NT = C::NT;

  __shared__ __attribute__((aligned(16))) T s_halo[C::HPMAX * PS];
  __shared__ __attribute__((aligned(16))) T s_w[9 * BN * PS];
  __shared__ int s_hidx[BM];
  __shared__ int s_gpix[BM];
  __shared__ float s_red[2 * WM * BN];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 31, h = lane >> 5;

  int bid = blockIdx.x;
  const int cot = bid % p.nCoT; bid /= p.nCoT;
  const int tx = bid % p.tilesX; bid /= p.tilesX;
  const int ty = bid % p.tilesY;
  const int tg = bid / p.tilesY;
  const int x0 = tx * p.TW, y0 = ty * p.TH, n0 = tg * p.NI;
  const int co0 = cot * BN;
  const int HW2 = p.TW + 2, HH2 = p.TH + 2;
  const int HP = p.NI * HH2 * HW2;
  const int THW = p.TH * p.TW;

  for (int m = tid; m < BM; m += NT) {
    const int ni = m / THW;
    const int rem = m - ni * THW;
    const int ly = rem / p.TW, lx = rem - ly * p.TW;
    const int n = n0 + ni, y = y0 + ly, x = x0 + lx;
    const bool ok = ni < p.NI && n < p.N && y < p.H && x < p.W;
    s_hidx[m] = ok ? ((ni * HH2 + ly + 1) * HW2 + lx + 1) : (HW2 + 1);
    s_gpix[m] = ok ? ((n * p.H + y) * p.W + x) : -1;
  }

  // per-thread staging units (fixed across channel chunks)
  int hgp[C::UH];
#pragma unroll
  for (int k = 0; k < C::UH; ++k) {
    const int u = tid + k * NT;
    const int hp = u >> 2;
    int gp = -2;
    if (hp < HP) {
      const int ni = hp / (HH2 * HW2);
      const int rem = hp - ni * (HH2 * HW2);
      const int hy = rem / HW2, hx = rem - hy * HW2;
      const int n = n0 + ni, y = y0 + hy - 1, x = x0 + hx - 1;
      gp = (n < p.N && y >= 0 && y < p.H && x >= 0 && x < p.W) ? ((n * p.H + y) * p.W + x) : -1;
    }
    hgp[k] = gp;
  }

  Vec16<T> hreg[C::UH];
  Vec16<T> wreg[C::UW];

  auto load_regs = [&](int kb, int kc) {
    const T* src; int ch, pitch;
    if (kb < p.C0) { src = (const T*)p.src0; ch = kb; pitch = p.P0; }
    else { src = (const T*)p.src1; ch = kb - p.C0; pitch = p.P1; }
#pragma unroll
    for (int k = 0; k < C::UH; ++k) {
      const int seg = (tid + k * NT) & 3;
      if (hgp[k] >= 0 && seg * EPV < kc) hreg[k] = ld16(src + (size_t)hgp[k] * pitch + ch + seg * EPV);
      else hreg[k] = zero16<T>();
    }
#pragma unroll
    for (int k = 0; k < C::UW; ++k) {
      const int u = tid + k * NT;
      const int row = u >> 2, seg = u & 3;
      if (row < 9 * BN && seg * EPV < kc) {
        const int tap = row / BN, co = row - tap * BN;
        wreg[k] = ld16((const T*)p.w + ((size_t)(tap * p.Cout + co0 + co)) * p.Cin + kb + seg * EPV);
      } else wreg[k] = zero16<T>();
    }
  };
  auto write_lds = [&]() {
#pragma unroll
    for (int k = 0; k < C::UH; ++k) {
      const int u = tid + k * NT;
      if (hgp[k] != -2) st16(&s_halo[(u >> 2) * PS + (u & 3) * EPV], hreg[k]);
    }
#pragma unroll
    for (int k = 0; k < C::UW; ++k) {
      const int u = tid + k * NT;
      if ((u >> 2) < 9 * BN) st16(&s_w[(u >> 2) * PS + (u & 3) * EPV], wreg[k]);
    }
  };

  f32x16 acc[SM][SN];
#pragma unroll
  for (int a = 0; a < SM; ++a)
#pragma unroll
    for (int b = 0; b < SN; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

  int kb = 0;
  int kc = min(C::KC, (kb < p.C0 ? p.C0 : p.Cin) - kb);
  load_regs(kb, kc);
  __syncthreads();  // tables visible
  int abase[SM];
#pragma unroll
  for (int a = 0; a < SM; ++a) abase[a] = s_hidx[(wm * SM + a) * 32 + r] * PS;

  while (kb < p.Cin) {
    __syncthreads();  // previous chunk's reads done
    write_lds();
    __syncthreads();
    const int kc_cur = kc;
    kb += kc_cur;
    if (kb < p.Cin) {
      kc = min(C::KC, (kb < p.C0 ? p.C0 : p.Cin) - kb);
      load_regs(kb, kc);
    }
    const int ksteps = kc_cur >> 4;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int toff = ((tap / 3 - 1) * HW2 + (tap % 3 - 1)) * PS;
      for (int ks = 0; ks < ksteps; ++ks) {
        typename M::Frag fb[SN], fa[SM];
#pragma unroll
        for (int b = 0; b < SN; ++b)
          fb[b] = M::load(&s_w[(tap * BN + (wn * SN + b) * 32 + r) * PS + ks * 16 + 8 * h]);
#pragma unroll
        for (int a = 0; a < SM; ++a) fa[a] = M::load(&s_halo[abase[a] + toff + ks * 16 + 8 * h]);
#pragma unroll
        for (int a = 0; a < SM; ++a)
#pragma unroll
          for (int b = 0; b < SN; ++b) M::mma(acc[a][b], fa[a], fb[b]);
      }
    }
  }

  // ---- epilogue: bias, optional accumulate, store, BN partial sums ----------
#pragma unroll
  for (int b = 0; b < SN; ++b) {
    const int co = co0 + (wn * SN + b) * 32 + r;
    const float bias = p.bias ? p.bias[co] : 0.f;
    T* dst; int pitch, cc; bool accum;
    if (co < p.D0) {
      dst = (T*)p.dst0; pitch = p.Q0; cc = co;
      accum = (p.acc0_mask >> (p.slot_w > 0 ? co / p.slot_w : 0)) & 1u;
    } else {
      dst = (T*)p.dst1; pitch = p.Q1; cc = co - p.D0; accum = p.acc1 != 0;
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int a = 0; a < SM; ++a) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int m = (wm * SM + a) * 32 + acc_row(i, h);
        const int gp = s_gpix[m];
        if (gp >= 0) {
          T* q = dst + (size_t)gp * pitch + cc;
          float v = acc[a][b][i] + bias;
          if (accum) v += to_f32(*q);
          const T tv = from_f32<T>(v);
          *q = tv;
          const float d = to_f32(tv) - bias;
          s1 += d; s2 += d * d;
        }
      }
    }
    if (p.stats) {
      s1 += __shfl_xor(s1, 32);
      s2 += __shfl_xor(s2, 32);
      if (h == 0) {
        s_red[(wm * BN + (wn * SN + b) * 32 + r) * 2 + 0] = s1;
        s_red[(wm * BN + (wn * SN + b) * 32 + r) * 2 + 1] = s2;
      }
    }
  }
  if (p.stats) {
    __syncthreads();
    for (int c = tid; c < BN; c += NT) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int k = 0; k < WM; ++k) { s1 += s_red[(k * BN + c) * 2]; s2 += s_red[(k * BN + c) * 2 + 1]; }
      atomicAdd(&p.stats[co0 + c], s1);
      atomicAdd(&p.stats[p.Cout + co0 + c], s2);
    }
  }
}

// tile-geometry chooser shared by fwd and wgrad
struct TileGeom { int NI, TH, TW, tilesX, tilesY, tilesG; };
TileGeom nunet_choose_tile(int N, int H, int W, int BM, int HPMAX) {
  TileGeom best{1, 1, 1, W, H, N};
  double best_score = -1.0;
  for (int tw = 1; tw <= W && tw <= BM; ++tw) {
    for (int th = 1; th <= H && th * tw <= BM; ++th) {
      int nimax = (th == H && tw == W) ? BM / (H * W) : 1;
      if (nimax > N) nimax = N;
      for (int ni = 1; ni <= nimax; ++ni) {
        const long hp = (long)ni * (th + 2) * (tw + 2);
        if (hp > HPMAX) continue;
        const int tX = ceil_div(W, tw), tY = ceil_div(H, th), tG = ceil_div(N, ni);
        const double util = (double)N * H * W / ((double)tX * tY * tG * BM);
        const double halo = (double)hp / ((double)ni * th * tw);
        // prefer utilisation, then low halo overhead, then wide rows (coalescing)
        const double score = util * 1000.0 - halo * 10.0 + (tw >= 16 ? 1.0 : 0.0);
        if (score > best_score) { best_score = score; best = TileGeom{ni, th, tw, tX, tY, tG}; }
      }
    }
  }
  return best;
}

template <typename T, int WM, int WN, int SM, int SN>
static int launch_conv_cfg(const nunet_conv_desc* d, hipStream_t st) {
  typedef ConvCfg<T, WM, WN, SM, SN> C;
  ConvP p;
  p.src0 = d->src0; p.src1 = d->src1; p.C0 = d->C0; p.C1 = d->C1; p.P0 = d->P0; p.P1 = d->P1;
  p.w = d->wpack; p.bias = d->bias;
  p.dst0 = d->dst0; p.dst1 = d->dst1; p.D0 = d->D0; p.D1 = d->D1; p.Q0 = d->Q0; p.Q1 = d->Q1;
  p.slot_w = d->acc_slot_w; p.acc0_mask = d->acc0_mask; p.acc1 = d->acc1;
  p.stats = d->stats;
  p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->C0 + d->C1; p.Cout = d->D0 + d->D1;
  const TileGeom g = nunet_choose_tile(d->N, d->H, d->W, C::BM, C::HPMAX);
  p.NI = g.NI; p.TH = g.TH; p.TW = g.TW; p.tilesX = g.tilesX; p.tilesY = g.tilesY; p.tilesG = g.tilesG;
  p.nCoT = p.Cout / C::BN;
  const long grid = (long)p.nCoT * g.tilesX * g.tilesY * g.tilesG;
  const double px = (double)d->N * d->H * d->W;
  const int acin = g_prof_alg_cin > 0 ? g_prof_alg_cin : p.Cin;
  ProfScope ps(C::BN == 32 ? PC_CONV_M256N32 : PC_CONV_M128N64, 2.0 * 9 * acin * p.Cout * px,
               (px * (acin + p.Cout) + 9.0 * acin * p.Cout) * sizeof(T), st);
  hipLaunchKernelGGL((conv3x3_kernel<T, WM, WN, SM, SN>), dim3((unsigned)grid), dim3(C::NT), 0, st, p);
  return nunet_check_launch("conv3x3");
}

template <typename T> static int launch_conv(const nunet_conv_desc* d, hipStream_t st) {
  const int cout = d->D0 + d->D1;
  if (cout % 64 == 0) return launch_conv_cfg<T, 2, 2, 2, 1>(d, st);  // BM 128, BN 64
  return launch_conv_cfg<T, 4, 1, 2, 1>(d, st);                      // BM 256, BN 32
}

extern "C" int nunet_conv3x3_fwd(const nunet_conv_desc* d, nunet_stream_t s) {
  NUNET_REQUIRE(d && d->src0 && d->wpack && d->dst0, "conv3x3: null pointer");
  const int cin = d->C0 + d->C1, cout = d->D0 + d->D1;
  NUNET_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0, "conv3x3: bad extent %dx%dx%d", d->N, d->H, d->W);
  NUNET_REQUIRE(d->C0 > 0 && d->C0 % 16 == 0 && d->C1 % 16 == 0, "conv3x3: C0=%d C1=%d must be multiples of 16", d->C0, d->C1);
  NUNET_REQUIRE(d->C1 == 0 || d->src1, "conv3x3: src1 null with C1=%d", d->C1);
  NUNET_REQUIRE(cout % 32 == 0 && d->D0 % 32 == 0 && d->D1 % 32 == 0, "conv3x3: D0=%d D1=%d must be multiples of 32", d->D0, d->D1);
  NUNET_REQUIRE(d->D1 == 0 || d->dst1, "conv3x3: dst1 null with D1=%d", d->D1);
  NUNET_REQUIRE(d->P0 >= d->C0 && (d->C1 == 0 || d->P1 >= d->C1) && d->Q0 >= d->D0 && (d->D1 == 0 || d->Q1 >= d->D1), "conv3x3: pitch smaller than channels");
  const int epv = 16 / dtype_size(d->dtype);
  NUNET_REQUIRE(d->P0 % epv == 0 && (d->C1 == 0 || d->P1 % epv == 0), "conv3x3: source pitch must keep 16-byte alignment");
  NUNET_REQUIRE(d->acc_slot_w == 0 || d->acc_slot_w % 32 == 0, "conv3x3: acc_slot_w %d", d->acc_slot_w);
  NUNET_REQUIRE((long)d->N * d->H * d->W < (1L << 30) && cin <= 4096, "conv3x3: problem too large for 32-bit pixel indices");
  return NUNET_DISPATCH(d->dtype, launch_conv, d, (hipStream_t)s);
}

// ---------------------------------------------------------------------------
// wgrad kernel
// ---------------------------------------------------------------------------
struct WgP {
  const void* src0; const void* src1;
  int C0, C1, P0, P1;
  const void* dy; int Cout, PY;
  float* dw;
  int N, H, W, Cin;
  int NI, TH, TW, tilesX, tilesY, tilesG;
  int nCoT, nCiT, ksplit, nMT;
};

// smallest row stride (elements) >= c with (2*stride) % 128 == 64: conflict-free
// for ds_read_b64_tr_b16 (4 rows x 64 B per 32-lane half land on distinct banks)
constexpr int tr_stride(int c) { return ((c - 32 + 63) / 64) * 64 + 32; }

template <typename T, int WCO_, int WCI_> struct WgCfg {
  static constexpr int WCO = WCO_, WCI = WCI_;
  static constexpr int NT = 64 * WCO * WCI;
  static constexpr int BCO = 32 * WCO, BCI = 32 * WCI;
  static constexpr int BM = 128, HPMAX = 192;
  static constexpr int EPV = Tr<T>::EPV;
  static constexpr bool F32 = std::is_same<T, float>::value;
  static constexpr int SY = F32 ? BCO : tr_stride(BCO);
  static constexpr int SA = F32 ? BCI : tr_stride(BCI);
};

template <typename T>
__device__ __forceinline__ s16x4 tr_read(const T* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4*)(const_cast<T*>(p)));
}

template <typename T> struct Frag16;
template <> struct Frag16<bf16_t> { typedef bf16x8 V; };
template <> struct Frag16<f16_t> { typedef f16x8 V; };

template <typename T, int WCO, int WCI>
__global__ __launch_bounds__(64 * WCO * WCI) void wgrad_kernel(WgP p) {
  typedef WgCfg<T, WCO, WCI> C;
  constexpr int NT = C::NT, BM = C::BM, EPV = C::EPV, SY = C::SY, SA = C::SA;
  __shared__ __attribute__((aligned(16))) T s_dy[BM * SY];
  __shared__ __attribute__((aligned(16))) T s_a[C::HPMAX * SA];
  __shared__ int s_hidx[BM];
  __shared__ int s_gpix[BM];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wco = wave / WCI, wci = wave % WCI;
  const int r = lane & 31, h = lane >> 5;

  int bid = blockIdx.x;
  const int cit = bid % p.nCiT; bid /= p.nCiT;
  const int cot = bid % p.nCoT;
  const int split = bid / p.nCoT;
  const int co0 = cot * C::BCO, ci0 = cit * C::BCI;
  const int HW2 = p.TW + 2, HH2 = p.TH + 2;
  const int HP = p.NI * HH2 * HW2;
  const int THW = p.TH * p.TW;

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

  for (int mt = split; mt < p.nMT; mt += p.ksplit) {
    int b2 = mt;
    const int tx = b2 % p.tilesX; b2 /= p.tilesX;
    const int ty = b2 % p.tilesY;
    const int tg = b2 / p.tilesY;
    const int x0 = tx * p.TW, y0 = ty * p.TH, n0 = tg * p.NI;

    __syncthreads();  // previous tile fully consumed
    for (int m = tid; m < BM; m += NT) {
      const int ni = m / THW;
      const int rem = m - ni * THW;
      const int ly = rem / p.TW, lx = rem - ly * p.TW;
      const int n = n0 + ni, y = y0 + ly, x = x0 + lx;
      const bool ok = ni < p.NI && n < p.N && y < p.H && x < p.W;
      s_hidx[m] = ok ? ((ni * HH2 + ly + 1) * HW2 + lx + 1) : (HW2 + 1);
      s_gpix[m] = ok ? ((n * p.H + y) * p.W + x) : -1;
    }
    __syncthreads();
    // stage dY tile [BM][BCO]
    constexpr int UY = C::BCO / EPV;
    for (int u = tid; u < BM * UY; u += NT) {
      const int m = u / UY, seg = u - m * UY;
      const int gp = s_gpix[m];
      const int co = co0 + seg * EPV;
      Vec16<T> v = zero16<T>();
      if (gp >= 0 && co < p.Cout) v = ld16((const T*)p.dy + (size_t)gp * p.PY + co);
      st16(&s_dy[m * SY + seg * EPV], v);
    }
    // stage X halo [HP][BCI]
    constexpr int UA = C::BCI / EPV;
    for (int u = tid; u < HP * UA; u += NT) {
      const int hp = u / UA, seg = u - hp * UA;
      const int ni = hp / (HH2 * HW2);
      const int rem = hp - ni * (HH2 * HW2);
      const int hy = rem / HW2, hx = rem - hy * HW2;
      const int n = n0 + ni, y = y0 + hy - 1, x = x0 + hx - 1;
      const int c = ci0 + seg * EPV;
      Vec16<T> v = zero16<T>();
      if (n < p.N && y >= 0 && y < p.H && x >= 0 && x < p.W && c < p.Cin) {
        const size_t gp = ((size_t)n * p.H + y) * p.W + x;
        v = (c < p.C0) ? ld16((const T*)p.src0 + gp * p.P0 + c)
                       : ld16((const T*)p.src1 + gp * p.P1 + (c - p.C0));
      }
      st16(&s_a[hp * SA + seg * EPV], v);
    }
    __syncthreads();

    if constexpr (C::F32) {
      for (int k0 = 0; k0 < BM; k0 += 2) {
        const int m = k0 + h;
        const float av = s_dy[m * SY + wco * 32 + r];
        const int hx = s_hidx[m];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          const int toff = (tap / 3 - 1) * HW2 + (tap % 3 - 1);
          const float bv = s_a[(hx + toff) * SA + wci * 32 + r];
          acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[tap], 0, 0, 0);
        }
      }
    } else {
      typedef typename Frag16<T>::V FV;
      const int q = (lane >> 2) & 3;                       // row within the 4-row block
      const int colo = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);  // column offset supplied by this lane
      for (int k0 = 0; k0 < BM; k0 += 16) {
        const int m0 = k0 + 8 * h + q, m1 = m0 + 4;
        const int hx0 = s_hidx[m0], hx1 = s_hidx[m1];
        const s16x4 a0 = tr_read(&s_dy[m0 * SY + wco * 32 + colo]);
        const s16x4 a1 = tr_read(&s_dy[m1 * SY + wco * 32 + colo]);
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        const s16x8 av = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          const int toff = (tap / 3 - 1) * HW2 + (tap % 3 - 1);
          const s16x4 b0 = tr_read(&s_a[(hx0 + toff) * SA + wci * 32 + colo]);
          const s16x4 b1 = tr_read(&s_a[(hx1 + toff) * SA + wci * 32 + colo]);
          const s16x8 bv = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
          Mma<T>::mma(acc[tap], __builtin_bit_cast(FV, av), __builtin_bit_cast(FV, bv));
        }
      }
    }
  }

  // atomically accumulate the partial dW tile
  const int ci = ci0 + wci * 32 + r;
  if (ci < p.Cin) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int co = co0 + wco * 32 + acc_row(i, h);
        if (co < p.Cout) atomicAdd(&p.dw[((size_t)tap * p.Cout + co) * p.Cin + ci], acc[tap][i]);
      }
    }
  }
}

template <typename T, int WCO, int WCI>
static int launch_wgrad_cfg(const nunet_wgrad_desc* d, hipStream_t st) {
  typedef WgCfg<T, WCO, WCI> C;
  WgP p;
  p.src0 = d->src0; p.src1 = d->src1; p.C0 = d->C0; p.C1 = d->C1; p.P0 = d->P0; p.P1 = d->P1;
  p.dy = d->dy; p.Cout = d->Cout; p.PY = d->PY; p.dw = d->dw;
  p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->C0 + d->C1;
  const TileGeom g = nunet_choose_tile(d->N, d->H, d->W, C::BM, C::HPMAX);
  p.NI = g.NI; p.TH = g.TH; p.TW = g.TW; p.tilesX = g.tilesX; p.tilesY = g.tilesY; p.tilesG = g.tilesG;
  p.nCoT = ceil_div(p.Cout, C::BCO);
  p.nCiT = ceil_div(p.Cin, C::BCI);
  p.nMT = g.tilesX * g.tilesY * g.tilesG;
  const int otiles = p.nCoT * p.nCiT;
  int ks = ceil_div(512, otiles);
  if (ks > p.nMT) ks = p.nMT;
  if (ks < 1) ks = 1;
  p.ksplit = ks;
  const long grid = (long)otiles * ks;
  const double px = (double)d->N * d->H * d->W;
  const int acin = g_prof_alg_cin > 0 ? g_prof_alg_cin : p.Cin;
  ProfScope ps(WCO == 1 ? PC_WGRAD_1x4 : PC_WGRAD_2x2, 2.0 * 9 * acin * p.Cout * px,
               px * (acin + p.Cout) * sizeof(T) + 9.0 * acin * p.Cout * 4, st);
  hipLaunchKernelGGL((wgrad_kernel<T, WCO, WCI>), dim3((unsigned)grid), dim3(C::NT), 0, st, p);
  return nunet_check_launch("wgrad3x3");
}

template <typename T> static int launch_wgrad(const nunet_wgrad_desc* d, hipStream_t st) {
  if (d->Cout % 64 == 0) return launch_wgrad_cfg<T, 2, 2>(d, st);
  return launch_wgrad_cfg<T, 1, 4>(d, st);
}

extern "C" int nunet_conv3x3_wgrad(const nunet_wgrad_desc* d, nunet_stream_t s) {
  NUNET_REQUIRE(d && d->src0 && d->dy && d->dw, "wgrad: null pointer");
  NUNET_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0, "wgrad: bad extent");
  NUNET_REQUIRE(d->C0 > 0 && d->C0 % 16 == 0 && d->C1 % 16 == 0, "wgrad: C0=%d C1=%d must be multiples of 16", d->C0, d->C1);
  NUNET_REQUIRE(d->C1 == 0 || d->src1, "wgrad: src1 null");
  NUNET_REQUIRE(d->Cout % 32 == 0, "wgrad: Cout=%d must be a multiple of 32", d->Cout);
  const int epv = 16 / dtype_size(d->dtype);
  NUNET_REQUIRE(d->P0 % epv == 0 && (d->C1 == 0 || d->P1 % epv == 0) && d->PY % epv == 0, "wgrad: pitch alignment");
  NUNET_REQUIRE((long)d->N * d->H * d->W < (1L << 30), "wgrad: too many pixels");
  return NUNET_DISPATCH(d->dtype, launch_wgrad, d, (hipStream_t)s);
}
