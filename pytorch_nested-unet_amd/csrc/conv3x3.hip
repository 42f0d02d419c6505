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
#include <stdlib.h>

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
// Tile-space -> global pixel (or -1 = zero padding / masked row).
// Regular tiling: (n, y, x) inside image n. Stacked-rows tiling (SH = H + 1): the batch is ONE virtual
// image of N * SH rows, virtual row v = n * SH + y + 1, rows with v % SH == 0 are zero separators that
// serve as bottom halo of image n - 1 and top halo of image n. Tiles are TH virtual rows x full width,
// so small images (12 x 12 at level 3 of a 96 x 96 input) fill a 128-pixel tile to 86 % instead of 56 %.
// SHinv = ceil(2^32 / SH): y / SH == umulhi(y, SHinv) exactly while y * SH < 2^32 (y < N * SH <= 2^20, SH <= 2^12 by the
// chooser's bound) - the compiler's generic 32-bit division is ~30 VALU instructions, paid per staging unit and tile.
__device__ __forceinline__ int map_pixel(int n, int y, int x, int N, int H, int W, int SH, unsigned SHinv) {
  // branch-free for both modes (selects, no early returns): callers map several staging units back to back
  const bool st = SH != 0;
  const int yc = y < 0 ? 0 : y;
  const int n2 = (int)__umulhi((unsigned)yc, SHinv);
  const int yy = yc - n2 * SH - 1;                 // stacked: row inside image n2, -1 on a separator row
  const int ni = st ? n2 : n, yi = st ? yy : y;
  const bool ok = (y >= 0) & (x >= 0) & (x < W) & (ni < N) & (yi >= 0) & (st | (y < H));
  return ok ? (ni * H + yi) * W + x : -1;
}

// n / d for a divisor fixed per launch: inv = ceil(2^32 / d) (0 for d == 1); exact while n * d < 2^32 (tile and item
// counts are < 2^20 here). The compiler's generic 32-bit division is ~30 instructions, also for uniform operands.
__host__ __device__ __forceinline__ unsigned fastdiv_inv(int d) { return d > 1 ? (unsigned)(((1ull << 32) + (unsigned)d - 1) / (unsigned)d) : 0u; }
__device__ __forceinline__ int fastdiv(int n, unsigned inv) { return inv ? (int)__umulhi((unsigned)n, inv) : n; }

// Branch-free, tiling mode known at compile time (wgrad maps seven staging units per 128-pixel tile: guarded early
// returns cost it a basic block and an exec-mask round trip per unit).
template <bool ST>
__device__ __forceinline__ int map_pixel_t(int n, int y, int x, int N, int H, int W, int SH, unsigned SHinv) {
  if constexpr (ST) {
    const int yc = y < 0 ? 0 : y;
    const int n2 = (int)__umulhi((unsigned)yc, SHinv), yy = yc - n2 * SH - 1;
    const bool ok = (y >= 0) & (x >= 0) & (x < W) & (n2 < N) & (yy >= 0);
    const int gp = (n2 * H + yy) * W + x;
    return ok ? gp : -1;
  } else {
    const bool ok = (n < N) & (y >= 0) & (y < H) & (x >= 0) & (x < W);
    const int gp = (n * H + y) * W + x;
    return ok ? gp : -1;
  }
}

// zeros that a staging load of a halo pixel outside the image reads instead of branching (FK kernels): one page covers
// any channel offset of a 16-byte unit (Cin <= 4096 elements of 2 bytes / 2048 of 4)
__device__ __attribute__((aligned(64))) uint32_t g_zero_page[2048 + 16];

struct ConvP {
  const void* src0; const void* src1;
  int C0, C1, P0, P1;
  const void* w; const float* bias;
  void* dst0; void* dst1;
  int D0, D1, Q0, Q1;
  int slot_w; unsigned acc0_mask; int acc1;
  unsigned inv_slot_w;   // fastdiv_inv(slot_w): the destination slot of a channel without a division per store unit
  float* stats;
  int N, H, W, Cin, Cout;
  int NI, TH, TW, tilesX, tilesY, tilesG, nCoT, nItems;
  int SH;                // stacked-rows tiling: H + 1 (0 = off), see map_pixel
  unsigned SHinv;        // ceil(2^32 / SH)
  unsigned invS, invCoT, invTX, invTY;   // fastdiv_inv of S, nCoT, tilesX, tilesY (item decode)
  int S, nch0, nch;      // K-split: slices, channel chunks of source 0 / total (SK kernels only)
  float* slabs;          // [S][pixels][Cout] fp32 partial sums
  unsigned* sk_cnt;      // per (tile, Cout-tile) arrival counters (zero before and after every launch); NULL: separate finalize kernel
  long long slab_stride; // pixels * Cout
  // BNR kernels (dgrad of a block's second conv): the BatchNorm+ReLU backward REDUCE pass of the first
  // conv's BN is taken in the epilogue, on the values just stored (dst0 must be dense and assign-only)
  const void* bn_y; int bn_py;            // raw output of the first conv (what that BN normalised)
  const float* bn_mi; const float* bn_gamma; const float* bn_beta;   // saved mean | invstd, affine
  float* bn_sums;                         // [2][Cout]: sum dz, sum dz * xhat (atomically accumulated)
};

template <typename T, int WM_, int WN_, int SM_, int SN_, int KG_ = 1> struct ConvCfg {
  static constexpr int WM = WM_, WN = WN_, SM = SM_, SN = SN_, KG = KG_;
  static constexpr int NT = 64 * WM * WN * KG;     // KG wave groups share the tile and split the MFMA steps of every chunk
  static constexpr int BM = 32 * SM * WM;
  static constexpr int BN = 32 * SN * WN;
  static constexpr int EPV = Tr<T>::EPV;
  static constexpr int KC = 64 / (int)sizeof(T);  // channels per LDS chunk (64 B per pixel)
  static constexpr int KS = KC / 16;              // MFMA k-steps per tap per chunk
  static constexpr int PS = KC + EPV;             // padded pixel stride (80 B)
  static constexpr int HPMAX = BM + BM / 2;       // halo pixel capacity
  static constexpr int UH = (HPMAX * 4 + NT - 1) / NT;
  static constexpr int UW = (9 * BN * 4 + NT - 1) / NT;
  static constexpr int OS = BN + EPV;             // epilogue staging row stride (elements)
  static constexpr int UO = (BM * (BN / EPV) + NT - 1) / NT;
  static constexpr int HALO_ELEMS = HPMAX * PS;
  static constexpr int W_ELEMS = 9 * BN * PS;
  static constexpr int STAGE_ELEMS = (HALO_ELEMS + W_ELEMS) > BM * OS ? (HALO_ELEMS + W_ELEMS) : BM * OS;
};

// Persistent kernel: each workgroup walks (tile, Cout-tile) items with stride gridDim.x.
// The global loads of the NEXT (item, channel chunk) are issued into registers before the
// current chunk's MFMA sweep, so HBM latency hides under compute and under the previous
// tile's epilogue, and co-resident workgroups de-synchronise their load/compute/store phases.
template <typename T, int WM, int WN, int SM, int SN, bool SK, bool BNR = false, bool DB = false, int KG = 1, bool FK = false>
__global__ __launch_bounds__(64 * WM * WN * KG) void conv3x3_kernel(ConvP p) {
  typedef ConvCfg<T, WM, WN, SM, SN, KG> C;
  typedef Mma<T> M;
  constexpr int PS = C::PS, EPV = C::EPV, BN = C::BN, BM = C::BM, NT = C::NT, KS = C::KS, OS = C::OS;

  // one LDS arena: [halo | weights] during the K loop, [BM][OS] output staging in the epilogue
  __shared__ __attribute__((aligned(16))) T s_buf[(DB ? 2 : 1) * C::STAGE_ELEMS];   // DB: two [halo | weights] stages
  __shared__ int s_hidx[BM];         // tile-invariant: halo index of output row m
  __shared__ int s_mxy[BM];          // tile-invariant: packed (ni, ly, lx) of row m, -1 unused
  __shared__ int s_hxy[C::HPMAX];    // tile-invariant: packed (ni, hy, hx) of halo pixel, -1 unused
  __shared__ int s_gpix[BM + 1];     // per item: global pixel of row m, -1 masked; [BM]: K-split "this slice arrived last" flag
  __shared__ float s_red[2 * WM * BN];
  __shared__ float s_bnc[BNR ? 4 * 512 : 1];   // BNR: [mean | invstd | scale | shift][Cout <= 512], loaded once
  T* const s_halo = s_buf;
  T* const s_w = s_buf + C::HALO_ELEMS;

  const int tid = threadIdx.x, lane = tid & 63, wave_all = tid >> 6;
  // KG > 1: wave group kg owns the MFMA steps st with st % KG == kg of every chunk (twice the waves per SIMD to hide
  // LDS and MFMA latency when the grid offers one workgroup per CU); group 0 collects the partial sums at the end
  const int kg = KG > 1 ? wave_all / (WM * WN) : 0;
  const int wave = KG > 1 ? wave_all % (WM * WN) : wave_all;
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 31, h = lane >> 5;
  const int HW2 = p.TW + 2, HH2 = p.TH + 2;
  const int HP = p.NI * HH2 * HW2;
  const int THW = p.TH * p.TW;

  for (int m = tid; m < BM; m += NT) {
    const int ni = m / THW;
    const int rem = m - ni * THW;
    const int ly = rem / p.TW, lx = rem - ly * p.TW;
    const bool ok = ni < p.NI;
    s_hidx[m] = ok ? ((ni * HH2 + ly + 1) * HW2 + lx + 1) : (HW2 + 1);
    s_mxy[m] = ok ? ((ni << 20) | (ly << 10) | lx) : -1;
  }
  for (int hp = tid; hp < C::HPMAX; hp += NT) {
    int code = -1;
    if (hp < HP) {
      const int ni = hp / (HH2 * HW2);
      const int rem = hp - ni * (HH2 * HW2);
      const int hy = rem / HW2, hx = rem - hy * HW2;
      code = (ni << 20) | (hy << 10) | hx;
    }
    s_hxy[hp] = code;
  }
  if constexpr (BNR) {
    for (int c = tid; c < p.Cout; c += NT) {
      const float mean = p.bn_mi[c], istd = p.bn_mi[p.Cout + c];
      const float sc = p.bn_gamma[c] * istd;
      s_bnc[c] = mean; s_bnc[512 + c] = istd; s_bnc[1024 + c] = sc; s_bnc[1536 + c] = p.bn_beta[c] - mean * sc;
    }
  }
  __syncthreads();

  int abase[SM];
#pragma unroll
  for (int a = 0; a < SM; ++a) abase[a] = s_hidx[(wm * SM + a) * 32 + r] * PS + 8 * h;
  int toff[9];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) toff[tap] = ((tap / 3 - 1) * HW2 + (tap % 3 - 1)) * PS;
  const int bbase = ((wn * SN) * 32 + r) * PS + 8 * h;
  int hcode[C::UH];
#pragma unroll
  for (int k = 0; k < C::UH; ++k) {
    const int hp = (tid + k * NT) >> 2;
    hcode[k] = hp < C::HPMAX ? s_hxy[hp] : -1;
  }

  struct Item { int co0, n0, y0, x0, ks; };
  auto decode = [&](int item) {
    Item it;
    it.ks = 0;
    if constexpr (SK) { const int q = fastdiv(item, p.invS); it.ks = item - q * p.S; item = q; }
    int q = fastdiv(item, p.invCoT);
    it.co0 = (item - q * p.nCoT) * BN; item = q;
    q = fastdiv(item, p.invTX);
    it.x0 = (item - q * p.tilesX) * p.TW; item = q;
    q = fastdiv(item, p.invTY);
    it.y0 = (item - q * p.tilesY) * p.TH;
    it.n0 = q * p.NI;
    return it;
  };

  int hgp[C::UH];   // global pixel of each staging unit for the item being LOADED: >=0, -1 zero pad
  auto set_hgp = [&](const Item& it) {
#pragma unroll
    for (int k = 0; k < C::UH; ++k) {
      int gp = -1;
      if (hcode[k] >= 0) {
        const int n = it.n0 + (hcode[k] >> 20), y = it.y0 + ((hcode[k] >> 10) & 1023) - 1, x = it.x0 + (hcode[k] & 1023) - 1;
        gp = map_pixel(n, y, x, p.N, p.H, p.W, p.SH, p.SHinv);
      }
      hgp[k] = gp;
    }
  };

  Vec16<T> hreg[C::UH];
  Vec16<T> wreg[C::UW];
  // units past the valid channels of a chunk are zero-filled, so every chunk runs KS full k-steps
#ifdef NUNET_ABL
  bool abl_loaded = false, abl_written = false;   // diagnostic builds (tools/build_ablations.sh) only
#endif
  // FK ("full K": C0 and C1 multiples of the chunk width, every staging unit exists): the address of every unit is
  // fixed for an item up to the chunk's channel offset, which is uniform -> pointers are set up once per item
  // (a pixel outside the image points at a page of zeros) and a chunk's staging is 12 plain loads: no per-unit
  // branches, multiplies or zero selects. PMC had counted 828 VALU + 305 SALU instructions per 36 MFMAs per wave
  // in this loop; with one wave per SIMD that, not the MFMAs or the LDS reads, sets the time of a chunk.
  const T* hp0[FK ? C::UH : 1]; const T* hp1[FK ? C::UH : 1]; const T* wp[FK ? C::UW : 1];
  auto set_ptrs = [&](const Item& it) {
    if constexpr (FK) {
      const T* const zp = reinterpret_cast<const T*>(g_zero_page);
#pragma unroll
      for (int k = 0; k < C::UH; ++k) {
        const int seg = (tid + k * NT) & 3;
        const bool ok = hgp[k] >= 0;   // (-1 also for units past the halo capacity)
        hp0[k] = ok ? (const T*)p.src0 + (size_t)hgp[k] * p.P0 + seg * EPV : zp + seg * EPV;
        hp1[k] = (ok && p.C1 > 0) ? (const T*)p.src1 + (size_t)hgp[k] * p.P1 + seg * EPV : zp + seg * EPV;
      }
#pragma unroll
      for (int k = 0; k < C::UW; ++k) {
        const int u = tid + k * NT;
        const int row = u >> 2, seg = u & 3;
        const int tap = row / BN, co = row - tap * BN;
        // (a unit past the last weight row exists when 9*BN*4 is not a multiple of NT; it is never written to LDS.
        //  The zero page pointer is biased so that "+ kb" stays inside the page.)
        wp[k] = row < 9 * BN ? (const T*)p.w + ((size_t)(tap * p.Cout + it.co0 + co)) * p.Cin + seg * EPV : zp;
      }
    }
  };
  auto load_regs = [&](const Item& it, int kb, int kc) {
#if defined(NUNET_ABL) && (NUNET_ABL & 1)
    if (abl_loaded) return;
    abl_loaded = true;
#endif
    if constexpr (FK) {
      const bool s0 = kb < p.C0;
      const int ch = s0 ? kb : kb - p.C0;
#pragma unroll
      for (int k = 0; k < C::UH; ++k) hreg[k] = ld16((s0 ? hp0[k] : hp1[k]) + ch);
#pragma unroll
      for (int k = 0; k < C::UW; ++k) wreg[k] = ld16(wp[k] + kb);
      return;
    }
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
        wreg[k] = ld16((const T*)p.w + ((size_t)(tap * p.Cout + it.co0 + co)) * p.Cin + kb + seg * EPV);
      } else wreg[k] = zero16<T>();
    }
  };
  auto write_lds = [&](T* const s_halo, T* const s_w) {
#if defined(NUNET_ABL) && (NUNET_ABL & 2)
    if (abl_written) return;
    abl_written = true;
#endif
#pragma unroll
    for (int k = 0; k < C::UH; ++k) {
      const int u = tid + k * NT;
      // (the bound check folds away for every k whose whole NT-unit run exists: no branch per unit)
      if ((k + 1) * NT <= C::HPMAX * 4 || (u >> 2) < C::HPMAX) st16(&s_halo[(u >> 2) * PS + (u & 3) * EPV], hreg[k]);
    }
#pragma unroll
    for (int k = 0; k < C::UW; ++k) {
      const int u = tid + k * NT;
      if ((k + 1) * NT <= 9 * BN * 4 || (u >> 2) < 9 * BN) st16(&s_w[(u >> 2) * PS + (u & 3) * EPV], wreg[k]);
    }
  };

  f32x16 acc[SM][SN];
#pragma unroll
  for (int a = 0; a < SM; ++a)
#pragma unroll
    for (int b = 0; b < SN; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

  // K-split kernels walk channel CHUNKS [c_lo, c_hi) of their slice; plain kernels walk channels
  auto chunk_kb = [&](int c) { return c < p.nch0 ? c * C::KC : p.C0 + (c - p.nch0) * C::KC; };
  auto chunk_kc = [&](int c) { return min(C::KC, (c < p.nch0 ? p.C0 : p.Cin) - chunk_kb(c)); };
  constexpr int NSTEP = 9 * KS;
  // steps of wave group G: st = G, G + KG, ...; fragments of the group's next step are read while the current one multiplies
  auto sweep_g = [&](const T* const s_halo, const T* const s_w, auto gtag) {
    constexpr int G = decltype(gtag)::value;
    constexpr int NS = (NSTEP - G + KG - 1) / KG;       // steps of this group
#if defined(NUNET_ABL) && (NUNET_ABL & 4)
    if (p.N < 0)
#endif
    {
      typename M::Frag fa[2][SM], fb[2][SN];
      {
        constexpr int st0 = G, tap0 = st0 / KS, ks0 = st0 % KS;
#pragma unroll
        for (int a = 0; a < SM; ++a) fa[0][a] = M::load(&s_halo[abase[a] + toff[tap0] + ks0 * 16]);
#pragma unroll
        for (int b = 0; b < SN; ++b) fb[0][b] = M::load(&s_w[bbase + (tap0 * BN + b * 32) * PS + ks0 * 16]);
      }
#pragma unroll
      for (int j = 0; j < NS; ++j) {
        const int cu = j & 1;
        if (j + 1 < NS) {
          const int st = (j + 1) * KG + G;
          const int tap = st / KS, ks = st % KS;
#pragma unroll
          for (int a = 0; a < SM; ++a) fa[cu ^ 1][a] = M::load(&s_halo[abase[a] + toff[tap] + ks * 16]);
#pragma unroll
          for (int b = 0; b < SN; ++b) fb[cu ^ 1][b] = M::load(&s_w[bbase + (tap * BN + b * 32) * PS + ks * 16]);
        }
#pragma unroll
        for (int a = 0; a < SM; ++a)
#pragma unroll
          for (int b = 0; b < SN; ++b) M::mma(acc[a][b], fa[cu][a], fb[cu][b]);
      }
    }
  };
  auto sweep = [&](const T* const s_halo, const T* const s_w) {
    if constexpr (KG == 1) sweep_g(s_halo, s_w, std::integral_constant<int, 0>{});
    else {
      if (kg == 0) sweep_g(s_halo, s_w, std::integral_constant<int, 0>{});
      else sweep_g(s_halo, s_w, std::integral_constant<int, 1>{});
    }
  };
  // KG > 1, end of an item: group 1 hands its partial accumulators to group 0 through LDS (placed behind the
  // epilogue's output staging so that the two never overlap), then every register->memory step below is group 0's
  auto collect_groups = [&]() {
    if constexpr (KG > 1) {
      __syncthreads();   // every wave is done with the halo / weights of the last chunk
      float* const s_kg = reinterpret_cast<float*>(s_buf + BM * OS) ;   // [WM*WN waves][SM*SN*16][64 lanes]
      if (kg == 1) {
#pragma unroll
        for (int a = 0; a < SM; ++a)
#pragma unroll
          for (int b = 0; b < SN; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) { s_kg[((wave * SM * SN + a * SN + b) * 16 + i) * 64 + lane] = acc[a][b][i]; acc[a][b][i] = 0.f; }
      }
      __syncthreads();
      if (kg == 0) {
#pragma unroll
        for (int a = 0; a < SM; ++a)
#pragma unroll
          for (int b = 0; b < SN; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] += s_kg[((wave * SM * SN + a * SN + b) * 16 + i) * 64 + lane];
      }
    }
  };
  auto epi_sk = [&](const Item& cur, const int item) {
      // ---- K-split epilogue: this slice's fp32 partial tile goes to its slab (plain stores,
      // 128-byte runs per half-wave); splitk_finalize_kernel sums the slabs deterministically
      float* slab = p.slabs + (size_t)cur.ks * p.slab_stride;
      if (kg == 0) {
#pragma unroll
      for (int b = 0; b < SN; ++b) {
        const int co = cur.co0 + (wn * SN + b) * 32 + r;
#pragma unroll
        for (int a = 0; a < SM; ++a) {
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int gp = s_gpix[(wm * SM + a) * 32 + acc_row(i, h)];
            if (gp >= 0) slab[(size_t)gp * p.Cout + co] = acc[a][b][i];
            acc[a][b][i] = 0.f;
          }
        }
      }
      }
      if constexpr (SK) {
        if (p.sk_cnt) {
          // ---- last arriver finalizes the tile (no separate finalize launch): every slice publishes its slab
          // (device-scope fence), then counts itself in; the slice that completes the count sums the S slabs in
          // fixed order (deterministic), adds the bias, converts, routes, and takes the BatchNorm statistics
          __threadfence();
          __syncthreads();
          if (tid == 0) {
            const int tix = item / p.S;
            const unsigned old = atomicAdd(&p.sk_cnt[tix], 1u);
            const bool last = old == (unsigned)(p.S - 1);
            if (last) p.sk_cnt[tix] = 0u;             // all S arrivals are in: leave the counter clean for the next launch
            s_gpix[BM] = last ? 1 : 0;
          }
          __syncthreads();
          if (s_gpix[BM]) {
            __threadfence();
            constexpr int SEGS = BN / EPV;
            float q1[EPV], q2[EPV];
#pragma unroll
            for (int e = 0; e < EPV; ++e) { q1[e] = 0.f; q2[e] = 0.f; }
#pragma unroll
            for (int k = 0; k < C::UO; ++k) {
              const int u = tid + k * NT;
              const int m = u / SEGS, seg = u - m * SEGS;
              const int gp = m < BM ? s_gpix[m] : -1;
              if (gp >= 0) {
                const int co = cur.co0 + seg * EPV;
                float x[EPV];
#pragma unroll
                for (int e = 0; e < EPV; ++e) x[e] = 0.f;
                for (int sl = 0; sl < p.S; ++sl) {
                  const f32x4* w4 = reinterpret_cast<const f32x4*>(p.slabs + (size_t)sl * p.slab_stride + (size_t)gp * p.Cout + co);
#pragma unroll
                  for (int v4 = 0; v4 < EPV / 4; ++v4) {
                    const f32x4 t4 = w4[v4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[v4 * 4 + e] += t4[e];
                  }
                }
                T* q; bool accum;
                if (co < p.D0) { q = (T*)p.dst0 + (size_t)gp * p.Q0 + co; accum = (p.acc0_mask >> (p.slot_w > 0 ? fastdiv(co, p.inv_slot_w) : 0)) & 1u; }
                else { q = (T*)p.dst1 + (size_t)gp * p.Q1 + (co - p.D0); accum = p.acc1 != 0; }
                const Vec16<T> o = accum ? ld16(q) : zero16<T>();
                Vec16<T> v;
#pragma unroll
                for (int e = 0; e < EPV; ++e) {
                  const float bb = p.bias ? p.bias[co + e] : 0.f;
                  float y = x[e] + bb;
                  if (accum) y += o.get(e);
                  v.set(e, y);
                  const float d = to_f32(from_f32<T>(y)) - bb;
                  q1[e] += d; q2[e] += d * d;
                }
                st16(q, v);
              }
            }
            if (p.stats) {
#pragma unroll
              for (int off = SEGS; off < 64; off <<= 1) {
#pragma unroll
                for (int e = 0; e < EPV; ++e) { q1[e] += __shfl_xor(q1[e], off); q2[e] += __shfl_xor(q2[e], off); }
              }
              float* s_fin = reinterpret_cast<float*>(s_buf);     // [waves][2][BN]; the staging arena is idle here
              if (lane < SEGS) {
#pragma unroll
                for (int e = 0; e < EPV; ++e) {
                  s_fin[(wave * 2 + 0) * BN + lane * EPV + e] = q1[e];
                  s_fin[(wave * 2 + 1) * BN + lane * EPV + e] = q2[e];
                }
              }
              __syncthreads();
              float* const stp = p.stats + (size_t)(blockIdx.x & (bn_sum_replicas(p.Cout) - 1)) * 2 * p.Cout;
              for (int t = tid; t < 2 * BN; t += NT) {
                const int vsel = t / BN, c = t - vsel * BN;
                float sum = 0.f;
#pragma unroll
                for (int wv = 0; wv < WM * WN; ++wv) sum += s_fin[(wv * 2 + vsel) * BN + c];
                atomicAdd(&stp[vsel * p.Cout + cur.co0 + c], sum);
              }
            }
          }
        }
      }
  };
  auto epi_plain = [&](const Item& cur, T* const s_out) {
      // ---- epilogue: bias, BN partial sums from registers, LDS transpose, 16-byte stores ----
      // BNR: the y1 vectors of this thread's store units are requested NOW, so their latency hides under
      // the accumulator -> LDS transposition below instead of being exposed once per unit in the store loop
      Vec16<T> byv[BNR ? C::UO : 1];
      if constexpr (BNR) {
        constexpr int SEGS0 = BN / EPV;
#pragma unroll
        for (int k = 0; k < C::UO; ++k) {
          const int u = tid + k * NT;
          const int m = u / SEGS0, seg = u - m * SEGS0;
          const int gp = m < BM ? s_gpix[m] : -1;
#if defined(NUNET_ABL) && (NUNET_ABL & 16)
          if (gp >= 0 && p.N < 0) byv[k] = ld16((const T*)p.bn_y + (size_t)gp * p.bn_py + cur.co0 + seg * EPV);
#else
          if (gp >= 0) byv[k] = ld16((const T*)p.bn_y + (size_t)gp * p.bn_py + cur.co0 + seg * EPV);
#endif
        }
      }
      __syncthreads();  // every wave finished reading halo/weights: the arena becomes staging
      if (kg == 0) {
#pragma unroll
      for (int b = 0; b < SN; ++b) {
        const int cl = (wn * SN + b) * 32 + r;  // channel within the tile
        const float bias = p.bias ? p.bias[cur.co0 + cl] : 0.f;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int a = 0; a < SM; ++a) {
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int m = (wm * SM + a) * 32 + acc_row(i, h);
            const T tv = from_f32<T>(acc[a][b][i] + bias);
            s_out[m * OS + cl] = tv;
            if (p.stats && s_gpix[m] >= 0) {
              const float d = to_f32(tv) - bias;
              s1 += d; s2 += d * d;
            }
            acc[a][b][i] = 0.f;
          }
        }
        if (p.stats) {
          s1 += __shfl_xor(s1, 32);
          s2 += __shfl_xor(s2, 32);
          if (h == 0) { s_red[(wm * BN + cl) * 2 + 0] = s1; s_red[(wm * BN + cl) * 2 + 1] = s2; }
        }
      }
      }
      __syncthreads();
      constexpr int SEGS = BN / EPV;
      // BNR: every unit of a thread has the same channel segment (NT % SEGS == 0), so the BN-backward
      // partial sums of its 8 channels live in registers across the store loop
      float bmean[BNR ? EPV : 1], bistd[BNR ? EPV : 1], bsc[BNR ? EPV : 1], bsh[BNR ? EPV : 1], r1[BNR ? EPV : 1], r2[BNR ? EPV : 1];
      if constexpr (BNR) {
        const int c0 = cur.co0 + (tid % SEGS) * EPV;
#pragma unroll
        for (int e = 0; e < EPV; ++e) {
          bmean[e] = s_bnc[c0 + e]; bistd[e] = s_bnc[512 + c0 + e]; bsc[e] = s_bnc[1024 + c0 + e]; bsh[e] = s_bnc[1536 + c0 + e];
          r1[e] = 0.f; r2[e] = 0.f;
        }
      }
#pragma unroll
      for (int k = 0; k < C::UO; ++k) {
        const int u = tid + k * NT;
        const int m = u / SEGS, seg = u - m * SEGS;
        if (m < BM) {
          const int gp = s_gpix[m];
#if defined(NUNET_ABL) && (NUNET_ABL & 8)
          if (gp >= 0 && p.N < 0) {
#else
          if (gp >= 0) {
#endif
            const int co = cur.co0 + seg * EPV;
            T* q; bool accum;
            if (co < p.D0) {
              q = (T*)p.dst0 + (size_t)gp * p.Q0 + co;
              accum = (p.acc0_mask >> (p.slot_w > 0 ? fastdiv(co, p.inv_slot_w) : 0)) & 1u;
            } else {
              q = (T*)p.dst1 + (size_t)gp * p.Q1 + (co - p.D0);
              accum = p.acc1 != 0;
            }
            Vec16<T> v = ld16(&s_out[m * OS + seg * EPV]);
            if (accum) {
              const Vec16<T> o = ld16(q);
#pragma unroll
              for (int e = 0; e < EPV; ++e) v.set(e, v.get(e) + o.get(e));
            }
            st16(q, v);
#if defined(NUNET_ABL) && (NUNET_ABL & 32)
            if constexpr (BNR) if (p.N < 0) {
#else
            if constexpr (BNR) {
#endif
              const Vec16<T> yv = byv[k];
#pragma unroll
              for (int e = 0; e < EPV; ++e) {
                const float yy = yv.get(e);
                const float dz = (yy * bsc[e] + bsh[e]) > 0.f ? v.get(e) : 0.f;   // the stored (rounded) gradient
                r1[e] += dz;
                r2[e] += dz * ((yy - bmean[e]) * bistd[e]);
              }
            }
          }
        }
      }
      if constexpr (BNR) {
        // lanes that share a channel segment (lane % SEGS) are summed with xor-shuffles, the four waves
        // through a small LDS table, then one atomic per channel and sum
#if !(defined(NUNET_ABL) && (NUNET_ABL & 64))
#pragma unroll
        for (int off = SEGS; off < 64; off <<= 1) {
#pragma unroll
          for (int e = 0; e < EPV; ++e) { r1[e] += __shfl_xor(r1[e], off); r2[e] += __shfl_xor(r2[e], off); }
        }
#endif
        __syncthreads();                                   // s_out reads of the store loop are done
        float* s_bn = reinterpret_cast<float*>(s_buf);     // [waves][2][BN]
        if (lane < SEGS) {
#pragma unroll
          for (int e = 0; e < EPV; ++e) {
            s_bn[(wave * 2 + 0) * BN + lane * EPV + e] = r1[e];
            s_bn[(wave * 2 + 1) * BN + lane * EPV + e] = r2[e];
          }
        }
        __syncthreads();
        for (int t = tid; t < 2 * BN; t += NT) {
          const int vsel = t / BN, c = t - vsel * BN;
          float sum = 0.f;
#pragma unroll
          for (int wv = 0; wv < WM * WN; ++wv) sum += s_bn[(wv * 2 + vsel) * BN + c];
#if defined(NUNET_ABL) && (NUNET_ABL & 128)
          if (p.N < 0)
#endif
          atomicAdd(&p.bn_sums[((blockIdx.x & (bn_sum_replicas(p.Cout) - 1)) * 2 + vsel) * p.Cout + cur.co0 + c], sum);
        }
      }
      if (p.stats) {
        for (int c = tid; c < BN; c += NT) {
          float s1 = 0.f, s2 = 0.f;
#pragma unroll
          for (int k = 0; k < WM; ++k) { s1 += s_red[(k * BN + c) * 2]; s2 += s_red[(k * BN + c) * 2 + 1]; }
          float* const st = p.stats + (size_t)(blockIdx.x & (bn_sum_replicas(p.Cout) - 1)) * 2 * p.Cout;   // replica of this workgroup
          atomicAdd(&st[cur.co0 + c], s1);
          atomicAdd(&st[p.Cout + cur.co0 + c], s2);
        }
      }
  };
  int item = blockIdx.x;
  if (item >= p.nItems) return;
  Item cur = decode(item);
  set_hgp(cur);
  set_ptrs(cur);
  int cc = 0, c_hi = 0;
  int kb = 0, kc = 0;
  if constexpr (SK) { cc = cur.ks * p.nch / p.S; c_hi = (cur.ks + 1) * p.nch / p.S; kb = chunk_kb(cc); kc = chunk_kc(cc); }
  else { kc = min(C::KC, (kb < p.C0 ? p.C0 : p.Cin) - kb); }
  load_regs(cur, kb, kc);
  bool first_chunk = true;

  while (true) {
    __syncthreads();  // previous chunk's fragment reads / previous item's epilogue reads are done
    write_lds(s_halo, s_w);
    if (first_chunk) {
      for (int m = tid; m < BM; m += NT) {
        const int code = s_mxy[m];
        int gp = -1;
        if (code >= 0) {
          const int n = cur.n0 + (code >> 20), y = cur.y0 + ((code >> 10) & 1023), x = cur.x0 + (code & 1023);
          gp = map_pixel(n, y, x, p.N, p.H, p.W, p.SH, p.SHinv);
        }
        s_gpix[m] = gp;
      }
    }
    __syncthreads();
    // prefetch the next (item, chunk) into registers
    int nkb = kb + kc, nkc = 0, nitem = item, ncc = cc + 1, nc_hi = c_hi;
    Item nxt = cur;
    bool have_next = true;
    bool last_chunk;
    if constexpr (SK) last_chunk = ncc >= c_hi; else last_chunk = nkb >= p.Cin;
    if (last_chunk) {
      nkb = 0;
      nitem = item + gridDim.x;
      if (nitem < p.nItems) {
        nxt = decode(nitem); set_hgp(nxt); set_ptrs(nxt);
        if constexpr (SK) { ncc = nxt.ks * p.nch / p.S; nc_hi = (nxt.ks + 1) * p.nch / p.S; }
      } else have_next = false;
    }
    if (have_next) {
      if constexpr (SK) { nkb = chunk_kb(ncc); nkc = chunk_kc(ncc); }
      else nkc = min(C::KC, (nkb < p.C0 ? p.C0 : p.Cin) - nkb);
      load_regs(nxt, nkb, nkc);
    }
    sweep(s_halo, s_w);
    first_chunk = false;
    if (SK && last_chunk) {
      collect_groups();
      epi_sk(cur, item);
      if (!have_next) break;
      cur = nxt; item = nitem; first_chunk = true;
    } else if (last_chunk) {
      collect_groups();
      epi_plain(cur, s_buf);
      if (!have_next) break;
      cur = nxt; item = nitem; first_chunk = true;
    }
    kb = nkb; kc = nkc; cc = ncc; c_hi = nc_hi;
  }
}

// Finishes a K-split convolution: sum of the S fp32 slabs (fixed order: deterministic) -> (+bias)
// -> T, routed to the two destinations with the per-slot accumulate mask, BatchNorm partial sums.
struct SplitFinP {
  const float* slabs; long long slab_stride; int S; const float* bias;
  void* dst0; void* dst1; int D0, D1, Q0, Q1; int slot_w; unsigned acc0_mask; int acc1; unsigned inv_slot_w;
  float* stats; long long npix; int Cout;
  const void* bn_y; int bn_py; const float* bn_mi; const float* bn_gamma; const float* bn_beta; float* bn_sums;   // see ConvP
};
template <typename T, bool BNR = false>
__global__ __launch_bounds__(256) void splitk_finalize_kernel(SplitFinP p) {
  constexpr int EPV = Tr<T>::EPV;
  __shared__ float s_st[2 * 1024];
  __shared__ float s_bn[BNR ? 6 * 1024 : 1];   // [mean | invstd | scale | shift | sum dz | sum dz*xhat][Cout]
  const int G = p.Cout / EPV;
  if (p.stats) for (int c = threadIdx.x; c < 2 * p.Cout; c += blockDim.x) s_st[c] = 0.f;
  if constexpr (BNR) {
    for (int c = threadIdx.x; c < p.Cout; c += blockDim.x) {
      const float mean = p.bn_mi[c], istd = p.bn_mi[p.Cout + c];
      const float sc = p.bn_gamma[c] * istd;
      s_bn[c] = mean; s_bn[p.Cout + c] = istd; s_bn[2 * p.Cout + c] = sc; s_bn[3 * p.Cout + c] = p.bn_beta[c] - mean * sc;
      s_bn[4 * p.Cout + c] = 0.f; s_bn[5 * p.Cout + c] = 0.f;
    }
  }
  __syncthreads();
  const long long total = p.npix * G;
  // when the block size is a multiple of G every element a thread visits has the same channel group: the statistics are
  // summed in registers and meet in LDS once per thread (per-element LDS atomics on G*EPV addresses were most of this kernel)
  const bool fixed_cg = (blockDim.x % G) == 0;
  float q1[EPV], q2[EPV];
#pragma unroll
  for (int e = 0; e < EPV; ++e) { q1[e] = 0.f; q2[e] = 0.f; }
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cg = (int)(i % G);
    const long long pix = i / G;
    const int co = cg * EPV;
    float x[EPV];
#pragma unroll
    for (int e = 0; e < EPV; ++e) x[e] = 0.f;
    for (int k = 0; k < p.S; ++k) {
      const f32x4* w = reinterpret_cast<const f32x4*>(p.slabs + (size_t)k * p.slab_stride + pix * p.Cout + co);
#pragma unroll
      for (int v4 = 0; v4 < EPV / 4; ++v4) {
        const f32x4 q4 = w[v4];
#pragma unroll
        for (int e = 0; e < 4; ++e) x[v4 * 4 + e] += q4[e];
      }
    }
    T* q; bool accum;
    if (co < p.D0) { q = (T*)p.dst0 + pix * p.Q0 + co; accum = (p.acc0_mask >> (p.slot_w > 0 ? fastdiv(co, p.inv_slot_w) : 0)) & 1u; }
    else { q = (T*)p.dst1 + pix * p.Q1 + (co - p.D0); accum = p.acc1 != 0; }
    const Vec16<T> o = accum ? ld16(q) : zero16<T>();
    Vec16<T> v;
#pragma unroll
    for (int e = 0; e < EPV; ++e) {
      const float b = p.bias ? p.bias[co + e] : 0.f;
      float y = x[e] + b;
      if (accum) y += o.get(e);
      v.set(e, y);
      if (p.stats) {
        const float d = to_f32(from_f32<T>(y)) - b;   // the stored (rounded) value; not v.get(e) right after v.set(e)
        if (fixed_cg) { q1[e] += d; q2[e] += d * d; }
        else { atomicAdd(&s_st[co + e], d); atomicAdd(&s_st[p.Cout + co + e], d * d); }
      }
    }
    st16(q, v);
    if constexpr (BNR) {
      const Vec16<T> yv = ld16((const T*)p.bn_y + pix * p.bn_py + co);
#pragma unroll
      for (int e = 0; e < EPV; ++e) {
        const int c = co + e;
        const float yy = yv.get(e);
        const float dz = (yy * s_bn[2 * p.Cout + c] + s_bn[3 * p.Cout + c]) > 0.f ? v.get(e) : 0.f;
        atomicAdd(&s_bn[4 * p.Cout + c], dz);
        atomicAdd(&s_bn[5 * p.Cout + c], dz * ((yy - s_bn[c]) * s_bn[p.Cout + c]));
      }
    }
  }
  if constexpr (BNR) {
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * p.Cout; c += blockDim.x)
      atomicAdd(&p.bn_sums[(blockIdx.x & (bn_sum_replicas(p.Cout) - 1)) * 2 * p.Cout + c], s_bn[4 * p.Cout + c]);
  }
  if (p.stats) {
    if (fixed_cg) {
      const int co = (threadIdx.x % G) * EPV;
#pragma unroll
      for (int e = 0; e < EPV; ++e) { atomicAdd(&s_st[co + e], q1[e]); atomicAdd(&s_st[p.Cout + co + e], q2[e]); }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * p.Cout; c += blockDim.x)
      atomicAdd(&p.stats[(size_t)(blockIdx.x & (bn_sum_replicas(p.Cout) - 1)) * 2 * p.Cout + c], s_st[c]);
  }
}


// tile-geometry chooser shared by fwd and wgrad
struct TileGeom { int NI, TH, TW, tilesX, tilesY, tilesG, SH; };
TileGeom nunet_choose_tile(int N, int H, int W, int BM, int HPMAX) {
  TileGeom best{1, 1, 1, W, H, N, 0};
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
        if (score > best_score) { best_score = score; best = TileGeom{ni, th, tw, tX, tY, tG, 0}; }
      }
    }
  }
  // stacked-rows candidates (map_pixel): TH virtual rows x full width over N * (H + 1) virtual rows;
  // taken only when they beat the best regular tiling by a clear margin (they waste the separator rows)
  static int stacked = -1;
  if (stacked < 0) { const char* e = getenv("NUNET_STACKED_TILES"); stacked = e ? atoi(e) : 1; }
  if (stacked && W <= BM && (long)N * (H + 1) < (1 << 20) && H + 1 <= 4096) {
    const double best_util = (double)N * H * W / ((double)best.tilesX * best.tilesY * best.tilesG * BM);
    const int VH = N * (H + 1);
    double su = 0.0; int sth = 0;
    for (int th = 1; th * W <= BM && th < 1000; ++th) {
      if ((long)(th + 2) * (W + 2) > HPMAX) break;
      const int tY = ceil_div(VH, th);
      const double util = (double)N * H * W / ((double)tY * BM);
      if (util > su) { su = util; sth = th; }
    }
    if (sth > 0 && su > best_util * 1.08) best = TileGeom{1, sth, W, 1, ceil_div(VH, sth), 1, H + 1};
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
  p.slot_w = d->acc_slot_w; p.acc0_mask = d->acc0_mask; p.acc1 = d->acc1; p.inv_slot_w = fastdiv_inv(d->acc_slot_w);
  p.stats = d->stats;
  p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->C0 + d->C1; p.Cout = d->D0 + d->D1;
  const bool bnr = d->bn_y != nullptr;
  p.bn_y = d->bn_y; p.bn_py = d->bn_py; p.bn_mi = d->bn_mean_invstd; p.bn_gamma = d->bn_gamma; p.bn_beta = d->bn_beta; p.bn_sums = d->bn_sums;
  const TileGeom g = nunet_choose_tile(d->N, d->H, d->W, C::BM, C::HPMAX);
  p.NI = g.NI; p.TH = g.TH; p.TW = g.TW; p.tilesX = g.tilesX; p.tilesY = g.tilesY; p.tilesG = g.tilesG; p.SH = g.SH;
  p.SHinv = g.SH ? (unsigned)(((1ull << 32) + g.SH - 1) / g.SH) : 0u;
  p.nCoT = p.Cout / C::BN;
  long items = (long)p.nCoT * g.tilesX * g.tilesY * g.tilesG;
  // K-split for the grid-starved deep levels: slices of the channel-chunk loop become extra items, each
  // writes an fp32 partial slab; splitk_finalize_kernel sums them (fixed order, deterministic)
  p.S = 1; p.slabs = nullptr; p.slab_stride = 0; p.sk_cnt = nullptr;
  p.nch0 = ceil_div(p.C0, C::KC);
  p.nch = p.nch0 + (p.C1 > 0 ? ceil_div(p.C1, C::KC) : 0);
  static int sk_max_items = -1;
  if (sk_max_items < 0) { const char* e = getenv("NUNET_SK_MAXITEMS"); sk_max_items = e ? atoi(e) : 100; }
  // the first NUNET_SPLITK_COUNTER_FLOATS floats of the workspace are the arrival counters of the in-kernel
  // finalize (zero before the first use, left zero by every launch); the slabs follow
  static int sk_inkernel = -1;
  if (sk_inkernel < 0) { const char* e = getenv("NUNET_SK_INKERNEL"); sk_inkernel = e ? atoi(e) : 0;   /* measured: the device-scope fences (L2 write-back + invalidate on every workgroup) cost 470 us per step: off */ }
  if (d->splitk_ws && d->splitk_ws_floats > NUNET_SPLITK_COUNTER_FLOATS && items <= sk_max_items && p.nch >= 8 && p.Cout <= 1024) {
    int S = (int)((320 + items - 1) / items);
    if (S > p.nch / 2) S = p.nch / 2;
    const long long need = (long long)S * d->N * d->H * d->W * p.Cout;
    if (S > 1 && need <= d->splitk_ws_floats - NUNET_SPLITK_COUNTER_FLOATS) {
      p.S = S; p.slabs = d->splitk_ws + NUNET_SPLITK_COUNTER_FLOATS; p.slab_stride = (long long)d->N * d->H * d->W * p.Cout;
      if (sk_inkernel && !bnr && items <= NUNET_SPLITK_COUNTER_FLOATS) p.sk_cnt = reinterpret_cast<unsigned*>(d->splitk_ws);
      items *= S;
    }
  }
  p.nItems = (int)items;
  p.invS = fastdiv_inv(p.S); p.invCoT = fastdiv_inv(p.nCoT); p.invTX = fastdiv_inv(p.tilesX); p.invTY = fastdiv_inv(p.tilesY);
  // KG = 2 (8 waves share the tile, each group takes every other MFMA step of a chunk): when the grid offers at most
  // one workgroup per CU anyway, twice the waves per SIMD hide LDS/MFMA latency. 16-bit types only (LDS budget of the
  // cross-group hand-over). NUNET_CONV_KG: 0 off, 1 auto (items <= NUNET_CONV_KG_MAXITEMS), 2 always.
  static int kg_mode = -1, kg_max_items = 0;
  if (kg_mode < 0) {
    const char* e = getenv("NUNET_CONV_KG"); kg_mode = e ? atoi(e) : 0;
    e = getenv("NUNET_CONV_KG_MAXITEMS"); kg_max_items = e ? atoi(e) : 300;
  }
  const bool use_kg = sizeof(T) == 2 && !bnr && !p.sk_cnt && (kg_mode == 2 || (kg_mode == 1 && items <= kg_max_items));
  // persistent grid: resident workgroups only, item counts balanced across them
  const size_t lds_bytes = sizeof(T) * C::STAGE_ELEMS + 4 * (3 * C::BM + C::HPMAX) + 8 * WM * C::BN + (d->bn_y ? 8192 : 0);
  long per_cu = (long)(160 * 1024 / lds_bytes);
  if (per_cu < 1) per_cu = 1;
  if (per_cu > 2048 / C::NT) per_cu = 2048 / C::NT;
  if (use_kg) per_cu = 1;   // 8 waves x > 128 registers: one workgroup per CU
  const long resident = 256 * per_cu;
  const long rounds = (items + resident - 1) / resident;
  const long grid = (items + rounds - 1) / rounds;
  const double px = (double)d->N * d->H * d->W;
  const int acin = g_prof_alg_cin > 0 ? g_prof_alg_cin : p.Cin;
  ProfScope ps(C::BN == 32 ? PC_CONV_M256N32 : PC_CONV_M128N64,  /* BN 64 configs share a class */ 2.0 * 9 * acin * p.Cout * px,
               (px * (acin + p.Cout) + 9.0 * acin * p.Cout) * sizeof(T), st);
  // FK: branch-free staging when every chunk is full (see the kernel). NUNET_CONV_FK=0 turns it off.
  static int fk_mode = -1;
  if (fk_mode < 0) { const char* e = getenv("NUNET_CONV_FK"); fk_mode = e ? atoi(e) : 1; }
  const bool use_fk = fk_mode && !bnr && !use_kg && d->C0 % C::KC == 0 && d->C1 % C::KC == 0 && (size_t)p.Cin * sizeof(T) <= 8192;
  if (use_fk) {
    if (p.S > 1) {
      hipLaunchKernelGGL((conv3x3_kernel<T, WM, WN, SM, SN, true, false, false, 1, true>), dim3((unsigned)grid), dim3(C::NT), 0, st, p);
    } else {
      hipLaunchKernelGGL((conv3x3_kernel<T, WM, WN, SM, SN, false, false, false, 1, true>), dim3((unsigned)grid), dim3(C::NT), 0, st, p);
      return nunet_check_launch("conv3x3 (full-K staging)");
    }
  }
  if (p.S > 1) {
    if constexpr (sizeof(T) == 2) {
      if (use_kg) hipLaunchKernelGGL((conv3x3_kernel<T, WM, WN, SM, SN, true, false, false, 2>), dim3((unsigned)grid), dim3(2 * C::NT), 0, st, p);
    }
    if (!use_kg && !use_fk) hipLaunchKernelGGL((conv3x3_kernel<T, WM, WN, SM, SN, true>), dim3((unsigned)grid), dim3(C::NT), 0, st, p);
    if (p.sk_cnt) return nunet_check_launch("conv3x3 (K-split, in-kernel finalize)");
    SplitFinP f;
    f.slabs = p.slabs; f.slab_stride = p.slab_stride; f.S = p.S; f.bias = p.bias;
    f.dst0 = p.dst0; f.dst1 = p.dst1; f.D0 = p.D0; f.D1 = p.D1; f.Q0 = p.Q0; f.Q1 = p.Q1;
    f.slot_w = p.slot_w; f.acc0_mask = p.acc0_mask; f.acc1 = p.acc1; f.inv_slot_w = p.inv_slot_w; f.stats = p.stats;
    f.npix = (long long)d->N * d->H * d->W; f.Cout = p.Cout;
    f.bn_y = p.bn_y; f.bn_py = p.bn_py; f.bn_mi = p.bn_mi; f.bn_gamma = p.bn_gamma; f.bn_beta = p.bn_beta; f.bn_sums = p.bn_sums;
    long long fg = (f.npix * (p.Cout / C::EPV) + 255) / 256;
    if (fg > 1024) fg = 1024;
    if (bnr) hipLaunchKernelGGL((splitk_finalize_kernel<T, true>), dim3((unsigned)fg), dim3(256), 0, st, f);
    else hipLaunchKernelGGL((splitk_finalize_kernel<T, false>), dim3((unsigned)fg), dim3(256), 0, st, f);
    return nunet_check_launch("conv3x3 (K-split)");
  }
  if constexpr (sizeof(T) == 2) {
    if (use_kg) {
      hipLaunchKernelGGL((conv3x3_kernel<T, WM, WN, SM, SN, false, false, false, 2>), dim3((unsigned)grid), dim3(2 * C::NT), 0, st, p);
      return nunet_check_launch("conv3x3 (2 wave groups)");
    }
  }
  if (bnr) hipLaunchKernelGGL((conv3x3_kernel<T, WM, WN, SM, SN, false, true>), dim3((unsigned)grid), dim3(C::NT), 0, st, p);
  else hipLaunchKernelGGL((conv3x3_kernel<T, WM, WN, SM, SN, false, false>), dim3((unsigned)grid), dim3(C::NT), 0, st, p);
  return nunet_check_launch("conv3x3");
}

template <typename T> static int launch_conv(const nunet_conv_desc* d, hipStream_t st) {
  const int cout = d->D0 + d->D1;
  static int big = -1;   // measured: the 256x64 tile drops to 1 wave/SIMD (296 registers) and loses; off by default
  if (big < 0) { const char* e = getenv("NUNET_CONV_BIG"); big = e ? atoi(e) : 0; }
  const long px = (long)d->N * d->H * d->W;
  if (cout % 64 == 0) {
    // deep pyramid levels are weight-streaming bound: every M-tile re-reads the layer's whole weight
    // slab, so few pixels -> the largest tile (256 px x 64 co, 64x64 per wave: 1.0 LDS reads per MFMA),
    // parallelism restored by the K-split
    if (big && px <= big * 10000L && d->splitk_ws) return launch_conv_cfg<T, 4, 1, 2, 2>(d, st);   // BM 256, BN 64
    return launch_conv_cfg<T, 2, 2, 2, 1>(d, st);                                       // BM 128, BN 64
  }
  return launch_conv_cfg<T, 4, 1, 2, 1>(d, st);                                         // BM 256, BN 32
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
  if (d->bn_y) {
    NUNET_REQUIRE(d->bn_mean_invstd && d->bn_gamma && d->bn_beta && d->bn_sums, "conv3x3: fused BN-backward reduce needs mean/invstd, gamma, beta and sums");
    NUNET_REQUIRE(d->D1 == 0 && d->Q0 == d->D0 && d->acc0_mask == 0 && d->bn_py % (16 / dtype_size(d->dtype)) == 0 && cout <= 512,
                  "conv3x3: fused BN-backward reduce needs one dense, assign-only destination");
  }
  return NUNET_DISPATCH(d->dtype, launch_conv, d, (hipStream_t)s);
}

// ---------------------------------------------------------------------------
// wgrad kernel
//
// Work item = (32 Cout) x (32 Cin) x 9 taps of dW, over a slice of the pixel tiles.
// All 4 waves of a workgroup accumulate the SAME output block over different pixels of
// each 128-pixel tile (intra-workgroup split of the contraction), so no wave idles for
// narrow layers; partials are summed through LDS once, then added atomically to the fp32
// gradient. Tiles are double-buffered in LDS; the next tile's global loads are in flight
// (in registers) while the current one multiplies.
// ---------------------------------------------------------------------------
struct WgP {
  const void* src0; const void* src1;
  int C0, C1, P0, P1;
  const void* dy; int Cout, PY;
  float* dw;
  int N, H, W, Cin;
  int NI, TH, TW, tilesX, tilesY, tilesG;
  int nCoT, nCiT, ksplit, nMT;
  int SH; unsigned SHinv;
  unsigned invTX, invTY;   // fastdiv_inv(tilesX), fastdiv_inv(tilesY)
};

template <typename T> struct WgCfg {
  static constexpr int NT = 192;                      // 3 waves: wave w owns taps 3w..3w+2
  static constexpr int BM = 128, HPMAX = 192;
  static constexpr int EPV = Tr<T>::EPV;
  static constexpr bool F32 = std::is_same<T, float>::value;
  static constexpr int SR = 32;                       // row stride (elements): 32 channels, no pad
  static constexpr int UPP = 32 / EPV;                // 16-byte units per pixel row
  static constexpr int NUD = (BM * UPP + NT - 1) / NT;      // dY units per thread
  static constexpr int NUA = (HPMAX * UPP + NT - 1) / NT;   // halo units per thread
  static constexpr int STAGE = (BM + HPMAX) * SR;     // elements per LDS stage
};

template <typename T>
__device__ __forceinline__ s16x4 tr_read(const T* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4*)(const_cast<T*>(p)));
}

template <typename T> struct Frag16;
template <> struct Frag16<bf16_t> { typedef bf16x8 V; };
template <> struct Frag16<f16_t> { typedef f16x8 V; };
typedef __attribute__((ext_vector_type(8))) short s16x8;

// Work item = (32 Cout) x (32 Cin) x 9 taps of dW over a slice of the 128-pixel tiles. The three
// waves of a workgroup split the TAPS (3 each) and all walk every pixel of the tile: no cross-wave
// reduction, 48 accumulator registers per lane. Tiles are double-buffered in LDS; the next tile's
// global loads are in flight (registers) while the current one multiplies.
template <typename T, bool ST>
__device__ __forceinline__ void wgrad_body(const WgP& p, int bid) {
  typedef WgCfg<T> C;
  constexpr int NT = C::NT, BM = C::BM, EPV = C::EPV, SR = C::SR, UPP = C::UPP;
  __shared__ __attribute__((aligned(16))) T s_stage[2 * C::STAGE];
  __shared__ int s_hidx[BM];
  __shared__ int s_mxy[BM];
  __shared__ int s_hxy[C::HPMAX];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;

  const int cit = bid % p.nCiT; bid /= p.nCiT;
  const int cot = bid % p.nCoT;
  const int split = bid / p.nCoT;
  const int co0 = cot * 32, ci0 = cit * 32;
  const int HW2 = p.TW + 2, HH2 = p.TH + 2;
  const int HP = p.NI * HH2 * HW2;
  const int THW = p.TH * p.TW;

  for (int m = tid; m < BM; m += NT) {
    const int ni = m / THW;
    const int rem = m - ni * THW;
    const int ly = rem / p.TW, lx = rem - ly * p.TW;
    const bool ok = ni < p.NI;
    s_hidx[m] = ok ? ((ni * HH2 + ly + 1) * HW2 + lx + 1) : (HW2 + 1);
    s_mxy[m] = ok ? ((ni << 20) | (ly << 10) | lx) : -1;
  }
  for (int hp = tid; hp < C::HPMAX; hp += NT) {
    int code = -1;
    if (hp < HP) {
      const int ni = hp / (HH2 * HW2);
      const int rem = hp - ni * (HH2 * HW2);
      const int hy = rem / HW2, hx = rem - hy * HW2;
      code = (ni << 20) | (hy << 10) | hx;
    }
    s_hxy[hp] = code;
  }
  __syncthreads();

  // tile-invariant staging codes of this thread's units (-2: no such unit)
  int dcode[C::NUD], acode[C::NUA];
#pragma unroll
  for (int k = 0; k < C::NUD; ++k) { const int u = tid + k * NT; dcode[k] = u < BM * UPP ? s_mxy[u / UPP] : -2; }
#pragma unroll
  for (int k = 0; k < C::NUA; ++k) { const int u = tid + k * NT; acode[k] = u < C::HPMAX * UPP ? s_hxy[u / UPP] : -2; }
  // the channel slice of the forward input this item reads: one source per 16-byte unit
  // (C0 is a multiple of 16, so a unit never straddles the two concat sources)
  const int seg = tid % UPP;                 // same for all units of a thread (NT % UPP == 0)
  const int cch = ci0 + seg * EPV;
  const bool cvalid = cch < p.Cin;
  const T* asrc; int apitch, ach;
  if (cch < p.C0) { asrc = (const T*)p.src0; apitch = p.P0; ach = cch; }
  else { asrc = (const T*)p.src1; apitch = p.P1; ach = cch - p.C0; }
  const T* dsrc = (const T*)p.dy + co0 + seg * EPV;

  Vec16<T> dreg[C::NUD], areg[C::NUA];
  auto load_tile = [&](int mt) {
    const int q1 = fastdiv(mt, p.invTX), q2 = fastdiv(q1, p.invTY);
    const int x0 = (mt - q1 * p.tilesX) * p.TW;
    const int y0 = (q1 - q2 * p.tilesY) * p.TH;
    const int n0 = q2 * p.NI;
    // branch-free: a unit outside the image (or past the tile / channel range) reads a page of zeros, so the seven
    // loads of a tile are independent instructions of one basic block instead of seven guarded blocks
    const T* const zp = reinterpret_cast<const T*>(g_zero_page);
#pragma unroll
    for (int k = 0; k < C::NUD; ++k) {
      const int cde = dcode[k] >= 0 ? dcode[k] : 0;
      const int n = n0 + (cde >> 20), y = y0 + ((cde >> 10) & 1023), x = x0 + (cde & 1023);
      int gp = map_pixel_t<ST>(n, y, x, p.N, p.H, p.W, p.SH, p.SHinv);
      if (dcode[k] < 0) gp = -1;
      dreg[k] = ld16(gp >= 0 ? dsrc + (size_t)gp * p.PY : zp);
    }
#pragma unroll
    for (int k = 0; k < C::NUA; ++k) {
      const int cde = acode[k] >= 0 ? acode[k] : 0;
      const int n = n0 + (cde >> 20), y = y0 + ((cde >> 10) & 1023) - 1, x = x0 + (cde & 1023) - 1;
      int gp = map_pixel_t<ST>(n, y, x, p.N, p.H, p.W, p.SH, p.SHinv);
      if (acode[k] < 0 || !cvalid) gp = -1;
      areg[k] = ld16(gp >= 0 ? asrc + (size_t)gp * apitch + ach : zp);
    }
  };
  auto write_tile = [&](int buf) {
    T* sd = s_stage + buf * C::STAGE;
    T* sa = sd + BM * SR;
#pragma unroll
    for (int k = 0; k < C::NUD; ++k) if ((k + 1) * NT <= BM * UPP || dcode[k] != -2) st16(&sd[((tid + k * NT) / UPP) * SR + seg * EPV], dreg[k]);
#pragma unroll
    for (int k = 0; k < C::NUA; ++k) if ((k + 1) * NT <= C::HPMAX * UPP || acode[k] != -2) st16(&sa[((tid + k * NT) / UPP) * SR + seg * EPV], areg[k]);
  };

  f32x16 acc[3];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

  // this wave's three taps: rows 3w..3w+2 of the 3x3 window = kernel row `wave`
  int toff[3];
#pragma unroll
  for (int t = 0; t < 3; ++t) toff[t] = ((wave - 1) * HW2 + (t - 1)) * SR;

  int mt = split;
  if (mt < p.nMT) load_tile(mt);
  int buf = 0;
  if (mt < p.nMT) write_tile(0);
  __syncthreads();
  while (mt < p.nMT) {
    const int nmt = mt + p.ksplit;
    if (nmt < p.nMT) load_tile(nmt);   // in flight during the MFMAs below
    const T* sd = s_stage + buf * C::STAGE;
    const T* sa = sd + BM * SR;
    if constexpr (C::F32) {
#pragma unroll 4
      for (int kk = 0; kk < BM / 2; ++kk) {
        const int m = kk * 2 + h;
        const float av = sd[m * SR + r];
        const int hx = s_hidx[m] * SR + r;
#pragma unroll
        for (int t = 0; t < 3; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, sa[hx + toff[t]], acc[t], 0, 0, 0);
      }
    } else {
      typedef typename Frag16<T>::V FV;
      const int q = (lane >> 2) & 3;                             // row within the 4-row block
      const int colo = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);  // column offset supplied by this lane
#pragma unroll
      for (int kk = 0; kk < BM / 16; ++kk) {
        const int m0 = kk * 16 + 8 * h + q, m1 = m0 + 4;
        const int b0 = s_hidx[m0] * SR + colo, b1 = s_hidx[m1] * SR + colo;
        const s16x4 a0 = tr_read(&sd[m0 * SR + colo]);
        const s16x4 a1 = tr_read(&sd[m1 * SR + colo]);
        const s16x8 av = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
        s16x4 x0[3], x1[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) { x0[t] = tr_read(&sa[b0 + toff[t]]); x1[t] = tr_read(&sa[b1 + toff[t]]); }
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          const s16x8 bv = __builtin_shufflevector(x0[t], x1[t], 0, 1, 2, 3, 4, 5, 6, 7);
          Mma<T>::mma(acc[t], __builtin_bit_cast(FV, av), __builtin_bit_cast(FV, bv));
        }
      }
    }
    if (nmt < p.nMT) write_tile(buf ^ 1);   // the other stage was last read one iteration ago
    __syncthreads();
    buf ^= 1;
    mt = nmt;
  }

  // one atomic per output element of this wave's three taps
  const int ci = ci0 + r;
  if (ci < p.Cin) {
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int tap = wave * 3 + t;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int co = co0 + acc_row(i, h);
#if defined(NUNET_ABL) && (NUNET_ABL & 256)
        if (co < p.Cout) p.dw[((size_t)tap * p.Cout + co) * p.Cin + ci] = acc[t][i];    // timing experiment: no atomics
#elif defined(NUNET_ABL) && (NUNET_ABL & 512)
        if (co < p.Cout) atomicAdd(&p.dw[(size_t)(split & 7) * 9 * p.Cout * p.Cin + ((size_t)tap * p.Cout + co) * p.Cin + ci], acc[t][i]);   // timing experiment: 8 replicas (dw must be 8x)
#else
        if (co < p.Cout) atomicAdd(&p.dw[((size_t)tap * p.Cout + co) * p.Cin + ci], acc[t][i]);
#endif
      }
    }
  }
}

template <typename T, bool ST>
__global__ __launch_bounds__(192) void wgrad_kernel(WgP p) { wgrad_body<T, ST>(p, blockIdx.x); }

// Two independent weight-gradient problems in ONE launch (the two convolutions of a VGGBlock finish
// their dY at the same point of the backward pass): one kernel boundary less per block and twice
// the workgroups to fill the chip.
template <typename T, bool ST>
__global__ __launch_bounds__(192) void wgrad_pair_kernel(WgP pa, WgP pb, int na) {
  if ((int)blockIdx.x < na) wgrad_body<T, ST>(pa, blockIdx.x);
  else wgrad_body<T, ST>(pb, blockIdx.x - na);
}

template <typename T> static long wgrad_setup(const nunet_wgrad_desc* d, WgP& p, int target_override = 0) {
  typedef WgCfg<T> C;
  p.src0 = d->src0; p.src1 = d->src1; p.C0 = d->C0; p.C1 = d->C1; p.P0 = d->P0; p.P1 = d->P1;
  p.dy = d->dy; p.Cout = d->Cout; p.PY = d->PY; p.dw = d->dw;
  p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->C0 + d->C1;
  const TileGeom g = nunet_choose_tile(d->N, d->H, d->W, C::BM, C::HPMAX);
  p.NI = g.NI; p.TH = g.TH; p.TW = g.TW; p.tilesX = g.tilesX; p.tilesY = g.tilesY; p.tilesG = g.tilesG; p.SH = g.SH;
  p.SHinv = g.SH ? (unsigned)(((1ull << 32) + g.SH - 1) / g.SH) : 0u;
  p.invTX = fastdiv_inv(g.tilesX); p.invTY = fastdiv_inv(g.tilesY);
  p.nCoT = ceil_div(p.Cout, 32);
  p.nCiT = ceil_div(p.Cin, 32);
  p.nMT = g.tilesX * g.tilesY * g.tilesG;
  const int otiles = p.nCoT * p.nCiT;
  static int wg_target = 0;
  if (!wg_target) { const char* e = getenv("NUNET_WG_TARGET"); wg_target = e ? atoi(e) : 256; }
  int ks = ceil_div(target_override > 0 ? target_override : wg_target, otiles);
  if (ks > p.nMT) ks = p.nMT;
  if (ks < 1) ks = 1;
  p.ksplit = ks;
  return (long)otiles * ks;
}
template <typename T> static void wgrad_prof(const nunet_wgrad_desc* d, const WgP& p, double& flops, double& bytes) {
  const double px = (double)d->N * d->H * d->W;
  const int acin = g_prof_alg_cin > 0 ? g_prof_alg_cin : p.Cin;
  flops = 2.0 * 9 * acin * p.Cout * px;
  bytes = px * (acin + p.Cout) * sizeof(T) + 9.0 * acin * p.Cout * 4;
}

template <typename T> static int launch_wgrad(const nunet_wgrad_desc* d, hipStream_t st) {
  typedef WgCfg<T> C;
  WgP p;
  const long grid = wgrad_setup<T>(d, p);
  double fl, by; wgrad_prof<T>(d, p, fl, by);
  ProfScope ps(p.Cout == 32 ? PC_WGRAD_1x4 : PC_WGRAD_2x2, fl, by, st);
  if (p.SH) hipLaunchKernelGGL((wgrad_kernel<T, true>), dim3((unsigned)grid), dim3(C::NT), 0, st, p);
  else hipLaunchKernelGGL((wgrad_kernel<T, false>), dim3((unsigned)grid), dim3(C::NT), 0, st, p);
  return nunet_check_launch("wgrad3x3");
}

struct WgPairArgs { const nunet_wgrad_desc* a; const nunet_wgrad_desc* b; };
template <typename T> static int launch_wgrad_pair(const WgPairArgs* w, hipStream_t st) {
  typedef WgCfg<T> C;
  WgP pa, pb;
  // in a pair the chip is filled by both problems together: the smaller one (fewer input channels) can take
  // fewer K-split slices, i.e. fewer same-address atomics (NUNET_WG_TARGET_SMALL, default 128: +0.7 % on the step)
  static int small_target = -1;
  if (small_target < 0) { const char* e = getenv("NUNET_WG_TARGET_SMALL"); small_target = e ? atoi(e) : 128; }
  const int cin_a = w->a->C0 + w->a->C1, cin_b = w->b->C0 + w->b->C1;
  const long ga = wgrad_setup<T>(w->a, pa, cin_a < cin_b ? small_target : 0), gb = wgrad_setup<T>(w->b, pb, cin_b < cin_a ? small_target : 0);
  double fa, ba, fb, bb; wgrad_prof<T>(w->a, pa, fa, ba);
  { const int keep = g_prof_alg_cin; g_prof_alg_cin = 0; wgrad_prof<T>(w->b, pb, fb, bb); g_prof_alg_cin = keep; }
  ProfScope ps(pa.Cout == 32 ? PC_WGRAD_1x4 : PC_WGRAD_2x2, fa + fb, ba + bb, st);
  if ((pa.SH != 0) != (pb.SH != 0)) {   // different tiling modes (different extents): two launches
    int rc = launch_wgrad<T>(w->a, st);
    return rc ? rc : launch_wgrad<T>(w->b, st);
  }
  if (pa.SH) hipLaunchKernelGGL((wgrad_pair_kernel<T, true>), dim3((unsigned)(ga + gb)), dim3(C::NT), 0, st, pa, pb, (int)ga);
  else hipLaunchKernelGGL((wgrad_pair_kernel<T, false>), dim3((unsigned)(ga + gb)), dim3(C::NT), 0, st, pa, pb, (int)ga);
  return nunet_check_launch("wgrad3x3 (pair)");
}

static int wgrad_check(const nunet_wgrad_desc* d) {
  NUNET_REQUIRE(d && d->src0 && d->dy && d->dw, "wgrad: null pointer");
  NUNET_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0, "wgrad: bad extent");
  NUNET_REQUIRE(d->C0 > 0 && d->C0 % 16 == 0 && d->C1 % 16 == 0, "wgrad: C0=%d C1=%d must be multiples of 16", d->C0, d->C1);
  NUNET_REQUIRE(d->C1 == 0 || d->src1, "wgrad: src1 null");
  NUNET_REQUIRE(d->Cout % 32 == 0, "wgrad: Cout=%d must be a multiple of 32", d->Cout);
  const int epv = 16 / dtype_size(d->dtype);
  NUNET_REQUIRE(d->P0 % epv == 0 && (d->C1 == 0 || d->P1 % epv == 0) && d->PY % epv == 0, "wgrad: pitch alignment");
  NUNET_REQUIRE((long)d->N * d->H * d->W < (1L << 30), "wgrad: too many pixels");
  return NUNET_OK;
}
extern "C" int nunet_conv3x3_wgrad(const nunet_wgrad_desc* d, nunet_stream_t s) {
  int rc = wgrad_check(d);
  if (rc) return rc;
  return NUNET_DISPATCH(d->dtype, launch_wgrad, d, (hipStream_t)s);
}
extern "C" int nunet_conv3x3_wgrad_pair(const nunet_wgrad_desc* a, const nunet_wgrad_desc* b, nunet_stream_t s) {
  int rc = wgrad_check(a);
  if (rc) return rc;
  rc = wgrad_check(b);
  if (rc) return rc;
  NUNET_REQUIRE(a->dtype == b->dtype, "wgrad_pair: the two problems must share the dtype");
  WgPairArgs w{a, b};
  return NUNET_DISPATCH(a->dtype, launch_wgrad_pair, &w, (hipStream_t)s);
}
