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
#include <type_traits>

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
  static __device__ __forceinline__ void sink(const Frag& a) { asm volatile("" :: "v"(a)); }   // diagnostic builds
};
template <> struct Mma<f16_t> {
  typedef f16x8 Frag;
  static __device__ __forceinline__ Frag load(const f16_t* p) { return *reinterpret_cast<const f16x8*>(p); }
  static __device__ __forceinline__ void mma(f32x16& acc, const Frag& a, const Frag& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  }
  static __device__ __forceinline__ void sink(const Frag& a) { asm volatile("" :: "v"(a)); }
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
  static __device__ __forceinline__ void sink(const Frag& a) { asm volatile("" :: "v"(a.lo), "v"(a.hi)); }
};

// value of the lane N places further on in the same row of 16 lanes (wrapping): after `x += dpp_row_ror<4>(x); x += dpp_row_ror<8>(x)`
// every lane holds the sum over the four lanes of its row that share lane % 4 (and with <8> alone: over lane % 8 pairs)
template <int N> __device__ __forceinline__ float dpp_row_ror(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x120 + N, 0xf, 0xf, false));
}

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
  long long* stats;      // fixed-point BatchNorm sums [rep][2][Cout] (common.h fx_add) or NULL
  int N, H, W, Cin, Cout;
  int NI, TH, TW, tilesX, tilesY, tilesG, nCoT, nItems;
  int SH;                // stacked-rows tiling: H + 1 (0 = off), see map_pixel
  unsigned SHinv;        // ceil(2^32 / SH)
  unsigned invS, invCoT, invTX, invTY;   // fastdiv_inv of S, nCoT, tilesX, tilesY (item decode)
  unsigned invTHW, invTW, invHH2HW2, invHW2;   // ... of TH*TW, TW, (TH+2)*(TW+2), TW+2 (tile-invariant tables of the prologue)
  int S, nch0, nch;      // K-split: slices, channel chunks of source 0 / total (SK kernels only)
  float* slabs;          // [S][pixels][Cout] fp32 partial sums
  long long slab_stride; // pixels * Cout
  // BNR kernels (dgrad of a block's second conv): the BatchNorm+ReLU backward REDUCE pass of the first
  // conv's BN is taken in the epilogue, on the values just stored (dst0 must be dense and assign-only)
  const void* bn_y; int bn_py;            // raw output of the first conv (what that BN normalised)
  const float* bn_mi; const float* bn_gamma; const float* bn_beta;   // saved mean | invstd, affine
  long long* bn_sums;                     // fixed point [rep][2][Cout]: sum dz, sum dz * xhat
  // LT kernels: the input (source 0, dense, C1 == 0) is TRANSFORMED between the global load and the LDS write
  //   LT 1: relu(bn(src0))                      - the BatchNorm+ReLU between the two convs of a VGGBlock
  //   LT 2: BatchNorm+ReLU backward apply        - src0 = dA, tf_y = raw y: dy = sc*(dz - k1 - xhat*k2)
  // so that neither the activation nor the gradient makes a round trip through HBM on the dependency chain;
  // the Cout-tile-0 workgroups store the transformed interior pixels for the weight gradient (tf_store).
  const void* tf_y; int tf_py;
  const long long* tf_fx;                 // LT 1 (training): sums of the producing conv; LT 2: sum dz, sum dz*xhat
  const float* tf_gamma; const float* tf_beta; const float* tf_bias;
  float* tf_rm; float* tf_rv; long long* tf_nbt;
  float* tf_save;                         // [2][Cin] mean | invstd: written by workgroup 0 (LT 1, training), read (LT 2)
  int tf_training; float tf_momentum, tf_eps;
  float* tf_dgamma; float* tf_dbeta; float* tf_dbias;   // LT 2: written by workgroup 0
  void* tf_store; int tf_ps;
  float M;                                // N * H * W
};

template <typename T, int WM_, int WN_, int SM_, int SN_> struct ConvCfg {
  static constexpr int WM = WM_, WN = WN_, SM = SM_, SN = SN_;
  static constexpr int NT = 64 * WM * WN;
  static constexpr int BM = 32 * SM * WM;
  static constexpr int BN = 32 * SN * WN;
  static constexpr int EPV = Tr<T>::EPV;
  static constexpr int KC = 64 / (int)sizeof(T);  // channels per LDS chunk (64 B per pixel)
  static constexpr int KS = KC / 16;              // MFMA k-steps per tap per chunk
  static constexpr int PS = KC + EPV;             // padded pixel stride (80 B)
  // halo pixel capacity. The 256 x 64 tile keeps it at 344 (a 16 x 16 tile has 324 halo pixels, 8 x 32 has 340): with 384 its
  // arena (halo + 46 KB of weights) is 2 KB past half of the CU's LDS and only one workgroup would be resident
  static constexpr int HPMAX = (SM * SN == 4) ? 344 : BM + BM / 2;
  static constexpr int UH = (HPMAX * 4 + NT - 1) / NT;
  static constexpr int UW = (9 * BN * 4 + NT - 1) / NT;
  static constexpr int OS = BN + EPV;             // epilogue staging row stride (elements)
  static constexpr int UO = (BM * (BN / EPV) + NT - 1) / NT;
  static constexpr int HALO_ELEMS = HPMAX * PS;
  static constexpr int W_ELEMS = 9 * BN * PS;
  static constexpr int STAGE_ELEMS = (HALO_ELEMS + W_ELEMS) > BM * OS ? (HALO_ELEMS + W_ELEMS) : BM * OS;
};

// floats of dynamic LDS a launch needs for its per-channel coefficient tables
// Diagnostic build only (-DNUNET_KSTAMP, tools/kstamp_build.sh): wave 0 of the first workgroups writes s_memtime
// at the phase boundaries of its first items, so a per-phase cycle budget of the persistent loop can be read back.
#ifdef NUNET_KSTAMP
__device__ unsigned long long* g_kstamp; __device__ int g_kstamp_wgs;
extern "C" int nunet_kstamp_set(void* buf, int wgs) {
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_kstamp), &buf, sizeof(buf)) != hipSuccess) return 1;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_kstamp_wgs), &wgs, sizeof(wgs)) != hipSuccess;
}
#define KSTAMP(i) do { if (threadIdx.x == 0 && (int)blockIdx.x < g_kstamp_wgs && (i) < 30) g_kstamp[blockIdx.x * 32 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
// slots 30 / 31: the chip-wide 100 MHz counter at workgroup entry / exit (s_memtime is per-CU and not comparable across workgroups)
#define KSTAMP_RT(i) do { if (threadIdx.x == 0 && (int)blockIdx.x < g_kstamp_wgs) g_kstamp[blockIdx.x * 32 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define KSTAMP(i) do {} while (0)
#define KSTAMP_RT(i) do {} while (0)
#endif

// Diagnostic build only (-DNUNET_ABLATE=bits, tools/ablate_build.sh): parts of the conv kernel are left out to see which
// resource bounds a layer (results are wrong by construction). 1 no MFMA | 2 no fragment reads | 4 no staging loads after
// the first | 8 no LDS staging writes after the first | 16 no epilogue | 32 no statistics atomics | 64 no output stores | 128 no y1 loads | 256 no BNR math | 512 no BNR cross-lane reduce | 1024 no transposition (LDS writes + stats) 
#ifndef NUNET_ABLATE
#define NUNET_ABLATE 0
#endif

static inline int conv_coef_floats(int lt, int cin, bool bnr, int cout) { return (lt == 1 ? 2 : lt == 2 ? 4 : 0) * cin + (bnr ? 4 * cout : 0) + cout; }

// Persistent kernel: each workgroup walks (tile, Cout-tile) items with stride gridDim.x.
// The global loads of the NEXT (item, channel chunk) are issued into registers before the
// current chunk's MFMA sweep, so HBM latency hides under compute and under the previous
// tile's epilogue, and co-resident workgroups de-synchronise their load/compute/store phases.
// Staging is branch-free: C0 and C1 are multiples of the chunk width, so the address of every staging unit is
// fixed for an item up to the chunk's (uniform) channel offset: pointers are set up once per item, a pixel
// outside the image points at a page of zeros, and a chunk's staging is 12 plain 16-byte loads.
// A launch carries up to CONV_GROUP_MAX independent problems of one kernel variant (same tile configuration, same fused
// BatchNorm work): the plan's single-stream schedule groups ready convolutions of different blocks of the x_{i,j} grid, whose
// workgroups then overlap each other's load / compute / store phases (two such launches side by side cost 1.3-1.4 x one,
// tools/conv_concurrency_probe.py). Workgroups are dealt to the problems round-robin (problem k owns `grid[k]` of them, sorted
// ascending), so all problems start together instead of one after the other.
struct ConvGroup { int n; int grid[CONV_GROUP_MAX]; ConvP p[CONV_GROUP_MAX]; };

// XCD-aware placement (for speed only, any placement is correct): workgroups are dealt round-robin over the 8 XCDs, each with its
// own 4 MB L2, so blocks b and b + 8 share an L2 and b, b + 1 never do. Work items that read the same data are CONSECUTIVE in item
// order (the Cout tiles of one pixel tile of a conv; the Cout x Cin tiles of one pixel slice of a weight gradient; neighbouring
// pixel tiles, whose halos overlap): taking item = block would send them to eight different L2s, each of which fetches the tile
// from memory again. The map below hands every XCD one contiguous range of the items instead: block b (XCD b % 8, the (b / 8)-th
// block that XCD receives) takes item base(b % 8) + b / 8.
#ifndef NUNET_XCD_REMAP
#define NUNET_XCD_REMAP 1
#endif
__device__ __forceinline__ int xcd_remap(int b, int G) {
  if (!NUNET_XCD_REMAP) return b;
  const int x = b & 7, l = b >> 3, q = G >> 3, rem = G & 7;
  return (x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + l;
}
__device__ __forceinline__ void conv_group_decode(const ConvGroup& g, int b, int& k, int& v) {
  int base = 0, prevg = 0, live = g.n;
  k = 0; v = b;
#pragma unroll
  for (int j = 0; j < CONV_GROUP_MAX; ++j) {
    if (j < g.n) {
      const int span = (g.grid[j] - prevg) * live;
      if (b >= base && b < base + span) { const int r = b - base; const int q = r / live; k = j + (r - q * live); v = prevg + q; }
      base += span; prevg = g.grid[j]; --live;
    }
  }
}

template <typename T, int WM, int WN, int SM, int SN, bool SK, bool BNR, int LT>
__global__ __launch_bounds__(64 * WM * WN, 2) void conv3x3_kernel(ConvGroup grp) {
  typedef ConvCfg<T, WM, WN, SM, SN> C;
  int gk = 0, vbid = (int)blockIdx.x;
  if (grp.n > 1) conv_group_decode(grp, (int)blockIdx.x, gk, vbid);
  gk = __builtin_amdgcn_readfirstlane(gk); vbid = __builtin_amdgcn_readfirstlane(vbid);
  const ConvP& p = grp.p[gk];
  const int vgrid = grp.grid[gk];
  typedef Mma<T> M;
  constexpr int PS = C::PS, EPV = C::EPV, BN = C::BN, BM = C::BM, NT = C::NT, KS = C::KS, OS = C::OS;

  // one LDS arena: [halo | weights] during the K loop, [BM][OS] output staging in the epilogue
  __shared__ __attribute__((aligned(16))) T s_buf[C::STAGE_ELEMS];
  __shared__ int s_hidx[BM];         // tile-invariant: halo index of output row m
  __shared__ int s_mxy[BM];          // tile-invariant: packed (ni, ly, lx) of row m, -1 unused
  __shared__ int s_hxy[C::HPMAX];    // tile-invariant: packed (ni, hy, hx) of halo pixel, -1 unused
  __shared__ int s_gpix[BM];         // per item: global pixel of row m, -1 masked
  __shared__ float s_red[2 * WM * BN];
  extern __shared__ __attribute__((aligned(16))) float s_coef[];   // LT tables [2 or 4][Cin], then BNR tables [4][Cout], then the bias [Cout]
  T* const s_halo = s_buf;
  T* const s_w = s_buf + C::HALO_ELEMS;
  float* const s_bnc = s_coef + (LT == 1 ? 2 : LT == 2 ? 4 : 0) * p.Cin;
  float* const s_bias = s_bnc + (BNR ? 4 : 0) * p.Cout;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 31, h = lane >> 5;
  const int HW2 = p.TW + 2, HH2 = p.TH + 2;
  const int HP = p.NI * HH2 * HW2;
  const int THW = p.TH * p.TW;
  KSTAMP(0); KSTAMP_RT(30);
  [[maybe_unused]] int kst = 4;   // NUNET_KSTAMP: next stamp slot of the loop

  // (divisions by launch constants as multiplies by host-computed inverses: the generic 32-bit division is ~30 instructions,
  //  and this prologue is a quarter of a workgroup's lifetime on the one-chunk layers of the first level)
  for (int m = tid; m < BM; m += NT) {
    // Which pixel of the tile does MFMA row m (lane m % 32 of its 32-row block) own? Not raster pixel m: gfx950 serves a
    // ds_read_b128 in the lane groups {0-3,12-15,20-27} | {4-11,16-19,28-31} (and the same + 32), and the 80-byte pixel stride is
    // conflict-free only for 16 CONSECUTIVE halo pixels per group. With raster order a group takes pixels 0-3, 12-15 of one tile
    // row and 4-11 of the next (18 halo pixels further on a 16-wide tile: residues 6-13 mod 16 meet 12, 13 - two 2-way conflicts
    // per fragment read). So the first group's lanes take raster pixels 0-15 of the
    // block and the second group's 16-31: every group reads one run of consecutive pixels whenever the tile width is a multiple
    // of 16 (measured: -1.6 % over the 59 conv launches of the step, +0.6 % on the step). Everything downstream (fragment bases,
    // output rows, statistics masks) goes through these tables.
    const int r32 = m & 31;
    const int q = (r32 < 4) ? r32 : (r32 < 12) ? r32 + 12 : (r32 < 16) ? r32 - 8 : (r32 < 20) ? r32 + 8 : (r32 < 28) ? r32 - 12 : r32;
    const int t = (m & ~31) + q;
    const int ni = fastdiv(t, p.invTHW);
    const int rem = t - ni * THW;
    const int ly = fastdiv(rem, p.invTW), lx = rem - ly * p.TW;
    const bool ok = ni < p.NI;
    s_hidx[m] = ok ? ((ni * HH2 + ly + 1) * HW2 + lx + 1) : (HW2 + 1);
    s_mxy[m] = ok ? ((ni << 20) | (ly << 10) | lx) : -1;
  }
  for (int hp = tid; hp < C::HPMAX; hp += NT) {
    int code = -1;
    if (hp < HP) {
      const int ni = fastdiv(hp, p.invHH2HW2);
      const int rem = hp - ni * (HH2 * HW2);
      const int hy = fastdiv(rem, p.invHW2), hx = rem - hy * HW2;
      code = (ni << 20) | (hy << 10) | hx;
    }
    s_hxy[hp] = code;
  }
  __syncthreads();
  KSTAMP(1);

  int abase[SM];
#pragma unroll
  for (int a = 0; a < SM; ++a) abase[a] = s_hidx[(wm * SM + a) * 32 + r] * PS + 8 * h;
  int toff[9];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) toff[tap] = ((tap / 3 - 1) * HW2 + (tap % 3 - 1)) * PS;
  const int bbase = ((wn * SN) * 32 + r) * PS + 8 * h;
  int hcode[C::UH];
  unsigned interior = 0u;            // LT: bit k = staging unit k is a pixel of the tile itself (not its halo ring)
#pragma unroll
  for (int k = 0; k < C::UH; ++k) {
    const int hp = (tid + k * NT) >> 2;
    hcode[k] = hp < C::HPMAX ? s_hxy[hp] : -1;
    if (LT != 0 && hcode[k] >= 0) {
      const int hy = (hcode[k] >> 10) & 1023, hx = hcode[k] & 1023;
      if (hy >= 1 && hy <= p.TH && hx >= 1 && hx <= p.TW) interior |= 1u << k;
    }
  }
  const int seg = tid & 3;           // 16-byte segment of a pixel's chunk row: the same for every unit of a thread (NT % 4 == 0)

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
  Vec16<T> yreg[LT == 2 ? C::UH : 1];
  Vec16<T> wreg[C::UW];
  // Staging addresses are 32-bit byte offsets from uniform bases (host-checked to fit): a unit's offset is
  // pixel * pitch + its 16-byte segment + the chunk's (uniform) channel offset - two VALU per unit and chunk and no
  // 64-bit pointer per unit held across the MFMA sweep. A pixel outside the image reads pixel 0 and is zeroed on the
  // way to LDS (write_lds), so a chunk's staging stays 12 plain 16-byte loads without branches.
  unsigned woff[C::UW];              // tile-invariant: weight row (tap, co within the Cout tile) + segment
#pragma unroll
  for (int k = 0; k < C::UW; ++k) {
    const int u = tid + k * NT;
    const int row = u >> 2;
    const int tap = row / BN, co = row - tap * BN;
    // (a unit past the last weight row exists when 9*BN*4 is not a multiple of NT; it re-reads row 0 and is never written to LDS)
    woff[k] = row < 9 * BN ? (unsigned)(((tap * p.Cout + co) * p.Cin + seg * EPV) * (int)sizeof(T)) : (unsigned)(seg * 16);
  }
  unsigned wco = 0u;                 // the loaded item's Cout tile: co0 * Cin elements, in bytes (uniform)
  auto set_ptrs = [&](const Item& it) { wco = (unsigned)(it.co0 * p.Cin * (int)sizeof(T)); };
  // The staging loads of a chunk: uniform part (source, pitch, channel offset) once, then one 16-byte load per unit.
  // The first chunk of a workgroup issues them back to back (load_regs); every later chunk's loads are spread over the
  // MFMA sweep of the chunk before it (sweep<true>), one per step: a wave's 12 loads take ~16 clocks each of the CU's
  // address/L1 path, and issued as a block they stall the wave for ~1200 clocks with the matrix pipe idle.
  struct LoadCtx { const char* base; unsigned pb, cb, yb, ycb, wb; };
  auto load_ctx = [&](int kb) {
    LoadCtx c;
    const bool s0 = LT != 0 || kb < p.C0;
    c.base = (const char*)(s0 ? p.src0 : p.src1);
    c.pb = (unsigned)(s0 ? p.P0 : p.P1) * (unsigned)sizeof(T);
    c.cb = (unsigned)((s0 ? kb : kb - p.C0) * (int)sizeof(T) + seg * 16);
    c.yb = (unsigned)p.tf_py * (unsigned)sizeof(T);
    c.ycb = (unsigned)(kb * (int)sizeof(T) + seg * 16);
    c.wb = (unsigned)(kb * (int)sizeof(T)) + wco;
    return c;
  };
  constexpr int NU = C::UH + C::UW;
  auto load_unit = [&](const LoadCtx& c, int u) {       // u is a compile-time constant after unrolling
    if (u < C::UH) {
      const unsigned gp = hgp[u] < 0 ? 0u : (unsigned)hgp[u];
      hreg[u].raw = *reinterpret_cast<const u32x4*>(c.base + (gp * c.pb + c.cb));
      if constexpr (LT == 2) yreg[u].raw = *reinterpret_cast<const u32x4*>((const char*)p.tf_y + (gp * c.yb + c.ycb));
    } else {
      wreg[u - C::UH].raw = *reinterpret_cast<const u32x4*>((const char*)p.w + (woff[u - C::UH] + c.wb));
    }
  };
  auto load_regs = [&](int kb) {
    const LoadCtx c = load_ctx(kb);
#pragma unroll
    for (int u = 0; u < NU; ++u) load_unit(c, u);
  };
  // registers -> LDS; LT kernels transform the input on the way. `kb` / `store` belong to the chunk held in the
  // registers (hgp[] still describes its item: the next item's pixels are mapped only after this write).
  auto write_lds = [&](int kb, bool store) {
    if constexpr (LT == 0) {
#pragma unroll
      for (int k = 0; k < C::UH; ++k) {
        const int u = tid + k * NT;
        // (the bound check folds away for every k whose whole NT-unit run exists: no branch per unit)
        if ((k + 1) * NT <= C::HPMAX * 4 || (u >> 2) < C::HPMAX) st16(&s_halo[(u >> 2) * PS + seg * EPV], hgp[k] >= 0 ? hreg[k] : zero16<T>());
      }
    } else {
      typedef typename FV<T>::type V;
      const int c0 = kb + seg * EPV;
      const V sc = ldf<T>(&s_coef[c0]), sh = ldf<T>(&s_coef[p.Cin + c0]);
      V cA, cB;
      if constexpr (LT == 2) { cA = ldf<T>(&s_coef[2 * p.Cin + c0]); cB = ldf<T>(&s_coef[3 * p.Cin + c0]); }
#pragma unroll
      for (int k = 0; k < C::UH; ++k) {
        const int u = tid + k * NT;
        if ((k + 1) * NT <= C::HPMAX * 4 || (u >> 2) < C::HPMAX) {
          const bool ok = hgp[k] >= 0;        // zero padding stays zero: the conv pads the ACTIVATION, not the raw tensor
          V v;
          if constexpr (LT == 1) v = bn_relu_apply<V>(vec_to_f<T>(hreg[k]), sc, sh);
          else v = bn_relu_bwd_apply<V>(vec_to_f<T>(hreg[k]), vec_to_f<T>(yreg[k]), sc, sh, cA, cB);
          Vec16<T> o = vec_from_f<T>(v);
          if (!ok) o = zero16<T>();
          st16(&s_halo[(u >> 2) * PS + seg * EPV], o);
          if (store && ok && ((interior >> k) & 1u)) st16((T*)p.tf_store + (size_t)hgp[k] * p.tf_ps + c0, o);
        }
      }
    }
#pragma unroll
    for (int k = 0; k < C::UW; ++k) {
      const int u = tid + k * NT;
      if ((k + 1) * NT <= 9 * BN * 4 || (u >> 2) < 9 * BN) st16(&s_w[(u >> 2) * PS + seg * EPV], wreg[k]);
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
  constexpr int NSTEP = 9 * KS;
  // fragments of the next step are read while the current one multiplies; LOADS: the next chunk's staging loads are
  // issued one per step between the fragment reads and the MFMAs (the scheduling barriers keep them there)
  auto sweep = [&](auto LOADS, const LoadCtx& lc) {
    constexpr bool WITH_LOADS = decltype(LOADS)::value;
    // Fragment reads run PD steps ahead of the MFMAs that consume them (PD + 1 register buffers). Measured: 2 or 3
    // steps ahead cost 12-30 registers and change no layer's time (tools/conv_layers.py), so one step it is. (Round 3, again and
    // together with a second accumulator chain for the one-tile-per-wave configuration - two independent MFMA chains per wave -:
    // 1192 us over the 59 launches either way; the second chain alone costs the BatchNorm-loader kernels 14 %: its fold and
    // registers. Neither the fragment-read latency nor the MFMA dependency is what a sweep waits for.)
    constexpr int PD = 1;
    constexpr int NB = PD + 1;
    typename M::Frag fa[NB][SM], fb[NB][SN];
    auto read_step = [&](int j) {                 // j: compile-time after unrolling
      const int tap = j / KS, ks = j % KS, bu = j % NB;
#pragma unroll
      for (int a = 0; a < SM; ++a) fa[bu][a] = M::load(&s_halo[abase[a] + toff[tap] + ks * 16]);
#pragma unroll
      for (int b = 0; b < SN; ++b) fb[bu][b] = M::load(&s_w[bbase + (tap * BN + b * 32) * PS + ks * 16]);
    };
#pragma unroll
    for (int j = 0; j < PD; ++j) read_step(j);
#pragma unroll
    for (int j = 0; j < NSTEP; ++j) {
      if (j + PD < NSTEP && !(NUNET_ABLATE & 2)) read_step(j + PD);
      if constexpr (WITH_LOADS && !(NUNET_ABLATE & 4)) {
#pragma unroll
        for (int u = 0; u < NU; ++u)
          if ((u * NSTEP) / NU == j) load_unit(lc, u);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int a = 0; a < SM; ++a)
#pragma unroll
        for (int b = 0; b < SN; ++b) {
          if constexpr (NUNET_ABLATE & 1) { M::sink(fa[j % NB][a]); M::sink(fb[j % NB][b]); }
          else M::mma(acc[a][b], fa[j % NB][a], fb[j % NB][b]);
        }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto epi_sk = [&](const Item& cur) {
    // ---- K-split epilogue: this slice's fp32 partial tile goes to its slab (plain stores,
    // 128-byte runs per half-wave); splitk_finalize_kernel sums the slabs deterministically
    float* slab = p.slabs + (size_t)cur.ks * p.slab_stride;
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
  };
  static_assert(SM * 16 <= 64, "row-validity mask of the epilogue is one 64-bit word");
  // BNR: the y1 vectors of this thread's store units. They are requested BEFORE the last chunk's sweep, i.e. before the
  // next item's staging loads: loads return in order, so the epilogue's wait for them leaves those staging loads in
  // flight (requested in the epilogue, behind them, the wait drained all twelve and cost a memory latency per item).
  Vec16<T> byv[BNR ? C::UO : 1];
  auto issue_byv = [&](const Item& cur) {
    if constexpr (BNR) {
      constexpr int SEGS = BN / EPV;
      const int sg = tid % SEGS, m0 = tid / SEGS;
#pragma unroll
      for (int k = 0; k < C::UO; ++k) {
        const int m = m0 + k * (NT / SEGS);
        const int gp = m < BM ? s_gpix[m] : -1;
        byv[k] = ld16((const T*)p.bn_y + (size_t)(gp < 0 ? 0 : gp) * p.bn_py + cur.co0 + sg * EPV);
      }
    }
  };
  auto epi_plain = [&](const Item& cur, T* const s_out) {
    // ---- epilogue: bias, BN partial sums from registers, LDS transpose, 16-byte stores ----
    // Nothing in here waits on a global LOAD unless a destination accumulates (the bias sits in an LDS table since the
    // prologue): a wait would also drain the next item's staging loads already in flight (the counter is in-order).
    constexpr int SEGS = BN / EPV;
    const int sg = tid % SEGS, m0 = tid / SEGS;       // NT % SEGS == 0: every store unit of a thread has the same channel segment
    int gpu[C::UO];
#pragma unroll
    for (int k = 0; k < C::UO; ++k) {
      const int m = m0 + k * (NT / SEGS);
      gpu[k] = m < BM ? s_gpix[m] : -1;
    }
    // which of this thread's accumulator rows are pixels of the image (all 16*SM reads in flight at once)
    unsigned long long rowmask = 0ull;
    if (p.stats) {
#pragma unroll
      for (int a = 0; a < SM; ++a)
#pragma unroll
        for (int i = 0; i < 16; ++i) rowmask |= (s_gpix[(wm * SM + a) * 32 + acc_row(i, h)] >= 0 ? 1ull : 0ull) << (a * 16 + i);
    }
    KSTAMP(kst); ++kst;       // g: row masks / unit pixels read
    __syncthreads();  // every wave finished reading halo/weights: the arena becomes staging
    KSTAMP(kst); ++kst;       // h: barrier passed
#pragma unroll
    for (int b = 0; b < SN; ++b) {
      const int cl = (wn * SN + b) * 32 + r;  // channel within the tile
      const float bias = s_bias[cur.co0 + cl];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int a = 0; a < SM; ++a) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int m = (wm * SM + a) * 32 + acc_row(i, h);
          const T tv = from_f32<T>(acc[a][b][i] + bias);
          if (!(NUNET_ABLATE & 1024)) s_out[m * OS + cl] = tv;
          const float d = ((rowmask >> (a * 16 + i)) & 1ull) ? to_f32(tv) - bias : 0.f;
          s1 += d; s2 += d * d;
          acc[a][b][i] = 0.f;
        }
      }
      if (p.stats) {
        s1 += __shfl_xor(s1, 32);
        s2 += __shfl_xor(s2, 32);
        if (h == 0) { s_red[(wm * BN + cl) * 2 + 0] = s1; s_red[(wm * BN + cl) * 2 + 1] = s2; }
      }
    }
    KSTAMP(kst); ++kst;       // i: accumulators transposed into LDS
    __syncthreads();
    KSTAMP(kst); ++kst;       // j: barrier passed
    // BNR: the BN-backward partial sums of the thread's 8 channels live in registers across the store loop
    typedef typename FV<T>::type V;
    V bmean, bistd, bsc, bsh, r1 = V(0.f), r2 = V(0.f);
    if constexpr (BNR) {
      const int c0 = cur.co0 + sg * EPV;
      bmean = ldf<T>(&s_bnc[c0]); bistd = ldf<T>(&s_bnc[p.Cout + c0]); bsc = ldf<T>(&s_bnc[2 * p.Cout + c0]); bsh = ldf<T>(&s_bnc[3 * p.Cout + c0]);
    }
    // destination of the thread's channel segment (the same for every unit)
    const int co = cur.co0 + sg * EPV;
    const bool to0 = co < p.D0;
    T* const qb = to0 ? (T*)p.dst0 + co : (T*)p.dst1 + (co - p.D0);
    // (readfirstlane: a per-lane select between two kernel arguments otherwise becomes a per-lane LOAD from the
    //  argument segment, and the wait for it drains the staging loads in flight)
    const int qs = to0 ? __builtin_amdgcn_readfirstlane(p.Q0) : __builtin_amdgcn_readfirstlane(p.Q1);
    const bool accum = to0 ? ((p.acc0_mask >> (p.slot_w > 0 ? fastdiv(co, p.inv_slot_w) : 0)) & 1u) != 0u : p.acc1 != 0;
    const bool any_accum = p.acc0_mask != 0u || p.acc1 != 0;     // uniform: plain launches never branch per unit
    Vec16<T> vv[C::UO];
#pragma unroll
    for (int k = 0; k < C::UO; ++k) {
      const int m = m0 + k * (NT / SEGS);
      vv[k] = ld16(&s_out[(m < BM ? m : 0) * OS + sg * EPV]);
    }
#pragma unroll
    for (int k = 0; k < C::UO; ++k) {
      if (gpu[k] >= 0) {
        T* const q = qb + (size_t)gpu[k] * qs;
        Vec16<T> v = vv[k];
        if (any_accum && accum) v = vec_from_f<T>(vec_to_f<T>(v) + vec_to_f<T>(ld16(q)));
        if (!(NUNET_ABLATE & 64)) st16(q, v); else asm volatile("" :: "v"(v.raw));
        if constexpr (BNR && !(NUNET_ABLATE & 256)) {
          // whole-vector math (the element accessors cost ~3x the instructions): dz = the stored (rounded) gradient
          // where the forward activation was positive; sums of dz and dz * xhat
          const V yf = vec_to_f<T>(byv[k]);
          const V dz = __builtin_elementwise_fma(yf, bsc, bsh) > V(0.f) ? vec_to_f<T>(v) : V(0.f);
          r1 += dz;
          r2 += dz * ((yf - bmean) * bistd);
        }
      }
    }
    KSTAMP(kst); ++kst;       // k: stores issued
    if constexpr (BNR && !(NUNET_ABLATE & 512)) {
      // lanes that share a channel segment (lane % SEGS) are summed with xor-shuffles, the four waves
      // through a small LDS table (fixed order), then one fixed-point add per channel and sum
      // (strides 4 and 8 stay inside a row of 16 lanes: DPP row rotates, no LDS crossbar round trip; 16 and 32 shuffle)
#pragma unroll
      for (int off = SEGS; off < 64; off <<= 1) {
#pragma unroll
        for (int e = 0; e < EPV; ++e) {
          if (off == 4) { r1[e] += dpp_row_ror<4>(r1[e]); r2[e] += dpp_row_ror<4>(r2[e]); }
          else if (off == 8) { r1[e] += dpp_row_ror<8>(r1[e]); r2[e] += dpp_row_ror<8>(r2[e]); }
          else { r1[e] += __shfl_xor(r1[e], off); r2[e] += __shfl_xor(r2[e], off); }
        }
      }
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
        if (!(NUNET_ABLATE & 32)) fx_add(p.bn_sums + ((size_t)((vbid & (bn_sum_replicas(p.Cout) - 1)) * 2 + vsel) * p.Cout + cur.co0 + c) * NUNET_FX_WORDS, sum); else asm volatile("" :: "v"(sum));
      }
    }
    if (p.stats) {
      for (int t = tid; t < 2 * BN; t += NT) {
        const int vsel = t / BN, c = t - vsel * BN;
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < WM; ++k) sum += s_red[(k * BN + c) * 2 + vsel];
        if (!(NUNET_ABLATE & 32)) fx_add(p.stats + ((size_t)((vbid & (bn_sum_replicas(p.Cout) - 1)) * 2 + vsel) * p.Cout + cur.co0 + c) * NUNET_FX_WORDS, sum); else asm volatile("" :: "v"(sum));
      }
    }
  };
  // (a group's problems interleave their workgroups, so block -> XCD is no longer vbid % 8: no remap there)
  int item = grp.n == 1 ? xcd_remap(vbid, vgrid) : vbid;
  if (item >= p.nItems) return;
  Item cur = decode(item);
  set_hgp(cur);
  set_ptrs(cur);
  int cc = 0, c_hi = 0;
  int kb = 0;
  if constexpr (SK) { cc = cur.ks * p.nch / p.S; c_hi = (cur.ks + 1) * p.nch / p.S; kb = chunk_kb(cc); }
  KSTAMP(2);
  load_regs(kb);
  bool first_chunk = true;
  // per-channel coefficient tables, AFTER the first tile's loads were issued (the two global-memory latencies overlap);
  // the barrier at the top of the loop orders them before the first write_lds / epilogue
  for (int c = tid; c < p.Cout; c += NT) s_bias[c] = p.bias ? p.bias[c] : 0.f;
  if constexpr (BNR) {
    for (int c = tid; c < p.Cout; c += NT) {
      const float mean = p.bn_mi[c], istd = p.bn_mi[p.Cout + c];
      const float sc = p.bn_gamma[c] * istd;
      s_bnc[c] = mean; s_bnc[p.Cout + c] = istd; s_bnc[2 * p.Cout + c] = sc; s_bnc[3 * p.Cout + c] = __builtin_fmaf(-mean, sc, p.bn_beta[c]);
    }
  }
  // (measured and not kept, round 3: requesting those sums at the very top of the kernel - into registers, 64 VGPRs and a
  //  workgroup less per CU, or as a one-dword-per-line touch - so that their round trip runs under the index-table prologue:
  //  the lines were last written by the producer's memory-side atomics and come back slowly, the vector-memory counter is
  //  in order, and the first tile's staging loads - issued later - then wait behind them: +1.7 % over the 59 conv launches)
  if constexpr (LT == 1) {
    // BatchNorm coefficients of the INPUT channels, from the producing conv's fixed-point sums (every workgroup
    // derives the same values; workgroup 0 also owns the running-statistics update and the saved mean / invstd)
    BnStatArgs a; a.fx = p.tf_fx; a.conv_bias = p.tf_bias; a.rm = p.tf_rm; a.rv = p.tf_rv; a.C = p.Cin; a.training = p.tf_training; a.M = p.M; a.eps = p.tf_eps;
    for (int c = tid; c < p.Cin; c += NT) {
      float mean, invstd, var, mf;
      bn_stat_coeffs(a, c, mean, invstd, var, mf);
      const float sc = p.tf_gamma[c] * invstd;
      s_coef[c] = sc; s_coef[p.Cin + c] = __builtin_fmaf(-mean, sc, p.tf_beta[c]);
      if (vbid == 0 && p.tf_training) {
        if (p.tf_save) { p.tf_save[c] = mean; p.tf_save[p.Cin + c] = invstd; }
        if (p.tf_rm) bn_running_update(p.tf_rm, p.tf_rv, c, p.tf_momentum, mf, var, p.M);
      }
    }
    if (vbid == 0 && p.tf_training && tid == 0 && p.tf_nbt) *p.tf_nbt += 1;
  }
  if constexpr (LT == 2) {
    const int nrep = bn_sum_replicas(p.Cin);
    for (int c = tid; c < p.Cin; c += NT) {
      const float mean = p.tf_save[c], istd = p.tf_save[p.Cin + c];
      const float sc = p.tf_gamma[c] * istd;
      double t1, t2;
      fx_totals(p.tf_fx, p.Cin, nrep, c, t1, t2);
      float A, B;
      bn_bwd_AB(mean, istd, sc, (float)(t1 / (double)p.M), (float)(t2 / (double)p.M), A, B);
      s_coef[c] = sc; s_coef[p.Cin + c] = __builtin_fmaf(-mean, sc, p.tf_beta[c]); s_coef[2 * p.Cin + c] = A; s_coef[3 * p.Cin + c] = B;
      if (vbid == 0) {
        // d beta = sum dz, d gamma = sum dz * xhat; the conv bias in front of a BatchNorm has gradient sum(dy) == 0
        if (p.tf_dbeta) p.tf_dbeta[c] = (float)t1;
        if (p.tf_dgamma) p.tf_dgamma[c] = (float)t2;
        if (p.tf_dbias) p.tf_dbias[c] = 0.f;
      }
    }
  }

  KSTAMP(3);
  while (true) {
    __syncthreads();  // previous chunk's fragment reads / previous item's epilogue reads are done
    KSTAMP(kst); ++kst;       // a: barrier passed
    if (!(NUNET_ABLATE & 8) || (item == vbid && first_chunk)) write_lds(kb, LT != 0 && p.tf_store != nullptr && cur.co0 == 0);
    KSTAMP(kst); ++kst;       // b: loads arrived, transformed, written to LDS
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
    KSTAMP(kst); ++kst;       // c: barrier passed
    // prefetch the next (item, chunk) into registers
    int nkb = kb + C::KC, nitem = item, ncc = cc + 1, nc_hi = c_hi;
    Item nxt = cur;
    bool have_next = true;
    bool last_chunk;
    if constexpr (SK) last_chunk = ncc >= c_hi; else last_chunk = nkb >= p.Cin;
    if (last_chunk) {
      nkb = 0;
      nitem = item + vgrid;
      if (nitem < p.nItems) {
        nxt = decode(nitem); set_hgp(nxt); set_ptrs(nxt);
        if constexpr (SK) { ncc = nxt.ks * p.nch / p.S; nc_hi = (nxt.ks + 1) * p.nch / p.S; }
      } else have_next = false;
    }
    if (have_next) {
      if constexpr (SK) nkb = chunk_kb(ncc);
    }
    const LoadCtx lc = load_ctx(nkb);
    if constexpr (BNR && !SK && !(NUNET_ABLATE & 128)) { if (last_chunk) issue_byv(cur); }
    KSTAMP(kst); ++kst;       // d: next item decoded
    if (have_next) sweep(std::true_type{}, lc); else sweep(std::false_type{}, lc);
    KSTAMP(kst); ++kst;       // e: MFMA sweep done
    first_chunk = false;
    if (last_chunk) {
      if constexpr (NUNET_ABLATE & 16) {
#pragma unroll
        for (int a = 0; a < SM; ++a)
#pragma unroll
          for (int b = 0; b < SN; ++b) asm volatile("" : "+v"(acc[a][b]));
      } else if constexpr (SK) epi_sk(cur); else epi_plain(cur, s_buf);
      KSTAMP(kst); ++kst;     // f: epilogue done
      if (!have_next) { KSTAMP_RT(31); break; }
      cur = nxt; item = nitem; first_chunk = true;
    }
    kb = nkb; cc = ncc; c_hi = nc_hi;
  }
}

// Finishes a K-split convolution: sum of the S fp32 slabs (fixed order: deterministic) -> (+bias)
// -> T, routed to the two destinations with the per-slot accumulate mask, BatchNorm partial sums.
// The block size is a multiple of the channel groups G, so every element a thread visits has the same
// channel group: statistics are summed in registers, meet in LDS in a fixed order, one fixed-point add per channel.
struct SplitFinP {
  const float* slabs; long long slab_stride; int S; const float* bias;
  void* dst0; void* dst1; int D0, D1, Q0, Q1; int slot_w; unsigned acc0_mask; int acc1; unsigned inv_slot_w;
  long long* stats; long long npix; int Cout;
  const void* bn_y; int bn_py; const float* bn_mi; const float* bn_gamma; const float* bn_beta; long long* bn_sums;   // see ConvP
};
template <typename T, bool BNR>
__global__ __launch_bounds__(256) void splitk_finalize_kernel(SplitFinP p) {
  constexpr int EPV = Tr<T>::EPV;
  constexpr int NV = BNR ? 4 : 2;
  __shared__ float s_part[256 * NV * EPV];
  const int G = p.Cout / EPV;                 // blockDim.x % G == 0
  const int cg = threadIdx.x % G, co = cg * EPV;
  const int ppb = blockDim.x / G, pl = threadIdx.x / G;
  float q1[EPV], q2[EPV], r1[BNR ? EPV : 1], r2[BNR ? EPV : 1];
  float bmean[BNR ? EPV : 1], bistd[BNR ? EPV : 1], bsc[BNR ? EPV : 1], bsh[BNR ? EPV : 1], bias[EPV];
#pragma unroll
  for (int e = 0; e < EPV; ++e) {
    q1[e] = 0.f; q2[e] = 0.f;
    bias[e] = p.bias ? p.bias[co + e] : 0.f;
    if constexpr (BNR) {
      const int c = co + e;
      const float mean = p.bn_mi[c], istd = p.bn_mi[p.Cout + c];
      const float sc = p.bn_gamma[c] * istd;
      bmean[e] = mean; bistd[e] = istd; bsc[e] = sc; bsh[e] = __builtin_fmaf(-mean, sc, p.bn_beta[c]);
      r1[e] = 0.f; r2[e] = 0.f;
    }
  }
  T* q; bool accum; long long qstride;
  if (co < p.D0) { q = (T*)p.dst0 + co; qstride = p.Q0; accum = (p.acc0_mask >> (p.slot_w > 0 ? fastdiv(co, p.inv_slot_w) : 0)) & 1u; }
  else { q = (T*)p.dst1 + (co - p.D0); qstride = p.Q1; accum = p.acc1 != 0; }
  for (long long pix = (long long)blockIdx.x * ppb + pl; pix < p.npix; pix += (long long)gridDim.x * ppb) {
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
    T* qq = q + pix * qstride;
    const Vec16<T> o = accum ? ld16(qq) : zero16<T>();
    Vec16<T> v;
    float stored[EPV];
#pragma unroll
    for (int e = 0; e < EPV; ++e) {
      float y = x[e] + bias[e];
      if (accum) y += o.get(e);
      v.set(e, y);
      stored[e] = to_f32(from_f32<T>(y));   // the stored (rounded) value; not v.get(e) right after v.set(e)
      const float d = stored[e] - bias[e];
      q1[e] += d; q2[e] += d * d;
    }
    st16(qq, v);
    if constexpr (BNR) {
      const Vec16<T> yv = ld16((const T*)p.bn_y + pix * p.bn_py + co);
#pragma unroll
      for (int e = 0; e < EPV; ++e) {
        const float yy = yv.get(e);
        const float dz = __builtin_fmaf(yy, bsc[e], bsh[e]) > 0.f ? stored[e] : 0.f;
        r1[e] += dz; r2[e] += dz * ((yy - bmean[e]) * bistd[e]);
      }
    }
  }
  if (!p.stats && !BNR) return;
#pragma unroll
  for (int e = 0; e < EPV; ++e) {
    s_part[(threadIdx.x * NV + 0) * EPV + e] = q1[e];
    s_part[(threadIdx.x * NV + 1) * EPV + e] = q2[e];
    if constexpr (BNR) { s_part[(threadIdx.x * NV + 2) * EPV + e] = r1[e]; s_part[(threadIdx.x * NV + 3) * EPV + e] = r2[e]; }
  }
  __syncthreads();
  const int rep = blockIdx.x & (bn_sum_replicas(p.Cout) - 1);
  for (int t = threadIdx.x; t < NV * p.Cout; t += blockDim.x) {
    const int v = t / p.Cout, c = t - v * p.Cout;
    const int g = c / EPV, e = c - g * EPV;
    float sum = 0.f;
    for (int qd = 0; qd < ppb; ++qd) sum += s_part[((qd * G + g) * NV + v) * EPV + e];
    if (v < 2) { if (p.stats) fx_add(p.stats + ((size_t)(rep * 2 + v) * p.Cout + c) * NUNET_FX_WORDS, sum); }
    else fx_add(p.bn_sums + ((size_t)(rep * 2 + (v - 2)) * p.Cout + c) * NUNET_FX_WORDS, sum);
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
  if (W <= BM && (long)N * (H + 1) < (1 << 20) && H + 1 <= 4096) {
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

template <typename T, int WM, int WN, int SM, int SN, bool SK, bool BNR>
static void launch_conv_lt(int lt, unsigned grid, size_t dyn, hipStream_t st, const ConvGroup& g) {
  typedef ConvCfg<T, WM, WN, SM, SN> C;
  if (lt == 1) NUNET_LAUNCH((conv3x3_kernel<T, WM, WN, SM, SN, SK, BNR, 1>), dim3(grid), dim3(C::NT), dyn, st, g);
  else if (lt == 2) NUNET_LAUNCH((conv3x3_kernel<T, WM, WN, SM, SN, SK, BNR, 2>), dim3(grid), dim3(C::NT), dyn, st, g);
  else NUNET_LAUNCH((conv3x3_kernel<T, WM, WN, SM, SN, SK, BNR, 0>), dim3(grid), dim3(C::NT), dyn, st, g);
}

// everything a launch of one problem needs, for one tile configuration
struct ConvSetup { ConvP p; long grid; size_t dyn; bool bnr; int lt; double flops, bytes; };
template <typename T, int WM, int WN, int SM, int SN>
static void conv_setup(const nunet_conv_desc* d, ConvSetup& S) {
  typedef ConvCfg<T, WM, WN, SM, SN> C;
  ConvP& p = S.p;
  p.src0 = d->src0; p.src1 = d->src1; p.C0 = d->C0; p.C1 = d->C1; p.P0 = d->P0; p.P1 = d->P1;
  p.w = d->wpack; p.bias = d->bias;
  p.dst0 = d->dst0; p.dst1 = d->dst1; p.D0 = d->D0; p.D1 = d->D1; p.Q0 = d->Q0; p.Q1 = d->Q1;
  p.slot_w = d->acc_slot_w; p.acc0_mask = d->acc0_mask; p.acc1 = d->acc1; p.inv_slot_w = fastdiv_inv(d->acc_slot_w);
  p.stats = (long long*)d->stats;
  p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->C0 + d->C1; p.Cout = d->D0 + d->D1;
  p.M = (float)d->N * d->H * d->W;
  const bool bnr = d->bn_y != nullptr;
  const int lt = d->in_tf;
  p.bn_y = d->bn_y; p.bn_py = d->bn_py; p.bn_mi = d->bn_mean_invstd; p.bn_gamma = d->bn_gamma; p.bn_beta = d->bn_beta; p.bn_sums = (long long*)d->bn_sums;
  p.tf_y = d->tf_y; p.tf_py = d->tf_py; p.tf_fx = (const long long*)d->tf_fx; p.tf_gamma = d->tf_gamma; p.tf_beta = d->tf_beta; p.tf_bias = d->tf_conv_bias;
  p.tf_rm = d->tf_running_mean; p.tf_rv = d->tf_running_var; p.tf_nbt = (long long*)d->tf_nbt; p.tf_save = d->tf_mean_invstd;
  p.tf_training = d->tf_training; p.tf_momentum = d->tf_momentum; p.tf_eps = d->tf_eps;
  p.tf_dgamma = d->tf_dgamma; p.tf_dbeta = d->tf_dbeta; p.tf_dbias = d->tf_dbias; p.tf_store = d->tf_store; p.tf_ps = d->tf_ps;
  const TileGeom g = nunet_choose_tile(d->N, d->H, d->W, C::BM, C::HPMAX);
  p.NI = g.NI; p.TH = g.TH; p.TW = g.TW; p.tilesX = g.tilesX; p.tilesY = g.tilesY; p.tilesG = g.tilesG; p.SH = g.SH;
  p.SHinv = g.SH ? (unsigned)(((1ull << 32) + g.SH - 1) / g.SH) : 0u;
  p.nCoT = p.Cout / C::BN;
  long items = (long)p.nCoT * g.tilesX * g.tilesY * g.tilesG;
  // K-split for the grid-starved deep levels: slices of the channel-chunk loop become extra items, each
  // writes an fp32 partial slab; splitk_finalize_kernel sums them (fixed order, deterministic)
  p.S = 1; p.slabs = nullptr; p.slab_stride = 0;
  p.nch0 = p.C0 / C::KC;
  p.nch = p.nch0 + p.C1 / C::KC;
  // (measured on the 96x96 workload, tools/conv_layers.py: below ~60 items a split pays for its finalize launch, above it does not)
#ifndef NUNET_SK_ITEMS
#define NUNET_SK_ITEMS 60
#define NUNET_SK_TARGET 320
#endif
  if (d->splitk_ws && items <= NUNET_SK_ITEMS && p.nch >= 8 && p.Cout <= 1024 && (256 / (p.Cout / C::EPV)) >= 1) {
    int Sn = (int)((NUNET_SK_TARGET + items - 1) / items);
    if (Sn > p.nch / 2) Sn = p.nch / 2;
    const long long need = (long long)Sn * d->N * d->H * d->W * p.Cout;
    if (Sn > 1 && need <= d->splitk_ws_floats) {
      p.S = Sn; p.slabs = d->splitk_ws; p.slab_stride = (long long)d->N * d->H * d->W * p.Cout;
      items *= Sn;
    }
  }
  p.nItems = (int)items;
  p.invS = fastdiv_inv(p.S); p.invCoT = fastdiv_inv(p.nCoT); p.invTX = fastdiv_inv(p.tilesX); p.invTY = fastdiv_inv(p.tilesY);
  p.invTHW = fastdiv_inv(g.TH * g.TW); p.invTW = fastdiv_inv(g.TW); p.invHH2HW2 = fastdiv_inv((g.TH + 2) * (g.TW + 2)); p.invHW2 = fastdiv_inv(g.TW + 2);
  // persistent grid: resident workgroups only, item counts balanced across them
  S.bnr = bnr; S.lt = lt;
  S.dyn = sizeof(float) * (size_t)conv_coef_floats(lt, p.Cin, bnr && p.S == 1, p.Cout);
  const size_t lds_bytes = sizeof(T) * C::STAGE_ELEMS + 4 * (3 * C::BM + C::HPMAX) + 8 * WM * C::BN + S.dyn;
  long per_cu = (long)(160 * 1024 / lds_bytes);
  if (per_cu < 1) per_cu = 1;
  if (per_cu > 2048 / C::NT) per_cu = 2048 / C::NT;
  const long resident = 256 * per_cu;
  const long rounds = (items + resident - 1) / resident;
  S.grid = (items + rounds - 1) / rounds;
  const double px = (double)d->N * d->H * d->W;
  const int acin = g_prof_alg_cin > 0 ? g_prof_alg_cin : p.Cin;
  S.flops = 2.0 * 9 * acin * p.Cout * px;
  S.bytes = (px * (acin * (lt == 2 ? 2 : 1) + p.Cout) + 9.0 * acin * p.Cout) * sizeof(T);
}

template <typename T, int WM, int WN, int SM, int SN>
static int launch_conv_cfg(const nunet_conv_desc* const* ds, int n, hipStream_t st) {
  typedef ConvCfg<T, WM, WN, SM, SN> C;
  ConvSetup S[CONV_GROUP_MAX];
  for (int k = 0; k < n; ++k) conv_setup<T, WM, WN, SM, SN>(ds[k], S[k]);
  const int cls = C::BN == 64 ? PC_CONV_M128N64 : C::BM == 256 ? PC_CONV_M256N32 : PC_CONV_M128N32;
  if (n == 1 && S[0].p.S > 1) {
    const ConvSetup& s0 = S[0];
    const ConvP& p = s0.p;
    const nunet_conv_desc* d = ds[0];
    ProfScope ps(cls, s0.flops, s0.bytes, st);
    ConvGroup g; g.n = 1; g.grid[0] = (int)s0.grid; g.p[0] = p;
    launch_conv_lt<T, WM, WN, SM, SN, true, false>(s0.lt, (unsigned)s0.grid, s0.dyn, st, g);
    SplitFinP f;
    f.slabs = p.slabs; f.slab_stride = p.slab_stride; f.S = p.S; f.bias = p.bias;
    f.dst0 = p.dst0; f.dst1 = p.dst1; f.D0 = p.D0; f.D1 = p.D1; f.Q0 = p.Q0; f.Q1 = p.Q1;
    f.slot_w = p.slot_w; f.acc0_mask = p.acc0_mask; f.acc1 = p.acc1; f.inv_slot_w = p.inv_slot_w; f.stats = p.stats;
    f.npix = (long long)d->N * d->H * d->W; f.Cout = p.Cout;
    f.bn_y = p.bn_y; f.bn_py = p.bn_py; f.bn_mi = p.bn_mi; f.bn_gamma = p.bn_gamma; f.bn_beta = p.bn_beta; f.bn_sums = p.bn_sums;
    const int G = p.Cout / C::EPV;
    const int blk = G * (256 / G);
    long long fg = (f.npix + (blk / G) - 1) / (blk / G);
    if (fg > 1024) fg = 1024;
    if (s0.bnr) NUNET_LAUNCH((splitk_finalize_kernel<T, true>), dim3((unsigned)fg), dim3(blk), 0, st, f);
    else NUNET_LAUNCH((splitk_finalize_kernel<T, false>), dim3((unsigned)fg), dim3(blk), 0, st, f);
    return nunet_check_launch("conv3x3 (K-split)");
  }
  // one launch for the n problems (the caller grouped only problems of one variant, none of them K-split): sorted by grid size
  // for the round-robin workgroup map, coefficient tables sized for the largest
  int order[CONV_GROUP_MAX];
  for (int k = 0; k < n; ++k) order[k] = k;
  for (int a = 0; a < n; ++a) for (int b = a + 1; b < n; ++b) if (S[order[b]].grid < S[order[a]].grid) { const int t = order[a]; order[a] = order[b]; order[b] = t; }
  ConvGroup g; g.n = n;
  long total = 0; size_t dyn = 0; double fl = 0, by = 0;
  for (int k = 0; k < n; ++k) {
    const ConvSetup& s = S[order[k]];
    g.grid[k] = (int)s.grid; g.p[k] = s.p; total += s.grid; if (s.dyn > dyn) dyn = s.dyn; fl += s.flops; by += s.bytes;
  }
  ProfScope ps(cls, fl, by, st);
  if (S[0].bnr) launch_conv_lt<T, WM, WN, SM, SN, false, true>(S[0].lt, (unsigned)total, dyn, st, g);
  else launch_conv_lt<T, WM, WN, SM, SN, false, false>(S[0].lt, (unsigned)total, dyn, st, g);
  return nunet_check_launch(n > 1 ? "conv3x3 (group)" : "conv3x3");
}

// Tile choice (measured per layer on MI355X, tools/conv_layers.py): the standard tiles are 128 pixels x 64 channels
// (Cout multiple of 64: 2 x 2 waves of 64 x 32) or 256 x 32. They leave a deep level (few pixels) with fewer work
// items than the chip has CUs; there the 128 x 32 tile (4 workgroups per CU, twice the items) wins by up to 2x and
// needs no K-split. It also wins for the plain / BN-forward Cout = 32 convs of the first level (one wave per SIMD
// with the 256-pixel tile). With the BN-backward input transform the small tile loses: every Cout tile repeats the
// transform of its input tile. Also measured and not kept: the same tiles on 8 waves (32 x 32 per wave: every layer
// slower, up to 1.35x), one workgroup per CU to leave room for another lane's kernel (-4 % on the step).
// (round 3, measured and not kept: 576 pixels x 32 channels on 6 waves of 96 x 32 - one balanced round of 256 items at level 0,
//  the weight stage shared by 4.5 x the pixels, every B fragment feeding three MFMAs - is 10-30 % SLOWER on every level-0 layer:
//  six waves land 2-2-1-1 on the four SIMDs, the nine staging units per thread push the kernel to 256 registers with spills, and
//  one workgroup per CU loses what co-resident workgroups still overlap; capping today's kernel at one workgroup per CU costs 20 %.)
// (round 3, measured and not kept: a double-buffered LDS loop for the 128 x 32 tile - the next chunk's staging (wait, transform,
//  16-byte LDS writes) and the loads of the chunk after it spread over the MFMA steps of the current sweep, ONE barrier per chunk,
//  2 x 38 KB - is bit-correct and 8.5 % SLOWER over the 59 launches (level-0 conv1 +17 %, level-2 conv2 +40 %): a chunk's time is
//  set by the latency of its 12 staging loads per thread against the bytes a CU has in flight (2-3 workgroups x 30 KB per ~2 us),
//  not by the write / barrier / sweep phases being serial; the second image halves the resident workgroups and buys no distance -
//  a load still has one sweep to arrive. Then built as well: an LDS-DMA loader for the plain 16-bit 128 x 32 convs - three or four
//  lane-linear 32 KB chunk images (64-byte rows, the bank swizzle on the SOURCE address: slot s of row r fetches channel segment
//  s ^ ((r >> 2) & 3), the fragment reads XOR the same), `global_load_lds_dwordx4` through inline asm (hipcc puts an s_waitcnt
//  vmcnt(0) in front of the first fragment read after every LDS-DMA it knows of), a counted vmcnt(8 / 16) and ONE raw s_barrier per
//  chunk, no staging registers, one workgroup per CU. Bit-correct (118 op tests) and 39 % SLOWER over the 15 first convs of the
//  blocks (314 -> 436 us), identically with 3 and with 4 images: not latency either. With one wave per SIMD nothing overlaps the
//  wave's own instruction stream - address generation of the 8 DMA instructions (~0.4 us), the sweep's 18 dependent MFMAs (~0.5 us),
//  the epilogue (2.5 us per item) run one after the other -, whereas the register-staged kernel's 2-3 co-resident workgroups fill
//  each other's gaps: 1.2 us per chunk and CU against 1.5-1.9. What this kernel needs is the staging work INSIDE the MFMA gaps of
//  the same wave and two accumulator chains per wave, not more loads in flight.)
// 0: 128 x 32, 1: 128 x 64, 2: 256 x 32
static int conv_cfg_of(const nunet_conv_desc* d) {
  const int cout = d->D0 + d->D1;
  if (d->tile >= 1 && d->tile <= 4) return d->tile - 1;
  const long px = (long)d->N * d->H * d->W;
  const long items_std = cout % 64 == 0 ? ceil_div64(px, 128) * (cout / 64) : ceil_div64(px, 256) * (cout / 32);
  // (256x256 bs32, tools/conv_layers.py with nunet_conv_desc.tile: with >= 1M pixels the 256-pixel tile wins for every Cout = 32
  //  layer - 8 or more items per resident workgroup, no round quantisation left, the weight stage shared by twice the pixels:
  //  level-0 conv1 -13 %, conv2 -3 %)
  //  (re-measured after the XCD remap, 96x96 bs16: the plain / BN-forward Cout = 64 convs of level 1 - 36864 pixels - are 8 % faster
  //   on the small tile as well: conv1 21.5 -> 19.7, conv2 14.1 -> 12.8 us; their input-gradient convs are not: 20.7 -> 32.6)
  const bool small = items_std < 256 || (cout == 32 && d->in_tf != NUNET_TF_BN_RELU_BWD && px < (1L << 20)) ||
                     (cout == 64 && d->in_tf != NUNET_TF_BN_RELU_BWD && px < (1L << 16));
  if (small) return 0;
  return cout % 64 == 0 ? 1 : 2;
}
template <typename T> static int launch_conv_n(const nunet_conv_desc* const* ds, int n, hipStream_t st) {
  const int cfg = conv_cfg_of(ds[0]);
  if (cfg == 0) return launch_conv_cfg<T, 4, 1, 1, 1>(ds, n, st);                            // 128 pixels x 32 channels
  if (cfg == 1) return launch_conv_cfg<T, 2, 2, 2, 1>(ds, n, st);
  if (cfg == 3) return launch_conv_cfg<T, 4, 1, 2, 2>(ds, n, st);                            // 256 x 64: 64 x 64 per wave, every fragment feeds two MFMAs
  return launch_conv_cfg<T, 4, 1, 2, 1>(ds, n, st);
}
template <typename T> static int launch_conv(const nunet_conv_desc* d, hipStream_t st) { return launch_conv_n<T>(&d, 1, st); }

// would this problem be K-split (then it is launched alone: its finalize launch follows it)?
template <typename T> static int conv_is_split(const nunet_conv_desc* d) {
  ConvSetup S;
  const int cfg = conv_cfg_of(d);
  if (cfg == 0) conv_setup<T, 4, 1, 1, 1>(d, S); else if (cfg == 1) conv_setup<T, 2, 2, 2, 1>(d, S); else if (cfg == 3) conv_setup<T, 4, 1, 2, 2>(d, S); else conv_setup<T, 4, 1, 2, 1>(d, S);
  return S.p.S > 1 ? 1 : 0;
}

static int conv_check(const nunet_conv_desc* d);
// Grouping key of a problem: problems with equal keys (>= 0) may share a launch. -1: launch it alone.
int nunet_conv_group_key(const nunet_conv_desc* d) {
  if (!d || conv_check(d) != NUNET_OK) return -1;
  if (NUNET_DISPATCH(d->dtype, conv_is_split, d)) return -1;
  return ((d->dtype * 4 + conv_cfg_of(d)) * 2 + (d->bn_y ? 1 : 0)) * 3 + d->in_tf;
}
// n problems of one key in one launch (internal: the plan's single-stream schedule)
int nunet_conv3x3_group(const nunet_conv_desc* const* ds, int n, hipStream_t st) {
  NUNET_REQUIRE(ds && n >= 1 && n <= CONV_GROUP_MAX, "conv3x3 group: 1..%d problems", CONV_GROUP_MAX);
  const int key = nunet_conv_group_key(ds[0]);
  for (int k = 0; k < n; ++k) { const int rc = conv_check(ds[k]); if (rc) return rc; }
  if (n > 1) for (int k = 0; k < n; ++k) NUNET_REQUIRE(key >= 0 && nunet_conv_group_key(ds[k]) == key, "conv3x3 group: problems of different kernel variants");
  return NUNET_DISPATCH(ds[0]->dtype, launch_conv_n, ds, n, st);
}

static int conv_check(const nunet_conv_desc* d) {
  NUNET_REQUIRE(d && d->src0 && d->wpack && d->dst0, "conv3x3: null pointer");
  const int cin = d->C0 + d->C1, cout = d->D0 + d->D1;
  NUNET_REQUIRE(d->dtype >= 0 && d->dtype <= 2, "conv3x3: bad dtype %d", d->dtype);
  const int kc = 64 / dtype_size(d->dtype);
  NUNET_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0, "conv3x3: bad extent %dx%dx%d", d->N, d->H, d->W);
  NUNET_REQUIRE(d->C0 > 0 && d->C0 % kc == 0 && d->C1 % kc == 0, "conv3x3: C0=%d C1=%d must be multiples of %d (64-byte channel chunks)", d->C0, d->C1, kc);
  NUNET_REQUIRE(d->C1 == 0 || d->src1, "conv3x3: src1 null with C1=%d", d->C1);
  NUNET_REQUIRE(cout % 32 == 0 && d->D0 % 32 == 0 && d->D1 % 32 == 0, "conv3x3: D0=%d D1=%d must be multiples of 32", d->D0, d->D1);
  NUNET_REQUIRE(d->D1 == 0 || d->dst1, "conv3x3: dst1 null with D1=%d", d->D1);
  NUNET_REQUIRE(d->P0 >= d->C0 && (d->C1 == 0 || d->P1 >= d->C1) && d->Q0 >= d->D0 && (d->D1 == 0 || d->Q1 >= d->D1), "conv3x3: pitch smaller than channels");
  const int epv = 16 / dtype_size(d->dtype);
  NUNET_REQUIRE(d->P0 % epv == 0 && (d->C1 == 0 || d->P1 % epv == 0), "conv3x3: source pitch must keep 16-byte alignment");
  NUNET_REQUIRE(d->acc_slot_w == 0 || d->acc_slot_w % 32 == 0, "conv3x3: acc_slot_w %d", d->acc_slot_w);
  {
    // staging uses 32-bit byte offsets from the tensor bases
    const unsigned long long px = (unsigned long long)d->N * d->H * d->W, es = dtype_size(d->dtype);
    const unsigned long long lim = 1ull << 32;
    NUNET_REQUIRE(px < (1ull << 30) && px * d->P0 * es < lim && (d->C1 == 0 || px * d->P1 * es < lim) && 9ull * cout * cin * es < lim &&
                  (d->in_tf != 2 || px * d->tf_py * es < lim), "conv3x3: problem too large (every input tensor must span < 4 GB)");
  }
  if (d->bn_y) {
    NUNET_REQUIRE(d->bn_mean_invstd && d->bn_gamma && d->bn_beta && d->bn_sums, "conv3x3: fused BN-backward reduce needs mean/invstd, gamma, beta and sums");
    NUNET_REQUIRE(d->D1 == 0 && d->Q0 == d->D0 && d->acc0_mask == 0 && d->bn_py % epv == 0 && cout <= 512,
                  "conv3x3: fused BN-backward reduce needs one dense, assign-only destination");
  }
  NUNET_REQUIRE(d->in_tf >= 0 && d->in_tf <= 2, "conv3x3: in_tf %d", d->in_tf);
  NUNET_REQUIRE(d->tile >= 0 && d->tile <= 4 && ((d->tile != 2 && d->tile != 4) || cout % 64 == 0), "conv3x3: tile %d with Cout=%d", d->tile, cout);
  if (d->in_tf) {
    NUNET_REQUIRE(d->C1 == 0 && cin <= 1024, "conv3x3: an input transform needs a single source (C1 == 0) of at most 1024 channels");
    NUNET_REQUIRE(d->tf_gamma && d->tf_beta, "conv3x3: input transform needs gamma and beta");
    NUNET_REQUIRE(!d->tf_store || d->tf_ps % epv == 0, "conv3x3: tf_store pitch alignment");
    if (d->in_tf == 1) {
      if (d->tf_training) NUNET_REQUIRE(d->tf_fx, "conv3x3: BN input transform in training mode needs the producing conv's sums");
      else NUNET_REQUIRE(d->tf_running_mean && d->tf_running_var, "conv3x3: BN input transform in eval mode needs running statistics");
    } else {
      NUNET_REQUIRE(d->tf_y && d->tf_fx && d->tf_mean_invstd && d->tf_py % epv == 0, "conv3x3: BN-backward input transform needs y, sums and saved mean/invstd");
    }
  }
  return NUNET_OK;
}
extern "C" int nunet_conv3x3_fwd(const nunet_conv_desc* d, nunet_stream_t s) {
  const int rc = conv_check(d);
  if (rc) return rc;
  return NUNET_DISPATCH(d->dtype, launch_conv, d, (hipStream_t)s);
}

// ---------------------------------------------------------------------------
// wgrad kernel
//
// Work item = (32 Cout) x (32 Cin) x 9 taps of dW, over a slice of the pixel tiles (K-split). Every
// slice writes its partial block to its own fp32 slab with plain stores; the slabs are summed in a
// fixed order afterwards (nunet_wgrad_reduce), so the gradient is bit-identical from run to run.
// ---------------------------------------------------------------------------
struct WgP {
  const void* src0; const void* src1;
  int C0, C1, P0, P1;
  const void* dy; int Cout, PY;
  float* dw; long long slab_stride;   // slab s of the K-split at dw + s * slab_stride, [9][Cout][Cin] each
  int N, H, W, Cin;
  int NI, TH, TW, tilesX, tilesY, tilesG;
  int nCoT, nCiT, ksplit, nMT;
  int SH; unsigned SHinv;
  unsigned invTX, invTY;   // fastdiv_inv(tilesX), fastdiv_inv(tilesY)
};

// A work item covers (32 * A Cout) x (32 * B Cin) x 9 taps. In LDS the two operand tiles are stored as A (B) PLANES of
// [pixels][32 channels] - each plane is exactly the 64-byte-row image the transposing reads were laid out for (conflict-free),
// so wider items need no new bank analysis: a fragment of sub-tile a / b is the old read at a plane offset.
template <typename T, int A_, int B_> struct WgCfg {
  static constexpr int A = A_, B = B_;
  static constexpr int NT = 192;                      // 3 waves: wave w owns taps 3w..3w+2
  static constexpr int BM = 128, HPMAX = 192;
  static constexpr int EPV = Tr<T>::EPV;
  static constexpr bool F32 = std::is_same<T, float>::value;
  static constexpr int SR = 32;                       // row stride of a plane (elements): 32 channels, no pad
  static constexpr int UPR = 32 / EPV;                // 16-byte units per plane row
  static constexpr int UPD = UPR * A, UPX = UPR * B;  // units per pixel of the dY / input tile
  static constexpr int NUD = (BM * UPD + NT - 1) / NT;      // dY units per thread
  static constexpr int NUA = (HPMAX * UPX + NT - 1) / NT;   // halo units per thread
  static constexpr int STAGE = (BM * A + HPMAX * B) * SR;   // elements per LDS stage
  static_assert(NT % UPD == 0 && NT % UPX == 0, "every staging unit of a thread has the same channel segment");
  static_assert(NUD <= 16 && NUA <= 32, "validity masks");
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

// Work item = (32 A Cout) x (32 B Cin) x 9 taps of dW over a slice of the 128-pixel tiles. The three
// waves of a workgroup split the TAPS (3 each) and all walk every pixel of the tile: no cross-wave
// reduction, 48 A B accumulator registers per lane. Tiles are double-buffered in LDS; the next tile's
// global loads are in flight (registers) while the current one multiplies. Compared with 32 x 32 items
// (A = B = 1) an item reads its dY tile once for 32 B input channels and its input tile once for 32 A
// output channels, and one fragment feeds A (or B) MFMAs.
template <typename T, bool ST, int A, int B>
__device__ __forceinline__ void wgrad_body(const WgP& p, int bid) {
  typedef WgCfg<T, A, B> C;
  constexpr int NT = C::NT, BM = C::BM, EPV = C::EPV, SR = C::SR, UPR = C::UPR, UPD = C::UPD, UPX = C::UPX;
  extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
  T* const s_stage = reinterpret_cast<T*>(s_dyn);                                   // [2][STAGE]
  int* const s_hidx = reinterpret_cast<int*>(s_dyn + 2 * C::STAGE * sizeof(T));     // [BM]
  int* const s_mxy = s_hidx + BM;                                                   // [BM]
  int* const s_hxy = s_mxy + BM;                                                    // [HPMAX]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;

  const int cit = bid % p.nCiT; bid /= p.nCiT;
  const int cot = bid % p.nCoT;
  const int split = bid / p.nCoT;
  const int co0 = cot * 32 * A, ci0 = cit * 32 * B;
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

  // Tile-invariant part of every staging unit's address, computed ONCE: in-kernel cycle stamps (round 2, the method of tools/kstamp_build.sh)
  // showed the per-tile address generation (seven map_pixel() calls with 64-bit pointer arithmetic per thread) at
  // 2300 cycles - more than the tile's MFMAs and fragment reads together (1300-2100). A unit is (image-in-tile ni,
  // row dy, column dx) relative to the tile origin: its pixel is gp0(tile) + uoff, its validity two or three compares,
  // its address a 32-bit byte offset from a uniform base (host-checked to fit). Units outside the image read pixel 0
  // and are zeroed on the way into LDS.
  //   code: (ni << 20) | ((dy + 1) << 10) | (dx + 1) with dy, dx >= -1;  -2: no such unit
  int dcode[C::NUD], acode[C::NUA], duoff[C::NUD], auoff[C::NUA];
#pragma unroll
  for (int k = 0; k < C::NUD; ++k) {
    const int u = tid + k * NT;
    const int cde = u < BM * UPD ? s_mxy[u / UPD] : -2;                 // (ni, ly, lx), -1: row past the tile
    dcode[k] = cde >= 0 ? cde + (1 << 10) + 1 : cde;
    duoff[k] = cde >= 0 ? ((cde >> 20) * p.H + ((cde >> 10) & 1023)) * p.W + (cde & 1023) : 0;
  }
#pragma unroll
  for (int k = 0; k < C::NUA; ++k) {
    const int u = tid + k * NT;
    const int cde = u < C::HPMAX * UPX ? s_hxy[u / UPX] : -2;           // (ni, hy, hx): halo coordinates = tile coordinates + 1
    acode[k] = cde;
    auoff[k] = cde >= 0 ? ((cde >> 20) * p.H + ((cde >> 10) & 1023) - 1) * p.W + (cde & 1023) - 1 : 0;
  }
  // the channel slices this item reads: one source per 16-byte unit of the forward input
  // (C0 is a multiple of 16, so a unit never straddles the two concat sources)
  const int segx = tid % UPX, segd = tid % UPD;   // same for all units of a thread (NT % UPX == 0, NT % UPD == 0)
  const int cch = ci0 + segx * EPV;
  const bool cvalid = cch < p.Cin;
  const int dch = co0 + segd * EPV;
  const bool dvalid = dch < p.Cout;
  const char* abase; unsigned apb, acb;      // uniform base, pitch in bytes, this thread's channel byte offset
  // (a thread whose channels lie past Cin - the ragged last input tile of a wide item - loads pixel 0 / channel 0 of source 0
  //  and writes zeros: never an address formed from a null second source)
  if (cch < p.C0 || !cvalid) { abase = (const char*)p.src0; apb = (unsigned)p.P0 * (unsigned)sizeof(T); acb = (unsigned)(cvalid ? cch : 0) * (unsigned)sizeof(T); }
  else { abase = (const char*)p.src1; apb = (unsigned)p.P1 * (unsigned)sizeof(T); acb = (unsigned)(cch - p.C0) * (unsigned)sizeof(T); }
  const char* const dbase = (const char*)p.dy;
  const unsigned dpb = (unsigned)p.PY * (unsigned)sizeof(T), dcb = (unsigned)(dvalid ? dch : 0) * (unsigned)sizeof(T);
  // LDS position of the thread's units inside a stage: plane (32-channel block) and 16-byte segment of the plane row
  const int dplane = segd / UPR, dseg = segd % UPR;
  const int xplane = segx / UPR, xseg = segx % UPR;

  Vec16<T> dreg[C::NUD], areg[C::NUA];
  unsigned vmd = 0u, vma = 0u;               // bit k: dY / input unit k valid (of the tile held in the registers)
  // pixel of a unit (code, uoff) of the tile at (n0, y0, x0); -1: outside the image
  auto unit_pixel = [&](int code, int uoff, int n0, int y0, int x0, int gp0) {
    const int ni = code >> 20, dy = ((code >> 10) & 1023) - 1, dx = (code & 1023) - 1;
    if constexpr (ST) {
      // stacked rows: virtual row v = y0 + dy of ONE image of N * SH rows; rows with v % SH == 0 separate the images
      const int v = y0 + dy, x = dx;
      const int vc = v < 0 ? 0 : v;
      const int n2 = (int)__umulhi((unsigned)vc, p.SHinv), yy = vc - n2 * p.SH - 1;
      const bool ok = (v >= 0) & (yy >= 0) & (n2 < p.N) & (x >= 0) & (x < p.W);
      return ok ? (n2 * p.H + yy) * p.W + x : -1;
    } else {
      const int y = y0 + dy, x = x0 + dx;
      const bool ok = (n0 + ni < p.N) & (y >= 0) & (y < p.H) & (x >= 0) & (x < p.W);
      return ok ? gp0 + uoff : -1;
    }
  };
  auto load_tile = [&](int mt) {
    const int q1 = fastdiv(mt, p.invTX), q2 = fastdiv(q1, p.invTY);
    const int x0 = (mt - q1 * p.tilesX) * p.TW;
    const int y0 = (q1 - q2 * p.tilesY) * p.TH;
    const int n0 = q2 * p.NI;
    const int gp0 = (n0 * p.H + y0) * p.W + x0;
    unsigned md = 0u, ma = 0u;
#pragma unroll
    for (int k = 0; k < C::NUD; ++k) {
      const int gp = (dcode[k] >= 0 && dvalid) ? unit_pixel(dcode[k], duoff[k], n0, y0, x0, gp0) : -1;
      md |= gp >= 0 ? (1u << k) : 0u;
      dreg[k].raw = *reinterpret_cast<const u32x4*>(dbase + ((unsigned)(gp < 0 ? 0 : gp) * dpb + dcb));
    }
#pragma unroll
    for (int k = 0; k < C::NUA; ++k) {
      const int gp = (acode[k] >= 0 && cvalid) ? unit_pixel(acode[k], auoff[k], n0, y0, x0, gp0) : -1;
      ma |= gp >= 0 ? (1u << k) : 0u;
      areg[k].raw = *reinterpret_cast<const u32x4*>(abase + ((unsigned)(gp < 0 ? 0 : gp) * apb + acb));
    }
    vmd = md; vma = ma;
  };
  auto write_tile = [&](int buf) {
    T* sd = s_stage + buf * C::STAGE;
    T* sa = sd + A * BM * SR;
#pragma unroll
    for (int k = 0; k < C::NUD; ++k)
      if ((k + 1) * NT <= BM * UPD || dcode[k] != -2) st16(&sd[(dplane * BM + (tid + k * NT) / UPD) * SR + dseg * EPV], (vmd >> k) & 1u ? dreg[k] : zero16<T>());
#pragma unroll
    for (int k = 0; k < C::NUA; ++k)
      if ((k + 1) * NT <= C::HPMAX * UPX || acode[k] != -2) st16(&sa[(xplane * C::HPMAX + (tid + k * NT) / UPX) * SR + xseg * EPV], (vma >> k) & 1u ? areg[k] : zero16<T>());
  };

  f32x16 acc[A][B][3];
#pragma unroll
  for (int a = 0; a < A; ++a)
#pragma unroll
    for (int b = 0; b < B; ++b)
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[a][b][t][i] = 0.f;

  // this wave's three taps: rows 3w..3w+2 of the 3x3 window = kernel row `wave`
  int toff[3];
#pragma unroll
  for (int t = 0; t < 3; ++t) toff[t] = ((wave - 1) * HW2 + (t - 1)) * SR;

  // tile-invariant LDS offsets of this lane's fragment rows (16-bit types): A rows m = kk*16 + 8h + q (+4), B = their halo
  // pixel shifted to this wave's kernel row, tap t at + t * SR
  const int fq = (lane >> 2) & 3, colo = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  const int aoff = (8 * h + fq) * SR + colo;
  int boff0[BM / 16], boff1[BM / 16];
#pragma unroll
  for (int kk = 0; kk < BM / 16; ++kk) {
    const int m0 = kk * 16 + 8 * h + fq;
    boff0[kk] = s_hidx[m0] * SR + colo + toff[0];
    boff1[kk] = s_hidx[m0 + 4] * SR + colo + toff[0];
  }

  int mt = split;
  if (mt < p.nMT) load_tile(mt);
  int buf = 0;
  if (mt < p.nMT) write_tile(0);
  __syncthreads();
  while (mt < p.nMT) {
    const int nmt = mt + p.ksplit;
    if (nmt < p.nMT) load_tile(nmt);   // in flight during the MFMAs below
    const T* sd = s_stage + buf * C::STAGE;
    const T* sa = sd + A * BM * SR;
    if constexpr (C::F32) {
#pragma unroll 2
      for (int kk = 0; kk < BM / 2; ++kk) {
        const int m = kk * 2 + h;
        const int hx = s_hidx[m] * SR + r;
        float av[A];
#pragma unroll
        for (int a = 0; a < A; ++a) av[a] = sd[(a * BM + m) * SR + r];
#pragma unroll
        for (int b = 0; b < B; ++b)
#pragma unroll
          for (int t = 0; t < 3; ++t) {
            const float xv = sa[b * C::HPMAX * SR + hx + toff[t]];
#pragma unroll
            for (int a = 0; a < A; ++a) acc[a][b][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a], xv, acc[a][b][t], 0, 0, 0);
          }
      }
    } else {
      // fragments of k-step kk + 1 are read while k-step kk multiplies (two named register sets); the halo offsets of a
      // lane's rows are tile-invariant and were hoisted out of the tile loop (boff): the k-step used to open with two
      // LDS index reads and a full lgkmcnt(0) round trip before it could even form its addresses
      typedef typename Frag16<T>::V FV;
      s16x4 fa0[2][A], fa1[2][A], fx0[2][B][3], fx1[2][B][3];
      auto rd = [&](int kk, int sl) {
#pragma unroll
        for (int a = 0; a < A; ++a) {
          fa0[sl][a] = tr_read(&sd[a * BM * SR + aoff + kk * 16 * SR]);
          fa1[sl][a] = tr_read(&sd[a * BM * SR + aoff + (kk * 16 + 4) * SR]);
        }
#pragma unroll
        for (int b = 0; b < B; ++b)
#pragma unroll
          for (int t = 0; t < 3; ++t) {
            fx0[sl][b][t] = tr_read(&sa[b * C::HPMAX * SR + boff0[kk] + t * SR]);
            fx1[sl][b][t] = tr_read(&sa[b * C::HPMAX * SR + boff1[kk] + t * SR]);
          }
      };
      rd(0, 0);
#pragma unroll
      for (int kk = 0; kk < BM / 16; ++kk) {
        const int cu = kk & 1;
        if (kk + 1 < BM / 16) rd(kk + 1, cu ^ 1);
        s16x8 av[A];
#pragma unroll
        for (int a = 0; a < A; ++a) av[a] = __builtin_shufflevector(fa0[cu][a], fa1[cu][a], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
        for (int b = 0; b < B; ++b)
#pragma unroll
          for (int t = 0; t < 3; ++t) {
            const s16x8 bv = __builtin_shufflevector(fx0[cu][b][t], fx1[cu][b][t], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
            for (int a = 0; a < A; ++a) Mma<T>::mma(acc[a][b][t], __builtin_bit_cast(FV, av[a]), __builtin_bit_cast(FV, bv));
          }
      }
    }
    if (nmt < p.nMT) write_tile(buf ^ 1);   // the other stage was last read one iteration ago
    __syncthreads();
    buf ^= 1;
    mt = nmt;
  }

  // this slice's partial gradient block goes to ITS slab with plain stores (no atomics: 256 workgroups adding
  // 9216 floats each to the same 36.8 KB block serialise memory-side, and the sum would depend on arrival order);
  // nunet_wgrad_reduce / the plan's reduce launch sums the slabs in fixed order
  float* const slab = p.dw + (size_t)split * p.slab_stride;
#pragma unroll
  for (int b = 0; b < B; ++b) {
    const int ci = ci0 + 32 * b + r;
    if (ci < p.Cin) {
#pragma unroll
      for (int a = 0; a < A; ++a)
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          const int tap = wave * 3 + t;
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int co = co0 + 32 * a + acc_row(i, h);
            if (co < p.Cout) slab[((size_t)tap * p.Cout + co) * p.Cin + ci] = acc[a][b][t][i];
          }
        }
    }
  }
}

template <typename T, bool ST, int A, int B>
__global__ __launch_bounds__(192) void wgrad_kernel(WgP p) { wgrad_body<T, ST, A, B>(p, xcd_remap((int)blockIdx.x, (int)gridDim.x)); }

// Two independent weight-gradient problems in ONE launch (the two convolutions of a VGGBlock finish
// their dY at the same point of the backward pass): one kernel boundary less per block and twice
// the workgroups to fill the chip.
template <typename T, bool ST, int A1, int B1, int A2, int B2>
__global__ __launch_bounds__(192) void wgrad_pair_kernel(WgP pa, WgP pb, int na, int na_pad) {
  // (problem a's share of the grid is padded to a multiple of 8 - at most 7 idle blocks - so that block -> XCD is (local block) % 8
  //  in both problems and each gets its own xcd_remap)
  if ((int)blockIdx.x < na_pad) {
    const int v = xcd_remap((int)blockIdx.x, na_pad);
    if (v < na) wgrad_body<T, ST, A1, B1>(pa, v);
  } else wgrad_body<T, ST, A2, B2>(pb, xcd_remap((int)blockIdx.x - na_pad, (int)gridDim.x - na_pad));
}

// Item shape of a problem, a pure function of its extents. The kernel is bound by the latency of its staging loads,
// i.e. by how many workgroups are resident (measured, tools/wgrad_layers.py: the time of a launch is inversely
// proportional to its workgroup count until registers / LDS cap the residency). 64 output channels per item (2 x 1: the
// input tile, the bigger of the two, is read once per 64 output channels; 256 registers, two workgroups per CU) win
// 10-15 % on the levels with enough pixel tiles to keep the K-split wide; 64 input channels per item (1 x 2, for the
// Cout = 32 level) need 260 registers - one workgroup per CU - and lose 40 %: kept as an instantiation for diagnostics only.
struct WgTile { int A, B; };
// (inside the multi-lane step the 2 x 1 items LOSE what they win alone - 8160 vs 8330 img/s at 96x96 bs16: at two workgroups per
//  CU they leave the chain's kernels less room - so the plan asks for 32 x 32 items everywhere; the wider shapes are descriptor
//  options - nunet_wgrad_desc.item_shape - covered by the op tests and tools/wgrad_layers.py)
static WgTile wgrad_tile(const nunet_wgrad_desc* d) {
  const int cout = d->Cout, cin = d->C0 + d->C1;
  if (d->item_shape == 21 && cout % 64 == 0) return WgTile{2, 1};
  if (d->item_shape == 12 && cin >= 64) return WgTile{1, 2};
  return WgTile{1, 1};
}
template <typename T> static size_t wgrad_lds_bytes(WgTile t) {
  return ((size_t)2 * (128 * t.A + 192 * t.B) * 32) * sizeof(T) + (size_t)(2 * 128 + 192) * sizeof(int);
}

// K-split slices of a weight-gradient problem: enough work items for `target` workgroups, at most one
// slice per 128-pixel tile and at most `max_slabs` (the caller's slab capacity). Pure function of the descriptor.
static int wgrad_slices(const nunet_wgrad_desc* d, const TileGeom& g) {
  const WgTile wt = wgrad_tile(d);
  const int otiles = ceil_div(d->Cout, 32 * wt.A) * ceil_div(d->C0 + d->C1, 32 * wt.B);
  const int nMT = g.tilesX * g.tilesY * g.tilesG;
  int ks = ceil_div(d->target_wgs > 0 ? d->target_wgs : 256, otiles);
  if (ks > nMT) ks = nMT;
  if (d->max_slabs > 0 && ks > d->max_slabs) ks = d->max_slabs;
  if (ks < 1) ks = 1;
  return ks;
}
extern "C" int32_t nunet_conv3x3_wgrad_slabs(const nunet_wgrad_desc* d) {
  if (!d || d->N <= 0 || d->H <= 0 || d->W <= 0 || d->Cout <= 0 || d->C0 <= 0) return 0;
  const TileGeom g = nunet_choose_tile(d->N, d->H, d->W, 128, 192);
  return wgrad_slices(d, g);
}

template <typename T> static long wgrad_setup(const nunet_wgrad_desc* d, WgP& p) {
  p.src0 = d->src0; p.src1 = d->src1; p.C0 = d->C0; p.C1 = d->C1; p.P0 = d->P0; p.P1 = d->P1;
  p.dy = d->dy; p.Cout = d->Cout; p.PY = d->PY; p.dw = d->dw;
  p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->C0 + d->C1;
  p.slab_stride = d->slab_stride > 0 ? d->slab_stride : 9LL * p.Cout * p.Cin;
  const TileGeom g = nunet_choose_tile(d->N, d->H, d->W, 128, 192);     // WgCfg::BM, HPMAX
  p.NI = g.NI; p.TH = g.TH; p.TW = g.TW; p.tilesX = g.tilesX; p.tilesY = g.tilesY; p.tilesG = g.tilesG; p.SH = g.SH;
  p.SHinv = g.SH ? (unsigned)(((1ull << 32) + g.SH - 1) / g.SH) : 0u;
  p.invTX = fastdiv_inv(g.tilesX); p.invTY = fastdiv_inv(g.tilesY);
  const WgTile wt = wgrad_tile(d);
  p.nCoT = ceil_div(p.Cout, 32 * wt.A);
  p.nCiT = ceil_div(p.Cin, 32 * wt.B);
  p.nMT = g.tilesX * g.tilesY * g.tilesG;
  p.ksplit = wgrad_slices(d, g);
  return (long)p.nCoT * p.nCiT * p.ksplit;
}
template <typename T> static void wgrad_prof(const nunet_wgrad_desc* d, const WgP& p, double& flops, double& bytes) {
  const double px = (double)d->N * d->H * d->W;
  const int acin = g_prof_alg_cin > 0 ? g_prof_alg_cin : p.Cin;
  flops = 2.0 * 9 * acin * p.Cout * px;
  bytes = px * (acin + p.Cout) * sizeof(T) + 9.0 * acin * p.Cout * 4;
}

// (dynamic LDS: the wide items exceed the 64 KB static limit; the attribute is set once per instantiation)
template <typename K> static void wgrad_allow_lds(K kernel, size_t bytes) {
  (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}
template <typename T, bool ST, int A, int B> static void launch_wgrad_one(long grid, const WgP& p, hipStream_t st) {
  const size_t lds = wgrad_lds_bytes<T>(WgTile{A, B});
  static bool once = false;
  if (!once) { wgrad_allow_lds(wgrad_kernel<T, ST, A, B>, lds); once = true; }
  NUNET_LAUNCH((wgrad_kernel<T, ST, A, B>), dim3((unsigned)grid), dim3(192), lds, st, p);
}
template <typename T, bool ST, int A1, int B1, int A2, int B2> static void launch_wgrad_two(long ga, long gb, const WgP& pa, const WgP& pb, hipStream_t st) {
  const size_t l1 = wgrad_lds_bytes<T>(WgTile{A1, B1}), l2 = wgrad_lds_bytes<T>(WgTile{A2, B2});
  const size_t lds = l1 > l2 ? l1 : l2;
  static bool once = false;
  if (!once) { wgrad_allow_lds(wgrad_pair_kernel<T, ST, A1, B1, A2, B2>, lds); once = true; }
  const long ga_pad = (ga + 7) / 8 * 8;
  NUNET_LAUNCH((wgrad_pair_kernel<T, ST, A1, B1, A2, B2>), dim3((unsigned)(ga_pad + gb)), dim3(192), lds, st, pa, pb, (int)ga, (int)ga_pad);
}

template <typename T> static int launch_wgrad(const nunet_wgrad_desc* d, hipStream_t st) {
  WgP p;
  const long grid = wgrad_setup<T>(d, p);
  double fl, by; wgrad_prof<T>(d, p, fl, by);
  ProfScope ps(p.Cout == 32 ? PC_WGRAD_1x4 : PC_WGRAD_2x2, fl, by, st);
  const WgTile wt = wgrad_tile(d);
  const int key = wt.A * 10 + wt.B;
  if (p.SH) {
    if (key == 21) launch_wgrad_one<T, true, 2, 1>(grid, p, st);
    else if (key == 12) launch_wgrad_one<T, true, 1, 2>(grid, p, st);
    else launch_wgrad_one<T, true, 1, 1>(grid, p, st);
  } else {
    if (key == 21) launch_wgrad_one<T, false, 2, 1>(grid, p, st);
    else if (key == 12) launch_wgrad_one<T, false, 1, 2>(grid, p, st);
    else launch_wgrad_one<T, false, 1, 1>(grid, p, st);
  }
  return nunet_check_launch("wgrad3x3");
}

struct WgPairArgs { const nunet_wgrad_desc* a; const nunet_wgrad_desc* b; };
template <typename T> static int launch_wgrad_pair(const WgPairArgs* w, hipStream_t st) {
  WgP pa, pb;
  const long ga = wgrad_setup<T>(w->a, pa), gb = wgrad_setup<T>(w->b, pb);
  double fa, ba, fb, bb; wgrad_prof<T>(w->a, pa, fa, ba);
  { const int keep = g_prof_alg_cin; g_prof_alg_cin = 0; wgrad_prof<T>(w->b, pb, fb, bb); g_prof_alg_cin = keep; }
  const WgTile ta = wgrad_tile(w->a), tb = wgrad_tile(w->b);
  const int key = (ta.A * 10 + ta.B) * 100 + tb.A * 10 + tb.B;
  // the item-shape pairs a VGGBlock produces (conv1 | conv2): 1x1 | 1x1, 2x1 | 2x1, and the mixed forms of the diagnostic rules;
  // anything else (different tiling modes, other shape pairs): two launches
  const bool fused = (pa.SH != 0) == (pb.SH != 0) && (key == 1111 || key == 1211 || key == 2121 || key == 2111 || key == 1121);
  if (!fused) {
    int rc = launch_wgrad<T>(w->a, st);
    if (rc) return rc;
    const int keep = g_prof_alg_cin; g_prof_alg_cin = 0;
    rc = launch_wgrad<T>(w->b, st);
    g_prof_alg_cin = keep;
    return rc;
  }
  ProfScope ps(pa.Cout == 32 ? PC_WGRAD_1x4 : PC_WGRAD_2x2, fa + fb, ba + bb, st);
  if (pa.SH) {
    if (key == 1111) launch_wgrad_two<T, true, 1, 1, 1, 1>(ga, gb, pa, pb, st);
    else if (key == 1211) launch_wgrad_two<T, true, 1, 2, 1, 1>(ga, gb, pa, pb, st);
    else if (key == 2111) launch_wgrad_two<T, true, 2, 1, 1, 1>(ga, gb, pa, pb, st);
    else if (key == 1121) launch_wgrad_two<T, true, 1, 1, 2, 1>(ga, gb, pa, pb, st);
    else launch_wgrad_two<T, true, 2, 1, 2, 1>(ga, gb, pa, pb, st);
  } else {
    if (key == 1111) launch_wgrad_two<T, false, 1, 1, 1, 1>(ga, gb, pa, pb, st);
    else if (key == 1211) launch_wgrad_two<T, false, 1, 2, 1, 1>(ga, gb, pa, pb, st);
    else if (key == 2111) launch_wgrad_two<T, false, 2, 1, 1, 1>(ga, gb, pa, pb, st);
    else if (key == 1121) launch_wgrad_two<T, false, 1, 1, 2, 1>(ga, gb, pa, pb, st);
    else launch_wgrad_two<T, false, 2, 1, 2, 1>(ga, gb, pa, pb, st);
  }
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
  {
    const unsigned long long px = (unsigned long long)d->N * d->H * d->W, es = dtype_size(d->dtype), lim = 1ull << 32;   // 32-bit staging offsets
    NUNET_REQUIRE(px * d->P0 * es < lim && (d->C1 == 0 || px * d->P1 * es < lim) && px * d->PY * es < lim, "wgrad: every input tensor must span < 4 GB");
  }
  NUNET_REQUIRE(d->slab_stride == 0 || d->slab_stride >= 9LL * d->Cout * (d->C0 + d->C1), "wgrad: slab_stride smaller than one slab");
  NUNET_REQUIRE(d->max_slabs >= 0 && d->target_wgs >= 0, "wgrad: max_slabs / target_wgs");
  NUNET_REQUIRE(d->item_shape == 0 || d->item_shape == 11 || d->item_shape == 12 || d->item_shape == 21, "wgrad: item_shape %d (0 | 11 | 12 | 21)", d->item_shape);
  {
    // every slab the launch will write must fit what the caller says `dw` holds
    const long long stride = d->slab_stride > 0 ? d->slab_stride : 9LL * d->Cout * (d->C0 + d->C1);
    const long long need = (long long)(nunet_conv3x3_wgrad_slabs(d) - 1) * stride + 9LL * d->Cout * (d->C0 + d->C1);
    NUNET_REQUIRE(d->dw_floats >= need, "wgrad: dw holds %lld floats, the %d slabs of this launch need %lld", (long long)d->dw_floats, (int)nunet_conv3x3_wgrad_slabs(d), need);
  }
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

// Sum of the K-split slabs of a weight gradient (fixed order: bit-reproducible): out[i] (+)= sum_s slabs[s * stride + i].
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slabs, long long stride, int ns, float* __restrict__ out, long long n4, int accumulate) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < ns; ++s) { const f32x4 v = *reinterpret_cast<const f32x4*>(slabs + (size_t)s * stride + i * 4); a[0] += v[0]; a[1] += v[1]; a[2] += v[2]; a[3] += v[3]; }
    f32x4* q = reinterpret_cast<f32x4*>(out + i * 4);
    if (accumulate) { const f32x4 o = *q; a[0] += o[0]; a[1] += o[1]; a[2] += o[2]; a[3] += o[3]; }
    *q = a;
  }
}
extern "C" int nunet_wgrad_reduce(const float* slabs, int64_t slab_stride, int32_t nslabs, int64_t n, float* out, int32_t accumulate, nunet_stream_t s) {
  NUNET_REQUIRE(slabs && out && nslabs >= 1 && n > 0 && n % 4 == 0 && slab_stride % 4 == 0, "wgrad_reduce: bad args (n and slab_stride multiples of 4)");
  NUNET_REQUIRE(((uintptr_t)slabs & 15) == 0 && ((uintptr_t)out & 15) == 0, "wgrad_reduce: 16-byte alignment");
  long long g = (n / 4 + 255) / 256; if (g > 4096) g = 4096;
  NUNET_LAUNCH(wgrad_reduce_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)s, slabs, (long long)slab_stride, nslabs, out, (long long)(n / 4), accumulate);
  return nunet_check_launch("wgrad_reduce");
}
