// gemm_dma.h — the fp32 MFMA GEMMs of the hot shapes, staged by LDS-DMA (buffer_load ... lds).
//
// One tile body, three operand combinations (KC = k-contiguous: the reduction index is the fastest-moving index in
// memory; RM = reduction-major: the reduction index is the row index in memory):
//   NT  A KC, B KC :  Y[M,N]  = act(X)[M,K] * W[N,K]^T + b          torch.nn.Linear forward
//   NN  A KC, B RM :  dX[M,K] = dY[M,N] * W[N,K]                     its input gradient
//   TN  A RM, B RM :  dW[N,K] = dY[M,N]^T * act(X)[M,K] (+ db)       its weight gradient, split over M into slabs
//
// Call sites replaced: every torch.nn.Linear / GINEConv.lin of /root/reference/run_graphcount.py:54-121,183-189
// (zinc_models.py:513-576, ogb_mol_gnn.py:330-345) whose reduction length is a multiple of 32 — the H-wide layers,
// >99 % of a training step's flops.  Ragged shapes (in_dim 10, ...) stay on linear_mfma.hip.
//
// Why a second GEMM family: these shapes are SHORT-K (K = 256: 8 K-steps per tile), so prologue, per-step issue
// overhead and epilogue are first-order costs (measured on MI355X for a 128x128x256 tile of the r01 kernel design:
// K loop 37 k cycles for 32.8 k cycles of MFMA, epilogue 17.7 k cycles).  Here
//   * tiles land in LDS by DMA (one 1-KiB piece per wave-instruction, no VGPR staging, no ds_write); row bounds come
//     from the buffer descriptor (out-of-range rows read as zero): no clamps, no masks, no per-step VALU;
//   * optional LOADER waves issue the DMA, so a compute wave's stream is barrier / ds_read / MFMA only (a piece costs
//     its issuing wave 60-180 cycles; 8 per K-step is 10-35 % of the step's 4096 MFMA cycles): K loop 34.3 k cycles;
//   * KC images are unpadded and XOR-swizzled on the SOURCE address (the DMA writes lane-linear): conflict-free
//     ds_read_b128 fragments; RM images are read with ds_read_b32 (consecutive lanes, consecutive banks);
//   * STAGES-deep ring with counted vmcnt and ONE raw s_barrier per K-step;
//   * the output tile is staged through LDS (dead by then) and leaves as whole rows, 16 bytes per lane (16 dword
//     stores per 32x32 block straight from the accumulators cost 17.7 k cycles, staged: 4.5 k, HBM-bound);
//   * v_mfma_f32_32x32x2_f32: exact fp32 fma chain (no TF32 on gfx950); lane half h owns k = 8c+4h+t of every
//     8-chunk, the same permutation for both operands.
#pragma once
#include "common.h"

namespace esc {
namespace dma {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
using lds_void = __attribute__((address_space(3))) void;

// raw buffer descriptor (stride 0): accesses at byte offsets >= `bytes` read as zero / are dropped
__device__ __forceinline__ i32x4 make_rsrc(const void* p, unsigned bytes) {
  const uint64_t a = reinterpret_cast<uint64_t>(p);
  i32x4 r;
  r[0] = (int)(uint32_t)a; r[1] = (int)(uint32_t)((a >> 32) & 0xffffu); r[2] = (int)bytes; r[3] = 0x00020000;
  return r;
}
__device__ __forceinline__ unsigned lds_addr(const float* p) {
  return (unsigned)(uintptr_t)(lds_void*)const_cast<float*>(p);
}
// one 1-KiB piece: lane i moves 16 bytes from (base + voff + soff) to LDS byte (dst + 16 i); dst wave-uniform.
// (voff is bounds-checked against the descriptor, soff is not.)
// Issued from inline asm on purpose: hipcc would otherwise drain vmcnt(0) before every LDS read that follows a DMA it can
// see, i.e. once per K-step, and the ring below would never overlap a transfer with the MFMAs.  The statement is
// therefore absent from the compiler's vmcnt bookkeeping: completion is counted by hand (wait_vmcnt + s_barrier).
// M0 is compiler-reserved: saved and restored inside the statement; s_nop 4 covers an SGPR operand freshly written by SALU.
__device__ __forceinline__ void dma16(i32x4 r, unsigned dst, int voff, int soff) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(r), "s"(dst), "s"(soff)
      : "memory");
}
template <int N> __device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// swizzle of the 16-byte chunk index inside a k-contiguous LDS row: the 16 lanes of a ds_read_b128 group read 16
// different rows at one logical chunk; with 128-byte rows (BK 32) two rows share a 256-byte bank line, so the key is
// (row >> 1) & 7; with 256-byte rows (BK 64) it is row & 15.
template <int BK> __device__ __forceinline__ int swz(int row) {
  static_assert(BK == 32 || BK == 64, "BK is 32 or 64");
  return BK == 32 ? ((row >> 1) & 7) : (row & 15);
}

struct GArgs {
  const float* A; int lda;           // KC: [M, R]   RM: [R, M]
  const float* B; int ldb;           // KC: [N, R]   RM: [R, N]
  float* C; int ldc;                 // [M, N]; with splits > 1: slabs [split][M][N] (ldc = N)
  const float* bias;                 // [N] or null
  const float* pro_scale;            // PRO 1: [R] on A (KC)   PRO 2: [N] on B (RM):  x := relu(x * scale + shift)
                                     // PRO 3: as 1, the coefficients merged from `fold` by every workgroup itself
  const float* pro_shift;
  float2* col_stats;                 // [ceil(M/BM)][N] (mean, M2) of the outputs per ROW TILE, or null (STATS)
  BnFoldDev fold;                    // PRO 3: the BatchNorm in front of A, still as partials (merged in the prologue)
  float* db_part;                    // DB: per-split sums over the reduction of A's columns, [split][M]
  int M, N, R;                       // output rows, output cols, reduction length
  int red_per_split;                 // multiple of BK; splits = ceil(R / red_per_split)
  int accumulate;                    // C += ...
  const float* c_init; int ld_init;  // forward tiles: start the accumulators from this [M, N] partial result (see gemm_body)
  int c_vec;                         // set by the launcher: 16-byte row stores are legal
  int ntile_m, ntile_n;              // set by the launcher
  unsigned long long* stamps;        // diagnostics (tools/gemm_lab): per workgroup clock readings, or null
  BnbDev bnb;                        // PRO bit 2
  BnStatDev bst;                     // PRO bit 3
};

__device__ __forceinline__ void stamp(unsigned long long* base, int slot) {
  if (base != nullptr) {                                  // wave-uniform
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t = __builtin_readcyclecounter();
    const unsigned long long r = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { base[blockIdx.x * 12 + slot] = t; base[blockIdx.x * 12 + 6 + slot] = r; }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// tile sequence number of a workgroup: workgroups b and b+8 share an XCD (dealt round-robin), so the sequence is cut
// into 8 contiguous runs — consecutive tiles (same row panel of A) meet in one L2.  Bijective for any nwg.
__device__ __forceinline__ int xcd_tile_id(int b, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = b & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}

// DMA piece geometry of one operand tile.  KC tile [T][BK]: 1-KiB pieces of 1024 / (4 BK) whole rows.  RM tile [BK][T]: a piece is
// as many WHOLE rows of 4 T bytes as fit 1 KiB (T = 256, 128, 64, 32: the piece is full; T = 160 (r03: 300 / 600-wide layers):
// one row of 640 bytes, 40 of the 64 lanes active — the rest are masked off, their LDS slots belong to the next piece).
template <int T, int BK, bool RM> struct Pieces {
  static constexpr int RB = RM ? T * 4 : BK * 4;              // bytes of a tile row in LDS
  static constexpr int RPP = 1024 / RB;                       // rows per piece
  static constexpr int BYTES = RPP * RB;                      // LDS bytes a piece covers
  static constexpr int LANES = BYTES / 16;                    // active lanes
  static constexpr int COUNT = (RM ? BK : T) / RPP;           // pieces per tile
  static_assert(RB <= 1024 && RB % 16 == 0 && (RM ? BK : T) % RPP == 0, "tile rows must fit a DMA piece and tile it evenly");
};

template <int BM, int BN, int BK, int WM, int WN, int STAGES, int LW, bool A_RM, bool B_RM, int PRO, bool STATS, bool DB>
struct Cfg {
  static constexpr int P = PRO & 3;                                     // operand prologue (see GArgs)
  static constexpr bool BNB = (PRO & 4) != 0, BSTAT = (PRO & 8) != 0;
  static constexpr int NW = WM * WN, NWT = NW + LW, NTHR = NWT * 64, DW = LW > 0 ? LW : NW;
  static constexpr int A_FLOATS = BM * BK, B_FLOATS = BN * BK, A2_FLOATS = BNB ? A_FLOATS : 0;   // A2: the BatchNorm input rows beside dOut
  static constexpr int STAGE_FLOATS = A_FLOATS + A2_FLOATS + B_FLOATS;
  using PcA = Pieces<BM, BK, A_RM>; using PcB = Pieces<BN, BK, B_RM>;
  static constexpr int PA = PcA::COUNT, PB = PcB::COUNT;                // DMA pieces per tile
  static constexpr int PPWA = PA / DW, PPWB = PB / DW, PPW = PPWA * (BNB ? 2 : 1) + PPWB;
  static constexpr int TM = BM / WM, TN = BN / WN, MT = TM / 32, NT = TN / 32;
  static constexpr int OUT_LD = BN + 4, OUT_FLOATS = BM * OUT_LD;       // epilogue staging image [BM][BN+4]
  static constexpr int STAT_FLOATS = STATS ? WM * BN * 3 : 0;           // (n, mean, M2) per wave row and column
  static constexpr int RING_FLOATS = STAGES * STAGE_FLOATS > OUT_FLOATS + STAT_FLOATS ? STAGES * STAGE_FLOATS : OUT_FLOATS + STAT_FLOATS;
  static constexpr int PRO_MAXK = 1280, BNB_MAXK = 640;
  static constexpr bool PRO_A = P == 1 || P == 3;                       // per-k affine + ReLU on a KC A operand
  static constexpr bool BNB_KC = BNB && !A_RM;                          // KC A: mean | scale | mask scale | mask shift | k1 | invstd k2 per k,
                                                                        // 6 arrays of the (padded) reduction length BEHIND the static part: bnb_lds_floats(R)
  static constexpr int BST_FLOATS = BSTAT ? NTHR * 8 : 0;               // epilogue: one (s1, s2) float4 pair per thread, inside the dead ring
  static constexpr int PRO_FLOATS = PRO_A ? 2 * PRO_MAXK : 0;           // scale | shift of the whole reduction range
  static constexpr size_t LDS_BYTES = (size_t)(RING_FLOATS + PRO_FLOATS) * 4;
  static constexpr int bnb_lds_floats(int R) { return BNB_KC ? 6 * (((R + BK - 1) / BK) * BK) : 0; }
  static_assert(OUT_FLOATS + STAT_FLOATS + BST_FLOATS <= RING_FLOATS, "the epilogue's images must fit the ring they reuse");
  static constexpr int C4 = BN / 4;                                     // float4 per output row of the tile
  static_assert(PA % DW == 0 && PB % DW == 0, "pieces must split evenly over the DMA waves");
  static_assert(MT >= 1 && NT >= 1 && STAGES >= 2 && STAGES <= 4, "bad tile");
  static_assert(NTHR >= C4, "a thread keeps one column quad through the epilogue");
  static_assert(P == 0 || (PRO_A && !A_RM) || (P == 2 && B_RM), "prologue: per-k on a KC A, or per-column on an RM B");
  static_assert(!(BNB && PRO_A), "the BatchNorm backward and the affine prologue both transform A");
  static_assert(!BSTAT || (!A_RM && B_RM && !STATS), "BatchNorm-backward partials ride on the input-gradient tiles");
  static_assert(!DB || A_RM, "bias gradient = column sums of a reduction-major A");
  static_assert(LDS_BYTES <= 160 * 1024, "tile does not fit the 160 KiB LDS");
};

// ---- per-lane DMA source offsets of one operand tile -------------------------------------------------------------
// KC tile [T][BK]: piece p holds rows p RPP .. ; lane: row = p RPP + l / CPR, physical chunk l % CPR holding logical
//   chunk (l % CPR) ^ swz(row).  Offset = (o0 + row) ld 4 + chunk 16; the K-step goes through soff = kt BK 4.
// RM tile [BK][T]: piece p holds reduction rows p RPP .. ; offset = (red0 + row) ld 4 + (o0 + col) 4; the K-step is ADDED
//   to the offsets (it must be bounds-checked: rows past the split's end read as zero).
// kc[j] (KC operands): k of this lane's 16-byte chunk inside a K-step — a reduction length that is not a multiple of BK
// ends in a partial K-step whose chunks at k >= R are fetched with an out-of-range offset (they read as zero) instead of
// running on into the next row.
template <int T, int BK, int DW, int NP, bool RM>
__device__ __forceinline__ void dma_offsets(int (&vo)[NP], int (&kc)[NP], int dw, int l, int o0, int red0, int ld) {
  if constexpr (!RM) {
    constexpr int RB = BK * 4, CPR = RB / 16, RPP = 1024 / RB;
    const int prow = l / CPR, pc = l % CPR;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int row = (dw + DW * j) * RPP + prow;
      kc[j] = (pc ^ swz<BK>(row)) << 2;
      vo[j] = ((o0 + row) * ld) * 4 + (kc[j] << 2);
    }
  } else {
    constexpr int RB = T * 4, CPR = RB / 16, RPP = 1024 / RB;
    static_assert(RB <= 1024, "RM tile rows longer than one DMA piece are not supported");
    const int prow = l / CPR, pc = l % CPR;                  // (lanes >= RPP * CPR are masked off by the issuer)
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int row = (dw + DW * j) * RPP + prow;
      kc[j] = 0;
      vo[j] = ((red0 + row) * ld + o0) * 4 + (pc << 4);
    }
  }
}

template <int BM, int BN, int BK, int WM, int WN, int STAGES, int LW, bool A_RM, bool B_RM, int PRO, bool STATS, bool DB, bool INIT = false>
__device__ __forceinline__ void gemm_body(const GArgs& g, float* __restrict__ lds, int wg) {
  using C_ = Cfg<BM, BN, BK, WM, WN, STAGES, LW, A_RM, B_RM, PRO, STATS, DB>;
  constexpr int NW = C_::NW, DW = C_::DW;
  constexpr int P = C_::P;
  constexpr bool BNB = C_::BNB, BSTAT = C_::BSTAT;
  constexpr int PPWA = C_::PPWA, PPWB = C_::PPWB, PPW = C_::PPW;
  constexpr int TM = C_::TM, TN = C_::TN, MT = C_::MT, NT = C_::NT;
  constexpr int RB = BK * 4;

  const int per = g.ntile_m * g.ntile_n;
  const int split = wg / per;
  const int tile = xcd_tile_id(wg % per, per);
  const int m0 = (tile / g.ntile_n) * BM, n0 = (tile % g.ntile_n) * BN;
  const int red0 = split * g.red_per_split;
  const int red1 = min(g.R, red0 + g.red_per_split);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int l = lane_id(), h = l >> 5, lr = l & 31;
  const bool loader = LW > 0 && wave >= NW;                 // wave-uniform
  const bool issuing = LW == 0 || loader;
  const int dw = LW > 0 ? wave - NW : wave;                 // index among the DMA-issuing waves
  const int wm = (wave % NW) / WN, wn = wave % WN;

  const i32x4 ra = A_RM ? make_rsrc(g.A, (unsigned)(((size_t)(red1 - 1) * g.lda + g.M) * 4))
                        : make_rsrc(g.A, (unsigned)(((size_t)(g.M - 1) * g.lda + g.R) * 4));
  const i32x4 rb = B_RM ? make_rsrc(g.B, (unsigned)(((size_t)(red1 - 1) * g.ldb + g.N) * 4))
                        : make_rsrc(g.B, (unsigned)(((size_t)(g.N - 1) * g.ldb + g.R) * 4));
  int voa[PPWA], vob[PPWB], kca[PPWA], kcb[PPWB];
  dma_offsets<BM, BK, DW, PPWA, A_RM>(voa, kca, dw, l, m0, red0, g.lda);
  dma_offsets<BN, BK, DW, PPWB, B_RM>(vob, kcb, dw, l, n0, red0, g.ldb);
  // BNB: the BatchNorm's input rows travel beside dOut — same tile, own base / leading dimension, image right behind A's.
  // (Fetching them from global memory into registers in the fragment layout instead, one K-step ahead, keeps the ring at
  // two images and three stages but measured SLOWER than this: the fragment-shaped loads are 32-byte pieces of 32 rows.)
  i32x4 ra2 = ra;
  int voa2[BNB ? PPWA : 1], kca2[BNB ? PPWA : 1];
  if constexpr (BNB) {
    ra2 = A_RM ? make_rsrc(g.bnb.x, (unsigned)(((size_t)(red1 - 1) * g.bnb.ldx + g.M) * 4))
               : make_rsrc(g.bnb.x, (unsigned)(((size_t)(g.M - 1) * g.bnb.ldx + g.R) * 4));
    dma_offsets<BM, BK, DW, PPWA, A_RM>(voa2, kca2, dw, l, m0, red0, g.bnb.ldx);
  }

  constexpr int OOB = 0x7FFFFFF0;                           // beyond every descriptor's num_records: the load returns 0
  const int stepa = A_RM ? BK * g.lda * 4 : 0, stepb = B_RM ? BK * g.ldb * 4 : 0;   // RM: K-step inside the checked offset
  const int stepa2 = (BNB && A_RM) ? BK * g.bnb.ldx * 4 : 0;

  const unsigned lds_base = lds_addr(lds);
  int issued = 0;                                           // K-steps issued so far by this wave
  auto stage = [&](int buf) {
    const unsigned a_dst0 = lds_base + (unsigned)(buf * C_::STAGE_FLOATS) * 4u;
    const unsigned a_dst = a_dst0 + (unsigned)dw * (unsigned)C_::PcA::BYTES;
    const unsigned b_dst = a_dst0 + (C_::A_FLOATS + C_::A2_FLOATS) * 4u + (unsigned)dw * (unsigned)C_::PcB::BYTES;
    const int k0 = red0 + issued * BK;                      // first reduction index of this K-step
    const int soff = k0 * 4;                                // KC operands: K offset (an SGPR offset is not range-checked)
    const bool tail = k0 + BK > red1;                       // wave-uniform: the partial last K-step of a KC operand
#pragma unroll
    for (int j = 0; j < PPWA; ++j) {
      int vo = voa[j];
      if constexpr (!A_RM) { if (tail && k0 + kca[j] >= red1) vo = OOB; }
      if (C_::PcA::LANES == 64 || l < C_::PcA::LANES) dma16(ra, a_dst + (unsigned)(DW * j) * (unsigned)C_::PcA::BYTES, vo, A_RM ? 0 : soff);
      if constexpr (A_RM) voa[j] += stepa;
    }
    if constexpr (BNB) {
#pragma unroll
      for (int j = 0; j < PPWA; ++j) {
        int vo = voa2[j];
        if constexpr (!A_RM) { if (tail && k0 + kca2[j] >= red1) vo = OOB; }
        if (C_::PcA::LANES == 64 || l < C_::PcA::LANES) dma16(ra2, a_dst + C_::A_FLOATS * 4u + (unsigned)(DW * j) * (unsigned)C_::PcA::BYTES, vo, A_RM ? 0 : soff);
        if constexpr (A_RM) voa2[j] += stepa2;
      }
    }
#pragma unroll
    for (int j = 0; j < PPWB; ++j) {
      int vo = vob[j];
      if constexpr (!B_RM) { if (tail && k0 + kcb[j] >= red1) vo = OOB; }
      if (C_::PcB::LANES == 64 || l < C_::PcB::LANES) dma16(rb, b_dst + (unsigned)(DW * j) * (unsigned)C_::PcB::BYTES, vo, B_RM ? 0 : soff);
      if constexpr (B_RM) vob[j] += stepb;
    }
    ++issued;
  };
  // tile kt of this wave's pieces has landed once at most `ahead` younger groups are still in flight
  auto wait_tile = [&](int ahead) {
    if (ahead <= 0) wait_vmcnt<0>();
    else if (ahead == 1) wait_vmcnt<PPW>();
    else wait_vmcnt<(STAGES >= 4 ? 2 : 1) * PPW>();
  };

  stamp(g.stamps, 0);
  const int nk = (red1 - red0 + BK - 1) / BK;
  if (issuing) {
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
      if (s < nk) stage(s);
  }

  float* pro = lds + C_::RING_FLOATS;     // PRO 1 / 3: [scale R | shift R]
  if constexpr (P == 3) {                 // merge the producer's BatchNorm partials here (see common.h); tile 0 keeps the results
    for (int k = threadIdx.x; k < g.R; k += C_::NTHR) {
      float sc, sh;
      bn_fold_column(g.fold, k, wg == 0, sc, sh);
      pro[k] = sc; pro[C_::PRO_MAXK + k] = sh;
    }
    for (int k = g.R + threadIdx.x; k < ((g.R + BK - 1) / BK) * BK; k += C_::NTHR) { pro[k] = 0.f; pro[C_::PRO_MAXK + k] = 0.f; }   // partial last K-step: relu(0*0+0) = 0
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  }
  if constexpr (P == 1) {
    // with loader waves only the COMPUTE waves stage the coefficients: a loader's vmcnt(0) here would drain the ring's
    // prologue transfers it has just issued before it may run ahead
    const int kfirst = LW > 0 ? (loader ? 1 << 30 : (int)threadIdx.x) : (int)threadIdx.x;
    constexpr int kstride = LW > 0 ? NW * 64 : C_::NTHR;
    const int rpad1 = ((g.R + BK - 1) / BK) * BK;
    for (int k = kfirst; k < rpad1; k += kstride) {
      const bool in = k < g.R;                                            // partial last K-step: relu(0*0+0) = 0
      pro[k] = in ? g.pro_scale[k] : 0.f; pro[C_::PRO_MAXK + k] = in ? g.pro_shift[k] : 0.f;
    }
    // the raw s_barrier of the first K-step publishes these writes: they must have LANDED before this wave arrives there
    if (LW == 0 || !loader) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  }
  // BNB on a KC A (input-gradient tiles): six per-k arrays over the whole reduction range; k >= R: scale 0 -> operand 0
  float* bnbc = lds + C_::RING_FLOATS + C_::PRO_FLOATS;
  const bool bnb_elu = BNB && g.bnb.relu == 2;               // (wave-uniform) ELU behind the fused BatchNorm backward
  const int bnb_s = ((g.R + BK - 1) / BK) * BK;              // stride of the six coefficient arrays
  if constexpr (BNB && !A_RM) {
    const int S = bnb_s;
    const int rpad = bnb_s;
    // with loader waves the compute waves stage the coefficients: their vmcnt holds no DMA, so the wait below does not drain
    // the ring's prologue transfers
    const int kstart = LW > 0 ? (loader ? rpad : (int)threadIdx.x) : (int)threadIdx.x;
    for (int k = kstart; k < rpad; k += (LW > 0 ? NW * 64 : C_::NTHR)) {
      const bool in = k < g.R;
      const float sc = in ? g.bnb.scale[k] : 0.f;
      const float2 kk = in ? g.bnb.coef[k] : make_float2(0.f, 0.f);
      bnbc[k] = in ? g.bnb.mean[k] : 0.f;
      bnbc[S + k] = sc;
      bnbc[2 * S + k] = g.bnb.relu ? sc : 0.f;                           // mask: fmaf(x, msc, msh) > 0 (no activation: always)
      bnbc[3 * S + k] = g.bnb.relu ? (in ? g.bnb.shift[k] : 0.f) : 1.f;
      bnbc[4 * S + k] = kk.x;
      bnbc[5 * S + k] = in ? g.bnb.invstd[k] * kk.y : 0.f;
    }
    if (LW == 0 || !loader) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // published by the first K-step's raw s_barrier
  }
  // BNB on an RM A (weight-gradient tiles): the channel is the lane's column of every 32-column block
  float q_mu[BNB && A_RM ? MT : 1], q_a[BNB && A_RM ? MT : 1], q_ms[BNB && A_RM ? MT : 1], q_mh[BNB && A_RM ? MT : 1],
        q_k1[BNB && A_RM ? MT : 1], q_k2[BNB && A_RM ? MT : 1];
  if constexpr (BNB && A_RM) {
    auto load = [&](int col, float& mu, float& a, float& ms, float& mh, float& k1, float& k2) {
      const bool in = col < g.M;
      const float sc = in ? g.bnb.scale[col] : 0.f;
      const float2 kk = in ? g.bnb.coef[col] : make_float2(0.f, 0.f);
      mu = in ? g.bnb.mean[col] : 0.f; a = sc;
      ms = g.bnb.relu ? sc : 0.f; mh = g.bnb.relu ? (in ? g.bnb.shift[col] : 0.f) : 1.f;
      k1 = kk.x; k2 = in ? g.bnb.invstd[col] * kk.y : 0.f;
    };
#pragma unroll
    for (int i = 0; i < MT; ++i) load(m0 + wm * TM + i * 32 + lr, q_mu[i], q_a[i], q_ms[i], q_mh[i], q_k1[i], q_k2[i]);
  }
  float cs[NT], ch_[NT];                  // PRO 2: this lane's column coefficients (column = lane & 31 of each block)
  if constexpr (P == 2) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int col = n0 + wn * TN + j * 32 + lr;
      cs[j] = col < g.N ? g.pro_scale[col] : 0.f;
      ch_[j] = col < g.N ? g.pro_shift[col] : 0.f;
    }
  }

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // c_init: the tile starts from a partial result computed elsewhere (an earlier slice of the reduction: the readout Linear's columns
  // of the layers that were already final) instead of from zero — statistics and bias then see the complete sums.  C/D map of the 32x32
  // block: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 h.  (Its own instantiation: with the loads in the common kernels the
  // compiler keeps the accumulators in 60-100 more VGPRs — 135 -> 194 for the edge-row tile — and every forward GEMM pays.)
  if constexpr (INIT) if (g.c_init != nullptr && !loader) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int col = n0 + wn * TN + j * 32 + lr;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + wm * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (row < g.M && col < g.N) acc[i][j][r] = g.c_init[(size_t)row * g.ld_init + col];
        }
      }
  }
  float dbsum = 0.f;
  float dbfrag[(BNB && A_RM && DB) ? MT : 1] = {};          // BNB: per-lane column sums of the transformed A fragments

  // fragment addressing.  KC: row (lane & 31) of a 32-row block, logical chunk 2 c8 + h -> physical (2 c8) ^ y.
  // RM: reduction row 8 c8 + 4 h + t, column (lane & 31) of a 32-column block.
  const int y16 = (h ^ swz<BK>(lr)) << 4;
  const int a_base = A_RM ? (h * 4 * BM + wm * TM + lr) * 4 : (wm * TM + lr) * RB;
  const int b_base = B_RM ? (h * 4 * BN + wn * TN + lr) * 4 : (wn * TN + lr) * RB;

  auto compute = [&](int buf, int kt) {
    const char* a_l = reinterpret_cast<const char*>(lds + buf * C_::STAGE_FLOATS);
    const char* a2_l = a_l + C_::A_FLOATS * 4;
    const char* b_l = a_l + (C_::A_FLOATS + C_::A2_FLOATS) * 4;
    const int kstep0 = red0 + kt * BK;                                   // first reduction index of this K-step
    if constexpr (DB && !BNB) {     // column sums of the reduction-major A tile (bias gradient): first column tile only
      if (n0 == 0 && (int)threadIdx.x < BM) {
        const float* col = reinterpret_cast<const float*>(a_l) + threadIdx.x;
#pragma unroll 8
        for (int kk = 0; kk < BK; ++kk) dbsum += col[kk * BM];
      }
    }
    // (DB && BNB: the transformed fragments are summed where they are made — see transform)
    float4 af[2][MT], bf[2][NT], s4[2], h4[2];
    float4 xf[2][BNB ? MT : 1], b6[2][(BNB && !A_RM) ? 6 : 1];          // BNB: the BatchNorm input fragments; KC A: per-k coefficients
    int cq[2] = {0, 0};                                                  // the 8-chunk each fragment slot holds
    auto frags = [&](int c8, int q) {
      const int ch = (c8 << 5) ^ y16;
      cq[q] = c8;
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        if constexpr (!A_RM) {
          af[q][i] = *reinterpret_cast<const float4*>(a_l + a_base + i * 32 * RB + ch);
          if constexpr (BNB) xf[q][i] = *reinterpret_cast<const float4*>(a2_l + a_base + i * 32 * RB + ch);
        } else {
          const float* p = reinterpret_cast<const float*>(a_l + a_base + (c8 * 8 * BM + i * 32) * 4);
          af[q][i] = make_float4(p[0], p[BM], p[2 * BM], p[3 * BM]);
          if constexpr (BNB) {
            const float* px = reinterpret_cast<const float*>(a2_l + a_base + (c8 * 8 * BM + i * 32) * 4);
            xf[q][i] = make_float4(px[0], px[BM], px[2 * BM], px[3 * BM]);
          }
        }
      }
      if constexpr (BNB && !A_RM) {
        const int k = kstep0 + c8 * 8 + h * 4;
#pragma unroll
        for (int u = 0; u < 6; ++u) b6[q][u] = *reinterpret_cast<const float4*>(bnbc + u * bnb_s + k);
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        if constexpr (!B_RM) {
          bf[q][j] = *reinterpret_cast<const float4*>(b_l + b_base + j * 32 * RB + ch);
        } else {
          const float* p = reinterpret_cast<const float*>(b_l + b_base + (c8 * 8 * BN + j * 32) * 4);
          bf[q][j] = make_float4(p[0], p[BN], p[2 * BN], p[3 * BN]);
        }
      }
      if constexpr (C_::PRO_A) {
        const int k = red0 + kt * BK + c8 * 8 + h * 4;
        s4[q] = *reinterpret_cast<const float4*>(pro + k);
        h4[q] = *reinterpret_cast<const float4*>(pro + C_::PRO_MAXK + k);
      }
    };
    constexpr int NREADS = (A_RM ? 4 : 1) * MT * (BNB ? 2 : 1) + (B_RM ? 4 : 1) * NT + (C_::PRO_A ? 2 : 0) + ((BNB && !A_RM) ? 6 : 0);
    constexpr int NVALU = (C_::PRO_A ? 8 * MT : (P == 2 ? 8 * NT : 0)) + (BNB ? 28 * MT : 0);
    constexpr int NMFMA = 4 * MT * NT;
    auto transform = [&](int q) {         // consumer-side BatchNorm + ReLU on the staged operand
      if constexpr (C_::PRO_A) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          float4& v = af[q][i];
          v.x = fmaxf(fmaf(v.x, s4[q].x, h4[q].x), 0.f); v.y = fmaxf(fmaf(v.y, s4[q].y, h4[q].y), 0.f);
          v.z = fmaxf(fmaf(v.z, s4[q].z, h4[q].z), 0.f); v.w = fmaxf(fmaf(v.w, s4[q].w, h4[q].w), 0.f);
        }
      }
      if constexpr (BNB && !A_RM) {       // BatchNorm backward on the staged gradient: per-k coefficients
#define ESC_BNB_KC(FN)                                                                                                         \
        _Pragma("unroll") for (int i = 0; i < MT; ++i) {                                                                       \
          float4& v = af[q][i];                                                                                                \
          const float4 x = xf[q][i];                                                                                           \
          v.x = FN(v.x, x.x, b6[q][0].x, b6[q][1].x, b6[q][2].x, b6[q][3].x, b6[q][4].x, b6[q][5].x);                           \
          v.y = FN(v.y, x.y, b6[q][0].y, b6[q][1].y, b6[q][2].y, b6[q][3].y, b6[q][4].y, b6[q][5].y);                           \
          v.z = FN(v.z, x.z, b6[q][0].z, b6[q][1].z, b6[q][2].z, b6[q][3].z, b6[q][4].z, b6[q][5].z);                           \
          v.w = FN(v.w, x.w, b6[q][0].w, b6[q][1].w, b6[q][2].w, b6[q][3].w, b6[q][4].w, b6[q][5].w);                           \
        }
        if (bnb_elu) { ESC_BNB_KC(bnb_apply_elu) } else { ESC_BNB_KC(bnb_apply) }       // (wave-uniform: one scalar branch per fragment set)
#undef ESC_BNB_KC
      }
      if constexpr (BNB && A_RM) {        // ... per-column coefficients; reduction rows past the split's end contribute nothing
        const int r0 = kstep0 + cq[q] * 8 + h * 4;
#define ESC_BNB_RM(FN)                                                                                                         \
        _Pragma("unroll") for (int i = 0; i < MT; ++i) {                                                                       \
          float4& v = af[q][i];                                                                                                \
          const float4 x = xf[q][i];                                                                                           \
          v.x = r0 + 0 < red1 ? FN(v.x, x.x, q_mu[i], q_a[i], q_ms[i], q_mh[i], q_k1[i], q_k2[i]) : 0.f;                         \
          v.y = r0 + 1 < red1 ? FN(v.y, x.y, q_mu[i], q_a[i], q_ms[i], q_mh[i], q_k1[i], q_k2[i]) : 0.f;                         \
          v.z = r0 + 2 < red1 ? FN(v.z, x.z, q_mu[i], q_a[i], q_ms[i], q_mh[i], q_k1[i], q_k2[i]) : 0.f;                         \
          v.w = r0 + 3 < red1 ? FN(v.w, x.w, q_mu[i], q_a[i], q_ms[i], q_mh[i], q_k1[i], q_k2[i]) : 0.f;                         \
          if constexpr (DB) dbfrag[i] += (v.x + v.y) + (v.z + v.w);                                                            \
        }
        if (bnb_elu) { ESC_BNB_RM(bnb_apply_elu) } else { ESC_BNB_RM(bnb_apply) }
#undef ESC_BNB_RM
      }
      if constexpr (P == 2) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          float4& v = bf[q][j];
          v.x = fmaxf(fmaf(v.x, cs[j], ch_[j]), 0.f); v.y = fmaxf(fmaf(v.y, cs[j], ch_[j]), 0.f);
          v.z = fmaxf(fmaf(v.z, cs[j], ch_[j]), 0.f); v.w = fmaxf(fmaf(v.w, cs[j], ch_[j]), 0.f);
        }
      }
    };
    frags(0, 0);
    transform(0);
    __builtin_amdgcn_sched_group_barrier(0x100, NREADS, 0);
    if constexpr (NVALU > 0) __builtin_amdgcn_sched_group_barrier(0x002, NVALU, 0);
#pragma unroll
    for (int c8 = 0; c8 < BK / 8; ++c8) {
      const int cur = c8 & 1;
      const bool more = c8 + 1 < BK / 8;
      if (more) frags(c8 + 1, cur ^ 1);     // the next chunk's fragments fly under this chunk's MFMAs ...
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i].x, bf[cur][j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i].y, bf[cur][j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i].z, bf[cur][j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i].w, bf[cur][j].w, acc[i][j], 0, 0, 0);
        }
      if (more) transform(cur ^ 1);         // ... and are transformed in the shadow of its second half
      // pin the order (hipcc otherwise sinks the reads below the MFMAs to save registers): reads, MFMAs, transform, MFMAs
      if (more) __builtin_amdgcn_sched_group_barrier(0x100, NREADS, 0);
      if constexpr (NVALU > 0) {
        __builtin_amdgcn_sched_group_barrier(0x008, NMFMA / 2, 0);
        if (more) __builtin_amdgcn_sched_group_barrier(0x002, NVALU, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, NMFMA - NMFMA / 2, 0);
      } else {
        __builtin_amdgcn_sched_group_barrier(0x008, NMFMA, 0);
      }
    }
  };

  // ring: at the top of step kt the DMA groups of tiles kt .. kt+STAGES-2 are in flight; the issuing waves wait for
  // the oldest, everyone meets (all pieces of tile kt have landed AND everyone is done reading tile kt-1), the buffer
  // tile kt-1 lived in is refilled with tile kt+STAGES-1, tile kt is computed.
  if (loader) {
    for (int kt = 0; kt < nk; ++kt) {
      wait_tile(min(STAGES - 2, nk - 1 - kt));
      __builtin_amdgcn_s_barrier();
      if (kt + STAGES - 1 < nk) stage((kt + STAGES - 1) % STAGES);
    }
  } else {
    for (int kt0 = 0; kt0 < nk; kt0 += STAGES) {
#pragma unroll
      for (int s = 0; s < STAGES; ++s) {
        const int kt = kt0 + s;
        if (kt < nk) {
          if constexpr (LW == 0) wait_tile(min(STAGES - 2, nk - 1 - kt));
          __builtin_amdgcn_s_barrier();
          __builtin_amdgcn_sched_barrier(0);
          if (kt == 0) stamp(g.stamps, 1);
          if constexpr (LW == 0) {
            if (kt + STAGES - 1 < nk) stage((s + STAGES - 1) % STAGES);
          }
          compute(s, kt);
        }
      }
    }
  }
  stamp(g.stamps, 2);

  // ---- BatchNorm partials of the outputs (bias included): one (mean, M2) per column and ROW TILE.  Each wave merges
  // its MT 32-row blocks (registers); the WM wave rows meet through LDS between the two epilogue barriers.
  float st_n[NT], st_mean[NT], st_m2[NT];
  if constexpr (STATS) {
    if (g.col_stats != nullptr && !loader) {
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int col = n0 + wn * TN + j * 32 + lr;
        const float bv = (g.bias && col < g.N) ? g.bias[col] : 0.f;
        st_n[j] = 0.f; st_mean[j] = 0.f; st_m2[j] = 0.f;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const int row0 = m0 + wm * TM + i * 32;
          const int nvalid = max(0, min(32, g.M - row0));
          float mean, m2 = 0.f;
          if (nvalid >= 32) {             // interior block (wave-uniform): no row masks
            float s1 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) s1 += acc[i][j][r] + bv;
            s1 += __shfl_xor(s1, 32, 64);
            mean = s1 * (1.f / 32.f);
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float d = acc[i][j][r] + bv - mean; m2 = fmaf(d, d, m2); }
          } else {
            float s1 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * h;
              if (row < g.M) s1 += acc[i][j][r] + bv;
            }
            s1 += __shfl_xor(s1, 32, 64);
            mean = nvalid > 0 ? s1 / (float)nvalid : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * h;
              if (row < g.M) { const float d = acc[i][j][r] + bv - mean; m2 = fmaf(d, d, m2); }
            }
          }
          m2 += __shfl_xor(m2, 32, 64);
          if (nvalid > 0) {               // Chan merge (wave-uniform branch)
            const float nb = (float)nvalid, tot = st_n[j] + nb, delta = mean - st_mean[j];
            st_mean[j] += delta * (nb / tot);
            st_m2[j] += m2 + delta * delta * (st_n[j] * nb / tot);
            st_n[j] = tot;
          }
        }
      }
    }
  }
  if constexpr (DB && !BNB) {
    if (n0 == 0 && (int)threadIdx.x < BM && m0 + (int)threadIdx.x < g.M)
      g.db_part[(size_t)split * g.M + m0 + threadIdx.x] = dbsum;
  }
  if constexpr (DB && BNB) {      // the two lane halves hold the k = 4h.. shares of the same column; wave column 0 of every wave row writes
    if (n0 == 0 && !loader && wn == 0) {
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const float tsum = dbfrag[i] + __shfl_xor(dbfrag[i], 32, 64);
        const int col = m0 + wm * TM + i * 32 + lr;
        if (h == 0 && col < g.M) g.db_part[(size_t)split * g.M + col] = tsum;
      }
    }
  }
  // ---- epilogue: the tile is staged through LDS (C/D map: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 h) and
  // leaves as whole rows, 16 bytes per lane
  __syncthreads();                                           // everyone is done reading the last K tile
  float* stat = lds + C_::OUT_FLOATS;                        // [WM][BN][3]
  if (!loader) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          lds[(wm * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * C_::OUT_LD + wn * TN + j * 32 + lr] = acc[i][j][r];
    if constexpr (STATS && WM > 1) {
      if (g.col_stats != nullptr && wm > 0 && l < 32) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          float* q = stat + (wm * BN + wn * TN + j * 32 + lr) * 3;
          q[0] = st_n[j]; q[1] = st_mean[j]; q[2] = st_m2[j];
        }
      }
    }
  }
  __syncthreads();
  if constexpr (STATS) {
    if (g.col_stats != nullptr && !loader && wm == 0 && l < 32) {
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int col = n0 + wn * TN + j * 32 + lr;
        float n = st_n[j], mean = st_mean[j], m2 = st_m2[j];
#pragma unroll
        for (int q = 1; q < WM; ++q) {                       // wave rows in order: a fixed association
          const float* p = stat + (q * BN + wn * TN + j * 32 + lr) * 3;
          const float nb = p[0];
          if (nb > 0.f) {
            const float tot = n + nb, delta = p[1] - mean;
            mean += delta * (nb / tot);
            m2 += p[2] + delta * delta * (n * nb / tot);
            n = tot;
          }
        }
        if (col < g.N && n > 0.f) g.col_stats[(size_t)(m0 / BM) * g.N + col] = make_float2(mean, m2);
      }
    }
  }
  {
    constexpr int C4 = C_::C4, RPI = C_::NTHR / C4, ITERS = (BM + RPI - 1) / RPI;      // rows per pass, passes
    const int c4 = threadIdx.x % C4, rr = threadIdx.x / C4;
    const bool epi = rr < RPI;                               // (NTHR need not be a multiple of the tile's column quads: 160-wide tiles)
    const int col = epi ? n0 + c4 * 4 : g.N;                 // an idle thread owns no column
    float* Cb = g.C + (size_t)split * g.M * g.ldc;           // split > 0 only for slab outputs
    // BSTAT: C is the gradient of a BatchNorm(+ReLU) output; every thread sums (g, g*xhat) of its column quad over the rows
    // it stores, the workgroup adds the per-thread sums in row order and writes ONE partial per tile row block and column
    float4 t1 = make_float4(0.f, 0.f, 0.f, 0.f), t2 = t1, e_mu = t1, e_is = t1, e_ms = t1, e_mh = make_float4(1.f, 1.f, 1.f, 1.f);
    bool bstat = false;
    float4 ypre[BSTAT ? ITERS : 1];                          // the BatchNorm input rows of this thread's quads, fetched up front
    if constexpr (BSTAT) {
      bstat = g.bst.partial != nullptr && g.c_vec;           // workgroup-uniform
      if (bstat && col < g.N) {
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
          const int row = m0 + rr + it * RPI;
          ypre[it] = (row < g.M && (BM % RPI == 0 || rr + it * RPI < BM)) ? *reinterpret_cast<const float4*>(g.bst.x + (size_t)row * g.bst.ldx + col)
                                                                          : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        e_mu = *reinterpret_cast<const float4*>(g.bst.mean + col);
        e_is = *reinterpret_cast<const float4*>(g.bst.invstd + col);
        if (g.bst.relu) { e_ms = *reinterpret_cast<const float4*>(g.bst.scale + col); e_mh = *reinterpret_cast<const float4*>(g.bst.shift + col); }
      }
    }
    if (g.c_vec) {
      if (col < g.N) {                                       // N % 4 == 0: whole quads only
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (g.bias) bv = *reinterpret_cast<const float4*>(g.bias + col);
        float* cp = Cb + (size_t)(m0 + rr) * g.ldc + col;
        const size_t step = (size_t)RPI * g.ldc;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
          const int row = m0 + rr + it * RPI;
          if (row < g.M && (BM % RPI == 0 || rr + it * RPI < BM)) {
            float4 v = *reinterpret_cast<const float4*>(lds + (rr + it * RPI) * C_::OUT_LD + c4 * 4);
            v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
            float4* dst = reinterpret_cast<float4*>(cp + it * step);
            if (g.accumulate) { const float4 o = *dst; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
            *dst = v;
            if constexpr (BSTAT) {
              if (bstat) {
                const float4 y = ypre[it];
                const float px = fmaf(y.x, e_ms.x, e_mh.x), py = fmaf(y.y, e_ms.y, e_mh.y), pz = fmaf(y.z, e_ms.z, e_mh.z), pw = fmaf(y.w, e_ms.w, e_mh.w);
                float gx, gy, gz, gw;
                if (g.bst.relu == 2) {                       // (workgroup-uniform) ELU: d act / d v = v > 0 ? 1 : exp(v)
                  gx = px > 0.f ? v.x : v.x * expf(px); gy = py > 0.f ? v.y : v.y * expf(py);
                  gz = pz > 0.f ? v.z : v.z * expf(pz); gw = pw > 0.f ? v.w : v.w * expf(pw);
                } else {
                  gx = px > 0.f ? v.x : 0.f; gy = py > 0.f ? v.y : 0.f; gz = pz > 0.f ? v.z : 0.f; gw = pw > 0.f ? v.w : 0.f;
                }
                t1.x += gx; t1.y += gy; t1.z += gz; t1.w += gw;
                t2.x = fmaf(gx, (y.x - e_mu.x) * e_is.x, t2.x); t2.y = fmaf(gy, (y.y - e_mu.y) * e_is.y, t2.y);
                t2.z = fmaf(gz, (y.z - e_mu.z) * e_is.z, t2.z); t2.w = fmaf(gw, (y.w - e_mu.w) * e_is.w, t2.w);
              }
            }
          }
        }
      }
      if constexpr (BSTAT) {
        if (bstat) {
          float4* red = reinterpret_cast<float4*>(lds + C_::OUT_FLOATS + C_::STAT_FLOATS);   // (behind the output image, inside the dead ring)
          red[threadIdx.x * 2] = t1; red[threadIdx.x * 2 + 1] = t2;
          __syncthreads();
          if (rr == 0 && col < g.N) {
#pragma unroll 4
            for (int r = 1; r < RPI; ++r) {
              const float4 u1 = red[(r * C4 + c4) * 2], u2 = red[(r * C4 + c4) * 2 + 1];
              t1.x += u1.x; t1.y += u1.y; t1.z += u1.z; t1.w += u1.w;
              t2.x += u2.x; t2.y += u2.y; t2.z += u2.z; t2.w += u2.w;
            }
            float2* dstp = g.bst.partial + (size_t)(m0 / BM) * g.N + col;
            *reinterpret_cast<float4*>(dstp) = make_float4(t1.x, t2.x, t1.y, t2.y);
            *reinterpret_cast<float4*>(dstp + 2) = make_float4(t1.z, t2.z, t1.w, t2.w);
          }
        }
      }
    } else {
#pragma unroll 1
      for (int it = 0; it < ITERS; ++it) {
        const int row = m0 + rr + it * RPI;
        if (row >= g.M || rr + it * RPI >= BM) break;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (col + q < g.N) {
            float v = lds[(rr + it * RPI) * C_::OUT_LD + c4 * 4 + q] + (g.bias ? g.bias[col + q] : 0.f);
            float* dst = Cb + (size_t)row * g.ldc + col + q;
            if (g.accumulate) v += *dst;
            *dst = v;
          }
        }
      }
    }
  }
  if (g.stamps != nullptr) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
  stamp(g.stamps, 3);
}

template <int BM, int BN, int BK, int WM, int WN, int STAGES, int LW, bool A_RM, bool B_RM, int PRO, bool STATS, bool DB, bool INIT = false>
__global__ __launch_bounds__((WM * WN + LW) * 64) void gemm_kernel(GArgs g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  if constexpr (BM * BN < 128 * 128) ESC_PRIO();            // node-sized tiles: see common.h
  gemm_body<BM, BN, BK, WM, WN, STAGES, LW, A_RM, B_RM, PRO, STATS, DB, INIT>(g, lds, (int)blockIdx.x);
}

// Backward of one Linear in ONE launch: the first workgroups compute the dX tiles (NN), the rest the split-M dW slabs
// (TN).  Both stream the same dY; one launch instead of two removes a boundary and lets the two under-filled grids
// of the node-sized layers share the chip.
struct DualArgs { GArgs dx, dw; int n_dx; int b0; };   // b0: first job of this launch (a launch may carry only the dX or only the dW tiles)
// BNB: dY is still the gradient of the BatchNorm(+ReLU) OUTPUT; its backward is applied to the operand as it is staged
// (both jobs); BSTAT: the dX tiles also emit the column sums of the NEXT BatchNorm backward (see BnbDev / BnStatDev).
template <int BM, int BN, int BK, int WM, int WN, int STAGES, int LW, bool PRO, bool BNB = false, bool BSTAT = false>
__global__ __launch_bounds__((WM * WN + LW) * 64) void gemm_dual_kernel(DualArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  if constexpr (BM * BN < 128 * 128) ESC_PRIO();
  const int b = (int)blockIdx.x + a.b0;
  if (b < a.n_dx) gemm_body<BM, BN, BK, WM, WN, STAGES, LW, false, true, (BNB ? 4 : 0) | (BSTAT ? 8 : 0), false, false>(a.dx, lds, b);
  else gemm_body<BM, BN, BK, WM, WN, STAGES, LW, true, true, (PRO ? 2 : 0) | (BNB ? 4 : 0), false, true>(a.dw, lds, b - a.n_dx);
}

inline int splits_of(const GArgs& g) { return g.red_per_split >= g.R ? 1 : (int)cdiv(g.R, g.red_per_split); }

template <typename K>
inline hipError_t raise_lds(K kern, size_t lds, size_t& raised_to) {
  if (lds > raised_to) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    raised_to = lds;
  }
  return hipSuccess;
}

template <int BM, int BN>
inline void finish_args(GArgs& g) {
  g.c_vec = (g.N % 4 == 0 && g.ldc % 4 == 0 && aligned16(g.C) && (g.bias == nullptr || aligned16(g.bias))) ? 1 : 0;
  g.ntile_m = (int)cdiv(g.M, BM);
  g.ntile_n = (int)cdiv(g.N, BN);
  if (g.red_per_split <= 0 || g.red_per_split > g.R) g.red_per_split = g.R;
}

template <int BM, int BN, int BK, int WM, int WN, int STAGES, int LW, bool A_RM, bool B_RM, int PRO, bool STATS, bool DB, bool INIT = false>
inline hipError_t launch_gemm(GArgs g, size_t lds_floor, hipStream_t s, int kind = ESC_K_LINEAR) {
  using C_ = Cfg<BM, BN, BK, WM, WN, STAGES, LW, A_RM, B_RM, PRO, STATS, DB>;
  auto kern = gemm_kernel<BM, BN, BK, WM, WN, STAGES, LW, A_RM, B_RM, PRO, STATS, DB, INIT>;
  if (BM < 128 && gemm_lds_floor() == 0 && (size_t)node_lds_floor() > lds_floor) lds_floor = (size_t)node_lds_floor();
  const size_t lds = C_::LDS_BYTES > lds_floor ? C_::LDS_BYTES : lds_floor;
  static size_t raised_to = 64 * 1024;      // per instantiation
  hipError_t e = raise_lds(kern, lds, raised_to);
  if (e != hipSuccess) return e;
  finish_args<BM, BN>(g);
  const unsigned nwg = (unsigned)(g.ntile_m * g.ntile_n * splits_of(g));
  esc::launch(BM >= 128 && kind == ESC_K_LINEAR ? ESC_K_GEMM_EDGE : kind, kern, dim3(nwg), dim3(C_::NTHR), lds, s, g);
  return hipSuccess;
}

// s_dw: nullptr / s = one launch; another stream = the dX tiles on s and the dW tiles on s_dw (the caller has ordered s_dw
// behind the operands' producers and keeps the operands untouched until s_dw has drained)
template <int BM, int BN, int BK, int WM, int WN, int STAGES, int LW, bool PRO, bool BNB = false, bool BSTAT = false>
inline hipError_t launch_dual(DualArgs a, size_t lds_floor, hipStream_t s, int kind = ESC_K_LINEAR, hipStream_t s_dw = nullptr) {
  using CX = Cfg<BM, BN, BK, WM, WN, STAGES, LW, false, true, (BNB ? 4 : 0) | (BSTAT ? 8 : 0), false, false>;
  using CW = Cfg<BM, BN, BK, WM, WN, STAGES, LW, true, true, (PRO ? 2 : 0) | (BNB ? 4 : 0), false, true>;
  auto kern = gemm_dual_kernel<BM, BN, BK, WM, WN, STAGES, LW, PRO, BNB, BSTAT>;
  const size_t lds_x = CX::LDS_BYTES + (size_t)CX::bnb_lds_floats(a.dx.R) * 4;        // (+ the dX job's per-k BatchNorm coefficients)
  size_t lds = lds_x > CW::LDS_BYTES ? lds_x : CW::LDS_BYTES;
  if (BM < 128 && gemm_lds_floor() == 0 && (size_t)node_lds_floor() > lds_floor) lds_floor = (size_t)node_lds_floor();
  if (lds_floor > lds) lds = lds_floor;
  static size_t raised_to = 64 * 1024;
  hipError_t e = raise_lds(kern, lds, raised_to);
  if (e != hipSuccess) return e;
  finish_args<BM, BN>(a.dx);
  finish_args<BM, BN>(a.dw);
  a.n_dx = a.dx.ntile_m * a.dx.ntile_n;
  a.b0 = 0;
  const unsigned n_dw = (unsigned)(a.dw.ntile_m * a.dw.ntile_n * splits_of(a.dw));
  const int k = BM >= 128 && kind == ESC_K_LINEAR ? ESC_K_GEMM_EDGE : kind;
  if (s_dw != nullptr && s_dw != s && a.n_dx > 0 && n_dw > 0) {
    esc::launch(k, kern, dim3((unsigned)a.n_dx), dim3(CX::NTHR), lds, s, a);
    a.b0 = a.n_dx;
    esc::launch(k, kern, dim3(n_dw), dim3(CX::NTHR), lds, s_dw, a);
    return hipSuccess;
  }
  esc::launch(k, kern, dim3((unsigned)a.n_dx + n_dw), dim3(CX::NTHR), lds, s, a);
  return hipSuccess;
}

}  // namespace dma
}  // namespace esc
