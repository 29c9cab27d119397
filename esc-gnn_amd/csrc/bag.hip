// bag.hip — the ESC structural-encoding bag  z_emb = P * W  and its table gradient  dW = P^T * dZ.
//
// P is the batch's sparse E x 1800 integer count matrix (the per-edge ego-net histograms built by
// create_subgraphs), W the 1800 x H embedding table (z_initial.weight).  Replaces
//   global_add_pool(z_initial.weight[pos_index] * pos_enc.view(-1,1), pos_batch)
// at /root/reference/run_graphcount.py:155 (= zinc_models.py:590, ogb_mol_gnn.py:716,
// kernel/gin.py:342-344), which materialises a Z x H temporary (563 MB @cfg1) and scatter-adds it.
//
// Roofline: HBM-bound on paper (table 1.8 MB is L2-resident; algorithmic bytes = entries + output
// rows); in practice bounded by L2->CU row traffic (Z rows of H floats).  One wave owns one output
// row (an edge): the entry list of the row is wave-uniform, so indices/counts travel through the
// scalar unit and every table row is one coalesced 16 B/lane read (H=256 => exactly 1 KiB/wave).
#include "common.h"

#include <cstdlib>

// Bitwise contract with the sequential CPU scatter: this file is compiled with -ffp-contract=off
// (see Makefile) so a*b+c is never fused behind our back; explicit fmaf() calls still emit FMAs
// where the order is free.

namespace esc {

// ---- forward -----------------------------------------------------------------------------------
// Bitwise contract: out = (((0 + w0*v0) + w1*v1) + ...) with separately rounded products, i.e.
// what a sequential scatter_add_ of the rounded products gives.  __fmul_rn/__fadd_rn are never
// contracted into an fma.
template <int VEC, bool ACC = false>
__global__ __launch_bounds__(256) void bag_fwd_kernel(const float* __restrict__ table, int H,
                                                      const int* __restrict__ row_ptr,
                                                      const int* __restrict__ idx,
                                                      const int* __restrict__ val, int E,
                                                      float* __restrict__ out, int64_t ld_out) {
  const int row = uniform((int)((blockIdx.x * (unsigned)blockDim.x + threadIdx.x) >> 6));
  if (row >= E) return;
  const int lane = lane_id();
  const int beg = uniform(row_ptr[row]);
  const int end = uniform(row_ptr[row + 1]);
  for (int c = lane * VEC; c < H; c += WAVE * VEC) {
    float acc[VEC];
#pragma unroll
    for (int t = 0; t < VEC; ++t) acc[t] = ACC ? out[(size_t)row * ld_out + c + t] : 0.f;   // ACC: add onto what is there
    int j = beg;
    // 8 table rows in flight per wave
    for (; j + 8 <= end; j += 8) {
      float w[8][VEC];
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int r = uniform(idx[j + u]);
        v[u] = (float)uniform(val[j + u]);
        const float* p = table + (size_t)r * H + c;
        if constexpr (VEC == 4) {
          const float4 q = *reinterpret_cast<const float4*>(p);
          w[u][0] = q.x; w[u][1] = q.y; w[u][2] = q.z; w[u][3] = q.w;
        } else {
          w[u][0] = *p;
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int t = 0; t < VEC; ++t) acc[t] = __fadd_rn(acc[t], __fmul_rn(w[u][t], v[u]));
    }
    if (j + 4 <= end) {
      float w[4][VEC];
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int r = uniform(idx[j + u]);
        v[u] = (float)uniform(val[j + u]);
        const float* p = table + (size_t)r * H + c;
        if constexpr (VEC == 4) {
          const float4 q = *reinterpret_cast<const float4*>(p);
          w[u][0] = q.x; w[u][1] = q.y; w[u][2] = q.z; w[u][3] = q.w;
        } else {
          w[u][0] = *p;
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int t = 0; t < VEC; ++t) acc[t] = __fadd_rn(acc[t], __fmul_rn(w[u][t], v[u]));
      j += 4;
    }
    for (; j < end; ++j) {
      const int r = uniform(idx[j]);
      const float v = (float)uniform(val[j]);
      const float* p = table + (size_t)r * H + c;
      if constexpr (VEC == 4) {
        const float4 q = *reinterpret_cast<const float4*>(p);
        acc[0] = __fadd_rn(acc[0], __fmul_rn(q.x, v));
        acc[1] = __fadd_rn(acc[1], __fmul_rn(q.y, v));
        acc[2] = __fadd_rn(acc[2], __fmul_rn(q.z, v));
        acc[3] = __fadd_rn(acc[3], __fmul_rn(q.w, v));
      } else {
        acc[0] = __fadd_rn(acc[0], __fmul_rn(*p, v));
      }
    }
    float* o = out + (size_t)row * ld_out + c;
    if constexpr (VEC == 4) {
      *reinterpret_cast<float4*>(o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    } else {
      *o = acc[0];
    }
  }
}


// ---- forward, LDS-staged table slices (r03) ------------------------------------------------------------------------------
// The wave-per-row kernel above pulls Z rows of H floats through the L2 (563 MB for 21.9 MB of algorithmic bytes at config
// 1: 22.7 us at ~24.8 TB/s of L2 traffic).  A block of consecutive edges touches few DISTINCT table rows (the edges of one
// or two graphs share their histogram bins: ~60-150 of the 1800 rows), so a workgroup owns BAG_EB edges x a 64-column slice and
//   A. reads its edges' entry lists ONCE (they are contiguous: one coalesced sweep), marks the table rows they use, numbers
//      those rows (any injective numbering will do) and rewrites the entries in LDS as (slot, count) pairs;
//   B. stages the used rows' slices in LDS (<= BAG_CAP rows x 256 B);
//   C. serves every entry from LDS: 16 lanes x float4 cover the 64 columns of one edge, a wave walks 4 edges at a time,
//      8 entries per batch (8 entry reads, then 8 row reads, then the adds in entry order).
// The L2 sees each used (row, slice) once per workgroup.  Per column the sum still runs over the entries in order with
// separately rounded products: bit-identical to the kernel above (and to the sequential scatter).  A workgroup whose edges
// use more rows / carry more entries than fit (or a count >= 65536) reads rows and entries from global memory in the same lane
// layout.  STATS: the workgroup also leaves the (mean, M2) of its BAG_EB rows per column — the BatchNorm partials
// esc_bn_stats_from_partials_rows(block_rows = BAG_EB) merges, i.e. the statistics pass over the output is gone.
// (v1 of this kernel kept the entry lists in global memory and broadcast them with ds_bpermute: 70.9 us.)
constexpr int BAG_EB = 128;          // edges per workgroup (32 per wave, 4 at a time)
constexpr int BAG_CAP = 192;         // table rows a workgroup can stage: 192 x 256 B = 48 KB
constexpr int BAG_ENT = 6144;        // entries a workgroup can keep in LDS (4 bytes each: slot | count << 16)
constexpr int BAG_MAXROWS = 4096;    // table height the row map serves

template <bool ACC, bool STATS>
__global__ __launch_bounds__(256) void bag_fwd_tiled(const float* __restrict__ table, int rows, int H,
                                                     const int* __restrict__ row_ptr, const int* __restrict__ idx,
                                                     const int* __restrict__ val, int E, float* __restrict__ out, int64_t ld_out,
                                                     float2* __restrict__ stats) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* tile = reinterpret_cast<float*>(smem);                                           // [BAG_CAP][64]
  unsigned* ent = reinterpret_cast<unsigned*>(smem + BAG_CAP * 256);                      // [BAG_ENT]
  unsigned short* map = reinterpret_cast<unsigned short*>(smem + BAG_CAP * 256 + BAG_ENT * 4);   // [rows] -> slot, 0xFFFF = unused
  unsigned short* list = map + ((rows + 7) & ~7);                                         // [BAG_CAP] slot -> row
  __shared__ int n_active, bad;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c0 = blockIdx.x * 64;                                  // this workgroup's column slice
  const int e0 = blockIdx.y * BAG_EB, e1 = min(E, e0 + BAG_EB);
  const int zb = row_ptr[e0], ze = row_ptr[e1], nent = ze - zb;
  // A. entries -> LDS, used rows marked
  for (int r = tid; r < rows; r += 256) map[r] = 0xFFFF;
  if (tid == 0) { n_active = 0; bad = nent > BAG_ENT ? 1 : 0; }
  __syncthreads();
  if (nent <= BAG_ENT) {
    for (int z = tid; z < nent; z += 256) {
      const int r = idx[zb + z], v = val[zb + z];
      if ((unsigned)v >= 65536u) bad = 1;                          // (benign race: every writer stores 1)
      ent[z] = (unsigned)r | ((unsigned)v << 16);
      map[r] = 0xFFFE;                                             // (benign race: the same mark)
    }
  }
  __syncthreads();
  if (!bad) {
    for (int r = tid; r < rows; r += 256) {
      if (map[r] == 0xFFFE) {
        const int sl = atomicAdd(&n_active, 1);
        map[r] = (unsigned short)min(sl, 0xFFF0);
        if (sl < BAG_CAP) list[sl] = (unsigned short)r;
      }
    }
  }
  __syncthreads();
  const int na = n_active;
  const bool staged = !bad && na <= BAG_CAP;                       // workgroup-uniform
  const int g = lane >> 4, t = lane & 15;                          // 16-lane group = one edge, lane t owns columns c0 + 4t .. +3
  const bool col_ok = c0 + t * 4 < H;
  if (staged) {
    for (int z = tid; z < nent; z += 256) { const unsigned e = ent[z]; ent[z] = (unsigned)map[e & 0xFFFFu] | (e & 0xFFFF0000u); }
    // B. the used rows' slices: a 16-lane group copies one 256-byte piece
    for (int sl = wave * 4 + g; sl < na; sl += 16) {
      const int r = list[sl];
      float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
      if (col_ok) q = *reinterpret_cast<const float4*>(table + (size_t)r * H + c0 + t * 4);
      *reinterpret_cast<float4*>(tile + sl * 64 + t * 4) = q;
    }
  }
  __syncthreads();
  // C. the edges: wave w takes e0 + 32 w .. + 31, four at a time; its 33 row pointers live in one register per lane
  const int my_ptr = (lane <= 32 && e0 + wave * 32 + lane <= E) ? row_ptr[min(e0 + wave * 32 + lane, E)] : ze;
  float s_n = 0.f, s_mean[4] = {0.f, 0.f, 0.f, 0.f}, s_m2[4] = {0.f, 0.f, 0.f, 0.f};     // STATS: Welford over this lane's edges
  for (int it = 0; it < 8; ++it) {
    const int e = e0 + wave * 32 + it * 4 + g;
    const bool live = e < e1;
    const int pb = __shfl(my_ptr, it * 4 + g, 64), pe = __shfl(my_ptr, it * 4 + g + 1, 64);
    const int beg = live ? pb : 0, len = live ? pe - pb : 0;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (ACC && live && col_ok) {
      const float4 q = *reinterpret_cast<const float4*>(out + (size_t)e * ld_out + c0 + t * 4);
      acc[0] = q.x; acc[1] = q.y; acc[2] = q.z; acc[3] = q.w;
    }
    int longest = len;                                             // the four edges of the wave run in lockstep
    longest = max(longest, __shfl_xor(longest, 16, 64));
    longest = max(longest, __shfl_xor(longest, 32, 64));
    if (staged) {
      const unsigned* my = ent + (beg - zb);
      for (int j = 0; j < longest; j += 8) {
        unsigned ev[8];
        float4 w[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) ev[k] = (j + k < len) ? my[j + k] : 0u;             // (one address per 16-lane group)
#pragma unroll
        for (int k = 0; k < 8; ++k) w[k] = *reinterpret_cast<const float4*>(tile + (ev[k] & 0xFFFFu) * 64 + t * 4);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          if (j + k < len) {                                        // (a padded slot must not touch the sum: -0 + 0 = +0)
            const float v = (float)(ev[k] >> 16);
            acc[0] = __fadd_rn(acc[0], __fmul_rn(w[k].x, v)); acc[1] = __fadd_rn(acc[1], __fmul_rn(w[k].y, v));
            acc[2] = __fadd_rn(acc[2], __fmul_rn(w[k].z, v)); acc[3] = __fadd_rn(acc[3], __fmul_rn(w[k].w, v));
          }
        }
      }
    } else {                                                        // rows and entries from global memory, same lane layout
      for (int j = 0; j < longest; j += 16) {
        int my_r = 0; float my_v = 0.f;
        if (j + t < len) { my_r = idx[beg + j + t]; my_v = (float)val[beg + j + t]; }
        const int todo = min(16, longest - j);
        for (int k = 0; k < todo; ++k) {
          const int r = __shfl(my_r, (g << 4) + k, 64);
          const float v = __shfl(my_v, (g << 4) + k, 64);
          const float4 w = col_ok ? *reinterpret_cast<const float4*>(table + (size_t)r * H + c0 + t * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
          if (j + k < len) {
            acc[0] = __fadd_rn(acc[0], __fmul_rn(w.x, v)); acc[1] = __fadd_rn(acc[1], __fmul_rn(w.y, v));
            acc[2] = __fadd_rn(acc[2], __fmul_rn(w.z, v)); acc[3] = __fadd_rn(acc[3], __fmul_rn(w.w, v));
          }
        }
      }
    }
    if (live && col_ok) *reinterpret_cast<float4*>(out + (size_t)e * ld_out + c0 + t * 4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    if constexpr (STATS) {
      if (live) {
        s_n += 1.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) { const float d = acc[q] - s_mean[q]; s_mean[q] += d / s_n; s_m2[q] = fmaf(d, acc[q] - s_mean[q], s_m2[q]); }
      }
    }
  }
  if constexpr (STATS) {      // (count, mean, M2) of the 16 lane groups x waves that share a column: Chan merge in a fixed order
    __syncthreads();                                               // the tile is dead: reuse it, [16 contributors][64 columns][3]
    float* red = tile;
    const int contrib = wave * 4 + g;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float* p = red + (contrib * 64 + t * 4 + q) * 3;
      p[0] = s_n; p[1] = s_mean[q]; p[2] = s_m2[q];
    }
    __syncthreads();
    if (tid < 64 && c0 + tid < H && stats != nullptr) {
      float n = 0.f, mean = 0.f, m2 = 0.f;
      for (int u = 0; u < 16; ++u) {
        const float* p = red + (u * 64 + tid) * 3;
        const float nb = p[0];
        if (nb > 0.f) {
          const float tot = n + nb, delta = p[1] - mean;
          mean += delta * (nb / tot);
          m2 += p[2] + delta * delta * (n * nb / tot);
          n = tot;
        }
      }
      stats[(size_t)blockIdx.y * H + c0 + tid] = make_float2(mean, m2);
    }
  }
}

// ---- table gradient ----------------------------------------------------------------------------
// CSC view (entries sorted by column, stable).  Pass 1: one wave per chunk of CH consecutive
// sorted entries; a column that lies entirely inside the chunk is written straight to dtable,
// a column that crosses a chunk border leaves a partial in slot[chunk][0] (segment touching the
// chunk's first entry) or slot[chunk][1] (segment touching the last entry, if different).
// Pass 2: one wave per column sums its partials in chunk order (=> bitwise reproducible) and
// zero-fills columns without entries.
constexpr int BAG_CH = 64;

template <int VEC>
__device__ __forceinline__ void bag_chunk(int chunk, const float* __restrict__ dz, int64_t ld_dz, int H,
                                          const int* __restrict__ col_ptr, const int* __restrict__ c_row,
                                          const int* __restrict__ c_val, const int* __restrict__ c_col, int Z,
                                          float* __restrict__ dtable, float* __restrict__ partials) {
  const int beg = chunk * BAG_CH;
  if (beg >= Z) return;
  const int end = min(beg + BAG_CH, Z);
  const int lane = lane_id();
  for (int c0 = lane * VEC; c0 < H; c0 += WAVE * VEC) {
    int j = beg;
    while (j < end) {
      const int col = uniform(c_col[j]);
      const int cb = uniform(col_ptr[col]);
      const int ce = uniform(col_ptr[col + 1]);
      const int seg_end = min(ce, end);
      if (seg_end <= j) { ++j; continue; }                       // a column pointer that contradicts c_col must not stall the wave
      float acc[VEC];
#pragma unroll
      for (int t = 0; t < VEC; ++t) acc[t] = 0.f;
      // 8 gradient rows in flight per wave (loads first, then the fmas in entry order: same sums as one by one)
      for (; j + 8 <= seg_end; j += 8) {
        float w[8][VEC], v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int r = uniform(c_row[j + u]);
          v[u] = (float)uniform(c_val[j + u]);
          const float* p = dz + (size_t)r * ld_dz + c0;
          if constexpr (VEC == 4) {
            const float4 q = *reinterpret_cast<const float4*>(p);
            w[u][0] = q.x; w[u][1] = q.y; w[u][2] = q.z; w[u][3] = q.w;
          } else {
            w[u][0] = *p;
          }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int t = 0; t < VEC; ++t) acc[t] = fmaf(w[u][t], v[u], acc[t]);
      }
#pragma unroll 4
      for (; j < seg_end; ++j) {
        const int r = uniform(c_row[j]);
        const float v = (float)uniform(c_val[j]);
        const float* p = dz + (size_t)r * ld_dz + c0;
        if constexpr (VEC == 4) {
          const float4 q = *reinterpret_cast<const float4*>(p);
          acc[0] = fmaf(q.x, v, acc[0]); acc[1] = fmaf(q.y, v, acc[1]);
          acc[2] = fmaf(q.z, v, acc[2]); acc[3] = fmaf(q.w, v, acc[3]);
        } else {
          acc[0] = fmaf(*p, v, acc[0]);
        }
      }
      float* o;
      if (cb >= beg && ce <= end) {
        o = dtable + (size_t)col * H + c0;                      // column interior to this chunk
      } else {
        const int slot = (cb < beg) ? 0 : 1;                    // continues from previous chunk : runs into next
        o = partials + ((size_t)chunk * 2 + slot) * H + c0;
      }
      if constexpr (VEC == 4) {
        *reinterpret_cast<float4*>(o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
      } else {
        *o = acc[0];
      }
    }
  }
}

template <int VEC>
__global__ __launch_bounds__(256) void bag_bwd_pass1(const float* __restrict__ dz, int64_t ld_dz, int H,
                                                     const int* __restrict__ col_ptr,
                                                     const int* __restrict__ c_row,
                                                     const int* __restrict__ c_val,
                                                     const int* __restrict__ c_col, int Z,
                                                     float* __restrict__ dtable,
                                                     float* __restrict__ partials) {
  const int chunk = uniform((int)((blockIdx.x * (unsigned)blockDim.x + threadIdx.x) >> 6));
  bag_chunk<VEC>(chunk, dz, ld_dz, H, col_ptr, c_row, c_val, c_col, Z, dtable, partials);
}

// ---- L2-local scheduling of pass 1 -----------------------------------------------------------------------
// The gradient rows dz (E x H, 15.6 MB at cfg1) do not fit one XCD's 4 MB L2, and a column's entries walk the rows in
// ascending order: 94 % of the 64-entry chunks touch a single eighth of the rows (the histogram has few, long columns).
// Chunks are therefore bucketed by the row eighth of their middle entry, and workgroup b — which shares an XCD, hence
// an L2, with every workgroup b' = b (mod 8) — only takes chunks of bucket b % 8: each L2 then serves ~2 MB of rows at
// ~36x reuse instead of streaming all 15.6 MB from HBM / Infinity Cache.  Placement affects speed only: every chunk is
// still processed exactly once and writes its own slots.
__global__ __launch_bounds__(256) void bag_bwd_classify(const int* __restrict__ c_row, int Z, int rows, int chunks,
                                                        int* __restrict__ order, int* __restrict__ bucket_cnt) {
  // one thread per chunk, many workgroups (the scattered middle-row reads miss: one CU alone needs ~20 us for them);
  // per wave and key ONE device atomic reserves the slots, the list of bucket k is order[k*chunks ...] in arrival
  // order — which does not matter: every chunk is processed once and writes its own slots
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = lane_id();
  const unsigned long long below = (1ull << lane) - 1ull;
  const unsigned eighth = ((unsigned)rows + 7u) / 8u;        // key = row / ceil(rows/8): 0..7
  int key = -1;
  if (q < chunks) {
    const int beg = q * BAG_CH, end = min(beg + BAG_CH, Z);
    key = min(7, (int)((unsigned)c_row[(beg + end) >> 1] / eighth));
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const unsigned long long m = __ballot(key == k);
    if (m == 0) continue;
    int at = 0;
    if (lane == 0) at = atomicAdd(&bucket_cnt[k], __popcll(m));
    at = __shfl(at, 0, 64);
    if (key == k) order[(size_t)k * chunks + at + __popcll(m & below)] = q;
  }
}

template <int VEC>
__global__ __launch_bounds__(256) void bag_bwd_pass1_local(const float* __restrict__ dz, int64_t ld_dz, int H,
                                                           const int* __restrict__ col_ptr,
                                                           const int* __restrict__ c_row,
                                                           const int* __restrict__ c_val,
                                                           const int* __restrict__ c_col, int Z,
                                                           float* __restrict__ dtable, float* __restrict__ partials,
                                                           const int* __restrict__ order,
                                                           const int* __restrict__ bucket_cnt, int chunks) {
  const int group = blockIdx.x & 7;                         // workgroups of one group share an XCD
  const int n = uniform(bucket_cnt[group]);
  const int* __restrict__ mine = order + (size_t)group * chunks;
  const int stride = (int)(gridDim.x >> 3) * 4;
  for (int slot = (int)(blockIdx.x >> 3) * 4 + (int)(threadIdx.x >> 6); slot < n; slot += stride)
    bag_chunk<VEC>(uniform(mine[slot]), dz, ld_dz, H, col_ptr, c_row, c_val, c_col, Z, dtable, partials);
}

// Pass 2: one WORKGROUP per column.  Its 4 waves take the column's chunk partials round-robin
// (wave w sums chunks q0+w, q0+w+4, ...), then the 4 wave sums are added in wave order through LDS:
// a fixed summation tree => bitwise reproducible, and a 15 000-entry column no longer serialises
// 230 dependent row reads in one wave.
template <int VEC>
__global__ __launch_bounds__(256) void bag_bwd_pass2(int H, const int* __restrict__ col_ptr, int n_cols,
                                                     float* __restrict__ dtable,
                                                     const float* __restrict__ partials) {
  extern __shared__ float sh[];                    // [3][H] : sums of waves 1..3
  const int col = blockIdx.x;
  if (col >= n_cols) return;
  const int cb = col_ptr[col];
  const int ce = col_ptr[col + 1];
  const int lane = lane_id(), wave = threadIdx.x >> 6;
  const bool empty = ce == cb;
  const int q0 = cb / BAG_CH;
  const int q1 = empty ? q0 : (ce - 1) / BAG_CH;
  if (!empty && q0 == q1) return;                  // column confined to one chunk: pass 1 wrote it
  for (int c0 = lane * VEC; c0 < H; c0 += WAVE * VEC) {
    float acc[VEC];
#pragma unroll
    for (int t = 0; t < VEC; ++t) acc[t] = 0.f;
    if (!empty) {
#pragma unroll 4
      for (int q = q0 + wave; q <= q1; q += 4) {
        const int slot = (cb < q * BAG_CH) ? 0 : 1;
        const float* p = partials + ((size_t)q * 2 + slot) * H + c0;
        if constexpr (VEC == 4) {
          const float4 v = *reinterpret_cast<const float4*>(p);
          acc[0] += v.x; acc[1] += v.y; acc[2] += v.z; acc[3] += v.w;
        } else {
          acc[0] += *p;
        }
      }
    }
    if (wave > 0) {
#pragma unroll
      for (int t = 0; t < VEC; ++t) sh[(wave - 1) * H + c0 + t] = acc[t];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int t = 0; t < VEC; ++t) acc[t] = ((acc[t] + sh[c0 + t]) + sh[H + c0 + t]) + sh[2 * H + c0 + t];
      float* o = dtable + (size_t)col * H + c0;
      if constexpr (VEC == 4) {
        *reinterpret_cast<float4*>(o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
      } else {
        *o = acc[0];
      }
    }
    __syncthreads();
  }
}

}  // namespace esc

extern "C" {

int esc_bag_fwd(const float* table, int64_t H, const int32_t* row_ptr, const int32_t* idx32,
                const int32_t* val32, int64_t E, float* out, int64_t ld_out, void* stream) {
  ESC_REQUIRE(table && row_ptr && out && idx32 && val32, "esc_bag_fwd: null pointer");
  ESC_REQUIRE(H > 0 && E >= 0 && ld_out >= H, "esc_bag_fwd: bad sizes H=%ld E=%ld ld=%ld", (long)H, (long)E, (long)ld_out);
  ESC_REQUIRE(E < (1LL << 31) / 64, "esc_bag_fwd: E too large");
  if (E == 0) return ESC_OK;
  hipStream_t s = (hipStream_t)stream;
  const bool vec = (H % 4 == 0) && (ld_out % 4 == 0) && esc::aligned16(table) && esc::aligned16(out);
  const int64_t blocks = esc::cdiv(E, 4);
  if (vec)
    esc::launch(ESC_K_BAG_FWD, esc::bag_fwd_kernel<4>, dim3(blocks), dim3(256), 0, s, table, (int)H, row_ptr, idx32, val32, (int)E, out, ld_out);
  else
    esc::launch(ESC_K_BAG_FWD, esc::bag_fwd_kernel<1>, dim3(blocks), dim3(256), 0, s, table, (int)H, row_ptr, idx32, val32, (int)E, out, ld_out);
  ESC_CHECK_LAUNCH("esc_bag_fwd");
  return ESC_OK;
}

int esc_bag_fwd_acc(const float* table, int64_t H, const int32_t* row_ptr, const int32_t* idx32,
                    const int32_t* val32, int64_t E, float* out, int64_t ld_out, void* stream) {
  ESC_REQUIRE(table && row_ptr && out && idx32 && val32, "esc_bag_fwd_acc: null pointer");
  ESC_REQUIRE(H > 0 && E >= 0 && ld_out >= H && E < (1LL << 31) / 64, "esc_bag_fwd_acc: bad sizes H=%ld E=%ld ld=%ld", (long)H, (long)E, (long)ld_out);
  if (E == 0) return ESC_OK;
  hipStream_t s = (hipStream_t)stream;
  const bool vec = (H % 4 == 0) && (ld_out % 4 == 0) && esc::aligned16(table) && esc::aligned16(out);
  const int64_t blocks = esc::cdiv(E, 4);
  if (vec)
    esc::launch(ESC_K_BAG_FWD, esc::bag_fwd_kernel<4, true>, dim3(blocks), dim3(256), 0, s, table, (int)H, row_ptr, idx32, val32, (int)E, out, ld_out);
  else
    esc::launch(ESC_K_BAG_FWD, esc::bag_fwd_kernel<1, true>, dim3(blocks), dim3(256), 0, s, table, (int)H, row_ptr, idx32, val32, (int)E, out, ld_out);
  ESC_CHECK_LAUNCH("esc_bag_fwd_acc");
  return ESC_OK;
}


/* the LDS-staged forward: needs the table height (rows), H % 4 == 0, 16-byte aligned table / out, ld_out % 4 == 0 */
static bool bag_tiled_ok(const float* table, int64_t rows, int64_t H, const float* out, int64_t ld_out, int64_t E) {
  // Measured on a config-1 batch (profiles/r03_kernel_roofline.txt): 70.9 us against 22.7 us of the wave-per-row kernel — 1 904
  // waves walking 32 edges each through dependent entry-list loads cannot hide what 15 200 one-row waves hide by sheer
  // numbers, and every workgroup pays the row map.  OFF by default (ESC_BAG_TILED=1 enables it for experiments); what it
  // would take to win — the workgroup's entry lists staged in LDS too, batched reads — is in DESIGN.md.
  static const int on = getenv("ESC_BAG_TILED") ? atoi(getenv("ESC_BAG_TILED")) : 0;
  return on && rows > 0 && rows <= esc::BAG_MAXROWS && H % 4 == 0 && ld_out % 4 == 0 && esc::aligned16(table) && esc::aligned16(out) &&
         E >= 4 * esc::BAG_EB && esc::cdiv(H, 64) <= 65535;
}
static size_t bag_tiled_lds(int64_t rows) { return (size_t)esc::BAG_CAP * 256 + (size_t)esc::BAG_ENT * 4 + (size_t)((rows + 7) & ~7) * 2 + (size_t)esc::BAG_CAP * 2 + 64; }

int esc_bag_fwd_rows(const float* table, int64_t rows, int64_t H, const int32_t* row_ptr, const int32_t* idx32, const int32_t* val32,
                     int64_t E, float* out, int64_t ld_out, int accumulate, float* stats, void* stream) {
  ESC_REQUIRE(table && row_ptr && out && idx32 && val32, "esc_bag_fwd_rows: null pointer");
  ESC_REQUIRE(H > 0 && E >= 0 && ld_out >= H && rows > 0 && E < (1LL << 31) / 64, "esc_bag_fwd_rows: bad sizes H=%ld E=%ld ld=%ld rows=%ld", (long)H, (long)E, (long)ld_out, (long)rows);
  ESC_REQUIRE(stats == nullptr || (bag_tiled_ok(table, rows, H, out, ld_out, E) && !accumulate && esc::aligned16(stats)),
              "esc_bag_fwd_rows: the statistics epilogue needs the tiled kernel (esc_bag_fwd_stats_block_rows() != 0) and no accumulation");
  if (E == 0) return ESC_OK;
  if (!bag_tiled_ok(table, rows, H, out, ld_out, E))
    return accumulate ? esc_bag_fwd_acc(table, H, row_ptr, idx32, val32, E, out, ld_out, stream)
                      : esc_bag_fwd(table, H, row_ptr, idx32, val32, E, out, ld_out, stream);
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((unsigned)esc::cdiv(H, 64), (unsigned)esc::cdiv(E, esc::BAG_EB));
  const size_t lds = bag_tiled_lds(rows);
  float2* st = reinterpret_cast<float2*>(stats);
  {      // more than the default 64 KB of dynamic LDS: raise the limit once per kernel
    static bool raised = false;
    if (!raised) {
      const int want = (int)bag_tiled_lds(esc::BAG_MAXROWS);
      bool ok = hipFuncSetAttribute((const void*)esc::bag_fwd_tiled<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, want) == hipSuccess;
      ok = ok && hipFuncSetAttribute((const void*)esc::bag_fwd_tiled<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, want) == hipSuccess;
      ok = ok && hipFuncSetAttribute((const void*)esc::bag_fwd_tiled<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, want) == hipSuccess;
      if (!ok) { esc::set_error("esc_bag_fwd_rows: cannot raise the dynamic LDS limit"); return ESC_ELAUNCH; }
      raised = true;
    }
  }
  if (accumulate)      esc::launch(ESC_K_BAG_FWD, esc::bag_fwd_tiled<true, false>, grid, dim3(256), lds, s, table, (int)rows, (int)H, row_ptr, idx32, val32, (int)E, out, ld_out, st);
  else if (stats)      esc::launch(ESC_K_BAG_FWD, esc::bag_fwd_tiled<false, true>, grid, dim3(256), lds, s, table, (int)rows, (int)H, row_ptr, idx32, val32, (int)E, out, ld_out, st);
  else                 esc::launch(ESC_K_BAG_FWD, esc::bag_fwd_tiled<false, false>, grid, dim3(256), lds, s, table, (int)rows, (int)H, row_ptr, idx32, val32, (int)E, out, ld_out, st);
  ESC_CHECK_LAUNCH("esc_bag_fwd_rows");
  return ESC_OK;
}

/* rows per BatchNorm partial the statistics epilogue of esc_bag_fwd_rows leaves (0: this shape is not served by it) */
int64_t esc_bag_fwd_stats_block_rows(const float* table, int64_t rows, int64_t H, const float* out, int64_t ld_out, int64_t E) {
  return bag_tiled_ok(table, rows, H, out, ld_out, E) ? esc::BAG_EB : 0;
}

int64_t esc_bag_bwd_scratch(int64_t Z, int64_t H) {      // chunk partials + (chunk order, bucket starts) of the local schedule
  return 2 * esc::cdiv(Z, esc::BAG_CH) * H + 8 * esc::cdiv(Z, esc::BAG_CH) + 64;
}

static int bag_bwd_impl(const float* dz, int64_t ld_dz, int64_t H, const int32_t* col_ptr,
                        const int32_t* c_row, const int32_t* c_val, const int32_t* c_col, int64_t Z,
                        int64_t n_cols, int64_t rows, int classified, float* dtable, float* partials, void* stream);

static inline bool bag_local_schedule(int64_t Z, int64_t H, int64_t rows) {
  // rows given, the gradient matrix larger than one XCD's L2, enough chunks to spread over 8 groups
  return rows > 0 && rows * H * (int64_t)sizeof(float) > (4 << 20) && esc::cdiv(Z, esc::BAG_CH) >= 64;
}

int esc_bag_bwd_classify(const int32_t* c_row, int64_t Z, int64_t H, int64_t rows, float* partials, void* stream) {
  ESC_REQUIRE(Z == 0 || (c_row && partials), "esc_bag_bwd_classify: null pointer");
  ESC_REQUIRE(H > 0 && Z >= 0 && rows >= 0 && rows < (1LL << 28) && Z < (1LL << 31) - 64, "esc_bag_bwd_classify: bad sizes");
  if (!bag_local_schedule(Z, H, rows)) return ESC_OK;        // esc_bag_bwd_table_rows will not use a schedule either
  hipStream_t s = (hipStream_t)stream;
  const int64_t chunks = esc::cdiv(Z, esc::BAG_CH);
  int* order = reinterpret_cast<int*>(partials + 2 * chunks * H);       // [8][chunks]
  int* bucket_cnt = order + 8 * chunks;                                 // [8]
  if (hipMemsetAsync(bucket_cnt, 0, 8 * sizeof(int), s) != hipSuccess) {
    esc::set_error("esc_bag_bwd_classify: memset failed");
    return ESC_ELAUNCH;
  }
  esc::launch(ESC_K_BAG_BWD, esc::bag_bwd_classify, dim3((unsigned)esc::cdiv(chunks, 256)), dim3(256), 0, s, c_row, (int)Z, (int)rows, (int)chunks, order, bucket_cnt);
  ESC_CHECK_LAUNCH("esc_bag_bwd_classify");
  return ESC_OK;
}

int esc_bag_bwd_table(const float* dz, int64_t ld_dz, int64_t H, const int32_t* col_ptr,
                      const int32_t* c_row, const int32_t* c_val, const int32_t* c_col, int64_t Z,
                      int64_t n_cols, float* dtable, float* partials, void* stream) {
  return bag_bwd_impl(dz, ld_dz, H, col_ptr, c_row, c_val, c_col, Z, n_cols, 0, 0, dtable, partials, stream);
}

int esc_bag_bwd_table_rows(const float* dz, int64_t ld_dz, int64_t H, const int32_t* col_ptr,
                           const int32_t* c_row, const int32_t* c_val, const int32_t* c_col, int64_t Z,
                           int64_t n_cols, int64_t rows, int classified, float* dtable, float* partials,
                           void* stream) {
  ESC_REQUIRE(rows >= 0 && rows < (1LL << 28), "esc_bag_bwd_table_rows: bad row count");
  return bag_bwd_impl(dz, ld_dz, H, col_ptr, c_row, c_val, c_col, Z, n_cols, rows, classified, dtable, partials, stream);
}

static int bag_bwd_impl(const float* dz, int64_t ld_dz, int64_t H, const int32_t* col_ptr,
                        const int32_t* c_row, const int32_t* c_val, const int32_t* c_col, int64_t Z,
                        int64_t n_cols, int64_t rows, int classified, float* dtable, float* partials, void* stream) {
  ESC_REQUIRE(dz && col_ptr && dtable, "esc_bag_bwd_table: null pointer");
  ESC_REQUIRE(Z == 0 || (c_row && c_val && c_col && partials), "esc_bag_bwd_table: null entry arrays");
  ESC_REQUIRE(H > 0 && Z >= 0 && n_cols > 0 && ld_dz >= H, "esc_bag_bwd_table: bad sizes");
  ESC_REQUIRE(Z < (1LL << 31) - 64, "esc_bag_bwd_table: Z too large");
  hipStream_t s = (hipStream_t)stream;
  const bool vec = (H % 4 == 0) && (ld_dz % 4 == 0) && esc::aligned16(dz) && esc::aligned16(dtable) && esc::aligned16(partials);
  if (Z > 0) {
    const int64_t chunks = esc::cdiv(Z, esc::BAG_CH);
    const int64_t blocks = esc::cdiv(chunks, 4);
    if (bag_local_schedule(Z, H, rows)) {          // chunks bucketed by row eighth (see bag_bwd_classify)
      if (!classified) {
        const int rc = esc_bag_bwd_classify(c_row, Z, H, rows, partials, stream);
        if (rc != ESC_OK) return rc;               // the message is already set
      }
      int* order = reinterpret_cast<int*>(partials + 2 * chunks * H);
      int* bucket_cnt = order + 8 * chunks;
      const unsigned per_group = (unsigned)(esc::cdiv(esc::cdiv(chunks, 8) * 5 / 4 + 4, 4));   // 25 % slack; the kernel strides beyond
      if (vec)
        esc::launch(ESC_K_BAG_BWD, esc::bag_bwd_pass1_local<4>, dim3(per_group * 8), dim3(256), 0, s, dz, ld_dz, (int)H, col_ptr, c_row, c_val, c_col, (int)Z, dtable, partials, (const int*)order, (const int*)bucket_cnt, (int)chunks);
      else
        esc::launch(ESC_K_BAG_BWD, esc::bag_bwd_pass1_local<1>, dim3(per_group * 8), dim3(256), 0, s, dz, ld_dz, (int)H, col_ptr, c_row, c_val, c_col, (int)Z, dtable, partials, (const int*)order, (const int*)bucket_cnt, (int)chunks);
      ESC_CHECK_LAUNCH("esc_bag_bwd_table.pass1_local");
    } else if (vec)
      esc::launch(ESC_K_BAG_BWD, esc::bag_bwd_pass1<4>, dim3(blocks), dim3(256), 0, s, dz, ld_dz, (int)H, col_ptr, c_row, c_val, c_col, (int)Z, dtable, partials);
    else
      esc::launch(ESC_K_BAG_BWD, esc::bag_bwd_pass1<1>, dim3(blocks), dim3(256), 0, s, dz, ld_dz, (int)H, col_ptr, c_row, c_val, c_col, (int)Z, dtable, partials);
    ESC_CHECK_LAUNCH("esc_bag_bwd_table.pass1");
  }
  const size_t lds2 = (size_t)3 * H * sizeof(float);
  ESC_REQUIRE(lds2 <= 64 * 1024, "esc_bag_bwd_table: H too large");
  if (vec)
    esc::launch(ESC_K_BAG_BWD, esc::bag_bwd_pass2<4>, dim3((unsigned)n_cols), dim3(256), lds2, s, (int)H, col_ptr, (int)n_cols, dtable, partials);
  else
    esc::launch(ESC_K_BAG_BWD, esc::bag_bwd_pass2<1>, dim3((unsigned)n_cols), dim3(256), lds2, s, (int)H, col_ptr, (int)n_cols, dtable, partials);
  ESC_CHECK_LAUNCH("esc_bag_bwd_table.pass2");
  return ESC_OK;
}

}  // extern "C"
