// embed.hip — lookups in SMALL embedding tables (ZINC's node / edge type tables: zinc_models.py:563-564,587,592;
// 100 rows x 32) for the step engine.  The general sum-of-embeddings path stays on the bag kernels (ops.embedding_sum);
// here the table has few rows, so the gradient is one workgroup per TABLE ROW scanning the index vector — no sort,
// no atomics, a fixed summation order (bitwise reproducible).
#include "common.h"

namespace esc {

// out[i, :] = table[idx[i], :]   (an index outside the table yields a zero row and raises *bad)
__global__ __launch_bounds__(256) void embed_fwd_kernel(const float* __restrict__ table, int64_t rows, int64_t C,
                                                        const int64_t* __restrict__ idx, int64_t M,
                                                        float* __restrict__ out, int64_t ld, int* __restrict__ bad) {
  const int64_t c4 = C >> 2;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= M * c4) return;
  const int64_t i = t / c4, c = (t % c4) << 2;
  const int64_t r = idx[i];
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (r >= 0 && r < rows) v = *reinterpret_cast<const float4*>(table + r * C + c);
  else if (bad) *bad = 1;
  *reinterpret_cast<float4*>(out + i * ld + c) = v;
}

// dtable[r, c] = sum_{i: idx[i] == r} g[i, c]; one workgroup per table row.  Pass over the index vector in slices of
// EMB_SLICE positions: each wave scans its quarter of the slice with coalesced loads and appends the matching positions
// to its own LDS list (ballot compaction, ascending); then thread (rg, c) adds the listed rows rg, rg+RG, ... and the RG
// partial sums are combined in ascending rg — a fixed order, bitwise reproducible.
constexpr int EMB_SLICE = 8192;
__global__ __launch_bounds__(256) void embed_bwd_kernel(const float* __restrict__ g, int64_t ld, const int64_t* __restrict__ idx,
                                                        int64_t M, int64_t C, int CW, float* __restrict__ dtable) {
  __shared__ int list[4][EMB_SLICE / 4];
  __shared__ int cnt[4];
  __shared__ float part[256];
  const int r = blockIdx.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int c = threadIdx.x % CW, rg = threadIdx.x / CW, RG = 256 / CW;
  for (int64_t c0 = 0; c0 < C; c0 += CW) {
    const int64_t col = c0 + c;
    float acc = 0.f;
    for (int64_t s0 = 0; s0 < M; s0 += EMB_SLICE) {
      const int64_t q0 = s0 + (int64_t)wave * (EMB_SLICE / 4);
      int n = 0;
      for (int t = 0; t < EMB_SLICE / 4; t += 64) {
        const int64_t i = q0 + t + lane;
        const bool hit = i < M && idx[i] == r;
        const unsigned long long m = __ballot(hit);
        if (hit) list[wave][n + __popcll(m & ((1ull << lane) - 1ull))] = (int)(i - s0);
        n += __popcll(m);
        if (q0 + t + 64 >= M) break;                      // wave-uniform
      }
      if (lane == 0) cnt[wave] = n;
      __syncthreads();
      if (col < C) {
        for (int w = 0; w < 4; ++w) {                      // wave-major = ascending position
          const int nw = cnt[w];
          int j = rg;
          for (; j + 3 * RG < nw; j += 4 * RG) {
            const float a0 = g[(s0 + list[w][j]) * ld + col], a1 = g[(s0 + list[w][j + RG]) * ld + col];
            const float a2 = g[(s0 + list[w][j + 2 * RG]) * ld + col], a3 = g[(s0 + list[w][j + 3 * RG]) * ld + col];
            acc += a0; acc += a1; acc += a2; acc += a3;
          }
          for (; j < nw; j += RG) acc += g[(s0 + list[w][j]) * ld + col];
        }
      }
      __syncthreads();
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    if (rg == 0 && col < C) {
      float sum = part[c];
      for (int k = 1; k < RG; ++k) sum += part[k * CW + c];
      dtable[(int64_t)r * C + col] = sum;
    }
    __syncthreads();
  }
}

// float4 form for narrow tables (C % 4 == 0, C <= 256: the 32-wide node / edge type embeddings of ZINC): C/4 lanes cover a
// gradient row, so 256 / (C/4) row groups (32 at C = 32, against 8 above) walk the match list side by side with four rows in
// flight each — a type with 1 600 matches is 13 dependent batches per thread instead of 50.  Same fixed summation tree
// for a given (M, C): partial of row group rg = rows rg, rg+RG, ... in ascending order, groups combined in ascending rg.
__global__ __launch_bounds__(256) void embed_bwd_kernel_v4(const float* __restrict__ g, int64_t ld, const int64_t* __restrict__ idx,
                                                           int64_t M, int C, float* __restrict__ dtable) {
  __shared__ int list[4][EMB_SLICE / 4];
  __shared__ int cnt[4];
  __shared__ float4 part[256];
  const int r = blockIdx.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int CW = C >> 2;                                   // lanes per row (1 .. 64)
  const int RG = 256 / CW;                                 // row groups; threads beyond RG * CW idle (C/4 not a divisor of 256)
  const int c = threadIdx.x % CW, rg = threadIdx.x / CW;
  const bool active = rg < RG;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t s0 = 0; s0 < M; s0 += EMB_SLICE) {
    const int64_t q0 = s0 + (int64_t)wave * (EMB_SLICE / 4);
    int n = 0;
    for (int t = 0; t < EMB_SLICE / 4; t += 64) {
      const int64_t i = q0 + t + lane;
      const bool hit = i < M && idx[i] == r;
      const unsigned long long m = __ballot(hit);
      if (hit) list[wave][n + __popcll(m & ((1ull << lane) - 1ull))] = (int)(i - s0);
      n += __popcll(m);
      if (q0 + t + 64 >= M) break;                        // wave-uniform
    }
    if (lane == 0) cnt[wave] = n;
    __syncthreads();
    if (active) {
      for (int w = 0; w < 4; ++w) {                        // wave-major = ascending position
        const int nw = cnt[w];
        int j = rg;
        for (; j + 3 * RG < nw; j += 4 * RG) {
          const float4 a0 = *reinterpret_cast<const float4*>(g + (s0 + list[w][j]) * ld + 4 * c);
          const float4 a1 = *reinterpret_cast<const float4*>(g + (s0 + list[w][j + RG]) * ld + 4 * c);
          const float4 a2 = *reinterpret_cast<const float4*>(g + (s0 + list[w][j + 2 * RG]) * ld + 4 * c);
          const float4 a3 = *reinterpret_cast<const float4*>(g + (s0 + list[w][j + 3 * RG]) * ld + 4 * c);
          acc.x += a0.x; acc.y += a0.y; acc.z += a0.z; acc.w += a0.w;
          acc.x += a1.x; acc.y += a1.y; acc.z += a1.z; acc.w += a1.w;
          acc.x += a2.x; acc.y += a2.y; acc.z += a2.z; acc.w += a2.w;
          acc.x += a3.x; acc.y += a3.y; acc.z += a3.z; acc.w += a3.w;
        }
        for (; j < nw; j += RG) {
          const float4 a = *reinterpret_cast<const float4*>(g + (s0 + list[w][j]) * ld + 4 * c);
          acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w;
        }
      }
    }
    __syncthreads();
  }
  part[threadIdx.x] = acc;
  __syncthreads();
  if (rg == 0) {
    float4 sum = part[c];
    for (int k = 1; k < RG; ++k) {
      const float4 v = part[k * CW + c];
      sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
    }
    *reinterpret_cast<float4*>(dtable + (int64_t)r * C + 4 * c) = sum;
  }
}

// out[i,:] = x[i,:] + rows[graph(i),:] — the virtual-node broadcast h + vn[batch] (ogb_mol_gnn.py:739); one thread per
// float4 of the output, the graph of a row found by bisection of seg_ptr (G+1 ints, L1/L2 resident)
__global__ __launch_bounds__(256) void segment_broadcast_add_kernel(const float* __restrict__ x, int64_t ld_x,
                                                                    const float* __restrict__ rows, int64_t ld_r,
                                                                    const int* __restrict__ seg_ptr, int G, int C,
                                                                    float* __restrict__ out, int64_t ld_o) {
  const int c4 = C >> 2;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int n = seg_ptr[G];
  if (t >= (int64_t)n * c4) return;
  const int r = (int)(t / c4), c = (int)(t % c4) << 2;
  int lo = 0, hi = G;                           // last g with seg_ptr[g] <= r
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (seg_ptr[mid] <= r) lo = mid; else hi = mid;
  }
  const float4 v = *reinterpret_cast<const float4*>(rows + (size_t)lo * ld_r + c);
  float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
  if (x) q = *reinterpret_cast<const float4*>(x + (size_t)r * ld_x + c);
  *reinterpret_cast<float4*>(out + (size_t)r * ld_o + c) = make_float4(q.x + v.x, q.y + v.y, q.z + v.z, q.w + v.w);
}

// dtable[0, c] = sum_i g[i, c]: the gradient of a ONE-row table (virtualnode_embedding).  64 columns x 4 row groups per
// workgroup; group q sums rows q, q+4, ... and the four partial sums are added in order
__global__ __launch_bounds__(256) void embed_bwd_one_row_kernel(const float* __restrict__ g, int64_t ld, int64_t M, int64_t C,
                                                               float* __restrict__ dtable) {
  __shared__ float part[256];
  const int64_t c = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
  const int q = threadIdx.x >> 6;
  float acc = 0.f;
  if (c < C) {
    int64_t i = q;
    for (; i + 28 < M; i += 32) {                 // eight rows in flight (loads first, then the adds in row order)
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = g[(i + 4 * u) * ld + c];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; i < M; i += 4) acc += g[i * ld + c];
  }
  part[threadIdx.x] = acc;
  __syncthreads();
  if (q == 0 && c < C) dtable[c] = ((part[threadIdx.x] + part[threadIdx.x + 64]) + part[threadIdx.x + 128]) + part[threadIdx.x + 192];
}

// counter-based uniform in [0,1): two rounds of a 64-bit mix of (seed, element index) — stateless, so the backward
// could regenerate the mask; it is stored instead (one byte per element) to keep the two passes independent
__device__ __forceinline__ float uniform01(unsigned long long seed, unsigned long long i) {
  unsigned long long z = seed + i * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (float)(z >> 40) * (1.0f / 16777216.0f);
}

// y = dropout_p(x) + res:  keep with probability 1-p, multiply by scale = 1/(1-p) (torch.nn.functional.dropout); p == 0: y = x + res
__global__ __launch_bounds__(256) void dropout_fwd_kernel(const float* __restrict__ x, int64_t ld_x, int64_t M, int C, float p,
                                                          float scale, unsigned long long seed, const float* __restrict__ res, int64_t ld_r,
                                                          float* __restrict__ y, int64_t ld_y, unsigned char* __restrict__ mask) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= M * C) return;
  const int64_t r = t / C;
  const int c = (int)(t % C);
  float v = x[r * ld_x + c];
  if (p > 0.f) {
    const bool keep = uniform01(seed, (unsigned long long)t) >= p;
    mask[t] = keep ? 1 : 0;
    v = keep ? v * scale : 0.f;
  }
  if (res) v += res[r * ld_r + c];
  y[r * ld_y + c] = v;
}

// y = dropout_p(act(x * scale_c + shift_c)) + res: a BatchNorm in coefficient form, its activation (0 none, 1 ReLU), the
// dropout and the residual add of one OGB layer update (ogb_mol_gnn.py:744-755) in ONE pass; element numbering, masks and
// rounding are those of esc_affine_act followed by esc_dropout_fwd
__global__ __launch_bounds__(256) void affine_dropout_fwd_kernel(const float* __restrict__ x, int64_t ld_x, int64_t M, int C,
                                                                 const float* __restrict__ sc, const float* __restrict__ sh, int act,
                                                                 float p, float scale, unsigned long long seed,
                                                                 const float* __restrict__ res, int64_t ld_r, float* __restrict__ y,
                                                                 int64_t ld_y, unsigned char* __restrict__ mask) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= M * C) return;
  const int64_t r = t / C;
  const int c = (int)(t % C);
  float v = fmaf(x[r * ld_x + c], sc[c], sh[c]);
  if (act == 1) v = fmaxf(v, 0.f);
  if (p > 0.f) {
    const bool keep = uniform01(seed, (unsigned long long)t) >= p;
    mask[t] = keep ? 1 : 0;
    v = keep ? v * scale : 0.f;
  }
  if (res) v += res[r * ld_r + c];
  y[r * ld_y + c] = v;
}

// dx = dy * mask / (1-p)  (+ add, when given: the other branch of a residual sum)
__global__ __launch_bounds__(256) void dropout_bwd_kernel(const float* __restrict__ dy, int64_t ld_dy, int64_t M, int C, float p,
                                                          float scale, const unsigned char* __restrict__ mask, const float* __restrict__ add,
                                                          int64_t ld_a, float* __restrict__ dx, int64_t ld_dx) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= M * C) return;
  const int64_t r = t / C;
  const int c = (int)(t % C);
  float v = dy[r * ld_dy + c];
  if (p > 0.f) v = mask[t] ? v * scale : 0.f;
  if (add) v += add[r * ld_a + c];
  dx[r * ld_dx + c] = v;
}

// gather many small tables into one [total_rows, C] buffer (the sum-of-embeddings encoders look rows up in ONE table),
// or scatter the gradient of that buffer back to the tables' own gradient slots
__global__ __launch_bounds__(256) void table_pack_kernel(esc_table_list tl, int C, float* __restrict__ cat, int unpack) {
  int row = blockIdx.x, j = 0;
  while (j < tl.count && row >= tl.rows[j]) { row -= tl.rows[j]; ++j; }
  if (j >= tl.count) return;
  for (int c = threadIdx.x; c < C; c += 256) {
    if (unpack) tl.dw[j][(size_t)row * C + c] = cat[(size_t)blockIdx.x * C + c];
    else        cat[(size_t)blockIdx.x * C + c] = tl.w[j][(size_t)row * C + c];
  }
}

}  // namespace esc

using namespace esc;

extern "C" {

int esc_embed_fwd(const float* table, int64_t rows, int64_t C, const int64_t* idx, int64_t M, float* out, int64_t ld_out,
                  int32_t* bad_flag, void* stream) {
  ESC_REQUIRE(table && out && (idx || M == 0), "esc_embed_fwd: null pointer");
  ESC_REQUIRE(rows > 0 && C > 0 && C % 4 == 0 && ld_out >= C && ld_out % 4 == 0 && M >= 0, "esc_embed_fwd: bad shape rows=%ld C=%ld ld=%ld", (long)rows, (long)C, (long)ld_out);
  ESC_REQUIRE(aligned16(table) && aligned16(out), "esc_embed_fwd: pointers must be 16-byte aligned");
  if (M == 0) return ESC_OK;
  esc::launch(ESC_K_BAG_FWD, embed_fwd_kernel, dim3((unsigned)cdiv(M * (C / 4), 256)), dim3(256), 0, (hipStream_t)stream, table,
              rows, C, idx, M, out, ld_out, (int*)bad_flag);
  ESC_CHECK_LAUNCH("esc_embed_fwd");
  return ESC_OK;
}

int esc_embed_bwd(const float* g, int64_t ld_g, const int64_t* idx, int64_t M, int64_t rows, int64_t C, float* dtable,
                  void* stream) {
  ESC_REQUIRE(dtable && ((g && idx) || M == 0), "esc_embed_bwd: null pointer");
  ESC_REQUIRE(rows > 0 && rows <= 4096 && C > 0 && ld_g >= C && M >= 0, "esc_embed_bwd: table of %ld rows x %ld is not a small one", (long)rows, (long)C);
  if (rows == 1) {
    esc::launch(ESC_K_BAG_BWD, embed_bwd_one_row_kernel, dim3((unsigned)cdiv(C, 64)), dim3(256), 0, (hipStream_t)stream, g, ld_g, M, C, dtable);
    ESC_CHECK_LAUNCH("esc_embed_bwd");
    return ESC_OK;
  }
  if (C % 4 == 0 && C <= 256 && ld_g % 4 == 0 && aligned16(g) && aligned16(dtable)) {
    esc::launch(ESC_K_BAG_BWD, embed_bwd_kernel_v4, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, g, ld_g, idx, M, (int)C, dtable);
    ESC_CHECK_LAUNCH("esc_embed_bwd");
    return ESC_OK;
  }
  int CW = 1;
  while (CW < C && CW < 256) CW <<= 1;
  esc::launch(ESC_K_BAG_BWD, embed_bwd_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, g, ld_g, idx, M, C, CW, dtable);
  ESC_CHECK_LAUNCH("esc_embed_bwd");
  return ESC_OK;
}

int esc_segment_broadcast_add(const float* x, int64_t ld_x, const float* rows, int64_t ld_rows, const int32_t* seg_ptr,
                              int64_t G, int64_t n_rows, int64_t C, float* out, int64_t ld_out, void* stream) {
  ESC_REQUIRE(rows && seg_ptr && out, "esc_segment_broadcast_add: null pointer");
  ESC_REQUIRE(G > 0 && C > 0 && C % 4 == 0 && ld_rows % 4 == 0 && ld_out % 4 == 0 && (!x || ld_x % 4 == 0) && ld_out >= C,
              "esc_segment_broadcast_add: C and the leading dimensions must be multiples of 4");
  ESC_REQUIRE(aligned16(rows) && aligned16(out) && (!x || aligned16(x)), "esc_segment_broadcast_add: pointers must be 16-byte aligned");
  ESC_REQUIRE(n_rows >= 0, "esc_segment_broadcast_add: negative row count");
  if (n_rows == 0) return ESC_OK;
  esc::launch(-1, segment_broadcast_add_kernel, dim3((unsigned)cdiv(n_rows * (C / 4), 256)), dim3(256), 0, (hipStream_t)stream, x, ld_x,
              rows, ld_rows, seg_ptr, (int)G, (int)C, out, ld_out);
  ESC_CHECK_LAUNCH("esc_segment_broadcast_add");
  return ESC_OK;
}

int esc_dropout_fwd(const float* x, int64_t ld_x, int64_t M, int64_t C, float p, uint64_t seed, const float* res, int64_t ld_res,
                    float* y, int64_t ld_y, uint8_t* mask, void* stream) {
  ESC_REQUIRE(x && y && (p <= 0.f || mask), "esc_dropout_fwd: null pointer");
  ESC_REQUIRE(M >= 0 && C > 0 && C < (1LL << 31) && p >= 0.f && p < 1.f, "esc_dropout_fwd: bad arguments (p=%g)", (double)p);
  if (M == 0) return ESC_OK;
  esc::launch(-1, dropout_fwd_kernel, dim3((unsigned)cdiv(M * C, 256)), dim3(256), 0, (hipStream_t)stream, x, ld_x, M, (int)C, p,
              (float)(1.0 / (1.0 - (double)p)), (unsigned long long)seed, res, ld_res, y, ld_y, (unsigned char*)mask);
  ESC_CHECK_LAUNCH("esc_dropout_fwd");
  return ESC_OK;
}

int esc_affine_act_dropout_fwd(const float* x, int64_t ld_x, int64_t M, int64_t C, const float* scale, const float* shift, int act,
                               float p, uint64_t seed, const float* res, int64_t ld_res, float* y, int64_t ld_y, uint8_t* mask,
                               void* stream) {
  ESC_REQUIRE(x && y && scale && shift && (p <= 0.f || mask), "esc_affine_act_dropout_fwd: null pointer");
  ESC_REQUIRE(M >= 0 && C > 0 && C < (1LL << 31) && p >= 0.f && p < 1.f && (act == 0 || act == 1),
              "esc_affine_act_dropout_fwd: bad arguments (p=%g, act=%d)", (double)p, act);
  if (M == 0) return ESC_OK;
  esc::launch(-1, affine_dropout_fwd_kernel, dim3((unsigned)cdiv(M * C, 256)), dim3(256), 0, (hipStream_t)stream, x, ld_x, M, (int)C,
              scale, shift, act, p, (float)(1.0 / (1.0 - (double)p)), (unsigned long long)seed, res, ld_res, y, ld_y, (unsigned char*)mask);
  ESC_CHECK_LAUNCH("esc_affine_act_dropout_fwd");
  return ESC_OK;
}

int esc_dropout_bwd(const float* dy, int64_t ld_dy, int64_t M, int64_t C, float p, const uint8_t* mask, const float* add,
                    int64_t ld_add, float* dx, int64_t ld_dx, void* stream) {
  ESC_REQUIRE(dy && dx && (p <= 0.f || mask), "esc_dropout_bwd: null pointer");
  ESC_REQUIRE(M >= 0 && C > 0 && C < (1LL << 31) && p >= 0.f && p < 1.f, "esc_dropout_bwd: bad arguments (p=%g)", (double)p);
  if (M == 0) return ESC_OK;
  esc::launch(-1, dropout_bwd_kernel, dim3((unsigned)cdiv(M * C, 256)), dim3(256), 0, (hipStream_t)stream, dy, ld_dy, M, (int)C, p,
              (float)(1.0 / (1.0 - (double)p)), (const unsigned char*)mask, add, ld_add, dx, ld_dx);
  ESC_CHECK_LAUNCH("esc_dropout_bwd");
  return ESC_OK;
}

static int table_pack(const esc_table_list* tl, int64_t C, float* cat, int unpack, void* stream) {
  ESC_REQUIRE(tl && cat && tl->count >= 0 && tl->count <= ESC_MAX_TABLES && C > 0, "esc_table_pack: bad table list");
  int64_t total = 0;
  for (int j = 0; j < tl->count; ++j) {
    ESC_REQUIRE(tl->rows[j] > 0 && tl->w[j] && (!unpack || tl->dw[j]), "esc_table_pack: table %d is incomplete", j);
    total += tl->rows[j];
  }
  if (total == 0) return ESC_OK;
  esc::launch(-1, table_pack_kernel, dim3((unsigned)total), dim3(256), 0, (hipStream_t)stream, *tl, (int)C, cat, unpack);
  ESC_CHECK_LAUNCH("esc_table_pack");
  return ESC_OK;
}
int esc_table_pack(const esc_table_list* tl, int64_t C, float* cat, void* stream) { return table_pack(tl, C, cat, 0, stream); }
int esc_table_unpack_grad(const esc_table_list* tl, int64_t C, const float* dcat, void* stream) {
  return table_pack(tl, C, const_cast<float*>(dcat), 1, stream);
}

}  // extern "C"
