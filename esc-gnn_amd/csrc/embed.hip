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

// dtable[r, c] = sum_{i: idx[i] == r} g[i, c]; one workgroup per table row, thread (rg, c) walks i = rg, rg+RG, ...
// and the RG partial sums are added in ascending rg.
__global__ __launch_bounds__(256) void embed_bwd_kernel(const float* __restrict__ g, int64_t ld, const int64_t* __restrict__ idx,
                                                        int64_t M, int64_t C, int CW, float* __restrict__ dtable) {
  __shared__ float part[256];
  const int r = blockIdx.x;
  const int c = threadIdx.x % CW, rg = threadIdx.x / CW, RG = 256 / CW;
  for (int64_t c0 = 0; c0 < C; c0 += CW) {
    const int64_t col = c0 + c;
    float acc = 0.f;
    if (col < C)
      for (int64_t i = rg; i < M; i += RG)
        if (idx[i] == r) acc += g[i * ld + col];
    part[threadIdx.x] = acc;
    __syncthreads();
    if (rg == 0 && col < C) {
      float s = part[c];
      for (int k = 1; k < RG; ++k) s += part[k * CW + c];
      dtable[(int64_t)r * C + col] = s;
    }
    __syncthreads();
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
  int CW = 1;
  while (CW < C && CW < 256) CW <<= 1;
  esc::launch(ESC_K_BAG_BWD, embed_bwd_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, g, ld_g, idx, M, C, CW, dtable);
  ESC_CHECK_LAUNCH("esc_embed_bwd");
  return ESC_OK;
}

}  // extern "C"
