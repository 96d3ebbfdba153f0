// util.hip -- error state, device facts, small elementwise kernels, HIP-graph capture.
#include <stdarg.h>

#include "common.h"

namespace arx {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int cu_count() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess)
      n = p.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}

// ---- kernels ---------------------------------------------------------------
template <typename T>
__global__ void k_fill(T* __restrict__ p, int64_t n, T v) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = v;
}

__global__ void k_axpby(float a, const float* __restrict__ x, float b, float* __restrict__ y,
                        int64_t n) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  if (b == 0.f) {
    for (; i < n; i += stride) y[i] = a * x[i];
  } else {
    for (; i < n; i += stride) y[i] = a * x[i] + b * y[i];
  }
}

// y[r, c] = a * x[r % xrows, c] + b * y[r, c]; one thread per float4
__global__ void k_add_rows_bcast(float a, const float* __restrict__ x, int64_t ldx,
                                 int64_t xrows, float b, float* __restrict__ y, int64_t ldy,
                                 int64_t rows, int d4) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t total = rows * d4;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < total; i += stride) {
    int64_t r = i / d4;
    int c = (int)(i % d4) * 4;
    float4 xv = *reinterpret_cast<const float4*>(x + (r % xrows) * ldx + c);
    float4* yp = reinterpret_cast<float4*>(y + r * ldy + c);
    float4 yv = (b == 0.f) ? make_float4(0, 0, 0, 0) : *yp;
    yv.x = a * xv.x + b * yv.x;
    yv.y = a * xv.y + b * yv.y;
    yv.z = a * xv.z + b * yv.z;
    yv.w = a * xv.w + b * yv.w;
    *yp = yv;
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// one wave per row
__global__ void k_row_sum(const float* __restrict__ x, int64_t ld, int64_t rows, int64_t cols,
                          float* __restrict__ out, int accumulate) {
  int lane = threadIdx.x & 63;
  int64_t w = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t r = w; r < rows; r += nw) {
    const float* p = x + r * ld;
    float s = 0.f;
    for (int64_t c = lane; c < cols; c += 64) s += p[c];
    s = wave_sum(s);
    if (lane == 0) out[r] = accumulate ? out[r] + s : s;
  }
}

// column sums, deterministic two-stage: grid (cols/64, row chunks) -> partials
// [chunks, cols] in the caller workspace -> fixed-order final add.  (A single
// stage with cols/64 workgroups left 240 of 256 CUs idle at [4096 x 1024].)
__global__ void k_col_sum_partial(const float* __restrict__ x, int64_t ld, int64_t rows,
                                  int64_t cols, int64_t rows_per_chunk,
                                  float* __restrict__ partial) {
  __shared__ float part[4][64];
  const int cx = threadIdx.x & 63;
  const int ry = threadIdx.x >> 6;
  const int64_t c = blockIdx.x * 64 + cx;
  const int64_t r0 = blockIdx.y * rows_per_chunk;
  const int64_t r1 = min(rows, r0 + rows_per_chunk);
  float s = 0.f;
  if (c < cols) {
    float s1 = 0.f, s2 = 0.f, s3 = 0.f;          // 4 independent loads in flight
    int64_t r = r0 + ry;
    for (; r + 12 < r1; r += 16) {
      s += x[r * ld + c];
      s1 += x[(r + 4) * ld + c];
      s2 += x[(r + 8) * ld + c];
      s3 += x[(r + 12) * ld + c];
    }
    for (; r < r1; r += 4) s += x[r * ld + c];
    s = (s + s1) + (s2 + s3);
  }
  part[ry][cx] = s;
  __syncthreads();
  if (ry == 0 && c < cols)
    partial[blockIdx.y * cols + c] = part[0][cx] + part[1][cx] + part[2][cx] + part[3][cx];
}

// 256 threads = 64 columns x 4 partial-row lanes, fixed-order combine through LDS
__global__ void k_col_sum_final(const float* __restrict__ partial, int nchunk, int64_t cols,
                                float* __restrict__ out) {
  __shared__ float part[4][64];
  const int cx = threadIdx.x & 63;
  const int ry = threadIdx.x >> 6;
  const int64_t c = blockIdx.x * 64 + cx;
  float s = 0.f;
  if (c < cols)
    for (int k = ry; k < nchunk; k += 4) s += partial[(int64_t)k * cols + c];
  part[ry][cx] = s;
  __syncthreads();
  if (ry == 0 && c < cols) out[c] = part[0][cx] + part[1][cx] + part[2][cx] + part[3][cx];
}

static inline int col_sum_chunks(int64_t rows, int64_t cols) {
  const int64_t colblocks = ceil_div(cols > 0 ? cols : 1, 64);
  int64_t want = ceil_div(1024, colblocks);          // ~1024 workgroups in flight
  int64_t maxc = ceil_div(rows > 0 ? rows : 1, 16);  // >= 16 rows per chunk
  int64_t n = want < maxc ? want : maxc;
  if (n < 1) n = 1;
  if (n > 64) n = 64;
  return (int)n;
}

// single block deterministic sum
__global__ void k_sum_scaled(const float* __restrict__ x, int64_t n, float scale,
                             float* __restrict__ out) {
  __shared__ float part[16];
  float s = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x) s += x[i];
  s = wave_sum(s);
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) part[w] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += part[i];
    *out = t * scale;
  }
}

__device__ __forceinline__ uint32_t mix32(uint64_t z) {
  // splitmix64 finaliser -> 32 bits; counter-based, order independent
  z += 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  z = z ^ (z >> 31);
  return (uint32_t)(z >> 32);
}

// `step` (nullable): device-side step counter added into the seed, so that a captured
// (hipGraph-replayed) launch draws a fresh mask every replay
__global__ void k_dropout_fwd(const float* __restrict__ x, int64_t n, float keep_prob,
                              uint64_t seed, const uint64_t* __restrict__ step,
                              float* __restrict__ y, uint8_t* __restrict__ keep) {
  if (step) seed += *step * 0x9e3779b97f4a7c15ull;
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  float inv = 1.f / keep_prob;
  for (; i < n; i += stride) {
    float u = (mix32(seed * 0x100000001b3ull + (uint64_t)i) >> 8) * (1.0f / 16777216.0f);
    uint8_t k = (u < keep_prob) ? 1 : 0;
    if (keep) keep[i] = k;
    y[i] = k ? x[i] * inv : 0.f;
  }
}

__global__ void k_counter_add(uint64_t* c, uint64_t v) { *c += v; }

// up to 8 device-to-device copies of 4-byte words in ONE launch (placeholder feeds: the
// runtime's blit kernel costs ~5-10 us per hipMemcpyAsync at these sizes)
struct CopySet {
  const uint32_t* src[8];
  uint32_t* dst[8];
  int64_t n[8];
  int count;
};
__global__ __launch_bounds__(256) void k_copy_words(CopySet cs) {
  for (int a = 0; a < cs.count; ++a) {
    const uint32_t* __restrict__ s = cs.src[a];
    uint32_t* __restrict__ d = cs.dst[a];
    const int64_t n = cs.n[a];
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
      d[i] = s[i];
  }
}

__global__ void k_dropout_bwd(const float* __restrict__ dy, const uint8_t* __restrict__ keep,
                              int64_t n, float keep_prob, float* __restrict__ dx) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  float inv = 1.f / keep_prob;
  for (; i < n; i += stride) dx[i] = keep[i] ? dy[i] * inv : 0.f;
}

__global__ void k_act_fwd(const float* __restrict__ x, int64_t n, int kind,
                          float* __restrict__ y) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) y[i] = kind == 0 ? fmaxf(x[i], 0.f) : tanhf(x[i]);
}

__global__ void k_act_bwd(const float* __restrict__ y, const float* __restrict__ dy, int64_t n,
                          int kind, float* __restrict__ dx) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    float yy = y[i];
    dx[i] = dy[i] * (kind == 0 ? (yy > 0.f ? 1.f : 0.f) : (1.f - yy * yy));
  }
}

__global__ void k_add_col_bias(float* __restrict__ y, int64_t ld, int64_t rows, int64_t cols,
                               const float* __restrict__ b) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t total = rows * cols;
  for (; i < total; i += stride) {
    int64_t r = i / cols, c = i % cols;
    y[r * ld + c] += b[c];
  }
}

static inline int grid_for(int64_t n, int block) {
  int64_t g = ceil_div(n, block);
  int64_t cap = (int64_t)cu_count() * 8;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

// dst[c, r] = src[r, c]: 32x32 tiles through LDS (padded: conflict-free both ways)
__global__ __launch_bounds__(256) void k_transpose(const float* __restrict__ src, int64_t lds,
                                                   int64_t rows, int64_t cols, float* __restrict__ dst,
                                                   int64_t ldd) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8
  const int64_t r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int64_t r = r0 + ty + 8 * k, c = c0 + tx;
    if (r < rows && c < cols) tile[ty + 8 * k][tx] = src[r * lds + c];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int64_t c = c0 + ty + 8 * k, r = r0 + tx;
    if (r < rows && c < cols) dst[c * ldd + r] = tile[tx][ty + 8 * k];
  }
}

}  // namespace arx

using namespace arx;

namespace arx {
namespace {
// dst[r, 0:width] = src[rows[r], 0:width] (0 for a row index outside [0, src_rows)): whole rows of any width --
// the block rows <-> pool slots permutations of the sharded step's logits exchange (arx/dist.py _step_logits),
// whose rows are B_loc floats wide (arx_gather_onehot_fwd stops at d = 1024)
__global__ __launch_bounds__(256) void k_gather_rows_wide(const float* __restrict__ src, int64_t lds, int64_t src_rows,
                                                          const int32_t* __restrict__ rows, int64_t n, int64_t width,
                                                          float* __restrict__ dst, int64_t ldd) {
  const int64_t c0 = (int64_t)blockIdx.x * 1024 + threadIdx.x * 4;
  for (int64_t r = blockIdx.y; r < n; r += gridDim.y) {
    const int64_t sr = rows[r];
    const bool ok = sr >= 0 && sr < src_rows;
    const float* sp = src + (ok ? sr : 0) * lds;
    float* dp = dst + r * ldd;
    if (c0 + 4 <= width && ((lds | ldd) & 3) == 0 && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0) {
      const float4 v = ok ? *reinterpret_cast<const float4*>(sp + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4*>(dp + c0) = v;
    } else {
      for (int64_t c = c0; c < width && c < c0 + 4; ++c) dp[c] = ok ? sp[c] : 0.f;
    }
  }
}
}  // namespace
}  // namespace arx

extern "C" {

const char* arx_last_error(void) { return g_err; }

int arx_version(void) { return 100; }

int arx_device_info(int* cu, int* wave, int* lds_bytes, char* arch, int arch_len) {
  int dev = 0;
  ARX_CHECK_HIP(hipGetDevice(&dev));
  hipDeviceProp_t p;
  ARX_CHECK_HIP(hipGetDeviceProperties(&p, dev));
  if (cu) *cu = p.multiProcessorCount;
  if (wave) *wave = p.warpSize;
  if (lds_bytes) *lds_bytes = (int)p.sharedMemPerBlock;
  if (arch && arch_len > 0) {
    strncpy(arch, p.gcnArchName, arch_len - 1);
    arch[arch_len - 1] = 0;
  }
  return ARX_OK;
}

}  // extern "C"
namespace {
__global__ void k_take_i32(const int32_t* __restrict__ table, const int32_t* __restrict__ idx, int64_t n, int32_t fill,
                           int32_t* __restrict__ out) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) out[i] = idx[i] >= 0 ? table[idx[i]] : fill;
}
}  // namespace
extern "C" {

int arx_fill_f32(float* p, int64_t n, float v, void* stream) {
  ARX_CHECK_ARG(p || n == 0, "arx_fill_f32: null pointer");
  if (n <= 0) return ARX_OK;
  k_fill<float><<<grid_for(n, 256), 256, 0, as_stream(stream)>>>(p, n, v);
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

int arx_fill_i32(int32_t* p, int64_t n, int32_t v, void* stream) {
  ARX_CHECK_ARG(p || n == 0, "arx_fill_i32: null pointer");
  if (n <= 0) return ARX_OK;
  k_fill<int32_t><<<grid_for(n, 256), 256, 0, as_stream(stream)>>>(p, n, v);
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

int arx_take_i32(const int32_t* table, const int32_t* idx, int64_t n, int32_t fill, int32_t* out, void* stream) {
  ARX_CHECK_ARG(n == 0 || (table && idx && out), "arx_take_i32: null pointer");
  if (n <= 0) return ARX_OK;
  k_take_i32<<<grid_for(n, 256), 256, 0, as_stream(stream)>>>(table, idx, n, fill, out);
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

int arx_fill_u8(uint8_t* p, int64_t n, int v, void* stream) {
  ARX_CHECK_ARG(p || n == 0, "arx_fill_u8: null pointer");
  if (n <= 0) return ARX_OK;
  ARX_CHECK_HIP(hipMemsetAsync(p, v, (size_t)n, as_stream(stream)));
  return ARX_OK;
}

int arx_axpby(float a, const float* x, float b, float* y, int64_t n, void* stream) {
  ARX_CHECK_ARG(x && y, "arx_axpby: null pointer");
  if (n <= 0) return ARX_OK;
  k_axpby<<<grid_for(n, 256), 256, 0, as_stream(stream)>>>(a, x, b, y, n);
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

int arx_add_rows_bcast(float a, const float* x, int64_t ldx, int64_t xrows, float b, float* y,
                       int64_t ldy, int64_t rows, int d, void* stream) {
  ARX_CHECK_ARG(x && y && xrows > 0, "arx_add_rows_bcast: bad argument");
  ARX_CHECK_ARG(d % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0, "arx_add_rows_bcast: d, ld must be multiples of 4");
  if (rows <= 0) return ARX_OK;
  k_add_rows_bcast<<<grid_for(rows * (d / 4), 256), 256, 0, as_stream(stream)>>>(
      a, x, ldx, xrows, b, y, ldy, rows, d / 4);
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

int arx_row_sum(const float* x, int64_t ld, int64_t rows, int64_t cols, float* out,
                int accumulate, void* stream) {
  ARX_CHECK_ARG(x && out, "arx_row_sum: null pointer");
  if (rows <= 0) return ARX_OK;
  k_row_sum<<<grid_for(rows * 64, 256), 256, 0, as_stream(stream)>>>(x, ld, rows, cols, out,
                                                                      accumulate);
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

size_t arx_col_sum_workspace_bytes(int64_t rows, int64_t cols) {
  return (size_t)col_sum_chunks(rows, cols) * (size_t)(cols > 0 ? cols : 1) * sizeof(float);
}

int arx_col_sum(const float* x, int64_t ld, int64_t rows, int64_t cols, float* out,
                void* workspace, size_t workspace_bytes, void* stream) {
  ARX_CHECK_ARG(x && out, "arx_col_sum: null pointer");
  if (cols <= 0) return ARX_OK;
  const int nchunk = col_sum_chunks(rows, cols);
  if (!workspace || workspace_bytes < arx_col_sum_workspace_bytes(rows, cols)) {
    set_error("arx_col_sum: workspace too small");
    return ARX_EWORKSPACE;
  }
  float* partial = reinterpret_cast<float*>(workspace);
  const int64_t rpc = ceil_div(rows > 0 ? rows : 1, nchunk);
  dim3 grid((unsigned)ceil_div(cols, 64), (unsigned)nchunk);
  k_col_sum_partial<<<grid, 256, 0, as_stream(stream)>>>(x, ld, rows, cols, rpc, partial);
  ARX_CHECK_LAUNCH();
  k_col_sum_final<<<(int)ceil_div(cols, 64), 256, 0, as_stream(stream)>>>(partial, nchunk, cols, out);
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

int arx_gather_rows_wide(const float* src, int64_t lds, int64_t src_rows, const int32_t* rows, int64_t n,
                         int64_t width, float* dst, int64_t ldd, void* stream) {
  ARX_CHECK_ARG(src && rows && dst, "arx_gather_rows_wide: null pointer");
  ARX_CHECK_ARG(lds >= width && ldd >= width && width > 0, "arx_gather_rows_wide: leading dimension too small");
  if (n <= 0) return ARX_OK;
  // one workgroup per 1024-float piece of a row
  dim3 grid((unsigned)ceil_div(width, 1024), (unsigned)(n < 65535 ? n : 65535));
  k_gather_rows_wide<<<grid, 256, 0, as_stream(stream)>>>(src, lds, src_rows, rows, n, width, dst, ldd);
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

int arx_transpose_f32(const float* src, int64_t lds, int64_t rows, int64_t cols, float* dst,
                      int64_t ldd, void* stream) {
  ARX_CHECK_ARG(src && dst, "arx_transpose_f32: null pointer");
  ARX_CHECK_ARG(lds >= cols && ldd >= rows, "arx_transpose_f32: leading dimension too small");
  if (rows <= 0 || cols <= 0) return ARX_OK;
  dim3 grid((unsigned)ceil_div(cols, 32), (unsigned)ceil_div(rows, 32));
  k_transpose<<<grid, 256, 0, as_stream(stream)>>>(src, lds, rows, cols, dst, ldd);
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

int arx_sum_scaled(const float* x, int64_t n, float scale, float* out, void* stream) {
  ARX_CHECK_ARG(x && out, "arx_sum_scaled: null pointer");
  k_sum_scaled<<<1, 1024, 0, as_stream(stream)>>>(x, n, scale, out);
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

int arx_dropout_fwd(const float* x, int64_t n, float keep_prob, uint64_t seed, float* y,
                    uint8_t* keep_mask, void* stream) {
  ARX_CHECK_ARG(x && y, "arx_dropout_fwd: null pointer");
  ARX_CHECK_ARG(keep_prob > 0.f && keep_prob <= 1.f, "arx_dropout_fwd: keep_prob must be in (0,1]");
  if (n <= 0) return ARX_OK;
  k_dropout_fwd<<<grid_for(n, 256), 256, 0, as_stream(stream)>>>(x, n, keep_prob, seed, nullptr, y,
                                                                 keep_mask);
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

int arx_dropout_fwd_step(const float* x, int64_t n, float keep_prob, uint64_t seed,
                         const uint64_t* step_dev, float* y, uint8_t* keep_mask, void* stream) {
  ARX_CHECK_ARG(x && y && step_dev, "arx_dropout_fwd_step: null pointer");
  ARX_CHECK_ARG(keep_prob > 0.f && keep_prob <= 1.f, "arx_dropout_fwd_step: keep_prob must be in (0,1]");
  if (n <= 0) return ARX_OK;
  k_dropout_fwd<<<grid_for(n, 256), 256, 0, as_stream(stream)>>>(x, n, keep_prob, seed, step_dev, y,
                                                                 keep_mask);
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

int arx_copy_words(int count, const void* const* src, void* const* dst, const int64_t* n_words,
                   void* stream) {
  ARX_CHECK_ARG(count >= 0 && count <= 8, "arx_copy_words: at most 8 copies per call");
  ARX_CHECK_ARG(count == 0 || (src && dst && n_words), "arx_copy_words: null pointer");
  CopySet cs = {};
  int64_t nmax = 0;
  for (int a = 0; a < count; ++a) {
    ARX_CHECK_ARG(n_words[a] >= 0 && (n_words[a] == 0 || (src[a] && dst[a])), "arx_copy_words: bad entry");
    cs.src[a] = reinterpret_cast<const uint32_t*>(src[a]);
    cs.dst[a] = reinterpret_cast<uint32_t*>(dst[a]);
    cs.n[a] = n_words[a];
    if (n_words[a] > nmax) nmax = n_words[a];
  }
  cs.count = count;
  if (nmax == 0) return ARX_OK;
  k_copy_words<<<grid_for(nmax, 256), 256, 0, as_stream(stream)>>>(cs);
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

int arx_counter_add(uint64_t* counter_dev, uint64_t v, void* stream) {
  ARX_CHECK_ARG(counter_dev, "arx_counter_add: null pointer");
  k_counter_add<<<1, 1, 0, as_stream(stream)>>>(counter_dev, v);
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

int arx_dropout_bwd(const float* dy, const uint8_t* keep_mask, int64_t n, float keep_prob,
                    float* dx, void* stream) {
  ARX_CHECK_ARG(dy && keep_mask && dx, "arx_dropout_bwd: null pointer");
  if (n <= 0) return ARX_OK;
  k_dropout_bwd<<<grid_for(n, 256), 256, 0, as_stream(stream)>>>(dy, keep_mask, n, keep_prob,
                                                                 dx);
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

int arx_act_fwd(const float* x, int64_t n, int kind, float* y, void* stream) {
  ARX_CHECK_ARG(x && y && (kind == 0 || kind == 1), "arx_act_fwd: bad argument");
  if (n <= 0) return ARX_OK;
  k_act_fwd<<<grid_for(n, 256), 256, 0, as_stream(stream)>>>(x, n, kind, y);
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

int arx_act_bwd(const float* y, const float* dy, int64_t n, int kind, float* dx, void* stream) {
  ARX_CHECK_ARG(y && dy && dx && (kind == 0 || kind == 1), "arx_act_bwd: bad argument");
  if (n <= 0) return ARX_OK;
  k_act_bwd<<<grid_for(n, 256), 256, 0, as_stream(stream)>>>(y, dy, n, kind, dx);
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

int arx_add_col_bias(float* y, int64_t ld, int64_t rows, int64_t cols, const float* b,
                     void* stream) {
  ARX_CHECK_ARG(y && b, "arx_add_col_bias: null pointer");
  if (rows * cols <= 0) return ARX_OK;
  k_add_col_bias<<<grid_for(rows * cols, 256), 256, 0, as_stream(stream)>>>(y, ld, rows, cols, b);
  ARX_CHECK_LAUNCH();
  return ARX_OK;
}

// ---- HIP graph capture -------------------------------------------------------
int arx_capture_begin(void* stream) {
  ARX_CHECK_HIP(hipStreamBeginCapture(as_stream(stream), hipStreamCaptureModeThreadLocal));
  return ARX_OK;
}

// ARX_GRAPH_PRIO=1 (experiment, round 6): the kernel nodes of the step's ids-only sort branch (names matching
// ARX_GRAPH_PRIO_MATCH, default the radix / extraction / expansion kernels) get the device's greatest stream priority
// as their hipKernelNodeAttributePriority before the graph is instantiated -- the branch is C3's critical path (190 us
// under contention with the scorer against 109 us alone) and its kernels wait for wave slots.
static void prio_nodes(hipGraph_t g, hipStream_t s) {
  static const char* on = getenv("ARX_GRAPH_PRIO");
  if (!on || !*on || *on == '0') return;
  const char* match = getenv("ARX_GRAPH_PRIO_MATCH");
  if (!match || !*match) match = "k_rs_,k_runs_extract,k_bag_expand,k_head_len,k_site_keys";
  int least = 0, greatest = 0;
  if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) return;
  size_t nn = 0;
  if (hipGraphGetNodes(g, nullptr, &nn) != hipSuccess || nn == 0) return;
  hipGraphNode_t* all = new hipGraphNode_t[nn];
  int hit = 0;
  if (hipGraphGetNodes(g, all, &nn) == hipSuccess) {
    for (size_t i = 0; i < nn; ++i) {
      hipGraphNodeType ty;
      if (hipGraphNodeGetType(all[i], &ty) != hipSuccess || ty != hipGraphNodeTypeKernel) continue;
      hipKernelNodeParams kp;
      if (hipGraphKernelNodeGetParams(all[i], &kp) != hipSuccess || !kp.func) continue;
      const char* name = hipKernelNameRefByPtr(kp.func, s);
      if (!name) continue;
      bool m = false;
      for (const char* p = match; *p && !m;) {
        const char* q = strchr(p, ',');
        const size_t len = q ? (size_t)(q - p) : strlen(p);
        if (len > 0 && len < 128) {
          char tok[128];
          memcpy(tok, p, len);
          tok[len] = 0;
          m = strstr(name, tok) != nullptr;
        }
        p = q ? q + 1 : p + len;
      }
      if (!m) continue;
      hipKernelNodeAttrValue v = {};
      v.priority = greatest;
      if (hipGraphKernelNodeSetAttribute(all[i], hipKernelNodeAttributePriority, &v) == hipSuccess) ++hit;
    }
  }
  delete[] all;
  if (getenv("ARX_GRAPH_PRIO_VERBOSE")) fprintf(stderr, "arx: %d of %zu graph nodes at priority %d\n", hit, nn, greatest);
}

int arx_capture_end(void* stream, void** graph_exec_out) {
  ARX_CHECK_ARG(graph_exec_out, "arx_capture_end: null out pointer");
  hipGraph_t g = nullptr;
  ARX_CHECK_HIP(hipStreamEndCapture(as_stream(stream), &g));
  prio_nodes(g, as_stream(stream));
  hipGraphExec_t e = nullptr;
  hipError_t err = hipGraphInstantiate(&e, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (err != hipSuccess) {
    set_error("hipGraphInstantiate failed: %s", hipGetErrorString(err));
    return ARX_EHIP;
  }
  *graph_exec_out = (void*)e;
  return ARX_OK;
}

// ---- placeholder feeds as nodes of the step graph -------------------------------------------------------------
// The feed of a step (arx_copy_words: the batch's id / weight arrays into the tensors the captured kernels read)
// was an eager launch in front of every graph launch: two submissions per step, and the graph's first kernel
// started ~7 us behind the copy.  Captured as the graph's first node(s), the copy's sources are swapped per step
// with hipGraphExecKernelNodeSetParams: one submission per step.
struct FeedNodes {
  hipGraph_t g;
  int n;
  hipGraphNode_t* node;      // [n]
};

int arx_capture_end_feeds(void* stream, void** graph_exec_out, void** feeds_out, int* n_feed_nodes) {
  ARX_CHECK_ARG(graph_exec_out && feeds_out && n_feed_nodes, "arx_capture_end_feeds: null out pointer");
  hipGraph_t g = nullptr;
  ARX_CHECK_HIP(hipStreamEndCapture(as_stream(stream), &g));
  FeedNodes* fn = new FeedNodes();
  fn->g = g;
  fn->n = 0;
  fn->node = nullptr;
  size_t nn = 0;
  hipError_t err = hipGraphGetNodes(g, nullptr, &nn);
  if (err == hipSuccess && nn > 0) {
    hipGraphNode_t* all = new hipGraphNode_t[nn];
    fn->node = new hipGraphNode_t[nn];
    err = hipGraphGetNodes(g, all, &nn);
    for (size_t i = 0; err == hipSuccess && i < nn; ++i) {
      hipGraphNodeType ty;
      if (hipGraphNodeGetType(all[i], &ty) != hipSuccess || ty != hipGraphNodeTypeKernel) continue;
      hipKernelNodeParams kp;
      if (hipGraphKernelNodeGetParams(all[i], &kp) != hipSuccess) continue;
      if (kp.func == reinterpret_cast<void*>(k_copy_words)) fn->node[fn->n++] = all[i];
    }
    delete[] all;
  }
  prio_nodes(g, as_stream(stream));
  hipGraphExec_t e = nullptr;
  if (err == hipSuccess) err = hipGraphInstantiate(&e, g, nullptr, nullptr, 0);
  if (err != hipSuccess) {
    (void)hipGraphDestroy(g);
    delete[] fn->node;
    delete fn;
    set_error("arx_capture_end_feeds: %s", hipGetErrorString(err));
    return ARX_EHIP;
  }
  *graph_exec_out = (void*)e;
  *feeds_out = (void*)fn;
  *n_feed_nodes = fn->n;
  return ARX_OK;
}

/* first destination pointer of feed node idx as captured (the caller matches nodes to its feed groups by it) */
int arx_graph_feed_dst0(void* feeds, int idx, void** dst0) {
  FeedNodes* fn = reinterpret_cast<FeedNodes*>(feeds);
  ARX_CHECK_ARG(fn && dst0 && idx >= 0 && idx < fn->n, "arx_graph_feed_dst0: bad argument");
  hipKernelNodeParams kp;
  ARX_CHECK_HIP(hipGraphKernelNodeGetParams(fn->node[idx], &kp));
  ARX_CHECK_ARG(kp.kernelParams && kp.kernelParams[0], "arx_graph_feed_dst0: the node carries no argument block");
  *dst0 = reinterpret_cast<const CopySet*>(kp.kernelParams[0])->dst[0];
  return ARX_OK;
}

/* feed node idx of the instantiated graph copies these (count 0: nothing) at its next launches */
int arx_graph_set_feed(void* graph_exec, void* feeds, int idx, int count, const void* const* src, void* const* dst,
                       const int64_t* n_words) {
  FeedNodes* fn = reinterpret_cast<FeedNodes*>(feeds);
  ARX_CHECK_ARG(graph_exec && fn && idx >= 0 && idx < fn->n && count >= 0 && count <= 8,
                "arx_graph_set_feed: bad argument");
  ARX_CHECK_ARG(count == 0 || (src && dst && n_words), "arx_graph_set_feed: null pointer");
  CopySet cs = {};
  int64_t nmax = 1;
  for (int a = 0; a < count; ++a) {
    ARX_CHECK_ARG(n_words[a] >= 0 && (n_words[a] == 0 || (src[a] && dst[a])), "arx_graph_set_feed: bad entry");
    cs.src[a] = reinterpret_cast<const uint32_t*>(src[a]);
    cs.dst[a] = reinterpret_cast<uint32_t*>(dst[a]);
    cs.n[a] = n_words[a];
    if (n_words[a] > nmax) nmax = n_words[a];
  }
  cs.count = count;
  hipKernelNodeParams kp;
  ARX_CHECK_HIP(hipGraphKernelNodeGetParams(fn->node[idx], &kp));
  void* args[1] = {&cs};
  kp.kernelParams = args;
  kp.extra = nullptr;
  kp.gridDim = dim3((unsigned)grid_for(nmax, 256), 1, 1);
  kp.blockDim = dim3(256, 1, 1);
  ARX_CHECK_HIP(hipGraphExecKernelNodeSetParams((hipGraphExec_t)graph_exec, fn->node[idx], &kp));
  return ARX_OK;
}

int arx_graph_feeds_destroy(void* feeds) {
  FeedNodes* fn = reinterpret_cast<FeedNodes*>(feeds);
  if (fn) {
    (void)hipDeviceSynchronize();            // (see arx_graph_destroy)
    (void)hipGraphDestroy(fn->g);
    delete[] fn->node;
    delete fn;
  }
  return ARX_OK;
}

int arx_graph_launch(void* graph_exec, void* stream) {
  ARX_CHECK_ARG(graph_exec, "arx_graph_launch: null graph");
  ARX_CHECK_HIP(hipGraphLaunch((hipGraphExec_t)graph_exec, as_stream(stream)));
  return ARX_OK;
}

int arx_graph_destroy(void* graph_exec) {
  // Destroyed at a QUIET device: an executable graph destroyed while launches of ANOTHER executable graph were still in
  // flight made that graph's next hipGraphLaunch segfault inside the runtime (round 6, ROCm 7.0: bench.py's last
  // workload, whenever Python's cyclic collector freed an earlier workload's plan in the middle of its replay loop).
  // Destruction is rare (a plan dies with its model); the synchronisation costs one pipeline bubble.
  if (graph_exec) {
    (void)hipDeviceSynchronize();
    ARX_CHECK_HIP(hipGraphExecDestroy((hipGraphExec_t)graph_exec));
  }
  return ARX_OK;
}

}  // extern "C"
