// prims.hpp — wave64 / workgroup primitives shared by the kernels (gfx950, wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cjs {

constexpr int WAVE = 64;

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

// ---- gfx950 DPP building blocks (wave64 = four rows of 16 lanes).  One DPP move feeds a lane the value of another lane
// of its row without touching LDS (a __shfl is a ds_bpermute: an LDS-pipe round trip per step):
//   row_shr:n   lane i of a row reads lane i-n of the same row (n = 1, 2, 4, 8 -> Hillis-Steele inside the row)
//   row_bcast:15 / row_bcast:31   lane 15 of every row -> the whole next row / lane 31 -> rows 2 and 3
// Lanes whose source is out of the row, or whose row is masked off, receive `idv` (the identity of the operation).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_move(uint32_t idv, uint32_t x) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)idv, (int)x, CTRL, ROW_MASK, 0xF, false);
}
constexpr int DPP_ROW_SHR1 = 0x111, DPP_ROW_SHR2 = 0x112, DPP_ROW_SHR4 = 0x114, DPP_ROW_SHR8 = 0x118, DPP_ROW_BCAST15 = 0x142, DPP_ROW_BCAST31 = 0x143;

// inclusive wave scan of a 32-bit value under an associative op with identity idv: 6 DPP moves
template <typename Op>
__device__ __forceinline__ uint32_t wave_incl_scan_dpp(uint32_t x, uint32_t idv, Op op) {
  x = op(x, dpp_move<DPP_ROW_SHR1, 0xF>(idv, x));
  x = op(x, dpp_move<DPP_ROW_SHR2, 0xF>(idv, x));
  x = op(x, dpp_move<DPP_ROW_SHR4, 0xF>(idv, x));
  x = op(x, dpp_move<DPP_ROW_SHR8, 0xF>(idv, x));        // every row now holds its own inclusive scan
  x = op(x, dpp_move<DPP_ROW_BCAST15, 0xA>(idv, x));     // rows 1, 3 += total of rows 0, 2
  x = op(x, dpp_move<DPP_ROW_BCAST31, 0xC>(idv, x));     // rows 2, 3 += total of rows 0..1
  return x;
}
struct OpAddU32 { __device__ __forceinline__ uint32_t operator()(uint32_t a, uint32_t b) const { return a + b; } };
struct OpMaxU32 { __device__ __forceinline__ uint32_t operator()(uint32_t a, uint32_t b) const { return a > b ? a : b; } };
struct OpMaxI32 { __device__ __forceinline__ uint32_t operator()(uint32_t a, uint32_t b) const { return (int32_t)a > (int32_t)b ? a : b; } };
struct OpMinU32 { __device__ __forceinline__ uint32_t operator()(uint32_t a, uint32_t b) const { return a < b ? a : b; } };

// inclusive wave scan (sum).  32-bit integers: DPP; other types: shuffle ladder
template <typename T>
__device__ __forceinline__ T wave_incl_sum(T x) {
  const int lane = lane_id();
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    T y = __shfl_up(x, d, 64);
    if (lane >= d) x += y;
  }
  return x;
}
__device__ __forceinline__ uint32_t wave_incl_sum(uint32_t x) { return wave_incl_scan_dpp(x, 0u, OpAddU32()); }
__device__ __forceinline__ int32_t wave_incl_sum(int32_t x) { return (int32_t)wave_incl_scan_dpp((uint32_t)x, 0u, OpAddU32()); }
template <typename T>
__device__ __forceinline__ T wave_incl_max(T x) {
  const int lane = lane_id();
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    T y = __shfl_up(x, d, 64);
    if (lane >= d) x = y > x ? y : x;
  }
  return x;
}
__device__ __forceinline__ uint32_t wave_incl_max(uint32_t x) { return wave_incl_scan_dpp(x, 0u, OpMaxU32()); }
__device__ __forceinline__ int32_t wave_incl_max(int32_t x) { return (int32_t)wave_incl_scan_dpp((uint32_t)x, 0x80000000u, OpMaxI32()); }
// wave reductions: the last lane of the inclusive scan holds the result (one scalar readlane broadcasts it)
template <typename T>
__device__ __forceinline__ T wave_sum(T x) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d, 64);
  return x;
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t x) { return (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan_dpp(x, 0u, OpAddU32()), 63); }
__device__ __forceinline__ int32_t wave_sum(int32_t x) { return __builtin_amdgcn_readlane((int)wave_incl_scan_dpp((uint32_t)x, 0u, OpAddU32()), 63); }
template <typename T>
__device__ __forceinline__ T wave_max(T x) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { T y = __shfl_xor(x, d, 64); x = y > x ? y : x; }
  return x;
}
__device__ __forceinline__ uint32_t wave_max(uint32_t x) { return (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan_dpp(x, 0u, OpMaxU32()), 63); }
__device__ __forceinline__ int32_t wave_max(int32_t x) { return __builtin_amdgcn_readlane((int)wave_incl_scan_dpp((uint32_t)x, 0x80000000u, OpMaxI32()), 63); }
template <typename T>
__device__ __forceinline__ T wave_min(T x) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { T y = __shfl_xor(x, d, 64); x = y < x ? y : x; }
  return x;
}
__device__ __forceinline__ uint32_t wave_min(uint32_t x) { return (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan_dpp(x, 0xFFFFFFFFu, OpMinU32()), 63); }

// Workgroup exclusive sum.  smem: at least BLOCK/64 entries of T.  Safe to call repeatedly
// (trailing barrier protects smem reuse).  Returns the exclusive prefix; total in `total`.
template <int BLOCK, typename T>
__device__ __forceinline__ T block_excl_sum(T v, T* smem, T& total) {
  constexpr int NW = BLOCK / 64;
  const int lane = lane_id(), w = wave_id();
  T incl = wave_incl_sum(v);
  if (lane == 63) smem[w] = incl;
  __syncthreads();
  if (w == 0) {
    T s = lane < NW ? smem[lane] : T(0);
    s = wave_incl_sum(s);
    if (lane < NW) smem[lane] = s;
  }
  __syncthreads();
  T base = w ? smem[w - 1] : T(0);
  total = smem[NW - 1];
  __syncthreads();
  return base + incl - v;
}
// Workgroup inclusive max-scan.
template <int BLOCK, typename T>
__device__ __forceinline__ T block_incl_max(T v, T* smem) {
  constexpr int NW = BLOCK / 64;
  const int lane = lane_id(), w = wave_id();
  T incl = wave_incl_max(v);
  if (lane == 63) smem[w] = incl;
  __syncthreads();
  if (w == 0) {
    T s = lane < NW ? smem[lane] : T(0);
    s = wave_incl_max(s);
    if (lane < NW) smem[lane] = s;
  }
  __syncthreads();
  if (w) { T b = smem[w - 1]; incl = b > incl ? b : incl; }
  __syncthreads();
  return incl;
}
template <int BLOCK, typename T>
__device__ __forceinline__ T block_sum(T v, T* smem) {
  constexpr int NW = BLOCK / 64;
  const int lane = lane_id(), w = wave_id();
  T s = wave_sum(v);
  if (lane == 0) smem[w] = s;
  __syncthreads();
  T r = T(0);
#pragma unroll
  for (int i = 0; i < NW; i++) r += smem[i];
  __syncthreads();
  return r;
}

// Device fill on the caller's stream (hipMemsetAsync goes through the runtime's blit path: on this stack its fill kernel
// starts 30-90 us after the kernel in front of it; a plain launch follows within a few us).  `p` 16-byte aligned; the bytes
// up to the next multiple of 16 behind `bytes` are written too (every workspace array is padded to 256 bytes).
static __global__ __launch_bounds__(256) void dev_fill_kernel(uint4* __restrict__ p, uint32_t word, size_t n16) {
  const uint4 v = make_uint4(word, word, word, word);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = v;
}
static inline void dev_fill(hipStream_t s, void* p, uint8_t byte, size_t bytes) {
  const size_t n16 = (bytes + 15) / 16;
  if (!n16) return;
  const size_t wg = (n16 + 255) / 256;
  hipLaunchKernelGGL(dev_fill_kernel, dim3((unsigned)(wg < 4096 ? wg : 4096)), dim3(256), 0, s, (uint4*)p, 0x01010101u * byte, n16);
}

}  // namespace cjs
