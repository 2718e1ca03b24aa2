// prims.hpp — wave64 / workgroup primitives shared by the kernels (gfx950, wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cjs {

constexpr int WAVE = 64;

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

// inclusive wave scan (sum)
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
template <typename T>
__device__ __forceinline__ T wave_sum(T x) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d, 64);
  return x;
}
template <typename T>
__device__ __forceinline__ T wave_max(T x) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { T y = __shfl_xor(x, d, 64); x = y > x ? y : x; }
  return x;
}
template <typename T>
__device__ __forceinline__ T wave_min(T x) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { T y = __shfl_xor(x, d, 64); x = y < x ? y : x; }
  return x;
}

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

}  // namespace cjs
