// shw_sinkhorn.hip -- log-domain Sinkhorn distance (comparison metric of main_rotation.py).
//
// Replaces log_Sinkhorn_Distance_Loss / log_N_Sinkhorn_Distance_Loss.forward
// (/root/reference/Comparison_Wasserstein_with_Chamfer_distance/losses/sinkhorn.py:14-63, :104-157):
//   C_ij = (sum_d |x_i - y_j|_d^p)^N ;  u, v = 0 ;  repeat max_iter times
//     u_i = eps (log(a + 1e-8) - LSE_j M_ij) + u_i ,  v_j = eps (log(b + 1e-8) - LSE_i M_ij) + v_j ,
//     M_ij = (-C_ij + u_i + v_j) / eps ,  a = 1/n, b = 1/m ;  stop when mean_b sum_i |u - u_old| < 1e-9 ;
//   cost_b = sum_ij exp(M_ij) C_ij.
// The reference materialises C (and every M) as dense (B, n, m) tensors -- 1 GB per sweep at config-3 sizes,
// ~400 sweeps -- and synchronises with the host once per iteration (`err.item()`, :43).  Here C is never
// stored: every pass recomputes c_ij from the 3-d points (6 flops) while streaming the other cloud through
// LDS as broadcast reads, the log-sum-exp is evaluated on line in chunks of 8 candidates (one rescale per
// chunk), and the convergence test is a device-side flag that turns the remaining launches into no-ops, so
// the whole solve is enqueued without a host round trip.  K = 3: no MFMA; the passes are VALU +
// transcendental bound.  Forward only (the reference differentiates through its unrolled loop; main_rotation
// only evaluates the value).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/shw.h"
#include "wave_sort.hpp"

namespace shw {

constexpr int kSkTile = 512;       // candidates staged per LDS tile: (x, y, z, dual) = 16 B each
constexpr float kLog2e = 1.44269504088896341f;
constexpr float kLn2 = 0.693147180559945309f;

struct SinkArgs {
  const float* x;     // (pairs, n, 3)
  const float* y;     // (pairs, m, 3)
  float* u;           // (pairs, n)
  float* v;           // (pairs, m)
  float* err;         // (pairs): sum_i |u_new - u_old| of the current iteration
  int* done;          // [0]: convergence flag
  int n, m;
  float eps, inv_eps;
  int norm_p;         // 1, 2 or other (powf)
  int cost_pow;       // N >= 1
};

// FAST: the reference's only configuration in use ('L2', N = 1): squared Euclidean distance, no branches
template <bool FAST>
__device__ __forceinline__ float pair_cost(float dx, float dy, float dz, int norm_p, int cost_pow) {
  if constexpr (FAST) return fmaf(dz, dz, fmaf(dy, dy, dx * dx));
  float c;
  if (norm_p == 2) c = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
  else if (norm_p == 1) c = fabsf(dx) + fabsf(dy) + fabsf(dz);
  else c = powf(fabsf(dx), (float)norm_p) + powf(fabsf(dy), (float)norm_p) + powf(fabsf(dz), (float)norm_p);
  float r = c;
  for (int k = 1; k < cost_pow; ++k) r *= c;
  return r;
}

// 2^x for x <= 0 (every exponent here has the running maximum subtracted): the bare v_exp_f32, 1 ulp; results
// below 2^-126 flush to zero, which is what the sum wants anyway
__device__ __forceinline__ float exp2_neg(float x) { return __builtin_amdgcn_exp2f(x); }

// One half-iteration: the rows are the points of the cloud whose dual is being updated.
//   TRANSPOSE = false: rows = x (dual u), candidates = y (dual v);  true: the other way round.
// grid (ceil(rows/256), pairs).
template <bool TRANSPOSE, bool FAST>
__global__ __launch_bounds__(256) void sinkhorn_pass_kernel(SinkArgs A) {
  __shared__ float4 tile[kSkTile];
  if (*A.done) return;                                       // converged earlier: this launch is a no-op
  const int b = blockIdx.y;
  const int rows = TRANSPOSE ? A.m : A.n, cands = TRANSPOSE ? A.n : A.m;
  const float* R = (TRANSPOSE ? A.y : A.x) + (long)b * rows * 3;
  const float* Cn = (TRANSPOSE ? A.x : A.y) + (long)b * cands * 3;
  float* dual_r = (TRANSPOSE ? A.v : A.u) + (long)b * rows;
  const float* dual_c = (TRANSPOSE ? A.u : A.v) + (long)b * cands;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int ic = min(i, rows - 1);
  const float rx = R[3 * ic], ry = R[3 * ic + 1], rz = R[3 * ic + 2];
  const float old = dual_r[ic];
  const float scale = A.inv_eps * kLog2e;                    // exponent in base 2
  float run_max = -__builtin_inff(), run_sum = 0.f;
  for (int base = 0; base < cands; base += kSkTile) {
    const int cnt = min(kSkTile, cands - base);
    __syncthreads();
    for (int t = threadIdx.x; t < cnt; t += 256) {
      const int j = base + t;
      tile[t] = make_float4(Cn[3 * j], Cn[3 * j + 1], Cn[3 * j + 2], dual_c[j]);
    }
    __syncthreads();
    int t = 0;
    for (; t + 8 <= cnt; t += 8) {                           // on-line log-sum-exp, one rescale per 8 candidates
      float mval[8];
      float cmax = -__builtin_inff();
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float4 q = tile[t + k];
        const float c = pair_cost<FAST>(rx - q.x, ry - q.y, rz - q.z, A.norm_p, A.cost_pow);
        mval[k] = ((old - c) + q.w) * scale;                // M_ij * log2(e), reference order: -C + u + v
        cmax = fmaxf(cmax, mval[k]);
      }
      const float nmax = fmaxf(run_max, cmax);
      float part = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) part += exp2_neg(mval[k] - nmax);
      run_sum = fmaf(run_sum, exp2_neg(run_max - nmax), part);
      run_max = nmax;
    }
    for (; t < cnt; ++t) {
      const float4 q = tile[t];
      const float c = pair_cost<FAST>(rx - q.x, ry - q.y, rz - q.z, A.norm_p, A.cost_pow);
      const float mv = ((old - c) + q.w) * scale;
      const float nmax = fmaxf(run_max, mv);
      run_sum = fmaf(run_sum, exp2_neg(run_max - nmax), exp2_neg(mv - nmax));
      run_max = nmax;
    }
  }
  const float lse = (run_max + log2f(run_sum)) * kLn2;       // natural-log LSE_j M_ij
  const float marg = TRANSPOSE ? 1.f / (float)A.m : 1.f / (float)A.n;
  const float fresh = A.eps * (logf(marg + 1e-8f) - lse) + old;
  float delta = 0.f;
  if (i < rows) {
    dual_r[i] = fresh;
    delta = fabsf(fresh - old);
  }
  if constexpr (!TRANSPOSE) {                                // convergence statistic: sum_i |u - u_old| (:42)
    delta = wave_sum(delta, threadIdx.x & 63);
    if ((threadIdx.x & 63) == 0) atomicAdd(&A.err[b], delta);
  }
}

// after each (u, v) sweep: mean over pairs of the statistic, set the flag, clear the statistic (:42-44)
__global__ __launch_bounds__(64) void sinkhorn_check_kernel(float* err, int* done, int pairs, float thresh) {
  if (*done) return;
  float acc = 0.f;
  for (int b = threadIdx.x; b < pairs; b += 64) { acc += err[b]; err[b] = 0.f; }
  acc = wave_sum(acc, threadIdx.x);
  if (threadIdx.x == 0 && acc / (float)pairs < thresh) *done = 1;
}

// cost_b = sum_ij exp(M_ij) C_ij ; optionally writes the dense plan P and cost matrix C (drop-in return values)
// grid (ceil(n/256), pairs); per-block partial sums, reduced in a fixed order by sinkhorn_cost_reduce_kernel
__global__ __launch_bounds__(256) void sinkhorn_cost_kernel(SinkArgs A, float* __restrict__ partial,
                                                            float* __restrict__ P, float* __restrict__ Cm) {
  __shared__ float4 tile[kSkTile];
  __shared__ float red[4];
  const int b = blockIdx.y;
  const float* X = A.x + (long)b * A.n * 3;
  const float* Y = A.y + (long)b * A.m * 3;
  const float* U = A.u + (long)b * A.n;
  const float* V = A.v + (long)b * A.m;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int ic = min(i, A.n - 1);
  const float rx = X[3 * ic], ry = X[3 * ic + 1], rz = X[3 * ic + 2], ui = U[ic];
  float acc = 0.f;
  for (int base = 0; base < A.m; base += kSkTile) {
    const int cnt = min(kSkTile, A.m - base);
    __syncthreads();
    for (int t = threadIdx.x; t < cnt; t += 256) {
      const int j = base + t;
      tile[t] = make_float4(Y[3 * j], Y[3 * j + 1], Y[3 * j + 2], V[j]);
    }
    __syncthreads();
    for (int t = 0; t < cnt; ++t) {
      const float4 q = tile[t];
      const float c = pair_cost<false>(rx - q.x, ry - q.y, rz - q.z, A.norm_p, A.cost_pow);
      const float p = expf(((ui - c) + q.w) * A.inv_eps);
      if (i < A.n) {
        acc = fmaf(p, c, acc);
        if (P) P[((long)b * A.n + i) * A.m + base + t] = p;
        if (Cm) Cm[((long)b * A.n + i) * A.m + base + t] = c;
      }
    }
  }
  acc = wave_sum(acc, threadIdx.x & 63);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[(long)b * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(64) void sinkhorn_cost_reduce_kernel(const float* __restrict__ partial, int blocks,
                                                                  float* __restrict__ cost) {
  const int b = blockIdx.x;
  float acc = 0.f;
  for (int k = threadIdx.x; k < blocks; k += 64) acc += partial[(long)b * blocks + k];
  acc = wave_sum(acc, threadIdx.x);
  if (threadIdx.x == 0) cost[b] = acc;
}

}  // namespace shw

extern "C" {

size_t shw_sinkhorn_workspace_bytes(int pairs, int n, int m) {
  if (pairs < 0 || n < 1 || m < 1) return 0;
  const size_t blocks = (size_t)(n + 255) / 256;
  // u, v, err, partial sums, flag (padded to 16 bytes)
  return ((size_t)pairs * ((size_t)n + (size_t)m + 1 + blocks)) * sizeof(float) + 16;
}

int shw_sinkhorn_forward(const float* x, const float* y, int pairs, int n, int m, float eps, int max_iter,
                         int norm_p, int cost_pow, float thresh, void* workspace, float* cost, float* plan,
                         float* cost_matrix, void* stream) {
  if (!x || !y || !workspace || !cost) return (int)hipErrorInvalidValue;
  if (pairs < 0 || pairs > 65535 || n < 1 || m < 1 || !(eps > 0.f) || max_iter < 0 || norm_p < 1 || cost_pow < 1)
    return (int)hipErrorInvalidValue;
  if (pairs == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const size_t blocks = (size_t)(n + 255) / 256;
  float* u = (float*)workspace;
  float* v = u + (size_t)pairs * n;
  float* err = v + (size_t)pairs * m;
  float* partial = err + pairs;
  int* done = (int*)(partial + (size_t)pairs * blocks);
  hipError_t e = hipMemsetAsync(workspace, 0, shw_sinkhorn_workspace_bytes(pairs, n, m), st);   // u = v = 0 (:28-29)
  if (e != hipSuccess) return (int)e;
  shw::SinkArgs A{x, y, u, v, err, done, n, m, eps, 1.f / eps, norm_p, cost_pow};
  const dim3 grid_u((n + 255) / 256, pairs), grid_v((m + 255) / 256, pairs);
  const bool fast = norm_p == 2 && cost_pow == 1;
  for (int it = 0; it < max_iter; ++it) {
    if (fast) {
      hipLaunchKernelGGL((shw::sinkhorn_pass_kernel<false, true>), grid_u, dim3(256), 0, st, A);
      hipLaunchKernelGGL((shw::sinkhorn_pass_kernel<true, true>), grid_v, dim3(256), 0, st, A);
    } else {
      hipLaunchKernelGGL((shw::sinkhorn_pass_kernel<false, false>), grid_u, dim3(256), 0, st, A);
      hipLaunchKernelGGL((shw::sinkhorn_pass_kernel<true, false>), grid_v, dim3(256), 0, st, A);
    }
    hipLaunchKernelGGL(shw::sinkhorn_check_kernel, dim3(1), dim3(64), 0, st, err, done, pairs, thresh);
  }
  hipLaunchKernelGGL(shw::sinkhorn_cost_kernel, grid_u, dim3(256), 0, st, A, partial, plan, cost_matrix);
  hipLaunchKernelGGL(shw::sinkhorn_cost_reduce_kernel, dim3(pairs), dim3(64), 0, st, partial, (int)blocks, cost);
  return (int)hipGetLastError();
}

}  // extern "C"
