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
// transcendental bound.  Round 2: the backward through the unrolled iterations (shw_sinkhorn_forward_train /
// shw_sinkhorn_backward below).
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
  float* u;           // (pairs, n) dual read by the u-pass ("old" u) -- in place: the same buffer as u_out
  float* v;           // (pairs, m) dual read by both passes ("old" v)
  float* u_out;       // where the u-pass writes (training: the next slot of the trajectory)
  float* v_out;       // where the v-pass writes
  float* err;         // (pairs): sum_i |u_new - u_old| of the current iteration
  int* done;          // [0]: convergence flag; [1]: sweeps executed (training)
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
  const float* dual_in = (TRANSPOSE ? A.v : A.u) + (long)b * rows;
  float* dual_r = (TRANSPOSE ? A.v_out : A.u_out) + (long)b * rows;
  const float* dual_c = (TRANSPOSE ? A.u_out : A.v) + (long)b * cands;      // the v-pass sees the fresh u
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int ic = min(i, rows - 1);
  const float rx = R[3 * ic], ry = R[3 * ic + 1], rz = R[3 * ic + 2];
  const float old = dual_in[ic];
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
  if (threadIdx.x == 0) done[1] += 1;                        // one more sweep executed
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
  const float* U = A.u_out + (long)b * A.n;
  const float* V = A.v_out + (long)b * A.m;
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


// the value from the last executed slot of the trajectory (training forward)
__global__ __launch_bounds__(256) void sinkhorn_cost_traj_kernel(const float* x, const float* y, const float* tu,
                                                                 const float* tv, const int* done, int pairs, int n, int m,
                                                                 float inv_eps, int norm_p, int cost_pow,
                                                                 float* __restrict__ partial, float* __restrict__ P,
                                                                 float* __restrict__ Cm) {
  __shared__ float4 tile[kSkTile];
  __shared__ float red[4];
  const int b = blockIdx.y;
  const int T = done[1];
  const float* X = x + (long)b * n * 3;
  const float* Y = y + (long)b * m * 3;
  const float* U = tu + ((long)T * pairs + b) * n;
  const float* V = tv + ((long)T * pairs + b) * m;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int ic = min(i, n - 1);
  const float rx = X[3 * ic], ry = X[3 * ic + 1], rz = X[3 * ic + 2], ui = U[ic];
  float acc = 0.f;
  for (int base = 0; base < m; base += kSkTile) {
    const int cnt = min(kSkTile, m - base);
    __syncthreads();
    for (int t = threadIdx.x; t < cnt; t += 256) {
      const int j = base + t;
      tile[t] = make_float4(Y[3 * j], Y[3 * j + 1], Y[3 * j + 2], V[j]);
    }
    __syncthreads();
    for (int t = 0; t < cnt; ++t) {
      const float4 q = tile[t];
      const float c = pair_cost<false>(rx - q.x, ry - q.y, rz - q.z, norm_p, cost_pow);
      const float p = expf(((ui - c) + q.w) * inv_eps);
      if (i < n) {
        acc = fmaf(p, c, acc);
        // the dense plan and cost matrix the reference returns (sinkhorn.py:52-58), on request, from the SAME solve
        if (P) P[((long)b * n + i) * m + base + t] = p;
        if (Cm) Cm[((long)b * n + i) * m + base + t] = c;
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


// ---------------------------------------------------------------------------------------------
// Backward through the unrolled iterations (the reference's forward is differentiable, sinkhorn.py:35-49).
// With la = log(a + 1e-8), lb = log(b + 1e-8) the updates are  u_t = eps la - eps LSE_j((-C_ij + v_{t-1,j})/eps)  and
// v_t = eps lb - eps LSE_i((-C_ij + u_{t,i})/eps)  (the "+ u" of the reference's line cancels the u inside M), so
//   d v_{t,j} / d u_{t,i} = -w_ij ,   d v_{t,j} / d C_ij = +w_ij ,    w_ij  = exp(M_ij(u_t, v_t))     / (b + 1e-8)
//   d u_{t,i} / d v_{t-1,j} = -w'_ij, d u_{t,i} / d C_ij = +w'_ij ,   w'_ij = exp(M_ij(u_t, v_{t-1})) / (a + 1e-8)
// and cost = sum_ij P_ij C_ij, P = exp(M(u_T, v_T)):  d cost / d C_ij = P_ij (1 - C_ij/eps),  d cost / d u_i = sum_j P_ij C_ij / eps.
// The forward keeps the trajectory (u_t, v_t), t = 0..T (two (T+1) x pairs x points arrays); nothing dense is stored.
// Per iteration, from t = T down to 1, two kernels: rows = x (adjoint of u_t completed, gradient of x accumulated),
// then rows = y (adjoint of v_{t-1}, gradient of y).  Every row is owned by one thread: no atomics, deterministic.
// ---------------------------------------------------------------------------------------------
struct SinkBwdArgs {
  const float* x;
  const float* y;
  const float* traj_u;      // (T+1, pairs, n), slot 0 = zeros
  const float* traj_v;      // (T+1, pairs, m)
  const int* done;          // [1] = sweeps executed
  float* ubar;              // (pairs, n) adjoint of u_t
  float* vbar;              // (pairs, m) adjoint of v_t
  float* gx;                // (pairs, n, 3)
  float* gy;                // (pairs, m, 3)
  const float* gcost;       // (pairs) upstream gradient of the per-pair cost
  int pairs, n, m;
  float eps, inv_eps;
  int norm_p, cost_pow;
};

// cost c and its derivative w.r.t. the difference vector d = row - candidate
template <bool FAST>
__device__ __forceinline__ float pair_cost_grad(float dx, float dy, float dz, int norm_p, int cost_pow, float& gx, float& gy,
                                                float& gz) {
  if constexpr (FAST) {
    gx = 2.f * dx; gy = 2.f * dy; gz = 2.f * dz;
    return fmaf(dz, dz, fmaf(dy, dy, dx * dx));
  }
  float c0, hx, hy, hz;
  if (norm_p == 2) { c0 = fmaf(dz, dz, fmaf(dy, dy, dx * dx)); hx = 2.f * dx; hy = 2.f * dy; hz = 2.f * dz; }
  else if (norm_p == 1) {
    c0 = fabsf(dx) + fabsf(dy) + fabsf(dz);
    hx = dx > 0.f ? 1.f : (dx < 0.f ? -1.f : 0.f); hy = dy > 0.f ? 1.f : (dy < 0.f ? -1.f : 0.f); hz = dz > 0.f ? 1.f : (dz < 0.f ? -1.f : 0.f);
  } else {
    const float p = (float)norm_p;
    c0 = powf(fabsf(dx), p) + powf(fabsf(dy), p) + powf(fabsf(dz), p);
    hx = copysignf(p * powf(fabsf(dx), p - 1.f), dx); hy = copysignf(p * powf(fabsf(dy), p - 1.f), dy);
    hz = copysignf(p * powf(fabsf(dz), p - 1.f), dz);
  }
  float r = c0, rm1 = 1.f;
  for (int k = 1; k < cost_pow; ++k) { rm1 = r; r *= c0; }
  const float outer = (float)cost_pow * rm1;                 // d c0^N / d c0
  gx = outer * hx; gy = outer * hy; gz = outer * hz;
  return r;
}

// gradient of the final cost: ROWS_X: rows = x (writes ubar, gx); else rows = y (writes vbar, gy)
template <bool ROWS_X, bool FAST>
__global__ __launch_bounds__(256) void sinkhorn_bwd_cost_kernel(SinkBwdArgs A) {
  __shared__ float4 tile[kSkTile];
  const int b = blockIdx.y;
  const int T = A.done[1];
  const int rows = ROWS_X ? A.n : A.m, cands = ROWS_X ? A.m : A.n;
  const float* R = (ROWS_X ? A.x : A.y) + (long)b * rows * 3;
  const float* Cn = (ROWS_X ? A.y : A.x) + (long)b * cands * 3;
  const float* dr = (ROWS_X ? A.traj_u + ((long)T * A.pairs + b) * A.n : A.traj_v + ((long)T * A.pairs + b) * A.m);
  const float* dc = (ROWS_X ? A.traj_v + ((long)T * A.pairs + b) * A.m : A.traj_u + ((long)T * A.pairs + b) * A.n);
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int ic = min(i, rows - 1);
  const float rx = R[3 * ic], ry = R[3 * ic + 1], rz = R[3 * ic + 2], own = dr[ic];
  float acc = 0.f, ax = 0.f, ay = 0.f, az = 0.f;
  for (int base = 0; base < cands; base += kSkTile) {
    const int cnt = min(kSkTile, cands - base);
    __syncthreads();
    for (int t = threadIdx.x; t < cnt; t += 256) {
      const int j = base + t;
      tile[t] = make_float4(Cn[3 * j], Cn[3 * j + 1], Cn[3 * j + 2], dc[j]);
    }
    __syncthreads();
    for (int t = 0; t < cnt; ++t) {
      const float4 q = tile[t];
      float hx, hy, hz;
      const float c = pair_cost_grad<FAST>(rx - q.x, ry - q.y, rz - q.z, A.norm_p, A.cost_pow, hx, hy, hz);
      const float P = expf(((own - c) + q.w) * A.inv_eps);
      acc = fmaf(P, c, acc);
      const float k = P * (1.f - c * A.inv_eps);
      ax = fmaf(k, hx, ax); ay = fmaf(k, hy, ay); az = fmaf(k, hz, az);
    }
  }
  if (i < rows) {
    const float g = A.gcost[b];
    (ROWS_X ? A.ubar : A.vbar)[(long)b * rows + i] = g * acc * A.inv_eps;
    float* G = (ROWS_X ? A.gx : A.gy) + ((long)b * rows + i) * 3;
    G[0] = g * ax; G[1] = g * ay; G[2] = g * az;             // (d = row - candidate: the derivative w.r.t. the row point)
  }
}

// iteration t, rows = x:  s_i = sum_j vbar_j w_ij ;  ubar_i -= s_i ;  gx_i += sum_j vbar_j w_ij dC + ubar_i sum_j w'_ij dC
template <bool FAST>
__global__ __launch_bounds__(256) void sinkhorn_bwd_x_kernel(SinkBwdArgs A, int t) {
  __shared__ float4 tile[kSkTile];
  __shared__ float2 tile2[kSkTile];
  if (t > A.done[1]) return;                                 // that sweep was never executed (converged earlier)
  const int b = blockIdx.y;
  const float* X = A.x + (long)b * A.n * 3;
  const float* Y = A.y + (long)b * A.m * 3;
  const float* ut = A.traj_u + ((long)t * A.pairs + b) * A.n;
  const float* vt = A.traj_v + ((long)t * A.pairs + b) * A.m;
  const float* vp = A.traj_v + ((long)(t - 1) * A.pairs + b) * A.m;
  const float* vbar = A.vbar + (long)b * A.m;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int ic = min(i, A.n - 1);
  const float rx = X[3 * ic], ry = X[3 * ic + 1], rz = X[3 * ic + 2], ui = ut[ic];
  const float inv_a = 1.f / (1.f / (float)A.n + 1e-8f), inv_b = 1.f / (1.f / (float)A.m + 1e-8f);
  float s = 0.f, ax = 0.f, ay = 0.f, az = 0.f, bx = 0.f, by = 0.f, bz = 0.f;
  for (int base = 0; base < A.m; base += kSkTile) {
    const int cnt = min(kSkTile, A.m - base);
    __syncthreads();
    for (int q = threadIdx.x; q < cnt; q += 256) {
      const int j = base + q;
      tile[q] = make_float4(Y[3 * j], Y[3 * j + 1], Y[3 * j + 2], vt[j]);
      tile2[q] = make_float2(vp[j], vbar[j]);
    }
    __syncthreads();
    for (int q = 0; q < cnt; ++q) {
      const float4 c4 = tile[q];
      const float2 c2 = tile2[q];
      float hx, hy, hz;
      const float c = pair_cost_grad<FAST>(rx - c4.x, ry - c4.y, rz - c4.z, A.norm_p, A.cost_pow, hx, hy, hz);
      const float w = expf(((ui - c) + c4.w) * A.inv_eps) * inv_b;      // v_t from u_t
      const float wp = expf(((ui - c) + c2.x) * A.inv_eps) * inv_a;     // u_t from v_{t-1}
      const float k = c2.y * w;
      s += k;
      ax = fmaf(k, hx, ax); ay = fmaf(k, hy, ay); az = fmaf(k, hz, az);
      bx = fmaf(wp, hx, bx); by = fmaf(wp, hy, by); bz = fmaf(wp, hz, bz);
    }
  }
  if (i < A.n) {
    float* ub = A.ubar + (long)b * A.n + i;
    // u_t feeds v_t only (and, for the last executed sweep, the cost): its adjoint starts from the cost's contribution
    // at t = T and from zero otherwise -- what this slot held before (the adjoint of u_{t+1}) is consumed
    const float ubar = (t == A.done[1] ? *ub : 0.f) - s;
    *ub = ubar;
    float* G = A.gx + ((long)b * A.n + i) * 3;
    G[0] += fmaf(ubar, bx, ax); G[1] += fmaf(ubar, by, ay); G[2] += fmaf(ubar, bz, az);
  }
}

// iteration t, rows = y:  gy_j += vbar_j sum_i w_ij dC/dy + sum_i ubar_i w'_ij dC/dy ;  vbar_j <- -sum_i ubar_i w'_ij
template <bool FAST>
__global__ __launch_bounds__(256) void sinkhorn_bwd_y_kernel(SinkBwdArgs A, int t) {
  __shared__ float4 tile[kSkTile];
  __shared__ float tile1[kSkTile];
  if (t > A.done[1]) return;
  const int b = blockIdx.y;
  const float* X = A.x + (long)b * A.n * 3;
  const float* Y = A.y + (long)b * A.m * 3;
  const float* ut = A.traj_u + ((long)t * A.pairs + b) * A.n;
  const float* vt = A.traj_v + ((long)t * A.pairs + b) * A.m;
  const float* vp = A.traj_v + ((long)(t - 1) * A.pairs + b) * A.m;
  const float* ubar = A.ubar + (long)b * A.n;
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int jc = min(j, A.m - 1);
  const float rx = Y[3 * jc], ry = Y[3 * jc + 1], rz = Y[3 * jc + 2], vtj = vt[jc], vpj = vp[jc];
  const float inv_a = 1.f / (1.f / (float)A.n + 1e-8f), inv_b = 1.f / (1.f / (float)A.m + 1e-8f);
  float s = 0.f, ax = 0.f, ay = 0.f, az = 0.f, bx = 0.f, by = 0.f, bz = 0.f;
  for (int base = 0; base < A.n; base += kSkTile) {
    const int cnt = min(kSkTile, A.n - base);
    __syncthreads();
    for (int q = threadIdx.x; q < cnt; q += 256) {
      const int i = base + q;
      tile[q] = make_float4(X[3 * i], X[3 * i + 1], X[3 * i + 2], ut[i]);
      tile1[q] = ubar[i];
    }
    __syncthreads();
    for (int q = 0; q < cnt; ++q) {
      const float4 c4 = tile[q];
      const float ub = tile1[q];
      float hx, hy, hz;                                         // derivative w.r.t. x - y: negate for y
      const float c = pair_cost_grad<FAST>(c4.x - rx, c4.y - ry, c4.z - rz, A.norm_p, A.cost_pow, hx, hy, hz);
      const float w = expf(((c4.w - c) + vtj) * A.inv_eps) * inv_b;
      const float wp = expf(((c4.w - c) + vpj) * A.inv_eps) * inv_a;
      const float k = ub * wp;
      s += k;
      ax = fmaf(w, hx, ax); ay = fmaf(w, hy, ay); az = fmaf(w, hz, az);
      bx = fmaf(k, hx, bx); by = fmaf(k, hy, by); bz = fmaf(k, hz, bz);
    }
  }
  if (j < A.m) {
    float* vb = A.vbar + (long)b * A.m + j;
    const float vbar_t = *vb;
    float* G = A.gy + ((long)b * A.m + j) * 3;
    G[0] -= fmaf(vbar_t, ax, bx); G[1] -= fmaf(vbar_t, ay, by); G[2] -= fmaf(vbar_t, az, bz);
    *vb = -s;                                                  // adjoint of v_{t-1}
  }
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
  shw::SinkArgs A{x, y, u, v, u, v, err, done, n, m, eps, 1.f / eps, norm_p, cost_pow};
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

size_t shw_sinkhorn_train_workspace_bytes(int pairs, int n, int m, int max_iter) {
  if (pairs < 0 || n < 1 || m < 1 || max_iter < 0) return 0;
  const size_t blocks = (size_t)(n + 255) / 256;
  // trajectory (max_iter + 1 slots of u and v), adjoints ubar / vbar, err, partial sums, flags
  return ((size_t)(max_iter + 2) * pairs * ((size_t)n + (size_t)m) + (size_t)pairs * (1 + blocks)) * sizeof(float) + 16;
}

static void sink_train_layout(void* workspace, int pairs, int n, int m, int max_iter, float*& tu, float*& tv, float*& ubar,
                              float*& vbar, float*& err, float*& partial, int*& done) {
  const size_t blocks = (size_t)(n + 255) / 256;
  tu = (float*)workspace;
  tv = tu + (size_t)(max_iter + 1) * pairs * n;
  ubar = tv + (size_t)(max_iter + 1) * pairs * m;
  vbar = ubar + (size_t)pairs * n;
  err = vbar + (size_t)pairs * m;
  partial = err + pairs;
  done = (int*)(partial + (size_t)pairs * blocks);
}

int shw_sinkhorn_forward_train(const float* x, const float* y, int pairs, int n, int m, float eps, int max_iter,
                               int norm_p, int cost_pow, float thresh, void* workspace, float* cost, float* plan,
                               float* cost_matrix, void* stream) {
  if (!x || !y || !workspace || !cost) return (int)hipErrorInvalidValue;
  if (pairs < 0 || pairs > 65535 || n < 1 || m < 1 || !(eps > 0.f) || max_iter < 0 || norm_p < 1 || cost_pow < 1)
    return (int)hipErrorInvalidValue;
  if (pairs == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const size_t blocks = (size_t)(n + 255) / 256;
  float *tu, *tv, *ubar, *vbar, *err, *partial;
  int* done;
  sink_train_layout(workspace, pairs, n, m, max_iter, tu, tv, ubar, vbar, err, partial, done);
  // slot 0 of both trajectories = 0 (u = v = 0, :28-29); statistics and flags cleared
  hipError_t e = hipMemsetAsync(tu, 0, (size_t)pairs * n * sizeof(float), st);
  if (e == hipSuccess) e = hipMemsetAsync(tv, 0, (size_t)pairs * m * sizeof(float), st);
  if (e == hipSuccess) e = hipMemsetAsync(err, 0, (size_t)pairs * (1 + blocks) * sizeof(float) + 16, st);
  if (e != hipSuccess) return (int)e;
  const dim3 grid_u((n + 255) / 256, pairs), grid_v((m + 255) / 256, pairs);
  const bool fast = norm_p == 2 && cost_pow == 1;
  const size_t su = (size_t)pairs * n, sv = (size_t)pairs * m;
  for (int it = 1; it <= max_iter; ++it) {
    shw::SinkArgs A{x, y, tu + (it - 1) * su, tv + (it - 1) * sv, tu + it * su, tv + it * sv, err, done, n, m, eps, 1.f / eps,
                    norm_p, cost_pow};
    if (fast) {
      hipLaunchKernelGGL((shw::sinkhorn_pass_kernel<false, true>), grid_u, dim3(256), 0, st, A);
      hipLaunchKernelGGL((shw::sinkhorn_pass_kernel<true, true>), grid_v, dim3(256), 0, st, A);
    } else {
      hipLaunchKernelGGL((shw::sinkhorn_pass_kernel<false, false>), grid_u, dim3(256), 0, st, A);
      hipLaunchKernelGGL((shw::sinkhorn_pass_kernel<true, false>), grid_v, dim3(256), 0, st, A);
    }
    hipLaunchKernelGGL(shw::sinkhorn_check_kernel, dim3(1), dim3(64), 0, st, err, done, pairs, thresh);
  }
  // the value: cost kernel on the last EXECUTED slot (device-side count)
  hipLaunchKernelGGL(shw::sinkhorn_cost_traj_kernel, grid_u, dim3(256), 0, st, x, y, tu, tv, done, pairs, n, m, 1.f / eps,
                     norm_p, cost_pow, partial, plan, cost_matrix);
  hipLaunchKernelGGL(shw::sinkhorn_cost_reduce_kernel, dim3(pairs), dim3(64), 0, st, partial, (int)blocks, cost);
  return (int)hipGetLastError();
}

int shw_sinkhorn_backward(const float* x, const float* y, int pairs, int n, int m, float eps, int max_iter, int norm_p,
                          int cost_pow, void* workspace, const float* grad_cost, float* grad_x, float* grad_y,
                          void* stream) {
  if (!x || !y || !workspace || !grad_cost || !grad_x || !grad_y) return (int)hipErrorInvalidValue;
  if (pairs < 0 || pairs > 65535 || n < 1 || m < 1 || !(eps > 0.f) || max_iter < 0 || norm_p < 1 || cost_pow < 1)
    return (int)hipErrorInvalidValue;
  if (pairs == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  float *tu, *tv, *ubar, *vbar, *err, *partial;
  int* done;
  sink_train_layout(workspace, pairs, n, m, max_iter, tu, tv, ubar, vbar, err, partial, done);
  shw::SinkBwdArgs A{x, y, tu, tv, done, ubar, vbar, grad_x, grad_y, grad_cost, pairs, n, m, eps, 1.f / eps, norm_p, cost_pow};
  const dim3 grid_x((n + 255) / 256, pairs), grid_y((m + 255) / 256, pairs);
  const bool fast = norm_p == 2 && cost_pow == 1;
  if (fast) {
    hipLaunchKernelGGL((shw::sinkhorn_bwd_cost_kernel<true, true>), grid_x, dim3(256), 0, st, A);
    hipLaunchKernelGGL((shw::sinkhorn_bwd_cost_kernel<false, true>), grid_y, dim3(256), 0, st, A);
  } else {
    hipLaunchKernelGGL((shw::sinkhorn_bwd_cost_kernel<true, false>), grid_x, dim3(256), 0, st, A);
    hipLaunchKernelGGL((shw::sinkhorn_bwd_cost_kernel<false, false>), grid_y, dim3(256), 0, st, A);
  }
  for (int t = max_iter; t >= 1; --t) {
    if (fast) {
      hipLaunchKernelGGL((shw::sinkhorn_bwd_x_kernel<true>), grid_x, dim3(256), 0, st, A, t);
      hipLaunchKernelGGL((shw::sinkhorn_bwd_y_kernel<true>), grid_y, dim3(256), 0, st, A, t);
    } else {
      hipLaunchKernelGGL((shw::sinkhorn_bwd_x_kernel<false>), grid_x, dim3(256), 0, st, A, t);
      hipLaunchKernelGGL((shw::sinkhorn_bwd_y_kernel<false>), grid_y, dim3(256), 0, st, A, t);
    }
  }
  return (int)hipGetLastError();
}

}  // extern "C"
