// shw_capi.hip -- the extern "C" boundary (include/shw.h): argument validation, dispatch, and the
// small deterministic reduction kernels.
#include "ssw_common.hpp"

namespace shw {

// ---------------------------------------------------------------------------------------------
// reductions: per-pair scaled sum over slices, then total over pairs.  Fixed order, no atomics:
// one wavefront per pair, lane j adds slices j, j+64, ... in order, then a butterfly over the lanes.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float pair_sum(const float* __restrict__ row, int slices, int lane) {
  float acc = 0.f;
  int l = lane;
  for (; l + 192 < slices; l += 256) {             // four independent loads in flight
    const float a = row[l], b = row[l + 64], c = row[l + 128], d = row[l + 192];
    acc += a; acc += b; acc += c; acc += d;
  }
  for (; l < slices; l += 64) acc += row[l];
  return wave_sum(acc, lane);
}

__global__ __launch_bounds__(256) void ssw_reduce_pairs_kernel(const float* __restrict__ slice_cost, int pairs,
                                                               int slices, float scale,
                                                               float* __restrict__ pair_loss) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= pairs) return;
  const float v = pair_sum(slice_cost + (long)b * slices, slices, lane) * scale;
  if (lane == 0) pair_loss[b] = v;
}

__global__ __launch_bounds__(64) void ssw_reduce_total_kernel(const float* __restrict__ pair_loss, int pairs,
                                                              float* __restrict__ total) {
  const int lane = threadIdx.x;
  float acc = 0.f;
  for (int b = lane; b < pairs; b += 64) acc += pair_loss[b];
  acc = wave_sum(acc, lane);
  if (lane == 0) {
    total[0] = acc;
    total[1] = acc / (float)pairs;
  }
}

// Both reductions in ONE launch for small batches (pairs <= 256): the 16 waves of a single workgroup
// take pairs w, w+16, ...; after the barrier wave 0 adds the pair losses exactly like
// ssw_reduce_total_kernel.  Same arithmetic order as the two-kernel form, one kernel boundary less.
__global__ __launch_bounds__(1024) void ssw_reduce_fused_kernel(const float* __restrict__ slice_cost, int pairs,
                                                                int slices, float scale,
                                                                float* __restrict__ pair_loss,
                                                                float* __restrict__ total) {
  __shared__ float pl[256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // four pairs of a wave at a time, their loads interleaved (one pair after the other is a chain of ~8 load latencies
  // at 512 slices: the kernel took 5 us); each pair's lane sums run over its slices in the order of pair_sum
  for (int b0 = wave; b0 < pairs; b0 += 64) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const float* row[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) row[q] = slice_cost + (long)min(b0 + 16 * q, pairs - 1) * slices;
    int l = lane;
    for (; l + 64 < slices; l += 128) {
      float v[2][4];
#pragma unroll
      for (int q = 0; q < 4; ++q) { v[0][q] = row[q][l]; v[1][q] = row[q][l + 64]; }
#pragma unroll
      for (int q = 0; q < 4; ++q) { acc[q] += v[0][q]; acc[q] += v[1][q]; }
    }
    for (; l < slices; l += 64) {
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] += row[q][l];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int b = b0 + 16 * q;
      const float v = wave_sum(acc[q], lane) * scale;
      if (lane == 0 && b < pairs) { pair_loss[b] = v; pl[b] = v; }
    }
  }
  __syncthreads();
  if (wave == 0 && total) {
    float acc = 0.f;
    for (int b = lane; b < pairs; b += 64) acc += pl[b];
    acc = wave_sum(acc, lane);
    if (lane == 0) {
      total[0] = acc;
      total[1] = acc / (float)pairs;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Orthonormal 2-frames from Gaussian 3x2 matrices: the reduced QR of max_spherical_sliced_w.py:307-308
// (`U, _ = torch.linalg.qr(Z)`).  One thread per matrix, following LAPACK's sgeqr2 + sorg2r step by step
// (Householder reflectors, beta = -sign(alpha) * norm), so the frames agree with torch's CPU result in sign
// and to fp32 rounding.  torch.linalg.qr on the device takes ~0.94 s for the 32 768 frames of config 3
// (batched rocSOLVER on tiny matrices; measured) -- 2 700x the loss kernel -- hence this kernel.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void householder(float alpha, float x1, float x2, bool two, float& beta, float& tau,
                                            float& v1, float& v2) {
  const float xnorm = two ? sqrtf(x1 * x1 + x2 * x2) : fabsf(x1);
  if (xnorm == 0.f) { beta = alpha; tau = 0.f; v1 = 0.f; v2 = 0.f; return; }
  beta = -copysignf(sqrtf(alpha * alpha + xnorm * xnorm), alpha);
  tau = (beta - alpha) / beta;
  const float scale = 1.f / (alpha - beta);
  v1 = x1 * scale;
  v2 = two ? x2 * scale : 0.f;
}

__global__ __launch_bounds__(256) void stiefel_frames_kernel(const float* __restrict__ z, int count,
                                                             float* __restrict__ u) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= count) return;
  const float* Z = z + (long)i * 6;                    // (3,2) row-major: Z[2*d + k]
  float a11 = Z[0], a12 = Z[1], a21 = Z[2], a22 = Z[3], a31 = Z[4], a32 = Z[5];
  float beta1, tau1, v1, v2;
  householder(a11, a21, a31, true, beta1, tau1, v1, v2);
  {                                                    // H1 applied to the second column
    const float w = a12 + v1 * a22 + v2 * a32;
    a12 -= tau1 * w;
    a22 -= tau1 * w * v1;
    a32 -= tau1 * w * v2;
  }
  float beta2, tau2, w1, unused;
  householder(a22, a32, 0.f, false, beta2, tau2, w1, unused);
  // sorg2r: Q = H1 H2 [e1 e2]
  float q2x = 0.f, q2y = 1.f - tau2, q2z = -tau2 * w1;
  {
    const float w = q2x + v1 * q2y + v2 * q2z;
    q2x -= tau1 * w;
    q2y -= tau1 * w * v1;
    q2z -= tau1 * w * v2;
  }
  float* U = u + (long)i * 6;
  U[0] = 1.f - tau1; U[1] = q2x;
  U[2] = -tau1 * v1; U[3] = q2y;
  U[4] = -tau1 * v2; U[5] = q2z;
}

}  // namespace shw

extern "C" {

int shw_stiefel_frames(const float* z, long count, float* u, void* stream) {
  if (!z || !u || count < 0 || count > 0x7fffffffL) return (int)hipErrorInvalidValue;
  if (count == 0) return 0;
  hipLaunchKernelGGL(shw::stiefel_frames_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, z, (int)count, u);
  return (int)hipGetLastError();
}

int shw_abi_version(void) { return SHW_ABI_VERSION; }
int shw_max_points(void) { return SHW_MAX_POINTS; }

int shw_ssw_forward(const float* xs, const float* xt, const float* dirs, int pairs, int n, int m, int slices,
                    long u_pair_stride, float p, float* slice_cost, int32_t* slice_shift, void* stream) {
  if (!xs || !xt || !dirs || !slice_cost) return (int)hipErrorInvalidValue;
  if (pairs < 0 || slices < 0 || n < 1 || m < 1 || n > SHW_MAX_POINTS || m > SHW_MAX_POINTS) return (int)hipErrorInvalidValue;
  if (!(p >= 1.f)) return (int)hipErrorInvalidValue;
  if (u_pair_stride != 0 && u_pair_stride < (long)slices * 6) return (int)hipErrorInvalidValue;
  if (p != 1.f && n != m) return (int)hipErrorInvalidValue;       // n != m with p != 1: shw_ssw_forward_general
  if (pairs == 0 || slices == 0) return 0;
  shw::SswArgs A{};
  A.xs = xs; A.xt = xt; A.dirs = dirs; A.slice_cost = slice_cost; A.slice_shift = slice_shift;
  A.pairs = pairs; A.n = n; A.m = m; A.slices = slices; A.u_pair_stride = u_pair_stride; A.pstride = 3;
  A.p = p; A.p_int = shw::small_integer_power(p);
  if (p == 1.f) return shw::dispatch_level_median(A, (hipStream_t)stream);
  return shw::dispatch_forward(A, (hipStream_t)stream);
}

int shw_ssw_reduce(const float* slice_cost, int pairs, int slices, float scale, float* pair_loss, float* total,
                   void* stream) {
  if (!slice_cost || !pair_loss || pairs < 0 || slices < 0) return (int)hipErrorInvalidValue;
  if (pairs == 0) return 0;
  if (pairs <= 256) {
    hipLaunchKernelGGL(shw::ssw_reduce_fused_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, slice_cost, pairs,
                       slices, scale, pair_loss, total);
    return (int)hipGetLastError();
  }
  hipLaunchKernelGGL(shw::ssw_reduce_pairs_kernel, dim3((pairs + 3) / 4), dim3(256), 0, (hipStream_t)stream, slice_cost,
                     pairs, slices, scale, pair_loss);
  int rc = (int)hipGetLastError();
  if (rc) return rc;
  if (total) {
    hipLaunchKernelGGL(shw::ssw_reduce_total_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, pair_loss, pairs, total);
    rc = (int)hipGetLastError();
  }
  return rc;
}

size_t shw_ssw_coef_bytes(int pairs, int n, int m, int slices) {
  if (pairs < 0 || n < 0 || m < 0 || slices < 0) return 0;
  return (size_t)pairs * (size_t)slices * ((size_t)n + (size_t)m) * sizeof(float);
}

int shw_ssw_forward_grad(const float* xs, const float* xt, const float* dirs, int pairs, int n, int m, int slices,
                         long u_pair_stride, float p, float* slice_cost, int32_t* slice_shift, float* coef_s,
                         float* coef_t, void* stream) {
  if (!xs || !xt || !dirs || !slice_cost || !coef_s || !coef_t) return (int)hipErrorInvalidValue;
  if (pairs < 0 || slices < 0 || n < 1 || m < 1 || n > SHW_MAX_POINTS || m > SHW_MAX_POINTS) return (int)hipErrorInvalidValue;
  if (!(p >= 1.f)) return (int)hipErrorInvalidValue;
  if (u_pair_stride != 0 && u_pair_stride < (long)slices * 6) return (int)hipErrorInvalidValue;
  if (p != 1.f && n != m) return (int)hipErrorInvalidValue;
  if (pairs == 0 || slices == 0) return 0;
  shw::SswArgs A{};
  A.xs = xs; A.xt = xt; A.dirs = dirs; A.slice_cost = slice_cost; A.slice_shift = slice_shift;
  A.coef_s = coef_s; A.coef_t = coef_t;
  A.pairs = pairs; A.n = n; A.m = m; A.slices = slices; A.u_pair_stride = u_pair_stride; A.pstride = 3;
  A.p = p; A.p_int = shw::small_integer_power(p);
  if (p == 1.f) return shw::dispatch_level_median(A, (hipStream_t)stream);
  return shw::dispatch_forward_grad(A, (hipStream_t)stream);
}

int shw_ssw_forward_general(const float* xs, const float* xt, const float* dirs, const float* wu, const float* wv,
                            long wu_pair_stride, long wv_pair_stride, int pairs, int n, int m, int slices,
                            long u_pair_stride, float p, float* slice_cost, float* slice_theta, float* coef_s,
                            float* coef_t, void* stream) {
  if (!xs || !xt || !dirs || !slice_cost) return (int)hipErrorInvalidValue;
  if ((coef_s == nullptr) != (coef_t == nullptr)) return (int)hipErrorInvalidValue;
  if (pairs < 0 || slices < 0 || n < 1 || m < 1 || n > 4096 || m > 4096) return (int)hipErrorInvalidValue;
  if (!(p >= 1.f)) return (int)hipErrorInvalidValue;               // p == 1: weighted level-median kernel
  if (u_pair_stride != 0 && u_pair_stride < (long)slices * 6) return (int)hipErrorInvalidValue;
  if ((wu_pair_stride != 0 && wu_pair_stride < n) || (wv_pair_stride != 0 && wv_pair_stride < m)) return (int)hipErrorInvalidValue;
  if (pairs == 0 || slices == 0) return 0;
  shw::SswArgs A{};
  A.xs = xs; A.xt = xt; A.dirs = dirs; A.slice_cost = slice_cost; A.slice_shift = nullptr;
  A.coef_s = coef_s; A.coef_t = coef_t;
  A.pairs = pairs; A.n = n; A.m = m; A.slices = slices; A.u_pair_stride = u_pair_stride; A.pstride = 3;
  A.p = p; A.p_int = shw::small_integer_power(p);
  return shw::dispatch_general(A, wu, wv, wu_pair_stride, wv_pair_stride, slice_theta, (hipStream_t)stream);
}

int shw_circle_ot(const float* u, const float* v, const float* wu, const float* wv, long wu_row_stride,
                  long wv_row_stride, int rows, int n, int m, float p, int method, float* cost, float* aux, float* grad_u,
                  float* grad_v, void* stream) {
  if (!u || !v || !cost) return (int)hipErrorInvalidValue;
  if (method != SHW_CIRCLE_AS_SLICED && method != SHW_CIRCLE_BISECTION && method != SHW_CIRCLE_LEVEL_MEDIAN) return (int)hipErrorInvalidValue;
  if (method == SHW_CIRCLE_LEVEL_MEDIAN && p != 1.f) return (int)hipErrorInvalidValue;   // emd1D_circle has no p != 1 branch
  if ((grad_u == nullptr) != (grad_v == nullptr)) return (int)hipErrorInvalidValue;
  if (rows < 0 || n < 1 || m < 1 || !(p >= 1.f)) return (int)hipErrorInvalidValue;
  if (rows == 0) return 0;
  const bool bisect = p != 1.f || method == SHW_CIRCLE_BISECTION;
  const bool general = wu || wv || (bisect && n != m);
  const int limit = general ? 4096 : SHW_MAX_POINTS;
  if (n > limit || m > limit) return (int)hipErrorInvalidValue;
  if ((wu_row_stride != 0 && wu_row_stride < n) || (wv_row_stride != 0 && wv_row_stride < m)) return (int)hipErrorInvalidValue;
  // a row is a "pair" with ONE slice whose atoms already are circle coordinates (dirs = NULL, one float per atom)
  shw::SswArgs A{};
  A.xs = u; A.xt = v; A.dirs = nullptr; A.slice_cost = cost; A.slice_shift = nullptr;
  A.coef_s = grad_u; A.coef_t = grad_v;
  A.pairs = rows; A.n = n; A.m = m; A.slices = 1; A.u_pair_stride = 0; A.pstride = 1;
  A.p = p; A.p_int = shw::small_integer_power(p);
  A.bisect_p1 = (p == 1.f && bisect) ? 1 : 0;
  if (general) return shw::dispatch_general(A, wu, wv, wu_row_stride, wv_row_stride, aux, (hipStream_t)stream);
  A.slice_shift = reinterpret_cast<int32_t*>(aux);
  if (!bisect) return shw::dispatch_level_median(A, (hipStream_t)stream);
  return grad_u ? shw::dispatch_forward_grad(A, (hipStream_t)stream) : shw::dispatch_forward(A, (hipStream_t)stream);
}

int shw_ssw_backward_points(const float* xs, const float* xt, const float* dirs, const float* coef_s,
                            const float* coef_t, int pairs, int n, int m, int slices, long u_pair_stride, float scale,
                            const float* pair_w, const float* total_w, float* grad_xs, float* grad_xt, void* stream) {
  if (!xs || !xt || !dirs || !coef_s || !coef_t || !grad_xs || !grad_xt) return (int)hipErrorInvalidValue;
  if (pairs < 0 || slices < 0 || n < 1 || m < 1) return (int)hipErrorInvalidValue;
  if (pairs == 0) return 0;
  return shw::launch_backward_points(xs, xt, dirs, coef_s, coef_t, pairs, n, m, slices, u_pair_stride, scale, pair_w,
                                     total_w, grad_xs, grad_xt, (hipStream_t)stream);
}

}  // extern "C"
