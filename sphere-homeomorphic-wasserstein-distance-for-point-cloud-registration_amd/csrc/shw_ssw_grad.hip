// shw_ssw_grad.hip -- loss + gradient-coefficient kernel for p != 1 (key+index register sort) and
// the coefficient -> point-gradient streaming kernel.  See ssw_common.hpp.
#include "ssw_common.hpp"

namespace shw {

// ---------------------------------------------------------------------------------------------
// forward + gradient coefficients.  Same flow as ssw_forward_kernel, but both sorts carry the
// original point index (wave_sort_kv), the sorted target indices are parked in LDS next to the
// sorted target coordinates, and after the shift solve every sorted source position e writes
//     coef_s[slice, idx_u(e)]        = +g ,   g = (1/n) d|D|^p/dD ,  D = u_(e) - v_ext(e + k*)
//     coef_t[slice, idx_v(e + k*)]   = -g
// i.e. d cost / d coordinate in ORIGINAL point order.  Each wave writes every
// entry of its 2 x n coefficient rows exactly once (the sort permutations are bijections), so the
// scratch needs no zero fill and the result is deterministic.
// ---------------------------------------------------------------------------------------------
template <int EPT, int WAVES, int PMODE>
__global__ __launch_bounds__(WAVES * 64) void ssw_forward_grad_kernel(SswArgs A) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* vbuf = lds + wave * (2 * EPT * kWave);
  int* vidx = reinterpret_cast<int*>(vbuf + EPT * kWave);

  const int vid = xcd_contiguous_id(blockIdx.x, A.num_groups);
  const int s = vid * WAVES + wave;
  if (s >= A.pairs * A.slices) return;
  const int b = s / A.slices, l = s - b * A.slices;
  const int n = A.n;

  const float* Ul = A.dirs + (long)b * A.u_pair_stride + (long)l * 6;
  float U[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) U[i] = Ul[i];

  item_t item[EPT];
  float sum_v = 0.f, sum_u = 0.f;
#pragma nounroll
  for (int which = 0; which < 2; ++which) {
    const float* X = which == 0 ? A.xt + (long)b * A.m * 3 : A.xs + (long)b * n * 3;
    const int count = which == 0 ? A.m : n;
    int ln = lane;
    asm volatile("" : "+v"(ln));
    float key[EPT];
    const float part = load_coords<EPT>(X, count, ln, U, key);
#pragma unroll
    for (int r = 0; r < EPT; ++r) item[r] = make_item(key[r], r * kWave + ln);
    wave_sort_kv<EPT>(item, ln);
    if (which == 0) {
      sum_v = wave_sum(part, lane);
#pragma unroll
      for (int r = 0; r < EPT; ++r) {
        vbuf[r * kWave + lane] = item_key(item[r]);
        vidx[r * kWave + lane] = item_idx(item[r]);
      }
    } else {
      sum_u = wave_sum(part, lane);
    }
  }
  __builtin_amdgcn_wave_barrier();

  float u[EPT];
#pragma unroll
  for (int r = 0; r < EPT; ++r) u[r] = item_key(item[r]);
  float best;
  const int k = solve_shift<EPT, PMODE>(u, vbuf, lane, n, sum_u, sum_v, A.p, A.p_int, best);
  const float inv_n = 1.f / (float)n;
  if (lane == 0) {
    A.slice_cost[s] = best * inv_n;
    if (A.slice_shift) A.slice_shift[s] = k;
  }
  float* cs = A.coef_s + (long)s * n;
  float* ct = A.coef_t + (long)s * A.m;
#pragma unroll
  for (int r = 0; r < EPT; ++r) {
    const int e = lane * EPT + r;
    if (e < n) {
      int q = e + k;                                 // in [-n, 2n): one turn at most
      float off = 0.f;
      if (q < 0) { q += n; off = -1.f; }
      else if (q >= n) { q -= n; off = 1.f; }
      const int slot = lds_slot<EPT>(q);
      const float d = u[r] - (vbuf[slot] + off);
      const float g = dpow_abs<PMODE>(d, A.p, A.p_int) * inv_n;
      cs[item_idx(item[r])] = g;
      ct[vidx[slot]] = -g;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// coefficient rows -> point gradients.  One thread per (pair, cloud, point) walks the slices:
//   grad[b,i,:] = scale * sum_l coef[b,l,i] * (-bb U_l[:,0] + a U_l[:,1]) / (2 pi (a^2 + bb^2)),
//   (a, bb) = U_l^T x[b,i].
// Streams the coefficient scratch once, coalesced over i (the HBM-bound kernel of this path);
// directions are wave-uniform scalar loads.  Four interleaved partial sums per component keep the
// slice sum's rounding error at the sqrt(L/4) level and the loads in flight.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ssw_backward_points_kernel(const float* __restrict__ xs,
                                                                  const float* __restrict__ xt,
                                                                  const float* __restrict__ dirs,
                                                                  const float* __restrict__ coef_s,
                                                                  const float* __restrict__ coef_t, int n, int m,
                                                                  int slices, long u_pair_stride, float scale,
                                                                  float* __restrict__ grad_xs,
                                                                  float* __restrict__ grad_xt, int chunks_s) {
  const int b = blockIdx.y;
  const bool is_t = (int)blockIdx.x >= chunks_s;
  const int chunk = is_t ? blockIdx.x - chunks_s : blockIdx.x;
  const int cnt = is_t ? m : n;
  const int i = chunk * 256 + threadIdx.x;
  const float* X = (is_t ? xt : xs) + (long)b * cnt * 3;
  const float* C = (is_t ? coef_t : coef_s) + (long)b * slices * cnt;
  float* G = (is_t ? grad_xt : grad_xs) + (long)b * cnt * 3;
  const float* Ub = dirs + (long)b * u_pair_stride;
  const int ic = min(i, cnt - 1);
  const float px = X[3 * ic], py = X[3 * ic + 1], pz = X[3 * ic + 2];
  float gx[4] = {0.f, 0.f, 0.f, 0.f}, gy[4] = {0.f, 0.f, 0.f, 0.f}, gz[4] = {0.f, 0.f, 0.f, 0.f};
  const float inv_two_pi = 0.159154936671257019f;
  int l = 0;
  for (; l + 4 <= slices; l += 4) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float* U = Ub + (long)(l + j) * 6;
      const float c = C[(long)(l + j) * cnt + ic];
      const float a = fmaf(pz, U[4], fmaf(py, U[2], px * U[0]));
      const float bb = fmaf(pz, U[5], fmaf(py, U[3], px * U[1]));
      const float w = c * inv_two_pi / fmaf(a, a, bb * bb);
      gx[j] = fmaf(w, fmaf(a, U[1], -bb * U[0]), gx[j]);
      gy[j] = fmaf(w, fmaf(a, U[3], -bb * U[2]), gy[j]);
      gz[j] = fmaf(w, fmaf(a, U[5], -bb * U[4]), gz[j]);
    }
  }
  for (; l < slices; ++l) {
    const float* U = Ub + (long)l * 6;
    const float c = C[(long)l * cnt + ic];
    const float a = fmaf(pz, U[4], fmaf(py, U[2], px * U[0]));
    const float bb = fmaf(pz, U[5], fmaf(py, U[3], px * U[1]));
    const float w = c * inv_two_pi / fmaf(a, a, bb * bb);
    gx[0] = fmaf(w, fmaf(a, U[1], -bb * U[0]), gx[0]);
    gy[0] = fmaf(w, fmaf(a, U[3], -bb * U[2]), gy[0]);
    gz[0] = fmaf(w, fmaf(a, U[5], -bb * U[4]), gz[0]);
  }
  if (i < cnt) {
    G[3 * i] = ((gx[0] + gx[1]) + (gx[2] + gx[3])) * scale;
    G[3 * i + 1] = ((gy[0] + gy[1]) + (gy[2] + gy[3])) * scale;
    G[3 * i + 2] = ((gz[0] + gz[1]) + (gz[2] + gz[3])) * scale;
  }
}

template <int EPT, int WAVES>
static int launch_forward_grad(SswArgs& A, hipStream_t stream) {
  const long total = (long)A.pairs * A.slices;
  const long groups = (total + WAVES - 1) / WAVES;
  if (groups > 0x7fffffffL) return (int)hipErrorInvalidValue;
  A.num_groups = (int)groups;
  const size_t lds = (size_t)WAVES * 2 * EPT * kWave * sizeof(float);
  if (A.p_int == 2) {
    hipLaunchKernelGGL((ssw_forward_grad_kernel<EPT, WAVES, 2>), dim3((unsigned)groups), dim3(WAVES * 64), lds, stream, A);
  } else {
    hipLaunchKernelGGL((ssw_forward_grad_kernel<EPT, WAVES, 0>), dim3((unsigned)groups), dim3(WAVES * 64), lds, stream, A);
  }
  return (int)hipGetLastError();
}

int dispatch_forward_grad(SswArgs& A, hipStream_t stream) {
  switch (ept_for(A.n, A.m)) {
#ifdef SHW_DEV_ONLY_EPT
    case SHW_DEV_ONLY_EPT: return launch_forward_grad<SHW_DEV_ONLY_EPT, (SHW_DEV_ONLY_EPT <= 32 ? 2 : 1)>(A, stream);
#else
    case 1: return launch_forward_grad<1, 4>(A, stream);
    case 2: return launch_forward_grad<2, 4>(A, stream);
    case 4: return launch_forward_grad<4, 4>(A, stream);
    case 8: return launch_forward_grad<8, 4>(A, stream);
    case 16: return launch_forward_grad<16, 4>(A, stream);
    case 32: return launch_forward_grad<32, 2>(A, stream);
    case 64: return launch_forward_grad<64, 1>(A, stream);
    case 128: return launch_forward_grad<128, 1>(A, stream);
#endif
    default: return (int)hipErrorInvalidValue;
  }
}


int launch_backward_points(const float* xs, const float* xt, const float* dirs, const float* coef_s,
                           const float* coef_t, int pairs, int n, int m, int slices, long u_pair_stride,
                           float scale, float* grad_xs, float* grad_xt, hipStream_t stream) {
  const int chunks_s = (n + 255) / 256, chunks_t = (m + 255) / 256;
  hipLaunchKernelGGL(ssw_backward_points_kernel, dim3(chunks_s + chunks_t, pairs), dim3(256), 0, stream, xs, xt, dirs,
                     coef_s, coef_t, n, m, slices, u_pair_stride, scale, grad_xs, grad_xt, chunks_s);
  return (int)hipGetLastError();
}

}  // namespace shw
