// shw_ssw_grad_kv.hip -- training kernel for the largest size class (4096 < n <= 8192, 128 keys per
// lane).  The packed-key kernel of shw_ssw_grad.hip needs ~3x128 live registers per lane there; this
// variant sorts 64-bit (coordinate bits << 32 | original index) items instead (wave_sort_kv): more
// compare-exchange work per item, but a third of the live state.  Same outputs, same stable order.
#include "ssw_common.hpp"

namespace shw {

template <int EPT, int WAVES, int PMODE>
__global__ __launch_bounds__(WAVES * 64) void ssw_forward_grad_kv_kernel(SswArgs A) {
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

  float U[6];
  load_frame(A.dirs, (long)b * A.u_pair_stride + (long)l * 6, U);   // (3,2) row-major: U[2*d + k]

  item_t item[EPT];
  float sum_v = 0.f, sum_u = 0.f;
#pragma nounroll
  for (int which = 0; which < 2; ++which) {
    const float* X = which == 0 ? A.xt + (long)b * A.m * A.pstride : A.xs + (long)b * n * A.pstride;
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

int launch_forward_grad_kv128(SswArgs& A, hipStream_t stream) {
  constexpr int EPT = 128, WAVES = 1;
  const long groups = (long)A.pairs * A.slices;
  if (groups > 0x7fffffffL) return (int)hipErrorInvalidValue;
  A.num_groups = (int)groups;
  const size_t lds = (size_t)WAVES * 2 * EPT * kWave * sizeof(float);
  if (A.p_int == 2) {
    hipLaunchKernelGGL((ssw_forward_grad_kv_kernel<EPT, WAVES, 2>), dim3((unsigned)groups), dim3(WAVES * 64), lds, stream, A);
  } else {
    hipLaunchKernelGGL((ssw_forward_grad_kv_kernel<EPT, WAVES, 0>), dim3((unsigned)groups), dim3(WAVES * 64), lds, stream, A);
  }
  return (int)hipGetLastError();
}

}  // namespace shw
