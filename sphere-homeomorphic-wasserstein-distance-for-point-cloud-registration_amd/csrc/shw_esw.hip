// shw_esw.hip -- Euclidean sliced-Wasserstein (the notebooks' SWD baseline): project both clouds on L
// unit directions of R^3, sort both projected sequences, sum |u_(i) - v_(i)|^p.
//
// Replaces `sliced_wasserstein_distance` (Wasserstein_flow_problem/Flow_cube.ipynb:280-292; the same cell
// in Flow_ellipsoid*.ipynb), which needs equal sample counts.  It is the spherical kernel without the
// circle: no atan2, no cyclic shift (k = 0), no wrap -- one wavefront per (pair, slice), the same
// register-resident sort.  Per slice it returns S_l = sum_i |u_(i) - v_(i)|^p; the outer
// (mean_l S_l)^(1/p) of the notebook is host-side arithmetic on L numbers.
#include "ssw_common.hpp"

namespace shw {

struct EswArgs {
  const float* xs;
  const float* xt;
  const float* thetas;     // (slices, 3) or (pairs, slices, 3) unit directions
  float* slice_sum;        // (pairs*slices)
  float* coef_s;           // optional (pairs*slices*n): d S_l / d projection, original point order
  float* coef_t;
  int pairs, n, slices;
  long theta_pair_stride;  // 0 = shared
  float p;
  int p_int;
  int num_groups;
};

// projections of one cloud on one direction; padding keys are +inf
template <int EPT>
__device__ __forceinline__ void load_projections(const float* __restrict__ X, int count, int lane,
                                                 float tx, float ty, float tz, float (&key)[EPT]) {
  constexpr int CH = EPT < 8 ? EPT : 8;
#pragma unroll
  for (int r0 = 0; r0 < EPT; r0 += CH) {
    float px[CH], py[CH], pz[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const int i = min((r0 + j) * kWave + lane, count - 1);
      px[j] = X[3 * i]; py[j] = X[3 * i + 1]; pz[j] = X[3 * i + 2];
    }
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const int i = (r0 + j) * kWave + lane;
      const float d = fmaf(pz[j], tz, fmaf(py[j], ty, px[j] * tx));
      key[r0 + j] = (i < count) ? d : __builtin_inff();
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// order-preserving map float -> unsigned (negative floats included), for the key+index items
__device__ __forceinline__ float orderable(float x) {
  const int b = as_i(x);
  return as_f(b ^ ((b >> 31) | (int)0x80000000));
}
__device__ __forceinline__ float from_orderable(float x) {
  const int b = as_i(x);
  return as_f(b ^ (((~b) >> 31) | (int)0x80000000));
}

template <int EPT, int WAVES, int PMODE, bool GRAD>
__global__ __launch_bounds__(WAVES * 64) void esw_kernel(EswArgs A) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int ROW = EPT * kWave;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* vbuf = lds + wave * ((GRAD ? 2 : 1) * ROW);
  int* vidx = reinterpret_cast<int*>(vbuf + ROW);            // GRAD only

  const int vid = xcd_contiguous_id(blockIdx.x, A.num_groups);
  const int s = vid * WAVES + wave;
  if (s >= A.pairs * A.slices) return;
  const int b = s / A.slices, l = s - b * A.slices;
  const int n = A.n;
  const float* T = A.thetas + (long)b * A.theta_pair_stride + (long)l * 3;
  const float tx = T[0], ty = T[1], tz = T[2];

  float u[EPT];
  int uidx[EPT];
#pragma nounroll
  for (int which = 0; which < 2; ++which) {
    const float* X = (which == 0 ? A.xt : A.xs) + (long)b * n * 3;
    int ln = lane;
    asm volatile("" : "+v"(ln));
    load_projections<EPT>(X, n, ln, tx, ty, tz, u);
    if constexpr (GRAD) {
      item_t item[EPT];
#pragma unroll
      for (int r = 0; r < EPT; ++r) item[r] = make_item(orderable(u[r]), r * kWave + ln);
      wave_sort_kv<EPT>(item, ln);
#pragma unroll
      for (int r = 0; r < EPT; ++r) {
        u[r] = from_orderable(item_key(item[r]));
        uidx[r] = item_idx(item[r]);
      }
    } else {
      wave_sort<EPT>(u, ln);
    }
    if (which == 0) {
#pragma unroll
      for (int r = 0; r < EPT; ++r) {
        vbuf[r * kWave + lane] = u[r];
        if constexpr (GRAD) vidx[r * kWave + lane] = uidx[r];
      }
    }
  }
  __builtin_amdgcn_wave_barrier();

  float acc = 0.f;
  float* cs = GRAD ? A.coef_s + (long)s * n : nullptr;
  float* ct = GRAD ? A.coef_t + (long)s * n : nullptr;
#pragma unroll
  for (int r = 0; r < EPT; ++r) {
    const int e = lane * EPT + r;                            // sorted position; target e sits in the same slot
    const float d = u[r] - vbuf[r * kWave + lane];
    if (e < n) {
      acc += pow_abs<PMODE>(d, A.p, A.p_int);
      if constexpr (GRAD) {
        const float g = dpow_abs<PMODE>(d, A.p, A.p_int);
        cs[uidx[r]] = g;
        ct[vidx[r * kWave + lane]] = -g;
      }
    }
  }
  acc = wave_sum(acc, lane);
  if (lane == 0) A.slice_sum[s] = acc;
}

// grad[b,i,:] = sum_l w[b,l] * coef[b,l,i] * theta[b,l,:]   (w = upstream gradient of the per-slice sums)
__global__ __launch_bounds__(256) void esw_backward_points_kernel(const float* __restrict__ thetas,
                                                                  const float* __restrict__ coef_s,
                                                                  const float* __restrict__ coef_t,
                                                                  const float* __restrict__ slice_w, int n,
                                                                  int slices, long theta_pair_stride,
                                                                  float* __restrict__ grad_xs,
                                                                  float* __restrict__ grad_xt, int chunks) {
  const int b = blockIdx.y;
  const bool is_t = (int)blockIdx.x >= chunks;
  const int chunk = is_t ? blockIdx.x - chunks : blockIdx.x;
  const int i = chunk * 256 + threadIdx.x;
  const int ic = min(i, n - 1);
  const float* C = (is_t ? coef_t : coef_s) + (long)b * slices * n;
  float* G = (is_t ? grad_xt : grad_xs) + (long)b * n * 3;
  const float* Tb = thetas + (long)b * theta_pair_stride;
  const float* W = slice_w + (long)b * slices;
  float gx = 0.f, gy = 0.f, gz = 0.f;
  for (int l = 0; l < slices; ++l) {
    const float c = C[(long)l * n + ic] * W[l];
    gx = fmaf(c, Tb[3 * l], gx);
    gy = fmaf(c, Tb[3 * l + 1], gy);
    gz = fmaf(c, Tb[3 * l + 2], gz);
  }
  if (i < n) { G[3 * i] = gx; G[3 * i + 1] = gy; G[3 * i + 2] = gz; }
}

// grad_theta[b,l,:] = w[b,l] * sum_i ( coef_s[b,l,i] * xs[b,i,:] + coef_t[b,l,i] * xt[b,i,:] )
// (the projections are x . theta, so d S_l / d theta is the coefficient-weighted sum of the points; used by
// the notebooks' max-sliced-W, which ascends on the direction).  One workgroup per (slice, pair), fixed-order tree.
__global__ __launch_bounds__(256) void esw_backward_dirs_kernel(const float* __restrict__ xs,
                                                                const float* __restrict__ xt,
                                                                const float* __restrict__ coef_s,
                                                                const float* __restrict__ coef_t,
                                                                const float* __restrict__ slice_w, int n, int slices,
                                                                float* __restrict__ grad_thetas) {
  __shared__ float red[3][4];
  const int l = blockIdx.x, b = blockIdx.y;
  const float* Xs = xs + (long)b * n * 3;
  const float* Xt = xt + (long)b * n * 3;
  const float* Cs = coef_s + ((long)b * slices + l) * n;
  const float* Ct = coef_t + ((long)b * slices + l) * n;
  float g[3] = {0.f, 0.f, 0.f};
  for (int i = threadIdx.x; i < n; i += 256) {
    const float cs = Cs[i], ct = Ct[i];
#pragma unroll
    for (int d = 0; d < 3; ++d) g[d] = fmaf(cs, Xs[3 * i + d], fmaf(ct, Xt[3 * i + d], g[d]));
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const float v = wave_sum(g[d], lane);
    if (lane == 0) red[d][wave] = v;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int d = threadIdx.x;
    grad_thetas[((long)b * slices + l) * 3 + d] =
        ((red[d][0] + red[d][1]) + (red[d][2] + red[d][3])) * slice_w[(long)b * slices + l];
  }
}

template <int EPT, int WAVES>
static int launch_esw(EswArgs& A, hipStream_t stream) {
  const long total = (long)A.pairs * A.slices;
  const long groups = (total + WAVES - 1) / WAVES;
  if (groups > 0x7fffffffL) return (int)hipErrorInvalidValue;
  A.num_groups = (int)groups;
  const bool grad = A.coef_s != nullptr;
  const size_t lds = (size_t)WAVES * (grad ? 2 : 1) * EPT * kWave * sizeof(float);
  const dim3 grid((unsigned)groups), block(WAVES * 64);
  if (A.p_int == 2) {
    if (grad) hipLaunchKernelGGL((esw_kernel<EPT, WAVES, 2, true>), grid, block, lds, stream, A);
    else hipLaunchKernelGGL((esw_kernel<EPT, WAVES, 2, false>), grid, block, lds, stream, A);
  } else {
    if (grad) hipLaunchKernelGGL((esw_kernel<EPT, WAVES, 0, true>), grid, block, lds, stream, A);
    else hipLaunchKernelGGL((esw_kernel<EPT, WAVES, 0, false>), grid, block, lds, stream, A);
  }
  return (int)hipGetLastError();
}

}  // namespace shw

extern "C" {

int shw_esw_forward(const float* xs, const float* xt, const float* thetas, int pairs, int n, int slices,
                    long theta_pair_stride, float p, float* slice_sum, float* coef_s, float* coef_t, void* stream) {
  if (!xs || !xt || !thetas || !slice_sum) return (int)hipErrorInvalidValue;
  if ((coef_s == nullptr) != (coef_t == nullptr)) return (int)hipErrorInvalidValue;
  if (pairs < 0 || slices < 0 || n < 1 || n > 4096 || !(p >= 1.f)) return (int)hipErrorInvalidValue;
  if (theta_pair_stride != 0 && theta_pair_stride < (long)slices * 3) return (int)hipErrorInvalidValue;
  if (pairs == 0 || slices == 0) return 0;
  shw::EswArgs A{};
  A.xs = xs; A.xt = xt; A.thetas = thetas; A.slice_sum = slice_sum; A.coef_s = coef_s; A.coef_t = coef_t;
  A.pairs = pairs; A.n = n; A.slices = slices; A.theta_pair_stride = theta_pair_stride;
  A.p = p; A.p_int = shw::small_integer_power(p);
  if (p == 1.f) A.p_int = 1;
  switch (shw::ept_for(n, n)) {
    case 1: return shw::launch_esw<1, 4>(A, (hipStream_t)stream);
    case 2: return shw::launch_esw<2, 4>(A, (hipStream_t)stream);
    case 4: return shw::launch_esw<4, 4>(A, (hipStream_t)stream);
    case 8: return shw::launch_esw<8, 4>(A, (hipStream_t)stream);
    case 16: return shw::launch_esw<16, 4>(A, (hipStream_t)stream);
    case 32: return shw::launch_esw<32, 2>(A, (hipStream_t)stream);
    case 64: return shw::launch_esw<64, 1>(A, (hipStream_t)stream);
    default: return (int)hipErrorInvalidValue;
  }
}

int shw_esw_backward_points(const float* thetas, const float* coef_s, const float* coef_t, const float* slice_w,
                            int pairs, int n, int slices, long theta_pair_stride, float* grad_xs, float* grad_xt,
                            void* stream) {
  if (!thetas || !coef_s || !coef_t || !slice_w || !grad_xs || !grad_xt) return (int)hipErrorInvalidValue;
  if (pairs < 0 || pairs > 65535 || slices < 0 || n < 1) return (int)hipErrorInvalidValue;
  if (pairs == 0) return 0;
  const int chunks = (n + 255) / 256;
  hipLaunchKernelGGL(shw::esw_backward_points_kernel, dim3(2 * chunks, pairs), dim3(256), 0, (hipStream_t)stream,
                     thetas, coef_s, coef_t, slice_w, n, slices, theta_pair_stride, grad_xs, grad_xt, chunks);
  return (int)hipGetLastError();
}

int shw_esw_backward_dirs(const float* xs, const float* xt, const float* coef_s, const float* coef_t,
                          const float* slice_w, int pairs, int n, int slices, float* grad_thetas, void* stream) {
  if (!xs || !xt || !coef_s || !coef_t || !slice_w || !grad_thetas) return (int)hipErrorInvalidValue;
  if (pairs < 0 || pairs > 65535 || slices < 0 || n < 1) return (int)hipErrorInvalidValue;
  if (pairs == 0 || slices == 0) return 0;
  hipLaunchKernelGGL(shw::esw_backward_dirs_kernel, dim3(slices, pairs), dim3(256), 0, (hipStream_t)stream, xs, xt,
                     coef_s, coef_t, slice_w, n, slices, grad_thetas);
  return (int)hipGetLastError();
}

}  // extern "C"
