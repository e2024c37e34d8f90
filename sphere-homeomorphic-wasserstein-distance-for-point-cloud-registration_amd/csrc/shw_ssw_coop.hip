// shw_ssw_coop.hip -- loss-only kernel for p != 1, W wavefronts of one workgroup per (pair, slice), sorted by the
// cooperative distribution sort of coop_sort.hpp.  See ssw_common.hpp for the path being replaced
// (sliced_cost, max_spherical_sliced_w.py:251-286; binary_search_circle :117-207 via the shift equivalence A8).
//
// Per slice: project the source (every lane EPT points), sort it (registers: sorted positions gl*EPT + r), project
// and sort the target, lay the sorted target out in LDS as [r][64 W] (conflict-free rows for the shift evaluation),
// then minimise the convex sequence c(k) exactly like the multi-wave kernel of shw_ssw_fwd.hip: every wave evaluates
// c(k-1), c(k), c(k+1) on its own 64*EPT source atoms, the partial sums are added across waves in wave order through
// LDS (all waves take identical decisions).
#include "coop_sort.hpp"
#include "ssw_common.hpp"

namespace shw {

#ifndef SHW_COOP_MINW
#define SHW_COOP_MINW 0      // 0: let the register allocator choose
#endif

// (any n keeps the sorted target in registers for re-centring the extended rows: without the bound the allocator
//  takes 259 registers and halves the occupancy -- 0.43 instead of 0.25 ms at n = 3000, B = 43, L = 256)
template <int EPT, int W, int PMODE, bool FULL>
__global__ __launch_bounds__(W * 64, FULL ? 1 : 2) void ssw_forward_coop_kernel(SswArgs A) {
  typedef Coop<EPT, W> C;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  unsigned* cnt = reinterpret_cast<unsigned*>(lds);
  float* buf = lds + C::NB;
  int* red = reinterpret_cast<int*>(lds + C::NB + C::CAP);
  float* redf = reinterpret_cast<float*>(red) + 2 * W;        // [2 parities][W][4] partial sums, then [W][2] coordinate sums
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int gl = wave * 64 + lane;
  const int s = xcd_contiguous_id(blockIdx.x, A.num_groups);   // one workgroup per (pair, slice)
  const int b = s / A.slices, l = s - b * A.slices;
  const int n = A.n;

  float U[6];
  load_frame(A.dirs, (long)b * A.u_pair_stride + (long)l * 6, U);   // (3,2) row-major: U[2*d + k]

  coop_zero_counters<EPT, W>(cnt, gl);
  float u[EPT], key[EPT];
  float part_u = 0.f, part_v = 0.f;
#pragma nounroll
  for (int which = 0; which < 2; ++which) {                     // 0: source -> registers, 1: target -> LDS rows
    const float* X = (which == 0 ? A.xs : A.xt) + (long)b * n * A.pstride;
    int g2 = gl;
    asm volatile("" : "+v"(g2));
    const float part = load_coords<EPT, FULL, false, C::NCOL>(X, n, g2, U, key);
    __syncthreads();                                            // counters zeroed (and the previous cloud's rows read)
    coop_sort<EPT, W, FULL>(key, wave, lane, n, cnt, buf, red);
    if (which == 0) {
      part_u = wave_sum_uniform(part, lane);
#pragma unroll
      for (int r = 0; r < EPT; ++r) u[r] = key[r];
    } else {
      part_v = wave_sum_uniform(part, lane);
      if constexpr (W > 1) __syncthreads();                     // every wave has read its keys back from buf
      if constexpr (FULL) {
#pragma unroll
        for (int r = 0; r < EPT; ++r) buf[r * C::NCOL + gl] = key[r];
      }
      // (any n: the sorted target stays in `key` and is written as pre-rotated extended rows once the first guess
      //  of the shift is known -- ssw_common.hpp, ExtRows)
    }
  }
  float* sums = redf + 8 * W;
  if (lane == 0) { sums[wave * 2] = part_u; sums[wave * 2 + 1] = part_v; }
  __syncthreads();
  float sum_u = 0.f, sum_v = 0.f;
#pragma unroll
  for (int q = 0; q < W; ++q) { sum_u += sums[q * 2]; sum_v += sums[q * 2 + 1]; }

  // minimise the convex sequence c(k), |k| <= n (solve_shift with the partial sums added across waves)
  int lo = -n, hi = n;
  float guess = rintf(sum_u - sum_v);
  guess = fminf(fmaxf(guess, (float)lo), (float)hi);
  int k = __builtin_amdgcn_readfirstlane((int)guess);
  bool lo_tight = false, hi_tight = false;
  int step = 1;
  float cm = 0.f, c0 = 0.f, cp = 0.f;
  typedef ExtRows<EPT, C::NCOL> X;
  static_assert(X::FLOATS <= C::NB + C::CAP, "extended rows take the counters' and the buffer's place");
  float* ext = lds;
  int kc = k;
  if constexpr (!FULL) ext_rows_write<EPT, C::NCOL>(key, ext, gl, n, kc);   // (the barrier above: cnt / buf are free)
  for (int it = 0; it < 64; ++it) {
    int g2 = gl;
    asm volatile("" : "+v"(g2));
    float pm, p0, pp;
    if constexpr (FULL) {
      shift_costs3_full<EPT, PMODE, C::NCOL>(u, buf, g2, k, A.p, A.p_int, pm, p0, pp);
    } else {
      // (every wave passed the barrier of the previous evaluation's sum after its last read of the rows)
      if (k - kc >= X::M || kc - k >= X::M) {                   // uniform over the workgroup: re-centre the rows
        kc = k;
        ext_rows_write<EPT, C::NCOL>(key, ext, g2, n, kc);
      }
      shift_costs3_ext<EPT, PMODE, EPT, C::NCOL>(u, ext, g2, n, k - kc, A.p, A.p_int, pm, p0, pp);
    }
    if constexpr (W > 1) {
      float* slot = redf + (it & 1) * 4 * W;                    // two parities: one barrier per evaluation
      if (lane == 0) { slot[wave * 4] = pm; slot[wave * 4 + 1] = p0; slot[wave * 4 + 2] = pp; }
      __syncthreads();
      cm = c0 = cp = 0.f;
#pragma unroll
      for (int q = 0; q < W; ++q) { cm += slot[q * 4]; c0 += slot[q * 4 + 1]; cp += slot[q * 4 + 2]; }
      cm = as_f(__builtin_amdgcn_readfirstlane(as_i(cm)));
      c0 = as_f(__builtin_amdgcn_readfirstlane(as_i(c0)));
      cp = as_f(__builtin_amdgcn_readfirstlane(as_i(cp)));
    } else {
      cm = pm; c0 = p0; cp = pp;
    }
    const bool right = (cp < c0) && (k < hi);
    const bool left = !right && (cm < c0) && (k > lo);
    if (!right && !left) break;
    if (right) {
      lo = k + 1; lo_tight = true;
      if (hi_tight) { k = lo + ((hi - lo) >> 1); } else { k = min(k + step, hi); step <<= 1; }
    } else {
      hi = k - 1; hi_tight = true;
      if (lo_tight) { k = lo + ((hi - lo) >> 1); } else { k = max(k - step, lo); step <<= 1; }
    }
    k = __builtin_amdgcn_readfirstlane(k);
  }
  if (threadIdx.x == 0) {
    A.slice_cost[s] = c0 / (float)n;
    if (A.slice_shift) A.slice_shift[s] = k;
  }
}

template <int EPT, int W>
static int launch_forward_coop(SswArgs& A, hipStream_t stream) {
  typedef Coop<EPT, W> C;
  const long total = (long)A.pairs * A.slices;
  if (total > 0x7fffffffL) return (int)hipErrorInvalidValue;
  A.num_groups = (int)total;
  const size_t lds = (size_t)C::LDS_FLOATS * sizeof(float);
  const bool full = is_pow2(EPT) && (A.n == C::CAP) && (A.m == C::CAP);
  const dim3 grid((unsigned)total), block(W * 64);
  if constexpr (is_pow2(EPT)) {                            // (the mask-free forms: power-of-two classes only)
    if (full) {
      if (A.p_int == 2) hipLaunchKernelGGL((ssw_forward_coop_kernel<EPT, W, 2, true>), grid, block, lds, stream, A);
      else hipLaunchKernelGGL((ssw_forward_coop_kernel<EPT, W, 0, true>), grid, block, lds, stream, A);
      return (int)hipGetLastError();
    }
  }
  if (A.p_int == 2) hipLaunchKernelGGL((ssw_forward_coop_kernel<EPT, W, 2, false>), grid, block, lds, stream, A);
  else hipLaunchKernelGGL((ssw_forward_coop_kernel<EPT, W, 0, false>), grid, block, lds, stream, A);
  return (int)hipGetLastError();
}

#ifndef SHW_COOP_EPT
#define SHW_COOP_EPT 32     // keys per lane (measured at 2048 points: 32 / W=1: 0.267 ms, 16 / W=2: 0.288, 8 / W=4: 0.292)
#endif

// point count -> cooperative kernel: W = 2 waves per slice up to 4096 points, 4 up to 8192, and 20 / 24 / 32 keys per
// lane (round 3: a cloud of 3000 points pays for 3072 slots, not 4096; SHW_KPL_CLASSES=0 keeps 32)
int dispatch_forward_coop(SswArgs& A, hipStream_t stream) {
  constexpr int E = SHW_COOP_EPT;
  const int big = A.n > A.m ? A.n : A.m;
  const int padded = next_pow2(big);
  const int W = padded / (64 * E);
  const int kpl = (W >= 2) ? coop_kpl_for(big, W, false) : E;
  switch (W * 100 + kpl) {
#ifdef SHW_DEV_ONLY_EPT      // developer switch: only the 2048-point class
    case (2048 / (64 * E)) * 100 + E: return launch_forward_coop<E, 2048 / (64 * E)>(A, stream);
#else
    case 100 + E: return launch_forward_coop<E, 1>(A, stream);
    case 220: return launch_forward_coop<20, 2>(A, stream);
    case 224: return launch_forward_coop<24, 2>(A, stream);
    case 232: return launch_forward_coop<32, 2>(A, stream);
    case 420: return launch_forward_coop<20, 4>(A, stream);
    case 424: return launch_forward_coop<24, 4>(A, stream);
    case 432: return launch_forward_coop<32, 4>(A, stream);
#endif
    default: return (int)hipErrorInvalidValue;
  }
}

// Small grids (round 3; see dispatch_forward_grad_small_grid): fewer (pair, slice) problems than SIMDs -- 8 keys per lane,
// W = padded / 512 waves per slice
int dispatch_forward_small_grid(SswArgs& A, hipStream_t stream) {
  const int padded = next_pow2(A.n > A.m ? A.n : A.m);
  switch (padded / 512) {
#ifndef SHW_DEV_ONLY_EPT
    case 1: return launch_forward_coop<8, 1>(A, stream);
    case 2: return launch_forward_coop<8, 2>(A, stream);
    case 4: return launch_forward_coop<8, 4>(A, stream);
#endif
    default: return (int)hipErrorInvalidValue;
  }
}

}  // namespace shw
