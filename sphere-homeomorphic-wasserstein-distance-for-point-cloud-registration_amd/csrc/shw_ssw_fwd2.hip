// shw_ssw_fwd2.hip -- loss-only kernel for p != 1, 257..2048 points: TWO wavefronts per (pair, slice), one cloud each
// (the structure of shw_ssw_grad2.hip without the permutation).
//
// The one-wave kernel of shw_ssw_fwd.hip keeps the sorted source in registers across the target's sort: at the 3 waves
// per SIMD its 12 KB of LDS allow it spills 25 VGPRs (62 MB of scratch writes per launch at config 3, 16x the kernel's
// algorithmic traffic).  Here wave 0 projects and sorts the source while wave 1 does the target, each with its own
// scratch; both publish their sorted cloud as [r][lane] rows; the shift solve is split by source registers (wave h
// evaluates registers [h*EPT/2, (h+1)*EPT/2) of every lane, partial sums added through LDS, one barrier per evaluation).
// Clouds that do not fill the size class use the pre-rotated extended target rows (ssw_common.hpp), written by wave 1.
#include "bin_sort.hpp"
#include "ssw_common.hpp"

namespace shw {

#ifndef SHW_FWD2_MINW
#define SHW_FWD2_MINW 3
#endif

template <int EPT, int PMODE, bool FULL>
__global__ __launch_bounds__(128, SHW_FWD2_MINW) void ssw_forward2_kernel(SswArgs A) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int NBINS = binsort_bins<EPT>();             // 32*EPT for the power-of-two classes
  constexpr int SCR = NBINS + 64 * EPT;                  // per wave: the counters + 64*EPT staging floats
  constexpr int HALF = EPT / 2;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* scr_s = lds;                                      // wave 0: source
  float* scr_t = lds + SCR;                                // wave 1: target
  float* row_s = scr_s + NBINS;                            // published rows = the staging buffers
  float* row_t = scr_t + NBINS;
  float* red = lds + 2 * SCR;                              // [2 parities][2 waves][4] partial sums, [2] coordinate sums
  float* my_scr = wave ? scr_t : scr_s;

  const int s = xcd_contiguous_id(blockIdx.x, A.num_groups);
  const int b = s / A.slices, l = s - b * A.slices;
  const int n = A.n;                                       // == A.m on this path

  float U[6];
  load_frame(A.dirs, (long)b * A.u_pair_stride + (long)l * 6, U);

  float key[EPT];
  {
    const float* X = (wave ? A.xt : A.xs) + (long)b * n * A.pstride;
    const float part = load_coords<EPT, FULL, true>(X, n, lane, U, key);
    wave_sort_binned<EPT, FULL>(key, lane, n, my_scr);
    const float total = wave_sum_uniform(part, lane);
    if (FULL || wave == 0) {                               // (partial sizes: the target goes out as extended rows below)
#pragma unroll
      for (int r = 0; r < EPT; ++r) my_scr[NBINS + r * kWave + lane] = key[r];
    }
    if (lane == 0) red[16 + wave] = total;
  }
  __syncthreads();
  const float sum_u = red[16], sum_v = red[17];
  int lo = -n, hi = n;
  float guess = rintf(sum_u - sum_v);
  guess = fminf(fmaxf(guess, (float)lo), (float)hi);
  int k = __builtin_amdgcn_readfirstlane((int)guess);
  int kc = k;
  if constexpr (!FULL) {
    static_assert(ExtRows<EPT>::FLOATS <= SCR, "extended rows must fit a wave's scratch");
    if (wave == 1) ext_rows_write<EPT>(key, scr_t, lane, n, kc);
    __syncthreads();
  }

  const int r_base = wave * HALF;
  float u[HALF];
#pragma unroll
  for (int j = 0; j < HALF; ++j) u[j] = row_s[(r_base + j) * kWave + lane];
  bool lo_tight = false, hi_tight = false;
  int step = 1;
  float c0 = 0.f;
  for (int it = 0; it < 64; ++it) {
    int ln = lane;
    asm volatile("" : "+v"(ln));
    float pm, p0, pp;
    if constexpr (FULL) {
      shift_costs3_full<EPT, PMODE, 64, HALF>(u, row_t, ln, k, A.p, A.p_int, pm, p0, pp, r_base);
    } else {
      if (k - kc >= ExtRows<EPT>::M || kc - k >= ExtRows<EPT>::M) {      // uniform over the workgroup: re-centre
        __syncthreads();                                                 // everyone is done reading the old rows
        kc = k;
        if (wave == 1) ext_rows_write<EPT>(key, scr_t, ln, n, kc);
        __syncthreads();
      }
      shift_costs3_ext<EPT, PMODE, HALF>(u, scr_t, ln, n, k - kc, A.p, A.p_int, pm, p0, pp, r_base);
    }
    float* slot = red + (it & 1) * 8;
    if (lane == 0) { slot[wave * 4] = pm; slot[wave * 4 + 1] = p0; slot[wave * 4 + 2] = pp; }
    __syncthreads();
    float cm = slot[0] + slot[4], cp = slot[2] + slot[6];
    c0 = slot[1] + slot[5];
    cm = as_f(__builtin_amdgcn_readfirstlane(as_i(cm)));
    c0 = as_f(__builtin_amdgcn_readfirstlane(as_i(c0)));
    cp = as_f(__builtin_amdgcn_readfirstlane(as_i(cp)));
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

template <int EPT>
static int launch_forward2(SswArgs& A, hipStream_t stream) {
  const long total = (long)A.pairs * A.slices;
  if (total > 0x7fffffffL) return (int)hipErrorInvalidValue;
  A.num_groups = (int)total;
  const size_t lds = (size_t)(2 * (binsort_bins<EPT>() + 64 * EPT) + 32) * sizeof(float);
  // (the mask-free forms index with shifts and masks: power-of-two classes only)
  const bool full = is_pow2(EPT) && (A.n == EPT * kWave) && (A.m == EPT * kWave);
  const dim3 grid((unsigned)total), block(128);
  if constexpr (is_pow2(EPT)) {
    if (full) {
      if (A.p_int == 2) hipLaunchKernelGGL((ssw_forward2_kernel<EPT, 2, true>), grid, block, lds, stream, A);
      else hipLaunchKernelGGL((ssw_forward2_kernel<EPT, 0, true>), grid, block, lds, stream, A);
      return (int)hipGetLastError();
    }
  }
  if (A.p_int == 2) hipLaunchKernelGGL((ssw_forward2_kernel<EPT, 2, false>), grid, block, lds, stream, A);
  else hipLaunchKernelGGL((ssw_forward2_kernel<EPT, 0, false>), grid, block, lds, stream, A);
  return (int)hipGetLastError();
}

int dispatch_forward2(SswArgs& A, hipStream_t stream) {
  switch (kpl_for(A.n, A.m)) {
#ifdef SHW_DEV_ONLY_EPT
    case SHW_DEV_ONLY_EPT: return launch_forward2<SHW_DEV_ONLY_EPT>(A, stream);
#else
    case 8: return launch_forward2<8>(A, stream);
    case 12: return launch_forward2<12>(A, stream);
    case 16: return launch_forward2<16>(A, stream);
    case 20: return launch_forward2<20>(A, stream);
    case 24: return launch_forward2<24>(A, stream);
    case 28: return launch_forward2<28>(A, stream);
    case 32: return launch_forward2<32>(A, stream);
#endif
    default: return (int)hipErrorInvalidValue;
  }
}

}  // namespace shw
