// shw_ssw_grad2.hip -- loss + gradient coefficients for p != 1, 257..2048 points: TWO wavefronts per (pair, slice),
// one cloud each (the layout of shw_ssw_p1_merge.hip).
//
// The one-wave kernel of shw_ssw_grad.hip carries the sorted source (32 coordinates + 16 packed index pairs per lane)
// in registers ACROSS the target's sort: 26 spilled VGPRs and ~0.4 GB of scratch traffic per launch at config 3 with
// the network sort, 84 with the distribution sort.  Here nothing is carried across a sort:
//   phase 1  wave 0 projects and sorts the source, wave 1 the target, at the same time, each with its own LDS scratch
//            (sorted_with_indices_binned: 32*EPT counters + a 64*EPT-word buffer);
//   phase 2  both publish their sorted coordinates as rows [r][lane] (sorted position lane*EPT + r) and their sorted
//            original indices as 16-bit rows in the counters' place;                                        barrier
//   phase 3  the shift solve, split by source registers: wave h evaluates c(k-1), c(k), c(k+1) on registers
//            [h*EPT/2, (h+1)*EPT/2) of every lane; the two partial triples are added in wave order through LDS
//            (one barrier per evaluation, two-slot parity buffer), so both waves take identical decisions;
//   phase 4  coefficients of the wave's own half: g = (1/n) d|D|^p/dD, D = u_(e) - v_ext(e + k*);       barrier
//            the two coordinate rows become staging rows: row_s[idx_u(e)] = g, row_t[idx_v(e + k*)] = -g; barrier
//            wave 0 stores the source row, wave 1 the target row, coalesced.  Every coefficient is written exactly
//            once (the permutations are bijections): no zero fill, no atomics, deterministic.
// LDS: 2 x 6 bytes per atom (24 KB per slice at 2048 points, i.e. the 12 KB per wave of the one-wave kernel).
#include "bin_sort_idx.hpp"
#include "ssw_common.hpp"

namespace shw {

#ifndef SHW_GRAD2_MINW
#define SHW_GRAD2_MINW 3
#endif
#ifndef SHW_GRAD2_MINW_PARTIAL
#define SHW_GRAD2_MINW_PARTIAL 3
#endif
// counters of the sort with indices (a power of two, bin_sort_idx.hpp) or half a row of 16-bit indices, whichever is larger
constexpr int grad2_counter_floats(int ept) {
  return next_pow2_c(SHW_BINSORT_NB_PER_EPT * ept) > 32 * ept ? next_pow2_c(SHW_BINSORT_NB_PER_EPT * ept) : 32 * ept;
}
// (20 .. 28 keys per lane need the 168 registers of three waves per SIMD like 32: asked for four they spill 27 .. 64)
constexpr int grad2_waves_per_simd(int ept, bool full) { return ept == 32 ? (full ? SHW_GRAD2_MINW : SHW_GRAD2_MINW_PARTIAL) : (ept >= 20 ? 3 : 4); }

template <int EPT, int PMODE, bool FULL>
__global__ __launch_bounds__(128, grad2_waves_per_simd(EPT, FULL)) void ssw_forward_grad2_kernel(SswArgs A) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int ROW = EPT * kWave;
  constexpr int HALF = EPT / 2;
  constexpr int CNT = grad2_counter_floats(EPT);                   // the sort's counters, then the 16-bit index row
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* row_s = lds;                                              // source: coordinates row, then staging row
  float* row_t = lds + ROW + CNT;                                  // target
  unsigned short* idx_s = reinterpret_cast<unsigned short*>(row_s + ROW);   // sorted original indices (counters first)
  unsigned short* idx_t = reinterpret_cast<unsigned short*>(row_t + ROW);
  float* red = lds + 2 * (ROW + CNT);                              // [2 parities][2 waves][4] partial sums, [2] sums
  float* my_row = wave ? row_t : row_s;
  unsigned short* my_idx = wave ? idx_t : idx_s;

  const int s = xcd_contiguous_id(blockIdx.x, A.num_groups);      // one workgroup per (pair, slice)
  const int b = s / A.slices, l = s - b * A.slices;
  const int n = A.n;                                               // == A.m on this path

  float U[6];
  load_frame(A.dirs, (long)b * A.u_pair_stride + (long)l * 6, U);   // (3,2) row-major: U[2*d + k]

  // ---- phase 1 + 2 ------------------------------------------------------------------------------------------
  {
    float val[EPT];
    int idx[EPT];
    const float* X = (wave ? A.xt : A.xs) + (long)b * n * A.pstride;
    const float part = sorted_with_indices_binned<EPT, true, FULL>(X, n, lane, U, reinterpret_cast<unsigned*>(my_row + ROW),
                                                       my_row, val, idx);
    const float total = wave_sum_uniform(part, lane);
    // (the wave has gathered its exact coordinates out of my_row: LDS operations of a wave execute in order)
#pragma unroll
    for (int r = 0; r < EPT; ++r) {
      my_row[r * kWave + lane] = val[r];
      my_idx[r * kWave + lane] = (unsigned short)idx[r];
    }
    if (lane == 0) red[16 + wave] = total;
  }
  __syncthreads();
  const float sum_u = red[16], sum_v = red[17];

  // ---- phase 3: shift solve on registers [r_base, r_base + HALF) ---------------------------------------------
  const int r_base = wave * HALF;
  float u[HALF];
#pragma unroll
  for (int j = 0; j < HALF; ++j) u[j] = row_s[(r_base + j) * kWave + lane];
  int lo = -n, hi = n;
  float guess = rintf(sum_u - sum_v);
  guess = fminf(fmaxf(guess, (float)lo), (float)hi);
  int k = __builtin_amdgcn_readfirstlane((int)guess);
  bool lo_tight = false, hi_tight = false;
  int step = 1;
  float c0 = 0.f;
  for (int it = 0; it < 64; ++it) {
    int ln = lane;
    asm volatile("" : "+v"(ln));
    float pm, p0, pp;
    if constexpr (FULL) shift_costs3_full<EPT, PMODE, 64, HALF>(u, row_t, ln, k, A.p, A.p_int, pm, p0, pp, r_base);
    else if constexpr (SHW_GRAD2_INC && HALF + 2 <= EPT) shift_costs3_inc<EPT, PMODE, 64, HALF>(u, row_t, ln, n, k, A.p, A.p_int, pm, p0, pp, r_base);
    else shift_costs3<EPT, PMODE, 64, HALF>(u, row_t, ln, n, k, A.p, A.p_int, pm, p0, pp, r_base);
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
  const float inv_n = 1.f / (float)n;
  if (threadIdx.x == 0) {
    A.slice_cost[s] = c0 * inv_n;
    if (A.slice_shift) A.slice_shift[s] = k;
  }

  // ---- phase 4: coefficients of this wave's half ---------------------------------------------------------------
  auto target_slot = [&](int e, float& off) -> int {
    const int q = min(e, n - 1) + k;                   // in [-n, 2n): one turn at most
    const int turn = (q < 0) ? -1 : ((q >= n) ? 1 : 0);
    off = (float)turn;
    return lds_slot<EPT>(q - turn * n);
  };
  unsigned pair_idx[HALF];                             // (source index) | (target index) << 16
#pragma unroll
  for (int j = 0; j < HALF; ++j) {
    float off;
    const int slot = target_slot(lane * EPT + r_base + j, off);
    const float d = u[j] - (row_t[slot] + off);
    u[j] = dpow_abs<PMODE>(d, A.p, A.p_int) * inv_n;
    pair_idx[j] = (unsigned)idx_s[(r_base + j) * kWave + lane] | ((unsigned)idx_t[slot] << 16);
  }
  __syncthreads();                                     // every coordinate has been read: the rows become staging rows
#pragma unroll
  for (int j = 0; j < HALF; ++j) {
    if (lane * EPT + r_base + j < n) {
      row_s[pair_idx[j] & 0xffffu] = u[j];
      row_t[pair_idx[j] >> 16] = -u[j];
    }
  }
  __syncthreads();
  float* out = (wave ? A.coef_t : A.coef_s) + (long)s * n;
#pragma unroll
  for (int r = 0; r < EPT; ++r) {
    const int i = r * kWave + lane;
    if (FULL || i < n) out[i] = my_row[i];
  }
}

// The masked (partially filled) 32-keys-per-lane form lives in its own translation unit, shw_ssw_grad2_m32.hip (this file
// compiled with SHW_GRAD2_MASKED32_UNIT): it is the one kernel of the library that is faster WITH the compiler's SLP
// vectorisation (without the packed forms it spills 45 registers: N = 2000 training step 0.62 -> 0.74 ms), every other
// instantiation is built with -fno-slp-vectorize like the rest of the library (Makefile, SLP_UNITS).
#ifdef SHW_GRAD2_MASKED32_UNIT
int launch_forward_grad2_masked32(SswArgs& A, hipStream_t stream) {
  constexpr int EPT = 32;
  const long total = (long)A.pairs * A.slices;
  if (total > 0x7fffffffL) return (int)hipErrorInvalidValue;
  A.num_groups = (int)total;
  const size_t lds = (size_t)(2 * (EPT * kWave + grad2_counter_floats(EPT)) + 32) * sizeof(float);
  const dim3 grid((unsigned)total), block(128);
  if (A.p_int == 2) hipLaunchKernelGGL((ssw_forward_grad2_kernel<EPT, 2, false>), grid, block, lds, stream, A);
  else hipLaunchKernelGGL((ssw_forward_grad2_kernel<EPT, 0, false>), grid, block, lds, stream, A);
  return (int)hipGetLastError();
}
#else
int launch_forward_grad2_masked32(SswArgs& A, hipStream_t stream);   // shw_ssw_grad2_m32.hip

template <int EPT>
static int launch_forward_grad2(SswArgs& A, hipStream_t stream) {
  const long total = (long)A.pairs * A.slices;
  if (total > 0x7fffffffL) return (int)hipErrorInvalidValue;
  A.num_groups = (int)total;
  const size_t lds = (size_t)(2 * (EPT * kWave + grad2_counter_floats(EPT)) + 32) * sizeof(float);
  // (the mask-free forms index with shifts and masks: power-of-two classes only)
  const bool full = is_pow2(EPT) && (A.n == EPT * kWave) && (A.m == EPT * kWave);
  const dim3 grid((unsigned)total), block(128);
  if constexpr (is_pow2(EPT)) {
    if (full) {
      if (A.p_int == 2) hipLaunchKernelGGL((ssw_forward_grad2_kernel<EPT, 2, true>), grid, block, lds, stream, A);
      else hipLaunchKernelGGL((ssw_forward_grad2_kernel<EPT, 0, true>), grid, block, lds, stream, A);
      return (int)hipGetLastError();
    }
  }
  if constexpr (EPT == 32) {
    return launch_forward_grad2_masked32(A, stream);
  } else {
    if (A.p_int == 2) hipLaunchKernelGGL((ssw_forward_grad2_kernel<EPT, 2, false>), grid, block, lds, stream, A);
    else hipLaunchKernelGGL((ssw_forward_grad2_kernel<EPT, 0, false>), grid, block, lds, stream, A);
    return (int)hipGetLastError();
  }
}

int dispatch_forward_grad2(SswArgs& A, hipStream_t stream) {
  switch (kpl_for(A.n, A.m, true)) {
#ifdef SHW_DEV_ONLY_EPT
    case SHW_DEV_ONLY_EPT: return launch_forward_grad2<SHW_DEV_ONLY_EPT>(A, stream);
#else
    case 8: return launch_forward_grad2<8>(A, stream);
    case 12: return launch_forward_grad2<12>(A, stream);
    case 16: return launch_forward_grad2<16>(A, stream);
    case 20: return launch_forward_grad2<20>(A, stream);
    case 24: return launch_forward_grad2<24>(A, stream);
    case 32: return launch_forward_grad2<32>(A, stream);
#endif
    default: return (int)hipErrorInvalidValue;
  }
}
#endif

}  // namespace shw
