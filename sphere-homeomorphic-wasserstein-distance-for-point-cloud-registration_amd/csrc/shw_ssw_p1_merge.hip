// shw_ssw_p1_merge.hip -- p == 1, loss only, max(n, m) <= 2048: the level-median closed form of
// emd1D_circle (max_spherical_sliced_w.py:210-247; see shw_ssw_p1.hip for the formula and the reference's
// quirks) with the MERGE done by the sorting network instead of by searches.
//
// One workgroup of TWO wavefronts per (pair, slice).  Wave 0 projects and sorts the source, wave 1 the
// target, at the same time, each in registers exactly like the p != 1 kernels; the cloud a key came from is
// recorded in its lowest mantissa bit (source 0, target 1: on equal coordinates the source atom comes first,
// the reference's stable merge order :232-235).  One more bitonic merge level -- a flip stage between the two
// waves through LDS, then the single-wave cross-lane and in-lane stages -- leaves the 2*64*EPT keys in merged
// order, 64*EPT consecutive positions per wave.  From there everything is arithmetic on registers:
//     #target atoms up to position g  = prefix sum of the tag bits (in-lane, lane scan, wave offset)
//     level numerator                 = (#source)*(m/g) - (#target)*(n/g)      (exact integers, g = gcd)
//     gap to the merged successor     = next key - key  (last live atom: 1 - key; [0, first atom) is not
//                                       integrated, as in the reference)
//     weighted median                 = integer bisection, one masked sum per step, added across the two waves
// The binary searches of the one-wave kernel (12 LDS probes per atom) are gone: 0.96 -> 0.53 ms per launch at
// config-3 sizes.  Clearing the tag bit moves a coordinate by at most one ulp (6e-8): the value changes by
// less than 1e-7 relative, far inside the 1e-5 parity tolerance; the one-wave kernel (exact coordinates) still
// serves clouds above 2048 points.
//
// GRAD = true (training): each wave sorts its cloud WITH the permutation (packed 32-bit words + exact gather +
// fix-up, sorted_with_indices: the exact stable order of torch.sort), parks the sorted original indices in LDS
// (2 B per atom), and the merge proceeds on the exact coordinates' bit patterns as above.  A merged atom's rank
// inside its own cloud is (#atoms of that cloud up to it) - 1 -- already known from the level computation -- so
// its original index is ONE LDS read, and   d cost / d coordinate = (|level_before - med| - |level - med|) / lcm
// (first merged atom: -|level - med| / lcm) is un-permuted through two LDS staging rows and stored coalesced:
// every coefficient written exactly once, no atomics (rows feed ssw_backward_points_kernel).  The one-wave
// search kernel it replaces sorted 64-bit (key, index) items: 3.5 -> see DESIGN.md ms per training step.
#include "bin_sort.hpp"
#include "bin_sort_idx.hpp"
#include "ssw_common.hpp"

#ifndef SHW_P1M_GRAD_BINS
#define SHW_P1M_GRAD_BINS 1     // training: each wave sorts its cloud by the distribution sort with indices (>= 8 keys per lane)
#endif

namespace shw {

// exclusive prefix sum over the 64 lanes of a wave
__device__ __forceinline__ int wave_exclusive_scan(int v, int lane) {
  int incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __builtin_amdgcn_ds_bpermute(max(lane - d, 0) << 2, incl);
    incl += (lane >= d) ? t : 0;
  }
  return incl - v;
}

__device__ __forceinline__ int wave_min_int(int v, int lane) {
  v = min(v, as_i(lane_xor<1>(as_f(v), lane)));
  v = min(v, as_i(lane_xor<2>(as_f(v), lane)));
  v = min(v, as_i(lane_xor<4>(as_f(v), lane)));
  v = min(v, as_i(lane_xor<8>(as_f(v), lane)));
  v = min(v, as_i(lane_xor<16>(as_f(v), lane)));
  v = min(v, as_i(lane_xor<32>(as_f(v), lane)));
  return v;
}

#ifndef SHW_P1M_WAVES
#define SHW_P1M_WAVES 4
#endif
#ifndef SHW_P1M_CHAINED
#define SHW_P1M_CHAINED true
#endif
#ifndef SHW_P1M_LOSS_BINS
#define SHW_P1M_LOSS_BINS 8     // loss only: distribution sort from this many keys per lane on (0: never)
#endif
constexpr bool loss_sort_binned(int ept) { return SHW_P1M_LOSS_BINS != 0 && ept >= SHW_P1M_LOSS_BINS; }
constexpr int merge_waves_per_simd(int ept) { return loss_sort_binned(ept) ? 3 : (ept <= 16 ? 6 : SHW_P1M_WAVES); }
// loss-only LDS floats: the exchange buffer, or the two waves' sort scratch where that is larger
template <int EPT>
constexpr int merge_loss_lds_floats() {
  return loss_sort_binned(EPT) && 2 * (binsort_bins<EPT>() + EPT * kWave) > EPT * 128 ? 2 * (binsort_bins<EPT>() + EPT * kWave)
                                                                                          : EPT * 128;
}

template <int EPT, bool GRAD>
__global__ __launch_bounds__(128, GRAD ? (EPT <= 16 ? 4 : 3) : merge_waves_per_simd(EPT)) void
ssw_level_median_merge_kernel(SswArgs A, int mg, int ng, float inv_lcm) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int CHUNK = EPT * kWave;                     // keys per wave
  // [EPT][128] words: exchange buffer of the merge; GRAD: before that the two clouds' coordinates by original
  // index (sorted_with_indices), after it the two coefficient staging rows
  unsigned* buf = reinterpret_cast<unsigned*>(lds);
  // GRAD only, behind the 16 scratch words: original indices of the two sorted clouds, [2][CHUNK] (lds_slot layout)
  unsigned short* sidx = reinterpret_cast<unsigned short*>(lds + EPT * 128 + 16);
  // 16 words of cross-wave scratch, one slot per wave each: [0..3] median partial sums (two parities),
  // [4,5] tag counts, [6,7] first keys, [8,9] smallest / [10,11] largest level, [12,13] gap totals, [14,15] costs
  float* red = lds + (GRAD ? EPT * 128 : merge_loss_lds_floats<EPT>());
  int* redi = reinterpret_cast<int*>(red);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int s = xcd_contiguous_id(blockIdx.x, A.num_groups);
  const int b = s / A.slices, l = s - b * A.slices;
  const int n = A.n, m = A.m, total_live = n + m;

  float U[6];
  load_frame(A.dirs, (long)b * A.u_pair_stride + (long)l * 6, U);   // (3,2) row-major: U[2*d + k]

  // ---- project + sort: wave 0 the source, wave 1 the target -------------------------------------
  unsigned pk[EPT];
  {
    const float* X = wave == 0 ? A.xs + (long)b * n * A.pstride : A.xt + (long)b * m * A.pstride;
    const int count = wave == 0 ? n : m;
    if constexpr (GRAD) {
      float val[EPT];
      int idx[EPT];
      // round 3: the distribution sort with indices of the p != 1 training kernel (bin_sort_idx.hpp) instead of the
      // bitonic network on packed words; its 32 EPT counters sit where the wave's 16-bit index row goes afterwards
      if constexpr (SHW_P1M_GRAD_BINS != 0 && EPT >= 8)
        sorted_with_indices_binned<EPT, true, false>(X, count, lane, U, reinterpret_cast<unsigned*>(sidx + wave * CHUNK),
                                                     lds + wave * CHUNK, val, idx);
      else
        sorted_with_indices<EPT>(X, count, lane, U, lds + wave * CHUNK, val, idx);
#pragma unroll
      for (int r = 0; r < EPT; ++r) {
        sidx[wave * CHUNK + r * kWave + lane] = (unsigned short)idx[r];     // sorted position lane*EPT + r
        pk[r] = (idx[r] < count) ? (((unsigned)as_i(val[r]) & ~1u) | (unsigned)wave) : 0xffffffffu;
      }
      __syncthreads();                                    // both clouds gathered: their rows become the exchange buffer
    } else {
      float key[EPT];
      load_coords<EPT, false, SHW_P1M_CHAINED>(X, count, lane, U, key);
      if constexpr (loss_sort_binned(EPT)) {
        // the distribution sort of the p != 1 kernels on the tagged keys (pads stay +inf: the largest words of the merge);
        // each wave's counters + staging buffer (12 EPT * 32 B) take the place of the exchange buffer until the barrier
#pragma unroll
        for (int r = 0; r < EPT; ++r) {
          const bool live = key[r] != __builtin_inff();
          key[r] = live ? as_f((int)(((unsigned)as_i(key[r]) & ~1u) | (unsigned)wave)) : key[r];
        }
        wave_sort_binned<EPT, false>(key, lane, count, lds + wave * (binsort_bins<EPT>() + CHUNK));
#pragma unroll
        for (int r = 0; r < EPT; ++r) pk[r] = (unsigned)as_i(key[r]);
        __syncthreads();                                    // both sorts done: their scratch becomes the exchange buffer
      } else {
#pragma unroll
        for (int r = 0; r < EPT; ++r) {
          const bool live = key[r] != __builtin_inff();       // load_coords pads with +inf; no coordinate is +inf
          pk[r] = live ? (((unsigned)as_i(key[r]) & ~1u) | (unsigned)wave) : 0xffffffffu;
        }
        wave_sort<EPT>(pk, lane);
      }
    }
  }
  // ---- merge the two sorted sequences: flip stage between the waves, the rest inside each wave ----
#pragma unroll
  for (int r = 0; r < EPT; ++r) buf[r * 128 + wave * 64 + lane] = pk[r];
  __syncthreads();
#pragma unroll
  for (int r = 0; r < EPT; ++r) {
    const unsigned p = buf[(EPT - 1 - r) * 128 + (1 - wave) * 64 + (63 - lane)];
    pk[r] = wave ? (pk[r] > p ? pk[r] : p) : (pk[r] < p ? pk[r] : p);
  }
  xlane_stages<U32Keys, EPT, 32>(pk, lane);
  if constexpr (EPT > 1) lane_stages<U32Keys, EPT, EPT / 2>(pk);
  // merged position of pk[r]: g = wave*CHUNK + lane*EPT + r; live iff g < n + m (pads are the largest words)
  unsigned tagmask = 0u;                                  // GRAD: bit r = cloud of merged atom r (1 = target)
  if constexpr (GRAD) {
#pragma unroll
    for (int r = 0; r < EPT; ++r) tagmask |= (pk[r] & 1u) << r;
  }

  // ---- level numerators and gaps -----------------------------------------------------------------
  const int g0 = wave * CHUNK + lane * EPT;
  int tags_in_lane = 0;
#pragma unroll
  for (int r = 0; r < EPT; ++r) tags_in_lane += (g0 + r < total_live) ? (int)(pk[r] & 1u) : 0;
  int before = wave_exclusive_scan(tags_in_lane, lane);                  // target atoms in lower lanes
  const int wave_tags = __builtin_amdgcn_readlane(before + tags_in_lane, 63);
  const float first_val = as_f((int)(pk[0] & ~1u));                      // this lane's first key, for its left neighbour
  const float next_lane_first = as_f(__builtin_amdgcn_ds_bpermute(min(lane + 1, 63) << 2, as_i(first_val)));
  if (lane == 0) {
    redi[4 + wave] = wave_tags;
    red[6 + wave] = first_val;                                           // wave 1's first key is wave 0's last successor
  }
  __syncthreads();
  if (wave == 1) before += redi[4];
  const float other_first = red[7];
  // num[r]: level numerator after merged atom g0 + r.  val[r]: its coordinate, 1 for a pad; val[EPT]: the
  // successor of the lane's last atom (1 past the last live atom: the reference's pad value, :237).  The gap
  // of atom r is then val[r + 1] - val[r] (0 for pads) and is recomputed where it is needed: keeping it in
  // registers next to num and val costs the fifth wave per SIMD.
  int num[EPT];
  float val[EPT + 1];
  int lo_num = 0x7fffffff, hi_num = -0x7fffffff;
  {
    int cv = before;
#pragma unroll
    for (int r = 0; r < EPT; ++r) {
      const int g = g0 + r;
      const bool live = g < total_live;
      cv += live ? (int)(pk[r] & 1u) : 0;
      const int cu = g + 1 - cv;
      num[r] = cu * mg - cv * ng;
      val[r] = live ? as_f((int)(pk[r] & ~1u)) : 1.f;
      lo_num = live ? min(lo_num, num[r]) : lo_num;
      hi_num = live ? max(hi_num, num[r]) : hi_num;
    }
    const float nxt = (lane < 63) ? next_lane_first : other_first;   // (wave 1, lane 63: g0 + EPT >= n + m always)
    val[EPT] = (g0 + EPT < total_live) ? nxt : 1.f;
  }
  float wsum = val[EPT] - val[0];                                       // the lane's gaps telescope
  lo_num = wave_min_int(lo_num, lane);
  hi_num = -wave_min_int(-hi_num, lane);
  wsum = wave_sum(wsum, lane);
  if (lane == 0) { redi[8 + wave] = lo_num; redi[10 + wave] = hi_num; red[12 + wave] = wsum; }
  __syncthreads();
  int lo = __builtin_amdgcn_readfirstlane(min(redi[8], redi[9]));
  int hi = __builtin_amdgcn_readfirstlane(max(redi[10], redi[11]));
  {
    const float total = red[12] + red[13];
    if (!(total >= 0.5f)) hi = lo;                   // degenerate (reference: argmin of an all-inf row = index 0)
    hi = __builtin_amdgcn_readfirstlane(hi);
  }

  // ---- weighted median: smallest level whose cumulated gap weight reaches 0.5 (:239-245) ----------
  int parity = 0;
  for (int it = 0; it < 34 && lo < hi; ++it) {       // <= ceil(log2(range)) <= 32 trips; both waves agree
    const int mid = lo + ((hi - lo) >> 1);
    float w = 0.f;
#pragma unroll
    for (int r = 0; r < EPT; ++r) w += (num[r] <= mid) ? (val[r + 1] - val[r]) : 0.f;
    w = wave_sum(w, lane);
    if (lane == 0) red[parity * 2 + wave] = w;
    __syncthreads();
    const float below = red[parity * 2] + red[parity * 2 + 1];
    if (below >= 0.5f) hi = mid; else lo = mid + 1;
    lo = __builtin_amdgcn_readfirstlane(lo);
    hi = __builtin_amdgcn_readfirstlane(hi);
    parity ^= 1;
  }
  const int med = lo;

  // ---- gradient coefficients ----------------------------------------------------------------------
  if constexpr (GRAD) {
    float* stage = lds;                                   // [2][CHUNK]: source row, target row (by original index)
    int cv = before;
#pragma unroll
    for (int r = 0; r < EPT; ++r) {
      const int g = g0 + r;
      const bool live = g < total_live;
      const int t = (int)((tagmask >> r) & 1u);
      cv += live ? t : 0;
      const int rank = t ? cv - 1 : g - cv;               // position inside the atom's own sorted cloud
      const int pred = t ? num[r] + ng : num[r] - mg;     // level before the atom's own weight
      const float before_abs = (g == 0) ? 0.f : (float)abs(pred - med);
      const float coef = (before_abs - (float)abs(num[r] - med)) * inv_lcm;
      if (live) {
        const int id = sidx[t * CHUNK + lds_slot<EPT>(rank)];
        stage[t * CHUNK + id] = coef;
      }
    }
    __syncthreads();
    float* cs = A.coef_s + (long)s * n;
    float* ct = A.coef_t + (long)s * m;
#pragma unroll
    for (int r = 0; r < EPT / 2 + (EPT == 1 ? 1 : 0); ++r) {
      const int i = r * 128 + (int)threadIdx.x;
      if (i < n) cs[i] = stage[i];
      if (i < m) ct[i] = stage[CHUNK + i];
    }
  }

  // ---- cost ------------------------------------------------------------------------------------------
  float acc = 0.f;
#pragma unroll
  for (int r = 0; r < EPT; ++r) acc += (val[r + 1] - val[r]) * (float)abs(num[r] - med);
  acc = wave_sum(acc, lane);
  if (lane == 0) red[14 + wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    A.slice_cost[s] = (red[14] + red[15]) * inv_lcm;
    if (A.slice_shift) A.slice_shift[s] = med;
  }
}

template <int EPT>
static int launch_level_median_merge(SswArgs& A, int mg, int ng, float inv_lcm, hipStream_t stream) {
  const long total = (long)A.pairs * A.slices;
  if (total > 0x7fffffffL) return (int)hipErrorInvalidValue;
  A.num_groups = (int)total;
  if (A.coef_s != nullptr) {
    const size_t lds = (size_t)(EPT * 128 + 16) * sizeof(float) + (size_t)2 * EPT * kWave * sizeof(unsigned short);
    hipLaunchKernelGGL((ssw_level_median_merge_kernel<EPT, true>), dim3((unsigned)total), dim3(128), lds, stream, A, mg,
                       ng, inv_lcm);
  } else {
    const size_t lds = (size_t)(merge_loss_lds_floats<EPT>() + 16) * sizeof(float);
    hipLaunchKernelGGL((ssw_level_median_merge_kernel<EPT, false>), dim3((unsigned)total), dim3(128), lds, stream, A,
                       mg, ng, inv_lcm);
  }
  return (int)hipGetLastError();
}

// p = 1 for max(n, m) <= 2048, with or without coefficients (called from dispatch_level_median, shw_ssw_p1.hip)
int dispatch_level_median_merge(SswArgs& A, int mg, int ng, float inv_lcm, hipStream_t stream) {
  switch (ept_for(A.n, A.m)) {
#ifdef SHW_DEV_ONLY_EPT
    case SHW_DEV_ONLY_EPT: return launch_level_median_merge<SHW_DEV_ONLY_EPT>(A, mg, ng, inv_lcm, stream);
#else
    case 1: return launch_level_median_merge<1>(A, mg, ng, inv_lcm, stream);
    case 2: return launch_level_median_merge<2>(A, mg, ng, inv_lcm, stream);
    case 4: return launch_level_median_merge<4>(A, mg, ng, inv_lcm, stream);
    case 8: return launch_level_median_merge<8>(A, mg, ng, inv_lcm, stream);
    case 16: return launch_level_median_merge<16>(A, mg, ng, inv_lcm, stream);
    case 32: return launch_level_median_merge<32>(A, mg, ng, inv_lcm, stream);
#endif
    default: return (int)hipErrorInvalidValue;
  }
}

}  // namespace shw
