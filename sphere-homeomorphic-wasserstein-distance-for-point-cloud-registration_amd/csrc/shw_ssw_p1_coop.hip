// shw_ssw_p1_coop.hip -- p == 1, 2048 < max(n, m) <= 8192 (loss and training): the level-median closed form of emd1D_circle
// (max_spherical_sliced_w.py:210-247; shw_ssw_p1.hip has the formula and the reference's quirks) with the MERGE of
// the two clouds done by ONE cooperative distribution sort (the one-wave search kernel it
// replaces runs 64 / 128 atoms per lane and 12-13 LDS probes per atom: 1.7 / 3.8 ms per launch at 4096 / 8192 points).
//
// W = next_pow2(n + m) / 2048 wavefronts (4 or 8) of one workgroup per (pair, slice), 32 merged atoms per lane.
// Every lane projects its share of the CONCATENATED clouds (atoms [0, n): source, [n, n + m): target) and records the
// cloud in the lowest mantissa bit of the coordinate (source 0, target 1: on equal coordinates the source atom comes
// first, the reference's stable merge order :232-235 -- the device of shw_ssw_p1_merge.hip).  Sorting those n + m
// tagged keys IS the merge: coop_sort (coop_sort.hpp) leaves them in merged order, 32 consecutive positions per lane.
// From there everything is arithmetic on registers, as in the two-wave merge kernel, with W-way exchanges through LDS:
//     #target atoms up to position g  = prefix sum of the tag bits (in-lane, lane scan, wave offsets)
//     level numerator                 = (#source)*(m/g) - (#target)*(n/g)      (exact integers, g = gcd)
//     gap to the merged successor     = next key - key  (last live atom: 1 - key; [0, first atom) is not integrated)
//     weighted median                 = integer bisection, one masked sum per step, added across the waves in wave order
// Clearing the tag bit moves a coordinate by at most one ulp (6e-8): far inside the 1e-5 parity tolerance.
#include "coop_sort_kv.hpp"
#include "ssw_common.hpp"

namespace shw {

__device__ __forceinline__ int p1c_wave_exclusive_scan(int v, int lane) {
  const int incl = wave_inclusive_scan_dpp(v);
  return incl - v;
}

__device__ __forceinline__ int p1c_wave_min_int(int v, int lane) {
  v = min(v, as_i(lane_xor<1>(as_f(v), lane)));
  v = min(v, as_i(lane_xor<2>(as_f(v), lane)));
  v = min(v, as_i(lane_xor<4>(as_f(v), lane)));
  v = min(v, as_i(lane_xor<8>(as_f(v), lane)));
  v = min(v, as_i(lane_xor<16>(as_f(v), lane)));
  v = min(v, as_i(lane_xor<32>(as_f(v), lane)));
  return v;
}

// GRAD (training): the same on 64-bit items (tagged coordinate bits << 32 | index inside the atom's own
// cloud; coop_sort_kv.hpp): the merged atom carries its original index, so
//     d cost / d coordinate = (|level_before - med| - |level - med|) / lcm      (first merged atom: -|level - med| / lcm)
// goes straight into the cloud's staging row (the item buffer's place) and is stored coalesced: every coefficient
// written exactly once, no atomics (rows feed ssw_backward_points_kernel).
// (training at W = 8: four keys per bin instead of two -- 16 KB of counters beside the 128 KB item buffer)
#ifndef SHW_P1C_GRAD_KPB
#define SHW_P1C_GRAD_KPB 4     // keys per bin of the training form (items: 4096 points 0.98 ms per step, with 2: 1.59)
#endif
template <int W, bool GRAD>
constexpr int p1c_keys_per_bin() { return GRAD ? (W == 8 ? 4 : SHW_P1C_GRAD_KPB) : SHW_COOP_KEYS_PER_BIN; }

template <int EPT, int W, bool GRAD>
__global__ __launch_bounds__(W * 64) void ssw_level_median_coop_kernel(SswArgs A, int mg, int ng, float inv_lcm) {
  constexpr int KPB = p1c_keys_per_bin<W, GRAD>();
  typedef Coop<EPT, W, KPB> C;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  unsigned* cnt = reinterpret_cast<unsigned*>(lds);
  float* buf = lds + C::NB;                                     // C::CAP floats, GRAD: 2 C::CAP (items)
  int* red = reinterpret_cast<int*>(lds + C::NB + (GRAD ? 2 : 1) * C::CAP);   // coop_sort: [0, 2W); then the exchanges below
  float* redf = reinterpret_cast<float*>(red);
  static_assert(C::RED >= 9 * W, "cross-wave scratch");
  int* x_tags = red;                                            // [W] target atoms per wave
  float* x_first = redf + W;                                    // [W] first key of each wave
  int* x_lo = red + 2 * W;                                      // [W] smallest / [W] largest level numerator
  int* x_hi = red + 3 * W;
  float* x_gap = redf + 4 * W;                                  // [W] gap totals
  float* x_med = redf + 5 * W;                                  // [2][W] median partial sums (two parities)
  float* x_cost = redf + 7 * W;                                 // [W] cost partial sums

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int gl = wave * 64 + lane;
  const int s = xcd_contiguous_id(blockIdx.x, A.num_groups);   // one workgroup per (pair, slice)
  const int b = s / A.slices, l = s - b * A.slices;
  const int n = A.n, m = A.m, total_live = n + m;

  float U[6];
  load_frame(A.dirs, (long)b * A.u_pair_stride + (long)l * 6, U);   // (3,2) row-major: U[2*d + k]
  const bool rows = U[0] != U[0];                               // coordinate-row mode (shw_circle_ot)
  const int wide = rows ? 0 : -1;
  const int o1 = rows ? 0 : 1, o2 = rows ? 0 : 2;
  const float* Xs = A.xs + (long)b * n * A.pstride;
  const float* Xt = A.xt + (long)b * m * A.pstride;

  coop_zero_counters<EPT, W, KPB>(cnt, gl);
  // ---- project the concatenated clouds: lane owns atoms r*64W + gl ------------------------------------------------
  float key[EPT];
  constexpr int CH = chunk_of(EPT);
#pragma unroll
  for (int r0 = 0; r0 < EPT; r0 += CH) {
    float px[CH], py[CH], pz[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const int i = (r0 + j) * C::NCOL + gl;
      const bool src = i < n;
      const int q = src ? i : min(i - n, m - 1);                // (clamp: branch-free, always in bounds)
      const float* X = src ? Xs : Xt;
      const int q3 = q + ((q & wide) << 1);                     // 3 q, or q for rows of coordinates
      px[j] = X[q3]; py[j] = X[q3 + o1]; pz[j] = X[q3 + o2];
    }
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const int i = (r0 + j) * C::NCOL + gl;
      const float a = fmaf(pz[j], U[4], fmaf(py[j], U[2], fmaf(px[j], U[0], 0.f)));   // (see load_coords)
      const float bb = fmaf(pz[j], U[5], fmaf(py[j], U[3], fmaf(px[j], U[1], 0.f)));
      float c = circle_coord(a, bb);
      c = rows ? px[j] : c;
      const unsigned tagged = ((unsigned)as_i(c) & ~1u) | (i < n ? 0u : 1u);
      key[r0 + j] = i < total_live ? as_f((int)tagged) : __builtin_inff();
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  __syncthreads();                                              // counters zeroed
  int idx[GRAD ? EPT : 1];
  if constexpr (GRAD) {
    item_t it[EPT];
#pragma unroll
    for (int r = 0; r < EPT; ++r) {
      const int i = r * C::NCOL + gl;
      it[r] = make_item(key[r], i < n ? i : i - n);
    }
    coop_sort_kv<EPT, W, false, KPB>(it, wave, lane, total_live, cnt, reinterpret_cast<item_t*>(buf), red);
#pragma unroll
    for (int r = 0; r < EPT; ++r) { key[r] = item_key(it[r]); idx[r] = item_idx(it[r]); }
  } else {
    coop_sort<EPT, W, false>(key, wave, lane, total_live, cnt, buf, red);
  }
  // merged position of key[r]: g = gl*EPT + r; live iff g < n + m (pads are +inf)

  // ---- level numerators and gaps -------------------------------------------------------------------------------------
  const int g0 = gl * EPT;
  int tags_in_lane = 0;
#pragma unroll
  for (int r = 0; r < EPT; ++r) tags_in_lane += (g0 + r < total_live) ? (as_i(key[r]) & 1) : 0;
  int before = p1c_wave_exclusive_scan(tags_in_lane, lane);     // target atoms in lower lanes of this wave
  const int wave_tags = __builtin_amdgcn_readlane(before + tags_in_lane, 63);
  const float first_val = as_f(as_i(key[0]) & ~1);              // this lane's first key, for its left neighbour
  const float next_lane_first = as_f(__builtin_amdgcn_ds_bpermute(min(lane + 1, 63) << 2, as_i(first_val)));
  __syncthreads();                                              // (the sort's last use of `red`)
  if (lane == 0) { x_tags[wave] = wave_tags; x_first[wave] = first_val; }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < W; ++q) before += (q < wave) ? x_tags[q] : 0;
  const float other_first = x_first[min(wave + 1, W - 1)];      // the next wave's first key: this wave's last successor
  int num[EPT];
  float val[EPT + 1];
  int lo_num = 0x7fffffff, hi_num = -0x7fffffff;
  {
    int cv = before;
#pragma unroll
    for (int r = 0; r < EPT; ++r) {
      const int g = g0 + r;
      const bool live = g < total_live;
      cv += live ? (as_i(key[r]) & 1) : 0;
      const int cu = g + 1 - cv;
      num[r] = cu * mg - cv * ng;
      val[r] = live ? as_f(as_i(key[r]) & ~1) : 1.f;
      lo_num = live ? min(lo_num, num[r]) : lo_num;
      hi_num = live ? max(hi_num, num[r]) : hi_num;
    }
    const float nxt = (lane < 63) ? next_lane_first : other_first;   // (last wave, lane 63: g0 + EPT >= n + m always)
    val[EPT] = (g0 + EPT < total_live) ? nxt : 1.f;
  }
  float wsum = val[EPT] - val[0];                               // the lane's gaps telescope
  lo_num = p1c_wave_min_int(lo_num, lane);
  hi_num = -p1c_wave_min_int(-hi_num, lane);
  wsum = wave_sum(wsum, lane);
  if (lane == 0) { x_lo[wave] = lo_num; x_hi[wave] = hi_num; x_gap[wave] = wsum; }
  __syncthreads();
  int lo = 0x7fffffff, hi = -0x7fffffff;
  float total = 0.f;
#pragma unroll
  for (int q = 0; q < W; ++q) { lo = min(lo, x_lo[q]); hi = max(hi, x_hi[q]); total += x_gap[q]; }
  if (!(total >= 0.5f)) hi = lo;                                // degenerate (reference: argmin of an all-inf row = index 0)
  lo = __builtin_amdgcn_readfirstlane(lo);
  hi = __builtin_amdgcn_readfirstlane(hi);

  // ---- weighted median: smallest level whose cumulated gap weight reaches 0.5 (:239-245) ------------------------------
  int parity = 0;
  for (int it = 0; it < 34 && lo < hi; ++it) {                  // <= ceil(log2(range)) <= 32 trips; all waves agree
    const int mid = lo + ((hi - lo) >> 1);
    float w = 0.f;
#pragma unroll
    for (int r = 0; r < EPT; ++r) w += (num[r] <= mid) ? (val[r + 1] - val[r]) : 0.f;
    w = wave_sum(w, lane);
    if (lane == 0) x_med[parity * W + wave] = w;
    __syncthreads();
    float below = 0.f;
#pragma unroll
    for (int q = 0; q < W; ++q) below += x_med[parity * W + q];
    if (below >= 0.5f) hi = mid; else lo = mid + 1;
    lo = __builtin_amdgcn_readfirstlane(lo);
    hi = __builtin_amdgcn_readfirstlane(hi);
    parity ^= 1;
  }
  const int med = lo;

  // ---- gradient coefficients --------------------------------------------------------------------------------------------
  if constexpr (GRAD) {
    float* stage_s = buf;                                       // by original index; the item buffer is free (the sort's
    float* stage_t = buf + C::CAP;                              // last barrier is behind every wave)
#pragma unroll
    for (int r = 0; r < EPT; ++r) {
      const int g = g0 + r;
      const bool live = g < total_live;
      const int t = as_i(key[r]) & 1;
      const int pred = t ? num[r] + ng : num[r] - mg;           // level before the atom's own weight
      const float before_abs = (g == 0) ? 0.f : (float)abs(pred - med);
      const float coef = (before_abs - (float)abs(num[r] - med)) * inv_lcm;
      if (live) (t ? stage_t : stage_s)[idx[r]] = coef;
    }
    __syncthreads();
    float* cs = A.coef_s + (long)s * n;
    float* ct = A.coef_t + (long)s * m;
#pragma unroll
    for (int r = 0; r < EPT; ++r) {
      const int i = r * C::NCOL + gl;
      if (i < n) cs[i] = stage_s[i];
      if (i < m) ct[i] = stage_t[i];
    }
  }

  // ---- cost ------------------------------------------------------------------------------------------------------------
  float acc = 0.f;
#pragma unroll
  for (int r = 0; r < EPT; ++r) acc += (val[r + 1] - val[r]) * (float)abs(num[r] - med);
  acc = wave_sum(acc, lane);
  if (lane == 0) x_cost[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float c = 0.f;
#pragma unroll
    for (int q = 0; q < W; ++q) c += x_cost[q];
    A.slice_cost[s] = c * inv_lcm;
    if (A.slice_shift) A.slice_shift[s] = med;
  }
}

template <int EPT, int W, bool GRAD>
static int launch_level_median_coop(SswArgs& A, int mg, int ng, float inv_lcm, hipStream_t stream) {
  typedef Coop<EPT, W, p1c_keys_per_bin<W, GRAD>()> C;
  const long total = (long)A.pairs * A.slices;
  if (total > 0x7fffffffL) return (int)hipErrorInvalidValue;
  A.num_groups = (int)total;
  const size_t lds = (size_t)(C::LDS_FLOATS + (GRAD ? C::CAP : 0)) * sizeof(float);
  auto kern = ssw_level_median_coop_kernel<EPT, W, GRAD>;
  static bool raised[64] = {};          // once per instantiation and device (not inside a later stream capture)
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (lds > 64 * 1024 && !raised[dev & 63]) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)lds);
    if (e != hipSuccess) return (int)e;
    raised[dev & 63] = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(W * 64), lds, stream, A, mg, ng, inv_lcm);
  return (int)hipGetLastError();
}

// training form available?  (12 bytes of LDS per merged slot up to n + m = 8192, 9 above: four keys per bin)
bool level_median_coop_trains(int n, int m) { return next_pow2(n + m) <= 16384; }

// p = 1 (called from dispatch_level_median, shw_ssw_p1.hip); coef_s != NULL: + coefficients.  W waves of 20 / 24 / 32 merged
// atoms per lane: the smallest of the 12 classes (1280 ... 16384 slots) that holds n + m (round 3: 1200 + 1200 points
// pay for 2560 slots, not 4096; SHW_KPL_CLASSES=0 keeps 32 per lane)
static void level_median_coop_class(int total, int& W, int& ept) {
  static const bool fine = [] { const char* v = getenv("SHW_KPL_CLASSES"); return !(v && v[0] == '0'); }();
  W = 0; ept = 0;
  for (int w = 1; w <= 8 && W == 0; w *= 2) {
    for (int e : {20, 24, 32}) {
      if ((e == 32 || fine) && 64 * w * e >= total) { W = w; ept = e; break; }
    }
  }
}
// merged slots of the class that serves n + m atoms (0: none)
int level_median_coop_slots(int total) {
  int W, ept;
  level_median_coop_class(total, W, ept);
  return 64 * W * ept;
}

int dispatch_level_median_coop(SswArgs& A, int mg, int ng, float inv_lcm, hipStream_t stream) {
  const bool grad = A.coef_s != nullptr;
  int W, ept;
  level_median_coop_class(A.n + A.m, W, ept);
#define SHW_P1C_CASE(WW, EE)                                                                        \
  case WW * 100 + EE:                                                                               \
    return grad ? launch_level_median_coop<EE, WW, true>(A, mg, ng, inv_lcm, stream)                \
                : launch_level_median_coop<EE, WW, false>(A, mg, ng, inv_lcm, stream)
  switch (W * 100 + ept) {
#ifndef SHW_DEV_ONLY_EPT
    SHW_P1C_CASE(1, 20); SHW_P1C_CASE(1, 24); SHW_P1C_CASE(1, 32);
    SHW_P1C_CASE(2, 20); SHW_P1C_CASE(2, 24); SHW_P1C_CASE(2, 32);
    SHW_P1C_CASE(4, 20); SHW_P1C_CASE(4, 24); SHW_P1C_CASE(4, 32);
    SHW_P1C_CASE(8, 20); SHW_P1C_CASE(8, 24); SHW_P1C_CASE(8, 32);
#endif
    default: return (int)hipErrorInvalidValue;
  }
#undef SHW_P1C_CASE
}

}  // namespace shw
