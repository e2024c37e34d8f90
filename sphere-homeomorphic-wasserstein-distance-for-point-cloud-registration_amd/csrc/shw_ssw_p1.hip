// shw_ssw_p1.hip -- p == 1: the reference's level-median closed form for W_1 on the circle
// (emd1D_circle, max_spherical_sliced_w.py:210-247), one wavefront per (pair, slice), any n and m.
//
// The reference sorts u and v, merges them (stable: u before v on equal values), accumulates the
// signed weights (+1/n per source atom, -1/m per target atom) into the CDF difference `level`,
// weights every merged atom with the gap to its successor (the last one with 1 - value; the segment
// [0, first atom) is NOT integrated -- SURVEY.md 8a row A7, a quirk this kernel reproduces), takes
// the weighted median of the levels at the fixed threshold 0.5 and returns sum gap * |level - median|.
//
// Here nothing is merged or re-sorted.  Both sorted arrays are parked in LDS; every atom finds its
// place in the *other* array by a branch-free binary search, which gives, in closed form,
//     source atom i :  level = (i+1)/n - lb/m ,  lb = #{v <  u_i},  successor = min(u_{i+1}, v_lb)
//     target atom j :  level = ub/n - (j+1)/m ,  ub = #{u <= v_j},  successor = min(v_{j+1}, u_ub)
// Levels are kept as exact integers  num = (#u)*(m/g) - (#v)*(n/g),  g = gcd(n, m),  level = num / lcm(n, m),
// so the weighted median is a bisection over an integer range with one wave-wide masked sum per step.
//
// With coefficients requested (training), the sorts carry the original point index and every atom
// also writes  d cost / d coordinate = (|level_pred - med| - |level - med|) / lcm  (first merged atom:
// -|level - med| / lcm), where level_pred is the level before the atom's own weight was added.
#include "ssw_common.hpp"

namespace shw {

// number of keys < val (STRICT = true) or <= val (STRICT = false) among the ascending, +inf padded
// array of 64*EPT keys in LDS (layout lds_slot)
template <int EPT, bool STRICT>
__device__ __forceinline__ int count_below(const float* buf, float val) {
  constexpr int P = EPT * kWave;
  int pos = 0;
#pragma unroll
  for (int s = P / 2; s >= 1; s >>= 1) {
    const float probe = buf[lds_slot<EPT>(pos + s - 1)];
    const bool go = STRICT ? (probe < val) : (probe <= val);
    pos += go ? s : 0;
  }
  const float probe = buf[lds_slot<EPT>(pos)];
  const bool go = STRICT ? (probe < val) : (probe <= val);
  return pos + (go ? 1 : 0);
}

template <int EPT>
__device__ __forceinline__ float key_at(const float* buf, int pos, int count) {
  // value of sorted position pos, +inf past the end (clamped address keeps the read in bounds)
  const float v = buf[lds_slot<EPT>(min(pos, EPT * kWave - 1))];
  return pos < count ? v : __builtin_inff();
}

__device__ __forceinline__ int wave_min_i(int v, int lane) {
  v = min(v, as_i(lane_xor<1>(as_f(v), lane)));
  v = min(v, as_i(lane_xor<2>(as_f(v), lane)));
  v = min(v, as_i(lane_xor<4>(as_f(v), lane)));
  v = min(v, as_i(lane_xor<8>(as_f(v), lane)));
  v = min(v, as_i(lane_xor<16>(as_f(v), lane)));
  v = min(v, as_i(lane_xor<32>(as_f(v), lane)));
  return v;
}

template <int EPT, int WAVES, bool GRAD>
__global__ __launch_bounds__(WAVES * 64) void ssw_level_median_kernel(SswArgs A, int mg, int ng, float inv_lcm) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int ROW = EPT * kWave;
  constexpr int ARRAYS = GRAD ? 4 : 2;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* ubuf = lds + wave * (ARRAYS * ROW);
  float* vbuf = ubuf + ROW;
  int* uidx = reinterpret_cast<int*>(vbuf + ROW);     // GRAD only
  int* vidx = uidx + ROW;

  const int vid = xcd_contiguous_id(blockIdx.x, A.num_groups);
  const int s = vid * WAVES + wave;
  if (s >= A.pairs * A.slices) return;
  const int b = s / A.slices, l = s - b * A.slices;
  const int n = A.n, m = A.m;

  float U[6];
  load_frame(A.dirs, (long)b * A.u_pair_stride + (long)l * 6, U);   // (3,2) row-major: U[2*d + k]

  // ---- project + sort both clouds, park them in LDS --------------------------------------------
#pragma nounroll
  for (int which = 0; which < 2; ++which) {
    const float* X = which == 0 ? A.xt + (long)b * m * A.pstride : A.xs + (long)b * n * A.pstride;
    const int count = which == 0 ? m : n;
    float* dst = which == 0 ? vbuf : ubuf;
    int ln = lane;
    asm volatile("" : "+v"(ln));
    float key[EPT];
    load_coords<EPT>(X, count, ln, U, key);
    if constexpr (GRAD) {
      int* dsti = which == 0 ? vidx : uidx;
      item_t item[EPT];
#pragma unroll
      for (int r = 0; r < EPT; ++r) item[r] = make_item(key[r], r * kWave + ln);
      wave_sort_kv<EPT>(item, ln);
#pragma unroll
      for (int r = 0; r < EPT; ++r) {
        dst[r * kWave + lane] = item_key(item[r]);
        dsti[r * kWave + lane] = item_idx(item[r]);
      }
    } else {
      wave_sort<EPT>(key, ln);
#pragma unroll
      for (int r = 0; r < EPT; ++r) dst[r * kWave + lane] = key[r];
    }
  }
  __builtin_amdgcn_wave_barrier();

  // ---- every atom: level numerator and gap to its merged successor ------------------------------
  // index 0..EPT-1: source atoms (sorted position lane*EPT + r), EPT..2*EPT-1: target atoms
  int num[2 * EPT];
  float gap[2 * EPT];
  int lo_num = 0x7fffffff, hi_num = -0x7fffffff;
#pragma unroll
  for (int r = 0; r < EPT; ++r) {
    const int e = lane * EPT + r;
    {  // source atom
      const float val = ubuf[r * kWave + lane];
      const int lb = count_below<EPT, true>(vbuf, val);             // padded +inf keys are never < val
      const float nxt = fminf(key_at<EPT>(ubuf, e + 1, n), key_at<EPT>(vbuf, lb, m));
      const bool live = e < n;
      const float succ = (nxt == __builtin_inff()) ? 1.f : nxt;     // last merged atom: pad value 1 (:237)
      gap[r] = live ? succ - val : 0.f;
      num[r] = (e + 1) * mg - min(lb, m) * ng;
      lo_num = live ? min(lo_num, num[r]) : lo_num;
      hi_num = live ? max(hi_num, num[r]) : hi_num;
    }
    {  // target atom
      const float val = vbuf[r * kWave + lane];
      const int ub = count_below<EPT, false>(ubuf, val);
      const float nxt = fminf(key_at<EPT>(vbuf, e + 1, m), key_at<EPT>(ubuf, ub, n));
      const bool live = e < m;
      const float succ = (nxt == __builtin_inff()) ? 1.f : nxt;
      gap[EPT + r] = live ? succ - val : 0.f;
      num[EPT + r] = min(ub, n) * mg - (e + 1) * ng;
      lo_num = live ? min(lo_num, num[EPT + r]) : lo_num;
      hi_num = live ? max(hi_num, num[EPT + r]) : hi_num;
    }
  }
  lo_num = __builtin_amdgcn_readfirstlane(wave_min_i(lo_num, lane));
  hi_num = __builtin_amdgcn_readfirstlane(-wave_min_i(-hi_num, lane));

  // ---- weighted median: smallest level whose cumulated gap weight reaches 0.5 -------------------
  // (reference :239-245; if the total never reaches 0.5 its argmin over an all-inf row is index 0,
  //  i.e. the smallest level)
  int lo = lo_num, hi = hi_num;          // invariant: answer in [lo, hi] if W(hi) >= 0.5
  {
    float w = 0.f;
#pragma unroll
    for (int t = 0; t < 2 * EPT; ++t) w += gap[t];
    const float total = wave_sum(w, lane);
    if (!(total >= 0.5f)) hi = lo;       // degenerate: median = smallest level
  }
  while (lo < hi) {                      // wave-uniform: <= ceil(log2(hi_num - lo_num + 1)) <= 27 steps
    const int mid = lo + ((hi - lo) >> 1);
    float w = 0.f;
#pragma unroll
    for (int t = 0; t < 2 * EPT; ++t) w += (num[t] <= mid) ? gap[t] : 0.f;
    const float below = wave_sum(w, lane);
    if (below >= 0.5f) hi = mid; else lo = mid + 1;
    lo = __builtin_amdgcn_readfirstlane(lo);
    hi = __builtin_amdgcn_readfirstlane(hi);
  }
  const int med = lo;

  // ---- cost (and coefficients) ------------------------------------------------------------------
  float acc = 0.f;
#pragma unroll
  for (int t = 0; t < 2 * EPT; ++t) acc += gap[t] * (float)abs(num[t] - med);
  const float cost = wave_sum(acc, lane) * inv_lcm;
  if (lane == 0) {
    A.slice_cost[s] = cost;
    if (A.slice_shift) A.slice_shift[s] = med;
  }
  if constexpr (GRAD) {
    float* cs = A.coef_s + (long)s * n;
    float* ct = A.coef_t + (long)s * m;
#pragma unroll
    for (int r = 0; r < EPT; ++r) {
      const int e = lane * EPT + r;
      if (e < n) {
        // merged rank = e + lb; lb recovered from the level: lb*ng = (e+1)*mg - num
        const int lb_ng = (e + 1) * mg - num[r];
        const bool first = (e == 0) && (lb_ng == 0);
        const float before = first ? 0.f : (float)abs(num[r] - mg - med);
        cs[uidx[r * kWave + lane]] = (before - (float)abs(num[r] - med)) * inv_lcm;
      }
      if (e < m) {
        const int ub_mg = num[EPT + r] + (e + 1) * ng;
        const bool first = (e == 0) && (ub_mg == 0);
        const float before = first ? 0.f : (float)abs(num[EPT + r] + ng - med);
        ct[vidx[r * kWave + lane]] = (before - (float)abs(num[EPT + r] - med)) * inv_lcm;
      }
    }
  }
}

static int gcd_int(int a, int b) {
  while (b) { const int t = a % b; a = b; b = t; }
  return a;
}

template <int EPT, int WAVES>
static int launch_level_median(SswArgs& A, hipStream_t stream) {
  const long total = (long)A.pairs * A.slices;
  const long groups = (total + WAVES - 1) / WAVES;
  if (groups > 0x7fffffffL) return (int)hipErrorInvalidValue;
  A.num_groups = (int)groups;
  const int g = gcd_int(A.n, A.m);
  const int mg = A.m / g, ng = A.n / g;
  const float inv_lcm = 1.f / ((float)A.n * (float)mg);
  const bool grad = A.coef_s != nullptr;
  const size_t lds = (size_t)WAVES * (grad ? 4 : 2) * EPT * kWave * sizeof(float);
  if (grad) {
    hipLaunchKernelGGL((ssw_level_median_kernel<EPT, WAVES, true>), dim3((unsigned)groups), dim3(WAVES * 64), lds,
                       stream, A, mg, ng, inv_lcm);
  } else {
    hipLaunchKernelGGL((ssw_level_median_kernel<EPT, WAVES, false>), dim3((unsigned)groups), dim3(WAVES * 64), lds,
                       stream, A, mg, ng, inv_lcm);
  }
  return (int)hipGetLastError();
}

int dispatch_level_median_merge(SswArgs& A, int mg, int ng, float inv_lcm, hipStream_t stream);
int dispatch_level_median_coop(SswArgs& A, int mg, int ng, float inv_lcm, hipStream_t stream);   // shw_ssw_p1_coop.hip
bool level_median_coop_trains(int n, int m);
int level_median_coop_slots(int total);

// SHW_P1_SEARCH_KERNEL=1 (diagnostic): use the one-wave search kernel at every size
static bool p1_search_kernel_forced() {
  static const bool forced = [] { const char* e = getenv("SHW_P1_SEARCH_KERNEL"); return e && e[0] == '1'; }();
  return forced;
}

// SHW_P1_KERNEL=coop | merge (diagnostic / A-B): the cooperative kernel of shw_ssw_p1_coop.hip, or the two-wave merge
// kernel, wherever they can run
static int p1_kernel_forced() {
  static const int forced = [] { const char* e = getenv("SHW_P1_KERNEL"); return e ? (e[0] == 'c' ? 1 : (e[0] == 'm' ? 2 : 0)) : 0; }();
  return forced;
}

int dispatch_level_median(SswArgs& A, hipStream_t stream) {
  const bool small = A.n <= 2048 && A.m <= 2048;
  const bool grad = A.coef_s != nullptr;
  // cooperative kernel (one distribution sort of the tagged concatenation, shw_ssw_p1_coop.hip): every shape above 2048
  // points; at or below, the loss from 1025 merged atoms on (measured at n = m = 2048 / 1024: 0.42 / 0.20 ms against the
  // merge kernel's 0.54 / 0.25) -- training stays with the merge kernel there (0.97 / 0.42 against 0.94 / 0.45 ms)
  // round 3: ... unless the cooperative kernel's class (20 / 24 / 32 merged atoms per lane) is smaller than the merge
  // kernel's two power-of-two halves by more than the 12 % the merge kernel is faster per slot (n = m = 1200: 2560 against
  // 4096 slots, 0.52 against 0.74 ms per step)
  bool coop = small ? (A.n + A.m > 1024 && (!grad || 9 * level_median_coop_slots(A.n + A.m) < 8 * 128 * ept_for(A.n, A.m)))
                    : (!grad || level_median_coop_trains(A.n, A.m));
  if (p1_kernel_forced() == 1 && A.n + A.m > 1024) coop = true;
  if (p1_kernel_forced() == 2 && small) coop = false;
  if (p1_search_kernel_forced()) coop = false;
  if (coop || (small && !p1_search_kernel_forced())) {
    const int g = gcd_int(A.n, A.m);
    const int mg = A.m / g, ng = A.n / g;
    const float inv_lcm = 1.f / ((float)A.n * (float)mg);
    // (merge kernel: two waves per slice, merge by the sorting network, shw_ssw_p1_merge.hip)
    return coop ? dispatch_level_median_coop(A, mg, ng, inv_lcm, stream) : dispatch_level_median_merge(A, mg, ng, inv_lcm, stream);
  }
  switch (ept_for(A.n, A.m)) {
#ifdef SHW_DEV_ONLY_EPT
    case SHW_DEV_ONLY_EPT: return launch_level_median<SHW_DEV_ONLY_EPT, 1>(A, stream);
#else
    case 1: return launch_level_median<1, 4>(A, stream);
    case 2: return launch_level_median<2, 4>(A, stream);
    case 4: return launch_level_median<4, 4>(A, stream);
    case 8: return launch_level_median<8, 4>(A, stream);
    case 16: return launch_level_median<16, 2>(A, stream);
    case 32: return launch_level_median<32, 1>(A, stream);
    case 64: return launch_level_median<64, 1>(A, stream);
    case 128: return launch_level_median<128, 1>(A, stream);
#endif
    default: return (int)hipErrorInvalidValue;
  }
}

}  // namespace shw
