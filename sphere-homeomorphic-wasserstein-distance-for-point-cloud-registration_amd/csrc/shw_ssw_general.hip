// shw_ssw_general.hip -- general circular OT for p != 1: different sizes (n != m) and/or non-uniform
// weights.  One wavefront per (pair, slice), like the fast kernels, but the solve follows the
// reference's algorithm step for step instead of the equal-size shortcut:
//
//   binary_search_circle (max_spherical_sliced_w.py:117-207): bisection over the cut theta in [-1, 1]
//   on the sign of the one-sided derivatives dCost (:25-65); exit when dC+ * dC- <= 0 or, once the
//   bracket is narrower than eps/L = 1e-7, through the tangent intersection (:189-200); the value is
//   Cost(theta) (:68-113), whose gradient w.r.t. the atoms is the loss gradient (theta is detached).
//
// Both sorted clouds live in LDS as (value, CDF) arrays.  The reference materialises the rotated
// target arrays, a merged CDF grid and a 2N sort per Cost call; here every atom locates itself in the
// other cloud's CDF by binary search (the rotated target CDF is evaluated on the fly, with the same
// fp32 operations the reference applies to v_cdf: subtract frac(theta), add 1 where negative), so one
// Cost / dCost evaluation is n + 2m independent searches and nothing is re-sorted.
//
// This is the compatibility path (~10x the work of the equal-size kernel): the trainers' "different
// source / target density" option (train_W_COS.py:292-293,334-336) and the u_weights / v_weights
// arguments (:289) land here.  Gradients are accumulated per sorted atom with LDS float atomics (a
// handful of terms per atom; their order, hence the last bit, may vary between runs).
#include "ssw_common.hpp"

namespace shw {

struct GeneralArgs {
  SswArgs base;
  const float* wu;        // (n) or (pairs, n) source weights, NULL = uniform 1/n
  const float* wv;        // (m) or (pairs, m) target weights, NULL = uniform 1/m
  long wu_pair_stride;    // 0 = shared by all pairs
  long wv_pair_stride;
  float* slice_theta;     // optional: the cut the solve ended on
};

// one cloud as the solver sees it: ascending atom values and their inclusive CDF, lds_slot layout
template <int EPT>
struct Side {
  const float* val;
  const float* cdf;
  int count;
  __device__ __forceinline__ float v(int i) const { return val[lds_slot<EPT>(i)]; }
  __device__ __forceinline__ float c(int i) const { return cdf[lds_slot<EPT>(i)]; }
  // number of atom VALUES < key (strict) or <= key (the p = 1 formula merges by value, not by CDF level)
  __device__ __forceinline__ int values_below(float key, bool strict) const {
    int lo = 0, hi = count;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      const float x = v(mid);
      const bool go = strict ? (x < key) : (x <= key);
      lo = go ? mid + 1 : lo;
      hi = go ? hi : mid;
    }
    return lo;
  }
  // number of CDF entries < key (strict) or <= key  == torch.searchsorted(cdf, key, right = !strict)
  __device__ __forceinline__ int below(float key, bool strict) const {
    int lo = 0, hi = count;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      const float x = c(mid);
      const bool go = strict ? (x < key) : (x <= key);
      lo = go ? mid + 1 : lo;
      hi = go ? hi : mid;
    }
    return lo;
  }
};

// the target after moving mass theta around the circle (reference :31-48, evaluated lazily)
template <int EPT>
struct Rotated {
  Side<EPT> t;
  float turns, frac;
  int start;                               // number of wrapped atoms = first atom of the rotated order
  __device__ __forceinline__ void set(const Side<EPT>& target, float theta) {
    t = target;
    turns = floorf(theta);
    frac = theta - turns;
    start = target.below(frac, true);      // (cdf - frac) < 0  <=>  cdf < frac
    if (start >= target.count) start = 0;  // degenerate (no atom left unwrapped): argmin over all-inf = 0
  }
  // atom j of the sorted target: shifted CDF and position unrolled onto the real line
  __device__ __forceinline__ void atom(int j, float& cdf, float& pos) const {
    const float sh = t.c(j) - frac;
    const bool wrapped = sh < 0.f;
    cdf = wrapped ? sh + 1.f : sh;
    pos = t.v(j) + (turns + (wrapped ? 1.f : 0.f));
  }
  // rotated index rho in [0, m]: rho = m is the appended copy of the first atom, one turn later
  __device__ __forceinline__ int source_index(int rho) const {
    const int j = rho + start;
    return j >= t.count ? j - t.count : j;
  }
  __device__ __forceinline__ float cdf_at(int rho) const { float c, p; atom(source_index(rho), c, p); return c; }
  __device__ __forceinline__ float pos_at(int rho) const {
    float c, p;
    if (rho >= t.count) { atom(start, c, p); return p + 1.f; }
    atom(source_index(rho), c, p);
    return p;
  }
  // number of rotated CDF entries strictly below key  == searchsorted(v_cdf_theta_rolled, key)
  __device__ __forceinline__ int below(float key) const {
    int lo = 0, hi = t.count;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      const bool go = cdf_at(mid) < key;
      lo = go ? mid + 1 : lo;
      hi = go ? hi : mid;
    }
    return lo;
  }
};

template <int PMODE>
__device__ __forceinline__ float powp(float d, float p, int p_int) { return pow_abs<PMODE>(d, p, p_int); }

// one-sided derivatives of the cost w.r.t. theta (reference dCost, :50-63), wave-uniform results
template <int EPT, int PMODE>
__device__ void cut_slopes(const Side<EPT>& S, const Side<EPT>& T, float theta, int lane, float p, int p_int,
                           float& d_plus, float& d_minus) {
  Rotated<EPT> R;
  R.set(T, theta);
  const int n = S.count, m = T.count;
  float sp = 0.f, sm = 0.f;
#pragma nounroll
  for (int r = 0; r < EPT; ++r) {
    const int j = lane * EPT + r;
    if (j < m) {
      float cdf, pos, ncdf, npos;
      R.atom(j, cdf, pos);
      const int jn = (j + 1 == m) ? 0 : j + 1;
      R.atom(jn, ncdf, npos);
      if (jn == R.start) npos += 1.f;                       // successor of the last rotated atom: first atom + 1
      const int il = min(S.below(cdf, true), n - 1);        // left-continuous source quantile (:50-51)
      const float al = S.v(il);
      int ir = S.below(cdf, false);                          // right-continuous on the extended arrays (:54-57)
      if (ir == n && S.c(0) + 1.f <= cdf) ir = n + 1;
      ir = min(ir, n);
      const float ar = ir < n ? S.v(ir) : S.v(0) + 1.f;
      sp += powp<PMODE>(al - npos, p, p_int) - powp<PMODE>(al - pos, p, p_int);
      sm += powp<PMODE>(ar - npos, p, p_int) - powp<PMODE>(ar - pos, p, p_int);
    }
  }
  d_plus = wave_sum_uniform(sp, lane);
  d_minus = wave_sum_uniform(sm, lane);
}

// transport cost at a fixed cut (reference Cost, :94-112).  GRAD: also accumulates
// d cost / d (sorted source atom) into gs and d cost / d (sorted target atom) into gt.
template <int EPT, int PMODE, bool GRAD>
__device__ float cut_cost(const Side<EPT>& S, const Side<EPT>& T, float theta, int lane, float p, int p_int,
                          float* gs, float* gt) {
  Rotated<EPT> R;
  R.set(T, theta);
  const int n = S.count, m = T.count;
  float acc = 0.f;
#pragma nounroll
  for (int r = 0; r < EPT; ++r) {
    const int e = lane * EPT + r;
    if (e < n) {                                             // grid point = source CDF level A_e
      const float g = S.c(e);
      const int cnt = R.below(g);                            // rotated target atom active at g
      const float b = R.pos_at(min(cnt, m));
      const float prev_a = e > 0 ? S.c(e - 1) : 0.f;
      const float prev_c = cnt > 0 ? R.cdf_at(cnt - 1) : 0.f;
      const float width = g - fmaxf(prev_a, prev_c);
      const float d = S.v(e) - b;
      acc += width * powp<PMODE>(d, p, p_int);
      if constexpr (GRAD) {
        const float w = width * dpow_abs<PMODE>(d, p, p_int);
        const int jt = (cnt >= m) ? R.start : R.source_index(cnt);
        atomicAdd(&gs[lds_slot<EPT>(e)], w);
        atomicAdd(&gt[lds_slot<EPT>(jt)], -w);
      }
    }
    if (e < m) {                                             // grid point = shifted target CDF level C_e
      float g, b;
      R.atom(e, g, b);
      const int rho = e >= R.start ? e - R.start : e - R.start + m;
      const int il = min(S.below(g, true), n - 1);
      const float a = S.v(il);
      const int na = S.below(g, false);                      // source levels <= g sort before g in the merged grid
      const float prev_a = na > 0 ? S.c(na - 1) : 0.f;
      const float prev_c = rho > 0 ? R.cdf_at(rho - 1) : 0.f;
      const float width = g - fmaxf(prev_a, prev_c);
      const float d = a - b;
      acc += width * powp<PMODE>(d, p, p_int);
      if constexpr (GRAD) {
        const float w = width * dpow_abs<PMODE>(d, p, p_int);
        atomicAdd(&gs[lds_slot<EPT>(il)], w);
        atomicAdd(&gt[lds_slot<EPT>(e)], -w);
      }
    }
  }
  return wave_sum_uniform(acc, lane);
}

// inclusive prefix sum over the wave's sorted positions lane*EPT + r  (the CDF, :169-170)
template <int EPT>
__device__ __forceinline__ void sorted_cdf(float (&w)[EPT], int lane) {
  float run = 0.f;
#pragma unroll
  for (int r = 0; r < EPT; ++r) { run += w[r]; w[r] = run; }
  float incl = run;                                          // inclusive scan of the lane totals
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const float up = as_f(__builtin_amdgcn_ds_bpermute(max(lane - d, 0) << 2, as_i(incl)));
    incl += (lane >= d) ? up : 0.f;
  }
  const float offset = incl - run;
#pragma unroll
  for (int r = 0; r < EPT; ++r) w[r] += offset;
}

// project, sort (with indices), gather weights and build the CDFs of both clouds of slice s; leaves the sorted
// values / CDFs in LDS and the sorted->original index maps in registers
template <int EPT>
__device__ __forceinline__ void prepare_sides(const GeneralArgs& G, int s, int lane, float* s_val, float* s_cdf,
                                              float* t_val, float* t_cdf, float* scratch, int (&sidx)[EPT],
                                              int (&tidx)[EPT]) {
  const SswArgs& A = G.base;
  const int b = s / A.slices, l = s - b * A.slices;
  const int n = A.n, m = A.m;
  const float* Ul = A.dirs + (long)b * A.u_pair_stride + (long)l * 6;
  float U[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) U[i] = Ul[i];
#pragma nounroll
  for (int which = 0; which < 2; ++which) {                  // 0: target, 1: source
    const float* X = which == 0 ? A.xt + (long)b * m * 3 : A.xs + (long)b * n * 3;
    const int count = which == 0 ? m : n;
    const float* W = which == 0 ? G.wv : G.wu;
    const long wstride = which == 0 ? G.wv_pair_stride : G.wu_pair_stride;
    float* dval = which == 0 ? t_val : s_val;
    float* dcdf = which == 0 ? t_cdf : s_cdf;
    int ln = lane;
    asm volatile("" : "+v"(ln));
    float val[EPT];
    int idx[EPT];
    sorted_with_indices<EPT>(X, count, ln, U, scratch, val, idx);
    float w[EPT];
#pragma unroll
    for (int r = 0; r < EPT; ++r) {
      const int e = lane * EPT + r;
      const bool live = e < count;
      w[r] = !live ? 0.f : (W ? W[(long)b * wstride + idx[r]] : 1.f / (float)count);
    }
    sorted_cdf<EPT>(w, lane);
#pragma unroll
    for (int r = 0; r < EPT; ++r) {                          // sorted position lane*EPT + r -> slot r*64 + lane
      dval[r * kWave + lane] = val[r];
      dcdf[r * kWave + lane] = w[r];
    }
    if (which == 0) {
#pragma unroll
      for (int r = 0; r < EPT; ++r) tidx[r] = idx[r];
    } else {
#pragma unroll
      for (int r = 0; r < EPT; ++r) sidx[r] = idx[r];
    }
    __builtin_amdgcn_wave_barrier();
  }
}

template <int EPT, int PMODE, bool GRAD>
__global__ __launch_bounds__(64) void ssw_general_kernel(GeneralArgs G) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int ROW = EPT * kWave;
  const SswArgs& A = G.base;
  const int lane = threadIdx.x & 63;
  float* s_val = lds;
  float* s_cdf = lds + ROW;
  float* t_val = lds + 2 * ROW;
  float* t_cdf = lds + 3 * ROW;
  float* scratch = lds + 4 * ROW;                            // coordinates by original index; later gs
  float* gt = lds + 5 * ROW;                                 // GRAD only

  const int s = xcd_contiguous_id(blockIdx.x, A.num_groups);
  if (s >= A.pairs * A.slices) return;
  const int n = A.n, m = A.m;
  int sidx[EPT], tidx[EPT];
  prepare_sides<EPT>(G, s, lane, s_val, s_cdf, t_val, t_cdf, scratch, sidx, tidx);

  Side<EPT> S{s_val, s_cdf, n}, T{t_val, t_cdf, m};

  // ---- bisection over the cut (reference :174-205) ----------------------------------------------
  float t_lo = -1.f, t_hi = 1.f, t_mid = 0.f;
  for (int it = 0; it < 40; ++it) {                          // widths halve: 2^-25 < 1e-7 after 25 steps
    float dp, dm;
    cut_slopes<EPT, PMODE>(S, T, t_mid, lane, A.p, A.p_int, dp, dm);
    if (dp * dm <= 0.f) break;                               // settled on a kink / flat piece
    if (!(dp * dm > 0.f)) break;                             // non-finite input: stop
    if ((t_hi - t_lo) < 1e-6f / 10.f) {                      // eps / L, :189
      float dp_lo, dm_lo, dp_hi, dm_hi;
      cut_slopes<EPT, PMODE>(S, T, t_lo, lane, A.p, A.p_int, dp_lo, dm_lo);
      cut_slopes<EPT, PMODE>(S, T, t_hi, lane, A.p, A.p_int, dp_hi, dm_hi);
      const float c_lo = cut_cost<EPT, PMODE, false>(S, T, t_lo, lane, A.p, A.p_int, nullptr, nullptr);
      const float c_hi = cut_cost<EPT, PMODE, false>(S, T, t_hi, lane, A.p, A.p_int, nullptr, nullptr);
      if (fabsf(dp_lo - dm_hi) > 1e-3f)                      // tangent intersection, :198-199
        t_mid = (c_hi - c_lo + t_lo * dp_lo - t_hi * dm_hi) / (dp_lo - dm_hi);
      break;
    }
    if (dp < 0.f) t_lo = t_mid; else t_hi = t_mid;
    t_mid = (t_lo + t_hi) * 0.5f;
  }

  if constexpr (GRAD) {
#pragma unroll
    for (int r = 0; r < EPT; ++r) {
      scratch[r * kWave + lane] = 0.f;
      gt[r * kWave + lane] = 0.f;
    }
    __builtin_amdgcn_wave_barrier();
  }
  const float cost = cut_cost<EPT, PMODE, GRAD>(S, T, t_mid, lane, A.p, A.p_int, scratch, gt);
  if (lane == 0) {
    A.slice_cost[s] = cost;
    if (G.slice_theta) G.slice_theta[s] = t_mid;
  }
  if constexpr (GRAD) {
    __builtin_amdgcn_wave_barrier();
    float* cs = A.coef_s + (long)s * n;
    float* ct = A.coef_t + (long)s * m;
#pragma unroll
    for (int r = 0; r < EPT; ++r) {
      const int e = lane * EPT + r;
      if (e < n) cs[sidx[r]] = scratch[r * kWave + lane];
      if (e < m) ct[tidx[r]] = gt[r * kWave + lane];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// p == 1 with weights: the reference's level-median formula (emd1D_circle, :210-247) on weighted atoms.
// level = CDF difference after the atom in merged-by-value order (source before target on equal values),
// gap = distance to the merged successor (the last atom: 1 - value; [0, first atom) is not integrated),
// median = smallest level whose cumulated gap weight reaches 0.5 (the smallest level if the total never does),
// cost = sum gap * |level - median|.  Levels are floats here, so the median is a float bisection followed by a
// snap to the smallest level above the bracket.  Coefficients (GRAD): |level_before - med| - |level - med|,
// the first merged atom -|level - med|.
// ---------------------------------------------------------------------------------------------
template <int EPT, bool GRAD>
__global__ __launch_bounds__(64) void ssw_general_p1_kernel(GeneralArgs G) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int ROW = EPT * kWave;
  const SswArgs& A = G.base;
  const int lane = threadIdx.x & 63;
  float* s_val = lds;
  float* s_cdf = lds + ROW;
  float* t_val = lds + 2 * ROW;
  float* t_cdf = lds + 3 * ROW;
  float* scratch = lds + 4 * ROW;
  float* lev_s = lds + 4 * ROW;                              // reuses scratch once the sorts are done
  float* gap_s = lds + 5 * ROW;
  float* lev_t = lds + 6 * ROW;
  float* gap_t = lds + 7 * ROW;

  const int s = xcd_contiguous_id(blockIdx.x, A.num_groups);
  if (s >= A.pairs * A.slices) return;
  const int n = A.n, m = A.m;
  int sidx[EPT], tidx[EPT];
  prepare_sides<EPT>(G, s, lane, s_val, s_cdf, t_val, t_cdf, scratch, sidx, tidx);
  Side<EPT> S{s_val, s_cdf, n}, T{t_val, t_cdf, m};

  float lo_lev = __builtin_inff(), hi_lev = -__builtin_inff(), total = 0.f;
#pragma nounroll
  for (int r = 0; r < EPT; ++r) {
    const int e = lane * EPT + r;
    if (e < n) {                                             // source atom e
      const float val = S.v(e);
      const int lb = T.values_below(val, true);
      const float lev = S.c(e) - (lb > 0 ? T.c(lb - 1) : 0.f);
      const float nxt = fminf(e + 1 < n ? S.v(e + 1) : __builtin_inff(), lb < m ? T.v(lb) : __builtin_inff());
      const float gap = (nxt == __builtin_inff() ? 1.f : nxt) - val;
      lev_s[r * kWave + lane] = lev;
      gap_s[r * kWave + lane] = gap;
      lo_lev = fminf(lo_lev, lev); hi_lev = fmaxf(hi_lev, lev); total += gap;
    }
    if (e < m) {                                             // target atom e
      const float val = T.v(e);
      const int ub = S.values_below(val, false);
      const float lev = (ub > 0 ? S.c(ub - 1) : 0.f) - T.c(e);
      const float nxt = fminf(e + 1 < m ? T.v(e + 1) : __builtin_inff(), ub < n ? S.v(ub) : __builtin_inff());
      const float gap = (nxt == __builtin_inff() ? 1.f : nxt) - val;
      lev_t[r * kWave + lane] = lev;
      gap_t[r * kWave + lane] = gap;
      lo_lev = fminf(lo_lev, lev); hi_lev = fmaxf(hi_lev, lev); total += gap;
    }
  }
  __builtin_amdgcn_wave_barrier();
  lo_lev = as_f(__builtin_amdgcn_readfirstlane(as_i(-wave_max(-lo_lev, lane))));
  hi_lev = as_f(__builtin_amdgcn_readfirstlane(as_i(wave_max(hi_lev, lane))));
  total = wave_sum_uniform(total, lane);

  auto weight_below = [&](float t) -> float {               // sum of gaps of atoms with level <= t
    float w = 0.f;
#pragma nounroll
    for (int r = 0; r < EPT; ++r) {
      const int e = lane * EPT + r;
      if (e < n && lev_s[r * kWave + lane] <= t) w += gap_s[r * kWave + lane];
      if (e < m && lev_t[r * kWave + lane] <= t) w += gap_t[r * kWave + lane];
    }
    return wave_sum_uniform(w, lane);
  };
  float med = lo_lev;
  if (total >= 0.5f) {
    float lo = lo_lev - 1.f, hi = hi_lev;                    // W(lo) = 0 < 0.5 <= W(hi) = total
    for (int it = 0; it < 48 && lo < hi; ++it) {
      const float mid = lo + (hi - lo) * 0.5f;
      if (!(mid > lo && mid < hi)) break;                    // bracket exhausted at fp32 resolution
      if (weight_below(mid) >= 0.5f) hi = mid; else lo = mid;
    }
    float best = __builtin_inff();                           // smallest level above the bracket's lower end
#pragma nounroll
    for (int r = 0; r < EPT; ++r) {
      const int e = lane * EPT + r;
      if (e < n) { const float l = lev_s[r * kWave + lane]; best = (l > lo) ? fminf(best, l) : best; }
      if (e < m) { const float l = lev_t[r * kWave + lane]; best = (l > lo) ? fminf(best, l) : best; }
    }
    med = as_f(__builtin_amdgcn_readfirstlane(as_i(-wave_max(-best, lane))));
  }

  float acc = 0.f;
  float* cs = GRAD ? A.coef_s + (long)s * n : nullptr;
  float* ct = GRAD ? A.coef_t + (long)s * m : nullptr;
#pragma unroll
  for (int r = 0; r < EPT; ++r) {
    const int e = lane * EPT + r;
    if (e < n) {
      const float lev = lev_s[r * kWave + lane], here = fabsf(lev - med);
      acc += gap_s[r * kWave + lane] * here;
      if constexpr (GRAD) {
        const float own = S.c(e) - (e > 0 ? S.c(e - 1) : 0.f);
        const bool first = (e == 0) && (T.values_below(S.v(0), true) == 0);
        cs[sidx[r]] = (first ? 0.f : fabsf(lev - own - med)) - here;
      }
    }
    if (e < m) {
      const float lev = lev_t[r * kWave + lane], here = fabsf(lev - med);
      acc += gap_t[r * kWave + lane] * here;
      if constexpr (GRAD) {
        const float own = T.c(e) - (e > 0 ? T.c(e - 1) : 0.f);
        const bool first = (e == 0) && (S.values_below(T.v(0), false) == 0);
        ct[tidx[r]] = (first ? 0.f : fabsf(lev + own - med)) - here;
      }
    }
  }
  const float cost = wave_sum_uniform(acc, lane);
  if (lane == 0) {
    A.slice_cost[s] = cost;
    if (G.slice_theta) G.slice_theta[s] = med;
  }
}

template <int EPT>
static int launch_general(GeneralArgs& G, hipStream_t stream) {
  SswArgs& A = G.base;
  const long total = (long)A.pairs * A.slices;
  if (total > 0x7fffffffL) return (int)hipErrorInvalidValue;
  A.num_groups = (int)total;
  const bool grad = A.coef_s != nullptr;
  const dim3 grid((unsigned)total), block(64);
  if (A.p == 1.f) {
    const size_t lds1 = (size_t)8 * EPT * kWave * sizeof(float);
    if (lds1 > 160 * 1024) return (int)hipErrorInvalidValue;
    if (grad) hipLaunchKernelGGL((ssw_general_p1_kernel<EPT, true>), grid, block, lds1, stream, G);
    else hipLaunchKernelGGL((ssw_general_p1_kernel<EPT, false>), grid, block, lds1, stream, G);
    return (int)hipGetLastError();
  }
  const size_t lds = (size_t)(grad ? 6 : 5) * EPT * kWave * sizeof(float);
  if (A.p_int == 2) {
    if (grad) hipLaunchKernelGGL((ssw_general_kernel<EPT, 2, true>), grid, block, lds, stream, G);
    else hipLaunchKernelGGL((ssw_general_kernel<EPT, 2, false>), grid, block, lds, stream, G);
  } else {
    if (grad) hipLaunchKernelGGL((ssw_general_kernel<EPT, 0, true>), grid, block, lds, stream, G);
    else hipLaunchKernelGGL((ssw_general_kernel<EPT, 0, false>), grid, block, lds, stream, G);
  }
  return (int)hipGetLastError();
}

int dispatch_general(SswArgs& A, const float* wu, const float* wv, long wu_pair_stride, long wv_pair_stride,
                     float* slice_theta, hipStream_t stream) {
  GeneralArgs G{A, wu, wv, wu_pair_stride, wv_pair_stride, slice_theta};
  switch (ept_for(A.n, A.m)) {
#ifdef SHW_DEV_ONLY_EPT
    case SHW_DEV_ONLY_EPT: return launch_general<SHW_DEV_ONLY_EPT>(G, stream);
#else
    case 1: return launch_general<1>(G, stream);
    case 2: return launch_general<2>(G, stream);
    case 4: return launch_general<4>(G, stream);
    case 8: return launch_general<8>(G, stream);
    case 16: return launch_general<16>(G, stream);
    case 32: return launch_general<32>(G, stream);
    case 64: return launch_general<64>(G, stream);
#endif
    default: return (int)hipErrorInvalidValue;               // > 4096 points: not built for this path
  }
}

}  // namespace shw
