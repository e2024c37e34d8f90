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
// arguments (:289) land here.
//
// Round 3: W wavefronts of one workgroup share a slice (W = 2 from 1024 points on, 4 at 4096).  The LDS arrays are the
// slice's, so W waves per slice put W times the waves on a CU without another byte of LDS (round 2 ran ONE wave per
// SIMD at n = 2048 with weights: 7.5 clocks per instruction of a dependent chain).  Wave 0 sorts the source while wave 1
// sorts the target; every evaluation of the solve is split by atoms (thread t of the slice owns sorted atoms
// [t AP, (t+1) AP), AP = EPT / W) and its partial sums are added in wave order through LDS, so all waves take the same
// decisions.  Gradients are OWNER-COMPUTED: every sorted atom's coefficient is accumulated in registers by one thread
// that walks the merged CDF grid over the atom's own mass interval, and written once -- no zero fill, no float
// atomics, bit-identical from run to run like the reference's autograd on the CPU.
#include "bin_sort_idx.hpp"
#include "ssw_common.hpp"

#ifndef SHW_DBG_EXTRA_LDS
#define SHW_DBG_EXTRA_LDS 0
#endif
#ifndef SHW_GENERAL_FIRST_GAIN
#define SHW_GENERAL_FIRST_GAIN 0.55f  // first step of the cut search at p = 2, in units of |slope|: Newton at curvature 2 is 0.5; stepping 10 % past it brackets the root at the second evaluation (measured 2.43 -> 2.35 ms; 0.6: 2.38, 0.7: 2.43)
#endif
#ifndef SHW_GENERAL_CHAINS
#define SHW_GENERAL_CHAINS 2    // interleaved rank walks per lane in the weighted slope evaluation
#endif

namespace shw {

struct GeneralArgs {
  SswArgs base;
  const float* wu;        // (n) or (pairs, n) source weights, NULL = uniform 1/n
  const float* wv;        // (m) or (pairs, m) target weights, NULL = uniform 1/m
  long wu_pair_stride;    // 0 = shared by all pairs
  long wv_pair_stride;
  float* slice_theta;     // optional: the cut the solve ended on
  float first_step;       // weights: first step of the bracket search around the mean-difference guess
  float min_width;        // weights: bracket width below which the tangent intersection finishes the solve
  int lcm, lcm_a, lcm_b;  // no weights: lcm(n, m), lcm / n, lcm / m  (n, m <= 4096: lcm < 2^24) -- the integer grid below
  // training runs as TWO launches: the solve at the loss-only kernel's occupancy (it leaves the cut of slice s in
  // cut_scratch[s * cut_stride] -- the first word of the slice's own coefficient row), then the gradient kernel
  // with cut_given = 1, which skips the solve and evaluates Cost and its gradient at that cut.
  float* cut_scratch;
  float* cut_scratch_t;   // index hand-off only: the target coefficient rows
  long cut_stride;
  int cut_given;
  // index hand-off (no weights, n, m >= 2): the solve launch also leaves the sort permutations of slice s in the
  // slice's coefficient rows (16-bit original indices by sorted position; the cut then goes to the LAST word of the
  // target row), and the gradient launch rebuilds the sorted coordinates from them -- a gather and a projection
  // instead of a second pair of sorts at the gradient kernel's low occupancy.
  int idx_handoff;
};

// The W waves of the workgroup that owns a slice.  sum(): wave-uniform partial sums -> sums over the slice, added in wave
// order (every wave gets the same bits, so control flow that depends on them stays uniform over the workgroup).  One
// barrier per call: the slots alternate between two parities, and a wave can only be one call ahead of another.
template <int W>
struct SliceTeam {
  float* red;               // [2 parities][W][4] floats
  int wave;
  int parity;
  template <int K>
  __device__ __forceinline__ void sum(float (&v)[K], int lane) {
    static_assert(K <= 4, "four sums per call");
    if constexpr (W > 1) {
      float* slot = red + parity * (4 * W);
      parity ^= 1;
      if (lane == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) slot[wave * 4 + k] = v[k];
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < K; ++k) {
        float acc = 0.f;
#pragma unroll
        for (int q = 0; q < W; ++q) acc += slot[q * 4 + k];
        v[k] = as_f(__builtin_amdgcn_readfirstlane(as_i(acc)));
      }
    }
  }
};
template <int W>
__device__ __forceinline__ void team_sum2_min2(SliceTeam<W>& team, float (&sums)[2], float (&mins)[2], int lane) {
  if constexpr (W > 1) {
    float* slot = team.red + team.parity * (4 * W);
    team.parity ^= 1;
    if (lane == 0) { slot[team.wave * 4] = sums[0]; slot[team.wave * 4 + 1] = sums[1]; slot[team.wave * 4 + 2] = mins[0]; slot[team.wave * 4 + 3] = mins[1]; }
    __syncthreads();
    float a0 = 0.f, a1 = 0.f, m0 = __builtin_inff(), m1 = __builtin_inff();
#pragma unroll
    for (int q = 0; q < W; ++q) {
      a0 += slot[q * 4]; a1 += slot[q * 4 + 1];
      m0 = fminf(m0, slot[q * 4 + 2]); m1 = fminf(m1, slot[q * 4 + 3]);
    }
    sums[0] = as_f(__builtin_amdgcn_readfirstlane(as_i(a0)));
    sums[1] = as_f(__builtin_amdgcn_readfirstlane(as_i(a1)));
    mins[0] = as_f(__builtin_amdgcn_readfirstlane(as_i(m0)));
    mins[1] = as_f(__builtin_amdgcn_readfirstlane(as_i(m1)));
  }
}
constexpr int kTeamFloats = 48;   // two parities of four waves' sums + the means and the tail coefficient

// waves per slice by size class: the evaluations split by atoms, the two sorts take one wave each
#ifndef SHW_GENERAL_W32
#define SHW_GENERAL_W32 2       // waves per slice at 1025..2048 points
#endif
#ifndef SHW_GENERAL_MINW_UNIFORM
#define SHW_GENERAL_MINW_UNIFORM 3   // waves per SIMD asked of the register allocator, kernels without weights
#endif
constexpr int general_waves(int ept) { return ept >= 64 ? 4 : (ept == 32 ? SHW_GENERAL_W32 : (ept >= 16 ? 2 : 1)); }
// one cloud as the solver sees it (weights given): ascending atom values and their inclusive CDF, lds_slot layout.
// (Clouds WITHOUT weights never get here: their CDFs are (i+1)/count and the solve runs on the integer grid of lcm(n, m),
//  grid_* below.)
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

// ---------------------------------------------------------------------------------------------
// Batched, branch-free binary searches.  The searches of one atom are a chain of dependent LDS reads (12 probes
// at 2048 atoms); a lane owns up to 64 atoms and the first version ran their searches one after the other with
// data-dependent loops: ~1 500 dependent LDS round trips per lane per evaluation, 25 evaluations per slice,
// 51 ms per loss at config-3 sizes.  Here NB atoms are searched TOGETHER with a fixed trip count, so that each
// level issues NB (or 2 NB) independent reads.
// lower_bounds2: for every key, the number of entries < key (lt) and <= key (le) among the first `count`
// entries of an ascending array in lds_slot layout  (= torch.searchsorted(..., right=False / True)).
// ---------------------------------------------------------------------------------------------
template <int EPT, int NB>
__device__ __forceinline__ void lower_bounds2_arr(const float* arr, int count, const float (&key)[NB], int (&lt)[NB],
                                                  int (&le)[NB]) {
  constexpr int P = EPT * kWave;
  // one fixed-trip search for #{< key}; #{<= key} then differs only by the entries EQUAL to key, which two more
  // probes count in all but degenerate inputs (three or more equal entries: a second full search, rare branch)
#pragma unroll
  for (int b = 0; b < NB; ++b) lt[b] = 0;
#pragma unroll
  for (int st = P / 2; st >= 1; st >>= 1) {
    float x[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) x[b] = arr[lds_slot<EPT>(lt[b] + st - 1)];
#pragma unroll
    for (int b = 0; b < NB; ++b) lt[b] += ((lt[b] + st - 1 < count) && (x[b] < key[b])) ? st : 0;
  }
  bool again = false;
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const float x = arr[lds_slot<EPT>(min(lt[b], P - 1))];
    lt[b] += ((lt[b] < count) && (x < key[b])) ? 1 : 0;
    const float e0 = arr[lds_slot<EPT>(min(lt[b], P - 1))];
    const float e1 = arr[lds_slot<EPT>(min(lt[b] + 1, P - 1))];
    const float e2 = arr[lds_slot<EPT>(min(lt[b] + 2, P - 1))];
    const bool q0 = (lt[b] < count) && (e0 == key[b]);
    const bool q1 = q0 && (lt[b] + 1 < count) && (e1 == key[b]);
    const bool q2 = q1 && (lt[b] + 2 < count) && (e2 == key[b]);
    le[b] = lt[b] + (q0 ? 1 : 0) + (q1 ? 1 : 0);
    again |= q2;
  }
  if (again) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      int pos = 0;
#pragma unroll
      for (int st = P / 2; st >= 1; st >>= 1) {
        const float y = arr[lds_slot<EPT>(pos + st - 1)];
        pos += ((pos + st - 1 < count) && (y <= key[b])) ? st : 0;
      }
      const float y = arr[lds_slot<EPT>(min(pos, P - 1))];
      le[b] = pos + (((pos < count) && (y <= key[b])) ? 1 : 0);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Walking searches (round 2, weighted clouds).  A lane's atoms are CONSECUTIVE sorted atoms, so their CDF levels
// ascend and so do their ranks in the other cloud's CDF: after one binary search for the lane's first atom the rank
// of every further atom is found by WALKING forward from its predecessor's -- four entries are read at once and the
// entries below the key counted; with weights of comparable size the walk advances ~1 entry per atom and one round
// of four reads settles it (the rare lane that needs more loops, wave-uniformly; a walk longer than kWalkRounds
// rounds falls back to the binary search).  The ranks are the binary search's, entry for entry: #{< key} is monotone
// in the key.  12 + 4 probes per atom become 4 + 3, and the chain of dependent reads per evaluation 32 instead
// of 4 x 15.  `prev` (the previous key) detects the one place where the keys of a lane do not ascend -- the rotated
// target's wrap from level ~1 to level ~0 -- and restarts the walk at entry 0.
// ---------------------------------------------------------------------------------------------
constexpr int kWalkRounds = 6;
// Window reads at constant offsets: an array in lds_slot layout ([r][lane], entry i at row i % EPT, column i / EPT)
// keeps entries i, i+1, ... of one column one row (256 bytes) apart -- until the column ends.  kWalkExt extra rows
// under the array repeat the first kWalkExt rows one column to the left (ext[r][c] = arr[r - EPT][c + 1], +inf past the
// last column), so that the kWalkExt entries from ANY index are base + q * 256 bytes: one address, kWalkExt reads.
constexpr int kWalkExt = 6;

// weighted clouds with >= 8 atoms per lane evaluate their slopes by walking (cut_slopes_walk)
template <int EPT>
constexpr bool general_walks() { return EPT >= 8; }
template <int EPT>
constexpr int general_ext_floats() { return general_walks<EPT>() ? kWalkExt * kWave : 0; }


template <int EPT>
__device__ __forceinline__ void fill_walk_ext(float* arr, int lane) {
#pragma unroll
  for (int q = 0; q < kWalkExt; ++q) {
    const float x = arr[q * kWave + min(lane + 1, kWave - 1)];
    arr[(EPT + q) * kWave + lane] = lane + 1 < kWave ? x : __builtin_inff();
  }
}

template <int EPT>
__device__ __forceinline__ int upper_bound_arr(const float* arr, int count, float key) {
  constexpr int P = EPT * kWave;
  int le = 0;
#pragma unroll
  for (int st = P / 2; st >= 1; st >>= 1) {
    const float x = arr[lds_slot<EPT>(le + st - 1)];
    le += ((le + st - 1 < count) && (x <= key)) ? st : 0;
  }
  const float x = arr[lds_slot<EPT>(min(le, P - 1))];
  return le + (((le < count) && (x <= key)) ? 1 : 0);
}

template <int EPT>
__device__ __forceinline__ int lower_bound_arr(const float* arr, int count, float key) {
  constexpr int P = EPT * kWave;
  int lt = 0;
#pragma unroll
  for (int st = P / 2; st >= 1; st >>= 1) {
    const float x = arr[lds_slot<EPT>(lt + st - 1)];
    lt += ((lt + st - 1 < count) && (x < key)) ? st : 0;
  }
  const float x = arr[lds_slot<EPT>(min(lt, P - 1))];
  return lt + (((lt < count) && (x < key)) ? 1 : 0);
}

// #{entries < key} for ONE key common to the wave: two rounds of 64 probes instead of 12 dependent ones
template <int EPT>
__device__ __forceinline__ int wave_lower_bound_arr(const float* arr, int count, float key, int lane) {
  static_assert(EPT <= kWave, "one probe per lane covers a block of EPT entries");
  const int i1 = lane * EPT + EPT - 1;                       // last entry of block `lane`
  const bool b1 = (i1 < count) && (arr[lds_slot<EPT>(i1)] < key);
  const int blk = __builtin_popcountll(__builtin_amdgcn_ballot_w64(b1));   // blocks entirely below the key
  const int i2 = min(blk, kWave - 1) * EPT + min(lane, EPT - 1);
  const bool b2 = (blk < kWave) && (lane < EPT) && (i2 < count) && (arr[lds_slot<EPT>(i2)] < key);
  return blk * EPT + __builtin_popcountll(__builtin_amdgcn_ballot_w64(b2));
}

// ranks #{< k} (ptr, updated) and #{<= k} (le) of C ascending key chains in `arr` (lds_slot layout with the
// fill_walk_ext rows, dead entries +inf), each from its chain's previous rank on: both are counted among the kWalkExt
// entries from ptr on and are settled unless all of those are <= k (then another round, wave-uniformly; binary
// searches after kWalkRounds rounds).
template <int EPT, int C>
__device__ __forceinline__ void walk_window(const float* arr, int count, const float (&k)[C], int (&ptr)[C],
                                            int (&le)[C]) {
  constexpr int P = EPT * kWave;
  int rounds = 0;
  for (;;) {
    bool more = false;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const float* w = arr + lds_slot<EPT>(min(ptr[c], P - 1));
      int lta = 0, lea = 0;
#pragma unroll
      for (int q = 0; q < kWalkExt; ++q) {
        const float x = w[q * kWave];
        lta += (x < k[c]) ? 1 : 0;
        lea += (x <= k[c]) ? 1 : 0;
      }
      const bool inside = ptr[c] < P;                        // ptr == P (every entry below the key): nothing to read
      lta = inside ? lta : 0;
      lea = inside ? lea : 0;
      le[c] = ptr[c] + lea;
      ptr[c] += lta;
      more |= lea == kWalkExt;
    }
    if (__builtin_amdgcn_ballot_w64(more) == 0) break;
    if (++rounds >= kWalkRounds) {
#pragma unroll
      for (int c = 0; c < C; ++c) {
        ptr[c] = lower_bound_arr<EPT>(arr, count, k[c]);
        le[c] = upper_bound_arr<EPT>(arr, count, k[c]);
      }
      break;
    }
  }
}

// ranks of NA keys that ascend (except where key < prev: restart).  ptr: in, a rank not above key[0]'s unless the
// keys restart; out, the rank of the last key.  Dead keys (live[a] false) are not searched: they take the running rank.
template <int EPT, int NA>
__device__ __forceinline__ void walk_lower_bounds2(const float* arr, int count, const float (&key)[NA],
                                                   const bool (&live)[NA], float& prev, int& ptr, int (&lt)[NA],
                                                   int (&le)[NA]) {
  constexpr int P = EPT * kWave;
  bool again = false;
#pragma unroll
  for (int a = 0; a < NA; ++a) {
    const float k = live[a] ? key[a] : prev;
    ptr = k < prev ? 0 : ptr;
    prev = k;
    int rounds = 0;
    for (;;) {
      float x[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) x[q] = arr[lds_slot<EPT>(min(ptr + q, P - 1))];
      int adv = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) adv += ((ptr + q < count) && (x[q] < k)) ? 1 : 0;
      ptr += adv;
      if (__builtin_amdgcn_ballot_w64(adv == 4) == 0) break;
      if (++rounds >= kWalkRounds) { ptr = lower_bound_arr<EPT>(arr, count, k); break; }
    }
    lt[a] = ptr;
    const float e0 = arr[lds_slot<EPT>(min(ptr, P - 1))];
    const float e1 = arr[lds_slot<EPT>(min(ptr + 1, P - 1))];
    const float e2 = arr[lds_slot<EPT>(min(ptr + 2, P - 1))];
    const bool q0 = (ptr < count) && (e0 == k);
    const bool q1 = q0 && (ptr + 1 < count) && (e1 == k);
    const bool q2 = q1 && (ptr + 2 < count) && (e2 == k);
    le[a] = ptr + (q0 ? 1 : 0) + (q1 ? 1 : 0);
    again |= q2;
  }
  if (again) {                                               // three or more equal entries: degenerate weights
#pragma unroll
    for (int a = 0; a < NA; ++a) {
      const float k = live[a] ? key[a] : prev;
      int pos = 0;
#pragma unroll
      for (int st = P / 2; st >= 1; st >>= 1) {
        const float y = arr[lds_slot<EPT>(pos + st - 1)];
        pos += ((pos + st - 1 < count) && (y <= k)) ? st : 0;
      }
      const float y = arr[lds_slot<EPT>(min(pos, P - 1))];
      le[a] = live[a] ? pos + (((pos < count) && (y <= k)) ? 1 : 0) : le[a];
    }
  }
}

// the target after moving mass theta around the circle (reference :31-48, evaluated lazily)
template <int EPT>
struct Rotated {
  Side<EPT> t;
  float turns, frac;
  int start;                               // number of wrapped atoms = first atom of the rotated order
  __device__ __forceinline__ void set(const Side<EPT>& target, float theta, int lane) {
    t = target;
    turns = floorf(theta);
    frac = theta - turns;
    // (cdf - frac) < 0  <=>  cdf < frac
    start = wave_lower_bound_arr<EPT>(target.cdf, target.count, frac, lane);
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
  // the same for NB ASCENDING keys by a forward walk over the rotated entries (see walk_lower_bounds2; no restart:
  // the keys are source levels).  cnt_io: in, a count not above key[0]'s; out, the count of the last key.
  template <int NB>
  __device__ __forceinline__ void below_walk(const float (&key)[NB], int& cnt_io, int (&cnt)[NB]) const {
    const int m = t.count;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      int rounds = 0;
      for (;;) {
        float x[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) x[q] = cdf_at(min(cnt_io + q, m - 1));
        int adv = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) adv += ((cnt_io + q < m) && (x[q] < key[b])) ? 1 : 0;
        cnt_io += adv;
        if (__builtin_amdgcn_ballot_w64(adv == 4) == 0) break;
        if (++rounds >= kWalkRounds) {
          const float k1[1] = {key[b]};
          int c1[1];
          below_batch<1>(k1, c1);
          cnt_io = c1[0];
          break;
        }
      }
      cnt[b] = cnt_io;
    }
  }
  // the same for NB keys at once, fixed trip count (see lower_bounds2)
  template <int NB>
  __device__ __forceinline__ void below_batch(const float (&key)[NB], int (&cnt)[NB]) const {
    constexpr int P = EPT * kWave;
    const int m = t.count;
#pragma unroll
    for (int b = 0; b < NB; ++b) cnt[b] = 0;
#pragma unroll
    for (int st = P / 2; st >= 1; st >>= 1) {
      float x[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) x[b] = cdf_at(min(cnt[b] + st - 1, m - 1));
#pragma unroll
      for (int b = 0; b < NB; ++b) cnt[b] += ((cnt[b] + st - 1 < m) && (x[b] < key[b])) ? st : 0;
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const float x = cdf_at(min(cnt[b], m - 1));
      cnt[b] += ((cnt[b] < m) && (x < key[b])) ? 1 : 0;
    }
  }
};

#ifdef SHW_DEV_NO_ATOMICS   // developer timing experiment only (wrong gradients): plain stores instead of LDS atomics
#define SHW_LDS_ADD(ptr, v) (*(ptr) = (v))
#else
#define SHW_LDS_ADD(ptr, v) atomicAdd((ptr), (v))
#endif

template <int PMODE>
__device__ __forceinline__ float powp(float d, float p, int p_int) { return pow_abs<PMODE>(d, p, p_int); }

// one-sided derivatives of the cost w.r.t. theta (reference dCost, :50-63), uniform over the slice's waves.
// tid: index of the thread among the 64 W threads of the slice; it owns target atoms [tid AP, (tid+1) AP), AP = EPT / W.
template <int EPT, int PMODE, int W>
__device__ void cut_slopes(const Side<EPT>& S, const Side<EPT>& T, float theta, int lane, int tid, float p,
                           int p_int, SliceTeam<W>& team, float& d_plus, float& d_minus) {
  constexpr int AP = EPT / W;
  Rotated<EPT> R;
  R.set(T, theta, lane);
  const int n = S.count, m = T.count;
  float sp = 0.f, sm = 0.f;
  constexpr int NA = AP < 8 ? AP : 8;                        // atoms searched together
  int walk_ptr = 0;                                          // rank of the thread's previous atom
  float walk_prev = 0.f;
  {
    float c0, p0;
    R.atom(min(tid * AP, m - 1), c0, p0);
    walk_ptr = lower_bound_arr<EPT>(S.cdf, n, c0);
    walk_prev = c0;
  }
#pragma nounroll
  for (int r0 = 0; r0 < AP; r0 += NA) {
    // NA + 1 consecutive atoms: atom a and its successor a + 1 (the atom after the last one is atom 0; indices
    // past the end repeat the last atom and are masked below)
    float wc[NA + 1], wp[NA + 1];
    int wj[NA + 1];
#pragma unroll
    for (int a = 0; a <= NA; ++a) {
      const int q = tid * AP + r0 + a;
      wj[a] = q < m ? q : (q == m ? 0 : m - 1);
      R.atom(wj[a], wc[a], wp[a]);
    }
    float cdf[NA], pos[NA], npos[NA];
#pragma unroll
    for (int a = 0; a < NA; ++a) {
      cdf[a] = wc[a];
      pos[a] = wp[a];
      npos[a] = wp[a + 1] + ((wj[a + 1] == R.start) ? 1.f : 0.f);   // successor of the last rotated atom: first + 1
    }
    int lt[NA], le[NA];
    {
      bool alive[NA];
#pragma unroll
      for (int a = 0; a < NA; ++a) alive[a] = (tid * AP + r0 + a) < m;
      walk_lower_bounds2<EPT, NA>(S.cdf, n, cdf, alive, walk_prev, walk_ptr, lt, le);
    }
    const float v0 = S.v(0);
#pragma unroll
    for (int a = 0; a < NA; ++a) {
      const float al = S.v(min(lt[a], n - 1));               // left-continuous source quantile (:50-51)
      // right-continuous on the extended arrays (:54-57): past the last source level the quantile is the first
      // atom one turn later (the second extension, level c0 + 1, cannot be reached: cdf <= 1); unconditional read
      const float sv = S.v(min(le[a], n - 1));
      const float ar = le[a] < n ? sv : v0 + 1.f;
      const bool live = (tid * AP + r0 + a) < m;
      const float tp = powp<PMODE>(al - npos[a], p, p_int) - powp<PMODE>(al - pos[a], p, p_int);
      const float tm = powp<PMODE>(ar - npos[a], p, p_int) - powp<PMODE>(ar - pos[a], p, p_int);
      sp += live ? tp : 0.f;
      sm += live ? tm : 0.f;
    }
  }
  float sums[2] = {wave_sum_uniform(sp, lane), wave_sum_uniform(sm, lane)};
  team.sum(sums, lane);
  d_plus = sums[0];
  d_minus = sums[1];
}

// cut_slopes for weighted clouds as C interleaved walks (see walk_lower_bounds2): chain c covers atoms
// [c * AP/C, (c+1) * AP/C) of the thread's AP atoms, the C chains advance together -- 4 C independent reads per round, EPT/C
// rounds per evaluation -- and each chain's first rank is carried from one evaluation of the solve to the next
// (`anchor`; warm = false: binary search): the cut moves by less than a level spacing between late evaluations, so
// the carried rank is put right by one backward and one forward round instead of a 12-probe search.
template <int EPT, int PMODE, int C, int W>
__device__ void cut_slopes_walk(const Side<EPT>& S, const Side<EPT>& T, float theta, int lane, int tid, float p,
                                int p_int, SliceTeam<W>& team, float& d_plus, float& d_minus, int (&anchor)[C], bool warm,
                                float& cost_scale) {
  constexpr int AP = EPT / W;                                // atoms of the thread: [tid AP, (tid+1) AP)
  constexpr int LEN = AP / C;
  static_assert(AP % C == 0, "chains of equal length");
  Rotated<EPT> R;
  R.set(T, theta, lane);
  const int n = S.count, m = T.count;
  const float* arr = S.cdf;
  auto atom_q = [&](int q, float& c, float& ps, int& j) {    // atom q of the lane's run; q == m: atom 0, past it: the last
    j = q < m ? q : (q == m ? 0 : m - 1);
    R.atom(j, c, ps);
  };
  int ptr[C];
  float prev[C], own_c[C], own_p[C], mass[C];
  // ---- first ranks
#pragma unroll
  for (int c = 0; c < C; ++c) {
    int j;
    atom_q(tid * AP + c * LEN, own_c[c], own_p[c], j);
    prev[c] = own_c[c];
    ptr[c] = min(max(anchor[c], 0), n);
    float bc, bp;                                            // mass of the chain's first atom: level step from its predecessor
    R.atom(j > 0 ? j - 1 : m - 1, bc, bp);
    mass[c] = own_c[c] - bc;
    mass[c] += mass[c] < 0.f ? 1.f : 0.f;
  }
  if (warm) {                                                // backwards until the entry before ptr is below the key
    int rounds = 0;
    for (;;) {
      bool more = false;
#pragma unroll
      for (int c = 0; c < C; ++c) {
        int back = 0;
        bool run = true;
#pragma unroll
        for (int q = 1; q <= 4; ++q) {
          const float x = arr[lds_slot<EPT>(max(ptr[c] - q, 0))];
          run = run && (ptr[c] - q >= 0) && !(x < prev[c]);
          back += run ? 1 : 0;
        }
        ptr[c] -= back;
        more |= back == 4;
      }
      if (__builtin_amdgcn_ballot_w64(more) == 0) break;
      if (++rounds >= kWalkRounds) { warm = false; break; }
    }
  }
  if (!warm) {
#pragma unroll
    for (int c = 0; c < C; ++c) ptr[c] = lower_bound_arr<EPT>(arr, n, prev[c]);
  }
  float sp = 0.f, sm = 0.f, sc = 0.f;
  const float v0 = S.v(0);
#pragma nounroll
  for (int i = 0; i < LEN; ++i) {
    float k[C], pos[C], npos[C], w[C];
    bool live[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int q = tid * AP + c * LEN + i;
      live[c] = q < m;
      pos[c] = own_p[c];
      k[c] = live[c] ? own_c[c] : prev[c];
      w[c] = mass[c];
      const float before = own_c[c];
      int nj;
      atom_q(q + 1, own_c[c], own_p[c], nj);                 // the successor: the chain's own atom of the next round
      mass[c] = own_c[c] - before;                           // levels are rotated by a common shift: steps survive, mod 1
      mass[c] += mass[c] < 0.f ? 1.f : 0.f;
      npos[c] = own_p[c] + ((nj == R.start) ? 1.f : 0.f);    // successor of the last rotated atom: first + 1
      ptr[c] = k[c] < prev[c] ? 0 : ptr[c];                  // the wrap: levels restart at ~0
      prev[c] = k[c];
    }
    int le[C];
    walk_window<EPT, C>(arr, n, k, ptr, le);
    if (i == 0) {
#pragma unroll
      for (int c = 0; c < C; ++c) anchor[c] = ptr[c];
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const float al = S.v(min(ptr[c], n - 1));              // left-continuous source quantile (:50-51)
      const float sv = S.v(min(le[c], n - 1));               // right-continuous on the extended arrays (:54-57)
      const float ar = le[c] < n ? sv : v0 + 1.f;
      const float tp = powp<PMODE>(al - npos[c], p, p_int) - powp<PMODE>(al - pos[c], p, p_int);
      const float tm = powp<PMODE>(ar - npos[c], p, p_int) - powp<PMODE>(ar - pos[c], p, p_int);
      sp += live[c] ? tp : 0.f;
      sm += live[c] ? tm : 0.f;
      sc += live[c] ? w[c] * powp<PMODE>(al - pos[c], p, p_int) : 0.f;
    }
  }
  // sc: the cost with every target atom sent whole to the source quantile at its level: the size of the cost, for the
  // solve's exit test
  float sums[3] = {wave_sum_uniform(sp, lane), wave_sum_uniform(sm, lane), wave_sum_uniform(sc, lane)};
  team.sum(sums, lane);
  d_plus = sums[0];
  d_minus = sums[1];
  cost_scale = sums[2];
}

// transport cost at a fixed cut (reference Cost, :94-112), uniform over the slice's waves.  Thread tid evaluates the grid
// points of source atoms and of target atoms [tid AP, (tid+1) AP).
template <int EPT, int PMODE, int W>
__device__ float cut_cost(const Side<EPT>& S, const Side<EPT>& T, float theta, int lane, int tid, float p,
                          int p_int, SliceTeam<W>& team) {
  constexpr int AP = EPT / W;
  Rotated<EPT> R;
  R.set(T, theta, lane);
  const int n = S.count, m = T.count;
  float acc = 0.f;
  constexpr int NA = AP < 8 ? AP : 8;                        // atoms searched together
  int walk_cnt = 0, walk_ptr = 0;                            // ranks of the thread's previous atoms
  float walk_prev = 0.f;
  {
    const float k1[1] = {S.c(min(tid * AP, n - 1))};
    int c1[1];
    R.template below_batch<1>(k1, c1);
    walk_cnt = c1[0];
    float c0, p0;
    R.atom(min(tid * AP, m - 1), c0, p0);
    walk_ptr = lower_bound_arr<EPT>(S.cdf, n, c0);
    walk_prev = c0;
  }
#pragma nounroll
  for (int r0 = 0; r0 < AP; r0 += NA) {
    {  // grid points = source CDF levels A_e
      float g[NA];
      int cnt[NA];
#pragma unroll
      for (int a = 0; a < NA; ++a) g[a] = S.c(min(tid * AP + r0 + a, n - 1));
      R.template below_walk<NA>(g, walk_cnt, cnt);             // rotated target atom active at g
#pragma unroll
      for (int a = 0; a < NA; ++a) {
        const int e = tid * AP + r0 + a;
        const bool live = e < n;
        const int ec = min(e, n - 1);
        const float b = R.pos_at(min(cnt[a], m));
        const float prev_a = ec > 0 ? S.c(ec - 1) : 0.f;
        const float prev_c = cnt[a] > 0 ? R.cdf_at(cnt[a] - 1) : 0.f;
        const float width = g[a] - fmaxf(prev_a, prev_c);
        const float d = S.v(ec) - b;
        acc += live ? width * powp<PMODE>(d, p, p_int) : 0.f;
      }
    }
    {  // grid points = shifted target CDF levels C_e
      float g[NA], b[NA];
      int lt[NA], le[NA];
#pragma unroll
      for (int a = 0; a < NA; ++a) R.atom(min(tid * AP + r0 + a, m - 1), g[a], b[a]);
      if constexpr (general_walks<EPT>()) {                  // window reads (the rows under the source CDF exist)
#pragma unroll
        for (int a = 0; a < NA; ++a) {
          const float k1[1] = {g[a]};
          int p1[1] = {g[a] < walk_prev ? 0 : walk_ptr}, l1[1];    // (the wrap: levels restart at ~0)
          walk_prev = g[a];
          walk_window<EPT, 1>(S.cdf, n, k1, p1, l1);
          walk_ptr = p1[0];
          lt[a] = p1[0];
          le[a] = l1[0];
        }
      } else {
        bool alive[NA];
#pragma unroll
        for (int a = 0; a < NA; ++a) alive[a] = true;        // (indices past the end repeat the last atom: keys ascend)
        walk_lower_bounds2<EPT, NA>(S.cdf, n, g, alive, walk_prev, walk_ptr, lt, le);
      }
#pragma unroll
      for (int a = 0; a < NA; ++a) {
        const int e = tid * AP + r0 + a;
        const bool live = e < m;
        const int ec = min(e, m - 1);
        const int rho = ec >= R.start ? ec - R.start : ec - R.start + m;
        const int il = min(lt[a], n - 1);
        const float av = S.v(il);
        const int na = le[a];                                // source levels <= g sort before g in the merged grid
        const float prev_a = na > 0 ? S.c(na - 1) : 0.f;
        const float prev_c = rho > 0 ? R.cdf_at(rho - 1) : 0.f;
        const float width = g[a] - fmaxf(prev_a, prev_c);
        const float d = av - b[a];
        acc += live ? width * powp<PMODE>(d, p, p_int) : 0.f;
      }
    }
  }
  float sums[1] = {wave_sum_uniform(acc, lane)};
  team.sum(sums, lane);
  return sums[0];
}

// ---------------------------------------------------------------------------------------------
// Gradient of Cost at the (detached) cut, OWNER-COMPUTED (round 3).  The merged CDF grid of Cost (:95-105) cuts [0, 1]
// into segments; on each one source atom i and one rotated target atom rho are active, and the segment adds
// width * |u_i - v_rho|^p to the cost, width * d|D|^p/dD to the coefficient of atom i and its negative to atom rho's.
// Round 2 accumulated both with LDS float atomics (sum order, hence the last bits, varied between runs).  Here every
// atom has ONE owner that walks the segments of the atom's own mass interval -- a two-pointer merge of its interval
// with the other cloud's levels, one segment per step -- accumulates in a register and writes the coefficient once:
//   walk_source_atoms : thread tid owns sorted source atoms [tid AP, (tid+1) AP); atom e's interval is
//                       (A_{e-1}, A_e], crossed by the rotated target levels C_rho inside it.  Also returns the
//                       thread's share of the cost (every segment belongs to exactly one source atom; the segments
//                       above the last source level -- rounding -- go to the last atom like the reference's clip).
//   walk_target_atoms : thread tid owns ROTATED target atoms [tid AP, (tid+1) AP) (the rotated order is the order of
//                       their levels); the owner of the last one also walks the tail (C_{m-1}, 1], where the active
//                       target atom is the appended copy of the first rotated atom one turn later (:48, :103): its
//                       coefficient belongs to that first atom and is handed over in `tail` (added in a fixed order).
// A step costs ~20 VALU + 3 LDS reads; a thread takes ~2 AP steps (its atoms + the foreign levels in its range), and
// threads are balanced because equal counts of atoms hold nearly equal mass.  Ties (a source level equal to a target
// level) advance the source first; the leftover segment has width 0 -- the reference's merged grid gives the duplicate
// grid point a zero delta too.
// ---------------------------------------------------------------------------------------------
template <int EPT, int PMODE, int W>
__device__ float walk_source_atoms(const Side<EPT>& S, const Rotated<EPT>& R, int tid, float p, int p_int,
                                   float* gs) {
  constexpr int AP = EPT / W;
  const int n = S.count, m = R.t.count;
  const float inf = __builtin_inff();
  int e = tid * AP;
  const int e_end = min(e + AP, n);
  bool active = e < e_end;
  const int e0 = min(e, n - 1);
  float a_prev = e0 > 0 ? S.c(e0 - 1) : 0.f;
  int rho = active ? (e0 > 0 ? R.below(a_prev) : 0) : m;
  float c_prev = rho > 0 ? R.cdf_at(min(rho, m) - 1) : 0.f;
  float a = S.c(e0), u = S.v(e0);
  float c = rho < m ? R.cdf_at(rho) : inf;
  float pos = R.pos_at(min(rho, m));
  float acc = 0.f, cost = 0.f;
  bool extended = false;                                     // the last source atom also takes the levels above A_{n-1}
  for (int guard = 0; guard < 2 * kWave * EPT + 8; ++guard) {
    if (__builtin_amdgcn_ballot_w64(active) == 0) break;
    if (active) {
      const float end = fminf(a, c);                         // (+inf: no level left on either side -- nothing to add)
      const float width = end < inf ? fmaxf(end - fmaxf(a_prev, c_prev), 0.f) : 0.f;
      const float d = u - pos;
      acc = fmaf(width, dpow_abs<PMODE>(d, p, p_int), acc);
      cost = fmaf(width, powp<PMODE>(d, p, p_int), cost);
      if (c < a) {                                           // the segment ended on a target level: next target atom
        c_prev = c;
        ++rho;
        c = rho < m ? R.cdf_at(rho) : inf;
        pos = R.pos_at(min(rho, m));
      } else if (e == n - 1 && !extended) {                  // (u_index.clip(0, n-1), :101)
        extended = true;
        a_prev = a;
        a = inf;
      } else {                                               // the atom's interval is done: its coefficient, once
        gs[lds_slot<EPT>(e)] = acc;
        acc = 0.f;
        a_prev = a;
        ++e;
        active = e < e_end;
        const int ec = min(e, n - 1);
        a = S.c(ec);
        u = S.v(ec);
      }
    }
  }
  return cost;
}

template <int EPT, int PMODE, int W>
__device__ void walk_target_atoms(const Side<EPT>& S, const Rotated<EPT>& R, int tid, float p, int p_int,
                                  float* gt, float* tail) {
  constexpr int AP = EPT / W;
  const int n = S.count, m = R.t.count;
  const float inf = __builtin_inff();
  int rho = tid * AP;
  const int rho_end = min(rho + AP, m);
  const bool owns_tail = (rho < m) && (rho_end == m);        // owner of the last rotated atom
  bool active = rho < rho_end;
  const int r0 = min(rho, m - 1);
  float c_prev = r0 > 0 ? R.cdf_at(r0 - 1) : 0.f;
  int i = active ? (r0 > 0 ? S.below(c_prev, true) : 0) : n;
  float a_prev = i > 0 ? S.c(min(i, n) - 1) : 0.f;
  float a = i < n ? S.c(i) : inf, u = S.v(min(i, n - 1));
  float c = R.cdf_at(r0), pos = R.pos_at(r0);
  float acc = 0.f;
  for (int guard = 0; guard < 2 * kWave * EPT + 8; ++guard) {
    if (__builtin_amdgcn_ballot_w64(active) == 0) break;
    if (active) {
      const float end = fminf(a, c);                         // (+inf: the tail beyond the last source level is empty)
      const float width = end < inf ? fmaxf(end - fmaxf(a_prev, c_prev), 0.f) : 0.f;
      acc = fmaf(width, dpow_abs<PMODE>(u - pos, p, p_int), acc);
      if (i < n && a <= c) {                                 // the segment ended on a source level: next source atom
        a_prev = a;
        ++i;
        a = i < n ? S.c(i) : inf;
        u = S.v(min(i, n - 1));
      } else {                                               // the atom's interval is done
        if (rho < m) gt[lds_slot<EPT>(R.source_index(rho))] = -acc;
        else *tail = -acc;                                   // the appended copy: belongs to the first rotated atom
        acc = 0.f;
        c_prev = c;
        ++rho;
        active = rho < rho_end || (owns_tail && rho == m);
        c = rho < m ? R.cdf_at(rho) : inf;
        pos = R.pos_at(min(rho, m));
      }
    }
  }
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

// ---------------------------------------------------------------------------------------------
// No weights, n != m (round 3): the solve on the INTEGER grid of lcm(n, m).
//
// With masses 1/n and 1/m every CDF level is a multiple of 1/G, G = lcm(n, m) = a n = b m.  Cut the unit of mass into G
// cells: cell q belongs to source atom q / a, and -- after moving mass k / G around the circle -- to the extended target
// atom (q + k) / b (floor division; vx(t) = v[t mod m] + floor(t / m)).  The reference's Cost (:68-113) at theta = k / G is
//     c(k) = (1/G) sum_q | u[q / a] - vx((q + k) / b) |^p ,
// it is linear between grid points (both quantile functions are step functions whose steps sit on the grid), so the
// bisection of binary_search_circle (:117-207) converges to  min_k c(k),  a convex sequence -- the same statement as row
// A8 of SURVEY 8a, which is its special case a = b = 1.  Everything here is exact integer index arithmetic: no CDF
// arrays, no searches, no rounding questions about coinciding levels (round 2 evaluated the same thing with closed-form
// float ranks: ~100 VALU per atom and evaluation, a divergent tie path for every third atom when n and m share a factor).
//   * slope:  c(k+1) - c(k) = (1/G) sum over the m cells q = t b - k - 1 whose target atom changes, of
//             |u[q/a] - vx(t)|^p - |u[q/a] - vx(t-1)|^p   (the reference's dCost, :59-63).  One pass gives the forward
//             difference dp at k, the backward difference dm (= dp at k - 1) and how far k can move either way before
//             any term's source atom changes (the distance to the next kink of the sequence).
//   * search: secant / Illinois on the slope from k0 = round(G (mean u - mean v)) (exact for p = 2 and evenly spread
//             targets), every evaluation moving the bracket at least to the next kink; ends when dm <= 0 <= dp.
//   * cost and gradient at k*: every source atom walks the <= a/b + 2 target atoms that share cells with it (and every
//             target atom its sources): each coefficient is accumulated by its owner and written once.
// ---------------------------------------------------------------------------------------------
// q = floor(x / d), r = x - q d for 0 <= x < 2^24, 1 <= d, quotient < 2^13 (indices of atoms): the fp32 quotient is
// within one of the answer (x is exact in fp32, the quotient's error is < 2^13 * 2^-22), one correction each way
__device__ __forceinline__ void div_small(int x, int d, float inv_d, int& q, int& r) {
  q = (int)((float)x * inv_d);
  r = x - q * d;
  if (r < 0) { r += d; --q; }
  if (r >= d) { r -= d; ++q; }
}

// k = q d + r with 0 <= r < d for |k| < 2^24 (floor division)
__device__ __forceinline__ void floor_divmod(int k, int d, float inv_d, int& q, int& r) {
  int qa, ra;
  div_small(k < 0 ? -k : k, d, inv_d, qa, ra);
  q = k < 0 ? -qa - (ra > 0 ? 1 : 0) : qa;
  r = (k < 0 && ra > 0) ? d - ra : ra;
}

__device__ __forceinline__ int floor_div_m(int x, int m) {          // x in [-m, 2m)
  return x < 0 ? -1 : (x >= m ? 1 : 0);
}

struct Grid {
  int n, m, G, a, b;
  float inv_a, inv_b;
};


// forward / backward differences of G c(k) at k and the distances to the neighbouring kinks; uniform over the slice
template <int EPT, int PMODE, int W>
__device__ void grid_slopes(const float* s_val, const float* t_val, const Grid& gr, int k, int lane, int tid, float p, int p_int,
                            SliceTeam<W>& team, float& dm, float& dp, int& gap_left, int& gap_right) {
  constexpr int AP = EPT / W;
  const int a = gr.a, b = gr.b, n = gr.n, m = gr.m;
  int kb, krem;                                                    // k = kb b + krem, 0 <= krem < b
  floor_divmod(k, b, gr.inv_b, kb, krem);
  const int s1 = krem == 0 ? 1 : 0;                                // dp's cells sit one target atom further when b | k
  const int t_base = krem == 0 ? kb : kb + 1;                      // ceil(k / b)
  const int j0 = tid * AP;
  // atom j: t = t_base + j;  dm's cell q1 = t b - k,  dp's cell q2 = (t + s1) b - k - 1;  0 <= q < G for j < m
  const int q1 = min((t_base + j0) * b - k, gr.G - 1);             // (threads past the last atom: clamped, masked below)
  int i1, r1, i2, r2, bh, bl;
  div_small(q1, a, gr.inv_a, i1, r1);
  div_small(max(q1 + s1 * b - 1, 0), a, gr.inv_a, i2, r2);
  div_small(b, a, gr.inv_a, bh, bl);
  float vm = target_unrolled<EPT>(t_val, min(t_base + j0 - 1, 3 * m - 1), m);
  float v0 = target_unrolled<EPT>(t_val, min(t_base + j0, 3 * m - 1), m);
  float sm = 0.f, sp = 0.f;
  int gl = 0x7fffffff, grt = 0x7fffffff;
#pragma unroll 4
  for (int r = 0; r < AP; ++r) {
    const bool live = (j0 + r) < m;
    const float vp = target_unrolled<EPT>(t_val, min(t_base + j0 + r + 1, 3 * m - 1), m);
    const float um = s_val[lds_slot<EPT>(min(i1, n - 1))];
    const float up = s_val[lds_slot<EPT>(min(i2, n - 1))];
    const float hi = s1 ? vp : v0, lo = s1 ? v0 : vm;
    const float tm = powp<PMODE>(um - v0, p, p_int) - powp<PMODE>(um - vm, p, p_int);
    const float tp = powp<PMODE>(up - hi, p, p_int) - powp<PMODE>(up - lo, p, p_int);
    sm += live ? tm : 0.f;
    sp += live ? tp : 0.f;
    gl = live ? min(gl, a - r1) : gl;
    grt = live ? min(grt, r2 + 1) : grt;
    vm = v0; v0 = vp;
    r1 += bl; i1 += bh;
    if (r1 >= a) { r1 -= a; ++i1; }
    r2 += bl; i2 += bh;
    if (r2 >= a) { r2 -= a; ++i2; }
  }
  float sums[2] = {wave_sum_uniform(sm, lane), wave_sum_uniform(sp, lane)};
  // (distances are <= max(a, b) <= 4096: exact in fp32)
  float mins[2] = {-wave_max(-(float)min(gl, 1 << 23), lane), -wave_max(-(float)min(grt, 1 << 23), lane)};
  mins[0] = as_f(__builtin_amdgcn_readfirstlane(as_i(mins[0])));
  mins[1] = as_f(__builtin_amdgcn_readfirstlane(as_i(mins[1])));
  team_sum2_min2(team, sums, mins, lane);
  dm = sums[0];
  dp = sums[1];
  gap_left = (int)mins[0];
  gap_right = (int)mins[1];
}

// minimiser k* of the convex sequence c(k), |k| <= G (theta in [-1, 1], :174-177); uniform over the slice
template <int EPT, int PMODE, int W>
__device__ int grid_solve(const float* s_val, const float* t_val, const Grid& gr, float mean_s, float mean_t, int lane, int tid,
                          float p, int p_int, SliceTeam<W>& team, int& evals) {
  const float Gf = (float)gr.G;
  int lo = -gr.G, hi = gr.G;
  float guess = rintf((mean_s - mean_t) * Gf);
  if (!(guess >= (float)lo)) guess = (float)lo;                     // (also non-finite input)
  if (!(guess <= (float)hi)) guess = (float)hi;
  int k = __builtin_amdgcn_readfirstlane((int)guess);
  int k_neg = 0, k_pos = 0, k_prev = 0, last_side = 0, secant_steps = 0;
  float f_neg = 0.f, f_pos = 0.f, f_prev = 0.f, step = 1.f;
  bool have_neg = false, have_pos = false, have_prev = false;
  evals = 0;
  for (int it = 0; it < 96; ++it) {
    float dm, dp;
    int gl, grt;
    grid_slopes<EPT, PMODE, W>(s_val, t_val, gr, k, lane, tid, p, p_int, team, dm, dp, gl, grt);
    ++evals;
    const bool right = (dp < 0.f) && (k < hi);
    const bool left = !right && (dm > 0.f) && (k > lo);
    if (!right && !left) break;                                    // dm <= 0 <= dp: k is a minimiser (:186-187)
    const float f = right ? dp : dm;
    if (right) {
      lo = min(k + max(grt, 1), hi);                               // the slope cannot change before the next kink
      k_neg = k; f_neg = dp; have_neg = true;
      if (last_side > 0 && have_pos) f_pos *= 0.5f;                // Illinois: the end that stays put loses weight
      last_side = 1;
    } else {
      hi = max(k - max(gl, 1), lo);
      k_pos = k; f_pos = dm; have_pos = true;
      if (last_side < 0 && have_neg) f_neg *= 0.5f;
      last_side = -1;
    }
    if (lo >= hi) { k = lo; break; }                               // one candidate left: the minimiser
    float next;
    if (have_neg && have_pos) {
      const float w = (float)(k_pos - k_neg);
      next = (float)k_neg + rintf(w * (-f_neg) / (f_pos - f_neg));
      if (!(next >= (float)lo && next <= (float)hi) || ++secant_steps > 24) next = (float)(lo + ((hi - lo) >> 1));
    } else {
      // p = 2: G c is ~quadratic in theta = k / G with curvature ~2 for clouds spread around the circle
      if (PMODE == 2 && !have_prev) step = fmaxf(step, 0.5f * fabsf(f) * Gf);
      next = (float)k + (right ? step : -step);
      if (have_prev && (f - f_prev) * (float)(k - k_prev) > 0.f) {
        const float root = (float)k - f * (float)(k - k_prev) / (f - f_prev);
        const float over = (float)k + 1.25f * (root - (float)k);
        next = right ? fmaxf(next, over) : fminf(next, over);
      }
      step *= 2.f;
    }
    next = fminf(fmaxf(rintf(next), (float)lo), (float)hi);
    k_prev = k; f_prev = f; have_prev = true;
    k = __builtin_amdgcn_readfirstlane((int)next);
  }
  return k;
}

// G * Cost at the shift k, the thread's share (sum over its source atoms); GRAD: G * d Cost / d (sorted source atom) into gs
template <int EPT, int PMODE, bool GRAD, int W>
__device__ float grid_cost_source(const float* s_val, const float* t_val, const Grid& gr, int k, int tid, float p, int p_int,
                                  float* gs) {
  constexpr int AP = EPT / W;
  const int a = gr.a, b = gr.b, n = gr.n, m = gr.m;
  const int trips = (a + b - 2) / b + 1;                           // a source atom's a cells meet at most this many targets
  float cost = 0.f;
  int e = tid * AP;
  // cells [e a, (e+1) a): the first one belongs to target t = floor((e a + k) / b), rb cells into it
  int kb, krem, t, rb, ah, al;
  floor_divmod(k, b, gr.inv_b, kb, krem);
  div_small(min(e, n - 1) * a + krem, b, gr.inv_b, t, rb);         // (< G + b <= 2^24)
  t += kb;
  div_small(a, b, gr.inv_b, ah, al);
#pragma nounroll
  for (int r = 0; r < AP; ++r, ++e) {
    const bool live = e < n;
    const float u = s_val[lds_slot<EPT>(min(e, n - 1))];
    float acc = 0.f, part = 0.f;
    int left = a, tt = t, off = rb;
    for (int s = 0; s < trips; ++s) {
      const int len = min(left, b - off);                          // cells shared with target tt (0 once the atom is used up)
      const float d = u - target_unrolled<EPT>(t_val, min(tt, 3 * m - 1), m);
      part = fmaf((float)len, powp<PMODE>(d, p, p_int), part);
      if constexpr (GRAD) acc = fmaf((float)len, dpow_abs<PMODE>(d, p, p_int), acc);
      left -= len;
      off = 0;
      ++tt;
    }
    cost += live ? part : 0.f;
    if constexpr (GRAD) {
      if (live) gs[lds_slot<EPT>(e)] = acc;
    }
    rb += al; t += ah;
    if (rb >= b) { rb -= b; ++t; }
  }
  return cost;
}

// GRAD: G * d Cost / d (sorted target atom) into gt.  Thread tid owns the extended target atoms T0 + [tid AP, (tid+1) AP),
// T0 = floor(k / b); when b does not divide k the first of them holds only part of its cells and the rest sit one turn
// later at T0 + m -- the owner of the last atom walks that instance too and hands its sum over in *tail (it belongs to
// sorted atom T0 mod m, which adds it to its own part: a fixed order).
template <int EPT, int PMODE, int W>
__device__ void grid_grad_target(const float* s_val, const float* t_val, const Grid& gr, int k, int tid, float p, int p_int,
                                 float* gt, float* tail) {
  constexpr int AP = EPT / W;
  const int a = gr.a, b = gr.b, n = gr.n, m = gr.m, G = gr.G;
  const int trips = (a + b - 2) / a + 1;                           // a target atom's b cells meet at most this many sources
  int T0, krem;
  floor_divmod(k, b, gr.inv_b, T0, krem);                          // T0 = floor(k / b)
  const int rho0 = tid * AP;
  const bool owns_tail = rho0 < m && rho0 + AP >= m;               // owner of the last extended atom
  const int rho_end = owns_tail ? m + 1 : min(rho0 + AP, m);
  int j = T0 + rho0;                                               // sorted atom of instance rho: (T0 + rho) mod m
  j += j < 0 ? m : 0;
  j -= j >= m ? m : 0;
  j -= j >= m ? m : 0;
#pragma nounroll
  for (int rho = rho0; rho < rho_end; ++rho) {
    const int t = T0 + rho;
    const float v = target_unrolled<EPT>(t_val, t, m);             // t in [-m, 2m]
    const int q_lo = max(t * b - k, 0);
    const int q_hi = min((t + 1) * b - k, G);                      // (rho = m with b | k: no cells, the sum is 0)
    int i, ra;
    div_small(min(q_lo, G - 1), a, gr.inv_a, i, ra);
    int left = max(q_hi - q_lo, 0);
    float acc = 0.f;
    for (int s = 0; s < trips; ++s) {
      const int len = min(left, a - ra);
      const float d = s_val[lds_slot<EPT>(min(i, n - 1))] - v;
      acc = fmaf((float)len, dpow_abs<PMODE>(d, p, p_int), acc);
      left -= len;
      ra = 0;
      ++i;
    }
    if (rho < m) gt[lds_slot<EPT>(j)] = -acc;
    else *tail = -acc;
    ++j;
    j -= j >= m ? m : 0;
  }
}

// project, sort (with indices), gather weights and build the CDF of ONE cloud of slice s (which = 0: target, 1: source);
// leaves the sorted values / CDF in LDS (dval, dcdf) and the sorted->original index map in registers.  `scratch` is a row
// for the coordinates by original index (it may be dval itself: the gather out of it is complete before the sorted values
// are written, LDS operations of a wave execute in order), `counters` 32 EPT words for the distribution sort.
template <int EPT, bool UNIFORM = false>
__device__ __forceinline__ void prepare_one(const GeneralArgs& G, int s, int lane, int which, float* dval, float* dcdf,
                                            float* scratch, unsigned* counters, int (&idx)[EPT], float& mean_out) {
  const SswArgs& A = G.base;
  const int b = s / A.slices, l = s - b * A.slices;
  const int n = A.n, m = A.m;
  float U[6];
  load_frame(A.dirs, (long)b * A.u_pair_stride + (long)l * 6, U);   // (3,2) row-major: U[2*d + k]
  const float* X = which == 0 ? A.xt + (long)b * m * A.pstride : A.xs + (long)b * n * A.pstride;
  const int count = which == 0 ? m : n;
  const float* Wt = which == 0 ? G.wv : G.wu;
  const long wstride = which == 0 ? G.wv_pair_stride : G.wu_pair_stride;
  int ln = lane;
  asm volatile("" : "+v"(ln));
  float val[EPT];
  // weighted, >= 8 atoms per lane: the distribution sort of bin_sort_idx.hpp (32 EPT counters beside the staging
  // row).  Without weights the one-wave kernel ran two waves per SIMD on 248 registers and the distribution sort's extra
  // live words spilled (measured in round 2: 2.1 -> 3.3 ms at n = 2048, m = 1536): it keeps the network.
  float part;
  if constexpr (EPT >= 8 && !UNIFORM) part = sorted_with_indices_binned<EPT, false, false>(X, count, ln, U, counters, scratch, val, idx);
  else part = sorted_with_indices<EPT>(X, count, ln, U, scratch, val, idx);
  float mean = 0.f;                                        // mass-weighted mean coordinate (first guess of the cut)
  if constexpr (UNIFORM) {
    mean = wave_sum_uniform(part, lane) / (float)count;                                 // CDF = (i+1)/count in closed form: no array
#pragma unroll
    for (int r = 0; r < EPT; ++r) dval[r * kWave + lane] = val[r];
  } else {
    float w[EPT];
#pragma unroll
    for (int r = 0; r < EPT; ++r) {
      const int e = lane * EPT + r;
      const bool live = e < count;
      w[r] = !live ? 0.f : (Wt ? Wt[(long)b * wstride + idx[r]] : 1.f / (float)count);
      mean += live ? w[r] * val[r] : 0.f;
    }
    mean = wave_sum_uniform(mean, lane);
    sorted_cdf<EPT>(w, lane);
#pragma unroll
    for (int r = 0; r < EPT; ++r) {                        // sorted position lane*EPT + r -> slot r*64 + lane
      dval[r * kWave + lane] = val[r];
      dcdf[r * kWave + lane] = (lane * EPT + r < count) ? w[r] : __builtin_inff();   // (window reads count on it)
    }
  }
  mean_out = mean;
  __builtin_amdgcn_wave_barrier();
}

// both clouds by ONE wave, the target first (the p = 1 kernels and the classes below 1024 points)
template <int EPT, bool UNIFORM = false>
__device__ __forceinline__ void prepare_sides(const GeneralArgs& G, int s, int lane, float* s_val, float* s_cdf,
                                              float* t_val, float* t_cdf, float* scratch, int (&sidx)[EPT],
                                              int (&tidx)[EPT], float& mean_s, float& mean_t,
                                              unsigned* counters = nullptr) {
  int idx[EPT];
#pragma nounroll
  for (int which = 0; which < 2; ++which) {                  // 0: target, 1: source
    float mean;
    prepare_one<EPT, UNIFORM>(G, s, lane, which, which == 0 ? t_val : s_val, which == 0 ? t_cdf : s_cdf, scratch, counters,
                              idx, mean);
    if (which == 0) {
      mean_t = mean;
#pragma unroll
      for (int r = 0; r < EPT; ++r) tidx[r] = idx[r];
    } else {
      mean_s = mean;
#pragma unroll
      for (int r = 0; r < EPT; ++r) sidx[r] = idx[r];
    }
  }
}

// gradient launch with index hand-off: sorted coordinates of one cloud (which = 0: target, 1: source) from the permutation
// the solve launch left in the coefficient rows (same projection arithmetic as load_coords, so the values are
// bit-identical to the ones that were sorted)
template <int EPT>
__device__ __forceinline__ void prepare_one_from_indices(const GeneralArgs& G, int s, int lane, int which, float* dval,
                                                         int (&idx)[EPT]) {
  const SswArgs& A = G.base;
  const int b = s / A.slices, l = s - b * A.slices;
  float U[6];
  load_frame(A.dirs, (long)b * A.u_pair_stride + (long)l * 6, U);   // (3,2) row-major: U[2*d + k]
  const int count = which == 0 ? A.m : A.n;
  const float* X = which == 0 ? A.xt + (long)b * count * A.pstride : A.xs + (long)b * count * A.pstride;
  const unsigned short* perm = reinterpret_cast<const unsigned short*>(
      which == 0 ? A.coef_t + (long)s * A.m : A.coef_s + (long)s * A.n);
#pragma unroll
  for (int r = 0; r < EPT; ++r) {
    const int e = lane * EPT + r;
    idx[r] = e < count ? (int)perm[min(e, count - 1)] : 0;
  }
#pragma unroll
  for (int r = 0; r < EPT; ++r) {
    const int e = lane * EPT + r;
    const float px = X[3 * idx[r]], py = X[3 * idx[r] + 1], pz = X[3 * idx[r] + 2];
    const float a = fmaf(pz, U[4], fmaf(py, U[2], fmaf(px, U[0], 0.f)));
    const float bb = fmaf(pz, U[5], fmaf(py, U[3], fmaf(px, U[1], 0.f)));
    dval[r * kWave + lane] = e < count ? circle_coord(a, bb) : __builtin_inff();
  }
  __builtin_amdgcn_wave_barrier();
}

template <int EPT, int PMODE, bool GRAD, bool UNIFORM, int W>
__global__ __launch_bounds__(64 * W, W > 1 ? (UNIFORM ? SHW_GENERAL_MINW_UNIFORM : 2) : 1) void ssw_general_kernel(GeneralArgs G) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int ROW = EPT * kWave;
  constexpr int AP = EPT / W;                                // sorted atoms per thread in the evaluations
  static_assert(EPT % W == 0 && AP >= 1, "atoms split evenly over the slice's threads");
  const SswArgs& A = G.base;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int tid = wave * kWave + lane;                       // thread of the slice: owns sorted atoms [tid AP, (tid+1) AP)
  // rows: sorted values of both clouds, their CDFs (weighted only), coordinates by original index (later the
  // source gradient row), target gradient row (GRAD only)
  float* s_val = lds;
  float* t_val = lds + ROW;
  constexpr int EXT = (UNIFORM ? 0 : general_ext_floats<EPT>());     // window rows under the source CDF (fill_walk_ext)
  float* s_cdf = UNIFORM ? nullptr : lds + 2 * ROW;
  float* t_cdf = UNIFORM ? nullptr : lds + 3 * ROW + EXT;
  float* grad_rows = lds + (UNIFORM ? 2 : 4) * ROW + EXT;    // GRAD only: two rows of coefficients by sorted position
  float* gs = grad_rows;
  float* gt = grad_rows + ROW;
  float* team_mem = lds + ((UNIFORM ? 2 : 4) + (GRAD ? 2 : 0)) * ROW + EXT;
  SliceTeam<W> team{team_mem, wave, 0};
  float* shared = team_mem + 8 * W;                          // [0], [1]: the two mean coordinates; [2]: the tail coefficient

  const int s = xcd_contiguous_id(blockIdx.x, A.num_groups);
  if (s >= A.pairs * A.slices) return;                       // (uniform over the workgroup)
  const int n = A.n, m = A.m;
  // sorted -> original index maps.  One wave per slice: [0] target, [1] source.  W waves: wave 0 sorts the source and
  // wave 1 the target at the same time, each keeps its own cloud's map in [0].
  constexpr int NI = W == 1 ? 2 : 1;
  int oidx[NI][EPT];
  float mean_s = 0.f, mean_t = 0.f;
  float handed_cut = 0.f;
  const bool from_indices = GRAD && UNIFORM && G.idx_handoff;
  if (from_indices) handed_cut = A.coef_t[(long)s * m + (m - 1)];   // read before the rows are reused
  if constexpr (W == 1) {
    if (from_indices) {
      prepare_one_from_indices<EPT>(G, s, lane, 0, t_val, oidx[0]);
      prepare_one_from_indices<EPT>(G, s, lane, 1, s_val, oidx[NI - 1]);
    } else {
      // sort counters (32 EPT words, weighted only): the source CDF row, which is written after both sorts' scatters;
      // loss only: the coordinates-by-original-index row of the sorts shares the source row
      unsigned* counters = reinterpret_cast<unsigned*>(s_cdf);
      float* scratch = GRAD ? gs : s_val;
#pragma nounroll
      for (int which = 0; which < 2; ++which) {
        float mean;
        int idx[EPT];
        prepare_one<EPT, UNIFORM>(G, s, lane, which, which == 0 ? t_val : s_val, which == 0 ? t_cdf : s_cdf, scratch,
                                  counters, idx, mean);
        if (which == 0) {
          mean_t = mean;
#pragma unroll
          for (int r = 0; r < EPT; ++r) oidx[0][r] = idx[r];
        } else {
          mean_s = mean;
#pragma unroll
          for (int r = 0; r < EPT; ++r) oidx[NI - 1][r] = idx[r];
        }
      }
    }
  } else {
    if (wave < 2) {
      const int which = 1 - wave;                            // wave 0: source, wave 1: target
      float* dval = which == 0 ? t_val : s_val;
      float* dcdf = which == 0 ? t_cdf : s_cdf;
      if (from_indices) {
        prepare_one_from_indices<EPT>(G, s, lane, which, dval, oidx[0]);
      } else {
        float mean;
        // each wave's own rows serve as its sort scratch: the value row takes the coordinates by original index, the
        // CDF row the counters; both are written with their final contents after the wave's gather
        // (without weights the sorts stay on the network: the distribution sort with indices needs half a row of counters
        //  per cloud -- 24 instead of 16 KB per slice -- and measured 2.06 against 1.79 ms at 2048 vs 1536 points)
        prepare_one<EPT, UNIFORM>(G, s, lane, which, dval, dcdf, dval, reinterpret_cast<unsigned*>(dcdf), oidx[0], mean);
        if (lane == 0) shared[which == 0 ? 1 : 0] = mean;
      }
    }
    __syncthreads();
    mean_s = shared[0];
    mean_t = shared[1];
  }
  if (!GRAD && G.idx_handoff) {                              // solve launch: leave the permutations for the gradient launch
    unsigned short* ps = reinterpret_cast<unsigned short*>(G.cut_scratch + (long)s * n);
    unsigned short* pt = reinterpret_cast<unsigned short*>(G.cut_scratch_t + (long)s * m);
    if (W == 1 || wave < 2) {
#pragma unroll
      for (int r = 0; r < EPT; ++r) {
        const int e = lane * EPT + r;
        if constexpr (W == 1) {
          if (e < n) ps[e] = (unsigned short)oidx[NI - 1][r];
          if (e < m) pt[e] = (unsigned short)oidx[0][r];
        } else {
          if (wave == 0 && e < n) ps[e] = (unsigned short)oidx[0][r];
          if (wave == 1 && e < m) pt[e] = (unsigned short)oidx[0][r];
        }
      }
    }
  }

  if constexpr (UNIFORM) {
    // ---- no weights: the solve on the integer grid of lcm(n, m) (grid_* above) ------------------------------------
    const Grid gr{n, m, G.lcm, G.lcm_a, G.lcm_b, 1.f / (float)G.lcm_a, 1.f / (float)G.lcm_b};
    const float inv_G = 1.f / (float)gr.G;
    int k;
    if constexpr (GRAD) {
      // (the gradient launch never solves: launch_general runs the loss-only kernel first, which hands the shift over as
      //  the bits of an int in the slice's own coefficient row)
      k = __float_as_int(G.idx_handoff ? handed_cut : G.cut_scratch[(long)s * G.cut_stride]);
    } else {
      int evals;
      k = grid_solve<EPT, PMODE, W>(s_val, t_val, gr, mean_s, mean_t, lane, tid, A.p, A.p_int, team, evals);
      // (index hand-off: the 16-bit permutation takes the first half of the target row, the shift its last word; m >= 2)
      if (G.idx_handoff) { if (tid == 0) G.cut_scratch_t[(long)s * m + (m - 1)] = __int_as_float(k); }
      else if (G.cut_scratch && tid == 0) G.cut_scratch[(long)s * G.cut_stride] = __int_as_float(k);
#ifdef SHW_DBG_EVALS
      if (G.slice_theta && tid == 0) G.slice_theta[s] = (float)evals;
#endif
    }
    float sums[1] = {wave_sum_uniform(grid_cost_source<EPT, PMODE, GRAD, W>(s_val, t_val, gr, k, tid, A.p, A.p_int, gs), lane)};
    if constexpr (GRAD) grid_grad_target<EPT, PMODE, W>(s_val, t_val, gr, k, tid, A.p, A.p_int, gt, shared + 2);
    team.sum(sums, lane);                                    // (W > 1: also the barrier that publishes gs, gt, tail)
    if (tid == 0) {
      A.slice_cost[s] = sums[0] * inv_G;
#ifndef SHW_DBG_EVALS
      if (G.slice_theta) G.slice_theta[s] = (float)k * inv_G;
#endif
    }
    if constexpr (GRAD) {
      if constexpr (W == 1) __builtin_amdgcn_wave_barrier();
      // un-permute through LDS (the value rows are dead now -- every wave is past its loops) and store coalesced
      const float tail = shared[2];
      int jstart, unused;
      floor_divmod(k, gr.b, gr.inv_b, jstart, unused);        // floor(k / b): its atom owns the tail
      jstart -= floor_div_m(jstart, m) * m;
      float* by_index_s = s_val;
      float* by_index_t = t_val;
      if (W == 1 || wave == 0) {
#pragma unroll
        for (int r = 0; r < EPT; ++r) {
          const int e = lane * EPT + r;
          if (e < n) by_index_s[oidx[NI - 1][r]] = gs[r * kWave + lane] * inv_G;
        }
      }
      if (W == 1 || wave == 1) {
#pragma unroll
        for (int r = 0; r < EPT; ++r) {
          const int e = lane * EPT + r;
          // (the atom of the first extended instance also owns the cells one turn later: own part first, then the tail)
          if (e < m) by_index_t[oidx[0][r]] = (gt[r * kWave + lane] + (e == jstart ? tail : 0.f)) * inv_G;
        }
      }
      if constexpr (W > 1) __syncthreads(); else __builtin_amdgcn_wave_barrier();
      float* cs = A.coef_s + (long)s * n;
      float* ct = A.coef_t + (long)s * m;
      for (int i = tid; i < max(n, m); i += kWave * W) {
        if (i < n) cs[i] = by_index_s[i];
        if (i < m) ct[i] = by_index_t[i];
      }
    }
  } else {
  Side<EPT> S{s_val, s_cdf, n}, T{t_val, t_cdf, m};
  if constexpr (general_walks<EPT>()) {
    if (wave == 0) fill_walk_ext<EPT>(s_cdf, lane);
    if constexpr (W > 1) __syncthreads(); else __builtin_amdgcn_wave_barrier();
  }

  // ---- the cut: minimiser of the convex, piecewise LINEAR cost over theta in [-1, 1] -------------------
  // The reference bisects [-1, 1] from theta = 0 on the sign of dCost until the bracket is below eps/L = 1e-7,
  // then intersects the two end tangents (:174-205): ~27 derivative evaluations.  Same exits here (kink:
  // dC+ * dC- <= 0; tangent intersection), but the bracket is grown around a first guess instead of halved
  // from [-1, 1]: moving the cut by theta moves every target atom by theta, so for p = 2 the optimum is the
  // difference of the mean coordinates (up to the kink spacing) and for other p it is near it; the search
  // steps out from there (doubling) until dCost changes sign, then bisects.  Both quantile functions are step
  // functions, so the cost is linear between kinks, and once the bracket is narrower than the smallest kink
  // spacing (G.min_width: the reference's eps/L = 1e-7) it holds at most one kink and the tangent intersection IS the
  // minimiser.  (Clouds without weights do not come here: the integer grid above.)
  float t_mid = 0.f;
#ifdef SHW_DBG_EVALS
  int dbg_evals = 0, dbg_bracket_at = -1;
#endif
  if constexpr (GRAD) {
    // the gradient launch never solves: launch_general always runs the loss-only kernel first and hands the cut over
    // (cut_given = 1) -- keeping the search out of this instantiation keeps its registers for the walks
    t_mid = G.idx_handoff ? handed_cut : G.cut_scratch[(long)s * G.cut_stride];
  } else {
    float t_lo = -1.f, t_hi = 1.f;
    t_mid = fminf(fmaxf(mean_s - mean_t, -1.f), 1.f);
    if (!(t_mid >= -1.f)) t_mid = 0.f;                         // non-finite input
    bool lo_tight = false, hi_tight = false;
    float step = G.first_step, dp_lo = 0.f, dm_hi = 0.f;
    float f_lo = 0.f, f_hi = 0.f, t_prev = 0.f, f_prev = 0.f;  // secant state (weighted clouds)
    constexpr int kChains = AP >= SHW_GENERAL_CHAINS ? SHW_GENERAL_CHAINS : AP;
    int anchors[kChains] = {};                                 // first ranks of the previous evaluation (cut_slopes_walk)
    float cost_scale = 0.f;                                    // size of the cost (cut_slopes_walk)
    int last_side = 0, secant_steps = 0;
    bool have_prev = false;
#ifndef SHW_DBG_MAX_EVALS
#define SHW_DBG_MAX_EVALS 96
#endif
    for (int it = 0; it < SHW_DBG_MAX_EVALS; ++it) {
#ifdef SHW_DBG_EVALS
      dbg_evals = it + 1;
      if (dbg_bracket_at < 0 && lo_tight && hi_tight) dbg_bracket_at = it;
#endif                          // <= ~25 doublings + ~25 halvings
      float dp, dm;
      if constexpr (!general_walks<EPT>()) {
        cut_slopes<EPT, PMODE, W>(S, T, t_mid, lane, tid, A.p, A.p_int, team, dp, dm);
      } else {
        cut_slopes_walk<EPT, PMODE, kChains, W>(S, T, t_mid, lane, tid, A.p, A.p_int, team, dp, dm, anchors, it > 0, cost_scale);
      }
#ifdef SHW_DBG_TRACE
      if (s < 4 && tid == 0) printf("slice %d it %d t %.9f dp %.4e dm %.4e lo %.9f hi %.9f\n", s, it, t_mid, dp, dm, t_lo, t_hi);
#endif
      if (dp * dm <= 0.f) break;                               // settled on a kink / flat piece (:186-187)
      if (!(dp * dm > 0.f)) break;                             // non-finite input: stop
      if (dp < 0.f) { t_lo = t_mid; lo_tight = true; dp_lo = dp; }
      else { t_hi = t_mid; hi_tight = true; dm_hi = dm; }
      {
        // by convexity an end of the bracket is within  width * |slope at that end|  of the minimum
        // COST.  Below one fp32 ulp of the cost (and below 1e-12 in any case) nothing is left to gain and the end is the
        // answer: with ~n*m micro-kinks the slope near the optimum is a noisy ~1e-6 and the search would spend its
        // last evaluations inside that noise; this ends it a few halvings before eps/L, and without the three cost
        // evaluations of the reference's finish, which resolve nothing at that scale.  Coinciding levels (equal
        // weights given explicitly: few, large kinks) keep large slopes on both sides and take the reference's exit.
        const float kGain = fmaxf(1e-12f, 1.2e-7f * cost_scale);
        if (lo_tight && hi_tight) {
          const float w = t_hi - t_lo;
          const float g_lo = -w * dp_lo, g_hi = w * dm_hi;
          if (fminf(g_lo, g_hi) < kGain) { t_mid = g_lo < g_hi ? t_lo : t_hi; break; }
        }
      }
      if ((t_hi - t_lo) < G.min_width) {                       // :189-200
        float unused;
        if (!lo_tight) cut_slopes<EPT, PMODE, W>(S, T, t_lo, lane, tid, A.p, A.p_int, team, dp_lo, unused);
        if (!hi_tight) cut_slopes<EPT, PMODE, W>(S, T, t_hi, lane, tid, A.p, A.p_int, team, unused, dm_hi);
        const float c_lo = cut_cost<EPT, PMODE, W>(S, T, t_lo, lane, tid, A.p, A.p_int, team);
        const float c_hi = cut_cost<EPT, PMODE, W>(S, T, t_hi, lane, tid, A.p, A.p_int, team);
        float t_c = (t_lo + t_hi) * 0.5f;
        if (fabsf(dp_lo - dm_hi) > 1e-3f) {                    // tangent intersection, :198-199 (written relative to
          // t_lo: the reference's form cancels terms of size theta * slope against each other)
          const float t_x = t_lo + (c_hi - c_lo - dm_hi * (t_hi - t_lo)) / (dp_lo - dm_hi);
          if (t_x == t_x) t_c = fminf(fmaxf(t_x, t_lo), t_hi);
        }
        // never end above a bracket end: an evaluation that lands within rounding of a kink can put that kink
        // ON an end, and the candidate then sits on the wrong side of it
        const float c_c = cut_cost<EPT, PMODE, W>(S, T, t_c, lane, tid, A.p, A.p_int, team);
        t_mid = t_c;
        float best = c_c;
        if (c_lo < best) { best = c_lo; t_mid = t_lo; }
        if (c_hi < best) { best = c_hi; t_mid = t_hi; }
        break;
      }
      {
        // Weighted clouds (round 2).  The cost has n*m kinks (every coincidence of a source level with a target
        // level), ~2.4e-7 apart at 2048 points: its one-sided slope is, at every scale above that, a smooth increasing
        // function -- exactly linear for p = 2 with the masses fixed.  Halving the bracket down to eps/L = 1e-7 as
        // the reference does takes ~17 evaluations after ~6 doublings; a secant on the slope gets there in 3-5:
        //   * no bracket yet: the line through the last two evaluations, over-stepped by a quarter (at least `step`);
        //   * bracket: regula falsi with the Illinois rule (an end that stays put twice has its slope halved, which
        //     throws the next point across the root) so that BOTH ends close in; the midpoint when the secant point
        //     is not strictly inside, and plain halving after kSecant steps.
        // The exit (width < eps/L, then the reference's tangent intersection) is unchanged.
        constexpr int kSecant = 24;
        const float f = dp < 0.f ? dp : dm;
        const int side = dp < 0.f ? -1 : 1;
        if (side < 0) { f_lo = f; if (last_side < 0 && hi_tight) f_hi *= 0.5f; }
        else { f_hi = f; if (last_side > 0 && lo_tight) f_lo *= 0.5f; }
        float t_next;
        if (lo_tight && hi_tight) {
          const float w = t_hi - t_lo;
          t_next = t_lo + w * (-f_lo) / (f_hi - f_lo);
          if (!(t_next > t_lo && t_next < t_hi) || ++secant_steps > kSecant) t_next = t_lo + 0.5f * w;
        } else {
          // p = 2: the cost is ~quadratic in the cut with curvature ~2 for clouds spread around the circle
          if (PMODE == 2 && !have_prev) step = fmaxf(step, SHW_GENERAL_FIRST_GAIN * fabsf(f));
          t_next = t_mid - (float)side * step;
          if (have_prev && (f - f_prev) * (t_mid - t_prev) > 0.f) {
            const float root = t_mid - f * (t_mid - t_prev) / (f - f_prev);
            const float over = t_mid + 1.25f * (root - t_mid);
            t_next = side < 0 ? fmaxf(t_next, over) : fminf(t_next, over);
          }
          t_next = fminf(fmaxf(t_next, t_lo), t_hi);
          step *= 2.f;
        }
        t_prev = t_mid; f_prev = f; have_prev = true; last_side = side;
        t_mid = t_next;
      }
    }
    if (G.cut_scratch && tid == 0) G.cut_scratch[(long)s * G.cut_stride] = t_mid;   // training: the gradient launch reads it
  }

  float cost;
  if constexpr (GRAD) {
    // Cost and its gradient at the cut: every atom's coefficient by its owner, once (walk_source_atoms)
    Rotated<EPT> R;
    R.set(T, t_mid, lane);
#ifndef SHW_DBG_WALK
#define SHW_DBG_WALK 0      // developer timing experiments only (wrong gradients): 1 = no target walk, 2 = no walk at all
#endif
    float part = 0.f;
    if (SHW_DBG_WALK < 2) part = walk_source_atoms<EPT, PMODE, W>(S, R, tid, A.p, A.p_int, gs);
    float sums[1] = {wave_sum_uniform(part, lane)};
    if (SHW_DBG_WALK < 1) walk_target_atoms<EPT, PMODE, W>(S, R, tid, A.p, A.p_int, gt, shared + 2);
    team.sum(sums, lane);                                    // (W > 1: also the barrier that publishes gs, gt, tail)
    cost = sums[0];
    if constexpr (W == 1) __builtin_amdgcn_wave_barrier();
    // un-permute through LDS (the value rows are dead now -- every wave is past its walks) and store coalesced: a
    // direct scatter writes one 4-byte word per cache line -- 2.3 ms per launch at n=2048, m=1536
    const float tail = shared[2];
    float* by_index_s = s_val;
    float* by_index_t = t_val;
    if (W == 1 || wave == 0) {
#pragma unroll
      for (int r = 0; r < EPT; ++r) {
        const int e = lane * EPT + r;
        if (e < n) by_index_s[oidx[NI - 1][r]] = gs[r * kWave + lane];
      }
    }
    if (W == 1 || wave == 1) {
#pragma unroll
      for (int r = 0; r < EPT; ++r) {
        const int e = lane * EPT + r;
        // (the first rotated atom also owns the coefficient of its copy one turn later: own part first, then the tail)
        if (e < m) by_index_t[oidx[0][r]] = gt[r * kWave + lane] + (e == R.start ? tail : 0.f);
      }
    }
    if constexpr (W > 1) __syncthreads(); else __builtin_amdgcn_wave_barrier();
    float* cs = A.coef_s + (long)s * n;
    float* ct = A.coef_t + (long)s * m;
    for (int i = tid; i < max(n, m); i += kWave * W) {
      if (i < n) cs[i] = by_index_s[i];
      if (i < m) ct[i] = by_index_t[i];
    }
  } else {
    cost = cut_cost<EPT, PMODE, W>(S, T, t_mid, lane, tid, A.p, A.p_int, team);
  }
  if (tid == 0) {
    A.slice_cost[s] = cost;
    if (G.slice_theta) G.slice_theta[s] = t_mid;
#ifdef SHW_DBG_EVALS                                          // developer aid: evaluations + 100 * (evaluations before the bracket)
    if (G.slice_theta && !G.cut_given) G.slice_theta[s] = (float)(dbg_evals + 100 * (dbg_bracket_at < 0 ? 0 : dbg_bracket_at));
#endif
  }
  }   // weighted clouds
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
  float mean_s_unused = 0.f, mean_t_unused = 0.f;
  prepare_sides<EPT>(G, s, lane, s_val, s_cdf, t_val, t_cdf, scratch, sidx, tidx, mean_s_unused, mean_t_unused);
  Side<EPT> S{s_val, s_cdf, n}, T{t_val, t_cdf, m};

  float lo_lev = __builtin_inff(), hi_lev = -__builtin_inff(), total = 0.f;
  constexpr int NA = EPT < 4 ? EPT : 4;                      // atoms searched together (see lower_bounds2)
#pragma nounroll
  for (int r0 = 0; r0 < EPT; r0 += NA) {
    float su[NA], sv[NA];
    int lt_u[NA], le_u[NA], lt_v[NA], le_v[NA];
#pragma unroll
    for (int a = 0; a < NA; ++a) {
      su[a] = S.v(min(lane * EPT + r0 + a, n - 1));
      sv[a] = T.v(min(lane * EPT + r0 + a, m - 1));
    }
    lower_bounds2_arr<EPT, NA>(t_val, m, su, lt_u, le_u);    // source atom: target values <  it
    lower_bounds2_arr<EPT, NA>(s_val, n, sv, lt_v, le_v);    // target atom: source values <= it
#pragma unroll
    for (int a = 0; a < NA; ++a) {
      const int r = r0 + a, e = lane * EPT + r;
      if (e < n) {                                           // source atom e
        const float val = su[a];
        const int lb = lt_u[a];
        const float lev = S.c(e) - (lb > 0 ? T.c(lb - 1) : 0.f);
        const float nxt = fminf(e + 1 < n ? S.v(e + 1) : __builtin_inff(), lb < m ? T.v(lb) : __builtin_inff());
        const float gap = (nxt == __builtin_inff() ? 1.f : nxt) - val;
        lev_s[r * kWave + lane] = lev;
        gap_s[r * kWave + lane] = gap;
        lo_lev = fminf(lo_lev, lev); hi_lev = fmaxf(hi_lev, lev); total += gap;
      }
      if (e < m) {                                           // target atom e
        const float val = sv[a];
        const int ub = le_v[a];
        const float lev = (ub > 0 ? S.c(ub - 1) : 0.f) - T.c(e);
        const float nxt = fminf(e + 1 < m ? T.v(e + 1) : __builtin_inff(), ub < n ? S.v(ub) : __builtin_inff());
        const float gap = (nxt == __builtin_inff() ? 1.f : nxt) - val;
        lev_t[r * kWave + lane] = lev;
        gap_t[r * kWave + lane] = gap;
        lo_lev = fminf(lo_lev, lev); hi_lev = fmaxf(hi_lev, lev); total += gap;
      }
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

// ---------------------------------------------------------------------------------------------
// p == 1 with weights, >= 8 atoms per lane: the same formula as ssw_general_p1_kernel with
//   * the two cross searches (target values below a source atom, source values not above a target atom) done by
//     walking (walk_window: the lane's atoms ascend, so do their ranks in the other cloud's values; round 2);
//   * levels and gaps in REGISTERS: the median bisection reads no LDS;
//   * TWO waves per slice (round 3): wave 0 owns the source cloud, wave 1 the target -- its sort (at the same time as the
//     other's), the levels and gaps of its atoms (one walk each, at the same time, straight into registers: the loop over a
//     lane's atoms is unrolled), its share of every masked sum of the median bisection (added in wave order through LDS, one
//     barrier per step) and its coefficient row.  Round 2's one-wave kernel kept the levels and gaps of BOTH clouds in
//     registers (256 VGPRs + AGPRs, one wave per SIMD, three slices per CU): 3.3 -> 1.6 ms per loss, 3.8 -> 2.4 per
//     training step at B = 64, n = m = 2048, L = 512; the loss-only form needs no staging rows (4 slices per CU);
//   * coefficients un-permuted through LDS and stored coalesced.
// ---------------------------------------------------------------------------------------------
template <int EPT, int C, bool SRC>
__device__ __forceinline__ int p1_levels_walk_regs(const Side<EPT>& O, const Side<EPT>& X, int lane, float (&lev_out)[EPT],
                                                   float (&gap_out)[EPT]) {
  constexpr int P = EPT * kWave;
  constexpr int LEN = EPT / C;
  const int no = O.count, nx = X.count;
  const float inf = __builtin_inff();
  int ptr[C];
  float prev[C], own_v[C];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const int e0 = lane * EPT + c * LEN;
    own_v[c] = O.val[lds_slot<EPT>(min(e0, P - 1))];
    prev[c] = O.val[lds_slot<EPT>(min(e0, no - 1))];
    ptr[c] = lower_bound_arr<EPT>(X.val, nx, prev[c]);
  }
  int first_rank = 0;
#pragma unroll
  for (int i = 0; i < LEN; ++i) {
    float k[C], val[C], nxt_own[C];
    bool live[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int e = lane * EPT + c * LEN + i;
      live[c] = e < no;
      val[c] = own_v[c];
      own_v[c] = e + 1 < P ? O.val[lds_slot<EPT>(min(e + 1, P - 1))] : inf;   // dead values are +inf in the row
      nxt_own[c] = own_v[c];
      k[c] = live[c] ? val[c] : prev[c];
      prev[c] = k[c];
    }
    int le[C];
    walk_window<EPT, C>(X.val, nx, k, ptr, le);
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int e = lane * EPT + c * LEN + i;
      const int rank = SRC ? ptr[c] : le[c];
      if (i == 0 && c == 0) first_rank = rank;
      const float below = rank > 0 ? X.c(rank - 1) : 0.f;
      const float mine = O.c(min(e, no - 1));
      const float lev = SRC ? mine - below : below - mine;
      const float cross = rank < nx ? X.v(min(rank, P - 1)) : inf;
      const float nxt = fminf(nxt_own[c], cross);
      const float gap = (nxt == inf ? 1.f : nxt) - val[c];
      lev_out[c * LEN + i] = live[c] ? lev : inf;
      gap_out[c * LEN + i] = live[c] ? gap : 0.f;
    }
  }
  return first_rank;
}

template <int EPT, bool GRAD>
__global__ __launch_bounds__(128, 2) void ssw_general_p1_walk2_kernel(GeneralArgs G) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int ROW = EPT * kWave, EXT = kWalkExt * kWave;
  constexpr int C = 2;
  const SswArgs& A = G.base;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  float* s_val = lds;                                        // each value row with its window rows
  float* t_val = s_val + ROW + EXT;
  float* s_cdf = t_val + ROW + EXT;
  float* t_cdf = s_cdf + ROW;
  float* stage = t_cdf + ROW;                                // GRAD only: [2][ROW] coefficients by original index
  float* team_mem = stage + (GRAD ? 2 * ROW : 0);
  SliceTeam<2> team{team_mem, wave, 0};
  float* shared = team_mem + 16;                             // [0..3] minima / maxima of the two waves

  const int s = xcd_contiguous_id(blockIdx.x, A.num_groups);
  if (s >= A.pairs * A.slices) return;
  const int n = A.n, m = A.m;
  const bool src = wave == 0;                                // wave 0: the source cloud, wave 1: the target
  int oidx[EPT];
  {
    float mean_unused;
    float* dval = src ? s_val : t_val;
    float* dcdf = src ? s_cdf : t_cdf;
    // the wave's own rows serve as its sort scratch (see ssw_general_kernel)
    prepare_one<EPT, false>(G, s, lane, src ? 1 : 0, dval, dcdf, dval, reinterpret_cast<unsigned*>(dcdf), oidx, mean_unused);
    fill_walk_ext<EPT>(dval, lane);
  }
  __syncthreads();
  Side<EPT> S{s_val, s_cdf, n}, T{t_val, t_cdf, m};
  const Side<EPT>& O = src ? S : T;                          // own cloud
  const int no = src ? n : m;

  float lev[EPT], gap[EPT];
  int rank0;
  if (src) rank0 = p1_levels_walk_regs<EPT, C, true>(S, T, lane, lev, gap);
  else rank0 = p1_levels_walk_regs<EPT, C, false>(T, S, lane, lev, gap);

  const float inf = __builtin_inff();
  float lo_lev = inf, hi_lev = -inf, total = 0.f;
#pragma unroll
  for (int r = 0; r < EPT; ++r) {
    lo_lev = fminf(lo_lev, lev[r]);
    hi_lev = fmaxf(hi_lev, lev[r] < inf ? lev[r] : -inf);
    total += gap[r];
  }
  lo_lev = -wave_max(-lo_lev, lane);
  hi_lev = wave_max(hi_lev, lane);
  if (lane == 0) { shared[wave] = lo_lev; shared[2 + wave] = hi_lev; }
  float sums[1] = {wave_sum_uniform(total, lane)};
  team.sum(sums, lane);                                      // (its barrier also publishes the minima / maxima)
  total = sums[0];
  lo_lev = as_f(__builtin_amdgcn_readfirstlane(as_i(fminf(shared[0], shared[1]))));
  hi_lev = as_f(__builtin_amdgcn_readfirstlane(as_i(fmaxf(shared[2], shared[3]))));

  auto weight_below = [&](float t) -> float {               // sum of gaps of the slice's atoms with level <= t
    float w = 0.f;
#pragma unroll
    for (int r = 0; r < EPT; ++r) w += (lev[r] <= t) ? gap[r] : 0.f;
    float sw[1] = {wave_sum_uniform(w, lane)};
    team.sum(sw, lane);
    return sw[0];
  };
  float med = lo_lev;
  if (total >= 0.5f) {
    float lo = lo_lev - 1.f, hi = hi_lev;                    // W(lo) = 0 < 0.5 <= W(hi) = total
    for (int it = 0; it < 48 && lo < hi; ++it) {
      const float mid = lo + (hi - lo) * 0.5f;
      if (!(mid > lo && mid < hi)) break;                    // bracket exhausted at fp32 resolution
      if (weight_below(mid) >= 0.5f) hi = mid; else lo = mid;
    }
    float best = inf;                                        // smallest level above the bracket's lower end
#pragma unroll
    for (int r = 0; r < EPT; ++r) best = (lev[r] > lo) ? fminf(best, lev[r]) : best;
    best = -wave_max(-best, lane);
    __syncthreads();                                         // (everyone has read the minima of the first exchange)
    if (lane == 0) shared[wave] = best;
    __syncthreads();
    med = as_f(__builtin_amdgcn_readfirstlane(as_i(fminf(shared[0], shared[1]))));
  }

  float acc = 0.f;
#pragma unroll
  for (int r = 0; r < EPT; ++r) {
    const int e = lane * EPT + r;
    if (e < no) {
      const float here = fabsf(lev[r] - med);
      acc += gap[r] * here;
      if constexpr (GRAD) {
        const float own = O.c(e) - (e > 0 ? O.c(e - 1) : 0.f);
        const bool first = (e == 0) && (rank0 == 0);         // no atom of the other cloud before (source) / at or before it
        // source: the level before the atom's own weight is lev - own; target: lev + own
        const float before = src ? lev[r] - own : lev[r] + own;
        stage[(src ? 0 : ROW) + oidx[r]] = (first ? 0.f : fabsf(before - med)) - here;
      }
    }
  }
  float cs1[1] = {wave_sum_uniform(acc, lane)};
  team.sum(cs1, lane);                                       // (GRAD: its barrier also publishes the staging rows)
  if (threadIdx.x == 0) {
    A.slice_cost[s] = cs1[0];
    if (G.slice_theta) G.slice_theta[s] = med;
  }
  if constexpr (GRAD) {
    float* cs = A.coef_s + (long)s * n;
    float* ct = A.coef_t + (long)s * m;
    for (int i = (int)threadIdx.x; i < max(n, m); i += 128) {
      if (i < n) cs[i] = stage[i];
      if (i < m) ct[i] = stage[ROW + i];
    }
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
  if (A.p == 1.f && !A.bisect_p1) {
    if constexpr (EPT >= 8) {
      const size_t lds2 = ((size_t)(grad ? 6 : 4) * EPT * kWave + 2 * kWalkExt * kWave + kTeamFloats) * sizeof(float);
      if (lds2 > 160 * 1024) return (int)hipErrorInvalidValue;
      if (grad) hipLaunchKernelGGL((ssw_general_p1_walk2_kernel<EPT, true>), grid, dim3(128), lds2, stream, G);
      else hipLaunchKernelGGL((ssw_general_p1_walk2_kernel<EPT, false>), grid, dim3(128), lds2, stream, G);
    } else {
      const size_t lds1 = (size_t)8 * EPT * kWave * sizeof(float);
      if (grad) hipLaunchKernelGGL((ssw_general_p1_kernel<EPT, true>), grid, block, lds1, stream, G);
      else hipLaunchKernelGGL((ssw_general_p1_kernel<EPT, false>), grid, block, lds1, stream, G);
    }
    return (int)hipGetLastError();
  }
  const bool uniform = G.wu == nullptr && G.wv == nullptr;   // no weights: CDFs in closed form, no searches
  constexpr int W = general_waves(EPT);                      // waves per slice
  const dim3 wblock(64 * W);
#define SHW_LAUNCH_GENERAL(PM, GR, ARGS)                                                                       \
  do {                                                                                                         \
    const size_t lds_ = ((size_t)((uniform ? 2 : 4) + ((GR) ? 2 : 0)) * EPT * kWave + kTeamFloats +            \
                         (uniform ? 0 : general_ext_floats<EPT>())) * sizeof(float) + SHW_DBG_EXTRA_LDS; \
    if (uniform) hipLaunchKernelGGL((ssw_general_kernel<EPT, PM, GR, true, W>), grid, wblock, lds_, stream, ARGS); \
    else hipLaunchKernelGGL((ssw_general_kernel<EPT, PM, GR, false, W>), grid, wblock, lds_, stream, ARGS);        \
  } while (0)
  if (!grad) {
    if (A.p_int == 2) SHW_LAUNCH_GENERAL(2, false, G);
    else SHW_LAUNCH_GENERAL(0, false, G);
    return (int)hipGetLastError();
  }
  // training: solve with the loss-only kernel (2 LDS rows, twice the waves per CU), then one gradient evaluation
  GeneralArgs solve = G;
  solve.base.coef_s = nullptr;
  solve.base.coef_t = nullptr;
  solve.cut_scratch = A.coef_s;                              // first word of each slice's own coefficient row
  solve.cut_stride = A.n;
  solve.cut_scratch_t = A.coef_t;
  // (coordinate-row mode re-sorts in the gradient launch: the hand-off re-projects gathered POINTS)
  const int handoff = (uniform && A.n >= 2 && A.m >= 2 && A.pstride == 3) ? 1 : 0;
  solve.idx_handoff = handoff;
  GeneralArgs eval = G;
  eval.cut_scratch = A.coef_s;
  eval.cut_scratch_t = A.coef_t;
  eval.cut_stride = A.n;
  eval.cut_given = 1;
  eval.idx_handoff = handoff;
  if (A.p_int == 2) { SHW_LAUNCH_GENERAL(2, false, solve); SHW_LAUNCH_GENERAL(2, true, eval); }
  else { SHW_LAUNCH_GENERAL(0, false, solve); SHW_LAUNCH_GENERAL(0, true, eval); }
#undef SHW_LAUNCH_GENERAL
  return (int)hipGetLastError();
}

static int gcd_general(int a, int b) {
  while (b) { const int t = a % b; a = b; b = t; }
  return a;
}

int dispatch_general(SswArgs& A, const float* wu, const float* wv, long wu_pair_stride, long wv_pair_stride,
                     float* slice_theta, hipStream_t stream) {
  GeneralArgs G{A, wu, wv, wu_pair_stride, wv_pair_stride, slice_theta, 0.f, 0.f, 0, 0, 0, nullptr, nullptr, 0, 0, 0};
  if (wu == nullptr && wv == nullptr) {                      // no weights: the integer grid of lcm(n, m)
    const long lcm = (long)A.n / gcd_general(A.n, A.m) * (long)A.m;
    G.lcm = (int)lcm;                                        // n, m <= 4096: < 2^24
    G.lcm_a = G.lcm / A.n;
    G.lcm_b = G.lcm / A.m;
  } else {
    G.first_step = 0.25f / (float)(A.n + A.m);
    G.min_width = 1e-7f;                                     // eps / L, :189
  }
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
