// shw_ssw_fwd.hip -- loss-only kernel for p != 1 (key-only register sort).  See ssw_common.hpp.
#include <cstdlib>

#include "ssw_common.hpp"
#include "bin_sort.hpp"

#ifndef SHW_BINSORT
#define SHW_BINSORT 1      // 1: distribution sort through LDS (bin_sort.hpp) for >= 8 keys per lane; 0: bitonic network only
#endif
#ifndef SHW_FWD_MINW
#define SHW_FWD_MINW 3     // waves per SIMD asked of the register allocator at 2048 points (bin sort)
#endif
#ifndef SHW_BINSORT_NB_PER_EPT
#define SHW_BINSORT_NB_PER_EPT 32   // bins = this * keys-per-lane (32: two keys per bin on average)
#endif
#ifndef SHW_FWD_WAVES
#define SHW_FWD_WAVES 1    // wavefronts per workgroup of the one-wave-per-slice kernel
#endif

namespace shw {

// register budget: the LDS footprint (EPT*256 B per wave) admits 20 waves per CU at EPT = 32, so
// ask the allocator for 5 waves per SIMD there (<= 96 VGPRs); larger EPT take what they need.
// The general-power variant (powf) is left at 4.  At EPT = 64 the unconstrained allocator takes all 256
// registers (1 wave per SIMD); asked for 3 waves it needs 144 with no spill, which is what 16 KB of LDS per
// wave can use.
constexpr int min_waves_per_simd(int ept, int pmode) {
  return ept <= 16 ? (pmode == 2 ? 6 : 4) : (ept == 32 ? (pmode == 2 ? 5 : 3) : (ept == 64 ? 3 : 1));
}

constexpr bool forward_uses_bins(int ept) { return SHW_BINSORT != 0 && ept >= 8; }
// LDS floats per wave: the sorted target row (64*EPT) and, with the bin sort, its 32*EPT counters in front of it
constexpr int forward_lds_floats(int ept) { return forward_uses_bins(ept) ? (64 + SHW_BINSORT_NB_PER_EPT) * ept : 64 * ept; }
#ifndef SHW_FWD_MINW_PARTIAL
#define SHW_FWD_MINW_PARTIAL 2     // clouds that do not fill the size class: 2 waves per SIMD, no spills (measured 0.371 -> 0.307 ms at N=2000)
#endif
constexpr int forward_min_waves(int ept, int pmode, bool full = true) {
  if (forward_uses_bins(ept) && ept == 32 && !full) return SHW_FWD_MINW_PARTIAL;
  // bin sort: 12 KB of LDS per wave at EPT = 32 -> 13 waves per CU: ask the register allocator for 3 per SIMD
  return forward_uses_bins(ept) ? (ept == 32 ? SHW_FWD_MINW : (ept == 16 ? 5 : 6)) : min_waves_per_simd(ept, pmode);
}

// FULL: n == m == 64*EPT (no padding atoms): mask-free projection and the fast shift evaluation.
template <int EPT, int WAVES, int PMODE, bool FULL>
__global__ __launch_bounds__(WAVES * 64, forward_min_waves(EPT, PMODE, FULL)) void ssw_forward_kernel(SswArgs A) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr bool BINS = forward_uses_bins(EPT);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* scratch = lds + wave * forward_lds_floats(EPT);
  float* vbuf = BINS ? scratch + SHW_BINSORT_NB_PER_EPT * EPT : scratch;       // the staging buffer of the sort becomes the target row

  const int vid = xcd_contiguous_id(blockIdx.x, A.num_groups);
  const int s = vid * WAVES + wave;
  if (s >= A.pairs * A.slices) return;            // wave-uniform
  const int b = s / A.slices, l = s - b * A.slices;

  float U[6];
  load_frame(A.dirs, (long)b * A.u_pair_stride + (long)l * 6, U);   // (3,2) row-major: U[2*d + k]

  float key[EPT], u[EPT];
  float sum_v = 0.f, sum_u = 0.f;
#ifdef SHW_DBG_RUNLEN
  int dbg_run = 0;
#endif
#pragma nounroll
  for (int which = 0; which < 2; ++which) {        // 0: source -> registers, 1: target -> LDS
    const float* X = which == 0 ? A.xs + (long)b * A.n * A.pstride : A.xt + (long)b * A.m * A.pstride;
    const int count = which == 0 ? A.n : A.m;
    // Opaque copy of the lane id: keeps the compiler from hoisting the lane-dependent constants of the sort (and
    // the point offsets) out of this loop and holding them in VGPRs.
    int ln = lane;
    asm volatile("" : "+v"(ln));
#ifdef SHW_ABL_NO_COORDS       // developer ablation: pseudo-random keys instead of load + project + atan2
    float part = 0.f;
#pragma unroll
    for (int r = 0; r < EPT; ++r) {
      unsigned h = (unsigned)(ln * 2654435761u) ^ (unsigned)((r + 33 * which + s) * 40503u);
      h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
      key[r] = (float)(h >> 8) * (1.0f / 16777216.0f);
      part += key[r];
    }
#else
    const float part = load_coords<EPT, FULL>(X, count, ln, U, key);
#endif
#ifdef SHW_DBG_RUNLEN          // developer build (tools/nonuniform_time.py): slice_shift reports the longest equal-bin run
    if constexpr (BINS) dbg_run = max(dbg_run, wave_sort_binned<EPT, FULL>(key, ln, count, scratch));
    else wave_sort<EPT>(key, ln);
#else
    if constexpr (BINS) wave_sort_binned<EPT, FULL>(key, ln, count, scratch);
    else wave_sort<EPT>(key, ln);
#endif
    if (which == 0) {
      sum_u = wave_sum(part, lane);
#pragma unroll
      for (int r = 0; r < EPT; ++r) u[r] = key[r];
    } else {
      sum_v = wave_sum(part, lane);
      if constexpr (FULL || !BINS) {
#pragma unroll
        for (int r = 0; r < EPT; ++r) vbuf[r * kWave + lane] = key[r];
      }
    }
  }
  __builtin_amdgcn_wave_barrier();

  float best;
#ifdef SHW_ABL_NO_SOLVE
  best = u[0] + vbuf[lane] + sum_u - sum_v;
  const int k = 0;
#else
  int k;
  if constexpr (FULL || !BINS) {
    k = solve_shift<EPT, PMODE, FULL>(u, vbuf, lane, A.n, sum_u, sum_v, A.p, A.p_int, best);
  } else {
    // any n: pre-rotated extended rows over the sort's scratch (counters + staging buffer are dead), target kept in
    // registers for the rare rewrite
    static_assert(ExtRows<EPT>::FLOATS <= forward_lds_floats(EPT), "extended rows must fit the sort's scratch");
    k = solve_shift_ext<EPT, PMODE>(u, key, scratch, lane, A.n, sum_u, sum_v, A.p, A.p_int, best);
  }
#endif
  if (lane == 0) {
    A.slice_cost[s] = best / (float)A.n;
    if (A.slice_shift) A.slice_shift[s] = k;
#ifdef SHW_DBG_RUNLEN
    if (A.slice_shift) A.slice_shift[s] = dbg_run;
#endif
  }
}

// ---------------------------------------------------------------------------------------------
// Multi-wave form for 2048 < n <= 8192: W = 2 or 4 wavefronts of one workgroup share a slice.  Each wave
// sorts its chunk of 2048 keys in registers exactly as above; the chunks are then merged by the remaining
// levels of the same bitonic network, whose first stages pair elements of DIFFERENT waves: those go through
// one LDS exchange each (write 32 registers, barrier, read the partner wave's slot, barrier), and because the
// "lower / upper" role of such a stage is the same for a whole wave they are plain v_min / v_max.  The stages
// below the wave level are the single-wave code (xlane_stages, lane_stages).  The source is sorted first and
// stays in registers while the target is sorted; the exchange buffer then becomes the sorted target, so LDS
// stays at 8 KB per wave and registers at the 2048-point budget (the one-wave kernels for these sizes need 64
// and 128 keys per lane: 1 wave per SIMD, and spills at 8192).  The shift solve runs in every wave on its own
// 2048 source atoms; the three partial sums are added across waves in wave order through LDS, so all waves
// take identical decisions.
// ---------------------------------------------------------------------------------------------
template <int W>
__device__ __forceinline__ void cross_wave_exchange(float (&key)[32], float* buf, int wave, int lane, int partner,
                                                    bool mirror, bool upper) {
  constexpr int NCOL = 64 * W;
#pragma unroll
  for (int r = 0; r < 32; ++r) buf[r * NCOL + wave * 64 + lane] = key[r];
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 32; ++r) {
    const float p = mirror ? buf[(31 - r) * NCOL + partner * 64 + (63 - lane)] : buf[r * NCOL + partner * 64 + lane];
    key[r] = upper ? __builtin_fmaxf(key[r], p) : __builtin_fminf(key[r], p);
  }
  __syncthreads();
}

template <int W>
__device__ __forceinline__ void merge_across_waves(float (&key)[32], float* buf, int wave, int lane) {
#pragma unroll
  for (int c = 1; (1 << c) <= W; ++c) {                       // merge blocks of 2^c waves
    cross_wave_exchange<W>(key, buf, wave, lane, wave ^ ((1 << c) - 1), true, (wave & (1 << (c - 1))) != 0);
#pragma unroll
    for (int t = c - 2; t >= 0; --t)
      cross_wave_exchange<W>(key, buf, wave, lane, wave ^ (1 << t), false, (wave & (1 << t)) != 0);
    xlane_stages<F32Keys, 32, 32>(key, lane);
    lane_stages<F32Keys, 32, 16>(key);
  }
}

template <int W, int PMODE, bool FULL>
__global__ __launch_bounds__(W * 64, 4) void ssw_forward_mw_kernel(SswArgs A) {
  constexpr int EPT = 32, NCOL = 64 * W, CHUNK = EPT * kWave;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* buf = lds;                                            // [EPT][NCOL]: exchange buffer, then sorted target
  float* red = lds + EPT * NCOL;                               // [W][4] partial sums
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int s = xcd_contiguous_id(blockIdx.x, A.num_groups);  // one workgroup per (pair, slice)
  const int b = s / A.slices, l = s - b * A.slices;
  const int n = A.n;
  float U[6];
  load_frame(A.dirs, (long)b * A.u_pair_stride + (long)l * 6, U);   // (3,2) row-major: U[2*d + k]

  float u[EPT], key[EPT];
  float part_u = 0.f, part_v = 0.f;
  const int first = wave * CHUNK;                              // first point of this wave's chunk
  const int chunk_live = max(0, min(CHUNK, n - first));
#pragma nounroll
  for (int which = 0; which < 2; ++which) {                    // 0: source -> registers, 1: target -> LDS
    const float* X = (which == 0 ? A.xs : A.xt) + (long)b * n * A.pstride;
    const int base = min(first, n - 1);                        // keep the addresses of an empty chunk in bounds
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const float part = load_coords<EPT, FULL>(X + (long)base * A.pstride, n - base, ln, U, key, chunk_live);
    wave_sort<EPT>(key, ln);
    merge_across_waves<W>(key, buf, wave, ln);
    if (which == 0) {
      part_u = wave_sum_uniform(part, lane);
#pragma unroll
      for (int r = 0; r < EPT; ++r) u[r] = key[r];
    } else {
      part_v = wave_sum_uniform(part, lane);
#pragma unroll
      for (int r = 0; r < EPT; ++r) buf[r * NCOL + wave * 64 + lane] = key[r];
    }
  }
  if (lane == 0) { red[wave * 4] = part_u; red[wave * 4 + 1] = part_v; }
  __syncthreads();
  float sum_u = 0.f, sum_v = 0.f;
#pragma unroll
  for (int w = 0; w < W; ++w) { sum_u += red[w * 4]; sum_v += red[w * 4 + 1]; }
  __syncthreads();

  // minimise the convex sequence c(k), |k| <= n: solve_shift with the partial sums added across waves
  int lo = -n, hi = n;
  float guess = rintf(sum_u - sum_v);
  guess = fminf(fmaxf(guess, (float)lo), (float)hi);
  int k = __builtin_amdgcn_readfirstlane((int)guess);
  bool lo_tight = false, hi_tight = false;
  int step = 1;
  float cm = 0.f, c0 = 0.f, cp = 0.f;
  const int glane = wave * 64 + lane;
  for (int it = 0; it < 64; ++it) {
    int gl = glane;
    asm volatile("" : "+v"(gl));
    float pm, p0, pp;
    if constexpr (FULL) shift_costs3_full<EPT, PMODE, NCOL>(u, buf, gl, k, A.p, A.p_int, pm, p0, pp);
    else shift_costs3<EPT, PMODE, NCOL>(u, buf, gl, n, k, A.p, A.p_int, pm, p0, pp);
    if (lane == 0) { red[wave * 4] = pm; red[wave * 4 + 1] = p0; red[wave * 4 + 2] = pp; }
    __syncthreads();
    cm = c0 = cp = 0.f;
#pragma unroll
    for (int w = 0; w < W; ++w) { cm += red[w * 4]; c0 += red[w * 4 + 1]; cp += red[w * 4 + 2]; }
    cm = as_f(__builtin_amdgcn_readfirstlane(as_i(cm)));
    c0 = as_f(__builtin_amdgcn_readfirstlane(as_i(c0)));
    cp = as_f(__builtin_amdgcn_readfirstlane(as_i(cp)));
    __syncthreads();
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

template <int W>
static int launch_forward_mw(SswArgs& A, hipStream_t stream) {
  const long total = (long)A.pairs * A.slices;
  if (total > 0x7fffffffL) return (int)hipErrorInvalidValue;
  A.num_groups = (int)total;
  const size_t lds = (size_t)(32 * 64 * W + 4 * W) * sizeof(float);
  const bool full = (A.n == 2048 * W) && (A.m == 2048 * W);
  const dim3 grid((unsigned)total), block(W * 64);
  if (A.p_int == 2) {
    if (full) hipLaunchKernelGGL((ssw_forward_mw_kernel<W, 2, true>), grid, block, lds, stream, A);
    else hipLaunchKernelGGL((ssw_forward_mw_kernel<W, 2, false>), grid, block, lds, stream, A);
  } else {
    if (full) hipLaunchKernelGGL((ssw_forward_mw_kernel<W, 0, true>), grid, block, lds, stream, A);
    else hipLaunchKernelGGL((ssw_forward_mw_kernel<W, 0, false>), grid, block, lds, stream, A);
  }
  return (int)hipGetLastError();
}

template <int EPT, int WAVES>
static int launch_forward(SswArgs& A, hipStream_t stream) {
  const long total = (long)A.pairs * A.slices;
  const long groups = (total + WAVES - 1) / WAVES;
  if (groups > 0x7fffffffL) return (int)hipErrorInvalidValue;
  A.num_groups = (int)groups;
  size_t lds = (size_t)WAVES * forward_lds_floats(EPT) * sizeof(float);
#ifdef SHW_DEV_OCCUPANCY_EXPERIMENT   // developer build only: pad the LDS request to lower the waves per CU
  if (const char* extra = getenv("SHW_DEV_EXTRA_LDS")) lds += (size_t)atoi(extra);
#endif
  const bool full = (A.n == EPT * kWave) && (A.m == EPT * kWave);
  const dim3 grid((unsigned)groups), block(WAVES * 64);
  if (A.p_int == 2) {
    if (full) hipLaunchKernelGGL((ssw_forward_kernel<EPT, WAVES, 2, true>), grid, block, lds, stream, A);
    else hipLaunchKernelGGL((ssw_forward_kernel<EPT, WAVES, 2, false>), grid, block, lds, stream, A);
  } else {
    if (full) hipLaunchKernelGGL((ssw_forward_kernel<EPT, WAVES, 0, true>), grid, block, lds, stream, A);
    else hipLaunchKernelGGL((ssw_forward_kernel<EPT, WAVES, 0, false>), grid, block, lds, stream, A);
  }
  return (int)hipGetLastError();
}

int dispatch_forward_coop(SswArgs& A, hipStream_t stream);       // shw_ssw_coop.hip
int dispatch_forward2(SswArgs& A, hipStream_t stream);           // shw_ssw_fwd2.hip

// Which loss-only kernel serves p != 1 (measured, profiles/r02_ab_twowave_fwd.txt):
//   n == m == 2048 exactly    : one wave per slice (ssw_forward_kernel, in-wave distribution sort) -- 0.228 ms at config 3
//                               against 0.240 for two waves, at the price of 25 spilled VGPRs
//   512..2048 (padded) points : otherwise two waves per slice, one cloud each (shw_ssw_fwd2.hip): no spills, and faster
//                               wherever the cloud does not fill its size class (N=2000: 0.306 vs 0.330 ms)
//   > 2048                    : W = padded / 2048 waves per slice, cooperative distribution sort (shw_ssw_coop.hip)
//   < 512                     : one wave per slice, register network below 8 keys per lane
// SHW_FORWARD_KERNEL (diagnostic, read once; the tests run every family): onewave = never two waves;
// twowave = two waves also at 2048 exactly; network = the bitonic multi-wave kernel above 2048 points;
// coop = the cooperative kernel from 2048 points on.
static int forward_family() {
  static const int fam = [] {
    const char* e = getenv("SHW_FORWARD_KERNEL");
    if (e && e[0] == 'n') return 1;       // network
    if (e && e[0] == 'c') return 2;       // coop from 2048
    if (e && e[0] == 't') return 3;       // twowave
    if (e && e[0] == 'o') return 4;       // onewave
    return 0;
  }();
  return fam;
}

int dispatch_forward_small_grid(SswArgs& A, hipStream_t stream);   // shw_ssw_coop.hip

// launches with at most this many (pair, slice) problems take the small-grid kernels (SHW_SMALL_GRID overrides; 0 = never)
static long small_grid_slices() {
  static const long v = [] {
    const char* e = getenv("SHW_SMALL_GRID");
    return e ? atol(e) : 1024L;
  }();
  return v;
}

int dispatch_forward(SswArgs& A, hipStream_t stream) {
#ifndef SHW_NO_COOP
  {
    const int padded = next_pow2(A.n > A.m ? A.n : A.m);
    const int fam = forward_family();
    // fewer problems than SIMDs: latency-bound, W waves per slice of 8 keys per lane
    if (fam == 0 && padded >= 512 && padded <= 2048 && (long)A.pairs * A.slices <= small_grid_slices())
      return dispatch_forward_small_grid(A, stream);
    if ((fam == 0 && padded > 2048) || (fam == 2 && padded >= 2048)) return dispatch_forward_coop(A, stream);
    if (padded >= 512 && padded <= 2048 && fam != 4 && fam != 1) {
      const bool headline = (A.n == 2048 && A.m == 2048);
      if (fam == 3 || (fam == 0 && !headline)) return dispatch_forward2(A, stream);
    }
  }
#endif
  switch (ept_for(A.n, A.m)) {
#ifdef SHW_DEV_ONLY_EPT   // developer switch: compile a single size class quickly
#ifndef SHW_DEV_FWD_WAVES
#define SHW_DEV_FWD_WAVES (SHW_DEV_ONLY_EPT <= 32 ? SHW_FWD_WAVES : (SHW_DEV_ONLY_EPT == 64 ? 2 : 1))
#endif
    case SHW_DEV_ONLY_EPT: return launch_forward<SHW_DEV_ONLY_EPT, SHW_DEV_FWD_WAVES>(A, stream);
#else
    case 1: return launch_forward<1, 4>(A, stream);
    case 2: return launch_forward<2, 4>(A, stream);
    case 4: return launch_forward<4, 4>(A, stream);
    case 8: return launch_forward<8, 4>(A, stream);
    case 16: return launch_forward<16, 4>(A, stream);
    case 32: return launch_forward<32, SHW_FWD_WAVES>(A, stream);
    case 64: return launch_forward_mw<2>(A, stream);            // 2049..4096 points: two waves per slice
    case 128: return launch_forward_mw<4>(A, stream);           // 4097..8192 points: four waves per slice
#endif
    default: return (int)hipErrorInvalidValue;
  }
}


}  // namespace shw
