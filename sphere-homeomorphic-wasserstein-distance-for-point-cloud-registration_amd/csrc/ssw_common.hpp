// ssw_common.hpp -- shared device code of the spherical sliced-Wasserstein hot path for MI355X (gfx950, wave64).
//
// One 64-lane wavefront owns one (pair, slice): it projects both clouds of the pair onto the
// slice's great circle, sorts both coordinate arrays in registers (wave_sort.hpp), parks the sorted
// target in LDS and solves the circular optimal-transport problem against the sorted source that
// stays in registers.  No barrier, no inter-wave communication, no MFMA (there is no contraction).
//
// Reference being replaced (paths relative to /root/reference/Point_Cloud_Resistration/losses/):
//   sliced_cost            max_spherical_sliced_w.py:251-286, _fast.py:258-295
//   binary_search_circle   :117-207   (p != 1)      emd1D_circle :210-247 (p == 1)
// The p != 1 solve uses the equivalence of SURVEY.md 8a row A8: for n == m and uniform weights the
// value the reference's bisection converges to is  min_k c(k),
//     c(k) = (1/n) sum_i |u_(i) - v_ext(i+k)|^p ,  v_ext(q) = v_(q mod n) + floor(q/n),
// a convex sequence in k.  The kernel starts at k0 = round(sum u - sum v) (exact minimiser when the
// target atoms are equally spaced, since sum_i (u_(i) - v_ext(i+k)) = sum u - sum v - k) and walks
// the convex sequence by galloping + bisection on the sign of c(k+1) - c(k).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdlib>

#include "../../include/shw.h"
#include "wave_sort.hpp"

#ifndef SHW_INC_MASK_BY_EXEC
#define SHW_INC_MASK_BY_EXEC 1
#endif
#ifndef SHW_SOLVE_INC
#define SHW_SOLVE_INC 1
#endif
#ifndef SHW_GRAD2_INC
#define SHW_GRAD2_INC 1
#endif
#ifndef SHW_MASK_BY_EXEC
#define SHW_MASK_BY_EXEC 1      // pads of a partially filled class masked by exec (a kept branch) instead of three selects
#endif

namespace shw {


struct SswArgs {
  const float* xs;
  const float* xt;
  const float* dirs;
  float* slice_cost;
  int32_t* slice_shift;
  float* coef_s;   // forward_grad only
  float* coef_t;
  int pairs, n, m, slices;
  int pstride;     // floats per point: 3 (clouds, projected on the slice's frame) or 1 (rows of circle coordinates: the
                   // circle-level entry point shw_circle_ot; dirs is NULL then and a "pair" is one row)
  long u_pair_stride;
  float p;
  int p_int;       // p if p is a small integer (1..8), else 0
  int bisect_p1;   // p == 1 only: 1 = the reference's BISECTION at p = 1 (binary_search_circle's default p, :117, ending in
                   // Cost's p == 1 branch :107-108) instead of the level-median formula of emd1D_circle; set by shw_circle_ot
  int num_groups;  // workgroups launched (for the XCD remap)
};

// ---------------------------------------------------------------------------------------------
// XCD-aware workgroup renumbering: hardware deals workgroups round-robin over the 8 XCDs
// (blockIdx % 8 labels the XCD); give every XCD one contiguous range of (pair, slice) work so the
// two clouds of a pair are fetched into ONE XCD's L2 instead of eight.  Bijective for any count.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int xcd_contiguous_id(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

// The slice's frame U (3,2) row-major, wave-uniform.  Coordinate-row mode (dirs == NULL): U[0] = NaN tells load_coords to
// take the input values as circle coordinates.
__device__ __forceinline__ void load_frame(const float* dirs, long offset, float (&U)[6]) {
  if (dirs) {
    const float* Ul = dirs + offset;
#pragma unroll
    for (int i = 0; i < 6; ++i) U[i] = Ul[i];
  } else {
#pragma unroll
    for (int i = 0; i < 6; ++i) U[i] = __builtin_nanf("");
  }
}

// circle coordinate of one projected point (reference :274-279):
//     coord = (atan2(-b, -a) + pi) / (2 pi)
// F.normalize (:274-275) is a positive rescale and cannot change the angle, so it is skipped.
// atan2 is evaluated as atan(min/max) with a degree-15 odd minimax polynomial
// (t + t^3 Q(t^2); with the 1.5-ulp quotient the total error is <= 1.9e-7 rad, rms 5.5e-8 rad, i.e.
// <= 3e-8 in coordinate units = half an fp32 ulp of a coordinate in [0.5, 1)) and octant fix-ups that
// follow the IEEE
// signed-zero rules the reference relies on: a = b = +0 gives atan2(-0,-0) = -pi, i.e. coord 0.
// Domain note: |a|,|b| below 1e-37 (denormal-scale projections) are treated as if max(|a|,|b|) were
// 1e-37, i.e. the angle of such a vector is not resolved; exact zeros are handled exactly.
__device__ __forceinline__ float circle_coord(float a, float b) {
  const float ax = fabsf(a), ay = fabsf(b);
  const float mx = fmaxf(fmaxf(ax, ay), 1e-37f), mn = fminf(ax, ay);
  const float t = mn * __builtin_amdgcn_rcpf(mx);  // v_rcp_f32 (1 ulp) * mn: t = mn/mx to 1.5 ulp
  const float s = t * t;
  float q = -4.3554045260e-03f;
  q = fmaf(q, s, 2.3040132597e-02f);
  q = fmaf(q, s, -5.7773582637e-02f);
  q = fmaf(q, s, 9.7942344844e-02f);
  q = fmaf(q, s, -1.3976581395e-01f);
  q = fmaf(q, s, 1.9962704182e-01f);
  q = fmaf(q, s, -3.3331659436e-01f);
  float r = fmaf(t * s, q, t);                     // atan(mn/mx) in [0, pi/4]
  r = (ay > ax) ? 1.57079637050628662f - r : r;    // angle from the x-axis, [0, pi/2]
  // x = -a is "negative" (incl. -0) exactly when the sign bit of a is clear
  r = (__builtin_bit_cast(int, a) >= 0) ? 3.14159274101257324f - r : r;
  const float ang = copysignf(r, -b);              // sign of y = -b (incl. signed zero)
  return (ang + 3.14159274101257324f) * 0.159154936671257019f;
}

// |d|^p.  PMODE 2: p == 2 (the reference's own call sites all use p = 2); PMODE 0: any p >= 1,
// small integer powers by repeated multiplication, otherwise powf.
template <int PMODE>
__device__ __forceinline__ float pow_abs(float d, float p, int p_int) {
  if constexpr (PMODE == 2) {
    return d * d;
  } else {
    const float a = fabsf(d);
    if (p_int > 0) {
      float r = a;
      for (int i = 1; i < p_int; ++i) r *= a;
      return r;
    }
    return powf(a, p);
  }
}

// d/dD |D|^p
template <int PMODE>
__device__ __forceinline__ float dpow_abs(float d, float p, int p_int) {
  if constexpr (PMODE == 2) {
    return 2.f * d;
  } else {
    const float a = fabsf(d);
    float r;
    if (p_int > 0) {
      r = 1.f;
      for (int i = 1; i < p_int; ++i) r *= a;
    } else {
      r = (a > 0.f) ? powf(a, p - 1.f) : 0.f;
    }
    return (a > 0.f) ? copysignf(p * r, d) : 0.f;
  }
}

// sorted target in LDS: sorted position q lives at [(q % EPT) * 64 + q / EPT]  (= register r of
// lane q/EPT, written with one conflict-free ds_write per register).
// NCOL = number of columns = lanes that hold the sorted array: 64 for one wave, 64*W when W waves of a
// workgroup sort one slice together (shw_ssw_fwd.hip, multi-wave kernel).
template <int EPT, int NCOL = 64>
__device__ __forceinline__ int lds_slot(int q) {
  if constexpr (is_pow2(EPT)) {
    constexpr int LOG = __builtin_ctz(EPT);
    return (q & (EPT - 1)) * NCOL + (q >> LOG);
  } else {                                          // (q >= 0: division by a constant)
    const unsigned col = (unsigned)q / (unsigned)EPT;
    return (int)(((unsigned)q - col * (unsigned)EPT) * (unsigned)NCOL + col);
  }
}

// floor(x / EPT) and x - EPT floor(x / EPT) for x >= -1024 EPT: shift and mask for the power-of-two classes
template <int EPT>
__device__ __forceinline__ int ediv(int x) {
  if constexpr (is_pow2(EPT)) return x >> __builtin_ctz(EPT);
  else return (int)((unsigned)(x + 1024 * EPT) / (unsigned)EPT) - 1024;
}
template <int EPT>
__device__ __forceinline__ int emod(int x) {
  if constexpr (is_pow2(EPT)) return x & (EPT - 1);
  else return x - ediv<EPT>(x) * EPT;
}

// v_ext(q) for q in [-2n, 3n): branch-free wrap onto [0, n) with the turn offset (two turns each
// way: the shift k ranges over [-n, n] and its neighbours k-1, k+1 are evaluated alongside)
template <int EPT, int NCOL = 64>
__device__ __forceinline__ float target_unrolled(const float* vbuf, int q, int n) {
  const int t1 = (q < 0) ? -1 : ((q >= n) ? 1 : 0);
  q -= t1 * n;
  const int t2 = (q < 0) ? -1 : ((q >= n) ? 1 : 0);
  q -= t2 * n;
  return vbuf[lds_slot<EPT, NCOL>(q)] + (float)(t1 + t2);
}

// c(k-1), c(k), c(k+1) (sums, not yet divided by n), valid in every lane.
// Lane owns sorted positions e0 .. e0+EPT-1 and needs v_ext(e0+k-1 .. e0+k+EPT): a sliding window,
// fetched 8 positions at a time to bound the registers in flight.
// `lane` is the index of the lane among the NCOL lanes that hold the slice (wave * 64 + lane in the multi-wave
// kernel, whose waves then add their partial sums).
// NR / r_base: the evaluation may cover only registers [r_base, r_base + NR) of every lane's EPT sorted positions
// (u then holds those NR atoms): the two-wave training kernel gives each wave half of the source.
template <int EPT, int PMODE, int NCOL = 64, int NR = EPT>
__device__ __forceinline__ void shift_costs3(const float (&u)[NR], const float* vbuf, int lane, int n,
                                             int k, float p, int p_int, float& cm, float& c0, float& cp,
                                             int r_base = 0) {
  float sm = 0.f, s0 = 0.f, sp = 0.f;
  const int e0 = lane * EPT + r_base;
  const int last = n - 1;
  float prev = target_unrolled<EPT, NCOL>(vbuf, min(e0, last) + k - 1, n);
  float cur = target_unrolled<EPT, NCOL>(vbuf, min(e0, last) + k, n);
  constexpr int CH = chunk_of(NR);
#pragma unroll
  for (int r0 = 0; r0 < NR; r0 += CH) {
    float nxt[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) nxt[j] = target_unrolled<EPT, NCOL>(vbuf, min(e0 + r0 + j, last) + k + 1, n);
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const bool live = (e0 + r0 + j) < n;
      const float a = pow_abs<PMODE>(u[r0 + j] - prev, p, p_int);
      const float b = pow_abs<PMODE>(u[r0 + j] - cur, p, p_int);
      const float c = pow_abs<PMODE>(u[r0 + j] - nxt[j], p, p_int);
      sm += live ? a : 0.f;
      s0 += live ? b : 0.f;
      sp += live ? c : 0.f;
      prev = cur;
      cur = nxt[j];
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  cm = wave_sum_uniform(sm, lane & 63);
  c0 = wave_sum_uniform(s0, lane & 63);
  cp = wave_sum_uniform(sp, lane & 63);
}

// one fetch of shift_costs3_inc (arguments by value: as captured references the offsets end up as a table in scratch
// memory read through flat pointers)
__device__ __forceinline__ float inc_fetch(const char* row, bool before_turn, bool first_column, int a0, int a1, int b0,
                                           float t0, float tw) {
  const int a = before_turn ? (first_column ? a0 : a1) : b0;
  return *reinterpret_cast<const float*>(row + a) + (before_turn ? t0 : tw);
}
// the last fetch of a window of EPT + 2 positions (one-wave kernels): it may lie in the third column of the window, or in
// the second one after the end of the circle
__device__ __forceinline__ float inc_fetch_last(const char* row, bool before_turn, bool second_column, bool second_after_turn,
                                                int a1, int a2, int b0, int b1, float t0, float tw) {
  const int a = before_turn ? (second_column ? a1 : a2) : (second_after_turn ? b1 : b0);
  return *reinterpret_cast<const float*>(row + a) + (before_turn ? t0 : tw);
}

// shift_costs3 for any n with 6 VALU per fetched target atom instead of the ~15 of target_unrolled.  The window of a
// lane (NR + 2 consecutive positions) starts at slot (row0, col0) of the plain rows; while it is no longer than one column
// (NR + 2 <= EPT: the two-wave training kernel, NR = EPT/2) it meets at most one end of column (the address drops by a
// constant) and at most one end of the circle (position n: it restarts at slot (0, 0) one turn further on, and is then too
// short to meet anything else).  With the two per-lane thresholds jr = EPT - row0 and jn = n - q0 prepared once, fetch j is
//     ds_read(j*256 + (j < jn ? (j < jr ? A0 : A1) : B0)) + (j < jn ? t0 : t0 + 1).
// A window of EPT + 2 positions (NR = EPT: the one-wave kernels) can reach one column further with its LAST position only
// (jr >= 1, jn >= 1), which gets its own form.  Pads are masked by exec as in shift_costs3_ext.
template <int EPT, int PMODE, int NCOL = 64, int NR = EPT>
__device__ __forceinline__ void shift_costs3_inc(const float (&u)[NR], const float* vbuf, int lane, int n,
                                                 int k, float p, int p_int, float& cm, float& c0, float& cp,
                                                 int r_base = 0) {
  static_assert(NR <= EPT, "a window of at most EPT + 2 positions");
  int q = lane * EPT + r_base + k - 1;              // in [-n - 1, 64 EPT + n) within [-2n, 3n): two turns at most
  const int t1 = (q < 0) ? -1 : ((q >= n) ? 1 : 0);
  q -= t1 * n;
  const int t2 = (q < 0) ? -1 : ((q >= n) ? 1 : 0);
  q -= t2 * n;
  const float t0 = (float)(t1 + t2), tw = t0 + 1.f;
  const int col0 = ediv<EPT>(q), row0 = q - col0 * EPT;
  const int jr = EPT - row0, jn = n - q;
  constexpr int RB = NCOL * 4;                      // bytes per row
  const int A0 = row0 * RB + (col0 << 2);
  const int A1 = A0 + 4 - EPT * RB;
  const int B0 = -jn * RB;
  const char* rows = reinterpret_cast<const char*>(vbuf);
  auto fetch = [&](int j) -> float {                // j compile-time after unrolling
    if (j <= EPT) return inc_fetch(rows + j * RB, j < jn, j < jr, A0, A1, B0, t0, tw);
    return inc_fetch_last(rows + j * RB, j < jn, j < jr + EPT, j >= jn + EPT, A1, A1 + 4 - EPT * RB, B0, B0 + 4 - EPT * RB,
                          t0, tw);
  };
  const int live_regs = n - lane * EPT - r_base;
  float sm = 0.f, s0 = 0.f, sp = 0.f;
  float prev = fetch(0), cur = fetch(1);
  constexpr int CH = chunk_of(NR);
#pragma unroll
  for (int r0 = 0; r0 < NR; r0 += CH) {
    float nxt[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) nxt[j] = fetch(r0 + j + 2);
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const float a = pow_abs<PMODE>(u[r0 + j] - prev, p, p_int);
      const float b = pow_abs<PMODE>(u[r0 + j] - cur, p, p_int);
      const float c = pow_abs<PMODE>(u[r0 + j] - nxt[j], p, p_int);
#if SHW_INC_MASK_BY_EXEC
      if ((r0 + j) < live_regs) {
        asm volatile("" : "+v"(sm), "+v"(s0), "+v"(sp));
        sm += a; s0 += b; sp += c;
      }
#else
      const bool live = (r0 + j) < live_regs;
      sm += live ? a : 0.f;
      s0 += live ? b : 0.f;
      sp += live ? c : 0.f;
#endif
      prev = cur;
      cur = nxt[j];
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  cm = wave_sum_uniform(sm, lane & 63);
  c0 = wave_sum_uniform(s0, lane & 63);
  cp = wave_sum_uniform(sp, lane & 63);
}

// Fast form of shift_costs3 for n == 64*EPT exactly (every register slot is a real atom, n a power of
// two).  With base = k-1 = kh*EPT + kl (0 <= kl < EPT) the window position j of lane l is sorted
// position (l + kh + c)*EPT + row with  row = (kl + j) mod EPT  and carry  c = (kl + j) div EPT in
// {0,1,2}: row and c are WAVE-UNIFORM, so each fetch is  ds_read(addr_c + row*256) + turn_c  with the
// three per-lane (address, turn) pairs prepared once per evaluation -- 4 VALU per fetch instead of ~15.
template <int EPT, int PMODE, int NCOL = 64, int NR = EPT>
__device__ __forceinline__ void shift_costs3_full(const float (&u)[NR], const float* vbuf, int lane, int k,
                                                  float p, int p_int, float& cm, float& c0, float& cp,
                                                  int r_base = 0) {
  constexpr int LOG = __builtin_ctz(EPT);
  const int base = k - 1 + r_base;            // registers [r_base, r_base + NR): the same window, r_base further on
  const int kl = base & (EPT - 1);
  const int kh = base >> LOG;                                   // floor division (arithmetic shift)
  int addr[3];
  float turn[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int col = lane + kh + c;                              // unwrapped lane index of the atom
    addr[c] = (col & (NCOL - 1)) << 2;                          // byte offset inside an LDS row
    turn[c] = (float)(col >> __builtin_ctz(NCOL));              // whole turns around the circle
  }
  const char* rows = reinterpret_cast<const char*>(vbuf);
  auto fetch = [&](int j) -> float {                            // j compile-time after unrolling
    const int rj = kl + j;                                      // scalar
    const int row = rj & (EPT - 1);
    const bool carry = (j < EPT) ? (rj >= EPT) : (rj >= 2 * EPT);
    const int a = (j < EPT) ? (carry ? addr[1] : addr[0]) : (carry ? addr[2] : addr[1]);
    const float t = (j < EPT) ? (carry ? turn[1] : turn[0]) : (carry ? turn[2] : turn[1]);
    return *reinterpret_cast<const float*>(rows + a + row * (NCOL * 4)) + t;
  };
  float sm = 0.f, s0 = 0.f, sp = 0.f;
  float prev = fetch(0), cur = fetch(1);
  constexpr int CH = NR < 8 ? NR : 8;
#pragma unroll
  for (int r0 = 0; r0 < NR; r0 += CH) {
    float nxt[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) nxt[j] = fetch(r0 + j + 2);
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      sm += pow_abs<PMODE>(u[r0 + j] - prev, p, p_int);
      s0 += pow_abs<PMODE>(u[r0 + j] - cur, p, p_int);
      sp += pow_abs<PMODE>(u[r0 + j] - nxt[j], p, p_int);
      prev = cur;
      cur = nxt[j];
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  cm = wave_sum_uniform(sm, lane & 63);
  c0 = wave_sum_uniform(s0, lane & 63);
  cp = wave_sum_uniform(sp, lane & 63);
}

// Minimise the convex sequence c(k), |k| <= n (theta in [-1, 1], the reference's bracket :174-177).
// Returns k*, writes c(k*) (sum form).
template <int EPT, int PMODE, bool FULL = false>
__device__ __forceinline__ int solve_shift(const float (&u)[EPT], const float* vbuf, int lane, int n,
                                           float sum_u, float sum_v, float p, int p_int, float& best) {
  int lo = -n, hi = n;
  float guess = rintf(sum_u - sum_v);
  guess = fminf(fmaxf(guess, (float)lo), (float)hi);
  int k = __builtin_amdgcn_readfirstlane((int)guess);         // k lives in an SGPR from here on
  bool lo_tight = false, hi_tight = false;
  int step = 1;
  float cm, c0, cp;
  // every iteration removes k from [lo, hi]; galloping doubles, bisection halves: <= ~2 log2(2n)+2
  // iterations.  The hard cap only guards against non-finite input (comparisons all false -> exit).
  for (int it = 0; it < 64; ++it) {
    int ln = lane;                       // opaque copy: no lane-derived constants held across iterations
    asm volatile("" : "+v"(ln));
    if constexpr (FULL) shift_costs3_full<EPT, PMODE>(u, vbuf, ln, k, p, p_int, cm, c0, cp);
    else if constexpr (SHW_SOLVE_INC && EPT <= 16) shift_costs3_inc<EPT, PMODE>(u, vbuf, ln, n, k, p, p_int, cm, c0, cp);
    else shift_costs3<EPT, PMODE>(u, vbuf, ln, n, k, p, p_int, cm, c0, cp);   // (diagnostic one-wave kernels of the big classes)
    const bool right = (cp < c0) && (k < hi);
    const bool left = !right && (cm < c0) && (k > lo);
    if (!right && !left) break;
    if (right) {
      lo = k + 1;
      lo_tight = true;
      if (hi_tight) { k = lo + ((hi - lo) >> 1); }
      else { k = min(k + step, hi); step <<= 1; }
    } else {
      hi = k - 1;
      hi_tight = true;
      if (lo_tight) { k = lo + ((hi - lo) >> 1); }
      else { k = max(k - step, lo); step <<= 1; }
    }
    k = __builtin_amdgcn_readfirstlane(k);
  }
  best = c0;
  return k;
}

// ---------------------------------------------------------------------------------------------
// Any n (not only n == 64*EPT): the sorted target as PRE-ROTATED, EXTENDED rows.
//
// The generic evaluation (shift_costs3 / target_unrolled) pays ~15 VALU per fetched target atom for the wrap-around
// arithmetic (two conditional turns, a multiply, the slot computation).  Instead the target is written to LDS already
// rotated to the first guess kc of the shift and with the turns added:
//     ext[E] = v_ext(E - M + kc),   E in [0, n + 2M),   v_ext(q) = v[q mod n] + floor(q / n),   M = EPT,
// laid out [E mod EPT][E div EPT] with 66 columns per row (64 lanes + the halo).  A shift k = kc + d with |d| < M
// then reads ext[e + d + delta + M] for source atom e: no wrap, no turn, every address = lane*4 + a wave-uniform
// offset -- 1 VALU per fetch.  The search seldom leaves |d| < M (the first guess is exact for evenly spaced
// targets and galloping starts with steps 1, 2, 4, ...); when it does the rows are rewritten around the new k.
// ---------------------------------------------------------------------------------------------
// NCOL > 64 (cooperative kernel): NCOL lanes of NCOL/64 waves hold the slice, `lane` is the lane number within the
// slice and the writers synchronise with workgroup barriers.
template <int EPT, int NCOL = 64>
struct ExtRows {
  static constexpr int M = EPT;                 // half width of the halo = reach of the shift without a rewrite
  static constexpr int RS = NCOL + 2;           // columns per row: (NCOL*EPT + 2M) / EPT
  static constexpr int FLOATS = EPT * RS;
  static constexpr int TRASH = ((EPT - 1) * RS + NCOL + 1) * 4;   // byte offset of the last slot: never a valid E for n < NCOL*EPT
};

// v[r] = sorted target at position lane*EPT + r (anything at positions >= n).  n > M required (size classes give n > 32*EPT).
template <int EPT, int NCOL = 64>
__device__ __forceinline__ void ext_rows_write(const float (&v)[EPT], float* ext, int lane, int n, int kc) {
  typedef ExtRows<EPT, NCOL> X;
  // c1 = (M - kc) mod n and the turn s it absorbs: ext[p + c1] = v[p] - s, or, past the end, ext[p + c1 - n] = v[p] - s - 1
  const int t = X::M - kc;                                       // in [M - n, M + n]
  const int s = t < 0 ? -1 : (t >= n ? 1 : 0);
  const int c1 = t - s * n;                                      // in [0, n)
  const int c2 = c1 - n;                                         // in [-n, 0)
  const int c1h = ediv<EPT>(c1), c1l = emod<EPT>(c1);
  const int c2h = ediv<EPT>(c2), c2l = emod<EPT>(c2);            // floor division
  const float vn = (float)(-s), vw = (float)(-s - 1);
  char* bytes = reinterpret_cast<char*>(ext);
  const int lane4 = lane << 2;
  const int first_wrapped = n - c1;                              // positions p >= this wrap
#pragma unroll
  for (int r = 0; r < EPT; ++r) {
    const int a1 = r + c1l, a2 = r + c2l;                        // scalars
    const int off_n = ((emod<EPT>(a1) * X::RS) + c1h + ediv<EPT>(a1)) << 2;
    const int off_w = ((emod<EPT>(a2) * X::RS) + c2h + ediv<EPT>(a2)) << 2;
    const int pos = lane * EPT + r;
    const bool wrap = pos >= first_wrapped;
    int off = wrap ? off_w : off_n;
    off = pos < n ? off + lane4 : X::TRASH;
    *reinterpret_cast<float*>(bytes + off) = v[r] + (wrap ? vw : vn);
  }
  if constexpr (NCOL > 64) __syncthreads(); else __builtin_amdgcn_wave_barrier();
  // halo: ext[n + l] = ext[l] + 1 for l < 2M (one entry per lane; 2M <= 64)
  {
    const int src = lane, dst = n + lane;
    const int so = ((emod<EPT>(src) * X::RS) + ediv<EPT>(src)) << 2;
    const int d_o = ((emod<EPT>(dst) * X::RS) + ediv<EPT>(dst)) << 2;
    const float x = *reinterpret_cast<const float*>(bytes + so);
    *reinterpret_cast<float*>(bytes + (lane < 2 * X::M ? d_o : X::TRASH)) = x + 1.f;
  }
  if constexpr (NCOL > 64) __syncthreads(); else __builtin_amdgcn_wave_barrier();
}

// c(k-1), c(k), c(k+1) for k = kc + d, |d| < M, on registers [r_base, r_base + NR) of every lane (u holds those NR
// source atoms; atoms at positions >= n are masked out).  Sums over the wave, valid in every lane.
template <int EPT, int PMODE, int NR = EPT, int NCOL = 64>
__device__ __forceinline__ void shift_costs3_ext(const float (&u)[NR], const float* ext, int lane, int n, int d,
                                                 float p, int p_int, float& cm, float& c0, float& cp, int r_base = 0) {
  typedef ExtRows<EPT, NCOL> X;
  const int base = d - 1 + X::M + r_base;                        // >= 0, wave-uniform
  const int bl = emod<EPT>(base), bh = ediv<EPT>(base);
  const char* rows = reinterpret_cast<const char*>(ext) + (lane << 2);
  auto fetch = [&](int j) -> float {                             // j compile-time after unrolling
    const int a = bl + j;                                        // scalar
    const int off = ((emod<EPT>(a) * X::RS) + bh + ediv<EPT>(a)) << 2;
    return *reinterpret_cast<const float*>(rows + off);
  };
  const int live_regs = n - lane * EPT - r_base;                 // registers j < live_regs hold real atoms
  float sm = 0.f, s0 = 0.f, sp = 0.f;
  float prev = fetch(0), cur = fetch(1);
  constexpr int CH = chunk_of(NR);
#pragma unroll
  for (int r0 = 0; r0 < NR; r0 += CH) {
    float nxt[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) nxt[j] = fetch(r0 + j + 2);
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const float a = pow_abs<PMODE>(u[r0 + j] - prev, p, p_int);
      const float b = pow_abs<PMODE>(u[r0 + j] - cur, p, p_int);
      const float c = pow_abs<PMODE>(u[r0 + j] - nxt[j], p, p_int);
#if SHW_MASK_BY_EXEC
      // pads add nothing: a branch the compiler must keep (the empty asm), i.e. one v_cmp and an exec mask around three
      // full-rate adds -- as selects the three accumulations cost a quarter-rate v_cndmask each (op rates)
      if ((r0 + j) < live_regs) {
        asm volatile("" : "+v"(sm), "+v"(s0), "+v"(sp));
        sm += a; s0 += b; sp += c;
      }
#else
      const bool live = (r0 + j) < live_regs;
      sm += live ? a : 0.f;
      s0 += live ? b : 0.f;
      sp += live ? c : 0.f;
#endif
      prev = cur;
      cur = nxt[j];
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  cm = wave_sum_uniform(sm, lane & 63);
  c0 = wave_sum_uniform(s0, lane & 63);
  cp = wave_sum_uniform(sp, lane & 63);
}

// solve_shift on extended rows (one wave owns the slice).  v = the sorted target in registers (kept for rewrites).
template <int EPT, int PMODE>
__device__ __forceinline__ int solve_shift_ext(const float (&u)[EPT], const float (&v)[EPT], float* ext, int lane, int n,
                                               float sum_u, float sum_v, float p, int p_int, float& best) {
  typedef ExtRows<EPT> X;
  int lo = -n, hi = n;
  float guess = rintf(sum_u - sum_v);
  guess = fminf(fmaxf(guess, (float)lo), (float)hi);
  int k = __builtin_amdgcn_readfirstlane((int)guess);
  int kc = k;
  ext_rows_write<EPT>(v, ext, lane, n, kc);
  bool lo_tight = false, hi_tight = false;
  int step = 1;
  float cm, c0, cp;
  for (int it = 0; it < 64; ++it) {
    int ln = lane;
    asm volatile("" : "+v"(ln));
    if (k - kc >= X::M || kc - k >= X::M) {                      // wave-uniform: re-centre the rows on k
      kc = k;
      ext_rows_write<EPT>(v, ext, ln, n, kc);
    }
    shift_costs3_ext<EPT, PMODE>(u, ext, ln, n, k - kc, p, p_int, cm, c0, cp);
    const bool right = (cp < c0) && (k < hi);
    const bool left = !right && (cm < c0) && (k > lo);
    if (!right && !left) break;
    if (right) {
      lo = k + 1;
      lo_tight = true;
      if (hi_tight) { k = lo + ((hi - lo) >> 1); }
      else { k = min(k + step, hi); step <<= 1; }
    } else {
      hi = k - 1;
      hi_tight = true;
      if (lo_tight) { k = lo + ((hi - lo) >> 1); }
      else { k = max(k - step, lo); step <<= 1; }
    }
    k = __builtin_amdgcn_readfirstlane(k);
  }
  best = c0;
  return k;
}

// Project the cloud onto the slice's circle: lane owns points r*64 + lane (coalesced 12-byte
// records).  Padding keys are +inf so that they sort behind every real coordinate.
// CHAINED: make the addresses of a chunk depend (through an empty asm) on the last coordinate of the chunk
// before it.  Needed where the loads are not inside a loop: the compiler otherwise issues all 3*EPT loads up
// front and the raw points take 96 registers (shw_ssw_p1_merge.hip).
// NCOL: lanes that share the cloud (64 for one wave; 64*W when W waves of a workgroup own a slice together, `lane`
// then being the index among those lanes): lane owns points r*NCOL + lane.
// FOLD: how the masked classes take coordinate-row mode, see below.
template <int EPT, bool FULL = false, bool CHAINED = false, int NCOL = kWave, bool FOLD = false>
__device__ __forceinline__ float load_coords(const float* __restrict__ X, int count, int lane,
                                             const float (&U)[6], float (&key)[EPT], int live_count = -1) {
  // `count` bounds the addresses (clamp), `live_count` (default: count) says how many of the 64*EPT slots are
  // real atoms; they differ only for the trailing chunks of the multi-wave kernel
  if (live_count < 0) live_count = count;
  float acc = 0.f;
  // coordinate-row mode (shw_circle_ot): X holds circle coordinates, one float per atom (U[0] is NaN then): a branch
  // -- or, FOLD, part of the one code path (record stride 1 or 3 and a final select).  As a branch it left both forms'
  // state live at the join in the two-wave training kernel's masked classes: 14 -> 39 spilled registers, N=2000
  // training 0.62 -> 0.80 ms; folded it costs 2 + 1 VALU per point.  (The loss kernels spill less with the branch.)
  const bool rows = U[0] != U[0];
  constexpr bool kFold = FOLD && !FULL;
  if constexpr (!kFold) {
    if (rows) {
#pragma unroll
      for (int r = 0; r < EPT; ++r) {
        const int i = r * NCOL + lane;
        const float c = X[FULL ? i : min(i, count - 1)];
        const bool live = FULL || (i < live_count);
        acc += live ? c : 0.f;
        key[r] = live ? c : __builtin_inff();
      }
      return acc;
    }
  }
  const int wide = (kFold && rows) ? 0 : -1;       // all ones: 12-byte records
  const int o1 = (kFold && rows) ? 0 : 1, o2 = (kFold && rows) ? 0 : 2;
  constexpr int CH = chunk_of(EPT);                // 8 points (24 loads) in flight per lane (4 or 5 for the odd classes)
#pragma unroll
  for (int r0 = 0; r0 < EPT; r0 += CH) {
    if constexpr (!FULL && !kFold) {
      // lane owns points r*NCOL + lane: from row ceil(live_count / NCOL) on EVERY lane's point is a pad -- a wave-uniform
      // test per chunk skips their loads, projections and arctangents (N=1200 in the 2048 class: a quarter of them;
      // loss 0.290 -> 0.263 ms).  Not in the indexed sort's loader (FOLD): there the branches cost the nearly full
      // classes more than they save the others (N=2000 training 0.654 -> 0.681 ms).
      if (r0 * NCOL >= live_count) {
#pragma unroll
        for (int j = 0; j < CH; ++j) key[r0 + j] = __builtin_inff();
        continue;
      }
    }
    float px[CH], py[CH], pz[CH];
    if constexpr (CHAINED) {
      if (r0 > 0) asm volatile("" : "+v"(lane) : "v"(key[r0 > 0 ? r0 - 1 : 0]));
    }
    // lane owns points r*NCOL + lane: when every point of the chunk's rows exists (uniform over the wave) the chunk is
    // the code of a full class -- no clamped addresses, no masks; only the one mixed chunk of a cloud pays for them
    const bool whole = FULL || (!is_pow2(EPT) && (r0 + CH) * NCOL <= (live_count < count ? live_count : count));   // (classes: bin_sort.hpp)
    if (whole) {
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        const int i = (r0 + j) * NCOL + lane;
        const int i3 = kFold ? i + ((i & wide) << 1) : 3 * i;
        px[j] = X[i3]; py[j] = X[i3 + (kFold ? o1 : 1)]; pz[j] = X[i3 + (kFold ? o2 : 2)];
      }
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        const float a = fmaf(pz[j], U[4], fmaf(py[j], U[2], fmaf(px[j], U[0], 0.f)));
        const float b = fmaf(pz[j], U[5], fmaf(py[j], U[3], fmaf(px[j], U[1], 0.f)));
        float c = circle_coord(a, b);
        if constexpr (kFold) c = rows ? px[j] : c;
        acc += c;
        key[r0 + j] = c;
      }
    } else {
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        const int raw = (r0 + j) * NCOL + lane;
        const int i = min(raw, count - 1);                        // clamp: branch-free, always in bounds
        const int i3 = kFold ? i + ((i & wide) << 1) : 3 * i;     // 3 i, or i for rows of coordinates
        px[j] = X[i3]; py[j] = X[i3 + (kFold ? o1 : 1)]; pz[j] = X[i3 + (kFold ? o2 : 2)];
      }
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        const int i = (r0 + j) * NCOL + lane;
        // fma(x, u, +0): a sum that starts from +0 like the reference's matmul accumulator (:270), so that an
        // all-zero point projects to (+0, +0) -- never -0 -- and lands on coordinate 0 (G4 fixture)
        const float a = fmaf(pz[j], U[4], fmaf(py[j], U[2], fmaf(px[j], U[0], 0.f)));
        const float b = fmaf(pz[j], U[5], fmaf(py[j], U[3], fmaf(px[j], U[1], 0.f)));
        float c = circle_coord(a, b);
        if constexpr (kFold) c = rows ? px[j] : c;
        const bool live = i < live_count;
        acc += live ? c : 0.f;
        key[r0 + j] = live ? c : __builtin_inff();
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  return acc;
}


// ---------------------------------------------------------------------------------------------
// sort WITH the permutation (training and weighted kernels): packed 32-bit keys, see shw_ssw_grad.hip
// ---------------------------------------------------------------------------------------------
template <int EPT>
struct Packing {
  static constexpr int IDX_BITS = log2_ceil_c(EPT * kWave);
  static constexpr int QBITS = 32 - IDX_BITS;
  static constexpr unsigned IDX_MASK = (1u << IDX_BITS) - 1u;
  static __device__ __forceinline__ unsigned pack(float coord, int idx, bool live) {
    // coord in [0, 1]; 2^QBITS * coord is exact in fp32 for QBITS <= 26 and fits 32 bits
    const float scaled = coord * (float)(1u << (QBITS > 26 ? 26 : QBITS));
    unsigned q = (unsigned)scaled;
    if constexpr (QBITS > 26) q <<= (QBITS - 26);
    const unsigned qmax = (QBITS >= 32) ? 0xffffffffu : ((1u << QBITS) - 1u);
    q = q < qmax ? q : qmax;
    return live ? ((q << IDX_BITS) | (unsigned)idx) : 0xffffffffu;
  }
};

// (value, index) pair order: ascending value, ties by ascending index
__device__ __forceinline__ bool pair_after(float va, int ia, float vb, int ib) {
  return (va > vb) || (va == vb && ia > ib);
}

// Put a nearly sorted (val, idx) sequence -- sorted position lane*EPT + r -- into exact stable order.
// One round = exchange of pairs (2j, 2j+1) then (2j+1, 2j+2); rounds repeat until a round is clean.
template <int EPT>
__device__ __forceinline__ void exact_order_fixup(float (&val)[EPT], int (&idx)[EPT], int lane,
                                                  int max_rounds = EPT * kWave) {
  for (int round = 0; round < max_rounds; ++round) {        // bound: odd-even transposition sorts in n rounds
    bool any = false;
    auto exch = [&](float& va, int& ia, float& vb, int& ib) {
      const bool sw = pair_after(va, ia, vb, ib);
      const float tv = va; const int ti = ia;
      va = sw ? vb : va; ia = sw ? ib : ia;
      vb = sw ? tv : vb; ib = sw ? ti : ib;
      any |= sw;
    };
#pragma unroll
    for (int r = 0; r + 1 < EPT; r += 2) exch(val[r], idx[r], val[r + 1], idx[r + 1]);
#pragma unroll
    for (int r = 1; r + 1 < EPT; r += 2) exch(val[r], idx[r], val[r + 1], idx[r + 1]);
    {  // boundary pair: this lane's last atom against the next lane's first
      const int up = min(lane + 1, 63) << 2, dn = max(lane - 1, 0) << 2;
      const float nv = as_f(__builtin_amdgcn_ds_bpermute(up, as_i(val[0])));
      const int ni = __builtin_amdgcn_ds_bpermute(up, idx[0]);
      const float pv = as_f(__builtin_amdgcn_ds_bpermute(dn, as_i(val[EPT - 1])));
      const int pi = __builtin_amdgcn_ds_bpermute(dn, idx[EPT - 1]);
      const bool sw_up = (lane < 63) && pair_after(val[EPT - 1], idx[EPT - 1], nv, ni);
      const bool sw_dn = (lane > 0) && pair_after(pv, pi, val[0], idx[0]);
      if constexpr (EPT == 1) {
        // a lane's single atom can be wanted by both neighbours: alternate even / odd boundaries
        const bool even_phase = (round & 1) == 0;
        const bool do_up = sw_up && (((lane & 1) == 0) == even_phase);
        const bool do_dn = sw_dn && (((lane & 1) == 1) == even_phase);
        val[0] = do_up ? nv : (do_dn ? pv : val[0]);
        idx[0] = do_up ? ni : (do_dn ? pi : idx[0]);
        any |= sw_up || sw_dn;
      } else {
        val[EPT - 1] = sw_up ? nv : val[EPT - 1]; idx[EPT - 1] = sw_up ? ni : idx[EPT - 1];
        val[0] = sw_dn ? pv : val[0]; idx[0] = sw_dn ? pi : idx[0];
        any |= sw_up || sw_dn;
      }
    }
    if (__builtin_amdgcn_readfirstlane((int)(__ballot(any) != 0ull)) == 0) break;
  }
}

// From the sorted packed words to (exact coordinate, original index) in exact stable order: gather the fp32
// coordinates by index from `orig` (coordinates by ORIGINAL index, pads +inf) and repair the order of atoms whose
// quantised coordinates collide.  Shared by the network sort (sorted_with_indices) and the distribution sort
// (bin_sort_idx.hpp).
// FULL: count == 64*EPT (no pads): the pad tests fold away.
template <int EPT, bool FULL = false>
__device__ __forceinline__ void unpack_sorted_words(const unsigned (&pk)[EPT], const float* orig, int count, int lane,
                                                    float (&val)[EPT], int (&idx)[EPT]) {
  typedef Packing<EPT> PK;
  // A pad is recognised by its index field (all ones, >= count whenever pads exist), NOT by the key value:
  // the real atom with index 64*EPT-1 and a coordinate in the top quantisation cell packs to the same word
  // 0xffffffff when the row is full.
  // Equal quantised coordinate on adjacent atoms?  (`chain`: on three in a row -- only then can one repair round be
  // not enough.)
  bool collide = false, chain = false;
  bool prev_eq;
  {
    const unsigned nxt = (unsigned)__builtin_amdgcn_ds_bpermute(min(lane + 1, 63) << 2, (int)pk[0]);
    const unsigned prv = (unsigned)__builtin_amdgcn_ds_bpermute(max(lane - 1, 0) << 2, (int)pk[EPT - 1]);
    collide = (lane < 63) && (((pk[EPT - 1] ^ nxt) >> PK::IDX_BITS) == 0) &&
              (FULL || (int)(pk[EPT - 1] & PK::IDX_MASK) < count);
    prev_eq = (lane > 0) && (((pk[0] ^ prv) >> PK::IDX_BITS) == 0) && (FULL || (int)(pk[0] & PK::IDX_MASK) < count);
  }
  const bool next_eq = collide;                           // (last atom of this lane, first atom of the next)
#pragma unroll
  for (int r = 1; r < EPT; ++r) {
    const bool eq = (((pk[r] ^ pk[r - 1]) >> PK::IDX_BITS) == 0) && (FULL || (int)(pk[r] & PK::IDX_MASK) < count);
    chain |= eq && prev_eq;
    collide |= eq;
    prev_eq = eq;
  }
  chain |= prev_eq && next_eq;                            // ... (EPT-2, EPT-1, first of the next lane)
  if constexpr (EPT == 1) chain |= collide;               // one atom per lane: every collision touches a lane boundary
  // (the packed words are dead from here on: index and gathered coordinate take their registers)
#pragma unroll
  for (int r = 0; r < EPT; ++r) {
    idx[r] = (int)(pk[r] & PK::IDX_MASK);
    val[r] = (!FULL && idx[r] >= count) ? __builtin_inff() : orig[FULL ? idx[r] : min(idx[r], count - 1)];
  }
#ifndef SHW_ABL_NO_EXACTFIX
  if (__builtin_amdgcn_readfirstlane((int)(__ballot(collide) != 0ull)) != 0) {
    // one round repairs every colliding PAIR (ties inside a pair are already in index order: the packed order); longer
    // chains of equal quantised coordinates (rare) get the loop-until-clean form
    const bool chained = __builtin_amdgcn_readfirstlane((int)(__ballot(chain) != 0ull)) != 0;
    exact_order_fixup<EPT>(val, idx, lane, chained ? EPT * kWave : 1);
  }
#endif
  __builtin_amdgcn_wave_barrier();
}

// project one cloud, sort it (packed keys), recover exact sorted coordinates + original indices
template <int EPT>
__device__ __forceinline__ float sorted_with_indices(const float* __restrict__ X, int count, int lane,
                                                     const float (&U)[6], float* orig, float (&val)[EPT],
                                                     int (&idx)[EPT]) {
  typedef Packing<EPT> PK;
  unsigned pk[EPT];
  float part;
  {
    float key[EPT];
    part = load_coords<EPT>(X, count, lane, U, key);
    // pin the coordinate sum HERE: left alone, the compiler sinks the 64*EPT additions below the sort and
    // keeps every unsorted coordinate alive in a register across it
    asm volatile("" : "+v"(part));
#pragma unroll
    for (int r = 0; r < EPT; ++r) {
      const int i = r * kWave + lane;
      orig[i] = key[r];                                  // exact coordinate by ORIGINAL index (pads: +inf)
      pk[r] = PK::pack(key[r], i, i < count);
    }
  }
  wave_sort<EPT>(pk, lane);
  __builtin_amdgcn_wave_barrier();
  unpack_sorted_words<EPT>(pk, orig, count, lane, val, idx);
  return part;
}

// ---------------------------------------------------------------------------------------------
// host-side helpers shared by the per-kernel translation units
// ---------------------------------------------------------------------------------------------
inline int small_integer_power(float p) {
  const int q = (int)p;
  return ((float)q == p && q >= 1 && q <= 8) ? q : 0;
}

inline int next_pow2(int v) {
  int r = 1;
  while (r < v) r <<= 1;
  return r;
}


// size class: registers per lane (EPT) for the padded point count
inline int ept_for(int n, int m) {
  const int padded = next_pow2(n > m ? n : m);
  return padded <= 64 ? 1 : padded / 64;
}

// keys per lane of the two-wave kernels of 513..2048 points (round 3): the power-of-two classes plus 12, 20, 24 and 28, so
// that a cloud pays for the next multiple of 256 points (768: of 256 x 3) and not for the next power of two -- the
// notebooks' 1200 points (Flow_cube.ipynb:200) take 1280 slots instead of 2048.  SHW_KPL_CLASSES=0 keeps powers of two.
inline int kpl_for(int n, int m, bool training = false) {
  const int big = n > m ? n : m;
  const int e = ept_for(n, m);
  static const bool fine = [] { const char* v = getenv("SHW_KPL_CLASSES"); return !(v && v[0] == '0'); }();
  if (!fine || e < 16) return e;
  for (int k = e / 2 + 4; k < e; k += 4) {           // 16: 12;  32: 20, 24, 28
    // (28 keys per lane: the loss kernel gains -- N = 1700: 0.299 -> 0.277 ms -- the training kernel does not: 0.632 -> 0.642)
    if (k * 64 >= big && !(training && k == 28)) return k;
  }
  return e;
}

// keys per lane of the cooperative kernels above 2048 points (W = 2 or 4 waves per slice): 20, 24 or 32.  Measured per
// pair at B N ~ 131 k, L = 512 (profiles/r03_size_sweep.txt): 28 keys per lane is never faster than 32 (the partially
// filled 32 class costs the same), nor is 24 at W = 4 in training; 20 and 24 pay (N = 3000: training 1.40 -> 0.83 ms,
// N = 5000: 1.88 -> 0.89).
inline int coop_kpl_for(int points, int W, bool training) {
  static const bool fine = [] { const char* v = getenv("SHW_KPL_CLASSES"); return !(v && v[0] == '0'); }();
  if (!fine) return 32;
  if (20 * 64 * W >= points) return 20;
  if (24 * 64 * W >= points && !(training && W == 4)) return 24;
  return 32;
}

// dispatchers, one per translation unit (SswArgs validated by the C entry points in shw_capi.hip)
int dispatch_forward(SswArgs& A, hipStream_t stream);        // shw_ssw_fwd.hip   p != 1, loss only
int dispatch_forward_grad(SswArgs& A, hipStream_t stream);   // shw_ssw_grad.hip  p != 1, loss + coefficients
int dispatch_forward_grad2(SswArgs& A, hipStream_t stream);  // shw_ssw_grad2.hip two waves per slice, 257..2048 points
int dispatch_forward_grad_coop(SswArgs& A, hipStream_t stream);   // shw_ssw_grad_coop.hip 2 / 4 waves per slice, 2049..8192 points
int dispatch_level_median(SswArgs& A, hipStream_t stream);   // shw_ssw_p1.hip    p == 1 (coef_s != NULL: + coefficients)
int dispatch_general(SswArgs& A, const float* wu, const float* wv, long wu_pair_stride, long wv_pair_stride,
                     float* slice_theta, hipStream_t stream);   // shw_ssw_general.hip  p != 1, n != m / weights
int launch_backward_points(const float* xs, const float* xt, const float* dirs, const float* coef_s,
                           const float* coef_t, int pairs, int n, int m, int slices, long u_pair_stride,
                           float scale, const float* pair_w, const float* total_w, float* grad_xs, float* grad_xt,
                           hipStream_t stream);

}  // namespace shw
