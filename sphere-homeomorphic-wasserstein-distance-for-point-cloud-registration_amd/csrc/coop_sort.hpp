// coop_sort.hpp -- the distribution sort of bin_sort.hpp run by W wavefronts of ONE workgroup on one slice
// (gfx950).  Same five steps (histogram with ds_add_rtn, scan, scatter, read back, odd-even fix-up of the equal-bin
// runs), but the counters and the staging buffer are shared by the W waves of the slice:
//
//   * a lane keeps EPT = keys/(64 W) keys instead of keys/64 -- at 2048 points and W = 4: 8 registers per array
//     instead of 32, so the kernel needs ~64 VGPRs, and LDS per WAVE is (bins + keys) * 4 / W bytes (4 KB), against
//     168 VGPRs and 12 KB for the one-wave form: the CU fills up with waves;
//   * lane gl = wave*64 + lane (0 <= gl < 64 W) owns points r*64W + gl and, after the sort, sorted positions
//     gl*EPT + r -- the layout of wave_sort / the multi-wave bitonic kernel;
//   * every wave fixes up its own 64*EPT positions; a run of equal-bin keys that straddles the seam between two
//     waves is finished by a 64-key window sort around the seam (one key per lane, the cross-lane network of
//     wave_sort.hpp): runs are at most SHW_COOP_MAX_RUN (< 32) long, so the window holds them whole;
//   * workgroup barriers separate the steps (7 per sort).
// Data with longer runs falls back to the bitonic network (in-wave sort + merge across waves through LDS).
#pragma once
#include <hip/hip_runtime.h>

#include "bin_sort.hpp"

#ifndef SHW_COOP_BINS_PER_KEY
#define SHW_COOP_BINS_PER_KEY 1      // bins = slice capacity * SHW_COOP_BINS_PER_KEY / SHW_COOP_KEYS_PER_BIN
#endif
#ifndef SHW_COOP_KEYS_PER_BIN
#define SHW_COOP_KEYS_PER_BIN 2      // (2: two keys per bin on average -- measured best, as for the one-wave sort)
#endif

namespace shw {

// KPB: keys per bin on average (the p = 1 training kernel at 16384 merged atoms takes 4 to fit the 160 KB of LDS)
template <int EPT, int W, int KPB = SHW_COOP_KEYS_PER_BIN>
struct Coop {
  static constexpr int NCOL = 64 * W;                       // lanes per slice
  static constexpr int CAP = EPT * NCOL;                    // keys per slice (padded)
  // bins per lane in the scan: EPT / KPB for the power-of-two classes; rounded DOWN to a multiple of four otherwise
  // (128-bit accesses; 20 keys per lane: 8 bins per lane, 2.5 keys per bin -- never more counters than half a row)
  static constexpr int BPL = is_pow2(EPT) ? SHW_COOP_BINS_PER_KEY * EPT / KPB : (SHW_COOP_BINS_PER_KEY * EPT / KPB) / 4 * 4;
  static constexpr int NB = BPL * NCOL;                     // bins
  static constexpr int RED = 12 * W + 16;                   // ints / floats of cross-wave scratch
  static constexpr int LDS_FLOATS = NB + CAP + RED;
  static_assert(EPT % 4 == 0 && BPL % 4 == 0, "128-bit LDS accesses need multiples of four");
  static_assert(NB <= 65536, "the bin number travels in 16 bits");
};

// address permutation of the staging buffer, generalised from binsort_addr: lane stride is EPT*4 bytes; lanes
// 128/(EPT*4) apart... the 16 lanes served together by a ds_read_b128 must hit 16 different 16-byte bank groups
template <int EPT>
__device__ __forceinline__ unsigned coop_addr(unsigned pos) {
  if constexpr (EPT >= 8) return binsort_addr<EPT>(pos);
  else return (pos << 2) ^ ((pos >> 3) & 0x10u);             // EPT = 4: one chunk per lane, rows of 8 lanes
}

// ---- cross-wave bitonic merge (fallback path; generalises the multi-wave loss kernel's merge to any EPT) -------
template <int EPT, int W>
__device__ __forceinline__ void coop_exchange(float (&key)[EPT], float* buf, int wave, int lane, int partner,
                                              bool mirror, bool upper) {
  constexpr int NCOL = 64 * W;
#pragma unroll
  for (int r = 0; r < EPT; ++r) buf[r * NCOL + wave * 64 + lane] = key[r];
  __syncthreads();
#pragma unroll
  for (int r = 0; r < EPT; ++r) {
    const float p = mirror ? buf[(EPT - 1 - r) * NCOL + partner * 64 + (63 - lane)] : buf[r * NCOL + partner * 64 + lane];
    key[r] = upper ? __builtin_fmaxf(key[r], p) : __builtin_fminf(key[r], p);
  }
  __syncthreads();
}

template <int EPT, int W>
__device__ __forceinline__ void coop_bitonic(float (&key)[EPT], float* buf, int wave, int lane) {
  wave_sort<EPT>(key, lane);
#pragma unroll
  for (int c = 1; (1 << c) <= W; ++c) {                       // merge blocks of 2^c waves
    coop_exchange<EPT, W>(key, buf, wave, lane, wave ^ ((1 << c) - 1), true, (wave & (1 << (c - 1))) != 0);
#pragma unroll
    for (int t = c - 2; t >= 0; --t)
      coop_exchange<EPT, W>(key, buf, wave, lane, wave ^ (1 << t), false, (wave & (1 << t)) != 0);
    xlane_stages<F32Keys, EPT, 32>(key, lane);
    lane_stages<F32Keys, EPT, EPT / 2>(key);
  }
}

// The fallback of a class that is not a power of two (long runs: clustered data, duplicates): a bitonic network on the
// keys where they sit in LDS.  `a` holds `cap` keys in plain order; the network runs over next_pow2(cap) positions whose
// tail is virtual +inf -- in the flip form of the network every comparator sends the larger key to the higher index, so
// the virtual keys never move and are never touched.  One barrier per stage (~80 of them at 4096 slots): slow, rare.
template <class T>
__device__ __forceinline__ void lds_bitonic_stage(T* a, int cap, int tid, int nthr, int mask) {
  for (int i = tid; i < cap; i += nthr) {
    const int j = i ^ mask;
    if (j > i && j < cap) {                                  // (j >= cap: the partner is a virtual +inf)
      const T x = a[i], y = a[j];
      if (y < x) { a[i] = y; a[j] = x; }
    }
  }
  __syncthreads();
}

template <class T>
__device__ __forceinline__ void lds_bitonic_sort(T* a, int cap, int tid, int nthr) {
  int P = 1;
  while (P < cap) P <<= 1;
  for (int k = 2; k <= P; k <<= 1) {
    lds_bitonic_stage(a, cap, tid, nthr, k - 1);             // merge of blocks of k: i against its mirror image
    for (int s = k >> 2; s >= 1; s >>= 1) lds_bitonic_stage(a, cap, tid, nthr, s);
  }
}

// Zero the counters (every wave its share).  The caller places a barrier between this and the next histogram.
template <int EPT, int W, int KPB = SHW_COOP_KEYS_PER_BIN>
__device__ __forceinline__ void coop_zero_counters(unsigned* cnt, int gl) {
  typedef Coop<EPT, W, KPB> C;
#pragma unroll
  for (int j = 0; j < C::BPL / 4; ++j)
    *reinterpret_cast<u32x4*>(cnt + j * (C::NCOL * 4) + gl * 4) = u32x4{0u, 0u, 0u, 0u};
}

// Sort the slice's keys ascending.  On entry key[r] belongs to point r*64W + gl; on return to sorted position
// gl*EPT + r (pads = +inf behind the n live keys).  cnt must be zero on entry (coop_zero_counters + barrier) and is
// left zeroed for the next sort.  Every wave of the workgroup must call this (it contains barriers).
template <int EPT, int W, bool FULL>
__device__ __forceinline__ void coop_sort(float (&key)[EPT], int wave, int lane, int n, unsigned* cnt, float* buf,
                                          int* red) {
  typedef Coop<EPT, W> C;
  const int gl = wave * 64 + lane;
  char* bytes = reinterpret_cast<char*>(buf);
  // ---- 1. histogram ------------------------------------------------------------------------------------------
  unsigned w[EPT];
  {
    unsigned b[EPT], rank[EPT];
#pragma unroll
    for (int r = 0; r < EPT; ++r) {
      const unsigned t = (unsigned)(key[r] * (float)C::NB);          // saturating convert: NaN -> 0, +inf -> max
      b[r] = t < (unsigned)(C::NB - 1) ? t : (unsigned)(C::NB - 1);
      // pads (they add 0) go to different counters: atomics on ONE address would be served one after the other
      if constexpr (!FULL) b[r] = (r * C::NCOL + gl < n) ? b[r] : (unsigned)gl;
    }
#pragma unroll
    for (int r = 0; r < EPT; ++r) {
      const unsigned inc = (FULL || (r * C::NCOL + gl < n)) ? 1u : 0u;       // pads add 0: no divergent branch
      rank[r] = __hip_atomic_fetch_add(cnt + b[r], inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
#pragma unroll
    for (int r = 0; r < EPT; ++r) w[r] = (rank[r] << 16) | b[r];
  }
  __syncthreads();
  // ---- 2. scan: lane gl owns bins [gl*BPL, (gl+1)*BPL) --------------------------------------------------------
  unsigned c[C::BPL];
#pragma unroll
  for (int j = 0; j < C::BPL / 4; ++j) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(cnt + gl * C::BPL + j * 4);
    c[4 * j] = v.x; c[4 * j + 1] = v.y; c[4 * j + 2] = v.z; c[4 * j + 3] = v.w;
  }
  unsigned run = 0, total = 0;
#pragma unroll
  for (int j = 0; j < C::BPL; ++j) {
    run = c[j] > run ? c[j] : run;
    const unsigned t = c[j];
    c[j] = total;
    total += t;
  }
  const int incl = wave_inclusive_scan_dpp((int)total);
  int gw = (int)run;
  gw = max(gw, __builtin_amdgcn_update_dpp(0, gw, 0x111, 0xf, 0xf, false));
  gw = max(gw, __builtin_amdgcn_update_dpp(0, gw, 0x112, 0xf, 0xf, false));
  gw = max(gw, __builtin_amdgcn_update_dpp(0, gw, 0x114, 0xf, 0xf, false));
  gw = max(gw, __builtin_amdgcn_update_dpp(0, gw, 0x118, 0xf, 0xf, false));
  gw = max(gw, __builtin_amdgcn_update_dpp(0, gw, 0x142, 0xa, 0xf, false));
  gw = max(gw, __builtin_amdgcn_update_dpp(0, gw, 0x143, 0xc, 0xf, false));
  if (lane == 63) { red[wave] = incl; red[W + wave] = gw; }
  __syncthreads();
  int base = 0, g = 0;
#pragma unroll
  for (int q = 0; q < W; ++q) {
    const int t = red[q], gq = red[W + q];
    base += (q < wave) ? t : 0;
    g = max(g, gq);
  }
  g = __builtin_amdgcn_readfirstlane(g);
  const unsigned off = (unsigned)(incl - (int)total + base);
#pragma unroll
  for (int j = 0; j < C::BPL / 4; ++j)
    *reinterpret_cast<u32x4*>(cnt + gl * C::BPL + j * 4) =
        u32x4{c[4 * j] + off, c[4 * j + 1] + off, c[4 * j + 2] + off, c[4 * j + 3] + off};
  __syncthreads();
  if (g > SHW_COOP_MAX_RUN) {
    // long runs (clustered data, duplicates): the network sorts it; counters re-zeroed for the next sort
    coop_zero_counters<EPT, W>(cnt, gl);
    if constexpr (is_pow2(EPT)) {
      coop_bitonic<EPT, W>(key, buf, wave, lane);
    } else {
#pragma unroll
      for (int r = 0; r < EPT; ++r) buf[r * C::NCOL + gl] = key[r];
      __syncthreads();
      lds_bitonic_sort<float>(buf, C::CAP, gl, C::NCOL);
#pragma unroll
      for (int r = 0; r < EPT; ++r) key[r] = buf[gl * EPT + r];
      __syncthreads();
    }
    return;
  }
  // ---- 3. scatter ----------------------------------------------------------------------------------------------
  {
    unsigned start[EPT];
#pragma unroll
    for (int r = 0; r < EPT; ++r) start[r] = cnt[w[r] & 0xffffu];
#pragma unroll
    for (int r = 0; r < EPT; ++r) {
      // a pad (key +inf, original index i >= n) goes to position i: behind the live keys, each once
      const unsigned i = (unsigned)(r * C::NCOL + gl);
      const unsigned pos = (FULL || (int)i < n) ? start[r] + (w[r] >> 16) : i;
      *reinterpret_cast<float*>(bytes + coop_addr<EPT>(pos)) = key[r];
    }
  }
  __syncthreads();
  coop_zero_counters<EPT, W>(cnt, gl);                       // the offsets are dead: ready for the next sort
  // ---- 4. read back EPT consecutive positions ------------------------------------------------------------------
  auto read_back = [&]() {
#pragma unroll
    for (int j = 0; j < EPT / 4; ++j) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(bytes + coop_addr<EPT>((unsigned)gl * EPT + 4u * j));
      key[4 * j] = v.x; key[4 * j + 1] = v.y; key[4 * j + 2] = v.z; key[4 * j + 3] = v.w;
    }
  };
  read_back();
  // ---- 5. fix-up inside the wave: g phases of odd-even transposition -------------------------------------------
  for (int phase = 0; phase < g; phase += 2) {
#pragma unroll
    for (int r = 0; r + 1 < EPT; r += 2) cmp_swap<F32Keys>(key[r], key[r + 1]);
    if (phase + 1 < g) {
#pragma unroll
      for (int r = 1; r + 1 < EPT; r += 2) cmp_swap<F32Keys>(key[r], key[r + 1]);
      binsort_boundary<EPT>(key, lane);
    }
  }
  // ---- 6. seams between waves ----------------------------------------------------------------------------------
  if constexpr (W > 1) {
#pragma unroll
    for (int j = 0; j < EPT / 4; ++j)
      *reinterpret_cast<f32x4*>(bytes + coop_addr<EPT>((unsigned)gl * EPT + 4u * j)) =
          f32x4{key[4 * j], key[4 * j + 1], key[4 * j + 2], key[4 * j + 3]};
    __syncthreads();
    if (wave > 0) {
      const unsigned pos = (unsigned)(wave * 64 * EPT - 32 + lane);
      float x[1] = {*reinterpret_cast<const float*>(bytes + coop_addr<EPT>(pos))};
      wave_sort<1>(x, lane);
      *reinterpret_cast<float*>(bytes + coop_addr<EPT>(pos)) = x[0];
    }
    __syncthreads();
    read_back();
  }
}

}  // namespace shw
