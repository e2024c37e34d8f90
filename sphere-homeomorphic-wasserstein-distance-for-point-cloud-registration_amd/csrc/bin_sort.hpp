// bin_sort.hpp -- one-pass distribution sort of <= 64*EPT circle coordinates by ONE wavefront (gfx950), through LDS.
//
// Why: on MI355X every v_min/v_max/v_med3/v_cmp (f32 or integer) issues at 1.78 ns per wave-instruction per SIMD,
// add/fma/xor at 1.0 ns (tools/ubench/op_rate.hip, round 2).  The register bitonic network of wave_sort.hpp costs
// 66 stages x 32 keys = 2112 slow-rate instructions + 672 crossbar moves per 2048-key sort, and two of them are
// 70 % of the loss kernel.  Circle coordinates are numbers in [0, 1] -- a key's value says where it belongs:
//
//   1. histogram   b = min(floor(key * NB), NB-1);  rank = ds_add_rtn(cnt[b], 1)        (NB = 32*EPT bins: 2 keys per
//                                                                                          bin on average)
//   2. scan        exclusive prefix sum over the NB counters (16 per lane in-lane, then a wave scan); the largest
//                  counter g is the longest run of keys that share a bin
//   3. scatter     buf[start[b] + rank] = key          (keys are now ordered by bin; inside a bin in arrival order)
//   4. read back   32 consecutive positions per lane (sorted position of x[r] in lane `lane` is lane*EPT + r, the
//                  layout wave_sort leaves)
//   5. fix-up      g phases of odd-even transposition (in-lane compare-exchanges + one exchange across each lane
//                  boundary per odd phase): every run of equal-bin keys is at most g long and no key has to leave its
//                  run, so g phases put every run -- hence the whole array -- into exact ascending order.
//
// ~8 VALU + 3 LDS instructions per key for steps 1-4 and 33 slow-rate instructions per lane per phase: with g ~ 8
// (uniformly spread coordinates) a quarter of the network's VALU work and an eighth of its crossbar traffic.
// The result is the exact ascending order of the fp32 keys whatever order the atomics were served in.
// Data with long runs (clustered clouds, duplicates, an all-zero cloud: g > SHW_BINSORT_MAX_RUN) is detected
// after step 2 -- before any key has moved -- and the caller sorts it with the network instead (wave-uniform branch).
//
// LDS per wave: NB counters (128*EPT bytes) + a 64*EPT-float staging buffer (256*EPT bytes).  LDS operations of one
// wave execute in order and nothing here is shared with another wave: no barrier.
#pragma once
#include <hip/hip_runtime.h>

#include "wave_sort.hpp"

#ifndef SHW_BINSORT_NB_PER_EPT
#define SHW_BINSORT_NB_PER_EPT 32
#endif
// Longest equal-bin run the one-wave sorts fix up by odd-even phases (one phase ~ 33 quarter-rate instructions per lane,
// ~60 ns) before they prefer the bitonic network (~4.4 us per 2048-key sort, after the histogram has been paid): round 2
// used 24, which sent 23 % of the slices of the notebooks' cube-surface clouds and 95 % of tightly clustered clouds to
// the network (profiles/r03_nonuniform.txt).  Measured break-even: ~45 phases (at 64 clouds of 32-fold duplicate
// points took 0.68 instead of 0.53 ms per launch, at 24 sixteen tight clusters 0.56 instead of 0.52).
#ifndef SHW_BINSORT_MAX_RUN
#define SHW_BINSORT_MAX_RUN 40
#endif
// the cooperative sorts (several waves per slice) finish a run that straddles two waves inside a 64-key window: < 32
#ifndef SHW_COOP_MAX_RUN
#define SHW_COOP_MAX_RUN 24
#endif

namespace shw {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// bins of the key-only sort: 32 per key slot of a lane (two keys per bin) for the power-of-two classes; otherwise the next
// count whose share per lane in the scan is a multiple of four (128-bit accesses): 768 for 20 and 24 keys per lane, 1024 for 28
template <int EPT>
constexpr int binsort_bins() {
  return is_pow2(EPT) ? SHW_BINSORT_NB_PER_EPT * EPT : 64 * ((((EPT + 1) / 2) + 3) / 4 * 4);
}

// byte address of sorted position `pos` in the staging buffer: rows of 32 floats (one lane's read-back), the eight
// 16-byte chunks of a row XOR-permuted by bits 1..3 of the row number so that the 16 lanes a ds_read_b128 serves
// together touch 16 different bank groups (lane stride is 128 B: without the permutation 8 lanes share each group)
template <int EPT>
__device__ __forceinline__ unsigned binsort_addr(unsigned pos) {
  // a row is EPT floats = EPT/4 chunks; lanes 2^k apart (k = 1 at EPT 32, 2 at 16, 3 at 8) share bank groups, and
  // bit 6 of pos is bit k of the row number in every size class: (pos >> 2) puts it on the chunk-index bits
  return (pos << 2) ^ ((pos >> 2) & (unsigned)((EPT / 4 - 1) << 4));
}

// inclusive prefix sum over the lanes of a wave (row_shr DPP steps inside rows of 16, then the two row broadcasts)
__device__ __forceinline__ int wave_inclusive_scan_dpp(int v) {
  // classic GCN scan: row_shr:1,2,3 / 4 / 8 then row_bcast:15 and row_bcast:31 with the matching row masks
  int t;
  t = __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false); v += t;            // row_shr:1
  t = __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false); v += t;            // row_shr:2
  t = __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false); v += t;            // row_shr:4
  t = __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false); v += t;            // row_shr:8
  t = __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false); v += t;            // row_bcast:15 -> rows 1, 3
  t = __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false); v += t;            // row_bcast:31 -> rows 2, 3
  return v;
}

// Steps 1-2.  Returns the longest equal-bin run g (wave-uniform); on return cnt[] holds the exclusive prefix sums and
// w[r] = (rank << 16) | bin for live keys.  `n` = number of live keys; key[r] belongs to point r*64 + lane.
template <int EPT, bool FULL>
__device__ __forceinline__ int binsort_histogram(const float (&key)[EPT], unsigned (&w)[EPT], int lane, int n,
                                                 unsigned* cnt) {
  constexpr int NB = binsort_bins<EPT>();
  constexpr int BPL = NB / 64;                          // bins per lane in the scan (EPT/2: 16 at EPT = 32)
  static_assert(BPL >= 4 && BPL % 4 == 0, "bin sort needs >= 4 bins per lane");
  // zero the counters: every ds_write_b128 covers 1 KB of consecutive addresses (conflict-free)
#pragma unroll
  for (int j = 0; j < BPL / 4; ++j)
    *reinterpret_cast<u32x4*>(cnt + j * 256 + lane * 4) = u32x4{0u, 0u, 0u, 0u};
  __builtin_amdgcn_wave_barrier();
  constexpr int CH = chunk_of(EPT);                      // atomics in flight per lane; bounds the live registers
#pragma unroll
  for (int r0 = 0; r0 < EPT; r0 += CH) {
    // key[r] belongs to point r*64 + lane: the pads of a class that is not full are its LAST rows -- a chunk of rows is
    // entirely live (no masks: the code of a full class), entirely pads (nothing to count) or the one mixed chunk;
    // which of the three is uniform over the wave
    // (measured, N = 1200 / 2000 at B = 64, L = 512: the branches pay in the classes of 12 .. 28 keys per lane -- the loads
    //  are no longer hoisted together, 150 -> 111 VGPRs, a fourth wave per SIMD -- and cost 6 % at 32, where LDS holds the
    //  occupancy at three: power-of-two classes keep the masked form)
    constexpr bool kByChunk = !is_pow2(EPT);
    const bool all_live = FULL || (kByChunk && (r0 + CH) * kWave <= n);
    const bool none_live = !FULL && kByChunk && r0 * kWave >= n;
    if (none_live) {
#pragma unroll
      for (int j = 0; j < CH; ++j) w[r0 + j] = 0u;
    } else if (all_live) {
      unsigned b[CH], rank[CH];
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        // v_cvt_u32_f32 saturates (NaN -> 0, +inf -> 0xffffffff): the bin is always inside [0, NB)
        const unsigned t = (unsigned)(key[r0 + j] * (float)NB);
        b[j] = t < (unsigned)(NB - 1) ? t : (unsigned)(NB - 1);
      }
#pragma unroll
      for (int j = 0; j < CH; ++j)
        rank[j] = __hip_atomic_fetch_add(cnt + b[j], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
#pragma unroll
      for (int j = 0; j < CH; ++j) w[r0 + j] = (rank[j] << 16) | b[j];
    } else {
      unsigned b[CH], rank[CH];
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        const unsigned t = (unsigned)(key[r0 + j] * (float)NB);
        b[j] = t < (unsigned)(NB - 1) ? t : (unsigned)(NB - 1);
        // pads (they add 0) go to 64 different counters: 64 atomics on ONE address would be served one after the other
        b[j] = ((r0 + j) * kWave + lane < n) ? b[j] : (unsigned)lane;
      }
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        // pads add 0: no divergent branch around the atomic
        const unsigned inc = ((r0 + j) * kWave + lane < n) ? 1u : 0u;
        rank[j] = __hip_atomic_fetch_add(cnt + b[j], inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      }
#pragma unroll
      for (int j = 0; j < CH; ++j) w[r0 + j] = (rank[j] << 16) | b[j];
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  __builtin_amdgcn_wave_barrier();
  // scan: lane owns bins [lane*BPL, (lane+1)*BPL)
  unsigned c[BPL];
#pragma unroll
  for (int j = 0; j < BPL / 4; ++j) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(cnt + lane * BPL + j * 4);
    c[4 * j] = v.x; c[4 * j + 1] = v.y; c[4 * j + 2] = v.z; c[4 * j + 3] = v.w;
  }
  unsigned run = 0, total = 0;
#pragma unroll
  for (int j = 0; j < BPL; ++j) {
    run = c[j] > run ? c[j] : run;
    const unsigned t = c[j];
    c[j] = total;                                        // exclusive inside the lane
    total += t;
  }
  const int incl = wave_inclusive_scan_dpp((int)total);
  const unsigned base = (unsigned)incl - total;
#pragma unroll
  for (int j = 0; j < BPL / 4; ++j)
    *reinterpret_cast<u32x4*>(cnt + lane * BPL + j * 4) =
        u32x4{c[4 * j] + base, c[4 * j + 1] + base, c[4 * j + 2] + base, c[4 * j + 3] + base};
  __builtin_amdgcn_wave_barrier();
  // wave maximum of the run lengths
  int g = (int)run;
  g = max(g, __builtin_amdgcn_update_dpp(0, g, 0x111, 0xf, 0xf, false));
  g = max(g, __builtin_amdgcn_update_dpp(0, g, 0x112, 0xf, 0xf, false));
  g = max(g, __builtin_amdgcn_update_dpp(0, g, 0x114, 0xf, 0xf, false));
  g = max(g, __builtin_amdgcn_update_dpp(0, g, 0x118, 0xf, 0xf, false));
  g = max(g, __builtin_amdgcn_update_dpp(0, g, 0x142, 0xa, 0xf, false));
  g = max(g, __builtin_amdgcn_update_dpp(0, g, 0x143, 0xc, 0xf, false));
  return __builtin_amdgcn_readlane(g, 63);
}

// one compare-exchange between the last key of every lane and the first key of the next lane
template <int EPT>
__device__ __forceinline__ void binsort_boundary(float (&x)[EPT], int lane) {
  const float nxt = as_f(__builtin_amdgcn_ds_bpermute(min(lane + 1, 63) << 2, as_i(x[0])));
  const float prv = as_f(__builtin_amdgcn_ds_bpermute(max(lane - 1, 0) << 2, as_i(x[EPT - 1])));
  const float hi = __builtin_fminf(x[EPT - 1], lane < 63 ? nxt : __builtin_inff());
  const float lo = __builtin_fmaxf(x[0], lane > 0 ? prv : -__builtin_inff());
  x[EPT - 1] = hi;
  x[0] = lo;
}

// Steps 3-5 (only after binsort_histogram returned g <= SHW_BINSORT_MAX_RUN).
template <int EPT, bool FULL>
__device__ __forceinline__ void binsort_place(float (&key)[EPT], const unsigned (&w)[EPT], int lane, int n, int g,
                                              const unsigned* cnt, float* buf) {
  char* bytes = reinterpret_cast<char*>(buf);
  constexpr int CH = chunk_of(EPT);
#pragma unroll
  for (int r0 = 0; r0 < EPT; r0 += CH) {
    constexpr bool kByChunk = !is_pow2(EPT);                       // (see binsort_histogram)
    const bool all_live = FULL || (kByChunk && (r0 + CH) * kWave <= n);
    const bool none_live = !FULL && kByChunk && r0 * kWave >= n;
    if (none_live) {
      // pads (key +inf, original index i >= n) go to position i: the positions behind the n live keys, each once
#pragma unroll
      for (int j = 0; j < CH; ++j)
        *reinterpret_cast<float*>(bytes + binsort_addr<EPT>((unsigned)((r0 + j) * kWave + lane))) = key[r0 + j];
    } else {
      unsigned start[CH];
#pragma unroll
      for (int j = 0; j < CH; ++j) start[j] = cnt[w[r0 + j] & 0xffffu];
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        const unsigned i = (unsigned)((r0 + j) * kWave + lane);
        const unsigned pos = (all_live || (int)i < n) ? start[j] + (w[r0 + j] >> 16) : i;
        *reinterpret_cast<float*>(bytes + binsort_addr<EPT>(pos)) = key[r0 + j];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int j = 0; j < EPT / 4; ++j) {
    const unsigned pos0 = (unsigned)lane * EPT + 4u * j;          // logical chunk j of row `lane`
    const f32x4 v = *reinterpret_cast<const f32x4*>(bytes + binsort_addr<EPT>(pos0));
    key[4 * j] = v.x; key[4 * j + 1] = v.y; key[4 * j + 2] = v.z; key[4 * j + 3] = v.w;
  }
  // odd-even transposition, g phases (wave-uniform trip count)
  for (int phase = 0; phase < g; phase += 2) {
#pragma unroll
    for (int r = 0; r + 1 < EPT; r += 2) cmp_swap<F32Keys>(key[r], key[r + 1]);
    if (phase + 1 < g) {
#pragma unroll
      for (int r = 1; r + 1 < EPT; r += 2) cmp_swap<F32Keys>(key[r], key[r + 1]);
      binsort_boundary<EPT>(key, lane);
    }
  }
  __builtin_amdgcn_wave_barrier();
}

// The bitonic network for a class that is not a power of two (the rare fallback of such a class): sort next_pow2(EPT) keys
// per lane with +inf / all-ones behind the real ones, then take the 64 EPT smallest through the staging buffer into the
// layout "sorted position lane*EPT + r".  MAXKEY sorts behind every real key.
template <int EPT, class T>
__device__ __forceinline__ void wave_sort_relayout(T (&key)[EPT], int lane, T maxkey, void* buf) {
  constexpr int P2 = next_pow2_c(EPT);
  T tmp[P2];
#pragma unroll
  for (int r = 0; r < P2; ++r) tmp[r] = r < EPT ? key[r < EPT ? r : 0] : maxkey;
  wave_sort<P2>(tmp, lane);
  char* bytes = reinterpret_cast<char*>(buf);
#pragma unroll
  for (int r = 0; r < P2; ++r) {
    const unsigned pos = (unsigned)(lane * P2 + r);
    if (pos < (unsigned)(kWave * EPT)) *reinterpret_cast<T*>(bytes + binsort_addr<EPT>(pos)) = tmp[r];
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int r = 0; r < EPT; ++r) key[r] = *reinterpret_cast<const T*>(bytes + binsort_addr<EPT>((unsigned)(lane * EPT + r)));
  __builtin_amdgcn_wave_barrier();
}

// Sort the 64*EPT keys of a wave ascending (pads = +inf behind the n live keys).  Falls back to the bitonic network
// when the data has runs longer than SHW_BINSORT_MAX_RUN.  scratch: 32*EPT counters followed by 64*EPT floats.
// Returns the longest run of keys that share a bin (> SHW_BINSORT_MAX_RUN: the slice took the network).
template <int EPT, bool FULL>
__device__ __forceinline__ int wave_sort_binned(float (&key)[EPT], int lane, int n, float* counters, float* buf) {
  unsigned* cnt = reinterpret_cast<unsigned*>(counters);
  unsigned w[EPT];
  const int g = binsort_histogram<EPT, FULL>(key, w, lane, n, cnt);
  if (g <= SHW_BINSORT_MAX_RUN) {
    binsort_place<EPT, FULL>(key, w, lane, n, g, cnt, buf);
  } else {
#ifndef SHW_ABL_NO_FALLBACK
    if constexpr (is_pow2(EPT)) wave_sort<EPT>(key, lane);
    else wave_sort_relayout<EPT, float>(key, lane, __builtin_inff(), buf);
#endif
  }
  return g;
}

// counters and staging buffer contiguous: binsort_bins<EPT>() counters followed by 64*EPT floats
template <int EPT, bool FULL>
__device__ __forceinline__ int wave_sort_binned(float (&key)[EPT], int lane, int n, float* scratch) {
  return wave_sort_binned<EPT, FULL>(key, lane, n, scratch, scratch + binsort_bins<EPT>());
}

}  // namespace shw
