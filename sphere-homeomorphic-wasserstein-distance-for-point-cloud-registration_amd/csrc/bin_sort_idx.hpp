// bin_sort_idx.hpp -- sort WITH the permutation by the distribution sort of bin_sort.hpp (training kernels).
//
// Same packed words as the network version (ssw_common.hpp: floor(coord * 2^QBITS) << IDX_BITS | original index):
// their top bits ARE the bin number, so the histogram needs no multiply; the words are scattered by bin, read back
// EPT consecutive positions per lane and put into exact packed order by the odd-even fix-up on unsigned words (one
// v_min_u32 / v_max_u32 pair per compare-exchange, as cheap as the float form).  The staging buffer then takes the
// exact coordinates by ORIGINAL index (they wait in registers while the words travel), and unpack_sorted_words
// gathers them and repairs the few pairs whose quantised coordinates collide -- the result is the stable ascending
// order torch.sort gives the reference (:163-164).
// LDS per wave: 32*EPT counters + 64*EPT words; nothing is needed beside them (the network version keeps a
// 64*EPT-float row of coordinates by original index: the same 4 bytes per atom).
#pragma once
#include "bin_sort.hpp"
#include "ssw_common.hpp"

namespace shw {

// bins of the sort with indices: the top bits of a packed word ARE its bin, so the count is a power of two -- 32 per key
// slot of a lane for the power-of-two classes, the next power of two above that otherwise (1024 for 20 .. 28 keys per lane)
template <int EPT>
constexpr int binsort_idx_bins() { return next_pow2_c(SHW_BINSORT_NB_PER_EPT * EPT); }

template <int EPT>
__device__ __forceinline__ void binsort_boundary_u32(unsigned (&x)[EPT], int lane) {
  const unsigned nxt = (unsigned)__builtin_amdgcn_ds_bpermute(min(lane + 1, 63) << 2, (int)x[0]);
  const unsigned prv = (unsigned)__builtin_amdgcn_ds_bpermute(max(lane - 1, 0) << 2, (int)x[EPT - 1]);
  const unsigned up = lane < 63 ? nxt : 0xffffffffu;
  const unsigned dn = lane > 0 ? prv : 0u;
  x[EPT - 1] = x[EPT - 1] < up ? x[EPT - 1] : up;
  x[0] = x[0] > dn ? x[0] : dn;
}

// project one cloud, sort it with its permutation; returns the coordinate sum of the lane.
//   cnt : binsort_idx_bins<EPT>() counters (32*EPT for the power-of-two classes), buf : 64*EPT words (scatter target, then coordinates by original index -- still holding
//   them on return, pads +inf).
// CHAINED: see load_coords -- needed wherever this is not inside a loop (the compiler otherwise hoists all 3*EPT point
// loads to the top and the raw points take 96 registers).
template <int EPT, bool CHAINED = true, bool FULL = false>
__device__ __forceinline__ float sorted_with_indices_binned(const float* __restrict__ X, int count, int lane,
                                                            const float (&U)[6], unsigned* cnt, float* buf,
                                                            float (&val)[EPT], int (&idx)[EPT]) {
  typedef Packing<EPT> PK;
  constexpr int NB = binsort_idx_bins<EPT>();
  constexpr int BPL = NB / 64;
  constexpr int BIN_SHIFT = 32 - __builtin_ctz(NB);
  static_assert(BPL >= 4 && BPL % 4 == 0, "bin sort needs >= 4 bins per lane");
  float key[EPT];
  unsigned pk[EPT];
  float part = load_coords<EPT, FULL, CHAINED, kWave, true>(X, count, lane, U, key);
  asm volatile("" : "+v"(part));                           // see sorted_with_indices
#pragma unroll
  for (int r = 0; r < EPT; ++r) pk[r] = PK::pack(key[r], r * kWave + lane, FULL || r * kWave + lane < count);

  // ---- histogram of the top bits; ranks travel four to a register (runs longer than 255 never get past the check)
#pragma unroll
  for (int j = 0; j < BPL / 4; ++j) *reinterpret_cast<u32x4*>(cnt + j * 256 + lane * 4) = u32x4{0u, 0u, 0u, 0u};
  __builtin_amdgcn_wave_barrier();
  unsigned rk[(EPT + 3) / 4];
#pragma unroll
  for (int q = 0; q < (EPT + 3) / 4; ++q) rk[q] = 0u;
  constexpr int CH = chunk_of(EPT);
#pragma unroll
  for (int r0 = 0; r0 < EPT; r0 += CH) {
    unsigned rank[CH];
    // (uniform over the wave: every row of the chunk is live -- the code of a full class -- or the one mixed / pad chunk)
    if (FULL || (!is_pow2(EPT) && (r0 + CH) * kWave <= count)) {
#pragma unroll
      for (int j = 0; j < CH; ++j)
        rank[j] = __hip_atomic_fetch_add(cnt + (pk[r0 + j] >> BIN_SHIFT), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    } else {
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        // pads add 0 (no divergent branch), each lane to its own counter (64 atomics on one address would serialise)
        const bool live = (r0 + j) * kWave + lane < count;
        const unsigned bin = live ? (pk[r0 + j] >> BIN_SHIFT) : (unsigned)lane;
        rank[j] = __hip_atomic_fetch_add(cnt + bin, live ? 1u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      }
    }
#pragma unroll
    for (int j = 0; j < CH; ++j) rk[(r0 + j) / 4] |= (rank[j] & 0xffu) << (8 * ((r0 + j) & 3));
    __builtin_amdgcn_sched_barrier(0);
  }
  __builtin_amdgcn_wave_barrier();
  // ---- scan (as binsort_histogram)
  int g;
  {
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
      c[j] = total;
      total += t;
    }
    const int incl = wave_inclusive_scan_dpp((int)total);
    const unsigned base = (unsigned)incl - total;
#pragma unroll
    for (int j = 0; j < BPL / 4; ++j)
      *reinterpret_cast<u32x4*>(cnt + lane * BPL + j * 4) =
          u32x4{c[4 * j] + base, c[4 * j + 1] + base, c[4 * j + 2] + base, c[4 * j + 3] + base};
    int gw = (int)run;
    gw = max(gw, __builtin_amdgcn_update_dpp(0, gw, 0x111, 0xf, 0xf, false));
    gw = max(gw, __builtin_amdgcn_update_dpp(0, gw, 0x112, 0xf, 0xf, false));
    gw = max(gw, __builtin_amdgcn_update_dpp(0, gw, 0x114, 0xf, 0xf, false));
    gw = max(gw, __builtin_amdgcn_update_dpp(0, gw, 0x118, 0xf, 0xf, false));
    gw = max(gw, __builtin_amdgcn_update_dpp(0, gw, 0x142, 0xa, 0xf, false));
    gw = max(gw, __builtin_amdgcn_update_dpp(0, gw, 0x143, 0xc, 0xf, false));
    g = __builtin_amdgcn_readlane(gw, 63);
  }
  __builtin_amdgcn_wave_barrier();
  char* bytes = reinterpret_cast<char*>(buf);
  if (g <= SHW_BINSORT_MAX_RUN) {
    // ---- scatter the words, read back EPT consecutive positions, fix up
#pragma unroll
    for (int r0 = 0; r0 < EPT; r0 += CH) {
      unsigned start[CH];
#pragma unroll
      for (int j = 0; j < CH; ++j) start[j] = cnt[pk[r0 + j] >> BIN_SHIFT];
      const bool whole = FULL || (!is_pow2(EPT) && (r0 + CH) * kWave <= count);       // (uniform over the wave)
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        // a pad (word 0xffffffff, original index i >= count) goes to position i: behind the live words, each once
        const unsigned i = (unsigned)((r0 + j) * kWave + lane);
        const unsigned pos = (whole || (int)i < count) ? start[j] + ((rk[(r0 + j) / 4] >> (8 * ((r0 + j) & 3))) & 0xffu) : i;
        *reinterpret_cast<unsigned*>(bytes + binsort_addr<EPT>(pos)) = pk[r0 + j];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < EPT / 4; ++j) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(bytes + binsort_addr<EPT>((unsigned)lane * EPT + 4u * j));
      pk[4 * j] = v.x; pk[4 * j + 1] = v.y; pk[4 * j + 2] = v.z; pk[4 * j + 3] = v.w;
    }
    __builtin_amdgcn_wave_barrier();
    // the buffer is free again: exact coordinates by ORIGINAL index (LDS operations of a wave execute in order)
#pragma unroll
    for (int r = 0; r < EPT; ++r) buf[r * kWave + lane] = key[r];
    for (int phase = 0; phase < g; phase += 2) {
#pragma unroll
      for (int r = 0; r + 1 < EPT; r += 2) cmp_swap<U32Keys>(pk[r], pk[r + 1]);
      if (phase + 1 < g) {
#pragma unroll
        for (int r = 1; r + 1 < EPT; r += 2) cmp_swap<U32Keys>(pk[r], pk[r + 1]);
        binsort_boundary_u32<EPT>(pk, lane);
      }
    }
  } else {
#ifndef SHW_ABL_NO_FALLBACK
    // (a class that is not a power of two sorts through the buffer: before the coordinates move in)
    if constexpr (!is_pow2(EPT)) wave_sort_relayout<EPT, unsigned>(pk, lane, 0xffffffffu, buf);
#endif
#pragma unroll
    for (int r = 0; r < EPT; ++r) buf[r * kWave + lane] = key[r];
#ifndef SHW_ABL_NO_FALLBACK
    if constexpr (is_pow2(EPT)) wave_sort<EPT>(pk, lane);
#endif
  }
  __builtin_amdgcn_wave_barrier();
  unpack_sorted_words<EPT, FULL>(pk, buf, count, lane, val, idx);
  return part;
}

}  // namespace shw
