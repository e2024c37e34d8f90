// coop_sort_kv.hpp -- the cooperative distribution sort of coop_sort.hpp on 64-bit (coordinate bits << 32 | original
// index) items: the sort WITH the permutation for more than 2048 points (training kernel of shw_ssw_grad_coop.hip).
//
// Why items and not the packed 32-bit words of the one- and two-wave training kernels: at 8192 points a word has 13
// index bits and 19 coordinate bits -- ~1.6 % of adjacent pairs share a quantised coordinate (hundreds per slice, with
// chains of three) and the repair of those would have to cross wave seams.  An item carries the exact coordinate, the
// index makes it unique, and unsigned 64-bit order IS the stable ascending order torch.sort gives the reference
// (:163-164): nothing to repair.  Price: 8 bytes per slot in the staging buffer and one v_cmp_u64 + four v_cndmask
// per compare-exchange of the fix-up instead of a v_min / v_max pair.
//
// Same steps and layout as coop_sort: lane gl = wave*64 + lane owns points r*64W + gl on entry and sorted positions
// gl*EPT + r on return; an item at position p sits at the 8 bytes of 4-byte slots 2p, 2p+1 of coop_addr<2 EPT>
// (the same XOR permutation of 16-byte chunks: a ds_read_b128 returns two items).
#pragma once
#include "coop_sort.hpp"

namespace shw {

// ---- cross-wave bitonic merge on items (fallback path) ---------------------------------------------------------
template <int EPT, int W>
__device__ __forceinline__ void coop_exchange_kv(item_t (&it)[EPT], item_t* buf, int wave, int lane, int partner,
                                                 bool mirror, bool upper) {
  constexpr int NCOL = 64 * W;
#pragma unroll
  for (int r = 0; r < EPT; ++r) buf[r * NCOL + wave * 64 + lane] = it[r];
  __syncthreads();
#pragma unroll
  for (int r = 0; r < EPT; ++r) {
    const item_t p = mirror ? buf[(EPT - 1 - r) * NCOL + partner * 64 + (63 - lane)] : buf[r * NCOL + partner * 64 + lane];
    it[r] = upper ? U64Items::hi(it[r], p) : U64Items::lo(it[r], p);
  }
  __syncthreads();
}

template <int EPT, int W>
__device__ __forceinline__ void coop_bitonic_kv(item_t (&it)[EPT], item_t* buf, int wave, int lane) {
  wave_sort_kv<EPT>(it, lane);
#pragma unroll
  for (int c = 1; (1 << c) <= W; ++c) {                       // merge blocks of 2^c waves
    coop_exchange_kv<EPT, W>(it, buf, wave, lane, wave ^ ((1 << c) - 1), true, (wave & (1 << (c - 1))) != 0);
#pragma unroll
    for (int t = c - 2; t >= 0; --t)
      coop_exchange_kv<EPT, W>(it, buf, wave, lane, wave ^ (1 << t), false, (wave & (1 << t)) != 0);
    xlane_stages<U64Items, EPT, 32>(it, lane);
    lane_stages<U64Items, EPT, EPT / 2>(it);
  }
}

template <int EPT>
__device__ __forceinline__ void binsort_boundary_kv(item_t (&x)[EPT], int lane) {
  const unsigned a_lo = (unsigned)x[0], a_hi = (unsigned)(x[0] >> 32);
  const unsigned z_lo = (unsigned)x[EPT - 1], z_hi = (unsigned)(x[EPT - 1] >> 32);
  const unsigned n_lo = (unsigned)__builtin_amdgcn_ds_bpermute(min(lane + 1, 63) << 2, (int)a_lo);
  const unsigned n_hi = (unsigned)__builtin_amdgcn_ds_bpermute(min(lane + 1, 63) << 2, (int)a_hi);
  const unsigned p_lo = (unsigned)__builtin_amdgcn_ds_bpermute(max(lane - 1, 0) << 2, (int)z_lo);
  const unsigned p_hi = (unsigned)__builtin_amdgcn_ds_bpermute(max(lane - 1, 0) << 2, (int)z_hi);
  const item_t nxt = lane < 63 ? (((item_t)n_hi << 32) | n_lo) : ~0ull;
  const item_t prv = lane > 0 ? (((item_t)p_hi << 32) | p_lo) : 0ull;
  x[EPT - 1] = U64Items::lo(x[EPT - 1], nxt);
  x[0] = U64Items::hi(x[0], prv);
}

// Sort the slice's items ascending.  On entry it[r] = make_item(coordinate, r*64W + gl) (pads: +inf, any index >= n);
// on return it[r] is the item at sorted position gl*EPT + r.  cnt must be zero on entry and is left zeroed.
// buf: 64 W EPT items.  Every wave of the workgroup must call this (it contains barriers).
template <int EPT, int W, bool FULL, int KPB = SHW_COOP_KEYS_PER_BIN>
__device__ __forceinline__ void coop_sort_kv(item_t (&it)[EPT], int wave, int lane, int n, unsigned* cnt, item_t* buf,
                                             int* red) {
  typedef Coop<EPT, W, KPB> C;
  const int gl = wave * 64 + lane;
  char* bytes = reinterpret_cast<char*>(buf);
  // ---- 1. histogram ------------------------------------------------------------------------------------------
  unsigned w[EPT];
  {
    unsigned b[EPT], rank[EPT];
#pragma unroll
    for (int r = 0; r < EPT; ++r) {
      const unsigned t = (unsigned)(item_key(it[r]) * (float)C::NB);   // saturating convert: NaN -> 0, +inf -> max
      b[r] = t < (unsigned)(C::NB - 1) ? t : (unsigned)(C::NB - 1);
      if constexpr (!FULL) b[r] = (r * C::NCOL + gl < n) ? b[r] : (unsigned)gl;   // pads: see coop_sort
    }
#pragma unroll
    for (int r = 0; r < EPT; ++r) {
      const unsigned inc = (FULL || (r * C::NCOL + gl < n)) ? 1u : 0u;
      rank[r] = __hip_atomic_fetch_add(cnt + b[r], inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
#pragma unroll
    for (int r = 0; r < EPT; ++r) w[r] = (rank[r] << 16) | b[r];
  }
  __syncthreads();
  // ---- 2. scan: lane gl owns bins [gl*BPL, (gl+1)*BPL) --------------------------------------------------------
  int g = 0;
  {
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
    int base = 0;
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
  }
  __syncthreads();
  if (g > SHW_COOP_MAX_RUN) {
    // long runs (clustered data, duplicates): the network sorts it; counters re-zeroed for the next sort
    coop_zero_counters<EPT, W, KPB>(cnt, gl);
    if constexpr (is_pow2(EPT)) {
      coop_bitonic_kv<EPT, W>(it, buf, wave, lane);
    } else {                                                 // (see lds_bitonic_sort; items are unique: plain <)
#pragma unroll
      for (int r = 0; r < EPT; ++r) buf[r * C::NCOL + gl] = it[r];
      __syncthreads();
      lds_bitonic_sort<item_t>(buf, C::CAP, gl, C::NCOL);
#pragma unroll
      for (int r = 0; r < EPT; ++r) it[r] = buf[gl * EPT + r];
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
      const unsigned i = (unsigned)(r * C::NCOL + gl);       // a pad goes to position i: behind the live items, each once
      const unsigned pos = (FULL || (int)i < n) ? start[r] + (w[r] >> 16) : i;
      *reinterpret_cast<item_t*>(bytes + coop_addr<2 * EPT>(2u * pos)) = it[r];
    }
  }
  __syncthreads();
  coop_zero_counters<EPT, W, KPB>(cnt, gl);                       // the offsets are dead: ready for the next sort
  // ---- 4. read back EPT consecutive positions ------------------------------------------------------------------
  auto read_back = [&]() {
#pragma unroll
    for (int j = 0; j < EPT / 2; ++j) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(bytes + coop_addr<2 * EPT>((unsigned)gl * (2 * EPT) + 4u * j));
      it[2 * j] = ((item_t)v.y << 32) | v.x;
      it[2 * j + 1] = ((item_t)v.w << 32) | v.z;
    }
  };
  read_back();
  // ---- 5. fix-up inside the wave: g phases of odd-even transposition -------------------------------------------
  for (int phase = 0; phase < g; phase += 2) {
#pragma unroll
    for (int r = 0; r + 1 < EPT; r += 2) cmp_swap<U64Items>(it[r], it[r + 1]);
    if (phase + 1 < g) {
#pragma unroll
      for (int r = 1; r + 1 < EPT; r += 2) cmp_swap<U64Items>(it[r], it[r + 1]);
      binsort_boundary_kv<EPT>(it, lane);
    }
  }
  // ---- 6. seams between waves ----------------------------------------------------------------------------------
  if constexpr (W > 1) {
#pragma unroll
    for (int j = 0; j < EPT / 2; ++j)
      *reinterpret_cast<u32x4*>(bytes + coop_addr<2 * EPT>((unsigned)gl * (2 * EPT) + 4u * j)) =
          u32x4{(unsigned)it[2 * j], (unsigned)(it[2 * j] >> 32), (unsigned)it[2 * j + 1], (unsigned)(it[2 * j + 1] >> 32)};
    __syncthreads();
    if (wave > 0) {
      const unsigned pos = (unsigned)(wave * 64 * EPT - 32 + lane);
      item_t x[1] = {*reinterpret_cast<const item_t*>(bytes + coop_addr<2 * EPT>(2u * pos))};
      wave_sort_kv<1>(x, lane);
      *reinterpret_cast<item_t*>(bytes + coop_addr<2 * EPT>(2u * pos)) = x[0];
    }
    __syncthreads();
    read_back();
  }
}

}  // namespace shw
