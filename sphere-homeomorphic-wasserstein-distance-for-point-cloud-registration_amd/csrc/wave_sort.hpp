// wave_sort.hpp -- register-resident bitonic sort of 64*EPT floats by ONE 64-lane wavefront (gfx950).
//
// Element index e = lane*EPT + r (r = register index, the LOW bits), so that of the
// log2(64*EPT)*(log2(64*EPT)+1)/2 compare-exchange stages only the 21 whose stride reaches across
// lanes need a cross-lane move; every other stage is a v_min/v_max pair on two registers of the
// same lane.  The network is the "flip" form of bitonic sort (first stage of every merge pairs e
// with its mirror image inside the block, later stages pair e with e^stride): every
// compare-exchange then sends the smaller key to the smaller index, so a cross-lane stage is
//     x = med3(x, partner, lower ? -inf : +inf)        (one v_med3_f32)
// with `lower` a per-lane constant of the stage.
//
// Cross-lane moves use the cheapest gfx950 mechanism per lane-xor mask:
//   1, 2, 3 (quad_perm), 7 (row_half_mirror), 15 (row_mirror)  -> DPP v_mov  (VALU, no LDS port)
//   4, 8, 16, 31                                                -> ds_swizzle (LDS crossbar, no memory)
//   32, 63                                                      -> ds_bpermute
// No LDS memory and no barrier is used: a wave sorts on its own.
#pragma once
#include <hip/hip_runtime.h>

namespace shw {

constexpr int kWave = 64;

// Keys per lane need not be a power of two (round 3: classes of 12, 20, 24, 28 keys per lane, so that a 1200-point cloud
// pays for 1280 slots and not 2048).  Compile-time helpers for the code that is generic in the class:
constexpr bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
constexpr int next_pow2_c(int v) { int r = 1; while (r < v) r <<= 1; return r; }
constexpr int log2_ceil_c(int v) { int r = 0; while ((1 << r) < v) ++r; return r; }
// largest divisor of `count` that is <= most: unrolled loops work in chunks of it (8 for the power-of-two classes)
constexpr int chunk_of(int count, int most = 8) {
  int best = 1;
  for (int d = 1; d <= most; ++d) best = (count % d == 0) ? d : best;
  return best;
}

__device__ __forceinline__ float as_f(int v) { return __builtin_bit_cast(float, v); }
__device__ __forceinline__ int as_i(float v) { return __builtin_bit_cast(int, v); }

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
  // mov_dpp leaves the "old" operand undefined: with row_mask = bank_mask = 0xf and a permutation
  // that stays inside its row every lane is written, and the compiler emits no v_mov to seed it.
  return as_f(__builtin_amdgcn_mov_dpp(as_i(x), CTRL, 0xf, 0xf, true));
}

// value of `x` held by lane (lane ^ MASK); every lane of the wave must be active.
#ifndef SHW_XLANE_BATCH
#define SHW_XLANE_BATCH 8
#endif
#ifndef SHW_XLANE_DPP
#define SHW_XLANE_DPP 0   // 0: every cross-lane move on the LDS crossbar (ds_swizzle / ds_bpermute);
#endif                    // 1: DPP v_mov for the masks DPP can express.  Measured on MI355X: the kernels
                          // are VALU-issue bound (SQ_ACTIVE_INST_VALU ~ 100 %) with the LDS pipe < 10 %
                          // busy, so moving the 13 DPP stages to ds_swizzle removes ~9 % of the VALU work.
template <int MASK>
__device__ __forceinline__ float lane_xor(float x, int lane) {
#ifdef SHW_ABL_XLANE_NOSWZ      // developer ablation: no LDS crossbar traffic (results are wrong)
  { float y = x; asm volatile("v_mov_b32 %0, %1" : "=v"(y) : "v"(x)); return y; }
#endif
  if constexpr (SHW_XLANE_DPP == 0 && MASK < 32) {
    return as_f(__builtin_amdgcn_ds_swizzle(as_i(x), (MASK << 10) | 0x1F));
  } else if constexpr (MASK == 1) {
    return dpp_mov<0xB1>(x);                       // quad_perm:[1,0,3,2]
  } else if constexpr (MASK == 2) {
    return dpp_mov<0x4E>(x);                       // quad_perm:[2,3,0,1]
  } else if constexpr (MASK == 3) {
    return dpp_mov<0x1B>(x);                       // quad_perm:[3,2,1,0]
  } else if constexpr (MASK == 7) {
    return dpp_mov<0x141>(x);                      // row_half_mirror
  } else if constexpr (MASK == 15) {
    return dpp_mov<0x140>(x);                      // row_mirror
  } else if constexpr (MASK < 32) {
    // ds_swizzle bit-mask mode: lane' = ((lane & and) | or) ^ xor inside each group of 32
    return as_f(__builtin_amdgcn_ds_swizzle(as_i(x), (MASK << 10) | 0x1F));
  } else {
    return as_f(__builtin_amdgcn_ds_bpermute((lane ^ MASK) << 2, as_i(x)));
  }
}

// ---- key policies: what a compare-exchange is for a 32-bit key type ------------------------------
struct F32Keys {                      // float keys (circle coordinates, +inf padding)
  typedef float type;
  static __device__ __forceinline__ float lo(float a, float b) { return __builtin_fminf(a, b); }
  static __device__ __forceinline__ float hi(float a, float b) { return __builtin_fmaxf(a, b); }
  // lower lane keeps min(x, p), upper lane keeps max(x, p): one v_med3_f32 against -inf / +inf
  static __device__ __forceinline__ float bound(bool upper) { return upper ? __builtin_inff() : -__builtin_inff(); }
  static __device__ __forceinline__ float pick(float x, float p, float bnd) { return __builtin_amdgcn_fmed3f(x, p, bnd); }
};
struct U32Keys {                      // unsigned keys (packed quantised coordinate | original index)
  typedef unsigned type;
  static __device__ __forceinline__ unsigned lo(unsigned a, unsigned b) { return a < b ? a : b; }
  static __device__ __forceinline__ unsigned hi(unsigned a, unsigned b) { return a < b ? b : a; }
  static __device__ __forceinline__ unsigned bound(bool upper) { return upper ? 0xffffffffu : 0u; }
  static __device__ __forceinline__ unsigned pick(unsigned x, unsigned p, unsigned bnd) {
    unsigned r;                        // v_med3_u32 has no clang builtin; a plain VALU op needs no wait states
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(p), "v"(bnd));
    return r;
  }
};

template <int MASK>
__device__ __forceinline__ unsigned lane_xor(unsigned x, int lane) {
  return (unsigned)as_i(lane_xor<MASK>(as_f((int)x), lane));
}

// 64-bit items (key bits << 32 | original index); policy U64Items further down
typedef unsigned long long item_t;

__device__ __forceinline__ item_t make_item(float key, int idx) {
  return ((item_t)(unsigned)as_i(key) << 32) | (unsigned)idx;
}
__device__ __forceinline__ float item_key(item_t it) { return as_f((int)(it >> 32)); }
__device__ __forceinline__ int item_idx(item_t it) { return (int)(unsigned)it; }

template <int MASK>
__device__ __forceinline__ item_t lane_xor(item_t x, int lane) {
  const float lo = lane_xor<MASK>(as_f((int)(unsigned)x), lane);
  const float hi = lane_xor<MASK>(as_f((int)(x >> 32)), lane);
  return ((item_t)(unsigned)as_i(hi) << 32) | (unsigned)as_i(lo);
}

template <class P>
__device__ __forceinline__ void cmp_swap(typename P::type& lo, typename P::type& hi) {
  const typename P::type a = lo, b = hi;
  lo = P::lo(a, b);
  hi = P::hi(a, b);
}

// in-lane half-cleaner stages with strides J, J/2, ..., 1 (ascending, compile-time register pairs)
template <class P, int EPT, int J>
__device__ __forceinline__ void lane_stages(typename P::type (&x)[EPT]) {
  if constexpr (J >= 1) {
#pragma unroll
    for (int r = 0; r < EPT; ++r) {
      if ((r & J) == 0) cmp_swap<P>(x[r], x[r | J]);
    }
    lane_stages<P, EPT, J / 2>(x);
  }
}

// in-lane merges of size K = 2, 4, ..., EPT
template <class P, int EPT, int K>
__device__ __forceinline__ void lane_merges(typename P::type (&x)[EPT]) {
  if constexpr (K <= EPT) {
#pragma unroll
    for (int r = 0; r < EPT; ++r) {
      const int q = r ^ (K - 1);
      if (q > r) cmp_swap<P>(x[r], x[q]);
    }
    lane_stages<P, EPT, K / 4>(x);
    lane_merges<P, EPT, K * 2>(x);
  }
}

// cross-lane xor stages with lane masks M, M/2, ..., 1
template <class P, int EPT, int M>
__device__ __forceinline__ void xlane_stages(typename P::type (&x)[EPT], int lane) {
  if constexpr (M >= 1) {
    const typename P::type bnd = P::bound((lane & M) != 0);
    // batches of SHW_XLANE_BATCH moves in flight before their compare-exchanges: a wave then stalls once per
    // batch for the LDS-crossbar latency instead of once per handful of elements (hipcc on its own keeps 5)
    constexpr int CH = EPT < SHW_XLANE_BATCH ? EPT : SHW_XLANE_BATCH;
#pragma unroll
    for (int r0 = 0; r0 < EPT; r0 += CH) {
      typename P::type part[CH];
#pragma unroll
      for (int j = 0; j < CH; ++j) part[j] = lane_xor<M>(x[r0 + j], lane);
#pragma unroll
      for (int j = 0; j < CH; ++j) x[r0 + j] = P::pick(x[r0 + j], part[j], bnd);
    }
    xlane_stages<P, EPT, M / 2>(x, lane);
  }
}

// merges spanning 2^C lanes, C = 1 .. 6
template <class P, int EPT, int C>
__device__ __forceinline__ void xlane_merges(typename P::type (&x)[EPT], int lane) {
  if constexpr (C <= 6) {
    constexpr int MASK = (1 << C) - 1;
    const typename P::type bnd = P::bound((lane & (1 << (C - 1))) != 0);
#ifdef SHW_ABL_NO_XLANE
    if constexpr (false) {
#else
    if constexpr (EPT == 1) {
#endif
      x[0] = P::pick(x[0], lane_xor<MASK>(x[0], lane), bnd);
    } else {
#pragma unroll
#ifdef SHW_ABL_NO_XLANE
      for (int r = 0; r < 0; ++r) {
#else
      for (int r = 0; r < EPT / 2; ++r) {          // mirror pairs (r, EPT-1-r): two temporaries live
#endif
        const typename P::type pa = lane_xor<MASK>(x[EPT - 1 - r], lane);
        const typename P::type pb = lane_xor<MASK>(x[r], lane);
        x[r] = P::pick(x[r], pa, bnd);
        x[EPT - 1 - r] = P::pick(x[EPT - 1 - r], pb, bnd);
      }
    }
#ifndef SHW_ABL_NO_XLANE
    xlane_stages<P, EPT, (1 << C) / 4>(x, lane);
#endif
#ifndef SHW_ABL_NO_INLANE
    lane_stages<P, EPT, EPT / 2>(x);
#endif
    xlane_merges<P, EPT, C + 1>(x, lane);
  }
}

// ascending sort of the 64*EPT keys of a wave; sorted position of x[r] in lane `lane` is lane*EPT + r.
template <int EPT>
__device__ __forceinline__ void wave_sort(float (&x)[EPT], int lane) {
#ifndef SHW_ABL_NO_SORT
  lane_merges<F32Keys, EPT, 2>(x);
  xlane_merges<F32Keys, EPT, 1>(x, lane);
#endif
}
template <int EPT>
__device__ __forceinline__ void wave_sort(unsigned (&x)[EPT], int lane) {
  lane_merges<U32Keys, EPT, 2>(x);
  xlane_merges<U32Keys, EPT, 1>(x, lane);
}

// ---------------------------------------------------------------------------------------------
// Key + payload variant: items are 64-bit words (float key bits << 32 | original index).  Keys are
// non-negative floats (circle coordinates, +inf padding) or order-preserving transforms of signed floats,
// whose bit patterns order like unsigned integers, and the index makes every item unique, so the result is
// the STABLE ascending order of the keys (ties by original index) -- the order torch.sort yields in the
// reference (:163-164).  Same network, third key policy: a compare-exchange costs one v_cmp_*_u64 and two
// v_cndmask per item instead of one v_min/v_max/v_med3, which is why the loss-only kernels sort bare keys
// and the training kernel sorts packed 32-bit words (shw_ssw_grad.hip).
// ---------------------------------------------------------------------------------------------
struct U64Items {
  typedef item_t type;
  static __device__ __forceinline__ item_t lo(item_t a, item_t b) { return a > b ? b : a; }
  static __device__ __forceinline__ item_t hi(item_t a, item_t b) { return a > b ? a : b; }
  static __device__ __forceinline__ item_t bound(bool upper) { return upper ? ~0ull : 0ull; }
  // the lower lane keeps the smaller item, the upper lane the larger; items are unique: no tie case
  static __device__ __forceinline__ item_t pick(item_t x, item_t p, item_t bnd) {
    const bool take = (p < x) != (bnd != 0ull);
    return take ? p : x;
  }
};

template <int EPT>
__device__ __forceinline__ void wave_sort_kv(item_t (&x)[EPT], int lane) {
  lane_merges<U64Items, EPT, 2>(x);
  xlane_merges<U64Items, EPT, 1>(x, lane);
}

// ----- wave-wide sums (result valid in every lane) ---------------------------------------------
__device__ __forceinline__ float wave_sum(float v, int lane) {
  v += lane_xor<1>(v, lane);
  v += lane_xor<2>(v, lane);
  v += lane_xor<4>(v, lane);
  v += lane_xor<8>(v, lane);
  v += lane_xor<16>(v, lane);
  v += lane_xor<32>(v, lane);
  return v;
}

__device__ __forceinline__ float wave_max(float v, int lane) {
  v = fmaxf(v, lane_xor<1>(v, lane));
  v = fmaxf(v, lane_xor<2>(v, lane));
  v = fmaxf(v, lane_xor<4>(v, lane));
  v = fmaxf(v, lane_xor<8>(v, lane));
  v = fmaxf(v, lane_xor<16>(v, lane));
  v = fmaxf(v, lane_xor<32>(v, lane));
  return v;
}

// same, but the result is handed back through an SGPR so that the compiler's divergence analysis
// knows it is wave-uniform (branches on it become scalar branches, values derived from it stay in
// SGPRs instead of being recomputed per lane)
__device__ __forceinline__ float wave_sum_uniform(float v, int lane) {
  return as_f(__builtin_amdgcn_readfirstlane(as_i(wave_sum(v, lane))));
}

}  // namespace shw
