// shw_ssw_grad_coop.hip -- loss + gradient coefficients for p != 1, 2049..8192 points: W = 2 or 4 wavefronts of one
// workgroup per (pair, slice), 32 atoms per lane (the one-wave forms of shw_ssw_grad.hip /
// shw_ssw_grad_kv.hip keep 64 / 128 atoms per lane and run at 1.6 / 8 ms per step at 4096 / 8192 points).
//
//   sorts     both clouds with their permutations by the cooperative distribution sort on 64-bit items
//             (coop_sort_kv.hpp: exact coordinates, stable order, nothing to repair), source first; its sorted
//             coordinates stay in registers, its sorted original indices go to a 16-bit row;
//   solve     the target's sorted coordinates become rows [r][64 W] (n == 2048 W) or pre-rotated extended rows (any n,
//             ssw_common.hpp ExtRows); every wave evaluates c(k-1), c(k), c(k+1) on its own atoms, the partial sums are
//             added in wave order through LDS (shw_ssw_coop.hip);
//   gradient  g = (1/n) d|D|^p/dD at D = u_(e) - v_ext(e + k*), scattered by original index into two staging rows
//             (every entry written exactly once: the permutations are bijections) and stored coalesced
//             (shw_ssw_grad2.hip, phase 4).  No atomics, deterministic.
// LDS per slice: 12 bytes per atom slot (8: the item buffer, later the two staging rows; 4: counters + index row).
#include "coop_sort_kv.hpp"
#include "ssw_common.hpp"

namespace shw {

#ifndef SHW_GRADCOOP_MINW
#define SHW_GRADCOOP_MINW 2      // waves per SIMD asked of the register allocator at W = 2 (measured 0.84 against 0.96 ms
                                 // with 1 at 4096 points; W = 4: one workgroup per CU whatever the registers)
#endif

#ifndef SHW_GRADCOOP_KPB
#define SHW_GRADCOOP_KPB 2       // keys per bin of the item sort
#endif

template <int EPT, int W, int PMODE, bool FULL>
__global__ __launch_bounds__(W * 64, W == 2 ? SHW_GRADCOOP_MINW : 1) void ssw_forward_grad_coop_kernel(SswArgs A) {
  constexpr int KPB = SHW_GRADCOOP_KPB;
  typedef Coop<EPT, W, KPB> C;
  typedef ExtRows<EPT, C::NCOL> X;
  static_assert(C::NB * 2 <= C::CAP, "counters take at most half a row: the index row takes the other half");
  static_assert(X::FLOATS <= 2 * C::CAP, "extended rows fit the item buffer");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  unsigned* cnt = reinterpret_cast<unsigned*>(lds);                              // counters, then the target's index row
  unsigned short* idx_t = reinterpret_cast<unsigned short*>(lds);
  unsigned short* idx_s = reinterpret_cast<unsigned short*>(lds + C::CAP / 2);   // the source's index row
  float* area = lds + C::CAP;                                                    // 2 CAP floats:
  item_t* buf = reinterpret_cast<item_t*>(area);                                 //   the sorts' item buffer,
  float* rows = area;                                                            //   then the target rows,
  float* stage_s = area;                                                         //   then the two staging rows
  float* stage_t = area + C::CAP;
  int* red = reinterpret_cast<int*>(lds + 3 * C::CAP);
  float* redf = reinterpret_cast<float*>(red) + 2 * W;       // [2 parities][W][4] partial sums, then [W][2] coordinate sums

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int gl = wave * 64 + lane;
  const int s = xcd_contiguous_id(blockIdx.x, A.num_groups);   // one workgroup per (pair, slice)
  const int b = s / A.slices, l = s - b * A.slices;
  const int n = A.n;                                            // == A.m on this path

  float U[6];
  load_frame(A.dirs, (long)b * A.u_pair_stride + (long)l * 6, U);   // (3,2) row-major: U[2*d + k]

  coop_zero_counters<EPT, W, KPB>(cnt, gl);
  item_t it[EPT];
  float u[EPT];
  float part_u = 0.f, part_v = 0.f;
#pragma nounroll
  for (int which = 0; which < 2; ++which) {                     // 0: source, 1: target
    const float* Xp = (which == 0 ? A.xs : A.xt) + (long)b * n * A.pstride;
    int g2 = gl;
    asm volatile("" : "+v"(g2));
    float part;
    {
      float key[EPT];
      part = load_coords<EPT, FULL, false, C::NCOL>(Xp, n, g2, U, key);
#pragma unroll
      for (int r = 0; r < EPT; ++r) it[r] = make_item(key[r], r * C::NCOL + g2);
    }
    __syncthreads();                                            // counters zeroed; the previous sort's buffer read
    coop_sort_kv<EPT, W, FULL, KPB>(it, wave, lane, n, cnt, buf, red);
    if (which == 0) {
      part_u = wave_sum_uniform(part, lane);
#pragma unroll
      for (int r = 0; r < EPT; ++r) {
        u[r] = item_key(it[r]);
        idx_s[r * C::NCOL + gl] = (unsigned short)item_idx(it[r]);
      }
    } else {
      part_v = wave_sum_uniform(part, lane);
    }
  }
  float v[EPT];                                                 // sorted target coordinates (kept for re-centring, any n)
#pragma unroll
  for (int r = 0; r < EPT; ++r) v[r] = item_key(it[r]);
  float* sums = redf + 8 * W;
  if (lane == 0) { sums[wave * 2] = part_u; sums[wave * 2 + 1] = part_v; }
  __syncthreads();                                              // every wave has read its items back: buffer and counters free
#pragma unroll
  for (int r = 0; r < EPT; ++r) idx_t[r * C::NCOL + gl] = (unsigned short)item_idx(it[r]);
  if constexpr (FULL) {
#pragma unroll
    for (int r = 0; r < EPT; ++r) rows[r * C::NCOL + gl] = v[r];
  }
  float sum_u = 0.f, sum_v = 0.f;
#pragma unroll
  for (int q = 0; q < W; ++q) { sum_u += sums[q * 2]; sum_v += sums[q * 2 + 1]; }

  // ---- the shift: minimise the convex sequence c(k), |k| <= n (shw_ssw_coop.hip) -------------------------------
  int lo = -n, hi = n;
  float guess = rintf(sum_u - sum_v);
  guess = fminf(fmaxf(guess, (float)lo), (float)hi);
  int k = __builtin_amdgcn_readfirstlane((int)guess);
  bool lo_tight = false, hi_tight = false;
  int step = 1;
  float cm = 0.f, c0 = 0.f, cp = 0.f;
  int kc = k;
  if constexpr (FULL) __syncthreads();                          // rows and index row written
  else ext_rows_write<EPT, C::NCOL>(v, rows, gl, n, kc);       // (its barriers also publish the index row)
  for (int itn = 0; itn < 64; ++itn) {
    int g2 = gl;
    asm volatile("" : "+v"(g2));
    float pm, p0, pp;
    if constexpr (FULL) {
      shift_costs3_full<EPT, PMODE, C::NCOL>(u, rows, g2, k, A.p, A.p_int, pm, p0, pp);
    } else {
      if (k - kc >= X::M || kc - k >= X::M) {                   // uniform over the workgroup: re-centre the rows
        kc = k;
        ext_rows_write<EPT, C::NCOL>(v, rows, g2, n, kc);
      }
      shift_costs3_ext<EPT, PMODE, EPT, C::NCOL>(u, rows, g2, n, k - kc, A.p, A.p_int, pm, p0, pp);
    }
    float* slot = redf + (itn & 1) * 4 * W;                     // two parities: one barrier per evaluation
    if (lane == 0) { slot[wave * 4] = pm; slot[wave * 4 + 1] = p0; slot[wave * 4 + 2] = pp; }
    __syncthreads();
    cm = c0 = cp = 0.f;
#pragma unroll
    for (int q = 0; q < W; ++q) { cm += slot[q * 4]; c0 += slot[q * 4 + 1]; cp += slot[q * 4 + 2]; }
    cm = as_f(__builtin_amdgcn_readfirstlane(as_i(cm)));
    c0 = as_f(__builtin_amdgcn_readfirstlane(as_i(c0)));
    cp = as_f(__builtin_amdgcn_readfirstlane(as_i(cp)));
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
  const float inv_n = 1.f / (float)n;
  if (threadIdx.x == 0) {
    A.slice_cost[s] = c0 * inv_n;
    if (A.slice_shift) A.slice_shift[s] = k;
  }

  // ---- coefficients ------------------------------------------------------------------------------------------------
  unsigned pair_idx[EPT];                                       // (source index) | (target index) << 16
#pragma unroll
  for (int r = 0; r < EPT; ++r) {
    const int e = gl * EPT + r;
    const int q = min(e, n - 1) + k;                            // in [-n, 2n): one turn at most
    const int turn = (q < 0) ? -1 : ((q >= n) ? 1 : 0);
    const int qq = q - turn * n;
    const int tslot = lds_slot<EPT, C::NCOL>(qq);
    float tv;
    if constexpr (FULL) {
      tv = rows[tslot] + (float)turn;
    } else {
      const int E = min(e, n - 1) + (k - kc) + X::M;            // the rows hold the turns already
      tv = rows[emod<EPT>(E) * X::RS + ediv<EPT>(E)];
    }
    const float d = u[r] - tv;
    u[r] = dpow_abs<PMODE>(d, A.p, A.p_int) * inv_n;
    pair_idx[r] = (unsigned)idx_s[r * C::NCOL + gl] | ((unsigned)idx_t[tslot] << 16);
  }
  __syncthreads();                                              // every coordinate has been read: the area becomes staging rows
#pragma unroll
  for (int r = 0; r < EPT; ++r) {
    if (FULL || gl * EPT + r < n) {
      stage_s[pair_idx[r] & 0xffffu] = u[r];
      stage_t[pair_idx[r] >> 16] = -u[r];
    }
  }
  __syncthreads();
  float* cs = A.coef_s + (long)s * n;
  float* ct = A.coef_t + (long)s * n;
#pragma unroll
  for (int r = 0; r < EPT; ++r) {
    const int i = r * C::NCOL + gl;
    if (FULL || i < n) { cs[i] = stage_s[i]; ct[i] = stage_t[i]; }
  }
}

template <int EPT, int W>
static int launch_forward_grad_coop(SswArgs& A, hipStream_t stream) {
  typedef Coop<EPT, W> C;
  const long total = (long)A.pairs * A.slices;
  if (total > 0x7fffffffL) return (int)hipErrorInvalidValue;
  A.num_groups = (int)total;
  const size_t lds = (size_t)(3 * C::CAP + C::RED) * sizeof(float);
  const bool full = is_pow2(EPT) && (A.n == C::CAP) && (A.m == C::CAP);
  const dim3 grid((unsigned)total), block(W * 64);
#define SHW_LAUNCH_GRAD_COOP(PM, FL)                                                                              \
  do {                                                                                                            \
    auto kern = ssw_forward_grad_coop_kernel<EPT, W, PM, FL>;                                                     \
    static bool raised[64] = {};      /* once per instantiation and device (not inside a later stream capture) */   \
    int dev_ = 0;                                                                                                 \
    (void)hipGetDevice(&dev_);                                                                                    \
    if (lds > 64 * 1024 && !raised[dev_ & 63]) {                                                                  \
      const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                               \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);             \
      if (e != hipSuccess) return (int)e;                                                                         \
      raised[dev_ & 63] = true;                                                                                   \
    }                                                                                                             \
    hipLaunchKernelGGL(kern, grid, block, lds, stream, A);                                                        \
  } while (0)
  if constexpr (is_pow2(EPT)) {                            // (the mask-free forms: power-of-two classes only)
    if (full) {
      if (A.p_int == 2) SHW_LAUNCH_GRAD_COOP(2, true); else SHW_LAUNCH_GRAD_COOP(0, true);
      return (int)hipGetLastError();
    }
  }
  if (A.p_int == 2) SHW_LAUNCH_GRAD_COOP(2, false); else SHW_LAUNCH_GRAD_COOP(0, false);
#undef SHW_LAUNCH_GRAD_COOP
  return (int)hipGetLastError();
}

// 2049..4096 points: W = 2 waves per slice, 4097..8192: W = 4; 20 / 24 / 32 atoms per lane (round 3: 3000 points pay
// for 3072 slots, 5000 for 5120)
int dispatch_forward_grad_coop(SswArgs& A, hipStream_t stream) {
  if (A.n != A.m) return (int)hipErrorInvalidValue;
  const int W = next_pow2(A.n) / 2048;
  switch (W * 100 + coop_kpl_for(A.n, W, true)) {
#ifndef SHW_DEV_ONLY_EPT
    case 220: return launch_forward_grad_coop<20, 2>(A, stream);
    case 224: return launch_forward_grad_coop<24, 2>(A, stream);
    case 232: return launch_forward_grad_coop<32, 2>(A, stream);
    case 420: return launch_forward_grad_coop<20, 4>(A, stream);
    case 432: return launch_forward_grad_coop<32, 4>(A, stream);
#endif
    default: return (int)hipErrorInvalidValue;
  }
}

// Small grids (round 3): when a launch has fewer (pair, slice) problems than the chip has SIMDs -- the notebooks' gradient
// flow is ONE pair x 100 slices (Flow_cube.ipynb:1381) -- a slice is latency, not throughput: 8 atoms per lane and
// W = padded / 512 waves per slice (4 at 1025..2048 points) cut the dependent chain of the sort and the solve and put
// 400 waves on the chip instead of 200.  (For full grids the same form is slower: wave scans, barriers and the seam sort
// are paid W times -- profiles/r02_ab_coop_keys_per_lane.txt.)
int dispatch_forward_grad_small_grid(SswArgs& A, hipStream_t stream) {
  if (A.n != A.m) return (int)hipErrorInvalidValue;
  const int padded = next_pow2(A.n);
  switch (padded / 512) {
#ifndef SHW_DEV_ONLY_EPT
    case 1: return launch_forward_grad_coop<8, 1>(A, stream);
    case 2: return launch_forward_grad_coop<8, 2>(A, stream);
    case 4: return launch_forward_grad_coop<8, 4>(A, stream);
#endif
    default: return (int)hipErrorInvalidValue;
  }
}

}  // namespace shw
