// shw_ssw_grad.hip -- loss + gradient-coefficient kernel for p != 1 (packed-key register sort) and
// the coefficient -> point-gradient streaming kernel.  See ssw_common.hpp.
#include <cstdlib>

#include "bin_sort_idx.hpp"
#include "ssw_common.hpp"

#ifndef SHW_GRAD_BINSORT
#define SHW_GRAD_BINSORT 1     // 1: distribution sort with indices (bin_sort_idx.hpp) at >= 8 keys per lane
#endif

namespace shw {

// ---------------------------------------------------------------------------------------------
// forward + gradient coefficients (training).
//
// The permutation is needed as well as the sorted values.  Carrying a 64-bit (key, index) item through
// the network costs 2.6x the VALU work of the key-only sort (v_cmp_u64 + v_cndmask pairs instead of
// v_min/v_max/v_med3; measured 1.46 ms vs 0.35 ms per launch at config 3), so the sort runs on ONE
// 32-bit word per atom instead:
//     packed = (floor(coord * 2^QBITS) << IDX_BITS) | original index ,  IDX_BITS = log2(64*EPT)
// (QBITS = 21 at N = 2048) with the unsigned-integer form of the same register network.  Packed keys are
// unique, so the result is ordered by quantised coordinate, ties by original index.  The exact fp32
// coordinates are then gathered by index from an LDS copy, and atoms whose quantised coordinates
// collide (distance < 2^-QBITS; ~1 pair per slice at N = 2048) are put into exact order by an
// odd-even transposition fix-up that runs only when a collision is detected and loops until no
// exchange happens.  The final order is the stable ascending order of the fp32 coordinates -- the
// order torch.sort gives the reference (:163-164) -- and deterministic.
//
// After the shift solve, sorted source position e writes
//     coef_s[slice, idx_u(e)]        = +g ,   g = (1/n) d|D|^p/dD ,  D = u_(e) - v_ext(e + k*)
//     coef_t[slice, idx_v(e + k*)]   = -g
// i.e. d cost / d coordinate in ORIGINAL point order.  Every entry of the two
// coefficient rows is written exactly once (the permutations are bijections): no zero fill, no atomics.
// ---------------------------------------------------------------------------------------------
// waves per SIMD asked of the register allocator: what 6 B of LDS per atom allow (160 KB per CU)
constexpr int grad_waves_per_simd(int ept) { return ept <= 16 ? 4 : (ept == 32 ? 3 : 1); }
constexpr bool grad_uses_bins(int ept) { return SHW_GRAD_BINSORT != 0 && ept >= 8 && ept <= 32; }

template <int EPT, int WAVES, int PMODE, bool FULL>
__global__ __launch_bounds__(WAVES * 64, grad_waves_per_simd(EPT)) void ssw_forward_grad_kernel(SswArgs A) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int ROW = EPT * kWave;
  constexpr int HALF = (EPT + 1) / 2;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // LDS per wave, 6 B per atom (12 KB at N = 2048 -> three waves per SIMD):
  //   row  (4 B): coordinates by ORIGINAL index while a cloud is being sorted, then the sorted target
  //               coordinates for the shift solve, then the staging row of the coefficient un-permutation
  //   vidx (2 B): original indices of the sorted target
  // The SOURCE is sorted first and stays in registers (coordinates + 16-bit index pairs) while the target
  // is sorted, so that one row serves both clouds.
  // (distribution sort: the 2-byte index row doubles as the sort's counters -- 32*EPT of them, the same 128*EPT bytes --
  //  which are dead before the target's sorted indices are written)
  float* row = lds + wave * (ROW * 3 / 2);
  unsigned short* vidx = reinterpret_cast<unsigned short*>(row + ROW);
  unsigned* cnt = reinterpret_cast<unsigned*>(row + ROW);

  const int vid = xcd_contiguous_id(blockIdx.x, A.num_groups);
  const int s = vid * WAVES + wave;
  if (s >= A.pairs * A.slices) return;
  const int b = s / A.slices, l = s - b * A.slices;
  const int n = A.n;

  float U[6];
  load_frame(A.dirs, (long)b * A.u_pair_stride + (long)l * 6, U);   // (3,2) row-major: U[2*d + k]

  float u[EPT];
  unsigned upair[HALF];                              // original indices of the sorted source, two per word
  float sum_v = 0.f, sum_u = 0.f;
#pragma nounroll
  for (int which = 0; which < 2; ++which) {
    const float* X = which == 0 ? A.xs + (long)b * n * A.pstride : A.xt + (long)b * A.m * A.pstride;
    const int count = which == 0 ? n : A.m;
    int ln = lane;
    asm volatile("" : "+v"(ln));
    float val[EPT];
    int idx[EPT];
    float part;
    if constexpr (grad_uses_bins(EPT)) part = sorted_with_indices_binned<EPT>(X, count, ln, U, cnt, row, val, idx);
    else part = sorted_with_indices<EPT>(X, count, ln, U, row, val, idx);
    const float total = wave_sum(part, lane);
    if (which == 0) {
      sum_u = total;
#pragma unroll
      for (int r = 0; r < EPT; ++r) u[r] = val[r];
#pragma unroll
      for (int h = 0; h < HALF; ++h)
        upair[h] = (unsigned)idx[2 * h] | ((2 * h + 1 < EPT) ? ((unsigned)idx[(2 * h + 1) % EPT] << 16) : 0u);
    } else {
      sum_v = total;
      // (every lane has gathered its exact coordinates out of `row` by now: LDS operations of a wave
      //  execute in order)
#pragma unroll
      for (int r = 0; r < EPT; ++r) {
        row[r * kWave + lane] = val[r];
        vidx[r * kWave + lane] = (unsigned short)idx[r];
      }
    }
  }
  __builtin_amdgcn_wave_barrier();

  float best;
  const int k = solve_shift<EPT, PMODE, FULL>(u, row, lane, n, sum_u, sum_v, A.p, A.p_int, best);
  const float inv_n = 1.f / (float)n;
  if (lane == 0) {
    A.slice_cost[s] = best * inv_n;
    if (A.slice_shift) A.slice_shift[s] = k;
  }
  // coefficients: g replaces u in place; un-permute through the staging row, then store coalesced (a direct
  // scatter to global memory costs one address per lane per store: ~0.2 ms per launch at config 3)
  float* cs = A.coef_s + (long)s * n;
  float* ct = A.coef_t + (long)s * A.m;
  auto target_slot = [&](int e, float& off) -> int {
    const int q = min(e, n - 1) + k;                   // in [-n, 2n): one turn at most
    const int turn = (q < 0) ? -1 : ((q >= n) ? 1 : 0);
    off = (float)turn;
    return lds_slot<EPT>(q - turn * n);
  };
#pragma unroll
  for (int r = 0; r < EPT; ++r) {
    float off;
    const int slot = target_slot(lane * EPT + r, off);
    const float d = u[r] - (row[slot] + off);
    u[r] = dpow_abs<PMODE>(d, A.p, A.p_int) * inv_n;
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int r = 0; r < EPT; ++r) {
    const int e = lane * EPT + r;
    const int iu = (int)((r & 1) ? (upair[r / 2] >> 16) : (upair[r / 2] & 0xffffu));
    if (e < n) row[iu] = u[r];
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int r = 0; r < EPT; ++r) {
    const int i = r * kWave + lane;
    if (i < n) cs[i] = row[i];
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int r = 0; r < EPT; ++r) {
    const int e = lane * EPT + r;
    float off;
    const int slot = target_slot(e, off);
    if (e < n) row[vidx[slot]] = -u[r];
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int r = 0; r < EPT; ++r) {
    const int i = r * kWave + lane;
    if (i < A.m) ct[i] = row[i];
  }
}

// ---------------------------------------------------------------------------------------------
// coefficient rows -> point gradients:
//   grad[b,i,:] = scale * sum_l coef[b,l,i] * (-bb U_l[:,0] + a U_l[:,1]) / (2 pi (a^2 + bb^2)),
//   (a, bb) = U_l^T x[b,i].
// The HBM-bound kernel of the path: it streams the coefficient scratch exactly once.  A workgroup owns
// 64 consecutive points of one cloud of one pair; its 4 waves split the slices 4 ways (wave w takes
// slices w, w+4, ...), each lane keeps 8 coalesced loads in flight, and the 4 partial sums are added
// in wave order through LDS -- fixed order, no atomics, bitwise reproducible.  Directions are
// wave-uniform scalar loads.
// ---------------------------------------------------------------------------------------------
template <int NW>
__global__ __launch_bounds__(64 * NW) void ssw_backward_points_kernel(const float* __restrict__ xs,
                                                                  const float* __restrict__ xt,
                                                                  const float* __restrict__ dirs,
                                                                  const float* __restrict__ coef_s,
                                                                  const float* __restrict__ coef_t, int n, int m,
                                                                  int slices, long u_pair_stride, float scale,
                                                                  const float* __restrict__ pair_w,
                                                                  const float* __restrict__ total_w,
                                                                  float* __restrict__ grad_xs,
                                                                  float* __restrict__ grad_xt, int chunks_s) {
  // NW waves split the slices NW ways: 4 for grids that fill the chip, 16 for small ones (the notebooks' one pair x 100
  // slices: 6 slices per wave, one round of loads instead of four)
  __shared__ float part[3][NW][64];
  const int b = blockIdx.y;
  const bool is_t = (int)blockIdx.x >= chunks_s;
  const int chunk = is_t ? blockIdx.x - chunks_s : blockIdx.x;
  const int cnt = is_t ? m : n;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = chunk * 64 + lane;
  const float* X = (is_t ? xt : xs) + (long)b * cnt * 3;
  const float* C = (is_t ? coef_t : coef_s) + (long)b * slices * cnt;
  float* G = (is_t ? grad_xt : grad_xs) + (long)b * cnt * 3;
  const float* Ub = dirs + (long)b * u_pair_stride;
  const int ic = min(i, cnt - 1);
  const float px = X[3 * ic], py = X[3 * ic + 1], pz = X[3 * ic + 2];
  const float inv_two_pi = 0.159154936671257019f;
  float gx = 0.f, gy = 0.f, gz = 0.f;
  auto add = [&](float c, const float* U) {
    const float a = fmaf(pz, U[4], fmaf(py, U[2], px * U[0]));
    const float bb = fmaf(pz, U[5], fmaf(py, U[3], px * U[1]));
    // a point whose projection on the slice plane is exactly (0, 0) (an all-zero / zero-padded point) has no
    // angle: the reference's autograd gives it a ZERO gradient there (torch's atan2 backward masks 0/0 and
    // F.normalize clamps the norm, :274-279), so the quotient is guarded instead of producing inf * 0 = NaN
    const float r2 = fmaf(a, a, bb * bb);
    const float w = r2 > 0.f ? c * inv_two_pi / r2 : 0.f;
    gx = fmaf(w, fmaf(a, U[1], -bb * U[0]), gx);
    gy = fmaf(w, fmaf(a, U[3], -bb * U[2]), gy);
    gz = fmaf(w, fmaf(a, U[5], -bb * U[4]), gz);
  };
  int l = wave;
  for (; l + 7 * NW < slices; l += 8 * NW) {         // 8 slices of this wave per trip, loads first
    float c[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) c[j] = C[(long)(l + NW * j) * cnt + ic];
#pragma unroll
    for (int j = 0; j < 8; ++j) add(c[j], Ub + (long)(l + NW * j) * 6);
  }
  if (l < slices) {                                  // the last, partial trip: the same loads, clamped, then only the live ones
    float c[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) c[j] = C[(long)min(l + NW * j, slices - 1) * cnt + ic];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (l + NW * j < slices) add(c[j], Ub + (long)(l + NW * j) * 6);   // (wave-uniform)
    }
  }
  part[0][wave][lane] = gx;
  part[1][wave][lane] = gy;
  part[2][wave][lane] = gz;
  __syncthreads();
  if (wave == 0 && i < cnt) {
    // upstream gradient folded in: d loss / d pair_loss[b] (+ d loss / d total[0], which every pair feeds)
    float up = (pair_w || total_w) ? 0.f : 1.f;
    if (pair_w) up += pair_w[b];
    if (total_w) up += total_w[0];
    const float sc = scale * up;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      float acc = part[d][0][lane];
#pragma unroll
      for (int w = 1; w < NW; ++w) acc += part[d][w][lane];                // wave order: fixed
      G[3 * i + d] = acc * sc;
    }
  }
}

// The same for clouds whose size is a multiple of 4 (round 3, late): a lane owns FOUR consecutive points and reads their
// coefficients of one slice as one 16-byte load (1 KB per wave-load instead of 256 B), SHW_BWD4_INFLIGHT slices in flight:
// four times the bytes in flight per lane.  A workgroup owns 256 consecutive points; its 4 waves split the slices 4 ways;
// partial sums added in wave order through LDS as above.
#ifndef SHW_BWD4_INFLIGHT
#define SHW_BWD4_INFLIGHT 4
#endif
#ifndef SHW_BWD4_WAVES
#define SHW_BWD4_WAVES 4      // waves per workgroup = ways the slices are split
#endif
__global__ __launch_bounds__(64 * SHW_BWD4_WAVES) void ssw_backward_points4_kernel(const float* __restrict__ xs,
                                                                   const float* __restrict__ xt,
                                                                   const float* __restrict__ dirs,
                                                                   const float* __restrict__ coef_s,
                                                                   const float* __restrict__ coef_t, int n, int m,
                                                                   int slices, long u_pair_stride, float scale,
                                                                   const float* __restrict__ pair_w,
                                                                   const float* __restrict__ total_w,
                                                                   float* __restrict__ grad_xs,
                                                                   float* __restrict__ grad_xt, int chunks_s) {
  constexpr int NW = SHW_BWD4_WAVES;
  __shared__ float part[NW][12][64];
  const int b = blockIdx.y;
  const bool is_t = (int)blockIdx.x >= chunks_s;
  const int chunk = is_t ? blockIdx.x - chunks_s : blockIdx.x;
  const int cnt = is_t ? m : n;                              // a multiple of 4
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i0 = chunk * 256 + lane * 4;                     // first of the lane's four points
  const float* X = (is_t ? xt : xs) + (long)b * cnt * 3;
  const float* C = (is_t ? coef_t : coef_s) + (long)b * slices * cnt;
  float* G = (is_t ? grad_xt : grad_xs) + (long)b * cnt * 3;
  const float* Ub = dirs + (long)b * u_pair_stride;
  const int ic = min(i0, cnt - 4);
  float p[12], g[12];
  {
    const float4* X4 = reinterpret_cast<const float4*>(X + 3 * ic);   // 12 floats = 3 x 16 B (ic a multiple of 4)
    const float4 q0 = X4[0], q1 = X4[1], q2 = X4[2];
    p[0] = q0.x; p[1] = q0.y; p[2] = q0.z; p[3] = q0.w; p[4] = q1.x; p[5] = q1.y; p[6] = q1.z; p[7] = q1.w;
    p[8] = q2.x; p[9] = q2.y; p[10] = q2.z; p[11] = q2.w;
  }
#pragma unroll
  for (int k = 0; k < 12; ++k) g[k] = 0.f;
  const float inv_two_pi = 0.159154936671257019f;
  auto add = [&](const float4 c4, const float* U) {
    const float c[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float px = p[3 * q], py = p[3 * q + 1], pz = p[3 * q + 2];
      const float a = fmaf(pz, U[4], fmaf(py, U[2], px * U[0]));
      const float bb = fmaf(pz, U[5], fmaf(py, U[3], px * U[1]));
      const float r2 = fmaf(a, a, bb * bb);                  // (0, 0) projection: zero gradient, see above
      const float w = r2 > 0.f ? c[q] * inv_two_pi / r2 : 0.f;
      g[3 * q] = fmaf(w, fmaf(a, U[1], -bb * U[0]), g[3 * q]);
      g[3 * q + 1] = fmaf(w, fmaf(a, U[3], -bb * U[2]), g[3 * q + 1]);
      g[3 * q + 2] = fmaf(w, fmaf(a, U[5], -bb * U[4]), g[3 * q + 2]);
    }
  };
  constexpr int F = SHW_BWD4_INFLIGHT;
  int l = wave;
  for (; l + NW * (F - 1) < slices; l += NW * F) {             // F slices of this wave per trip, loads first
    float4 c[F];
#pragma unroll
    for (int j = 0; j < F; ++j) c[j] = *reinterpret_cast<const float4*>(C + (long)(l + NW * j) * cnt + ic);
#pragma unroll
    for (int j = 0; j < F; ++j) add(c[j], Ub + (long)(l + NW * j) * 6);
  }
  for (; l < slices; l += NW) add(*reinterpret_cast<const float4*>(C + (long)l * cnt + ic), Ub + (long)l * 6);
#pragma unroll
  for (int k = 0; k < 12; ++k) part[wave][k][lane] = g[k];
  __syncthreads();
  if (wave == 0 && i0 < cnt) {
    float up = (pair_w || total_w) ? 0.f : 1.f;
    if (pair_w) up += pair_w[b];
    if (total_w) up += total_w[0];
    const float sc = scale * up;
    float o[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      float acc = part[0][k][lane];
#pragma unroll
      for (int w = 1; w < NW; ++w) acc += part[w][k][lane];  // wave order: fixed
      o[k] = acc * sc;
    }
    float4* G4 = reinterpret_cast<float4*>(G + 3 * i0);
    G4[0] = make_float4(o[0], o[1], o[2], o[3]);
    G4[1] = make_float4(o[4], o[5], o[6], o[7]);
    G4[2] = make_float4(o[8], o[9], o[10], o[11]);
  }
}

int launch_forward_grad_kv128(SswArgs& A, hipStream_t stream);

template <int EPT, int WAVES>
static int launch_forward_grad(SswArgs& A, hipStream_t stream) {
  const long total = (long)A.pairs * A.slices;
  const long groups = (total + WAVES - 1) / WAVES;
  if (groups > 0x7fffffffL) return (int)hipErrorInvalidValue;
  A.num_groups = (int)groups;
  const size_t lds = (size_t)WAVES * (EPT * kWave * 6);
  const bool full = (A.n == EPT * kWave) && (A.m == EPT * kWave);
  const dim3 grid((unsigned)groups), block(WAVES * 64);
  if (A.p_int == 2) {
    if (full) hipLaunchKernelGGL((ssw_forward_grad_kernel<EPT, WAVES, 2, true>), grid, block, lds, stream, A);
    else hipLaunchKernelGGL((ssw_forward_grad_kernel<EPT, WAVES, 2, false>), grid, block, lds, stream, A);
  } else {
    if (full) hipLaunchKernelGGL((ssw_forward_grad_kernel<EPT, WAVES, 0, true>), grid, block, lds, stream, A);
    else hipLaunchKernelGGL((ssw_forward_grad_kernel<EPT, WAVES, 0, false>), grid, block, lds, stream, A);
  }
  return (int)hipGetLastError();
}

// SHW_GRAD_KERNEL=onewave (diagnostic, used by the tests): the one-wave kernel of this file at every size
static bool grad_one_wave_forced() {
  static const bool forced = [] { const char* e = getenv("SHW_GRAD_KERNEL"); return e && e[0] == 'o'; }();
  return forced;
}

int dispatch_forward_grad_small_grid(SswArgs& A, hipStream_t stream);   // shw_ssw_grad_coop.hip

// launches with at most this many (pair, slice) problems take the small-grid kernels (SHW_SMALL_GRID overrides; 0 = never)
static long small_grid_slices() {
  static const long v = [] {
    const char* e = getenv("SHW_SMALL_GRID");
    return e ? atol(e) : 1024L;
  }();
  return v;
}

int dispatch_forward_grad(SswArgs& A, hipStream_t stream) {
  {
    const int ept = ept_for(A.n, A.m);
    // fewer problems than SIMDs: latency-bound, W waves per slice (shw_ssw_grad_coop.hip, small grids)
    if (ept >= 8 && ept <= 32 && (long)A.pairs * A.slices <= small_grid_slices() && !grad_one_wave_forced())
      return dispatch_forward_grad_small_grid(A, stream);
    if (ept >= 8 && ept <= 32 && !grad_one_wave_forced()) return dispatch_forward_grad2(A, stream);   // shw_ssw_grad2.hip
    if ((ept == 64 || ept == 128) && !grad_one_wave_forced()) return dispatch_forward_grad_coop(A, stream);   // shw_ssw_grad_coop.hip
  }
  switch (ept_for(A.n, A.m)) {
#ifdef SHW_DEV_ONLY_EPT
    case SHW_DEV_ONLY_EPT: return launch_forward_grad<SHW_DEV_ONLY_EPT, 1>(A, stream);
#else
    case 1: return launch_forward_grad<1, 4>(A, stream);
    case 2: return launch_forward_grad<2, 4>(A, stream);
    case 4: return launch_forward_grad<4, 4>(A, stream);
    case 8: return launch_forward_grad<8, 4>(A, stream);
    case 16: return launch_forward_grad<16, 4>(A, stream);
    case 32: return launch_forward_grad<32, 1>(A, stream);
    case 64: return launch_forward_grad<64, 1>(A, stream);
    case 128: return launch_forward_grad_kv128(A, stream);      // shw_ssw_grad_kv.hip
#endif
    default: return (int)hipErrorInvalidValue;
  }
}


int launch_backward_points(const float* xs, const float* xt, const float* dirs, const float* coef_s,
                           const float* coef_t, int pairs, int n, int m, int slices, long u_pair_stride,
                           float scale, const float* pair_w, const float* total_w, float* grad_xs, float* grad_xt,
                           hipStream_t stream) {
  // sizes that are multiples of 4 (16-byte aligned rows and points): four points per lane, 16-byte loads
  // SHW_BWD_WIDE=0: never; =2: whenever sizes and alignment allow (tests); default: when the grid also fills the chip
  static const int wide_mode = [] { const char* v = getenv("SHW_BWD_WIDE"); return v ? (v[0] == '0' ? 0 : (v[0] == '2' ? 2 : 1)) : 1; }();
  const bool wide_ok = wide_mode != 0;
  const bool aligned = ((reinterpret_cast<uintptr_t>(xs) | reinterpret_cast<uintptr_t>(xt) | reinterpret_cast<uintptr_t>(coef_s) |
                         reinterpret_cast<uintptr_t>(coef_t) | reinterpret_cast<uintptr_t>(grad_xs) |
                         reinterpret_cast<uintptr_t>(grad_xt)) & 15) == 0;
  // ... and launches that fill the chip with 256-point workgroups (two per CU): a small grid -- the notebooks' one pair of
  // 1200 points -- is latency, and the one-point-per-lane kernel has four times the workgroups (9.3 against 15.2 us there)
  const long wide_groups = (long)((n + 255) / 256 + (m + 255) / 256) * pairs;
  if (wide_ok && aligned && n % 4 == 0 && m % 4 == 0 && n >= 4 && m >= 4 && (wide_groups >= 512 || wide_mode == 2)) {
    const int c_s = (n + 255) / 256, c_t = (m + 255) / 256;
    for (int b0 = 0; b0 < pairs; b0 += 65535) {
      const int nb = pairs - b0 < 65535 ? pairs - b0 : 65535;
      hipLaunchKernelGGL(ssw_backward_points4_kernel, dim3(c_s + c_t, nb), dim3(64 * SHW_BWD4_WAVES), 0, stream,
                         xs + (long)b0 * n * 3, xt + (long)b0 * m * 3, dirs + (long)b0 * u_pair_stride,
                         coef_s + (long)b0 * slices * n, coef_t + (long)b0 * slices * m, n, m, slices, u_pair_stride,
                         scale, pair_w ? pair_w + b0 : nullptr, total_w, grad_xs + (long)b0 * n * 3, grad_xt + (long)b0 * m * 3, c_s);
    }
    return (int)hipGetLastError();
  }
  const int chunks_s = (n + 63) / 64, chunks_t = (m + 63) / 64;
  // pairs ride on gridDim.y (<= 65535): larger batches go out as several launches over pair blocks
  for (int b0 = 0; b0 < pairs; b0 += 65535) {
    const int nb = pairs - b0 < 65535 ? pairs - b0 : 65535;
    // fewer workgroups than CUs: sixteen waves per workgroup share the slices
    if ((long)(chunks_s + chunks_t) * pairs < 256 && slices >= 32)
      hipLaunchKernelGGL(ssw_backward_points_kernel<16>, dim3(chunks_s + chunks_t, nb), dim3(1024), 0, stream,
                         xs + (long)b0 * n * 3, xt + (long)b0 * m * 3, dirs + (long)b0 * u_pair_stride,
                         coef_s + (long)b0 * slices * n, coef_t + (long)b0 * slices * m, n, m, slices, u_pair_stride,
                         scale, pair_w ? pair_w + b0 : nullptr, total_w, grad_xs + (long)b0 * n * 3, grad_xt + (long)b0 * m * 3, chunks_s);
    else
      hipLaunchKernelGGL(ssw_backward_points_kernel<4>, dim3(chunks_s + chunks_t, nb), dim3(256), 0, stream,
                         xs + (long)b0 * n * 3, xt + (long)b0 * m * 3, dirs + (long)b0 * u_pair_stride,
                         coef_s + (long)b0 * slices * n, coef_t + (long)b0 * slices * m, n, m, slices, u_pair_stride,
                         scale, pair_w ? pair_w + b0 : nullptr, total_w, grad_xs + (long)b0 * n * 3, grad_xt + (long)b0 * m * 3, chunks_s);
  }
  return (int)hipGetLastError();
}

}  // namespace shw
