// shw_ssw_fwd.hip -- loss-only kernel for p != 1 (key-only register sort).  See ssw_common.hpp.
#include <cstdlib>

#include "ssw_common.hpp"

namespace shw {

// register budget: the LDS footprint (EPT*256 B per wave) admits 20 waves per CU at EPT = 32, so
// ask the allocator for 5 waves per SIMD there (<= 96 VGPRs); larger EPT take what they need.
// The general-power variant (powf) is left at 4.  At EPT = 64 the unconstrained allocator takes all 256
// registers (1 wave per SIMD); asked for 3 waves it needs 144 with no spill, which is what 16 KB of LDS per
// wave can use.
constexpr int min_waves_per_simd(int ept, int pmode) {
  return ept <= 16 ? (pmode == 2 ? 6 : 4) : (ept == 32 ? (pmode == 2 ? 5 : 3) : (ept == 64 ? 3 : 1));
}

// FULL: n == m == 64*EPT (no padding atoms): mask-free projection and the fast shift evaluation.
template <int EPT, int WAVES, int PMODE, bool FULL>
__global__ __launch_bounds__(WAVES * 64, min_waves_per_simd(EPT, PMODE)) void ssw_forward_kernel(SswArgs A) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* vbuf = lds + wave * (EPT * kWave);

  const int vid = xcd_contiguous_id(blockIdx.x, A.num_groups);
  const int s = vid * WAVES + wave;
  if (s >= A.pairs * A.slices) return;            // wave-uniform
  const int b = s / A.slices, l = s - b * A.slices;

  const float* Ul = A.dirs + (long)b * A.u_pair_stride + (long)l * 6;
  float U[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) U[i] = Ul[i];       // (3,2) row-major: U[2*d + k]

  float key[EPT];
  float sum_v = 0.f, sum_u = 0.f;
#pragma nounroll
  for (int which = 0; which < 2; ++which) {        // 0: target -> LDS, 1: source -> registers
    const float* X = which == 0 ? A.xt + (long)b * A.m * 3 : A.xs + (long)b * A.n * 3;
    const int count = which == 0 ? A.m : A.n;
    // Opaque copy of the lane id: keeps the compiler from hoisting the ~60 lane-dependent stage
    // constants of the sort (and the point offsets) out of this loop and holding them in VGPRs.
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const float part = load_coords<EPT, FULL>(X, count, ln, U, key);
    wave_sort<EPT>(key, ln);
    if (which == 0) {
      sum_v = wave_sum(part, lane);
#pragma unroll
      for (int r = 0; r < EPT; ++r) vbuf[r * kWave + lane] = key[r];
    } else {
      sum_u = wave_sum(part, lane);
    }
  }
  __builtin_amdgcn_wave_barrier();

  float best;
  const int k = solve_shift<EPT, PMODE, FULL>(key, vbuf, lane, A.n, sum_u, sum_v, A.p, A.p_int, best);
  if (lane == 0) {
    A.slice_cost[s] = best / (float)A.n;
    if (A.slice_shift) A.slice_shift[s] = k;
  }
}

template <int EPT, int WAVES>
static int launch_forward(SswArgs& A, hipStream_t stream) {
  const long total = (long)A.pairs * A.slices;
  const long groups = (total + WAVES - 1) / WAVES;
  if (groups > 0x7fffffffL) return (int)hipErrorInvalidValue;
  A.num_groups = (int)groups;
  size_t lds = (size_t)WAVES * EPT * kWave * sizeof(float);
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

int dispatch_forward(SswArgs& A, hipStream_t stream) {
  switch (ept_for(A.n, A.m)) {
#ifdef SHW_DEV_ONLY_EPT   // developer switch: compile a single size class quickly
#ifndef SHW_DEV_FWD_WAVES
#define SHW_DEV_FWD_WAVES (SHW_DEV_ONLY_EPT <= 32 ? 4 : (SHW_DEV_ONLY_EPT == 64 ? 2 : 1))
#endif
    case SHW_DEV_ONLY_EPT: return launch_forward<SHW_DEV_ONLY_EPT, SHW_DEV_FWD_WAVES>(A, stream);
#else
    case 1: return launch_forward<1, 4>(A, stream);
    case 2: return launch_forward<2, 4>(A, stream);
    case 4: return launch_forward<4, 4>(A, stream);
    case 8: return launch_forward<8, 4>(A, stream);
    case 16: return launch_forward<16, 4>(A, stream);
    case 32: return launch_forward<32, 4>(A, stream);
    case 64: return launch_forward<64, 2>(A, stream);
    case 128: return launch_forward<128, 1>(A, stream);
#endif
    default: return (int)hipErrorInvalidValue;
  }
}


}  // namespace shw
