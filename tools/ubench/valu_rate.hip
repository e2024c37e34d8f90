// Micro-benchmark: VALU issue rate per SIMD on gfx950 for the instruction kinds the sort uses.
// build: hipcc -O3 --offload-arch=gfx950 -o valu_rate valu_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int KIND>
__global__ void kern(float* out, int iters, float seed) {
  float x[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = seed + i + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 4; ++rep) {
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        if (KIND == 0) {            // min/max pair (in-lane compare-exchange)
          float a = x[i], b = x[i + 1];
          x[i] = __builtin_fminf(a, b);
          x[i + 1] = __builtin_fmaxf(a, b);
        } else if (KIND == 1) {     // fma
          x[i] = fmaf(x[i], 1.0001f, 0.5f);
          x[i + 1] = fmaf(x[i + 1], 0.9999f, 0.25f);
        } else if (KIND == 2) {     // med3
          x[i] = __builtin_amdgcn_fmed3f(x[i], x[i + 1], seed);
          x[i + 1] = __builtin_amdgcn_fmed3f(x[i + 1], x[i], -seed);
        } else if (KIND == 3) {     // ds_swizzle + med3 (cross-lane stage on the LDS crossbar)
          float p = __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, x[i]), (4 << 10) | 0x1f));
          float q = __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, x[i + 1]), (4 << 10) | 0x1f));
          x[i] = __builtin_amdgcn_fmed3f(x[i], p, seed);
          x[i + 1] = __builtin_amdgcn_fmed3f(x[i + 1], q, seed);
        } else if (KIND == 4) {     // dpp mov + med3
          float p = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x[i]), 0xB1, 0xf, 0xf, true));
          float q = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x[i + 1]), 0xB1, 0xf, 0xf, true));
          x[i] = __builtin_amdgcn_fmed3f(x[i], p, seed);
          x[i + 1] = __builtin_amdgcn_fmed3f(x[i + 1], q, seed);
        } else if (KIND == 5) {     // gfx950 v_permlane32_swap on a register pair + min/max + swap back: the
          // lane^32 compare-exchange of TWO keys with full-rate VALU only (4 instructions per 2 keys)
          int a = __builtin_bit_cast(int, x[i]), b = __builtin_bit_cast(int, x[i + 1]);
          asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
          float fa = __builtin_bit_cast(float, a), fb = __builtin_bit_cast(float, b);
          float lo = __builtin_fminf(fa, fb), hi = __builtin_fmaxf(fa, fb);
          a = __builtin_bit_cast(int, lo); b = __builtin_bit_cast(int, hi);
          asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
          x[i] = __builtin_bit_cast(float, a);
          x[i + 1] = __builtin_bit_cast(float, b);
        } else if (KIND == 6) {     // the same with v_permlane16_swap (lane^16 ... rows of 16)
          int a = __builtin_bit_cast(int, x[i]), b = __builtin_bit_cast(int, x[i + 1]);
          asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
          float fa = __builtin_bit_cast(float, a), fb = __builtin_bit_cast(float, b);
          float lo = __builtin_fminf(fa, fb), hi = __builtin_fmaxf(fa, fb);
          a = __builtin_bit_cast(int, lo); b = __builtin_bit_cast(int, hi);
          asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
          x[i] = __builtin_bit_cast(float, a);
          x[i + 1] = __builtin_bit_cast(float, b);
        }
      }
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
void run(const char* name, int valu_per_iter, int ds_per_iter) {
  float* out;
  hipMalloc(&out, 256 * 8 * 1024 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000;
  for (int wps : {1, 2, 3, 4, 6, 8}) {           // waves per SIMD
    const int threads = 256;                       // 4 waves per block = 1 per SIMD
    const int blocks = 256 * wps;                  // one block per CU per wave-per-SIMD
    kern<KIND><<<blocks, threads>>>(out, 10, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kern<KIND><<<blocks, threads>>>(out, iters, 1.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double valu = (double)valu_per_iter * iters * wps;      // VALU wave-instr per SIMD
    const double ds = (double)ds_per_iter * iters * wps * 4;      // DS wave-instr per CU
    printf("%-22s waves/SIMD=%d  %.3f ms  VALU/SIMD per us: %.1f  (cycles/VALU @2.4GHz: %.2f)  DS/CU per us: %.1f\n",
           name, wps, ms, valu / (ms * 1e3), ms * 1e-3 * 2.4e9 / valu, ds / (ms * 1e3));
  }
  hipFree(out);
}

int main() {
  run<0>("min/max", 64, 0);
  run<1>("fma", 64, 0);
  run<2>("med3", 64, 0);
  run<3>("ds_swizzle+med3", 64, 64);
  run<4>("dpp_mov+med3", 128, 0);
  run<5>("permlane32_swap x2+minmax", 128, 0);
  run<6>("permlane16_swap x2+minmax", 128, 0);
  return 0;
}
