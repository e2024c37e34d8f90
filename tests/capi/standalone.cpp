// Standalone consumer of the C ABI (no Python, no torch): links libshw_hip.so against the SYSTEM HIP runtime,
// reads clouds + frames from a binary file, runs shw_ssw_forward + shw_ssw_reduce + shw_chamfer_forward on
// hipMalloc'ed buffers and prints the results.  tests/test_ssw_gpu.py compiles and runs it and compares with the
// values obtained through the Python mirror.
//   file layout: int32 B, n, L; float xs[B*n*3], xt[B*n*3], dirs[B*L*6]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "shw.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); return 2; } } while (0)

int main(int argc, char** argv) {
  if (argc < 2) return 1;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 1;
  int hdr[3];
  if (fread(hdr, sizeof(int), 3, f) != 3) return 1;
  const int B = hdr[0], n = hdr[1], L = hdr[2];
  std::vector<float> xs((size_t)B * n * 3), xt(xs.size()), dirs((size_t)B * L * 6);
  if (fread(xs.data(), 4, xs.size(), f) != xs.size() || fread(xt.data(), 4, xt.size(), f) != xt.size() ||
      fread(dirs.data(), 4, dirs.size(), f) != dirs.size()) return 1;
  fclose(f);
  if (shw_abi_version() != SHW_ABI_VERSION) return 3;
  float *dxs, *dxt, *dd, *cost, *pair, *total, *mxy, *myx, *cpair;
  int32_t *shift, *nxy, *nyx;
  CK(hipMalloc(&dxs, xs.size() * 4)); CK(hipMalloc(&dxt, xt.size() * 4)); CK(hipMalloc(&dd, dirs.size() * 4));
  CK(hipMalloc(&cost, (size_t)B * L * 4)); CK(hipMalloc(&shift, (size_t)B * L * 4));
  CK(hipMalloc(&pair, B * 4)); CK(hipMalloc(&total, 8)); CK(hipMalloc(&cpair, B * 4));
  CK(hipMalloc(&mxy, (size_t)B * n * 4)); CK(hipMalloc(&myx, (size_t)B * n * 4));
  CK(hipMalloc(&nxy, (size_t)B * n * 4)); CK(hipMalloc(&nyx, (size_t)B * n * 4));
  CK(hipMemcpy(dxs, xs.data(), xs.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dxt, xt.data(), xt.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dd, dirs.data(), dirs.size() * 4, hipMemcpyHostToDevice));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  int rc = shw_ssw_forward(dxs, dxt, dd, B, n, n, L, (long)L * 6, 2.0f, cost, shift, st);
  if (!rc) rc = shw_ssw_reduce(cost, B, L, 1.0f / L, pair, total, st);
  if (!rc) rc = shw_chamfer_forward(dxs, dxt, B, n, n, mxy, nxy, myx, nyx, cpair, st);
  if (rc) { fprintf(stderr, "shw call failed: %d\n", rc); return 4; }
  CK(hipStreamSynchronize(st));
  std::vector<float> hp(B), hc(B);
  float ht[2];
  CK(hipMemcpy(hp.data(), pair, B * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(hc.data(), cpair, B * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(ht, total, 8, hipMemcpyDeviceToHost));
  printf("total %.9g %.9g\n", ht[0], ht[1]);
  for (int b = 0; b < B; ++b) printf("pair %d %.9g %.9g\n", b, hp[b], hc[b]);
  // invalid arguments are rejected, not executed
  if (shw_ssw_forward(dxs, dxt, dd, B, n, n + 1, L, (long)L * 6, 2.0f, cost, shift, st) != 1) return 5;
  return 0;
}
