// Where do the workgroups of a small grid land?  (development aid)
// Every workgroup records the XCD and CU it ran on and when it started / ended; the host prints how many
// workgroups each CU got and how the starts spread.  Build: mkdir -p tools/bin && hipcc --offload-arch=gfx950 -O2 -w -o tools/bin/dispatchbench tools/dispatchbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#include <algorithm>

template <int LDS_BYTES>
__global__ __launch_bounds__(256) void k_probe(unsigned *out, int iters) {
  __shared__ float S[LDS_BYTES / 4];
  unsigned hwid, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const unsigned long long t0 = wall_clock64();
  float a = threadIdx.x * 0.001f;
  for (int i = 0; i < iters; i++) {
    S[(threadIdx.x + i) % (LDS_BYTES / 4)] = a;
    __syncthreads();
    a = a * 1.0001f + S[(threadIdx.x * 7 + i) % (LDS_BYTES / 4)];
    __syncthreads();
  }
  const unsigned long long t1 = wall_clock64();
  if (threadIdx.x == 0) {
    unsigned *o = out + 6 * blockIdx.x;
    o[0] = hwid; o[1] = xcc; o[2] = (unsigned)t0; o[3] = (unsigned)(t0 >> 32); o[4] = (unsigned)(t1 - t0); o[5] = __float_as_uint(a);
  }
}

template <int LDS_BYTES>
static void run(int nblocks, int iters) {
  unsigned *d;
  hipMalloc(&d, sizeof(unsigned) * 6 * nblocks);
  for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL(k_probe<LDS_BYTES>, dim3(nblocks), dim3(256), 0, 0, d, iters);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k_probe<LDS_BYTES>, dim3(nblocks), dim3(256), 0, 0, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned> h(6 * nblocks);
  hipMemcpy(h.data(), d, sizeof(unsigned) * h.size(), hipMemcpyDeviceToHost);
  std::map<unsigned, int> per_cu;
  unsigned long long tmin = ~0ull, tmax = 0; double dur = 0;
  for (int b = 0; b < nblocks; b++) {
    const unsigned hw = h[6 * b], xcc = h[6 * b + 1] & 0xF;
    const unsigned cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    per_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu]++;
    const unsigned long long t0 = h[6 * b + 2] | ((unsigned long long)h[6 * b + 3] << 32);
    tmin = std::min(tmin, t0); tmax = std::max(tmax, t0); dur += h[6 * b + 4];
  }
  int hist[16] = {0}; int mx = 0;
  for (auto &kv : per_cu) { hist[std::min(kv.second, 15)]++; mx = std::max(mx, kv.second); }
  printf("LDS %3d KB, %4d workgroups x %d iters: kernel %.1f us | CUs used %zu, workgroups per CU:", LDS_BYTES / 1024, nblocks, iters, ms * 1e3, per_cu.size());
  for (int i = 1; i <= mx; i++) printf(" %dx:%d", i, hist[i]);
  printf(" | starts spread %.2f us, mean workgroup %.2f us (100 MHz clock)\n", (tmax - tmin) / 100.0, dur / nblocks / 100.0);
  hipFree(d);
}

int main(int argc, char **argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 400;
  for (int nb : {256, 308, 528, 924, 1024}) run<40960>(nb, iters);
  for (int nb : {264, 528}) run<61440>(nb, iters);
  for (int nb : {256, 528}) run<8192>(nb, iters);
  return 0;
}
