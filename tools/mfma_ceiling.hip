// What does a box of this pool sustain on the bf16 matrix cores when NOTHING else runs?  A register-only loop of
// v_mfma_f32_32x32x16_bf16 (one wave per SIMD, 1024 SIMDs, operands fixed in registers) on random, on post-ReLU-like (half zeros)
// and on all-zero operands, and the same loop with ds_read_b128 fragment reads beside the MFMAs at the rates the convolution
// kernels run (0.7 and 1.5 reads per MFMA).  Prints TFLOP/s and the shader clock each variant ran at (s_memtime ticks of a wave /
// wall time of its launch), so that the convolution kernels' rates can be read against THIS part's ceiling instead of the nominal
// 2.5 PFLOP/s (MI355X_MICROARCH.md, DVFS give-back: 1 247 TFLOP/s at 1.90-1.95 GHz for a tuned GEMM loop on random data).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_ceiling.hip -o /tmp/mfma_ceiling && /tmp/mfma_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <cstring>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// 10 independent accumulators (the wide-tile kernel's 5 pixel blocks x 2 row blocks), 2 + 5 operand fragments per "tap".
// LDSR: how many of the 7 operand fragments of a tap are re-read from LDS (ds_read_b128, random content) before its 10 MFMAs:
// 0 = register-only loop, 7 = 0.7 reads per MFMA (the wide-tile kernel), and LDSX extra reads whose results are dropped
// (LDSR 7 + LDSX 8 = 1.5 reads per MFMA, the ws kernel's small tiles).
template <int LDSR, int LDSX>
__global__ __launch_bounds__(256, 1) void mfma_loop(const u32x4* __restrict__ src, float* __restrict__ sink, unsigned long long* ticks, int iters) {
  __shared__ u32x4 lds[4096];  // 64 KB
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = src[((blockIdx.x & 3) * 4096 + i) & 32767];
  __syncthreads();
  u32x4 op[7];
#pragma unroll
  for (int i = 0; i < 7; ++i) op[i] = src[(8192 + i * 1024 + threadIdx.x * 3) & 32767];
  f32x16 acc[2][5];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const unsigned base = (unsigned)(size_t)lds + (unsigned)((wave * 64 + lane) * 16);
  const unsigned long long t0 = __builtin_readcyclecounter();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
      for (int r = 0; r < LDSR; ++r)
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(op[r]) : "v"(base), "n"((t * 7 + r) * 1024 % 49152));
#pragma unroll
      for (int r = 0; r < LDSX; ++r) {
        u32x4 v;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(base), "n"((t * 8 + r) * 1024 % 49152 + 4096));
      }
      if constexpr (LDSR + LDSX > 0)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(op[0]), "+v"(op[1]), "+v"(op[2]), "+v"(op[3]), "+v"(op[4]), "+v"(op[5]), "+v"(op[6]));
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, op[i]), __builtin_bit_cast(bf16x8, op[2 + j]), acc[i][j], 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  if (s == 12345.678f) sink[threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { ticks[0] = t1 - t0; ticks[1] = r1 - r0; }
}

static unsigned short f2bf(float f) {
  unsigned u;
  memcpy(&u, &f, 4);
  return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

int main(int argc, char** argv) {
  const bool json = argc > 1 && !strcmp(argv[1], "--json");  // one line: random operands, registers only / +0.7 reads
  CK(hipSetDevice(0));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  if (!json) printf("# %s, %d CUs, clockRate %d kHz\n", prop.gcnArchName, cus, prop.clockRate);
  const size_t nvec = 32768;  // 512 KB of operand data
  std::vector<unsigned short> h(nvec * 8);
  u32x4* d_src;
  float* d_sink;
  unsigned long long* d_ticks;
  CK(hipMalloc(&d_src, nvec * 16));
  CK(hipMalloc(&d_sink, 4096));
  CK(hipMalloc(&d_ticks, 64));
  CK(hipMemset(d_ticks, 0, 64));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int iters = 40;  // 40 x 90 MFMAs x 32 cycles = 115 k cycles ~ 60 us per launch
  const double flop_per_launch = (double)cus * 4 * iters * 90 * 2.0 * 32 * 32 * 16;
  const char* kinds[3] = {"random N(0,1)", "post-ReLU-like (B operand half zeros)", "all zeros"};
  double jt[2] = {0, 0}, jl[2] = {0, 0}, jc[2] = {0, 0};
  for (int rep = 0; rep < (json ? 1 : 2); ++rep)
    for (int kind = 0; kind < (json ? 1 : 3); ++kind) {
      srand(1234 + kind);
      for (size_t i = 0; i < h.size(); ++i) {
        float u1 = (rand() + 1.0f) / (RAND_MAX + 2.0f), u2 = (rand() + 1.0f) / (RAND_MAX + 2.0f);
        float g = sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);
        if (kind == 1 && i >= 4096 * 8 && g < 0.f) g = 0.f;
        if (kind == 2) g = 0.f;
        h[i] = f2bf(g);
      }
      CK(hipMemcpy(d_src, h.data(), nvec * 16, hipMemcpyHostToDevice));
      for (int v = 0; v < (json ? 2 : 3); ++v) {
        auto launch = [&]() {
          if (v == 0) hipLaunchKernelGGL((mfma_loop<0, 0>), dim3(cus), dim3(256), 0, 0, d_src, d_sink, d_ticks, iters);
          else if (v == 1) hipLaunchKernelGGL((mfma_loop<7, 0>), dim3(cus), dim3(256), 0, 0, d_src, d_sink, d_ticks, iters);
          else hipLaunchKernelGGL((mfma_loop<7, 8>), dim3(cus), dim3(256), 0, 0, d_src, d_sink, d_ticks, iters);
        };
        for (int i = 0; i < 2000; ++i) launch();  // ~0.12 s: lets the clock settle under load
        CK(hipDeviceSynchronize());
        const int N = 4000;
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < N; ++i) launch();
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long t[2];
        CK(hipMemcpy(t, d_ticks, 16, hipMemcpyDeviceToHost));
        const double us = ms * 1e3 / N;
        const double loop_us = t[1] / 100.0;  // s_memrealtime: 100 MHz
        if (json) { jt[v] = flop_per_launch / us / 1e6; jl[v] = flop_per_launch / loop_us / 1e6; jc[v] = t[0] / loop_us / 1e3; continue; }
        printf("%-34s %-38s: %7.2f us/launch %7.1f TFLOP/s (%.3f of 2500) | in-loop: %6.2f us, %7.1f TFLOP/s, %llu cycle-counter ticks = %.3f GHz, %.1f ticks per MFMA\n",
               v == 0 ? "registers only" : v == 1 ? "+0.7 ds_read_b128 per MFMA" : "+1.5 ds_read_b128 per MFMA", kinds[kind], us,
               flop_per_launch / us / 1e6, flop_per_launch / us / 1e6 / 2500.0, loop_us, flop_per_launch / loop_us / 1e6, t[0],
               t[0] / loop_us / 1e3, (double)t[0] / (iters * 90.0));
        fflush(stdout);
      }
    }
  if (json)
    printf("{\"registers_only\": {\"tflops_per_launch\": %.1f, \"tflops_in_loop\": %.1f, \"clock_GHz\": %.3f}, "
           "\"with_0p7_lds_reads_per_mfma\": {\"tflops_per_launch\": %.1f, \"tflops_in_loop\": %.1f, \"clock_GHz\": %.3f}, \"cus\": %d}\n",
           jt[0], jl[0], jc[0], jt[1], jl[1], jc[1], cus);
  return 0;
}
