// Probe: does a wave that streams MFMAs leave room for ANOTHER wave of the same SIMD to run VALU / LDS / VMEM work?
// One 512-thread workgroup per CU (8 waves = 2 per SIMD; 100 KiB of LDS requested so that only one fits).  Waves 0-3 run
// role A, waves 4-7 role B.  Roles: 0 idle, 1 MFMA 32x32x2 f32 stream (8 independent accumulators), 2 VALU FMA stream,
// 3 LDS ds_read_b128 stream, 4 global 16-byte store stream, 5 global 16-byte load stream.
// Prints the time of A alone, B alone and both: "both ~ max" = the two overlap, "both ~ sum" = they serialise.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_overlap_probe.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int ROLE>
__device__ __forceinline__ void run_role(int iters, float *gbuf, float *lds, float &sink) {
  const int lane = threadIdx.x & 63;
  if (ROLE == 1) {
    f32x16 acc[8];
    for (int t = 0; t < 8; ++t) for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    float a = 1.0f + lane, b = 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 16; ++k)
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
    }
    for (int t = 0; t < 8; ++t) sink += acc[t][0];
  } else if (ROLE == 6) {
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float a = 1.0f + lane, b = 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 32; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    sink += acc[0];
  } else if (ROLE == 7) {
    f32x4 acc[8];
    for (int t = 0; t < 8; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a = 1.0f + lane, b = 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 32; ++k)
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
    }
    for (int t = 0; t < 8; ++t) sink += acc[t][0];
  } else if (ROLE == 2) {
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = lane + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 64; ++k)
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = x[i] * 1.0001f + 0.5f;
    }
    for (int i = 0; i < 16; ++i) sink += x[i];
  } else if (ROLE == 3) {
    f32x4 s = {0, 0, 0, 0};
    const f32x4 *p = reinterpret_cast<const f32x4 *>(lds) + lane;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 64; ++k) { f32x4 v = p[(k * 64) & 1023]; s += v; }
    }
    sink += s.x + s.y + s.z + s.w;
  } else if (ROLE == 4) {
    f32x4 v = {1.f, 2.f, 3.f, (float)lane};
    f32x4 *p = reinterpret_cast<f32x4 *>(gbuf) + ((size_t)blockIdx.x * 512 + threadIdx.x);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 16; ++k) p[(size_t)((it * 16 + k) & 1023) * 256 * 512] = v;
    }
  } else if (ROLE == 5) {
    f32x4 s = {0, 0, 0, 0};
    const f32x4 *p = reinterpret_cast<const f32x4 *>(gbuf) + ((size_t)blockIdx.x * 512 + threadIdx.x);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 16; ++k) s += p[(size_t)((it * 16 + k) & 1023) * 256 * 512];
    }
    sink += s.x + s.y + s.z + s.w;
  }
}

template <int RA, int RB>
__global__ __launch_bounds__(512, 2) void probe(int ia, int ib, float *gbuf, float *out, int prio) {
  extern __shared__ float lds[];
  if (prio) { if ((threadIdx.x >> 6) < 4) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(3); }
  for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = i;
  __syncthreads();
  float sink = 0.f;
  if ((threadIdx.x >> 6) < 4) run_role<RA>(ia, gbuf, lds, sink);
  else run_role<RB>(ib, gbuf, lds, sink);
  if (sink == 123.456f) out[threadIdx.x] = sink;
}

template <int RA, int RB>
static float time_it(int ia, int ib, float *gbuf, float *out, int prio = 0) {
  hipFuncSetAttribute(reinterpret_cast<const void *>(probe<RA, RB>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<RA, RB><<<256, 512, 100 * 1024>>>(ia, ib, gbuf, out, prio);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<RA, RB><<<256, 512, 100 * 1024>>>(ia, ib, gbuf, out, prio);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}

template <int RB, int RA = 1>
static void pair(const char *name, int ia, int ib, float *gbuf, float *out) {
  const float a = time_it<RA, 0>(ia, 0, gbuf, out), b = time_it<0, RB>(0, ib, gbuf, out), ab = time_it<RA, RB>(ia, ib, gbuf, out);
  const float abp = time_it<RA, RB>(ia, ib, gbuf, out, 1);
  printf("A[%d] %.3f ms | %-12s %.3f ms | both %.3f ms  (max %.3f, sum %.3f) | B at s_setprio 3: %.3f ms\n", RA, a, name, b, ab, a > b ? a : b, a + b, abp);
}

int main() {
  float *gbuf, *out;
  hipMalloc(&gbuf, (size_t)1024 * 256 * 512 * 16);   // 2 GiB: each thread walks 1024 slots, 2 MiB apart
  hipMalloc(&out, 4096);
  hipMemset(gbuf, 0, (size_t)1024 * 256 * 512 * 16);
  pair<1>("MFMA", 2000, 2000, gbuf, out);
  pair<2>("VALU fma", 2000, 1000, gbuf, out);
  pair<3>("LDS b128", 2000, 8000, gbuf, out);
  pair<4>("global store", 2000, 600, gbuf, out);
  pair<5>("global load", 2000, 600, gbuf, out);
  printf("-- A = dependent MFMA chain (one accumulator: the pipe has bubbles)\n");
  pair<2, 6>("VALU fma", 2000, 1000, gbuf, out);
  pair<5, 6>("global load", 2000, 600, gbuf, out);
  printf("-- A = MFMA 16x16x4 f32 (8 passes), 8 accumulators\n");
  pair<2, 7>("VALU fma", 4000, 1000, gbuf, out);
  pair<5, 7>("global load", 4000, 600, gbuf, out);
  printf("-- A = VALU fma stream\n");
  pair<5, 2>("global load", 6000, 600, gbuf, out);
  pair<3, 2>("LDS b128", 6000, 8000, gbuf, out);
  return 0;
}
