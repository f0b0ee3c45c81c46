// Microbenchmark: what VALU issue rate can a gfx950 SIMD actually sustain for the instruction
// mixes of the march kernel?  (DESIGN.md section 3 quotes the result.)
//   hipcc --offload-arch=gfx950 -O3 -o valu_issue valu_issue.hip && ./valu_issue
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
  float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5,
        a6 = a0 + 6, a7 = a0 + 7;
  const float m = 0.999f, c = 0.001f;
  int sacc = 0;
  typedef float float2v __attribute__((ext_vector_type(2)));
  float2v p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pm = {m, m}, pc = {c, c};
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      if (MODE == 0) {  // 8 independent chains
        asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n"
                     "v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n"
                     "v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(m), "v"(c));
      } else if (MODE == 1) {  // one dependent chain
        asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                     "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                     "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                     : "+v"(a0) : "v"(m), "v"(c));
      } else if (MODE == 2) {  // 2 dependent chains interleaved (ILP 2)
        asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %0, %0, %2, %3\n"
                     "v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n"
                     "v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n"
                     : "+v"(a0), "+v"(a1) : "v"(m), "v"(c));
      } else if (MODE == 3) {  // 8 VALU (independent) + 4 SALU
        asm volatile("v_fma_f32 %0, %0, %9, %10\n s_add_u32 %8, %8, 1\n v_fma_f32 %1, %1, %9, %10\n"
                     "v_fma_f32 %2, %2, %9, %10\n s_add_u32 %8, %8, 1\n v_fma_f32 %3, %3, %9, %10\n"
                     "v_fma_f32 %4, %4, %9, %10\n s_add_u32 %8, %8, 1\n v_fma_f32 %5, %5, %9, %10\n"
                     "v_fma_f32 %6, %6, %9, %10\n s_add_u32 %8, %8, 1\n v_fma_f32 %7, %7, %9, %10\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7),
                       "+s"(sacc)
                     : "v"(m), "v"(c));
      } else if (MODE == 4) {  // dependent chain + 4 SALU
        asm volatile("v_fma_f32 %0, %0, %2, %3\n s_add_u32 %1, %1, 1\n v_fma_f32 %0, %0, %2, %3\n"
                     "v_fma_f32 %0, %0, %2, %3\n s_add_u32 %1, %1, 1\n v_fma_f32 %0, %0, %2, %3\n"
                     "v_fma_f32 %0, %0, %2, %3\n s_add_u32 %1, %1, 1\n v_fma_f32 %0, %0, %2, %3\n"
                     "v_fma_f32 %0, %0, %2, %3\n s_add_u32 %1, %1, 1\n v_fma_f32 %0, %0, %2, %3\n"
                     : "+v"(a0), "+s"(sacc) : "v"(m), "v"(c));
      } else if (MODE == 5) {  // 6 fma (dependent) + v_sqrt + v_rcp (transcendental, dependent)
        asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                     "v_sqrt_f32 %0, %0\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                     "v_rcp_f32 %0, %0\n v_fma_f32 %0, %0, %1, %2\n"
                     : "+v"(a0) : "v"(m), "v"(c));
      } else if (MODE == 7) {  // packed fp32: 4 independent v_pk_fma_f32 chains (2 flops-pairs per lane)
        asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n"
                     "v_pk_fma_f32 %3, %3, %4, %5\n v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n"
                     "v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                     : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pm), "v"(pc));
      } else if (MODE == 8) {  // v_pk_mul_f32 + v_pk_add_f32 alternating, 4 chains
        asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %5\n v_pk_mul_f32 %2, %2, %4\n"
                     "v_pk_add_f32 %3, %3, %5\n v_pk_mul_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %5\n"
                     "v_pk_mul_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %5\n"
                     : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pm), "v"(pc));
      } else if (MODE == 9) {  // 8 dependent v_sqrt_f32: the transcendental rate alone
        asm volatile("v_sqrt_f32 %0, %0\n s_nop 0\n v_sqrt_f32 %0, %0\n s_nop 0\n v_sqrt_f32 %0, %0\n s_nop 0\n v_sqrt_f32 %0, %0\n s_nop 0\n"
                     "v_sqrt_f32 %0, %0\n s_nop 0\n v_sqrt_f32 %0, %0\n s_nop 0\n v_sqrt_f32 %0, %0\n s_nop 0\n v_sqrt_f32 %0, %0\n s_nop 0\n"
                     : "+v"(a0));
      } else if (MODE == 10) {  // 7 fma + 1 sqrt, dependent: does the root overlap other waves' fma?
        asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                     "v_sqrt_f32 %0, %0\n s_nop 0\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                     "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                     : "+v"(a0) : "v"(m), "v"(c));
      } else if (MODE == 11) {  // 4 fma + 4 sqrt alternating, dependent
        asm volatile("v_fma_f32 %0, %0, %1, %2\n v_sqrt_f32 %0, %0\n s_nop 0\n v_fma_f32 %0, %0, %1, %2\n v_sqrt_f32 %0, %0\n s_nop 0\n"
                     "v_fma_f32 %0, %0, %1, %2\n v_sqrt_f32 %0, %0\n s_nop 0\n v_fma_f32 %0, %0, %1, %2\n v_sqrt_f32 %0, %0\n s_nop 0\n"
                     : "+v"(a0) : "v"(m), "v"(c));
      } else if (MODE == 12) {  // the march's event mix: 25 fma-class + 2 sqrt, dependent (x8 per iteration -> scaled below)
        asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                     "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                     "v_fma_f32 %0, %0, %1, %2\n v_sqrt_f32 %0, %0\n s_nop 0\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                     "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                     "v_sqrt_f32 %0, %0\n s_nop 0\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                     "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                     "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                     : "+v"(a0) : "v"(m), "v"(c));
      } else if (MODE == 13) {  // event mix, the two roots ADJACENT (the second one's input does not need the first)
        asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %0, %0, %4, %5\n v_add_f32 %1, %0, %5\n v_sqrt_f32 %2, %0\n v_sqrt_f32 %3, %1\n s_nop 0\n"
                     "v_fma_f32 %0, %2, %4, %3\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %0, %0, %4, %5\n "
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(c));
      } else if (MODE == 14) {  // event mix, three independent fma between each root and its first use
        asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %0, %0, %4, %5\n v_sqrt_f32 %2, %0\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %0, %2, %4, %5\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %0, %0, %4, %5\n "
                     "v_sqrt_f32 %3, %0\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %0, %3, %4, %5\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %0, %0, %4, %5\n "
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(c));
      } else if (MODE == 15) {  // event mix, roots adjacent AND three independent fma before the first use
        asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %0, %0, %4, %5\n v_add_f32 %1, %0, %5\n v_sqrt_f32 %2, %0\n v_sqrt_f32 %3, %1\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %1, %1, %4, %5\n "
                     "v_fma_f32 %0, %2, %4, %3\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %0, %0, %4, %5\n "
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(c));
      } else if (MODE == 6) {  // cmp + cndmask pairs through VCC, dependent
        asm volatile("v_cmp_gt_f32 vcc, %0, %2\n v_cndmask_b32 %0, %0, %1, vcc\n v_fma_f32 %0, %0, %1, %2\n"
                     "v_cmp_gt_f32 vcc, %0, %2\n v_cndmask_b32 %0, %0, %1, vcc\n v_fma_f32 %0, %0, %1, %2\n"
                     "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                     : "+v"(a0) : "v"(m), "v"(c) : "vcc");
      }
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + sacc + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}

template <int MODE>
double run(float* d, int blocks, int iters, int valu_per_iter) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 100, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double winst = (double)blocks * 4 * iters * 8.0 * valu_per_iter;
  return winst / (ms * 1e-3);
}

int main() {
  float* d;
  hipMalloc(&d, sizeof(float) * 256 * 256 * 16);
  const char* names[] = {"8 independent fma chains", "1 dependent fma chain", "2 interleaved chains",
                         "8 indep fma + 4 SALU per 8", "dependent fma + 4 SALU per 8",
                         "dependent: 6 fma + sqrt + rcp", "dependent: 2x(cmp+cndmask) + 4 fma",
                         "4 indep v_pk_fma_f32 chains", "v_pk_mul/v_pk_add alternating",
                         "dependent: 8 v_sqrt_f32", "dependent: 7 fma + 1 sqrt", "dependent: 4 x (fma, sqrt)",
                         "dependent: event mix 25 fma + 2 sqrt", "event mix, roots adjacent",
                         "event mix, 3 indep fma after each root", "event mix, adjacent + 3 indep"};
  for (int wps : {2, 8}) {  // waves per SIMD = blocks per CU (256 threads = 4 waves = 1/SIMD)
    int blocks = 256 * wps;
    double r[16];
    r[0] = run<0>(d, blocks, 20000, 8);
    r[1] = run<1>(d, blocks, 20000, 8);
    r[2] = run<2>(d, blocks, 20000, 8);
    r[3] = run<3>(d, blocks, 20000, 8);
    r[4] = run<4>(d, blocks, 20000, 8);
    r[5] = run<5>(d, blocks, 20000, 8);
    r[6] = run<6>(d, blocks, 20000, 8);
    r[7] = run<7>(d, blocks, 20000, 8);
    r[8] = run<8>(d, blocks, 20000, 8);
    r[9] = run<9>(d, blocks, 20000, 8);
    r[10] = run<10>(d, blocks, 20000, 8);
    r[11] = run<11>(d, blocks, 20000, 8);
    r[12] = run<12>(d, blocks, 6000, 27);
    r[13] = run<13>(d, blocks, 6000, 27);
    r[14] = run<14>(d, blocks, 6000, 27);
    r[15] = run<15>(d, blocks, 6000, 27);
    for (int i = 0; i < 16; i++)
      if (i != 3 && i != 4) printf("waves/SIMD %d  %-36s %.3e VALU wave-instr/s  (%.2f cyc/instr/SIMD @2.4GHz)\n", wps, names[i],
             r[i], 1024 * 2.4e9 / r[i]);
  }
  return 0;
}
