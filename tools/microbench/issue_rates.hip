// What a CU's scalar pipe sustains beside its vector pipes ("nobody has yet asked what the scalar pipe
// costs").  The grid kernels run one wave per env with wave-uniform control flow on SGPRs: 710-870 SALU beside 985 VALU per
// env-step, 32 waves per CU.  This program measures, on an otherwise idle chip, the issue rate per CU of
//   salu      independent s_add_u32 / s_and_b64 / s_lshl_b64 mixes on 8 accumulators
//   salu_dep  ONE dependent chain of s_add_u32 (the latency a wave-uniform walk pays per step)
//   valu      independent v_add_u32 on 8 accumulators
//   valu_dep  one dependent chain of v_add_u32
//   mix11     1 SALU : 1 VALU interleaved (the grid kernels' ratio is ~0.75 : 1)
//   ballot    v_cmp (writes an SGPR pair) -> s_and_b64 -> s_ff1 -> v_readlane chain: the shuffle walk's inner link
// for 1, 2, 4, 8, 16, 32 waves per CU (one 64-lane workgroup per wave, as the step kernels launch).
//   hipcc --offload-arch=gfx950 -O3 -o issue_rates issue_rates.hip && ./issue_rates
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef unsigned int u32;
typedef unsigned long long u64;

#define CK(x)                                                                          \
  do {                                                                                 \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess) {                                                            \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(2);                                                                         \
    }                                                                                  \
  } while (0)

constexpr int kUnroll = 64;  // instructions of the measured kind per loop iteration and wave

#define REP8(X) X X X X X X X X
#define REP64(X) REP8(REP8(X))

template <int MODE> __global__ __launch_bounds__(64) void k_issue(u32* out, u32 iters) {
  u32 s0 = blockIdx.x, s1 = 1, s2 = 2, s3 = 3, s4 = 4, s5 = 5, s6 = 6, s7 = 7;
  u32 v0 = threadIdx.x, v1 = 1, v2 = 2, v3 = 3, v4 = 4, v5 = 5, v6 = 6, v7 = 7;
  for (u32 it = 0; it < iters; ++it) {
    if (MODE == 0) {  // 64 independent SALU (8 accumulators x 8)
      REP8(asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n"
                        "s_add_u32 %4, %4, 1\n s_add_u32 %5, %5, 1\n s_add_u32 %6, %6, 1\n s_add_u32 %7, %7, 1"
                        : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7)::"scc");)
    } else if (MODE == 1) {  // 64 dependent SALU
      REP64(asm volatile("s_add_u32 %0, %0, 1" : "+s"(s0)::"scc");)
    } else if (MODE == 2) {  // 64 independent VALU
      REP8(asm volatile("v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n v_add_u32 %2, %2, 1\n v_add_u32 %3, %3, 1\n"
                        "v_add_u32 %4, %4, 1\n v_add_u32 %5, %5, 1\n v_add_u32 %6, %6, 1\n v_add_u32 %7, %7, 1"
                        : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));)
    } else if (MODE == 3) {  // 64 dependent VALU
      REP64(asm volatile("v_add_u32 %0, %0, 1" : "+v"(v0));)
    } else if (MODE == 4) {  // 32 SALU + 32 VALU interleaved, all independent
      REP8(asm volatile("s_add_u32 %0, %0, 1\n v_add_u32 %4, %4, 1\n s_add_u32 %1, %1, 1\n v_add_u32 %5, %5, 1\n"
                        "s_add_u32 %2, %2, 1\n v_add_u32 %6, %6, 1\n s_add_u32 %3, %3, 1\n v_add_u32 %7, %7, 1"
                        : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3)::"scc");)
    } else if (MODE == 5) {  // the shuffle walk's link, 16 per iteration (4 instructions each = 64): cmp -> and -> ff1 -> readlane
      u64 avail = ~0ull;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        u64 hit;
        u32 pos, j;
        asm volatile("v_cmp_le_u32 %0, %3, %4\n s_and_b64 %0, %0, %5\n s_ff1_i32_b64 %1, %0\n s_nop 3\n v_readlane_b32 %2, %3, %1"
                     : "=&s"(hit), "=&s"(pos), "=s"(j)
                     : "v"(v0), "v"(v1), "s"(avail)
                     : "scc", "vcc");
        avail ^= (u64)j << 7;
        v1 += j & 1u;
      }
      s0 += (u32)avail;
    } else if (MODE == 6) {  // 64-bit scalar ops (mask bookkeeping): s_and_b64 / s_lshl_b64 / s_bcnt1 independent
      u64 a = s0, b = s1, c = s2, d = s3;
      REP8(asm volatile("s_and_b64 %0, %0, %1\n s_lshl_b64 %1, %1, 1\n s_or_b64 %2, %2, %3\n s_andn2_b64 %3, %3, %0\n"
                        "s_and_b64 %0, %0, %2\n s_lshl_b64 %1, %1, 1\n s_or_b64 %2, %2, %1\n s_andn2_b64 %3, %3, %2"
                        : "+s"(a), "+s"(b), "+s"(c), "+s"(d)::"scc");)
      s0 = (u32)a, s1 = (u32)b, s2 = (u32)c, s3 = (u32)d;
    }
  }
  if ((s0 ^ s1 ^ s2 ^ s3 ^ s4 ^ s5 ^ s6 ^ s7 ^ v0 ^ v1 ^ v2 ^ v3 ^ v4 ^ v5 ^ v6 ^ v7) == 0x12345u) out[blockIdx.x * 64 + threadIdx.x] = 1;
}

template <int MODE> static double run(u32* out, int blocks, u32 iters) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k_issue<MODE>, dim3(blocks), dim3(64), 0, 0, out, 16u);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 5; ++r) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_issue<MODE>, dim3(blocks), dim3(64), 0, 0, out, iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  return best * 1e-3;
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const double clk = prop.clockRate * 1e3;  // Hz (the boost clock the runtime reports; the sustained clock may be lower)
  printf("# %s, %d CUs, reported clock %.0f MHz; instructions per CYCLE PER CU at the reported clock (per wave in brackets)\n", prop.name, cus,
         clk / 1e6);
  u32* out;
  CK(hipMalloc(&out, (size_t)cus * 32 * 64 * 4 + 4096));
  const char* names[] = {"salu independent", "salu dependent chain", "valu independent", "valu dependent chain", "1 salu : 1 valu",
                         "cmp->and->ff1->readlane", "salu 64-bit mask ops"};
  const u32 iters = 4000;
  printf("%-26s", "waves per CU");
  for (int w : {1, 2, 4, 8, 16, 32}) printf(" %14d", w);
  printf("\n");
  for (int m = 0; m < 7; ++m) {
    printf("%-26s", names[m]);
    for (int w : {1, 2, 4, 8, 16, 32}) {
      const int blocks = cus * w;
      double s = 0;
      switch (m) {
        case 0: s = run<0>(out, blocks, iters); break;
        case 1: s = run<1>(out, blocks, iters); break;
        case 2: s = run<2>(out, blocks, iters); break;
        case 3: s = run<3>(out, blocks, iters); break;
        case 4: s = run<4>(out, blocks, iters); break;
        case 5: s = run<5>(out, blocks, iters); break;
        case 6: s = run<6>(out, blocks, iters); break;
      }
      const double per_wave = (double)iters * (m == 5 ? 16 : kUnroll);  // instructions of the kind(s) per wave (mode 5: LINKS, 16 per iteration)
      const double ipc_cu = per_wave * w / (s * clk);
      printf(" %7.3f (%5.3f)", ipc_cu, ipc_cu / w);
    }
    printf("\n");
  }
  CK(hipFree(out));
  return 0;
}
