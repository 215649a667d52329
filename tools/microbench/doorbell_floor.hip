// Floor of a RESIDENT per-step path beside a launch per step (ce_server_*).
//
// A "server" launch keeps W waves on the chip; per step the caller's stream releases a doorbell word, every workgroup of the
// server takes the step (reads its action word fresh, runs a body, writes its outputs write-through), arrives on a sharded
// counter, and the last arriver publishes `done = seq`, which the caller's stream waits for.  This program measures the period
// of such a step for several geometries / bodies / ways to ring and wait, next to the same body as one plain launch per step.
//   hipcc --offload-arch=gfx950 -O3 -o doorbell_floor doorbell_floor.hip && ./doorbell_floor
// Every spin is bounded by s_memrealtime (100 MHz): a wave that sees no doorbell for 50 ms exits and counts a timeout; the host
// never waits without a deadline.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

typedef unsigned int u32;
typedef unsigned long long u64;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(2);                                                                     \
    }                                                                              \
  } while (0)

constexpr int kShards = 8;
constexpr int kLine = 32;  // u32 words per 128-byte line
struct Ctl {               // every word on a line of its own
  u32 doorbell[kLine];
  u32 stop[kLine];
  u32 done[kLine];
  u32 top[kLine];
  u32 timeouts[kLine];
  u32 shard[kShards][kLine];
};

__device__ __forceinline__ u32 ld_fresh(const u32* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_through(u32* p, u32 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u64 now_ticks() { return __builtin_amdgcn_s_memrealtime(); }
__device__ __forceinline__ void store16_sc1(void* p, u32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}

// the body of one step for one wave: `spin` dependent multiply-adds per lane, then `quads` 16-byte write-through stores per lane
__device__ __forceinline__ u32 body(u32 act, u32 spin, u32 quads, u32x4* out, u32 wave, u32 lane, bool through) {
  u32 x = act | 1u;
  for (u32 i = 0; i < spin; ++i) x = x * 1664525u + 1013904223u;
  for (u32 q = 0; q < quads; ++q) {
    u32x4 v = {x, x + q, wave, lane};
    u32x4* dst = out + ((size_t)wave * quads + q) * 64 + lane;
    if (through) store16_sc1(dst, v);
    else __builtin_nontemporal_store(v, dst);
  }
  return x;
}

__global__ void k_server(Ctl* c, const u32* actions, u32 action_planes, u32x4* out, u32* sink, u32 spin, u32 quads, u32 max_steps,
                         u32 groups_per_shard, u64 timeout_ticks) {
  __shared__ u32 go;
  const u32 lane = threadIdx.x & 63u, w = threadIdx.x >> 6, waves = blockDim.x >> 6;
  const u32 wave = blockIdx.x * waves + w;
  u32 acc = 0;
  for (u32 seq = 1; seq <= max_steps; ++seq) {
    if (w == 0) {
      const u64 t0 = now_ticks();
      u32 ok = 0;
      for (;;) {
        if (ld_fresh(c->doorbell) >= seq) { ok = 1; break; }
        if (ld_fresh(c->stop)) break;
        if (now_ticks() - t0 > timeout_ticks) {
          if (lane == 0) atomicAdd(c->timeouts, 1u);
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
      if (lane == 0) go = ok;
    }
    __syncthreads();
    if (!go) break;
    // the step's input, fresh from memory (another kernel wrote it, possibly through another XCD's L2)
    const u32 act = ld_fresh(actions + (size_t)(seq % action_planes) * gridDim.x * waves + wave);
    acc += body(act, spin, quads, out, wave, lane, true);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's write-through stores have left
    __syncthreads();
    if (threadIdx.x == 0) {
      const u32 sh = blockIdx.x % kShards;
      const u32 a = __hip_atomic_fetch_add(&c->shard[sh][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
      if (a == groups_per_shard * seq) {  // last group of this shard for this step
        const u32 t = __hip_atomic_fetch_add(c->top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
        if (t == (u32)kShards * seq) st_through(c->done, seq);
      }
    }
  }
  if (acc == 0x12345u) sink[wave] = acc;
}

// caller's stream, one per step: release step `seq`, wait (bounded) until the server says it is complete
__global__ void k_ring_wait(Ctl* c, u32 seq, u64 timeout_ticks) {
  if (threadIdx.x == 0) {
    st_through(c->doorbell, seq);
    const u64 t0 = now_ticks();
    while (ld_fresh(c->done) < seq) {
      if (now_ticks() - t0 > timeout_ticks) {
        atomicAdd(c->timeouts, 1u);
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
  }
}
__global__ void k_ring(Ctl* c, u32 seq) {
  if (threadIdx.x == 0) st_through(c->doorbell, seq);
}
__global__ void k_wait(Ctl* c, u32 seq, u64 timeout_ticks) {
  if (threadIdx.x == 0) {
    const u64 t0 = now_ticks();
    while (ld_fresh(c->done) < seq) {
      if (now_ticks() - t0 > timeout_ticks) {
        atomicAdd(c->timeouts, 1u);
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
  }
}
__global__ void k_stop(Ctl* c) {
  if (threadIdx.x == 0) st_through(c->stop, 1u);
}
// the same body as one plain launch per step
__global__ void k_step(const u32* actions, u32x4* out, u32* sink, u32 spin, u32 quads) {
  const u32 lane = threadIdx.x & 63u, waves = blockDim.x >> 6;
  const u32 wave = blockIdx.x * waves + (threadIdx.x >> 6);
  const u32 acc = body(actions[wave], spin, quads, out, wave, lane, false);
  if (acc == 0x12345u) sink[wave] = acc;
}

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static bool wait_stream(hipStream_t s, double seconds) {  // host wait with a deadline
  const double t0 = now_s();
  while (hipStreamQuery(s) == hipErrorNotReady) {
    if (now_s() - t0 > seconds) return false;
    std::this_thread::sleep_for(std::chrono::microseconds(50));
  }
  return true;
}

int main(int argc, char** argv) {
  const int K = argc > 1 ? atoi(argv[1]) : 3000;
  int lo = 0, hi = 0;
  CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  hipStream_t s_srv, s_call;
  CK(hipStreamCreateWithPriority(&s_srv, hipStreamNonBlocking, lo));  // lowest priority: a queue of its own, and the caller's
  CK(hipStreamCreateWithFlags(&s_call, hipStreamNonBlocking));        // kernels are dispatched ahead of it
  printf("stream priority range: lowest %d, highest %d\n", lo, hi);
  Ctl* c;
  CK(hipMalloc(&c, sizeof(Ctl)));
  const u32 max_waves = 8192;
  u32 *actions, *sink;
  u32x4* out;
  const u32 planes = 4, max_quads = 8;
  CK(hipMalloc(&actions, (size_t)planes * max_waves * 4));
  CK(hipMemset(actions, 1, (size_t)planes * max_waves * 4));
  CK(hipMalloc(&sink, max_waves * 4));
  CK(hipMalloc(&out, (size_t)max_waves * max_quads * 64 * 16));
  const u64 tmo = 5000000ull;  // 50 ms of 100 MHz ticks
  struct Geo { int groups, threads; };
  const Geo geos[] = {{2048, 64}, {512, 256}, {256, 512}, {128, 1024}, {4096, 64}, {1024, 256}, {256, 1024}};
  struct Body { u32 spin, quads; const char* what; };
  const Body bodies[] = {{0, 0, "empty"}, {0, 4, "4 KB/wave out"}, {1500, 0, "1500-deep chain"}, {1500, 4, "chain + 4 KB/wave"}, {4000, 3, "4000-deep chain + 3 KB/wave"}};
  printf("%-14s %-28s %10s %10s %10s %10s %8s\n", "geometry", "body", "launch/us", "ring+wait", "ring,wait", "memops", "timeouts");
  for (const Geo& g : geos) {
    if (g.groups % kShards) continue;
    for (const Body& b : bodies) {
      // plain launches of the same body
      for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(k_step, dim3(g.groups), dim3(g.threads), 0, s_call, actions, out, sink, b.spin, b.quads);
      CK(hipStreamSynchronize(s_call));
      double t0 = now_s();
      for (int i = 0; i < K; ++i) hipLaunchKernelGGL(k_step, dim3(g.groups), dim3(g.threads), 0, s_call, actions, out, sink, b.spin, b.quads);
      CK(hipStreamSynchronize(s_call));
      const double us_launch = (now_s() - t0) / K * 1e6;
      double us_mode[3] = {-1, -1, -1};
      u32 timeouts = 0;
      for (int mode = 0; mode < 3; ++mode) {
        CK(hipMemset(c, 0, sizeof(Ctl)));
        CK(hipDeviceSynchronize());
        const u32 warm = 200, total = warm + K;
        hipLaunchKernelGGL(k_server, dim3(g.groups), dim3(g.threads), 0, s_srv, c, actions, planes, out, sink, b.spin, b.quads, total,
                           (u32)(g.groups / kShards), tmo);
        CK(hipGetLastError());
        bool ok = true;
        double t1 = 0;
        for (u32 seq = 1; seq <= total && ok; ++seq) {
          if (seq == warm + 1) {
            ok = wait_stream(s_call, 2.0);
            t1 = now_s();
          }
          if (mode == 0) {
            hipLaunchKernelGGL(k_ring_wait, dim3(1), dim3(64), 0, s_call, c, seq, tmo);
          } else if (mode == 1) {  // a tick whose policy kernel rings at its end and whose consumer waits at its head
            hipLaunchKernelGGL(k_ring, dim3(1), dim3(64), 0, s_call, c, seq);
            hipLaunchKernelGGL(k_wait, dim3(1), dim3(64), 0, s_call, c, seq, tmo);
          } else {  // the command processor rings and waits: no wave at all on the caller's side
            if (hipStreamWriteValue32(s_call, c->doorbell, seq, 0) != hipSuccess ||
                hipStreamWaitValue32(s_call, c->done, seq, hipStreamWaitValueGte, 0xffffffffu) != hipSuccess) {
              (void)hipGetLastError();
              ok = false;
            }
          }
        }
        if (ok) ok = wait_stream(s_call, 5.0);
        const double t2 = now_s();
        hipLaunchKernelGGL(k_stop, dim3(1), dim3(64), 0, s_call, c);
        const bool ended = wait_stream(s_srv, 2.0) && wait_stream(s_call, 2.0);
        u32 h_to = 0;
        CK(hipMemcpy(&h_to, c->timeouts, 4, hipMemcpyDeviceToHost));
        timeouts += h_to;
        if (!ended) {
          printf("server did not end within its deadline: giving up\n");
          return 3;
        }
        if (ok && h_to == 0) us_mode[mode] = (t2 - t1) / K * 1e6;
      }
      char gname[32];
      snprintf(gname, sizeof gname, "%d x %d", g.groups, g.threads);
      printf("%-14s %-28s %10.2f %10.2f %10.2f %10.2f %8u\n", gname, b.what, us_launch, us_mode[0], us_mode[1], us_mode[2], timeouts);
      fflush(stdout);
    }
  }
  return 0;
}
