// ce_selfdrive_kernels.hip — gfx950 kernels for the continuous-state merge domain
// (environments/self_driving_car_accelerate.py) with the SelfdriveContractDistprop transfer
// (contract/contract_list.py:69-102) and the wrapper's reward application / theta sampling
// (environments/two_stage_train.py:62-121,159-187).
//
// The per-env state is a few dozen doubles and the step is a short, branchy float64 recurrence
// over n <= 10 cars, so here ONE LANE owns one env (64 envs per wavefront), templated on n so
// that the car table lives in registers.  All arithmetic is float64 in the reference's order of
// operations (compiled with -ffp-contract=off); Python's min()/max() tie behaviour is kept.
// The two MT19937 generators of the reference process (np.random for theta, `random` for the
// start positions) are only touched by reset and are advanced in place in HBM by the owning lane.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ce_device.h"

namespace ce {

typedef uint32_t u32;
typedef uint64_t u64;

#define DEVINL __device__ __forceinline__

// ---- MT19937 advanced in place in global memory by one lane (reset only) ----
DEVINL u32 mtg_next(u32* mt) {
  u32 pos = mt[kMtN];
  if (pos >= (u32)kMtN) {
    for (int i = 0; i < kMtN; ++i) {
      const u32 a = mt[i], b = mt[(i + 1) % kMtN], c = mt[(i + kMtM) % kMtN];
      const u32 y = (a & 0x80000000u) | (b & 0x7fffffffu);
      mt[i] = c ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    pos = 0;
  }
  u32 y = mt[pos];
  mt[kMtN] = pos + 1;
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}
DEVINL double mtg_double(u32* mt) {
  const u32 a = mtg_next(mt) >> 5, b = mtg_next(mt) >> 6;
  return ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
}

DEVINL double py_min2(double x, double y) { return y < x ? y : x; }  // min([x, y])
DEVINL double py_max2(double x, double y) { return y > x ? y : x; }  // max([x, y])

template <int N> struct Cars {
  double pos[N], vel[N], dtf[N];
  bool done[N];
  bool done_all;
  int ncross;
  int cross[N];
  double transfers;
};

template <int N> DEVINL void sd_load(const SdParams& p, u32 e, Cars<N>& c) {
  const double* s = p.sd_state + (size_t)e * CE_SD_STATE_DOUBLES(N);
#pragma unroll
  for (int a = 0; a < N; ++a) {
    c.pos[a] = s[a];
    c.vel[a] = s[N + a];
    c.dtf[a] = s[2 * N + a];
    c.done[a] = s[3 * N + a] != 0.0;
    c.cross[a] = (int)s[4 * N + 2 + a];
  }
  c.done_all = s[4 * N] != 0.0;
  c.ncross = (int)s[4 * N + 1];
  c.transfers = s[5 * N + 2];
}
template <int N> DEVINL void sd_store(const SdParams& p, u32 e, const Cars<N>& c) {
  double* s = p.sd_state + (size_t)e * CE_SD_STATE_DOUBLES(N);
#pragma unroll
  for (int a = 0; a < N; ++a) {
    s[a] = c.pos[a];
    s[N + a] = c.vel[a];
    s[2 * N + a] = c.dtf[a];
    s[3 * N + a] = c.done[a] ? 1.0 : 0.0;
    s[4 * N + 2 + a] = a < c.ncross ? (double)c.cross[a] : -1.0;
  }
  s[4 * N] = c.done_all ? 1.0 : 0.0;
  s[4 * N + 1] = (double)c.ncross;
  s[5 * N + 2] = c.transfers;
}

// obs of the acting cars (…accelerate.py:241-247) + [theta, 0] (two_stage_train.py:113-117)
template <int N> DEVINL void sd_write_obs(const SdParams& p, u32 e, const Cars<N>& c, const bool* active, double theta, double last) {
  constexpr int L = 2 * N + 7;
  const double nan = __longlong_as_double(0x7ff8000000000000ll);
#pragma unroll
  for (int k = 0; k < N; ++k) {
    double* ob = p.obs_f64 + ((size_t)e * N + k) * L;
    if (!active[k]) {
      for (int i = 0; i < L; ++i) ob[i] = nan;
      continue;
    }
    ob[0] = c.pos[k];
    ob[1] = c.vel[k];
#pragma unroll
    for (int i = 0; i < N; ++i) ob[2 + i] = c.pos[i] - c.pos[k];
#pragma unroll
    for (int i = 0; i < N; ++i) ob[2 + N + i] = c.vel[i];
    ob[2 + 2 * N] = c.pos[0] > 0 ? 1.0 : 0.0;
    ob[3 + 2 * N] = c.pos[k] > 0 ? 1.0 : 0.0;
    ob[4 + 2 * N] = last;
    ob[5 + 2 * N] = theta;
    ob[6 + 2 * N] = 0.0;
  }
}

template <int N> DEVINL void sd_zero_metrics(const SdParams& p, u32 e) {
  for (int k = 0; k < (int)CE_MF_COUNT(N); ++k) p.f64_metrics[(size_t)e * CE_MF_COUNT(N) + k] = 0.0;
  for (int k = 0; k < (int)CE_MI_COUNT(N); ++k) p.int_metrics[(size_t)e * CE_MI_COUNT(N) + k] = 0;
}

// SelfAcceleratingCarEnv.reset (:49-79) + SeparateContractSubgameStage.reset theta draw
template <int N> DEVINL void sd_reset_env(const SdParams& p, u32 e, Cars<N>& c, double& theta) {
  u32* np_mt = p.rng + (size_t)e * CE_RNG_WORDS_SELFDRIVE;
  u32* py_mt = np_mt + CE_RNG_WORDS_GRID;
  const double low = p.low_bound;
#pragma unroll
  for (int a = 0; a < N; ++a) {
    const double u = mtg_double(py_mt);
    if (a == 0) {
      c.pos[a] = u * low / 2 + low / 2;
      c.vel[a] = p.start_vel_ambulance;
    } else {
      c.pos[a] = u * low / 16 + low * 3 / 16;
      c.vel[a] = p.start_vel;
    }
    c.done[a] = false;
    c.dtf[a] = -1.0;
    c.cross[a] = -1;
  }
  c.done_all = false;
  c.ncross = 0;
  c.transfers = 0.0;
  if (p.flags & CE_FLAG_EXTERNAL_THETA) {  // the caller owns the theta buffer: a reset neither draws nor changes it
    theta = p.theta[e];
  } else if (p.contract == CE_CONTRACT_NONE) {
    theta = 0.0;
  } else {
    const double u0 = mtg_double(np_mt);
    if (u0 > p.null_prob) {
      const double u1 = mtg_double(np_mt);
      theta = p.contract_low + (p.contract_high - p.contract_low) * u1;
    } else {
      theta = p.contract_low;
    }
  }
}

template <int N> __global__ void k_sd_construct(SdParams p) {
  const u32 e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= p.E) return;
  if (p.mask && p.mask[e] == 0) return;
  Cars<N> c;
#pragma unroll
  for (int a = 0; a < N; ++a) {  // __init__ :29-35, no RNG use
    c.pos[a] = p.low_bound;
    c.vel[a] = p.start_vel;
    c.done[a] = false;
    c.dtf[a] = -1.0;
    c.cross[a] = -1;
  }
  c.done_all = false;
  c.ncross = 0;
  c.transfers = 0.0;
  sd_store(p, e, c);
  sd_zero_metrics<N>(p, e);
  p.theta[e] = 0.0;
  p.done[e] = 0;
  p.error_flags[e] = 0;
}

template <int N> __global__ void k_sd_reset(SdParams p) {
  const u32 e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= p.E) return;
  if (p.mask && p.mask[e] == 0) return;
  Cars<N> c;
  double theta;
  sd_reset_env(p, e, c, theta);
  bool active[N];
#pragma unroll
  for (int a = 0; a < N; ++a) active[a] = true;
  sd_write_obs(p, e, c, active, theta, 0.0);
  sd_store(p, e, c);
  sd_zero_metrics<N>(p, e);
  p.theta[e] = theta;
  p.done[e] = 0;
  p.sd_info[(size_t)e * 2] = p.sd_info[(size_t)e * 2 + 1] = 0.0;
  p.error_flags[e] = 0;  // a reset starts a clean episode (faults are sticky until then)
#pragma unroll
  for (int a = 0; a < N; ++a) {
    p.reward[(size_t)e * N + a] = 0.0;
    p.base_reward[(size_t)e * N + a] = 0;
    p.info[((size_t)e * N + a) * 2] = 0;
    p.info[((size_t)e * N + a) * 2 + 1] = 0;
    p.done_agents[(size_t)e * N + a] = 0;
  }
}

// ========================================================================================
// The step kernels: agent-parallel lanes.  An env is a group of G = 2^ceil(log2 n) adjacent lanes (lane k of the group =
// car k), 64 / G envs per wavefront — at n = 4 the 32 768 envs of BASELINE config 5 are 2 048 waves instead of the 512
// of a lane-per-env mapping.  A car's position / velocity / done flag live in its own lane; whatever the reference does
// across cars (merge-order bookkeeping, the ordered consistency pass, the Distprop transfer) is a short group-uniform
// loop that fetches the other cars with ds_bpermute.  The observation block of a wave's envs is contiguous in HBM
// ([env][car][2n+7] doubles): rows are assembled in LDS and leave as full-width 16-byte stores, 1 KB per instruction.
// k_sd_rollout (ce_rollout_fused) keeps the cars in registers across the steps of a launch.
// ========================================================================================
template <int N> struct SdGeo {
  static constexpr int G = N <= 1 ? 1 : N <= 2 ? 2 : N <= 4 ? 4 : N <= 8 ? 8 : 16;  // lanes per env
  static constexpr int EPW = 64 / G;                                                // envs per wave
  static constexpr int L = 2 * N + 7;                                               // doubles per observation row
  static constexpr int OBS_DOUBLES = EPW * N * L;                                   // a wave's observation block
};

DEVINL u32 sd_lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
DEVINL u32 sd_bperm(u32 v, u32 src_lane) { return (u32)__builtin_amdgcn_ds_bpermute((int)(src_lane << 2), (int)v); }
DEVINL double sd_shfl(double v, u32 src_lane) {
  const u64 b = (u64)__double_as_longlong(v);
  const u32 lo = sd_bperm((u32)b, src_lane), hi = sd_bperm((u32)(b >> 32), src_lane);
  return __longlong_as_double((long long)(((u64)hi << 32) | lo));
}
DEVINL void sd_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
DEVINL u32 sd_temper(u32 y) {
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}
DEVINL double sd_words_to_double(u32 w0, u32 w1) {
  return ((double)(sd_temper(w0) >> 5) * 67108864.0 + (double)(sd_temper(w1) >> 6)) / 9007199254740992.0;
}

// one env's cars as seen by lane k of its group
template <int N> struct Car {
  double pos, vel, dtf;  // own car
  bool done;
  // group-uniform (every lane of the group holds the same value)
  u64 crossed;  // merge order so far: car index of entry i in bits [4i, 4i+4)
  u32 ncross;
  bool done_all;
  double transfers, theta;
};
DEVINL u32 sd_nib(u64 list, u32 i) { return (u32)(list >> (4 * i)) & 15u; }

// where a step's outputs go (see StepOutDirect / StepOutPlane of the grid kernels)
struct SdOut {
  CE_GPTR(double) obs_f64;
  CE_GPTR(int32_t) base_reward;
  CE_GPTR(double) reward;
  CE_GPTR(uint8_t) done;
  CE_GPTR(uint8_t) done_agents;
  CE_GPTR(uint8_t) info;
  CE_GPTR(double) sd_info;
};

template <int N> struct SdLane {
  u32 lane, k, gb, e;  // lane id, car index in the group, first lane of the group, env index
  bool live, is_car;   // env inside the launch range; k < N
  u32 e0, live_envs;   // first env of the wave, envs of the wave inside the range
};

template <int N> DEVINL void sd_load_state(const SdParams& p, const SdLane<N>& ln, Car<N>& c) {
  typedef SdGeo<N> Gm;
  const auto s = p.sd_state + (size_t)ln.e * CE_SD_STATE_DOUBLES(N);
  const u32 k = ln.is_car ? ln.k : 0u;
  c.pos = s[k];
  c.vel = s[N + k];
  c.dtf = s[2 * N + k];
  c.done = s[3 * N + k] != 0.0;
  const double cr = s[4 * N + 2 + k];
  c.done_all = s[4 * N] != 0.0;
  c.ncross = (u32)s[4 * N + 1];
  c.transfers = s[5 * N + 2];
  c.theta = p.theta[ln.e];
  const u32 mine = cr >= 0.0 ? (u32)cr : 0u;
  c.crossed = 0;
#pragma unroll
  for (int i = 0; i < N; ++i) c.crossed |= (u64)(sd_bperm(mine, ln.gb + i) & 15u) << (4 * i);
  if (Gm::G == 1) c.crossed = mine;
}
template <int N> DEVINL void sd_store_state(const SdParams& p, const SdLane<N>& ln, const Car<N>& c, bool theta_too) {
  if (!ln.live) return;
  const auto s = p.sd_state + (size_t)ln.e * CE_SD_STATE_DOUBLES(N);
  if (ln.is_car) {
    s[ln.k] = c.pos;
    s[N + ln.k] = c.vel;
    s[2 * N + ln.k] = c.dtf;
    s[3 * N + ln.k] = c.done ? 1.0 : 0.0;
    s[4 * N + 2 + ln.k] = ln.k < c.ncross ? (double)sd_nib(c.crossed, ln.k) : -1.0;
  }
  if (ln.k == 0) {
    s[4 * N] = c.done_all ? 1.0 : 0.0;
    s[4 * N + 1] = (double)c.ncross;
    s[5 * N + 2] = c.transfers;
    if (theta_too) p.theta[ln.e] = c.theta;
  }
}

// Writes one observation row per car into the wave's LDS block (acting cars; the others get NaN rows) and streams the
// block out: the envs of a wave are adjacent, so the block is one contiguous span of HBM.  Called by the whole wave;
// `emit` (group-uniform) = this env's rows are (re)written — the rows of the other envs of the wave are left alone.
template <int N> DEVINL void sd_emit_obs(const SdLane<N>& ln, const Car<N>& c, bool emit, bool active, double last,
                                         double* stage, CE_GPTR(double) obs_base) {
  typedef SdGeo<N> Gm;
  constexpr int L = Gm::L;
  const double nan = __longlong_as_double(0x7ff8000000000000ll);
  const u32 slot = ln.lane / Gm::G;  // env slot of the wave
  double* ob = stage + (size_t)(slot * N + ln.k) * L;
  const double p0 = sd_shfl(c.pos, ln.gb);
  // which env slots of the wave are written: bit per slot, taken from the first lane of each group
  const u64 em = __builtin_amdgcn_ballot_w64(emit && ln.live && ln.k == 0);
  sd_wave_sync();
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const double pi = sd_shfl(c.pos, ln.gb + i), vi = sd_shfl(c.vel, ln.gb + i);
    if (ln.is_car) {
      ob[2 + i] = active ? pi - c.pos : nan;
      ob[2 + N + i] = active ? vi : nan;
    }
  }
  if (ln.is_car) {
    ob[0] = active ? c.pos : nan;
    ob[1] = active ? c.vel : nan;
    ob[2 + 2 * N] = active ? (p0 > 0 ? 1.0 : 0.0) : nan;
    ob[3 + 2 * N] = active ? (c.pos > 0 ? 1.0 : 0.0) : nan;
    ob[4 + 2 * N] = active ? last : nan;
    ob[5 + 2 * N] = active ? c.theta : nan;
    ob[6 + 2 * N] = active ? 0.0 : nan;
  }
  sd_wave_sync();
  // copy-out: consecutive lanes write consecutive doubles (512 bytes per instruction), streaming stores
  const auto dst = obs_base + (size_t)ln.e0 * (N * L);
#pragma unroll
  for (u32 r = 0; r < (u32)(Gm::OBS_DOUBLES + 63) / 64; ++r) {
    const u32 j = r * 64 + ln.lane;
    const u32 js = j / (u32)(N * L);  // env slot double j belongs to
    if (j < (u32)Gm::OBS_DOUBLES && ((em >> (js * Gm::G)) & 1ull)) __builtin_nontemporal_store(stage[j], dst + j);
  }
}

// ---- a stream that runs over the end of its generation, handled by the whole wave ----
// A reset draws 2n words of the CPython stream and 2 or 4 of numpy's.  Almost always the window lies inside the current
// generation and the cars fetch their words themselves (sd_reset_group); once per 624 words it does not, and the
// generation has to be regenerated.  Left to one lane on global memory (624 dependent read-modify-write rounds) that
// took ~0.3 ms — with 32 768 envs a few waves of EVERY launch hit it, and the steady-state step time was theirs
// (55 us per step instead of 12).  Here the wave stages the env's 624 words in its LDS block (free between two
// observation emits), twists them with 64 lanes in three dependent chunks like the grid kernels, and writes them back.
DEVINL u32 sd_mix(u32 a, u32 b, u32 c) {
  const u32 y = (a & 0x80000000u) | (b & 0x7fffffffu);
  return c ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}
__device__ __noinline__ void sd_twist_lds(u32* mt, u32 lane) {
  sd_wave_sync();
  u32 v[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {  // words 0..226 read old words only
    const u32 i = 64 * r + lane, ic = i < 227 ? i : 0;
    v[r] = sd_mix(mt[ic], mt[ic + 1], mt[ic + kMtM]);
  }
  sd_wave_sync();
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (64 * r + lane < 227) mt[64 * r + lane] = v[r];
  sd_wave_sync();
#pragma unroll
  for (int r = 0; r < 4; ++r) {  // 227..453 read chunk-1 results
    const u32 i = 227 + 64 * r + lane, ic = i < 454 ? i : 227;
    v[r] = sd_mix(mt[ic], mt[ic + 1], mt[ic - 227]);
  }
  sd_wave_sync();
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (227 + 64 * r + lane < 454) mt[227 + 64 * r + lane] = v[r];
  sd_wave_sync();
#pragma unroll
  for (int r = 0; r < 3; ++r) {  // 454..622 read chunk-2 results
    const u32 i = 454 + 64 * r + lane, ic = i < 623 ? i : 454;
    v[r] = sd_mix(mt[ic], mt[ic + 1], mt[ic - 227]);
  }
  sd_wave_sync();
#pragma unroll
  for (int r = 0; r < 3; ++r)
    if (454 + 64 * r + lane < 623) mt[454 + 64 * r + lane] = v[r];
  sd_wave_sync();
  if (lane == 0) mt[623] = sd_mix(mt[623], mt[0], mt[396]);
  sd_wave_sync();
}
// SelfAcceleratingCarEnv.reset (:49-79) + the wrapper's theta draw for the groups flagged `go`, lane-parallel: the 2n
// words of the CPython `random` stream one reset consumes are fetched by the cars themselves (and the <= 4 numpy words
// by lane 0) unless the window runs over the end of the generation — then the whole wave serves that group's stream
// (sd_stage_stream / sd_twist_lds above).  Called by the whole wave; `lds` = the wave's staging block (no observation
// rows pending in it).
template <int N> DEVINL void sd_reset_group(const SdParams& p, const SdLane<N>& ln, Car<N>& c, bool go, u32* lds) {
  const u32 e = ln.e;
  u32* np_mt = (u32*)p.rng + (size_t)e * CE_RNG_WORDS_SELFDRIVE;
  u32* py_mt = np_mt + CE_RNG_WORDS_GRID;
  const double low = p.low_bound;
  double u = 0.0, theta = c.theta;
  const bool draws_theta = !(p.flags & CE_FLAG_EXTERNAL_THETA) && p.contract != CE_CONTRACT_NONE;
  bool py_fast = true, np_fast = true;
  if (go) {
    const u32 ppos = py_mt[kMtN];
    py_fast = ppos + 2u * N <= (u32)kMtN;
    const u32 npos = np_mt[kMtN];
    np_fast = !draws_theta || npos + 4u <= (u32)kMtN;
    if (py_fast) {
      if (ln.is_car) u = sd_words_to_double(py_mt[ppos + 2 * ln.k], py_mt[ppos + 2 * ln.k + 1]);
    }
    // everything that writes a stream position: the group's first lane
    if (ln.k == 0) {
      if (py_fast) py_mt[kMtN] = ppos + 2u * N;
      if (p.flags & CE_FLAG_EXTERNAL_THETA) {  // the caller owns the theta buffer: a reset neither draws nor changes it
        theta = p.theta[e];
      } else if (p.contract == CE_CONTRACT_NONE) {
        theta = 0.0;
      } else if (np_fast) {
        const double u0 = sd_words_to_double(np_mt[npos], np_mt[npos + 1]);
        if (u0 > p.null_prob) {
          theta = p.contract_low + (p.contract_high - p.contract_low) * sd_words_to_double(np_mt[npos + 2], np_mt[npos + 3]);
          np_mt[kMtN] = npos + 4;
        } else {
          theta = p.contract_low;
          np_mt[kMtN] = npos + 2;
        }
      }
    }
  }
  // The rare groups whose window crosses a generation end, one (group, stream) at a time, served by the whole wave.
  // genrand semantics: words are consumed in order and the generation is regenerated exactly when a word past its end
  // is consumed — the window's first `tail` words come from the old generation, the rest from the new one.  Lane i
  // holds word i of the window; ONE code instance serves both streams (the twist is 40 instructions x 3 chunks).
  const u64 slow_py = __builtin_amdgcn_ballot_w64(go && ln.k == 0 && !py_fast);
  const u64 slow_np = __builtin_amdgcn_ballot_w64(go && ln.k == 0 && !np_fast);
  for (u64 it = slow_py | slow_np; it; it &= it - 1) {
    const u32 L = (u32)__builtin_ctzll(it);  // first lane of the group
    const u32 eL = (u32)__builtin_amdgcn_readlane((int)e, (int)L);
    const bool mine = ln.gb == L;
#pragma unroll 1
    for (u32 which = 0; which < 2; ++which) {  // 0: CPython `random` (2n words), 1: np.random (2 or 4 words)
      if ((((which ? slow_np : slow_py) >> L) & 1ull) == 0) continue;
      u32* mt = (u32*)p.rng + (size_t)eL * CE_RNG_WORDS_SELFDRIVE + (which ? 0u : (u32)CE_RNG_WORDS_GRID);
      const u32 want = which ? 4u : 2u * N;
      sd_wave_sync();
      for (u32 k2 = ln.lane; k2 < (u32)kMtN; k2 += 64) lds[k2] = mt[k2];
      const u32 pos = (u32)__builtin_amdgcn_readfirstlane((int)mt[kMtN]);
      const u32 tail = (u32)kMtN - pos;  // words of the window that lie in the old generation
      sd_wave_sync();
      u32 x = (ln.lane < want && ln.lane < tail) ? lds[pos + ln.lane] : 0u;
      u32 consumed = want;
      if (which && tail >= 2u) {  // u0 is decided by old words: maybe only two words are consumed, and no twist
        const double u0 = sd_words_to_double((u32)__builtin_amdgcn_readlane((int)x, 0), (u32)__builtin_amdgcn_readlane((int)x, 1));
        consumed = u0 > p.null_prob ? 4u : 2u;
      }
      const bool twist = consumed > tail;
      if (twist) {
        sd_twist_lds(lds, ln.lane);
        if (ln.lane < want && ln.lane >= tail) x = lds[ln.lane - tail];
        sd_wave_sync();
        for (u32 k2 = ln.lane; k2 < (u32)kMtN; k2 += 64) mt[k2] = lds[k2];
      }
      if (which) {
        const u32 x0 = (u32)__builtin_amdgcn_readlane((int)x, 0), x1 = (u32)__builtin_amdgcn_readlane((int)x, 1);
        const u32 x2 = (u32)__builtin_amdgcn_readlane((int)x, 2), x3 = (u32)__builtin_amdgcn_readlane((int)x, 3);
        const double u0 = sd_words_to_double(x0, x1);
        consumed = u0 > p.null_prob ? 4u : 2u;  // (tail < 2: decided by the new words; a twist was due either way)
        const double th = consumed == 4u ? p.contract_low + (p.contract_high - p.contract_low) * sd_words_to_double(x2, x3)
                                         : p.contract_low;
        if (mine) theta = th;
      } else {
        const u32 w0 = sd_bperm(x, 2u * ln.k), w1 = sd_bperm(x, 2u * ln.k + 1u);
        if (mine && ln.is_car) u = sd_words_to_double(w0, w1);
      }
      if (ln.lane == 0) mt[kMtN] = twist ? consumed - tail : pos + consumed;
      sd_wave_sync();
    }
  }
  theta = sd_shfl(theta, ln.gb);  // (every lane takes part in the permute; only `go` groups use the value)
  if (go) {
    if (ln.k == 0) {
      c.pos = u * low / 2 + low / 2;
      c.vel = p.start_vel_ambulance;
    } else {
      c.pos = u * low / 16 + low * 3 / 16;
      c.vel = p.start_vel;
    }
    c.done = false;
    c.dtf = -1.0;
    c.crossed = 0;
    c.ncross = 0;
    c.done_all = false;
    c.transfers = 0.0;
    c.theta = theta;
  }
}

// One step of the groups of a wave.  `act` = this car's acceleration, `active_in` < 0: acting = not done (what RLlib
// sends), else the caller's flag.  Returns with the cars updated (and reset where an episode ended under AUTO_RESET).
template <int N, bool FUSED>
DEVINL void sd_step_core(const SdParams& p, const SdOut& out, const SdLane<N>& ln, Car<N>& c, float act, int active_in,
                         double* stage, u32& fault, bool& did_reset) {
  typedef SdGeo<N> Gm;
  constexpr u32 G = Gm::G;
  const u32 k = ln.k, gb = ln.gb;
  const double high = p.high_bound;
  const double nan = __longlong_as_double(0x7ff8000000000000ll);
  const u64 gmask = G == 64 ? ~0ull : ((1ull << G) - 1ull);
  auto group_mask = [&](bool pred) -> u32 { return (u32)((__builtin_amdgcn_ballot_w64(pred) >> gb) & gmask); };

  const bool active = ln.is_car && (active_in < 0 ? !c.done : active_in != 0);
  const u32 am = group_mask(active);
  // the reference raises on a step after __all__ (:154-167, collision_check_all is undefined): nothing is stepped
  const bool skip = c.done_all || am == 0 || !ln.live;
  if (c.done_all || am == 0) fault |= CE_FAULT_STEP_AFTER_DONE;
  const bool act_ok = active && !skip;

  // velocities / tentative positions (:172-180)
  double new_pos = c.pos;
  if (act_ok) {
    const double a = (double)act;
    const double vmax = k == 0 ? 1.0 : 0.25;
    c.vel = py_max2(py_min2(py_max2(py_min2(a, 0.1), -0.1) + c.vel, vmax), 0.0);
    new_pos = c.vel + c.pos;
  }
  const bool jp = act_ok && c.pos < 0.0 && new_pos > 0.0;
  const u32 jm = group_mask(jp);
  // infos of the first acting key (:183-189): the ambulance's rank in the merge order so far, its recorded distance
  double amb_rank = (double)N;
#pragma unroll
  for (int i = 0; i < N; ++i)
    if ((u32)i < c.ncross && sd_nib(c.crossed, i) == 0) amb_rank = (double)(i + 1);
  const double dtf0 = sd_shfl(c.dtf, gb);
  double amb_dtf = dtf0 > -1 ? dtf0 : high - p.low_bound;
  // update_rel_rank (:110-125): the sort key is the id's 2nd character => agent index order
#pragma unroll
  for (int i = 0; i < N; ++i)
    if ((jm >> i) & 1u) {
      c.crossed |= (u64)i << (4 * c.ncross);
      c.ncross += 1;
    }
  // update_infos (:127-149); the dist_to_front quirks are kept: :143 subtracts a value from itself, :144 writes index n-1
  if (__builtin_amdgcn_ballot_w64(jp) != 0) {
    double d = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const double npi = sd_shfl(new_pos, gb + i);
      const bool acti = ((am >> i) & 1u) != 0;
      if ((u32)i != k && (!acti || npi > new_pos)) {
        if (!acti) {
          if (high - new_pos > d) d = high + 1 - new_pos;
        } else {
          if (npi - new_pos > d) d = npi - npi;
        }
      }
    }
    const u32 hi_jp = jm ? 31u - (u32)__builtin_clz(jm) : 0u;  // the last just-passed car in key order writes last
    const double d_last = sd_shfl(d, gb + hi_jp), d_a0 = sd_shfl(d, gb);
    if (jm != 0 && k == N - 1) c.dtf = d_last;
    if (jm & 1u) {  // :146-149 ambulance stats
#pragma unroll
      for (int i = 0; i < N; ++i)
        if ((u32)i < c.ncross && sd_nib(c.crossed, i) == 0) amb_rank = (double)(i + 1);
      amb_dtf = d_a0;
    }
  }
  // merge-order pass over consecutive pairs of the crossed list: crash test (collision_on, :81-90) or the ordered
  // consistency correction (:92-108; a corrected car is the front of the next pair)
  double eff = new_pos;  // positions the pass sees: new for acting cars, stored for the others
  bool crashed = false;
  u32 pre = 0;
  const bool collide = (p.flags & CE_FLAG_COLLISION_ON) != 0;
#pragma unroll
  for (int i = 0; i + 1 < N; ++i) {
    const bool valid = (u32)i + 1 < c.ncross && !skip;
    const u32 f = valid ? sd_nib(c.crossed, i) : 0u, b2 = valid ? sd_nib(c.crossed, i + 1) : 0u;
    const double pf = sd_shfl(eff, gb + f), pb = sd_shfl(eff, gb + b2);
    if (collide) {
      crashed = crashed || (valid && pf < pb);
    } else {
      const bool fix = valid && pf < pb && ((am >> f) & 1u) && ((am >> b2) & 1u);
      const double np2 = pf - 0.01;
      if (fix && k == b2) eff = np2;
      if (fix && np2 < 0) pre |= 1u << b2;
    }
  }
  double rew = -1.0;
  if (crashed) {  // :196-215 — the observation keeps the OLD positions, velocities are already updated
    amb_rank = (double)N;
    c.done_all = true;
    if (active) c.done = true;
    rew = -10000.0;
  } else if (!skip) {
    if (pre) {  // cars pushed back behind the line leave the merge order
      u64 nl = 0;
      u32 nc = 0;
#pragma unroll
      for (int i = 0; i < N; ++i)
        if ((u32)i < c.ncross) {
          const u32 a = sd_nib(c.crossed, i);
          if (!((pre >> a) & 1u)) {
            nl |= (u64)a << (4 * nc);
            nc += 1;
          }
        }
      c.crossed = nl;
      c.ncross = nc;
    }
    if (active) c.pos = eff;
    if (k == 0 && active) rew -= 99.0;  // the ambulance's higher penalty
    if (ln.is_car && c.pos > high) {
      c.pos = high + 1;  // sentinel, avoids overflow
      c.done = true;
    }
    c.done_all = group_mask(active && !c.done) == 0;
  }
  // ---- outputs ----
  sd_emit_obs<N>(ln, c, !skip, active, crashed ? 1.0 : 0.0, stage, out.obs_f64);
  const int32_t base_rew = active ? (int32_t)rew : 0;  // -1 / -100 / -10000
  // SelfdriveContractDistprop.compute_transfer on a0's observation (contract_list.py:69-102), applied in the
  // wrapper's order (two_stage_train.py:69-92): only on the step the ambulance passes the line
  if (p.contract == CE_CONTRACT_SELFDRIVE_DISTPROP && __builtin_amdgcn_ballot_w64((jm & 1u) != 0 && !skip) != 0) {
    const bool a0jp = (jm & 1u) != 0 && !skip;
    const double p0 = sd_shfl(c.pos, gb);
    const double rel = c.pos - p0;  // ob0[2 + k] (crash branch: the old positions); slots n, n+1 of the reference's loop
    const bool behind = a0jp && ln.is_car && k >= 1 && rel < 0;  // read velocities (>= 0): never "behind"
    const u32 bm = group_mask(behind);
    double sum_d = 0.0;  // python int 0 + floats, in index order
#pragma unroll
    for (int i = 1; i < N; ++i) {
      const double ri = sd_shfl(rel, gb + i);
      if ((bm >> i) & 1u) sum_d += -ri;
    }
    const double tval0 = c.theta * sum_d;  // (value, {behind car: share}) when any car is behind, else 0
    const bool tuple = a0jp && active && (k == 0 ? bm != 0 : !behind);
    const double tval = k == 0 ? tval0 : c.theta * rel;  // cars not behind pay theta * distance ahead back to a0
    double total = 0.0;
    if (a0jp) {
      // agent 0 first, then the others in index order: each pays its own transfer; a0 collects in index order
      if (k == 0) {
        if (tuple) rew -= tval0;
        total = tuple ? tval0 : 0.0;
      } else if (active) {
        if (behind && bm != 0) rew += tval0 * ((-rel) / sum_d);
        if (tuple) rew -= tval;
      }
    }
#pragma unroll
    for (int i = 1; i < N; ++i) {
      const double ti = sd_shfl(tuple ? tval : 0.0, gb + i);
      const bool is_t = ((group_mask(tuple) >> i) & 1u) != 0;
      if (a0jp && is_t) {
        if (k == 0) rew += ti * 1.0;
        total += ti;
      }
    }
    if (a0jp) c.transfers += sd_shfl(total, gb);
  }
  if (ln.live && !skip) {
    if (ln.is_car) {
      const size_t ea = (size_t)ln.e * N + k;
      out.base_reward[ea] = base_rew;
      out.reward[ea] = active ? rew : nan;
      out.info[2 * ea] = jp ? 1 : 0;
      out.info[2 * ea + 1] = crashed ? 1 : 0;
      out.done_agents[ea] = c.done ? 1 : 0;
    }
    if (k == 0) {
      out.done[ln.e] = c.done_all ? 1 : 0;
      out.sd_info[(size_t)ln.e * 2] = amb_rank;
      out.sd_info[(size_t)ln.e * 2 + 1] = amb_dtf;
      p.f64_metrics[(size_t)ln.e * CE_MF_COUNT(N) + CE_MF_TRANSFERS] = c.transfers;
      if (c.done_all) p.final_f64_metrics[(size_t)ln.e * CE_MF_COUNT(N) + CE_MF_TRANSFERS] = c.transfers;
    }
  }
  // in-launch auto-reset: the terminal step's rewards / infos / dones stay, the observation becomes the reset one
  const bool go = !skip && c.done_all && (p.flags & CE_FLAG_AUTO_RESET) != 0;
  if (__builtin_amdgcn_ballot_w64(go) != 0) {
    sd_wave_sync();  // the terminal observation has left the staging block
    sd_reset_group<N>(p, ln, c, go && ln.live, (u32*)stage);
    sd_emit_obs<N>(ln, c, go, ln.is_car, 0.0, stage, out.obs_f64);  // every car of a reset env is acting
    if (go && ln.live && k == 0) p.f64_metrics[(size_t)ln.e * CE_MF_COUNT(N) + CE_MF_TRANSFERS] = 0.0;
    did_reset = did_reset || go;
  }
}

template <int N> DEVINL SdLane<N> sd_lane_setup(u32 env_first, u32 env_end) {
  typedef SdGeo<N> Gm;
  SdLane<N> ln;
  ln.lane = sd_lane_id();
  ln.k = ln.lane % Gm::G;
  ln.gb = ln.lane - ln.k;
  ln.e0 = env_first + blockIdx.x * Gm::EPW;
  const u32 e = ln.e0 + ln.lane / Gm::G;
  ln.live = e < env_end;
  ln.e = ln.live ? e : env_end - 1;  // out-of-range groups shadow the last env (loads stay in bounds, nothing is stored)
  ln.is_car = ln.k < (u32)N;
  ln.live_envs = env_end - ln.e0 < (u32)Gm::EPW ? env_end - ln.e0 : (u32)Gm::EPW;
  return ln;
}

template <int N> __global__ __launch_bounds__(64) void k_sd_step(SdParams p) {
  typedef SdGeo<N> Gm;
  __shared__ __attribute__((aligned(16))) double stage[Gm::OBS_DOUBLES];
  const u32 env_end = p.env_first + (p.env_count ? p.env_count : p.E - p.env_first);
  if (p.env_first + blockIdx.x * Gm::EPW >= env_end) return;
  const SdLane<N> ln = sd_lane_setup<N>(p.env_first, env_end);
  Car<N> c;
  sd_load_state<N>(p, ln, c);
  const size_t ea = (size_t)ln.e * N + (ln.is_car ? ln.k : 0u);
  const float act = p.actions[ea];
  const int active_in = p.active ? (int)p.active[ea] : -1;
  const SdOut out = {p.obs_f64, p.base_reward, p.reward, p.done, p.done_agents, p.info, p.sd_info};
  u32 fault = 0;
  bool did_reset = false;
  sd_step_core<N, false>(p, out, ln, c, act, active_in, stage, fault, did_reset);
  if (fault) {
    if (ln.live && ln.k == 0) p.error_flags[ln.e] |= fault;
    return;  // nothing was stepped
  }
  sd_store_state<N>(p, ln, c, did_reset);
}

// Fused multi-step rollout (ce_rollout_fused): the cars stay in registers for num_steps steps, every step reads its
// own action plane and writes its own outputs (plane (plane0 + s) mod num_planes of the trajectory arrays); acting =
// not done.  The two MT19937 streams stay in HBM: only resets touch them.
template <int N> __global__ __launch_bounds__(64) void k_sd_rollout(SdParams p, const RolloutArgs ra) {
  typedef SdGeo<N> Gm;
  __shared__ __attribute__((aligned(16))) double stage[Gm::OBS_DOUBLES];
  if (ra.env_first + blockIdx.x * Gm::EPW >= ra.env_end) return;
  const SdLane<N> ln = sd_lane_setup<N>(ra.env_first, ra.env_end);
  Car<N> c;
  sd_load_state<N>(p, ln, c);
  const size_t ea = (size_t)ln.e * N + (ln.is_car ? ln.k : 0u);
  const auto acts = (CE_GPTR(const float))ra.actions;
  float act = acts[ea];
  u32 fault = 0, pl = ra.plane0;
  bool any_reset = false;
  for (u32 s = 0; s < ra.num_steps; ++s) {
    const u32 sn = s + 1 < ra.num_steps ? s + 1 : s;
    const float act_next = acts[(size_t)sn * ra.action_plane + ea];  // in flight while this step runs
    const SdOut out = {ra.obs_f64 + (size_t)pl * ra.obs_f64_plane, ra.base_reward + (size_t)pl * ra.agent_plane,
                       ra.reward + (size_t)pl * ra.reward_plane, ra.done + (size_t)pl * ra.done_plane,
                       ra.done_agents + (size_t)pl * ra.done_agents_plane, ra.info + (size_t)pl * ra.info_plane,
                       ra.sd_info + (size_t)pl * ra.sd_info_plane};
    sd_step_core<N, true>(p, out, ln, c, act, -1, stage, fault, any_reset);
    act = act_next;
    pl = pl + 1 == ra.num_planes ? 0u : pl + 1;
  }
  sd_store_state<N>(p, ln, c, any_reset);
  if (fault && ln.live && ln.k == 0) p.error_flags[ln.e] |= fault;
}

__global__ void k_synth_f32(float* out, u64 key, u64 env_base, u32 E, u32 n, u32 t0, u32 T) {
  const size_t total = (size_t)T * E * n;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const u32 a = (u32)(i % n);
    const size_t r = i / n;
    const u32 e = (u32)(r % E);
    const u32 t = (u32)(r / E);
    // uniform in [-0.1, 0.1): 24 random bits -> float32
    const u32 bits = (u32)(synth_hash(key, env_base + e, t0 + t, a) >> 40);
    out[i] = ((float)bits * (1.0f / 16777216.0f)) * 0.2f - 0.1f;
  }
}

#define CE_SD_DISPATCH(kern)                                                                      \
  do {                                                                                            \
    const u32 cnt_ = p.env_count ? p.env_count : p.E - p.env_first;                               \
    dim3 grid((cnt_ + 63) / 64), block(64);                                                       \
    switch (p.n) {                                                                                \
      case 1: hipLaunchKernelGGL(kern<1>, grid, block, 0, (hipStream_t)stream, p); break;         \
      case 2: hipLaunchKernelGGL(kern<2>, grid, block, 0, (hipStream_t)stream, p); break;         \
      case 3: hipLaunchKernelGGL(kern<3>, grid, block, 0, (hipStream_t)stream, p); break;         \
      case 4: hipLaunchKernelGGL(kern<4>, grid, block, 0, (hipStream_t)stream, p); break;         \
      case 5: hipLaunchKernelGGL(kern<5>, grid, block, 0, (hipStream_t)stream, p); break;         \
      case 6: hipLaunchKernelGGL(kern<6>, grid, block, 0, (hipStream_t)stream, p); break;         \
      case 7: hipLaunchKernelGGL(kern<7>, grid, block, 0, (hipStream_t)stream, p); break;         \
      case 8: hipLaunchKernelGGL(kern<8>, grid, block, 0, (hipStream_t)stream, p); break;         \
      case 9: hipLaunchKernelGGL(kern<9>, grid, block, 0, (hipStream_t)stream, p); break;         \
      case 10: hipLaunchKernelGGL(kern<10>, grid, block, 0, (hipStream_t)stream, p); break;       \
      default: break;                                                                             \
    }                                                                                             \
  } while (0)

void launch_sd_construct(const SdParams& p, void* stream) { CE_SD_DISPATCH(k_sd_construct); }
void launch_sd_reset(const SdParams& p, void* stream) { CE_SD_DISPATCH(k_sd_reset); }
#define CE_SD_STEP_DISPATCH(N_)                                                                                       \
  case N_:                                                                                                           \
    if (ra) hipLaunchKernelGGL(k_sd_rollout<N_>, dim3((cnt_ + SdGeo<N_>::EPW - 1) / SdGeo<N_>::EPW), dim3(64), 0,      \
                               (hipStream_t)stream, p, *ra);                                                         \
    else hipLaunchKernelGGL(k_sd_step<N_>, dim3((cnt_ + SdGeo<N_>::EPW - 1) / SdGeo<N_>::EPW), dim3(64), 0,            \
                            (hipStream_t)stream, p);                                                                 \
    break;
static void sd_step_dispatch(const SdParams& p, const RolloutArgs* ra, void* stream) {
  const u32 cnt_ = ra ? ra->env_end - ra->env_first : (p.env_count ? p.env_count : p.E - p.env_first);
  switch (p.n) {
    CE_SD_STEP_DISPATCH(1)
    CE_SD_STEP_DISPATCH(2)
    CE_SD_STEP_DISPATCH(3)
    CE_SD_STEP_DISPATCH(4)
    CE_SD_STEP_DISPATCH(5)
    CE_SD_STEP_DISPATCH(6)
    CE_SD_STEP_DISPATCH(7)
    CE_SD_STEP_DISPATCH(8)
    CE_SD_STEP_DISPATCH(9)
    CE_SD_STEP_DISPATCH(10)
    default: break;
  }
}
void launch_sd_step(const SdParams& p, void* stream) { sd_step_dispatch(p, nullptr, stream); }
void launch_sd_rollout(const SdParams& p, const RolloutArgs& ra, void* stream) { sd_step_dispatch(p, &ra, stream); }

void launch_synth_actions_f32(float* out, u64 key, u64 env_base, u32 E, u32 n, u32 t0, u32 T, void* stream) {
  hipLaunchKernelGGL(k_synth_f32, dim3(2048), dim3(256), 0, (hipStream_t)stream, out, key, env_base, E, n, t0, T);
}

}  // namespace ce
