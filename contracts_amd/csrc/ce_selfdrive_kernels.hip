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

template <int N> __global__ void k_sd_step(SdParams p) {
  const u32 e = p.env_first + blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= p.env_first + (p.env_count ? p.env_count : p.E - p.env_first)) return;
  Cars<N> c;
  sd_load(p, e, c);
  double theta = p.theta[e];
  const double high = p.high_bound;
  const double nan = __longlong_as_double(0x7ff8000000000000ll);
  bool active[N];
  int n_active = 0;
#pragma unroll
  for (int a = 0; a < N; ++a) {
    active[a] = p.active ? p.active[(size_t)e * N + a] != 0 : !c.done[a];
    n_active += active[a] ? 1 : 0;
  }
  if (c.done_all || n_active == 0) {  // the reference raises here (:154-167, collision_check_all undefined)
    p.error_flags[e] |= CE_FAULT_STEP_AFTER_DONE;
    return;
  }
  double new_pos[N];
#pragma unroll
  for (int k = 0; k < N; ++k) {
    new_pos[k] = c.pos[k];
    if (active[k]) {  // :172-180
      const double a = (double)p.actions[(size_t)e * N + k];
      const double vmax = k == 0 ? 1.0 : 0.25;
      const double v = py_max2(py_min2(py_max2(py_min2(a, 0.1), -0.1) + c.vel[k], vmax), 0.0);
      c.vel[k] = v;
      new_pos[k] = c.vel[k] + c.pos[k];
    }
  }
  bool just_passed[N];
  // update_rel_rank :110-125 (sort key = 2nd char of the id => agent index order)
#pragma unroll
  for (int k = 0; k < N; ++k) {
    just_passed[k] = active[k] && c.pos[k] < 0.0 && new_pos[k] > 0.0;
    if (just_passed[k]) {
#pragma unroll
      for (int q = 0; q < N; ++q)
        if (q == c.ncross) c.cross[q] = k;
      c.ncross++;
    }
  }
  // update_infos :127-149 (dist_to_front quirks kept: :143 subtracts a value from itself, :144 writes index n-1)
#pragma unroll
  for (int k = 0; k < N; ++k)
    if (just_passed[k]) {
      double dtf = 0.0;
#pragma unroll
      for (int i = 0; i < N; ++i)
        if (i != k) {
          if (!active[i] || new_pos[i] > new_pos[k]) {
            if (!active[i]) {
              if (high - new_pos[k] > dtf) dtf = high + 1 - new_pos[k];
            } else {
              if (new_pos[i] - new_pos[k] > dtf) dtf = new_pos[i] - new_pos[i];
            }
          }
        }
      c.dtf[N - 1] = dtf;
    }
  auto cross_at = [&](int idx) {
    int v = 0;
#pragma unroll
    for (int q = 0; q < N; ++q)
      if (q == idx) v = c.cross[q];
    return v;
  };
  auto eff_pos = [&](int car) {  // new position if acting, else stored position
    double v = 0.0;
#pragma unroll
    for (int q = 0; q < N; ++q)
      if (q == car) v = active[q] ? new_pos[q] : c.pos[q];
    return v;
  };
  auto is_active = [&](int car) {
    bool v = false;
#pragma unroll
    for (int q = 0; q < N; ++q)
      if (q == car) v = active[q];
    return v;
  };
  bool crashed = false;
  if (p.flags & CE_FLAG_COLLISION_ON) {  // check_if_crashed :81-90
    for (int i = 0; i + 1 < c.ncross; ++i)
      if (eff_pos(cross_at(i)) < eff_pos(cross_at(i + 1))) crashed = true;
  }
  double rews[N];
  if (crashed) {  // :196-215 — obs keep the OLD positions, velocities are already updated
    c.done_all = true;
#pragma unroll
    for (int k = 0; k < N; ++k) {
      if (active[k]) c.done[k] = true;
      rews[k] = -10000.0;
    }
    sd_write_obs(p, e, c, active, theta, 1.0);
  } else {
    // make_new_pos_consistent :92-108
    bool pre[N];
#pragma unroll
    for (int k = 0; k < N; ++k) pre[k] = false;
    for (int i = 0; i + 1 < c.ncross; ++i) {
      const int f = cross_at(i), b2 = cross_at(i + 1);
      const double pf = eff_pos(f), pb = eff_pos(b2);
      if (pf < pb && is_active(f) && is_active(b2)) {
        const double np2 = pf - 0.01;
#pragma unroll
        for (int q = 0; q < N; ++q)
          if (q == b2) {
            new_pos[q] = np2;
            if (np2 < 0) pre[q] = true;
          }
      }
    }
    {
      int nc = 0;
      int cr[N];
#pragma unroll
      for (int q = 0; q < N; ++q) cr[q] = -1;
      for (int i = 0; i < c.ncross; ++i) {
        const int a = cross_at(i);
        bool drop = false;
#pragma unroll
        for (int q = 0; q < N; ++q)
          if (q == a) drop = pre[q];
        if (!drop) {
#pragma unroll
          for (int q = 0; q < N; ++q)
            if (q == nc) cr[q] = a;
          nc++;
        }
      }
#pragma unroll
      for (int q = 0; q < N; ++q) c.cross[q] = cr[q];
      c.ncross = nc;
    }
#pragma unroll
    for (int k = 0; k < N; ++k) {
      if (active[k]) c.pos[k] = new_pos[k];
      rews[k] = -1.0;
    }
    if (active[0]) rews[0] -= 99.0;
#pragma unroll
    for (int i = 0; i < N; ++i)
      if (c.pos[i] > high) {
        c.pos[i] = high + 1;
        c.done[i] = true;
      }
    bool all_done = true;
#pragma unroll
    for (int k = 0; k < N; ++k)
      if (active[k] && !c.done[k]) all_done = false;
    c.done_all = all_done;
    sd_write_obs(p, e, c, active, theta, 0.0);
  }
#pragma unroll
  for (int k = 0; k < N; ++k) p.base_reward[(size_t)e * N + k] = active[k] ? (int32_t)rews[k] : 0;  // -1 / -100 / -10000
  // SelfdriveContractDistprop.compute_transfer on a0's observation (contract_list.py:69-102)
  if (p.contract == CE_CONTRACT_SELFDRIVE_DISTPROP) {
    // ob0[2+i] = pos_i - pos_0 as written above (for the crash branch: the OLD positions)
    double rel[N];
#pragma unroll
    for (int i = 0; i < N; ++i) rel[i] = c.pos[i] - c.pos[0];
    bool is_tuple[N];
    double tval[N];
    bool behind[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
      is_tuple[i] = false;
      tval[i] = 0.0;
      behind[i] = false;
    }
    double sum_d = 0.0;
    if (active[0] && just_passed[0]) {
      // slots n and n+1 of the loop read velocities (>= 0): never "behind"
#pragma unroll
      for (int i = 1; i < N; ++i)
        if (rel[i] < 0) {
          behind[i] = true;
          sum_d += -rel[i];
        }
      bool any = false;
#pragma unroll
      for (int i = 1; i < N; ++i) any = any || behind[i];
      if (any) {
        is_tuple[0] = true;
        tval[0] = theta * sum_d;
      }
#pragma unroll
      for (int i = 1; i < N; ++i)
        if (active[i] && !behind[i]) {
          is_tuple[i] = true;
          tval[i] = theta * rel[i];
        }
    }
    double total = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i)
      if (active[i]) {
        if (is_tuple[i]) {
          rews[i] -= tval[i];
          total += tval[i];
          if (i == 0) {
#pragma unroll
            for (int j = 1; j < N; ++j)
              if (behind[j] && active[j]) rews[j] += tval[0] * ((-rel[j]) / sum_d);
          } else {
            if (active[0]) rews[0] += tval[i] * 1.0;
          }
        } else {
          rews[i] -= 0.0;
          total += 0.0;
#pragma unroll
          for (int j = 0; j < N; ++j)
            if (j != i && active[j]) rews[j] += 0.0 / (double)(n_active - 1);
        }
      }
    c.transfers += total;
  }
#pragma unroll
  for (int k = 0; k < N; ++k) {
    p.reward[(size_t)e * N + k] = active[k] ? rews[k] : nan;
    p.info[((size_t)e * N + k) * 2] = just_passed[k] ? 1 : 0;
    p.info[((size_t)e * N + k) * 2 + 1] = crashed ? 1 : 0;
    p.done_agents[(size_t)e * N + k] = c.done[k] ? 1 : 0;
  }
  p.done[e] = c.done_all ? 1 : 0;
  p.f64_metrics[(size_t)e * CE_MF_COUNT(N) + CE_MF_TRANSFERS] = c.transfers;
  if (c.done_all) {
    for (int k = 0; k < (int)CE_MF_COUNT(N); ++k)
      p.final_f64_metrics[(size_t)e * CE_MF_COUNT(N) + k] = k == CE_MF_TRANSFERS ? c.transfers : 0.0;
    if (p.flags & CE_FLAG_AUTO_RESET) {
      sd_reset_env(p, e, c, theta);
      bool all_active[N];
#pragma unroll
      for (int a = 0; a < N; ++a) all_active[a] = true;
      sd_write_obs(p, e, c, all_active, theta, 0.0);
      p.theta[e] = theta;
      p.f64_metrics[(size_t)e * CE_MF_COUNT(N) + CE_MF_TRANSFERS] = 0.0;
    }
  }
  sd_store(p, e, c);
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
void launch_sd_step(const SdParams& p, void* stream) { CE_SD_DISPATCH(k_sd_step); }

void launch_synth_actions_f32(float* out, u64 key, u64 env_base, u32 E, u32 n, u32 t0, u32 T, void* stream) {
  hipLaunchKernelGGL(k_synth_f32, dim3(2048), dim3(256), 0, (hipStream_t)stream, out, key, env_base, E, n, t0, T);
}

}  // namespace ce
