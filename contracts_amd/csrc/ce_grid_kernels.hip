// ce_grid_kernels.hip — gfx950 kernels for the Cleanup / Harvest grid families.
//
// Execution model: ONE 64-lane wavefront owns ONE env replica for the whole step.  The env's
// map (padded by the view radius), its MT19937 state and the step's random words live in the
// wave's private slice of LDS; the agent table lives in VGPRs with lane a = agent a, and all
// order-dependent logic of the reference (move conflict resolution, beam order, Fisher-Yates
// shuffles) runs as wave-uniform scalar control flow that addresses agents with
// v_readlane/v_writelane and takes set-membership decisions with 64-bit ballots.  Waves never
// talk to each other: no block barriers, no atomics, no inter-workgroup traffic.
//
// What each phase restates (reference paths relative to the reference root):
//   update_moves            environments/map_env.py:483-676   
//   consume                 map_env.py:244-247, Agent.py:189-195,228-234
//   update_custom_moves     map_env.py:678-693  + update_map_fire :721-814
//   spawn (cleanup)         cleanup_new.py:294-376 ; (harvest) harvest_new.py:251-317
//   color_view crop         map_env.py:397-411
//   infos/feature obs       cleanup_new.py:211-267,378-420 ; harvest_new.py:181-239,326-336
//   contract + wrapper      contract/contract_list.py:22-27,45-54 ; two_stage_train.py:62-121,159-187
//   numpy legacy RandomState shuffle / random_sample / randint / uniform (MT19937)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>

#include "ce_device.h"

// base[idx] for a wave-uniform global base and a per-lane 32-bit index: the byte offset is formed in 32 bits,
// which is what lets the access take the SGPR-base + VGPR-offset form (a 64-bit scaled index does not).
#define GAT(base, idx) (*(decltype(base))((CE_GPTR(char))(base) + (u32)((u32)(idx) * (u32)sizeof(*(base)))))

// Diagnostic builds (phase stamps, truncated kernels, ablations, issue probes: the scripts under tools/) compile with
// -DCE_DIAGNOSTIC and take their instrumentation from ce_grid_probe.inc; the shipped build has none of it.
#ifdef CE_DIAGNOSTIC
#include "ce_grid_probe.inc"
#else
#define CE_STAMP(k) ((void)0)
#define CE_SUBSTAMP(k) ((void)0)
#define CE_REALSTAMP(k) ((void)0)
#define CE_PROBE_POINT(T_, P_, LANE_, N_) ((void)0)
#define CE_TRUNCATE_SPAWN_RETURN() ((void)0)
namespace ce {
namespace diag {
constexpr bool ablate_moves = false, ablate_features = false, ablate_shuffle = false, ablate_obsstore = false,
               ablate_gridstore = false, ablate_rngstore = false, seq_shuffle = false, serial_apply = false,
               serial_small_shuffle = false, ablate_twist = false, ablate_half_narrow = false, ablate_featscan = false,
               ablate_obs = false;
}
}  // namespace ce
#endif

namespace ce {
// This file is compiled twice.  The second translation unit (ce_grid_kernels_ctr.hip: -DCE_RNG_COUNTER) holds the grid kernels of
// the counter-RNG mode (CE_FLAG_RNG_COUNTER, contracts_engine.h): same step logic, the env's random stream comes from
// Philox4x32-10 blocks instead of the numpy MT19937 state.  Its kernels, constant tables and launchers live in an inline
// namespace so the two sets of symbols never meet; the feature-vector kernels and the helpers at the end of the file are
// built in the first unit only.
#ifdef CE_RNG_COUNTER
inline namespace ctr {
#define CE_LAUNCHER(name) name##_ctr
constexpr bool kCounterRng = true;
#else
#define CE_LAUNCHER(name) name
constexpr bool kCounterRng = false;
#endif

typedef uint32_t u32;
typedef uint64_t u64;
typedef int32_t i32;
// words of one generation of the env's stream in LDS, and of an env's row of the `rng` buffer
constexpr u32 kGen = kCounterRng ? (u32)CE_RNG_COUNTER_GEN : (u32)kMtN;
constexpr u32 kRngRow = kCounterRng ? (u32)CE_RNG_WORDS_COUNTER : (u32)kRngStride;

#define DEVINL __device__ __forceinline__

__constant__ GridTables c_tab[2];
__constant__ u32 c_rgb[16];                                       // colour LUT, 0x00BBGGRR

// ----------------------------------------------------------------------------------------
// wave primitives
// ----------------------------------------------------------------------------------------
DEVINL u32 lane_id() {
  const u32 l = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  __builtin_assume(l < 64u);  // lets the compiler drop the `lane + 64 r < N` guards that are always true
  return l;
}
DEVINL u64 ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
DEVINL u32 rdl(u32 v, u32 l) { return (u32)__builtin_amdgcn_readlane((int)v, (int)l); }
// clang exposes no builtin for v_writelane_b32, but the LLVM intrinsic can be declared directly; the
// compiler then manages M0 (gfx9 needs the lane select in M0 when the value is an SGPR too) and the
// VALU-writes-SGPR -> lane-select hazard.  `val` and `l` must be wave-uniform.
extern "C" __device__ int ce_llvm_writelane(int, int, int) __asm("llvm.amdgcn.writelane.i32");
DEVINL u32 wrl(u32 val, u32 l, u32 old) { return (u32)ce_llvm_writelane((int)val, (int)l, (int)old); }
DEVINL u32 rfl(u32 v) { return (u32)__builtin_amdgcn_readfirstlane((int)v); }
DEVINL u32 bperm(u32 v, u32 src_lane) { return (u32)__builtin_amdgcn_ds_bpermute((int)(src_lane << 2), (int)v); }
DEVINL u32 popc64(u64 x) { return (u32)__builtin_popcountll(x); }
DEVINL u32 ctz64(u64 x) { return (u32)__builtin_ctzll(x); }
DEVINL u32 fls64(u64 x) { return 63u - (u32)__builtin_clzll(x); }
DEVINL bool bit(u64 m, u32 i) { return (m >> i) & 1ull; }

// Lanes of one wave exchange data through LDS in program order; this only stops the compiler
// from reordering one lane's LDS accesses around another lane's (no instruction is emitted).
DEVINL void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// wave-wide unsigned min via DPP row shifts + row broadcasts (result uniform)
DEVINL u32 wave_min_u32(u32 v) {
  const int id = (int)0xffffffffu;
#define CE_DPP_MIN(ctrl, rmask)                                                                        \
  {                                                                                                    \
    u32 t = (u32)__builtin_amdgcn_update_dpp(id, (int)v, ctrl, rmask, 0xf, false);                     \
    v = t < v ? t : v;                                                                                 \
  }
  CE_DPP_MIN(0x111, 0xf)  // row_shr:1
  CE_DPP_MIN(0x112, 0xf)  // row_shr:2
  CE_DPP_MIN(0x114, 0xf)  // row_shr:4
  CE_DPP_MIN(0x118, 0xf)  // row_shr:8
  CE_DPP_MIN(0x142, 0xa)  // row_bcast:15 -> rows 1,3
  CE_DPP_MIN(0x143, 0xc)  // row_bcast:31 -> rows 2,3
#undef CE_DPP_MIN
  return rdl(v, 63);
}

// ----------------------------------------------------------------------------------------
// MT19937 in LDS (numpy legacy RandomState stream)
// ----------------------------------------------------------------------------------------
DEVINL u32 mt_temper(u32 y) {
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}
DEVINL u32 mt_mix(u32 a, u32 b, u32 c) {  // new = c ^ twist(a,b)
  u32 y = (a & 0x80000000u) | (b & 0x7fffffffu);
  return c ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

// Regenerates the 624-word state with 64 lanes, four consecutive words (one 16-byte quad) per lane and phase.
// new[i] = mix(old[i], old[i+1], X[i]) with X[i] = old[i+397] for i < 227 and new[i-227] after that; 397 and -227 are
// both 1 mod 4, so with the state seen as 156 quads Q[0..155] a lane that owns quad q needs Q[q], the first word of
// Q[q+1], and the quad r = (q + 99) mod 156 shifted by one word (its last three words and the first of Q[r+1]).
// Three dependent phases of quads [0,56), [56,112), [112,156) keep every dependence in an earlier phase: phase B reads
// Q[155] (still old: words 621..623) and Q[0..55] (new), phase C reads Q[55..99] (new), and the wrap-around word of
// quad 155 is new[0], as in the sequential algorithm.  3 x (2 ds_read_b128 + 2 ds_read_b32 + 20 VALU + 1 ds_write_b128)
// instead of eleven 64-word rounds of dword accesses (it was ~140 VALU and 45 LDS instructions per twist).
// (noinline: it is reached from every place that can run the stream dry.  The state pointer is passed in the
// LDS address space so the body is ds_read / ds_write with 32-bit addresses instead of flat accesses.)
typedef __attribute__((address_space(3))) u32 lds_u32;
typedef u32 u32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) u32x4_t lds_u32x4;
DEVINL void mt_twist_body(lds_u32* mt, u32 lane) {
  lds_u32x4* Q = (lds_u32x4*)mt;
  constexpr u32 kQuads = (u32)kMtN / 4;  // 156
  wave_sync();
#pragma unroll
  for (int ph = 0; ph < 3; ++ph) {
    const u32 q0 = ph == 0 ? 0u : ph == 1 ? 56u : 112u, cnt = ph == 2 ? 44u : 56u;
    const bool on = lane < cnt;
    const u32 q = q0 + (on ? lane : 0u);  // idle lanes shadow the phase's first quad (reads only)
    u32 r = q + 99u;
    r = r >= kQuads ? r - kQuads : r;
    const u32 q1 = q + 1u == kQuads ? 0u : q + 1u, r1 = r + 1u == kQuads ? 0u : r + 1u;
    const u32x4_t A = Q[q], R = Q[r];
    const u32 nx = mt[4u * q1], rx = mt[4u * r1];
    u32x4_t N;
    N.x = mt_mix(A.x, A.y, R.y);
    N.y = mt_mix(A.y, A.z, R.z);
    N.z = mt_mix(A.z, A.w, R.w);
    N.w = mt_mix(A.w, nx, rx);
    wave_sync();  // every read of the phase precedes its writes (a lane's next word is its neighbour's quad)
    if (on) Q[q] = N;
    wave_sync();
  }
}

__device__ __noinline__ void mt_twist_lds(lds_u32* mt, u32 lane) { mt_twist_body(mt, lane); }

DEVINL void mt_twist(u32* mt, u32 lane) {
  if (!diag::ablate_twist) mt_twist_lds((lds_u32*)mt, lane);
}
// the same inlined: for a caller that holds most of its register budget live at the call (a real call saves and restores
// the caller's registers through scratch memory around it)
DEVINL void mt_twist_inline(u32* mt, u32 lane) { mt_twist_body((lds_u32*)mt, lane); }

// Counter mode (CE_FLAG_RNG_COUNTER, contracts_engine.h): Philox4x32-10 (Salmon et al., SC'11; Random123), ten rounds of
//   (c0, c1, c2, c3) <- (hi(M1 c2) ^ c1 ^ k0, lo(M1 c2), hi(M0 c0) ^ c3 ^ k1, lo(M0 c0)),  k0 += W0, k1 += W1.
// The 64-bit products are one v_mad_u64_u32 each; the key schedule is wave-uniform (scalar adds).
DEVINL void philox4x32_10(u32 k0, u32 k1, u32& c0, u32& c1, u32& c2, u32& c3) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const u64 p0 = (u64)0xD2511F53u * c0, p1 = (u64)0xCD9E8D57u * c2;
    const u32 n0 = (u32)(p1 >> 32) ^ c1 ^ k0, n2 = (u32)(p0 >> 32) ^ c3 ^ k1;
    c1 = (u32)p1;
    c3 = (u32)p0;
    c0 = n0;
    c2 = n2;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}
// Generation `gen` of an env's stream into LDS: 128 blocks of four words, two per lane, each one 16-byte store.
DEVINL void ctr_fill_body(lds_u32* mt, u32 lane, u32 k0, u32 k1, u32 gen) {
  static_assert(CE_RNG_COUNTER_GEN == 512, "two blocks per lane");
  lds_u32x4* Q = (lds_u32x4*)mt;
  wave_sync();
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    u32 c0 = lane + 64u * r, c1 = gen, c2 = 0, c3 = 0;
    philox4x32_10(k0, k1, c0, c1, c2, c3);
    u32x4_t c;
    c.x = c0;
    c.y = c1;
    c.z = c2;
    c.w = c3;
    Q[lane + 64u * r] = c;
  }
  wave_sync();
}
// ... as a called function for the places that run a generation dry in mid-operation (resets; rare)
__device__ __noinline__ void ctr_fill_lds(lds_u32* mt, u32 lane, u32 k0, u32 k1, u32 gen) {
  // arguments of a called function arrive in VGPRs; these three are wave-uniform, and the ten round keys derived from
  // them belong in SGPRs (as VGPRs they were twenty registers held through the whole function)
  ctr_fill_body(mt, lane, rfl(k0), rfl(k1), rfl(gen));
}

struct Rng {
  u32* mt;     // LDS, kGen words
  u32 pos;     // words consumed from the current generation (uniform)
  u32 cache;   // lane k: tempered word cbase + k
  u32 cbase;   // uniform
  u32 ccount;  // uniform; 0 = cache invalid
  u64 cvalid;  // lanes < ccount
  u32 twists;  // generations advanced since the state was loaded (uniform): 0 = the key words in HBM are still current
  u32 k0, k1, gen0;  // counter mode: the env's key and the generation its `rng` row named at load (uniform)
};
// the stream has run dry: next generation into LDS (callers reset pos)
DEVINL void rng_advance(Rng& r, u32 lane) {
  r.twists += 1;
  if (kCounterRng) ctr_fill_lds((lds_u32*)r.mt, lane, r.k0, r.k1, r.gen0 + r.twists);
  else mt_twist(r.mt, lane);
}
// Counter mode: every operation on an env (construct, reset, a step that is taken) opens a fresh generation and leaves nothing
// behind in it, so no state but the generation number travels between launches and a fused rollout draws what single steps
// draw.  The per-step kernel opens it while its state loads are in flight (load_env_state, inlined fill: a called function
// would wait for them first); a rollout's later steps open theirs at step entry.
DEVINL void rng_begin_op(Rng& r, u32 lane, bool inline_fill = false) {
  if (kCounterRng && r.pos >= kGen) {
    r.twists += 1;
    if (inline_fill) ctr_fill_body((lds_u32*)r.mt, lane, r.k0, r.k1, r.gen0 + r.twists);
    else ctr_fill_lds((lds_u32*)r.mt, lane, r.k0, r.k1, r.gen0 + r.twists);
    r.pos = 0;
    r.ccount = 0;
  }
}
DEVINL void rng_end_of_op(Rng& r) {
  if (kCounterRng) {
    r.pos = kGen;
    r.ccount = 0;
  }
}

// The stream position / cache window are wave-uniform by construction, but after inlined helpers with several
// exits merge, the compiler's divergence analysis can lose that and turn every loop over the stream into an
// exec-masked vector loop.  Re-asserting uniformity at the entry of those loops costs five v_readfirstlane.
DEVINL void rng_assert_uniform(Rng& r) {
  r.pos = rfl(r.pos);
  r.cbase = rfl(r.cbase);
  r.ccount = rfl(r.ccount);
  r.cvalid = ((u64)rfl((u32)(r.cvalid >> 32)) << 32) | rfl((u32)r.cvalid);
}

// make the per-lane cache cover stream words [pos, pos + ccount)
DEVINL void rng_refill(Rng& r, u32 lane) {
  if (r.pos >= kGen) {
    rng_advance(r, lane);
    r.pos = 0;
  }
  r.cbase = r.pos;
  const u32 left = kGen - r.pos;
  r.ccount = left < 64u ? left : 64u;
  r.cvalid = left < 64u ? ((1ull << left) - 1ull) : ~0ull;
  const u32 idx = r.pos + lane;
  r.cache = mt_temper(r.mt[idx < kGen ? idx : kGen - 1]);
}

DEVINL u32 rng_next(Rng& r, u32 lane) {
  u32 off = r.pos - r.cbase;
  if (off >= r.ccount) {
    rng_refill(r, lane);
    off = 0;
  }
  r.pos += 1;
  return rdl(r.cache, off);
}

// The rand(k) call of the spawn models: `count` stream words are consumed.  The first `keep` tempered words
// go to U (the doubles that are compared against a real threshold); for every double d the byte S[d] records
// "u_d < 0.5", i.e. bit 31 of its first word is clear (X < 2^52 <=> (a >> 5) < 2^26 <=> a < 2^31).
DEVINL void rng_bulk(Rng& r, u32* U, uint8_t* S, u32 count, u32 keep, bool want_s, u32 lane) {
  rng_assert_uniform(r);
  u32 done = 0;
  while (done < count) {
    if (r.pos >= kGen) {
      rng_advance(r, lane);
      r.pos = 0;
    }
    u32 chunk = kGen - r.pos;
    if (chunk > count - done) chunk = count - done;
    for (u32 k = lane; k < chunk; k += 64) {
      const u32 w = mt_temper(r.mt[r.pos + k]);
      const u32 sidx = done + k;
      if (sidx < keep) U[sidx] = w;
      if (want_s && (sidx & 1u) == 0) S[sidx >> 1] = (uint8_t)((w >> 31) ^ 1u);
    }
    r.pos += chunk;
    done += chunk;
  }
  r.ccount = 0;
  wave_sync();
}

DEVINL double rng_double(Rng& r, u32 lane) {
  u32 a = rng_next(r, lane) >> 5;
  u32 b = rng_next(r, lane) >> 6;
  return ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
}

template <int PAD> DEVINL u32 writelane(u32 val, u32 sel, u32 old) { return wrl(val, sel, old); }

// np.random.shuffle (untyped path): for i = len-1 .. 1: j = random_interval(i); swap(x[i], x[j]), over a
// list held across lanes (element k < 64 in lane k of L0, element 64 + k in lane k of L1).
// random_interval is masked rejection sampling, one stream word per attempt.  All cached words are tested
// at once: the draw for index i is the first unread cached word whose masked value is <= i (one ballot +
// find-first-set), the words before it are the rejected attempts.  Indices are walked in segments that
// share a mask (2^k .. 2^(k+1)-1) so the masked words are computed once per segment; the hot inner loop
// is ~16 instructions, all in registers; only the outermost loop may refill the cache (and call the twist).
// MODE 0: swap the list; 1: collect the draws (J[i] in lane i of L0); 2: only consume the stream words
template <bool TWO, int MODE = 0> DEVINL void shuffle_core(Rng& r, u32& L0, u32& L1, u32 len, u32 lane) {
  if (len < 2) return;
  rng_assert_uniform(r);
  u32 i = rfl(len) - 1;
  u32 pos = r.pos, cbase = r.cbase, ccount = r.ccount, cache = r.cache;
  u64 cvalid = r.cvalid;
  u32 l0 = L0, l1 = L1;
  while (i >= 1) {
    const u32 lo = 1u << (31 - __builtin_clz(i));  // segment [lo, i] shares mask 2*lo - 1
    const u32 mask = 2 * lo - 1;
    while (i >= lo) {
      u32 off = pos - cbase;
      if (off >= ccount) {
        r.pos = pos;
        rng_refill(r, lane);
        pos = r.pos;
        cbase = r.cbase;
        ccount = r.ccount;
        cvalid = r.cvalid;
        cache = r.cache;
        off = 0;
      }
      u64 avail = cvalid & (~0ull << off);
      const u32 v = cache & mask;
      u32 lastk = 0xffffffffu;  // last accepted cached word of this run (pos is derived at exit)
      bool exhausted = false;
      if (TWO && lo >= 64) {
        do {  // x[i] lives in L1, x[j] in either register
          const u64 hit = ballot(v <= i) & avail;
          if (hit == 0) {  // every remaining cached word is a rejected attempt for this i
            exhausted = true;
            break;
          }
          const u32 k = ctz64(hit);
          const u32 j = rdl(v, k);
          avail &= (~1ull << k);
          lastk = k;
          const u32 jm = j & 63, im = i - 64;
          const u32 vi = rdl(l1, im), a0 = rdl(l0, jm), b0 = rdl(l1, jm);
          const bool jlow = j < 64;
          const u32 vj = jlow ? a0 : b0;
          l0 = writelane<2>(jlow ? vi : a0, jm, l0);
          l1 = writelane<0>(jlow ? b0 : vi, jm, l1);
          l1 = writelane<0>(vj, im, l1);  // last: wins when jm == im with j < 64
          --i;
        } while (i >= lo);
      } else {
        do {
          const u64 hit = ballot(v <= i) & avail;
          if (hit == 0) {
            exhausted = true;
            break;
          }
          const u32 k = ctz64(hit);
          const u32 j = rdl(v, k);
          avail &= (~1ull << k);
          lastk = k;
          if (MODE == 2) {
          } else if (MODE == 1) {  // collect J[i] in lane i instead of swapping
            l0 = writelane<0>(j, i, l0);
          } else {
            const u32 vi = rdl(l0, i), vj = rdl(l0, j);
            l0 = writelane<0>(vj, i, l0);
            l0 = writelane<2>(vi, j, l0);
          }
          --i;
        } while (i >= lo);
      }
      if (exhausted) pos = cbase + ccount;
      else if (lastk != 0xffffffffu) pos = cbase + lastk + 1;
    }
  }
  r.pos = pos;
  L0 = l0;
  L1 = l1;
}
// The same walk for a list of at most 64 items held in one register, written as ONE flat loop (one exit, the
// cache refill as a rare `continue`): the nested form above costs ~17 scalar instructions per draw in compiler
// generated flag shuffling, this one ~10.  MODE as in shuffle_core.
// Lists of at most 8 elements (the agent-order shuffles of every step): the serial walk below costs ~540 cycles per
// draw on a full SIMD — a dozen dependent scalar instructions each — and these two shuffles were a fifth of a wave's
// life.  Here the acceptance masks of all seven possible indices (three bit masks, seven compares) are taken up
// front, independent of each other; what stays serial per draw is and / find-first / shift on 64-bit scalars.
// Returns the index still to be drawn (0 = done) when the cached words run out, for the serial walk to finish.
template <int MODE> DEVINL u32 shuffle_le8(Rng& r, u32& l0, u32 len, u32 lane) {
  u32 off = r.pos - r.cbase;
  if (off >= r.ccount || r.ccount - off < 24u) {  // too few cached words to make running out unlikely: recache at pos
    rng_refill(r, lane);
    off = 0;
  }
  const u32 w = r.cache;
  const u32 v7 = w & 7u, v3 = w & 3u, v1 = w & 1u;
  // indices 7, 3 and 1 equal their mask: every word is accepted (the next unread one); only 6, 5, 4 and 2 can reject
  const u64 B6 = ballot(v7 <= 6u), B5 = ballot(v7 <= 5u), B4 = ballot(v7 <= 4u), B2 = ballot(v3 <= 2u);
  // Straight-line walk: every step is and / find-first / shift / and on 64-bit scalars, no branch inside — running out of
  // cached words only sets a flag (the find-first then lands on a planted top bit), and the rare failure is redone by the
  // caller's serial walk from the untouched stream position.  (The earlier form returned from inside each step: the
  // compiler turned the chain into a state machine of ~20 scalar instructions per step.)
  u64 avail = r.cvalid & (~0ull << off);
  u32 klast = off - 1u;  // no step taken: the position stays
  bool ok = true;
  const u32 l0_in = l0;
#define CE_LE8_STEP(I, A, V)                                     \
  if (len > (I)) {                                               \
    const u64 hit = (A) & avail;                                 \
    ok = ok && hit != 0;                                         \
    const u32 k = ctz64(hit | (1ull << 63));                     \
    avail &= (~1ull) << k;                                       \
    klast = k;                                                   \
    if (MODE == 1) {                                             \
      l0 = wrl(rdl((V), k), (I), l0);                            \
    } else if (MODE == 0) {                                      \
      const u32 j = rdl((V), k);                                 \
      const u32 vi = rdl(l0, (I)), vj = rdl(l0, j);              \
      l0 = wrl(vj, (I), l0);                                     \
      l0 = wrl(vi, j, l0);                                       \
    }                                                            \
  }
  CE_LE8_STEP(7, ~0ull, v7)
  CE_LE8_STEP(6, B6, v7)
  CE_LE8_STEP(5, B5, v7)
  CE_LE8_STEP(4, B4, v7)
  CE_LE8_STEP(3, ~0ull, v3)
  CE_LE8_STEP(2, B2, v3)
  CE_LE8_STEP(1, ~0ull, v1)
#undef CE_LE8_STEP
  if (ok) {
    r.pos = r.cbase + klast + 1u;
    return 0;
  }
  l0 = l0_in;  // ran out of cached words (needs > 24 rejections in a row-ish: practically never): nothing was consumed
  return len - 1;
}

template <int MODE> DEVINL void shuffle_small(Rng& r, u32& L0, u32 len, u32 lane) {
  if (len < 2) return;
  rng_assert_uniform(r);
  u32 i = rfl(len) - 1;
  u32 l0 = L0;
  if (!diag::serial_small_shuffle && i <= 7u) {
    i = shuffle_le8<MODE>(r, l0, i + 1, lane);
    if (i == 0) {
      L0 = l0;
      return;
    }
  }
  u32 off = r.pos - r.cbase;
  if (off >= r.ccount) {
    rng_refill(r, lane);
    off = 0;
  }
  u64 avail = r.cvalid & (~0ull << off);  // unread words of the cache
  u32 nb = 1u << (31 - __builtin_clz(i));  // lowest index of the current mask segment
  u32 v = r.cache & (2 * nb - 1);
  for (;;) {
    const u64 hit = ballot(v <= i) & avail;
    if (hit == 0) {  // every remaining cached word is a rejected attempt for this i
      r.pos = r.cbase + r.ccount;
      rng_refill(r, lane);
      avail = r.cvalid;
      v = r.cache & (2 * nb - 1);
      continue;
    }
    const u32 k = ctz64(hit);
    avail &= (~1ull) << k;  // the words before k were rejected attempts, k is consumed
    if (MODE == 1) {        // collect J[i] in lane i
      l0 = wrl(rdl(v, k), i, l0);
    } else if (MODE == 0) {
      const u32 j = rdl(v, k);
      const u32 vi = rdl(l0, i), vj = rdl(l0, j);
      l0 = wrl(vj, i, l0);
      l0 = wrl(vi, j, l0);
    }
    if (i == nb) {  // segment done: the next index uses the next smaller mask
      if (i == 1) {
        r.pos = r.cbase + k + 1;
        break;
      }
      nb >>= 1;
      v = r.cache & (2 * nb - 1);
    }
    --i;
  }
  L0 = l0;
}

// Lanes below `lane` that are set in a wave-uniform 64-bit mask (v_mbcnt)
DEVINL u32 rank_in(u64 m, u32 /*lane*/) {
  return __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
}

// The draws of the waste-list shuffle (np.random.shuffle: for i = len-1 .. 1: j = random_interval(i), masked rejection
// sampling, one stream word per attempt; cleanup_new.py:339).  Word-parallel over the 64 cached stream words: word k is
// consumed at index i_k = i0 - a_k, a_k = the number of words accepted before it, and is accepted iff
// (w_k & mask(i_k)) <= i_k with mask(i) = 2^bitlen(i) - 1 -- a self-referential predicate whose solution is the unique fixed
// point of X <- { k : accept(w_k, i0 - rank_X(k)) }: if X is a fixed point then, by induction over k, every word's rank and
// therefore its decision is the sequential one.  The iteration from any start reaches it (the lowest word that is still wrong
// has a correct rank and is right after the next pass), and `X' == X` both ends the loop and proves the result, so neither
// monotone bounds nor a cut at the mask segments (round 5: one batch per segment [2^b, 2^(b+1)) and a serial walk below 8)
// are needed: a batch runs through as many segments as its words reach, down to index 1.  Three batches and ~14 passes of
// 7 VALU per 119-entry list where the segmented form took six batches, ~20 passes and a serial tail.
// The iteration runs on the complement: Y = the unread cached words that are REJECTED, because the index of word k is then
// i0 - (k - off) + |Y below k| — two v_mbcnt with the start index as their accumulator, no subtraction.  Past the end of
// the list the index runs to 0 and below: v_ffbh_i32 gives those a mask of 1 (index 0 or -1) or of the leading-ones run,
// the signed compare rejects every negative index and at index 0 accepts the first even word — a deterministic function of
// the ranks like the real draws, so the fixed point stays unique; real draws are the accepted words with index >= 1.
// Two passes per convergence test (a pass over a fixed point reproduces it): half the scalar compares and taken branches.
// J[i] (LDS) receives the draw of index i; r advances past every consumed word.
#ifndef CE_SHUFFLE_DRAWS_SEGMENTED
DEVINL i32 draw_mask(i32 idx) {  // idx >= 1: 2^bitlen(idx) - 1 (v_ffbh_i32: defined for every input, unlike __builtin_clz)
  u32 lead;
  asm("v_ffbh_i32 %0, %1" : "=v"(lead) : "v"(idx));
  return (i32)(0xffffffffu >> (lead & 31u));
}
template <bool STORE = true, u32 RECACHE = 32u> DEVINL void shuffle_draws(Rng& r, u32 len, u32* J, u32 lane) {
  rng_assert_uniform(r);
  u32 i0 = rfl(len) - 1;  // indices i0 .. 1 are still to be drawn
  while (i0 >= 1) {
    u32 off = r.pos - r.cbase;
    if (off >= r.ccount || r.ccount - off < RECACHE) {  // a short rest of an earlier consumer's cache: recache at pos (a batch per refill)
      rng_refill(r, lane);
      off = 0;
    }
    const u64 have = r.cvalid & (~0ull << off);  // unread cached words
    const i32 c7 = (i32)(r.cache & 127u);        // every mask is <= 127
    const i32 istart = (i32)i0 - (i32)(lane - off);  // the word's index if every unread word before it were accepted
    i32 idx = istart;
    u64 Y = ballot((c7 & draw_mask(idx)) > idx) & have;
    for (;;) {
      idx = istart + (i32)rank_in(Y, lane);
      const u64 Y1 = ballot((c7 & draw_mask(idx)) > idx) & have;
      idx = istart + (i32)rank_in(Y1, lane);
      Y = ballot((c7 & draw_mask(idx)) > idx) & have;
      if (Y == Y1) break;  // idx belongs to Y1 == Y: the fixed point
    }
    const u64 real = ~Y & have & ballot(idx >= 1);  // accepted, without whatever follows the end of the list
    const u32 total = popc64(real);
    if (total >= i0) {                            // the list completes inside this batch: stop behind the draw of index 1
      r.pos = r.cbase + ctz64(real & ballot(idx == 1)) + 1;
      i0 = 0;
    } else {                                      // every cached word is consumed (accepted or rejected)
      r.pos = r.cbase + r.ccount;
      i0 -= total;
    }
    if (STORE && ((real >> lane) & 1ull)) J[idx] = (u32)(c7 & draw_mask(idx));
  }
  if (STORE) wave_sync();
}
// The stream words of a shuffle whose order nobody looks at (an interaction-free update_moves, fewer than two beams): the same
// fixed point, nothing stored.  ~4 passes of 6 VALU for a list of <= 8 where the find-first walk (shuffle_le8<2>) spent 8
// dependent scalar instructions per draw on the CU's one scalar pipe, twice per step.
DEVINL void shuffle_consume(Rng& r, u32 len, u32 lane) {
  if (len < 2) return;
  shuffle_draws<false, 24u>(r, len, nullptr, lane);
}
#else
// The draws of the waste-list shuffle.  Indices >= 32 are vectorised: for one mask segment [lo, i0] and the
// cached words k = off.. (v_k = word_k & mask), sequential rejection sampling accepts word k iff
// v_k <= i0 - a_k, where a_k is the number of words accepted before k.  That self-referential predicate is
// solved by monotone bounds: L (surely accepted) and U (possibly accepted) are recomputed from each other's ranks
//   L' = { v_k <= i0 - |U below k| },  U' = { v_k <= i0 - |L below k| }
// until they meet (the lowest undecided lane always resolves, so it terminates; 1-3 rounds per 64 words from the
// start L = { v_k <= max(lo, i0 - (k - off)) }, U = { v_k <= i0 }).  The accepted word with rank a is the draw for
// index i0 - a; draws are scattered to J[index] in LDS.  The short segments below 32 cost a whole vector round
// each, so they take the one-draw-per-ballot walk instead (3 VALU per draw), collecting J[i] in lane i.
// Returns with r advanced past every consumed word.
DEVINL void shuffle_draws(Rng& r, u32 len, u32* J, u32 lane) {
  constexpr u32 kVecMin = 8;
  rng_assert_uniform(r);
  u32 i0 = len - 1;
  while (i0 >= kVecMin) {
    const u32 lo = 1u << (31 - __builtin_clz(i0));  // segment [lo, i0] shares mask 2*lo - 1
    const u32 mask = 2 * lo - 1;
    u32 off = r.pos - r.cbase;
    if (off >= r.ccount) {
      rng_refill(r, lane);
      off = 0;
    }
    const u32 d = lane - off;  // rank among the unread cached words
    const i32 v = d < r.ccount - off ? (i32)(r.cache & mask) : 0x7fffffff;  // read / missing words never accept
    const i32 t0 = (i32)i0 - (i32)d;
    u64 Lm = ballot(v <= (t0 > (i32)lo ? t0 : (i32)lo)), Um = ballot(v <= (i32)i0);
    while (Lm != Um) {  // each half-step uses the newest bound of the other side (Gauss-Seidel: fewer rounds than updating both from the old pair)
      const i32 aL = (i32)rank_in(Lm, lane);
      Um = ballot(v <= (i32)i0 - aL);  // a_k >= |L below k|  =>  accepted words lie in { v <= i0 - |L below k| }
      const i32 aU = (i32)rank_in(Um, lane);
      Lm = ballot(v <= (i32)i0 - aU);  // a_k <= |U below k|  =>  { v <= i0 - |U below k| } is surely accepted
    }
    const u32 a = rank_in(Lm, lane);
    const u32 idx = i0 - a;
    const u32 need = i0 - lo + 1;  // draws left in this segment
    const u32 total = popc64(Lm);
    u32 used;
    if (total >= need) {  // the segment completes inside this batch: stop after its last accepted word
      const u64 last = ballot(a == need - 1) & Lm;
      used = need;
      r.pos = r.cbase + ctz64(last) + 1;
    } else {  // every cached word is consumed (accepted or rejected)
      used = total;
      r.pos = r.cbase + r.ccount;
    }
    if (v <= (i32)idx && a < used) J[idx] = (u32)v;
    i0 -= used;
  }
  u32 JL = 0;
  shuffle_small<1>(r, JL, i0 + 1, lane);
  if (lane >= 1 && lane <= i0) J[lane] = JL;
  wave_sync();
}
#endif

// Applies swap(x[i], x[J[i]]) for i = len-1 .. 1 to the list held across lanes (L0: elements 0..63, L1: 64..)
DEVINL void shuffle_apply(u32& L0, u32& L1, u32 len, const u32* J, u32 lane) {
  const u32 J0 = J[lane], J1 = J[64 + lane];
  u32 l0 = L0, l1 = L1;
  u32 i = len - 1;
  for (; i >= 64; --i) {  // x[i] lives in L1, x[j] in either register
    const u32 j = rdl(J1, i - 64);
    const u32 jm = j & 63, im = i - 64;
    const u32 vi = rdl(l1, im), a0 = rdl(l0, jm), b0 = rdl(l1, jm);
    const bool jlow = j < 64;
    // x[j] <- vi in whichever register holds it (the other one is rewritten with its own value),
    // then x[i] <- old x[j]; the last write wins when both are the same slot
    l0 = writelane<0>(jlow ? vi : a0, jm, l0);
    l1 = writelane<0>(jlow ? b0 : vi, jm, l1);
    l1 = writelane<0>(jlow ? a0 : b0, im, l1);
  }
#pragma unroll 4
  for (; i >= 1; --i) {
    const u32 j = rdl(J0, i);
    const u32 vi = rdl(l0, i), vj = rdl(l0, j);
    l0 = writelane<0>(vj, i, l0);
    l0 = writelane<0>(vi, j, l0);
  }
  L0 = l0;
  L1 = l1;
}

// The same list update without the serial chain.  Position i is final right after step i (later steps have
// smaller i and j <= i), and what step i leaves there is the content of position J[i] at that time:
//   - C(x), the content of position x just before its own step x, is C(G[x]) with G[x] = the smallest step
//     i' > x that targets x (the last deposit before time x; a deposit at step i' is the old x[i'] = C(i')),
//     or the initial x[x] when no step targets it: a forest ascending in x, flattened by pointer doubling;
//   - final x[i] = C(F[i]) with F[i] = the smallest step i' > i with J[i'] == J[i], or the initial x[J[i]];
//   - a position that has no step of its own in the phase ends with C(smallest step targeting it).
// "Smallest step above i targeting p" is a find-first-set on a per-target 64-bit step mask built with LDS
// atomic ORs.  Two phases (steps len-1..64 with x[i] in L1, then steps 63..1 inside L0) keep the masks 64 bit.
// `scratch` is the LDS block holding J on entry; it is overwritten (needs 8 * len bytes).
DEVINL u32 ffs64_or(u64 m, u32 none) { return m ? (u32)__builtin_ctzll(m) : none; }
DEVINL u32 chase_roots(u32 parent) {  // parent[l] >= l; returns the root of every lane's chain
  u32 R = parent;
  for (int k = 0; k < 6; ++k) {
    const u32 R2 = bperm(R, R);
    if (ballot(R2 != R) == 0) break;
    R = R2;
  }
  return R;
}
DEVINL void shuffle_apply_par(u32& L0, u32& L1, u32 len, u32* scratch, u32 lane) {  // 64 < len <= 128
  const u32 n1 = len - 64;  // steps 64 .. len-1 live in lanes 0 .. n1-1
  const u32 J0 = lane >= 1 ? scratch[lane] : 0u;  // lane 0 has no step; J = 0 makes F = "first step targeting 0"
  const u32 J1 = lane < n1 ? scratch[64 + lane] : 0u;
  wave_sync();
  unsigned long long* M = reinterpret_cast<unsigned long long*>(scratch);
  const u64 above = (~1ull) << lane;  // steps after this lane's own
  // ---- phase 1
  M[lane] = 0ull;
  if (lane + 64 < len) M[64 + lane] = 0ull;
  wave_sync();
  if (lane < n1) atomicOr(&M[J1], 1ull << lane);
  wave_sync();
  {
    const u64 mF = lane < n1 ? (M[J1] & above) : 0ull;
    const u64 mG = lane < n1 ? (M[64 + lane] & above) : 0ull;
    const u64 mH = M[lane];
    const u32 R = chase_roots(ffs64_or(mG, lane));
    const u32 rF = bperm(R, ffs64_or(mF, 0u)), rH = bperm(R, ffs64_or(mH, 0u));
    const u32 v1 = bperm(L1, mF ? rF : (J1 & 63u)), v0 = bperm(L0, J1 & 63u), vH = bperm(L1, rH);
    const u32 nL1 = (mF == 0 && J1 < 64) ? v0 : v1;
    L1 = lane < n1 ? nL1 : L1;
    L0 = mH ? vH : L0;
  }
  wave_sync();
  // ---- phase 2
  M[lane] = 0ull;
  wave_sync();
  if (lane >= 1) atomicOr(&M[J0], 1ull << lane);
  wave_sync();
  {
    const u64 mF = M[J0] & above, mG = M[lane] & above;
    const u32 R = chase_roots(ffs64_or(mG, lane));
    const u32 rF = bperm(R, ffs64_or(mF, 0u));
    L0 = bperm(L0, mF ? rF : J0);
  }
  wave_sync();
}

DEVINL void consume_small(Rng& r, u32 len, u32 lane) {  // the stream words of a shuffle of len <= 64 entries, no swaps
#ifndef CE_SHUFFLE_DRAWS_SEGMENTED
  shuffle_consume(r, len, lane);
#else
  u32 d0 = 0;
  shuffle_small<2>(r, d0, len, lane);
#endif
}
DEVINL void shuffle_lanes1(Rng& r, u32& L0, u32 len, u32 lane) {  // len <= 64
  shuffle_small<0>(r, L0, len, lane);
}

// ----------------------------------------------------------------------------------------
// per-wave LDS
// ----------------------------------------------------------------------------------------
template <int KIND> struct alignas(16) WaveLds {
  u32 mt[kGen];
  u32 U[Geo<KIND>::UWORDS];
  uint8_t S[Geo<KIND>::SBYTES];
  uint8_t pmap[Geo<KIND>::PQUADS * 16];  // PCELLS bytes used; padded to whole 16-byte quads for the image copy
  u32 rgb[16];
};

// While a step is in flight, bit 7 of a padded-map byte marks "an agent stands here" (set right after
// update_moves / setup_agents); the cell code is the low 7 bits.  It turns every "is an agent on this
// cell" query of the beam / spawn code into the LDS read that is needed anyway.
constexpr uint8_t kAgentBit = 0x80;
constexpr u32 kCodeMask = 0x7fu;
// The LDS map of the grid kernels holds every cell code PRE-SCALED by 4 (bits 2..6): a map byte is then directly the
// byte offset of its colour in the LUT, which saves the observation pass one shift per pixel (32 VALU per step at
// n = 8).  HBM images, tables and the feature-vector kernels keep the plain CE_CELL_* codes.
constexpr u32 kScale = 2;
constexpr u32 kEmpty = CE_CELL_EMPTY << kScale, kWall = CE_CELL_WALL << kScale, kApple = CE_CELL_APPLE << kScale,
              kWaste = CE_CELL_WASTE << kScale, kRiver = CE_CELL_RIVER << kScale;
// Predicated store into the padded map without touching exec: lanes that are off aim at byte 0 — the corner of the
// view border, CE_CELL_EMPTY in both maps and never anything else — and write 0 there.  Two v_cndmask instead of a
// compare + exec save / restore + branch; the scalar unit is the busier one in these kernels.
// `a && load(...)` compiles to an exec save / branch / restore around the load; where the address is valid for every
// lane (list entries past the end point at byte 0) both sides are simply evaluated
DEVINL bool both(bool a, bool b) { return (bool)((u32)a & (u32)b); }
DEVINL u32 pm_sel(bool on, u32 idx) { return on ? idx : 0u; }
DEVINL void pm_put(uint8_t* pm, bool on, u32 idx, u32 val) { pm[on ? idx : 0u] = (uint8_t)(on ? val : 0u); }
constexpr u32 kCellPadMask = 0x7ffu;
DEVINL u32 cell_pad(u32 packed) { return packed & kCellPadMask; }
DEVINL u32 cell_rc(u32 packed) { return packed >> 16; }  // col | row << 8: one byte per coordinate

// env context: everything a wave keeps in registers for its env
template <int KIND> struct Env {
  typedef Geo<KIND> G;
  WaveLds<KIND>* L;
  Rng rng;
  u32 lane, n, e;
  bool is_agent;
  // lane a < n: agent a
  u32 P;    // padded cell index of the agent
  u32 O;    // orientation UP0 RIGHT1 DOWN2 LEFT3
  i32 RW;   // reward_this_turn
  // persistent lists across lanes
  u32 SP;      // spawn list entry k in lane k (k < 20)
  u32 WP0, WP1;  // waste list entries k / 64 + k
  // static lane data: packed cells handled by this lane in round r
  u32 AP[3];
  u32 WS[2];
  unsigned long long* dbg;  // diagnostic builds only
  bool waste_perm_dirty;    // the persistent waste list was shuffled in this launch
  bool wt;                  // this launch writes its views and the MT19937 row through the L2 (write_obs, store_rng); wave-uniform
  // The layout the env is built from (env_geometry): the static tables and the lengths of its cell lists.  For the shipped layout
  // these are literals — everything is inlined into the kernel, so they fold exactly like the Geo<KIND> constants they stand
  // for; a kernel instance for a caller's layout (ce_config.ascii_map, CM = true) reads them from the parameter block.
  const GridTables* T;
  u32 napple, nwaste, randw, nspawn;  // apple / waste cells, stream words of the spawn model's rand() call, 'P' cells
  u32 cells, mapw;                    // rows x columns and columns of the layout (the beam_map's shape)
};

DEVINL i32 dir_delta(int PW, u32 o) {  // ORIENTATIONS map_env.py:22 as padded-index deltas
  return o == 0 ? -PW : o == 1 ? 1 : o == 2 ? PW : -1;
}

template <int KIND, bool CM> DEVINL void env_geometry(Env<KIND>& E, const GridParams& p) {
  typedef Geo<KIND> G;
  if (CM) {
    E.T = (const GridTables*)p.tab;
    E.napple = p.napple;
    E.nwaste = p.nwaste;
    E.randw = 2u * (p.napple + p.nwaste);
    E.nspawn = p.nspawn;
    E.cells = p.map_h * p.map_w;
    E.mapw = p.map_w;
  } else {
    E.T = &c_tab[KIND];
    E.napple = (u32)G::NAPPLE;
    E.nwaste = (u32)G::NWASTE;
    E.randw = (u32)G::RANDW;
    E.nspawn = (u32)G::NSPAWN_CTOR;
    E.cells = (u32)(G::H * G::W);
    E.mapw = (u32)G::W;
  }
}

template <int KIND> DEVINL u32 pad_of(u32 row, u32 col) { return __umul24(row + kView, (u32)Geo<KIND>::PW) + col + kView; }
// Exact small-range divisions by multiply-shift with 24-bit multiplies (v_mul_u32_u24 is full rate, the
// 32-bit v_mul_lo/hi the compiler emits for `/ constant` are quarter rate): valid for idx < 640 / pad < 1600.
template <int KIND> DEVINL u32 div_pw(u32 pad) {  // pad / PW
  return KIND == CE_KIND_CLEANUP ? (__umul24(pad, 1821u) >> 16) : (__umul24(pad, 1261u) >> 16);  // / 36, / 52
}
template <int KIND> DEVINL u32 row_of(u32 pad) { return div_pw<KIND>(pad) - kView; }
template <int KIND> DEVINL u32 col_of(u32 pad) { return pad - __umul24(div_pw<KIND>(pad), (u32)Geo<KIND>::PW) - kView; }

// is some agent standing on padded cell `cell` (per-lane query); returns highest agent id + 1 or 0
template <int KIND> DEVINL void mark_agents(Env<KIND>& E) {
  wave_sync();
  pm_put(E.L->pmap, E.is_agent, E.P, E.L->pmap[pm_sel(E.is_agent, E.P)] | kAgentBit);
  wave_sync();
}
template <int KIND> DEVINL u32 agent_on(const Env<KIND>& E, u32 cell) {
  u32 hit = 0;
  for (u32 b = 0; b < E.n; ++b) {
    u32 pb = rdl(E.P, b);
    hit = (cell == pb) ? b + 1 : hit;
  }
  return hit;
}

// ----------------------------------------------------------------------------------------
// state load / store
// ----------------------------------------------------------------------------------------
template <int KIND> DEVINL void load_static(Env<KIND>& E) {
  const GridTables& T = *E.T;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    u32 idx = E.lane + 64 * r;
    E.AP[r] = idx < E.napple ? T.apple[idx < 160 ? idx : 0] : 0;
  }
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    u32 idx = E.lane + 64 * r;
    E.WS[r] = idx < E.nwaste ? T.waste[idx < 128 ? idx : 0] : 0;
  }
  if (E.lane < 16) E.L->rgb[E.lane] = c_rgb[E.lane];
}

// counter mode: the env's `rng` row is (key0, key1, generation, pad) — one scalar load; the stream starts "dry", the first
// draw of the operation fills LDS with generation + 1
DEVINL void load_rng_counter(Rng& r, const GridParams& p, u32 e) {
  const auto row = (CE_GPTR(const u32))(p.rng + (size_t)e * kRngRow);
  r.k0 = rfl(row[0]);
  r.k1 = rfl(row[1]);
  r.gen0 = rfl(row[2]);
  r.pos = kGen;
}
template <int KIND> DEVINL void load_rng(Env<KIND>& E, const GridParams& p) {
  E.rng.mt = E.L->mt;
  if (kCounterRng) {
    load_rng_counter(E.rng, p, E.e);
  } else {
    const uint4* src = (const uint4*)(p.rng + (size_t)E.e * kRngRow);
    uint4* dst = (uint4*)E.L->mt;
    for (u32 k = E.lane; k < kMtN / 4; k += 64) dst[k] = src[k];
    E.rng.pos = rfl(p.rng[(size_t)E.e * kRngRow + kMtN]);
  }
  E.rng.cbase = 0;
  E.rng.ccount = 0;
  E.rng.cvalid = 0;
  E.rng.cache = 0;
  E.rng.twists = 0;
  wave_sync();
  rng_begin_op(E.rng, E.lane);
}
// `wt`: the key words leave through the L2 (sc1) like the views of the same launch (write_obs: single-step launches of handles
// that fit the Infinity Cache) — 2.5 KB per env that would otherwise sit dirty in the L2 until the kernel's end
template <int KIND> DEVINL void store_rng(Env<KIND>& E, const GridParams& p, bool wt = false) {
  wave_sync();
  if (kCounterRng) {  // only the generation number moves
    if (E.lane == 0) p.rng[(size_t)E.e * kRngRow + 2] = E.rng.gen0 + E.rng.twists;
    return;
  }
  if (rfl(E.rng.twists) != 0 && !diag::ablate_twist) {  // the key words only change at a twist; otherwise just the position moves
    uint4* dst = (uint4*)(p.rng + (size_t)E.e * kRngRow);
    const uint4* src = (const uint4*)E.rng.mt;
    const u32 q2 = min(E.lane + 128u, (u32)kMtN / 4 - 1);  // unconditional: idle lanes repeat the last quad
    const uint4 r0 = src[E.lane], r1 = src[E.lane + 64], r2 = src[q2];
#ifndef CE_RNG_WT_OFF
    if (wt) {
      typedef u32 u32x4 __attribute__((ext_vector_type(4)));
      const auto base = (CE_GPTR(char))dst;
      const u32 o = E.lane << 4;
      const u32x4 v0 = {r0.x, r0.y, r0.z, r0.w}, v1 = {r1.x, r1.y, r1.z, r1.w}, v2 = {r2.x, r2.y, r2.z, r2.w};
      asm volatile("global_store_dwordx4 %0, %1, %2 sc1\n\ts_nop 1" ::"v"(o), "v"(v0), "s"(base) : "memory");
      asm volatile("global_store_dwordx4 %0, %1, %2 offset:1024 sc1\n\ts_nop 1" ::"v"(o), "v"(v1), "s"(base) : "memory");
      if (E.lane + 128u < (u32)kMtN / 4)  // (a write-through store is a fabric write of its own: no repeats here)
        asm volatile("global_store_dwordx4 %0, %1, %2 offset:2048 sc1\n\ts_nop 1" ::"v"(o), "v"(v2), "s"(base) : "memory");
    } else
#endif
    {
      dst[E.lane] = r0;
      dst[E.lane + 64] = r1;
      dst[q2] = r2;
    }
  }
  if (E.lane == 0) p.rng[(size_t)E.e * kRngRow + kMtN] = E.rng.pos;
}

template <int KIND> DEVINL void zero_pmap(Env<KIND>& E) {
  typedef Geo<KIND> G;
  u32* pm = (u32*)E.L->pmap;
  for (u32 k = E.lane; k < (u32)G::PCELLS / 4; k += 64) pm[k] = 0;
  wave_sync();
}

// lane's bit of a wave-uniform 64-bit mask
DEVINL bool lane_bit(u64 m, u32 lane) { return (((lane < 32 ? (u32)m : (u32)(m >> 32)) >> (lane & 31u)) & 1u) != 0; }
// if_set where the lane's bit of the wave-uniform mask is set, else if_clear: a 64-bit SGPR pair IS a lane mask, so this
// is one v_cndmask (the shift / and / compare form of lane_bit is four)
DEVINL u32 select_by_mask(u64 m, u32 if_set, u32 if_clear) { return __builtin_amdgcn_inverse_ballot_w64(m) ? if_set : if_clear; }

// The map image in LDS = constant base map + the env's presence bits (apples; cleanup: waste vs river).
// `bits` = the env's 8 state dwords in SGPRs (loaded by the caller with one scalar load).
template <int KIND> DEVINL void paint_presence(Env<KIND>& E, const u32 (&bits)[8]) {
  typedef Geo<KIND> G;
  uint8_t* pm = E.L->pmap;
  constexpr int AR = (G::NAPPLE + 63) / 64;
#pragma unroll
  for (int r = 0; r < AR; ++r) {
    const u64 m = (u64)bits[2 * r] | (u64)bits[2 * r + 1] << 32;
    pm_put(pm, E.lane + 64 * r < E.napple, cell_pad(E.AP[r]), select_by_mask(m, kApple, kEmpty));
  }
  if (KIND == CE_KIND_CLEANUP) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const u64 m = (u64)bits[4 + 2 * r] | (u64)bits[5 + 2 * r] << 32;
      pm_put(pm, E.lane + 64 * r < E.nwaste, cell_pad(E.WS[r]), select_by_mask(m, kWaste, kRiver));
    }
  }
}
// the env's 8 state dwords from presence ballots: presA[r] = apple on apple cell lane + 64 r, presW[r] likewise for waste
template <int KIND> DEVINL void store_grid_bits(Env<KIND>& E, const GridParams& p, const u64 (&presA)[3], const u64 (&presW)[2], bool blank = false) {
  typedef Geo<KIND> G;
  constexpr int AR = (G::NAPPLE + 63) / 64;
  u32 w = 0;  // lane k < 8 assembles state dword k
#pragma unroll
  for (int r = 0; r < AR; ++r) {
    if (E.lane == 2 * r) w = (u32)presA[r];
    if (E.lane == 2 * r + 1) w = (u32)(presA[r] >> 32);
  }
  if (KIND == CE_KIND_CLEANUP) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      if (E.lane == 4 + 2 * r) w = (u32)presW[r];
      if (E.lane == 5 + 2 * r) w = (u32)(presW[r] >> 32);
    }
  }
  if (blank && E.lane == 7) w |= 1u << (kGridBlankBit & 31);
  if (E.lane < 8) GAT((CE_GPTR(u32))(p.grid + (size_t)E.e * kGridStateBytes), E.lane) = w;
}
template <int KIND> DEVINL void store_grid(Env<KIND>& E, const GridParams& p, bool blank = false) {
  typedef Geo<KIND> G;
  wave_sync();
  const uint8_t* pm = E.L->pmap;
  constexpr int AR = (G::NAPPLE + 63) / 64;
  u64 presA[3] = {0, 0, 0}, presW[2] = {0, 0};
#pragma unroll
  for (int r = 0; r < AR; ++r)
    presA[r] = ballot(both(E.lane + 64 * r < E.napple, (pm[cell_pad(E.AP[r])] & kCodeMask) == kApple));
  if (KIND == CE_KIND_CLEANUP) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
      presW[r] = ballot(both(E.lane + 64 * r < E.nwaste, (pm[cell_pad(E.WS[r])] & kCodeMask) == kWaste));
  }
  store_grid_bits(E, p, presA, presW, blank);
}

template <int KIND> DEVINL void load_agents(Env<KIND>& E, const GridParams& p) {
  u32 w = 0;
  if (E.is_agent) w = GAT((CE_GPTR(const u32))p.agents + (size_t)E.e * E.n, E.lane);
  E.P = pad_of<KIND>(w & 0xff, (w >> 8) & 0xff);
  E.O = (w >> 16) & 3;
  E.RW = 0;
  if (!E.is_agent) E.P = 0xffffu;  // never equals a real cell
}
template <int KIND> DEVINL void store_agents(Env<KIND>& E, const GridParams& p) {
  if (E.is_agent) {
    u32 w = row_of<KIND>(E.P) | (col_of<KIND>(E.P) << 8) | (E.O << 16);
    GAT((CE_GPTR(u32))p.agents + (size_t)E.e * E.n, E.lane) = w;
  }
}
template <int KIND> DEVINL void load_perms(Env<KIND>& E, const GridParams& p) {
  E.SP = E.lane < 20 ? GAT(p.spawn_perm + (size_t)E.e * 20, E.lane) : 0;
  E.WP0 = E.WP1 = 0;
  if (KIND == CE_KIND_CLEANUP) {
    const auto wp = p.waste_perm + (size_t)E.e * 119;
    E.WP0 = GAT(wp, E.lane);
    E.WP1 = E.lane + 64 < 119 ? GAT(wp, E.lane + 64) : 0;
  }
}
template <int KIND> DEVINL void store_perms(Env<KIND>& E, const GridParams& p, bool spawn_too, bool waste_too = true) {
  if (spawn_too && E.lane < 20) GAT(p.spawn_perm + (size_t)E.e * 20, E.lane) = (uint8_t)E.SP;
  if (KIND == CE_KIND_CLEANUP && waste_too) {
    const auto wp = p.waste_perm + (size_t)E.e * 119;
    GAT(wp, E.lane) = (uint8_t)E.WP0;
    if (E.lane + 64 < 119) GAT(wp, E.lane + 64) = (uint8_t)E.WP1;
  }
}

// All of an env's state in one go: every global load is issued first (they overlap in flight), then the
// LDS image (MT words, padded map) is built.  Used by the step kernel.
template <int KIND> DEVINL void load_env_state(Env<KIND>& E, const GridParams& p) {
  typedef Geo<KIND> G;
  const GridTables& T = *E.T;
  const u32 lane = E.lane;
  // Every load and LDS store below is unconditional: a lane past the end of a list repeats the last element (clamped
  // index, same value to the same address).  A divergent `if` around a single load costs a compare, an exec save /
  // restore pair and a branch — scalar-unit work, which is what this kernel has least of.
  const uint4* rsrc = (const uint4*)(p.rng + (size_t)E.e * kRngRow);
  constexpr u32 kLastQuad = (u32)kMtN / 4 - 1;
  const u32 q2 = min(lane + 128u, kLastQuad);
  uint4 r0 = {}, r1 = {}, r2 = {};
  u32 rpos = kGen;
  if (kCounterRng) {
    load_rng_counter(E.rng, p, E.e);
  } else {
    r0 = rsrc[lane], r1 = rsrc[lane + 64], r2 = rsrc[q2];
    rpos = p.rng[(size_t)E.e * kRngRow + kMtN];
  }
  // map: the constant base image, codes pre-scaled (L2-resident, shared by every env), as two rounds of 16-byte copies
  // + this env's 8 presence dwords (one scalar load)
  const uint4* gsrc = (const uint4*)T.base_pmap4;
  static_assert(G::PQUADS > 64 && G::PQUADS <= 128 && G::PQUADS * 16 <= (int)sizeof(T.base_pmap4), "two quad rounds");
  constexpr u32 kLastMapQuad = (u32)G::PQUADS - 1;
  const u32 gq1 = min(lane + 64u, kLastMapQuad);  // unconditional: idle lanes repeat the last quad
  const uint4 gw0 = gsrc[lane], gw1 = gsrc[gq1];
  u32 gbits[8];
  {
    const auto bsrc = (CE_GPTR(const u32))(p.grid + (size_t)E.e * kGridStateBytes);
#pragma unroll
    for (int k = 0; k < 8; ++k) gbits[k] = bsrc[k];
  }
  const u32 aw = GAT((CE_GPTR(const u32))p.agents + (size_t)E.e * E.n, min(lane, E.n - 1u));
  E.SP = 0;  // the 20-entry spawn list is only needed by a reset: it stays in HBM and is fetched there (grid_step_core)
  E.WP0 = E.WP1 = 0;
  if (KIND == CE_KIND_CLEANUP) {
    const auto wp = p.waste_perm + (size_t)E.e * 119;
    E.WP0 = GAT(wp, lane);
    const u32 w1 = GAT(wp, min(lane + 64u, 118u));
    E.WP1 = lane + 64 < 119 ? w1 : 0;
  }
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const u32 idx = lane + 64 * r;
    const u32 v = T.apple[min(idx, E.napple - 1u)];
    E.AP[r] = idx < E.napple ? v : 0;
  }
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const u32 idx = lane + 64 * r;
    const u32 v = T.waste[G::NWASTE ? min(idx, E.nwaste - 1u) : 0u];
    E.WS[r] = idx < E.nwaste ? v : 0;
  }
  const u32 rgbv = c_rgb[lane & 15];
  if (kCounterRng) {  // the step's generation, computed under the loads above
    E.rng.mt = E.L->mt;
    E.rng.twists = 0;
    rng_begin_op(E.rng, lane, true);
  }
  // ---- LDS image ----
  uint4* pm128 = (uint4*)E.L->pmap;
  pm128[lane] = gw0;
  pm128[gq1] = gw1;
  wave_sync();
  paint_presence(E, gbits);
  E.L->rgb[lane & 15] = rgbv;
  if (!kCounterRng) {
    uint4* mt4 = (uint4*)E.L->mt;
    mt4[lane] = r0;
    mt4[lane + 64] = r1;
    mt4[q2] = r2;
  }
  E.rng.mt = E.L->mt;
  E.rng.pos = kCounterRng ? 0u : rfl(rpos);
  E.rng.cbase = 0;
  E.rng.ccount = 0;
  E.rng.cvalid = 0;
  E.rng.cache = 0;
  if (!kCounterRng) E.rng.twists = 0;
  E.P = pad_of<KIND>(aw & 0xff, (aw >> 8) & 0xff);
  E.O = (aw >> 16) & 3;
  E.RW = 0;
  if (!E.is_agent) E.P = 0xffffu;
  wave_sync();
}

// ----------------------------------------------------------------------------------------
// update_moves (map_env.py:483-676)
// ----------------------------------------------------------------------------------------
template <int KIND> DEVINL void update_moves(Env<KIND>& E, u32 ACT) {
  typedef Geo<KIND> G;
  const u32 lane = E.lane;
  const bool mover = E.is_agent && ACT <= 4;
  // phase A: targets (rotate_action + return_valid_pos), turns take effect immediately
  i32 v0 = ACT == 2 ? -1 : ACT == 3 ? 1 : 0;  // MOVE_UP / MOVE_DOWN row component
  i32 v1 = ACT == 0 ? -1 : ACT == 1 ? 1 : 0;  // MOVE_LEFT / MOVE_RIGHT col component
  i32 dr, dc;
  if (E.O == 0) { dr = v0; dc = v1; }          // UP: as is
  else if (E.O == 3) { dr = -v1; dc = v0; }    // LEFT: rotate_left
  else if (E.O == 1) { dr = v1; dc = -v0; }    // RIGHT: rotate_right
  else { dr = -v0; dc = -v1; }                 // DOWN: two rotate_left
  u32 tgt = E.P;
  if (mover) {
    u32 cand = (u32)((i32)E.P + dr * G::PW + dc);
    if (E.L->pmap[cand] != kWall) tgt = cand;
  }
  const u32 TGT0 = tgt;
  if (E.is_agent && (ACT == 5 || ACT == 6)) E.O = (E.O + (ACT == 5 ? 1u : 3u)) & 3u;
  const u64 M = ballot(mover);
  if (M == 0) return;
  const u32 m = popc64(M);
  // Interaction-free steps (the common case): no two movers share a target and no mover's target is another
  // agent's cell.  Then phase B has nothing to arbitrate and every pass-C entry either finds its cell empty
  // (targets are distinct from every live position) or is the agent itself (stay / wall), so the outcome is
  // P <- target whatever the shuffled order: only the stream words of the shuffle are consumed.
  if (E.n <= 8) {  // one (a, b) agent pair per lane
    const u32 a = lane >> 3, b = lane & 7u;
    const u32 ta = bperm(TGT0, a), pb = bperm(E.P, b), tb = bperm(TGT0, b);
    const u32 mlo = (u32)M;
    const bool ma = ((mlo >> a) & 1u) != 0, mb = ((mlo >> b) & 1u) != 0;
    const bool clash = ma && a != b && b < E.n && (ta == pb || (mb && ta == tb));
    if (ballot(clash) == 0) {
      consume_small(E.rng, m, lane);
      if (mover) E.P = TGT0;
      return;
    }
  }
  // slot list in agent order, then np.random.shuffle of the (agent, slot) pairs
  u32 SA = 0;
  {
    u64 mm = M;
    u32 s = 0;
    while (mm) {
      SA = wrl(ctz64(mm), s, SA);
      mm &= mm - 1;
      ++s;
    }
  }
  shuffle_lanes1(E.rng, SA, m, lane);
  const bool valid_s = lane < m;
  const u32 T = bperm(TGT0, valid_s ? SA : 0);  // target of the agent at shuffled position `lane`
  u32 cnt = 0;
  for (u32 s2 = 0; s2 < m; ++s2) cnt += (rdl(T, s2) == T) ? 1u : 0u;
  u32 MOVE = TGT0;  // agent_moves[a]
  u64 HM = M;       // keys of agent_moves
  u32 BYPOS = E.P;  // positions agent_by_pos was last built from
  u64 todo = ballot(valid_s && cnt > 1);
  // phase B: contested cells in ascending (row, col) order
  while (todo) {
    u32 cmin = 0xffffffffu;
    for (u64 tt = todo; tt; tt &= tt - 1) {
      u32 tv = rdl(T, ctz64(tt));
      cmin = tv < cmin ? tv : cmin;
    }
    const u64 Gm = ballot(valid_s && T == cmin);
    todo &= ~Gm;
    const u32 w = rdl(SA, ctz64(Gm));            // first contender in shuffled order
    const u64 CA = ballot(mover && TGT0 == cmin);  // contenders as agent lanes
    const u64 occ = ballot(E.is_agent && E.P == cmin);
    bool free_cell = true;
    if (occ) {
      const u64 bp = ballot(E.is_agent && BYPOS == cmin);
      const u32 conf = fls64(bp | 1ull);
      const bool conf_has = bit(HM, conf);
      const u32 cpos = rdl(E.P, conf);
      const u32 cmove = conf_has ? rdl(MOVE, conf) : cpos;
      if (bit(CA, conf)) free_cell = false;                      // (1) a contender is the occupant
      else if (!conf_has || cpos == cmove) free_cell = false;    // (2) occupant stays / has no move
      else if (cmin == cpos && ballot(bit(CA, lane) && E.P == cmove) != 0) free_cell = false;  // (3) swap
    }
    if (free_cell) {
      E.P = wrl(cmin, w, E.P);
      BYPOS = E.P;
    }
    if (bit(CA, lane)) MOVE = E.P;  // every contender's move := its live position
  }
  // phase C: passes over the remaining moves in insertion (agent) order
  while (HM) {
    BYPOS = E.P;
    const u64 snap = HM;
    const u32 before = popc64(HM);
    u64 del = 0;
    for (u64 it = snap; it; it &= it - 1) {
      const u32 x = ctz64(it);
      if (bit(del, x)) continue;
      const u32 mv = rdl(MOVE, x);
      const u64 occ = ballot(E.is_agent && E.P == mv);
      if (occ) {
        const u64 bp = ballot(E.is_agent && BYPOS == mv);
        if (bp == 0) {  // reference would raise KeyError; unreachable (targets are unique after phase B)
          HM &= ~(1ull << x);
          del |= 1ull << x;
          continue;
        }
        const u32 conf = fls64(bp);
        const u32 cpos = rdl(E.P, conf);
        const u32 cmove = bit(HM, conf) ? rdl(MOVE, conf) : cpos;
        const u32 xpos = rdl(E.P, x);
        if (x == conf) {
          HM &= ~(1ull << x);
          del |= 1ull << x;
        } else if (!bit(snap, conf) || cpos == cmove) {
          HM &= ~(1ull << x);
          del |= 1ull << x;
        } else if (cmove == xpos && mv == cpos) {
          HM &= ~((1ull << x) | (1ull << conf));
          del |= (1ull << x) | (1ull << conf);
        }
      } else {
        E.P = wrl(mv, x, E.P);
        HM &= ~(1ull << x);
        del |= 1ull << x;
      }
    }
    if (popc64(HM) == before) {  // cycle: move everyone that is left
      if (bit(HM, lane)) E.P = MOVE;
      break;
    }
  }
}

// ----------------------------------------------------------------------------------------
// beams (update_map_fire map_env.py:721-814); lanes 0..14 = 3 rays x 5 cells
// returns number of cells cleaned (CLEAN) — hits are applied to E.RW (FIRE)
// ----------------------------------------------------------------------------------------
// render support (CE_FLAG_BEAM_TRACE): MapEnv.beam_pos as a per-cell map, cleared at step entry and by reset()
template <int KIND> DEVINL void clear_beam_map(Env<KIND>& E, const GridParams& p) {
  const auto bm = p.beam_map + (size_t)E.e * E.cells;
  for (u32 k = E.lane; k < E.cells; k += 64) GAT(bm, k) = CE_BEAM_NONE;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the beams of this step overwrite these bytes from other lanes
}

template <int KIND> DEVINL u32 fire_beam(Env<KIND>& E, const GridParams& p, u32 firer, bool is_clean) {
  typedef Geo<KIND> G;
  const u32 lane = E.lane;
  const u32 o = rdl(E.O, firer), p0 = rdl(E.P, firer);
  const i32 dd = dir_delta(G::PW, o), rr = dir_delta(G::PW, (o + 1) & 3);  // right(dir(o)) == dir(o+1)
  const u32 ray = lane / 5, step = lane - 5 * ray;
  const bool in_beam = lane < 15;
  i32 start = (i32)p0;
  if (ray == 1) start += rr - dd;
  if (ray == 2) start += -rr - dd;
  const u32 cell = in_beam ? (u32)(start + (i32)(step + 1) * dd) : 0u;
  const u32 raw = E.L->pmap[cell];
  const u32 code = raw & kCodeMask;
  const bool invalid = code == kWall;
  const bool agent_here = in_beam && (raw & kAgentBit) != 0;
  const bool stopper = invalid || agent_here || (is_clean && code == kWaste);
  const u64 S = ballot(in_beam && stopper);
  const u32 rb = (u32)(S >> (5 * ray)) & 31u;
  const u32 f = rb ? (u32)__builtin_ctz(rb) : 5u;  // first stopping cell of this ray
  const bool processed = in_beam && (step < f || (step == f && !invalid));
  if (p.flags & CE_FLAG_BEAM_TRACE) {  // firing_points (map_env.py:788,813); the walls around the map keep them inside
    const auto bm = p.beam_map + (size_t)E.e * E.cells;
    if (processed) GAT(bm, __umul24(row_of<KIND>(cell), E.mapw) + col_of<KIND>(cell)) = is_clean ? CE_BEAM_CLEAN : CE_BEAM_FIRE;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // a later beam of this step overwrites in list order
  }
  u32 cleaned = 0;
  if (is_clean) {
    const bool upd = processed && code == kWaste;
    cleaned = popc64(ballot(upd));
    wave_sync();
    pm_put(E.L->pmap, upd, cell, kRiver | (raw & kAgentBit));
    wave_sync();
  } else {
    const u64 hm0 = ballot(processed && agent_here);
    if (hm0) {
      const u32 hit = agent_on(E, cell);  // highest agent id on the cell + 1 (agent_by_pos: later wins)
      for (u64 hm = hm0; hm; hm &= hm - 1) {
        const u32 hid = rdl(hit, ctz64(hm)) - 1;
        if (lane == hid) E.RW -= 50;  // Agent.hit(b"F")
      }
    }
  }
  return cleaned;
}

// ----------------------------------------------------------------------------------------
// spawn models
// ----------------------------------------------------------------------------------------
// The rand(k) call of the spawn models consumes RANDW stream words every step, but only a few are ever looked at:
// the first word of the double of each eligible apple cell (the 27 high bits decide x < threshold except on an
// exact tie, which falls back to the second word), and — for the waste walk — only whether a double is < 0.5,
// i.e. the sign of its tempered first word, which is the parity of four raw state bits (temper is GF(2)-linear:
// bit31(temper(w)) = w31 ^ w27 ^ w24 ^ w16).  So the window [pos, pos + RANDW) is never materialised: each lane
// fetches its own words straight from the MT state, before the in-place twist for the part of the window that
// lies in the current generation and after it for the rest.
struct StreamWindow {
  u32* mt;
  u32 pos, alen;  // window start in the current generation; words of the window that lie in it
};
DEVINL StreamWindow window_open(Rng& r, u32 lane, u32 RANDW) {
  rng_assert_uniform(r);
  if (r.pos >= kGen) {
    rng_advance(r, lane);
    r.pos = 0;
  }
  StreamWindow w;
  w.mt = r.mt;
  w.pos = r.pos;
  const u32 left = kGen - r.pos;
  w.alen = left < (u32)RANDW ? left : (u32)RANDW;
  return w;
}
DEVINL u32 window_read_old(const StreamWindow& w, bool need, u32 s) {
  const bool ok = both(need, s < w.alen);  // branch-free: lanes without a word read mt[0]
  const u32 v = w.mt[ok ? w.pos + s : 0u];
  return ok ? v : 0u;
}
DEVINL u32 window_read_new(const StreamWindow& w, bool need, u32 s, u32 old) {
  const bool ok = both(need, s >= w.alen);
  const u32 v = w.mt[ok ? s - w.alen : 0u];
  return ok ? v : old;
}
DEVINL void window_close(Rng& r, const StreamWindow& w, u32 RANDW) {
  r.pos = w.alen < (u32)RANDW ? (u32)RANDW - w.alen : w.pos + (u32)RANDW;
  r.ccount = 0;
}
// x < thr for the 53-bit X = (a >> 5) << 26 | (b >> 6) of a double, from the tempered first word alone unless tied
DEVINL bool below_hi(u32 a_tempered, u64 thr, bool& tie) {
  const u32 a27 = a_tempered >> 5, hi = (u32)(thr >> 26);
  tie = a27 == hi;
  return a27 < hi;
}
DEVINL bool below_lo(u32 b_raw, u64 thr) { return (mt_temper(b_raw) >> 6) < ((u32)thr & 0x3ffffffu); }

template <int KIND> DEVINL void custom_map_update(Env<KIND>& E) {
  typedef Geo<KIND> G;
  const GridTables& T = *E.T;
  const u32 lane = E.lane;
  const u64 lt = (1ull << lane) - 1ull;
  uint8_t* pm = E.L->pmap;
  constexpr int AR = (G::NAPPLE + 63) / 64;  // lane rounds over the apple cells
  // eligible apple cells (apple cell, no apple, no agent: bit 7) and the index of the double each one is handed
  bool elig[AR];
  u32 sa[AR];
  u64 thrA[AR];
  u32 rbase = 0;
  u32 nH = 0;
  u64 hmask[2] = {0, 0};  // cleanup: waste present on static waste cell lane + 64 r (the map as the spawn model sees it)
  bool waste_on = false;
  bool scan = true;  // wave-uniform: some cell can spawn this step
  if (KIND == CE_KIND_CLEANUP) {
    // compute_probabilities: #H on the map -> host-precomputed 53-bit threshold
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const bool v = lane + 64 * r < E.nwaste;
      hmask[r] = ballot(both(v, (pm[cell_pad(E.WS[r])] & kCodeMask) == kWaste));
      nH += popc64(hmask[r]);
    }
    const u64 th = T.apple_thresh[nH];
    waste_on = (th & kWasteOnBit) != 0;
    // waste density >= 0.4 (cleanup_new.py:357-359): both probabilities are zero, so nothing can spawn and the rand(222)
    // call only moves the stream — about half the steps of a steady-state episode (the density hovers at the threshold:
    // a spawned waste switches the model off until the next one is cleaned).  The per-cell work below is skipped then;
    // the stream bookkeeping (window, twist) is shared with the full path.
    scan = th != 0;
  }
#pragma unroll
  for (int r = 0; r < AR; ++r) {
    elig[r] = false;
    sa[r] = 0;
    thrA[r] = 0;
  }
  if (scan) {
#pragma unroll
  for (int r = 0; r < AR; ++r) {
    const bool v = lane + 64 * r < E.napple;
    const u32 cell = cell_pad(E.AP[r]);
    elig[r] = both(v, pm[cell] == kEmpty);
    if (KIND == CE_KIND_CLEANUP) {
      thrA[r] = T.apple_thresh[nH] & ~kWasteOnBit;
    } else {
      // apples in the 3x3 block around the cell (j^2 + k^2 <= APPLE_RADIUS = 2), pre-update map
      u32 num = 0;
      if (elig[r]) {
#pragma unroll
        for (int j = -1; j <= 1; ++j)
#pragma unroll
          for (int k = -1; k <= 1; ++k) num += pm[(i32)cell + j * G::PW + k] == kApple ? 1u : 0u;
      }
      thrA[r] = T.apple_thresh[num < 3 ? num : 3];
    }
    const u64 eb = ballot(elig[r]);
    sa[r] = 2 * (rbase + popc64(eb & lt));
    rbase += popc64(eb);
  }
  }
  CE_SUBSTAMP(11);
  // waste walk candidates: the t-th non-waste cell of the (shuffled) list gets double rbase + t
  const u32 ncand = E.nwaste - nH;
  bool needw[2] = {false, false};
  u32 sw[2] = {0, 0};
  if (KIND == CE_KIND_CLEANUP) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const u32 t = lane + 64 * r;
      needw[r] = waste_on && t < ncand;
      sw[r] = 2 * (rbase + t);
    }
  }
  StreamWindow W = window_open(E.rng, lane, E.randw);
  u32 wa[AR], wb[AR], ww[2] = {0, 0};
#pragma unroll
  for (int r = 0; r < AR; ++r) wa[r] = wb[r] = 0;
  if (scan) {
#pragma unroll
    for (int r = 0; r < AR; ++r) {
      wa[r] = window_read_old(W, elig[r], sa[r]);
      wb[r] = window_read_old(W, elig[r], sa[r] + 1);
    }
    if (KIND == CE_KIND_CLEANUP) {
#pragma unroll
      for (int r = 0; r < 2; ++r) ww[r] = window_read_old(W, needw[r], sw[r]);
    }
  }
  if (W.alen < E.randw) {  // the window runs into the next generation
    rng_advance(E.rng, lane);
    if (scan) {
#pragma unroll
      for (int r = 0; r < AR; ++r) {
        wa[r] = window_read_new(W, elig[r], sa[r], wa[r]);
        wb[r] = window_read_new(W, elig[r], sa[r] + 1, wb[r]);
      }
      if (KIND == CE_KIND_CLEANUP) {
#pragma unroll
        for (int r = 0; r < 2; ++r) ww[r] = window_read_new(W, needw[r], sw[r], ww[r]);
      }
    }
  }
  window_close(E.rng, W, E.randw);
  if (!scan) return;
  bool spawnA[AR], tie[AR];
  bool any_tie = false;
#pragma unroll
  for (int r = 0; r < AR; ++r) {
    spawnA[r] = below_hi(mt_temper(wa[r]), thrA[r], tie[r]) && elig[r];
    tie[r] = tie[r] && elig[r];
    any_tie = any_tie || tie[r];
  }
  if (ballot(any_tie) != 0) {  // exact tie of the 27 high bits (once in 2^27 doubles): the second word decides
#pragma unroll
    for (int r = 0; r < AR; ++r)
      if (tie[r]) spawnA[r] = below_lo(wb[r], thrA[r]);
  }
  CE_SUBSTAMP(12);
  u32 waste_cell = 0;
  bool waste_found = false;
  if (KIND == CE_KIND_CLEANUP) {
    if (waste_on && !diag::ablate_shuffle) {
      // The walk over the shuffled list stops at the first candidate whose double is < 0.5: t* is the first
      // candidate index with a clear tempered sign bit, independent of the permutation.
      u32 tstar = 0xffffffffu;
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const u64 sb = ballot(needw[r] && (__builtin_popcount(ww[r] & 0x89010000u) & 1) == 0);
        if (tstar == 0xffffffffu && sb) tstar = ctz64(sb) + 64 * r;
      }
      if (diag::seq_shuffle) {
        shuffle_core<true>(E.rng, E.WP0, E.WP1, E.nwaste, lane);
      } else if (E.nwaste <= 64u) {  // a caller's small layout (never the shipped one: 119): the list lives in one register
        E.waste_perm_dirty = true;
        shuffle_lanes1(E.rng, E.WP0, E.nwaste, lane);
      } else {
        // U is free scratch: first the draw list J[0..118], then the step-mask table of the list update (which
        // spills into S)
        E.waste_perm_dirty = true;
        shuffle_draws(E.rng, E.nwaste, E.L->U, lane);
        CE_SUBSTAMP(10);
        static_assert(offsetof(WaveLds<KIND>, S) == offsetof(WaveLds<KIND>, U) + sizeof(E.L->U), "S must follow U");
        static_assert(KIND != CE_KIND_CLEANUP || sizeof(E.L->U) + sizeof(E.L->S) >= 8 * G::NWASTE, "step-mask table does not fit");
        if (diag::serial_apply) shuffle_apply(E.WP0, E.WP1, E.nwaste, E.L->U, lane);
        else shuffle_apply_par(E.WP0, E.WP1, E.nwaste, E.L->U, lane);
      }
      CE_SUBSTAMP(13);
      if (tstar != 0xffffffffu) {  // the tstar-th candidate in shuffled order gets the waste
        // both halves of the list at once.  Whether list entry w (a static waste-cell index) is a candidate is bit w of
        // the presence ballots taken above — no lookup of the entry's cell, no map read; only the chosen entry's cell
        // is fetched (a readlane of the static table held in E.WS)
        static_assert(KIND != CE_KIND_CLEANUP || (G::NWASTE > 64 && G::NWASTE <= 128), "two lane rounds");
        const bool v1 = lane + 64 < E.nwaste;
        const u32 w0 = E.WP0, w1 = v1 ? E.WP1 : 0u;
        const u64 h0 = w0 < 64u ? hmask[0] : hmask[1], h1 = w1 < 64u ? hmask[0] : hmask[1];
        const bool cand0 = ((h0 >> (w0 & 63u)) & 1ull) == 0;
        const bool cand1 = both(v1, ((h1 >> (w1 & 63u)) & 1ull) == 0);
        const u64 cb0 = ballot(cand0), cb1 = ballot(cand1);
        const u64 sel0 = ballot(cand0 && popc64(cb0 & lt) == tstar);
        const u64 sel1 = ballot(cand1 && popc64(cb0) + popc64(cb1 & lt) == tstar);
        if (sel0 | sel1) {
          const u32 wsel = sel0 ? rdl(w0, ctz64(sel0)) : rdl(w1, ctz64(sel1));
          waste_found = true;
          waste_cell = cell_pad(wsel < 64u ? rdl(E.WS[0], wsel) : rdl(E.WS[1], wsel - 64u));
        }
      }
    }
  }
  wave_sync();
#pragma unroll
  for (int r = 0; r < AR; ++r)
    pm_put(pm, spawnA[r], cell_pad(E.AP[r]), kApple);
  {
    const bool wput = waste_found && lane == 0;
    pm_put(pm, wput, waste_cell, kWaste | (pm[pm_sel(wput, waste_cell)] & kAgentBit));
  }
  wave_sync();
}

// ----------------------------------------------------------------------------------------
// observation crop: the env's n*225 pixels as one byte stream, 4 pixels (12 B) per lane
// ----------------------------------------------------------------------------------------
// `obs` = base of the [E][obs_env_stride] plane this step's observation goes to.  RESTORE (fused rollouts): the map
// bytes under the painted agents are put back afterwards (cell code only, agent bit cleared), so the LDS map stays the
// env's map for the next step of the same launch.
struct NoHook {
  DEVINL void operator()() const {}
};
// `round_hook()`: called once per view round (two agents' views), between a round's gathers and its stores — independent LDS work
// of the caller's (the feature scan's chunks, compute_features) shares the round's waits
template <int KIND, bool RESTORE = false, class HOOK = NoHook>
DEVINL void write_obs(Env<KIND>& E, const GridParams& p, CE_GPTR(uint8_t) obs, bool paint_agents, HOOK&& round_hook = HOOK()) {
  typedef Geo<KIND> G;
  const u32 lane = E.lane;
  uint8_t* pm = E.L->pmap;
  wave_sync();
  u32 under = 0;
  if (RESTORE && paint_agents) {
    under = pm[pm_sel(E.is_agent, E.P)] & kCodeMask;
    wave_sync();
  }
  if (paint_agents) {
    // agents painted in agent order, the later agent wins on a shared cell (map_env.py:257-261)
    if (E.n <= 8) {
      // one (a, b) pair per lane: agent a is covered when a later agent b stands on its cell; every uncovered agent
      // paints itself in one store
      const u32 a = lane >> 3, b = lane & 7u;
      const u32 pa = bperm(E.P, a), pb = bperm(E.P, b);
      const u64 cov = ballot(a < b && b < E.n && pa == pb);
      const bool top = ((u32)(cov >> ((lane & 7u) << 3)) & 0xffu) == 0;  // lane a < 8 reads its own row of pairs
      pm_put(pm, E.is_agent && top, E.P, (6 + lane) << kScale);
      wave_sync();
    } else {
      for (u32 a = 0; a < E.n; ++a) {
        const u32 pa = rdl(E.P, a);
        pm_put(pm, lane == 0, pa, (6 + a) << kScale);
        wave_sync();
      }
    }
  } else {
    if (E.is_agent) pm[E.P] = (uint8_t)(pm[E.P] & kCodeMask);  // reset(): agents are not on the colour map
    wave_sync();
  }
  // crop address of view pixel (i, j): off = o0 + i*A + j*B, kept per agent in its own lane — three registers, so that a
  // view's parameters are three v_readlane and nothing else (packed into one word they cost a readlane + three scalar unpack
  // instructions per view, on the CU's one scalar pipe: DESIGN.md 4.8)
  i32 VO = 0, VA = 0, VB = 0;
  {
    const i32 base = (i32)(E.is_agent ? E.P : pad_of<KIND>(0, 0)) - kView * G::PW - kView;
    if (E.O == 0) { VO = base; VA = G::PW; VB = 1; }                                   // UP
    else if (E.O == 3) { VO = base + 14; VA = -1; VB = G::PW; }                        // LEFT  rot90(k=1)
    else if (E.O == 2) { VO = base + 14 * G::PW + 14; VA = -G::PW; VB = -1; }          // DOWN  rot90(k=2)
    else { VO = base + 14 * G::PW; VA = 1; VB = -G::PW; }                              // RIGHT rot90(k=1,(1,0))
  }
  wave_sync();
  // One lane round per agent: lane l < 60 produces the 4 horizontally adjacent pixels (row l / 4, columns
  // 4 * (l % 4) ..) of that agent's view as 12 bytes (rows are pitched to 16 pixels).  The view parameters of the
  // round are wave-uniform scalars, the lane's (row, column) never changes, and the store is base + lane * 12.
  const u32 row = (lane < 60 ? lane : 59u) >> 2, j0 = (lane & 3u) << 2;
  // column 15 is the row padding: those lanes' last-word byte selector takes zeros instead of the fourth pixel (one
  // per-lane selector instead of a select per view)
  const u32 selz = j0 == 12 ? 0x0c0c0c02u : 0x06050402u;
  const u32 voff = __umul24(lane, 12u);
  const u32* rgb = E.L->rgb;
  const auto dst_env = (CE_GPTR(char))(obs + (size_t)E.e * p.obs_env_stride);
  typedef u32 u32x3 __attribute__((ext_vector_type(3)));
  // the 4 pixels of this lane in agent a's view, packed as 12 bytes
  auto view_unit = [&](u32 a) -> u32x3 {
    const i32 A = (i32)rdl((u32)VA, a), B = (i32)rdl((u32)VB, a);
    const i32 off0 = __mul24((i32)row, A) + __mul24((i32)j0, B) + (i32)rdl((u32)VO, a);
    const i32 off1 = off0 + B, off2 = off1 + B, off3 = off2 + B;  // a chain of adds with the scalar B
    const auto lut = [&](i32 off) { return *(const u32*)((const char*)rgb + pm[off]); };  // a map byte is code * 4
    const u32 c0 = lut(off0);
    const u32 c1 = lut(off1);
    const u32 c2 = lut(off2);
    const u32 c3 = lut(off3);
    u32x3 d;
    d.x = __builtin_amdgcn_perm(c1, c0, 0x04020100u);  // R0 G0 B0 R1
    d.y = __builtin_amdgcn_perm(c2, c1, 0x05040201u);  // G1 B1 R2 G2
    d.z = __builtin_amdgcn_perm(c3, c2, selz);         // B2 R3 G3 B3 (padding lanes: B2 0 0 0)
    return d;
  };
  // Two cache policies for the same bytes, chosen ONCE per pass (the whole loop over the views exists twice: as a branch per
  // view — round 5 — the switch cost ~10 scalar instructions per store, 80 per step, on the pipe the step is shortest of):
  //  * nontemporal: the observation is write-once output and the bulk of the step's bytes; keeping it out of L2 / Infinity
  //    Cache leaves them to the env state that is re-read next step (+2 % at 16 384 envs, +23 % at 65 536 over plain stores)
  //  * write-through (sc1), single-step launches of handles that fit the Infinity Cache (Env::wt — bit 8 of the step kernel's
  //    num_agents argument, decided per launch by the host: ce_api.hip obs_write_through): the bytes leave the XCD's L2 as they are produced instead of at the kernel's end, when
  //    the launch's release writes every dirty line back at once — round 5, interleaved A/B: C4 +3.7 %, C3 +2.6 %, C2 +1.5 %;
  //    a fused rollout LOSES 40 % with it and a 32 768-env batch 30 % (sustained write bandwidth past the cache), hence the switch.
  //    Written as asm: the compiler has no 12-byte sc1 store but the buffer form, and that one (SGPR soffset) it follows
  //    with a write to the data registers without the wait state gfx950 needs — corrupted view rows in 2 % of the envs
  //    (tools/dbg_wt.py); the s_nop inside the string is that wait state.  ("memory": nothing may move across the store.)
  auto views = [&](auto WT) {
    auto put_unit = [&](u32 a, const u32x3& dv) {
      // the agent's view starts a * 720 bytes into the env's block: folded into the wave-uniform base (scalar add), the
      // per-lane part of the address stays lane * 12 for every agent
      if (diag::ablate_obsstore) {  // traffic experiment: the pixels are computed but not written
        asm volatile("" ::"v"(dv.x), "v"(dv.y), "v"(dv.z));
        return;
      }
      const auto dst_a = dst_env + (size_t)__umul24(a, (u32)kObsAgentStride);
      if (lane < 60) {
        if (decltype(WT)::value) asm volatile("global_store_dwordx3 %0, %1, %2 sc1\n\ts_nop 1" ::"v"(voff), "v"(dv), "s"(dst_a) : "memory");
        else __builtin_nontemporal_store(dv, (CE_GPTR(u32x3))(dst_a + voff));
      }
    };
    // two agents per round: their two dependent LDS lookups (map byte, then colour) overlap instead of queueing up;
    // with an odd n the last round repeats agent n - 1 (same bytes to the same place)
    for (u32 a = 0; a < E.n; a += 2) {
      const u32 a1 = min(a + 1u, E.n - 1u);
      const u32x3 d0 = view_unit(a), d1 = view_unit(a1);
      round_hook();
      put_unit(a, d0);
      put_unit(a1, d1);
    }
  };
  if (!RESTORE && E.wt) views(std::true_type{});
  else views(std::false_type{});
  if (RESTORE && paint_agents) {
    wave_sync();
    pm_put(pm, E.is_agent, E.P, under);
    wave_sync();
  }
}

// ----------------------------------------------------------------------------------------
// metrics helpers
// ----------------------------------------------------------------------------------------
DEVINL double shfl_f64(double v, u32 src_lane_uniform) {
  u64 b = (u64)__double_as_longlong(v);
  u32 lo = rdl((u32)b, src_lane_uniform), hi = rdl((u32)(b >> 32), src_lane_uniform);
  return __longlong_as_double((long long)(((u64)hi << 32) | lo));
}
DEVINL i32 shfl_i32(i32 v, u32 src_lane_uniform) { return (i32)rdl((u32)v, src_lane_uniform); }
DEVINL long long shfl_i64(long long v, u32 l) {
  u64 b = (u64)v;
  return (long long)(((u64)rdl((u32)(b >> 32), l) << 32) | rdl((u32)b, l));
}

// numpy pairwise sum of the n (< 128) per-agent values held in lanes 0..n-1 (np.mean of a list)
DEVINL double np_sum_lanes(double v, u32 n) {
  if (n < 8) {
    double res = 0.;
    for (u32 i = 0; i < n; ++i) res += shfl_f64(v, i);
    return res;
  }
  double r[8];
  for (u32 j = 0; j < 8; ++j) r[j] = shfl_f64(v, j);
  double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
  for (u32 i = 8; i < n; ++i) res += shfl_f64(v, i);  // n <= 9: tail only
  return res;
}

template <int KIND> DEVINL void zero_metrics(Env<KIND>& E, const GridParams& p) {
  const u32 nmi = CE_MI_COUNT(E.n), nmf = CE_MF_COUNT(E.n);
  for (u32 k = E.lane; k < nmi; k += 64) p.int_metrics[(size_t)E.e * nmi + k] = 0;
  for (u32 k = E.lane; k < nmf; k += 64) p.f64_metrics[(size_t)E.e * nmf + k] = 0.0;
}

// ----------------------------------------------------------------------------------------
// setup_agents (cleanup_new.py:302-320, map_env.py:816-832): n x (shuffle the persistent
// spawn list, take the LAST free entry, randint(4) orientation)
// ----------------------------------------------------------------------------------------
template <int KIND> DEVINL bool setup_agents(Env<KIND>& E, u32 list_len) {
  const GridTables& T = *E.T;
  const u32 lane = E.lane;
  bool ok = true;
  E.P = 0xffffu;
  E.O = 0;
  E.RW = 0;
  for (u32 i = 0; i < E.n; ++i) {
    shuffle_lanes1(E.rng, E.SP, list_len, lane);
    const u32 cell = cell_pad(T.spawn[lane < list_len ? E.SP : 0]);
    bool taken = false;
    for (u32 b = 0; b < i; ++b) taken = taken || (rdl(E.P, b) == cell);
    const u64 fb = ballot(lane < list_len && !taken);
    const u32 r = rng_next(E.rng, lane) & 3u;  // randint(4)
    if (fb == 0) {
      ok = false;
      continue;
    }
    const u32 pc = rdl(cell, fls64(fb));
    const u32 orient = r == 0 ? 3u : r == 1 ? 1u : r == 2 ? 0u : 2u;  // keys order LEFT,RIGHT,UP,DOWN
    E.P = wrl(pc, i, E.P);
    E.O = wrl(orient, i, E.O);
  }
  return ok;
}

template <int KIND> DEVINL void sample_theta(Env<KIND>& E, const GridParams& p, double& theta) {
  // SeparateContractSubgameStage.reset two_stage_train.py:163-166
  if (p.flags & CE_FLAG_EXTERNAL_THETA) {  // the caller owns the theta buffer: a reset neither draws nor changes it
    theta = p.theta[E.e];
    return;
  }
  if (p.contract == CE_CONTRACT_NONE) {
    theta = 0.0;
    return;
  }
  const double u0 = rng_double(E.rng, E.lane);
  if (u0 > p.null_prob) {
    const double u1 = rng_double(E.rng, E.lane);
    theta = p.contract_low + (p.contract_high - p.contract_low) * u1;
  } else {
    theta = p.contract_low;
  }
}

// MapEnv.reset + CleanupEnv/HarvestEnv.reset + wrapper reset (state ends up in E / LDS)
template <int KIND> DEVINL void reset_env(Env<KIND>& E, const GridParams& p, double& theta, u32& t, u32& fault) {
  typedef Geo<KIND> G;
  if (!setup_agents(E, KIND == CE_KIND_CLEANUP ? 2u * E.nspawn : E.nspawn)) fault |= CE_FAULT_NO_SPAWN;  // cleanup's doubled list (cleanup_new.py:114-115)
  if (!E.is_agent) E.P = 0xffffu;
  wave_sync();
  {  // reset_map + custom_reset: the static padded base map
    const u32* src = (const u32*)E.T->base_pmap;
    u32* dst = (u32*)E.L->pmap;
    for (u32 k = E.lane; k < (u32)G::PCELLS / 4; k += 64) dst[k] = src[k] << kScale;
  }
  mark_agents(E);
  zero_metrics(E, p);
  if (p.flags & CE_FLAG_BEAM_TRACE) clear_beam_map(E, p);  // self.beam_pos = [] (map_env.py:316)
  custom_map_update(E);
  t = 0;
  sample_theta(E, p, theta);
}

// ----------------------------------------------------------------------------------------
// infos[k]['feature_obs'] (cleanup_new.py:243-251 / harvest_new.py:215-222), also the feature-mode
// reset observation (cleanup_new.py:193-202).  Returns this lane's feature 8 (harvest
// total_close_apples).  Apples / wastes only ever sit on the static apple / waste cells, so the
// closest-cell searches scan those lists (2-3 lane rounds) instead of the whole map.
// ----------------------------------------------------------------------------------------
DEVINL u32 min3u(u32 a, u32 b, u32 c) { return min(a, min(b, c)); }  // v_min3_u32
// min over the aligned group of 2^sh lanes (sh = 2, 3, 4) this lane belongs to; every lane of the group gets it
DEVINL u32 group_min_u32(u32 v, u32 sh) {
#define CE_DPP_MIN(ctrl)                                                                    \
  {                                                                                         \
    const u32 t = (u32)__builtin_amdgcn_update_dpp((int)v, (int)v, ctrl, 0xf, 0xf, false);  \
    v = t < v ? t : v;                                                                      \
  }
  CE_DPP_MIN(0xB1)               // quad_perm [1,0,3,2]
  CE_DPP_MIN(0x4E)               // quad_perm [2,3,0,1]
  if (sh >= 3) CE_DPP_MIN(0x141)  // row_half_mirror
  if (sh >= 4) CE_DPP_MIN(0x140)  // row_mirror
#undef CE_DPP_MIN
  return v;
}

// two feature values as one dword when the row is dword aligned (num_features even), else two shorts
template <class P> DEVINL void store_feat2(P f, u32 idx, u32 lo, u32 hi, bool aligned) {
  if (aligned) {
    *(CE_GPTR(u32))(f + idx) = (lo & 0xffffu) | hi << 16;
  } else {
    f[idx] = (int16_t)lo;
    f[idx + 1] = (int16_t)hi;
  }
}

// presA / presW (out): the presence ballots of the map as it is now (apple on apple cell lane + 64 r, waste likewise) — the
// packed map state of a single-step launch is assembled from them (store_grid_bits) instead of scanning the map again
// `mid(scan_step)`: called once the key lists are published, with the scan's loop body as a callable (one 4-cell chunk per list
// and call, a no-op once the lists are through); chunks mid() did not take are scanned afterwards.  The hook exists for the
// round-6 experiment that ran the observation pass there (-DCE_FEATSCAN_INTERLEAVE, grid_step_core): the scan's LDS round
// trips — two 16-byte reads and a wait per chunk, a chain nothing else fills: ablated, the scan is worth + 8 % on the resident
// kernels for 3.5 % of their instructions (profiles/r06_phase_ablations.txt) — would travel with the views' own gathers.  It
// costs registers the step kernel does not have (see there); the default passes no mid.
struct NoMid {
  template <class F> DEVINL void operator()(F&&) const {}
};
template <int KIND, class MID = NoMid> DEVINL u32 compute_features(Env<KIND>& E, const GridParams& p, CE_GPTR(int16_t) features, u32 cleaned,
                                                                   u64 (&presA)[3], u64 (&presW)[2], MID&& mid = MID()) {
  typedef Geo<KIND> G;
  const GridTables& T = *E.T;
  const u32 lane = E.lane, n = E.n;
  uint8_t* pm = E.L->pmap;
  const u32 nf = p.num_features;
  const auto feat_env = features + (size_t)E.e * n * nf;  // wave-uniform base, 32-bit lane offsets below
  const bool al = (nf & 1u) == 0;  // feature rows dword aligned
  const u32 cp = n > 1 ? 1u : 0u;  // compute_closest_pos bug: a0 -> a1, everyone else -> a0
  const u32 p_a0 = rdl(E.P, 0), o_a0 = rdl(E.O, 0), p_cp = rdl(E.P, cp), o_cp = rdl(E.O, cp);
  const u32 myrow = row_of<KIND>(E.is_agent ? E.P : pad_of<KIND>(0, 0));
  const u32 mycol = col_of<KIND>(E.is_agent ? E.P : pad_of<KIND>(0, 0));
  const u32 cpp = lane == 0 ? p_cp : p_a0, cpo = lane == 0 ? o_cp : o_a0;

  // Closest apple / waste = min over keys  manhattan << 16 | row << 8 | col  (ties: smallest (row, col) ==
  // first in the row-major list, as np.argmin).  Coordinates sit one per byte, so the Manhattan distance is a
  // single v_sad_u8 against the agent's packed (row, col).  The present apples / wastes are first written as a
  // key list (absent cell = kNoKey, whose key stays above every real one) into the now idle random-word scratch; then the wave
  // splits into n groups of 2^sh lanes, group a scanning the whole list for agent a, 4 cells per LDS read.
  constexpr u32 NCHUNK = KIND == CE_KIND_CLEANUP ? 32u : 48u;  // 4-cell chunks per list (a multiple of the largest group, 16 lanes)
  constexpr u32 kNoKey = 0x7f000000u;  // list entry of an absent cell: its key (sad << 16) + entry stays above every real key, no wrap
  constexpr u32 NENT = NCHUNK * 4u;
  static_assert(NENT >= (u32)G::NAPPLE && (KIND != CE_KIND_CLEANUP || NENT >= (u32)G::NWASTE), "key list too short");  // (a caller's layout never has more cells than the shipped one)
  static_assert(KIND == CE_KIND_CLEANUP ? 2 * NENT * 4 <= sizeof(E.L->U) + sizeof(E.L->S) : NENT * 4 <= sizeof(E.L->U), "scratch");
  u32* keyA = E.L->U;
  u32* keyW = E.L->U + NENT;
  u32 napples = 0, nwaste = 0;
#pragma unroll
  for (u32 r = 0; r < NENT / 64u; ++r) {
    bool f = false;
    u32 rc = 0;
    if (r < 3) {
      f = both(lane + 64 * r < E.napple, pm[cell_pad(E.AP[r < 3 ? r : 0])] == kApple);
      rc = cell_rc(E.AP[r < 3 ? r : 0]);
    }
    const u64 fb = ballot(f);
    if (r < 3) presA[r < 3 ? r : 0] = fb;
#ifdef CE_FEATSCAN_FULL_LISTS
    if (!diag::ablate_featscan) keyA[lane + 64 * r] = f ? rc : kNoKey;
#else
    // COMPACTED key lists (round 6): only the present cells, in cell order (rank = present cells before this one) — the scan
    // below then runs ceil(present / cells-per-round) rounds instead of one per 32 slots (a steady-state cleanup map holds ~50
    // of 119 waste cells and a few dozen of 103 apples).  Every slot is first filled with kNoKey by its own lane, then the present
    // cells overwrite their rank's slot: two DS stores of one wave execute in order, so the lists need no padding logic and the
    // scan no per-list bounds
    if (!diag::ablate_featscan) {
      keyA[lane + 64 * r] = kNoKey;
      if (f) keyA[napples + rank_in(fb, lane)] = rc;
    }
#endif
    napples += popc64(fb);
  }
  if (KIND == CE_KIND_CLEANUP) {
#pragma unroll
    for (u32 r = 0; r < 2; ++r) {
      const bool f = both(lane + 64 * r < E.nwaste, (pm[cell_pad(E.WS[r])] & kCodeMask) == kWaste);
      presW[r] = ballot(f);
#ifdef CE_FEATSCAN_FULL_LISTS
      if (!diag::ablate_featscan) keyW[lane + 64 * r] = f ? cell_rc(E.WS[r]) : kNoKey;
#else
      if (!diag::ablate_featscan) {
        keyW[lane + 64 * r] = kNoKey;
        if (f) keyW[nwaste + rank_in(presW[r], lane)] = cell_rc(E.WS[r]);
      }
#endif
      nwaste += popc64(presW[r]);
    }
  }
  const u32 sh = n <= 4 ? 4u : n <= 8 ? 3u : 2u;
  const u32 ga = lane >> sh, gl = lane & ((1u << sh) - 1u);

  const u32 prc = bperm(mycol | myrow << 8, ga);  // also the wave_sync-free way to get agent ga's position
  // harvest: apples in each agent's 21-cell neighbourhood, read here — before mid() may paint the agents over the map
  u32 close_now = 0;
  if (KIND == CE_KIND_HARVEST) {
    if (n <= 8) {
      // all agents at once: lane = agent * 8 + j looks at offsets j, 8 + j, 16 + j of the 21-cell neighbourhood — three
      // independent map reads in flight instead of n dependent read / ballot rounds
      const u32 ag = lane >> 3, j = lane & 7u;
      const i32 pa = (i32)bperm(E.P, ag < n ? ag : 0u);
      const i32 o0 = (i32)T.close_off[j], o1 = (i32)T.close_off[8 + j], o2 = (i32)T.close_off[min(16u + j, 20u)];
      const bool live = ag < n;
      const u64 b0 = ballot(both(live, pm[live ? pa + o0 : 0] == kApple));
      const u64 b1 = ballot(both(live, pm[live ? pa + o1 : 0] == kApple));
      const u64 b2 = ballot(both(live && j < 5, pm[live ? pa + o2 : 0] == kApple));
      const u32 sh8 = (lane & 7u) << 3;  // lane a < 8: its own byte of the three pair masks
      close_now = __builtin_popcount((u32)(b0 >> sh8) & 0xffu) + __builtin_popcount((u32)(b1 >> sh8) & 0xffu) +
                  __builtin_popcount((u32)(b2 >> sh8) & 0xffu);
      if (!E.is_agent) close_now = 0;
    } else {
      for (u32 a = 0; a < n; ++a) {
        const u32 pa = rdl(E.P, a);
        const bool v = both(lane < 21, pm[(i32)pa + (i32)T.close_off[lane < 21 ? lane : 0]] == kApple);
        const u32 cnt = popc64(ballot(v));
        if (lane == a) close_now = cnt;
      }
    }
  }
  wave_sync();
  u32 ka = 0xffffffffu, kw = 0xffffffffu;
  u32 kc = 0;
#ifdef CE_FEATSCAN_FULL_LISTS
  const u32 chunks = diag::ablate_featscan ? 0u : (NCHUNK >> sh);  // wave-uniform trip count: every lane scans NCHUNK >> sh chunks
#else
  // rounds of the longer list (cells per round = 4 per lane x 2^sh lanes per agent group); the shorter one reads kNoKey slots
  const u32 longer = KIND == CE_KIND_CLEANUP ? max(napples, nwaste) : napples;
  const u32 chunks = diag::ablate_featscan ? 0u : (longer + (4u << sh) - 1u) >> (sh + 2u);
#endif
  auto scan_round = [&]() {
    const u32 c = gl + (kc << sh);
    // key = manhattan << 16 | row << 8 | col in one v_sad_hi_u8 per cell (absent entries land at >= kNoKey)
    const uint4 a4 = *reinterpret_cast<const uint4*>(keyA + 4 * c);
    const u32 k0 = __builtin_amdgcn_sad_hi_u8(a4.x, prc, a4.x), k1 = __builtin_amdgcn_sad_hi_u8(a4.y, prc, a4.y);
    const u32 k2 = __builtin_amdgcn_sad_hi_u8(a4.z, prc, a4.z), k3 = __builtin_amdgcn_sad_hi_u8(a4.w, prc, a4.w);
    ka = min3u(min3u(ka, k0, k1), k2, k3);
    if (KIND == CE_KIND_CLEANUP) {
      const uint4 w4 = *reinterpret_cast<const uint4*>(keyW + 4 * c);
      const u32 q0 = __builtin_amdgcn_sad_hi_u8(w4.x, prc, w4.x), q1 = __builtin_amdgcn_sad_hi_u8(w4.y, prc, w4.y);
      const u32 q2 = __builtin_amdgcn_sad_hi_u8(w4.z, prc, w4.z), q3 = __builtin_amdgcn_sad_hi_u8(w4.w, prc, w4.w);
      kw = min3u(min3u(kw, q0, q1), q2, q3);
    }
    ++kc;
  };
  auto scan_step = [&]() {  // (what mid() gets: a no-op once the lists are through)
    if (kc < chunks) scan_round();
  };
  mid(scan_step);
#pragma unroll 1  // unrolling keeps 8 b128 loads in flight and costs an occupancy step
  while (kc < chunks) scan_round();
  ka = group_min_u32(ka, sh);
  if (KIND == CE_KIND_CLEANUP) kw = group_min_u32(kw, sh);
  if (ka >= kNoKey) ka = 0;  // [0, 0] sentinel when there is none
  if (kw >= kNoKey) kw = 0;
  if (!al && gl == 0 && ga < n) {  // odd row pitch: the group's first lane writes agent ga's closest-apple / -waste features
    auto f = feat_env + __umul24(ga, nf);
    store_feat2(f, 6, (ka >> 8) & 0xffu, ka & 0xffu, al);
    if (KIND == CE_KIND_CLEANUP) store_feat2(f, 8, (kw >> 8) & 0xffu, kw & 0xffu, al);
  }
  // dword-aligned rows (every harvest row, cleanup with an even n): agent lane a fetches its group's minima and
  // writes its whole row as a few wide stores in ONE predicated block (it was ~10 dword stores in 3 blocks)
  const u32 lead = E.is_agent ? lane << sh : 0u;
  const u32 ka_a = bperm(ka, lead), kw_a = KIND == CE_KIND_CLEANUP ? bperm(kw, lead) : 0u;
  auto f = feat_env + __umul24(E.is_agent ? lane : 0u, nf);
  if (al) {
    typedef u32 u32x2 __attribute__((ext_vector_type(2)));
    typedef u32 u32x4 __attribute__((ext_vector_type(4)));
    const auto f32 = (CE_GPTR(u32))f;
    const u32x4 head = {myrow | mycol << 16, E.O | row_of<KIND>(cpp) << 16, col_of<KIND>(cpp) | cpo << 16,
                        ((ka_a >> 8) & 0xffu) | (ka_a & 0xffu) << 16};
    if (KIND == CE_KIND_CLEANUP) {
      const u32x2 mid = {((kw_a >> 8) & 0xffu) | (kw_a & 0xffu) << 16, napples | nwaste << 16};
      u32 cw[4] = {0, 0, 0, 0};  // the cleaned vector, two agents per dword (wave-uniform); all zero on most steps
      if (n == 8 && ballot(cleaned != 0) != 0) {
#pragma unroll
        for (u32 b = 0; b < 8; b += 2) cw[b >> 1] = rdl(cleaned, b) | rdl(cleaned, b + 1) << 16;
      }
      if (E.is_agent) {
        *(CE_GPTR(u32x4))(f32) = head;
        *(CE_GPTR(u32x2))(f32 + 4) = mid;
        if (n == 8) {
          const u32x4 tail = {cw[0], cw[1], cw[2], cw[3]};
          *(CE_GPTR(u32x4))(f32 + 6) = tail;
        }
      }
      if (n != 8)
        for (u32 b = 0; b < n; b += 2) {
          const u32 w = rdl(cleaned, b) | rdl(cleaned, b + 1) << 16;
          if (E.is_agent) f32[6 + (b >> 1)] = w;
        }
    } else {
      if (E.is_agent) {
        *(CE_GPTR(u32x4))(f32) = head;
        f32[4] = close_now | napples << 16;
        if (n == 8) {
          const u32x4 z = {0u, 0u, 0u, 0u};
          *(CE_GPTR(u32x4))(f32 + 5) = z;
          *(CE_GPTR(u32x4))(f32 + 9) = z;
        } else {
          for (u32 b = 0; b < n; ++b) f32[5 + b] = 0u;
        }
      }
    }
    return close_now;
  }
  if (KIND == CE_KIND_CLEANUP) {
    for (u32 b = 0; b < n; ++b) {
      const u32 cb = rdl(cleaned, b);
      if (E.is_agent) f[12 + b] = (int16_t)cb;
    }
  }
  if (E.is_agent) {
    store_feat2(f, 0, myrow, mycol, al);
    store_feat2(f, 2, E.O, row_of<KIND>(cpp), al);
    store_feat2(f, 4, col_of<KIND>(cpp), cpo, al);
    if (KIND == CE_KIND_CLEANUP) {
      store_feat2(f, 10, napples, nwaste, al);
    } else {
      store_feat2(f, 8, close_now, napples, al);
      for (u32 b = 0; b < n; ++b) *(CE_GPTR(u32))(f + 10 + 2 * b) = 0u;  // num_features = 10 + 2n: always aligned
    }
  }
  return close_now;
}

// ----------------------------------------------------------------------------------------
// kernels
// ----------------------------------------------------------------------------------------
#ifndef CE_HARVEST_WAVES
#define CE_HARVEST_WAVES 8
#endif
// Launch bounds are chosen to keep the kernels out of scratch: a kernel with spilled VGPRs ran at two speeds on this
// pool, process by process (the feature-env rollout with 17 spills: 1.57 G or 1.26 G agent-steps/s from the same binary
// on the same box), and the step kernel at 8 waves with two spilled dwords was no faster than at 7 waves with none
// (it is back at 8 waves without spills since the spawn skip shares the stream bookkeeping of the full path).
#ifndef CE_CLEANUP_WAVES
#define CE_CLEANUP_WAVES 8  // the step kernel: 62 VGPRs, no scratch
#endif
#ifndef CE_CLEANUP_ROLLOUT_WAVES
// 8 waves (64 VGPRs + 21 spilled): measured 3 % faster than 7 waves (72 + 13 spilled) in round 3 — 4.24 vs 4.12 G
// agent-steps/s, tools/ab.sh, two rounds interleaved — which in round 1 was itself 3 % faster than 5 waves without spills
#define CE_CLEANUP_ROLLOUT_WAVES 8
#endif
// launches that do not fill the machine (launch_grid_rollout): every wave is resident anyway, so the register budget is cut
// for the occupancy the launch can reach and no further — C2 (three launches of 1 365 envs): 7 waves 1.59 G, 6: 1.69, 5: 1.75,
// 4: 1.76; launches of 2 730 envs: 6 waves 2.81 / 3.94 G (n = 4 / 8), 5: 2.83 / 3.79, 4: 2.6 / 3.45
#ifndef CE_CLEANUP_ROLLOUT_WAVES_MID
#define CE_CLEANUP_ROLLOUT_WAVES_MID 6
#endif
#ifndef CE_CLEANUP_ROLLOUT_WAVES_SMALL
#define CE_CLEANUP_ROLLOUT_WAVES_SMALL 4
#endif
constexpr int kWavesPerBlock = 1;

template <int KIND> DEVINL bool env_begin(Env<KIND>& E, const GridParams& p, WaveLds<KIND>* lds, u32 env_first, u32 env_end) {
  const u32 wave = threadIdx.x >> 6;
  E.lane = lane_id();
  E.waste_perm_dirty = false;
  E.wt = false;  // (the single-step kernels set it from their launch: k_grid_step, k_grid_reset)
  E.e = rfl(env_first + blockIdx.x * kWavesPerBlock + wave);
  E.n = p.n;
  E.is_agent = E.lane < E.n;
  E.L = lds + wave;
#ifdef CE_INSTRUMENTED
  E.dbg = p.debug ? (unsigned long long*)p.debug + (size_t)E.e * 16 : nullptr;
#else
  E.dbg = nullptr;
#endif
  return E.e < env_end;
}

// --- seed + "construct": replay the RNG use of MapEnv.__init__ (map_env.py:122-131) ---
template <int KIND, bool CM> __global__ __launch_bounds__(64 * kWavesPerBlock) void k_grid_construct(const GridParams* __restrict__ pp, const uint8_t* __restrict__ call_actions, const uint8_t* __restrict__ call_mask,
                 u32 env_first, u32 env_end) {
  const GridParams& p = *pp;
  __shared__ WaveLds<KIND> lds[kWavesPerBlock];
  Env<KIND> E;
  if (!env_begin(E, p, lds, env_first, env_end)) return;
  env_geometry<KIND, CM>(E, p);
  if (call_mask && call_mask[E.e] == 0) return;
  load_static(E);
  load_rng(E, p);
  E.SP = E.lane < 20 ? E.lane : 0;
  E.WP0 = E.lane;
  E.WP1 = E.lane + 64;
  u32 fault = 0;
  if (!setup_agents(E, E.nspawn)) fault |= CE_FAULT_NO_SPAWN;
  zero_pmap(E);  // world_map is blank until the first reset
  store_grid(E, p, true);
  store_agents(E, p);
  store_perms(E, p, true);
  store_rng(E, p);
  zero_metrics(E, p);
  if (E.lane == 0) {
    p.timestep[E.e] = 0;
    p.theta[E.e] = 0.0;
    p.done[E.e] = 0;
    p.error_flags[E.e] = fault;
  }
}

template <int KIND> DEVINL void clear_step_outputs(Env<KIND>& E, const GridParams& p) {
  if (E.is_agent) {
    const size_t ea = (size_t)E.e * E.n;
    (p.base_reward + ea)[E.lane] = 0;
    (p.reward + ea)[E.lane] = 0.0;
    (p.info + 2 * ea)[2 * E.lane] = 0;
    (p.info + 2 * ea)[2 * E.lane + 1] = 0;
  }
}

template <int KIND, bool CM> __global__ __launch_bounds__(64 * kWavesPerBlock) void k_grid_reset(const GridParams* __restrict__ pp, const uint8_t* __restrict__ call_actions, const uint8_t* __restrict__ call_mask,
                 u32 env_first, u32 env_end) {
  const GridParams& p = *pp;
  __shared__ WaveLds<KIND> lds[kWavesPerBlock];
  Env<KIND> E;
  if (!env_begin(E, p, lds, env_first, env_end)) return;
  env_geometry<KIND, CM>(E, p);
  if (call_mask && call_mask[E.e] == 0) return;
  load_static(E);
  load_rng(E, p);
  load_perms(E, p);
  double theta = 0.0;
  u32 t = 0, fault = 0;
  reset_env(E, p, theta, t, fault);
  if (!diag::ablate_gridstore) store_grid(E, p);
  store_agents(E, p);
  store_perms(E, p, true);
  store_rng(E, p);
  clear_step_outputs(E, p);
  if (!E.is_agent) E.P = 0xffffu;
  {
    u64 pa_[3], pw_[2];
    compute_features(E, p, p.features, 0u, pa_, pw_);
  }
  write_obs(E, p, p.obs, false);  // reset() does not paint the agents on the colour map
  if (E.lane == 0) {
    p.timestep[E.e] = 0;
    p.theta[E.e] = theta;
    p.done[E.e] = 0;
    p.error_flags[E.e] = fault;  // a reset starts a clean episode (faults are sticky until then)
  }
}

// ----------------------------------------------------------------------------------------
// Where a step's per-step outputs go.  StepOutDirect: the handle's own [E] buffers (one launch per step).
// StepOutPlane: plane `pl` of the trajectory arrays of a fused rollout (ce_rollout_fused; an array the caller did not
// supply is the handle's buffer with plane stride 0, resolved on the host).
// ----------------------------------------------------------------------------------------
struct StepOutDirect {
  const GridParams& p;
  DEVINL CE_GPTR(uint8_t) obs() const { return p.obs; }
  DEVINL CE_GPTR(int32_t) base_reward() const { return p.base_reward; }
  DEVINL CE_GPTR(double) reward() const { return p.reward; }
  DEVINL CE_GPTR(uint8_t) done() const { return p.done; }
  DEVINL CE_GPTR(uint8_t) info() const { return p.info; }
  DEVINL CE_GPTR(int16_t) features() const { return p.features; }
};
struct StepOutPlane {
  const RolloutArgs& ra;
  u32 pl;
  DEVINL CE_GPTR(uint8_t) obs() const { return ra.obs + (size_t)pl * ra.obs_plane; }
  DEVINL CE_GPTR(int32_t) base_reward() const { return ra.base_reward + (size_t)pl * ra.agent_plane; }
  DEVINL CE_GPTR(double) reward() const { return ra.reward + (size_t)pl * ra.reward_plane; }
  DEVINL CE_GPTR(uint8_t) done() const { return ra.done + (size_t)pl * ra.done_plane; }
  DEVINL CE_GPTR(uint8_t) info() const { return ra.info + (size_t)pl * ra.info_plane; }
  DEVINL CE_GPTR(int16_t) features() const { return ra.features + (size_t)pl * ra.features_plane; }
};

// One env-step on the state held in E / LDS: MapEnv.step, infos, contract transfer, metrics, observation, in-launch
// auto-reset.  FUSED = false: the step is its own launch and the state is written back to HBM at the end.
// FUSED = true (k_grid_rollout): the state stays resident for the next step; only the per-step outputs leave.
template <int KIND, bool FUSED, class OUT>
DEVINL void grid_step_core(Env<KIND>& E, const GridParams& p, const OUT& out, u32 ACT, u32& t, double& theta, u32& fault,
                           bool& did_reset) {
  const GridTables& T = *E.T;
  const u32 lane = E.lane, n = E.n;
  const size_t ea = (size_t)E.e * n;  // wave-uniform: per-agent arrays are indexed base + 32-bit lane offset
  const u32 max_action = KIND == CE_KIND_CLEANUP ? 8u : 7u;
  if (ballot(E.is_agent && ACT > max_action) != 0) {  // KeyError in the reference (Agent.py:174,213): the step is not taken
    if (FUSED) fault |= CE_FAULT_BAD_ACTION;
    else if (lane == 0) p.error_flags[E.e] |= CE_FAULT_BAD_ACTION;
    return;
  }
  uint8_t* pm = E.L->pmap;
  if (FUSED) rng_begin_op(E.rng, lane);  // counter mode: steps after a launch's first one

  // ---------------- MapEnv.step ----------------
  t += 1;
  if (p.flags & CE_FLAG_BEAM_TRACE) clear_beam_map(E, p);  // self.beam_pos = [] (map_env.py:231)
  CE_PROBE_POINT(t, &p, lane, n);
  CE_STAMP(1);
  // -DCE_ABLATE_HALF_NARROW (timing only, wrong results): every other env skips the phases that keep <= 16 lanes busy — moves,
  // consume / id shuffle / beams, reward and metric bookkeeping — i.e. what two envs packed into one wave could at best share
  const bool narrow_skip = diag::ablate_half_narrow && (E.e & 1u) != 0;
  if (!diag::ablate_moves && !narrow_skip) update_moves(E, ACT);
  CE_STAMP(2);
  if (!E.is_agent) E.P = 0xffffu;

  // eaten_apples: final position held an apple when the step was entered (nothing has touched
  // the map yet, so the pre-consume map IS the reference's current_apple_points)
  const bool onA = both(E.is_agent, pm[E.is_agent ? E.P : 0] == kApple);
  u32 eaten = onA ? 1u : 0u, eaten_close = 0, cleaned = 0;
  if (KIND == CE_KIND_HARVEST) {
    for (u64 om = ballot(onA); om; om &= om - 1) {  // count_apples_in_radius(5, pos) < 4
      const u32 a = ctz64(om);
      const u32 pa = rdl(E.P, a);
      const bool v = both(lane < 21, pm[(i32)pa + (i32)T.close_off[lane < 21 ? lane : 0]] == kApple);
      const u32 cnt = popc64(ballot(v));
      if (lane == a && cnt < 4) eaten_close = 1;
    }
  }
  if (!narrow_skip) {  // consume in agent order: the first agent on the cell gets the apple
    bool first = true;
    const u64 onm = ballot(onA);
    if (onm & (onm - 1)) {  // two or more agents on apples: only then can a cell be shared
      for (u64 it = onm; it; it &= it - 1) {
        const u32 b = ctz64(it);
        const u32 pb = rdl(E.P, b);
        if (lane > b && E.P == pb) first = false;
      }
    }
    if (onA && first) E.RW += 1;
    wave_sync();
    pm_put(pm, onA, E.P, kEmpty);
    mark_agents(E);
  }
  if (!narrow_skip) {  // update_custom_moves: always shuffles the n ids, then fires in that order
    u32 IDS = lane;
    const u64 firing = ballot(E.is_agent && ACT >= 7);
    if (firing & (firing - 1)) {  // the shuffled order only matters between two or more beams
      shuffle_lanes1(E.rng, IDS, n, lane);
    } else {
      consume_small(E.rng, n, lane);  // same stream words, no swaps
    }
    if (firing != 0) {
      // positions of the shuffled list that hold a firing agent, visited in list order
      const u32 act_at = bperm(ACT, IDS);  // lane k: action of the agent at list position k
      for (u64 it = ballot(lane < n && act_at >= 7); it; it &= it - 1) {
        const u32 k = ctz64(it);
        const u32 a = rdl(IDS, k);
        const u32 act = rdl(act_at, k);
        if (KIND == CE_KIND_CLEANUP && act == 7) {
          const u32 c = fire_beam(E, p, a, true);
          if (lane == a) cleaned = c;
        } else {
          if (lane == a) E.RW -= 1;  // fire_beam(b"F")
          fire_beam(E, p, a, false);
        }
      }
    }
  }
  CE_STAMP(3);
  custom_map_update<KIND>(E);
  CE_TRUNCATE_SPAWN_RETURN();
  CE_STAMP(4);

  // ---------------- rewards ----------------
  i32 base_rew = E.is_agent ? E.RW : 0;
  E.RW = 0;
  if (p.flags & CE_FLAG_COLLECTIVE_REWARD) {
    i32 s = 0;
    for (u32 b = 0; b < n; ++b) s += shfl_i32(base_rew, b);
    base_rew = s;
  }
  double rew = (double)base_rew;
  if (p.flags & CE_FLAG_INEQUITY_AVERSE) {
    long long pos = 0, neg = 0;
    for (u32 b = 0; b < n; ++b) {
      const i32 d = shfl_i32(base_rew, b) - base_rew;
      if (d > 0) pos += d;
      if (d < 0) neg += d;
    }
    const double dis = p.alpha * (double)pos, adv = p.beta * (double)neg;
    rew = (double)base_rew - (dis + adv) / (double)(n - 1);
  }

  CE_STAMP(5);
  // ---------------- feature obs, infos, metrics ----------------
  u64 presA[3] = {0, 0, 0}, presW[2] = {0, 0};
  // The observation (the bulk of the step's stores) goes out as early as the map allows, so that its stores drain
  // under the epilogue's arithmetic instead of at the wave's very end.  It paints the agents over the map bytes,
  // hence after the feature pass has read the map and after the map state is packed; a done step with auto-reset writes
  // the reset observation instead and keeps the late path.  -DCE_FEATSCAN_INTERLEAVE runs it INSIDE the feature pass
  // (compute_features' `mid`: the closest-apple / -waste scan only reads its key lists, so its chunks could ride along with
  // the view rounds) — measured in round 6 and NOT the default: the merged live ranges spill 16 VGPRs in the headline
  // instance (C4 per-step 4.40 -> 3.0 G, counter 4.97 -> 4.27, fused - 2.5 %: profiles/r06_ab_scan_interleave.txt).
  const bool obs_early = t != p.horizon;
  auto obs_pass = [&](auto&& scan_step) {
    if (!obs_early) return;
    if (!FUSED) store_grid_bits(E, p, presA, presW);  // (a fused rollout keeps the map in LDS and packs it once, after its last step)
    if (!diag::ablate_obs) {
#ifdef CE_FEATSCAN_INTERLEAVE
      write_obs<KIND, FUSED>(E, p, out.obs(), true, scan_step);
#else
      write_obs<KIND, FUSED>(E, p, out.obs(), true);
#endif
    }
  };
  u32 feat8 = 0;
  if (diag::ablate_features) {
    if (obs_early) {
      if (!FUSED) store_grid(E, p);
      if (!diag::ablate_obs) write_obs<KIND, FUSED>(E, p, out.obs(), true);
    }
  } else {
#ifdef CE_FEATSCAN_INTERLEAVE
    feat8 = compute_features(E, p, out.features(), cleaned, presA, presW, obs_pass);
#else
    feat8 = compute_features(E, p, out.features(), cleaned, presA, presW);
    obs_pass([] {});
#endif
  }
  const double rew_env = rew;  // the env's own reward (after collective / inequity aversion), before the contract
  // A quiet step — nobody ate, cleaned, fired or was hit: every reward, transfer and metric increment is zero — skips the
  // contract arithmetic and the whole metric bookkeeping behind ONE wave-uniform test (two thirds of the steps of the
  // benchmark; it was half a dozen separate tests, each a ballot, a scalar compare and a branch).
  const bool done = t == p.horizon;
  const bool busy = (ballot(E.is_agent && (eaten | eaten_close | cleaned | (u32)base_rew) != 0) != 0 || done) && !narrow_skip;
  // ---------------- contract transfer (two_stage_train.py:69-92) ----------------
  double transfers_total = 0.0;
  if (busy && p.contract != CE_CONTRACT_NONE) {
    double tr;
    if (p.contract == CE_CONTRACT_CLEANUP) tr = -theta * (double)cleaned;  // contract_list.py:26
    else tr = (feat8 < 4 && eaten_close > 0) ? theta : 0.0;                // contract_list.py:50-53
    double total = 0.0;
    const double share = tr / (double)(n - 1);  // t_i / (len(acts) - 1), one division per agent
    // Agents with a zero transfer are skipped: x - (+-0) and x + (+-0) leave every x but -0.0 unchanged, and
    // neither a reward (an integer, or an integer minus a positive penalty) nor the running total is ever -0.0.
    for (u64 it = ballot(tr != 0.0); it; it &= it - 1) {
      const u32 i = ctz64(it);
      const double ti = shfl_f64(tr, i), qi = shfl_f64(share, i);
      if (lane == i) rew -= ti;
      else rew += qi;
      total += ti;
    }
    transfers_total = total;
  }
  // ---------------- running metrics: read-modify-write only the rows this step really changes ----------------
  // (most steps add zero to every accumulator; the rows are 64 B each, ~0.8 KB of traffic per env-step if touched
  // blindly.  The done step needs the per-agent sums for equality / sustainability and loads them regardless.)
  const u32 nmi = CE_MI_COUNT(n), nmf = CE_MF_COUNT(n);
  const auto mi = p.int_metrics + (size_t)E.e * nmi;
  const auto mf = p.f64_metrics + (size_t)E.e * nmf;
  const u32 inc_a = KIND == CE_KIND_CLEANUP ? cleaned : eaten;
  long long m_sr = 0, m_str = 0;
  double f_sr = 0.0, f_str = 0.0;
  CE_STAMP(6);
  if (busy) {
    // eaten / eaten_close are 0/1 flags; cleaned and the base rewards are zero for most agents
    const u32 sum_eaten = popc64(ballot(eaten != 0)), sum_close = popc64(ballot(eaten_close != 0));
    u32 sum_clean = 0;
    i32 sum_rew = 0;
    for (u64 it = ballot(cleaned != 0); it; it &= it - 1) sum_clean += rdl(cleaned, ctz64(it));
    const u64 rew_lanes = ballot(E.is_agent && base_rew != 0);
    for (u64 it = rew_lanes; it; it &= it - 1) sum_rew += shfl_i32(base_rew, ctz64(it));
    if ((sum_eaten | sum_close | sum_clean) != 0 || sum_rew != 0) {
      if (lane < 4) {
        long long g_m = GAT(mi, lane);  // lane k < 4 holds global metric k
        if (lane == CE_MI_TOTAL_APPLES_EATEN) g_m += sum_eaten;
        if (lane == CE_MI_RAW_ENV_REWARDS) g_m += sum_rew;
        if (lane == CE_MI_DIRT_CLEANED && KIND == CE_KIND_CLEANUP) g_m += sum_clean;
        if (lane == CE_MI_LOW_DENSITY_APPLES && KIND == CE_KIND_HARVEST) g_m += sum_close;
        GAT(mi, lane) = g_m;
      }
    }
    if (ballot(inc_a != 0) != 0 && E.is_agent) GAT(mi, CE_MI_AGENT(n, CE_MIA_A, lane)) += inc_a;
    if (KIND == CE_KIND_HARVEST && ballot(eaten_close != 0) != 0 && E.is_agent) GAT(mi, CE_MI_AGENT(n, CE_MIA_B, lane)) += eaten_close;
    if ((rew_lanes != 0 || done) && E.is_agent) {
      m_sr = GAT(mi, CE_MI_AGENT(n, CE_MIA_SUM_R, lane)) + base_rew;
      m_str = GAT(mi, CE_MI_AGENT(n, CE_MIA_SUM_TR, lane)) + (long long)(t - 1) * base_rew;
      if (rew_lanes != 0) {
        GAT(mi, CE_MI_AGENT(n, CE_MIA_SUM_R, lane)) = m_sr;
        GAT(mi, CE_MI_AGENT(n, CE_MIA_SUM_TR, lane)) = m_str;
      }
    }
  }
  double b_sr = 0.0, b_str = 0.0;
  if (busy && (p.flags & CE_FLAG_INEQUITY_AVERSE)) {
    // float env rewards: the reference appends / sums the floats themselves (cleanup_new.py:227-234): raw_rewards = 0;
    // raw_rewards += r[k] in agent order; metrics['raw_env_rewards'] += raw_rewards; total_reward_dict[k].append(r[k])
    double raw = 0.0;
    for (u32 b = 0; b < n; ++b) raw += shfl_f64(rew_env, b);
    if (lane == 0) mf[CE_MF_RAW_ENV_REWARDS_F(n)] += raw;
    if (E.is_agent) {
      b_sr = GAT(mf, CE_MF_BASE_AGENT(n, CE_MFA_BASE_SUM_R, lane)) + rew_env;
      b_str = GAT(mf, CE_MF_BASE_AGENT(n, CE_MFA_BASE_SUM_TR, lane)) + (double)(t - 1) * rew_env;
      GAT(mf, CE_MF_BASE_AGENT(n, CE_MFA_BASE_SUM_R, lane)) = b_sr;
      GAT(mf, CE_MF_BASE_AGENT(n, CE_MFA_BASE_SUM_TR, lane)) = b_str;
    }
  }
  if (busy && p.contract != CE_CONTRACT_NONE) {
    if (transfers_total != 0.0 && lane == 0) mf[CE_MF_TRANSFERS] += transfers_total;
    const bool any_rew = ballot(E.is_agent && rew != 0.0) != 0;  // adding +-0.0 leaves the (never -0.0) sums unchanged
    if ((any_rew || done) && E.is_agent) {
      f_sr = GAT(mf, CE_MF_AGENT(n, CE_MFA_SUM_R, lane)) + rew;
      f_str = GAT(mf, CE_MF_AGENT(n, CE_MFA_SUM_TR, lane)) + (double)(t - 1) * rew;
      if (any_rew) {
        GAT(mf, CE_MF_AGENT(n, CE_MFA_SUM_R, lane)) = f_sr;
        GAT(mf, CE_MF_AGENT(n, CE_MFA_SUM_TR, lane)) = f_str;
      }
    }
  }
  if (E.is_agent) {
    GAT(out.base_reward() + ea, lane) = base_rew;
    GAT(out.reward() + ea, lane) = rew;
    const u32 info2 = eaten | (KIND == CE_KIND_CLEANUP ? cleaned : eaten_close) << 8;  // info[a][0..1] as one short
    *(CE_GPTR(uint16_t))(out.info() + 2 * ea + 2 * lane) = (uint16_t)info2;
  }

  if (done) {
    // equality / sustainability (cleanup_new.py:422-445) and their transferred versions
    const long long sr = E.is_agent ? m_sr : 0;
    const long long str_ = E.is_agent ? m_str : 0;
    long long eq = 0, total = 0;
    for (u32 i = 0; i < n; ++i) {
      const long long ri = shfl_i64(sr, i);
      for (u32 j = 0; j < n; ++j) {
        const long long d = ri - shfl_i64(sr, j);
        eq += d < 0 ? -d : d;
      }
      total += ri;
    }
    const double ts = total == 0 ? 0.001 : (double)total;
    double equality = 1.0 - (double)eq / ((double)(2 * n) * ts);
    const long long den = sr < 1 ? 1 : sr;
    double sust = np_sum_lanes((double)str_ / (double)den, n) / (double)n;
    // the same two formulas on float per-agent sums (held in lanes 0..n-1), in the reference's accumulation order
    auto eq_sust_f64 = [&](double fr, double ftr, double& eq_out, double& sust_out) {
      double e2 = 0.0, tot = 0.0;
      for (u32 i = 0; i < n; ++i) {
        const double ri = shfl_f64(fr, i);
        for (u32 j = 0; j < n; ++j) e2 += fabs(ri - shfl_f64(fr, j));
        tot += ri;
      }
      if (tot == 0.0) tot = 0.001;
      eq_out = 1.0 - e2 / ((double)(2 * n) * tot);
      const double dn = fr > 1.0 ? fr : 1.0;
      sust_out = np_sum_lanes(ftr / dn, n) / (double)n;
    };
    if (p.flags & CE_FLAG_INEQUITY_AVERSE) eq_sust_f64(E.is_agent ? b_sr : 0.0, E.is_agent ? b_str : 0.0, equality, sust);
    double teq = 0.0, tsust = 0.0;
    if (p.contract != CE_CONTRACT_NONE) eq_sust_f64(E.is_agent ? f_sr : 0.0, E.is_agent ? f_str : 0.0, teq, tsust);
    if (lane == 0) {
      mf[CE_MF_EQUALITY] = equality;
      mf[CE_MF_SUSTAINABILITY] = sust;
      mf[CE_MF_TRANSFER_EQUALITY] = teq;
      mf[CE_MF_TRANSFER_SUSTAINABILITY] = tsust;
    }
    __threadfence_block();
    for (u32 k = lane; k < nmi; k += 64) p.final_int_metrics[(size_t)E.e * nmi + k] = mi[k];
    for (u32 k = lane; k < nmf; k += 64) p.final_f64_metrics[(size_t)E.e * nmf + k] = mf[k];
    if (p.flags & CE_FLAG_AUTO_RESET) {
      __threadfence_block();
      E.SP = GAT(p.spawn_perm + (size_t)E.e * 20, min(lane, 19u));  // the spawn list is only ever needed here
      reset_env(E, p, theta, t, fault);
      store_perms(E, p, true, false);
      did_reset = true;
    }
  }

  CE_STAMP(7);
  rng_end_of_op(E.rng);
  // ---------------- state out ----------------
  if (FUSED) {  // the env state stays in registers / LDS for the next step of this launch
    if (lane == 0) out.done()[E.e] = done ? 1 : 0;
    if (!obs_early) write_obs<KIND, true>(E, p, out.obs(), !did_reset);
    return;
  }
  if (!obs_early) store_grid(E, p);
  store_agents(E, p);
  store_perms(E, p, false, E.waste_perm_dirty);  // (the spawn list was written by the reset that changed it)
  if (!diag::ablate_rngstore) store_rng(E, p, !FUSED && E.wt);
  if (lane == 0) {
    p.timestep[E.e] = (i32)t;
    out.done()[E.e] = done ? 1 : 0;
    if (did_reset) p.theta[E.e] = theta;
    if (fault) p.error_flags[E.e] |= fault;
  }
  CE_STAMP(8);
  if (!obs_early) write_obs(E, p, out.obs(), !did_reset);
  CE_STAMP(9);
  CE_REALSTAMP(15);
}

// Entry latency.  A wave can do nothing before its env's state arrives, so the first vector loads should leave as early as
// possible.  Everything they need comes in as kernel ARGUMENTS — the state pointers, the env range, n — fourteen dwords,
// which gfx950 preloads into SGPRs when the wave is launched (-mllvm -amdgpu-kernarg-preload-count=16: the kernel's real
// entry follows a 256-byte compatibility prologue that older firmware runs to fetch them with scalar loads).  The wave
// starts with its addresses in registers instead of two dependent scalar-load round trips (the kernarg segment, then the
// head of the parameter block), ~300 cycles each under load.  The parameter block `pp` is still read for everything that
// is not on the way to the first load (output pointers, flags, contract bounds).
// NFIX: the number of agents as a compile-time constant (0 = the runtime argument).  The step logic is written for any
// n <= 9, with wave-uniform branches on n in every phase (pair tests, group sizes of the feature scan, loop bounds); the
// instance for the headline's n = 8 lets the compiler fold them all (everything is inlined into the kernel, so a literal
// E.n propagates through every phase).
// POLICY: what `call_actions` holds (ce_step_policy, contracts_engine.h: CE_POLICY_*).  0 = the action ids themselves (every
// other entry point); CE_POLICY_BYTES_MOD = one policy byte per agent, action = byte mod |A|; CE_POLICY_ARGMAX_F32 = |A| float
// scores per agent, action = index of the first maximum (Gumbel-max sampling when the policy adds the noise).  The action a
// policy step took is written to `actions_taken`.  Separate instances: the plain step kernel sits exactly at its 64 VGPRs.
// The plane a step launch takes its actions from.  Read-only (const, restrict: loads may be scalarised / hoisted) for every
// action source but CE_POLICY_AHEAD_NOISE, whose launch also WRITES the plane: that instance gets a plain pointer, so the
// compiler may assume nothing about it (the write went through a cast of a const __restrict__ argument).
template <int POLICY> struct ActionPlane { typedef const uint8_t* __restrict__ type; };
template <> struct ActionPlane<CE_POLICY_AHEAD_NOISE> { typedef uint8_t* type; };
template <int POLICY> DEVINL u32 policy_action(typename ActionPlane<POLICY>::type src, size_t ea, u32 lane, bool is_agent, u32 A,
                                                CE_GPTR(const uint8_t) prev_view) {
  if (POLICY == CE_POLICY_AHEAD_NOISE) {
    // the benchmark's closed-loop policy evaluated here: the env's noise byte moves on by the green channel of the cell in front
    // of the agent in the view the previous step (or reset) wrote — view pixel (6, 7) of a 15 x 15 egocentric crop — and the
    // action is the new byte mod |A|.  `prev_view` = this env's block of the observation buffer: read here, at entry, long
    // before this step's own views are written to the same place.
    const auto nz = (CE_GPTR(uint8_t))src + ea;
    const u32 l = is_agent ? lane : 0u;
    const u32 b = ((u32)GAT(nz, l) + (u32)GAT(prev_view, __umul24(l, (u32)kObsAgentStride) + 6u * kObsRowStride + 7u * 3u + 1u)) & 0xffu;
    if (is_agent) GAT(nz, lane) = (uint8_t)b;
    return !is_agent ? 4u : A == 8u ? (b & 7u) : A == 7u ? b % 7u : A == 9u ? b % 9u : b % A;
  }
  if (POLICY == CE_POLICY_BYTES_MOD) {
    const u32 b = is_agent ? (u32)GAT((CE_GPTR(const uint8_t))src + ea, lane) : 4u;
    return A == 8u ? (b & 7u) : A == 7u ? b % 7u : A == 9u ? b % 9u : b % A;
  }
  // CE_POLICY_ARGMAX_F32: scores [E][n][A]; strict '>' keeps the first maximum, a NaN never wins
  const auto sc = (CE_GPTR(const float))src + ea * A;
  const u32 row = __umul24(is_agent ? lane : 0u, A);
  float best = GAT(sc, row);
  u32 arg = 0;
#pragma unroll
  for (u32 k = 1; k < 9u; ++k) {
    const float v = GAT(sc, row + (k < A ? k : 0u));
    if (k < A && (v > best || best != best)) {
      best = v;
      arg = k;
    }
  }
  return is_agent ? arg : 4u;
}

template <int KIND, int NFIX, int POLICY = 0, bool CM = false>
__global__ __launch_bounds__(64 * kWavesPerBlock, KIND == CE_KIND_CLEANUP ? CE_CLEANUP_WAVES : CE_HARVEST_WAVES) void k_grid_step(
    typename ActionPlane<POLICY>::type call_actions, u32 env_first, u32 num_agents, u32* rng_base, uint8_t* grid_base, uint8_t* agents_base,
    uint8_t* waste_perm_base, const GridParams* __restrict__ pp) {
  // fourteen dwords — sixteen user SGPRs less the kernarg segment pointer — are preloaded: exactly these arguments.  The
  // launch has one workgroup per env of its range, so no upper bound travels (kWavesPerBlock == 1).
  static_assert(kWavesPerBlock == 1, "k_grid_step takes no env_end: the grid is the env range");
  const GridParams& p = *pp;
  __shared__ WaveLds<KIND> lds[kWavesPerBlock];
  Env<KIND> E;
  GridParams ph;  // the fields load_env_state reads, from the arguments
  ph.rng = (decltype(ph.rng))rng_base;
  ph.grid = (decltype(ph.grid))grid_base;
  ph.agents = (decltype(ph.agents))agents_base;
  ph.waste_perm = (decltype(ph.waste_perm))waste_perm_base;
#ifdef CE_INSTRUMENTED
  ph.debug = p.debug;  // diagnostic builds stamp through E.dbg
#else
  ph.debug = nullptr;
#endif
  ph.n = NFIX ? (u32)NFIX : (num_agents & 0xffu);  // bit 8 of the argument: this launch stores write-through (Env::wt)
  const auto acts = (CE_GPTR(const uint8_t))call_actions;
  env_begin(E, ph, lds, env_first, 0xffffffffu);
  E.wt = (num_agents & 0x100u) != 0;
  env_geometry<KIND, CM>(E, p);
  const u32 lane = E.lane, n = E.n;
  const size_t ea = (size_t)E.e * n;  // wave-uniform: per-agent arrays are indexed base + 32-bit lane offset

  // the action load is in flight together with the state loads; it is validated before anything is written
  u32 ACT;
  if (POLICY == 0) {
    ACT = E.is_agent ? (u32)GAT(acts + ea, lane) : 4u;
  } else {
    const u32 A = (KIND == CE_KIND_CLEANUP ? 8u : 7u) + ((p.flags & CE_FLAG_FIRING_ENABLED) ? 1u : 0u);
    ACT = policy_action<POLICY>(call_actions, ea, lane, E.is_agent, A, (CE_GPTR(const uint8_t))p.obs + (size_t)E.e * p.obs_env_stride);
    if (E.is_agent) GAT(p.actions_taken + ea, lane) = (uint8_t)ACT;
  }
  CE_STAMP(0);
  CE_REALSTAMP(14);
  load_env_state(E, ph);
  u32 t = (u32)p.timestep[E.e];
  double theta = p.theta[E.e];
  u32 fault = 0;
  bool did_reset = false;
  grid_step_core<KIND, false>(E, p, StepOutDirect{p}, ACT, t, theta, fault, did_reset);
}

// Fused multi-step rollout (ce_rollout_fused): the env's state — map, agent table, persistent lists, MT19937 — is loaded
// once, stays in LDS / registers for `num_steps` consecutive steps and is written back once; every step still reads its
// own action plane and writes its own observation / reward / info / feature / done outputs (plane (plane0 + s) mod
// num_planes of the trajectory arrays).  Results are those of num_steps single-step launches, bit for bit.
// Consumers in the reference that roll whole episodes per call: run_solver.py:35-65, two_stage_train.py:290-333.
// Loop-invariant scalar loads (parameter-block fields, trajectory pointers) and everything derived from the env index
// would be hoisted out of the step loop and then live in registers across the whole step body — far more than there
// are.  Each iteration therefore takes them through a value the compiler cannot see through: a real v_mov inside a
// volatile asm (an empty asm with a "+v" operand would leave the SGPR -> VGPR copy hoistable, i.e. a VGPR held across
// the loop per value), re-asserted wave-uniform with v_readfirstlane.  Two instructions per value and step.
// (Round 4: the value stays in an SGPR — an empty volatile asm with a "+s" operand is just as opaque to the optimiser and costs
// nothing; the v_mov + v_readfirstlane form it replaces was two VALU per value and step and cost spill slots: cleanup rollout
// 11 -> 2 spilled SGPRs, harvest rollout 6 -> 3 spilled VGPRs, feature rollout 10 -> 2.  -DCE_OPAQUE_VMOV keeps the old form for A/B.)
DEVINL u32 opaque_u32(u32 x) {
#ifdef CE_OPAQUE_VMOV
  u32 v;
  asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "s"(x));
  return rfl(v);
#else
  u32 y = rfl(x);  // (folds away when x already lives in an SGPR)
  asm volatile("" : "+s"(y));
  return y;
#endif
}
// ... and for the two read-only argument blocks the pointer is typed into the constant address space, so that the field
// reads stay scalar loads (an integer-built pointer is a flat one to the compiler).
template <class T> DEVINL const T& opaque_block(const T* q) {
  typedef const __attribute__((address_space(4))) T* cptr;
  return *(const T*)(cptr)(((u64)opaque_u32((u32)((u64)q >> 32)) << 32) | opaque_u32((u32)(u64)q));
}

#ifndef CE_HARVEST_ROLLOUT_WAVES
// round 3 (tools/ab.sh, harvest n = 8 x 16 384 envs): 6 waves (80 VGPRs + 2 spilled) 5.21 G, 7 waves (72 + 3) 5.46 G, 8 waves (64 + 7) 5.38 G
#define CE_HARVEST_ROLLOUT_WAVES 7
#endif
// WAVES = the occupancy the register budget is cut for.  The cleanup rollout exists three times: 8 waves / SIMD (64 VGPRs + 21
// spilled) for launches that oversubscribe the machine — occupancy is what hides a wave's dependent chain there: +3 % on the
// headline — and 6 / 4 waves (80 / 128 VGPRs, few / no spills) for launches that leave at most that many waves per SIMD, where
// every wave is resident anyway and spills only cost.  launch_grid_rollout picks by the size of the launch.
// NFIX as for k_grid_step: the instance for n = 8 folds every branch on the number of agents (and frees the register n lives in).
template <int KIND, int WAVES, int NFIX, bool CM = false> __global__ __launch_bounds__(64 * kWavesPerBlock, WAVES) void k_grid_rollout(const GridParams* __restrict__ pp, const RolloutArgs ra_) {
  // the by-value argument block is read in place from the kernarg segment (it follows the 8-byte pp)
  static_assert(alignof(RolloutArgs) == 8, "RolloutArgs sits at kernarg offset 8");
  const RolloutArgs* rap = (const RolloutArgs*)((const char*)__builtin_amdgcn_kernarg_segment_ptr() + 8);
  __shared__ WaveLds<KIND> lds[kWavesPerBlock];
  Env<KIND> E;
  if (!env_begin(E, *pp, lds, rap->env_first, rap->env_end)) return;
  env_geometry<KIND, CM>(E, *pp);
  if (NFIX) {
    E.n = (u32)NFIX;
    E.is_agent = E.lane < E.n;
  }
  u32 ACT = E.is_agent ? (u32)GAT((CE_GPTR(const uint8_t))rap->actions + (size_t)E.e * E.n, E.lane) : 4u;
  load_env_state(E, *pp);
  u32 t = rfl((u32)pp->timestep[E.e]);
  double theta = shfl_f64(pp->theta[E.e], 0);  // wave-uniform: kept in SGPRs across the loop
  u32 fault = 0, pl = rap->plane0;
  bool any_reset = false;
  const u32 num_steps = rap->num_steps;
  for (u32 s = 0; s < num_steps; ++s) {
    const GridParams& p = opaque_block(pp);
    const RolloutArgs& ra = opaque_block(rap);
    // the same for everything derived from the lane id / env index (address offsets, masks, group ids): recomputed per
    // step like in a single-step launch instead of living in registers across the loop (hoisted, they save 65 VALU
    // per step and cost 40 spilled registers: 2.4 G instead of 3.3 G).  The static cell tables (AP / WS) do stay
    // resident: re-fetching them per step cost 4 %
    asm volatile("" : "+v"(E.lane));
    E.n = NFIX ? (u32)NFIX : opaque_u32(E.n);
    E.e = opaque_u32(E.e);
    E.is_agent = E.lane < E.n;
    const u32 lane = E.lane;
    const size_t ea = (size_t)E.e * E.n;
    E.SP = 0;  // lives in HBM between resets (grid_step_core fetches it for an in-launch reset)
    // the next step's actions are in flight while this step runs (the last iteration re-reads its own plane)
    const u32 sn = s + 1 < num_steps ? s + 1 : s;
    const u32 ACTN = E.is_agent ? (u32)GAT((CE_GPTR(const uint8_t))ra.actions + (size_t)sn * ra.action_plane + ea, lane) : 4u;
    bool did_reset = false;
    grid_step_core<KIND, true>(E, p, StepOutPlane{ra, pl}, ACT, t, theta, fault, did_reset);
    ACT = ACTN;
    pl = pl + 1 == ra.num_planes ? 0u : pl + 1;
    // what is carried into the next step as a wave-uniform value must also be one for the compiler (a loop-carried
    // value that merges with anything it takes for divergent becomes a VGPR held across the whole body)
    rng_assert_uniform(E.rng);
    E.rng.twists = rfl(E.rng.twists);
    E.waste_perm_dirty = rfl((u32)E.waste_perm_dirty) != 0;
    t = rfl(t);
    fault = rfl(fault);
    any_reset = rfl((u32)(any_reset || did_reset)) != 0;
    if (did_reset) theta = shfl_f64(theta, 0);  // comes out of VALU double arithmetic
    E.RW = 0;
  }
  const GridParams& p = opaque_block(pp);
  store_grid(E, p);
  store_agents(E, p);
  store_perms(E, p, false, E.waste_perm_dirty);
  store_rng(E, p);
  if (E.lane == 0) {
    p.timestep[E.e] = (i32)t;
    if (any_reset) p.theta[E.e] = theta;
    if (fault) p.error_flags[E.e] |= fault;
  }
}

#ifndef CE_RNG_COUNTER  // the rest of the kernels is built in the first translation unit only (see the top of the file)
// ========================================================================================
// Feature-vector envs: HarvestFeatures (harvest_features.py:60-336) and CleanupFeatures
// (cleanup_features.py:48-309), BASELINE config 0 / SURVEY §8f.1.  Same maps and static tables as the two grid
// kinds, none of the MapEnv logic: moves are claimed in dict order (stayers first), apples / wastes are Python
// lists whose ORDER matters (np.argmin ties), respawn doubles and the spawn shuffle come from CPython's `random`,
// orientations from np.random.  One wave per env; lane a = agent a; lane c + 64 r = apple / waste cell c + 64 r.
// ========================================================================================
template <int GK> struct alignas(16) FeatLds {
  u32 mt_py[kMtN];  // CPython `random` stream
  union {           // the np.random stream is only needed for a moment at a reset (orientations, theta): it borrows
    u32 mt_np[kMtN];  // the space of the per-step working set
    struct {
      u32 U[256];                     // tempered words of up to 128 respawn doubles (one bulk pass; harvest may take two)
      uint8_t pmap[Geo<GK>::PQUADS * 16];  // padded map (PCELLS bytes used): walls + apples / wastes currently present
    } w;
  };
};
constexpr u32 kAbsent = CE_FEAT_ABSENT;
// doubles of the CPython stream a single-step launch of HarvestFeatures fetches speculatively (one per lane): a step
// draws one per absent apple cell — 5-20 in the steady state, 40 at the 99.9th percentile; more takes the general path
constexpr u32 kFeatWindow = 40;

template <int GK> struct FEnv {
  FeatLds<GK>* L;
  Rng py;
  u32 lane, n, e;
  bool is_agent;
  u32 P, O;         // lane a: padded cell, orientation
  u32 AP[3], AS[3];  // lane c, round r: packed apple cell c + 64 r and its list stamp (kAbsent = not present)
  u32 WC[2], WS[2];  // cleanup: waste cells and stamps
  u32 next_a, next_w;
  // Single-step launches do not stage the 2.5 KB CPython generator in LDS up front: a step of HarvestFeatures draws one
  // double per absent apple cell (5-20 in the steady state), so lane q fetches the two words of double q of the window
  // [pos, pos + 128) straight from HBM (win_a / win_b, raw) and only the position is written back.  The full state is
  // brought in (feat_ensure_py) when a step needs more than the window or runs over the generation end (a twist), or
  // for a reset.  py_resident (wave-uniform): the LDS copy is valid.
  bool py_resident;
  u32 win_a, win_b;
  bool stamps_dirty;  // wave-uniform: the list stamps changed since they were loaded
};

DEVINL void rng_bind(Rng& r, u32* mt, u32 pos) {
  r.mt = mt;
  r.pos = rfl(pos);
  r.cbase = 0;
  r.ccount = 0;
  r.cvalid = 0;
  r.cache = 0;
  r.twists = 0;
}
// skip k stream words (k <= 624); the twist is taken only when the position moves past the generation end
DEVINL void rng_skip(Rng& r, u32 k, u32 lane) {
  rng_assert_uniform(r);
  u32 np_ = r.pos + k;
  if (np_ > (u32)kMtN) {
    mt_twist(r.mt, lane);
    r.twists += 1;
    np_ -= (u32)kMtN;
  }
  r.pos = np_;
  r.ccount = 0;
}

template <int GK> DEVINL void feat_base_map(FEnv<GK>& E) {
  // two rounds of 16-byte copies (the LDS map is padded to whole quads); idle lanes of the second repeat the last quad
  const uint4* bsrc = (const uint4*)c_tab[GK].base_pmap;
  uint4* pm128 = (uint4*)E.L->w.pmap;
  static_assert(Geo<GK>::PQUADS > 64 && Geo<GK>::PQUADS <= 128, "two quad rounds");
  const u32 q1 = min(E.lane + 64u, (u32)Geo<GK>::PQUADS - 1u);
  const uint4 w0 = bsrc[E.lane], w1 = bsrc[q1];
  pm128[E.lane] = w0;
  pm128[q1] = w1;
}
// The np.random draws of a reset — n orientations (one masked word each), then the wrapper's theta — taken in one go
// on the stream loaded into the borrowed LDS block and written straight back: the two generators are independent,
// so taking these before the `random` draws of the same reset changes nothing.  Call before the working set is built.
template <int GK> DEVINL void feat_np_draws(FEnv<GK>& E, const GridParams& p, bool with_theta, u32& orient, double& theta) {
  const auto rsrc = p.rng + (size_t)E.e * CE_RNG_WORDS_SELFDRIVE;
  wave_sync();
  for (u32 k = E.lane; k < (u32)kMtN; k += 64) E.L->mt_np[k] = rsrc[k];
  Rng np;
  rng_bind(np, E.L->mt_np, rsrc[kMtN]);
  wave_sync();
  orient = 0;
  for (u32 a = 0; a < E.n; ++a) {
    const u32 w = rng_next(np, E.lane) & 3u;  // legacy randint(0, 4): one masked word
    if (E.lane == a) orient = w;
  }
  theta = (p.flags & CE_FLAG_EXTERNAL_THETA) ? p.theta[E.e] : 0.0;  // external: the caller owns the theta buffer
  if (with_theta && p.contract != CE_CONTRACT_NONE && !(p.flags & CE_FLAG_EXTERNAL_THETA)) {  // SeparateContractSubgameStage.reset two_stage_train.py:163-166
    const double u0 = rng_double(np, E.lane);
    if (u0 > p.null_prob) {
      const double u1 = rng_double(np, E.lane);
      theta = p.contract_low + (p.contract_high - p.contract_low) * u1;
    } else {
      theta = p.contract_low;
    }
  }
  wave_sync();
  if (rfl(np.twists) != 0)
    for (u32 k = E.lane; k < (u32)kMtN; k += 64) rsrc[k] = E.L->mt_np[k];
  if (E.lane == 0) rsrc[kMtN] = np.pos;
  wave_sync();
}
// the CPython generator's 624 key words into LDS (156 x 16 B); the position is bound separately
template <int GK> DEVINL void feat_fetch_py(FEnv<GK>& E, const GridParams& p) {
  const auto rsrc = p.rng + (size_t)E.e * CE_RNG_WORDS_SELFDRIVE;
  const auto src4 = (CE_GPTR(const uint4))(rsrc + CE_RNG_WORDS_GRID);
  uint4* dst4 = (uint4*)E.L->mt_py;
  // unconditional: the third round's idle lanes repeat the last quad (same value, same address)
  const u32 q2 = min(E.lane + 128u, (u32)kMtN / 4 - 1);
  const uint4 r0 = src4[E.lane], r1 = src4[E.lane + 64], r2 = src4[q2];
  dst4[E.lane] = r0;
  dst4[E.lane + 64] = r1;
  dst4[q2] = r2;
  E.py_resident = true;
}
template <int GK> DEVINL void feat_ensure_py(FEnv<GK>& E, const GridParams& p) {
  if (!E.py_resident) {
    wave_sync();
    feat_fetch_py(E, p);
    wave_sync();
  }
}
template <int GK, bool LIGHT = false> DEVINL void feat_load(FEnv<GK>& E, const GridParams& p, bool with_state) {
  typedef Geo<GK> G;
  const GridTables& T = c_tab[GK];
  const u32 lane = E.lane;
  const auto rsrc = p.rng + (size_t)E.e * CE_RNG_WORDS_SELFDRIVE;
  E.py_resident = false;
  E.stamps_dirty = false;
  E.win_a = E.win_b = 0;
  const u32 pypos = rsrc[CE_RNG_WORDS_GRID + kMtN];
  if (LIGHT) {  // lane q < kFeatWindow: the two raw words of double q after the current position (clamped inside the generation)
    const u32 i0 = min(rfl(pypos) + 2u * min(lane, kFeatWindow - 1u), (u32)kMtN - 2u);
    const auto w = rsrc + CE_RNG_WORDS_GRID;
    E.win_a = GAT(w, i0);
    E.win_b = GAT(w, i0 + 1u);
  } else {
    feat_fetch_py(E, p);
  }
  rng_bind(E.py, E.L->mt_py, pypos);
  feat_base_map(E);
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const u32 idx = lane + 64 * r;
    const u32 av = T.apple[min(idx, (u32)G::NAPPLE - 1u)];
    E.AP[r] = idx < (u32)G::NAPPLE ? av : 0;
    E.AS[r] = kAbsent;
  }
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    E.WC[r] = 0;
    E.WS[r] = kAbsent;
    if (G::NWASTE) {  // (HarvestFeatures has no waste cells: nothing to fetch)
      const u32 idx = lane + 64 * r;
      const u32 wv = T.waste[min(idx, (u32)(G::NWASTE ? G::NWASTE - 1 : 0))];
      E.WC[r] = idx < (u32)G::NWASTE ? wv : 0;
    }
  }
  E.next_a = E.next_w = 0;
  E.P = 0xffffu;
  E.O = 0;
  if (with_state) {
    const auto st = (CE_GPTR(const uint16_t))(p.grid + (size_t)E.e * CE_FEAT_STATE_BYTES);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const u32 v = st[min(lane + 64u * r, (u32)CE_FEAT_APPLE_SLOTS - 1u)];
      if (lane + 64 * r < (u32)G::NAPPLE) E.AS[r] = v;
    }
    if (G::NWASTE) {
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const u32 v = st[CE_FEAT_APPLE_SLOTS + min(lane + 64u * r, (u32)CE_FEAT_WASTE_SLOTS - 1u)];
        if (lane + 64 * r < (u32)G::NWASTE) E.WS[r] = v;
      }
    }
    const auto cnt = (CE_GPTR(const u32))(p.grid + (size_t)E.e * CE_FEAT_STATE_BYTES + 2 * (CE_FEAT_APPLE_SLOTS + CE_FEAT_WASTE_SLOTS));
    E.next_a = rfl(cnt[0]);
    E.next_w = rfl(cnt[1]);
    {
      const u32 w = GAT((CE_GPTR(const u32))p.agents + (size_t)E.e * E.n, min(lane, E.n - 1u));
      E.P = E.is_agent ? pad_of<GK>(w & 0xff, (w >> 8) & 0xff) : 0xffffu;
      E.O = E.is_agent ? (w >> 16) & 3 : 0u;
    }
  }
  wave_sync();
}
// presence map: apple / waste cells carry their code only while present
template <int GK> DEVINL void feat_paint(FEnv<GK>& E) {
  typedef Geo<GK> G;
  uint8_t* pm = E.L->w.pmap;
#pragma unroll
  for (int r = 0; r < 3; ++r)
    pm_put(pm, E.lane + 64 * r < (u32)G::NAPPLE, cell_pad(E.AP[r]), E.AS[r] != kAbsent ? CE_CELL_APPLE : CE_CELL_EMPTY);
  if (G::NWASTE) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
      pm_put(pm, E.lane + 64 * r < (u32)G::NWASTE, cell_pad(E.WC[r]), E.WS[r] != kAbsent ? CE_CELL_WASTE : CE_CELL_RIVER);
  }
  wave_sync();
}
// waste_block: also write the waste stamps (HarvestFeatures has none: only its construct / reset launches initialise them)
template <int GK> DEVINL void feat_store(FEnv<GK>& E, const GridParams& p, bool waste_block = true) {
  typedef Geo<GK> G;
  const u32 lane = E.lane;
  wave_sync();
  const auto rdst = p.rng + (size_t)E.e * CE_RNG_WORDS_SELFDRIVE;
  if (E.py_resident && rfl(E.py.twists) != 0) {  // the key words only change at a twist
    const auto dst4 = (CE_GPTR(uint4))(rdst + CE_RNG_WORDS_GRID);
    const uint4* src4 = (const uint4*)E.L->mt_py;
    for (u32 k = lane; k < (u32)kMtN / 4; k += 64) dst4[k] = src4[k];
  }
  if (lane == 0) rdst[CE_RNG_WORDS_GRID + kMtN] = E.py.pos;
  if (E.stamps_dirty) {  // most steps neither eat, spawn nor clean anything: the lists are as they were loaded
    const auto st = (CE_GPTR(uint16_t))(p.grid + (size_t)E.e * CE_FEAT_STATE_BYTES);
#pragma unroll
    for (int r = 0; r < 3; ++r)
      if (lane + 64 * r < CE_FEAT_APPLE_SLOTS) st[lane + 64 * r] = (uint16_t)(lane + 64 * r < (u32)G::NAPPLE ? E.AS[r] : kAbsent);
    if (GK == CE_KIND_CLEANUP || waste_block) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
        if (lane + 64 * r < CE_FEAT_WASTE_SLOTS)
          st[CE_FEAT_APPLE_SLOTS + lane + 64 * r] = (uint16_t)(lane + 64 * r < (u32)G::NWASTE ? E.WS[r] : kAbsent);
    }
    const auto cnt = (CE_GPTR(u32))(p.grid + (size_t)E.e * CE_FEAT_STATE_BYTES + 2 * (CE_FEAT_APPLE_SLOTS + CE_FEAT_WASTE_SLOTS));
    if (lane == 0) {
      cnt[0] = E.next_a;
      cnt[1] = E.next_w;
    }
  }
  if (E.is_agent) GAT((CE_GPTR(u32))p.agents + (size_t)E.e * E.n, lane) = row_of<GK>(E.P) | (col_of<GK>(E.P) << 8) | (E.O << 16);
}

// initialize_arrays: harvest starts with every apple, cleanup with no apple and the H cells as waste
template <int GK> DEVINL void feat_init_arrays(FEnv<GK>& E) {
  typedef Geo<GK> G;
  const GridTables& T = c_tab[GK];
  E.next_a = E.next_w = 0;
  E.stamps_dirty = true;
#pragma unroll
  for (int r = 0; r < 3; ++r) E.AS[r] = kAbsent;
#pragma unroll
  for (int r = 0; r < 2; ++r) E.WS[r] = kAbsent;
  if (GK == CE_KIND_HARVEST) {
#pragma unroll
    for (int r = 0; r < 3; ++r)
      if (E.lane + 64 * r < (u32)G::NAPPLE) E.AS[r] = E.lane + 64 * r;
    E.next_a = (u32)G::NAPPLE;
  } else {
    u32 base = 0;
#pragma unroll
    for (int r = 0; r < 2; ++r) {  // waste_start_points: the 'H' cells in row-major order
      const bool h = E.lane + 64 * r < (u32)G::NWASTE && T.base_pmap[cell_pad(E.WC[r])] == CE_CELL_WASTE;
      const u64 hb = ballot(h);
      if (h) E.WS[r] = base + popc64(hb & ((1ull << E.lane) - 1ull));
      base += popc64(hb);
    }
    E.next_w = base;
  }
}
// initialize_players: random.shuffle(list(range(len(spawn_points)))) with CPython's _randbelow_with_getrandbits
// (top bits of one word per attempt, k = (i + 1).bit_length()), then np.random.randint(0, 4) per agent
template <int GK> DEVINL void feat_init_players(FEnv<GK>& E, u32 orient) {
  const GridTables& T = c_tab[GK];
  const u32 L = GK == CE_KIND_HARVEST ? 20u : 10u;
  u32 IDX = E.lane;
  for (u32 i = L - 1; i >= 1; --i) {
    const u32 nn = i + 1, k = 32u - (u32)__builtin_clz(nn);
    u32 r = rng_next(E.py, E.lane) >> (32 - k);
    while (r >= nn) r = rng_next(E.py, E.lane) >> (32 - k);
    const u32 vi = rdl(IDX, i), vj = rdl(IDX, r);
    IDX = wrl(vj, i, IDX);
    IDX = wrl(vi, r, IDX);
  }
  const u32 cell = T.spawn[IDX < 20 ? IDX : 0];
  E.P = E.is_agent ? cell_pad(cell) : 0xffffu;
  E.O = orient;  // np.random.randint(0, 4) per agent, drawn by feat_np_draws
}

// r < p for a double given as tempered words (a, b): exact 53-bit integer compare against ceil(p * 2^53)
DEVINL bool dbl_below(u32 a, u32 b, u64 thr) { return ((((u64)(a >> 5)) << 26) | (u64)(b >> 6)) < thr; }

template <int GK> DEVINL void feat_spawn(FEnv<GK>& E, const GridParams& p) {
  typedef Geo<GK> G;
  const GridTables& T = c_tab[GK];
  const u32 lane = E.lane;
  const u64 lt = (1ull << lane) - 1ull;
  uint8_t* pm = E.L->w.pmap;
  constexpr int AR = (G::NAPPLE + 63) / 64;
  // agent presence on apple cells: `apple_pos not in self.agent_pos.values()`
  bool elig[AR];
  u32 ri[AR];
  u32 nelig = 0;
#pragma unroll
  for (int r = 0; r < AR; ++r) {
    bool on = false;
    const u32 cell = cell_pad(E.AP[r]);
    for (u32 a = 0; a < E.n; ++a) on = on || rdl(E.P, a) == cell;
    elig[r] = lane + 64 * r < (u32)G::NAPPLE && E.AS[r] == kAbsent && !on;
    const u64 eb = ballot(elig[r]);
    ri[r] = nelig + popc64(eb & lt);
    nelig += popc64(eb);
  }
  if (GK == CE_KIND_HARVEST) {
    // HarvestFeatures' common case: at most 64 eligible cells and their doubles inside the current generation.  The
    // eligible cells are compacted to one lane round — lane q = the q-th eligible cell in row-major order with double q
    // of the stream — so the neighbour counts, the fixed-point iteration and the list appends below run once instead
    // of once per 64 apple cells, and the stream words come straight from the window (registers, or the resident LDS
    // state of a fused rollout): nothing of the generator but its position changes.
    rng_assert_uniform(E.py);
    if (nelig == 0) return;
    if (nelig <= (E.py_resident ? 64u : kFeatWindow) && E.py.pos + 2u * nelig <= (u32)kMtN) {
      u32 wa = E.win_a, wb = E.win_b;
      if (E.py_resident) {
        const u32 i0 = min(E.py.pos + 2u * lane, (u32)kMtN - 2u);
        wa = E.py.mt[i0];
        wb = E.py.mt[i0 + 1u];
      }
      u32* U = E.L->w.U;  // [0, 64): the compacted list; [64, 256): the stamp each owner (cell index) gets back
      wave_sync();
#pragma unroll
      for (int r = 0; r < AR; ++r)
        if (elig[r]) U[ri[r]] = cell_pad(E.AP[r]) | (lane + 64u * r) << 16;
      wave_sync();
      const bool act = lane < nelig;
      const u32 ent = U[act ? lane : 0u];
      const i32 cell = (i32)(ent & 0xffffu);  // idle lanes alias entry 0: a valid interior cell, never written by them
      const u32 owner = ent >> 16;
      const u64 X = (((u64)(mt_temper(wa) >> 5)) << 26) | (u64)(mt_temper(wb) >> 6);
      const u64 t0 = T.apple_thresh[0], t1 = T.apple_thresh[1], t2 = T.apple_thresh[2], t3 = T.apple_thresh[3];
      bool sp = false;
      for (;;) {
        u32 num = 0;
#pragma unroll
        for (int j = -1; j <= 1; ++j)
#pragma unroll
          for (int k = -1; k <= 1; ++k) {
            const uint8_t c = pm[cell + j * G::PW + k];
            const bool earlier = j < 0 || (j == 0 && k < 0);  // row-major: this pass's spawns above / to the left count
            num += (c == CE_CELL_APPLE || (earlier && c == 0x42)) ? 1u : 0u;
          }
        const u64 thr = num == 0 ? t0 : num == 1 ? t1 : num == 2 ? t2 : t3;
        const bool s2 = act && X < thr;
        const bool changed = s2 != sp;
        sp = s2;
        if (ballot(changed) == 0) break;
        wave_sync();
        pm_put(pm, act, (u32)cell, sp ? 0x42u : (u32)CE_CELL_EMPTY);  // 0x42 = spawned in this pass
        wave_sync();
      }
      wave_sync();
      const u64 sb = ballot(sp);
      const u32 stamp = E.next_a + popc64(sb & lt);  // appended in row-major order: stamps continue the list
      if (sp) pm[cell] = CE_CELL_APPLE;
      if (act) U[64u + owner] = sp ? stamp : kAbsent;
      wave_sync();
#pragma unroll
      for (int r = 0; r < AR; ++r)
        if (elig[r]) E.AS[r] = U[64u + lane + 64u * r];
      E.next_a += popc64(sb);
      if (sb) E.stamps_dirty = true;
      E.py.pos += 2u * nelig;
      E.py.ccount = 0;
      wave_sync();
      return;
    }
  }
  feat_ensure_py(E, p);  // the general path walks the stream in LDS (and may twist it)
  E.stamps_dirty = true;
  u64 thr_uniform = 0;
  bool waste_on = false;
  u32 nwaste = 0;
  if (GK == CE_KIND_CLEANUP) {  // compute_probabilities on the current waste count (same constants as cleanup_new)
#pragma unroll
    for (int r = 0; r < 2; ++r) nwaste += popc64(ballot(lane + 64 * r < (u32)G::NWASTE && E.WS[r] != kAbsent));
    const u64 te = T.apple_thresh[nwaste];
    waste_on = (te & kWasteOnBit) != 0;
    thr_uniform = te & ~kWasteOnBit;
  }
  // every eligible cell consumes exactly one random.random(): temper the 2 * nelig words in stream order
  constexpr u32 kBulkDoubles = sizeof(E.L->w.U) / 8;  // doubles per bulk pass (the scratch is kept small: 8 waves/SIMD)
  bool spawn[AR];
  if (GK == CE_KIND_CLEANUP) {
    static_assert(GK != CE_KIND_CLEANUP || (u32)G::NAPPLE <= kBulkDoubles, "one bulk pass covers every apple cell");
    rng_bulk(E.py, E.L->w.U, nullptr, 2 * nelig, 2 * nelig, false, lane);
#pragma unroll
    for (int r = 0; r < AR; ++r) spawn[r] = elig[r] && dbl_below(E.L->w.U[2 * (elig[r] ? ri[r] : 0)], E.L->w.U[2 * (elig[r] ? ri[r] : 0) + 1], thr_uniform);
  } else {
    // spawn_apples (harvest_features.py:139-151): the neighbour count of a cell includes apples spawned EARLIER in
    // this very loop (row-major order).  More neighbours never lower the probability, so iterating the parallel
    // decision from "no new apples" upwards converges to the sequential result (the system is triangular).
    u32 a_w[AR], b_w[AR];
#pragma unroll
    for (int r = 0; r < AR; ++r) {
      a_w[r] = b_w[r] = 0;
      spawn[r] = false;
    }
    for (u32 base = 0; base < nelig; base += kBulkDoubles) {  // the stream is sequential: passes of <= 128 doubles
      const u32 cnt = nelig - base < kBulkDoubles ? nelig - base : kBulkDoubles;
      rng_bulk(E.py, E.L->w.U, nullptr, 2 * cnt, 2 * cnt, false, lane);
#pragma unroll
      for (int r = 0; r < AR; ++r) {
        const u32 q = ri[r] - base;
        if (elig[r] && q < cnt) {
          a_w[r] = E.L->w.U[2 * q];
          b_w[r] = E.L->w.U[2 * q + 1];
        }
      }
      wave_sync();  // the next pass overwrites the scratch
    }
    for (;;) {
      bool changed = false;
#pragma unroll
      for (int r = 0; r < AR; ++r) {
        u32 num = 0;
        if (elig[r]) {
          const i32 cell = (i32)cell_pad(E.AP[r]);
          // earlier cells in row-major order: the row above and the left neighbour see this pass's spawns
#pragma unroll
          for (int j = -1; j <= 1; ++j)
#pragma unroll
            for (int k = -1; k <= 1; ++k) {
              const uint8_t c = pm[cell + j * G::PW + k];
              const bool earlier = j < 0 || (j == 0 && k < 0);
              num += (c == CE_CELL_APPLE || (earlier && c == 0x42)) ? 1u : 0u;
            }
        }
        const bool s2 = elig[r] && dbl_below(a_w[r], b_w[r], T.apple_thresh[num < 3 ? num : 3]);
        changed = changed || (s2 != spawn[r]);
        spawn[r] = s2;
      }
      if (ballot(changed) == 0) break;
      wave_sync();
#pragma unroll
      for (int r = 0; r < AR; ++r)
        pm_put(pm, elig[r], cell_pad(E.AP[r]), spawn[r] ? 0x42u : (u32)CE_CELL_EMPTY);  // 0x42 = spawned in this pass
      wave_sync();
    }
  }
  // append in row-major order: stamps continue the list
  wave_sync();
#pragma unroll
  for (int r = 0; r < AR; ++r) {
    const u64 sb = ballot(spawn[r]);
    if (spawn[r]) {
      E.AS[r] = E.next_a + popc64(sb & lt);
      pm[cell_pad(E.AP[r])] = CE_CELL_APPLE;
    }
    E.next_a += popc64(sb);
  }
  if (GK == CE_KIND_CLEANUP) {
    // at most one waste: walk the absent waste cells in row-major order, one random.random() each, stop at the
    // first r < p_waste (p_waste is 0.5 or 0: decided by the sign of the tempered first word, or never)
    bool cand[2];
    u32 tq[2], ncand = 0;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      cand[r] = lane + 64 * r < (u32)G::NWASTE && E.WS[r] == kAbsent;
      const u64 cb = ballot(cand[r]);
      tq[r] = ncand + popc64(cb & lt);
      ncand += popc64(cb);
    }
    u32 done = 0, hit = 0xffffffffu;  // candidates walked so far; index of the successful one
    while (done < ncand && hit == 0xffffffffu) {
      rng_assert_uniform(E.py);
      if (E.py.pos >= (u32)kMtN) {
        mt_twist(E.py.mt, lane);
        E.py.twists += 1;
        E.py.pos = 0;
      }
      rng_refill(E.py, lane);  // cache = tempered words pos .. pos + ccount - 1 (nothing consumed yet)
      const u32 vis = (E.py.ccount + 1) / 2;  // doubles whose first word is in the cache
      const u32 take = ncand - done < vis ? ncand - done : vis;
      u64 sb = 0;
      if (waste_on) sb = ballot(lane < 2 * take && (lane & 1u) == 0 && (i32)E.py.cache >= 0);  // u < 0.5 <=> bit 31 clear
      u32 used = take;
      if (sb) {
        used = ctz64(sb) / 2 + 1;
        hit = done + used - 1;
      }
      rng_skip(E.py, 2 * used, lane);
      done += used;
    }
    if (hit != 0xffffffffu) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
        if (cand[r] && tq[r] == hit) {
          E.WS[r] = E.next_w;
          pm[cell_pad(E.WC[r])] = CE_CELL_WASTE;
        }
      E.next_w += 1;
    }
  }
  wave_sync();
}

// apples within j^2 + k^2 <= 5 of a padded cell (uniform), current presence map
template <int GK> DEVINL u32 feat_close_count(FEnv<GK>& E, u32 cell) {
  const GridTables& T = c_tab[CE_KIND_HARVEST];
  const bool v = E.lane < 21 && E.L->w.pmap[(i32)cell + (i32)T.close_off[E.lane < 21 ? E.lane : 0]] == CE_CELL_APPLE;
  return popc64(ballot(v));
}

// feature vector (also the observation); closest apple / waste = min over (manhattan, list stamp)
template <int GK> DEVINL u32 feat_features(FEnv<GK>& E, const GridParams& p, CE_GPTR(int16_t) features, u32 cleaned) {
  typedef Geo<GK> G;
  const u32 lane = E.lane, n = E.n, nf = p.num_features;
  const auto f = features + ((size_t)E.e * n + (E.is_agent ? lane : 0)) * nf;
  const u32 cp = n > 1 ? 1u : 0u;  // compute_closest_pos: a0 -> a1, everyone else -> a0 (inf - inf = nan argmin)
  const u32 p_a0 = rdl(E.P, 0), o_a0 = rdl(E.O, 0), p_cp = rdl(E.P, cp), o_cp = rdl(E.O, cp);
  const u32 cpp = lane == 0 ? p_cp : p_a0, cpo = lane == 0 ? o_cp : o_a0;
  u32 napples = 0, nwastes = 0;
#pragma unroll
  for (int r = 0; r < 3; ++r) napples += popc64(ballot(lane + 64 * r < (u32)G::NAPPLE && E.AS[r] != kAbsent));
#pragma unroll
  for (int r = 0; r < 2; ++r) nwastes += popc64(ballot(lane + 64 * r < (u32)G::NWASTE && E.WS[r] != kAbsent));
  u32 ca = 0, cw = 0, close_now = 0;
  for (u32 a = 0; a < n; ++a) {
    const u32 pa = rdl(E.P, a);
    const u32 prc = col_of<GK>(pa) | row_of<GK>(pa) << 8;
    u32 key = 0xffffffffu;
#pragma unroll
    for (int r = 0; r < 3; ++r)
      if (lane + 64 * r < (u32)G::NAPPLE && E.AS[r] != kAbsent) {
        const u32 k2 = __builtin_amdgcn_sad_u8(cell_rc(E.AP[r]), prc, 0u) << 24 | E.AS[r] << 8 | (u32)r << 6 | lane;
        key = k2 < key ? k2 : key;
      }
    const u32 best = wave_min_u32(key);
    u32 rc = 0;
    if (best != 0xffffffffu) {
      const u32 src = best & 63u, rr = (best >> 6) & 3u;
      const u32 c0 = rdl(E.AP[0], src), c1 = rdl(E.AP[1], src), c2 = rdl(E.AP[2], src);
      rc = cell_rc(rr == 0 ? c0 : rr == 1 ? c1 : c2);
    }
    if (lane == a) ca = rc;
    if (GK == CE_KIND_CLEANUP) {
      u32 keyw = 0xffffffffu;
#pragma unroll
      for (int r = 0; r < 2; ++r)
        if (lane + 64 * r < (u32)G::NWASTE && E.WS[r] != kAbsent) {
          const u32 k2 = __builtin_amdgcn_sad_u8(cell_rc(E.WC[r]), prc, 0u) << 24 | E.WS[r] << 8 | (u32)r << 6 | lane;
          keyw = k2 < keyw ? k2 : keyw;
        }
      const u32 bw = wave_min_u32(keyw);
      u32 rcw = 0;
      if (bw != 0xffffffffu) {
        const u32 src = bw & 63u;
        const u32 c0 = rdl(E.WC[0], src), c1 = rdl(E.WC[1], src);
        rcw = cell_rc(((bw >> 6) & 1u) ? c1 : c0);
      }
      if (lane == a) cw = rcw;
    } else {
      const u32 cnt = feat_close_count(E, pa);
      if (lane == a) close_now = cnt;
    }
  }
  if ((nf & 1u) == 0) {  // dword-aligned rows (every harvest row, cleanup with an even n): a few wide stores per agent
    typedef u32 u32x4 __attribute__((ext_vector_type(4)));
    const auto f32 = (CE_GPTR(u32))f;
    const u32x4 head = {row_of<GK>(E.P) | col_of<GK>(E.P) << 16, E.O | row_of<GK>(cpp) << 16, col_of<GK>(cpp) | cpo << 16,
                        (ca >> 8) | (ca & 0xffu) << 16};
    if (GK == CE_KIND_CLEANUP) {
      for (u32 b = 0; b < n; b += 2) {
        const u32 w = rdl(cleaned, b) | rdl(cleaned, b + 1) << 16;
        if (E.is_agent) f32[6 + (b >> 1)] = w;
      }
      if (E.is_agent) {
        *(CE_GPTR(u32x4))(f32) = head;
        f32[4] = (cw >> 8) | (cw & 0xffu) << 16;
        f32[5] = napples | nwastes << 16;
      }
    } else if (E.is_agent) {
      *(CE_GPTR(u32x4))(f32) = head;
      f32[4] = close_now | napples << 16;
      for (u32 b = 0; b < n; ++b) f32[5 + b] = 0u;
    }
    return close_now;
  }
  if (E.is_agent) {
    f[0] = (int16_t)row_of<GK>(E.P);
    f[1] = (int16_t)col_of<GK>(E.P);
    f[2] = (int16_t)E.O;
    f[3] = (int16_t)row_of<GK>(cpp);
    f[4] = (int16_t)col_of<GK>(cpp);
    f[5] = (int16_t)cpo;
    f[6] = (int16_t)(ca >> 8);
    f[7] = (int16_t)(ca & 0xffu);
  }
  if (GK == CE_KIND_CLEANUP) {
    if (E.is_agent) {
      f[8] = (int16_t)(cw >> 8);
      f[9] = (int16_t)(cw & 0xffu);
      f[10] = (int16_t)napples;
      f[11] = (int16_t)nwastes;
    }
    for (u32 b = 0; b < n; ++b) {
      const u32 cb = rdl(cleaned, b);
      if (E.is_agent) f[12 + b] = (int16_t)cb;
    }
  } else if (E.is_agent) {
    f[8] = (int16_t)close_now;
    f[9] = (int16_t)napples;
    for (u32 b = 0; b < 2 * n; ++b) f[10 + b] = 0;
  }
  return close_now;
}

template <int GK> DEVINL bool feat_begin(FEnv<GK>& E, const GridParams& p, FeatLds<GK>* lds, u32 env_first, u32 env_end) {
  E.lane = lane_id();
  E.e = rfl(env_first + blockIdx.x);
  E.n = p.n;
  E.is_agent = E.lane < E.n;
  E.L = lds;
  return E.e < env_end;
}
template <int GK> DEVINL void feat_zero_outputs(FEnv<GK>& E, const GridParams& p) {
  const u32 nmi = CE_MI_COUNT(E.n), nmf = CE_MF_COUNT(E.n);
  for (u32 k = E.lane; k < nmi; k += 64) p.int_metrics[(size_t)E.e * nmi + k] = 0;
  for (u32 k = E.lane; k < nmf; k += 64) p.f64_metrics[(size_t)E.e * nmf + k] = 0.0;
  if (E.is_agent) {
    const size_t ea = (size_t)E.e * E.n;
    (p.base_reward + ea)[E.lane] = 0;
    (p.reward + ea)[E.lane] = 0.0;
    (p.info + 2 * ea)[2 * E.lane] = 0;
    (p.info + 2 * ea)[2 * E.lane + 1] = 0;
  }
}
// reset(): initialize_arrays, initialize_players, spawn, metrics, theta (wrapper), observation
template <int GK> DEVINL void feat_reset_env(FEnv<GK>& E, const GridParams& p, u32 orient) {
  feat_init_arrays(E);
  feat_ensure_py(E, p);  // the spawn shuffle draws from the CPython stream word by word
  feat_init_players(E, orient);
  feat_paint(E);
  feat_spawn(E, p);
}

template <int GK> __global__ __launch_bounds__(64) void k_feat_construct(const GridParams* __restrict__ pp, const uint8_t* __restrict__ call_actions,
                                                                        const uint8_t* __restrict__ call_mask, u32 env_first, u32 env_end) {
  const GridParams& p = *pp;
  __shared__ FeatLds<GK> lds;
  FEnv<GK> E;
  if (!feat_begin(E, p, &lds, env_first, env_end)) return;
  if (call_mask && call_mask[E.e] == 0) return;
  u32 orient;
  double theta_unused;
  feat_np_draws(E, p, false, orient, theta_unused);
  feat_load(E, p, false);
  feat_init_arrays(E);  // __init__: initialize_arrays, (compute_probabilities), initialize_players
  feat_init_players(E, orient);
  feat_zero_outputs(E, p);
  E.stamps_dirty = true;
  feat_store(E, p);
  if (E.lane == 0) {
    p.timestep[E.e] = 0;
    p.theta[E.e] = 0.0;
    p.done[E.e] = 0;
    p.error_flags[E.e] = 0;
  }
}

template <int GK> __global__ __launch_bounds__(64) void k_feat_reset(const GridParams* __restrict__ pp, const uint8_t* __restrict__ call_actions,
                                                                    const uint8_t* __restrict__ call_mask, u32 env_first, u32 env_end) {
  const GridParams& p = *pp;
  __shared__ FeatLds<GK> lds;
  FEnv<GK> E;
  if (!feat_begin(E, p, &lds, env_first, env_end)) return;
  if (call_mask && call_mask[E.e] == 0) return;
  u32 orient;
  double theta;
  feat_np_draws(E, p, true, orient, theta);
  feat_load(E, p, false);
  feat_reset_env(E, p, orient);
  feat_zero_outputs(E, p);
  feat_features(E, p, p.features, 0u);
  E.stamps_dirty = true;
  feat_store(E, p);
  if (E.lane == 0) {
    p.timestep[E.e] = 0;
    p.theta[E.e] = theta;
    p.done[E.e] = 0;
    p.error_flags[E.e] = 0;
  }
}

// One step of a feature-vector env on the state held in E / LDS (see grid_step_core for FUSED / OUT)
template <int GK, bool FUSED, class OUT>
DEVINL void feat_step_core(FEnv<GK>& E, const GridParams& p, const OUT& out, u32 ACT, u32& t, double& theta, u32& fault,
                           bool& did_reset) {
  typedef Geo<GK> G;
  const u32 lane = E.lane, n = E.n;
  const size_t ea = (size_t)E.e * n;
  constexpr bool harvest = GK == CE_KIND_HARVEST;
  if (ballot(E.is_agent && ACT > (harvest ? 7u : 8u)) != 0) {  // the step is not taken
    fault |= CE_FAULT_BAD_ACTION;
    return;
  }
  if (FUSED) {  // the presence map is rebuilt from the list stamps held in registers (a single-step launch does the same after its load)
    feat_base_map(E);
    wave_sync();
  }
  feat_paint(E);
  uint8_t* pm = E.L->w.pmap;

  // ---- move_squares: an insertion-ordered dict — stayers first, then movers in key order; a mover is refused by
  // a wall or by any square claimed so far (harvest_features.py:176-196 / cleanup_features.py:165-183) ----
  const bool stay = E.is_agent && (harvest ? ACT > 3 : ACT == 4);
  const bool mover = E.is_agent && ACT < 4;
  const i32 delta = ACT == 0 ? -1 : ACT == 1 ? 1 : ACT == 2 ? -(i32)G::PW : (i32)G::PW;  // MOVE_ACTIONS
  const u32 tgt = mover ? (u32)((i32)E.P + delta) : E.P;
  const bool wall = mover && pm[mover ? tgt : 0] == CE_CELL_WALL;
  u32 SQ = E.P;
  u64 has = ballot(stay);
  for (u64 it = ballot(mover); it; it &= it - 1) {
    const u32 a = ctz64(it);
    const u32 ta = rdl(tgt, a);
    const bool blocked = bit(ballot(wall), a) || (ballot(bit(has, lane) && SQ == ta) != 0);
    if (lane == a && !blocked) SQ = ta;
    has |= 1ull << a;
  }
  if (bit(has, lane)) E.P = SQ;
  if (!E.is_agent) E.P = 0xffffu;

  // ---- consume apples in move_squares order (stayers, then movers) ----
  u32 rew = 0, eaten = 0, eaten_close = 0, cleaned = 0;
  for (int pass = 0; pass < 2; ++pass)
    for (u64 it = pass == 0 ? ballot(stay) : ballot(mover); it; it &= it - 1) {
      const u32 a = ctz64(it);
      const u32 pa = rdl(E.P, a);
      if (pm[pa] != CE_CELL_APPLE) continue;
      u32 close = 0;
      if (harvest) close = feat_close_count(E, pa);  // counted before the apple is removed
      if (lane == a) {
        rew += 1;
        if (harvest) {
          eaten += 1;
          if (close < 4) eaten_close += 1;
        }
      }
      wave_sync();
      pm_put(pm, lane == 0, pa, CE_CELL_EMPTY);
#pragma unroll
      for (int r = 0; r < 3; ++r)
        if (lane + 64 * r < (u32)G::NAPPLE && cell_pad(E.AP[r]) == pa) E.AS[r] = kAbsent;
      E.stamps_dirty = true;
      wave_sync();
    }
  // ---- rotations ----
  if (E.is_agent && (ACT == 5 || ACT == 6)) E.O = (E.O + (ACT == 5 ? 1u : 3u)) & 3u;
  // ---- clean beams (cleanup act 7): three rays of 6 cells starting in the agent's own row, through waste, until a wall ----
  if (!harvest) {
    for (u64 it = ballot(E.is_agent && ACT == 7); it; it &= it - 1) {
      const u32 a = ctz64(it);
      const u32 o = rdl(E.O, a), p0 = rdl(E.P, a);
      const i32 dd = dir_delta(G::PW, o), ss = dir_delta(G::PW, (o + 1) & 3);  // FIRE_DIRECTIONS[o], [(o+1)%4]
      const u32 ray = lane / 6, j = lane - 6 * ray;
      const bool in_beam = lane < 18;
      const i32 start = (i32)p0 + (ray == 1 ? ss : ray == 2 ? -ss : 0);
      const u32 cell = in_beam ? (u32)(start + (i32)j * dd) : 0u;
      const uint8_t code = pm[cell];
      const u64 wb = ballot(in_beam && code == CE_CELL_WALL);
      const u32 rb = (u32)(wb >> (6 * ray)) & 63u;
      const u32 first_wall = rb ? (u32)__builtin_ctz(rb) : 6u;
      const bool hitw = in_beam && j < first_wall && code == CE_CELL_WASTE;
      const u32 c = popc64(ballot(hitw));
      if (lane == a) cleaned = c;
      wave_sync();
      pm_put(pm, hitw, cell, CE_CELL_RIVER);
      if (c) E.stamps_dirty = true;
      for (u64 hw = ballot(hitw); hw; hw &= hw - 1) {
        const u32 cw = rdl(cell, ctz64(hw));
#pragma unroll
        for (int r = 0; r < 2; ++r)
          if (lane + 64 * r < (u32)G::NWASTE && cell_pad(E.WC[r]) == cw) E.WS[r] = kAbsent;
      }
      wave_sync();
    }
  }
  // ---- spawn, features ----  (-DCE_FEAT_ABLATE=1 / 2: instruction-count probes, wrong results, never shipped)
#if !defined(CE_FEAT_ABLATE) || CE_FEAT_ABLATE != 1
  feat_spawn(E, p);
#endif
#if defined(CE_FEAT_ABLATE) && CE_FEAT_ABLATE == 2
  const u32 feat8 = 0;
#else
  const u32 feat8 = feat_features(E, p, out.features(), cleaned);
#endif
  t += 1;
  const bool done = t == p.horizon;
  // ---- metrics (same layout as the grid kinds) ----
  const u32 nmi = CE_MI_COUNT(n), nmf = CE_MF_COUNT(n);
  const auto mi = p.int_metrics + (size_t)E.e * nmi;
  const auto mf = p.f64_metrics + (size_t)E.e * nmf;
  // read-modify-write only the rows this step really changes (most steps add zero to every accumulator); the done
  // step needs the per-agent sums for equality / sustainability and loads them regardless
  double rw = (double)rew;
  long long m_sr = 0, m_str = 0;
  double f_sr = 0.0, f_str = 0.0;
  {
    u32 sum_eaten = 0, sum_close = 0, sum_clean = 0, sum_rew = 0;
    const u64 touched = ballot(E.is_agent && (eaten | eaten_close | cleaned | rew) != 0);
    for (u64 it = touched; it; it &= it - 1) {
      const u32 b = ctz64(it);
      sum_eaten += rdl(eaten, b);
      sum_close += rdl(eaten_close, b);
      sum_clean += rdl(cleaned, b);
      sum_rew += rdl(rew, b);
    }
    if (touched != 0 && lane < 4) {
      long long g_m = GAT(mi, lane);  // lane k < 4 holds global metric k
      if (lane == CE_MI_TOTAL_APPLES_EATEN) g_m += sum_eaten;
      if (lane == CE_MI_RAW_ENV_REWARDS) g_m += sum_rew;
      if (lane == CE_MI_DIRT_CLEANED) g_m += sum_clean;
      if (lane == CE_MI_LOW_DENSITY_APPLES) g_m += sum_close;
      GAT(mi, lane) = g_m;
    }
    const u32 inc_a = harvest ? eaten : cleaned;
    if (ballot(inc_a != 0) != 0 && E.is_agent) GAT(mi, CE_MI_AGENT(n, CE_MIA_A, lane)) += inc_a;
    if (harvest && ballot(eaten_close != 0) != 0 && E.is_agent) GAT(mi, CE_MI_AGENT(n, CE_MIA_B, lane)) += eaten_close;
    const u64 rew_lanes = ballot(E.is_agent && rew != 0);
    if ((rew_lanes != 0 || done) && E.is_agent) {
      m_sr = GAT(mi, CE_MI_AGENT(n, CE_MIA_SUM_R, lane)) + rew;
      m_str = GAT(mi, CE_MI_AGENT(n, CE_MIA_SUM_TR, lane)) + (long long)(t - 1) * rew;
      if (rew_lanes != 0) {
        GAT(mi, CE_MI_AGENT(n, CE_MIA_SUM_R, lane)) = m_sr;
        GAT(mi, CE_MI_AGENT(n, CE_MIA_SUM_TR, lane)) = m_str;
      }
    }
  }
  if (p.contract != CE_CONTRACT_NONE) {  // SeparateContractEnv.step two_stage_train.py:62-121
    double tr;
    if (p.contract == CE_CONTRACT_CLEANUP) tr = -theta * (double)cleaned;
    else tr = (feat8 < 4 && eaten_close > 0) ? theta : 0.0;
    double total = 0.0;
    const double share = tr / (double)(n - 1);
    // agents with a zero transfer are skipped: adding +-0.0 changes neither a reward nor the total (never -0.0)
    for (u64 it = ballot(E.is_agent && tr != 0.0); it; it &= it - 1) {
      const u32 i = ctz64(it);
      const double ti = shfl_f64(tr, i), qi = shfl_f64(share, i);
      if (lane == i) rw -= ti;
      else rw += qi;
      total += ti;
    }
    if (total != 0.0 && lane == 0) mf[CE_MF_TRANSFERS] += total;
    const bool any_rew = ballot(E.is_agent && rw != 0.0) != 0;
    if ((any_rew || done) && E.is_agent) {
      f_sr = GAT(mf, CE_MF_AGENT(n, CE_MFA_SUM_R, lane)) + rw;
      f_str = GAT(mf, CE_MF_AGENT(n, CE_MFA_SUM_TR, lane)) + (double)(t - 1) * rw;
      if (any_rew) {
        GAT(mf, CE_MF_AGENT(n, CE_MFA_SUM_R, lane)) = f_sr;
        GAT(mf, CE_MF_AGENT(n, CE_MFA_SUM_TR, lane)) = f_str;
      }
    }
  }
  if (E.is_agent) {
    GAT(out.base_reward() + ea, lane) = (i32)rew;
    GAT(out.reward() + ea, lane) = rw;
    const u32 info2 = eaten | (harvest ? eaten_close : cleaned) << 8;  // info[a][0..1] as one short
    *(CE_GPTR(uint16_t))(out.info() + 2 * ea + 2 * lane) = (uint16_t)info2;
  }
  if (done) {  // compute_equality / compute_sustainability (+ the transferred versions of the wrapper)
    const long long sr = E.is_agent ? m_sr : 0, str_ = E.is_agent ? m_str : 0;
    long long eq = 0, total = 0;
    for (u32 i = 0; i < n; ++i) {
      const long long ri = shfl_i64(sr, i);
      for (u32 j = 0; j < n; ++j) {
        const long long d = ri - shfl_i64(sr, j);
        eq += d < 0 ? -d : d;
      }
      total += ri;
    }
    const double ts = total == 0 ? 0.001 : (double)total;
    const double equality = 1.0 - (double)eq / ((double)(2 * n) * ts);
    const long long den = sr < 1 ? 1 : sr;
    const double sust = np_sum_lanes((double)str_ / (double)den, n) / (double)n;
    double teq = 0.0, tsust = 0.0;
    if (p.contract != CE_CONTRACT_NONE) {
      const double fr = E.is_agent ? f_sr : 0.0, ftr = E.is_agent ? f_str : 0.0;
      double e2 = 0.0, tot = 0.0;
      for (u32 i = 0; i < n; ++i) {
        const double ri = shfl_f64(fr, i);
        for (u32 j = 0; j < n; ++j) e2 += fabs(ri - shfl_f64(fr, j));
        tot += ri;
      }
      if (tot == 0.0) tot = 0.001;
      teq = 1.0 - e2 / ((double)(2 * n) * tot);
      const double dn = fr > 1.0 ? fr : 1.0;
      tsust = np_sum_lanes(ftr / dn, n) / (double)n;
    }
    if (lane == 0) {
      mf[CE_MF_EQUALITY] = equality;
      mf[CE_MF_SUSTAINABILITY] = sust;
      mf[CE_MF_TRANSFER_EQUALITY] = teq;
      mf[CE_MF_TRANSFER_SUSTAINABILITY] = tsust;
    }
    __threadfence_block();
    for (u32 k = lane; k < nmi; k += 64) p.final_int_metrics[(size_t)E.e * nmi + k] = mi[k];
    for (u32 k = lane; k < nmf; k += 64) p.final_f64_metrics[(size_t)E.e * nmf + k] = mf[k];
    if (p.flags & CE_FLAG_AUTO_RESET) {
      __threadfence_block();
      u32 orient;
      feat_np_draws(E, p, true, orient, theta);  // the step's working set is dead by now: the block is free to borrow
      feat_base_map(E);
      wave_sync();
      feat_reset_env(E, p, orient);
      for (u32 k = lane; k < nmi; k += 64) p.int_metrics[(size_t)E.e * nmi + k] = 0;
      for (u32 k = lane; k < nmf; k += 64) p.f64_metrics[(size_t)E.e * nmf + k] = 0.0;
      t = 0;
      did_reset = true;
    }
  }
  if (lane == 0) out.done()[E.e] = done ? 1 : 0;
  if (FUSED) return;  // the state stays in registers / LDS for the next step of the launch
  feat_store(E, p, GK == CE_KIND_CLEANUP);
  if (lane == 0) {
    p.timestep[E.e] = (i32)t;
    if (did_reset) p.theta[E.e] = theta;
  }
}

// One single-step launch's work for env `e` on the whole wave.  NFIX: the number of agents as a compile-time constant
// (0 = p.n), as for k_grid_step; the instance is for n = 2 (BASELINE config 0)
template <int GK, int NFIX, class OUT>
DEVINL void feat_step_one(const GridParams& p, const uint8_t* __restrict__ call_actions, FeatLds<GK>* lds, u32 e, const OUT& out) {
  FEnv<GK> E;
  E.lane = lane_id();
  E.e = rfl(e);
  E.n = NFIX ? (u32)NFIX : p.n;
  E.is_agent = E.lane < E.n;
  E.L = lds;
  const u32 ACT = E.is_agent ? (u32)GAT((CE_GPTR(const uint8_t))call_actions + (size_t)E.e * E.n, E.lane) : 4u;
  if (ballot(E.is_agent && ACT > (GK == CE_KIND_HARVEST ? 7u : 8u)) != 0) {  // validated before anything is loaded or written
    if (E.lane == 0) p.error_flags[E.e] |= CE_FAULT_BAD_ACTION;
    return;
  }
  feat_load<GK, GK == CE_KIND_HARVEST>(E, p, true);  // HarvestFeatures: stream window instead of the whole generator
  u32 t = (u32)p.timestep[E.e], fault = 0;
  double theta = p.theta[E.e];
  bool did_reset = false;
  feat_step_core<GK, false>(E, p, out, ACT, t, theta, fault, did_reset);
}
template <int GK, int NFIX> __global__ __launch_bounds__(64, 8) void k_feat_step(const GridParams* __restrict__ pp, const uint8_t* __restrict__ call_actions,
                                                                   const uint8_t* __restrict__ call_mask, u32 env_first, u32 env_end) {
  __shared__ FeatLds<GK> lds;
  const u32 e = env_first + blockIdx.x;
  if (e >= env_end) return;
  feat_step_one<GK, NFIX>(*pp, call_actions, &lds, e, StepOutDirect{*pp});
}

// ----------------------------------------------------------------------------------------
// HarvestFeatures with two agents (BASELINE config 0), FOUR envs per wave.  One wave per env leaves 62 of 64 lanes idle in
// everything that is per agent or per env (moves, consume, metrics, transfers, stores): the kernel above is bound by the
// vector instructions a wave issues, not by bytes.  Here a row of 16 lanes (a DPP row) is one env: lane sl of a row owns the
// ten consecutive apple cells 10 sl .. 10 sl + 9 (row-major list order = lane-major order), per-env values live replicated in
// the row's lanes, counts and ranks come from row scans / the row's 16 bits of a ballot, broadcasts from ds_swizzle.
// Draws that run over the end of the CPython generator's 624 words (one step in ~30 per env) get the twist from the whole
// wave: the row's key words pass through LDS, the row takes its words from both generations, the new key goes back to HBM.
// The draws of the four rows share the wave's 64 lanes, row after row in list order, in as many rounds as it takes.
// A step that is not the common case otherwise — a bad action id, the horizon (metrics of the episode + reset), a second row
// of the same wave crossing the generation end in the same step — is left untouched by the packed pass and taken afterwards
// by the whole wave through the one-env code above, env by env (a launch lasts as long as its slowest wave: this has to
// stay rare).
// Same results as k_feat_step<HARVEST, 2> field for field (tests/test_feature_kinds_gpu.py runs both).
// ----------------------------------------------------------------------------------------
namespace quad {
typedef Geo<CE_KIND_HARVEST> G;
constexpr u32 kCellsPerLane = 10;
// list stamps are kept shifted left by 8: distance << 24 | stamp << 8 | cell index is the key of the closest-apple search.
// "No apple" also carries bit 31, which puts such a cell behind every present one in that search whatever its distance.
constexpr u32 kAbs8 = kAbsent << 8 | 0x80000000u;
static_assert(16 * kCellsPerLane >= (u32)G::NAPPLE && 16 * kCellsPerLane == CE_FEAT_APPLE_SLOTS, "a row covers the list slots");

// The presence maps keep only what a step can touch — padded rows 5 .. 24 (the playable rows plus two on either side), as the
// bytes [kMapLo, kMapHi) of the padded image — and are addressed through a pointer moved back by kMapLo, so that padded indices
// work unchanged.  kDump (row 5, column 0: never read) takes the stores of lanes that have nothing to write.
constexpr u32 kMapLo = 256, kMapHi = 1312, kDump = 260;
static_assert(kMapLo % 16 == 0 && kMapHi % 16 == 0 && kMapLo <= 5 * G::PW && kMapHi >= 25 * G::PW && kMapHi <= (u32)G::PQUADS * 16, "rows 5 .. 24");
struct alignas(16) Lds {
  uint8_t pm[4][kMapHi - kMapLo];  // presence map of each row's env
  uint16_t L[4][CE_FEAT_APPLE_SLOTS];  // compacted cells to draw for (padded index), then the cells that got an apple
  u32 tw[kMtN];                   // one CPython generator at a time, for a row whose draws run over the generation end
  uint16_t cell_pad[CE_FEAT_APPLE_SLOTS], cell_rc[CE_FEAT_APPLE_SLOTS];  // the static list of apple cells: padded index, col | row << 8
  uint8_t owner[kMapHi - kMapLo];  // padded cell -> its index in that list (0xff: not an apple cell): which lane and slot own a cell
};
static_assert(sizeof(Lds) <= 10240, "sixteen waves per CU");
DEVINL void put(uint8_t* pm, bool on, u32 idx, u32 val) { pm[on ? idx : kDump] = (uint8_t)(on ? val : 0u); }

template <int K> DEVINL u32 row_bcast(u32 v) { return (u32)__builtin_amdgcn_ds_swizzle((int)v, 0x10 | (K << 5)); }  // lane K of the row
template <int CTRL> DEVINL u32 dpp_zero(u32 v) { return (u32)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true); }
template <int CTRL> DEVINL u32 dpp_keep(u32 v) { return (u32)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xf, 0xf, false); }
DEVINL u32 row_scan_incl(u32 v) {  // inclusive prefix sum over the row's lanes
  v += dpp_zero<0x111>(v);
  v += dpp_zero<0x112>(v);
  v += dpp_zero<0x114>(v);
  v += dpp_zero<0x118>(v);
  return v;
}
DEVINL u32 row_min_all(u32 v) {  // butterfly over row rotations: every lane ends with the row's minimum
  u32 t = dpp_keep<0x128>(v);
  v = t < v ? t : v;
  t = dpp_keep<0x124>(v);
  v = t < v ? t : v;
  t = dpp_keep<0x122>(v);
  v = t < v ? t : v;
  t = dpp_keep<0x121>(v);
  v = t < v ? t : v;
  return v;
}
// the row's 16 bits of a ballot
DEVINL u32 row_bits(u64 b, u32 sub) {
  const u32 w = (sub & 2u) ? (u32)(b >> 32) : (u32)b;
  return (sub & 1u) ? w >> 16 : w & 0xffffu;
}
DEVINL u32 popc32(u32 x) { return (u32)__builtin_popcount(x); }
DEVINL void add_i64(CE_GPTR(int64_t) a, long long v) { (void)__hip_atomic_fetch_add((CE_GPTR(long long))a, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DEVINL void add_f64(CE_GPTR(double) a, double v) { (void)__builtin_amdgcn_global_atomic_fadd_f64(a, v); }

// the i-th of the 21 offsets (j, k) with j^2 + k^2 <= 5, row-major, as a padded-index delta (what GridTables::close_off holds:
// harvest_new.py:326-336; rows of 3, 5, 5, 5, 3 cells)
DEVINL i32 close_offset(u32 i) {
  const i32 j = i < 3u ? -2 : i < 8u ? -1 : i < 13u ? 0 : i < 18u ? 1 : 2;
  const i32 first = i < 3u ? 0 : i < 8u ? 3 : i < 13u ? 8 : i < 18u ? 13 : 18;
  const i32 k = (i32)i - first - (i < 3u || i >= 18u ? 1 : 2);
  return j * (i32)G::PW + k;
}
// apples within j^2 + k^2 <= 5 of padded cell `c` (row-uniform) in the row's map: 21 offsets over 16 lanes, two rounds
// (co0 / co1: this lane's offsets, close_offset(sl) and close_offset(16 + sl) — the latter only for sl < 5)
DEVINL u32 close_count(const uint8_t* pm, i32 co0, i32 co1, u32 c, u32 sl, u32 sub) {
  const bool v0 = pm[(i32)c + co0] == CE_CELL_APPLE;
  const bool v1 = sl < 5u && pm[(i32)c + co1] == CE_CELL_APPLE;
  return popc32(row_bits(ballot(v0), sub)) + popc32(row_bits(ballot(v1), sub));
}
}  // namespace quad

#if defined(CE_DIAGNOSTIC) && defined(CE_PHASE_STAMPS)  // tools/quad_profile.py: where a wave of the packed kernel spends its life
#define CE_QSTAMP(k)                                                               \
  do {                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                             \
    __builtin_amdgcn_s_waitcnt(0);                                                 \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime();                    \
    __builtin_amdgcn_sched_barrier(0);                                             \
    if (C.lane == 0 && C.pp->debug) C.pp->debug[(size_t)C.e_row0 * 16 + (k)] = t_;     \
  } while (0)
#elif defined(CE_DIAGNOSTIC) && defined(CE_ISA_MARKERS)  // static-ISA reading aid (never run): a comment line at every phase boundary
#define CE_QSTAMP(k) asm volatile("; ==== Q_PHASE " #k)
#else
#define CE_QSTAMP(k) ((void)0)
#endif
namespace quad {
union QuadLds {
  Lds q;
  FeatLds<CE_KIND_HARVEST> one;  // the one-env code borrows the block while the packed state is in HBM
};
// what a row's lanes keep in registers for their env (per-env values replicated over the row's 16 lanes)
struct Row {
  u32 AS8[kCellsPerLane];                       // list stamp << 8 of this lane's ten cells (kAbs8 = no apple)
  u32 P0, P1, O0, O1;                           // the two agents: padded cell, orientation
  u32 next_a, pos, t;                           // next list stamp, CPython stream position, timestep
  double theta;
  bool dirty;                                   // the list changed since it was last in HBM
};
struct Ctx {
  const GridParams* pp;  // (a pointer: a fused rollout re-reads the block through an opaque copy every step)
  QuadLds* L;
  u32 lane, sl, sub, below, e_row0, env_end, e, q2;
  bool live, lastl;  // lastl: lane 15 owns cells 150 .. 159, the last five are past the list end
  uint8_t* pm;       // the row's presence map, indexed by padded cell
  const uint16_t* pad;  // this lane's ten cells in the static list (LDS): padded index ...
  const uint16_t* rc;   // ... and col | row << 8
  const uint8_t* own;   // owner table, indexed by padded cell
  i32 co0, co1;         // this lane's two offsets of the 21-cell neighbourhood (close_count)
};
// the stream words a step starts with: doubles sl and sl + 16 after the row's position (clamped inside the generation),
// and — for ONE row of the wave that is within 48 words of the generation end — the whole key, to be parked in LDS
struct Window {
  u32 wa0, wb0, wa1, wb1;
  u32 tw_row;
  uint4 pf0, pf1, pf2;
};
#define CE_QVALID(C_, r) (!((C_).lastl && (r) >= (u32)G::NAPPLE - 150u))

DEVINL void derive(Ctx& C, u32 env_end);
DEVINL void begin(Ctx& C, u32 env_first, u32 env_end, u32 block) {
  C.lane = lane_id();
  C.e_row0 = env_first + 4u * block;
  derive(C, env_end);
  C.co0 = close_offset(C.sl);
  C.co1 = close_offset(16u + (C.sl < 5u ? C.sl : 0u));
}
// Everything derived from the lane id / env index (address offsets, masks, LDS pointers).  A fused rollout calls this at the
// top of every step on an opaque copy of the lane id: hoisted out of the step loop these values would live in registers
// across the whole body (the compiler then spills a hundred of them), recomputed they cost a handful of instructions.
DEVINL void rederive(Ctx& C, u32 env_end) {
  asm volatile("" : "+v"(C.lane));
  C.e_row0 = opaque_u32(C.e_row0);
  derive(C, env_end);
}
DEVINL void derive(Ctx& C, u32 env_end) {
  C.env_end = env_end;
  C.sl = C.lane & 15u;
  C.sub = C.lane >> 4;
  C.below = (1u << C.sl) - 1u;
  C.live = C.e_row0 + C.sub < env_end;
  C.e = C.live ? C.e_row0 + C.sub : env_end - 1u;  // rows past the end shadow the last env: they load, compute and store nothing
  C.q2 = min(C.lane + 128u, (u32)kMtN / 4u - 1u);
  C.lastl = C.sl == 15u;
  C.pm = C.L->q.pm[C.sub] - kMapLo;
  C.pad = C.L->q.cell_pad + kCellsPerLane * C.sl;
  C.rc = C.L->q.cell_rc + kCellsPerLane * C.sl;
  C.own = C.L->q.owner - kMapLo;
}
// the static list of apple cells into LDS, once per wave (the slots past the list end aim at the dump byte and are never valid)
DEVINL void load_static(const Ctx& C) {
  const GridTables& T = c_tab[CE_KIND_HARVEST];
  constexpr u32 kOwnQuads = (kMapHi - kMapLo) / 16u;
  uint4* o128 = (uint4*)C.L->q.owner;
#pragma unroll
  for (u32 k = 0; k < (kOwnQuads + 63u) / 64u; ++k) o128[min(C.lane + 64u * k, kOwnQuads - 1u)] = uint4{~0u, ~0u, ~0u, ~0u};
  wave_sync();
#pragma unroll
  for (u32 k = 0; k < (CE_FEAT_APPLE_SLOTS + 63u) / 64u; ++k) {
    const u32 idx = C.lane + 64u * k;
    const u32 v = T.apple[idx < (u32)G::NAPPLE ? idx : 0u];
    if (idx < CE_FEAT_APPLE_SLOTS) {
      C.L->q.cell_pad[idx] = (uint16_t)(idx < (u32)G::NAPPLE ? cell_pad(v) : kDump);
      C.L->q.cell_rc[idx] = (uint16_t)cell_rc(v);
    }
    if (idx < (u32)G::NAPPLE) C.L->q.owner[cell_pad(v) - kMapLo] = (uint8_t)idx;
  }
}
// list stamp of the cell with list index `idx` (row-uniform; 0xff = none) := v, in the registers of the lane that owns it
DEVINL void set_stamp(Row& R, const Ctx& C, u32 idx, u32 v) {
  const u32 d = idx - kCellsPerLane * C.sl;  // this lane's slot, if below ten (anything else — 0xff included — wraps far above)
#pragma unroll
  for (u32 r = 0; r < kCellsPerLane; ++r) R.AS8[r] = d == r ? v : R.AS8[r];
}
DEVINL u32 rng_base(const Ctx& C) { return C.e * CE_RNG_WORDS_SELFDRIVE + CE_RNG_WORDS_GRID; }
DEVINL u32 state_base(const Ctx& C) { return C.e * (CE_FEAT_STATE_BYTES / 4u); }
DEVINL void fetch_window(Window& W, const Row& R, const Ctx& C) {
  const GridParams& p = *C.pp;
  const u32 wbase = rng_base(C);
  W.tw_row = 0xffffffffu;
  W.pf0 = W.pf1 = W.pf2 = uint4{};
  // (the row furthest along its generation: the one most likely to cross its end in this step)
  const u32 p0 = rdl(R.pos, 0), p1 = C.e_row0 + 1u < C.env_end ? rdl(R.pos, 16) : 0u, p2 = C.e_row0 + 2u < C.env_end ? rdl(R.pos, 32) : 0u,
            p3 = C.e_row0 + 3u < C.env_end ? rdl(R.pos, 48) : 0u;
  const u32 m01 = p1 > p0 ? p1 : p0, m23 = p3 > p2 ? p3 : p2;
  const u32 far_row = m23 > m01 ? (p3 > p2 ? 3u : 2u) : (p1 > p0 ? 1u : 0u);
  if ((m23 > m01 ? m23 : m01) + 48u > (u32)kMtN) {
    W.tw_row = far_row;
    const auto key4 = (CE_GPTR(const uint4))(p.rng + (size_t)(C.e_row0 + W.tw_row) * CE_RNG_WORDS_SELFDRIVE + CE_RNG_WORDS_GRID);
    W.pf0 = key4[C.lane];
    W.pf1 = key4[C.lane + 64u];
    W.pf2 = key4[C.q2];
  }
  const u32 q0w = min(R.pos + 2u * C.sl, (u32)kMtN - 2u), q1w = min(R.pos + 2u * C.sl + 32u, (u32)kMtN - 2u);
  W.wa0 = GAT(p.rng, wbase + q0w);
  W.wb0 = GAT(p.rng, wbase + q0w + 1u);
  W.wa1 = GAT(p.rng, wbase + q1w);
  W.wb1 = GAT(p.rng, wbase + q1w + 1u);
}
// HBM -> registers: clocks, stream position, the two agents, the list stamps (20 bytes per lane)
DEVINL void load_row(Row& R, const Ctx& C) {
  const GridParams& p = *C.pp;
  const u32 e = C.e;
  R.t = (u32)GAT(p.timestep, e);
  R.theta = GAT(p.theta, e);
  R.pos = GAT(p.rng, rng_base(C) + (u32)kMtN);
  const auto st32 = (CE_GPTR(u32))p.grid;
  const u32 stw = state_base(C);
  u32 sd[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) sd[k] = GAT(st32, stw + 5u * C.sl + (u32)k);
  R.next_a = GAT(st32, stw + (CE_FEAT_APPLE_SLOTS + CE_FEAT_WASTE_SLOTS) / 2u);
  const u32 ag0 = GAT((CE_GPTR(const u32))p.agents, 2u * e), ag1 = GAT((CE_GPTR(const u32))p.agents, 2u * e + 1u);
#pragma unroll
  for (u32 r = 0; r < kCellsPerLane; ++r) {
    const u32 v = (r & 1u) ? (sd[r >> 1] >> 8) & 0xffff00u : (sd[r >> 1] << 8) & 0xffff00u;
    R.AS8[r] = CE_QVALID(C, r) && v != kAbsent << 8 ? v : kAbs8;
  }
  R.P0 = pad_of<CE_KIND_HARVEST>(ag0 & 0xffu, (ag0 >> 8) & 0xffu);
  R.O0 = (ag0 >> 16) & 3u;
  R.P1 = pad_of<CE_KIND_HARVEST>(ag1 & 0xffu, (ag1 >> 8) & 0xffu);
  R.O1 = (ag1 >> 16) & 3u;
  R.dirty = false;
}
// the row's presence map: the reset-time map (every apple present) minus the absent cells
DEVINL void build_map(const Row& R, const Ctx& C) {
  const GridTables& T = c_tab[CE_KIND_HARVEST];
  const uint4* bsrc = (const uint4*)(T.base_pmap + kMapLo);
  uint4* pm128 = (uint4*)C.L->q.pm[C.sub];
  constexpr u32 kQuads = (kMapHi - kMapLo) / 16u;
  wave_sync();
#pragma unroll
  for (u32 k = 0; k < (kQuads + 15u) / 16u; ++k) {  // 16-byte copies; the last round's idle lanes repeat the last quad
    const u32 qd = min(C.sl + 16u * k, kQuads - 1u);
    pm128[qd] = bsrc[qd];
  }
  wave_sync();
#pragma unroll
  for (u32 r = 0; r < kCellsPerLane; ++r) C.pm[R.AS8[r] == kAbs8 ? (u32)C.pad[r] : kDump] = CE_CELL_EMPTY;
  wave_sync();
}
// registers -> HBM for the rows in `rows`: what the next launch (or the one-env code) reads
DEVINL void flush_row(Row& R, const Ctx& C, bool rows) {
  const GridParams& p = *C.pp;
  const u32 e = C.e;
  const auto st32 = (CE_GPTR(u32))p.grid;
  const u32 stw = state_base(C);
  if (rows && C.sl < 2u) {
    const u32 Pm = C.sl == 0u ? R.P0 : R.P1, Om = C.sl == 0u ? R.O0 : R.O1;
    GAT((CE_GPTR(u32))p.agents, 2u * e + C.sl) = row_of<CE_KIND_HARVEST>(Pm) | col_of<CE_KIND_HARVEST>(Pm) << 8 | Om << 16;
  }
  if (rows && C.sl == 0u) {
    GAT(p.timestep, e) = (i32)R.t;
    GAT(p.rng, rng_base(C) + (u32)kMtN) = R.pos;
    if (R.dirty) GAT(st32, stw + (CE_FEAT_APPLE_SLOTS + CE_FEAT_WASTE_SLOTS) / 2u) = R.next_a;
  }
  if (rows && R.dirty) {
#pragma unroll
    for (int k = 0; k < 5; ++k) GAT(st32, stw + 5u * C.sl + (u32)k) = ((R.AS8[2 * k] >> 8) & 0xffffu) | ((R.AS8[2 * k + 1] << 8) & 0xffff0000u);
  }
  R.dirty = R.dirty && !rows;
}

// One step of every row that takes the packed path (returned per row; the others are left exactly as they were).
// OUT: where the step's outputs go (the handle's buffers or a plane of a fused rollout).
template <class OUT> DEVINL bool step(Row& R, const Ctx& C, const OUT& out, u32 act2, Window& W) {
  const GridParams& p = *C.pp;
  const GridTables& T = c_tab[CE_KIND_HARVEST];
  const u32 lane = C.lane, sl = C.sl, sub = C.sub, e = C.e, q2 = C.q2;
  uint8_t* pm = C.pm;
  // a bad action id, or the horizon (episode metrics + reset): not taken here
  const bool simple = C.live && (act2 & 0xffu) <= 7u && (act2 >> 8) <= 7u && R.t + 1u != p.horizon;
  const u32 ACT0 = simple ? act2 & 0xffu : 4u, ACT1 = simple ? act2 >> 8 : 4u;  // (a row that sits the step out: two agents that stay)
  const u32 t_old = R.t;

  CE_QSTAMP(2);
  // ---- move_squares in dict order: stayers first, then the movers by key; a mover is refused by a wall or a claimed square ----
  const bool m0 = ACT0 < 4u, m1 = ACT1 < 4u;
  const i32 d0 = ACT0 == 0 ? -1 : ACT0 == 1 ? 1 : ACT0 == 2 ? -(i32)G::PW : (i32)G::PW;
  const i32 d1 = ACT1 == 0 ? -1 : ACT1 == 1 ? 1 : ACT1 == 2 ? -(i32)G::PW : (i32)G::PW;
  const u32 tg0 = m0 ? (u32)((i32)R.P0 + d0) : R.P0, tg1 = m1 ? (u32)((i32)R.P1 + d1) : R.P1;
  const bool w0 = pm[tg0] == CE_CELL_WALL, w1 = pm[tg1] == CE_CELL_WALL;
  const u32 P0 = (m0 && !w0 && !(!m1 && tg0 == R.P1)) ? tg0 : R.P0;
  const u32 P1 = (m1 && !w1 && tg1 != P0) ? tg1 : R.P1;
  R.P0 = P0;
  R.P1 = P1;

  // ---- consume in move_squares order (a1 eats first only when it stays and a0 moves) ----
  const bool swap = m0 && !m1;
  const u32 pf = swap ? P1 : P0, ps = swap ? P0 : P1;
  u32 eat_f = 0, eat_s = 0, ecl_f = 0, ecl_s = 0;
  {
    const bool ef = pm[pf] == CE_CELL_APPLE;
    if (ballot(ef) != 0) {
      const u32 close = close_count(pm, C.co0, C.co1, pf, sl, sub);  // counted before the apple is removed
      eat_f = ef ? 1u : 0u;
      ecl_f = ef && close < 4u ? 1u : 0u;
      wave_sync();
      put(pm, ef && sl == 0u, pf, CE_CELL_EMPTY);
      set_stamp(R, C, ef ? (u32)C.own[pf] : 0xffu, kAbs8);
      R.dirty = R.dirty || ef;
      wave_sync();
    }
    const bool es = pm[ps] == CE_CELL_APPLE;
    if (ballot(es) != 0) {
      const u32 close = close_count(pm, C.co0, C.co1, ps, sl, sub);
      eat_s = es ? 1u : 0u;
      ecl_s = es && close < 4u ? 1u : 0u;
      wave_sync();
      put(pm, es && sl == 0u, ps, CE_CELL_EMPTY);
      set_stamp(R, C, es ? (u32)C.own[ps] : 0xffu, kAbs8);
      R.dirty = R.dirty || es;
      wave_sync();
    }
  }
  const u32 eat0 = swap ? eat_s : eat_f, eat1 = swap ? eat_f : eat_s;
  const u32 ecl0 = swap ? ecl_s : ecl_f, ecl1 = swap ? ecl_f : ecl_s;
  // ---- rotations ----
  R.O0 = (R.O0 + (ACT0 == 5u ? 1u : ACT0 == 6u ? 3u : 0u)) & 3u;
  R.O1 = (R.O1 + (ACT1 == 5u ? 1u : ACT1 == 6u ? 3u : 0u)) & 3u;

  CE_QSTAMP(3);
  if (W.tw_row != 0xffffffffu) {  // the key requested with the window: parked now that the map work no longer waits behind it
    uint4* tw4 = (uint4*)C.L->q.tw;
    tw4[lane] = W.pf0;
    tw4[lane + 64u] = W.pf1;
    tw4[q2] = W.pf2;
  }
  // ---- spawn_apples: one random.random() per absent cell no agent stands on, in list order ----
  // per lane: bit r of absmask = this lane's cell r has no apple; the cells the agents stand on come from the owner table
  u32 absmask = 0;
#pragma unroll
  for (u32 r = 0; r < kCellsPerLane; ++r) absmask |= R.AS8[r] == kAbs8 ? 1u << r : 0u;
  absmask &= C.lastl ? (1u << ((u32)G::NAPPLE - 150u)) - 1u : (1u << kCellsPerLane) - 1u;  // (lane 15: the slots past the list end)
  const u32 s0 = (u32)C.own[P0] - kCellsPerLane * sl, s1 = (u32)C.own[P1] - kCellsPerLane * sl;
  const u32 under = (s0 < kCellsPerLane ? 1u << s0 : 0u) | (s1 < kCellsPerLane ? 1u << s1 : 0u);
  const u32 elmask = simple ? absmask & ~under : 0u;
  const u32 cnt = popc32(elmask), nabs = popc32(absmask);
  const u32 incl = row_scan_incl(cnt | nabs << 16);  // both counts ride one scan
  const u32 tot = row_bcast<15>(incl);
  const u32 nelig = tot & 0xffffu;
  u32 napples = (u32)G::NAPPLE - (tot >> 16);
  // Draws that cross the generation end get the twist from the whole wave, the row's key passing through the LDS block.  The
  // block serves one row at a time: the first pass below takes every row that does not twist plus the first one that does,
  // each further twisting row of the wave (rare) gets a pass of its own.
  const bool over = simple && R.pos + 2u * nelig > (u32)kMtN;
#if defined(CE_QUAD_ABLATE) && CE_QUAD_ABLATE >= 1  // timing probes (wrong results): 1 = no twist, no one-env rows
  u64 tw_pending = 0;
#else
  u64 tw_pending = ballot(over && sl == 0u);
#endif
  CE_QSTAMP(4);
  for (bool first_pass = true;; first_pass = false) {
    const u32 tws = tw_pending ? ctz64(tw_pending) >> 4 : 0xffffffffu;
    const bool istw = over && sub == tws;
    const bool inpass = first_pass ? simple && (!over || istw) : istw;
    if (tw_pending != 0) {
      if (tws != W.tw_row) {  // (not the row whose key was parked in LDS ahead of time)
        const auto key4 = (CE_GPTR(const uint4))(p.rng + (size_t)(C.e_row0 + tws) * CE_RNG_WORDS_SELFDRIVE + CE_RNG_WORDS_GRID);
        uint4* tw4 = (uint4*)C.L->q.tw;
        wave_sync();
        const uint4 r0 = key4[lane], r1 = key4[lane + 64u], r2 = key4[q2];
        tw4[lane] = r0;
        tw4[lane + 64u] = r1;
        tw4[q2] = r2;
      }
      W.tw_row = 0xffffffffu;  // the block holds a twisted key after this pass
      wave_sync();
    }
    CE_QSTAMP(5);
#if defined(CE_QUAD_ABLATE) && CE_QUAD_ABLATE >= 2
    if (false) {
#else
    if (ballot(inpass && nelig != 0u) != 0) {
#endif
      // the cells to draw for, compacted in list order (the q-th one takes double q of the stream)
      uint16_t* L = C.L->q.L[sub];
      {
        const u32 rk0 = (incl & 0xffffu) - cnt;  // list rank of this lane's first eligible cell
        u32 padv[kCellsPerLane];
#pragma unroll
        for (u32 r = 0; r < kCellsPerLane; ++r) padv[r] = C.pad[r];
#pragma unroll
        for (u32 r = 0; r < kCellsPerLane; ++r) {  // (entry 159 is past every rank: it takes the stores of the other cells)
          const bool e = inpass && (elmask >> r & 1u) != 0u;
          L[e ? rk0 + popc32(elmask & ((1u << r) - 1u)) : CE_FEAT_APPLE_SLOTS - 1u] = (uint16_t)padv[r];
        }
      }
      wave_sync();
      const u64 th0 = T.apple_thresh[0], th1 = T.apple_thresh[1], th2 = T.apple_thresh[2], th3 = T.apple_thresh[3];
      const u32* tw = C.L->q.tw;
      // The draws of the four rows are dealt to the wave's 64 lanes together, row after row: slot = (draws of the rows before)
      // + q.  Four rows of 12-18 draws are one round of 64 slots where a round per row of 16 lanes would be two.
      const u32 ne = inpass ? nelig : 0u;
      const u32 n0 = rdl(ne, 0), n1 = rdl(ne, 16), n2 = rdl(ne, 32), n3 = rdl(ne, 48);
      const u32 b1 = n0, b2 = b1 + n1, b3 = b2 + n2, total = b3 + n3;
      const u32 pos0 = rdl(R.pos, 0), pos1 = rdl(R.pos, 16), pos2 = rdl(R.pos, 32), pos3 = rdl(R.pos, 48);
      const u32 pos_tw = tws == 0u ? pos0 : tws == 1u ? pos1 : tws == 2u ? pos2 : pos3;
      u32 s0 = 0, s1 = 0, s2 = 0, s3 = 0;  // apples of this pass so far, per row
      bool twisted = false;
      const u64 lt = (1ull << lane) - 1ull;
      for (u32 c = 0; 64u * c < total; ++c) {
        const u32 slot = 64u * c + lane;
        const bool act = slot < total;
        const u32 rho = (slot >= b1 ? 1u : 0u) + (slot >= b2 ? 1u : 0u) + (slot >= b3 ? 1u : 0u);  // the row this slot draws for
        const u32 q = slot - (rho == 0u ? 0u : rho == 1u ? b1 : rho == 2u ? b2 : b3);            // ... and which of its draws
        // the two words of double q of that row's stream: from the lane of the row that holds them (the window: q < 32) ...
        const u32 holder = rho << 4 | (q & 15u);
        const u32 la = bperm(W.wa0, holder), lb = bperm(W.wb0, holder), ha = bperm(W.wa1, holder), hb = bperm(W.wb1, holder);
        u32 wa = (q & 16u) ? ha : la, wb = (q & 16u) ? hb : lb;
        const u32 posr = rho == 0u ? pos0 : rho == 1u ? pos1 : rho == 2u ? pos2 : pos3;
        if (ballot(act && q >= 32u) != 0) {  // ... or, past the window, from the row's key in HBM
          const u32 i = min(posr + 2u * q, (u32)kMtN - 2u);
          const u32 kb = (C.e_row0 + rho) * CE_RNG_WORDS_SELFDRIVE + CE_RNG_WORDS_GRID;
          if (act && q >= 32u) {
            wa = GAT(p.rng, kb + i);
            wb = GAT(p.rng, kb + i + 1u);
          }
        }
        if (tw_pending != 0) {  // the twisting row: words below 624 from the current key, the rest from the twisted one
          const bool mine = act && rho == tws;
          const u32 w = pos_tw + 2u * q;
          if (!twisted) {
            if (mine && w < (u32)kMtN) wa = tw[w];
            if (mine && w + 1u < (u32)kMtN) wb = tw[w + 1u];
            if (ballot(mine && w + 1u >= (u32)kMtN) != 0) {  // this round passes word 624
              wave_sync();
              mt_twist_inline(C.L->q.tw, lane);
              wave_sync();
              const auto key4 = (CE_GPTR(uint4))(p.rng + (size_t)(C.e_row0 + tws) * CE_RNG_WORDS_SELFDRIVE + CE_RNG_WORDS_GRID);
              const uint4* tw4 = (const uint4*)C.L->q.tw;
              const uint4 r0 = tw4[lane], r1 = tw4[lane + 64u], r2 = tw4[q2];
              key4[lane] = r0;
              key4[lane + 64u] = r1;
              key4[q2] = r2;
              twisted = true;
            }
          }
          if (twisted) {
            if (mine && w >= (u32)kMtN && w < 2u * (u32)kMtN) wa = tw[w - (u32)kMtN];
            if (mine && w + 1u >= (u32)kMtN && w + 1u < 2u * (u32)kMtN) wb = tw[w + 1u - (u32)kMtN];
          }
        }
        uint16_t* Lr = C.L->q.L[0] + rho * CE_FEAT_APPLE_SLOTS;
        uint8_t* pmr = C.L->q.pm[0] + rho * (kMapHi - kMapLo) - kMapLo;
        const u32 cell_q = Lr[act ? q : 0u];
        const i32 cell = (i32)(act ? cell_q : kDump + (u32)G::PW + 1u);  // idle lanes look at a border cell and write nothing
        // r < p on the 53-bit integer form of random.random(): the upper 27 bits (first word) decide unless they tie with the
        // threshold's (once in 2^27 draws); only then is the second word tempered and compared
        const u32 xa = mt_temper(wa) >> 5;
        bool sp = false;
        // the neighbour count of a cell includes apples spawned EARLIER in this very pass (harvest_features.py:139-151): iterate
        // the parallel decision from "none spawned" upwards, as feat_spawn does.  (The cell itself holds no apple: eight reads.)
        for (;;) {
          u32 num = 0;
#pragma unroll
          for (int j = -1; j <= 1; ++j)
#pragma unroll
            for (int k = -1; k <= 1; ++k) {
              if (j == 0 && k == 0) continue;
              const bool earlier = j < 0 || (j == 0 && k < 0);
              const uint8_t x = pmr[cell + j * G::PW + k];
              num += (x == CE_CELL_APPLE || (earlier && x == 0x42)) ? 1u : 0u;
            }
          const u32 thi = num == 0 ? (u32)(th0 >> 26) : num == 1 ? (u32)(th1 >> 26) : num == 2 ? (u32)(th2 >> 26) : (u32)(th3 >> 26);
          bool z = act && xa < thi;
          if (ballot(act && xa == thi) != 0) {
            const u32 tlo = (u32)(num == 0 ? th0 : num == 1 ? th1 : num == 2 ? th2 : th3) & 0x3ffffffu;
            z = act && (xa < thi || (xa == thi && (mt_temper(wb) >> 6) < tlo));
          }
          const bool changed = z != sp;
          sp = z;
          if (ballot(changed) == 0) break;
          wave_sync();
          put(pmr, act, (u32)cell, sp ? 0x42u : (u32)CE_CELL_EMPTY);  // 0x42 = spawned in this pass
          wave_sync();
        }
        const u64 sb = ballot(sp);
        if (sb != 0) {  // each row's list of this pass's apples grows over the entries already consumed (rank <= q)
          const u64 m0 = ballot(act && rho == 0u), m1 = ballot(act && rho == 1u), m2 = ballot(act && rho == 2u), m3 = ballot(act && rho == 3u);
          const u64 mine_row = rho == 0u ? m0 : rho == 1u ? m1 : rho == 2u ? m2 : m3;
          const u32 before = rho == 0u ? s0 : rho == 1u ? s1 : rho == 2u ? s2 : s3;
          wave_sync();
          if (sp) Lr[before + popc64(sb & mine_row & lt)] = (uint16_t)cell;
          wave_sync();
          s0 += popc64(sb & m0);
          s1 += popc64(sb & m1);
          s2 += popc64(sb & m2);
          s3 += popc64(sb & m3);
        }
      }
      const u32 nsp = sub == 0u ? s0 : sub == 1u ? s1 : sub == 2u ? s2 : s3;
      if (ballot(nsp != 0u) != 0) {  // appended in list order: stamps continue the list; the owners of the cells take theirs
        for (u32 k = 0; ballot(k < nsp) != 0; ++k) {
          const bool on = k < nsp;
          const u32 cellk = on ? (u32)L[k] : kDump;
          put(pm, on && sl == 0u, cellk, CE_CELL_APPLE);
          set_stamp(R, C, on ? (u32)C.own[cellk] : 0xffu, (R.next_a + k) << 8);
        }
        wave_sync();
        R.next_a += nsp;
        napples += nsp;
        R.dirty = R.dirty || nsp != 0u;
      }
      R.pos += inpass ? 2u * nelig : 0u;
      R.pos -= R.pos > (u32)kMtN ? (u32)kMtN : 0u;  // (the twist above)
    }
    tw_pending &= tw_pending - 1;
    if (tw_pending == 0) break;
  }

  CE_QSTAMP(6);
  // ---- feature vectors (the observation): closest apple = min over (manhattan distance, list stamp) ----
  const u32 row0 = row_of<CE_KIND_HARVEST>(P0), row1 = row_of<CE_KIND_HARVEST>(P1);
  const u32 col0 = P0 - __umul24(row0 + kView, (u32)G::PW) - kView, col1 = P1 - __umul24(row1 + kView, (u32)G::PW) - kView;
  const u32 prc0 = col0 | row0 << 8, prc1 = col1 | row1 << 8;
  u32 best0 = 0xffffffffu, best1 = 0xffffffffu;
#if !defined(CE_QUAD_ABLATE) || CE_QUAD_ABLATE < 3
#pragma unroll
#else
#pragma unroll
  for (u32 r = 0; r < 1; ++r) best0 = best1 = R.AS8[r];
  if (false)
#endif
  for (u32 r = 0; r < kCellsPerLane; ++r) {
    const u32 kk = R.AS8[r] | sl << 4 | r;  // (low byte: which lane, which of its cells)
    const u32 k0 = __builtin_amdgcn_sad_u8((u32)C.rc[r], prc0, 0u) << 24 | kk, k1 = __builtin_amdgcn_sad_u8((u32)C.rc[r], prc1, 0u) << 24 | kk;
    best0 = k0 < best0 ? k0 : best0;
    best1 = k1 < best1 ? k1 : best1;
  }
  best0 = row_min_all(best0);
  best1 = row_min_all(best1);
  const u32 mine = sl == 0u ? best0 : best1;  // lane a of a row writes agent a's vector
  u32 ca = 0;
  if ((mine >> 24) < 128u) ca = C.L->q.cell_rc[kCellsPerLane * ((mine >> 4) & 15u) + (mine & 15u)];
  const u32 cn0 = close_count(pm, C.co0, C.co1, P0, sl, sub), cn1 = close_count(pm, C.co0, C.co1, P1, sl, sub);

  CE_QSTAMP(7);
  // ---- rewards, infos, transfers (two_stage_train.py:62-121), metrics ----
  const bool me0 = sl == 0u, agent = sl < 2u;
  const u32 rew = me0 ? eat0 : eat1, ecl = me0 ? ecl0 : ecl1, cn = me0 ? cn0 : cn1;
  const bool outl = simple && agent;
  double rw = (double)rew;
  const u32 nmi = CE_MI_COUNT(2), nmf = CE_MF_COUNT(2);
  const auto mi = p.int_metrics;
  const auto mf = p.f64_metrics;
  const u32 mib = e * nmi, mfb = e * nmf;
  if (ballot(simple && (eat0 | eat1) != 0u) != 0) {
    const bool touched = simple && (eat0 | eat1) != 0u;
    // one wave owns these rows for the launch: an atomic add without a return value is the same read-modify-write minus
    // the wait for the read (bitwise the same sums, integer and double alike)
    if (touched && sl < 4u) {  // lane k < 4 of the row holds global metric k
      const u32 add = sl == CE_MI_TOTAL_APPLES_EATEN || sl == CE_MI_RAW_ENV_REWARDS ? eat0 + eat1 : sl == CE_MI_LOW_DENSITY_APPLES ? ecl0 + ecl1 : 0u;
      if (add) add_i64(&GAT(mi, mib + sl), (long long)add);
    }
    if (touched && agent && rew) {
      add_i64(&GAT(mi, mib + CE_MI_AGENT(2, CE_MIA_A, sl)), (long long)rew);
      if (ecl) add_i64(&GAT(mi, mib + CE_MI_AGENT(2, CE_MIA_B, sl)), (long long)ecl);
      add_i64(&GAT(mi, mib + CE_MI_AGENT(2, CE_MIA_SUM_R, sl)), (long long)rew);
      if (t_old) add_i64(&GAT(mi, mib + CE_MI_AGENT(2, CE_MIA_SUM_TR, sl)), (long long)t_old * (long long)rew);
    }
    // (a transfer needs a close apple eaten, a non-zero reward an apple eaten: nothing of the wrapper's can change otherwise)
    if (p.contract != CE_CONTRACT_NONE) {
      double tr0, tr1;
      if (p.contract == CE_CONTRACT_CLEANUP) tr0 = tr1 = 0.0;  // HarvestFeatures cleans nothing
      else {
        tr0 = (cn0 < 4u && ecl0 > 0u) ? R.theta : 0.0;
        tr1 = (cn1 < 4u && ecl1 > 0u) ? R.theta : 0.0;
      }
      // agents with a zero transfer are skipped, in agent order (n = 2: the other agent's share is the whole transfer)
      double total = 0.0;
      if (tr0 != 0.0) {
        rw = me0 ? rw - tr0 : rw + tr0;
        total += tr0;
      }
      if (tr1 != 0.0) {
        rw = me0 ? rw + tr1 : rw - tr1;
        total += tr1;
      }
      if (simple && total != 0.0 && sl == 0u) add_f64(&GAT(mf, mfb + CE_MF_TRANSFERS), total);
      const bool any_rew = row_bits(ballot(agent && rw != 0.0), sub) != 0u;
      if (outl && any_rew) {
        add_f64(&GAT(mf, mfb + CE_MF_AGENT(2, CE_MFA_SUM_R, sl)), rw);
        add_f64(&GAT(mf, mfb + CE_MF_AGENT(2, CE_MFA_SUM_TR, sl)), (double)t_old * rw);
      }
    }
  }
  CE_QSTAMP(8);
  // ---- the step's outputs (rows that took the packed step) ----
  if (outl) {
    GAT(out.base_reward(), 2u * e + sl) = (i32)rew;
    GAT(out.reward(), 2u * e + sl) = rw;
    GAT((CE_GPTR(uint16_t))out.info(), 2u * e + sl) = (uint16_t)(rew | ecl << 8);  // info[a][0..1] as one short
    // own position / orientation, then the other agent's (compute_closest_pos: a0 -> a1, a1 -> a0), the closest apple, counts
    const u32 rowm = me0 ? row0 : row1, colm = me0 ? col0 : col1, Om = me0 ? R.O0 : R.O1;
    const u32 rowc = me0 ? row1 : row0, colc = me0 ? col1 : col0, Oc = me0 ? R.O1 : R.O0;
    typedef u32 u32x4 __attribute__((ext_vector_type(4)));
    typedef u32 u32x2 __attribute__((ext_vector_type(2)));
    const auto f32 = (CE_GPTR(u32))out.features();
    const u32 fb = (2u * e + sl) * 7u;  // 14 int16 per agent: 28-byte rows, dword aligned
    *(CE_GPTR(u32x4))&GAT(f32, fb) = u32x4{rowm | colm << 16, Om | rowc << 16, colc | Oc << 16, (ca >> 8) | (ca & 0xffu) << 16};
    *(CE_GPTR(u32x2))&GAT(f32, fb + 4u) = u32x2{cn | napples << 16, 0u};
    GAT(f32, fb + 6u) = 0u;
  }
  if (simple && sl == 0u) GAT(out.done(), e) = 0;
  R.t += simple ? 1u : 0u;
  return simple;
}
// the rows in `rest` (bit 16 r = row r), one env at a time on the whole wave through the one-env code: HBM state in, HBM state out
template <class OUT> DEVINL void one_env_rows(const Ctx& C, const OUT& out, const uint8_t* __restrict__ actions, u64 rest) {
  wave_sync();
  for (; rest; rest &= rest - 1) {
    feat_step_one<CE_KIND_HARVEST, 2>(*C.pp, actions, &C.L->one, C.e_row0 + (ctz64(rest) >> 4), out);
    wave_sync();
  }
}
// (a real call: the one-env code inlined into the step loop of the rollout costs the loop a hundred spilled registers; as a
// call it costs a save / restore of the caller's registers on the rare steps that take it)
__device__ __noinline__ void rollout_rest(const GridParams* pp, QuadLds* L, const RolloutArgs* rap, u32 pl, const uint8_t* actions, u32 e_row0,
                                          u64 rest) {
  Ctx C{pp, L};
  C.lane = lane_id();
  C.e_row0 = e_row0;
  one_env_rows(C, StepOutPlane{*rap, pl}, actions, rest);
}
__device__ __noinline__ void step_rest(const GridParams* pp, QuadLds* L, const uint8_t* actions, u32 e_row0, u64 rest) {
  Ctx C{pp, L};
  C.lane = lane_id();
  C.e_row0 = e_row0;
  one_env_rows(C, StepOutDirect{*pp}, actions, rest);
}
}  // namespace quad

__global__ __launch_bounds__(64, 4) void k_feat_step_quad(const GridParams* __restrict__ pp, const uint8_t* __restrict__ call_actions, u32 env_first,
                                                         u32 env_end) {
  using namespace quad;
  __shared__ QuadLds lds;
  if (env_first + 4u * blockIdx.x >= env_end) return;
  Ctx C{pp, &lds};
  begin(C, env_first, env_end, blockIdx.x);
  CE_QSTAMP(0);
  const u32 act2 = (u32)GAT((CE_GPTR(const uint16_t))call_actions, C.e);
  Row R;
  Window W;
  load_row(R, C);
  load_static(C);  // (the table and map loads below do not wait for the stream position the window does)
  CE_QSTAMP(1);
  build_map(R, C);
  fetch_window(W, R, C);
  const bool simple = step(R, C, StepOutDirect{*C.pp}, act2, W);
  flush_row(R, C, simple);
  CE_QSTAMP(9);
  u64 rest = ballot(C.live && !simple && C.sl == 0u);
#if defined(CE_QUAD_ABLATE) && CE_QUAD_ABLATE >= 1
  rest = 0;
#endif
  if (rest != 0) step_rest(C.pp, C.L, call_actions, C.e_row0, rest);
  CE_QSTAMP(10);
}

// The fused rollout of the same packing: the rows' state stays in registers / LDS for the steps of a launch, the stream
// window of step s + 1 is requested as soon as step s knows where the stream stands.  A step with a row that cannot take
// the packed path (horizon, bad action id) parks every row's state in HBM, lets the one-env code take those rows and
// loads everything back.
__global__ __launch_bounds__(64, 4) void k_feat_rollout_quad(const GridParams* __restrict__ pp, const RolloutArgs ra_) {
  using namespace quad;
  static_assert(alignof(RolloutArgs) == 8, "RolloutArgs sits at kernarg offset 8");
  const RolloutArgs* rap = (const RolloutArgs*)((const char*)__builtin_amdgcn_kernarg_segment_ptr() + 8);
  __shared__ QuadLds lds;
  if (rap->env_first + 4u * blockIdx.x >= rap->env_end) return;
  Ctx C{pp, &lds};
  begin(C, rap->env_first, rap->env_end, blockIdx.x);
  Row R;
  Window W;
  load_row(R, C);
  load_static(C);
  build_map(R, C);
  u32 pl = rap->plane0;
  const u32 num_steps = rap->num_steps;
  u32 act2 = (u32)GAT((CE_GPTR(const uint16_t))rap->actions, C.e);
  for (u32 s = 0; s < num_steps; ++s) {
    const RolloutArgs& ra = opaque_block(rap);
    C.pp = &opaque_block(pp);
    rederive(C, ra.env_end);
    const u32 sn = s + 1 < num_steps ? s + 1 : s;  // the next step's actions are in flight while this step runs
    const auto plane = (CE_GPTR(const uint8_t))ra.actions + (size_t)s * ra.action_plane;
    const u32 act2n = (u32)GAT((CE_GPTR(const uint16_t))((CE_GPTR(const uint8_t))ra.actions + (size_t)sn * ra.action_plane), C.e);
    const StepOutPlane out{ra, pl};
    CE_QSTAMP(11);
    fetch_window(W, R, C);
    CE_QSTAMP(12);
    const bool simple = step(R, C, out, act2, W);
    CE_QSTAMP(13);
    const u64 rest = ballot(C.live && !simple && C.sl == 0u);
    if (rest != 0) {
      flush_row(R, C, C.live);
      __threadfence();
      rollout_rest(C.pp, C.L, rap, pl, (const uint8_t*)plane, C.e_row0, rest);
      __threadfence();
      load_row(R, C);
      build_map(R, C);
    }
    act2 = act2n;
    pl = pl + 1 == ra.num_planes ? 0u : pl + 1;
  }
  flush_row(R, C, C.live);
}

// Fused multi-step rollout of the feature-vector envs (ce_rollout_fused): list stamps, agents and the CPython `random`
// stream stay on chip for the steps of a launch (the np.random stream is only touched by resets, straight in HBM)
#ifndef CE_FEAT_ROLLOUT_WAVES
// round 3, tools/ab.sh, two rounds interleaved (HarvestFeatures / CleanupFeatures n = 2 x 16 384 envs, G agent-steps/s):
// 5 waves (96 VGPRs, no scratch) 1.99 / 1.81, 6 waves 2.14 / 1.95, 7 waves (72 VGPRs + 12 / 20 spilled) 2.28 / 2.04, 8 waves 2.24 / 2.01.
// (Round 2 saw a spilling build of this kernel run at two speeds from process to process and chose 5; not reproduced since.)
#define CE_FEAT_ROLLOUT_WAVES 7
#endif
template <int GK, int NFIX> __global__ __launch_bounds__(64, CE_FEAT_ROLLOUT_WAVES) void k_feat_rollout(const GridParams* __restrict__ pp, const RolloutArgs ra_) {
  static_assert(alignof(RolloutArgs) == 8, "RolloutArgs sits at kernarg offset 8");
  const RolloutArgs* rap = (const RolloutArgs*)((const char*)__builtin_amdgcn_kernarg_segment_ptr() + 8);
  __shared__ FeatLds<GK> lds;
  FEnv<GK> E;
  if (!feat_begin(E, *pp, &lds, rap->env_first, rap->env_end)) return;
  if (NFIX) {
    E.n = (u32)NFIX;
    E.is_agent = E.lane < E.n;
  }
  u32 ACT = E.is_agent ? (u32)GAT((CE_GPTR(const uint8_t))rap->actions + (size_t)E.e * E.n, E.lane) : 4u;
  feat_load(E, *pp, true);
  u32 t = rfl((u32)pp->timestep[E.e]);
  double theta = shfl_f64(pp->theta[E.e], 0);
  u32 fault = 0, pl = rap->plane0;
  bool any_reset = false;
  const u32 num_steps = rap->num_steps;
  for (u32 s = 0; s < num_steps; ++s) {
    const GridParams& p = opaque_block(pp);
    const RolloutArgs& ra = opaque_block(rap);
    asm volatile("" : "+v"(E.lane));
    E.n = NFIX ? (u32)NFIX : opaque_u32(E.n);
    E.e = opaque_u32(E.e);
    E.is_agent = E.lane < E.n;
    const u32 sn = s + 1 < num_steps ? s + 1 : s;
    const u32 ACTN = E.is_agent ? (u32)GAT((CE_GPTR(const uint8_t))ra.actions + (size_t)sn * ra.action_plane + (size_t)E.e * E.n, E.lane) : 4u;
    bool did_reset = false;
    feat_step_core<GK, true>(E, p, StepOutPlane{ra, pl}, ACT, t, theta, fault, did_reset);
    ACT = ACTN;
    pl = pl + 1 == ra.num_planes ? 0u : pl + 1;
    rng_assert_uniform(E.py);
    E.py.twists = rfl(E.py.twists);
    E.py_resident = true;
    E.stamps_dirty = false;
    E.next_a = rfl(E.next_a);
    E.next_w = rfl(E.next_w);
    t = rfl(t);
    fault = rfl(fault);
    any_reset = rfl((u32)(any_reset || did_reset)) != 0;
    if (did_reset) theta = shfl_f64(theta, 0);
  }
  const GridParams& p = opaque_block(pp);
  E.stamps_dirty = true;
  feat_store(E, p, GK == CE_KIND_CLEANUP);  // (stamps and agents only: the static tables are not needed)
  if (E.lane == 0) {
    p.timestep[E.e] = (i32)t;
    if (any_reset) p.theta[E.e] = theta;
    if (fault) p.error_flags[E.e] |= fault;
  }
}

// ----------------------------------------------------------------------------------------
// ce_download / ce_upload("grid"): packed presence bits <-> the padded map image (one wave per env)
// ----------------------------------------------------------------------------------------
template <int KIND> __global__ __launch_bounds__(64) void k_grid_expand(const uint8_t* __restrict__ state, uint8_t* __restrict__ image,
                                                                     u32 env_first, u32 env_count, const GridTables* tab, u32 napple, u32 nwaste) {
  typedef Geo<KIND> G;
  const GridTables& T = tab ? *tab : c_tab[KIND];  // a handle built from a caller's layout carries its own tables
  if (blockIdx.x >= env_count) return;
  const u32 lane = lane_id(), e = env_first + blockIdx.x;
  const u32* bits = (const u32*)(state + (size_t)e * kGridStateBytes);
  u32* dst = (u32*)(image + (size_t)blockIdx.x * G::IMAGE_STRIDE);
  const bool blank = (bits[7] >> (kGridBlankBit & 31)) & 1u;
  const u32* base = (const u32*)T.base_pmap;
  for (u32 k = lane; k < (u32)G::IMAGE_STRIDE / 4; k += 64) dst[k] = (blank || k >= (u32)G::PCELLS / 4) ? 0u : base[k];
  if (blank) return;
  __syncthreads();
  uint8_t* img = image + (size_t)blockIdx.x * G::IMAGE_STRIDE;
  for (u32 c = lane; c < napple; c += 64) img[cell_pad(T.apple[c])] = (bits[c >> 5] >> (c & 31)) & 1u ? CE_CELL_APPLE : CE_CELL_EMPTY;
  if (KIND == CE_KIND_CLEANUP)
    for (u32 c = lane; c < nwaste; c += 64) img[cell_pad(T.waste[c])] = (bits[4 + (c >> 5)] >> (c & 31)) & 1u ? CE_CELL_WASTE : CE_CELL_RIVER;
}
// ----------------------------------------------------------------------------------------
// ce_global_view: the whole colour map of an env, agents painted — MapEnv.global_view (map_env.py:394-395:
// world_map_color without its padding), what JointEnv's `global_obs` hands the centralised agent
// (two_stage_train.py:531,572-586, cleanup_new.py:299-300).  world_map_color carries the agents a step painted
// (map_env.py:257-261: in agent order, the later agent wins a shared cell) and none after reset() (map_env.py:306-342
// rebuilds it from the map alone), so an env paints iff its timestep is past 0; beams never enter it.  One wave per env,
// dense uint8 [H][W][3] rows.  Only launched when asked for: rollouts do not pay for it.
// ----------------------------------------------------------------------------------------
template <int KIND> __global__ __launch_bounds__(64) void k_grid_global_view(const uint8_t* __restrict__ state, const uint8_t* __restrict__ agents,
                                                                          const i32* __restrict__ timestep, uint8_t* __restrict__ out,
                                                                          u32 env_first, u32 env_count, u32 n, const GridTables* tab, u32 napple,
                                                                          u32 nwaste, u32 H, u32 W) {
  typedef Geo<KIND> G;
  const GridTables& T = tab ? *tab : c_tab[KIND];
  if (blockIdx.x >= env_count) return;
  __shared__ uint8_t img[G::IMAGE_STRIDE];
  const u32 lane = lane_id(), e = env_first + blockIdx.x;
  const u32* bits = (const u32*)(state + (size_t)e * kGridStateBytes);
  const bool blank = (bits[7] >> (kGridBlankBit & 31)) & 1u;
  const u32* base = (const u32*)T.base_pmap;
  u32* img4 = (u32*)img;
  for (u32 k = lane; k < (u32)G::IMAGE_STRIDE / 4; k += 64) img4[k] = (blank || k >= (u32)G::PCELLS / 4) ? 0u : base[k];
  __syncthreads();
  if (!blank) {
    for (u32 c = lane; c < napple; c += 64) img[cell_pad(T.apple[c])] = (bits[c >> 5] >> (c & 31)) & 1u ? CE_CELL_APPLE : CE_CELL_EMPTY;
    if (KIND == CE_KIND_CLEANUP)
      for (u32 c = lane; c < nwaste; c += 64) img[cell_pad(T.waste[c])] = (bits[4 + (c >> 5)] >> (c & 31)) & 1u ? CE_CELL_WASTE : CE_CELL_RIVER;
  }
  __syncthreads();
  if (lane == 0 && timestep[e] > 0) {  // in agent order: the later agent wins
    const uint8_t* ag = agents + (size_t)e * n * 4;
    for (u32 a = 0; a < n; ++a) img[pad_of<KIND>(ag[4 * a], ag[4 * a + 1])] = (uint8_t)(6 + a);
  }
  __syncthreads();
  uint8_t* dst = out + (size_t)blockIdx.x * H * W * 3;
  for (u32 k = lane; k < H * W; k += 64) {
    const u32 r = k / W, c = k - r * W;
    const u32 rgbv = c_rgb[img[pad_of<KIND>(r, c)] & 15u];
    dst[3 * k] = (uint8_t)rgbv;
    dst[3 * k + 1] = (uint8_t)(rgbv >> 8);
    dst[3 * k + 2] = (uint8_t)(rgbv >> 16);
  }
}
template <int KIND> __global__ __launch_bounds__(64) void k_grid_pack(const uint8_t* __restrict__ image, uint8_t* __restrict__ state,
                                                                   u32* __restrict__ error_flags, u32 env_first, u32 env_count, const GridTables* tab,
                                                                   u32 napple, u32 nwaste) {
  typedef Geo<KIND> G;
  const GridTables& T = tab ? *tab : c_tab[KIND];
  if (blockIdx.x >= env_count) return;
  __shared__ uint8_t chk[G::PCELLS];
  const u32 lane = lane_id(), e = env_first + blockIdx.x;
  const uint8_t* img = image + (size_t)blockIdx.x * G::IMAGE_STRIDE;
  bool nonzero = false;
  for (u32 k = lane; k < (u32)G::PCELLS; k += 64) {
    chk[k] = img[k];
    nonzero = nonzero || img[k] != 0;
  }
  const bool blank = ballot(nonzero) == 0;
  __syncthreads();
  u32 w = 0;
  bool bad = false;
  for (u32 r = 0; r < (napple + 63) / 64; ++r) {
    const u32 c = lane + 64 * r;
    const u32 cell = cell_pad(T.apple[c < napple ? c : 0]);
    const uint8_t v = c < napple ? chk[cell] : (uint8_t)0;
    const u64 m = ballot(c < napple && v == CE_CELL_APPLE);
    bad = bad || (c < napple && v != CE_CELL_APPLE && v != CE_CELL_EMPTY);
    if (lane == 2 * r) w = (u32)m;
    if (lane == 2 * r + 1) w = (u32)(m >> 32);
  }
  if (KIND == CE_KIND_CLEANUP)
    for (u32 r = 0; r < 2; ++r) {
      const u32 c = lane + 64 * r;
      const u32 cell = cell_pad(T.waste[c < nwaste ? c : 0]);
      const uint8_t v = c < nwaste ? chk[cell] : (uint8_t)0;
      const u64 m = ballot(c < nwaste && v == CE_CELL_WASTE);
      bad = bad || (c < nwaste && v != CE_CELL_WASTE && v != CE_CELL_RIVER);
      if (lane == 4 + 2 * r) w = (u32)m;
      if (lane == 5 + 2 * r) w = (u32)(m >> 32);
    }
  __syncthreads();
  // every other cell must be the static map: neutralise the variable cells, then compare with the base image
  for (u32 c = lane; c < napple; c += 64) chk[cell_pad(T.apple[c])] = T.base_pmap[cell_pad(T.apple[c])];
  if (KIND == CE_KIND_CLEANUP)
    for (u32 c = lane; c < nwaste; c += 64) chk[cell_pad(T.waste[c])] = T.base_pmap[cell_pad(T.waste[c])];
  __syncthreads();
  for (u32 k = lane; k < (u32)G::PCELLS; k += 64) bad = bad || chk[k] != T.base_pmap[k];
  if (blank) {
    w = lane == 7 ? 1u << (kGridBlankBit & 31) : 0u;
    bad = false;
  }
  if (lane < 8) ((u32*)(state + (size_t)e * kGridStateBytes))[lane] = w;
  if (ballot(bad) != 0 && lane == 0) error_flags[e] |= CE_FAULT_BAD_GRID;
}

// ----------------------------------------------------------------------------------------
// MT seeding: one thread per env (sequential recurrence), numpy init_genrand or CPython
// init_by_array([seed])
// ----------------------------------------------------------------------------------------
__global__ void k_mt_seed(u32* rng, u32 stride, u32 block_off, const u64* seeds, const uint8_t* mask, u32 E, int python_seeding) {
  const u32 e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  if (mask && mask[e] == 0) return;
  u32* mt = rng + (size_t)e * stride + block_off;
  const u32 seed = (u32)seeds[e];
  u32 s = python_seeding ? 19650218u : seed;
  for (u32 i = 0; i < (u32)kMtN; ++i) {
    mt[i] = s;
    s = 1812433253u * (s ^ (s >> 30)) + i + 1u;
  }
  if (python_seeding) {
    u32 i = 1;
    for (u32 k = kMtN; k; --k) {  // key_length = 1, j stays 0
      mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525u)) + seed + 0u;
      ++i;
      if (i >= (u32)kMtN) {
        mt[0] = mt[kMtN - 1];
        i = 1;
      }
    }
    for (u32 k = kMtN - 1; k; --k) {
      mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941u)) - i;
      ++i;
      if (i >= (u32)kMtN) {
        mt[0] = mt[kMtN - 1];
        i = 1;
      }
    }
    mt[0] = 0x80000000u;
  }
  mt[kMtN] = kMtN;  // pos: state exhausted, first draw twists
  mt[kMtN + 1] = mt[kMtN + 2] = mt[kMtN + 3] = 0;
}

__global__ void k_synth_u8(uint8_t* out, u64 key, u64 env_base, u32 E, u32 n, u32 t0, u32 T, u32 num_actions) {
  const size_t total = (size_t)T * E * n;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const u32 a = (u32)(i % n);
    const size_t r = i / n;
    const u32 e = (u32)(r % E);
    const u32 t = (u32)(r / E);
    out[i] = (uint8_t)synth_action(key, env_base + e, t0 + t, a, num_actions);
  }
}

// ----------------------------------------------------------------------------------------
// self-test of the wave primitives (ce_selftest): bit i of out[0] set = check i failed
// ----------------------------------------------------------------------------------------
__global__ void k_selftest(u32* out) {
  __shared__ u32 mt[kMtN];
  __shared__ u32 ref[kMtN];
  const u32 lane = lane_id();
  u32 fail = 0;
  // (0) wave_min_u32 against a shuffle butterfly
  for (u32 round = 0; round < 8; ++round) {
    u32 v = (lane * 2654435761u + round * 40503u) ^ (round << 20);
    if (round == 3) v = 0xffffffffu;
    if (round == 4) v = lane == 63 ? 5u : 0xffffffffu;
    if (round == 5) v = lane == 0 ? 7u : 0xffffffffu;
    u32 b = v;
    for (int s = 1; s < 64; s <<= 1) {
      u32 o = bperm(b, lane ^ s);
      b = o < b ? o : b;
    }
    if (wave_min_u32(v) != rfl(b)) fail |= 1u;
  }
  // (1) parallel twist against the sequential recurrence
  for (u32 k = lane; k < (u32)kMtN; k += 64) {
    u32 s = 1234567u + k * 2246822519u;
    s ^= s >> 15;
    mt[k] = s;
    ref[k] = s;
  }
  wave_sync();
  if (lane == 0) {
    for (int i = 0; i < kMtN; ++i) {
      u32 y = (ref[i] & 0x80000000u) | (ref[(i + 1) % kMtN] & 0x7fffffffu);
      ref[i] = ref[(i + kMtM) % kMtN] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
  }
  wave_sync();
  mt_twist(mt, lane);
  bool bad = false;
  for (u32 k = lane; k < (u32)kMtN; k += 64) bad = bad || (mt[k] != ref[k]);
  if (ballot(bad)) fail |= 2u;
  // (2) readlane / writelane list swap
  {
    u32 L0 = lane, L1 = lane + 64;
    u32 vi = rdl(L0, 5), vj = rdl(L1, 3);
    L0 = wrl(vj, 5, L0);
    L1 = wrl(vi, 3, L1);
    if (ballot((lane == 5 && L0 != 67) || (lane == 3 && L1 != 5) || (lane != 5 && L0 != lane)) != 0) fail |= 4u;
  }
  // (3) pointer-chasing list update against the serial swap loop, over random and adversarial draw lists
  {
    __shared__ u32 jl[2][256];
    for (u32 trial = 0; trial < 64; ++trial) {
      const u32 len = trial < 48 ? 119u : 65u + ((trial * 37u) & 63u);
      for (u32 i = lane; i < 128; i += 64) {
        u32 h = (i * 2654435761u) ^ (trial * 0x9e3779b9u);
        h ^= h >> 13;
        h *= 0x85ebca6bu;
        h ^= h >> 16;
        u32 j = h % (i + 1);
        if (trial == 1) j = 0;
        if (trial == 2) j = i;
        if (trial == 3) j = i ? i - 1 : 0;
        if (trial == 4) j = i >= 64 ? 64 : 0;
        if (trial == 5) j = i >= 64 ? 63 : i / 2;
        if (trial == 6) j = i & 1 ? i : i / 3;
        if (trial >= 7 && trial < 16) j = h % ((i >> (trial - 6)) + 1);  // crowded low targets
        jl[0][i] = j;
        jl[1][i] = j;
      }
      wave_sync();
      u32 a0 = (lane * 7u + trial) & 127u, a1 = (lane * 5u + 3u * trial + 64u) & 127u;
      u32 b0 = a0, b1 = a1;
      shuffle_apply(a0, a1, len, jl[0], lane);
      shuffle_apply_par(b0, b1, len, jl[1], lane);
      if (ballot(a0 != b0 || (lane < len - 64 && a1 != b1)) != 0) fail |= 8u;
      wave_sync();
    }
  }
  // (6) [bits 4, 5 belong to the counter-mode self-test] the word-parallel draws of the waste-list shuffle (shuffle_draws + the serial list update) against the serial swap
  // walk (shuffle_core) from the same generator state: list contents and stream position, over start positions that put
  // the generation end before, inside and behind the draws, and over list lengths 65 .. 128
  {
    __shared__ u32 jd[128];
    for (u32 trial = 0; trial < 96; ++trial) {
      const u32 len = trial < 64 ? 119u : 65u + ((trial * 29u) & 63u);
      const u32 start = trial < 8 ? (u32)kMtN - 3u - 23u * trial : (trial * 2654435761u >> 7) % (u32)kMtN;
      for (u32 k = lane; k < (u32)kMtN; k += 64) {
        u32 h = (k + 1u) * 2246822519u ^ (trial + 1u) * 0x9e3779b9u;
        h ^= h >> 15;
        h *= 0x85ebca6bu;
        h ^= h >> 13;
        mt[k] = h;
        ref[k] = h;
      }
      wave_sync();
      Rng ra, rb;
      ra.mt = mt, rb.mt = ref;
      ra.pos = rb.pos = start;
      ra.cbase = rb.cbase = 0, ra.ccount = rb.ccount = 0, ra.cvalid = rb.cvalid = 0, ra.cache = rb.cache = 0;
      ra.twists = rb.twists = 0, ra.k0 = rb.k0 = 0, ra.k1 = rb.k1 = 0, ra.gen0 = rb.gen0 = 0;
      if (trial & 1u) {  // a consumer before the shuffle leaves a partly read cache behind
        (void)rng_next(ra, lane);
        (void)rng_next(rb, lane);
        for (u32 q = 0; q < (trial >> 1) % 40u; ++q) {
          (void)rng_next(ra, lane);
          (void)rng_next(rb, lane);
        }
      }
      u32 a0 = lane, a1 = lane + 64, b0 = lane, b1 = lane + 64;
      shuffle_core<true>(rb, b0, b1, len, lane);
      shuffle_draws(ra, len, jd, lane);
      shuffle_apply(a0, a1, len, jd, lane);
      if (ballot(a0 != b0 || (lane < len - 64 && a1 != b1)) != 0 || ra.pos != rb.pos || ra.twists != rb.twists) fail |= 64u;
      wave_sync();
    }
  }
  // (7) the stream words of a small shuffle (consume_small: the fixed-point form, nothing stored) against the find-first walk
  {
    for (u32 trial = 0; trial < 128; ++trial) {
      const u32 len = 2u + trial % 8u;  // 2 .. 9
      const u32 start = trial < 16 ? (u32)kMtN - 1u - trial : (trial * 2654435761u >> 9) % (u32)kMtN;
      for (u32 k = lane; k < (u32)kMtN; k += 64) {
        u32 h = (k + 7u) * 2246822519u ^ (trial + 3u) * 0x9e3779b9u;
        h ^= h >> 15;
        h *= 0x85ebca6bu;
        h ^= h >> 13;
        mt[k] = h;
        ref[k] = h;
      }
      wave_sync();
      Rng ra, rb;
      ra.mt = mt, rb.mt = ref;
      ra.pos = rb.pos = start;
      ra.cbase = rb.cbase = 0, ra.ccount = rb.ccount = 0, ra.cvalid = rb.cvalid = 0, ra.cache = rb.cache = 0;
      ra.twists = rb.twists = 0, ra.k0 = rb.k0 = 0, ra.k1 = rb.k1 = 0, ra.gen0 = rb.gen0 = 0;
      for (u32 q = 0; q < (trial * 5u) % 61u; ++q) {  // a partly read cache
        (void)rng_next(ra, lane);
        (void)rng_next(rb, lane);
      }
      u32 d0 = 0;
      shuffle_small<2>(rb, d0, len, lane);
      consume_small(ra, len, lane);
      if (ra.pos != rb.pos || ra.twists != rb.twists || rng_next(ra, lane) != rng_next(rb, lane)) fail |= 128u;
      wave_sync();
    }
  }
  if (lane == 0) out[0] = fail;
}

#else  // CE_RNG_COUNTER
// ----------------------------------------------------------------------------------------
// counter mode: seeding = writing the key; self-test = the Random123 known-answer vectors of Philox4x32-10
// ----------------------------------------------------------------------------------------
__global__ void k_ctr_seed(u32* rng, const u64* seeds, const uint8_t* mask, u32 E) {
  const u32 e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  if (mask && mask[e] == 0) return;
  u32* row = rng + (size_t)e * kRngRow;
  row[0] = (u32)seeds[e];
  row[1] = (u32)(seeds[e] >> 32);
  row[2] = 0;  // generation 0 counts as used up: the first draw opens generation 1
  row[3] = 0;
}
__global__ void k_ctr_selftest(u32* out) {
  __shared__ u32 gen[CE_RNG_COUNTER_GEN];
  const u32 lane = lane_id();
  // Random123 kat_vectors, philox4x32 10 rounds: counter / key / expected
  const u32 kat[3][10] = {
      {0, 0, 0, 0, 0, 0, 0x6627e8d5u, 0xe169c58du, 0xbc57ac4cu, 0x9b00dbd8u},
      {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0x408f276du, 0x41c83b0eu, 0xa20bc7c6u, 0x6d5451fdu},
      {0x243f6a88u, 0x85a308d3u, 0x13198a2eu, 0x03707344u, 0xa4093822u, 0x299f31d0u, 0xd16cfe09u, 0x94fdccebu, 0x5001e420u, 0x24126ea1u}};
  u32 fail = 0;
  for (int v = 0; v < 3; ++v) {
    u32 c0 = kat[v][0], c1 = kat[v][1], c2 = kat[v][2], c3 = kat[v][3];
    philox4x32_10(kat[v][4], kat[v][5], c0, c1, c2, c3);
    if (c0 != kat[v][6] || c1 != kat[v][7] || c2 != kat[v][8] || c3 != kat[v][9]) fail |= 16u;
  }
  // the LDS fill against per-block evaluation: word 4 q + j of generation g = block (q, g, 0, 0) word j
  ctr_fill_lds((lds_u32*)gen, lane, 0xa4093822u, 0x299f31d0u, 7u);
  for (u32 q = lane; q < CE_RNG_COUNTER_GEN / 4; q += 64) {
    u32 c0 = q, c1 = 7u, c2 = 0, c3 = 0;
    philox4x32_10(0xa4093822u, 0x299f31d0u, c0, c1, c2, c3);
    if (gen[4 * q] != c0 || gen[4 * q + 1] != c1 || gen[4 * q + 2] != c2 || gen[4 * q + 3] != c3) fail |= 32u;
  }
  fail = ballot(fail & 16u) ? (fail | 16u) : fail;
  fail = ballot(fail & 32u) ? (fail | 32u) : fail;
  if (lane == 0) out[0] = fail;
}
#endif  // CE_RNG_COUNTER

// ----------------------------------------------------------------------------------------
// host launchers
// ----------------------------------------------------------------------------------------
int CE_LAUNCHER(upload_grid_tables)(int kind, const GridTables& t, const u32* rgb16) {
  if (hipMemcpyToSymbol(HIP_SYMBOL(c_tab), &t, sizeof(GridTables), sizeof(GridTables) * kind) != hipSuccess) return -1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(c_rgb), rgb16, sizeof(u32) * 16) != hipSuccess) return -1;
  return 0;
}

#ifndef CE_RNG_COUNTER
void launch_mt_seed(u32* rng, u32 stride_words, u32 block_offset_words, const u64* seeds_dev, const uint8_t* mask_dev,
                    u32 E, int python_seeding, void* stream) {
  hipLaunchKernelGGL(k_mt_seed, dim3((E + 63) / 64), dim3(64), 0, (hipStream_t)stream, rng, stride_words,
                     block_offset_words, seeds_dev, mask_dev, E, python_seeding);
}
#else
void launch_seed_ctr(u32* rng, const u64* seeds_dev, const uint8_t* mask_dev, u32 E, void* stream) {
  hipLaunchKernelGGL(k_ctr_seed, dim3((E + 63) / 64), dim3(64), 0, (hipStream_t)stream, rng, seeds_dev, mask_dev, E);
}
int launch_selftest_ctr(u32* out_dev, void* stream) {
  hipLaunchKernelGGL(k_ctr_selftest, dim3(1), dim3(64), 0, (hipStream_t)stream, out_dev);
  return 0;
}
#endif

// CE_EXTRA_LDS (bytes of unused dynamic LDS per workgroup) is an occupancy-sweep knob for experiments
static unsigned extra_lds() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("CE_EXTRA_LDS");
    v = e ? atoi(e) : 0;
  }
  return (unsigned)v;
}
#define CE_LAUNCH_GRID(kern)                                                                          \
  do {                                                                                                \
    const u32 first = p.env_first, count = p.env_count ? p.env_count : p.E - p.env_first;             \
    dim3 grid((count + kWavesPerBlock - 1) / kWavesPerBlock), block(64 * kWavesPerBlock);             \
    if (kind == CE_KIND_CLEANUP && !p.custom_map)                                                     \
      hipLaunchKernelGGL((kern<CE_KIND_CLEANUP, false>), grid, block, extra_lds(), (hipStream_t)stream, dp, \
                         p.actions, p.mask, first, first + count);                                    \
    else if (kind == CE_KIND_CLEANUP)                                                                 \
      hipLaunchKernelGGL((kern<CE_KIND_CLEANUP, true>), grid, block, extra_lds(), (hipStream_t)stream, dp, \
                         p.actions, p.mask, first, first + count);                                    \
    else if (!p.custom_map)                                                                           \
      hipLaunchKernelGGL((kern<CE_KIND_HARVEST, false>), grid, block, extra_lds(), (hipStream_t)stream, dp, \
                         p.actions, p.mask, first, first + count);                                    \
    else                                                                                              \
      hipLaunchKernelGGL((kern<CE_KIND_HARVEST, true>), grid, block, extra_lds(), (hipStream_t)stream, dp, \
                         p.actions, p.mask, first, first + count);                                    \
  } while (0)

void CE_LAUNCHER(launch_grid_construct)(int kind, const GridParams& p, const GridParams* dp, void* stream) { CE_LAUNCH_GRID(k_grid_construct); }
void CE_LAUNCHER(launch_grid_reset)(int kind, const GridParams& p, const GridParams* dp, void* stream) { CE_LAUNCH_GRID(k_grid_reset); }
void CE_LAUNCHER(launch_grid_step)(int kind, const GridParams& p, const GridParams* dp, void* stream) {
  const u32 first = p.env_first, count = p.env_count ? p.env_count : p.E - p.env_first;
  dim3 grid((count + kWavesPerBlock - 1) / kWavesPerBlock), block(64 * kWavesPerBlock);
  // the state pointers travel as kernel arguments (preloaded into SGPRs at wave launch), see k_grid_step
#define CE_STEP_LAUNCH(K_, N_)                                                                                          \
  hipLaunchKernelGGL((k_grid_step<K_, N_>), grid, block, extra_lds(), (hipStream_t)stream, p.actions, first, p.n | (p.obs_wt ? 0x100u : 0u), (u32*)p.rng, \
                     (uint8_t*)p.grid, (uint8_t*)p.agents, (uint8_t*)p.waste_perm, dp)
  if (p.custom_map) {  // a caller's layout (ce_config.ascii_map): the instance that reads tables and list lengths from the block
    if (kind == CE_KIND_CLEANUP)
      hipLaunchKernelGGL((k_grid_step<CE_KIND_CLEANUP, 0, 0, true>), grid, block, extra_lds(), (hipStream_t)stream, p.actions, first, p.n | (p.obs_wt ? 0x100u : 0u),
                         (u32*)p.rng, (uint8_t*)p.grid, (uint8_t*)p.agents, (uint8_t*)p.waste_perm, dp);
    else
      hipLaunchKernelGGL((k_grid_step<CE_KIND_HARVEST, 0, 0, true>), grid, block, extra_lds(), (hipStream_t)stream, p.actions, first, p.n | (p.obs_wt ? 0x100u : 0u),
                         (u32*)p.rng, (uint8_t*)p.grid, (uint8_t*)p.agents, (uint8_t*)p.waste_perm, dp);
    return;
  }
  if (kind == CE_KIND_CLEANUP) {
    if (p.n == 8) CE_STEP_LAUNCH(CE_KIND_CLEANUP, 8);
    else if (p.n == 4) CE_STEP_LAUNCH(CE_KIND_CLEANUP, 4);  // BASELINE config 1
    else CE_STEP_LAUNCH(CE_KIND_CLEANUP, 0);
  } else {
    if (p.n == 8) CE_STEP_LAUNCH(CE_KIND_HARVEST, 8);
    else CE_STEP_LAUNCH(CE_KIND_HARVEST, 0);
  }
#undef CE_STEP_LAUNCH
}

// ce_step_policy: the same launch with the action taken from the policy's output inside the kernel (p.actions = that output)
void CE_LAUNCHER(launch_grid_step_policy)(int kind, int policy, const GridParams& p, const GridParams* dp, void* stream) {
  const u32 first = p.env_first, count = p.env_count ? p.env_count : p.E - p.env_first;
  dim3 grid((count + kWavesPerBlock - 1) / kWavesPerBlock), block(64 * kWavesPerBlock);
#define CE_STEP_LAUNCH(K_, N_, P_)                                                                                      \
  hipLaunchKernelGGL((k_grid_step<K_, N_, P_>), grid, block, extra_lds(), (hipStream_t)stream, (typename ActionPlane<P_>::type)p.actions, first, p.n | (p.obs_wt ? 0x100u : 0u), (u32*)p.rng, \
                     (uint8_t*)p.grid, (uint8_t*)p.agents, (uint8_t*)p.waste_perm, dp)
#define CE_STEP_POLICY(P_)                                         \
  do {                                                             \
    if (kind == CE_KIND_CLEANUP) { /* generic n: the n = 8 policy instance spills (10 VGPRs), the generic one does not */ \
      CE_STEP_LAUNCH(CE_KIND_CLEANUP, 0, P_);                      \
    } else {                                                       \
      if (p.n == 8) CE_STEP_LAUNCH(CE_KIND_HARVEST, 8, P_);        \
      else CE_STEP_LAUNCH(CE_KIND_HARVEST, 0, P_);                 \
    }                                                              \
  } while (0)
#define CE_STEP_POLICY_CM(K_, P_)                                                                                       \
  hipLaunchKernelGGL((k_grid_step<K_, 0, P_, true>), grid, block, extra_lds(), (hipStream_t)stream, (typename ActionPlane<P_>::type)p.actions, first, p.n | (p.obs_wt ? 0x100u : 0u), (u32*)p.rng, \
                     (uint8_t*)p.grid, (uint8_t*)p.agents, (uint8_t*)p.waste_perm, dp)
  if (p.custom_map) {
    if (kind == CE_KIND_CLEANUP) {
      if (policy == CE_POLICY_BYTES_MOD) CE_STEP_POLICY_CM(CE_KIND_CLEANUP, CE_POLICY_BYTES_MOD);
      else if (policy == CE_POLICY_AHEAD_NOISE) CE_STEP_POLICY_CM(CE_KIND_CLEANUP, CE_POLICY_AHEAD_NOISE);
      else CE_STEP_POLICY_CM(CE_KIND_CLEANUP, CE_POLICY_ARGMAX_F32);
    } else {
      if (policy == CE_POLICY_BYTES_MOD) CE_STEP_POLICY_CM(CE_KIND_HARVEST, CE_POLICY_BYTES_MOD);
      else if (policy == CE_POLICY_AHEAD_NOISE) CE_STEP_POLICY_CM(CE_KIND_HARVEST, CE_POLICY_AHEAD_NOISE);
      else CE_STEP_POLICY_CM(CE_KIND_HARVEST, CE_POLICY_ARGMAX_F32);
    }
  } else if (policy == CE_POLICY_BYTES_MOD) CE_STEP_POLICY(CE_POLICY_BYTES_MOD);
  else if (policy == CE_POLICY_AHEAD_NOISE) CE_STEP_POLICY(CE_POLICY_AHEAD_NOISE);
  else CE_STEP_POLICY(CE_POLICY_ARGMAX_F32);
#undef CE_STEP_POLICY_CM
#undef CE_STEP_POLICY
#undef CE_STEP_LAUNCH
}

void CE_LAUNCHER(launch_grid_rollout)(int kind, u32 num_agents, bool custom_map, const GridParams* dp, const RolloutArgs& ra, void* stream) {
  const u32 count = ra.env_end - ra.env_first;
  dim3 grid((count + kWavesPerBlock - 1) / kWavesPerBlock), block(64 * kWavesPerBlock);
  if (custom_map) {  // a caller's layout: one instance per kind, at the occupancy the shipped rollout runs at
    if (kind == CE_KIND_CLEANUP)
      hipLaunchKernelGGL((k_grid_rollout<CE_KIND_CLEANUP, CE_CLEANUP_ROLLOUT_WAVES, 0, true>), grid, block, extra_lds(), (hipStream_t)stream, dp, ra);
    else
      hipLaunchKernelGGL((k_grid_rollout<CE_KIND_HARVEST, CE_HARVEST_ROLLOUT_WAVES, 0, true>), grid, block, extra_lds(), (hipStream_t)stream, dp, ra);
    return;
  }
  // a launch of fewer waves than a third of the machine's 8 192 wave slots (three slices are in flight) never queues; one of
  // a sixth leaves four waves per SIMD
  constexpr u32 kMidLaunch = 2730, kSmallLaunch = 1366;
#define CE_ROLLOUT_LAUNCH(K_, W_, N_) \
  hipLaunchKernelGGL((k_grid_rollout<K_, W_, N_>), grid, block, extra_lds(), (hipStream_t)stream, dp, ra)
  if (kind == CE_KIND_CLEANUP) {
    // (the n = 8 instance of the cleanup rollout measured the same as the generic one — 4.35 G both, 35 spilled registers
    // instead of 24 — so only the generic one is built; harvest's gains 8 %: 5.44 -> 5.89 G)
    if (count <= kSmallLaunch && num_agents == 4) CE_ROLLOUT_LAUNCH(CE_KIND_CLEANUP, CE_CLEANUP_ROLLOUT_WAVES_SMALL, 4);  // BASELINE config 1
    else if (count <= kSmallLaunch) CE_ROLLOUT_LAUNCH(CE_KIND_CLEANUP, CE_CLEANUP_ROLLOUT_WAVES_SMALL, 0);
    else if (count <= kMidLaunch) CE_ROLLOUT_LAUNCH(CE_KIND_CLEANUP, CE_CLEANUP_ROLLOUT_WAVES_MID, 0);
    else CE_ROLLOUT_LAUNCH(CE_KIND_CLEANUP, CE_CLEANUP_ROLLOUT_WAVES, 0);
  } else {
    if (num_agents == 8) CE_ROLLOUT_LAUNCH(CE_KIND_HARVEST, CE_HARVEST_ROLLOUT_WAVES, 8);
    else CE_ROLLOUT_LAUNCH(CE_KIND_HARVEST, CE_HARVEST_ROLLOUT_WAVES, 0);
  }
#undef CE_ROLLOUT_LAUNCH
}

#ifndef CE_RNG_COUNTER
void launch_grid_expand(int kind, const uint8_t* state, uint8_t* image, u32 env_first, u32 env_count, const GridTables* tab, u32 napple,
                        u32 nwaste, void* stream) {
  if (kind == CE_KIND_CLEANUP) hipLaunchKernelGGL(k_grid_expand<CE_KIND_CLEANUP>, dim3(env_count), dim3(64), 0, (hipStream_t)stream, state, image, env_first, env_count, tab, napple, nwaste);
  else hipLaunchKernelGGL(k_grid_expand<CE_KIND_HARVEST>, dim3(env_count), dim3(64), 0, (hipStream_t)stream, state, image, env_first, env_count, tab, napple, nwaste);
}
void launch_grid_global_view(int kind, const uint8_t* state, const uint8_t* agents, const int32_t* timestep, uint8_t* out, u32 env_first,
                             u32 env_count, u32 n, const GridTables* tab, u32 napple, u32 nwaste, u32 H, u32 W, void* stream) {
  if (kind == CE_KIND_CLEANUP)
    hipLaunchKernelGGL(k_grid_global_view<CE_KIND_CLEANUP>, dim3(env_count), dim3(64), 0, (hipStream_t)stream, state, agents, timestep, out, env_first,
                       env_count, n, tab, napple, nwaste, H, W);
  else
    hipLaunchKernelGGL(k_grid_global_view<CE_KIND_HARVEST>, dim3(env_count), dim3(64), 0, (hipStream_t)stream, state, agents, timestep, out, env_first,
                       env_count, n, tab, napple, nwaste, H, W);
}
void launch_grid_pack(int kind, const uint8_t* image, uint8_t* state, u32* error_flags, u32 env_first, u32 env_count, const GridTables* tab,
                      u32 napple, u32 nwaste, void* stream) {
  if (kind == CE_KIND_CLEANUP) hipLaunchKernelGGL(k_grid_pack<CE_KIND_CLEANUP>, dim3(env_count), dim3(64), 0, (hipStream_t)stream, image, state, error_flags, env_first, env_count, tab, napple, nwaste);
  else hipLaunchKernelGGL(k_grid_pack<CE_KIND_HARVEST>, dim3(env_count), dim3(64), 0, (hipStream_t)stream, image, state, error_flags, env_first, env_count, tab, napple, nwaste);
}

#define CE_LAUNCH_FEAT(kern)                                                                                    \
  do {                                                                                                          \
    const u32 first = p.env_first, count = p.env_count ? p.env_count : p.E - p.env_first;                       \
    if (kind == CE_KIND_HARVEST_FEATURES)                                                                       \
      hipLaunchKernelGGL(kern<CE_KIND_HARVEST>, dim3(count), dim3(64), 0, (hipStream_t)stream, dp, p.actions,   \
                         p.mask, first, first + count);                                                         \
    else                                                                                                        \
      hipLaunchKernelGGL(kern<CE_KIND_CLEANUP>, dim3(count), dim3(64), 0, (hipStream_t)stream, dp, p.actions,   \
                         p.mask, first, first + count);                                                         \
  } while (0)
void launch_feat_construct(int kind, const GridParams& p, const GridParams* dp, void* stream) { CE_LAUNCH_FEAT(k_feat_construct); }
void launch_feat_reset(int kind, const GridParams& p, const GridParams* dp, void* stream) { CE_LAUNCH_FEAT(k_feat_reset); }
// Smallest launch (envs) of HarvestFeatures n = 2 that runs four envs per wave.  Below ~2 000 envs a launch is a handful of waves
// per CU either way and lasts as long as ONE wave's chain, which is shorter with one env per wave (measured, three slices:
// 1 365 envs per launch 9.5 vs 9.9 us per step, 2 048 equal, 2 730 and up the packed kernels win: 5 461 per launch 11.6 vs
// 14.4 us, fused 7.4 vs 12.5).  CE_FEAT_QUAD_MIN_ENVS=N moves the threshold, CE_FEAT_QUAD=0 switches the packing off (A/B runs).
static u32 feat_quad_min() {
  static const u32 n = [] {
    const char* v = getenv("CE_FEAT_QUAD");
    if (v && v[0] == '0') return 0xffffffffu;
    const char* m = getenv("CE_FEAT_QUAD_MIN_ENVS");
    return m ? (u32)strtoul(m, nullptr, 10) : 2048u;
  }();
  return n;
}
void launch_feat_step(int kind, const GridParams& p, const GridParams* dp, void* stream) {
  const u32 first = p.env_first, count = p.env_count ? p.env_count : p.E - p.env_first;
#define CE_FEAT_STEP_LAUNCH(K_, N_) \
  hipLaunchKernelGGL((k_feat_step<K_, N_>), dim3(count), dim3(64), 0, (hipStream_t)stream, dp, p.actions, p.mask, first, first + count)
  if (kind == CE_KIND_HARVEST_FEATURES) {
    if (p.n == 2 && count >= feat_quad_min())
      hipLaunchKernelGGL(k_feat_step_quad, dim3((count + 3u) / 4u), dim3(64), 0, (hipStream_t)stream, dp, p.actions, first, first + count);
    else if (p.n == 2) CE_FEAT_STEP_LAUNCH(CE_KIND_HARVEST, 2);
    else CE_FEAT_STEP_LAUNCH(CE_KIND_HARVEST, 0);
  } else {
    if (p.n == 2) CE_FEAT_STEP_LAUNCH(CE_KIND_CLEANUP, 2);
    else CE_FEAT_STEP_LAUNCH(CE_KIND_CLEANUP, 0);
  }
#undef CE_FEAT_STEP_LAUNCH
}
void launch_feat_rollout(int kind, u32 num_agents, const GridParams* dp, const RolloutArgs& ra, void* stream) {
  const u32 count = ra.env_end - ra.env_first;
#define CE_FEAT_ROLLOUT_LAUNCH(K_, N_) \
  hipLaunchKernelGGL((k_feat_rollout<K_, N_>), dim3(count), dim3(64), 0, (hipStream_t)stream, dp, ra)
  if (kind == CE_KIND_HARVEST_FEATURES) {
    if (num_agents == 2 && count >= feat_quad_min())
      hipLaunchKernelGGL(k_feat_rollout_quad, dim3((count + 3u) / 4u), dim3(64), 0, (hipStream_t)stream, dp, ra);
    else if (num_agents == 2) CE_FEAT_ROLLOUT_LAUNCH(CE_KIND_HARVEST, 2);
    else CE_FEAT_ROLLOUT_LAUNCH(CE_KIND_HARVEST, 0);
  } else {
    if (num_agents == 2) CE_FEAT_ROLLOUT_LAUNCH(CE_KIND_CLEANUP, 2);
    else CE_FEAT_ROLLOUT_LAUNCH(CE_KIND_CLEANUP, 0);
  }
#undef CE_FEAT_ROLLOUT_LAUNCH
}

void launch_synth_actions_u8(uint8_t* out, u64 key, u64 env_base, u32 E, u32 n, u32 t0, u32 T, u32 num_actions,
                             void* stream) {
  hipLaunchKernelGGL(k_synth_u8, dim3(2048), dim3(256), 0, (hipStream_t)stream, out, key, env_base, E, n, t0, T,
                     num_actions);
}

int launch_selftest(u32* out_dev, void* stream) {
  hipLaunchKernelGGL(k_selftest, dim3(1), dim3(64), 0, (hipStream_t)stream, out_dev);
  return 0;
}
#endif  // !CE_RNG_COUNTER

#ifdef CE_RNG_COUNTER
}  // inline namespace ctr
#endif
}  // namespace ce
