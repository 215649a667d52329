// ce_device.h — structures shared by the host API (ce_api.hip) and the gfx950 kernels
// (ce_grid_kernels.hip, ce_selfdrive_kernels.hip).  Not part of the public C-ABI.
#pragma once
#include <stdint.h>

#include "../../include/contracts_engine.h"

namespace ce {

constexpr int kWave = 64;
constexpr int kView = 7;    // CLEANUP_VIEW_SIZE / HARVEST_VIEW_SIZE (cleanup_new.py:51, harvest_new.py:36)
constexpr int kWin = 15;    // 2*view+1
constexpr int kPixPerAgent = kWin * kWin;  // 225
constexpr int kMaxGridAgents = 9;
// obs pitch: a view row is 16 pixels x 3 B (the 16th is padding), so the 12-byte / 4-pixel store units of the
// crop kernel tile rows exactly: 4 units per row, 60 per agent, no unit straddles a row or an agent
constexpr int kObsRowStride = 48, kObsAgentStride = kWin * kObsRowStride, kObsUnitsPerAgent = kObsAgentStride / 12;
constexpr int kMtN = 624, kMtM = 397;
// Persistent map state per env: 8 dwords.  Bits 0..127 = "apple present" per apple cell (row-major index), bits
// 128..246 = "waste present" per waste cell (harvest: apple bits 0..154), bit 255 = map still blank (constructed,
// never reset).  Everything else on the map is static; the image is rebuilt from the constant base map.
constexpr int kGridStateBytes = 32, kGridBlankBit = 255;
constexpr int kRngStride = CE_RNG_WORDS_GRID;  // words per env row

// Geometry of the two grid families.  The LDS copy of the map is padded by the view radius
// on every side (PW x PH) so the egocentric crop and every neighbourhood scan need no
// bounds checks; both maps have a closed wall perimeter, so moves and beams never leave
// the interior either.
template <int KIND> struct Geo;
template <> struct Geo<CE_KIND_CLEANUP> {
  // PW = 36 (9 dwords), not 32: the crop gathers of the observation pass read a 15 x 15 window with 60 lanes; with a
  // 32-byte pitch four lanes of every left / right facing view and two of every up / down facing one land on the same
  // LDS bank (ds_read_u8 banks: (a / 4) mod 32 per 32-lane half) — 128 of the step's 210 conflict cycles; 36 makes the
  // left / right views conflict free and leaves one extra cycle on the up / down ones
  static constexpr int H = 25, W = 18, CELLS = 450, PW = 36, PH = 39, PCELLS = PW * PH;
  static constexpr int PQUADS = (PCELLS + 15) / 16;  // the LDS map is padded to whole 16-byte quads
  static constexpr int NAPPLE = 103, NWASTE = 119, RANDW = 2 * (103 + 119), NSPAWN_CTOR = 10;
  // LDS keeps the words of the apple doubles only; of a waste double only "u < 0.5" matters, which is
  // bit 31 of its first word (kept as one byte per double)
  static constexpr int UWORDS = 2 * 103, SBYTES = 224;
  // HBM keeps the map in the SAME bordered layout as LDS: loading / storing an env's map is a straight
  // dword copy (no per-cell index arithmetic, no separate border zeroing)
  static constexpr int IMAGE_STRIDE = (PCELLS + 15) / 16 * 16;  // bytes of the padded image ce_download("grid") returns
};
template <> struct Geo<CE_KIND_HARVEST> {
  static constexpr int H = 16, W = 38, CELLS = 608, PW = 52, PH = 30, PCELLS = PW * PH;
  static constexpr int PQUADS = (PCELLS + 15) / 16;
  static constexpr int NAPPLE = 155, NWASTE = 0, RANDW = 2 * 155, NSPAWN_CTOR = 20;
  static constexpr int UWORDS = 192, SBYTES = 16;  // the feature pass's key list (48 four-cell chunks >= 155 apple cells); keeps the wave's LDS slice under 5 KB = 8 waves/SIMD
  static constexpr int IMAGE_STRIDE = (PCELLS + 15) / 16 * 16;  // bytes of the padded image ce_download("grid") returns
};

// Static per-family tables (host-built from the ASCII maps, uploaded to __constant__).
// Cell entries are packed:  padded_index (11 bits) | col << 16 | row << 24  (one byte per coordinate).
// Bit 63 of a cleanup apple_thresh entry: the waste spawn probability is non-zero at that waste count.  The flag
// rides in the threshold word because a second, byte-indexed scalar table made the compiler fold both accesses
// onto a common base + nH, and s_load drops the low two bits of a misaligned base.
constexpr uint64_t kWasteOnBit = 1ull << 63;

struct GridTables {
  uint32_t apple[160];         // apple spawn cells, row-major (cleanup 'B', harvest 'A')
  uint32_t waste[128];         // cleanup waste cells 'H' u 'R', row-major
  uint32_t spawn[20];          // 'P' cells; cleanup: entries 10..19 repeat 0..9 (cleanup_new.py:114-115)
  uint64_t apple_thresh[120];  // cleanup: by #H on the map -> ceil(p_apple * 2^53) | kWasteOnBit; harvest: [0..3] by neighbour count
  uint8_t waste_on[120];       // cleanup: by #H -> waste spawn probability is non-zero (host-side copy of the flag bit)
  uint8_t base_pmap[1568];     // padded reset-time map (walls + H/R/S, or harvest apples)
  uint8_t base_pmap4[1568];    // the same with every code pre-scaled by 4 (the grid kernels' LDS form: 16-byte copies, no shift)
  uint32_t close_off[24];      // harvest: 21 padded-index offsets with j^2+k^2 <= 5 (as int32)
};

// Device-side view of the parameter blocks: the pointers are typed into the global address space so that every
// access through them is a global_load / global_store with an SGPR base (no flat addressing, no 64-bit VALU
// address arithmetic).  Host code (CE_PLAIN_PARAM_POINTERS) sees the same layout with ordinary pointers.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(CE_PLAIN_PARAM_POINTERS)
#define CE_GPTR(T) __attribute__((address_space(1))) T*
#else
#define CE_GPTR(T) T*
#endif

struct GridParams {
  // persistent state
  CE_GPTR(uint8_t) grid;
  CE_GPTR(uint8_t) agents;
  CE_GPTR(uint8_t) spawn_perm;
  CE_GPTR(uint8_t) waste_perm;
  CE_GPTR(uint32_t) rng;
  CE_GPTR(int32_t) timestep;
  CE_GPTR(double) theta;
  // outputs
  CE_GPTR(uint8_t) obs;
  CE_GPTR(int32_t) base_reward;
  CE_GPTR(double) reward;
  CE_GPTR(uint8_t) done;
  CE_GPTR(uint8_t) info;
  CE_GPTR(int16_t) features;
  CE_GPTR(int64_t) int_metrics;
  CE_GPTR(double) f64_metrics;
  CE_GPTR(int64_t) final_int_metrics;
  CE_GPTR(double) final_f64_metrics;
  CE_GPTR(uint32_t) error_flags;
  CE_GPTR(unsigned long long) debug;  // [E][16] phase cycle stamps (only written by CE_PHASE_STAMPS builds)
  CE_GPTR(uint8_t) beam_map;          // [E][H*W] CE_BEAM_*, written under CE_FLAG_BEAM_TRACE only
  CE_GPTR(uint8_t) actions_taken;     // [E][n] the action ids a ce_step_policy launch derived from the policy's output
  CE_GPTR(const GridTables) tab;      // the handle's own static tables (a caller's layout, ce_config.ascii_map); null = c_tab[kind]
  // inputs
  CE_GPTR(const uint8_t) actions;  // [E][n]
  CE_GPTR(const uint8_t) mask;     // [E] or null (seed/reset)
  uint32_t E, n, horizon, contract, flags, obs_env_stride, num_features;
  uint32_t replay_constructor;
  uint32_t env_first, env_count;  // host-side launch range (count 0 = through the last env)
  uint32_t custom_map;            // 1 = the kernels' CM instances run: tables from `tab`, list lengths below
  uint32_t napple, nwaste, nspawn, map_h, map_w;
  uint32_t obs_wt;                // single-step launches write the views through the L2 (sc1) instead of nontemporal (write_obs)
  double contract_low, contract_high, null_prob, alpha, beta;
};

// By-value kernel argument of the fused rollout kernels (ce_rollout_fused).  Step s of the launch reads action plane s
// and writes plane (plane0 + s) mod num_planes of each output array; *_plane = distance between consecutive planes in
// ELEMENTS of that array (0 = the array is the handle's own per-step buffer and every step overwrites it).
struct RolloutArgs {
  CE_GPTR(const void) actions;  // [num_steps][E][n] uint8 ids (grid / feature kinds) or float32 accelerations (selfdrive)
  CE_GPTR(uint8_t) obs;
  CE_GPTR(double) obs_f64;      // selfdrive
  CE_GPTR(int32_t) base_reward;
  CE_GPTR(double) reward;
  CE_GPTR(uint8_t) done;
  CE_GPTR(uint8_t) done_agents;  // selfdrive
  CE_GPTR(uint8_t) info;
  CE_GPTR(int16_t) features;
  CE_GPTR(double) sd_info;      // selfdrive
  uint64_t action_plane, obs_plane, obs_f64_plane, agent_plane, reward_plane, done_plane, done_agents_plane, info_plane, features_plane,
      sd_info_plane;
  uint32_t num_steps, num_planes, plane0;
  uint32_t env_first, env_end;
};

struct SdParams {
  CE_GPTR(double) sd_state;
  CE_GPTR(uint32_t) rng;
  CE_GPTR(double) theta;
  CE_GPTR(double) obs_f64;
  CE_GPTR(int32_t) base_reward;
  CE_GPTR(double) reward;
  CE_GPTR(uint8_t) done;
  CE_GPTR(uint8_t) done_agents;
  CE_GPTR(uint8_t) info;
  CE_GPTR(double) sd_info;
  CE_GPTR(double) f64_metrics;
  CE_GPTR(double) final_f64_metrics;
  CE_GPTR(int64_t) int_metrics;
  CE_GPTR(int64_t) final_int_metrics;
  CE_GPTR(uint32_t) error_flags;
  CE_GPTR(const float) actions;
  CE_GPTR(const uint8_t) active;
  CE_GPTR(const uint8_t) mask;
  uint32_t E, n, contract, flags, replay_constructor;
  uint32_t env_first, env_count;  // host-side launch range (count 0 = through the last env)
  uint32_t obs_wt;                // single-step launches write the observation rows through the L2 (sc1), see GridParams
  double contract_low, contract_high, null_prob, low_bound, high_bound, start_vel, start_vel_ambulance;
};

// host-callable launchers implemented in the kernel translation units
void launch_feat_construct(int kind, const GridParams& p, const GridParams* dp, void* stream);
void launch_feat_reset(int kind, const GridParams& p, const GridParams* dp, void* stream);
void launch_feat_step(int kind, const GridParams& p, const GridParams* dp, void* stream);
void launch_feat_rollout(int kind, uint32_t num_agents, const GridParams* dp, const RolloutArgs& ra, void* stream);
int upload_grid_tables(int kind, const GridTables& t, const uint32_t* rgb16);
void launch_grid_expand(int kind, const uint8_t* state, uint8_t* image, uint32_t env_first, uint32_t env_count, const GridTables* tab,
                        uint32_t napple, uint32_t nwaste, void* stream);
void launch_grid_global_view(int kind, const uint8_t* state, const uint8_t* agents, const int32_t* timestep, uint8_t* out, uint32_t env_first,
                             uint32_t env_count, uint32_t n, const GridTables* tab, uint32_t napple, uint32_t nwaste, uint32_t H, uint32_t W,
                             void* stream);
void launch_grid_pack(int kind, const uint8_t* image, uint8_t* state, uint32_t* error_flags, uint32_t env_first, uint32_t env_count,
                      const GridTables* tab, uint32_t napple, uint32_t nwaste, void* stream);
void launch_mt_seed(uint32_t* rng, uint32_t stride_words, uint32_t block_offset_words, const uint64_t* seeds_dev,
                    const uint8_t* mask_dev, uint32_t E, int python_seeding, void* stream);
// `p` carries the per-call pointers (actions / mask) and the batch size; `dp` is the device-resident copy of
// the handle's parameter block that the kernels read field by field (keeps them out of long-lived SGPRs)
void launch_grid_construct(int kind, const GridParams& p, const GridParams* dp, void* stream);
void launch_grid_reset(int kind, const GridParams& p, const GridParams* dp, void* stream);
void launch_grid_step(int kind, const GridParams& p, const GridParams* dp, void* stream);
void launch_grid_step_policy(int kind, int policy, const GridParams& p, const GridParams* dp, void* stream);
void launch_grid_rollout(int kind, uint32_t num_agents, bool custom_map, const GridParams* dp, const RolloutArgs& ra, void* stream);
// the same kernels over the counter-RNG stream (CE_FLAG_RNG_COUNTER; ce_grid_kernels_ctr.hip), with their own constant tables
inline namespace ctr {
int upload_grid_tables_ctr(int kind, const GridTables& t, const uint32_t* rgb16);
void launch_seed_ctr(uint32_t* rng, const uint64_t* seeds_dev, const uint8_t* mask_dev, uint32_t E, void* stream);
void launch_grid_construct_ctr(int kind, const GridParams& p, const GridParams* dp, void* stream);
void launch_grid_reset_ctr(int kind, const GridParams& p, const GridParams* dp, void* stream);
void launch_grid_step_ctr(int kind, const GridParams& p, const GridParams* dp, void* stream);
void launch_grid_step_policy_ctr(int kind, int policy, const GridParams& p, const GridParams* dp, void* stream);
void launch_grid_rollout_ctr(int kind, uint32_t num_agents, bool custom_map, const GridParams* dp, const RolloutArgs& ra, void* stream);
int launch_selftest_ctr(uint32_t* out_dev, void* stream);
}  // namespace ctr
void launch_sd_construct(const SdParams& p, void* stream);
void launch_sd_reset(const SdParams& p, void* stream);
void launch_sd_step(const SdParams& p, void* stream);
void launch_sd_rollout(const SdParams& p, const RolloutArgs& ra, void* stream);
void launch_synth_actions_u8(uint8_t* out, uint64_t key, uint64_t env_base, uint32_t E, uint32_t n, uint32_t t0,
                             uint32_t T, uint32_t num_actions, void* stream);
void launch_synth_actions_f32(float* out, uint64_t key, uint64_t env_base, uint32_t E, uint32_t n, uint32_t t0,
                              uint32_t T, void* stream);
int launch_selftest(uint32_t* out_dev, void* stream);

// counter-based action hash shared by host and device (splitmix64 finaliser)
#if defined(__HIPCC__)
#define CE_HD __host__ __device__
#else
#define CE_HD
#endif
CE_HD inline uint64_t synth_hash(uint64_t key, uint64_t env, uint32_t t, uint32_t agent) {
  uint64_t z = key ^ (env * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)t << 32 | (uint64_t)agent) * 0xD1B54A32D192ED03ull;
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
CE_HD inline uint32_t synth_action(uint64_t key, uint64_t env, uint32_t t, uint32_t agent, uint32_t num_actions) {
  // multiply-shift range reduction of the high 32 bits (bias < 2^-29 for num_actions <= 9)
  return (uint32_t)(((synth_hash(key, env, t, agent) >> 32) * (uint64_t)num_actions) >> 32);
}

}  // namespace ce
