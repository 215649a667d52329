/*
 * contracts_engine.h — C-ABI of the MI355X-native batched multi-agent env engine.
 *
 * This is the drop-in boundary for the rollout hot path of
 * Algorithmic-Alignment-Lab/contracts.  The reference is pure Python and has no FFI;
 * the entry points below are what a ctypes binding on the reference side would load
 * (INTEGRATION.md shows that stub).  Each entry cites the reference interface it
 * replaces (paths relative to the reference root).
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on success or
 * a negative CE_E* code; no exceptions cross the boundary; no global state; one handle
 * per GPU; a handle is NOT thread-safe (single stream-ordered caller), distinct handles
 * are independent.  The library owns all device memory; pointers handed out by
 * ce_get_buffers stay valid until ce_destroy and are DEVICE pointers (HBM).
 *
 * One handle = E independent env replicas of one family ("kind"), n agents each,
 * stored struct-of-arrays in HBM (one array per field, each env's slice contiguous).
 */
#ifndef CONTRACTS_ENGINE_H
#define CONTRACTS_ENGINE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI 4 (round 6): ce_traj carries num_envs / num_agents (ce_rollout_fused refuses a ring sized for another batch);
 * ce_state_bytes / ce_get_state / ce_set_state (one-call snapshot); ce_set_cache_budget; ce_step_policy takes a non-const
 * plane (CE_POLICY_AHEAD_NOISE writes it).  Nothing else moved: ce_config / ce_buffers are those of ABI 3. */
#define CE_ABI_VERSION 4

/* error codes */
#define CE_OK 0
#define CE_EINVAL (-22)   /* bad argument / unsupported configuration           */
#define CE_ENOMEM (-12)   /* device or host allocation failed                    */
#define CE_ENODEV (-19)   /* no usable gfx950 device / HIP runtime error         */
#define CE_EIO (-5)       /* a kernel reported a runtime fault (see last_error)  */
#define CE_ERANGE (-34)   /* an action outside the action space was supplied     */

/* env family: which reference class the replicas restate */
enum ce_kind {
  CE_KIND_CLEANUP = 0,   /* environments/cleanup_new.py:59  CleanupEnv(MapEnv)            */
  CE_KIND_HARVEST = 1,   /* environments/harvest_new.py:48  HarvestEnv(MapEnv)            */
  CE_KIND_SELFDRIVE = 2, /* environments/self_driving_car_accelerate.py:18                */
  /* feature-vector envs of BASELINE config 0; same maps as the two grid kinds, no MapEnv logic,
   * respawn doubles and the spawn shuffle from CPython's `random`, orientations from np.random */
  CE_KIND_HARVEST_FEATURES = 3, /* environments/harvest_features.py:60  HarvestFeatures   */
  CE_KIND_CLEANUP_FEATURES = 4  /* environments/cleanup_features.py:48  CleanupFeatures   */
};

/* contract fused into the step epilogue (contract/contract_list.py) */
enum ce_contract {
  CE_CONTRACT_NONE = 0,               /* base env only (the "separate" baseline)          */
  CE_CONTRACT_CLEANUP = 1,            /* CleanupContract.compute_transfer  :22-27         */
  CE_CONTRACT_HARVEST_LOCAL = 2,      /* HarvestFeaturemodLocalContract    :45-54         */
  CE_CONTRACT_SELFDRIVE_DISTPROP = 3  /* SelfdriveContractDistprop         :69-102        */
};

/* ce_config.flags */
#define CE_FLAG_FIRING_ENABLED 0x1u   /* disable_firing=False: FIRE is in the action space
                                         (cleanup_new.py:90-95, harvest_new.py:85-90)      */
#define CE_FLAG_AUTO_RESET 0x2u       /* an env whose step returns done is reset inside the
                                         same launch (RLlib calls reset() right after a done
                                         step; same RNG order).  obs then holds the reset
                                         observation; done/reward/info/final metrics keep the
                                         terminal step's values                              */
#define CE_FLAG_COLLECTIVE_REWARD 0x4u /* map_env.py:289-292                                */
#define CE_FLAG_INEQUITY_AVERSE 0x8u   /* map_env.py:293-301 (alpha, beta)                  */
#define CE_FLAG_COLLISION_ON 0x10u     /* self_driving_car_accelerate.py:195-215            */
#define CE_FLAG_EXTERNAL_THETA 0x20u   /* the contract parameter is set by the caller (write the `theta` buffer) instead of
                                         being drawn at reset (resets leave it untouched): the negotiate / combined stages of two_stage_train.py
                                         (:215-358, :373-470) reset the base env only and take theta from an agent's action */
#define CE_FLAG_BEAM_TRACE 0x40u       /* grid kinds: every step also writes `beam_map`, the cells the step's FIRE / CLEAN beams
                                         covered (MapEnv.beam_pos, map_env.py:231,813) — what render() / full_map_to_colors()
                                         overlay on the map (map_env.py:371-373); off by default, rollouts do not need it */
#define CE_FLAG_RNG_COUNTER 0x80u      /* grid kinds, fixed at ce_create: the env's random stream is a counter-based generator
                                         (Philox4x32-10) instead of numpy's legacy MT19937.  NOT the reference's stream: a
                                         rollout in this mode is a valid sample of the same process (every draw site of
                                         map_env.py:546,685,821,831, cleanup_new.py:326,339, harvest_new.py:294 and
                                         two_stage_train.py:163-166 consumes the stream in the reference's order, through
                                         the same shuffle / rand(k) / randint / uniform arithmetic), but it reproduces no
                                         np.random.seed trace.  What it buys: 16 bytes of generator state per env in HBM
                                         instead of 2 512, and no state round trip per step.  Definition (oracle.c restates
                                         it; tests pin both to the Random123 known-answer vectors):
                                           key      = (seed & 0xffffffff, seed >> 32)   the env's seed (ce_seed takes 64 bits here)
                                           block    (g, q) = philox4x32_10(counter = (q, g, 0, 0), key)      four words
                                           generation g    = blocks q = 0 .. 127      CE_RNG_COUNTER_GEN = 512 words
                                           stream word     = MT19937's tempering applied to the block word (a bijection;
                                                             it keeps every consumer of the stream common to both modes)
                                         and every operation on an env (construct, reset, step) starts a fresh generation:
                                         the words a step leaves unread are dropped, a step that needs more than 512
                                         runs on into generation g + 1.  `rng` row: key0, key1, g = last generation used, pad */

typedef struct ce_config {
  uint32_t abi_version;    /* CE_ABI_VERSION                                               */
  uint32_t kind;           /* enum ce_kind                                                 */
  uint32_t num_envs;       /* E replicas on this handle                                    */
  uint32_t num_agents;     /* n: 1..9 grid envs (agent colours '1'..'9', map_env.py:33-41),
                              1..10 selfdrive (int(id[-1]) in contract_list.py:85)         */
  uint32_t horizon;        /* episode length (MapEnv horizon kwarg, map_env.py:74); 0 = 1000 */
  uint32_t contract;       /* enum ce_contract                                             */
  uint32_t flags;          /* CE_FLAG_*                                                    */
  int32_t device;          /* HIP device ordinal                                           */
  uint64_t env_index_base; /* global index of env 0 of this handle (multi-GPU shard offset;
                              only used by ce_reset's seed0+index convenience)             */
  double contract_low;     /* Box low of the contract space (two_stage_train.py:39)        */
  double contract_high;    /* Box high — pass the float32-rounded value the reference's Box
                              holds, e.g. (double)0.2f (two_stage_train.py:40,164)         */
  double null_prob;        /* two_stage_train.py:163 (default 0.0)                         */
  double alpha, beta;      /* inequity aversion coefficients (map_env.py:71-72)            */
  /* selfdrive constructor kwargs (self_driving_car_accelerate.py:19) */
  double low_bound, high_bound, start_vel, start_vel_ambulance;
  /* Grid kinds (CE_KIND_CLEANUP / CE_KIND_HARVEST): the ASCII layout the env is built from — the reference's `ascii_map`
   * constructor argument (map_env.py:63,117-130, cleanup_new.py:107-128).  NULL = the kind's shipped layout (ce_static_map).
   * Otherwise map_rows x map_cols characters, row-major, no terminators (copied by ce_create), in the reference's alphabet:
   * '@' wall, ' ' floor, 'P' spawn point; cleanup: 'B' apple cell, 'H' waste, 'R' river (waste can appear), 'S' stream; harvest:
   * 'A' apple.  Caps — a custom layout must fit the kind's frame and tables: at most the shipped layout's rows / columns
   * (25 x 18 cleanup, 16 x 38 harvest), apple cells (103 / 155), waste cells (119), spawn points (10 / 20), at least one of
   * each list the kind uses, and num_agents <= spawn points (the reference asserts "not enough spawn points", map_env.py:826).
   * The layout must be walled in ('@' on its whole perimeter, like the shipped ones): the reference indexes the map with whatever
   * cell a move names (Agent.return_valid_pos), so an open edge wraps around or raises there.  The view beyond the layout is
   * black, as in the reference.  CE_EINVAL (ce_last_error says which rule) otherwise.  ABI 3. */
  const char* ascii_map;
  uint32_t map_rows, map_cols;
} ce_config;

/* Shapes, filled by ce_get_buffers.  Grid families: obs is the 15x15x3 egocentric
 * uint8 crop (map_env.py:397-411) per agent; divide by 255 for the reference's float view
 * (cleanup_new.py:258).  Selfdrive: obs_f64 is the 2n+5 vector (…:241-247) plus the two
 * contract slots [theta, 0] appended by the wrapper (two_stage_train.py:113-117). */
typedef struct ce_buffers {
  uint32_t num_envs, num_agents;
  uint32_t grid_h, grid_w;       /* 25x18 cleanup, 16x38 harvest, 0 selfdrive              */
  uint32_t obs_agent_stride;     /* bytes between consecutive agents of an env in `obs` (720)  */
  uint32_t num_features;         /* 12+n cleanup (cleanup_new.py:243-251), 10+2n harvest
                                    (harvest_new.py:215-222), 2n+7 selfdrive obs length    */
  uint32_t num_int_metrics;      /* see CE_MI_* / CE_MIA_*                                  */
  uint32_t num_f64_metrics;      /* see CE_MF_*                                            */
  uint32_t obs_env_stride;       /* bytes between consecutive envs in `obs` (n * obs_agent_stride) */
  uint32_t rng_words;            /* CE_RNG_WORDS_GRID / CE_RNG_WORDS_SELFDRIVE / CE_RNG_WORDS_COUNTER */
  uint32_t grid_env_stride;      /* bytes between consecutive envs in `grid`                  */
  uint32_t grid_row_stride;      /* bytes between consecutive map rows inside an env's `grid` slice: the map
                                    is stored with the 7-cell view border on every side (32 cleanup, 52 harvest) */
  uint32_t grid_origin;          /* byte offset of cell (0, 0) inside the slice (7 * grid_row_stride + 7)   */
  uint32_t obs_row_stride;       /* bytes between consecutive rows of a 15x15x3 view (48: rows are
                                    pitched to 16 pixels so a 4-pixel store unit never straddles a row) */

  /* ---- persistent env state (host copies via ce_download / ce_upload) ---- */
  uint8_t* grid;        /* grid kinds: 32 bytes per env (grid_env_stride) — one "present" bit per apple cell (bits 0..)
                           and per waste cell (bits 128..) in row-major cell order, bit 255 = blank map; the rest
                           of the map is static.  ce_download / ce_upload("grid") convert from / to the padded
                           IMAGE: cell (r, c) of an env at grid_origin + r*grid_row_stride + c, image stride
                           round16((grid_h + 14) * grid_row_stride), cell codes CE_CELL_*, border bytes 0.
                           Feature kinds: CE_FEAT_STATE_BYTES per env instead (grid_env_stride) —
                           u16 apple_stamp[CE_FEAT_APPLE_SLOTS], u16 waste_stamp[CE_FEAT_WASTE_SLOTS], u32 next_apple_stamp,
                           u32 next_waste_stamp: the stamp of a cell is its rank in the reference's current_apple_points /
                           current_waste_points list (argmin ties break by list order), 0xffff = absent            */
  uint8_t* agents;      /* [E][n][4]  row, col, orientation (UP0 RIGHT1 DOWN2 LEFT3,
                                      Agent.py:18-23), 0                                   */
  uint8_t* spawn_perm;  /* [E][20]    persistent shuffled spawn list (map_env.py:821) as
                                      indices into the static P-cell table                 */
  uint8_t* waste_perm;  /* [E][119]   cleanup only: persistent shuffled waste list
                                      (cleanup_new.py:339) as indices into the static table */
  uint32_t* rng;        /* [E][CE_RNG_WORDS_*] numpy legacy MT19937 key[624], pos, pad[3]
                                      (+ Python `random` MT for selfdrive, second 628-word block) */
  int32_t* timestep;    /* [E]        MapEnv.timesteps                                      */
  double* theta;        /* [E]        contract parameter of the episode                     */
  double* sd_state;     /* selfdrive: [E][CE_SD_STATE_DOUBLES(n)]                           */

  /* ---- per-step outputs ---- */
  uint8_t* obs;          /* pitched image stack: pixel (e, a, i, j, ch) at e*obs_env_stride + a*obs_agent_stride
                            + i*obs_row_stride + 3*j + ch; a zero-copy [E][n][15][15][3] strided view */
  double* obs_f64;       /* selfdrive: [E][n][2n+7]                                         */
  int32_t* base_reward;  /* [E][n]  MapEnv reward before the contract (ints, Agent.py:87)   */
  double* reward;        /* [E][n]  reward after the contract transfer
                                    (two_stage_train.py:69-90); == base when no contract    */
  uint8_t* done;         /* [E]     dones['__all__']; selfdrive also [E][n+1] in done_agents */
  uint8_t* done_agents;  /* selfdrive: [E][n] per-agent done                                */
  uint8_t* info;         /* [E][n][2] grid: {eaten_apples, cleaned_squares|eaten_close_apples}
                                      selfdrive: {just_passed, 0}                           */
  int16_t* features;     /* [E][n][num_features] grid families: infos[k]['feature_obs'],
                                      all entries are small integers                        */
  /* ---- metrics (reference env.metrics, cleanup_new.py:186-188 / harvest_new.py:152-156) */
  int64_t* int_metrics;  /* [E][num_int_metrics]  running episode                           */
  double* f64_metrics;   /* [E][num_f64_metrics]  running episode                           */
  int64_t* final_int_metrics; /* same layout, latched at the step that returned done        */
  double* final_f64_metrics;
  uint32_t* error_flags; /* [E] per-env fault bits (CE_FAULT_*), sticky until the env is reset    */
  uint8_t* beam_map;     /* grid kinds: [E][grid_h][grid_w] CE_BEAM_* of the last step's beams (a later beam of the
                            shuffled firing order overwrites an earlier one, as the reference's beam_pos list does);
                            written only under CE_FLAG_BEAM_TRACE, zeroed by a reset                        */
  double* sd_info;       /* selfdrive: [E][2] per-step infos the reference attaches to the FIRST acting key of the step
                            (…accelerate.py:183-189, update_infos :127-149, crash branch :203-204): {ambulance_rank,
                            ambulance_dist_to_front}; every other key carries 0.0 there.  Together with `info`
                            ({just_passed, is_crashed}) this is the whole infos dict of a selfdrive step               */
  uint8_t* actions_taken; /* grid kinds: [E][n] the action ids the last ce_step_policy launch derived from the policy's output
                             (ABI 3; untouched by every other entry point)                                             */
} ce_buffers;

#define CE_BEAM_NONE 0
#define CE_BEAM_FIRE 1  /* 'F' */
#define CE_BEAM_CLEAN 2 /* 'C' */

/* cell codes of `grid` (reference world_map chars) */
#define CE_CELL_EMPTY 0 /* ' ' */
#define CE_CELL_WALL 1  /* '@' */
#define CE_CELL_APPLE 2 /* 'A' */
#define CE_CELL_WASTE 3 /* 'H' */
#define CE_CELL_RIVER 4 /* 'R' */
#define CE_CELL_STREAM 5 /* 'S' */

/* int_metrics layout: 4 globals then 4 per-agent blocks of n */
#define CE_MI_TOTAL_APPLES_EATEN 0
#define CE_MI_RAW_ENV_REWARDS 1
#define CE_MI_DIRT_CLEANED 2           /* cleanup */
#define CE_MI_LOW_DENSITY_APPLES 3     /* harvest 'low_density_apples_eaten' */
#define CE_MI_GLOBALS 4
#define CE_MIA_A 0  /* per agent: cleanup 'a{i}-waste_cleaned', harvest 'a{i}-apples_consumed' */
#define CE_MIA_B 1  /* per agent: harvest 'a{i}-close_apples_consumed'                          */
#define CE_MIA_SUM_R 2   /* per agent: sum_t r_i,t           (total_reward_dict)                */
#define CE_MIA_SUM_TR 3  /* per agent: sum_t t*r_i,t                                           */
#define CE_MI_COUNT(n) (CE_MI_GLOBALS + 4 * (n))
#define CE_MI_AGENT(n, which, i) (CE_MI_GLOBALS + (which) * (n) + (i))

/* f64_metrics layout */
#define CE_MF_TRANSFERS 0              /* metrics['transfers'] two_stage_train.py:92 */
#define CE_MF_EQUALITY 1               /* valid in final_* only (cleanup_new.py:422-434) */
#define CE_MF_SUSTAINABILITY 2         /* (cleanup_new.py:436-445) */
#define CE_MF_TRANSFER_EQUALITY 3      /* two_stage_train.py:97-99 */
#define CE_MF_TRANSFER_SUSTAINABILITY 4
#define CE_MF_GLOBALS 5
#define CE_MFA_SUM_R 0   /* per agent: sum of transferred rewards  */
#define CE_MFA_SUM_TR 1  /* per agent: sum of t * transferred reward */
#define CE_MF_AGENT(n, which, i) (CE_MF_GLOBALS + (which) * (n) + (i))
/* Float twins of the integer reward accumulators, maintained under CE_FLAG_INEQUITY_AVERSE only: there the env's own
 * rewards are floats, and the reference appends / sums THOSE (cleanup_new.py:227-234, harvest_new.py:200-205,
 * compute_equality / compute_sustainability cleanup_new.py:422-445), in step order.  With the flag set, equality and
 * sustainability are computed from these and metrics['raw_env_rewards'] is CE_MF_RAW_ENV_REWARDS_F; the integer
 * accumulators (CE_MI_RAW_ENV_REWARDS, CE_MIA_SUM_R / SUM_TR) keep summing the integer rewards before the aversion. */
#define CE_MF_RAW_ENV_REWARDS_F(n) (CE_MF_GLOBALS + 2 * (n))
#define CE_MFA_BASE_SUM_R 0   /* per agent: sum_t r_i,t of the env's (float) rewards  */
#define CE_MFA_BASE_SUM_TR 1  /* per agent: sum_t t * r_i,t                           */
#define CE_MF_BASE_AGENT(n, which, i) (CE_MF_GLOBALS + 2 * (n) + 1 + (which) * (n) + (i))
#define CE_MF_COUNT(n) (CE_MF_GLOBALS + 4 * (n) + 1)

/* selfdrive state block: pos[n], vel[n], dist_to_front[n], done[n] (0/1), done_all,
 * n_crossed, crossed[n] (agent indices in crossing order), transfers metric */
#define CE_SD_STATE_DOUBLES(n) (5 * (n) + 3)

#define CE_FEAT_APPLE_SLOTS 160u
#define CE_FEAT_WASTE_SLOTS 120u
#define CE_FEAT_STATE_BYTES (2u * (CE_FEAT_APPLE_SLOTS + CE_FEAT_WASTE_SLOTS) + 8u)
#define CE_FEAT_ABSENT 0xffffu
#define CE_RNG_WORDS_GRID 628u       /* key[624], pos, 3 pad words (16-byte aligned rows)   */
#define CE_RNG_WORDS_SELFDRIVE 1256u /* numpy MT block then Python `random` MT block        */
#define CE_RNG_WORDS_COUNTER 4u      /* CE_FLAG_RNG_COUNTER: key0, key1, generation, pad    */
#define CE_RNG_COUNTER_GEN 512u      /* stream words per generation in that mode            */

#define CE_FAULT_BAD_ACTION 0x1u     /* action id outside the family's table (Agent.py:161,198)
                                        — the reference raises KeyError                     */
#define CE_FAULT_BAD_GRID 0x8u        /* ce_upload("grid"): a cell differs from the static map where only apples / waste may vary */
#define CE_FAULT_NO_SPAWN 0x2u       /* map_env.py:826 assertion                             */
#define CE_FAULT_STEP_AFTER_DONE 0x4u /* selfdrive step after __all__ (…accelerate.py:154-167
                                        raises AttributeError in the reference)              */

typedef struct ce_engine* ce_handle;

/* Library / device probe.  Never fails loudly: returns the ABI version. */
int ce_abi_version(void);
/* Number of gfx950 devices visible, or a negative CE_E* code. */
int ce_device_count(void);

/* Replaces: Cls(**config) in utils/env_creator_functions.py:12-36 +
 * SeparateContractSubgameStage.__init__ (two_stage_train.py:152-157), for E replicas. */
int ce_create(const ce_config* cfg, ce_handle* out);
int ce_destroy(ce_handle h);

/* Replaces: wrapping an already constructed base env in SeparateContractSubgameStage
 * (two_stage_train.py:34-44,152-157): switches the fused contract epilogue of an existing
 * handle on/off and sets the contract space + null_prob.  Takes effect at the next reset/step. */
int ce_set_contract(ce_handle h, uint32_t contract, double contract_low, double contract_high, double null_prob);
/* flips run-time flags of a live handle: flags = (flags & ~mask) | (value & mask); only CE_FLAG_AUTO_RESET,
 * CE_FLAG_EXTERNAL_THETA and CE_FLAG_BEAM_TRACE may change after ce_create */
int ce_set_flags(ce_handle h, uint32_t mask, uint32_t value);

/* Replaces: np.random.seed(s) (+ random.seed(s)) followed by CONSTRUCTING the env
 * (MapEnv.__init__ -> setup_agents consumes RNG: map_env.py:131,816-832; CleanupEnv then
 * duplicates the spawn list, cleanup_new.py:114-115).  seeds: host pointer [E] (or NULL:
 * env b gets seed0 + env_index_base + b).  mask: host pointer [E] of 0/1 (NULL = all).
 * mode: CE_SEED_RESEED re-seeds both generators (np.random.seed / MapEnv.seed map_env.py:344-345,
 * random.seed); CE_SEED_CONSTRUCT replays the constructor's RNG use on the CURRENT generator
 * state; the usual "seed, then construct" is CE_SEED_RESEED | CE_SEED_CONSTRUCT. */
#define CE_SEED_RESEED 1
#define CE_SEED_CONSTRUCT 2
int ce_seed(ce_handle h, const uint64_t* seeds, uint64_t seed0, const uint8_t* mask, int mode);

/* Replaces: SeparateContractSubgameStage.reset (two_stage_train.py:159-187) ->
 * CleanupEnv/HarvestEnv.reset -> MapEnv.reset (map_env.py:306-342) /
 * SelfAcceleratingCarEnv.reset (…accelerate.py:49-79).  mask as above (host pointer).
 * Stream ordering: step / rollout launches issued later on a DIFFERENT stream wait for this reset by themselves (one
 * hipStreamWaitEvent per stream and reset) — "reset, then roll out env slices on side streams" needs no host sync.
 * Every other cross-stream dependency (the same envs stepped on two streams in turn) is the caller's to order. */
int ce_reset(ce_handle h, const uint8_t* mask, void* stream);

/* Replaces: SeparateContractEnv.step (two_stage_train.py:62-121) -> CleanupEnv.step /
 * HarvestEnv.step -> MapEnv.step (map_env.py:216-304) + contract.compute_transfer.
 * actions: DEVICE pointer. grid families: uint8 [E][n] action ids (Agent.py:8-16,161,198).
 * selfdrive: float [E][n] accelerations; active: DEVICE uint8 [E][n] (NULL = agents that are
 * not done act, which is what RLlib sends).  Asynchronous on `stream` (hipStream_t). */
int ce_step(ce_handle h, const void* actions, const uint8_t* active, void* stream);

/* ce_step restricted to the env slice [env_begin, env_begin + env_count).  `actions` / `active` still
 * point at the FULL [E][n] planes (the slice indexes into them).  Slices are independent: stepping
 * disjoint slices on different HIP streams lets one slice's tail overlap another's head (and, in an RL
 * loop, one slice's policy inference overlap the other slice's env step — double-buffered sampling).
 * ce_step / ce_step_range are one plain kernel launch on the caller's stream (no host synchronisation, no allocation), so a
 * caller may capture them into a hipGraph together with its own kernels (bench.py's closed loop does: one graph per slice).
 * Capture only on a stream that has already issued a step since the last ce_reset: the first step after a reset on another
 * stream inserts a one-time hipStreamWaitEvent (see ce_reset), which does not belong in a capture. */
int ce_step_range(ce_handle h, const void* actions, const uint8_t* active, uint32_t env_begin, uint32_t env_count,
                  void* stream);

/* ce_step_range with the action SELECTION fused into the step kernel's action load: `policy_out` is what the caller's policy
 * kernel left on the device, and a sampler tick is "one policy kernel + one step launch" with no argmax / modulo / cast
 * kernels in between (bench.py closed_loop; the sampler loop this stands in for: utils/ray_config_utils.py:126-214).
 *   CE_POLICY_BYTES_MOD   uint8 [E][n]: action = byte mod |A|  (|A| = 8 cleanup / 7 harvest, + 1 with CE_FLAG_FIRING_ENABLED)
 *   CE_POLICY_ARGMAX_F32  float [E][n][|A|] scores: action = index of the first maximum (NaNs never win) — exact categorical
 *                         sampling when the policy adds Gumbel noise to its logits, greedy otherwise
 * The action ids taken are written to ce_buffers.actions_taken (the sampler needs them for its batch).  Grid kinds only
 * (CE_EINVAL otherwise); slices, streams and graph capture exactly as ce_step_range. */
#define CE_POLICY_BYTES_MOD 1
#define CE_POLICY_ARGMAX_F32 2
/*   CE_POLICY_AHEAD_NOISE uint8 [E][n] noise plane, READ AND WRITTEN by the launch (round 5; an addition, no layout changed: ABI 3): noise[e][a] += green channel of view
 *                         pixel (6, 7) — the cell in front of the agent — of the observation the previous step or reset left in
 *                         ce_buffers.obs, modulo 256; action = the new byte mod |A|.  This is bench.py's closed-loop policy
 *                         evaluated inside the step kernel's action load: the loop observation -> action -> step closes on the
 *                         device with ONE launch per env slice and tick.  A policy that is a network keeps its own kernel and
 *                         hands its output over with one of the two modes above. */
#define CE_POLICY_AHEAD_NOISE 3
/* (policy_out is not const: CE_POLICY_AHEAD_NOISE updates the plane in place; the other two modes only read it) */
int ce_step_policy(ce_handle h, void* policy_out, uint32_t mode, uint32_t env_begin, uint32_t env_count, void* stream);
/* ce_step_policy over all envs as `num_slices` contiguous env slices, slice i launched on streams[i] (NULL = the null stream): the
 * launch loop of a sampler tick in C — ONE host call per tick (ce_rollout's slicing rule; round 5 addition). */
int ce_step_policy_sliced(ce_handle h, void* policy_out, uint32_t mode, uint32_t num_slices, void* const* streams);

/* Launch loop in C for pre-supplied actions (benchmarks, random-policy rollouts): `num_steps` consecutive
 * steps over all envs, each issued as `num_slices` ce_step_range launches on streams[0..num_slices-1] (NULL =
 * all on the null stream).  actions: DEVICE pointer to [num_steps][E][n] planes.  Every step is still its own
 * launch reading its own action plane; only the host-side loop moves out of the interpreter. */
int ce_rollout(ce_handle h, const void* actions, uint32_t num_steps, uint32_t num_slices, void* const* streams);

/* Trajectory arrays of a fused rollout (all DEVICE pointers, caller-owned; each holds `num_planes` planes with the
 * layout of the per-step buffer of the same name in ce_buffers: obs [P][E][obs_env_stride], base_reward [P][E][n], ...).
 * A NULL array means "not wanted as a trajectory": that output is written to the handle's own per-step buffer instead,
 * every step, and holds the last step's values on return — exactly what num_steps ce_step calls leave there.
 * The arrays are caller-owned and the library cannot see their sizes: each non-NULL array MUST hold num_planes planes of
 * THIS handle's E and n.  The struct says which batch they were allocated for (num_envs, num_agents) and a mismatch is
 * refused; a caller that states the right batch but allocates less is still on its own. */
typedef struct ce_traj {
  uint32_t num_planes;   /* planes per array; step s of a call goes to plane (first_plane + s) % num_planes   */
  uint32_t first_plane;
  uint32_t num_envs;     /* ABI 4: the batch the arrays were sized for — must equal the handle's E and n, or     */
  uint32_t num_agents;   /*        ce_rollout_fused returns CE_EINVAL before a byte is written                  */
  uint8_t* obs;          /* grid kinds                                                                         */
  double* obs_f64;       /* selfdrive                                                                          */
  int32_t* base_reward;
  double* reward;
  uint8_t* done;
  uint8_t* done_agents;  /* selfdrive                                                                          */
  uint8_t* info;
  int16_t* features;     /* grid / feature kinds                                                               */
  double* sd_info;       /* selfdrive                                                                          */
} ce_traj;

/* Fused multi-step rollout for pre-supplied actions: ONE launch per `steps_per_launch` consecutive env-steps.  Each
 * env's state (map, agent table, persistent shuffled lists, MT19937) is loaded once per launch, stays on chip for the
 * steps of that launch and is written back once; every step still reads its own action plane (actions: DEVICE pointer
 * to [num_steps][E][n]) and writes all of its per-step outputs (observation, rewards, infos, features, done; in-launch
 * auto-reset included) to its plane of `traj` (NULL = the handle's per-step buffers).  State, outputs, metrics and RNG
 * streams after the call are bit-identical to num_steps ce_step calls.  steps_per_launch = 0 means one launch for all
 * steps.  Reference callers that roll whole episodes per call: run_solver.py:35-65, two_stage_train.py:290-333.
 * Selfdrive: agents that are not done act (the `active = NULL` behaviour of ce_step).
 * num_slices / streams as in ce_rollout: the env axis is cut into num_slices contiguous slices, slice i is launched on
 * streams[i] (NULL = everything on the null stream).  Envs are independent, so results do not depend on the slicing; a
 * launch's waves all run equally long, and with one launch in flight the chip idles through the last, partly filled
 * round of wave slots — several slices in flight fill one slice's tail with the next launch of another. */
int ce_rollout_fused(ce_handle h, const void* actions, uint32_t num_steps, uint32_t steps_per_launch,
                     const ce_traj* traj, uint32_t num_slices, void* const* streams);

/* Same as ce_step with HOST action / active pointers: they are copied to an engine-owned
 * staging buffer on `stream` first (the per-env adapters use this). */
int ce_step_host(ce_handle h, const void* host_actions, const uint8_t* host_active, void* stream);

/* Synthetic uniform i.i.d. actions for benchmarks, generated on device by a counter-based
 * hash keyed (key, global env index, t, agent) — reproducible on host (ce_synth_action_host).
 * out: DEVICE pointer, uint8 [T][E][n] (grid) or float [T][E][n] (selfdrive). */
int ce_synth_actions(ce_handle h, uint64_t key, uint32_t t0, uint32_t T, void* out, void* stream);
uint32_t ce_synth_action_host(uint64_t key, uint64_t env_index, uint32_t t, uint32_t agent, uint32_t num_actions);

int ce_get_buffers(ce_handle h, ce_buffers* out);
int ce_synchronize(ce_handle h, void* stream);

/* Host copies (stream-synchronous helpers for tests / adapters without torch).
 * field names: "grid","agents","spawn_perm","waste_perm","rng","timestep","theta","sd_state",
 * "obs","obs_f64","base_reward","reward","done","done_agents","info","features",
 * "int_metrics","f64_metrics","final_int_metrics","final_f64_metrics","error_flags","beam_map","sd_info","actions_taken".
 * env_begin/env_count select a slice of the env axis; dst/src are host pointers. */
int ce_download(ce_handle h, const char* field, uint32_t env_begin, uint32_t env_count, void* dst, uint64_t dst_bytes);
int ce_upload(ce_handle h, const char* field, uint32_t env_begin, uint32_t env_count, const void* src, uint64_t src_bytes);

/* Replaces: MapEnv.global_view() (map_env.py:394-395) / get_global_obs (cleanup_new.py:299-300) — the whole colour map of
 * each env of the slice with the agents a step painted on it (agent order, the later agent wins a shared cell; none right
 * after a reset, as the reference's world_map_color), what JointEnv hands its centralised agent under `global_obs`
 * (two_stage_train.py:531,572-586).  out: DEVICE pointer, uint8 [env_count][grid_h][grid_w][3], dense; the reference's
 * float image is out / 255.  Asynchronous on `stream`, ordered after a ce_reset on another stream like a step.  Grid kinds
 * only.  A launch of its own, issued only by callers that want the view: rollouts do not pay for it.  ABI 4. */
int ce_global_view(ce_handle h, uint32_t env_begin, uint32_t env_count, uint8_t* out, void* stream);

/* ---- one-call state snapshot (ABI 4) ----
 * The reference never checkpoints env state; a batched engine that holds thousands of episodes must be able to.
 * ce_get_state writes ONE self-describing blob to a host buffer: a ce_state_header, a directory of ce_state_field entries,
 * then every persistent field of the handle's kind in its device layout (16-byte aligned) — for the grid kinds the presence
 * rows, agent table, persistent shuffled lists, generator rows, timestep, theta, running and latched metrics, done and fault
 * flags; selfdrive: sd_state, both generator blocks, ...; with CE_STATE_OUTPUTS also the last step's outputs (observations,
 * rewards, infos, feature rows: what a sampler needs to choose the NEXT action after a restore).  ce_set_state restores it:
 * the header must agree with the handle on ABI, kind, E, n, map layout (hash of the ascii_map), contract, horizon, every
 * step-relevant flag, env_index_base and the float parameters — stepping on from a restored handle is then bit-identical to
 * the run that was saved; any disagreement is CE_EINVAL (ce_last_error names it) and leaves the handle untouched.  A caller
 * needs no field list: ce_state_bytes, one buffer, two calls.  Both synchronize the device. */
#define CE_STATE_MAGIC 0x54534543u /* "CEST" */
#define CE_STATE_OUTPUTS 0x1u      /* `what`: also the per-step output buffers */
typedef struct ce_state_header {
  uint32_t magic, abi_version, header_bytes, kind;
  uint32_t num_envs, num_agents, contract, flags;
  uint32_t horizon, what, num_fields, reserved;
  uint64_t env_index_base, layout_hash, total_bytes;
  double params[9]; /* contract_low, contract_high, null_prob, alpha, beta, low_bound, high_bound, start_vel, start_vel_ambulance */
} ce_state_header;
typedef struct ce_state_field {
  char name[24];      /* field name as for ce_download */
  uint64_t offset;    /* of env 0's row, from the start of the blob */
  uint64_t env_bytes; /* row size; the field is num_envs consecutive rows */
} ce_state_field;
int ce_state_bytes(ce_handle h, uint32_t what, uint64_t* bytes);
int ce_get_state(ce_handle h, uint32_t what, void* dst, uint64_t dst_bytes);
int ce_set_state(ce_handle h, const void* src, uint64_t src_bytes);

/* How much of the GPU's last-level cache (MI355X: the 256 MiB Infinity Cache) this handle may assume for itself.  Single-step
 * launches write their observations through the L2 while the handle's device memory (+ the action planes a ce_rollout call
 * reads) is within the budget — the views then leave the chip as they are produced instead of at the launch's end (+4 % on the
 * headline, +20 % closed loop) — and keep them nontemporal beyond it, where a write-through store is an HBM write of its own
 * (-30 %).  Default: 7/8 of the cache (224 MiB; CE_OBS_WT_MAX_BYTES in the environment overrides the default), i.e. the
 * handle alone on the GPU.  An integrator whose policy network, other handles or other ranks share the cache passes the share
 * left for this handle; 0 = never write through.  Takes effect at the next launch. */
int ce_set_cache_budget(ce_handle h, uint64_t bytes);

/* Several fields of one env slice in ONE call (the per-env adapters fetch a whole step result this way): the device is
 * synchronized once; small requests are gathered on the device into a staging buffer and leave in a single copy.
 * "grid" (which needs the expand kernel) is not accepted here.  Same slice / size rules as ce_download. */
typedef struct ce_field_req {
  const char* field;   /* field name as for ce_download                       */
  void* dst;           /* host pointer                                        */
  uint64_t dst_bytes;  /* capacity of dst: >= env_count * bytes per env        */
} ce_field_req;
/* (uses a per-handle staging buffer: like every entry point taking a handle, not to be called from two threads on the
 * same handle at once) */
int ce_download_many(ce_handle h, uint32_t env_begin, uint32_t env_count, const ce_field_req* reqs, uint32_t count);

/* ---- host boundary helpers: what a host-side sampler (the RLlib vector hook, contracts_amd/vector_env.py) needs to take a
 * tick's results off the device without a synchronize per field and without pageable staging ----
 * ce_host_alloc / ce_host_free: page-locked host memory (DMA target of the asynchronous copies below).
 * ce_download_async: the field's env slice -> dst on `stream`, no host synchronization: ordered after the launches already
 *   issued on that stream, complete once the caller has synchronized the stream (ce_synchronize).  dst should be page-locked.
 * ce_step_host_async: ce_step_range with a HOST action plane (uint8 [E][n], page-locked): the slice's bytes are copied on
 *   `stream` ahead of its step launch and the call returns at once — the plane must stay untouched until the stream has
 *   passed the copy (double-buffer it).
 * ce_obs_u8_to_f64: format conversion on the host, `threads` worker threads: the pitched uint8 observation block (as
 *   downloaded: ce_buffers.obs strides) -> the reference's float64 images value / 255 (cleanup_new.py:258,
 *   harvest_new.py:229), dense [envs][n][15][15][3].
 * ce_download_obs_f64: the observation leg of a tick in one call — the slice's views travel to `staging` (page-locked,
 *   env_count * obs_env_stride bytes) in `parts` copies issued back to back on `stream`, and each part is converted into
 *   `out` (float64, dense, as ce_obs_u8_to_f64) while the later ones are still on the wire; returns when `out` is complete.
 * The conversions run on a process-wide pool of worker threads started on first use (`threads` = how many take part). */
int ce_host_alloc(uint64_t bytes, void** out);
int ce_host_free(void* p);
int ce_download_async(ce_handle h, const char* field, uint32_t env_begin, uint32_t env_count, void* dst, uint64_t dst_bytes,
                      void* stream);
int ce_step_host_async(ce_handle h, const void* host_actions, uint32_t env_begin, uint32_t env_count, void* stream);
int ce_i16_to_f64(const int16_t* src, double* out, uint64_t count, uint32_t threads); /* feature rows -> feature_obs floats */
int ce_obs_u8_to_f64(const uint8_t* pitched, double* out, uint32_t num_envs, uint32_t num_agents, uint32_t obs_env_stride,
                     uint32_t obs_agent_stride, uint32_t obs_row_stride, uint32_t threads);
int ce_download_obs_f64(ce_handle h, uint32_t env_begin, uint32_t env_count, void* staging, uint64_t staging_bytes, double* out,
                        uint64_t out_bytes, uint32_t parts, uint32_t threads, void* stream);

/* Timing of the last N ce_step launches measured with HIP events on the launch stream
 * (bench.py roofline leg).  ce_timing_begin arms recording, ce_timing_end returns the mean
 * step-kernel duration in milliseconds and the number of launches measured. */
int ce_timing_begin(ce_handle h, void* stream);
int ce_timing_end(ce_handle h, void* stream, double* mean_ms, uint32_t* launches);

/* Device self-test of the wave primitives the kernels rely on (DPP reduction, parallel MT
 * twist, cross-lane list swap; bits 4, 5: Philox4x32-10 against the Random123 known-answer vectors and the counter-mode LDS
 * fill against per-block evaluation; bits 6, 7: the fixed-point draws of the waste-list shuffle and of the small shuffles against
 * the serial rejection-sampling walk, list contents and stream position).  failed_mask: bit i set = check i failed.  0 on success. */
int ce_selftest(int device, uint32_t* failed_mask);

/* 64-bit counter hash behind ce_synth_actions (selfdrive: action = ((hash>>40) / 2^24) * 0.2f - 0.1f) */
uint64_t ce_synth_hash_host(uint64_t key, uint64_t env_index, uint32_t t, uint32_t agent);

/* The ASCII layout the engine's static tables of a map kind are built from — the reference's CLEANUP_MAP
 * (cleanup_new.py:10-36) / HARVEST_MAP (harvest_new.py:10-27), which the feature kinds share: rows*cols bytes,
 * row-major, no terminators.  out == NULL only reports the shape.  CE_EINVAL for selfdrive (no map) or a short buffer. */
int ce_static_map(uint32_t kind, char* out, uint64_t out_bytes, uint32_t* rows, uint32_t* cols);

const char* ce_last_error(ce_handle h);

#ifdef __cplusplus
}
#endif
#endif /* CONTRACTS_ENGINE_H */
