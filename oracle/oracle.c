/*
 * oracle.c — CPU restatement of the reference rollout hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity checker for the HIP engine and the timed `cpu_baseline` leg of
 * bench.py.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may load it;
 * the product (contracts_amd/) never links, imports or calls anything in oracle/.
 *
 * It restates, in plain serial C and in the reference's own order of operations, the
 * Python algorithm of (paths relative to the reference root):
 *   environments/map_env.py      MapEnv.step/reset/update_moves/update_custom_moves/
 *                                update_map_fire/spawn_point/spawn_rotation/color_view
 *   environments/Agent.py        action tables, consume/hit/fire_beam, walkability
 *   environments/cleanup_new.py  CleanupEnv step/reset/spawn model/feature obs/metrics
 *   environments/harvest_new.py  HarvestEnv step/reset/spawn model/feature obs/metrics
 *   environments/self_driving_car_accelerate.py  SelfAcceleratingCarEnv
 *   contract/contract_list.py    the three contracts' compute_transfer
 *   environments/two_stage_train.py:62-121,159-187  reward-transfer wrapper, theta sampling
 * plus the third-party arithmetic those call: numpy's legacy RandomState (MT19937,
 * shuffle / random_sample / randint / uniform) and CPython's `random` (seed, random()).
 * numpy is a pinned-era dependency absent from the reference tree (requirements.yml pins
 * ray 2.2.0 / tf 2.11 => numpy 1.2x); its legacy stream is frozen by numpy policy and is
 * restated here from the published algorithm.
 *
 * Parity pin: tests/test_oracle_golden.py checks this file, step by step, against the
 * fixtures under tests/golden/ (.npz) that were produced by running the reference itself
 * (tests/golden/make_golden.py).  The reference has no tests of its own for this path.
 *
 * Data layout mirrors include/contracts_engine.h (ce_buffers) with host pointers.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/contracts_engine.h"

#define MAXN 10
#define VIEW 7
#define WIN 15

/* ------------------------------------------------------------------------- */
/* MT19937 — numpy legacy RandomState / CPython random                        */
/* ------------------------------------------------------------------------- */
typedef struct {
  uint32_t key[624];
  uint32_t pos;
  /* counter mode (CE_FLAG_RNG_COUNTER, contracts_engine.h — the engine's own stream, not the reference's): key[0..511] holds
   * generation `gen` of Philox4x32-10 blocks under (k0, k1); everything downstream of mt_next is shared with MT19937 */
  int ctr;
  uint32_t k0, k1, gen;
} mt_t;

/* numpy: RandomState.seed(int) -> mt19937_seed == init_genrand */
static void mt_init_genrand(mt_t* m, uint32_t s) {
  for (int i = 0; i < 624; i++) {
    m->key[i] = s;
    s = 1812433253u * (s ^ (s >> 30)) + (uint32_t)i + 1u;
  }
  m->pos = 624;
}

/* CPython random.seed(int): init_by_array(key = 32-bit chunks of abs(seed)) */
static void mt_init_by_array(mt_t* m, const uint32_t* init_key, int key_length) {
  mt_init_genrand(m, 19650218u);
  uint32_t* mt = m->key;
  int i = 1, j = 0;
  int k = (624 > key_length ? 624 : key_length);
  for (; k; k--) {
    mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525u)) + init_key[j] + (uint32_t)j;
    i++;
    j++;
    if (i >= 624) {
      mt[0] = mt[623];
      i = 1;
    }
    if (j >= key_length) j = 0;
  }
  for (k = 623; k; k--) {
    mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
    i++;
    if (i >= 624) {
      mt[0] = mt[623];
      i = 1;
    }
  }
  mt[0] = 0x80000000u;
  m->pos = 624;
}

/* Philox4x32-10 as published (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11; Random123
 * philox.h): ten rounds of (c0,c1,c2,c3) <- (hi(M1*c2)^c1^k0, lo(M1*c2), hi(M0*c0)^c3^k1, lo(M0*c0)), key += (W0, W1) */
static void philox4x32_10(uint32_t k0, uint32_t k1, uint32_t c[4]) {
  for (int r = 0; r < 10; r++) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    c[1] = (uint32_t)p1;
    c[3] = (uint32_t)p0;
    c[0] = n0;
    c[2] = n2;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}
static uint32_t rng_gen_words(const mt_t* m) { return m->ctr ? CE_RNG_COUNTER_GEN : 624u; }
/* counter mode: the next generation = blocks (q, gen, 0, 0), q = 0 .. 127 */
static void ctr_next_generation(mt_t* m) {
  m->gen += 1;
  for (uint32_t q = 0; q < CE_RNG_COUNTER_GEN / 4; q++) {
    uint32_t c[4] = {q, m->gen, 0, 0};
    philox4x32_10(m->k0, m->k1, c);
    memcpy(m->key + 4 * q, c, 16);
  }
  m->pos = 0;
}
/* ... every operation on an env (construct, reset, a step that is taken) opens a fresh generation and ends by dropping
 * what is left of it */
static void rng_begin_op(mt_t* m) {
  if (m->ctr && m->pos >= CE_RNG_COUNTER_GEN) ctr_next_generation(m);
}
static void rng_end_of_op(mt_t* m) {
  if (m->ctr) m->pos = CE_RNG_COUNTER_GEN;
}

static void mt_twist(mt_t* m) {
  if (m->ctr) {
    ctr_next_generation(m);
    return;
  }
  uint32_t* k = m->key;
  int i;
  uint32_t y;
  for (i = 0; i < 624 - 397; i++) {
    y = (k[i] & 0x80000000u) | (k[i + 1] & 0x7fffffffu);
    k[i] = k[i + 397] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
  }
  for (; i < 623; i++) {
    y = (k[i] & 0x80000000u) | (k[i + 1] & 0x7fffffffu);
    k[i] = k[i + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
  }
  y = (k[623] & 0x80000000u) | (k[0] & 0x7fffffffu);
  k[623] = k[396] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
  m->pos = 0;
}

static uint32_t mt_next(mt_t* m) {
  if (m->pos >= rng_gen_words(m)) mt_twist(m);
  uint32_t y = m->key[m->pos++];
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}

/* R2: random_sample / random.random(): two words -> 53-bit double */
static double mt_double(mt_t* m) {
  uint32_t a = mt_next(m) >> 5, b = mt_next(m) >> 6;
  return (a * 67108864.0 + b) / 9007199254740992.0;
}

/* R1 helper: legacy random_interval(max): masked rejection, one word per attempt */
static uint32_t mt_interval(mt_t* m, uint32_t max) {
  if (max == 0) return 0;
  uint32_t mask = max;
  mask |= mask >> 1;
  mask |= mask >> 2;
  mask |= mask >> 4;
  mask |= mask >> 8;
  mask |= mask >> 16;
  uint32_t v;
  while ((v = (mt_next(m) & mask)) > max) {
  }
  return v;
}

/* ------------------------------------------------------------------------- */
/* static maps (cleanup_new.py:10-36, harvest_new.py:10-27)                   */
/* ------------------------------------------------------------------------- */
static const char* CLEANUP_MAP[25] = {
    "@@@@@@@@@@@@@@@@@@", "@RRRRRR     BBBBB@", "@HHHHHH      BBBB@", "@RRRRRR     BBBBB@", "@RRRRR  P    BBBB@",
    "@RRRRR    P BBBBB@", "@HHHHH       BBBB@", "@RRRRR      BBBBB@", "@HHHHHHSSSSSSBBBB@", "@HHHHHHSSSSSSBBBB@",
    "@RRRRR   P P BBBB@", "@HHHHH   P  BBBBB@", "@RRRRRR    P BBBB@", "@HHHHHH P   BBBBB@", "@RRRRR       BBBB@",
    "@HHHH    P  BBBBB@", "@RRRRR       BBBB@", "@HHHHH  P P BBBBB@", "@RRRRR       BBBB@", "@HHHH       BBBBB@",
    "@RRRRR       BBBB@", "@HHHHH      BBBBB@", "@RRRRR       BBBB@", "@HHHH       BBBBB@", "@@@@@@@@@@@@@@@@@@"};

static const char* HARVEST_MAP[16] = {
    "@@@@@@@@@@@@@@@@@@@@@@@@@@@@@@@@@@@@@@", "@ P   P      A    P AAAAA    P  A P  @",
    "@  P     A P AA    P    AAA    A  A  @", "@     A AAA  AAA    A    A AA AAAA   @",
    "@ A  AAA A    A  A AAA  A  A   A A   @", "@AAA  A A    A  AAA A  AAA        A P@",
    "@ A A  AAA  AAA  A A    A AA   AA AA @", "@  A A  AAA    A A  AAA    AAA  A    @",
    "@   AAA  A      AAA  A    AAAA       @", "@ P  A       A  A AAA    A  A      P @",
    "@A  AAA  A  A  AAA A    AAAA     P   @", "@    A A   AAA  A A      A AA   A  P @",
    "@     AAA   A A  AAA      AA   AAA P @", "@ A    A     AAA  A  P          A    @",
    "@       P     A         P  P P     P @", "@@@@@@@@@@@@@@@@@@@@@@@@@@@@@@@@@@@@@@"};

/* colour LUT: DEFAULT_COLOURS map_env.py:24-42 + CLEANUP_COLORS cleanup_new.py:42-47 */
static const uint8_t CELL_RGB[6][3] = {{0, 0, 0}, {180, 180, 180}, {0, 255, 0}, {99, 156, 194}, {113, 75, 24}, {113, 75, 24}};
static const uint8_t AGENT_RGB[9][3] = {{0, 0, 255},     {2, 81, 154},   {204, 0, 204},   {216, 30, 54}, {254, 151, 0},
                                        {100, 255, 255}, {99, 99, 255}, {250, 204, 255}, {238, 223, 16}};

enum { UP = 0, RIGHT = 1, DOWN = 2, LEFT = 3 }; /* Agent.py:18-23 int codes */
/* ORIENTATIONS vectors, map_env.py:22 */
static const int ORI_VEC[4][2] = {{-1, 0}, {0, 1}, {1, 0}, {0, -1}};
/* spawn_rotation: randint index into list(ORIENTATIONS.keys()) = LEFT,RIGHT,UP,DOWN (map_env.py:22,829-832) */
static const int SPAWN_ROT[4] = {LEFT, RIGHT, UP, DOWN};
/* _MAP_ENV_ACTIONS vectors, map_env.py:11-16: ids 0..4 = LEFT,RIGHT,UP,DOWN,STAY */
static const int MOVE_VEC[5][2] = {{0, -1}, {0, 1}, {-1, 0}, {1, 0}, {0, 0}};

/* rotate_left: TURN_COUNTERCLOCKWISE [[0,-1],[1,0]] . v ; rotate_right: [[0,1],[-1,0]] . v (map_env.py:17-18,855-859) */
static void rot_left(const int* v, int* o) {
  o[0] = 0 * v[0] + -1 * v[1];
  o[1] = 1 * v[0] + 0 * v[1];
}
static void rot_right(const int* v, int* o) {
  o[0] = 0 * v[0] + 1 * v[1];
  o[1] = -1 * v[0] + 0 * v[1];
}
/* rotate_action map_env.py:844-853 */
static void rotate_action(const int* v, int orient, int* o) {
  if (orient == UP) {
    o[0] = v[0];
    o[1] = v[1];
  } else if (orient == LEFT) {
    rot_left(v, o);
  } else if (orient == RIGHT) {
    rot_right(v, o);
  } else {
    int t[2];
    rot_left(v, t);
    rot_left(t, o);
  }
}
/* update_rotation map_env.py:862-880 */
static int update_rotation(int ccw, int o) {
  if (ccw) {
    if (o == LEFT) return DOWN;
    if (o == DOWN) return RIGHT;
    if (o == RIGHT) return UP;
    return LEFT;
  }
  if (o == LEFT) return UP;
  if (o == UP) return RIGHT;
  if (o == RIGHT) return DOWN;
  return LEFT;
}

/* ------------------------------------------------------------------------- */
/* per-env state                                                              */
/* ------------------------------------------------------------------------- */
typedef struct {
  int row, col;
} pt_t;

typedef struct {
  mt_t np_rng; /* process-global np.random of the reference worker */
  mt_t py_rng; /* process-global `random` (selfdrive reset only)    */
  /* grid families */
  uint8_t grid[16 * 38];
  pt_t pos[MAXN];
  int orient[MAXN];
  int reward_this_turn[MAXN];
  int cleaned_squares[MAXN];
  int n_spawn;
  uint8_t spawn_perm[20]; /* index into static spawn table */
  uint8_t waste_perm[119];
  int timesteps;
  double theta;
  /* feature-vector envs: list-order stamps of current_apple_points / current_waste_points */
  uint16_t apple_stamp[CE_FEAT_APPLE_SLOTS], waste_stamp[CE_FEAT_WASTE_SLOTS];
  uint32_t next_apple_stamp, next_waste_stamp;
  /* apples present when the step was entered == reference current_apple_points */
  uint8_t entry_apples[16 * 38];
  /* selfdrive */
  double sd_pos[MAXN], sd_vel[MAXN], sd_dtf[MAXN];
  int sd_done[MAXN], sd_done_all;
  int sd_ncross, sd_cross[MAXN];
} env_t;

typedef struct orc {
  ce_config cfg;
  int n, H, W, cells, kind;
  int n_apple, n_waste, n_spawn_static;
  pt_t apple_pts[160];
  pt_t waste_pts[119];
  pt_t spawn_pts[20];
  uint8_t is_wall[16 * 38];
  uint8_t base_cell[16 * 38]; /* reset-time code of each cell (walls + H/R/S, or apples for harvest) */
  env_t* envs;
  ce_buffers b; /* host arrays */
  char err[256];
} orc_t;

static int in_bounds(const orc_t* o, int r, int c) { return 0 <= r && r < o->H && 0 <= c && c < o->W; }

static int is_feat_kind(int kind);
/* MapEnv.__init__ / CleanupEnv.__init__ / HarvestEnv.__init__ over the layout (map_env.py:117-130, cleanup_new.py:107-128):
 * the shipped one, or the caller's `ascii_map` (ce_config.ascii_map, grid kinds) within the caps of contracts_engine.h */
static int build_static(orc_t* o) {
  const int cleanup_map = o->kind == CE_KIND_CLEANUP || o->kind == CE_KIND_CLEANUP_FEATURES; /* same maps as the feature envs */
  const char** map = cleanup_map ? CLEANUP_MAP : HARVEST_MAP;
  const char* custom = o->cfg.ascii_map;
  const int frame_h = cleanup_map ? 25 : 16, frame_w = cleanup_map ? 18 : 38;
  o->H = frame_h;
  o->W = frame_w;
  if (custom) {
    if (is_feat_kind(o->kind) || o->cfg.map_rows < 1 || o->cfg.map_cols < 1 || (int)o->cfg.map_rows > frame_h ||
        (int)o->cfg.map_cols > frame_w)
      return CE_EINVAL;
    o->H = (int)o->cfg.map_rows;
    o->W = (int)o->cfg.map_cols;
  }
  o->cells = o->H * o->W;
  o->n_apple = o->n_waste = o->n_spawn_static = 0;
  const int cap_apple = cleanup_map ? 103 : 155, cap_waste = 119, cap_spawn = cleanup_map ? 10 : 20;
  for (int r = 0; r < o->H; r++)
    for (int c = 0; c < o->W; c++) {
      char ch = custom ? custom[r * o->W + c] : map[r][c];
      if (custom && (r == 0 || c == 0 || r == o->H - 1 || c == o->W - 1) && ch != '@')
        return CE_EINVAL; /* a layout must be walled in, like the shipped ones: the reference indexes the map with whatever cell a
                             move names (Agent.return_valid_pos), so stepping over an open edge wraps around or raises there */
      if (custom) { /* caps first: the tables below are sized for the shipped layouts */
        if (ch == 'P' && o->n_spawn_static >= cap_spawn) return CE_EINVAL;
        if (ch == (cleanup_map ? 'B' : 'A') && o->n_apple >= cap_apple) return CE_EINVAL;
        if (cleanup_map && (ch == 'H' || ch == 'R') && o->n_waste >= cap_waste) return CE_EINVAL;
        if (!(ch == '@' || ch == ' ' || ch == 'P' || (cleanup_map ? (ch == 'B' || ch == 'H' || ch == 'R' || ch == 'S') : ch == 'A')))
          return CE_EINVAL;
      }
      int idx = r * o->W + c;
      o->is_wall[idx] = ch == '@';
      o->base_cell[idx] = ch == '@' ? CE_CELL_WALL : CE_CELL_EMPTY;
      if (ch == 'P') o->spawn_pts[o->n_spawn_static++] = (pt_t){r, c};
      if (cleanup_map) {
        if (ch == 'B') o->apple_pts[o->n_apple++] = (pt_t){r, c};
        if (ch == 'H' || ch == 'R') o->waste_pts[o->n_waste++] = (pt_t){r, c};
        if (ch == 'H') o->base_cell[idx] = CE_CELL_WASTE;
        if (ch == 'R') o->base_cell[idx] = CE_CELL_RIVER;
        if (ch == 'S') o->base_cell[idx] = CE_CELL_STREAM;
      } else {
        if (ch == 'A') {
          o->apple_pts[o->n_apple++] = (pt_t){r, c};
          o->base_cell[idx] = CE_CELL_APPLE;
        }
      }
    }
  if (cleanup_map) {
    /* cleanup_new.py:114-115 appends the P cells a second time */
    for (int i = 0; i < o->n_spawn_static; i++) o->spawn_pts[o->n_spawn_static + i] = o->spawn_pts[i];
  }
  if (custom && (o->n_spawn_static < 1 || o->n_apple < 1 || (cleanup_map && o->n_waste < 1) || o->n > o->n_spawn_static))
    return CE_EINVAL;
  return CE_OK;
}
/* length of the spawn list setup_agents shuffles: at construction the P cells; from the first reset on cleanup's doubled list */
static int spawn_len_ctor(const orc_t* o) { return o->n_spawn_static; }
static int spawn_len_reset(const orc_t* o) { return o->kind == CE_KIND_CLEANUP ? 2 * o->n_spawn_static : o->n_spawn_static; }

/* ------------------------------------------------------------------------- */
/* numpy list shuffle (untyped path of RandomState.shuffle)                   */
/* ------------------------------------------------------------------------- */
static void shuffle_u8(mt_t* m, uint8_t* x, int len) {
  for (int i = len - 1; i >= 1; i--) {
    uint32_t j = mt_interval(m, (uint32_t)i);
    uint8_t t = x[i];
    x[i] = x[j];
    x[j] = t;
  }
}

/* MapEnv.spawn_point map_env.py:816-827: shuffle the persistent list in place, pick the
 * LAST entry not occupied by an already placed agent. */
static int spawn_point(orc_t* o, env_t* e, int placed, int list_len, pt_t* out) {
  shuffle_u8(&e->np_rng, e->spawn_perm, list_len);
  int found = -1;
  for (int i = 0; i < list_len; i++) {
    pt_t p = o->spawn_pts[e->spawn_perm[i]];
    int taken = 0;
    for (int a = 0; a < placed; a++)
      if (e->pos[a].row == p.row && e->pos[a].col == p.col) taken = 1;
    if (!taken) found = i;
  }
  if (found < 0) return -1;
  *out = o->spawn_pts[e->spawn_perm[found]];
  return 0;
}

/* setup_agents cleanup_new.py:302-320 / harvest_new.py:132-141 */
static int setup_agents(orc_t* o, env_t* e, int list_len) {
  for (int i = 0; i < o->n; i++) {
    pt_t p;
    if (spawn_point(o, e, i, list_len, &p)) return -1;
    uint32_t r = mt_next(&e->np_rng) & 3u; /* R3 randint(4) */
    e->pos[i] = p;
    e->orient[i] = SPAWN_ROT[r];
    e->reward_this_turn[i] = 0;
    e->cleaned_squares[i] = 0;
  }
  return 0;
}

/* ------------------------------------------------------------------------- */
/* spawn models                                                               */
/* ------------------------------------------------------------------------- */
static int agent_at(const orc_t* o, const env_t* e, int r, int c) {
  for (int a = 0; a < o->n; a++)
    if (e->pos[a].row == r && e->pos[a].col == c) return 1;
  return 0;
}

/* CleanupEnv.compute_probabilities cleanup_new.py:351-368 */
static void cleanup_probs(const orc_t* o, const env_t* e, double* p_apple, double* p_waste) {
  int nH = 0;
  for (int i = 0; i < o->cells; i++) nH += e->grid[i] == CE_CELL_WASTE;
  double waste_density = 0;
  int potential = o->n_waste;
  if (potential > 0) {
    int free_area = potential - nH;
    waste_density = 1 - (double)free_area / (double)potential;
  }
  if (waste_density >= 0.4) {
    *p_apple = 0;
    *p_waste = 0;
  } else {
    *p_waste = 0.5;
    if (waste_density <= 0.0)
      *p_apple = 0.05;
    else
      *p_apple = (1 - (waste_density - 0.0) / (0.4 - 0.0)) * 0.05;
  }
}

/* CleanupEnv.custom_map_update -> spawn_apples_and_waste cleanup_new.py:294-349 */
static void cleanup_spawn(orc_t* o, env_t* e) {
  double p_apple, p_waste;
  cleanup_probs(o, e, &p_apple, &p_waste);
  double rnd[103 + 119];
  int total = o->n_apple + o->n_waste;
  for (int i = 0; i < total; i++) rnd[i] = mt_double(&e->np_rng);
  int r = 0;
  int ns = 0;
  int sp_idx[103 + 1];
  uint8_t sp_code[103 + 1];
  for (int i = 0; i < o->n_apple; i++) {
    pt_t p = o->apple_pts[i];
    int idx = p.row * o->W + p.col;
    if (!agent_at(o, e, p.row, p.col) && e->grid[idx] != CE_CELL_APPLE) {
      double u = rnd[r++];
      if (u < p_apple) {
        sp_idx[ns] = idx;
        sp_code[ns++] = CE_CELL_APPLE;
      }
    }
  }
  /* np.isclose(p_waste, 0): p_waste is exactly 0 or 0.5 */
  if (p_waste != 0) {
    shuffle_u8(&e->np_rng, e->waste_perm, o->n_waste);
    for (int i = 0; i < o->n_waste; i++) {
      pt_t p = o->waste_pts[e->waste_perm[i]];
      int idx = p.row * o->W + p.col;
      if (e->grid[idx] != CE_CELL_WASTE) {
        double u = rnd[r++];
        if (u < p_waste) {
          sp_idx[ns] = idx;
          sp_code[ns++] = CE_CELL_WASTE;
          break;
        }
      }
    }
  }
  for (int i = 0; i < ns; i++) e->grid[sp_idx[i]] = sp_code[i];
}

/* HarvestEnv.spawn_apples harvest_new.py:284-317 */
static void harvest_spawn(orc_t* o, env_t* e) {
  static const double SPAWN_PROB[4] = {0, 0.005, 0.02, 0.05};
  double rnd[160];
  for (int i = 0; i < o->n_apple; i++) rnd[i] = mt_double(&e->np_rng);
  int r = 0, ns = 0;
  int sp_idx[160];
  for (int i = 0; i < o->n_apple; i++) {
    pt_t p = o->apple_pts[i];
    int idx = p.row * o->W + p.col;
    if (!agent_at(o, e, p.row, p.col) && e->grid[idx] != CE_CELL_APPLE) {
      int num = 0;
      for (int j = -2; j <= 2; j++)
        for (int k = -2; k <= 2; k++)
          if (j * j + k * k <= 2) { /* `<= APPLE_RADIUS`, not radius^2: harvest_new.py:303 */
            int x = p.row + j, y = p.col + k;
            if (0 <= x && x < o->H && o->W > y && y >= 0)
              if (e->grid[x * o->W + y] == CE_CELL_APPLE) num++;
          }
      double prob = SPAWN_PROB[num < 3 ? num : 3];
      double u = rnd[r++];
      if (u < prob) sp_idx[ns++] = idx;
    }
  }
  for (int i = 0; i < ns; i++) e->grid[sp_idx[i]] = CE_CELL_APPLE;
}

static void custom_map_update(orc_t* o, env_t* e) {
  if (o->kind == CE_KIND_CLEANUP)
    cleanup_spawn(o, e);
  else
    harvest_spawn(o, e);
}

/* ------------------------------------------------------------------------- */
/* observation crop: color_view map_env.py:397-411                             */
/* ------------------------------------------------------------------------- */
static void cell_rgb_with_agents(const orc_t* o, const env_t* e, int r, int c, int paint_agents, uint8_t* rgb) {
  if (!in_bounds(o, r, c)) {
    rgb[0] = rgb[1] = rgb[2] = 0; /* zero padding of world_map_color */
    return;
  }
  const uint8_t* src = CELL_RGB[e->grid[r * o->W + c]];
  if (paint_agents) {
    /* painted in agent order, the later agent wins (map_env.py:257-261) */
    for (int a = 0; a < o->n; a++)
      if (e->pos[a].row == r && e->pos[a].col == c) src = AGENT_RGB[a];
  }
  rgb[0] = src[0];
  rgb[1] = src[1];
  rgb[2] = src[2];
}

static void write_obs(orc_t* o, env_t* e, int ei, int paint_agents) {
  for (int a = 0; a < o->n; a++) {
    uint8_t* out = o->b.obs + (size_t)ei * o->b.obs_env_stride + (size_t)a * (WIN * WIN * 3);
    int row = e->pos[a].row, col = e->pos[a].col;
    for (int i = 0; i < WIN; i++)
      for (int j = 0; j < WIN; j++) {
        /* m[a,b] = padded[row + a, col + b]  => world cell (row - 7 + a, col - 7 + b) */
        int ma, mb;
        switch (e->orient[a]) {
          case UP: ma = i; mb = j; break;
          case LEFT: ma = j; mb = WIN - 1 - i; break;         /* np.rot90(k=1)            */
          case DOWN: ma = WIN - 1 - i; mb = WIN - 1 - j; break; /* np.rot90(k=2)          */
          default: ma = WIN - 1 - j; mb = i; break;             /* RIGHT: rot90(k=1,(1,0)) */
        }
        cell_rgb_with_agents(o, e, row - VIEW + ma, col - VIEW + mb, paint_agents, out + (i * WIN + j) * 3);
      }
  }
}

/* ------------------------------------------------------------------------- */
/* update_moves map_env.py:483-676 (literal restatement)                       */
/* ------------------------------------------------------------------------- */
static int walkable(const orc_t* o, int r, int c) { return in_bounds(o, r, c) && !o->is_wall[r * o->W + c]; }

static void update_agent_pos(const orc_t* o, env_t* e, int a, pt_t np_) {
  if (walkable(o, np_.row, np_.col)) e->pos[a] = np_; /* Agent.update_agent_pos Agent.py:121-139 */
}

static int pt_eq(pt_t a, pt_t b) { return a.row == b.row && a.col == b.col; }

static int pos_occupied(const orc_t* o, const env_t* e, pt_t p) { /* `p in self.agent_pos` */
  for (int a = 0; a < o->n; a++)
    if (pt_eq(e->pos[a], p)) return 1;
  return 0;
}
static int by_pos_lookup(const orc_t* o, const pt_t* snap, pt_t p) { /* dict built in agent order: later wins */
  int id = -1;
  for (int a = 0; a < o->n; a++)
    if (pt_eq(snap[a], p)) id = a;
  return id;
}

static void update_moves(orc_t* o, env_t* e, const uint8_t* act) {
  int n = o->n;
  /* phase A */
  int slot_agent[MAXN], nslots = 0;
  pt_t slot_tgt[MAXN];
  for (int a = 0; a < n; a++) {
    int ac = act[a];
    if (ac <= 4) { /* MOVE_* / STAY */
      int rv[2];
      rotate_action(MOVE_VEC[ac], e->orient[a], rv);
      pt_t np_ = {e->pos[a].row + rv[0], e->pos[a].col + rv[1]};
      if (!walkable(o, np_.row, np_.col)) np_ = e->pos[a]; /* return_valid_pos Agent.py:111-119 */
      slot_agent[nslots] = a;
      slot_tgt[nslots++] = np_;
    } else if (ac == 5 || ac == 6) {
      e->orient[a] = update_rotation(ac == 6, e->orient[a]);
    }
  }
  pt_t by_pos[MAXN]; /* snapshot of positions the agent_by_pos dict was built from */
  memcpy(by_pos, e->pos, sizeof(by_pos));
  /* agent_moves: ordered dict keyed by agent, insertion order = slot order */
  int has_move[MAXN] = {0};
  pt_t move[MAXN];
  int order[MAXN], norder = 0; /* insertion order of keys */
  for (int s = 0; s < nslots; s++) {
    has_move[slot_agent[s]] = 1;
    move[slot_agent[s]] = slot_tgt[s];
    order[norder++] = slot_agent[s];
  }
  if (nslots == 0) return;

  /* shuffle the (agent, slot) pairs: np.random.shuffle(list) */
  {
    uint8_t idx[MAXN];
    for (int s = 0; s < nslots; s++) idx[s] = (uint8_t)s;
    shuffle_u8(&e->np_rng, idx, nslots);
    int sa[MAXN];
    pt_t st[MAXN];
    for (int s = 0; s < nslots; s++) {
      sa[s] = slot_agent[idx[s]];
      st[s] = slot_tgt[idx[s]];
    }
    memcpy(slot_agent, sa, sizeof(int) * nslots);
    memcpy(slot_tgt, st, sizeof(pt_t) * nslots);
  }
  /* np.unique(move_slots, axis=0): sorted unique rows, first index, counts */
  pt_t uniq[MAXN];
  int ufirst[MAXN], ucount[MAXN], nu = 0;
  for (int s = 0; s < nslots; s++) {
    int k;
    for (k = 0; k < nu; k++)
      if (pt_eq(uniq[k], slot_tgt[s])) break;
    if (k == nu) {
      uniq[nu] = slot_tgt[s];
      ufirst[nu] = s;
      ucount[nu] = 1;
      nu++;
    } else
      ucount[k]++;
  }
  for (int i = 1; i < nu; i++) /* insertion sort, lexicographic (row, col) */
    for (int j = i; j > 0; j--) {
      pt_t a = uniq[j - 1], b = uniq[j];
      if (a.row > b.row || (a.row == b.row && a.col > b.col)) {
        pt_t tp = uniq[j - 1];
        uniq[j - 1] = uniq[j];
        uniq[j] = tp;
        int t = ufirst[j - 1];
        ufirst[j - 1] = ufirst[j];
        ufirst[j] = t;
        t = ucount[j - 1];
        ucount[j - 1] = ucount[j];
        ucount[j] = t;
      } else
        break;
    }
  /* phase B: contested cells */
  for (int u = 0; u < nu; u++) {
    if (ucount[u] <= 1) continue;
    pt_t mv = uniq[u];
    int free_cell = 1;
    for (int s = 0; s < nslots; s++) {
      if (!pt_eq(slot_tgt[s], mv)) continue;
      int agent_id = slot_agent[s];
      if (pos_occupied(o, e, mv)) {
        int conf = by_pos_lookup(o, by_pos, mv);
        pt_t curr_pos = e->pos[agent_id];
        pt_t curr_conflict_pos = e->pos[conf];
        pt_t conflict_move = has_move[conf] ? move[conf] : curr_conflict_pos;
        if (agent_id == conf)
          free_cell = 0;
        else if (!has_move[conf] || pt_eq(curr_conflict_pos, conflict_move))
          free_cell = 0;
        else if (has_move[conf]) {
          if (pt_eq(move[conf], curr_pos) && pt_eq(mv, e->pos[conf])) free_cell = 0;
        }
      }
    }
    if (free_cell) {
      update_agent_pos(o, e, slot_agent[ufirst[u]], mv);
      memcpy(by_pos, e->pos, sizeof(by_pos));
    }
    for (int s = 0; s < nslots; s++)
      if (pt_eq(slot_tgt[s], mv)) move[slot_agent[s]] = e->pos[slot_agent[s]];
  }
  /* phase C */
  for (;;) {
    int nm = 0;
    for (int a = 0; a < n; a++) nm += has_move[a];
    if (nm == 0) break;
    memcpy(by_pos, e->pos, sizeof(by_pos));
    int snap_has[MAXN];
    pt_t snap_move[MAXN];
    memcpy(snap_has, has_move, sizeof(snap_has));
    memcpy(snap_move, move, sizeof(snap_move));
    int deleted[MAXN] = {0};
    for (int k = 0; k < norder; k++) {
      int agent_id = order[k];
      if (!snap_has[agent_id]) continue;
      if (deleted[agent_id]) continue;
      pt_t mv = snap_move[agent_id];
      if (pos_occupied(o, e, mv)) {
        int conf = by_pos_lookup(o, by_pos, mv);
        if (conf < 0) { /* reference would raise KeyError; unreachable */
          has_move[agent_id] = 0;
          deleted[agent_id] = 1;
          continue;
        }
        pt_t curr_pos = e->pos[agent_id];
        pt_t curr_conflict_pos = e->pos[conf];
        pt_t conflict_move = has_move[conf] ? move[conf] : curr_conflict_pos;
        if (agent_id == conf) {
          has_move[agent_id] = 0;
          deleted[agent_id] = 1;
        } else if (!snap_has[conf] || pt_eq(curr_conflict_pos, conflict_move)) {
          has_move[agent_id] = 0;
          deleted[agent_id] = 1;
        } else if (snap_has[conf]) {
          if (pt_eq(move[conf], curr_pos) && pt_eq(mv, e->pos[conf])) {
            has_move[conf] = 0;
            has_move[agent_id] = 0;
            deleted[agent_id] = 1;
            deleted[conf] = 1;
          }
        }
      } else {
        update_agent_pos(o, e, agent_id, mv);
        has_move[agent_id] = 0;
        deleted[agent_id] = 1;
      }
    }
    int nm2 = 0;
    for (int a = 0; a < n; a++) nm2 += has_move[a];
    if (nm2 == nm) {
      for (int k = 0; k < norder; k++)
        if (has_move[order[k]]) update_agent_pos(o, e, order[k], move[order[k]]);
      break;
    }
  }
}

/* ------------------------------------------------------------------------- */
/* beams: update_map_fire map_env.py:721-814                                   */
/* ------------------------------------------------------------------------- */
/* returns number of updates; fire_char 'F' or 'C' */
static int fire_beam(orc_t* o, env_t* e, int firer, int is_clean, int* upd_idx) {
  int n = o->n;
  pt_t snap[MAXN];
  memcpy(snap, e->pos, sizeof(snap)); /* agent_by_pos built at call time */
  int dir[2] = {ORI_VEC[e->orient[firer]][0], ORI_VEC[e->orient[firer]][1]};
  int right[2];
  rot_right(dir, right);
  pt_t start = e->pos[firer];
  pt_t starts[3] = {start,
                    {start.row + right[0] - dir[0], start.col + right[1] - dir[1]},
                    {start.row - right[0] - dir[0], start.col - right[1] - dir[1]}};
  int nupd = 0;
  for (int k = 0; k < 3; k++) {
    int r = starts[k].row + dir[0], c = starts[k].col + dir[1];
    for (int i = 0; i < 5; i++) { /* fire_len = all_actions["FIRE"] = 5 for both beams */
      if (in_bounds(o, r, c) && e->grid[r * o->W + c] != CE_CELL_WALL) {
        int idx = r * o->W + c;
        if (o->cfg.flags & CE_FLAG_BEAM_TRACE) /* firing_points -> beam_pos (map_env.py:788,813): later entries win */
          o->b.beam_map[(size_t)(e - o->envs) * o->cells + idx] = is_clean ? CE_BEAM_CLEAN : CE_BEAM_FIRE;
        if (is_clean && e->grid[idx] == CE_CELL_WASTE) upd_idx[nupd++] = idx; /* cell_types=[H] -> R */
        pt_t p = {r, c};
        if (pos_occupied(o, e, p)) {
          int hit = by_pos_lookup(o, snap, p);
          if (!is_clean && hit >= 0) e->reward_this_turn[hit] -= 50; /* hit(b"F") Agent.py:178-180,224-226 */
          break;
        }
        if (is_clean && e->grid[idx] == CE_CELL_WASTE) break; /* blocking_cells=[H] */
        r += dir[0];
        c += dir[1];
      } else
        break;
    }
  }
  (void)n;
  return nupd;
}

/* update_custom_moves map_env.py:678-693 + custom_action cleanup_new.py:269-292 / harvest_new.py:241-249 */
static void update_custom_moves(orc_t* o, env_t* e, const uint8_t* act) {
  uint8_t ids[MAXN];
  for (int a = 0; a < o->n; a++) ids[a] = (uint8_t)a;
  shuffle_u8(&e->np_rng, ids, o->n);
  for (int k = 0; k < o->n; k++) {
    int a = ids[k];
    int ac = act[a];
    if (ac <= 6) continue;
    int upd[16];
    if (o->kind == CE_KIND_CLEANUP) {
      if (ac == 8) { /* FIRE */
        e->reward_this_turn[a] -= 1;
        fire_beam(o, e, a, 0, upd);
      } else if (ac == 7) { /* CLEAN */
        int nu = fire_beam(o, e, a, 1, upd);
        e->cleaned_squares[a] = nu;
        for (int i = 0; i < nu; i++) e->grid[upd[i]] = CE_CELL_RIVER;
      }
    } else { /* harvest: any custom action fires F */
      e->reward_this_turn[a] -= 1;
      fire_beam(o, e, a, 0, upd);
    }
  }
}

/* ------------------------------------------------------------------------- */
/* feature obs + infos                                                        */
/* ------------------------------------------------------------------------- */
static void closest_of(const orc_t* o, const env_t* e, int code, pt_t from, int* out_r, int* out_c, int* count) {
  int best = 1 << 30, br = 0, bc = 0, cnt = 0;
  for (int r = 0; r < o->H; r++)
    for (int c = 0; c < o->W; c++)
      if (e->grid[r * o->W + c] == code) {
        cnt++;
        int d = abs(r - from.row) + abs(c - from.col);
        if (d < best) { /* np.argmin: first minimum in row-major list order */
          best = d;
          br = r;
          bc = c;
        }
      }
  *out_r = br;
  *out_c = bc; /* [0, 0] sentinel when none */
  *count = cnt;
}

/* count_apples_in_radius harvest_new.py:326-336 over a given apple mask */
static int apples_in_radius5(const orc_t* o, const uint8_t* apple_mask, pt_t loc) {
  int num = 0;
  for (int j = -5; j <= 5; j++)
    for (int k = -5; k <= 5; k++)
      if (j * j + k * k <= 5) {
        int r = loc.row + j, c = loc.col + k;
        if (in_bounds(o, r, c) && apple_mask[r * o->W + c]) num++;
      }
  return num;
}

/* numpy pairwise sum for float64 (np.mean of a short list) */
static double np_sum(const double* a, int n) {
  if (n < 8) {
    double res = 0.;
    for (int i = 0; i < n; i++) res += a[i];
    return res;
  }
  double r[8];
  for (int j = 0; j < 8; j++) r[j] = a[j];
  int i;
  for (i = 8; i < n - (n % 8); i += 8)
    for (int j = 0; j < 8; j++) r[j] += a[i + j];
  double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
  for (; i < n; i++) res += a[i];
  return res;
}

/* infos[k]['feature_obs'] (cleanup_new.py:243-251, harvest_new.py:215-222); also the feature-mode
 * reset observation (cleanup_new.py:193-202, harvest_new.py:160-168) with cleaned = 0 */
static void write_features(orc_t* o, env_t* e, int ei, const int* cleaned) {
  int n = o->n;
  int16_t* feat = o->b.features + (size_t)ei * n * o->b.num_features;
  uint8_t now_apples[16 * 38];
  for (int i = 0; i < o->cells; i++) now_apples[i] = e->grid[i] == CE_CELL_APPLE;
  for (int a = 0; a < n; a++) {
    int16_t* f = feat + (size_t)a * o->b.num_features;
    /* compute_closest_pos (cleanup_new.py:405-412): index 1 for a0, else 0; n==1 -> 0 */
    int cp = (a == 0 && n > 1) ? 1 : 0;
    int ar, ac2, napples, wr = 0, wc = 0, nwaste = 0;
    closest_of(o, e, CE_CELL_APPLE, e->pos[a], &ar, &ac2, &napples);
    f[0] = (int16_t)e->pos[a].row;
    f[1] = (int16_t)e->pos[a].col;
    f[2] = (int16_t)e->orient[a];
    f[3] = (int16_t)e->pos[cp].row;
    f[4] = (int16_t)e->pos[cp].col;
    f[5] = (int16_t)e->orient[cp];
    f[6] = (int16_t)ar;
    f[7] = (int16_t)ac2;
    if (o->kind == CE_KIND_CLEANUP) {
      closest_of(o, e, CE_CELL_WASTE, e->pos[a], &wr, &wc, &nwaste);
      f[8] = (int16_t)wr;
      f[9] = (int16_t)wc;
      f[10] = (int16_t)napples;
      f[11] = (int16_t)nwaste;
      for (int b2 = 0; b2 < n; b2++) f[12 + b2] = (int16_t)cleaned[b2];
    } else {
      f[8] = (int16_t)apples_in_radius5(o, now_apples, e->pos[a]);
      f[9] = (int16_t)napples;
      for (int b2 = 0; b2 < 2 * n; b2++) f[10 + b2] = 0;
    }
  }
}

/* ------------------------------------------------------------------------- */
/* grid env: seed+construct, reset, step                                      */
/* ------------------------------------------------------------------------- */
static void latch_zero_metrics(orc_t* o, int ei) {
  int n = o->n;
  memset(o->b.int_metrics + (size_t)ei * CE_MI_COUNT(n), 0, sizeof(int64_t) * CE_MI_COUNT(n));
  memset(o->b.f64_metrics + (size_t)ei * CE_MF_COUNT(n), 0, sizeof(double) * CE_MF_COUNT(n));
}

static void export_state(orc_t* o, int ei) { /* the last thing construct / reset / step do to an env */
  env_t* e = &o->envs[ei];
  int n = o->n;
  rng_end_of_op(&e->np_rng);
  if (o->kind != CE_KIND_SELFDRIVE) {
    memcpy(o->b.grid + (size_t)ei * o->cells, e->grid, o->cells);
    for (int a = 0; a < n; a++) {
      uint8_t* p = o->b.agents + ((size_t)ei * n + a) * 4;
      p[0] = (uint8_t)e->pos[a].row;
      p[1] = (uint8_t)e->pos[a].col;
      p[2] = (uint8_t)e->orient[a];
      p[3] = 0;
    }
    memcpy(o->b.spawn_perm + (size_t)ei * 20, e->spawn_perm, 20);
    if (o->kind == CE_KIND_CLEANUP) memcpy(o->b.waste_perm + (size_t)ei * 119, e->waste_perm, 119);
    o->b.timestep[ei] = e->timesteps;
  } else {
    double* s = o->b.sd_state + (size_t)ei * CE_SD_STATE_DOUBLES(n);
    for (int a = 0; a < n; a++) {
      s[a] = e->sd_pos[a];
      s[n + a] = e->sd_vel[a];
      s[2 * n + a] = e->sd_dtf[a];
      s[3 * n + a] = e->sd_done[a];
      s[4 * n + 2 + a] = a < e->sd_ncross ? e->sd_cross[a] : -1;
    }
    s[4 * n] = e->sd_done_all;
    s[4 * n + 1] = e->sd_ncross;
    s[5 * n + 2] = o->b.f64_metrics[(size_t)ei * CE_MF_COUNT(n) + CE_MF_TRANSFERS];
  }
  o->b.theta[ei] = e->theta;
  if (e->np_rng.ctr) { /* key0, key1, generation, pad */
    uint32_t* rw = o->b.rng + (size_t)ei * CE_RNG_WORDS_COUNTER;
    rw[0] = e->np_rng.k0;
    rw[1] = e->np_rng.k1;
    rw[2] = e->np_rng.gen;
    rw[3] = 0;
    return;
  }
  uint32_t* rw = o->b.rng + (size_t)ei * (o->kind == CE_KIND_SELFDRIVE ? CE_RNG_WORDS_SELFDRIVE : CE_RNG_WORDS_GRID);
  memcpy(rw, e->np_rng.key, 624 * 4);
  rw[624] = e->np_rng.pos;
  if (o->kind == CE_KIND_SELFDRIVE) {
    memcpy(rw + CE_RNG_WORDS_GRID, e->py_rng.key, 624 * 4);
    rw[CE_RNG_WORDS_GRID + 624] = e->py_rng.pos;
  }
}

static void grid_seed_construct(orc_t* o, int ei, uint64_t seed, int mode) {
  env_t* e = &o->envs[ei];
  if ((mode & CE_SEED_RESEED) && (o->cfg.flags & CE_FLAG_RNG_COUNTER)) {
    memset(&e->np_rng, 0, sizeof(e->np_rng));
    e->np_rng.ctr = 1;
    e->np_rng.k0 = (uint32_t)seed;
    e->np_rng.k1 = (uint32_t)(seed >> 32);
    e->np_rng.gen = 0; /* counts as used up: the first draw opens generation 1 */
    e->np_rng.pos = CE_RNG_COUNTER_GEN;
  } else if (mode & CE_SEED_RESEED) {
    mt_init_genrand(&e->np_rng, (uint32_t)seed);
    uint32_t k32 = (uint32_t)seed;
    mt_init_by_array(&e->py_rng, &k32, 1);
  }
  if (!(mode & CE_SEED_CONSTRUCT)) return;
  rng_begin_op(&e->np_rng);
  /* MapEnv.__init__: spawn_points = the P cells in row-major order, then setup_agents() */
  int base_len = spawn_len_ctor(o);
  for (int i = 0; i < 20; i++) e->spawn_perm[i] = (uint8_t)i;
  for (int i = 0; i < 119; i++) e->waste_perm[i] = (uint8_t)i;
  memset(e->grid, CE_CELL_EMPTY, sizeof(e->grid)); /* world_map is blank until the first reset */
  if (setup_agents(o, e, base_len)) o->b.error_flags[ei] |= CE_FAULT_NO_SPAWN;
  /* CleanupEnv.__init__ then appends the row-major P cells again (entries 10..19 = static 10..19) */
  e->timesteps = 0;
  e->theta = 0;
  latch_zero_metrics(o, ei);
}

static void sample_theta(orc_t* o, env_t* e) {
  /* SeparateContractSubgameStage.reset two_stage_train.py:163-166 */
  if (o->cfg.flags & CE_FLAG_EXTERNAL_THETA) return; /* the caller owns theta: a reset neither draws nor changes it */
  if (o->cfg.contract == CE_CONTRACT_NONE) {
    e->theta = 0;
    return;
  }
  double u0 = mt_double(&e->np_rng);
  if (u0 > o->cfg.null_prob) {
    double u1 = mt_double(&e->np_rng);
    e->theta = o->cfg.contract_low + (o->cfg.contract_high - o->cfg.contract_low) * u1; /* R4 */
  } else
    e->theta = o->cfg.contract_low;
}

static void grid_reset(orc_t* o, int ei) {
  env_t* e = &o->envs[ei];
  int n = o->n;
  /* MapEnv.reset map_env.py:306-342 */
  if (setup_agents(o, e, spawn_len_reset(o))) o->b.error_flags[ei] |= CE_FAULT_NO_SPAWN;
  memcpy(e->grid, o->base_cell, o->cells); /* reset_map: walls + custom_reset */
  if (o->cfg.flags & CE_FLAG_BEAM_TRACE) memset(o->b.beam_map + (size_t)ei * o->cells, CE_BEAM_NONE, o->cells); /* beam_pos = [] :316 */
  latch_zero_metrics(o, ei);               /* custom_reset re-creates metrics / total_reward_dict */
  custom_map_update(o, e);
  e->timesteps = 0;
  write_obs(o, e, ei, 0); /* agents are NOT painted on the colour map by reset() */
  sample_theta(o, e);
  for (int a = 0; a < n; a++) {
    o->b.base_reward[(size_t)ei * n + a] = 0;
    o->b.reward[(size_t)ei * n + a] = 0;
    o->b.info[((size_t)ei * n + a) * 2] = 0;
    o->b.info[((size_t)ei * n + a) * 2 + 1] = 0;
  }
  {
    int zero[MAXN] = {0};
    write_features(o, e, ei, zero);
  }
  o->b.done[ei] = 0;
  export_state(o, ei);
}

/* compute_equality / compute_sustainability (cleanup_new.py:422-445) on float per-agent sums */
static void eq_sust_f64(int n, const double* sum_r, const double* sum_tr, double* equality, double* sustainability) {
  double eq = 0, total = 0; /* python ints 0 promoted on first float add */
  for (int i = 0; i < n; i++) {
    for (int j = 0; j < n; j++) eq += fabs(sum_r[i] - sum_r[j]);
    total += sum_r[i];
  }
  if (total == 0) total = 0.001;
  *equality = 1 - eq / (2 * n * total);
  double av[MAXN];
  for (int i = 0; i < n; i++) {
    double denom = sum_r[i];
    if (!(denom > 1)) denom = 1; /* max(denom, 1) */
    av[i] = sum_tr[i] / denom;
  }
  *sustainability = np_sum(av, n) / n;
}

static void finalize_episode_metrics(orc_t* o, int ei) {
  int n = o->n;
  int64_t* mi = o->b.int_metrics + (size_t)ei * CE_MI_COUNT(n);
  double* mf = o->b.f64_metrics + (size_t)ei * CE_MF_COUNT(n);
  /* compute_equality cleanup_new.py:422-434 on the base (integer) rewards */
  {
    int64_t eq = 0, total = 0;
    for (int i = 0; i < n; i++) {
      for (int j = 0; j < n; j++) eq += llabs(mi[CE_MI_AGENT(n, CE_MIA_SUM_R, i)] - mi[CE_MI_AGENT(n, CE_MIA_SUM_R, j)]);
      total += mi[CE_MI_AGENT(n, CE_MIA_SUM_R, i)];
    }
    double ts = total == 0 ? 0.001 : (double)total;
    mf[CE_MF_EQUALITY] = 1 - (double)eq / (2 * n * ts);
    /* compute_sustainability :436-445 */
    double av[MAXN];
    for (int i = 0; i < n; i++) {
      int64_t denom = mi[CE_MI_AGENT(n, CE_MIA_SUM_R, i)];
      if (denom < 1) denom = 1;
      av[i] = (double)mi[CE_MI_AGENT(n, CE_MIA_SUM_TR, i)] / (double)denom;
    }
    mf[CE_MF_SUSTAINABILITY] = np_sum(av, n) / n;
  }
  if (o->cfg.flags & CE_FLAG_INEQUITY_AVERSE) /* float rewards: the same two formulas on the float sums */
    eq_sust_f64(n, mf + CE_MF_BASE_AGENT(n, CE_MFA_BASE_SUM_R, 0), mf + CE_MF_BASE_AGENT(n, CE_MFA_BASE_SUM_TR, 0),
                &mf[CE_MF_EQUALITY], &mf[CE_MF_SUSTAINABILITY]);
  if (o->cfg.contract != CE_CONTRACT_NONE) /* two_stage_train.py:97-99 on the transferred (float) rewards */
    eq_sust_f64(n, mf + CE_MF_AGENT(n, CE_MFA_SUM_R, 0), mf + CE_MF_AGENT(n, CE_MFA_SUM_TR, 0), &mf[CE_MF_TRANSFER_EQUALITY],
                &mf[CE_MF_TRANSFER_SUSTAINABILITY]);
  memcpy(o->b.final_int_metrics + (size_t)ei * CE_MI_COUNT(n), mi, sizeof(int64_t) * CE_MI_COUNT(n));
  memcpy(o->b.final_f64_metrics + (size_t)ei * CE_MF_COUNT(n), mf, sizeof(double) * CE_MF_COUNT(n));
}

static void grid_step(orc_t* o, int ei, const uint8_t* act) {
  env_t* e = &o->envs[ei];
  int n = o->n, W = o->W;
  int max_action = o->kind == CE_KIND_CLEANUP ? 8 : 7;
  for (int a = 0; a < n; a++)
    if (act[a] > max_action) {
      o->b.error_flags[ei] |= CE_FAULT_BAD_ACTION;
      return;
    }
  rng_begin_op(&e->np_rng);
  int64_t* mi = o->b.int_metrics + (size_t)ei * CE_MI_COUNT(n);
  double* mf = o->b.f64_metrics + (size_t)ei * CE_MF_COUNT(n);
  /* reference current_apple_points == apples on the grid when the step is entered */
  for (int i = 0; i < o->cells; i++) e->entry_apples[i] = e->grid[i] == CE_CELL_APPLE;

  /* ---- MapEnv.step map_env.py:216-304 ---- */
  e->timesteps += 1;
  if (o->cfg.flags & CE_FLAG_BEAM_TRACE) memset(o->b.beam_map + (size_t)ei * o->cells, CE_BEAM_NONE, o->cells); /* beam_pos = [] :231 */
  update_moves(o, e, act);
  for (int a = 0; a < n; a++) { /* consume map_env.py:244-247 */
    int idx = e->pos[a].row * W + e->pos[a].col;
    if (e->grid[idx] == CE_CELL_APPLE) {
      e->reward_this_turn[a] += 1;
      e->grid[idx] = CE_CELL_EMPTY;
    }
  }
  update_custom_moves(o, e, act);
  custom_map_update(o, e);
  write_obs(o, e, ei, 1);
  double rew_f[MAXN];
  int rew_i[MAXN];
  int float_rewards = 0;
  for (int a = 0; a < n; a++) { /* compute_reward Agent.py:87-90 */
    rew_i[a] = e->reward_this_turn[a];
    e->reward_this_turn[a] = 0;
  }
  if (o->cfg.flags & CE_FLAG_COLLECTIVE_REWARD) { /* map_env.py:289-292 */
    int s = 0;
    for (int a = 0; a < n; a++) s += rew_i[a];
    for (int a = 0; a < n; a++) rew_i[a] = s;
  }
  for (int a = 0; a < n; a++) rew_f[a] = rew_i[a];
  if (o->cfg.flags & CE_FLAG_INEQUITY_AVERSE) { /* map_env.py:293-301 */
    float_rewards = 1;
    double tmp[MAXN];
    for (int a = 0; a < n; a++) {
      long pos = 0, neg = 0;
      for (int b2 = 0; b2 < n; b2++) {
        int d = rew_i[b2] - rew_i[a];
        if (d > 0) pos += d;
        if (d < 0) neg += d;
      }
      double dis = o->cfg.alpha * (double)pos;
      double adv = o->cfg.beta * (double)neg;
      tmp[a] = rew_i[a] - (dis + adv) / (n - 1);
    }
    for (int a = 0; a < n; a++) rew_f[a] = tmp[a];
  }

  /* ---- CleanupEnv.step cleanup_new.py:211-267 / HarvestEnv.step harvest_new.py:181-239 ---- */
  int eaten[MAXN] = {0}, eaten_close[MAXN] = {0}, cleaned[MAXN] = {0};
  if (o->kind == CE_KIND_CLEANUP) {
    for (int a = 0; a < n; a++) {
      cleaned[a] = e->cleaned_squares[a];
      mi[CE_MI_DIRT_CLEANED] += cleaned[a];
      mi[CE_MI_AGENT(n, CE_MIA_A, a)] += cleaned[a];
      e->cleaned_squares[a] = 0;
    }
    for (int a = 0; a < n; a++)
      if (e->entry_apples[e->pos[a].row * W + e->pos[a].col]) {
        eaten[a] += 1;
        mi[CE_MI_TOTAL_APPLES_EATEN] += 1;
      }
  } else {
    for (int a = 0; a < n; a++)
      if (e->entry_apples[e->pos[a].row * W + e->pos[a].col]) {
        eaten[a] += 1;
        mi[CE_MI_AGENT(n, CE_MIA_A, a)] += 1;
        if (apples_in_radius5(o, e->entry_apples, e->pos[a]) < 4) {
          eaten_close[a] += 1;
          mi[CE_MI_LOW_DENSITY_APPLES] += 1;
          mi[CE_MI_AGENT(n, CE_MIA_B, a)] += 1;
        }
        mi[CE_MI_TOTAL_APPLES_EATEN] += 1;
      }
  }
  /* total_reward_dict / raw_env_rewards (cleanup_new.py:227-234, harvest_new.py:200-205): integer accumulators of the
   * integer rewards; with float rewards (inequity aversion) the reference appends / sums the floats themselves, in
   * step order — kept in the float twins CE_MF_RAW_ENV_REWARDS_F / CE_MF_BASE_AGENT */
  for (int a = 0; a < n; a++) {
    mi[CE_MI_AGENT(n, CE_MIA_SUM_R, a)] += rew_i[a];
    mi[CE_MI_AGENT(n, CE_MIA_SUM_TR, a)] += (int64_t)(e->timesteps - 1) * rew_i[a];
    mi[CE_MI_RAW_ENV_REWARDS] += rew_i[a];
  }
  if (float_rewards) {
    double raw = 0; /* raw_rewards = 0; for k, v in r.items(): raw_rewards += v */
    for (int a = 0; a < n; a++) raw += rew_f[a];
    mf[CE_MF_RAW_ENV_REWARDS_F(n)] += raw;
    for (int a = 0; a < n; a++) {
      mf[CE_MF_BASE_AGENT(n, CE_MFA_BASE_SUM_R, a)] += rew_f[a];
      mf[CE_MF_BASE_AGENT(n, CE_MFA_BASE_SUM_TR, a)] += (double)(e->timesteps - 1) * rew_f[a];
    }
  }
  write_features(o, e, ei, cleaned);
  int16_t* feat = o->b.features + (size_t)ei * n * o->b.num_features;
  int done = e->timesteps == (int)o->cfg.horizon;

  /* ---- contract + SeparateContractEnv.step two_stage_train.py:62-121 ---- */
  double rews[MAXN];
  for (int a = 0; a < n; a++) rews[a] = float_rewards ? rew_f[a] : (double)rew_i[a];
  if (o->cfg.contract != CE_CONTRACT_NONE) {
    double tr[MAXN];
    for (int a = 0; a < n; a++) {
      if (o->cfg.contract == CE_CONTRACT_CLEANUP)
        tr[a] = -e->theta * cleaned[a]; /* contract_list.py:26 */
      else                              /* contract_list.py:50-53 */
        tr[a] = (feat[(size_t)a * o->b.num_features + 8] < 4 && eaten_close[a] > 0) ? e->theta : 0.0;
    }
    double total = 0;
    for (int i = 0; i < n; i++) {
      rews[i] -= tr[i];
      total += tr[i];
      for (int j = 0; j < n; j++)
        if (i != j) rews[j] += tr[i] / (n - 1);
    }
    mf[CE_MF_TRANSFERS] += total;
    for (int a = 0; a < n; a++) {
      mf[CE_MF_AGENT(n, CE_MFA_SUM_R, a)] += rews[a];
      mf[CE_MF_AGENT(n, CE_MFA_SUM_TR, a)] += (double)(e->timesteps - 1) * rews[a];
    }
  }
  for (int a = 0; a < n; a++) {
    o->b.base_reward[(size_t)ei * n + a] = rew_i[a];
    o->b.reward[(size_t)ei * n + a] = rews[a];
    o->b.info[((size_t)ei * n + a) * 2] = (uint8_t)eaten[a];
    o->b.info[((size_t)ei * n + a) * 2 + 1] = (uint8_t)(o->kind == CE_KIND_CLEANUP ? cleaned[a] : eaten_close[a]);
  }
  o->b.done[ei] = (uint8_t)done;
  if (done) {
    finalize_episode_metrics(o, ei);
    if (o->cfg.flags & CE_FLAG_AUTO_RESET) {
      uint8_t d = o->b.done[ei];
      int32_t br[MAXN];
      double rr[MAXN];
      uint8_t inf[MAXN * 2];
      int16_t ft[MAXN * 40];
      memcpy(br, o->b.base_reward + (size_t)ei * n, sizeof(int32_t) * n);
      memcpy(rr, o->b.reward + (size_t)ei * n, sizeof(double) * n);
      memcpy(inf, o->b.info + (size_t)ei * n * 2, 2 * n);
      memcpy(ft, feat, sizeof(int16_t) * n * o->b.num_features);
      grid_reset(o, ei);
      o->b.done[ei] = d;
      memcpy(o->b.base_reward + (size_t)ei * n, br, sizeof(int32_t) * n);
      memcpy(o->b.reward + (size_t)ei * n, rr, sizeof(double) * n);
      memcpy(o->b.info + (size_t)ei * n * 2, inf, 2 * n);
      memcpy(feat, ft, sizeof(int16_t) * n * o->b.num_features);
      return;
    }
  }
  export_state(o, ei);
}


/* ------------------------------------------------------------------------- */
/* feature-vector envs: HarvestFeatures (harvest_features.py:60-336) and        */
/* CleanupFeatures (cleanup_features.py:48-309), + the SeparateContract wrapper */
/* The reference keeps current_apple_points / current_waste_points as Python    */
/* lists; np.argmin over them breaks distance ties by LIST order, so each       */
/* present cell carries the stamp of its list position (stamps only grow).      */
/* ------------------------------------------------------------------------- */
static int is_feat_kind(int kind) { return kind == CE_KIND_HARVEST_FEATURES || kind == CE_KIND_CLEANUP_FEATURES; }

/* CPython random (Lib/random.py): getrandbits(k <= 32) takes the TOP k bits of one word */
static uint32_t py_randbelow(mt_t* m, uint32_t nn) { /* _randbelow_with_getrandbits */
  int k = 0;
  for (uint32_t v = nn; v; v >>= 1) k++; /* n.bit_length() */
  uint32_t r = mt_next(m) >> (32 - k);
  while (r >= nn) r = mt_next(m) >> (32 - k);
  return r;
}
static void py_shuffle_u8(mt_t* m, uint8_t* x, int len) { /* random.shuffle: i = len-1 .. 1, j = randbelow(i + 1) */
  for (int i = len - 1; i >= 1; i--) {
    uint32_t j = py_randbelow(m, (uint32_t)i + 1);
    uint8_t t = x[i];
    x[i] = x[j];
    x[j] = t;
  }
}

static int feat_apple_index(const orc_t* o, int r, int c) {
  for (int i = 0; i < o->n_apple; i++)
    if (o->apple_pts[i].row == r && o->apple_pts[i].col == c) return i;
  return -1;
}
static int feat_waste_index(const orc_t* o, int r, int c) {
  for (int i = 0; i < o->n_waste; i++)
    if (o->waste_pts[i].row == r && o->waste_pts[i].col == c) return i;
  return -1;
}
static int feat_apple_at(const orc_t* o, const env_t* e, int r, int c) {
  int i = feat_apple_index(o, r, c);
  return i >= 0 && e->apple_stamp[i] != CE_FEAT_ABSENT;
}
/* count_apples_in_radius harvest_features.py:127-137 (j^2 + k^2 <= radius, current apples) */
static int feat_count_apples(const orc_t* o, const env_t* e, int radius, pt_t loc) {
  int num = 0;
  for (int j = -radius; j <= radius; j++)
    for (int k = -radius; k <= radius; k++)
      if (j * j + k * k <= radius && feat_apple_at(o, e, loc.row + j, loc.col + k)) num++;
  return num;
}
static int feat_agent_on(const orc_t* o, const env_t* e, pt_t p) { /* `p in self.agent_pos.values()` */
  for (int a = 0; a < o->n; a++)
    if (pt_eq(e->pos[a], p)) return 1;
  return 0;
}

/* initialize_arrays: harvest starts with every apple, cleanup with no apple and the H cells as waste */
static void feat_init_arrays(orc_t* o, env_t* e) {
  for (int i = 0; i < (int)CE_FEAT_APPLE_SLOTS; i++) e->apple_stamp[i] = CE_FEAT_ABSENT;
  for (int i = 0; i < (int)CE_FEAT_WASTE_SLOTS; i++) e->waste_stamp[i] = CE_FEAT_ABSENT;
  e->next_apple_stamp = e->next_waste_stamp = 0;
  if (o->kind == CE_KIND_HARVEST_FEATURES) {
    for (int i = 0; i < o->n_apple; i++) e->apple_stamp[i] = (uint16_t)e->next_apple_stamp++;
  } else {
    for (int i = 0; i < o->n_waste; i++) /* waste_start_points: the 'H' cells in row-major order */
      if (o->base_cell[o->waste_pts[i].row * o->W + o->waste_pts[i].col] == CE_CELL_WASTE)
        e->waste_stamp[i] = (uint16_t)e->next_waste_stamp++;
  }
}
/* initialize_players: random.shuffle(range(len(spawn_points))), then np.random.randint(0, 4) per agent */
static void feat_init_players(orc_t* o, env_t* e) {
  uint8_t idx[20];
  int L = o->n_spawn_static;
  for (int i = 0; i < L; i++) idx[i] = (uint8_t)i;
  py_shuffle_u8(&e->py_rng, idx, L);
  for (int a = 0; a < o->n; a++) e->pos[a] = o->spawn_pts[idx[a]];
  for (int a = 0; a < o->n; a++) e->orient[a] = (int)(mt_next(&e->np_rng) & 3u); /* legacy randint(0,4): masked 32-bit word */
}
static void feat_cleanup_probs(const orc_t* o, const env_t* e, double* p_apple, double* p_waste) {
  int cur = 0;
  for (int i = 0; i < o->n_waste; i++) cur += e->waste_stamp[i] != CE_FEAT_ABSENT;
  double density = 0;
  if (o->n_waste > 0) density = 1 - (double)(o->n_waste - cur) / o->n_waste;
  if (density >= 0.4) {
    *p_apple = 0;
    *p_waste = 0;
  } else {
    *p_waste = 0.5;
    if (density <= 0.0)
      *p_apple = 0.05;
    else
      *p_apple = (1 - (density - 0.0) / (0.4 - 0.0)) * 0.05;
  }
}
static void feat_spawn(orc_t* o, env_t* e) {
  static const double SPAWN_PROB[4] = {0, 0.005, 0.02, 0.05};
  if (o->kind == CE_KIND_HARVEST_FEATURES) { /* spawn_apples :139-151: neighbours include apples added earlier in this loop */
    for (int i = 0; i < o->n_apple; i++) {
      if (e->apple_stamp[i] != CE_FEAT_ABSENT || feat_agent_on(o, e, o->apple_pts[i])) continue;
      int num = feat_count_apples(o, e, 2, o->apple_pts[i]);
      double r = mt_double(&e->py_rng);
      if (r < SPAWN_PROB[num < 3 ? num : 3]) e->apple_stamp[i] = (uint16_t)e->next_apple_stamp++;
    }
  } else { /* spawn_apples_and_waste cleanup_features.py:115-130 */
    double pa, pw;
    feat_cleanup_probs(o, e, &pa, &pw);
    for (int i = 0; i < o->n_apple; i++) {
      if (e->apple_stamp[i] != CE_FEAT_ABSENT || feat_agent_on(o, e, o->apple_pts[i])) continue;
      double r = mt_double(&e->py_rng);
      if (r < pa) e->apple_stamp[i] = (uint16_t)e->next_apple_stamp++;
    }
    for (int i = 0; i < o->n_waste; i++) {
      if (e->waste_stamp[i] != CE_FEAT_ABSENT) continue;
      double r = mt_double(&e->py_rng);
      if (r < pw) {
        e->waste_stamp[i] = (uint16_t)e->next_waste_stamp++;
        break;
      }
    }
  }
}
/* closest present cell: min over (manhattan, stamp) == np.argmin over the list; [0, 0] when the list is empty */
static void feat_closest(const pt_t* pts, const uint16_t* stamp, int cnt, pt_t from, int* out_r, int* out_c, int* present) {
  long best = -1;
  int br = 0, bc = 0, np_ = 0;
  for (int i = 0; i < cnt; i++) {
    if (stamp[i] == CE_FEAT_ABSENT) continue;
    np_++;
    long key = ((long)(abs(pts[i].row - from.row) + abs(pts[i].col - from.col)) << 16) | stamp[i];
    if (best < 0 || key < best) {
      best = key;
      br = pts[i].row;
      bc = pts[i].col;
    }
  }
  *out_r = br;
  *out_c = bc;
  *present = np_;
}
static void feat_write_features(orc_t* o, env_t* e, int ei, const int* cleaned) {
  int n = o->n, F = (int)o->b.num_features;
  int cp = n > 1 ? 1 : 0; /* compute_closest_pos: a0 -> a1, everyone else -> a0 (the inf - inf = nan argmin quirk) */
  for (int a = 0; a < n; a++) {
    int16_t* f = o->b.features + ((size_t)ei * n + a) * F;
    int other = a == 0 ? cp : 0;
    int ar, ac, na, wr = 0, wc = 0, nw = 0;
    feat_closest(o->apple_pts, e->apple_stamp, o->n_apple, e->pos[a], &ar, &ac, &na);
    f[0] = (int16_t)e->pos[a].row;
    f[1] = (int16_t)e->pos[a].col;
    f[2] = (int16_t)e->orient[a];
    f[3] = (int16_t)e->pos[other].row;
    f[4] = (int16_t)e->pos[other].col;
    f[5] = (int16_t)e->orient[other];
    f[6] = (int16_t)ar;
    f[7] = (int16_t)ac;
    if (o->kind == CE_KIND_HARVEST_FEATURES) {
      f[8] = (int16_t)feat_count_apples(o, e, 5, e->pos[a]);
      f[9] = (int16_t)na;
      for (int b = 0; b < 2 * n; b++) f[10 + b] = 0;
    } else {
      feat_closest(o->waste_pts, e->waste_stamp, o->n_waste, e->pos[a], &wr, &wc, &nw);
      f[8] = (int16_t)wr;
      f[9] = (int16_t)wc;
      f[10] = (int16_t)na;
      f[11] = (int16_t)nw;
      for (int b = 0; b < n; b++) f[12 + b] = (int16_t)cleaned[b];
    }
  }
}
static void feat_export(orc_t* o, int ei) {
  env_t* e = &o->envs[ei];
  int n = o->n;
  uint8_t* st = o->b.grid + (size_t)ei * CE_FEAT_STATE_BYTES;
  memcpy(st, e->apple_stamp, 2 * CE_FEAT_APPLE_SLOTS);
  memcpy(st + 2 * CE_FEAT_APPLE_SLOTS, e->waste_stamp, 2 * CE_FEAT_WASTE_SLOTS);
  memcpy(st + 2 * (CE_FEAT_APPLE_SLOTS + CE_FEAT_WASTE_SLOTS), &e->next_apple_stamp, 4);
  memcpy(st + 2 * (CE_FEAT_APPLE_SLOTS + CE_FEAT_WASTE_SLOTS) + 4, &e->next_waste_stamp, 4);
  for (int a = 0; a < n; a++) {
    uint8_t* p = o->b.agents + ((size_t)ei * n + a) * 4;
    p[0] = (uint8_t)e->pos[a].row;
    p[1] = (uint8_t)e->pos[a].col;
    p[2] = (uint8_t)e->orient[a];
    p[3] = 0;
  }
  o->b.timestep[ei] = e->timesteps;
  o->b.theta[ei] = e->theta;
  uint32_t* rw = o->b.rng + (size_t)ei * CE_RNG_WORDS_SELFDRIVE;
  memcpy(rw, e->np_rng.key, 624 * 4);
  rw[624] = e->np_rng.pos;
  memcpy(rw + CE_RNG_WORDS_GRID, e->py_rng.key, 624 * 4);
  rw[CE_RNG_WORDS_GRID + 624] = e->py_rng.pos;
}
static void feat_import(orc_t* o, int ei) {
  env_t* e = &o->envs[ei];
  int n = o->n;
  const uint8_t* st = o->b.grid + (size_t)ei * CE_FEAT_STATE_BYTES;
  memcpy(e->apple_stamp, st, 2 * CE_FEAT_APPLE_SLOTS);
  memcpy(e->waste_stamp, st + 2 * CE_FEAT_APPLE_SLOTS, 2 * CE_FEAT_WASTE_SLOTS);
  memcpy(&e->next_apple_stamp, st + 2 * (CE_FEAT_APPLE_SLOTS + CE_FEAT_WASTE_SLOTS), 4);
  memcpy(&e->next_waste_stamp, st + 2 * (CE_FEAT_APPLE_SLOTS + CE_FEAT_WASTE_SLOTS) + 4, 4);
  for (int a = 0; a < n; a++) {
    const uint8_t* p = o->b.agents + ((size_t)ei * n + a) * 4;
    e->pos[a].row = p[0];
    e->pos[a].col = p[1];
    e->orient[a] = p[2];
  }
  e->timesteps = o->b.timestep[ei];
  e->theta = o->b.theta[ei];
  const uint32_t* rw = o->b.rng + (size_t)ei * CE_RNG_WORDS_SELFDRIVE;
  memcpy(e->np_rng.key, rw, 624 * 4);
  e->np_rng.pos = rw[624];
  memcpy(e->py_rng.key, rw + CE_RNG_WORDS_GRID, 624 * 4);
  e->py_rng.pos = rw[CE_RNG_WORDS_GRID + 624];
}
static void feat_seed_construct(orc_t* o, int ei, uint64_t seed, int mode) {
  env_t* e = &o->envs[ei];
  if (mode & CE_SEED_RESEED) {
    mt_init_genrand(&e->np_rng, (uint32_t)seed);
    uint32_t k32 = (uint32_t)seed;
    mt_init_by_array(&e->py_rng, &k32, 1);
  }
  if (!(mode & CE_SEED_CONSTRUCT)) return;
  feat_init_arrays(o, e); /* __init__: initialize_arrays, (compute_probabilities), initialize_players */
  feat_init_players(o, e);
  e->timesteps = 0;
  e->theta = 0;
  latch_zero_metrics(o, ei);
}
static void feat_reset(orc_t* o, int ei) {
  env_t* e = &o->envs[ei];
  int n = o->n;
  feat_init_arrays(o, e);
  feat_init_players(o, e);
  feat_spawn(o, e);
  latch_zero_metrics(o, ei);
  e->timesteps = 0;
  sample_theta(o, e);
  int zero[MAXN] = {0};
  feat_write_features(o, e, ei, zero);
  for (int a = 0; a < n; a++) {
    o->b.base_reward[(size_t)ei * n + a] = 0;
    o->b.reward[(size_t)ei * n + a] = 0;
    o->b.info[((size_t)ei * n + a) * 2] = 0;
    o->b.info[((size_t)ei * n + a) * 2 + 1] = 0;
  }
  o->b.done[ei] = 0;
  feat_export(o, ei);
}
static void feat_step(orc_t* o, int ei, const uint8_t* act) {
  static const int MOVE[4][2] = {{0, -1}, {0, 1}, {-1, 0}, {1, 0}};  /* MOVE_ACTIONS */
  static const int FIRE[4][2] = {{-1, 0}, {0, 1}, {1, 0}, {0, -1}};  /* FIRE_DIRECTIONS */
  env_t* e = &o->envs[ei];
  int n = o->n, W = o->W;
  int harvest = o->kind == CE_KIND_HARVEST_FEATURES;
  int max_action = harvest ? 7 : 8;
  for (int a = 0; a < n; a++)
    if (act[a] > max_action) {
      o->b.error_flags[ei] |= CE_FAULT_BAD_ACTION;
      return;
    }
  int64_t* mi = o->b.int_metrics + (size_t)ei * CE_MI_COUNT(n);
  double* mf = o->b.f64_metrics + (size_t)ei * CE_MF_COUNT(n);
  /* move_squares, an insertion-ordered dict: stayers first, then movers in key order */
  int order[MAXN], has[MAXN] = {0}, cnt = 0;
  pt_t sq[MAXN];
  for (int a = 0; a < n; a++)
    if (harvest ? act[a] > 3 : act[a] == 4) {
      sq[a] = e->pos[a];
      has[a] = 1;
      order[cnt++] = a;
    }
  for (int a = 0; a < n; a++)
    if (act[a] < 4) {
      pt_t t = {e->pos[a].row + MOVE[act[a]][0], e->pos[a].col + MOVE[act[a]][1]};
      int blocked = !in_bounds(o, t.row, t.col) || o->is_wall[t.row * W + t.col];
      for (int k = 0; k < cnt && !blocked; k++) blocked = pt_eq(sq[order[k]], t);
      sq[a] = blocked ? e->pos[a] : t;
      has[a] = 1;
      order[cnt++] = a;
    }
  for (int a = 0; a < n; a++)
    if (has[a]) e->pos[a] = sq[a];
  /* consume apples in move_squares order */
  int rew[MAXN] = {0}, eaten[MAXN] = {0}, eaten_close[MAXN] = {0}, cleaned[MAXN] = {0};
  for (int k = 0; k < cnt; k++) {
    int a = order[k];
    int ai = feat_apple_index(o, sq[a].row, sq[a].col);
    if (ai < 0 || e->apple_stamp[ai] == CE_FEAT_ABSENT) continue;
    rew[a] += 1;
    if (harvest) {
      eaten[a] += 1;
      if (feat_count_apples(o, e, 5, e->pos[a]) < 4) {
        eaten_close[a] += 1;
        mi[CE_MI_LOW_DENSITY_APPLES] += 1;
        mi[CE_MI_AGENT(n, CE_MIA_B, a)] += 1;
      }
      mi[CE_MI_TOTAL_APPLES_EATEN] += 1;
      mi[CE_MI_AGENT(n, CE_MIA_A, a)] += 1;
    }
    e->apple_stamp[ai] = CE_FEAT_ABSENT;
  }
  for (int a = 0; a < n; a++) { /* rotations */
    if (act[a] == 5) e->orient[a] = (e->orient[a] + 1) % 4;
    if (act[a] == 6) e->orient[a] = (e->orient[a] + 3) % 4;
  }
  if (!harvest) /* clean beams (act 7): three rays of 6 cells from the agent's own row, through waste, until a wall */
    for (int a = 0; a < n; a++) {
      if (act[a] != 7) continue;
      const int* d = FIRE[e->orient[a]];
      const int* s = FIRE[(e->orient[a] + 1) % 4];
      for (int ray = 0; ray < 3; ray++) {
        int r0 = e->pos[a].row + (ray == 1 ? s[0] : ray == 2 ? -s[0] : 0);
        int c0 = e->pos[a].col + (ray == 1 ? s[1] : ray == 2 ? -s[1] : 0);
        for (int j = 0; j < 6; j++) {
          int r = r0 + j * d[0], c = c0 + j * d[1];
          if (in_bounds(o, r, c) && o->is_wall[r * W + c]) break;
          int wi = in_bounds(o, r, c) ? feat_waste_index(o, r, c) : -1;
          if (wi >= 0 && e->waste_stamp[wi] != CE_FEAT_ABSENT) {
            e->waste_stamp[wi] = CE_FEAT_ABSENT;
            cleaned[a] += 1;
            mi[CE_MI_DIRT_CLEANED] += 1;
            mi[CE_MI_AGENT(n, CE_MIA_A, a)] += 1;
          }
        }
      }
    }
  feat_spawn(o, e);
  feat_write_features(o, e, ei, cleaned);
  int16_t* feat = o->b.features + (size_t)ei * n * o->b.num_features;
  e->timesteps += 1;
  int done = e->timesteps == (int)o->cfg.horizon;
  for (int a = 0; a < n; a++) { /* total_reward_dict / raw_env_rewards (floats holding small integers) */
    mi[CE_MI_AGENT(n, CE_MIA_SUM_R, a)] += rew[a];
    mi[CE_MI_AGENT(n, CE_MIA_SUM_TR, a)] += (int64_t)(e->timesteps - 1) * rew[a];
    mi[CE_MI_RAW_ENV_REWARDS] += rew[a];
  }
  double rews[MAXN];
  for (int a = 0; a < n; a++) rews[a] = (double)rew[a];
  if (o->cfg.contract != CE_CONTRACT_NONE) { /* SeparateContractEnv.step two_stage_train.py:62-121 */
    double tr[MAXN];
    for (int a = 0; a < n; a++) {
      if (o->cfg.contract == CE_CONTRACT_CLEANUP)
        tr[a] = -e->theta * cleaned[a];
      else
        tr[a] = (feat[(size_t)a * o->b.num_features + 8] < 4 && eaten_close[a] > 0) ? e->theta : 0.0;
    }
    double total = 0;
    for (int i = 0; i < n; i++) {
      rews[i] -= tr[i];
      total += tr[i];
      for (int j = 0; j < n; j++)
        if (i != j) rews[j] += tr[i] / (n - 1);
    }
    mf[CE_MF_TRANSFERS] += total;
    for (int a = 0; a < n; a++) {
      mf[CE_MF_AGENT(n, CE_MFA_SUM_R, a)] += rews[a];
      mf[CE_MF_AGENT(n, CE_MFA_SUM_TR, a)] += (double)(e->timesteps - 1) * rews[a];
    }
  }
  for (int a = 0; a < n; a++) {
    o->b.base_reward[(size_t)ei * n + a] = rew[a];
    o->b.reward[(size_t)ei * n + a] = rews[a];
    o->b.info[((size_t)ei * n + a) * 2] = (uint8_t)eaten[a];
    o->b.info[((size_t)ei * n + a) * 2 + 1] = (uint8_t)(harvest ? eaten_close[a] : cleaned[a]);
  }
  o->b.done[ei] = (uint8_t)done;
  if (done) {
    finalize_episode_metrics(o, ei);
    if (o->cfg.flags & CE_FLAG_AUTO_RESET) {
      int32_t br[MAXN];
      double rr[MAXN];
      uint8_t inf[MAXN * 2];
      int16_t ft[MAXN * 40];
      memcpy(br, o->b.base_reward + (size_t)ei * n, sizeof(int32_t) * n);
      memcpy(rr, o->b.reward + (size_t)ei * n, sizeof(double) * n);
      memcpy(inf, o->b.info + (size_t)ei * n * 2, 2 * n);
      memcpy(ft, feat, sizeof(int16_t) * n * o->b.num_features);
      feat_reset(o, ei);
      o->b.done[ei] = 1;
      memcpy(o->b.base_reward + (size_t)ei * n, br, sizeof(int32_t) * n);
      memcpy(o->b.reward + (size_t)ei * n, rr, sizeof(double) * n);
      memcpy(o->b.info + (size_t)ei * n * 2, inf, 2 * n);
      memcpy(feat, ft, sizeof(int16_t) * n * o->b.num_features);
      return;
    }
  }
  feat_export(o, ei);
}

/* ------------------------------------------------------------------------- */
/* selfdrive (self_driving_car_accelerate.py) + SelfdriveContractDistprop      */
/* ------------------------------------------------------------------------- */
static double py_min2(double x, double y) { return y < x ? y : x; } /* min([x, y]) */
static double py_max2(double x, double y) { return y > x ? y : x; } /* max([x, y]) */

static void sd_write_obs(orc_t* o, env_t* e, int ei, const int* active, const double* pos, double last) {
  int n = o->n, L = 2 * n + 7;
  for (int k = 0; k < n; k++) {
    double* ob = o->b.obs_f64 + ((size_t)ei * n + k) * L;
    if (!active[k]) {
      for (int i = 0; i < L; i++) ob[i] = NAN;
      continue;
    }
    ob[0] = pos[k];
    ob[1] = e->sd_vel[k];
    for (int i = 0; i < n; i++) ob[2 + i] = pos[i] - pos[k];
    for (int i = 0; i < n; i++) ob[2 + n + i] = e->sd_vel[i];
    ob[2 + 2 * n] = pos[0] > 0 ? 1.0 : 0.0;
    ob[3 + 2 * n] = pos[k] > 0 ? 1.0 : 0.0;
    ob[4 + 2 * n] = last;
    ob[5 + 2 * n] = e->theta; /* wrapper concat two_stage_train.py:113-117,183-187 */
    ob[6 + 2 * n] = 0.0;
  }
}

static void sd_seed_construct(orc_t* o, int ei, uint64_t seed, int mode) {
  env_t* e = &o->envs[ei];
  if (mode & CE_SEED_RESEED) {
    mt_init_genrand(&e->np_rng, (uint32_t)seed);
    uint32_t k32 = (uint32_t)seed;
    mt_init_by_array(&e->py_rng, &k32, 1);
  }
  if (!(mode & CE_SEED_CONSTRUCT)) return;
  for (int a = 0; a < o->n; a++) { /* __init__ :29-35 (no RNG use) */
    e->sd_pos[a] = o->cfg.low_bound;
    e->sd_vel[a] = o->cfg.start_vel;
    e->sd_done[a] = 0;
    e->sd_dtf[a] = -1;
  }
  e->sd_done_all = 0;
  e->sd_ncross = 0;
  e->theta = 0;
  latch_zero_metrics(o, ei);
}

static void sd_reset(orc_t* o, int ei) {
  env_t* e = &o->envs[ei];
  int n = o->n;
  double low = o->cfg.low_bound;
  for (int a = 0; a < n; a++) { /* reset :49-64 */
    double u = mt_double(&e->py_rng);
    if (a == 0) {
      e->sd_pos[a] = u * low / 2 + low / 2;
      e->sd_vel[a] = o->cfg.start_vel_ambulance;
    } else {
      e->sd_pos[a] = u * low / 16 + low * 3 / 16;
      e->sd_vel[a] = o->cfg.start_vel;
    }
    e->sd_done[a] = 0;
    e->sd_dtf[a] = -1;
  }
  e->sd_done_all = 0;
  e->sd_ncross = 0;
  latch_zero_metrics(o, ei);
  sample_theta(o, e);
  int active[MAXN] = {0};
  for (int a = 0; a < n; a++) active[a] = 1;
  sd_write_obs(o, e, ei, active, e->sd_pos, 0.0);
  for (int a = 0; a < n; a++) {
    o->b.reward[(size_t)ei * n + a] = 0;
    o->b.base_reward[(size_t)ei * n + a] = 0;
    o->b.info[((size_t)ei * n + a) * 2] = 0;
    o->b.info[((size_t)ei * n + a) * 2 + 1] = 0;
    o->b.done_agents[(size_t)ei * n + a] = 0;
  }
  o->b.sd_info[(size_t)ei * 2] = o->b.sd_info[(size_t)ei * 2 + 1] = 0;
  o->b.done[ei] = 0;
  export_state(o, ei);
}

static int in_list(const int* l, int n, int v) {
  for (int i = 0; i < n; i++)
    if (l[i] == v) return 1;
  return 0;
}

static void sd_step(orc_t* o, int ei, const float* act32, const uint8_t* active_in) {
  env_t* e = &o->envs[ei];
  int n = o->n;
  double high = o->cfg.high_bound;
  double* mf = o->b.f64_metrics + (size_t)ei * CE_MF_COUNT(n);
  int active[MAXN], n_active = 0;
  for (int a = 0; a < n; a++) {
    active[a] = active_in ? active_in[a] : !e->sd_done[a];
    n_active += active[a];
  }
  if (e->sd_done_all || n_active == 0) { /* reference raises (collision_check_all undefined, :160) */
    o->b.error_flags[ei] |= CE_FAULT_STEP_AFTER_DONE;
    return;
  }
  double new_pos[MAXN];
  for (int k = 0; k < n; k++)
    if (active[k]) { /* :172-180 */
      double a = (double)act32[k];
      double vmax = k == 0 ? 1.0 : 0.25;
      double v = py_max2(py_min2(py_max2(py_min2(a, 0.1), -0.1) + e->sd_vel[k], vmax), 0.0);
      e->sd_vel[k] = v;
      new_pos[k] = e->sd_vel[k] + e->sd_pos[k];
    }
  int just_passed[MAXN] = {0};
  /* infos of the first acting key (:183-189): the ambulance's rank in the merge order so far (n while it has not
   * crossed) and its recorded distance to the car in front (high - low while nothing was recorded) */
  double amb_rank = n, amb_dtf = e->sd_dtf[0] > -1 ? e->sd_dtf[0] : high - o->cfg.low_bound;
  for (int i = 0; i < e->sd_ncross; i++)
    if (e->sd_cross[i] == 0) amb_rank = i + 1;
  /* update_rel_rank :110-125 — sort key is the id's 2nd character => agent index order */
  for (int k = 0; k < n; k++)
    if (active[k] && e->sd_pos[k] < 0.0 && new_pos[k] > 0.0) e->sd_cross[e->sd_ncross++] = k;
  /* update_infos :127-149 */
  for (int k = 0; k < n; k++)
    if (active[k] && e->sd_pos[k] < 0.0 && new_pos[k] > 0.0) {
      just_passed[k] = 1;
      double dtf = 0.0;
      for (int i = 0; i < n; i++)
        if (i != k) {
          if (!active[i] || new_pos[i] > new_pos[k]) {
            if (!active[i]) {
              if (high - new_pos[k] > dtf) dtf = high + 1 - new_pos[k];
            } else {
              if (new_pos[i] - new_pos[k] > dtf) dtf = new_pos[i] - new_pos[i];
            }
          }
        }
      e->sd_dtf[n - 1] = dtf; /* :144 writes index n-1 (loop variable leak) */
      if (k == 0) {             /* :146-149 ambulance stats, on the first acting key (a0 itself: it is acting) */
        for (int i = 0; i < e->sd_ncross; i++)
          if (e->sd_cross[i] == 0) amb_rank = i + 1;
        amb_dtf = dtf;
      }
    }
  int crashed = 0;
  if (o->cfg.flags & CE_FLAG_COLLISION_ON) { /* check_if_crashed :81-90 */
    for (int i = 0; i + 1 < e->sd_ncross; i++) {
      int f = e->sd_cross[i], b2 = e->sd_cross[i + 1];
      double pf = active[f] ? new_pos[f] : e->sd_pos[f];
      double pb = active[b2] ? new_pos[b2] : e->sd_pos[b2];
      if (pf < pb) crashed = 1;
    }
  }
  double rews[MAXN];
  if (crashed) { /* :196-215 */
    amb_rank = n; /* :203 ambulance rank is set to the lowest value when crashed */
    e->sd_done_all = 1;
    for (int k = 0; k < n; k++)
      if (active[k]) e->sd_done[k] = 1;
    sd_write_obs(o, e, ei, active, e->sd_pos, 1.0);
    for (int k = 0; k < n; k++) rews[k] = -10000.0;
  } else {
    /* make_new_pos_consistent :92-108 */
    int pre[MAXN], npre = 0;
    for (int i = 0; i + 1 < e->sd_ncross; i++) {
      int f = e->sd_cross[i], b2 = e->sd_cross[i + 1];
      double pf = active[f] ? new_pos[f] : e->sd_pos[f];
      double pb = active[b2] ? new_pos[b2] : e->sd_pos[b2];
      if (pf < pb && active[f] && active[b2]) {
        new_pos[b2] = pf - 0.01;
        if (new_pos[b2] < 0) pre[npre++] = b2;
      }
    }
    int nc = 0;
    for (int i = 0; i < e->sd_ncross; i++)
      if (!in_list(pre, npre, e->sd_cross[i])) e->sd_cross[nc++] = e->sd_cross[i];
    e->sd_ncross = nc;
    for (int k = 0; k < n; k++)
      if (active[k]) e->sd_pos[k] = new_pos[k];
    for (int k = 0; k < n; k++) rews[k] = -1.0;
    if (active[0]) rews[0] -= 99.0;
    for (int i = 0; i < n; i++)
      if (e->sd_pos[i] > high) {
        e->sd_pos[i] = high + 1;
        e->sd_done[i] = 1;
      }
    int all_done = 1;
    for (int k = 0; k < n; k++)
      if (active[k] && !e->sd_done[k]) all_done = 0;
    e->sd_done_all = all_done;
    sd_write_obs(o, e, ei, active, e->sd_pos, 0.0);
  }
  for (int k = 0; k < n; k++) o->b.base_reward[(size_t)ei * n + k] = active[k] ? (int32_t)rews[k] : 0;
  double base[MAXN];
  memcpy(base, rews, sizeof(base));
  /* SelfdriveContractDistprop.compute_transfer contract_list.py:69-102 on a0's base obs */
  if (o->cfg.contract == CE_CONTRACT_SELFDRIVE_DISTPROP) {
    int L = 2 * n + 7;
    const double* ob0 = o->b.obs_f64 + ((size_t)ei * n + 0) * L;
    int is_tuple[MAXN + 2] = {0};
    double tval[MAXN + 2] = {0};
    double share[MAXN + 2][MAXN + 2];
    memset(share, 0, sizeof(share));
    int share_has[MAXN + 2][MAXN + 2];
    memset(share_has, 0, sizeof(share_has));
    if (active[0] && just_passed[0]) {
      int behind[MAXN + 2], nb = 0;
      for (int i = 1; i < (2 * n + 5) / 2; i++)
        if (ob0[2 + i] < 0) behind[nb++] = i;
      if (nb) {
        double sum_d = 0; /* python int 0 + floats, in order */
        double dist[MAXN + 2];
        for (int q = 0; q < nb; q++) {
          dist[q] = -ob0[2 + behind[q]];
          sum_d += dist[q];
        }
        is_tuple[0] = 1;
        tval[0] = e->theta * sum_d;
        for (int q = 0; q < nb; q++) {
          share[0][behind[q]] = dist[q] / sum_d;
          share_has[0][behind[q]] = 1;
        }
      }
      for (int i = 1; i < (2 * n + 5) / 2; i++)
        if (i < n && active[i] && !in_list(behind, nb, i)) {
          is_tuple[i] = 1;
          tval[i] = e->theta * ob0[2 + i];
          share[i][0] = 1;
          share_has[i][0] = 1;
        }
    }
    double total = 0;
    int nact = n_active;
    for (int i = 0; i < n; i++)
      if (active[i]) {
        if (is_tuple[i]) {
          rews[i] -= tval[i];
          total += tval[i];
          for (int j = 0; j < n; j++)
            if (share_has[i][j] && active[j]) rews[j] += tval[i] * share[i][j];
        } else {
          rews[i] -= 0;
          total += 0;
          for (int j = 0; j < n; j++)
            if (i != j && active[j]) rews[j] += 0.0 / (nact - 1);
        }
      }
    mf[CE_MF_TRANSFERS] += total;
  }
  for (int k = 0; k < n; k++) {
    o->b.reward[(size_t)ei * n + k] = active[k] ? rews[k] : NAN;
    o->b.info[((size_t)ei * n + k) * 2] = (uint8_t)just_passed[k];
    o->b.info[((size_t)ei * n + k) * 2 + 1] = (uint8_t)crashed;
    o->b.done_agents[(size_t)ei * n + k] = (uint8_t)e->sd_done[k];
  }
  (void)base;
  o->b.sd_info[(size_t)ei * 2] = amb_rank;
  o->b.sd_info[(size_t)ei * 2 + 1] = amb_dtf;
  o->b.done[ei] = (uint8_t)e->sd_done_all;
  if (e->sd_done_all) {
    memcpy(o->b.final_int_metrics + (size_t)ei * CE_MI_COUNT(n), o->b.int_metrics + (size_t)ei * CE_MI_COUNT(n),
           sizeof(int64_t) * CE_MI_COUNT(n));
    memcpy(o->b.final_f64_metrics + (size_t)ei * CE_MF_COUNT(n), mf, sizeof(double) * CE_MF_COUNT(n));
    if (o->cfg.flags & CE_FLAG_AUTO_RESET) {
      /* keep terminal reward/info/done, publish the reset observation */
      double rr[MAXN], si[2] = {o->b.sd_info[(size_t)ei * 2], o->b.sd_info[(size_t)ei * 2 + 1]};
      int32_t br[MAXN];
      uint8_t inf[MAXN * 2], da[MAXN];
      memcpy(br, o->b.base_reward + (size_t)ei * n, sizeof(int32_t) * n);
      memcpy(rr, o->b.reward + (size_t)ei * n, sizeof(double) * n);
      memcpy(inf, o->b.info + (size_t)ei * n * 2, 2 * n);
      memcpy(da, o->b.done_agents + (size_t)ei * n, n);
      sd_reset(o, ei);
      memcpy(o->b.base_reward + (size_t)ei * n, br, sizeof(int32_t) * n);
      memcpy(o->b.reward + (size_t)ei * n, rr, sizeof(double) * n);
      o->b.sd_info[(size_t)ei * 2] = si[0];
      o->b.sd_info[(size_t)ei * 2 + 1] = si[1];
      memcpy(o->b.info + (size_t)ei * n * 2, inf, 2 * n);
      memcpy(o->b.done_agents + (size_t)ei * n, da, n);
      o->b.done[ei] = 1;
      return;
    }
  }
  export_state(o, ei);
}

/* ------------------------------------------------------------------------- */
/* public API (orc_* mirrors ce_*; all pointers are HOST pointers)             */
/* ------------------------------------------------------------------------- */
#define ALLOC(field, type, count)                                   \
  do {                                                              \
    o->b.field = (type*)calloc((size_t)(count) != 0 ? (size_t)(count) : 1, sizeof(type)); \
    if (!o->b.field) return CE_ENOMEM;                              \
  } while (0)

int orc_create(const ce_config* cfg, orc_t** out) {
  if (!cfg || !out || cfg->abi_version != CE_ABI_VERSION) return CE_EINVAL;
  if (cfg->kind > CE_KIND_CLEANUP_FEATURES || cfg->num_envs == 0) return CE_EINVAL;
  int maxn = cfg->kind == CE_KIND_SELFDRIVE ? 10 : 9;
  if (cfg->num_agents < 1 || (int)cfg->num_agents > maxn) return CE_EINVAL;
  if ((cfg->flags & CE_FLAG_RNG_COUNTER) && cfg->kind != CE_KIND_CLEANUP && cfg->kind != CE_KIND_HARVEST) return CE_EINVAL;
  orc_t* o = (orc_t*)calloc(1, sizeof(orc_t));
  if (!o) return CE_ENOMEM;
  o->cfg = *cfg;
  if (o->cfg.horizon == 0) o->cfg.horizon = 1000;
  o->n = (int)cfg->num_agents;
  o->kind = (int)cfg->kind;
  size_t E = cfg->num_envs, n = cfg->num_agents;
  o->envs = (env_t*)calloc(E, sizeof(env_t));
  if (!o->envs) return CE_ENOMEM;
  o->b.num_envs = cfg->num_envs;
  o->b.num_agents = cfg->num_agents;
  o->b.num_int_metrics = CE_MI_COUNT(n);
  o->b.num_f64_metrics = CE_MF_COUNT(n);
  if (is_feat_kind(o->kind)) {
    if (build_static(o) != CE_OK) return CE_EINVAL;
    o->b.grid_h = o->H;
    o->b.grid_w = o->W;
    o->b.grid_env_stride = CE_FEAT_STATE_BYTES;
    o->b.rng_words = CE_RNG_WORDS_SELFDRIVE;
    o->b.num_features = o->kind == CE_KIND_CLEANUP_FEATURES ? 12 + n : 10 + 2 * n;
    ALLOC(grid, uint8_t, E * CE_FEAT_STATE_BYTES);
    ALLOC(agents, uint8_t, E * n * 4);
    ALLOC(rng, uint32_t, E * CE_RNG_WORDS_SELFDRIVE);
    ALLOC(features, int16_t, E * n * o->b.num_features);
  } else if (o->kind != CE_KIND_SELFDRIVE) {
    if (build_static(o) != CE_OK) return CE_EINVAL;
    o->cfg.ascii_map = NULL; /* the caller's string is not ours to keep */
    o->b.grid_h = o->H;
    o->b.grid_w = o->W;
    o->b.obs_agent_stride = WIN * WIN * 3;
    o->b.obs_row_stride = WIN * 3;
    o->b.obs_env_stride = (uint32_t)((n * WIN * WIN * 3 + 3) / 4 * 4);
    o->b.rng_words = (cfg->flags & CE_FLAG_RNG_COUNTER) ? CE_RNG_WORDS_COUNTER : CE_RNG_WORDS_GRID;
    o->b.grid_env_stride = (uint32_t)o->cells;
    o->b.grid_row_stride = (uint32_t)o->W;
    o->b.grid_origin = 0;
    o->b.num_features = o->kind == CE_KIND_CLEANUP ? 12 + n : 10 + 2 * n;
    ALLOC(grid, uint8_t, E * o->cells);
    ALLOC(agents, uint8_t, E * n * 4);
    ALLOC(spawn_perm, uint8_t, E * 20);
    ALLOC(waste_perm, uint8_t, E * 119);
    ALLOC(rng, uint32_t, E * CE_RNG_WORDS_GRID);
    ALLOC(obs, uint8_t, E * o->b.obs_env_stride);
    ALLOC(features, int16_t, E * n * o->b.num_features);
    ALLOC(beam_map, uint8_t, E * o->cells);
  } else {
    o->b.num_features = 2 * n + 7;
    o->b.rng_words = CE_RNG_WORDS_SELFDRIVE;
    ALLOC(rng, uint32_t, E * CE_RNG_WORDS_SELFDRIVE);
    ALLOC(sd_state, double, E * CE_SD_STATE_DOUBLES(n));
    ALLOC(obs_f64, double, E * n * (2 * n + 7));
    ALLOC(done_agents, uint8_t, E * n);
    ALLOC(sd_info, double, E * 2);
  }
  ALLOC(timestep, int32_t, E);
  ALLOC(theta, double, E);
  ALLOC(base_reward, int32_t, E * n);
  ALLOC(reward, double, E * n);
  ALLOC(done, uint8_t, E);
  ALLOC(info, uint8_t, E * n * 2);
  ALLOC(int_metrics, int64_t, E * CE_MI_COUNT(n));
  ALLOC(f64_metrics, double, E * CE_MF_COUNT(n));
  ALLOC(final_int_metrics, int64_t, E * CE_MI_COUNT(n));
  ALLOC(final_f64_metrics, double, E * CE_MF_COUNT(n));
  ALLOC(error_flags, uint32_t, E);
  *out = o;
  return CE_OK;
}

int orc_destroy(orc_t* o) {
  if (!o) return CE_EINVAL;
  void* ptrs[] = {o->b.grid, o->b.agents, o->b.spawn_perm, o->b.waste_perm, o->b.rng, o->b.timestep, o->b.theta,
                  o->b.sd_state, o->b.obs, o->b.obs_f64, o->b.base_reward, o->b.reward, o->b.done, o->b.done_agents,
                  o->b.info, o->b.features, o->b.int_metrics, o->b.f64_metrics, o->b.final_int_metrics,
                  o->b.final_f64_metrics, o->b.error_flags, o->b.beam_map, o->b.sd_info};
  for (size_t i = 0; i < sizeof(ptrs) / sizeof(ptrs[0]); i++) free(ptrs[i]);
  free(o->envs);
  free(o);
  return CE_OK;
}

int orc_seed(orc_t* o, const uint64_t* seeds, uint64_t seed0, const uint8_t* mask, int mode) {
  if (!o) return CE_EINVAL;
#pragma omp parallel for schedule(static)
  for (long ei = 0; ei < (long)o->cfg.num_envs; ei++) {
    if (mask && !mask[ei]) continue;
    uint64_t s = seeds ? seeds[ei] : seed0 + o->cfg.env_index_base + (uint64_t)ei;
    if (is_feat_kind(o->kind)) {
      feat_seed_construct(o, (int)ei, s, mode);
      feat_export(o, (int)ei);
      continue;
    }
    if (o->kind == CE_KIND_SELFDRIVE)
      sd_seed_construct(o, (int)ei, s, mode);
    else
      grid_seed_construct(o, (int)ei, s, mode);
    export_state(o, (int)ei);
  }
  return CE_OK;
}

int orc_reset(orc_t* o, const uint8_t* mask) {
  if (!o) return CE_EINVAL;
#pragma omp parallel for schedule(static)
  for (long ei = 0; ei < (long)o->cfg.num_envs; ei++) {
    if (mask && !mask[ei]) continue;
    if (is_feat_kind(o->kind))
      feat_reset(o, (int)ei);
    else if (o->kind == CE_KIND_SELFDRIVE)
      sd_reset(o, (int)ei);
    else {
      rng_begin_op(&o->envs[ei].np_rng);
      grid_reset(o, (int)ei);
    }
  }
  return CE_OK;
}

int orc_step(orc_t* o, const void* actions, const uint8_t* active) {
  if (!o || !actions) return CE_EINVAL;
  int n = o->n;
#pragma omp parallel for schedule(static)
  for (long ei = 0; ei < (long)o->cfg.num_envs; ei++) {
    if (is_feat_kind(o->kind))
      feat_step(o, (int)ei, (const uint8_t*)actions + (size_t)ei * n);
    else if (o->kind == CE_KIND_SELFDRIVE)
      sd_step(o, (int)ei, (const float*)actions + (size_t)ei * n, active ? active + (size_t)ei * n : NULL);
    else
      grid_step(o, (int)ei, (const uint8_t*)actions + (size_t)ei * n);
  }
  return CE_OK;
}

int orc_get_buffers(orc_t* o, ce_buffers* out) {
  if (!o || !out) return CE_EINVAL;
  *out = o->b;
  return CE_OK;
}

/* state injection for known-answer scenarios: copies the exported arrays back into env_t */
int orc_import_state(orc_t* o, uint32_t ei) {
  if (!o || ei >= o->cfg.num_envs) return CE_EINVAL;
  env_t* e = &o->envs[ei];
  int n = o->n;
  if (is_feat_kind(o->kind)) {
    feat_import(o, (int)ei);
    return CE_OK;
  }
  if (o->kind != CE_KIND_SELFDRIVE) {
    memcpy(e->grid, o->b.grid + (size_t)ei * o->cells, o->cells);
    for (int a = 0; a < n; a++) {
      const uint8_t* p = o->b.agents + ((size_t)ei * n + a) * 4;
      e->pos[a].row = p[0];
      e->pos[a].col = p[1];
      e->orient[a] = p[2];
    }
    memcpy(e->spawn_perm, o->b.spawn_perm + (size_t)ei * 20, 20);
    if (o->kind == CE_KIND_CLEANUP) memcpy(e->waste_perm, o->b.waste_perm + (size_t)ei * 119, 119);
    e->timesteps = o->b.timestep[ei];
  } else {
    const double* s = o->b.sd_state + (size_t)ei * CE_SD_STATE_DOUBLES(n);
    for (int a = 0; a < n; a++) {
      e->sd_pos[a] = s[a];
      e->sd_vel[a] = s[n + a];
      e->sd_dtf[a] = s[2 * n + a];
      e->sd_done[a] = (int)s[3 * n + a];
      e->sd_cross[a] = (int)s[4 * n + 2 + a];
    }
    e->sd_done_all = (int)s[4 * n];
    e->sd_ncross = (int)s[4 * n + 1];
  }
  e->theta = o->b.theta[ei];
  if (o->cfg.flags & CE_FLAG_RNG_COUNTER) {
    const uint32_t* row = o->b.rng + (size_t)ei * CE_RNG_WORDS_COUNTER;
    e->np_rng.ctr = 1;
    e->np_rng.k0 = row[0];
    e->np_rng.k1 = row[1];
    e->np_rng.gen = row[2];
    e->np_rng.pos = CE_RNG_COUNTER_GEN;
    return CE_OK;
  }
  const uint32_t* rw = o->b.rng + (size_t)ei * (o->kind == CE_KIND_SELFDRIVE ? CE_RNG_WORDS_SELFDRIVE : CE_RNG_WORDS_GRID);
  memcpy(e->np_rng.key, rw, 624 * 4);
  e->np_rng.pos = rw[624];
  if (o->kind == CE_KIND_SELFDRIVE) {
    memcpy(e->py_rng.key, rw + CE_RNG_WORDS_GRID, 624 * 4);
    e->py_rng.pos = rw[CE_RNG_WORDS_GRID + 624];
  }
  return CE_OK;
}

/* raw generator access for the RNG known-answer tests */
void orc_philox4x32_10(const uint32_t key[2], const uint32_t counter[4], uint32_t out[4]) {
  memcpy(out, counter, 16);
  philox4x32_10(key[0], key[1], out);
}
/* the first `count` words of an env's counter-mode stream as the draw sites see them (generation 1 onwards, tempered) */
void orc_counter_stream(uint64_t seed, uint32_t* out, int count) {
  mt_t m;
  memset(&m, 0, sizeof(m));
  m.ctr = 1;
  m.k0 = (uint32_t)seed;
  m.k1 = (uint32_t)(seed >> 32);
  m.pos = CE_RNG_COUNTER_GEN;
  for (int i = 0; i < count; i++) out[i] = mt_next(&m);
}
void orc_rng_words(uint32_t seed, int python_seeding, uint32_t* out, int count) {
  mt_t m;
  if (python_seeding)
    mt_init_by_array(&m, &seed, 1);
  else
    mt_init_genrand(&m, seed);
  for (int i = 0; i < count; i++) out[i] = mt_next(&m);
}
void orc_rng_shuffle(uint32_t seed, uint8_t* x, int len, int skip_words) {
  mt_t m;
  mt_init_genrand(&m, seed);
  for (int i = 0; i < skip_words; i++) mt_next(&m);
  shuffle_u8(&m, x, len);
}
double orc_rng_double(uint32_t seed, int python_seeding, int index) {
  mt_t m;
  if (python_seeding)
    mt_init_by_array(&m, &seed, 1);
  else
    mt_init_genrand(&m, seed);
  double d = 0;
  for (int i = 0; i <= index; i++) d = mt_double(&m);
  return d;
}

/* ------------------------------------------------------------------------- */
/* The engine's own entry-point names over this library ("device = cpu"): the core of the C-ABI in                 */
/* include/contracts_engine.h — create / seed / reset / step / step_range / rollout / buffers / download / upload / flags /    */
/* contract — with the SAME prototypes, so that a C harness written against libcontracts_engine.so can be linked against       */
/* liboracle.so instead and run the same calls on the host.  Every pointer is a HOST pointer, `stream`    */
/* arguments are ignored (every call is synchronous), layouts are the ones ce_get_buffers reports here (dense 15 x 15 x 3     */
/* views, the unpadded map image).  Still test infrastructure: the product never loads this file.  Entry points of the device  */
/* boundary that have no host meaning (page-locked staging, hipGraph-friendly slicing, fused rollouts, write-through budget,   */
/* the device-layout snapshot) are not restated.                                                                               */
/* ------------------------------------------------------------------------- */
int ce_abi_version(void) { return CE_ABI_VERSION; }
int ce_device_count(void) { return 1; } /* the host */
int ce_create(const ce_config* cfg, ce_handle* out) { return orc_create(cfg, (orc_t**)out); }
int ce_destroy(ce_handle h) { return orc_destroy((orc_t*)h); }
int ce_seed(ce_handle h, const uint64_t* seeds, uint64_t seed0, const uint8_t* mask, int mode) {
  if (!h || (mode & ~(CE_SEED_RESEED | CE_SEED_CONSTRUCT)) || mode == 0) return CE_EINVAL;
  return orc_seed((orc_t*)h, seeds, seed0, mask, mode);
}
int ce_reset(ce_handle h, const uint8_t* mask, void* stream) {
  (void)stream;
  return orc_reset((orc_t*)h, mask);
}
/* (calls between the entry points below go through static functions: with the engine loaded RTLD_GLOBAL in the same process —
 * the parity tests load both — a call through the PLT would land in the ENGINE's function of the same name) */
static int step_range_impl(orc_t* o, const void* actions, const uint8_t* active, uint32_t env_begin, uint32_t env_count) {
  if (!o || !actions || env_count == 0 || (uint64_t)env_begin + env_count > o->cfg.num_envs) return CE_EINVAL;
  int n = o->n;
#pragma omp parallel for schedule(static)
  for (long ei = (long)env_begin; ei < (long)(env_begin + env_count); ei++) {
    if (is_feat_kind(o->kind))
      feat_step(o, (int)ei, (const uint8_t*)actions + (size_t)ei * n);
    else if (o->kind == CE_KIND_SELFDRIVE)
      sd_step(o, (int)ei, (const float*)actions + (size_t)ei * n, active ? active + (size_t)ei * n : NULL);
    else
      grid_step(o, (int)ei, (const uint8_t*)actions + (size_t)ei * n);
  }
  return CE_OK;
}
int ce_step_range(ce_handle h, const void* actions, const uint8_t* active, uint32_t env_begin, uint32_t env_count, void* stream) {
  (void)stream;
  return step_range_impl((orc_t*)h, actions, active, env_begin, env_count);
}
int ce_step(ce_handle h, const void* actions, const uint8_t* active, void* stream) {
  (void)stream;
  if (!h) return CE_EINVAL;
  return step_range_impl((orc_t*)h, actions, active, 0, ((orc_t*)h)->cfg.num_envs);
}
int ce_step_host(ce_handle h, const void* host_actions, const uint8_t* host_active, void* stream) {
  (void)stream;
  if (!h) return CE_EINVAL;
  return step_range_impl((orc_t*)h, host_actions, host_active, 0, ((orc_t*)h)->cfg.num_envs);
}
int ce_rollout(ce_handle h, const void* actions, uint32_t num_steps, uint32_t num_slices, void* const* streams) {
  (void)streams;
  orc_t* o = (orc_t*)h;
  if (!o || !actions || num_steps == 0 || num_slices == 0 || num_slices > o->cfg.num_envs) return CE_EINVAL;
  const size_t plane = (size_t)o->cfg.num_envs * o->n * (o->kind == CE_KIND_SELFDRIVE ? 4 : 1);
  for (uint32_t t = 0; t < num_steps; t++) {
    int rc = step_range_impl(o, (const char*)actions + (size_t)t * plane, NULL, 0, o->cfg.num_envs);
    if (rc != CE_OK) return rc;
  }
  return CE_OK;
}
int ce_get_buffers(ce_handle h, ce_buffers* out) { return orc_get_buffers((orc_t*)h, out); }
int ce_synchronize(ce_handle h, void* stream) {
  (void)stream;
  return h ? CE_OK : CE_EINVAL;
}
int ce_set_contract(ce_handle h, uint32_t contract, double contract_low, double contract_high, double null_prob) {
  orc_t* o = (orc_t*)h;
  if (!o) return CE_EINVAL;
  const int ok = contract == CE_CONTRACT_NONE ||
                 (contract == CE_CONTRACT_CLEANUP && (o->kind == CE_KIND_CLEANUP || o->kind == CE_KIND_CLEANUP_FEATURES)) ||
                 (contract == CE_CONTRACT_HARVEST_LOCAL && (o->kind == CE_KIND_HARVEST || o->kind == CE_KIND_HARVEST_FEATURES)) ||
                 (contract == CE_CONTRACT_SELFDRIVE_DISTPROP && o->kind == CE_KIND_SELFDRIVE);
  if (!ok) return CE_EINVAL;
  o->cfg.contract = contract;
  o->cfg.contract_low = contract_low;
  o->cfg.contract_high = contract_high;
  o->cfg.null_prob = null_prob;
  return CE_OK;
}
int ce_set_flags(ce_handle h, uint32_t mask, uint32_t value) {
  orc_t* o = (orc_t*)h;
  if (!o || (mask & ~(CE_FLAG_AUTO_RESET | CE_FLAG_EXTERNAL_THETA | CE_FLAG_BEAM_TRACE))) return CE_EINVAL;
  if ((mask & value & CE_FLAG_BEAM_TRACE) && o->kind != CE_KIND_CLEANUP && o->kind != CE_KIND_HARVEST) return CE_EINVAL;
  o->cfg.flags = (o->cfg.flags & ~mask) | (value & mask);
  return CE_OK;
}
typedef struct {
  const char* name;
  void* base;
  size_t env_bytes;
} orc_field_t;
static int orc_find_field(orc_t* o, const char* name, orc_field_t* out) {
  const ce_buffers* b = &o->b;
  const size_t n = (size_t)o->n;
  const orc_field_t fields[] = {
      {"grid", b->grid, (size_t)b->grid_env_stride},
      {"agents", b->agents, n * 4},
      {"spawn_perm", b->spawn_perm, 20},
      {"waste_perm", b->waste_perm, 119},
      {"rng", b->rng, (size_t)b->rng_words * 4},
      {"timestep", b->timestep, 4},
      {"theta", b->theta, 8},
      {"sd_state", b->sd_state, CE_SD_STATE_DOUBLES(n) * 8},
      {"obs", b->obs, b->obs_env_stride},
      {"obs_f64", b->obs_f64, n * (2 * n + 7) * 8},
      {"base_reward", b->base_reward, n * 4},
      {"reward", b->reward, n * 8},
      {"done", b->done, 1},
      {"done_agents", b->done_agents, n},
      {"info", b->info, n * 2},
      {"features", b->features, n * b->num_features * 2},
      {"int_metrics", b->int_metrics, (size_t)b->num_int_metrics * 8},
      {"f64_metrics", b->f64_metrics, (size_t)b->num_f64_metrics * 8},
      {"final_int_metrics", b->final_int_metrics, (size_t)b->num_int_metrics * 8},
      {"final_f64_metrics", b->final_f64_metrics, (size_t)b->num_f64_metrics * 8},
      {"error_flags", b->error_flags, 4},
      {"beam_map", b->beam_map, (size_t)b->grid_h * b->grid_w},
      {"sd_info", b->sd_info, 16},
  };
  for (size_t i = 0; i < sizeof(fields) / sizeof(fields[0]); i++)
    if (strcmp(fields[i].name, name) == 0) {
      if (!fields[i].base || !fields[i].env_bytes) return 0;
      *out = fields[i];
      return 1;
    }
  return 0;
}
int ce_download(ce_handle h, const char* field, uint32_t env_begin, uint32_t env_count, void* dst, uint64_t dst_bytes) {
  orc_t* o = (orc_t*)h;
  orc_field_t f;
  if (!o || !field || !dst) return CE_EINVAL;
  if (!orc_find_field(o, field, &f)) {
    snprintf(o->err, sizeof(o->err), "unknown or absent field");
    return CE_EINVAL;
  }
  if ((uint64_t)env_begin + env_count > o->cfg.num_envs || dst_bytes < (uint64_t)env_count * f.env_bytes) {
    snprintf(o->err, sizeof(o->err), "slice out of range");
    return CE_EINVAL;
  }
  memcpy(dst, (const char*)f.base + (size_t)env_begin * f.env_bytes, (size_t)env_count * f.env_bytes);
  return CE_OK;
}
/* state fields only: what is uploaded is imported into the envs' working state (orc_import_state), as ce_upload on the engine
 * replaces the device rows a step reads */
int ce_upload(ce_handle h, const char* field, uint32_t env_begin, uint32_t env_count, const void* src, uint64_t src_bytes) {
  orc_t* o = (orc_t*)h;
  orc_field_t f;
  if (!o || !field || !src) return CE_EINVAL;
  static const char* const state[] = {"grid", "agents", "spawn_perm", "waste_perm", "rng", "timestep", "theta", "sd_state"};
  int is_state = 0;
  for (size_t i = 0; i < sizeof(state) / sizeof(state[0]); i++) is_state |= strcmp(state[i], field) == 0;
  if (!is_state || !orc_find_field(o, field, &f)) {
    snprintf(o->err, sizeof(o->err), "unknown, absent or not a state field");
    return CE_EINVAL;
  }
  if ((uint64_t)env_begin + env_count > o->cfg.num_envs || src_bytes < (uint64_t)env_count * f.env_bytes) {
    snprintf(o->err, sizeof(o->err), "slice out of range");
    return CE_EINVAL;
  }
  memcpy((char*)f.base + (size_t)env_begin * f.env_bytes, src, (size_t)env_count * f.env_bytes);
  for (uint32_t e = env_begin; e < env_begin + env_count; e++) {
    int rc = orc_import_state(o, e);
    if (rc != CE_OK) return rc;
  }
  return CE_OK;
}
const char* ce_last_error(ce_handle h) { return h ? ((orc_t*)h)->err : "null handle"; }
