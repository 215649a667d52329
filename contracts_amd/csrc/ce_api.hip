// ce_api.hip — host side of the C-ABI declared in include/contracts_engine.h.
// Owns device memory, builds the static map tables, validates arguments and launches the
// gfx950 kernels.  No torch types, no exceptions across the boundary, no global mutable
// state besides the per-process constant tables (idempotent uploads).
#include <hip/hip_runtime.h>

#if defined(__x86_64__)
#include <immintrin.h>  // streaming stores of the host-side conversions; other hosts take the plain-store loops below
#define CE_HOST_X86 1
#endif

#include <pthread.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#define CE_PLAIN_PARAM_POINTERS 1  // host side: ordinary pointers in the parameter blocks
#include "ce_device.h"

using namespace ce;

// ----------------------------------------------------------------------------------------
// the two maps (the engine's own data; cells as in environments/cleanup_new.py:10-36 and
// environments/harvest_new.py:10-27: '@' wall, 'P' spawn, 'B' cleanup apple area, 'H' waste,
// 'R' river, 'S' stream, 'A' harvest apple)
// ----------------------------------------------------------------------------------------
static const char* kCleanupMap[25] = {
    "@@@@@@@@@@@@@@@@@@", "@RRRRRR     BBBBB@", "@HHHHHH      BBBB@", "@RRRRRR     BBBBB@", "@RRRRR  P    BBBB@",
    "@RRRRR    P BBBBB@", "@HHHHH       BBBB@", "@RRRRR      BBBBB@", "@HHHHHHSSSSSSBBBB@", "@HHHHHHSSSSSSBBBB@",
    "@RRRRR   P P BBBB@", "@HHHHH   P  BBBBB@", "@RRRRRR    P BBBB@", "@HHHHHH P   BBBBB@", "@RRRRR       BBBB@",
    "@HHHH    P  BBBBB@", "@RRRRR       BBBB@", "@HHHHH  P P BBBBB@", "@RRRRR       BBBB@", "@HHHH       BBBBB@",
    "@RRRRR       BBBB@", "@HHHHH      BBBBB@", "@RRRRR       BBBB@", "@HHHH       BBBBB@", "@@@@@@@@@@@@@@@@@@"};
static const char* kHarvestMap[16] = {
    "@@@@@@@@@@@@@@@@@@@@@@@@@@@@@@@@@@@@@@", "@ P   P      A    P AAAAA    P  A P  @",
    "@  P     A P AA    P    AAA    A  A  @", "@     A AAA  AAA    A    A AA AAAA   @",
    "@ A  AAA A    A  A AAA  A  A   A A   @", "@AAA  A A    A  AAA A  AAA        A P@",
    "@ A A  AAA  AAA  A A    A AA   AA AA @", "@  A A  AAA    A A  AAA    AAA  A    @",
    "@   AAA  A      AAA  A    AAAA       @", "@ P  A       A  A AAA    A  A      P @",
    "@A  AAA  A  A  AAA A    AAAA     P   @", "@    A A   AAA  A A      A AA   A  P @",
    "@     AAA   A A  AAA      AA   AAA P @", "@ A    A     AAA  A  P          A    @",
    "@       P     A         P  P P     P @", "@@@@@@@@@@@@@@@@@@@@@@@@@@@@@@@@@@@@@@"};

// colours: DEFAULT_COLOURS (map_env.py:24-42) + CLEANUP_COLORS (cleanup_new.py:42-47), packed 0x00BBGGRR
static uint32_t rgb(uint32_t r, uint32_t g, uint32_t b) { return r | (g << 8) | (b << 16); }

// The static tables of a layout: the kind's shipped one (rows == nullptr) or a caller's (ce_config.ascii_map; H0 x W0 characters,
// row-major).  Returns nullptr, or the rule of contracts_engine.h the layout breaks.  `counts` = {apple, waste, spawn} cells.
template <int KIND> static const char* build_tables(GridTables& t, const char* rows, int H0, int W0, uint32_t num_agents, uint32_t counts[3]) {
  typedef Geo<KIND> G;
  const char** map = KIND == CE_KIND_CLEANUP ? kCleanupMap : kHarvestMap;
  std::memset(&t, 0, sizeof(t));
  if (!rows) {
    H0 = G::H;
    W0 = G::W;
  } else if (H0 < 1 || W0 < 1 || H0 > G::H || W0 > G::W) {
    return "ascii_map: the layout does not fit the kind's frame (25 x 18 cleanup, 16 x 38 harvest)";
  }
  int na = 0, nw = 0, ns = 0;
  for (int r = 0; r < H0; ++r)
    for (int c = 0; c < W0; ++c) {
      const char ch = rows ? rows[(size_t)r * W0 + c] : map[r][c];
      if (rows) {
        if ((r == 0 || c == 0 || r == H0 - 1 || c == W0 - 1) && ch != '@') return "ascii_map: the layout must be walled in ('@' on its whole perimeter)";
        const bool known = ch == '@' || ch == ' ' || ch == 'P' || (KIND == CE_KIND_CLEANUP ? (ch == 'B' || ch == 'H' || ch == 'R' || ch == 'S') : ch == 'A');
        if (!known) return "ascii_map: a character outside the kind's alphabet";
        if (ch == 'P' && ns >= G::NSPAWN_CTOR) return "ascii_map: more spawn points than the shipped layout (10 cleanup, 20 harvest)";
        if (ch == (KIND == CE_KIND_CLEANUP ? 'B' : 'A') && na >= G::NAPPLE) return "ascii_map: more apple cells than the shipped layout (103 cleanup, 155 harvest)";
        if (KIND == CE_KIND_CLEANUP && (ch == 'H' || ch == 'R') && nw >= G::NWASTE) return "ascii_map: more waste cells than the shipped layout (119)";
      }
      const uint32_t pad = (uint32_t)((r + kView) * G::PW + c + kView);
      const uint32_t packed = pad | ((uint32_t)c << 16) | ((uint32_t)r << 24);
      uint8_t code = CE_CELL_EMPTY;
      if (ch == '@') code = CE_CELL_WALL;
      if (ch == 'P') t.spawn[ns++] = packed;
      if (KIND == CE_KIND_CLEANUP) {
        if (ch == 'B') t.apple[na++] = packed;
        if (ch == 'H' || ch == 'R') t.waste[nw++] = packed;
        if (ch == 'H') code = CE_CELL_WASTE;
        if (ch == 'R') code = CE_CELL_RIVER;
        if (ch == 'S') code = CE_CELL_STREAM;
      } else if (ch == 'A') {
        t.apple[na++] = packed;
        code = CE_CELL_APPLE;  // custom_reset: every apple cell starts as 'A' (harvest_new.py:143-146)
      }
      t.base_pmap[pad] = code;
      t.base_pmap4[pad] = (uint8_t)(code << 2);
    }
  if (rows && (ns < 1 || na < 1 || (KIND == CE_KIND_CLEANUP && nw < 1))) return "ascii_map: the layout needs at least one spawn point, one apple cell and (cleanup) one waste cell";
  if (rows && num_agents > (uint32_t)ns) return "ascii_map: more agents than spawn points (the reference asserts 'not enough spawn points', map_env.py:826)";
  counts[0] = (uint32_t)na;
  counts[1] = (uint32_t)nw;
  counts[2] = (uint32_t)ns;
  if (KIND == CE_KIND_CLEANUP)
    for (int i = 0; i < ns; ++i) t.spawn[ns + i] = t.spawn[i];  // cleanup_new.py:114-115
  if (KIND == CE_KIND_CLEANUP) {
    // compute_probabilities (cleanup_new.py:351-368) for every possible #H, as exact 53-bit
    // thresholds: rand < p  <=>  X < ceil(p * 2^53) for the 53-bit integer X behind rand
    const int potential = nw;
    for (int nH = 0; nH <= potential; ++nH) {
      volatile double free_area = (double)(potential - nH);
      volatile double density = 1 - free_area / (double)potential;
      double p_apple, p_waste;
      if (density >= 0.4) {
        p_apple = 0;
        p_waste = 0;
      } else {
        p_waste = 0.5;
        if (density <= 0.0) {
          p_apple = 0.05;
        } else {
          volatile double frac = (density - 0.0) / (0.4 - 0.0);
          volatile double one_minus = 1 - frac;
          p_apple = one_minus * 0.05;
        }
      }
      t.apple_thresh[nH] = (uint64_t)std::ceil(std::ldexp(p_apple, 53));
      t.waste_on[nH] = p_waste != 0;
      if (p_waste != 0) t.apple_thresh[nH] |= kWasteOnBit;  // one scalar load yields both (see ce_device.h)
    }
  } else {
    const double spawn_prob[4] = {0, 0.005, 0.02, 0.05};  // SPAWN_PROB harvest_new.py:34
    for (int k = 0; k < 4; ++k) t.apple_thresh[k] = (uint64_t)std::ceil(std::ldexp(spawn_prob[k], 53));
    int k = 0;
    for (int j = -5; j <= 5; ++j)
      for (int kk = -5; kk <= 5; ++kk)
        if (j * j + kk * kk <= 5) t.close_off[k++] = (uint32_t)(int32_t)(j * G::PW + kk);  // harvest_new.py:326-336
  }
  return nullptr;
}

// ----------------------------------------------------------------------------------------
struct ce_engine {
  ce_config cfg;
  ce_buffers buf;  // device pointers
  int grid_stride;
  uint64_t* d_seeds;
  uint8_t* d_mask;
  uint8_t* d_stage_actions;  // E*n*4 bytes
  uint8_t* d_stage_active;   // E*n
  unsigned long long* d_debug;  // E*16 phase stamps (diagnostic builds)
  char* d_gather;               // ce_download_many: device-side staging of small requests (grown on demand)
  size_t gather_bytes;
  std::vector<char> h_gather;   // ... and its host landing buffer
  GridParams* d_gparams;        // device copy of the grid kernels' parameter block
  // a caller's layout (ce_config.ascii_map): the handle's own tables in HBM, the lengths of its cell lists, its text
  GridTables* d_tab;
  GridTables* h_tab;
  uint32_t map_counts[3];       // apple, waste, spawn cells
  uint32_t map_h, map_w;
  std::string map_text;
  std::vector<hipEvent_t> obs_events;  // ce_download_obs_f64: one behind each part of the copy
  std::vector<std::pair<void**, size_t>> allocs;
  std::string err;
  // timing
  hipEvent_t ev_start, ev_stop;
  hipEvent_t ev_mask;  // recorded behind the last launch that reads d_mask: the staging buffer is reused only after it
  bool mask_in_flight;
  // ce_reset(stream) followed by steps on OTHER streams (env slices on their own streams): those streams wait for the
  // reset once (hipStreamWaitEvent), so the common "reset, then roll out on side streams" sequence needs no host sync
  hipEvent_t ev_reset;
  void* reset_stream;
  uint64_t reset_gen;
  std::vector<std::pair<void*, uint64_t>> reset_seen;  // (stream, generation it has already waited for)
  bool timing_armed;
  uint32_t timed_launches;
  // write-through decision (obs_write_through): the handle's device bytes, summed once at ce_create, against the share of the
  // last-level cache the integrator grants this handle (ce_set_cache_budget; default from CE_OBS_WT_MAX_BYTES / the device)
  unsigned long long device_bytes;
  unsigned long long cache_budget;
};

static int fail(ce_engine* h, int code, const char* what, hipError_t e = hipSuccess) {
  if (h) {
    h->err = what;
    if (e != hipSuccess) {
      h->err += ": ";
      h->err += hipGetErrorString(e);
    }
  }
  return code;
}

template <class T> static int dalloc(ce_engine* h, T** p, size_t count) {
  const size_t bytes = (count ? count : 1) * sizeof(T);
  hipError_t e = hipMalloc((void**)p, bytes);
  if (e != hipSuccess) return fail(h, CE_ENOMEM, "hipMalloc", e);
  e = hipMemset(*p, 0, bytes);
  if (e != hipSuccess) return fail(h, CE_ENODEV, "hipMemset", e);
  h->allocs.push_back({(void**)p, bytes});
  return CE_OK;
}

extern "C" int ce_abi_version(void) { return CE_ABI_VERSION; }

extern "C" int ce_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return CE_ENODEV;
  int ok = 0;
  for (int i = 0; i < n; ++i) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, i) == hipSuccess && std::strncmp(prop.gcnArchName, "gfx950", 6) == 0) ++ok;
  }
  return ok;
}

struct ce_engine;
static int sync_device_params(ce_engine* h);
static unsigned long long default_cache_budget(int device);

static bool is_grid(const ce_config& c) { return c.kind == CE_KIND_CLEANUP || c.kind == CE_KIND_HARVEST; }
static bool is_feat(const ce_config& c) { return c.kind == CE_KIND_HARVEST_FEATURES || c.kind == CE_KIND_CLEANUP_FEATURES; }
static bool counter_rng(const ce_config& c) { return (c.flags & CE_FLAG_RNG_COUNTER) != 0; }  // grid kinds only (ce_create)
static bool u8_actions(const ce_config& c) { return is_grid(c) || is_feat(c); }  // one byte per agent (selfdrive: float32)

static int contract_ok(uint32_t kind, uint32_t contract) {
  if (contract == CE_CONTRACT_NONE) return 1;
  if (kind == CE_KIND_CLEANUP) return contract == CE_CONTRACT_CLEANUP;
  if (kind == CE_KIND_HARVEST || kind == CE_KIND_HARVEST_FEATURES) return contract == CE_CONTRACT_HARVEST_LOCAL;
  if (kind == CE_KIND_CLEANUP_FEATURES) return contract == CE_CONTRACT_CLEANUP;
  return contract == CE_CONTRACT_SELFDRIVE_DISTPROP;
}

static uint32_t grid_napple(const ce_config& c) { return c.kind == CE_KIND_CLEANUP ? Geo<CE_KIND_CLEANUP>::NAPPLE : Geo<CE_KIND_HARVEST>::NAPPLE; }
static uint32_t grid_nwaste(const ce_config& c) { return c.kind == CE_KIND_CLEANUP ? Geo<CE_KIND_CLEANUP>::NWASTE : Geo<CE_KIND_HARVEST>::NWASTE; }

extern "C" int ce_create(const ce_config* cfg, ce_handle* out) {
  if (!cfg || !out) return CE_EINVAL;
  *out = nullptr;
  if (cfg->abi_version != CE_ABI_VERSION || cfg->kind > CE_KIND_CLEANUP_FEATURES || cfg->num_envs == 0) return CE_EINVAL;
  const uint32_t maxn = cfg->kind == CE_KIND_SELFDRIVE ? 10 : kMaxGridAgents;
  if (cfg->num_agents < 1 || cfg->num_agents > maxn) return CE_EINVAL;
  if (!contract_ok(cfg->kind, cfg->contract)) return CE_EINVAL;
  if (is_feat(*cfg) && (cfg->flags & (CE_FLAG_COLLECTIVE_REWARD | CE_FLAG_INEQUITY_AVERSE | CE_FLAG_FIRING_ENABLED))) return CE_EINVAL;
  if ((cfg->flags & CE_FLAG_INEQUITY_AVERSE) && cfg->num_agents < 2) return CE_EINVAL;  // map_env.py:294 assertion
  if ((cfg->flags & CE_FLAG_BEAM_TRACE) && !is_grid(*cfg)) return CE_EINVAL;
  if ((cfg->flags & CE_FLAG_RNG_COUNTER) && !is_grid(*cfg)) return CE_EINVAL;  // the other kinds draw from CPython's `random` too

  ce_engine* h = new (std::nothrow) ce_engine();
  if (!h) return CE_ENOMEM;
  h->cfg = *cfg;
  if (h->cfg.horizon == 0) h->cfg.horizon = 1000;
  h->timing_armed = false;
  h->device_bytes = 0;
  h->cache_budget = 0;
  h->mask_in_flight = false;
  h->ev_reset = nullptr;
  h->reset_stream = nullptr;
  h->reset_gen = 0;
  h->timed_launches = 0;
  h->d_seeds = nullptr;
  h->d_mask = nullptr;
  h->d_stage_actions = nullptr;
  h->d_stage_active = nullptr;
  h->d_debug = nullptr;
  h->d_gather = nullptr;
  h->gather_bytes = 0;
  h->d_gparams = nullptr;
  h->d_tab = nullptr;
  h->h_tab = nullptr;
  h->map_h = h->map_w = 0;
  h->map_counts[0] = h->map_counts[1] = h->map_counts[2] = 0;
  h->cfg.ascii_map = nullptr;  // (the caller's string is copied below, never kept)
  std::memset(&h->buf, 0, sizeof(h->buf));
  *out = h;  // handed out even on failure so ce_last_error works; caller must ce_destroy
  if (cfg->ascii_map) {  // a caller's layout: checked (and its tables built) before the device is touched
    if (!is_grid(*cfg)) return fail(h, CE_EINVAL, "ascii_map belongs to the grid kinds (cleanup_new / harvest_new)");
    // the caller's dimensions are checked against the kind's frame BEFORE anything is read through the pointer (a C caller
    // with garbage dimensions gets CE_EINVAL, not an out-of-bounds read or an exception through the C boundary)
    const uint32_t fh = cfg->kind == CE_KIND_CLEANUP ? Geo<CE_KIND_CLEANUP>::H : Geo<CE_KIND_HARVEST>::H;
    const uint32_t fw = cfg->kind == CE_KIND_CLEANUP ? Geo<CE_KIND_CLEANUP>::W : Geo<CE_KIND_HARVEST>::W;
    if (cfg->map_rows < 1 || cfg->map_cols < 1 || cfg->map_rows > fh || cfg->map_cols > fw)
      return fail(h, CE_EINVAL, "ascii_map: the layout does not fit the kind's frame (25 x 18 cleanup, 16 x 38 harvest)");
    h->h_tab = new (std::nothrow) GridTables();
    if (!h->h_tab) return fail(h, CE_ENOMEM, "layout tables");
    try {
      h->map_text.assign(cfg->ascii_map, (size_t)cfg->map_rows * cfg->map_cols);
    } catch (const std::bad_alloc&) {
      return fail(h, CE_ENOMEM, "layout text");
    } catch (const std::length_error&) {
      return fail(h, CE_EINVAL, "ascii_map: dimensions out of range");
    }
    const char* why = cfg->kind == CE_KIND_CLEANUP
                          ? build_tables<CE_KIND_CLEANUP>(*h->h_tab, h->map_text.data(), (int)cfg->map_rows, (int)cfg->map_cols, cfg->num_agents, h->map_counts)
                          : build_tables<CE_KIND_HARVEST>(*h->h_tab, h->map_text.data(), (int)cfg->map_rows, (int)cfg->map_cols, cfg->num_agents, h->map_counts);
    if (why) return fail(h, CE_EINVAL, why);
    h->map_h = cfg->map_rows;
    h->map_w = cfg->map_cols;
  }

  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0) return fail(h, CE_ENODEV, "no HIP device visible (this engine has no CPU path)", e);
  if (cfg->device < 0 || cfg->device >= ndev) return fail(h, CE_EINVAL, "device ordinal out of range");
  if ((e = hipSetDevice(cfg->device)) != hipSuccess) return fail(h, CE_ENODEV, "hipSetDevice", e);
  hipDeviceProp_t prop;
  if ((e = hipGetDeviceProperties(&prop, cfg->device)) != hipSuccess) return fail(h, CE_ENODEV, "hipGetDeviceProperties", e);
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return fail(h, CE_ENODEV, "device is not gfx950 (MI355X); kernels are built for gfx950 only");
  if (hipEventCreate(&h->ev_start) != hipSuccess || hipEventCreate(&h->ev_stop) != hipSuccess ||
      hipEventCreateWithFlags(&h->ev_mask, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&h->ev_reset, hipEventDisableTiming) != hipSuccess)
    return fail(h, CE_ENODEV, "hipEventCreate");

  const size_t E = cfg->num_envs, n = cfg->num_agents;
  ce_buffers& b = h->buf;
  b.num_envs = cfg->num_envs;
  b.num_agents = cfg->num_agents;
  b.num_int_metrics = CE_MI_COUNT(n);
  b.num_f64_metrics = CE_MF_COUNT(n);
  int rc = CE_OK;
#define A(field, count) \
  if (rc == CE_OK) rc = dalloc(h, &b.field, (count))
  if (is_grid(*cfg)) {
    const bool cl = cfg->kind == CE_KIND_CLEANUP;
    b.grid_h = h->h_tab ? h->map_h : (uint32_t)(cl ? Geo<0>::H : Geo<1>::H);  // a caller's layout: its own shape, embedded at the
    b.grid_w = h->h_tab ? h->map_w : (uint32_t)(cl ? Geo<0>::W : Geo<1>::W);  // origin of the kind's frame (same strides / origin)
    h->grid_stride = cl ? Geo<0>::IMAGE_STRIDE : Geo<1>::IMAGE_STRIDE;  // of the ce_download / ce_upload image
    b.grid_env_stride = kGridStateBytes;                                  // of the packed state `grid` points to
    b.grid_row_stride = cl ? Geo<0>::PW : Geo<1>::PW;
    b.grid_origin = kView * b.grid_row_stride + kView;
    b.obs_row_stride = kObsRowStride;
    b.obs_agent_stride = kObsAgentStride;
    b.obs_env_stride = (uint32_t)(n * kObsAgentStride);
    b.num_features = (uint32_t)(cl ? 12 + n : 10 + 2 * n);
    b.rng_words = counter_rng(*cfg) ? CE_RNG_WORDS_COUNTER : CE_RNG_WORDS_GRID;
    A(grid, E * kGridStateBytes);
    A(agents, E * n * 4);
    A(spawn_perm, E * 20);
    A(waste_perm, E * 119 + 8);
    A(rng, E * b.rng_words);
    A(obs, E * b.obs_env_stride + 16);
    A(features, E * n * b.num_features);
    A(beam_map, E * b.grid_h * b.grid_w);
    A(actions_taken, E * n);
    if (h->h_tab) {
      if (rc == CE_OK) rc = dalloc(h, &h->d_tab, 1);
      if (rc == CE_OK && hipMemcpy(h->d_tab, h->h_tab, sizeof(GridTables), hipMemcpyHostToDevice) != hipSuccess)
        rc = fail(h, CE_ENODEV, "layout table upload failed");
    }
  } else if (is_feat(*cfg)) {
    const bool cl = cfg->kind == CE_KIND_CLEANUP_FEATURES;
    b.grid_h = cl ? Geo<0>::H : Geo<1>::H;
    b.grid_w = cl ? Geo<0>::W : Geo<1>::W;
    h->grid_stride = CE_FEAT_STATE_BYTES;
    b.grid_env_stride = CE_FEAT_STATE_BYTES;
    b.num_features = (uint32_t)(cl ? 12 + n : 10 + 2 * n);
    b.rng_words = CE_RNG_WORDS_SELFDRIVE;
    A(grid, E * CE_FEAT_STATE_BYTES);
    A(agents, E * n * 4);
    A(rng, E * CE_RNG_WORDS_SELFDRIVE);
    A(features, E * n * b.num_features);
  }
  if (is_grid(*cfg) || is_feat(*cfg)) {
    // static tables (process-wide constants; re-uploading identical bytes is harmless)
    if (rc == CE_OK) {
      struct BothTables {
        GridTables cleanup, harvest;
        BothTables() {
          uint32_t cnt[3];
          build_tables<CE_KIND_CLEANUP>(cleanup, nullptr, 0, 0, 0, cnt);
          build_tables<CE_KIND_HARVEST>(harvest, nullptr, 0, 0, 0, cnt);
        }
      };
      static const BothTables built;  // once per process (thread-safe static); uploaded to this handle's device below
      const GridTables &t0 = built.cleanup, &t1 = built.harvest;
      const uint32_t lut[16] = {rgb(0, 0, 0),       rgb(180, 180, 180), rgb(0, 255, 0),     rgb(99, 156, 194),
                                rgb(113, 75, 24),   rgb(113, 75, 24),   rgb(0, 0, 255),     rgb(2, 81, 154),
                                rgb(204, 0, 204),   rgb(216, 30, 54),   rgb(254, 151, 0),   rgb(100, 255, 255),
                                rgb(99, 99, 255),   rgb(250, 204, 255), rgb(238, 223, 16),  0};
      if (upload_grid_tables(CE_KIND_CLEANUP, t0, lut) ||
          upload_grid_tables(CE_KIND_HARVEST, t1, lut))
        rc = fail(h, CE_ENODEV, "constant table upload failed");
      // (the counter-mode kernels are a translation unit of their own, with their own copies of the tables)
      if (rc == CE_OK && counter_rng(*cfg) &&
          (upload_grid_tables_ctr(CE_KIND_CLEANUP, t0, lut) || upload_grid_tables_ctr(CE_KIND_HARVEST, t1, lut)))
        rc = fail(h, CE_ENODEV, "constant table upload failed");
    }
  }
  if (!is_grid(*cfg) && !is_feat(*cfg)) {
    b.num_features = (uint32_t)(2 * n + 7);
    b.rng_words = CE_RNG_WORDS_SELFDRIVE;
    A(rng, E * CE_RNG_WORDS_SELFDRIVE);
    A(sd_state, E * CE_SD_STATE_DOUBLES(n));
    A(obs_f64, E * n * (2 * n + 7));
    A(done_agents, E * n);
    A(sd_info, E * 2);
  }
  A(timestep, E);
  A(theta, E);
  A(base_reward, E * n);
  A(reward, E * n);
  A(done, E);
  A(info, E * n * 2);
  A(int_metrics, E * CE_MI_COUNT(n));
  A(f64_metrics, E * CE_MF_COUNT(n));
  A(final_int_metrics, E * CE_MI_COUNT(n));
  A(final_f64_metrics, E * CE_MF_COUNT(n));
  A(error_flags, E);
  if (rc == CE_OK) rc = dalloc(h, &h->d_seeds, E);
  if (rc == CE_OK) rc = dalloc(h, &h->d_mask, E);
  if (rc == CE_OK) rc = dalloc(h, &h->d_stage_actions, E * n * 4);
  if (rc == CE_OK) rc = dalloc(h, &h->d_stage_active, E * n);
  if (rc == CE_OK) rc = dalloc(h, &h->d_debug, E * 16);
  if (rc == CE_OK && u8_actions(*cfg)) rc = dalloc(h, &h->d_gparams, 1);
  h->device_bytes = 0;
  for (const auto& a : h->allocs) h->device_bytes += a.second;
  h->cache_budget = default_cache_budget(cfg->device);
  if (rc == CE_OK && u8_actions(*cfg)) rc = sync_device_params(h);
#undef A
  return rc;
}

extern "C" int ce_destroy(ce_handle h) {
  if (!h) return CE_EINVAL;
  (void)hipSetDevice(h->cfg.device);
  (void)hipDeviceSynchronize();
  for (auto& a : h->allocs)
    if (*a.first) (void)hipFree(*a.first);
  if (h->d_gather) (void)hipFree(h->d_gather);
  if (h->ev_start) (void)hipEventDestroy(h->ev_start);
  if (h->ev_stop) (void)hipEventDestroy(h->ev_stop);
  if (h->ev_mask) (void)hipEventDestroy(h->ev_mask);
  if (h->ev_reset) (void)hipEventDestroy(h->ev_reset);
  for (hipEvent_t ev : h->obs_events) (void)hipEventDestroy(ev);
  delete h->h_tab;
  delete h;
  return CE_OK;
}

extern "C" int ce_set_contract(ce_handle h, uint32_t contract, double contract_low, double contract_high, double null_prob) {
  if (!h) return CE_EINVAL;
  if (!contract_ok(h->cfg.kind, contract)) return fail(h, CE_EINVAL, "contract does not belong to this env family");
  h->cfg.contract = contract;
  h->cfg.contract_low = contract_low;
  h->cfg.contract_high = contract_high;
  h->cfg.null_prob = null_prob;
  (void)hipSetDevice(h->cfg.device);
  return sync_device_params(h);
}

extern "C" int ce_set_flags(ce_handle h, uint32_t mask, uint32_t value) {
  if (!h) return CE_EINVAL;
  if (mask & ~(CE_FLAG_AUTO_RESET | CE_FLAG_EXTERNAL_THETA | CE_FLAG_BEAM_TRACE))
    return fail(h, CE_EINVAL, "only AUTO_RESET, EXTERNAL_THETA and BEAM_TRACE can change on a live handle");
  if ((mask & value & CE_FLAG_BEAM_TRACE) && !is_grid(h->cfg)) return fail(h, CE_EINVAL, "BEAM_TRACE belongs to the grid kinds");
  h->cfg.flags = (h->cfg.flags & ~mask) | (value & mask);
  (void)hipSetDevice(h->cfg.device);
  return sync_device_params(h);
}

// Single-step launches write their observations (and the MT19937 row) through the L2 (GridParams.obs_wt / SdParams.obs_wt ->
// the launch; write_obs, store_rng, sd_emit_obs) when what the call works on fits the 256 MB Infinity Cache with room to spare:
// the handle's device memory — state, outputs, generator rows — plus, for ce_rollout, the action planes the call will read.  A
// write-through store that lands in that cache is cheap and shortens the launch's end (no dirty lines left to write back); one
// that goes on to HBM is not.  Round 5, interleaved A/B, agent-steps/s: cleanup n = 8 x 16 384 envs (174 MB) + 4 %, closed loop
// + 20 %; C3 + 3-4 %; selfdrive n = 4 x 32 768 (217 MB) + 5 %; but cleanup n = 8 x 32 768 (348 MB) - 30 %, and the headline batch
// stepped through 1 000 resident action planes (175 + 131 MB) - 3 to - 8 %.  CE_OBS_WT_MAX_BYTES overrides the limit (0 = never).
// The budget is per handle (ce_set_cache_budget): the default assumes the handle has the cache to itself — what a bench or a
// sampler without a network on the same GPU has; an integrator whose policy network, other handles or other ranks share the
// cache passes the share it wants the env to assume (0 = never write through).
static unsigned long long default_cache_budget(int device) {
  if (const char* e = getenv("CE_OBS_WT_MAX_BYTES")) return (unsigned long long)atoll(e);
  // MI355X: 256 MiB Infinity Cache behind eight 4 MiB L2s; HIP reports the L2 only, so the last-level size is taken as
  // 64 x the L2 (256 MiB on gfx950) and an eighth of it is left to everything that is not this handle
  int l2 = 0;
  if (hipDeviceGetAttribute(&l2, hipDeviceAttributeL2CacheSize, device) != hipSuccess || l2 <= 0) l2 = 4 << 20;
  const unsigned long long llc = (unsigned long long)l2 * 64ull;
  return llc - llc / 8ull;  // 224 MiB on an MI355X
}
static bool obs_write_through(const ce_engine* h, unsigned long long extra_bytes = 0) {
  return h->device_bytes + extra_bytes <= h->cache_budget;
}
extern "C" int ce_set_cache_budget(ce_handle h, uint64_t bytes) {
  if (!h) return CE_EINVAL;
  h->cache_budget = bytes;
  (void)hipSetDevice(h->cfg.device);
  return h->d_gparams ? sync_device_params(h) : CE_OK;
}

static GridParams grid_params(ce_engine* h) {
  GridParams p;
  std::memset(&p, 0, sizeof(p));
  const ce_buffers& b = h->buf;
  p.grid = b.grid;
  p.agents = b.agents;
  p.spawn_perm = b.spawn_perm;
  p.waste_perm = b.waste_perm;
  p.rng = b.rng;
  p.timestep = b.timestep;
  p.theta = b.theta;
  p.obs = b.obs;
  p.base_reward = b.base_reward;
  p.reward = b.reward;
  p.done = b.done;
  p.info = b.info;
  p.features = b.features;
  p.int_metrics = b.int_metrics;
  p.f64_metrics = b.f64_metrics;
  p.final_int_metrics = b.final_int_metrics;
  p.final_f64_metrics = b.final_f64_metrics;
  p.error_flags = b.error_flags;
  p.debug = h->d_debug;
  p.beam_map = b.beam_map;
  p.actions_taken = b.actions_taken;
  p.tab = (decltype(p.tab))h->d_tab;
  p.custom_map = h->d_tab ? 1u : 0u;
  p.napple = h->map_counts[0];
  p.nwaste = h->map_counts[1];
  p.nspawn = h->map_counts[2];
  p.map_h = h->map_h;
  p.map_w = h->map_w;
  p.obs_wt = obs_write_through(h) ? 1u : 0u;
  p.E = h->cfg.num_envs;
  p.n = h->cfg.num_agents;
  p.horizon = h->cfg.horizon;
  p.contract = h->cfg.contract;
  p.flags = h->cfg.flags;
  p.obs_env_stride = b.obs_env_stride;
  p.num_features = b.num_features;
  p.contract_low = h->cfg.contract_low;
  p.contract_high = h->cfg.contract_high;
  p.null_prob = h->cfg.null_prob;
  p.alpha = h->cfg.alpha;
  p.beta = h->cfg.beta;
  return p;
}

// (re)uploads the parameter block the grid kernels read; called at create and whenever cfg changes
static int sync_device_params(ce_engine* h) {
  if (!h->d_gparams) return CE_OK;
  const GridParams p = grid_params(h);
  hipError_t e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy(h->d_gparams, &p, sizeof(p), hipMemcpyHostToDevice);
  if (e != hipSuccess) return fail(h, CE_ENODEV, "parameter block upload", e);
  return CE_OK;
}

static SdParams sd_params(ce_engine* h) {
  SdParams p;
  std::memset(&p, 0, sizeof(p));
  const ce_buffers& b = h->buf;
  p.sd_state = b.sd_state;
  p.rng = b.rng;
  p.theta = b.theta;
  p.obs_f64 = b.obs_f64;
  p.base_reward = b.base_reward;
  p.reward = b.reward;
  p.done = b.done;
  p.done_agents = b.done_agents;
  p.info = b.info;
  p.sd_info = b.sd_info;
  p.f64_metrics = b.f64_metrics;
  p.final_f64_metrics = b.final_f64_metrics;
  p.int_metrics = b.int_metrics;
  p.final_int_metrics = b.final_int_metrics;
  p.error_flags = b.error_flags;
  p.E = h->cfg.num_envs;
  p.n = h->cfg.num_agents;
  p.contract = h->cfg.contract;
  p.flags = h->cfg.flags;
  p.obs_wt = obs_write_through(h) ? 1u : 0u;
  p.contract_low = h->cfg.contract_low;
  p.contract_high = h->cfg.contract_high;
  p.null_prob = h->cfg.null_prob;
  p.low_bound = h->cfg.low_bound;
  p.high_bound = h->cfg.high_bound;
  p.start_vel = h->cfg.start_vel;
  p.start_vel_ambulance = h->cfg.start_vel_ambulance;
  return p;
}

// Every entry point that launches starts here: the calling thread's current device, and a clean per-thread error
// state — hipGetLastError is sticky, and an error some earlier, unrelated HIP call of this thread left behind (another
// library's, or one already reported) must not be blamed on the launches that follow.
static void begin_call(ce_engine* h) {
  (void)hipSetDevice(h->cfg.device);
  (void)hipGetLastError();
}

// make `stream` wait for the last ce_reset if that ran on a different stream and this one has not waited for it yet.
// A stream only counts as ordered once the wait was accepted: if it is refused (e.g. `stream` is being captured and the
// reset's event was recorded outside the capture) nothing is remembered and the caller reports the error.
static hipError_t order_after_reset(ce_engine* h, void* stream) {
  if (h->reset_gen == 0 || stream == h->reset_stream) return hipSuccess;
  for (auto& it : h->reset_seen) {
    if (it.first == stream) {
      if (it.second == h->reset_gen) return hipSuccess;
      const hipError_t e = hipStreamWaitEvent((hipStream_t)stream, h->ev_reset, 0);
      if (e == hipSuccess) it.second = h->reset_gen;
      return e;
    }
  }
  const hipError_t e = hipStreamWaitEvent((hipStream_t)stream, h->ev_reset, 0);
  if (e != hipSuccess) return e;
  if (h->reset_seen.size() >= 64) h->reset_seen.clear();
  h->reset_seen.emplace_back(stream, h->reset_gen);
  return hipSuccess;
}

static int check_launch(ce_engine* h, const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(h, CE_ENODEV, what, e);
  return CE_OK;
}

static int stage_mask(ce_engine* h, const uint8_t* mask, hipStream_t s, const uint8_t** dmask) {
  *dmask = nullptr;
  if (!mask) return CE_OK;
  hipError_t e = hipSuccess;
  if (h->mask_in_flight) {  // a masked launch on another stream may not have read the previous mask yet
    if ((e = hipEventSynchronize(h->ev_mask)) != hipSuccess) return fail(h, CE_ENODEV, "mask staging wait", e);
    h->mask_in_flight = false;
  }
  e = hipMemcpyAsync(h->d_mask, mask, h->cfg.num_envs, hipMemcpyHostToDevice, s);
  if (e != hipSuccess) return fail(h, CE_ENODEV, "mask upload", e);
  // the host buffer may be reused by the caller right away
  if ((e = hipStreamSynchronize(s)) != hipSuccess) return fail(h, CE_ENODEV, "mask upload sync", e);
  *dmask = h->d_mask;
  return CE_OK;
}

extern "C" int ce_seed(ce_handle h, const uint64_t* seeds, uint64_t seed0, const uint8_t* mask, int mode) {
  if (!h) return CE_EINVAL;
  begin_call(h);
  const uint32_t E = h->cfg.num_envs;
  std::vector<uint64_t> tmp;
  if (!seeds) {
    tmp.resize(E);
    for (uint32_t i = 0; i < E; ++i) tmp[i] = seed0 + h->cfg.env_index_base + i;
    seeds = tmp.data();
  }
  if (!counter_rng(h->cfg))  // (the counter stream's key is the whole 64-bit seed)
    for (uint32_t i = 0; i < E; ++i)
      if ((!mask || mask[i]) && seeds[i] > 0xffffffffull) return fail(h, CE_EINVAL, "seed must fit 32 bits (np.random.seed range)");
  hipError_t e = hipMemcpy(h->d_seeds, seeds, sizeof(uint64_t) * E, hipMemcpyHostToDevice);
  if (e != hipSuccess) return fail(h, CE_ENODEV, "seed upload", e);
  const uint8_t* dmask;
  int rc = stage_mask(h, mask, nullptr, &dmask);
  if (rc) return rc;
  if ((mode & (CE_SEED_RESEED | CE_SEED_CONSTRUCT)) == 0) return fail(h, CE_EINVAL, "ce_seed: empty mode");
  const bool reseed = mode & CE_SEED_RESEED, replay_constructor = mode & CE_SEED_CONSTRUCT;
  if (is_grid(h->cfg)) {
    const bool ctr = counter_rng(h->cfg);
    if (reseed && ctr) launch_seed_ctr(h->buf.rng, h->d_seeds, dmask, E, nullptr);
    if (reseed && !ctr) launch_mt_seed(h->buf.rng, CE_RNG_WORDS_GRID, 0, h->d_seeds, dmask, E, 0, nullptr);
    if (replay_constructor) {
      GridParams p = grid_params(h);
      p.mask = dmask;
      (ctr ? launch_grid_construct_ctr : launch_grid_construct)((int)h->cfg.kind, p, h->d_gparams, nullptr);
    }
  } else if (is_feat(h->cfg)) {
    if (reseed) launch_mt_seed(h->buf.rng, CE_RNG_WORDS_SELFDRIVE, 0, h->d_seeds, dmask, E, 0, nullptr);
    if (reseed) launch_mt_seed(h->buf.rng, CE_RNG_WORDS_SELFDRIVE, CE_RNG_WORDS_GRID, h->d_seeds, dmask, E, 1, nullptr);
    if (replay_constructor) {
      GridParams p = grid_params(h);
      p.mask = dmask;
      launch_feat_construct((int)h->cfg.kind, p, h->d_gparams, nullptr);
    }
  } else {
    if (reseed) launch_mt_seed(h->buf.rng, CE_RNG_WORDS_SELFDRIVE, 0, h->d_seeds, dmask, E, 0, nullptr);
    if (reseed) launch_mt_seed(h->buf.rng, CE_RNG_WORDS_SELFDRIVE, CE_RNG_WORDS_GRID, h->d_seeds, dmask, E, 1, nullptr);
    if (replay_constructor) {
      SdParams p = sd_params(h);
      p.mask = dmask;
      launch_sd_construct(p, nullptr);
    }
  }
  if ((rc = check_launch(h, "seed kernels"))) return rc;
  if ((e = hipDeviceSynchronize()) != hipSuccess) return fail(h, CE_ENODEV, "seed sync", e);
  return CE_OK;
}

extern "C" int ce_reset(ce_handle h, const uint8_t* mask, void* stream) {
  if (!h) return CE_EINVAL;
  begin_call(h);
  const uint8_t* dmask;
  int rc = stage_mask(h, mask, (hipStream_t)stream, &dmask);
  if (rc) return rc;
  if (is_grid(h->cfg)) {
    GridParams p = grid_params(h);
    p.mask = dmask;
    (counter_rng(h->cfg) ? launch_grid_reset_ctr : launch_grid_reset)((int)h->cfg.kind, p, h->d_gparams, stream);
  } else if (is_feat(h->cfg)) {
    GridParams p = grid_params(h);
    p.mask = dmask;
    launch_feat_reset((int)h->cfg.kind, p, h->d_gparams, stream);
  } else {
    SdParams p = sd_params(h);
    p.mask = dmask;
    launch_sd_reset(p, stream);
  }
  if (dmask && hipEventRecord(h->ev_mask, (hipStream_t)stream) == hipSuccess) h->mask_in_flight = true;
  if (hipEventRecord(h->ev_reset, (hipStream_t)stream) == hipSuccess) {
    h->reset_stream = stream;
    h->reset_gen += 1;
  }
  return check_launch(h, "reset kernel");
}

// wt: -1 = decide from the handle alone; 0 / 1 = the caller (ce_rollout) has decided for its whole call
static int step_range_impl(ce_handle h, const void* actions, const uint8_t* active, uint32_t env_begin, uint32_t env_count, void* stream,
                           int wt) {
  if (!h || !actions) return CE_EINVAL;
  if (env_count == 0 || (uint64_t)env_begin + env_count > h->cfg.num_envs) return fail(h, CE_EINVAL, "env range out of bounds");
  begin_call(h);
  if (hipError_t e = order_after_reset(h, stream); e != hipSuccess)
    return fail(h, CE_ENODEV, "step: the stream could not be ordered after the last ce_reset (a capture begun before the reset finished?)", e);
  if (is_grid(h->cfg)) {
    GridParams p = grid_params(h);
    p.actions = (const uint8_t*)actions;
    p.env_first = env_begin;
    p.env_count = env_count;
    if (wt >= 0) p.obs_wt = (uint32_t)wt;
    (counter_rng(h->cfg) ? launch_grid_step_ctr : launch_grid_step)((int)h->cfg.kind, p, h->d_gparams, stream);
  } else if (is_feat(h->cfg)) {
    GridParams p = grid_params(h);
    p.actions = (const uint8_t*)actions;
    p.env_first = env_begin;
    p.env_count = env_count;
    launch_feat_step((int)h->cfg.kind, p, h->d_gparams, stream);
  } else {
    SdParams p = sd_params(h);
    p.actions = (const float*)actions;
    p.active = active;
    p.env_first = env_begin;
    p.env_count = env_count;
    if (wt >= 0) p.obs_wt = (uint32_t)wt;
    launch_sd_step(p, stream);
  }
  if (h->timing_armed) h->timed_launches++;
  return check_launch(h, "step kernel");
}

extern "C" int ce_step_range(ce_handle h, const void* actions, const uint8_t* active, uint32_t env_begin, uint32_t env_count,
                             void* stream) {
  return step_range_impl(h, actions, active, env_begin, env_count, stream, -1);
}

extern "C" int ce_step_policy(ce_handle h, void* policy_out, uint32_t mode, uint32_t env_begin, uint32_t env_count, void* stream) {
  if (!h || !policy_out) return CE_EINVAL;
  if (!is_grid(h->cfg)) return fail(h, CE_EINVAL, "ce_step_policy belongs to the grid kinds");
  if (mode != CE_POLICY_BYTES_MOD && mode != CE_POLICY_ARGMAX_F32 && mode != CE_POLICY_AHEAD_NOISE) return fail(h, CE_EINVAL, "unknown CE_POLICY_* mode");
  if (env_count == 0 || (uint64_t)env_begin + env_count > h->cfg.num_envs) return fail(h, CE_EINVAL, "env range out of bounds");
  begin_call(h);
  if (hipError_t e = order_after_reset(h, stream); e != hipSuccess)
    return fail(h, CE_ENODEV, "step: the stream could not be ordered after the last ce_reset (a capture begun before the reset finished?)", e);
  GridParams p = grid_params(h);
  p.actions = (const uint8_t*)policy_out;
  p.env_first = env_begin;
  p.env_count = env_count;
  (counter_rng(h->cfg) ? launch_grid_step_policy_ctr : launch_grid_step_policy)((int)h->cfg.kind, (int)mode, p, h->d_gparams, stream);
  if (h->timing_armed) h->timed_launches++;
  return check_launch(h, "policy step kernel");
}

extern "C" int ce_step_policy_sliced(ce_handle h, void* policy_out, uint32_t mode, uint32_t num_slices, void* const* streams) {
  if (!h || !policy_out || num_slices == 0 || num_slices > h->cfg.num_envs) return CE_EINVAL;
  const uint32_t E = h->cfg.num_envs;
  for (uint32_t s = 0; s < num_slices; ++s) {
    const uint32_t b0 = (uint32_t)((uint64_t)E * s / num_slices), b1 = (uint32_t)((uint64_t)E * (s + 1) / num_slices);
    const int rc = ce_step_policy(h, policy_out, mode, b0, b1 - b0, streams ? streams[s] : nullptr);
    if (rc != CE_OK) return rc;
  }
  return CE_OK;
}

extern "C" int ce_rollout(ce_handle h, const void* actions, uint32_t num_steps, uint32_t num_slices, void* const* streams) {
  if (!h || !actions || num_steps == 0 || num_slices == 0 || num_slices > h->cfg.num_envs) return CE_EINVAL;
  const uint32_t E = h->cfg.num_envs;
  const size_t plane = (size_t)E * h->cfg.num_agents * (u8_actions(h->cfg) ? 1 : 4);
  // the call's working set is the handle AND the action planes it will read: the store policy of its launches follows that
  const int wt = obs_write_through(h, (unsigned long long)num_steps * plane) ? 1 : 0;
  for (uint32_t t = 0; t < num_steps; ++t) {
    const char* a_t = (const char*)actions + (size_t)t * plane;
    for (uint32_t s = 0; s < num_slices; ++s) {
      const uint32_t b0 = (uint32_t)((uint64_t)E * s / num_slices), b1 = (uint32_t)((uint64_t)E * (s + 1) / num_slices);
      const int rc = step_range_impl(h, a_t, nullptr, b0, b1 - b0, streams ? streams[s] : nullptr, wt);
      if (rc != CE_OK) return rc;
    }
  }
  return CE_OK;
}

extern "C" int ce_rollout_fused(ce_handle h, const void* actions, uint32_t num_steps, uint32_t steps_per_launch,
                                const ce_traj* traj, uint32_t num_slices, void* const* streams) {
  if (!h || !actions || num_steps == 0 || num_slices == 0 || num_slices > h->cfg.num_envs) return CE_EINVAL;
  if (traj && traj->num_planes == 0) return fail(h, CE_EINVAL, "ce_traj.num_planes must be at least 1");
  if (traj && traj->first_plane >= traj->num_planes) return fail(h, CE_EINVAL, "ce_traj.first_plane out of range");
  if (traj && (traj->num_envs != h->cfg.num_envs || traj->num_agents != h->cfg.num_agents))
    return fail(h, CE_EINVAL, "ce_traj.num_envs / num_agents do not match this handle: the ring was sized for another batch");
  begin_call(h);
  const ce_buffers& b = h->buf;
  const uint64_t E = h->cfg.num_envs, n = h->cfg.num_agents;
  RolloutArgs ra;
  std::memset(&ra, 0, sizeof(ra));
  // an array the caller did not supply is the handle's own buffer, overwritten by every step (plane stride 0)
#define CE_TRAJ(field, elems)                                   \
  do {                                                          \
    if (traj && traj->field) {                                  \
      ra.field = traj->field;                                   \
      ra.field##_plane = (elems);                               \
    } else {                                                    \
      ra.field = b.field;                                       \
      ra.field##_plane = 0;                                     \
    }                                                           \
  } while (0)
  CE_TRAJ(obs, E * b.obs_env_stride);
  CE_TRAJ(obs_f64, E * n * (2 * n + 7));
  CE_TRAJ(reward, E * n);
  CE_TRAJ(done, E);
  CE_TRAJ(done_agents, E * n);
  CE_TRAJ(info, E * n * 2);
  CE_TRAJ(features, E * n * b.num_features);
  CE_TRAJ(sd_info, E * 2);
#undef CE_TRAJ
  if (traj && traj->base_reward) {
    ra.base_reward = traj->base_reward;
    ra.agent_plane = E * n;
  } else {
    ra.base_reward = b.base_reward;
    ra.agent_plane = 0;
  }
  ra.action_plane = E * n;  // elements (bytes for the u8 kinds, floats for selfdrive)
  ra.num_planes = traj ? traj->num_planes : 1;
  ra.env_first = 0;
  ra.env_end = (uint32_t)E;
  const uint32_t per = steps_per_launch ? steps_per_launch : num_steps;
  uint32_t plane = traj ? traj->first_plane : 0;
  for (uint32_t s0 = 0; s0 < num_steps; s0 += per) {
    const uint32_t cnt = num_steps - s0 < per ? num_steps - s0 : per;
    ra.actions = (const char*)actions + (size_t)s0 * E * n * (u8_actions(h->cfg) ? 1 : 4);
    ra.num_steps = cnt;
    ra.plane0 = plane;
    for (uint32_t sl = 0; sl < num_slices; ++sl) {
      ra.env_first = (uint32_t)(E * sl / num_slices);
      ra.env_end = (uint32_t)(E * (sl + 1) / num_slices);
      void* stream = streams ? streams[sl] : nullptr;
      if (hipError_t e = order_after_reset(h, stream); e != hipSuccess)
        return fail(h, CE_ENODEV, "fused rollout: the stream could not be ordered after the last ce_reset", e);
      if (is_grid(h->cfg))
        (counter_rng(h->cfg) ? launch_grid_rollout_ctr : launch_grid_rollout)((int)h->cfg.kind, h->cfg.num_agents, h->d_tab != nullptr, h->d_gparams, ra, stream);
      else if (h->cfg.kind == CE_KIND_SELFDRIVE) launch_sd_rollout(sd_params(h), ra, stream);
      else launch_feat_rollout((int)h->cfg.kind, h->cfg.num_agents, h->d_gparams, ra, stream);
      if (h->timing_armed) h->timed_launches++;
    }
    plane = (uint32_t)(((uint64_t)plane + cnt) % ra.num_planes);
  }
  return check_launch(h, "fused rollout kernel");
}

extern "C" int ce_step(ce_handle h, const void* actions, const uint8_t* active, void* stream) {
  if (!h) return CE_EINVAL;
  return ce_step_range(h, actions, active, 0, h->cfg.num_envs, stream);
}

extern "C" int ce_step_host(ce_handle h, const void* host_actions, const uint8_t* host_active, void* stream) {
  if (!h || !host_actions) return CE_EINVAL;
  begin_call(h);
  const size_t cnt = (size_t)h->cfg.num_envs * h->cfg.num_agents;
  const size_t abytes = cnt * (u8_actions(h->cfg) ? 1 : 4);
  hipError_t e = hipMemcpyAsync(h->d_stage_actions, host_actions, abytes, hipMemcpyHostToDevice, (hipStream_t)stream);
  if (e == hipSuccess && host_active) e = hipMemcpyAsync(h->d_stage_active, host_active, cnt, hipMemcpyHostToDevice, (hipStream_t)stream);
  if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);  // host buffers are the caller's again
  if (e != hipSuccess) return fail(h, CE_ENODEV, "action upload", e);
  return ce_step(h, h->d_stage_actions, host_active ? h->d_stage_active : nullptr, stream);
}

extern "C" int ce_synth_actions(ce_handle h, uint64_t key, uint32_t t0, uint32_t T, void* out, void* stream) {
  if (!h || !out || T == 0) return CE_EINVAL;
  begin_call(h);
  const ce_config& c = h->cfg;
  if (is_feat(c)) {  // Discrete(7) / Discrete(8) action spaces of the feature envs
    launch_synth_actions_u8((uint8_t*)out, key, c.env_index_base, c.num_envs, c.num_agents, t0, T, c.kind == CE_KIND_HARVEST_FEATURES ? 7 : 8, stream);
  } else if (is_grid(c)) {
    const bool firing = c.flags & CE_FLAG_FIRING_ENABLED;
    const uint32_t na = c.kind == CE_KIND_CLEANUP ? (firing ? 9 : 8) : (firing ? 8 : 7);
    launch_synth_actions_u8((uint8_t*)out, key, c.env_index_base, c.num_envs, c.num_agents, t0, T, na, stream);
  } else {
    launch_synth_actions_f32((float*)out, key, c.env_index_base, c.num_envs, c.num_agents, t0, T, stream);
  }
  return check_launch(h, "synth actions kernel");
}

extern "C" uint32_t ce_synth_action_host(uint64_t key, uint64_t env_index, uint32_t t, uint32_t agent, uint32_t num_actions) {
  return synth_action(key, env_index, t, agent, num_actions);
}

extern "C" uint64_t ce_synth_hash_host(uint64_t key, uint64_t env_index, uint32_t t, uint32_t agent) {
  return synth_hash(key, env_index, t, agent);
}

extern "C" int ce_static_map(uint32_t kind, char* out, uint64_t out_bytes, uint32_t* rows, uint32_t* cols) {
  if (kind == CE_KIND_SELFDRIVE || kind > CE_KIND_CLEANUP_FEATURES) return CE_EINVAL;
  const bool cleanup = kind == CE_KIND_CLEANUP || kind == CE_KIND_CLEANUP_FEATURES;
  const uint32_t H = cleanup ? Geo<CE_KIND_CLEANUP>::H : Geo<CE_KIND_HARVEST>::H;
  const uint32_t W = cleanup ? Geo<CE_KIND_CLEANUP>::W : Geo<CE_KIND_HARVEST>::W;
  if (rows) *rows = H;
  if (cols) *cols = W;
  if (!out) return CE_OK;  // size query
  if (out_bytes < (uint64_t)H * W) return CE_EINVAL;
  const char** map = cleanup ? kCleanupMap : kHarvestMap;
  for (uint32_t r = 0; r < H; ++r) std::memcpy(out + (size_t)r * W, map[r], W);
  return CE_OK;
}

extern "C" int ce_get_buffers(ce_handle h, ce_buffers* out) {
  if (!h || !out) return CE_EINVAL;
  *out = h->buf;
  return CE_OK;
}

extern "C" int ce_synchronize(ce_handle h, void* stream) {
  if (!h) return CE_EINVAL;
  begin_call(h);
  hipError_t e = hipStreamSynchronize((hipStream_t)stream);
  if (e != hipSuccess) return fail(h, CE_ENODEV, "hipStreamSynchronize", e);
  return CE_OK;
}

struct FieldDesc {
  const char* name;
  void* base;
  size_t env_bytes;
};

static bool find_field(ce_engine* h, const char* name, FieldDesc* out) {
  const ce_buffers& b = h->buf;
  const size_t n = b.num_agents;
  const FieldDesc fields[] = {
      {"grid", b.grid, (size_t)h->grid_stride},
      {"agents", b.agents, n * 4},
      {"spawn_perm", b.spawn_perm, 20},
      {"waste_perm", b.waste_perm, 119},
      {"rng", b.rng, (size_t)b.rng_words * 4},
      {"timestep", b.timestep, 4},
      {"theta", b.theta, 8},
      {"sd_state", b.sd_state, CE_SD_STATE_DOUBLES(n) * 8},
      {"obs", b.obs, b.obs_env_stride},
      {"obs_f64", b.obs_f64, n * (2 * n + 7) * 8},
      {"base_reward", b.base_reward, n * 4},
      {"reward", b.reward, n * 8},
      {"done", b.done, 1},
      {"done_agents", b.done_agents, n},
      {"info", b.info, n * 2},
      {"features", b.features, n * b.num_features * 2},
      {"int_metrics", b.int_metrics, (size_t)b.num_int_metrics * 8},
      {"f64_metrics", b.f64_metrics, (size_t)b.num_f64_metrics * 8},
      {"final_int_metrics", b.final_int_metrics, (size_t)b.num_int_metrics * 8},
      {"final_f64_metrics", b.final_f64_metrics, (size_t)b.num_f64_metrics * 8},
      {"error_flags", b.error_flags, 4},
      {"beam_map", b.beam_map, (size_t)b.grid_h * b.grid_w},
      {"sd_info", b.sd_info, 16},
      {"actions_taken", b.actions_taken, n},
      {"debug", h->d_debug, 128},
  };
  for (const FieldDesc& f : fields)
    if (std::strcmp(f.name, name) == 0) {
      if (!f.base) return false;
      *out = f;
      return true;
    }
  return false;
}

extern "C" int ce_download(ce_handle h, const char* field, uint32_t env_begin, uint32_t env_count, void* dst, uint64_t dst_bytes) {
  if (!h || !field || !dst) return CE_EINVAL;
  begin_call(h);
  FieldDesc f;
  if (!find_field(h, field, &f)) return fail(h, CE_EINVAL, "unknown or absent field");
  if ((uint64_t)env_begin + env_count > h->cfg.num_envs || dst_bytes < (uint64_t)env_count * f.env_bytes) return fail(h, CE_EINVAL, "slice out of range");
  if (is_grid(h->cfg) && std::strcmp(field, "grid") == 0) {  // packed presence bits -> padded map image
    uint8_t* tmp = nullptr;
    hipError_t e2 = hipMalloc((void**)&tmp, (size_t)env_count * f.env_bytes);
    if (e2 != hipSuccess) return fail(h, CE_ENOMEM, "grid image buffer", e2);
    // step kernels may still be in flight on non-blocking streams the null stream does not wait for: drain the
    // device first, as every other field does, so the presence bits are read between steps and not inside one
    e2 = hipDeviceSynchronize();
    if (e2 == hipSuccess) {
      launch_grid_expand((int)h->cfg.kind, h->buf.grid, tmp, env_begin, env_count, h->d_tab, h->d_tab ? h->map_counts[0] : grid_napple(h->cfg),
                         h->d_tab ? h->map_counts[1] : grid_nwaste(h->cfg), nullptr);
      e2 = hipDeviceSynchronize();
    }
    if (e2 == hipSuccess) e2 = hipMemcpy(dst, tmp, (size_t)env_count * f.env_bytes, hipMemcpyDeviceToHost);
    (void)hipFree(tmp);
    if (e2 != hipSuccess) return fail(h, CE_ENODEV, "grid download", e2);
    return CE_OK;
  }
  hipError_t e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy(dst, (const char*)f.base + (size_t)env_begin * f.env_bytes, (size_t)env_count * f.env_bytes, hipMemcpyDeviceToHost);
  if (e != hipSuccess) return fail(h, CE_ENODEV, "download", e);
  return CE_OK;
}

// MapEnv.global_view for an env slice, on the device (JointEnv's `global_obs`): see k_grid_global_view
extern "C" int ce_global_view(ce_handle h, uint32_t env_begin, uint32_t env_count, uint8_t* out, void* stream) {
  if (!h || !out) return CE_EINVAL;
  if (!is_grid(h->cfg)) return fail(h, CE_EINVAL, "ce_global_view belongs to the grid kinds (cleanup_new / harvest_new)");
  if (env_count == 0 || (uint64_t)env_begin + env_count > h->cfg.num_envs) return fail(h, CE_EINVAL, "env range out of bounds");
  begin_call(h);
  if (hipError_t e = order_after_reset(h, stream); e != hipSuccess) return fail(h, CE_ENODEV, "global view: the stream could not be ordered after the last ce_reset", e);
  launch_grid_global_view((int)h->cfg.kind, h->buf.grid, h->buf.agents, h->buf.timestep, out, env_begin, env_count, h->cfg.num_agents, h->d_tab,
                          h->d_tab ? h->map_counts[0] : grid_napple(h->cfg), h->d_tab ? h->map_counts[1] : grid_nwaste(h->cfg), h->buf.grid_h,
                          h->buf.grid_w, stream);
  return check_launch(h, "global view kernel");
}

// ---- host boundary helpers (the RLlib vector hook's fast path, contracts_amd/vector_env.py) ----
extern "C" int ce_host_alloc(uint64_t bytes, void** out) {
  if (!out) return CE_EINVAL;
  *out = nullptr;
  void* p = nullptr;
  const hipError_t e = hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return e == hipErrorOutOfMemory ? CE_ENOMEM : CE_ENODEV;
  }
  *out = p;
  return CE_OK;
}

extern "C" int ce_host_free(void* p) {
  if (!p) return CE_OK;
  return hipHostFree(p) == hipSuccess ? CE_OK : CE_EINVAL;
}

extern "C" int ce_download_async(ce_handle h, const char* field, uint32_t env_begin, uint32_t env_count, void* dst, uint64_t dst_bytes,
                                 void* stream) {
  if (!h || !field || !dst) return CE_EINVAL;
  begin_call(h);
  FieldDesc f;
  if (!find_field(h, field, &f)) return fail(h, CE_EINVAL, "unknown or absent field");
  if (is_grid(h->cfg) && std::strcmp(field, "grid") == 0) return fail(h, CE_EINVAL, "ce_download_async: fetch \"grid\" with ce_download");
  if ((uint64_t)env_begin + env_count > h->cfg.num_envs || dst_bytes < (uint64_t)env_count * f.env_bytes) return fail(h, CE_EINVAL, "slice out of range");
  if (hipError_t eo = order_after_reset(h, stream); eo != hipSuccess)  // a copy on a stream other than the last reset's reads after it
    return fail(h, CE_ENODEV, "download: the stream could not be ordered after the last ce_reset", eo);
  const hipError_t e = hipMemcpyAsync(dst, (const char*)f.base + (size_t)env_begin * f.env_bytes, (size_t)env_count * f.env_bytes,
                                      hipMemcpyDeviceToHost, (hipStream_t)stream);
  if (e != hipSuccess) return fail(h, CE_ENODEV, "asynchronous download", e);
  return CE_OK;
}

extern "C" int ce_step_host_async(ce_handle h, const void* host_actions, uint32_t env_begin, uint32_t env_count, void* stream) {
  if (!h || !host_actions) return CE_EINVAL;
  if (!u8_actions(h->cfg)) return fail(h, CE_EINVAL, "ce_step_host_async takes uint8 action ids (grid / feature kinds)");
  if (env_count == 0 || (uint64_t)env_begin + env_count > h->cfg.num_envs) return fail(h, CE_EINVAL, "env range out of bounds");
  begin_call(h);
  const size_t n = h->cfg.num_agents;
  const hipError_t e = hipMemcpyAsync((char*)h->d_stage_actions + (size_t)env_begin * n, (const char*)host_actions + (size_t)env_begin * n,
                                      (size_t)env_count * n, hipMemcpyHostToDevice, (hipStream_t)stream);
  if (e != hipSuccess) return fail(h, CE_ENODEV, "action upload", e);
  return ce_step_range(h, h->d_stage_actions, nullptr, env_begin, env_count, stream);
}

// ---- host worker pool: the conversions below run every tick, and creating + joining 16 threads per call costs more than a
// quarter of the work at the headline batch (measured: the 4-part pipeline was slower than the unsplit one).  Threads are
// started on first use, sleep on a condition variable between jobs and live until the process exits.
namespace {
class HostPool {
 public:
  // fn(begin, end) over [0, total) in pieces of `grain`, on up to `threads` threads (the caller is one of them)
  void run(uint32_t threads, size_t total, size_t grain, const std::function<void(size_t, size_t)>& fn) {
    if (total == 0) return;
    if (grain == 0) grain = 1;
    const size_t pieces = (total + grain - 1) / grain;
    const uint32_t T = (uint32_t)std::min<size_t>(std::min<uint32_t>(threads, 256u), pieces);
    if (T <= 1) {
      fn(0, total);
      return;
    }
    std::lock_guard<std::mutex> one_job(job_m_);
    {
      std::unique_lock<std::mutex> lk(m_);
      while (started_ < T - 1) {
        try {
          std::thread(&HostPool::worker, this, started_).detach();
        } catch (const std::system_error&) {  // no more threads to be had: the job runs on the ones that exist (+ the caller)
          break;
        }
        ++started_;
      }
      fn_ = &fn, total_ = total, grain_ = grain;
      next_.store(0, std::memory_order_relaxed);
      helpers_ = left_ = std::min(T - 1, started_);
      ++gen_;
    }
    wake_.notify_all();
    drain();
    std::unique_lock<std::mutex> lk(m_);
    done_.wait(lk, [&] { return left_ == 0; });
  }

 private:
  void drain() {
    for (;;) {
      const size_t b = next_.fetch_add(grain_, std::memory_order_relaxed);
      if (b >= total_) return;
      (*fn_)(b, std::min(b + grain_, total_));
    }
  }
  void worker(uint32_t id) {
    uint64_t seen = 0;
    std::unique_lock<std::mutex> lk(m_);
    for (;;) {
      wake_.wait(lk, [&] { return gen_ != seen; });
      seen = gen_;
      if (id >= helpers_) continue;
      lk.unlock();
      drain();
      lk.lock();
      if (--left_ == 0) done_.notify_one();
    }
  }
  std::mutex job_m_, m_;
  std::condition_variable wake_, done_;
  const std::function<void(size_t, size_t)>* fn_ = nullptr;
  size_t total_ = 0, grain_ = 1;
  std::atomic<size_t> next_{0};
  uint64_t gen_ = 0;
  uint32_t started_ = 0, helpers_ = 0, left_ = 0;
};

HostPool* g_host_pool = nullptr;  // never destroyed: its threads may be asleep in it when the process exits
HostPool& host_pool() {
  static std::once_flag once;
  std::call_once(once, [] {
    g_host_pool = new HostPool;
    // a forked child inherits the object but none of its threads: give it a fresh pool on first use
    pthread_atfork(nullptr, nullptr, [] { g_host_pool = new HostPool; });
  });
  return *g_host_pool;
}

// one view (675 doubles) from a buffer that shares dst's alignment modulo 32 bytes: scalar head up to the first 32-byte boundary,
// 32-byte streaming stores, scalar tail
#ifdef CE_HOST_X86
__attribute__((target("avx"))) void stream_view_avx(double* dst, const double* t, int head) {
  int i = 0;
  for (; i < head; ++i) dst[i] = t[i];
  for (; i + 3 < 15 * 45; i += 4) _mm256_stream_pd(dst + i, _mm256_load_pd(t + i));
  for (; i < 15 * 45; ++i) dst[i] = t[i];
}
#endif

// uint8 pitched observation views -> float64 value / 255, dense [envs][n][15][15][3], on the pool.  A 256-entry table of the
// same double division numpy performs.
void convert_views(const uint8_t* pitched, double* out, uint32_t num_envs, uint32_t num_agents, uint32_t obs_env_stride,
                   uint32_t obs_agent_stride, uint32_t obs_row_stride, uint32_t threads) {
  static double lut[256];
  static std::once_flag once;
  std::call_once(once, [] {
    for (int v = 0; v < 256; ++v) lut[v] = (double)v / 255.0;
  });
  const size_t views = (size_t)num_envs * num_agents;
  // Streaming stores: the block is ~43 KB per env (0.7 GB at the headline batch), written once and read by the caller much
  // later — ordinary stores would first READ every destination line into the cache (read-for-ownership), doubling the memory
  // traffic of what is a bandwidth-bound loop.
  // (a view is assembled in a cache-resident buffer first: its rows start 8 bytes off a 16-byte boundary every other time, and
  // an ordinary store into a line that streaming stores are filling stalls the write-combining buffers — measured 4 x slower)
  // 32-byte streaming stores where the host has them (half the store instructions); CE_HOST_SSE_ONLY=1 keeps the 16-byte form for A/B
#ifdef CE_HOST_X86
  static const bool wide = __builtin_cpu_supports("avx") && !getenv("CE_HOST_SSE_ONLY");
#endif
  const std::function<void(size_t, size_t)> work = [=](size_t v0, size_t v1) {
    alignas(32) double tmp[15 * 45 + 4];
    for (size_t v = v0; v < v1; ++v) {
      const size_t e = v / num_agents, a = v % num_agents;
      const uint8_t* src = pitched + e * obs_env_stride + a * obs_agent_stride;
      double* dst = out + v * (size_t)(15 * 45);
      // the view lands in tmp at the offset that gives tmp + k and dst + k the same alignment modulo 32 bytes
      const int head = (int)((32u - ((uintptr_t)dst & 31u)) & 31u) / 8;  // doubles before dst's first 32-byte boundary (0..3)
      double* t = tmp + ((4 - head) & 3);
      for (int r = 0; r < 15; ++r) {
        const uint8_t* s = src + (size_t)r * obs_row_stride;
        double* d = t + r * 45;
        for (int k = 0; k < 45; ++k) d[k] = lut[s[k]];
      }
#ifdef CE_HOST_X86
      if (wide) stream_view_avx(dst, t, head);
      else {
        int i = 0;
        if (((uintptr_t)dst & 15u) != 0) dst[0] = t[0], i = 1;  // odd views start 8 bytes off a 16-byte boundary
        for (; i + 1 < 15 * 45; i += 2) _mm_stream_pd(dst + i, _mm_loadu_pd(t + i));
        if (i < 15 * 45) dst[i] = t[i];
      }
#else
      std::memcpy(dst, t, sizeof(double) * 15 * 45);  // hosts without the x86 streaming stores: ordinary stores
      (void)head;
#endif
    }
#ifdef CE_HOST_X86
    _mm_sfence();
#endif
  };
  // pieces a few times smaller than a thread's share: the threads that start late or share a core take fewer of them
  host_pool().run(threads, views, std::max<size_t>(64, views / ((size_t)threads * 4 + 1)), work);
}
}  // namespace

// uint8 pitched observation block -> the reference's float64 images (value / 255: cleanup_new.py:258, harvest_new.py:229)
extern "C" int ce_obs_u8_to_f64(const uint8_t* pitched, double* out, uint32_t num_envs, uint32_t num_agents, uint32_t obs_env_stride,
                                uint32_t obs_agent_stride, uint32_t obs_row_stride, uint32_t threads) {
  if (!pitched || !out || obs_row_stride < 45 || threads == 0) return CE_EINVAL;
  convert_views(pitched, out, num_envs, num_agents, obs_env_stride, obs_agent_stride, obs_row_stride, threads);
  return CE_OK;
}

// The observation leg of a dict-protocol tick in one call: the slice's pitched uint8 views travel to `staging` (page-locked) in
// `parts` copies issued back to back on `stream`, an event behind each; part i is converted into `out` (float64, dense) on the
// pool while parts i + 1 .. are still on the wire.  Returns when `out` is complete.
extern "C" int ce_download_obs_f64(ce_handle h, uint32_t env_begin, uint32_t env_count, void* staging, uint64_t staging_bytes, double* out,
                                   uint64_t out_bytes, uint32_t parts, uint32_t threads, void* stream) {
  if (!h || !staging || !out || threads == 0) return CE_EINVAL;
  begin_call(h);
  if (!is_grid(h->cfg)) return fail(h, CE_EINVAL, "ce_download_obs_f64: image observations are a grid-kind field");
  if (env_count == 0 || (uint64_t)env_begin + env_count > h->cfg.num_envs) return fail(h, CE_EINVAL, "slice out of range");
  const ce_buffers& b = h->buf;
  const uint32_t n = h->cfg.num_agents;
  if (staging_bytes < (uint64_t)env_count * b.obs_env_stride || out_bytes < (uint64_t)env_count * n * (15 * 45 * 8))
    return fail(h, CE_EINVAL, "destination too small");
  parts = std::max(1u, std::min(std::min(parts, 16u), env_count));
  while (h->obs_events.size() < parts) {
    hipEvent_t ev = nullptr;
    const hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    if (e != hipSuccess) return fail(h, CE_ENODEV, "event for the observation copy", e);
    h->obs_events.push_back(ev);
  }
  if (hipError_t eo = order_after_reset(h, stream); eo != hipSuccess)
    return fail(h, CE_ENODEV, "observation copy: the stream could not be ordered after the last ce_reset", eo);
  auto cut = [&](uint32_t i) { return (uint32_t)((uint64_t)env_count * i / parts); };
  // on a failure the copies already issued are waited for before returning: none is still writing `staging` when the caller
  // gets the error (and frees or reuses the block)
  auto settle = [&](uint32_t issued) {
    for (uint32_t j = 0; j < issued; ++j) (void)hipEventSynchronize(h->obs_events[j]);
    (void)hipStreamSynchronize((hipStream_t)stream);
  };
  for (uint32_t i = 0; i < parts; ++i) {
    const size_t off = (size_t)cut(i) * b.obs_env_stride, len = (size_t)(cut(i + 1) - cut(i)) * b.obs_env_stride;
    hipError_t e = hipMemcpyAsync((char*)staging + off, (const char*)b.obs + (size_t)env_begin * b.obs_env_stride + off, len,
                                  hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipEventRecord(h->obs_events[i], (hipStream_t)stream);
    if (e != hipSuccess) {
      settle(i);
      return fail(h, CE_ENODEV, "observation copy", e);
    }
  }
  for (uint32_t i = 0; i < parts; ++i) {
    const hipError_t e = hipEventSynchronize(h->obs_events[i]);
    if (e != hipSuccess) {
      settle(parts);
      return fail(h, CE_ENODEV, "observation copy", e);
    }
    convert_views((const uint8_t*)staging + (size_t)cut(i) * b.obs_env_stride, out + (size_t)cut(i) * n * (15 * 45), cut(i + 1) - cut(i), n,
                  b.obs_env_stride, b.obs_agent_stride, b.obs_row_stride, threads);
  }
  return CE_OK;
}

// int16 feature rows -> float64 (the reference's feature_obs entries are floats), `threads` worker threads
extern "C" int ce_i16_to_f64(const int16_t* src, double* out, uint64_t count, uint32_t threads) {
  if (!src || !out || threads == 0) return CE_EINVAL;
  const std::function<void(size_t, size_t)> work = [=](size_t i0, size_t i1) {
    for (size_t i = i0; i < i1; ++i) out[i] = (double)src[i];
  };
  host_pool().run(threads, (size_t)count, 65536, work);
  return CE_OK;
}

extern "C" int ce_download_many(ce_handle h, uint32_t env_begin, uint32_t env_count, const ce_field_req* reqs, uint32_t count) {
  if (!h || !reqs || count == 0) return CE_EINVAL;
  begin_call(h);
  if ((uint64_t)env_begin + env_count > h->cfg.num_envs) return fail(h, CE_EINVAL, "slice out of range");
  std::vector<FieldDesc> f(count);
  size_t total = 0;
  for (uint32_t i = 0; i < count; ++i) {
    if (!reqs[i].field || !reqs[i].dst) return CE_EINVAL;
    if (is_grid(h->cfg) && std::strcmp(reqs[i].field, "grid") == 0) return fail(h, CE_EINVAL, "ce_download_many: fetch \"grid\" with ce_download");
    if (!find_field(h, reqs[i].field, &f[i])) return fail(h, CE_EINVAL, "unknown or absent field");
    if (reqs[i].dst_bytes < (uint64_t)env_count * f[i].env_bytes) return fail(h, CE_EINVAL, "destination too small");
    total += ((size_t)env_count * f[i].env_bytes + 15) & ~(size_t)15;
  }
  hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) return fail(h, CE_ENODEV, "download_many sync", e);
  constexpr size_t kGatherMax = 1u << 20;  // above this a staging pass would only double the traffic
  if (total <= kGatherMax) {
    if (h->gather_bytes < total) {
      if (h->d_gather) (void)hipFree(h->d_gather);
      h->d_gather = nullptr;
      h->gather_bytes = 0;
      const size_t want = total < 65536 ? 65536 : total;
      if ((e = hipMalloc((void**)&h->d_gather, want)) != hipSuccess) return fail(h, CE_ENOMEM, "gather buffer", e);
      (void)hipMemsetAsync(h->d_gather, 0, want, nullptr);  // the 16-byte alignment gaps between fields are copied back too
      h->gather_bytes = want;
    }
    size_t off = 0;
    for (uint32_t i = 0; i < count && e == hipSuccess; ++i) {
      const size_t bytes = (size_t)env_count * f[i].env_bytes;
      e = hipMemcpyAsync(h->d_gather + off, (const char*)f[i].base + (size_t)env_begin * f[i].env_bytes, bytes, hipMemcpyDeviceToDevice, nullptr);
      off += (bytes + 15) & ~(size_t)15;
    }
    if (h->h_gather.size() < total) h->h_gather.resize(total);
    if (e == hipSuccess) e = hipMemcpy(h->h_gather.data(), h->d_gather, total, hipMemcpyDeviceToHost);  // null stream: after the gathers
    if (e != hipSuccess) return fail(h, CE_ENODEV, "download_many", e);
    off = 0;
    for (uint32_t i = 0; i < count; ++i) {
      const size_t bytes = (size_t)env_count * f[i].env_bytes;
      std::memcpy(reqs[i].dst, h->h_gather.data() + off, bytes);
      off += (bytes + 15) & ~(size_t)15;
    }
    return CE_OK;
  }
  for (uint32_t i = 0; i < count; ++i) {
    e = hipMemcpy(reqs[i].dst, (const char*)f[i].base + (size_t)env_begin * f[i].env_bytes, (size_t)env_count * f[i].env_bytes, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return fail(h, CE_ENODEV, "download_many", e);
  }
  return CE_OK;
}

extern "C" int ce_upload(ce_handle h, const char* field, uint32_t env_begin, uint32_t env_count, const void* src, uint64_t src_bytes) {
  if (!h || !field || !src) return CE_EINVAL;
  begin_call(h);
  FieldDesc f;
  if (!find_field(h, field, &f)) return fail(h, CE_EINVAL, "unknown or absent field");
  if ((uint64_t)env_begin + env_count > h->cfg.num_envs || src_bytes < (uint64_t)env_count * f.env_bytes) return fail(h, CE_EINVAL, "slice out of range");
  if (is_grid(h->cfg) && std::strcmp(field, "grid") == 0) {  // padded map image -> packed presence bits (validated)
    uint8_t* tmp = nullptr;
    hipError_t e2 = hipMalloc((void**)&tmp, (size_t)env_count * f.env_bytes);
    if (e2 != hipSuccess) return fail(h, CE_ENOMEM, "grid image buffer", e2);
    e2 = hipDeviceSynchronize();  // no step kernel may still be writing the bits this upload replaces
    if (e2 == hipSuccess) e2 = hipMemcpy(tmp, src, (size_t)env_count * f.env_bytes, hipMemcpyHostToDevice);
    if (e2 == hipSuccess) {
      launch_grid_pack((int)h->cfg.kind, tmp, h->buf.grid, h->buf.error_flags, env_begin, env_count, h->d_tab,
                       h->d_tab ? h->map_counts[0] : grid_napple(h->cfg), h->d_tab ? h->map_counts[1] : grid_nwaste(h->cfg), nullptr);
      e2 = hipDeviceSynchronize();
    }
    (void)hipFree(tmp);
    if (e2 != hipSuccess) return fail(h, CE_ENODEV, "grid upload", e2);
    return CE_OK;
  }
  hipError_t e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy((char*)f.base + (size_t)env_begin * f.env_bytes, src, (size_t)env_count * f.env_bytes, hipMemcpyHostToDevice);
  if (e != hipSuccess) return fail(h, CE_ENODEV, "upload", e);
  return CE_OK;
}

// ---- one-call state snapshot (checkpoint / resume for a C caller that does not know the field list) ----
namespace {
struct StateField {
  const char* name;
  void* base;
  size_t env_bytes;
  bool output;  // per-step output (only with CE_STATE_OUTPUTS)
};
// every persistent field of the handle's kind in its DEVICE layout (grid kinds: the 32-byte presence rows, not the image),
// then the per-step outputs
size_t state_fields(ce_engine* h, StateField* out) {
  const ce_buffers& b = h->buf;
  const size_t n = b.num_agents;
  const StateField all[] = {
      {"error_flags", b.error_flags, 4, false},
      {"grid", b.grid, (size_t)b.grid_env_stride, false},
      {"agents", b.agents, n * 4, false},
      {"spawn_perm", b.spawn_perm, 20, false},
      {"waste_perm", b.waste_perm, 119, false},
      {"rng", b.rng, (size_t)b.rng_words * 4, false},
      {"timestep", b.timestep, 4, false},
      {"theta", b.theta, 8, false},
      {"sd_state", b.sd_state, CE_SD_STATE_DOUBLES(n) * 8, false},
      {"int_metrics", b.int_metrics, (size_t)b.num_int_metrics * 8, false},
      {"f64_metrics", b.f64_metrics, (size_t)b.num_f64_metrics * 8, false},
      {"final_int_metrics", b.final_int_metrics, (size_t)b.num_int_metrics * 8, false},
      {"final_f64_metrics", b.final_f64_metrics, (size_t)b.num_f64_metrics * 8, false},
      {"done", b.done, 1, false},
      {"done_agents", b.done_agents, n, false},
      {"obs", b.obs, b.obs_env_stride, true},
      {"obs_f64", b.obs_f64, n * (2 * n + 7) * 8, true},
      {"base_reward", b.base_reward, n * 4, true},
      {"reward", b.reward, n * 8, true},
      {"info", b.info, n * 2, true},
      {"features", b.features, n * b.num_features * 2, true},
      {"sd_info", b.sd_info, 16, true},
      {"beam_map", b.beam_map, (size_t)b.grid_h * b.grid_w, true},
      {"actions_taken", b.actions_taken, n, true},
  };
  size_t k = 0;
  for (const StateField& f : all)
    if (f.base && f.env_bytes) out[k++] = f;
  return k;
}
constexpr size_t kMaxStateFields = 32;
uint64_t fnv1a(const void* p, size_t len, uint64_t hsh = 1469598103934665603ull) {
  const unsigned char* c = (const unsigned char*)p;
  for (size_t i = 0; i < len; ++i) hsh = (hsh ^ c[i]) * 1099511628211ull;
  return hsh;
}
// what a continuation must agree on to be bit-identical: the layout text (or the shipped one's name) and the table sizes
uint64_t layout_hash(const ce_engine* h) {
  uint64_t v = fnv1a(&h->cfg.kind, sizeof(h->cfg.kind));
  if (!h->map_text.empty()) {
    v = fnv1a(&h->map_h, sizeof(h->map_h), v);
    v = fnv1a(&h->map_w, sizeof(h->map_w), v);
    v = fnv1a(h->map_text.data(), h->map_text.size(), v);
  }
  return v;
}
void fill_header(ce_engine* h, uint32_t what, ce_state_header* hd, const StateField* f, size_t nf, ce_state_field* dir) {
  std::memset(hd, 0, sizeof(*hd));
  hd->magic = CE_STATE_MAGIC;
  hd->abi_version = CE_ABI_VERSION;
  hd->header_bytes = (uint32_t)(sizeof(ce_state_header) + nf * sizeof(ce_state_field));
  hd->kind = h->cfg.kind;
  hd->num_envs = h->cfg.num_envs;
  hd->num_agents = h->cfg.num_agents;
  hd->contract = h->cfg.contract;
  hd->flags = h->cfg.flags;
  hd->horizon = h->cfg.horizon;
  hd->what = what;
  hd->num_fields = 0;
  hd->env_index_base = h->cfg.env_index_base;
  hd->layout_hash = layout_hash(h);
  const double prm[9] = {h->cfg.contract_low, h->cfg.contract_high, h->cfg.null_prob, h->cfg.alpha, h->cfg.beta,
                         h->cfg.low_bound, h->cfg.high_bound, h->cfg.start_vel, h->cfg.start_vel_ambulance};
  std::memcpy(hd->params, prm, sizeof(prm));
  uint64_t off = hd->header_bytes;
  for (size_t i = 0; i < nf; ++i) {
    if (f[i].output && !(what & CE_STATE_OUTPUTS)) continue;
    ce_state_field& d = dir[hd->num_fields++];
    std::memset(&d, 0, sizeof(d));
    std::strncpy(d.name, f[i].name, sizeof(d.name) - 1);
    d.env_bytes = f[i].env_bytes;
    off = (off + 15ull) & ~15ull;
    d.offset = off;
    off += (uint64_t)f[i].env_bytes * h->cfg.num_envs;
  }
  hd->header_bytes = (uint32_t)(sizeof(ce_state_header) + hd->num_fields * sizeof(ce_state_field));
  hd->total_bytes = (off + 15ull) & ~15ull;
}
}  // namespace

extern "C" int ce_state_bytes(ce_handle h, uint32_t what, uint64_t* bytes) {
  if (!h || !bytes) return CE_EINVAL;
  if (what & ~CE_STATE_OUTPUTS) return fail(h, CE_EINVAL, "ce_state_bytes: unknown bits in `what`");
  StateField f[kMaxStateFields];
  ce_state_field dir[kMaxStateFields];
  ce_state_header hd;
  fill_header(h, what, &hd, f, state_fields(h, f), dir);
  *bytes = hd.total_bytes;
  return CE_OK;
}

extern "C" int ce_get_state(ce_handle h, uint32_t what, void* dst, uint64_t dst_bytes) {
  if (!h || !dst) return CE_EINVAL;
  if (what & ~CE_STATE_OUTPUTS) return fail(h, CE_EINVAL, "ce_get_state: unknown bits in `what`");
  begin_call(h);
  StateField f[kMaxStateFields];
  ce_state_field dir[kMaxStateFields];
  ce_state_header hd;
  const size_t nf = state_fields(h, f);
  fill_header(h, what, &hd, f, nf, dir);
  if (dst_bytes < hd.total_bytes) return fail(h, CE_EINVAL, "ce_get_state: buffer smaller than ce_state_bytes");
  hipError_t e = hipDeviceSynchronize();  // between steps, whatever stream they ran on
  if (e != hipSuccess) return fail(h, CE_ENODEV, "ce_get_state", e);
  char* out = (char*)dst;
  std::memset(out, 0, hd.header_bytes + 16);
  std::memcpy(out, &hd, sizeof(hd));
  std::memcpy(out + sizeof(hd), dir, hd.num_fields * sizeof(ce_state_field));
  for (uint32_t i = 0; i < hd.num_fields; ++i) {
    const StateField* sf = nullptr;
    for (size_t k = 0; k < nf; ++k)
      if (std::strcmp(f[k].name, dir[i].name) == 0) sf = &f[k];
    e = hipMemcpy(out + dir[i].offset, sf->base, (size_t)dir[i].env_bytes * hd.num_envs, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return fail(h, CE_ENODEV, "ce_get_state: copy", e);
  }
  return CE_OK;
}

extern "C" int ce_set_state(ce_handle h, const void* src, uint64_t src_bytes) {
  if (!h || !src) return CE_EINVAL;
  begin_call(h);
  if (src_bytes < sizeof(ce_state_header)) return fail(h, CE_EINVAL, "ce_set_state: not a state blob (too short)");
  ce_state_header hd;
  std::memcpy(&hd, src, sizeof(hd));
  if (hd.magic != CE_STATE_MAGIC) return fail(h, CE_EINVAL, "ce_set_state: not a state blob (magic)");
  if (hd.abi_version != CE_ABI_VERSION) return fail(h, CE_EINVAL, "ce_set_state: the blob was written by another ABI version (field layouts differ)");
  if (hd.total_bytes > src_bytes || hd.header_bytes > src_bytes || hd.num_fields > kMaxStateFields ||
      hd.header_bytes != sizeof(ce_state_header) + hd.num_fields * sizeof(ce_state_field))
    return fail(h, CE_EINVAL, "ce_set_state: truncated or inconsistent blob");
  if (hd.kind != h->cfg.kind || hd.num_envs != h->cfg.num_envs || hd.num_agents != h->cfg.num_agents)
    return fail(h, CE_EINVAL, "ce_set_state: the blob is for another kind / num_envs / num_agents");
  if (hd.layout_hash != layout_hash(h)) return fail(h, CE_EINVAL, "ce_set_state: the blob was taken on another map layout");
  // every parameter that enters a step: continuing under different ones would silently not reproduce the saved run
  const uint32_t step_flags = ~(uint32_t)CE_FLAG_BEAM_TRACE;
  const double prm[9] = {h->cfg.contract_low, h->cfg.contract_high, h->cfg.null_prob, h->cfg.alpha, h->cfg.beta,
                         h->cfg.low_bound, h->cfg.high_bound, h->cfg.start_vel, h->cfg.start_vel_ambulance};
  if (hd.contract != h->cfg.contract || hd.horizon != h->cfg.horizon || ((hd.flags ^ h->cfg.flags) & step_flags) != 0 ||
      hd.env_index_base != h->cfg.env_index_base || std::memcmp(hd.params, prm, sizeof(prm)) != 0)
    return fail(h, CE_EINVAL, "ce_set_state: blob and handle disagree on contract / flags / horizon / env_index_base / parameters");
  StateField f[kMaxStateFields];
  const size_t nf = state_fields(h, f);
  const char* in = (const char*)src;
  const ce_state_field* dir = (const ce_state_field*)(in + sizeof(ce_state_header));
  // validate the whole directory before the first byte of the handle is touched
  uint32_t persistent_seen = 0, persistent_need = 0;
  for (size_t k = 0; k < nf; ++k) persistent_need += f[k].output ? 0u : 1u;
  for (uint32_t i = 0; i < hd.num_fields; ++i) {
    ce_state_field d;
    std::memcpy(&d, &dir[i], sizeof(d));
    d.name[sizeof(d.name) - 1] = 0;
    const StateField* sf = nullptr;
    for (size_t k = 0; k < nf; ++k)
      if (std::strcmp(f[k].name, d.name) == 0) sf = &f[k];
    if (!sf || sf->env_bytes != d.env_bytes || d.offset > hd.total_bytes || (uint64_t)d.env_bytes * hd.num_envs > hd.total_bytes - d.offset)
      return fail(h, CE_EINVAL, "ce_set_state: a field of the blob does not exist on this handle or has another row size");
    persistent_seen += sf->output ? 0u : 1u;
  }
  if (persistent_seen != persistent_need) return fail(h, CE_EINVAL, "ce_set_state: the blob lacks a persistent field of this handle");
  hipError_t e = hipDeviceSynchronize();  // no step may still be writing what this replaces
  for (uint32_t i = 0; i < hd.num_fields && e == hipSuccess; ++i) {
    ce_state_field d;
    std::memcpy(&d, &dir[i], sizeof(d));
    d.name[sizeof(d.name) - 1] = 0;
    for (size_t k = 0; k < nf; ++k)
      if (std::strcmp(f[k].name, d.name) == 0)
        e = hipMemcpy(f[k].base, in + d.offset, (size_t)d.env_bytes * hd.num_envs, hipMemcpyHostToDevice);
  }
  if (e != hipSuccess) return fail(h, CE_ENODEV, "ce_set_state: copy", e);
  return CE_OK;
}

extern "C" int ce_timing_begin(ce_handle h, void* stream) {
  if (!h) return CE_EINVAL;
  begin_call(h);
  hipError_t e = hipEventRecord(h->ev_start, (hipStream_t)stream);
  if (e != hipSuccess) return fail(h, CE_ENODEV, "hipEventRecord", e);
  h->timing_armed = true;
  h->timed_launches = 0;
  return CE_OK;
}

extern "C" int ce_timing_end(ce_handle h, void* stream, double* mean_ms, uint32_t* launches) {
  if (!h || !h->timing_armed) return CE_EINVAL;
  begin_call(h);
  hipError_t e = hipEventRecord(h->ev_stop, (hipStream_t)stream);
  if (e == hipSuccess) e = hipEventSynchronize(h->ev_stop);
  float ms = 0.f;
  if (e == hipSuccess) e = hipEventElapsedTime(&ms, h->ev_start, h->ev_stop);
  h->timing_armed = false;
  if (e != hipSuccess) return fail(h, CE_ENODEV, "event timing", e);
  if (launches) *launches = h->timed_launches;
  if (mean_ms) *mean_ms = h->timed_launches ? (double)ms / h->timed_launches : 0.0;
  return CE_OK;
}

extern "C" int ce_selftest(int device, uint32_t* failed_mask) {
  if (hipSetDevice(device) != hipSuccess) return CE_ENODEV;
  uint32_t* d = nullptr;
  if (hipMalloc((void**)&d, 16) != hipSuccess) return CE_ENOMEM;
  (void)hipMemset(d, 0xff, 16);
  launch_selftest(d, nullptr);
  launch_selftest_ctr(d + 1, nullptr);  // bits 4, 5: Philox4x32-10 known answers, the counter-mode LDS fill
  uint32_t both[2] = {0xffffffffu, 0xffffffffu};
  hipError_t e = hipMemcpy(both, d, 8, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (e != hipSuccess) return CE_ENODEV;
  const uint32_t r = both[0] | both[1];
  if (failed_mask) *failed_mask = r;
  return r == 0 ? CE_OK : CE_EIO;
}

extern "C" const char* ce_last_error(ce_handle h) { return h ? h->err.c_str() : "null handle"; }
