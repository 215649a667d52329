/* A C caller of the engine's C-ABI (include/contracts_engine.h) that knows nothing about which library it is linked with:
 * tests link it once against oracle/_build/liboracle.so ("device = cpu", the CPU restatement under the engine's own entry-point
 * names) and once against contracts_amd/csrc/libcontracts_engine.so (the HIP engine) and compare what it prints.  It only uses
 * calls whose pointer arguments are host pointers on both sides (ce_step_host, ce_download) and reads the observation through
 * the strides ce_get_buffers reports.   usage: cabi_harness KIND NUM_ENVS NUM_AGENTS CONTRACT STEPS */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "contracts_engine.h"

static uint64_t fnv(const void* p, size_t len, uint64_t h) {
  const unsigned char* c = (const unsigned char*)p;
  for (size_t i = 0; i < len; i++) h = (h ^ c[i]) * 1099511628211ull;
  return h;
}
#define CK(call)                                                                          \
  do {                                                                                    \
    int rc_ = (call);                                                                     \
    if (rc_ != 0) {                                                                       \
      fprintf(stderr, "%s -> %d (%s)\n", #call, rc_, h ? ce_last_error(h) : "no handle"); \
      return 2;                                                                           \
    }                                                                                     \
  } while (0)

int main(int argc, char** argv) {
  if (argc < 6) return 1;
  const uint32_t kind = (uint32_t)atoi(argv[1]), E = (uint32_t)atoi(argv[2]), n = (uint32_t)atoi(argv[3]);
  const uint32_t contract = (uint32_t)atoi(argv[4]), T = (uint32_t)atoi(argv[5]);
  ce_handle h = NULL;
  ce_config cfg;
  memset(&cfg, 0, sizeof(cfg));
  cfg.abi_version = CE_ABI_VERSION;
  cfg.kind = kind;
  cfg.num_envs = E;
  cfg.num_agents = n;
  cfg.horizon = 23;
  cfg.contract = contract;
  cfg.flags = CE_FLAG_AUTO_RESET;
  cfg.contract_low = 0.0;
  cfg.contract_high = contract == CE_CONTRACT_CLEANUP ? (double)0.2f : 10.0;
  if (ce_abi_version() != CE_ABI_VERSION || ce_device_count() < 1) return 3;
  CK(ce_create(&cfg, &h));
  CK(ce_seed(h, NULL, 4242, NULL, CE_SEED_RESEED | CE_SEED_CONSTRUCT));
  CK(ce_reset(h, NULL, NULL));
  ce_buffers b;
  CK(ce_get_buffers(h, &b));
  uint8_t* acts = (uint8_t*)malloc((size_t)E * n);
  uint64_t lcg = 88172645463325252ull;
  const uint32_t na = kind == CE_KIND_CLEANUP ? 8u : 7u;
  for (uint32_t t = 0; t < T; t++) {
    for (size_t i = 0; i < (size_t)E * n; i++) {
      lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
      acts[i] = (uint8_t)((lcg >> 33) % na);
    }
    CK(ce_step_host(h, acts, NULL, NULL));
  }
  CK(ce_synchronize(h, NULL));
  static const struct { const char* name; size_t unit; } fields[] = {
      {"agents", 4}, {"rng", 0}, {"timestep", 4}, {"theta", 8}, {"base_reward", 4}, {"reward", 8}, {"info", 2},
      {"done", 0}, {"spawn_perm", 0}, {"waste_perm", 0}, {"features", 0}, {"int_metrics", 0}, {"f64_metrics", 0},
      {"final_int_metrics", 0}, {"final_f64_metrics", 0}, {"error_flags", 0}};
  for (size_t k = 0; k < sizeof(fields) / sizeof(fields[0]); k++) {
    size_t env_bytes;
    const char* f = fields[k].name;
    if (!strcmp(f, "rng")) env_bytes = 625 * 4; /* key[624] + pos: the pad words are free */
    else if (!strcmp(f, "done")) env_bytes = 1;
    else if (!strcmp(f, "spawn_perm")) env_bytes = 20;
    else if (!strcmp(f, "waste_perm")) env_bytes = 119;
    else if (!strcmp(f, "features")) env_bytes = (size_t)n * b.num_features * 2;
    else if (!strcmp(f, "int_metrics") || !strcmp(f, "final_int_metrics")) env_bytes = (size_t)b.num_int_metrics * 8;
    else if (!strcmp(f, "f64_metrics") || !strcmp(f, "final_f64_metrics")) env_bytes = (size_t)b.num_f64_metrics * 8;
    else if (!strcmp(f, "error_flags")) env_bytes = 4;
    else env_bytes = fields[k].unit * n;
    if (!strcmp(f, "timestep")) env_bytes = 4;
    if (!strcmp(f, "theta")) env_bytes = 8;
    if (!strcmp(f, "waste_perm") && kind != CE_KIND_CLEANUP) continue;
    const size_t row = !strcmp(f, "rng") ? (size_t)b.rng_words * 4 : env_bytes;
    uint8_t* buf = (uint8_t*)malloc(row * E);
    CK(ce_download(h, f, 0, E, buf, (uint64_t)row * E));
    uint64_t d = 1469598103934665603ull;
    for (uint32_t e = 0; e < E; e++) d = fnv(buf + (size_t)e * row, env_bytes, d);
    printf("%s %016llx\n", f, (unsigned long long)d);
    free(buf);
  }
  { /* the views, pixel by pixel through the reported strides (the engine's rows are pitched, the restatement's dense) */
    uint8_t* obs = (uint8_t*)malloc((size_t)b.obs_env_stride * E);
    CK(ce_download(h, "obs", 0, E, obs, (uint64_t)b.obs_env_stride * E));
    uint64_t d = 1469598103934665603ull;
    for (uint32_t e = 0; e < E; e++)
      for (uint32_t a = 0; a < n; a++)
        for (uint32_t i = 0; i < 15; i++)
          d = fnv(obs + (size_t)e * b.obs_env_stride + (size_t)a * b.obs_agent_stride + (size_t)i * b.obs_row_stride, 45, d);
    printf("obs %016llx\n", (unsigned long long)d);
    free(obs);
  }
  free(acts);
  CK(ce_destroy(h));
  return 0;
}
