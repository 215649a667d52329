// ce_grid_kernels_ctr.hip — the grid kernels of the counter-RNG mode (CE_FLAG_RNG_COUNTER, include/contracts_engine.h):
// the second translation unit of ce_grid_kernels.hip, see the note at the top of that file's namespace.
#define CE_RNG_COUNTER 1
#include "ce_grid_kernels.hip"
