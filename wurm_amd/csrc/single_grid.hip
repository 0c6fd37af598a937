// single_grid.hip — the SimpleGridworld half of single_snake.hip's kernels (launch<false>: step / reset / observe / fused step /
// rollout / flagged kernels of the two-channel state at every cells-per-lane), compiled as a translation unit of its own so
// that the two halves build side by side.  Everything else of that file — the SingleSnake half, the entry points — is
// compiled with single_snake.hip itself.
#define WURM_TU_GRID
#include "single_snake.hip"
