/*
 * wurm_hip.h — C ABI of libwurm_hip.so, the MI355X (gfx950) implementation of the batched environment step of
 * oscarknagg/wurm: wurm.envs.SingleSnake / SimpleGridworld / MultiSnake .step() / .reset() / ._observe().
 *
 * The reference exposes no FFI for this path — its boundary is the Python class API of wurm/envs
 * (wurm/envs/__init__.py:1-3).  Each entry point below names the reference method it replaces; the Python
 * classes in wurm_amd/envs/ (same constructor signatures, attributes and return conventions as the
 * reference's) are thin callers of this ABI, and INTEGRATION.md shows the ctypes binding a maintainer of
 * the reference would add to call it from the wurm/envs modules directly.
 *
 * Conventions
 *  - All pointers are DEVICE pointers (HBM), caller-allocated and caller-owned.  Nothing is allocated, freed
 *    or synchronised here: each call enqueues kernels on `stream` (a hipStream_t passed as void*, NULL = the
 *    default stream) and returns.
 *  - State layout is the reference's: fp32 NCHW holding exact small integers.
 *      SingleSnake     envs (N,3,S,S) = [food, head, body]         (single_snake.py:87, config.py:7-9)
 *      SimpleGridworld envs (N,2,S,S) = [food, agent]              (simple_gridworld.py:74)
 *      MultiSnake      foods (N,1,S,S), heads/bodies (N*K,1,S,S), agent = env*K + i   (multi_snake.py:100-108)
 *  - done / info outputs are one byte per env holding 0 or 1 (torch.bool / torch.uint8 storage).
 *  - Return value: WURM_OK or a negative WURM_ERR_* code; the Python layer maps codes onto the reference's
 *    exception types (TypeError / RuntimeError / NotImplementedError / ValueError).
 *  - Randomness: the reference draws from torch's global RNG in a way no other implementation can restate
 *    (SURVEY.md §0 fact 6).  This ABI uses a counter-based generator, Philox4x32-10 keyed by `seed`, with
 *    counter (global env id = env_offset + i, `call`, purpose): results do not depend on how envs are
 *    sharded over GPUs.  `call` must be different for every step()/reset() call of one env object (the
 *    Python classes count calls).  Every `inject_*` argument is nullable; when non-NULL it supplies the
 *    random outcomes (used to replay outcomes recorded from the reference in parity tests).
 *  - Supported grid sizes: 3 <= S <= 64.  Env ids must be < 2^32.
 */
#ifndef WURM_HIP_H
#define WURM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WURM_OK 0
#define WURM_ERR_INVALID_ARG (-1) /* -> RuntimeError / ValueError   */
#define WURM_ERR_UNSUPPORTED (-2) /* -> NotImplementedError         */
#define WURM_ERR_HIP (-3)         /* a HIP launch failed            */
#define WURM_ERR_DTYPE (-4)       /* -> TypeError                   */

/* observation modes: single_snake.py:130-195, simple_gridworld.py:111-133, multi_snake.py:283-334 */
#define WURM_OBS_DEFAULT 0     /* RGB/255, (N,3,S,S)  (MultiSnake: 'full', per agent)         */
#define WURM_OBS_RAW 1         /* clone of the state                                          */
#define WURM_OBS_ONE_CHANNEL 2 /* (N,1,S,S)                                                   */
#define WURM_OBS_POSITIONS 3   /* (N,4) head y,x food y,x                                     */
#define WURM_OBS_PARTIAL 4     /* (N,3*(2n+1)^2) crop around the head                         */
#define WURM_OBS_NONE 5        /* no observation written                                      */

/* element type of the `actions` tensor (single_snake.py:198-200 accepts short/int/long) */
#define WURM_ACT_I64 0
#define WURM_ACT_I32 1

/* Library / build identification: "wurm_hip <version> gfx950". */
const char *wurm_version(void);

/* Number of fp32 elements one env's observation occupies (0 = invalid mode for that env family). */
int64_t wurm_single_obs_elems(int obs_mode, int obs_n, int size);
int64_t wurm_grid_obs_elems(int obs_mode, int obs_n, int size);

/* ------------------------------------------------------------------------------------------- SingleSnake */

/* SingleSnake.step (single_snake.py:197-304) + its _observe (:130-195) in one launch.
 *   envs            in/out (N,3,S,S) fp32
 *   actions         in/out (N) int64|int32 — reverse moves are rewritten in place (:221-222)
 *   reward          out (N) fp32; done, self_collision, edge_collision out (N) bytes
 *   obs             out, wurm_single_obs_elems() floats per env (ignored for WURM_OBS_NONE); the observation
 *                   of the post-step, pre-reset state (:304)
 *   inject_food     nullable (N) int32: flat cell (y*S+x) the respawned food goes to when env i eats, -1 = none */
int wurm_single_step(float *envs, void *actions, int actions_dtype, float *reward, uint8_t *done,
                     uint8_t *self_collision, uint8_t *edge_collision, float *obs, int obs_mode, int obs_n,
                     int64_t num_envs, int size, uint64_t seed, uint64_t call, int64_t env_offset,
                     const int32_t *inject_food, void *stream);

/* SingleSnake.reset (single_snake.py:322-342) + _create_envs (:344-387): envs with done[i] != 0 are rebuilt
 * (3-segment snake at a random seed cell/direction, one food), then every env is observed.
 *   inject_reset    nullable (N,4) int32: seed_y, seed_x, direction, food_cell
 * Returns WURM_ERR_UNSUPPORTED for size <= 8 (:346-347). */
int wurm_single_reset(float *envs, const uint8_t *done, float *obs, int obs_mode, int obs_n, int64_t num_envs,
                      int size, uint64_t seed, uint64_t call, int64_t env_offset, const int32_t *inject_reset,
                      void *stream);

/* SingleSnake._observe (single_snake.py:130-195) */
int wurm_single_observe(const float *envs, float *obs, int obs_mode, int obs_n, int64_t num_envs, int size,
                        void *stream);

/* T fused iterations of the caller loop of tests/test_single_snake_env.py:24-31 / experiments/main.py:212-227:
 *   for t: outputs[t] = step(actions[t]) with call = call0 + 2t;  reset(done[t]) with call = call0 + 2t + 1
 * in ONE launch with the env resident on chip.  actions (T,N) in/out; reward/done/... (T,N); obs (T,N,elems);
 * inject_food (T,N), inject_reset (T,N,4) nullable.  Bit-identical to T calls of the two entry points above. */
int wurm_single_rollout(float *envs, void *actions, int actions_dtype, float *reward, uint8_t *done,
                        uint8_t *self_collision, uint8_t *edge_collision, float *obs, int obs_mode, int obs_n,
                        int64_t num_envs, int size, int64_t num_steps, uint64_t seed, uint64_t call0,
                        int64_t env_offset, const int32_t *inject_food, const int32_t *inject_reset, void *stream);

/* wurm.utils.env_consistency / snake_consistency (wurm/utils.py:113-178) as a per-env error bitmask
 * (bit i = i-th check failed, WURM_CHK_*; 0 = consistent).  err out (N) uint32. */
#define WURM_CHK_FOOD_VALUE 1u
#define WURM_CHK_ONE_HEAD 2u
#define WURM_CHK_HAS_SNAKE 4u
#define WURM_CHK_HEAD_AT_END 8u
#define WURM_CHK_BODY_RANGE 16u
#define WURM_CHK_MIN_LENGTH 32u
#define WURM_CHK_HEAD_ON_FOOD 64u
#define WURM_CHK_ONE_FOOD 128u
int wurm_single_check(const float *envs, uint32_t *err, int64_t num_envs, int size, void *stream);

/* ------------------------------------------------------------------------------------------- SimpleGridworld */

/* SimpleGridworld.step (simple_gridworld.py:135-202) + _observe (:111-133); actions are read only. */
int wurm_grid_step(float *envs, const void *actions, int actions_dtype, float *reward, uint8_t *done,
                   uint8_t *edge_collision, float *obs, int obs_mode, int obs_n, int64_t num_envs, int size,
                   uint64_t seed, uint64_t call, int64_t env_offset, const int32_t *inject_food, void *stream);

/* SimpleGridworld.reset (simple_gridworld.py:225-268): agent at (start_y,start_x), one food.
 *   inject_reset nullable (N) int32 food cell.  WURM_ERR_UNSUPPORTED for size <= 4 or no start location (:249-260). */
int wurm_grid_reset(float *envs, const uint8_t *done, float *obs, int obs_mode, int obs_n, int64_t num_envs,
                    int size, int start_y, int start_x, uint64_t seed, uint64_t call, int64_t env_offset,
                    const int32_t *inject_reset, void *stream);

int wurm_grid_observe(const float *envs, float *obs, int obs_mode, int obs_n, int64_t num_envs, int size,
                      void *stream);

int wurm_grid_rollout(float *envs, const void *actions, int actions_dtype, float *reward, uint8_t *done,
                      uint8_t *edge_collision, float *obs, int obs_mode, int obs_n, int64_t num_envs, int size,
                      int64_t num_steps, int start_y, int start_x, uint64_t seed, uint64_t call0,
                      int64_t env_offset, const int32_t *inject_food, const int32_t *inject_reset, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* WURM_HIP_H */
