/*
 * oracle/single_snake.c — scalar CPU restatement of the reference's SingleSnake and SimpleGridworld
 * transition, reset and observation functions.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle_common.h).  Each function cites the reference lines it follows;
 * paths are relative to the reference root (oscarknagg/wurm).
 *
 * State layout is the reference's: fp32 NCHW, SingleSnake (N,3,S,S) = [food, head, body]
 * (config.py:7-9), SimpleGridworld (N,2,S,S) = [food, agent] (simple_gridworld.py:22-24).  All
 * arithmetic is done per cell in fp32 exactly as the reference's tensor expressions do it, so the
 * restatement is valid for any grid content, not only for well-formed snakes.
 */
#include "oracle_common.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define EPS 1e-6f /* config.py:11 */

/* taps of ORIENTATION_FILTERS (wurm/_filters.py:7-28): filter i has +1 at centre+TAP[i] and -1 at the centre */
static const int TAP_Y[4] = {-1, 0, +1, 0};
static const int TAP_X[4] = {0, +1, 0, -1};

static inline float at(const float *ch, int S, int y, int x)
{
    return (y >= 0 && y < S && x >= 0 && x < S) ? ch[y * S + x] : 0.0f; /* conv2d zero padding */
}

/* wurm/utils.py:36-65 determine_orientations, one env */
static int orientation_of(const float *body, int S)
{
    int C = S * S;
    float L = body[0];
    for (int c = 1; c < C; ++c) if (body[c] > L) L = body[c];           /* utils.py:50 */
    float shift = L - 2.0f;                                              /* utils.py:51 */
    float *neck = (float *)malloc(sizeof(float) * (size_t)C);
    for (int c = 0; c < C; ++c) {
        float r = body[c] - shift;                                       /* utils.py:53 relu */
        if (r < 0.0f) r = 0.0f;
        r = r - 1.5f * (1.0f * (r > 0.0f ? 1.0f : 0.0f));                /* utils.py:54 */
        neck[c] = r * 2.0f;                                              /* utils.py:55 */
    }
    int best = 0;
    float best_v = 0.0f;
    for (int i = 0; i < 4; ++i) {                                        /* utils.py:59-63 */
        float m = -INFINITY;
        for (int y = 0; y < S; ++y)
            for (int x = 0; x < S; ++x) {
                float v = at(neck, S, y + TAP_Y[i], x + TAP_X[i]) - neck[y * S + x];
                if (v > m) m = v;
            }
        if (i == 0 || m > best_v) { best_v = m; best = i; }              /* argmax: first maximum */
    }
    free(neck);
    return best;
}

static int64_t load_action(const void *actions, int dtype, int64_t i)
{
    return dtype == ORACLE_ACT_I64 ? ((const int64_t *)actions)[i] : (int64_t)((const int32_t *)actions)[i];
}

static void store_action(void *actions, int dtype, int64_t i, int64_t v)
{
    if (dtype == ORACLE_ACT_I64) ((int64_t *)actions)[i] = v;
    else ((int32_t *)actions)[i] = (int32_t)v;
}

/* head channel shift: single_snake.py:225-233 / simple_gridworld.py:149-157.
 * delta[p] = head[p + TAP[a]] - head[p]; head[p] = round(head[p] + delta[p]). */
static void move_head(float *head, int S, int a)
{
    int C = S * S;
    float *old = (float *)malloc(sizeof(float) * (size_t)C);
    memcpy(old, head, sizeof(float) * (size_t)C);
    for (int y = 0; y < S; ++y)
        for (int x = 0; x < S; ++x) {
            float delta = at(old, S, y + TAP_Y[a], x + TAP_X[a]) - old[y * S + x];
            head[y * S + x] = rintf(old[y * S + x] + delta);
        }
    free(old);
}

/* _get_food_addition: single_snake.py:306-320, simple_gridworld.py:209-223 (+ wurm/utils.py:205-232).
 * One uniformly random cell among the interior cells whose channel sum is < EPS gets +1 food.
 * inject >= 0: use that cell (replaying the reference's recorded pick); inject == -1: draw rank k from `word`.
 * Returns the chosen cell or -1 when no cell is free. */
static int add_food(float *env, int n_channels, int S, int use_inject, int inject, uint32_t word)
{
    int C = S * S;
    if (use_inject) {
        if (inject >= 0 && inject < C) env[inject] += 1.0f;
        return inject;
    }
    int n_free = 0;
    for (int y = 1; y < S - 1; ++y)
        for (int x = 1; x < S - 1; ++x) {
            float s = 0.0f;
            for (int ch = 0; ch < n_channels; ++ch) s += env[ch * C + y * S + x];
            if (s < EPS) ++n_free;
        }
    if (n_free == 0) return -1;
    int k = (int)oracle_mulhi(word, (uint32_t)n_free);
    for (int y = 1; y < S - 1; ++y)
        for (int x = 1; x < S - 1; ++x) {
            float s = 0.0f;
            for (int ch = 0; ch < n_channels; ++ch) s += env[ch * C + y * S + x];
            if (s < EPS) {
                if (k == 0) { env[y * S + x] += 1.0f; return y * S + x; }
                --k;
            }
        }
    return -1;
}

/* ------------------------------------------------------------------ observations */

/* single_snake.py:104-128 _get_rgb, one cell -> 3 shorts.  Later writes win: body, head, food, edge. */
static void single_rgb_cell(const float *env, int S, int y, int x, int rgb[3])
{
    int C = S * S, c = y * S + x;
    rgb[0] = rgb[1] = rgb[2] = 255;                                     /* :106 */
    if (env[2 * C + c] > EPS) { rgb[0] = 0; rgb[1] = 127; rgb[2] = 0; } /* :111-112, colour :99 */
    if (env[1 * C + c] > EPS) { rgb[0] = 0; rgb[1] = 255; rgb[2] = 0; } /* :114-115 */
    if (env[0 * C + c] > EPS) { rgb[0] = 255; rgb[1] = 0; rgb[2] = 0; } /* :117-118 */
    if (y == 0 || x == 0 || y == S - 1 || x == S - 1) rgb[0] = rgb[1] = rgb[2] = 0; /* :120-123 */
}

/* simple_gridworld.py:88-109 _get_rgb: black background, head green, food red, black edge */
static void grid_rgb_cell(const float *env, int S, int y, int x, int rgb[3])
{
    int C = S * S, c = y * S + x;
    rgb[0] = rgb[1] = rgb[2] = 0;                                       /* :90 */
    if (env[1 * C + c] > EPS) { rgb[0] = 0; rgb[1] = 255; rgb[2] = 0; } /* :95-96 */
    if (env[0 * C + c] > EPS) { rgb[0] = 255; rgb[1] = 0; rgb[2] = 0; } /* :98-99 */
    if (y == 0 || x == 0 || y == S - 1 || x == S - 1) rgb[0] = rgb[1] = rgb[2] = 0; /* :101-104 */
}

static int argmax_f(const float *v, int n)
{
    int b = 0;
    for (int i = 1; i < n; ++i) if (v[i] > v[b]) b = i;
    return b;
}

int64_t oracle_single_obs_elems(int obs_mode, int obs_n, int size)
{
    int64_t C = (int64_t)size * size;
    switch (obs_mode) {
    case ORACLE_OBS_DEFAULT: return 3 * C;
    case ORACLE_OBS_RAW: return 3 * C;
    case ORACLE_OBS_ONE_CHANNEL: return C;
    case ORACLE_OBS_POSITIONS: return 4;
    case ORACLE_OBS_PARTIAL: return 3 * (int64_t)(2 * obs_n + 1) * (2 * obs_n + 1);
    default: return 0;
    }
}

int64_t oracle_grid_obs_elems(int obs_mode, int obs_n, int size)
{
    int64_t C = (int64_t)size * size;
    (void)obs_n;
    switch (obs_mode) {
    case ORACLE_OBS_DEFAULT: return 3 * C;
    case ORACLE_OBS_RAW: return 2 * C;
    case ORACLE_OBS_POSITIONS: return 4;
    default: return 0;
    }
}

/* single_snake.py:130-195 _observe */
int oracle_single_observe(const float *envs, float *obs, int obs_mode, int obs_n, int64_t N, int S)
{
    int C = S * S;
    int64_t per = oracle_single_obs_elems(obs_mode, obs_n, S);
    if (obs_mode == ORACLE_OBS_NONE) return ORACLE_OK;
    if (per == 0) return ORACLE_ERR_INVALID;
    for (int64_t i = 0; i < N; ++i) {
        const float *env = envs + i * 3 * C;
        float *o = obs + i * per;
        if (obs_mode == ORACLE_OBS_DEFAULT) {                            /* :131-138 */
            for (int y = 0; y < S; ++y)
                for (int x = 0; x < S; ++x) {
                    int rgb[3];
                    single_rgb_cell(env, S, y, x, rgb);
                    for (int ch = 0; ch < 3; ++ch) o[ch * C + y * S + x] = (float)rgb[ch] / 255.0f;
                }
        } else if (obs_mode == ORACLE_OBS_RAW) {                         /* :139-141 */
            memcpy(o, env, sizeof(float) * 3 * (size_t)C);
        } else if (obs_mode == ORACLE_OBS_ONE_CHANNEL) {                 /* :142-151 */
            for (int y = 0; y < S; ++y)
                for (int x = 0; x < S; ++x) {
                    int c = y * S + x;
                    float v = (env[2 * C + c] > EPS ? 1.0f : 0.0f) * 0.5f;
                    v += env[1 * C + c] * 0.5f;
                    v += env[0 * C + c] * 1.5f;
                    if (y == 0 || x == 0 || y == S - 1 || x == S - 1) v = -1.0f;
                    o[c] = v;
                }
        } else if (obs_mode == ORACLE_OBS_POSITIONS) {                   /* :152-165 */
            int h = argmax_f(env + C, C), f = argmax_f(env, C);
            o[0] = (float)(h / S); o[1] = (float)(h % S); o[2] = (float)(f / S); o[3] = (float)(f % S);
        } else {                                                         /* partial_n :166-193 */
            int w = 2 * obs_n + 1;
            int h = -1;
            for (int c = 0; c < C; ++c) if (rintf(env[C + c]) != 0.0f) { h = c; break; }
            if (h < 0) { /* the reference raises at :191; this build writes zeros (deviations) */
                memset(o, 0, sizeof(float) * (size_t)per);
                continue;
            }
            int hy = h / S, hx = h % S;
            for (int ch = 0; ch < 3; ++ch)
                for (int j = 0; j < w; ++j)
                    for (int k = 0; k < w; ++k) {
                        int y = hy - obs_n + j, x = hx - obs_n + k;
                        float v = 0.0f;                                  /* F.pad zeros :179 */
                        if (y >= 0 && y < S && x >= 0 && x < S) {
                            int rgb[3];
                            single_rgb_cell(env, S, y, x, rgb);
                            v = (float)rgb[ch] / 255.0f;                 /* :174 */
                        }
                        o[(ch * w + j) * w + k] = v;
                    }
        }
    }
    return ORACLE_OK;
}

/* simple_gridworld.py:111-133 _observe */
int oracle_grid_observe(const float *envs, float *obs, int obs_mode, int obs_n, int64_t N, int S)
{
    int C = S * S;
    int64_t per = oracle_grid_obs_elems(obs_mode, obs_n, S);
    if (obs_mode == ORACLE_OBS_NONE) return ORACLE_OK;
    if (per == 0) return ORACLE_ERR_INVALID;
    for (int64_t i = 0; i < N; ++i) {
        const float *env = envs + i * 2 * C;
        float *o = obs + i * per;
        if (obs_mode == ORACLE_OBS_DEFAULT) {
            for (int y = 0; y < S; ++y)
                for (int x = 0; x < S; ++x) {
                    int rgb[3];
                    grid_rgb_cell(env, S, y, x, rgb);
                    for (int ch = 0; ch < 3; ++ch) o[ch * C + y * S + x] = (float)rgb[ch] / 255.0f;
                }
        } else if (obs_mode == ORACLE_OBS_RAW) {
            memcpy(o, env, sizeof(float) * 2 * (size_t)C);
        } else { /* positions :122-131 (the reference only handles num_envs == 1; generalised per env) */
            int h = argmax_f(env + C, C), f = argmax_f(env, C);
            o[0] = (float)(h / S); o[1] = (float)(h % S); o[2] = (float)(f / S); o[3] = (float)(f % S);
        }
    }
    return ORACLE_OK;
}

/* ------------------------------------------------------------------ SingleSnake.step */

/* single_snake.py:197-304, one call over N envs.  `actions` is sanitised in place (:222). */
int oracle_single_step(float *envs, void *actions, int act_dtype, float *reward, uint8_t *done,
                       uint8_t *self_collision, uint8_t *edge_collision, float *obs, int obs_mode, int obs_n,
                       int64_t N, int S, uint64_t seed, uint64_t call, int64_t env_offset,
                       const int32_t *inject_food)
{
    int C = S * S;
    if (S < 3 || N < 0) return ORACLE_ERR_INVALID;
    for (int64_t i = 0; i < N; ++i) {
        float *env = envs + i * 3 * C;
        float *food = env, *head = env + C, *body = env + 2 * C;

        float L = body[0];                                               /* :210 snake_sizes */
        for (int c = 1; c < C; ++c) if (body[c] > L) L = body[c];

        int o = orientation_of(body, S);                                 /* :212 */
        int64_t a = load_action(actions, act_dtype, i);
        a = (a + ((int64_t)o == a ? 2 : 0)) % 4;                         /* :221-222 add_, fmod_ */
        store_action(actions, act_dtype, i, a);
        int ai = (int)(((a % 4) + 4) % 4);                               /* negative actions: UB in the reference (scatter_ :229) */

        move_head(head, S, ai);                                          /* :225-233 */

        float overlap = 0.0f;                                            /* :242 */
        for (int c = 0; c < C; ++c) overlap += head[c] * food[c];

        if ((uint8_t)overlap == 0)                                       /* :246-249 decay where nothing eaten */
            for (int c = 0; c < C; ++c) { float b = body[c] - 1.0f; body[c] = b > 0.0f ? b : 0.0f; }

        float hb = 0.0f;                                                 /* :252 */
        for (int c = 0; c < C; ++c) hb += head[c] * body[c];
        uint8_t self_c = hb > EPS;

        for (int c = 0; c < C; ++c) body[c] += head[c] * (L + overlap);  /* :258-262 */

        float r = 0.0f;                                                  /* :270-272 */
        for (int c = 0; c < C; ++c) {
            float removal = head[c] * food[c] * -1.0f;
            r -= removal;
            food[c] += removal;
        }
        reward[i] = r;

        if (r > 0.0f) {                                                  /* :277-282 */
            uint32_t w[4];
            oracle_rng_words(seed, call, (uint64_t)(env_offset + i), RNG_FOOD, 0, w);
            add_food(env, 3, S, inject_food != NULL, inject_food ? inject_food[i] : -1, w[0]);
        }

        float interior = 0.0f;                                           /* :290-293 valid conv with NO_CHANGE_FILTER */
        for (int y = 1; y < S - 1; ++y)
            for (int x = 1; x < S - 1; ++x) interior += head[y * S + x];
        uint8_t edge_c = interior < EPS;

        for (int c = 0; c < 3 * C; ++c) env[c] = rintf(env[c]);          /* :300 */

        self_collision[i] = self_c;
        edge_collision[i] = edge_c;
        done[i] = self_c | edge_c;                                       /* :254,294 */
    }
    if (obs != NULL) return oracle_single_observe(envs, obs, obs_mode, obs_n, N, S); /* :304 */
    return ORACLE_OK;
}

/* ------------------------------------------------------------------ SingleSnake.reset */

/* single_snake.py:344-387 _create_envs for one env slab.
 * inject = {seed_y, seed_x, direction, food_cell} or NULL. */
static void single_create_env(float *env, int S, uint64_t seed, uint64_t call, uint64_t env_id, const int32_t *inject)
{
    int C = S * S;
    memset(env, 0, sizeof(float) * 3 * (size_t)C);                       /* :352 */
    uint32_t w[4];
    oracle_rng_words(seed, call, env_id, RNG_RESET, 0, w);
    int sy, sx, d;
    if (inject) { sy = inject[0]; sx = inject[1]; d = inject[2]; }
    else {
        /* randint(1+L0, S-(1+L0)) with L0 = 3: values 4..S-5 (:358-359); randint(4) (:366) */
        sy = 4 + (int)oracle_mulhi(w[0], (uint32_t)(S - 8));
        sx = 4 + (int)oracle_mulhi(w[1], (uint32_t)(S - 8));
        d = (int)(w[2] >> 30);
    }
    /* conv2d(seed, LENGTH_3_SNAKES[d], padding=1) (:372-376, _filters.py:38-59): head value 3 at seed + TAP[d],
     * 2 at the seed, tail 1 at seed - TAP[d] */
    float *body = env + 2 * C, *head = env + C;
    int hy = sy + TAP_Y[d], hx = sx + TAP_X[d], ty = sy - TAP_Y[d], tx = sx - TAP_X[d];
    body[hy * S + hx] = 3.0f;
    body[sy * S + sx] = 2.0f;
    body[ty * S + tx] = 1.0f;
    head[hy * S + hx] = 1.0f;                                            /* :379-381 head = (body == max) */
    add_food(env, 3, S, inject != NULL, inject ? inject[3] : -1, w[3]);  /* :384-385 */
}

/* single_snake.py:322-342 reset(done): rebuild flagged envs, then observe everything */
int oracle_single_reset(float *envs, const uint8_t *done, float *obs, int obs_mode, int obs_n, int64_t N, int S,
                        uint64_t seed, uint64_t call, int64_t env_offset, const int32_t *inject_reset)
{
    int C = S * S;
    int any = 0;
    for (int64_t i = 0; i < N; ++i) any |= done[i] != 0;
    if (any && S <= 8) return ORACLE_ERR_UNSUPPORTED;                    /* :346-347 NotImplementedError */
    for (int64_t i = 0; i < N; ++i)
        if (done[i])
            single_create_env(envs + i * 3 * C, S, seed, call, (uint64_t)(env_offset + i),
                              inject_reset ? inject_reset + 4 * i : NULL);
    if (obs != NULL) return oracle_single_observe(envs, obs, obs_mode, obs_n, N, S); /* :342 */
    return ORACLE_OK;
}

/* ------------------------------------------------------------------ SimpleGridworld */

/* simple_gridworld.py:135-202 */
int oracle_grid_step(float *envs, const void *actions, int act_dtype, float *reward, uint8_t *done,
                     uint8_t *edge_collision, float *obs, int obs_mode, int obs_n, int64_t N, int S, uint64_t seed,
                     uint64_t call, int64_t env_offset, const int32_t *inject_food)
{
    int C = S * S;
    if (S < 3 || N < 0) return ORACLE_ERR_INVALID;
    for (int64_t i = 0; i < N; ++i) {
        float *env = envs + i * 2 * C;
        float *food = env, *head = env + C;
        int64_t a = load_action(actions, act_dtype, i);                  /* not sanitised, not written back */
        int ai = (int)(((a % 4) + 4) % 4);
        move_head(head, S, ai);                                          /* :149-157 */
        float r = 0.0f;                                                  /* :168-170 */
        for (int c = 0; c < C; ++c) {
            float removal = head[c] * food[c] * -1.0f;
            r -= removal;
            food[c] += removal;
        }
        reward[i] = r;
        if (r > 0.0f) {                                                  /* :175-182 */
            uint32_t w[4];
            oracle_rng_words(seed, call, (uint64_t)(env_offset + i), RNG_FOOD, 0, w);
            add_food(env, 2, S, inject_food != NULL, inject_food ? inject_food[i] : -1, w[0]);
        }
        float interior = 0.0f;                                           /* :188-191 */
        for (int y = 1; y < S - 1; ++y)
            for (int x = 1; x < S - 1; ++x) interior += head[y * S + x];
        uint8_t edge_c = interior < EPS;
        for (int c = 0; c < 2 * C; ++c) env[c] = rintf(env[c]);          /* :198 */
        edge_collision[i] = edge_c;
        done[i] = edge_c;                                                /* :192 */
    }
    if (obs != NULL) return oracle_grid_observe(envs, obs, obs_mode, obs_n, N, S);
    return ORACLE_OK;
}

/* simple_gridworld.py:225-268 reset/_create_envs: agent at start_location, one food.
 * inject_reset: (N) food cell or NULL. */
int oracle_grid_reset(float *envs, const uint8_t *done, float *obs, int obs_mode, int obs_n, int64_t N, int S,
                      int start_y, int start_x, uint64_t seed, uint64_t call, int64_t env_offset,
                      const int32_t *inject_reset)
{
    int C = S * S;
    int any = 0;
    for (int64_t i = 0; i < N; ++i) any |= done[i] != 0;
    if (any && S <= 4) return ORACLE_ERR_UNSUPPORTED;                    /* :249-250 */
    if (any && (start_y < 0 || start_x < 0 || start_y >= S || start_x >= S)) return ORACLE_ERR_UNSUPPORTED; /* :254-260 */
    for (int64_t i = 0; i < N; ++i) {
        if (!done[i]) continue;
        float *env = envs + i * 2 * C;
        memset(env, 0, sizeof(float) * 2 * (size_t)C);                   /* :252 */
        env[C + start_y * S + start_x] = 1.0f;                           /* :262 */
        uint32_t w[4];
        oracle_rng_words(seed, call, (uint64_t)(env_offset + i), RNG_RESET, 0, w);
        add_food(env, 2, S, inject_reset != NULL, inject_reset ? inject_reset[i] : -1, w[3]); /* :265-266 */
    }
    if (obs != NULL) return oracle_grid_observe(envs, obs, obs_mode, obs_n, N, S);
    return ORACLE_OK;
}

/* ------------------------------------------------------------------ rollouts (loop of tests/test_single_snake_env.py:24-31)
 * for t: step(actions[t]) with call = call0 + 2t -> outputs[t] (pre-reset observation), then reset(done[t])
 * with call = call0 + 2t + 1 (its observation is discarded, experiments/main.py:227). */
int oracle_single_rollout(float *envs, void *actions, int act_dtype, float *reward, uint8_t *done,
                          uint8_t *self_collision, uint8_t *edge_collision, float *obs, int obs_mode, int obs_n,
                          int64_t N, int S, int64_t T, uint64_t seed, uint64_t call0, int64_t env_offset,
                          const int32_t *inject_food, const int32_t *inject_reset)
{
    int64_t per = oracle_single_obs_elems(obs_mode, obs_n, S);
    size_t asz = act_dtype == ORACLE_ACT_I64 ? 8 : 4;
    for (int64_t t = 0; t < T; ++t) {
        int rc = oracle_single_step(envs, (char *)actions + (size_t)(t * N) * asz, act_dtype, reward + t * N,
                                    done + t * N, self_collision + t * N, edge_collision + t * N,
                                    obs ? obs + t * N * per : NULL, obs_mode, obs_n, N, S, seed,
                                    call0 + 2 * (uint64_t)t, env_offset, inject_food ? inject_food + t * N : NULL);
        if (rc) return rc;
        rc = oracle_single_reset(envs, done + t * N, NULL, ORACLE_OBS_NONE, 0, N, S, seed,
                                 call0 + 2 * (uint64_t)t + 1, env_offset,
                                 inject_reset ? inject_reset + t * N * 4 : NULL);
        if (rc) return rc;
    }
    return ORACLE_OK;
}

int oracle_grid_rollout(float *envs, const void *actions, int act_dtype, float *reward, uint8_t *done,
                        uint8_t *edge_collision, float *obs, int obs_mode, int obs_n, int64_t N, int S, int64_t T,
                        int start_y, int start_x, uint64_t seed, uint64_t call0, int64_t env_offset,
                        const int32_t *inject_food, const int32_t *inject_reset)
{
    int64_t per = oracle_grid_obs_elems(obs_mode, obs_n, S);
    size_t asz = act_dtype == ORACLE_ACT_I64 ? 8 : 4;
    for (int64_t t = 0; t < T; ++t) {
        int rc = oracle_grid_step(envs, (const char *)actions + (size_t)(t * N) * asz, act_dtype, reward + t * N,
                                  done + t * N, edge_collision + t * N, obs ? obs + t * N * per : NULL, obs_mode,
                                  obs_n, N, S, seed, call0 + 2 * (uint64_t)t, env_offset,
                                  inject_food ? inject_food + t * N : NULL);
        if (rc) return rc;
        rc = oracle_grid_reset(envs, done + t * N, NULL, ORACLE_OBS_NONE, 0, N, S, start_y, start_x, seed,
                               call0 + 2 * (uint64_t)t + 1, env_offset, inject_reset ? inject_reset + t * N : NULL);
        if (rc) return rc;
    }
    return ORACLE_OK;
}

/* ------------------------------------------------------------------ invariant checkers
 * wurm/utils.py:113-178 snake_consistency / env_consistency as a per-env error bitmask
 * (bit i = i-th check of the reference failed; 0 = consistent). */
enum {
    CHK_FOOD_VALUE = 1,      /* utils.py:119-125 food not in {0,1}                     */
    CHK_ONE_HEAD = 2,        /* utils.py:127-131 head channel does not sum to 1        */
    CHK_HAS_SNAKE = 4,       /* utils.py:134-136 body channel sums to 0                */
    CHK_HEAD_AT_END = 8,     /* utils.py:139-143 body value under the head != max body */
    CHK_BODY_RANGE = 16,     /* utils.py:147-153 body values are not {1..L}            */
    CHK_MIN_LENGTH = 32,     /* utils.py:156-157 body total < 6                        */
    CHK_HEAD_ON_FOOD = 64,   /* utils.py:160-164 head overlaps food                    */
    CHK_ONE_FOOD = 128       /* utils.py:176-178 food channel does not sum to 1        */
};

int oracle_single_check(const float *envs, uint32_t *err, int64_t N, int S)
{
    int C = S * S;
    for (int64_t i = 0; i < N; ++i) {
        const float *food = envs + i * 3 * C, *head = food + C, *body = food + 2 * C;
        uint32_t e = 0;
        float hs = 0, bs = 0, bm = body[0], hb = 0, hf = 0, fs = 0;
        for (int c = 0; c < C; ++c) {
            if (!(food[c] == 0.0f || food[c] == 1.0f)) e |= CHK_FOOD_VALUE;
            hs += head[c]; bs += body[c]; hb += head[c] * body[c]; hf += head[c] * food[c]; fs += food[c];
            if (body[c] > bm) bm = body[c];
        }
        if (hs != 1.0f) e |= CHK_ONE_HEAD;
        if (!(bs > 0.0f)) e |= CHK_HAS_SNAKE;
        if (bm != hb) e |= CHK_HEAD_AT_END;
        if ((sqrtf(8.0f * bs + 1.0f) - 1.0f) / 2.0f != bm) e |= CHK_BODY_RANGE;
        if (!(bs >= 6.0f)) e |= CHK_MIN_LENGTH;
        if (hf != 0.0f) e |= CHK_HEAD_ON_FOOD;
        if (fs != 1.0f) e |= CHK_ONE_FOOD;
        err[i] = e;
    }
    return ORACLE_OK;
}
