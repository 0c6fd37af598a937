/*
 * oracle/multi_snake.c — scalar CPU restatement of the reference's MultiSnake step / reset / observation
 * (wurm/envs/multi_snake.py in oscarknagg/wurm).
 *
 * TEST INFRASTRUCTURE ONLY (see oracle_common.h).  State layout is the reference's (multi_snake.py:100-108):
 *   foods (N,1,S,S) fp32, heads / bodies (N*K,1,S,S) fp32 with agent = env*K + i, dones (N*K) bytes,
 *   orientations (N*K) int64, agent_colours (N*K,3) int16.
 * Arithmetic is per cell in fp32 exactly as the reference's tensor expressions.  The reference's batch-global
 * gates (`torch.any(boosted_agents)` :503, `boost_cost_agents.sum() > 0` :580) have no per-env effect
 *  and are evaluated per env here.
 *
 * Random outcomes: RNG mode draws from the Philox streams of oracle_common.h; injected mode replays the
 * Bernoulli outcomes / picks recorded from the reference (see oracle_multi_inject).
 */
#include "oracle_common.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define EPS 1e-6f /* config.py:11 */

static const int TAP_Y[4] = {-1, 0, +1, 0}; /* wurm/_filters.py:7-28, see single_snake.c */
static const int TAP_X[4] = {0, +1, 0, -1};

typedef struct {
    int boost;               /* self.boost                                   multi_snake.py:123 */
    int food_on_death;       /* self.food_on_death_prob > 0                  :565,662           */
    float death_threshold;   /* (float)(1 - food_on_death_prob)              :424               */
    float boost_cost_prob;   /*                                              :579               */
    int food_mode;           /* 0 'only_one', 1 'random_rate'                :369,380           */
    float food_rate;         /*                                              :403               */
    int max_food;            /* num_snakes * 8                               :127               */
    float reward_on_death;   /*                                              :684               */
    int respawn_any;         /* respawn_mode == 'any'                        :805               */
    int colour_random;       /* colour_mode == 'random'                      :800               */
} oracle_multi_cfg;

typedef struct {
    const uint8_t *death_a;  /* (N,S,S)  outcome of rand_like > 1-p in the boost phase   :424 via :574 */
    const uint8_t *cost;     /* (N*K)    outcome of rand < boost_cost_prob               :579          */
    const uint8_t *death_b;  /* (N,S,S)  outcome of rand_like > 1-p in the regular phase :424 via :671 */
    const uint8_t *rate;     /* (N,S,S)  outcome of rand < food_rate                     :401-403      */
    const int32_t *food_cell;/* (N)      'only_one' respawn cell, -1 = none              :374-379      */
} oracle_multi_inject;

typedef struct {
    const int32_t *create;      /* (N,K,2) seed cell, direction of every snake of a rebuilt env   :996-1019 */
    const int32_t *create_food; /* (N)     food cell of a rebuilt env                              :1016     */
    const int16_t *colours;     /* (N*K,3) colour given to a snake that is still dead after reset  :800-803  */
    const int32_t *respawn;     /* (N,2)   respawn_mode 'any': seed cell (-1 = no room), direction :805-829  */
} oracle_multi_reset_inject;

/* ------------------------------------------------------------------------------------------------ helpers */

/* _move_heads (multi_snake.py:341-353): heads += conv2d(heads, ORIENTATION_FILTERS)[direction] */
static void move_head(float *head, int S, int dir)
{
    int C = S * S;
    float *old = (float *)malloc(sizeof(float) * (size_t)C);
    memcpy(old, head, sizeof(float) * (size_t)C);
    for (int y = 0; y < S; ++y)
        for (int x = 0; x < S; ++x) {
            int yy = y + TAP_Y[dir], xx = x + TAP_X[dir];
            float nb = (yy >= 0 && yy < S && xx >= 0 && xx < S) ? old[yy * S + xx] : 0.0f;
            head[y * S + x] = old[y * S + x] + (nb - old[y * S + x]);
        }
    free(old);
}

static int is_edge(int S, int y, int x) { return y == 0 || x == 0 || y == S - 1 || x == S - 1; }

/* One phase of MultiSnake.step for the snakes in `who` (boost phase :509-563, regular phase :613-660). */
static void run_phase(float *food, float *heads, float *bodies, uint8_t *done, int K, int S, const uint8_t *who,
                      const int *dir, float *L, float *rewards, float *food_cons, uint8_t *snake_col,
                      uint8_t *edge_col)
{
    int C = S * S;
    for (int s = 0; s < K; ++s)
        if (who[s]) move_head(heads + s * C, S, dir[s]);                    /* :509 / :613 */

    /* food overlap of ALL snakes (:514 / :618), food removal clamped to 1 per cell (:517-518 / :622) */
    float *ov = (float *)calloc((size_t)K, sizeof(float));
    for (int c = 0; c < C; ++c) {
        float sum = 0.0f;
        for (int s = 0; s < K; ++s) {
            float o = (heads[s * C + c] * food[c]) > EPS ? 1.0f : 0.0f;
            ov[s] += o;
            sum += o;
        }
        if (sum > 1.0f) sum = 1.0f;
        if (sum < 0.0f) sum = 0.0f;
        food[c] -= sum;
    }
    /* decay movers that did not eat (:523-526 / :627-628); rewards (:527-529 / :629-631) */
    for (int s = 0; s < K; ++s) {
        if (!who[s]) continue;
        if (ov[s] < EPS)
            for (int c = 0; c < C; ++c) { float b = bodies[s * C + c] - 1.0f; bodies[s * C + c] = b > 0.0f ? b : 0.0f; }
        float eaten = ov[s] > EPS ? 1.0f : 0.0f;
        rewards[s] += eaten;
        food_cons[s] += eaten;
    }
    /* collisions against all bodies + the other snakes' heads (:534-547 / :636-644) */
    for (int s = 0; s < K; ++s) {
        if (!who[s]) continue;
        int coll = 0;
        for (int c = 0; c < C; ++c) {
            float pathing = 0.0f;
            for (int o = 0; o < K; ++o) if (o != s) pathing += heads[o * C + c];
            float allb = 0.0f;
            for (int o = 0; o < K; ++o) allb += bodies[o * C + c];
            pathing += allb;
            if (heads[s * C + c] * pathing > EPS) coll = 1;
        }
        done[s] |= (uint8_t)coll;
        snake_col[s] |= (uint8_t)coll;
    }
    /* new head segment (:552-555 / :649-652) */
    for (int s = 0; s < K; ++s) {
        if (!who[s]) continue;
        float growth = ov[s] > EPS ? 1.0f : 0.0f;
        for (int c = 0; c < C; ++c) bodies[s * C + c] += heads[s * C + c] * (L[s] + growth);
        L[s] += growth;
    }
    /* edge collisions (:560-562 / :657-659) */
    for (int s = 0; s < K; ++s) {
        if (!who[s]) continue;
        int e = 0;
        for (int y = 0; y < S; ++y)
            for (int x = 0; x < S; ++x)
                if (is_edge(S, y, x) && heads[s * C + y * S + x] * 1.0f > EPS) e = 1;
        done[s] |= (uint8_t)e;
        edge_col[s] |= (uint8_t)e;
    }
    free(ov);
}

/* _food_from_death (:416-428) applied as at :565-576 / :662-673.
 * outcome(c) = injected byte, or u01 > threshold with the per-cell Philox uniform. */
static void food_from_death(float *food, const float *bodies, const uint8_t *done, int K, int S,
                            const oracle_multi_cfg *cfg, const uint8_t *inj, uint64_t seed, uint64_t call,
                            uint64_t env_id, uint32_t purpose)
{
    int C = S * S;
    for (int y = 0; y < S; ++y)
        for (int x = 0; x < S; ++x) {
            int c = y * S + x;
            float dead = 0.0f, living = 0.0f;
            for (int s = 0; s < K; ++s) {
                if (done[s]) dead += bodies[s * C + c];
                else living += bodies[s * C + c];
            }
            if (y == 1 || x == 0 || y == S - 1 || x == S - 1) dead = 0.0f; /* :418-421 (row 1, sic) */
            float d01 = rintf(dead) > 0.0f ? 1.0f : 0.0f;                   /* :422 */
            int hit;
            if (inj) hit = d01 > 0.0f && inj[c];
            else hit = d01 * oracle_cell_u01(seed, call, env_id, purpose, (uint32_t)c) > cfg->death_threshold; /* :424 */
            if (hit && !(living > EPS)) food[c] += 1.0f;                    /* :426, :575 / :672 */
        }
}

static void round_state(float *food, float *heads, float *bodies, int K, int C)
{
    for (int c = 0; c < C; ++c) {                                            /* :599-603 / :688-692 */
        float f = rintf(food[c]);
        food[c] = f < 0.0f ? 0.0f : (f > 1.0f ? 1.0f : f);
    }
    for (int i = 0; i < K * C; ++i) { heads[i] = rintf(heads[i]); bodies[i] = rintf(bodies[i]); } /* :604-605 / :693-694 */
}

static void delete_done(float *heads, float *bodies, const uint8_t *done, int K, int C)
{
    for (int s = 0; s < K; ++s)
        if (done[s]) {                                                       /* :595-596 / :676-677 */
            memset(heads + s * C, 0, sizeof(float) * (size_t)C);
            memset(bodies + s * C, 0, sizeof(float) * (size_t)C);
        }
}

/* free = no food, head or body on the cell, and interior (:439-445) */
static int cell_free(const float *food, const float *heads, const float *bodies, int K, int S, int y, int x)
{
    int C = S * S, c = y * S + x;
    if (y < 1 || x < 1 || y > S - 2 || x > S - 2) return 0;
    float sum = food[c];
    for (int s = 0; s < K; ++s) sum += heads[s * C + c];
    for (int s = 0; s < K; ++s) sum += bodies[s * C + c];
    return sum < EPS;
}

/* _get_food_addition (:430-457): +1 on the rank-th free cell (row-major) or on the injected cell */
static void add_one_food(float *food, const float *heads, const float *bodies, int K, int S, int use_inject,
                         int inject_cell, uint32_t word)
{
    if (use_inject) {
        if (inject_cell >= 0 && inject_cell < S * S) food[inject_cell] += 1.0f;
        return;
    }
    int n_free = 0;
    for (int y = 0; y < S; ++y)
        for (int x = 0; x < S; ++x) n_free += cell_free(food, heads, bodies, K, S, y, x);
    if (n_free == 0) return;
    int k = (int)oracle_mulhi(word, (uint32_t)n_free);
    for (int y = 0; y < S; ++y)
        for (int x = 0; x < S; ++x)
            if (cell_free(food, heads, bodies, K, S, y, x)) {
                if (k == 0) { food[y * S + x] += 1.0f; return; }
                --k;
            }
}

/* ------------------------------------------------------------------------------------------------ step */

/* MultiSnake.step for one env (:462-731 without the observation and dict packaging). */
static void multi_step_env(float *food, float *heads, float *bodies, uint8_t *done, int64_t *orient,
                           const int64_t *act /* K */, uint8_t *boost_this_step, float *rewards,
                           uint8_t *snake_col, uint8_t *edge_col, float *food_cons, float *sizes, int K, int S,
                           const oracle_multi_cfg *cfg, uint64_t seed, uint64_t call, uint64_t env_id,
                           const oracle_multi_inject *inj, int64_t env_local)
{
    int C = S * S;
    float *L = sizes;
    uint8_t done0[64], boosted[64], all[64];
    int dir[64];
    int any_boosted = 0;
    for (int s = 0; s < K; ++s) {
        snake_col[s] = 0; edge_col[s] = 0; food_cons[s] = 0.0f; rewards[s] = 0.0f;   /* :475-478 */
        float m = bodies[s * C];
        for (int c = 1; c < C; ++c) if (bodies[s * C + c] > m) m = bodies[s * C + c]; /* :489 */
        L[s] = m;
        done0[s] = done[s];                                                          /* :490 */
        int64_t a = act[s];
        int64_t d = a % 4;                                                           /* :483 fmod */
        int b = a > 3;                                                               /* :484 */
        if (orient[s] == d) d = (d + 2) % 4;                                         /* :493, :336-339 */
        orient[s] = (d + 2) % 4;                                                     /* :494, :355-357 */
        dir[s] = (int)(((d % 4) + 4) % 4);
        boosted[s] = (uint8_t)(b && L[s] >= 4.0f);                                   /* :497-498 */
        boost_this_step[s] = boosted[s];                                             /* :499 */
        any_boosted |= boosted[s];
        all[s] = 1;
    }

    if (cfg->boost && any_boosted) {                                                 /* :503 */
        run_phase(food, heads, bodies, done, K, S, boosted, dir, L, rewards, food_cons, snake_col, edge_col);
        if (cfg->food_on_death)                                                      /* :565-576 */
            food_from_death(food, bodies, done, K, S, cfg, inj ? inj->death_a + env_local * C : NULL, seed, call,
                            env_id, RNG_DEATH_FOOD_A);
        /* boost cost (:579-592) */
        for (int s = 0; s < K; ++s) {
            if (!boosted[s]) continue;
            int pay;
            if (inj) pay = inj->cost[env_local * K + s] != 0;
            else {
                uint32_t w[4];
                oracle_rng_words(seed, call, env_id, RNG_BOOST_COST, (uint32_t)s, w);
                pay = oracle_u01(w[0]) < cfg->boost_cost_prob;
            }
            boosted[s] = (uint8_t)(pay ? 2 : 1); /* 2 = pays the cost */
        }
        for (int c = 0; c < C; ++c) {                                                /* :583-586 */
            float tails = 0.0f;
            for (int s = 0; s < K; ++s)
                if (boosted[s] == 2 && bodies[s * C + c] == 1.0f) tails += 1.0f;
            if (tails > EPS) food[c] += 1.0f;
        }
        for (int s = 0; s < K; ++s) {
            if (boosted[s] != 2) continue;
            for (int c = 0; c < C; ++c) { float b = bodies[s * C + c] - 1.0f; bodies[s * C + c] = b > 0.0f ? b : 0.0f; } /* :588-589 */
            rewards[s] -= 1.0f;                                                      /* :590 */
            L[s] -= 1.0f;                                                            /* :591 */
        }
        delete_done(heads, bodies, done, K, C);                                      /* :595-596 */
        round_state(food, heads, bodies, K, C);                                      /* :599-605 */
    }

    run_phase(food, heads, bodies, done, K, S, all, dir, L, rewards, food_cons, snake_col, edge_col); /* :613-660 */
    if (cfg->food_on_death)                                                          /* :662-673 */
        food_from_death(food, bodies, done, K, S, cfg, inj ? inj->death_b + env_local * C : NULL, seed, call, env_id,
                        RNG_DEATH_FOOD_B);
    delete_done(heads, bodies, done, K, C);                                          /* :676-677 */

    /* _add_food (:368-410) */
    float fsum = 0.0f;
    for (int c = 0; c < C; ++c) fsum += food[c];
    if (cfg->food_mode == 0) {
        if (fsum < EPS) {                                                            /* :371-379 */
            uint32_t w[4];
            oracle_rng_words(seed, call, env_id, RNG_FOOD, 0, w);
            add_one_food(food, heads, bodies, K, S, inj != NULL, inj ? inj->food_cell[env_local] : -1, w[0]);
        }
    } else {
        if (fsum < (float)cfg->max_food) {                                           /* :382 */
            /* :393-408 every free interior cell spawns food independently with probability food_rate.  Injected: the
             * reference's own rand() outcomes per cell.  RNG mode (this build's own specification): the NUMBER of cells is
             * Binomial(n free cells, food_rate), drawn by inversion from one uniform, and the cells are a uniformly random
             * subset of that size (the j-th pick: the mulhi(word, n - j)-th remaining free cell in row-major order) — the
             * same distribution as n independent draws; cell by cell when P(no food) is too small for the recurrence. */
            int nfree = 0;
            uint8_t *fr = (uint8_t *)malloc((size_t)C);
            for (int y = 0; y < S; ++y)
                for (int x = 0; x < S; ++x) {
                    fr[y * S + x] = (uint8_t)cell_free(food, heads, bodies, K, S, y, x);
                    nfree += fr[y * S + x];
                }
            const float pw = inj ? 0.0f : oracle_pow_n(1.0f - cfg->food_rate, nfree);
            if (inj || !(cfg->food_rate > 0.0f) || pw < ORACLE_BINOMIAL_MIN_P0) {
                for (int c = 0; c < C; ++c) {
                    if (!fr[c]) continue;
                    int hit = inj ? inj->rate[env_local * C + c] != 0
                                  : oracle_cell_u01(seed, call, env_id, RNG_RATE_FOOD, (uint32_t)c) < cfg->food_rate; /* :401-403 */
                    if (hit) food[c] += 1.0f;                                        /* :408 */
                }
            } else {
                uint32_t w[4];
                oracle_rng_words(seed, call, env_id, RNG_RATE_FOOD, 0, w);
                const int k = oracle_binomial_inverse(nfree, cfg->food_rate, pw, oracle_u01(w[0]));
                for (int j = 0; j < k; ++j) {
                    const int wi = j + 1;
                    if ((wi & 3) == 0) oracle_rng_words(seed, call, env_id, RNG_RATE_FOOD, (uint32_t)(wi >> 2), w);
                    int rank = (int)(((uint64_t)w[wi & 3] * (uint64_t)(nfree - j)) >> 32);
                    for (int c = 0; c < C; ++c) {
                        if (!fr[c]) continue;
                        if (rank-- == 0) { food[c] += 1.0f; fr[c] = 0; break; }
                    }
                }
            }
            free(fr);
        }
    }

    for (int s = 0; s < K; ++s)                                                      /* :683-685 */
        if (done[s] && !done0[s]) rewards[s] += 1.0f * cfg->reward_on_death;
    round_state(food, heads, bodies, K, C);                                          /* :688-694 */
}

/* ------------------------------------------------------------------------------------------------ observations */

/* _observe_agent (:268-281) for one cell: layers painted in dict order food, own body, own head, other bodies,
 * other heads on white, then the black edge (_make_generic_rgb :175-192).  self_colour/2 = (0,96,0),
 * other_colour/2 = (0,0,96) under torch-1.1 integer division (:276,278). */
static void full_rgb_cell(const float *food, const float *heads, const float *bodies, int K, int S, int agent, int y,
                          int x, int rgb[3])
{
    int C = S * S, c = y * S + x;
    rgb[0] = rgb[1] = rgb[2] = 255;
    if (food[c] > EPS) { rgb[0] = 255; rgb[1] = 0; rgb[2] = 0; }
    if (bodies[agent * C + c] > EPS) { rgb[0] = 0; rgb[1] = 96; rgb[2] = 0; }
    if (heads[agent * C + c] > EPS) { rgb[0] = 0; rgb[1] = 192; rgb[2] = 0; }
    float ob = 0.0f, oh = 0.0f;
    for (int s = 0; s < K; ++s)
        if (s != agent) { ob += bodies[s * C + c]; oh += heads[s * C + c]; }
    if (ob > EPS) { rgb[0] = 0; rgb[1] = 0; rgb[2] = 96; }
    if (oh > EPS) { rgb[0] = 0; rgb[1] = 0; rgb[2] = 192; }
    if (is_edge(S, y, x)) rgb[0] = rgb[1] = rgb[2] = 0;
}

/* _get_env_images (:194-227) for one cell of one env */
static void env_image_cell(const float *food, const float *heads, const float *bodies, const uint8_t *boost,
                           const int16_t *colours, int K, int S, int y, int x, int rgb[3])
{
    int C = S * S, c = y * S + x;
    float acc[3] = {0.0f, 0.0f, 0.0f};
    for (int s = 0; s < K; ++s) {
        float inten = (bodies[s * C + c] > EPS ? 1.0f : 0.0f) * 1.0f / 3.0f +
                      (heads[s * C + c] > EPS ? 1.0f : 0.0f) * 1.0f / 3.0f;          /* :197 */
        inten *= (1.0f + 0.5f * (boost[s] ? 1.0f : 0.0f));                           /* :198 */
        for (int ch = 0; ch < 3; ++ch) acc[ch] += inten * (float)colours[s * 3 + ch]; /* :201-205 */
    }
    for (int ch = 0; ch < 3; ++ch) rgb[ch] = (int)(int16_t)acc[ch];                  /* :206 .short() truncates */
    if (food[c] > EPS) rgb[0] += 255;                                                /* :208-209 */
    if (rgb[0] == 0 && rgb[1] == 0 && rgb[2] == 0) rgb[0] = rgb[1] = rgb[2] = 255;  /* :214-219 black -> white */
    if (is_edge(S, y, x)) rgb[0] = rgb[1] = rgb[2] = 0;                              /* :225 */
}

int64_t oracle_multi_obs_elems(int obs_mode, int obs_n, int size)
{
    if (obs_mode == ORACLE_OBS_DEFAULT) return 3 * (int64_t)size * size;
    if (obs_mode == ORACLE_OBS_PARTIAL) return 3 * (int64_t)(2 * obs_n + 1) * (2 * obs_n + 1);
    return 0;
}

/* _observe (:283-334).  obs is (K, N, elems): obs[i] is agent_i's (N,3,h,w) tensor. */
int oracle_multi_observe(const float *foods, const float *heads, const float *bodies, const uint8_t *dones,
                         const uint8_t *boost_this_step, const int16_t *colours, float *obs, int obs_mode, int obs_n,
                         int64_t N, int K, int S)
{
    int C = S * S;
    int64_t per = oracle_multi_obs_elems(obs_mode, obs_n, S);
    if (obs_mode == ORACLE_OBS_NONE) return ORACLE_OK;
    if (per == 0) return ORACLE_ERR_INVALID;
    for (int64_t e = 0; e < N; ++e) {
        const float *food = foods + e * C, *hd = heads + e * K * C, *bd = bodies + e * K * C;
        for (int a = 0; a < K; ++a) {
            float *o = obs + ((int64_t)a * N + e) * per;
            if (obs_mode == ORACLE_OBS_DEFAULT) {                                     /* 'full' :287-288 */
                for (int y = 0; y < S; ++y)
                    for (int x = 0; x < S; ++x) {
                        int rgb[3];
                        full_rgb_cell(food, hd, bd, K, S, a, y, x, rgb);
                        for (int ch = 0; ch < 3; ++ch) o[ch * C + y * S + x] = (float)rgb[ch] / 255.0f; /* :281 */
                    }
            } else {                                                                  /* partial_n :289-332 */
                int w = 2 * obs_n + 1;
                memset(o, 0, sizeof(float) * (size_t)per);                            /* :320 dead snakes: zeros */
                if (dones[e * K + a]) continue;                                       /* :323 */
                int h = -1;
                for (int c = 0; c < C; ++c) if (rintf(hd[a * C + c]) != 0.0f) { h = c; break; }
                if (h < 0) continue;
                int hy = h / S, hx = h % S;
                for (int j = 0; j < w; ++j)
                    for (int k = 0; k < w; ++k) {
                        int y = hy - obs_n + j, x = hx - obs_n + k;
                        if (y < 0 || y >= S || x < 0 || x >= S) continue;             /* F.pad zeros :302 */
                        int rgb[3];
                        env_image_cell(food, hd, bd, boost_this_step + e * K, colours + e * K * 3, K, S, y, x, rgb);
                        for (int ch = 0; ch < 3; ++ch) o[(ch * w + j) * w + k] = (float)rgb[ch] / 255.0f; /* :296 */
                    }
            }
        }
    }
    return ORACLE_OK;
}

/* MultiSnake.step (:462-731).  actions is (K,N): actions[i*N + e] = agent_i's action in env e.
 * Per-agent outputs are (N*K) in the reference's agent order env*K + i. */
int oracle_multi_step(float *foods, float *heads, float *bodies, uint8_t *dones, int64_t *orientations,
                      const int64_t *actions, uint8_t *boost_this_step, float *rewards, uint8_t *snake_collision,
                      uint8_t *edge_collision, float *food_consumed, float *sizes, uint8_t *all_done,
                      const int16_t *colours, float *obs, int obs_mode, int obs_n, int64_t N, int K, int S,
                      const oracle_multi_cfg *cfg, uint64_t seed, uint64_t call, int64_t env_offset,
                      const oracle_multi_inject *inj)
{
    int C = S * S;
    if (K < 1 || K > 64 || S < 3) return ORACLE_ERR_INVALID;
    for (int64_t e = 0; e < N; ++e) {
        int64_t act[64];
        for (int s = 0; s < K; ++s) act[s] = actions[(int64_t)s * N + e];
        multi_step_env(foods + e * C, heads + e * K * C, bodies + e * K * C, dones + e * K, orientations + e * K, act,
                       boost_this_step + e * K, rewards + e * K, snake_collision + e * K, edge_collision + e * K,
                       food_consumed + e * K, sizes + e * K, K, S, cfg, seed, call, (uint64_t)(env_offset + e), inj, e);
        uint8_t ad = 1;
        for (int s = 0; s < K; ++s) ad &= dones[e * K + s] != 0;                      /* :703 */
        all_done[e] = ad;
    }
    if (obs != NULL)
        return oracle_multi_observe(foods, heads, bodies, dones, boost_this_step, colours, obs, obs_mode, obs_n, N, K, S);
    return ORACLE_OK;
}

/* ------------------------------------------------------------------------------------------------ reset */

/* availability map of _add_snake (:927-941) / _get_snake_addition (:848-858): not occupied, not within the 3x3
 * dilation of anything occupied, at least l = 2 cells from the border */
static int spawn_available(const float *food, const float *heads, const float *bodies, int K, int S, int y, int x)
{
    int C = S * S;
    if (y < 2 || x < 2 || y > S - 3 || x > S - 3) return 0;
    for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
            int c = (y + dy) * S + x + dx;
            float sum = food[c];
            for (int s = 0; s < K; ++s) sum += heads[s * C + c] + bodies[s * C + c];
            if (sum > EPS) return 0;
        }
    return 1;
}

/* place one 3-segment snake (:959-983 / :879-906); returns 1 if a location existed */
static int spawn_snake(float *food, float *heads, float *bodies, int64_t *orient, int K, int S, int snake,
                       int use_inject, int inj_cell, int inj_dir, uint32_t w_cell, uint32_t w_dir)
{
    int C = S * S;
    int cell = -1, d;
    if (use_inject) {
        cell = inj_cell;
        d = inj_dir;
    } else {
        d = (int)(w_dir >> 30);
        int n = 0;
        for (int y = 0; y < S; ++y)
            for (int x = 0; x < S; ++x) n += spawn_available(food, heads, bodies, K, S, y, x);
        if (n > 0) {
            int k = (int)oracle_mulhi(w_cell, (uint32_t)n);
            for (int y = 0; y < S && cell < 0; ++y)
                for (int x = 0; x < S; ++x)
                    if (spawn_available(food, heads, bodies, K, S, y, x)) {
                        if (k == 0) { cell = y * S + x; break; }
                        --k;
                    }
        }
    }
    orient[snake] = d;                                  /* :793 / :828 (assigned even when nothing spawned) */
    float *hd = heads + snake * C, *bd = bodies + snake * C;
    memset(hd, 0, sizeof(float) * (size_t)C);
    memset(bd, 0, sizeof(float) * (size_t)C);
    if (cell < 0) return 0;
    int sy = cell / S, sx = cell % S;
    int hy = sy + TAP_Y[d], hx = sx + TAP_X[d], ty = sy - TAP_Y[d], tx = sx - TAP_X[d];
    bd[hy * S + hx] = 3.0f;                             /* LENGTH_3_SNAKES, see single_snake.c */
    bd[sy * S + sx] = 2.0f;
    bd[ty * S + tx] = 1.0f;
    hd[hy * S + hx] = 1.0f;
    return 1;
}

/* get_n_colours (:163-169) for one snake from three uniforms */
static void random_colour(const uint32_t w[4], int16_t out[3])
{
    float c0 = oracle_u01(w[0]) / 1.5f, c1 = oracle_u01(w[1]), c2 = oracle_u01(w[2]);
    float norm = sqrtf(c0 * c0 + c1 * c1 + c2 * c2);
    out[0] = (int16_t)(c0 / norm * 192.0f);
    out[1] = (int16_t)(c1 / norm * 192.0f);
    out[2] = (int16_t)(c2 / norm * 192.0f);
}

/* MultiSnake.reset (:771-836).  done_env (N): envs to rebuild.  status (nullable, 1 int32): incremented for
 * every env in which _create_envs found no room for a snake (the reference raises RuntimeError :946-947). */
int oracle_multi_reset(float *foods, float *heads, float *bodies, uint8_t *dones, int64_t *orientations,
                       int16_t *colours, const uint8_t *done_env, int32_t *status, int64_t N, int K, int S,
                       const oracle_multi_cfg *cfg, uint64_t seed, uint64_t call, int64_t env_offset,
                       const oracle_multi_reset_inject *inj)
{
    int C = S * S;
    if (K < 1 || K > 64 || S < 5) return S < 5 ? ORACLE_ERR_UNSUPPORTED : ORACLE_ERR_INVALID; /* seeds need rows 2..S-3 (:938-941) */
    for (int64_t e = 0; e < N; ++e) {
        float *food = foods + e * C, *hd = heads + e * K * C, *bd = bodies + e * K * C;
        uint8_t *dn = dones + e * K;
        int64_t *ori = orientations + e * K;
        uint64_t env_id = (uint64_t)(env_offset + e);
        if (done_env[e]) {                                                   /* :787-798 _create_envs :996-1019 */
            memset(food, 0, sizeof(float) * (size_t)C);
            memset(hd, 0, sizeof(float) * (size_t)K * C);
            memset(bd, 0, sizeof(float) * (size_t)K * C);
            for (int s = 0; s < K; ++s) {
                uint32_t w[4];
                oracle_rng_words(seed, call, env_id, RNG_SPAWN, (uint32_t)s, w);
                int ok = spawn_snake(food, hd, bd, ori, K, S, s, inj != NULL,
                                     inj ? inj->create[(e * K + s) * 2] : -1, inj ? inj->create[(e * K + s) * 2 + 1] : 0,
                                     w[0], w[1]);
                if (!ok && status) status[0] += 1;
                dn[s] = 0;                                                   /* :798 */
            }
            uint32_t w[4];
            oracle_rng_words(seed, call, env_id, RNG_RESET, 0, w);
            add_one_food(food, hd, bd, K, S, inj != NULL, inj ? inj->create_food[e] : -1, w[3]); /* :1016-1017 */
        }
        if (cfg->colour_random)                                              /* :800-803 */
            for (int s = 0; s < K; ++s) {
                if (!dn[s]) continue;
                int16_t *col = colours + (e * K + s) * 3;
                if (inj) memcpy(col, inj->colours + (e * K + s) * 3, 3 * sizeof(int16_t));
                else {
                    uint32_t w[4];
                    oracle_rng_words(seed, call, env_id, RNG_COLOUR, (uint32_t)s, w);
                    random_colour(w, col);
                }
            }
        if (cfg->respawn_any) {                                              /* :805-831 */
            int first = -1;
            for (int s = 0; s < K; ++s) if (dn[s]) { first = s; break; }     /* :812 */
            if (first >= 0) {
                uint32_t w[4];
                oracle_rng_words(seed, call, env_id, RNG_SPAWN, (uint32_t)K, w);
                int ok = spawn_snake(food, hd, bd, ori, K, S, first, inj != NULL, inj ? inj->respawn[e * 2] : -1,
                                     inj ? inj->respawn[e * 2 + 1] : 0, w[0], w[1]);
                dn[first] = (uint8_t)!ok;                                    /* :829 */
            }
        }
    }
    return ORACLE_OK;
}

/* MultiSnake.check_consistency (:733-769) as a per-env bitmask */
enum {
    MCHK_SNAKE = 0xff,        /* OR of the single-snake checks of every living snake (wurm/utils.py:113-164) */
    MCHK_OVERLAP = 0x100,     /* :746-758 two snakes on one cell                                         */
    MCHK_DEAD_NONZERO = 0x200 /* :766-769 a dead snake still has head/body cells                          */
};

int oracle_multi_check(const float *foods, const float *heads, const float *bodies, const uint8_t *dones,
                       uint32_t *err, int64_t N, int K, int S)
{
    int C = S * S;
    for (int64_t e = 0; e < N; ++e) {
        const float *food = foods + e * C;
        uint32_t m = 0;
        for (int s = 0; s < K; ++s) {
            const float *hd = heads + (e * K + s) * C, *bd = bodies + (e * K + s) * C;
            if (dones[e * K + s]) {
                float sum = 0.0f;
                for (int c = 0; c < C; ++c) sum += hd[c] + bd[c];
                if (sum != 0.0f) m |= MCHK_DEAD_NONZERO;
                continue;
            }
            float hs = 0, bs = 0, bm = bd[0], hb = 0, hf = 0;
            for (int c = 0; c < C; ++c) {
                if (!(food[c] == 0.0f || food[c] == 1.0f)) m |= 1;
                hs += hd[c]; bs += bd[c]; hb += hd[c] * bd[c]; hf += hd[c] * food[c];
                if (bd[c] > bm) bm = bd[c];
            }
            if (hs != 1.0f) m |= 2;
            if (!(bs > 0.0f)) m |= 4;
            if (bm != hb) m |= 8;
            if ((sqrtf(8.0f * bs + 1.0f) - 1.0f) / 2.0f != bm) m |= 16;
            if (!(bs >= 6.0f)) m |= 32;
            if (hf != 0.0f) m |= 64;
        }
        for (int c = 0; c < C; ++c) {
            int cnt = 0;
            for (int s = 0; s < K; ++s) cnt += bodies[(e * K + s) * C + c] > EPS;
            if (cnt > 1) m |= MCHK_OVERLAP;
        }
        err[e] = m;
    }
    return ORACLE_OK;
}

/* determine_orientations (wurm/utils.py:36-65) over a (n,3,S,S) batch — used by the reference's tests to seed
 * MultiSnake.orientations (tests/test_multi_snake_env.py:38-44) */
int oracle_orientations(const float *envs, int64_t *out, int64_t n, int S)
{
    int C = S * S;
    for (int64_t i = 0; i < n; ++i) {
        const float *body = envs + i * 3 * C + 2 * C;
        float L = body[0];
        for (int c = 1; c < C; ++c) if (body[c] > L) L = body[c];
        float shift = L - 2.0f;
        int best = 0;
        float best_v = 0.0f;
        for (int f = 0; f < 4; ++f) {
            float m = -INFINITY;
            for (int y = 0; y < S; ++y)
                for (int x = 0; x < S; ++x) {
                    float r = body[y * S + x] - shift;
                    if (r < 0.0f) r = 0.0f;
                    float own = (r - 1.5f * (r > 0.0f ? 1.0f : 0.0f)) * 2.0f;
                    int yy = y + TAP_Y[f], xx = x + TAP_X[f];
                    float nb = 0.0f;
                    if (yy >= 0 && yy < S && xx >= 0 && xx < S) {
                        float q = body[yy * S + xx] - shift;
                        if (q < 0.0f) q = 0.0f;
                        nb = (q - 1.5f * (q > 0.0f ? 1.0f : 0.0f)) * 2.0f;
                    }
                    if (nb - own > m) m = nb - own;
                }
            if (f == 0 || m > best_v) { best_v = m; best = f; }
        }
        out[i] = best;
    }
    return ORACLE_OK;
}

/* get_n_colours at construction (multi_snake.py:143-148): 'random' = one colour per agent, 'fixed' = one colour
 * per snake index shared by all envs (drawn with the reserved env id 0xffffffff). */
int oracle_multi_colours(int16_t *colours, int64_t N, int K, int fixed, uint64_t seed, uint64_t call, int64_t env_offset)
{
    for (int64_t e = 0; e < N; ++e)
        for (int s = 0; s < K; ++s) {
            uint32_t w[4];
            oracle_rng_words(seed, call, fixed ? 0xffffffffull : (uint64_t)(env_offset + e), RNG_COLOUR, (uint32_t)s, w);
            random_colour(w, colours + (e * K + s) * 3);
        }
    return ORACLE_OK;
}
