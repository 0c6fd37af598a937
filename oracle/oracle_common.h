/*
 * oracle_common.h — shared definitions for the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  Everything under oracle/ is a scalar CPU restatement of the reference's
 * batched env step (oscarknagg/wurm, the wurm/envs modules) used as the checker in tests/, in
 * __graft_entry__.smoke() and as bench.py's cpu_baseline leg.  The product (wurm_amd/) never imports,
 * links or calls it.
 *
 * Parity status: PINNED — the restatement is checked against golden fixtures recorded from the real
 * reference imported in the build container (tests/golden/make_golden.py, tests/test_oracle_golden.py) and
 * against the reference's own known-answer tests (tests/test_oracle_kats.py).
 * What no reference test pins (which free cell food respawns in, reset positions, Bernoulli draws) is
 * replayed from the reference's recorded outcomes through the `inject_*` arguments; the counter-based
 * RNG below is this build's own spec (the reference's torch RNG stream is not restatable).
 */
#ifndef WURM_ORACLE_COMMON_H
#define WURM_ORACLE_COMMON_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* observation modes (reference wurm/envs/single_snake.py:130-195, simple_gridworld.py:111-133,
 * multi_snake.py:283-334) */
enum {
    ORACLE_OBS_DEFAULT = 0,     /* RGB/255 (N,3,S,S)                       */
    ORACLE_OBS_RAW = 1,         /* clone of the state                      */
    ORACLE_OBS_ONE_CHANNEL = 2, /* (N,1,S,S)                               */
    ORACLE_OBS_POSITIONS = 3,   /* (N,4)                                   */
    ORACLE_OBS_PARTIAL = 4,     /* (N,3*(2n+1)^2), window around the head  */
    ORACLE_OBS_NONE = 5         /* skip the observation                    */
};

/* action tensor element types accepted by step() (reference single_snake.py:198-200) */
enum { ORACLE_ACT_I64 = 0, ORACLE_ACT_I32 = 1 };

enum { ORACLE_OK = 0, ORACLE_ERR_INVALID = -1, ORACLE_ERR_UNSUPPORTED = -2 };

/* RNG purposes — one Philox stream per (env, call counter, purpose, sub-block). */
enum {
    RNG_FOOD = 0,        /* word 0: food cell rank                               */
    RNG_RESET = 1,       /* words: seed y, seed x, direction, food cell rank     */
    RNG_DEATH_FOOD_A = 2,/* per-cell uniforms, boost phase  (multi_snake.py:565-576) */
    RNG_DEATH_FOOD_B = 3,/* per-cell uniforms, regular phase(multi_snake.py:662-673) */
    RNG_BOOST_COST = 4,  /* per-snake uniform (multi_snake.py:579)                */
    RNG_RATE_FOOD = 5,   /* per-cell uniforms (multi_snake.py:401)                */
    RNG_SPAWN = 6,       /* sub = snake index: words: cell rank, direction        */
    RNG_COLOUR = 7,      /* sub = snake index: 3 uniforms (multi_snake.py:163-169)*/
    RNG_POLICY = 8       /* word 0: the uniform of Categorical(probs).sample() (experiments/main.py:210) */
};

/* Philox4x32-10 (Salmon et al., SC'11).  counter = (env_id, call_lo, call_hi, purpose | sub<<8),
 * key = (seed_lo, seed_hi). */
static inline void oracle_philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1)
{
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)M0 * c[0];
        uint64_t p1 = (uint64_t)M1 * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += W0; k1 += W1;
    }
}

static inline void oracle_rng_words(uint64_t seed, uint64_t call, uint64_t env_id, uint32_t purpose,
                                    uint32_t sub, uint32_t out[4])
{
    out[0] = (uint32_t)env_id;
    out[1] = (uint32_t)call;
    out[2] = (uint32_t)(call >> 32);
    out[3] = (purpose & 0xffu) | (sub << 8);
    oracle_philox4x32_10(out, (uint32_t)seed, (uint32_t)(seed >> 32));
}

/* uniform integer in [0, n) */
static inline uint32_t oracle_mulhi(uint32_t w, uint32_t n) { return (uint32_t)(((uint64_t)w * n) >> 32); }

/* uniform float32 in [0,1) with 24 random bits (same lattice as torch.rand's float32) */
static inline float oracle_u01(uint32_t w) { return (float)(w >> 8) * (1.0f / 16777216.0f); }

/* q^n in fp32 by binary exponentiation: this operation order is part of this build's RNG specification
 * (wurm_amd/csrc/wurm_device.hpp: pow_n) */
static inline float oracle_pow_n(float q, int n)
{
    float pw = 1.0f, base = q;
    for (int e = n; e; e >>= 1) {
        if (e & 1) pw = pw * base;
        base = base * base;
    }
    return pw;
}

/* Number of successes among n independent Bernoulli(p) trials from ONE uniform u by inversion of the Binomial(n, p)
 * distribution function (wurm_device.hpp: binomial_inverse); pw = oracle_pow_n(1 - p, n) >= ORACLE_BINOMIAL_MIN_P0. */
#define ORACLE_BINOMIAL_MIN_P0 1e-6f
static inline int oracle_binomial_inverse(int n, float p, float pw, float u)
{
    const float r = p / (1.0f - p);
    float pmf = pw, cdf = pw;
    int k = 0;
    while (u >= cdf && k < n) {
        pmf = pmf * ((float)(n - k) * r) / (float)(k + 1);
        const float nc = cdf + pmf;
        ++k;
        if (nc == cdf) break;
        cdf = nc;
    }
    return k;
}

/* per-cell uniform for cell index `cell` of an env.  Cells c, c+64, c+128, c+192 share one Philox block
 * (sub = (c >> 8) * 64 + (c & 63), word = (c >> 6) & 3): on the GPU one lane owns all four. */
static inline float oracle_cell_u01(uint64_t seed, uint64_t call, uint64_t env_id, uint32_t purpose, uint32_t cell)
{
    uint32_t w[4];
    oracle_rng_words(seed, call, env_id, purpose, ((cell >> 8) << 6) | (cell & 63u), w);
    return oracle_u01(w[(cell >> 6) & 3u]);
}

#ifdef __cplusplus
}
#endif
#endif
