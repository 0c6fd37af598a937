/*
 * policy.c — CPU oracle for the policy-in-the-loop rollout.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle_common.h).
 *
 * Restates, per env and per step, the acting half of the reference's single-agent loop
 * (experiments/main.py:207-212,227):
 *     probs, value = model(state)            FeedforwardAgent, wurm/agents/feedforward.py:8-28
 *     action = Categorical(probs).sample()   main.py:208-210
 *     state, reward, done, info = env.step(action)          (`state` = the PRE-reset observation, :212)
 *     env.reset(done)                                        (its observation is discarded, :227)
 * with the architecture the reference's experiments use for partial observations: inputs -> 64 -> 64 (ReLU after
 * both, wurm/modules.py feedforward_block = Linear + ReLU) -> {4 action scores -> softmax, 1 state value}.
 *
 * What the reference fixes: the architecture, softmax, "sample from Categorical(probs)", the data flow.  What it does
 * not fix (torch's accumulation order inside addmm, libm's expf, the multinomial sampler's use of the global RNG
 * stream) is this build's own spec, stated here so that the HIP kernel can be bit-identical to this file:
 *   - a hidden unit: two interleaved fused-multiply-add chains in index order, added at the end:
 *         even = b[j]; odd = 0; for k = 0, 2, ..: even = fmaf(W[j][k], x[k], even); odd = fmaf(W[j][k+1], x[k+1], odd)
 *         unit = even + odd            (a missing last odd input leaves `odd` as it is)
 *     (one v_pk_fma_f32 per pair on the GPU);
 *   - the five head outputs (4 action scores, 1 value): bias added to tree_sum() of the 64 rounded products
 *     W[a][k] * h2[k] — pairs, quads, eights, sixteens, then (r0 + r1) + (r2 + r3) over the four groups of 16
 *     (the association of a DPP reduction over a 64-lane wave);
 *   - exp_spec(): exp on (-inf, 0] by Cody-Waite reduction + a degree-6 polynomial, all in fmaf / ldexpf;
 *   - softmax: e_a = exp_spec(l_a - max l), p_a = e_a * (1 / (((e_0 + e_1) + e_2) + e_3))  (one IEEE reciprocal);
 *   - sampling: u = u01(word 0 of Philox(seed; env id, step call counter, RNG_POLICY)), action = number of
 *     cumulative sums c_0 = p_0, c_1 = c_0 + p_1, c_2 = c_1 + p_2 that are <= u (inverse CDF, clamped to 3).
 * Parity status: PINNED — probabilities and values agree with the REAL reference agent (wurm.agents.FeedforwardAgent,
 * imported in the build container by tests/golden/make_golden_policy.py, weights as torch initialises them and a
 * sharpened copy, real 'partial_n' observations) to 2e-6 absolute / 1e-5 relative: tests/golden/policy_ff_*.npz,
 * tests/test_policy_rollout.py::test_spec_forward_matches_the_reference_agent; the sampling distribution is checked by
 * a chi-square test.  What cannot be pinned is WHICH action torch's multinomial draws from its global RNG stream.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "oracle_common.h"

enum { HIDDEN = 64, N_ACTIONS = 4 };

int oracle_single_step(float *envs, void *actions, int act_dtype, float *reward, uint8_t *done,
                       uint8_t *self_collision, uint8_t *edge_collision, float *obs, int obs_mode, int obs_n,
                       int64_t N, int S, uint64_t seed, uint64_t call, int64_t env_offset,
                       const int32_t *inject_food);
int oracle_single_reset(float *envs, const uint8_t *done, float *obs, int obs_mode, int obs_n, int64_t N, int S,
                        uint64_t seed, uint64_t call, int64_t env_offset, const int32_t *inject_reset);
int64_t oracle_single_obs_elems(int obs_mode, int obs_n, int S);

/* exp(x) for x <= 0 (returns 0 below -87.3): n = rint(x * log2 e), r = x - n ln 2 (two-constant Cody-Waite),
 * p(r) = 1 + r (1 + r (1/2 + r (1/6 + r (1/24 + r (1/120 + r / 720))))), result = ldexp(p, n) */
float oracle_exp_spec(float x)
{
    if (!(x > -87.3f)) return 0.0f;
    const float n = rintf(x * 1.44269504088896341f);
    float r = fmaf(-n, 0.693359375f, x);
    r = fmaf(-n, -2.12194440e-4f, r);
    float p = 1.0f / 720.0f;
    p = fmaf(p, r, 1.0f / 120.0f);
    p = fmaf(p, r, 1.0f / 24.0f);
    p = fmaf(p, r, 1.0f / 6.0f);
    p = fmaf(p, r, 0.5f);
    p = fmaf(p, r, 1.0f);
    p = fmaf(p, r, 1.0f);
    return ldexpf(p, (int)n);
}

/* bias + w . x as two interleaved fmaf chains (even indices from the bias, odd indices from 0) */
static float two_chain_dot(const float *w, const float *x, int K, float bias)
{
    float even = bias, odd = 0.0f;
    for (int k = 0; k < K; k += 2) {
        even = fmaf(w[k], x[k], even);
        if (k + 1 < K) odd = fmaf(w[k + 1], x[k + 1], odd);
    }
    return even + odd;
}

/* sum of 64 floats with the association of a DPP butterfly over a wave: within each group of 16, pairs (i, i^1), quads,
 * eights, sixteens; then (r0 + r1) + (r2 + r3) (row broadcasts) */
static float tree_sum(const float t[HIDDEN])
{
    float r[4];
    for (int g = 0; g < 4; ++g) {
        float q[4];
        for (int m = 0; m < 4; ++m) {
            const float *u = t + 16 * g + 4 * m;
            q[m] = (u[0] + u[1]) + (u[2] + u[3]);
        }
        r[g] = (q[0] + q[1]) + (q[2] + q[3]);
    }
    return (r[0] + r[1]) + (r[2] + r[3]);
}

/* params: W1 (64 x E, row-major [unit][input]), b1 (64), W2 (64 x 64), b2 (64), Wp (4 x 64), bp (4), Wv (64), bv (1)
 * — the layout of torch's Linear.weight / .bias tensors concatenated in that order. */
void oracle_policy_forward(const float *params, int E, const float *x, float probs[4], float *value)
{
    const float *W1 = params, *b1 = W1 + (size_t)HIDDEN * E, *W2 = b1 + HIDDEN, *b2 = W2 + HIDDEN * HIDDEN;
    const float *Wp = b2 + HIDDEN, *bp = Wp + N_ACTIONS * HIDDEN, *Wv = bp + N_ACTIONS, *bv = Wv + HIDDEN;
    float h1[HIDDEN], h2[HIDDEN], l[N_ACTIONS];
    for (int j = 0; j < HIDDEN; ++j) {
        const float acc = two_chain_dot(W1 + (size_t)j * E, x, E, b1[j]);
        h1[j] = acc > 0.0f ? acc : 0.0f;
    }
    for (int j = 0; j < HIDDEN; ++j) {
        const float acc = two_chain_dot(W2 + j * HIDDEN, h1, HIDDEN, b2[j]);
        h2[j] = acc > 0.0f ? acc : 0.0f;
    }
    float t[HIDDEN];
    for (int a = 0; a < N_ACTIONS; ++a) {
        for (int k = 0; k < HIDDEN; ++k) t[k] = Wp[a * HIDDEN + k] * h2[k];
        l[a] = tree_sum(t) + bp[a];
    }
    for (int k = 0; k < HIDDEN; ++k) t[k] = Wv[k] * h2[k];
    *value = tree_sum(t) + bv[0];
    float m = l[0];
    for (int a = 1; a < N_ACTIONS; ++a) m = l[a] > m ? l[a] : m;
    float e[N_ACTIONS];
    for (int a = 0; a < N_ACTIONS; ++a) e[a] = oracle_exp_spec(l[a] - m);
    const float s = ((e[0] + e[1]) + e[2]) + e[3];
    const float rs = 1.0f / s;
    for (int a = 0; a < N_ACTIONS; ++a) probs[a] = e[a] * rs;
}

int oracle_policy_sample(const float probs[4], uint64_t seed, uint64_t call, uint64_t env_id)
{
    uint32_t w[4];
    oracle_rng_words(seed, call, env_id, RNG_POLICY, 0, w);
    const float u = oracle_u01(w[0]);
    const float c0 = probs[0], c1 = c0 + probs[1], c2 = c1 + probs[2];
    return (u >= c0) + (u >= c1) + (u >= c2);
}

/* T iterations of the acting loop over N envs.  obs0 (N,E): the observation the policy acts on at step 0 (the
 * caller's `state`).  Outputs, all (T,N,...): actions (int64, sanitised by step as in the reference), probs (4),
 * values, reward, done, self/edge collision, obs (E) = the observation step t returned (pre-reset), which is the
 * policy input of step t+1. */
int oracle_single_policy_rollout(float *envs, const float *obs0, const float *params, int64_t *actions, float *probs,
                                 float *values, float *reward, uint8_t *done, uint8_t *self_collision,
                                 uint8_t *edge_collision, float *obs, int obs_n, int64_t N, int S, int64_t T,
                                 uint64_t seed, uint64_t call0, int64_t env_offset)
{
    const int64_t E = oracle_single_obs_elems(ORACLE_OBS_PARTIAL, obs_n, S);
    if (E <= 0 || N < 0 || T < 0) return ORACLE_ERR_INVALID;
    for (int64_t t = 0; t < T; ++t) {
        const float *x = t == 0 ? obs0 : obs + (t - 1) * N * E;
        const uint64_t call = call0 + 2 * (uint64_t)t;
        for (int64_t i = 0; i < N; ++i) {
            float p[4], v;
            oracle_policy_forward(params, (int)E, x + i * E, p, &v);
            memcpy(probs + (t * N + i) * 4, p, sizeof p);
            values[t * N + i] = v;
            actions[t * N + i] = oracle_policy_sample(p, seed, call, (uint64_t)(env_offset + i));
        }
        int rc = oracle_single_step(envs, actions + t * N, ORACLE_ACT_I64, reward + t * N, done + t * N,
                                    self_collision + t * N, edge_collision + t * N, obs + t * N * E,
                                    ORACLE_OBS_PARTIAL, obs_n, N, S, seed, call, env_offset, NULL);
        if (rc) return rc;
        rc = oracle_single_reset(envs, done + t * N, NULL, ORACLE_OBS_NONE, 0, N, S, seed, call + 1, env_offset, NULL);
        if (rc) return rc;
    }
    return ORACLE_OK;
}
