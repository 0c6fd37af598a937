/*
 * oracle/rl.c — scalar CPU restatement of the return computation of the reference's A2C
 * (wurm/rl/a2c.py:49-66 in oscarknagg/wurm): n-step discounted returns and GAE, as a reverse scan over time.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle_common.h).  fp32 arithmetic in exactly the reference's operation order (each
 * torch op rounds once; python-float scalars multiply float32 tensors as float32), so results are bit-identical to
 * the reference's torch-CPU path.  Pinned against vectors recorded from the real A2C.loss(return_returns=True)
 * (tests/golden/make_golden_rl.py).
 *
 * Layout: rewards / values / dones / returns are (T, N) row-major (the reference's TrajectoryStore stacks (N,1)
 * tensors along dim 0, wurm/rl/trajectory_store.py:59-89); bootstrap is (N).
 */
#include "oracle_common.h"

int oracle_a2c_returns(const float *bootstrap, const float *rewards, const float *values, const uint8_t *dones,
                       float gamma, int use_gae, float gamma_lambda, float *returns, int64_t T, int64_t N)
{
    if (T < 0 || N < 0) return ORACLE_ERR_INVALID;
    for (int64_t n = 0; n < N; ++n) {
        if (use_gae) {                                                       /* a2c.py:50-59 */
            float gae = 0.0f;
            for (int64_t t = T - 1; t >= 0; --t) {
                float nd = dones[t * N + n] ? 0.0f : 1.0f;                   /* (~dones[t]).to(dtype) */
                float next = (t == T - 1) ? bootstrap[n] : values[(t + 1) * N + n];
                float delta = rewards[t * N + n] + gamma * next * nd - values[t * N + n];   /* :53-55 */
                gae = delta + gamma_lambda * nd * gae;                       /* :56 (gamma*lambda is one python float) */
                returns[t * N + n] = gae + values[t * N + n];                /* :57 */
            }
        } else {                                                             /* a2c.py:60-64 */
            if (T == 0) continue;
            float R = bootstrap[n] * (dones[(T - 1) * N + n] ? 0.0f : 1.0f); /* :61 */
            for (int64_t t = T - 1; t >= 0; --t) {
                float nd = dones[t * N + n] ? 0.0f : 1.0f;
                R = rewards[t * N + n] + gamma * R * nd;                     /* :63 */
                returns[t * N + n] = R;
            }
        }
    }
    return ORACLE_OK;
}

/* Per-step logging reductions of the single-agent loop (experiments/main.py:252-274): sums over the batch of
 * done, reward, edge / self collisions and the snake length (max of the body channel).  out (5) doubles. */
int oracle_single_stats(const float *envs, const float *reward, const uint8_t *done, const uint8_t *self_collision,
                        const uint8_t *edge_collision, double *out, int64_t N, int S)
{
    int C = S * S;
    for (int i = 0; i < 5; ++i) out[i] = 0.0;
    for (int64_t n = 0; n < N; ++n) {
        out[0] += done[n] != 0;
        out[1] += reward[n];
        out[2] += edge_collision[n] != 0;
        out[3] += self_collision[n] != 0;
        const float *body = envs + n * 3 * C + 2 * C;
        float m = body[0];
        for (int c = 1; c < C; ++c) if (body[c] > m) m = body[c];
        out[4] += m;
    }
    return ORACLE_OK;
}
