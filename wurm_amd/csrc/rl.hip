// rl.hip — the learner-side glue that sits directly on the env's outputs:
//   * A2C return computation: n-step discounted returns and GAE as a reverse scan over time
//     (oscarknagg/wurm wurm/rl/a2c.py:49-66), plus the matching gradient scan so the op can sit in an autograd graph
//     (the reference builds `returns` from `values` / `bootstrap_values` with torch ops, so gradients flow through it);
//   * the per-step logging reductions of experiments/main.py:252-274 as one fused reduction.
// One thread per env walks its column of the (T,N) row-major tensors: every access is coalesced across the wave,
// each env's scan is sequential in t.  Pure streaming: HBM-bound, 13 B per element forward.  fp32, no FMA
// contraction (-ffp-contract=off) so the forward scan is bit-identical to the reference's torch-CPU op sequence.
#include "wurm_device.hpp"
#include "../../include/wurm_hip.h"

namespace wurm {

__global__ __launch_bounds__(256) void a2c_returns_kernel(const float *__restrict__ bootstrap,
                                                          const float *__restrict__ rewards,
                                                          const float *__restrict__ values,
                                                          const uint8_t *__restrict__ dones, float gamma, int use_gae,
                                                          float gamma_lambda, float *__restrict__ returns, long long T,
                                                          long long N)
{
    const long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N || T <= 0) return;
    if (use_gae) { // a2c.py:50-59
        float gae = 0.0f, next = bootstrap[n];
        for (long long t = T - 1; t >= 0; --t) {
            const long long i = t * N + n;
            const float nd = dones[i] ? 0.0f : 1.0f, v = values[i];
            const float delta = rewards[i] + gamma * next * nd - v; // :53-55
            gae = delta + gamma_lambda * nd * gae;                  // :56
            returns[i] = gae + v;                                   // :57
            next = v;
        }
    } else { // a2c.py:60-64
        float R = bootstrap[n] * (dones[(T - 1) * N + n] ? 0.0f : 1.0f); // :61
        for (long long t = T - 1; t >= 0; --t) {
            const long long i = t * N + n;
            const float nd = dones[i] ? 0.0f : 1.0f;
            R = rewards[i] + gamma * R * nd; // :63
            returns[i] = R;
        }
    }
}

// d(loss)/d(values), d(loss)/d(bootstrap) given G = d(loss)/d(returns).  With c_t = gamma_lambda * nd_t (GAE) or
// gamma * nd_t (n-step) and A_t = G_t + c_{t-1} A_{t-1} (forward scan):
//   GAE:    grad_values[m] = G_m - A_m + gamma * nd_{m-1} * A_{m-1};  grad_bootstrap = gamma * nd_{T-1} * A_{T-1}
//   n-step: grad_values = 0;                                          grad_bootstrap = gamma * nd_{T-1} * A_{T-1}
__global__ __launch_bounds__(256) void a2c_returns_backward_kernel(const float *__restrict__ G,
                                                                   const uint8_t *__restrict__ dones, float gamma,
                                                                   int use_gae, float gamma_lambda,
                                                                   float *__restrict__ grad_values,
                                                                   float *__restrict__ grad_bootstrap, long long T,
                                                                   long long N)
{
    const long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float A_prev = 0.0f, nd_prev = 0.0f;
    const float c = use_gae ? gamma_lambda : gamma;
    for (long long t = 0; t < T; ++t) {
        const long long i = t * N + n;
        const float g = G[i];
        const float A = g + c * nd_prev * A_prev;
        if (grad_values) grad_values[i] = use_gae ? (g - A + gamma * nd_prev * A_prev) : 0.0f;
        A_prev = A;
        nd_prev = dones[i] ? 0.0f : 1.0f;
    }
    if (grad_bootstrap) grad_bootstrap[n] = T > 0 ? gamma * nd_prev * A_prev : 0.0f;
}

// sums over the batch of: done, reward, edge collision, self collision, snake length (max of the body channel).
// One env per wavefront; lane 0 adds the env's five values to the accumulators (the caller zeroes them and may
// accumulate many steps before reading them back).
__global__ __launch_bounds__(256) void single_stats_kernel(const float *__restrict__ envs,
                                                           const float *__restrict__ reward,
                                                           const uint8_t *__restrict__ done,
                                                           const uint8_t *__restrict__ selfc,
                                                           const uint8_t *__restrict__ edgec,
                                                           double *__restrict__ accum, long long N, int S)
{
    const int wave = (int)(threadIdx.x >> 6), wpb = (int)(blockDim.x >> 6), lane = (int)(threadIdx.x & 63u);
    const long long env = xcd_block(blockIdx.x, gridDim.x) * wpb + wave;
    if (env >= N) return;
    const int C = S * S;
    const float *body = envs + env * 3 * C + 2 * C;
    float m = 0.0f;
    for (int c = lane; c < C; c += 64) m = fmaxf(m, body[c]);
    const int L = wave_max_i32(__float2int_rn(m));
    if (lane == 0) {
        if (done[env]) atomicAdd(&accum[0], 1.0);
        const float r = reward[env];
        if (r != 0.0f) atomicAdd(&accum[1], (double)r);
        if (edgec[env]) atomicAdd(&accum[2], 1.0);
        if (selfc[env]) atomicAdd(&accum[3], 1.0);
        atomicAdd(&accum[4], (double)L);
    }
}

} // namespace wurm

using namespace wurm;

extern "C" {

int wurm_a2c_returns(const float *bootstrap, const float *rewards, const float *values, const uint8_t *dones,
                     float gamma, int use_gae, float gamma_lambda, float *returns, int64_t num_steps, int64_t num_envs,
                     void *stream)
{
    if (num_steps < 0 || num_envs < 0) return WURM_ERR_INVALID_ARG;
    if (num_steps == 0 || num_envs == 0) return WURM_OK;
    if (!bootstrap || !rewards || !dones || !returns || (use_gae && !values)) return WURM_ERR_INVALID_ARG;
    (void)hipGetLastError();
    WURM_LAUNCH(a2c_returns_kernel, dim3((unsigned)((num_envs + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       bootstrap, rewards, values, dones, gamma, use_gae, gamma_lambda, returns, (long long)num_steps,
                       (long long)num_envs);
    return hipGetLastError() == hipSuccess ? WURM_OK : WURM_ERR_HIP;
}

int wurm_a2c_returns_backward(const float *grad_returns, const uint8_t *dones, float gamma, int use_gae,
                              float gamma_lambda, float *grad_values, float *grad_bootstrap, int64_t num_steps,
                              int64_t num_envs, void *stream)
{
    if (num_steps < 0 || num_envs < 0) return WURM_ERR_INVALID_ARG;
    if (num_envs == 0) return WURM_OK;
    if (!grad_returns || !dones) return WURM_ERR_INVALID_ARG;
    (void)hipGetLastError();
    WURM_LAUNCH(a2c_returns_backward_kernel, dim3((unsigned)((num_envs + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, grad_returns, dones, gamma, use_gae, gamma_lambda, grad_values,
                       grad_bootstrap, (long long)num_steps, (long long)num_envs);
    return hipGetLastError() == hipSuccess ? WURM_OK : WURM_ERR_HIP;
}

int wurm_single_stats(const float *envs, const float *reward, const uint8_t *done, const uint8_t *self_collision,
                      const uint8_t *edge_collision, double *accum, int64_t num_envs, int size, void *stream)
{
    if (num_envs < 0 || size < 3) return WURM_ERR_INVALID_ARG;
    if (num_envs == 0) return WURM_OK;
    if (!envs || !reward || !done || !self_collision || !edge_collision || !accum) return WURM_ERR_INVALID_ARG;
    (void)hipGetLastError();
    WURM_LAUNCH(single_stats_kernel, dim3((unsigned)((num_envs + 3) / 4)), dim3(256), 0, (hipStream_t)stream, envs,
                       reward, done, self_collision, edge_collision, accum, (long long)num_envs, size);
    return hipGetLastError() == hipSuccess ? WURM_OK : WURM_ERR_HIP;
}

} // extern "C"
