// policy_rollout.hpp — the acting half of the reference's single-agent loop as ONE kernel.
// Included by single_snake.hip (it uses that file's Env / Geo / Fast machinery); not a standalone header.
//
// Per env (= per wave) and per step, experiments/main.py:207-212,227:
//     probs, value = model(state)            FeedforwardAgent (wurm/agents/feedforward.py:8-28): E -> 64 -> 64 -> {4, 1}
//     action = Categorical(probs).sample()
//     state, reward, done, info = env.step(action)      (`state`: the PRE-reset observation)
//     env.reset(done)
// so the observation never leaves the chip between env and policy: the crop of step t is written to HBM (the learner
// wants it) and, from the same registers, to LDS, where the 64 lanes — one hidden unit each — read it back with
// broadcast ds_read_b128s as the input of step t+1.
//
// Arithmetic spec (what the reference leaves open; identical in oracle/policy.c, so the two are bit-identical):
// a hidden unit is TWO interleaved fmaf chains — even inputs starting from the bias, odd inputs starting from 0 — added
// at the end (one v_pk_fma_f32 advances both: it issues as fast as a scalar v_fma_f32); the five head outputs are
// bias + a fixed-shape tree sum of the 64 products (pairs, quads, eights, sixteens, then the four
// 16-lane rows as (r0+r1)+(r2+r3) — the shape of a DPP wave reduction); exp_spec (Cody-Waite + degree-6 polynomial in fmaf / ldexp); softmax
// with the sum ((e0+e1)+e2)+e3 and one IEEE reciprocal; inverse-CDF sampling with
// u = u01(Philox(seed; env, step call, RNG_POLICY).w0).
//
// Domain: SingleSnake, grids of at most 128 cells (S <= 11), partial_n crop with n <= 3, envs in a well-formed state
// (fast_init: what reset / step+reset produce).  An env outside the domain is left untouched and flagged in `status`.
#pragma once

namespace wurm {

struct PolicyArgs {
    float *envs;
    const float *obs0;   // (N,E) the observation the policy acts on at step 0
    const float *params; // W1 (64,E) b1 (64) W2 (64,64) b2 (64) Wp (4,64) bp (4) Wv (64) bv (1)
    long long *actions;  // (T,N) sanitised sampled actions
    float *probs;        // (T,N,4)
    float *values;       // (T,N)
    float *reward;       // (T,N)
    uint8_t *done, *selfc, *edgec; // (T,N)
    float *obs;          // (T,N,E) observation returned by step t (pre-reset) = policy input of step t+1
    uint8_t *status;     // (N) 0: done, 1: env outside the domain (left untouched, outputs not written)
    long long N, T;
    int S;
    u64 seed, call;
    long long env_offset;
};

__device__ __forceinline__ float exp_spec(float x) // oracle/policy.c: oracle_exp_spec
{
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
    return x > -87.3f ? ldexpf(p, (int)n) : 0.0f;
}

// Five wave-wide fp32 sums at once, each with the fixed association of oracle/policy.c: tree_sum — a DPP butterfly
// inside each row of 16 lanes (pairs, quads, eights, sixteens), then (r0 + r1) + (r2 + r3) over the four rows with the
// wave-level row broadcasts; the totals end up in lane 63.  Written out because the compiler turns every
// `v += dpp(v)` into mov + nop + mov_dpp + add; interleaving the five chains keeps every DPP read two or more
// instructions behind the write it depends on, so no wait states are needed inside the block.
__device__ __forceinline__ void tree_sum5(float &a, float &b, float &c, float &d, float &e)
{
#define WURM_DPP5(CTRL)                                   \
    "v_add_f32_dpp %0, %0, %0 " CTRL "\n\t"               \
    "v_add_f32_dpp %1, %1, %1 " CTRL "\n\t"               \
    "v_add_f32_dpp %2, %2, %2 " CTRL "\n\t"               \
    "v_add_f32_dpp %3, %3, %3 " CTRL "\n\t"               \
    "v_add_f32_dpp %4, %4, %4 " CTRL "\n\t"
    asm("s_nop 1\n\t"
        WURM_DPP5("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
        WURM_DPP5("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
        WURM_DPP5("row_half_mirror row_mask:0xf bank_mask:0xf")
        WURM_DPP5("row_mirror row_mask:0xf bank_mask:0xf")
        WURM_DPP5("row_bcast:15 row_mask:0xa bank_mask:0xf")
        WURM_DPP5("row_bcast:31 row_mask:0xc bank_mask:0xf")
        "s_nop 1"
        : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e));
#undef WURM_DPP5
    a = __int_as_float(lane_value(__float_as_int(a), 63));
    b = __int_as_float(lane_value(__float_as_int(b), 63));
    c = __int_as_float(lane_value(__float_as_int(c), 63));
    d = __int_as_float(lane_value(__float_as_int(d), 63));
    e = __int_as_float(lane_value(__float_as_int(e), 63));
}

template <int NOBS>
__global__ __launch_bounds__(64) void policy_rollout_kernel(PolicyArgs p)
{
    constexpr int CPL = 2, W = 2 * NOBS + 1, W2 = W * W, E = 3 * W2, EP = (E + 3) & ~3, H = 64;
    static_assert(W2 <= 64, "one window cell per lane");
    const long long env = blockIdx.x;
    const Geo g = make_geo<CPL>(p.S);
    const int lane = g.lane;
    float *envp = p.envs + env * 3 * g.C;
    Env<CPL> e;
    load_state<CPL, true>(envp, g, e);
    Fast f = {-1, 0, 0, 0, 0, -1};
    if (!uniform((int)fast_init<CPL>(e, g, f))) {
        if (lane == 0) p.status[env] = 1;
        return;
    }
    if (lane == 0) p.status[env] = 0;
    const u64 env_id = (u64)(p.env_offset + env);

    // LDS: x[EP] (policy input, zero padded to a multiple of 4), then 64 floats for the first hidden layer
    float *lds_x = (float *)wurm_lds, *lds_h1 = lds_x + EP;
    // weights of "my" unit in registers: lane j is hidden unit j of both layers
    const float *W1 = p.params, *b1 = W1 + (long long)H * E, *W2p = b1 + H, *b2 = W2p + H * H, *Wp = b2 + H, *bp = Wp + 4 * H,
                *Wv = bp + 4, *bv = Wv + H;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 w1[EP / 2], w2[H / 2]; // (even input, odd input) pairs
#pragma unroll
    for (int k = 0; k < EP / 2; ++k) {
        w1[k].x = 2 * k < E ? W1[(long long)lane * E + 2 * k] : 0.0f;
        w1[k].y = 2 * k + 1 < E ? W1[(long long)lane * E + 2 * k + 1] : 0.0f;
    }
#pragma unroll
    for (int k = 0; k < H / 2; ++k) {
        w2[k].x = W2p[lane * H + 2 * k];
        w2[k].y = W2p[lane * H + 2 * k + 1];
    }
    // heads: lane k holds column k of the five output rows (4 action scores, 1 value)
    const float wp0 = Wp[lane], wp1 = Wp[H + lane], wp2 = Wp[2 * H + lane], wp3 = Wp[3 * H + lane], wv = Wv[lane];
    const float bias1 = b1[lane], bias2 = b2[lane];
    const float bp0 = bp[0], bp1 = bp[1], bp2 = bp[2], bp3 = bp[3], bv0 = bv[0];

    for (int k = lane; k < EP; k += 64) lds_x[k] = k < E ? p.obs0[env * E + k] : 0.0f;
    const Crop cg = make_crop(lane, NOBS);
    const long long obs_stride = p.N * E;
    float *obs_t = p.obs + env * E;
    u64 call = p.call;

    for (long long t0 = 0; t0 < p.T; t0 += 64) {
        const int nt = (int)min((long long)64, p.T - t0);
        const long long my_t = t0 + lane;
        // the sampling uniform of step t0 + lane
        const float my_u = u01(rng_words(p.seed, p.call + 2ull * (u64)my_t, env_id, RNG_POLICY, 0).w[0]);
        int my_act = 0, my_flags = 0;
        float my_val = 0.0f, my_p0 = 0.0f, my_p1 = 0.0f, my_p2 = 0.0f, my_p3 = 0.0f;

        for (int j = 0; j < nt; ++j, obs_t += obs_stride, call += 2) {
            // ---- policy forward (wurm/agents/feedforward.py:24-28)
            wave_lds_sync();
            // The broadcast reads are issued in batches of 8 x 16 bytes and each batch is followed by its 32 fmafs
            // (sched_barrier): left alone the scheduler keeps two reads in flight and the lone wave eats one LDS
            // latency per 4 inputs; a whole layer in flight costs 76 registers and with them the occupancy that large
            // batches need.
            constexpr int BATCH = 8;
            f2 acc2 = {bias1, 0.0f};
#pragma unroll
            for (int k0 = 0; k0 < EP / 4; k0 += BATCH) {
                float4 xs[BATCH];
#pragma unroll
                for (int k = 0; k < BATCH; ++k)
                    if (k0 + k < EP / 4) xs[k] = *(const float4 *)(lds_x + 4 * (k0 + k));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < BATCH; ++k)
                    if (k0 + k < EP / 4) {
                        const f2 lo = {xs[k].x, xs[k].y}, hi = {xs[k].z, xs[k].w};
                        acc2 = __builtin_elementwise_fma(w1[2 * (k0 + k)], lo, acc2);
                        acc2 = __builtin_elementwise_fma(w1[2 * (k0 + k) + 1], hi, acc2);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
            float acc = acc2.x + acc2.y;
            lds_h1[lane] = acc > 0.0f ? acc : 0.0f;
            wave_lds_sync();
            acc2.x = bias2;
            acc2.y = 0.0f;
#pragma unroll
            for (int k0 = 0; k0 < H / 4; k0 += BATCH) {
                float4 hs[BATCH];
#pragma unroll
                for (int k = 0; k < BATCH; ++k) hs[k] = *(const float4 *)(lds_h1 + 4 * (k0 + k));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < BATCH; ++k) {
                    const f2 lo = {hs[k].x, hs[k].y}, hi = {hs[k].z, hs[k].w};
                    acc2 = __builtin_elementwise_fma(w2[2 * (k0 + k)], lo, acc2);
                    acc2 = __builtin_elementwise_fma(w2[2 * (k0 + k) + 1], hi, acc2);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            acc = acc2.x + acc2.y;
            const float h2 = acc > 0.0f ? acc : 0.0f;
            float t0 = wp0 * h2, t1 = wp1 * h2, t2 = wp2 * h2, t3 = wp3 * h2, t4 = wv * h2;
            tree_sum5(t0, t1, t2, t3, t4);
            const float l0 = t0 + bp0, l1 = t1 + bp1, l2 = t2 + bp2, l3 = t3 + bp3, value = t4 + bv0;
            // softmax (:28) and Categorical(probs).sample() (main.py:208-210)
            const float m = fmaxf(fmaxf(l0, l1), fmaxf(l2, l3));
            const float e0 = exp_spec(l0 - m), e1 = exp_spec(l1 - m), e2 = exp_spec(l2 - m), e3 = exp_spec(l3 - m);
            const float s = ((e0 + e1) + e2) + e3;
            const float rs = 1.0f / s;
            const float p0 = e0 * rs, p1 = e1 * rs, p2 = e2 * rs, p3 = e3 * rs;
            const float u = __int_as_float(lane_value(__float_as_int(my_u), j));
            const float c0 = p0, c1 = c0 + p1, c2 = c1 + p2;
            const int a = uniform((u >= c0 ? 1 : 0) + (u >= c1 ? 1 : 0) + (u >= c2 ? 1 : 0));

            // ---- env.step(action) (single_snake.py:197-304), crop to HBM and to LDS, env.reset(done)
            StepOut out;
            fast_step<CPL>(e, g, f, a, a, out, p.seed, call, env_id, false, -1);
            fast_partial_small<CPL>(e, g, f, obs_t, cg, lds_x);
            if (lane == j) {
                my_act = (int)out.action;
                my_flags = out.done | (out.selfc << 1) | (out.edgec << 2) | (out.reward != 0.0f ? 8 : 0);
                my_val = value; my_p0 = p0; my_p1 = p1; my_p2 = p2; my_p3 = p3;
            }
            if (out.done) fast_reset<CPL>(e, g, f, p.seed, call + 1ull, env_id, nullptr);
        }
        if (lane < nt) {
            const long long i = my_t * p.N + env;
            p.actions[i] = (long long)my_act;
            p.values[i] = my_val;
            *(float4 *)(p.probs + 4 * i) = make_float4(my_p0, my_p1, my_p2, my_p3);
            p.reward[i] = (my_flags & 8) ? 1.0f : 0.0f;
            p.done[i] = (uint8_t)(my_flags & 1);
            p.selfc[i] = (uint8_t)((my_flags >> 1) & 1);
            p.edgec[i] = (uint8_t)((my_flags >> 2) & 1);
        }
    }
    fast_sync_bits<CPL>(e, g, f);
    store_state<CPL, true>(envp, g, e);
}

static int launch_policy_rollout(const PolicyArgs &p, int obs_n, void *stream)
{
    const int W2 = (2 * obs_n + 1) * (2 * obs_n + 1), EP = (3 * W2 + 3) & ~3;
    const size_t lds = (size_t)(EP + 64) * sizeof(float);
    dim3 grid((unsigned)p.N), block(64);
    hipStream_t st = (hipStream_t)stream;
    (void)hipGetLastError();
    switch (obs_n) {
    case 0: hipLaunchKernelGGL(policy_rollout_kernel<0>, grid, block, lds, st, p); break;
    case 1: hipLaunchKernelGGL(policy_rollout_kernel<1>, grid, block, lds, st, p); break;
    case 2: hipLaunchKernelGGL(policy_rollout_kernel<2>, grid, block, lds, st, p); break;
    case 3: hipLaunchKernelGGL(policy_rollout_kernel<3>, grid, block, lds, st, p); break;
    default: return WURM_ERR_UNSUPPORTED;
    }
    return hipGetLastError() == hipSuccess ? WURM_OK : WURM_ERR_HIP;
}

} // namespace wurm
