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
#include <cstdlib>

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
    asm volatile("s_nop 1\n\t"
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

// The policy of one wave: lane j is hidden unit j of both layers and column j of the five output rows.
template <int NOBS>
struct Policy {
    static constexpr int W = 2 * NOBS + 1, W2 = W * W, E = 3 * W2, EP = (E + 3) & ~3, H = 64;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 w1[EP / 2], w2[H / 2]; // (even input, odd input) pairs of the lane's weight rows
    float wp0, wp1, wp2, wp3, wv, bias1, bias2, bp0, bp1, bp2, bp3, bv0;
    float *lds_x, *lds_h1;    // LDS: x[EP] (policy input, zero padded to a multiple of 4), then the 64 first-layer activations

    __device__ __forceinline__ void load(const float *params, const float *x0, int lane)
    {
        const float *W1 = params, *b1 = W1 + (long long)H * E, *W2p = b1 + H, *b2 = W2p + H * H, *Wp = b2 + H,
                    *bp = Wp + 4 * H, *Wv = bp + 4, *bv = Wv + H;
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
        wp0 = Wp[lane]; wp1 = Wp[H + lane]; wp2 = Wp[2 * H + lane]; wp3 = Wp[3 * H + lane]; wv = Wv[lane];
        bias1 = b1[lane]; bias2 = b2[lane];
        bp0 = bp[0]; bp1 = bp[1]; bp2 = bp[2]; bp3 = bp[3]; bv0 = bv[0];
        lds_x = (float *)wurm_lds;
        lds_h1 = lds_x + EP;
        for (int k = lane; k < EP; k += 64) lds_x[k] = k < E ? x0[k] : 0.0f;
    }

    // probs, value = model(x in LDS) (wurm/agents/feedforward.py:24-28); action = Categorical(probs).sample() with the
    // uniform u (experiments/main.py:208-210).  All results wave-uniform.
    __device__ __forceinline__ int act(int lane, float u, float &p0, float &p1, float &p2, float &p3, float &value) const
    {
        wave_lds_sync();
        // The broadcast reads are issued in batches of 8 x 16 bytes and each batch is followed by its 32 fmafs
        // (sched_barrier): left alone the scheduler keeps two reads in flight and the lone wave eats one LDS latency
        // per 4 inputs; a whole layer in flight costs 76 registers and with them the occupancy that large batches need.
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
        const float l0 = t0 + bp0, l1 = t1 + bp1, l2 = t2 + bp2, l3 = t3 + bp3;
        value = t4 + bv0;
        const float m = fmaxf(fmaxf(l0, l1), fmaxf(l2, l3));
        const float e0 = exp_spec(l0 - m), e1 = exp_spec(l1 - m), e2 = exp_spec(l2 - m), e3 = exp_spec(l3 - m);
        const float rs = 1.0f / (((e0 + e1) + e2) + e3);
        p0 = e0 * rs; p1 = e1 * rs; p2 = e2 * rs; p3 = e3 * rs;
        const float c0 = p0, c1 = c0 + p1, c2 = c1 + p2;
        return uniform((u >= c0 ? 1 : 0) + (u >= c1 ? 1 : 0) + (u >= c2 ? 1 : 0));
    }
};

// what lane j keeps of step t0 + j, flushed every 64 steps
struct PolicyRecord {
    int act, flags; // sanitised action; done | self collision << 1 | edge collision << 2 | ate << 3
    float val, p0, p1, p2, p3;

    __device__ __forceinline__ void flush(const PolicyArgs &p, long long i) const
    {
        p.actions[i] = (long long)act;
        p.values[i] = val;
        *(float4 *)(p.probs + 4 * i) = make_float4(p0, p1, p2, p3);
        p.reward[i] = (flags & 8) ? 1.0f : 0.0f;
        p.done[i] = (uint8_t)(flags & 1);
        p.selfc[i] = (uint8_t)((flags >> 1) & 1);
        p.edgec[i] = (uint8_t)((flags >> 2) & 1);
    }
};

// the acting loop on the generic scalar-carry path (S <= 11)
template <int NOBS>
__device__ __forceinline__ void policy_generic_loop(const PolicyArgs &p, long long env, float *__restrict__ envp, const Geo &g,
                                                    Env<2> &e, Fast &f, const Policy<NOBS> &pol)
{
    constexpr int CPL = 2, E = Policy<NOBS>::E;
    const int lane = g.lane;
    const u64 env_id = (u64)(p.env_offset + env);
    const Crop cg = make_crop(lane, NOBS);
    const long long obs_stride = p.N * E;
    float *obs_t = p.obs + env * E;
    u64 call = p.call;
    for (long long t0 = 0; t0 < p.T; t0 += 64) {
        const int nt = (int)min((long long)64, p.T - t0);
        const long long my_t = t0 + lane;
        // the sampling uniform of step t0 + lane
        const float my_u = u01(rng_words(p.seed, p.call + 2ull * (u64)my_t, env_id, RNG_POLICY, 0).w[0]);
        PolicyRecord rec = {0, 0, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        for (int j = 0; j < nt; ++j, obs_t += obs_stride, call += 2) {
            float p0, p1, p2, p3, value;
            const int a = pol.act(lane, __int_as_float(lane_value(__float_as_int(my_u), j)), p0, p1, p2, p3, value);
            // env.step(action) (single_snake.py:197-304), crop to HBM and to LDS, env.reset(done)
            StepOut out;
            fast_step<CPL>(e, g, f, a, a, out, p.seed, call, env_id, false, -1);
            fast_partial_small<CPL>(e, g, f, obs_t, cg, pol.lds_x);
            if (lane == j) {
                rec.act = (int)out.action;
                rec.flags = out.done | (out.selfc << 1) | (out.edgec << 2) | (out.reward != 0.0f ? 8 : 0);
                rec.val = value; rec.p0 = p0; rec.p1 = p1; rec.p2 = p2; rec.p3 = p3;
            }
            if (out.done) fast_reset<CPL>(e, g, f, p.seed, call + 1ull, env_id, nullptr);
        }
        if (lane < nt) rec.flush(p, my_t * p.N + env);
    }
    fast_sync_bits<CPL>(e, g, f);
    store_state<CPL, true>(envp, g, e);
}

template <int NOBS>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(NOBS <= 2 ? 2 : 1)))
void policy_rollout_kernel(PolicyArgs p)
{
    constexpr int CPL = 2;
    static_assert(Policy<NOBS>::W2 <= 64, "one window cell per lane");
    const long long env = xcd_block(blockIdx.x, gridDim.x);
    const Geo g = make_geo<CPL>(p.S);
    float *envp = p.envs + env * 3 * g.C;
    Env<CPL> e;
    load_state<CPL, true>(envp, g, e);
    Fast f = {-1, 0, 0, 0, 0, -1};
    if (!uniform((int)fast_init<CPL>(e, g, f))) {
        if (g.lane == 0) p.status[env] = 1;
        return;
    }
    if (g.lane == 0) p.status[env] = 0;
    Policy<NOBS> pol;
    pol.load(p.params, p.obs0 + env * Policy<NOBS>::E, g.lane);
    policy_generic_loop<NOBS>(p, env, envp, g, e, f, pol);
}

// The same loop on the 9x9 machinery of rollout_s9_kernel (single_snake.hip: cell codes 8 * row + column, one lane per
// interior cell, ring / body / food in one bit test, crop liveness from a per-lane table).  Envs that are well formed
// but outside that kernel's extra preconditions (body or food on the ring) take the generic loop.
template <int NOBS>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(NOBS <= 2 ? 2 : 1)))
void policy_rollout_s9_kernel(PolicyArgs p)
{
    constexpr int CPL = 2, S = 9, E = Policy<NOBS>::E, W = Policy<NOBS>::W, W2 = Policy<NOBS>::W2;
    const long long env = xcd_block(blockIdx.x, gridDim.x);
    const Geo g = make_geo<CPL>(S);
    const int lane = g.lane;
    float *envp = p.envs + env * 3 * (S * S);
    Env<CPL> e;
    load_state<CPL, true>(envp, g, e);
    Fast f = {-1, 0, 0, 0, 0, -1};
    if (!uniform((int)fast_init<CPL>(e, g, f))) {
        if (lane == 0) p.status[env] = 1;
        return;
    }
    if (lane == 0) p.status[env] = 0;
    Policy<NOBS> pol;
    pol.load(p.params, p.obs0 + env * E, lane);
    bool lean;
    {
        const bool head_in = f.hc >= 0 && (unsigned)(f.hy - 1) < 7u && (unsigned)(f.hx - 1) < 7u;
        const int fy = f.food >= 0 ? div_size(f.food, g.rcpS) : 1, fx = f.food >= 0 ? f.food - fy * S : 1;
        const bool food_in = (unsigned)(fy - 1) < 7u && (unsigned)(fx - 1) < 7u;
        int under_food = 0;
        if (f.food >= 0) under_food = lane_value(f.food >= 64 ? e.body[1] : e.body[0], f.food & 63);
        const bool ring_body = lane_mask((e.body[0] > 0 && !(g.interior & 1)) || (e.body[1] > 0 && !(g.interior & 2))) != 0;
        lean = head_in && food_in && under_food == 0 && !ring_body;
    }
    if (!uniform((int)lean)) {
        policy_generic_loop<NOBS>(p, env, envp, g, e, f, pol);
        return;
    }

    const u64 env_id = (u64)(p.env_offset + env);
    const int ly = lane >> 3, lx = lane & 7;
    const bool lane_in = ly >= 1 && lx >= 1;
    const int my_cell = ly * S + lx;
    const u64 RING = ~lane_mask(lane_in);
    int ex = lane_in ? __float2int_rn(envp[2 * S * S + my_cell]) : 0;
    int c = uniform(f.hy) * 8 + uniform(f.hx), L = uniform(f.L), o16 = uniform(f.o) << 4;
    int foodc = -1;
    if (f.food >= 0) {
        const int fy = uniform(div_size(f.food, g.rcpS));
        foodc = fy * 8 + (uniform(f.food) - fy * S);
    }
    u64 XF = RING | (foodc >= 0 ? 1ull << foodc : 0);
    int G = L;
    // move table: lane = orientation * 16 + action -> sanitised action & 7 | next orientation << 4 | (code step & 63) << 6
    int move_tab;
    {
        const int to = lane >> 4, ta = lane & 3;
        const int a_out = to == ta ? (to ^ 2) : ta;            // single_snake.py:221-222
        move_tab = (a_out & 7) | ((a_out ^ 2) << 4) | (((-tap_y(a_out) * 8 - tap_x(a_out)) & 63) << 6);
    }
    const int w = min(lane, W2 - 1), wy = div_size(w, 1.0f / (float)W), wx = w - wy * W;
    const int dy0 = wy - NOBS, dx0 = wx - NOBS, code_d = dy0 * 8 + dx0;
    const float green = (w == NOBS * W + NOBS) ? 1.0f : 127.0f / 255.0f;
    u64 live_tab = 0;
#pragma unroll
    for (int y = 1; y <= 7; ++y) {
        u32 cols = 0;
#pragma unroll
        for (int x = 1; x <= 7; ++x)
            if ((unsigned)(x + dx0 - 1) < 7u) cols |= 1u << x;
        if ((unsigned)(y + dy0 - 1) < 7u) live_tab |= (u64)cols << (8 * y);
    }
    const u32 off_r = (u32)w * 4u, off_g = (u32)(W2 + w) * 4u, off_b = (u32)(2 * W2 + w) * 4u;
    const long long obs_stride = p.N * E;
    float *obs_t = p.obs + env * E;

    for (long long t0 = 0; t0 < p.T; t0 += 64) {
        const int nt = (int)min((long long)64, p.T - t0);
        const long long my_t = t0 + lane;
        const u64 my_call = p.call + 2ull * (u64)my_t;
        const float my_u = u01(rng_words(p.seed, my_call, env_id, RNG_POLICY, 0).w[0]);
        const S9Reset my_reset = s9_reset_draw(p.seed, my_call + 1ull, env_id);
        const int my_food = (int)rng_words(p.seed, my_call, env_id, RNG_FOOD, 0).w[0];
        int my_rec = 0, my_ate = 0, my_fl = 0;
        float my_val = 0.0f, my_p0 = 0.0f, my_p1 = 0.0f, my_p2 = 0.0f, my_p3 = 0.0f;
        {
            const int T = G - L;
            ex = max(ex - T, 0);
            G = L;
        }
        for (int j = 0; j < nt; ++j) {
            const u64 lane_j = 1ull << j;
            float p0, p1, p2, p3, value;
            const int a = pol.act(lane, __int_as_float(lane_value(__float_as_int(my_u), j)), p0, p1, p2, p3, value);
            my_val = __int_as_float(keep_in_lane(__float_as_int(my_val), __float_as_int(value), lane_j));
            my_p0 = __int_as_float(keep_in_lane(__float_as_int(my_p0), __float_as_int(p0), lane_j));
            my_p1 = __int_as_float(keep_in_lane(__float_as_int(my_p1), __float_as_int(p1), lane_j));
            my_p2 = __int_as_float(keep_in_lane(__float_as_int(my_p2), __float_as_int(p2), lane_j));
            my_p3 = __int_as_float(keep_in_lane(__float_as_int(my_p3), __float_as_int(p3), lane_j));
            // ---- step: as rollout_s9_kernel, with the move looked up from the sampled action
            const int ent = lane_value(move_tab, o16 + a);
            o16 = ent & 48;
            c += (ent << 20) >> 26;
            G += 1;
            int ate;
            asm("s_cmp_eq_u32 %1, %2\n\ts_cselect_b32 %0, 1, 0" : "=s"(ate) : "s"(c), "s"(foodc) : "scc");
            L += ate;
            const int T = G - L;
            const u64 body = lane_mask(ex > T);
            const u64 head = 1ull << (c & 63);
            ex = keep_in_lane(ex, G, head);
            const u64 occ = body | head;
            unsigned inside;
            {
                u64 lv;
                asm("v_lshrrev_b64 %0, %1, %2" : "=v"(lv) : "s"(c), "v"(live_tab));
                inside = (u32)lv & 1u;
            }
            int code = c + code_d;
            u64 mask = occ;
            int event = 0;
            if (__builtin_expect((((body | XF) >> (c & 63)) & 1) != 0, 0)) {
                if (c == foodc) {
                    my_ate = keep_in_lane(my_ate, 1, lane_j);
                    const u64 fr = ~(occ | RING);
                    const int n_free = popc64(fr);
                    foodc = -1;
                    if (n_free > 0) {
                        const int K = (int)mulhi_range((u32)lane_value(my_food, j), (u32)n_free);
                        foodc = first_bit(lane_mask((int)((fr >> lane) & 1) & (int)(rank_below(fr) == K)));
                    }
                    XF = RING | (foodc >= 0 ? 1ull << foodc : 0);
                }
                event = ((RING >> (c & 63)) & 1) ? 2 : ((body >> (c & 63)) & 1) ? 1 : 0;
                if (event == 2) {
                    const int ai = ent & 3, pc = c - ((ent << 20) >> 26);
                    const int hy = (pc >> 3) - tap_y(ai), hx = (pc & 7) - tap_x(ai);
                    inside = max((unsigned)(hy + dy0 - 1), (unsigned)(hx + dx0 - 1)) < 7u ? 1u : 0u;
                    code = (hy + dy0) * 8 + hx + dx0;
                    mask = body;
                }
            }
            {   // crop to HBM and to LDS (the policy input of the next step)
                u64 sh;
                asm("v_lshrrev_b64 %0, %1, %2" : "=v"(sh) : "v"(code), "s"(mask));
                const unsigned taken = (u32)sh & 1u, both = inside & taken;
                float vr, vb, vg;
                u64 m_free, m_not_food, m_taken;
                asm("v_cmp_gt_u32_e64 %3, %6, %7\n\t"
                    "v_cmp_ne_u32_e64 %4, %8, %9\n\t"
                    "v_cmp_ne_u32_e64 %5, 0, %10\n\t"
                    "v_cndmask_b32_e64 %0, 0, 1.0, %3\n\t"
                    "v_cndmask_b32_e64 %1, 0, %0, %4\n\t"
                    "v_cndmask_b32_e64 %2, %1, %11, %5"
                    : "=&v"(vr), "=&v"(vb), "=&v"(vg), "=&s"(m_free), "=&s"(m_not_food), "=&s"(m_taken)
                    : "v"(inside), "v"(taken), "s"(foodc), "v"(code), "v"(both), "v"(green));
                asm volatile("global_store_dword %0, %1, %6\n\tglobal_store_dword %2, %3, %6\n\tglobal_store_dword %4, %5, %6"
                             : : "v"(off_r), "v"(vr), "v"(off_g), "v"(vg), "v"(off_b), "v"(vb), "s"(obs_t) : "memory");
                obs_t += obs_stride;
                pol.lds_x[w] = vr;
                pol.lds_x[W2 + w] = vg;
                pol.lds_x[2 * W2 + w] = vb;
            }
            my_rec = keep_in_lane(my_rec, ent, lane_j);
            if (__builtin_expect(event != 0, 0)) {
                my_fl = keep_in_lane(my_fl, event, lane_j);
                const int ra = lane_value(my_reset.a, j), rb = lane_value(my_reset.b, j);
                o16 = (ra & 3) << 4;
                foodc = ra >> 2;
                XF = RING | (1ull << foodc);
                c = rb & 127;
                const int sc = (rb >> 7) & 127, tc = rb >> 14;
                ex = lane == tc ? T + 1 : 0; ex = lane == sc ? T + 2 : ex; ex = lane == c ? T + 3 : ex;
                L = 3;
                G = T + 3;
            }
        }
        if (lane < nt) {
            PolicyRecord rec;
            rec.act = (my_rec << 29) >> 29;
            rec.flags = (my_fl != 0 ? 1 : 0) | ((my_fl & 1) << 1) | ((my_fl >> 1) << 2) | (my_ate << 3);
            rec.val = my_val; rec.p0 = my_p0; rec.p1 = my_p1; rec.p2 = my_p2; rec.p3 = my_p3;
            rec.flush(p, my_t * p.N + env);
        }
    }
    if (lane_in) {
        const int T = G - L;
        envp[my_cell] = lane == foodc ? 1.0f : 0.0f;
        envp[S * S + my_cell] = lane == c ? 1.0f : 0.0f;
        envp[2 * S * S + my_cell] = (float)max(ex - T, 0);
    }
}

static int launch_policy_rollout(const PolicyArgs &p, int obs_n, void *stream)
{
    const int W2 = (2 * obs_n + 1) * (2 * obs_n + 1), EP = (3 * W2 + 3) & ~3;
    const size_t lds = (size_t)(EP + 64) * sizeof(float);
    dim3 grid((unsigned)p.N), block(64);
    hipStream_t st = (hipStream_t)stream;
    (void)hipGetLastError();
    const bool s9 = p.S == 9 && !opt.policy_generic; // (debug switch: time / test the generic loop on 9x9 grids)
    switch (obs_n) {
    case 0:
        if (s9) WURM_LAUNCH(policy_rollout_s9_kernel<0>, grid, block, lds, st, p);
        else WURM_LAUNCH(policy_rollout_kernel<0>, grid, block, lds, st, p);
        break;
    case 1:
        if (s9) WURM_LAUNCH(policy_rollout_s9_kernel<1>, grid, block, lds, st, p);
        else WURM_LAUNCH(policy_rollout_kernel<1>, grid, block, lds, st, p);
        break;
    case 2:
        if (s9) WURM_LAUNCH(policy_rollout_s9_kernel<2>, grid, block, lds, st, p);
        else WURM_LAUNCH(policy_rollout_kernel<2>, grid, block, lds, st, p);
        break;
    case 3:
        if (s9) WURM_LAUNCH(policy_rollout_s9_kernel<3>, grid, block, lds, st, p);
        else WURM_LAUNCH(policy_rollout_kernel<3>, grid, block, lds, st, p);
        break;
    default: return WURM_ERR_UNSUPPORTED;
    }
    return hipGetLastError() == hipSuccess ? WURM_OK : WURM_ERR_HIP;
}

} // namespace wurm
