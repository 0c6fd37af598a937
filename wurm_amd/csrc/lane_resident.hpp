// lane_resident.hpp — the per-call SingleSnake step (`obs, r, d, info = env.step(a); env.reset(d)`, one launch per iteration)
// for LARGE batches of 9 x 9 envs with the state RESIDENT in compact form between calls.
//
// lane_step_kernel (lane_step.hpp) reads the whole (N, 3, 9, 9) fp32 state every call to find a dozen non-zero elements
// per env: at 65 536 envs that is 63.7 MB of the launch's ~100 MB, and its waves spend 40 % of their life in that read
// (in-kernel timeline).  Here the caller hands over a second, caller-owned buffer — the MIRROR, 32 bytes per
// env: lane_rollout.hpp's per-lane state (64-bit occupancy mask over cell codes 8 * row + column, the body as a queue of
// 2-bit moves, head / tail codes, length, orientation, food code) — which the step kernel reads instead of the state and
// keeps up to date; `envs` itself is still WRITTEN every call (the handful of elements that change: the cells of a
// decaying body, two head cells, the food; a rebuilt env whole), so it always is what the reference would show and any
// other entry point can be called in between.  The caller says whether the mirror is current (wurm_single_call.
// resident_valid: nobody else has written `envs` since the last call that maintained it); if not, a build kernel
// recreates it from `envs` first (the cooperative read + validation of lane_rollout.hpp).
// LAZY (wurm_single_call.resident_lazy): `envs` is not written at all — the ~12 scattered 4-byte stores per env are what
// the eager form's waves spend 40-55 % of their life on — and is brought up to date from the mirror by
// lane_resident_flush_kernel (wurm_single_resident_flush) before anything else looks at it.
//
// One env per lane for the transition; crops of the post-step state (`obs`) and, when asked for, of the state after the
// finished envs are rebuilt (`obs_after`, what reset(done) returns) as (which, env) pair lanes -> bit planes -> flat bit
// strings -> 256-entry float4 table -> aligned 16-byte stores, exactly as phases 3 and 4 of lane_rollout_kernel.
// Contract: fused_step_kernel's without post_reset (deferred reset: envs flagged in p.done_in are rebuilt in front of the
// step with call = p.pre_call).  Domain: S = 9, observation 'partial_2' or none, RNG mode; per env the domain of
// lane_rollout.hpp.  An env outside it — or one that finished and is stepped again without the reset — is stepped by
// fused_step_env (the one-env-per-wave code) on `envs` inside the same launch, and stays on that path until it is rebuilt.
// Follows single_snake.py:197-304 (step), :322-387 (reset), :130-195 (_observe) like the kernels it stands in for.
#pragma once

#include "lane_rollout.hpp"

namespace wurm {

// The mirror: two planes of N uint4.
//   plane 0 [env]: occupancy lo, hi, q0, q1
//   plane 1 [env]: q2, head code | tail code << 7 | length << 14 | orientation << 21 | (food code + 1) << 23, flags,
//                  head row | head column << 4 (true coordinates: the code of a head on the border ring is ambiguous)
constexpr u32 RES_ACT = 1u;       // the env is in the lane kernels' domain and the record describes it
constexpr u32 RES_TERMINAL = 2u;  // the last step finished the env: the record is void unless the next call rebuilds it
constexpr int RES_BYTES = 32;

struct ResidentArgs {
    StepArgs p;
    uint4 *res;
    uint32_t *check_mask; // nullable (N): wurm_single_call.check_mask
};

template <int EPW>
__global__ __launch_bounds__(256) void lane_resident_build_kernel(ResidentArgs a)
{
    typedef LaneRollLds<EPW> Lds;
    extern __shared__ __attribute__((aligned(16))) unsigned char res_lds[];
    const StepArgs &p = a.p;
    const int wave = uniform((int)(threadIdx.x >> 6)), wpb = (int)(blockDim.x >> 6), lane = (int)(threadIdx.x & 63u);
    const long long env0 = ((long long)blockIdx.x * wpb + wave) * EPW;
    if (env0 >= p.N) return;
    unsigned char *lds = res_lds + wave * Lds::BYTES;
    const int nenv = (int)min((long long)EPW, p.N - env0);
    u64 occ;
    u32 q0, q1, q2;
    int c, tc, L, o, food;
    bool act;
    const bool whole = nenv == EPW && (((size_t)p.envs) & 15u) == 0;
    lr_read_block<EPW>(p.envs + env0 * LR_C3, whole, nenv, lane, lds, occ, q0, q1, q2, c, tc, L, o, food, act);
    if (lane < nenv) {
        const u32 pk = (u32)(c & 127) | ((u32)(tc & 127) << 7) | ((u32)(L & 127) << 14) | ((u32)(o & 3) << 21) |
                       ((u32)((food + 1) & 127) << 23);
        a.res[env0 + lane] = make_uint4((u32)occ, (u32)(occ >> 32), q0, q1);
        a.res[p.N + env0 + lane] = make_uint4(q2, pk, act ? RES_ACT : 0u, (u32)(c >> 3) | ((u32)(c & 7) << 4));
    }
}

// `envs` of ONE env from its record, whole wave (uniform arguments): body values by walking the queue from the head —
// a cell the walk visits twice (the snake ran into itself: single_snake.py:252-262 adds the new head's value on top of
// what the cell held) gets the sum.  For the finished env that is stepped again without its reset, LAZY form.
__device__ __forceinline__ void res_materialise_env(float *ep, int lane, u32 w0, u32 w1, u32 w2, int L, int hy, int hx, int fidx)
{
    constexpr int S = 9, C = LR_C;
    int v0 = 0, v1 = 0; // body values of cells lane and lane + 64
    int y = hy, x = hx;
    for (int v = L; v >= 1; --v) {
        const int cell = y * S + x;
        if (cell == lane) v0 += v;
        if (cell == lane + 64) v1 += v;
        const int m = (int)(w0 & 3u);
        y -= lr_dy(m); x -= lr_dx(m);
        w0 = (w0 >> 2) | (w1 << 30); w1 = (w1 >> 2) | (w2 << 30); w2 >>= 2;
    }
    const int hcell = hy * S + hx;
    ep[lane] = lane == fidx ? 1.0f : 0.0f;
    ep[C + lane] = lane == hcell ? 1.0f : 0.0f;
    ep[2 * C + lane] = (float)v0;
    if (lane + 64 < C) {
        ep[lane + 64] = lane + 64 == fidx ? 1.0f : 0.0f;
        ep[C + lane + 64] = lane + 64 == hcell ? 1.0f : 0.0f;
        ep[2 * C + lane + 64] = (float)v1;
    }
}

// `envs` from the mirror (LAZY form): every env whose record describes it (RES_ACT) is written whole; the others are the
// ones the one-env-per-wave code steps on `envs` itself, which is current for them.
template <int EPW>
__global__ __launch_bounds__(256) void lane_resident_flush_kernel(ResidentArgs a)
{
    constexpr int S = 9, N4 = EPW * LR_C3 / 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char res_lds[];
    const StepArgs &p = a.p;
    const int wave = uniform((int)(threadIdx.x >> 6)), wpb = (int)(blockDim.x >> 6), lane = (int)(threadIdx.x & 63u);
    const long long env0 = ((long long)blockIdx.x * wpb + wave) * EPW;
    if (env0 >= p.N) return;
    const int nenv = (int)min((long long)EPW, p.N - env0);
    u32 *slab = (u32 *)(res_lds + wave * (EPW * LR_C3 + 16)); // the block's state, one byte per float
    unsigned char *sb8 = (unsigned char *)slab;
    uint4 r0 = make_uint4(0, 0, 0, 0), r1 = r0;
    if (lane < nenv) {
        r0 = a.res[env0 + lane];
        r1 = a.res[p.N + env0 + lane];
    }
    for (int i = lane; i < N4 + 4; i += 64) slab[i] = 0;
    wave_lds_sync();
    const bool actf = lane < nenv && (r1.z & RES_ACT) != 0;
    if (actf) {
        unsigned char *my = sb8 + lane * LR_C3;
        const int L = (int)((r1.y >> 14) & 127u), food = (int)((r1.y >> 23) & 127u) - 1;
        int y = (int)(r1.w & 15u), x = (int)((r1.w >> 4) & 15u);
        if (food >= 0) my[(food >> 3) * S + (food & 7)] = 1;
        my[LR_C + y * S + x] = 1;
        u32 w0 = r0.z, w1 = r0.w, w2 = r1.x;
        for (int v = L; v >= 1; --v) {
            my[2 * LR_C + y * S + x] += (unsigned char)v;
            const int m = (int)(w0 & 3u);
            y -= lr_dy(m); x -= lr_dx(m);
            w0 = (w0 >> 2) | (w1 << 30); w1 = (w1 >> 2) | (w2 << 30); w2 >>= 2;
        }
    }
    const u64 am = ballot(actf);
    wave_lds_sync();
    if (nenv == EPW && am == (EPW == 64 ? ~0ull : (1ull << EPW) - 1ull) && (((size_t)p.envs) & 15u) == 0) {
        float4 *out4 = (float4 *)(p.envs + env0 * LR_C3);
#pragma unroll 4
        for (int g = lane; g < N4; g += 64) {
            const u32 b = slab[g];
            out4[g] = make_float4((float)(b & 0xffu), (float)((b >> 8) & 0xffu), (float)((b >> 16) & 0xffu), (float)(b >> 24));
        }
    } else {
        float *sb = p.envs + env0 * LR_C3;
        for (int i = lane; i < nenv * LR_C3; i += 64) {
            const int e = i / LR_C3;
            if ((am >> e) & 1ull) sb[i] = (float)sb8[i];
        }
    }
}

// per-wave LDS (bytes) of the step kernel
struct ResLds {
    static constexpr int IO = 0;            // uint4 [64]     crop records of the (which, env) pairs
    static constexpr int BITS = IO + 1024;  // u32 [2][152]   flat bit strings (interleaved as in lane_rollout.hpp)
    static constexpr int SCR = BITS + 1216; // fused_step_env's class map (96 bytes at S = 9)
    static constexpr int BYTES = SCR + 128;
};
// per-wave LDS of lane_resident_step_kernel by observation: the grid / crop modes keep flat bit strings, 'raw' a byte slab
// behind ResLds::BYTES
constexpr int res_wave_bytes(int OBSK)
{
    return ResLds::BYTES + (OBSK == LR_OBS_RAW ? LR_RAW_SLAB : (OBSK == LR_OBS_GRID1 || OBSK == LR_OBS_GRID3 || OBSK == LR_OBS_CROP3) ? LR_GRID_BITS : 0);
}

// EPW envs per wave; NW = 2: crops of `obs` and of `obs_after` (EPW * 2 <= 64 pair lanes), NW = 1: `obs` only; LAZY: `envs`
// is left alone
template <int EPW, int NW, int OBSK, bool LAZY>
__global__ __launch_bounds__(256) void lane_resident_step_kernel(ResidentArgs a)
{
    static_assert(EPW == 16 || EPW == 32 || EPW == 64, "envs per wave");
    static_assert(EPW * NW <= 64 && (NW == 1 || NW == 2), "pair lanes");
    static_assert(OBSK == WURM_OBS_PARTIAL || OBSK == LR_OBS_GENERIC || OBSK == LR_OBS_GRID1 || OBSK == LR_OBS_GRID3 ||
                  OBSK == LR_OBS_CROP3 || OBSK == LR_OBS_RAW || (OBSK == WURM_OBS_NONE && NW == 1),
                  "partial_2, one_channel / default / partial_3 through bit planes, raw through bytes, any other mode at run time (lr_obs_value), or none");
    constexpr bool GRID = OBSK == LR_OBS_GRID1 || OBSK == LR_OBS_GRID3 || OBSK == LR_OBS_CROP3;
    constexpr bool RAW = OBSK == LR_OBS_RAW;
    constexpr bool GTAB = lr_grid_tables(OBSK);
    constexpr int GE = OBSK == LR_OBS_GRID1 ? LR_C : OBSK == LR_OBS_CROP3 ? LR_E3 : LR_C3; // floats per env of such a mode
    constexpr int S = 9, C = LR_C, C3 = LR_C3;
    constexpr int LOG_EPW = EPW == 16 ? 4 : EPW == 32 ? 5 : 6;
    constexpr int NP = EPW * NW;                 // pair lanes
    constexpr int GS = EPW * LR_E / 4;           // 16-byte groups of one observation of the wave's envs
    constexpr int NG = NW * GS, ITER = (NG + 63) / 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char res_lds[];
    const StepArgs &p = a.p;
    const int wave = uniform((int)(threadIdx.x >> 6)), wpb = (int)(blockDim.x >> 6), lane = (int)(threadIdx.x & 63u);

    float4 *tab = (float4 *)res_lds;
    u64 *wint = (u64 *)(res_lds + 4096);
    // (generic observations: float -> channel / row / column, in place of tab; grid modes: behind their two tables)
    unsigned short *lut = (unsigned short *)(res_lds + (GTAB ? 8192 : 0));
    float4 *tabB = (float4 *)(res_lds + 4096);
    if (OBSK == WURM_OBS_PARTIAL) {
        lr_build_tables(tab, wint);
        __syncthreads();
    } else if (OBSK == LR_OBS_GENERIC || GRID) {
        lr_build_lut(lut, p.obs_mode, p.obs_n, (int)p.obs_elems);
        if (GRID) lr_build_grid_tables<OBSK>(tab, tabB);
        if (OBSK == LR_OBS_CROP3) lr_build_wint7(wint); // (in the place of 'one_channel's second table)
        __syncthreads();
    }

    const long long env0 = (xcd_block(blockIdx.x, gridDim.x) * wpb + wave) * EPW;
    if (env0 >= p.N) return;
    unsigned char *lds = res_lds + (GTAB ? LR_TAB_GRID : LR_TAB) + wave * res_wave_bytes(OBSK);
    unsigned char *gb = lds + ResLds::BYTES;      // ('raw': the pairs' observations, one byte per float)
    const int nenv = (int)min((long long)EPW, p.N - env0);
    const bool mine = lane < nenv;                // env lanes: lane e owns env0 + e
    const long long env = env0 + lane;
    const u64 env_id = (u64)(p.env_offset + env);

    // ---- env lanes: the record, the action, the postponed reset
    WURM_TL_DECL;
    WURM_TL(0); // entry (tables built)
    uint4 r0 = make_uint4(0, 0, 0, 0), r1 = r0;
    long long a_in = 0;
    bool pre = false;
    if (mine) {
        r0 = a.res[env];
        r1 = a.res[p.N + env];
        a_in = load_action(p.actions, p.act_dtype, env);
        pre = p.done_in != nullptr && p.done_in[env] != 0;
    }
    u64 occ = (u64)r0.x | ((u64)r0.y << 32);
    u32 q0 = r0.z, q1 = r0.w, q2 = r1.x;
    int c = (int)(r1.y & 127u), tc = (int)((r1.y >> 7) & 127u), L = (int)((r1.y >> 14) & 127u), o = (int)((r1.y >> 21) & 3u);
    int food = (int)((r1.y >> 23) & 127u) - 1;
    bool act = mine && (r1.z & (RES_ACT | RES_TERMINAL)) == RES_ACT;
    int c3 = 0, c2 = 0, c1 = 0; // a rebuilt env: row-major cells of its head, middle and tail segments
    if (pre) {                  // reset_kernel with call = pre_call (single_snake.py:322-387): the env is rebuilt, whatever it held
        const S9Reset r = s9_reset_draw(p.seed, p.pre_call, env_id);
        const int hc = r.b & 127, sc = (r.b >> 7) & 127;
        tc = (r.b >> 14) & 127;
        c = hc; o = r.a & 3; L = 3;
        food = r.a >> 2;
        occ = (1ull << hc) | (1ull << sc) | (1ull << tc);
        q0 = (u32)((o ^ 2) * 5); q1 = 0; q2 = 0;
        c3 = (hc >> 3) * S + (hc & 7); c2 = (sc >> 3) * S + (sc & 7); c1 = (tc >> 3) * S + (tc & 7);
        act = true;
    }

    WURM_TL(1); // record, action and reset flag loaded (first use), postponed reset drawn
    // ---- the transition (single_snake.py:197-304) on the env lanes
    const int c_old = c, L_old = L;
    const u32 w0_old = q0, w1_old = q1, w2_old = q2;
    uint4 rec;
    {
        const u32 a_small = (a_in >= 0 && a_in < 4) ? (u32)a_in : 7u;
        const u32 a_mod = (a_in >= 0 && a_in < 4) ? (u32)a_in : ((u32)(int)(a_in % 4) & 7u);
        u32 fword = 0;
        if (act) fword = rng_words(p.seed, p.call, env_id, RNG_FOOD, 0).w[0];
        if (RAW) { // the pairs' byte slab: zero, then the env lanes leave the stepped state in it
            for (int i = lane; i < LR_RAW_SLAB / 16; i += 64) ((uint4 *)gb)[i] = make_uint4(0, 0, 0, 0);
            wave_lds_sync();
        }
        rec = lr_transition<false, false, RAW>(occ, q0, q1, q2, c, tc, L, o, food, act, make_uint4(a_small | (a_mod << 3), 0u, fword, 0u),
                                               gb + lane * LR_C3);
    }
    const u32 rz = rec.z, rw = rec.w;
    const bool eat = (rw & 0x100u) != 0, selfc = (rw & 0x200u) != 0, edgec = (rw & 0x400u) != 0;
    const bool fin = selfc || edgec;
    const int a_out = (int)(signed char)(rz >> 8), ai = a_out & 3;
    const int hy0 = c_old >> 3, hx0 = c_old & 7;
    const int ny = hy0 + lr_dy(ai), nx = hx0 + lr_dx(ai);     // the head after the move, also when it is on the ring
    const int nh = ny * S + nx, hc0 = hy0 * S + hx0;
    const int fidx = food >= 0 ? (food >> 3) * S + (food & 7) : -1;
    const int dec = eat ? 0 : 1, grow = L_old + (eat ? 1 : 0);

    WURM_TL(2); // transition done
    if (act) { // per-env outputs
        store_action(p.actions, p.act_dtype, env, (long long)a_out);
        p.selfc[env] = (uint8_t)selfc;
        p.reward[env] = eat ? 1.0f : 0.0f;
        p.done[env] = (uint8_t)fin;
        p.edgec[env] = (uint8_t)edgec;
        if (p.done_copy) p.done_copy[env] = (uint8_t)fin;
    }
    if (mine) { // the mirror
        const u32 pk = (u32)(c & 127) | ((u32)(tc & 127) << 7) | ((u32)(L & 127) << 14) | ((u32)(o & 3) << 21) |
                       ((u32)((food + 1) & 127) << 23);
        a.res[env] = make_uint4((u32)occ, (u32)(occ >> 32), q0, q1);
        a.res[p.N + env] = make_uint4(q2, pk, act ? (RES_ACT | (fin ? RES_TERMINAL : 0u)) : (r1.z & ~RES_ACT),
                                      act ? ((u32)ny | ((u32)nx << 4)) : r1.w);
    }

    if (mine && a.check_mask != nullptr) // wurm_single_check's mask of the stepped state: a live env of the domain is a well-formed snake
        a.check_mask[env] = act && !fin ? ((L < 3 ? WURM_CHK_MIN_LENGTH : 0u) | (food < 0 ? WURM_CHK_ONE_FOOD : 0u)) : WURM_CHK_NOT_COMPUTED;

    WURM_TL(3); // outputs and mirror stored
    // ---- `envs`: the elements that change (single_snake.py:246-282), straight from the env lanes
    if (!LAZY) {
        char *sb = (char *)(p.envs + env0 * C3);
        const unsigned so = 4u * (unsigned)(lane * C3);
        auto put = [&](int elem, float v) { *(float *)(sb + (so + 4u * (unsigned)elem)) = v; };
        const bool stepped = act && !pre;
        // body: every cell decays unless food was eaten (walk the old queue from the old head); the cell the head moves
        // onto grows — on top of what it held if the snake ran into itself
        int under = 0;
        {
            const bool walk = stepped && !eat;
            int v = L_old, code = c_old;
            u32 w0 = w0_old, w1 = w1_old, w2 = w2_old;
            while (ballot(walk && v >= 1) != 0) {
                if (walk && v >= 1) {
                    int nv = v - 1;
                    if (code == c) { under = v; nv += grow; }
                    put(2 * C + (code >> 3) * S + (code & 7), (float)nv);
                    code -= lr_dcode((int)(w0 & 3u));
                    w0 = (w0 >> 2) | (w1 << 30); w1 = (w1 >> 2) | (w2 << 30); w2 >>= 2;
                    --v;
                }
            }
        }
        if (stepped) {
            if (under == 0) put(2 * C + nh, (float)grow);
            put(C + hc0, 0.0f);
            put(C + nh, 1.0f);
            if (eat) {
                put(nh, 0.0f);
                if (fidx >= 0) put(fidx, 1.0f);
            }
        }
        // a rebuilt env is stored whole (its old contents are unrelated): cooperative, one env at a time
        for (u64 m = ballot(pre); m != 0; m &= m - 1) {
            const int src = first_bit(m);
            const int s3 = lane_value(c3, src), s2 = lane_value(c2, src), s1 = lane_value(c1, src);
            const int s_nh = lane_value(nh, src), s_dec = lane_value(dec, src), s_grow = lane_value(grow, src);
            const int s_f = lane_value(fidx, src);
            float *ep = p.envs + (env0 + src) * C3;
            constexpr float rcpC = 1.0f / (float)C;
            for (int i = lane; i < C3; i += 64) {
                const int ch = div_size(i, rcpC), cell = i - ch * C;
                float v;
                if (ch == 0) v = cell == s_f ? 1.0f : 0.0f;
                else if (ch == 1) v = cell == s_nh ? 1.0f : 0.0f;
                else v = (float)((cell == s3 ? 3 - s_dec : cell == s2 ? 2 - s_dec : cell == s1 ? 1 - s_dec : 0) +
                                 (cell == s_nh ? s_grow : 0));
                ep[i] = v;
            }
        }
    }

    WURM_TL(4); // state elements stored
    // ---- observations: crop records of the (which, env) pairs — which = 0: the stepped state, 1: that state once a
    // finished env is rebuilt with call + 1 (not stored: the next launch's postponed reset recreates it)
    const u64 odd = ballot(mine && !act);
    if (OBSK == WURM_OBS_PARTIAL) {
        uint4 *io = (uint4 *)(lds + ResLds::IO);
        u32 *bits = (u32 *)(lds + ResLds::BITS);
        // record: occupancy lo, hi, head row | head column << 4 | valid << 8, food code + 1
        uint4 cr = make_uint4(rec.x, rec.y, (u32)ny | ((u32)nx << 4) | (act ? 0x100u : 0u), rw & 127u);
        if (NW == 2) {
            uint4 cr2 = cr;
            if (act && fin) {
                const S9Reset r = s9_reset_draw(p.seed, p.call + 1ull, env_id);
                const int hc = r.b & 127, sc = (r.b >> 7) & 127, t2 = (r.b >> 14) & 127;
                const u64 oc2 = (1ull << hc) | (1ull << sc) | (1ull << t2);
                cr2 = make_uint4((u32)oc2, (u32)(oc2 >> 32), (u32)(hc >> 3) | ((u32)(hc & 7) << 4) | 0x100u, (u32)(r.a >> 2) + 1u);
            }
            if (lane < EPW) { io[lane] = cr; io[EPW + lane] = cr2; }
        }
#pragma unroll
        for (int i = lane; i < 2 * 152; i += 64) bits[i] = 0;
        wave_lds_sync();
        if (NW == 2) cr = io[lane]; // pair lane (which, env) = (lane >> LOG_EPW, lane & (EPW - 1))
        const int pe = lane & (EPW - 1);
        if (lane < NP && pe < nenv && (cr.z & 0x100u) != 0) {
            // crop (single_snake.py:166-193): a window cell that is off the grid or on the ring is (0,0,0); food (1,0,0),
            // head (0,1,0), body (0,127/255,0), background (1,1,1)
            const int hy = (int)(cr.z & 15u), hx = (int)((cr.z >> 4) & 15u);
            const int sh = 8 * hy + hx - 18;              // window bit 8 i + j <-> code sh + 8 i + j
            const u64 oc = (u64)cr.x | ((u64)cr.y << 32);
            const u64 V = sh >= 0 ? oc >> sh : oc << (-sh);
            const u64 W = wint[hy * S + hx];
            const int fpos_w = (int)cr.w - 1 - sh;
            const u64 F = cr.w != 0 && (unsigned)fpos_w < 40u ? (1ull << fpos_w) & W : 0ull;
            const u64 R = W & ~V;                         // free or food: red
            const u64 B = R & ~F;                         // free: blue (and green)
            const u64 CENTRE = 1ull << 18;
            const u64 G1 = B | (W & CENTRE);              // green 1: free, or the head inside the ring
            const u64 GH = V & W & ~CENTRE;               // green 127/255: body
            const u32 r25 = lr_compact((u32)R, (u32)(R >> 32)), b25 = lr_compact((u32)B, (u32)(B >> 32));
            const u32 g25 = lr_compact((u32)G1, (u32)(G1 >> 32)), h25 = lr_compact((u32)GH, (u32)(GH >> 32));
            const u32 d0 = r25 | (g25 << 25), d1 = (g25 >> 7) | (b25 << 18), d2 = b25 >> 14;
            const u32 e0 = h25 << 25, e1 = h25 >> 7;
            const int bit = LR_E * lane, w0 = bit >> 5, sb = bit & 31;
            const u64 x01 = ((u64)d0 << sb), x12 = (((u64)d2 << 32) | d1) << sb;
            const u64 y01 = ((u64)e0 << sb), y1 = ((u64)e1 << sb);
            u32 *P = bits + 2 * w0;
            atomicOr(&P[0], (u32)x01);
            atomicOr(&P[2], (u32)(x01 >> 32) | (u32)x12);
            atomicOr(&P[4], (u32)(x12 >> 32));
            atomicOr(&P[6], (u32)(((u64)d2 << sb) >> 32));
            atomicOr(&P[1], (u32)y01);
            atomicOr(&P[3], (u32)(y01 >> 32) | (u32)y1);
            atomicOr(&P[5], (u32)(y1 >> 32));
        }
        wave_lds_sync();
        WURM_TL(5); // bit strings complete
        float *ob0 = p.obs + env0 * LR_E, *ob1 = NW == 2 ? p.obs_after + env0 * LR_E : nullptr;
        if (nenv == EPW) {
            const int shn = (lane & 7) * 4;
            const uint2 *b2 = (const uint2 *)bits + (lane >> 3);
#pragma unroll
            for (int i = 0; i < ITER; ++i) {
                const uint2 w = b2[8 * i];
                const float4 v = tab[((w.x >> shn) & 15u) | (((w.y >> shn) & 15u) << 4)];
                const int j = 64 * i + lane;               // group j = floats 4 j .. 4 j + 3 of the pairs' crops
                const bool second = NW == 2 && j >= GS;
                char *dst = (char *)(second ? ob1 : ob0) + 16u * (unsigned)(second ? j - GS : j);
                if (64 * i + 63 < NG || j < NG) *(float4 *)dst = v;
            }
        } else { // the ragged last wave: float by float
            for (int f = lane; f < NP * LR_E; f += 64) {
                const int pr = f / LR_E, k2 = f - pr * LR_E, w = pr >> LOG_EPW, e = pr & (EPW - 1);
                if (e < nenv) {
                    const u32 w1 = bits[2 * (f >> 5)], wh = bits[2 * (f >> 5) + 1];
                    const float v = ((w1 >> (f & 31)) & 1u) ? 1.0f : ((wh >> (f & 31)) & 1u) ? 127.0f / 255.0f : 0.0f;
                    (w ? ob1 : ob0)[e * LR_E + k2] = v;
                }
            }
        }
    }

    if (OBSK == LR_OBS_GENERIC || GRID) {
        // every other observation: records of the (which, env) pairs, then — 'one_channel' / 'default' — bit planes, nibbles
        // and 16-byte stores (lane_rollout.hpp: lr_grid_planes), or float by float (lr_obs_value)
        uint4 *io = (uint4 *)(lds + ResLds::IO);
        u32 *gbits = (u32 *)(lds + ResLds::BYTES);
        uint4 cr = make_uint4(rec.x, rec.y, (u32)ny | ((u32)nx << 4) | (act ? 0x100u : 0u), rw & 127u);
        uint4 cr2 = cr;
        if (NW == 2 && act && fin) {
            const S9Reset r = s9_reset_draw(p.seed, p.call + 1ull, env_id);
            const int hc = r.b & 127, sc = (r.b >> 7) & 127, t2 = (r.b >> 14) & 127;
            const u64 oc2 = (1ull << hc) | (1ull << sc) | (1ull << t2);
            cr2 = make_uint4((u32)oc2, (u32)(oc2 >> 32), (u32)(hc >> 3) | ((u32)(hc & 7) << 4) | 0x100u, (u32)(r.a >> 2) + 1u);
        }
        if (lane < EPW) { io[lane] = cr; if (NW == 2) io[EPW + lane] = cr2; }
        if (GRID && nenv == EPW) {
            constexpr int GPL = OBSK == LR_OBS_GRID1 ? 4 : 2;
            constexpr int NWORDS = GPL * ((NP * GE + 31) / 32 + 4);
            static_assert(NWORDS * 4 + 16 <= LR_GRID_BITS, "flat bit strings of a grid mode");
            for (int i = lane; i < (NWORDS + 3) / 4; i += 64) ((uint4 *)gbits)[i] = make_uint4(0, 0, 0, 0);
            wave_lds_sync();
            if (lane < NP) {
                const uint4 q = io[lane]; // pair lane (which, env) = (lane >> LOG_EPW, lane & (EPW - 1))
                if (q.z & 0x100u) {
                    if constexpr (OBSK == LR_OBS_CROP3)
                        lr_crop3_planes(gbits, wint, lane, (u64)q.x | ((u64)q.y << 32), (int)(q.z & 15u), (int)((q.z >> 4) & 15u), (int)q.w - 1);
                    else
                        lr_grid_planes<OBSK>(gbits, lane, (u64)q.x | ((u64)q.y << 32), (int)(q.z & 15u), (int)((q.z >> 4) & 15u), (int)q.w - 1);
                }
            }
            wave_lds_sync();
            constexpr int GSG = EPW * GE / 4;   // 16-byte groups of one observation of the wave's envs
            // (float4 pointers indexed by group — through `char * + 16 j` the NW = 1 instantiations of 'default', 'raw' and
            // 'partial_3' came out with every 16-byte store split into four dword stores: 4.0 store instructions per env
            // instead of 1.1 and 25.7 us per call of 65 536 envs instead of 15 — tools/percall_size_only.py under --pmc)
            float4 *ob0 = (float4 *)(p.obs + env0 * GE), *ob1 = NW == 2 ? (float4 *)(p.obs_after + env0 * GE) : nullptr;
#pragma unroll 4
            for (int j = lane; j < NW * GSG; j += 64) {
                const float4 v = lr_grid_group<OBSK>(gbits, tab, tabB, j);
                const bool second = NW == 2 && j >= GSG;
                (second ? ob1 : ob0)[second ? j - GSG : j] = v;
            }
            wave_lds_sync();
        } else {
        wave_lds_sync();
        const int E = (int)p.obs_elems, mode = p.obs_mode, n = p.obs_n, row = nenv * E;
        const float rcpE = 1.0f / (float)E;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            float *orow = (w ? p.obs_after : p.obs) + env0 * E;
#pragma unroll 2
            for (int f = lane; f < row; f += 64) {
                const int e = div_size(f, rcpE), r = f - e * E;
                const uint4 q = io[w * EPW + e];
                const u32 l = lut[r];
                if (q.z & 0x100u) // (else an env outside the domain: fused_step_env below writes it)
                    orow[f] = lr_obs_value((u64)q.x | ((u64)q.y << 32), (int)(q.z & 15u), (int)((q.z >> 4) & 15u), (int)q.w - 1, l, mode, n);
            }
        }
        wave_lds_sync();
        }
    }

    if (RAW) {
        // 'raw' (round 5): row `lane` of the slab holds the stepped state (lr_transition); for `obs_after` row EPW + lane
        // holds the state a finished env is rebuilt to with call + 1 (three segments, :372-376), else the stepped state again
        if (NW == 2 && act) {
            unsigned char *r2 = gb + (EPW + lane) * LR_C3;
            if (fin) {
                const S9Reset r = s9_reset_draw(p.seed, p.call + 1ull, env_id);
                const int hc = r.b & 127, sc = (r.b >> 7) & 127, t2 = (r.b >> 14) & 127, f2 = r.a >> 2;
                if (f2 >= 0) r2[9 * (f2 >> 3) + (f2 & 7)] = 1;
                r2[LR_C + 9 * (hc >> 3) + (hc & 7)] = 1;
                r2[2 * LR_C + 9 * (hc >> 3) + (hc & 7)] = 3;
                r2[2 * LR_C + 9 * (sc >> 3) + (sc & 7)] = 2;
                r2[2 * LR_C + 9 * (t2 >> 3) + (t2 & 7)] = 1;
            } else {
                lr_raw_bytes(r2, food, ny, nx, c_old, (q0 >> 2) | (q1 << 30), (q1 >> 2) | (q2 << 30), q2 >> 2, L);
            }
        }
        wave_lds_sync();
        float *ob0 = p.obs + env0 * LR_C3, *ob1 = NW == 2 ? p.obs_after + env0 * LR_C3 : nullptr;
        if (nenv == EPW && odd == 0) {
            constexpr int GSG = EPW * LR_C3 / 4;   // 16-byte groups of one observation of the wave's envs
            const u32 *g4 = (const u32 *)gb;
            float4 *o40 = (float4 *)ob0, *o41 = (float4 *)ob1;
#pragma unroll 4
            for (int j = lane; j < NW * GSG; j += 64) {
                const u32 b = g4[j];
                const bool second = NW == 2 && j >= GSG;
                (second ? o41 : o40)[second ? j - GSG : j] =
                    make_float4((float)(b & 0xffu), (float)((b >> 8) & 0xffu), (float)((b >> 16) & 0xffu), (float)(b >> 24));
            }
        } else { // the ragged last wave, or one with an env outside the domain (fused_step_env below writes that env's rows)
            const u64 ok = ballot(act);
            for (int f = lane; f < NP * LR_C3; f += 64) {
                const int pr = f / LR_C3, k2 = f - pr * LR_C3, w = pr >> LOG_EPW, e = pr & (EPW - 1);
                if (e < nenv && ((ok >> e) & 1ull)) (w ? ob1 : ob0)[e * LR_C3 + k2] = (float)gb[f];
            }
        }
        wave_lds_sync();
    }

    WURM_TL(6); // crops issued; WURM_TL_STORE: drained
    if (OBSK == WURM_OBS_PARTIAL) WURM_TL_STORE(p.obs + env0 * LR_E, lane);
    // ---- envs outside the domain: the one-env-per-wave code reads and writes their state, outputs and crops itself
    // (nothing above touched them except crop bytes, which it overwrites)
    if (odd != 0) {
        if (LAZY) { // a finished env that is stepped again without its reset: `envs` has to show its last state first
            for (u64 m = ballot(mine && (r1.z & (RES_ACT | RES_TERMINAL)) == (RES_ACT | RES_TERMINAL) && !pre); m != 0; m &= m - 1) {
                const int src = first_bit(m);
                const int fd = (int)((lane_value((int)r1.y, src) >> 23) & 127) - 1;
                res_materialise_env(p.envs + (env0 + src) * C3, lane, (u32)lane_value((int)r0.z, src),
                                    (u32)lane_value((int)r0.w, src), (u32)lane_value((int)r1.x, src),
                                    (lane_value((int)r1.y, src) >> 14) & 127, lane_value((int)r1.w, src) & 15,
                                    (lane_value((int)r1.w, src) >> 4) & 15, fd >= 0 ? (fd >> 3) * S + (fd & 7) : -1);
            }
        }
        __threadfence();
        wave_lds_sync();
        for (u64 m = odd; m != 0; m &= m - 1)
            fused_step_env<2, true>(p, env0 + first_bit(m), (signed char *)(lds + ResLds::SCR));
    }
}

// ------------------------------------------------------------------------------------------------ host side

// the shapes the resident step serves (the caller opts in per call by passing wurm_single_call.resident)
bool lane_resident_shape(int S, int obs_mode, int obs_n)
{
    // every observation but crops of 9 x 9 and more (lane_rollout_eligible says why)
    return S == 9 && !(obs_mode == WURM_OBS_PARTIAL && (obs_n < 0 || obs_n > 3));
}

bool lane_resident_eligible(const StepArgs &p)
{
    if (!lane_resident_shape(p.S, p.obs_mode, p.obs_n)) return false;
    if (p.inject_food || p.inject_reset || p.inject_pre_reset || p.post_reset || p.only_flagged) return false;
    if (p.N * (long long)RES_BYTES >= (1ll << 40)) return false;
    return true;
}

hipError_t launch_lane_resident_flush(const StepArgs &p, void *resident, hipStream_t stream)
{
    ResidentArgs a;
    a.p = p;
    a.res = (uint4 *)resident;
    a.check_mask = nullptr;
    constexpr int EPW = 16;
    const long long waves = (p.N + EPW - 1) / EPW;
    const int wpb = waves >= 1024 ? 4 : 1;
    dim3 block(64 * wpb), grid((unsigned)((waves + wpb - 1) / wpb));
    (void)hipGetLastError();
    WURM_LAUNCH(lane_resident_flush_kernel<EPW>, grid, block, (size_t)((EPW * LR_C3 + 16) * wpb), stream, a);
    return hipGetLastError();
}

template <bool LAZY>
static hipError_t launch_lane_resident_form(const StepArgs &p, void *resident, bool valid, uint32_t *check_mask, hipStream_t stream)
{
    ResidentArgs a;
    a.p = p;
    a.res = (uint4 *)resident;
    a.check_mask = check_mask;
    (void)hipGetLastError();
    if (!valid) {
        constexpr int EPW = 16;
        const long long waves = (p.N + EPW - 1) / EPW;
        const int wpb = waves >= 1024 ? 4 : 1;
        dim3 block(64 * wpb), grid((unsigned)((waves + wpb - 1) / wpb));
        WURM_LAUNCH(lane_resident_build_kernel<EPW>, grid, block, (size_t)(LaneRollLds<EPW>::BYTES * wpb), stream, a);
        hipError_t err = hipGetLastError();
        if (err != hipSuccess) return err;
    }
    const bool crops = p.obs_mode == WURM_OBS_PARTIAL && p.obs_n == 2;
    const bool grid1 = p.obs_mode == WURM_OBS_ONE_CHANNEL, grid3 = p.obs_mode == WURM_OBS_DEFAULT;
    const bool crop3 = p.obs_mode == WURM_OBS_PARTIAL && p.obs_n == 3, raw = p.obs_mode == WURM_OBS_RAW;
    const bool generic = p.obs_mode != WURM_OBS_NONE && !crops && !grid1 && !grid3 && !crop3 && !raw;
    const int nw = (p.obs_mode != WURM_OBS_NONE && p.obs_after != nullptr) ? 2 : 1;
    // envs per wave (automatic unless the option WURM_RESIDENT_EPW forces it: tests and the tuning sweep)
    int epw = (int)opt.resident_epw;
    // measured (rocprofv3, us per launch at 16 / 32 / 64 envs per wave): 65 536 envs 8.5 / 7.3 / 7.6 without and 11.3 / 11.1 / -
    // with the reset observation; 32 768 envs 6.3 / 5.9 / 6.3 and 7.6 / 8.4 / -
    if (epw != 16 && epw != 32 && epw != 64) epw = nw == 1 ? (p.N >= 16384 ? 32 : 16) : (p.N >= 49152 ? 32 : 16);
    if (epw * nw > 64) epw = 32;
    const int obsk = grid1 ? LR_OBS_GRID1 : grid3 ? LR_OBS_GRID3 : crop3 ? LR_OBS_CROP3 : raw ? LR_OBS_RAW : 0;
    auto go = [&](auto kernel, int e) {
        const long long waves = (p.N + e - 1) / e;
        const int wpb = waves >= 1024 ? 4 : 1;
        dim3 block(64 * wpb), grid((unsigned)((waves + wpb - 1) / wpb));
        const size_t lds_bytes = (size_t)((lr_grid_tables(obsk) ? LR_TAB_GRID : LR_TAB) + res_wave_bytes(obsk) * wpb);
        if (lds_bytes > 65536) (void)hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        WURM_LAUNCH(kernel, grid, block, lds_bytes, stream, a);
    };
    if (crop3) {
        if (nw == 2) {
            if (epw == 16) go(lane_resident_step_kernel<16, 2, LR_OBS_CROP3, LAZY>, 16);
            else go(lane_resident_step_kernel<32, 2, LR_OBS_CROP3, LAZY>, 32);
        } else if (epw == 16) go(lane_resident_step_kernel<16, 1, LR_OBS_CROP3, LAZY>, 16);
        else if (epw == 32) go(lane_resident_step_kernel<32, 1, LR_OBS_CROP3, LAZY>, 32);
        else go(lane_resident_step_kernel<64, 1, LR_OBS_CROP3, LAZY>, 64);
    } else if (raw) {
        if (nw == 2) {
            if (epw == 16) go(lane_resident_step_kernel<16, 2, LR_OBS_RAW, LAZY>, 16);
            else go(lane_resident_step_kernel<32, 2, LR_OBS_RAW, LAZY>, 32);
        } else if (epw == 16) go(lane_resident_step_kernel<16, 1, LR_OBS_RAW, LAZY>, 16);
        else if (epw == 32) go(lane_resident_step_kernel<32, 1, LR_OBS_RAW, LAZY>, 32);
        else go(lane_resident_step_kernel<64, 1, LR_OBS_RAW, LAZY>, 64);
    } else if (grid1) {
        if (nw == 2) {
            if (epw == 16) go(lane_resident_step_kernel<16, 2, LR_OBS_GRID1, LAZY>, 16);
            else go(lane_resident_step_kernel<32, 2, LR_OBS_GRID1, LAZY>, 32);
        } else if (epw == 16) go(lane_resident_step_kernel<16, 1, LR_OBS_GRID1, LAZY>, 16);
        else if (epw == 32) go(lane_resident_step_kernel<32, 1, LR_OBS_GRID1, LAZY>, 32);
        else go(lane_resident_step_kernel<64, 1, LR_OBS_GRID1, LAZY>, 64);
    } else if (grid3) {
        if (nw == 2) {
            if (epw == 16) go(lane_resident_step_kernel<16, 2, LR_OBS_GRID3, LAZY>, 16);
            else go(lane_resident_step_kernel<32, 2, LR_OBS_GRID3, LAZY>, 32);
        } else if (epw == 16) go(lane_resident_step_kernel<16, 1, LR_OBS_GRID3, LAZY>, 16);
        else if (epw == 32) go(lane_resident_step_kernel<32, 1, LR_OBS_GRID3, LAZY>, 32);
        else go(lane_resident_step_kernel<64, 1, LR_OBS_GRID3, LAZY>, 64);
    } else if (generic) {
        if (nw == 2) {
            if (epw == 16) go(lane_resident_step_kernel<16, 2, LR_OBS_GENERIC, LAZY>, 16);
            else go(lane_resident_step_kernel<32, 2, LR_OBS_GENERIC, LAZY>, 32);
        } else if (epw == 16) go(lane_resident_step_kernel<16, 1, LR_OBS_GENERIC, LAZY>, 16);
        else if (epw == 32) go(lane_resident_step_kernel<32, 1, LR_OBS_GENERIC, LAZY>, 32);
        else go(lane_resident_step_kernel<64, 1, LR_OBS_GENERIC, LAZY>, 64);
    } else if (!crops) {
        if (epw == 16) go(lane_resident_step_kernel<16, 1, WURM_OBS_NONE, LAZY>, 16);
        else if (epw == 32) go(lane_resident_step_kernel<32, 1, WURM_OBS_NONE, LAZY>, 32);
        else go(lane_resident_step_kernel<64, 1, WURM_OBS_NONE, LAZY>, 64);
    } else if (nw == 1) {
        if (epw == 16) go(lane_resident_step_kernel<16, 1, WURM_OBS_PARTIAL, LAZY>, 16);
        else if (epw == 32) go(lane_resident_step_kernel<32, 1, WURM_OBS_PARTIAL, LAZY>, 32);
        else go(lane_resident_step_kernel<64, 1, WURM_OBS_PARTIAL, LAZY>, 64);
    } else {
        if (epw == 16) go(lane_resident_step_kernel<16, 2, WURM_OBS_PARTIAL, LAZY>, 16);
        else go(lane_resident_step_kernel<32, 2, WURM_OBS_PARTIAL, LAZY>, 32);
    }
    return hipGetLastError();
}

hipError_t launch_lane_resident(const StepArgs &p, void *resident, bool valid, bool lazy, uint32_t *check_mask, hipStream_t stream)
{
    return lazy ? launch_lane_resident_form<true>(p, resident, valid, check_mask, stream)
                : launch_lane_resident_form<false>(p, resident, valid, check_mask, stream);
}

} // namespace wurm
