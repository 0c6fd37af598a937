// gridworld_lane.hip — the fused rollout (T iterations of `step(a[t]); reset(done)`) of SimpleGridworld for LARGE batches:
// ONE ENV PER LANE (round 5; VERDICT r04 "what's missing" #1: every !snake call went to the one-env-per-wave kernels).
//
// Replaces wurm/envs/simple_gridworld.py:135-202 (step), :111-133 (_observe), :225-268 (reset) for envs in this domain:
// exactly one agent cell and at most one food cell, both exactly 1.0, everything else exactly 0 — closed under step +
// reset.  The whole state of such an env is two cell indices, so a lane holds its env in four registers and a wave steps
// 64 consecutive envs with one instruction stream:
//   * the move, edge test and reward are a handful of integer operations per lane;
//   * the food respawn after an eaten food and the reset of a finished env are CLOSED FORMS of the same Philox draws the
//     one-env-per-wave kernels make (single_snake.hip: step_core / reset_core / add_food): the free interior cells are all
//     interior cells but the agent's, so "the K-th free cell in row-major order, K = mulhi(word, n_free)" is the interior
//     cell of index K, shifted by one behind the agent's — no mask, no ballot, no rank select;
//   * the observation of the wave's envs is one contiguous run (obs is (T, N, ...): consecutive envs are adjacent) that
//     is black / zero but for at most two floats per env: the run is kept as one BYTE per float in LDS (all zero; the env
//     lanes set their two bytes, and clear them after the step) and goes out as 4 bytes -> one 16-byte store, 1 KB per
//     wave instruction; the state is read once, coalesced, and actions are loaded four steps ahead.  (Runs beyond 16 KB —
//     large grids — are zero-filled and patched instead, 7 % slower.)  Bound: the store stream, 12 S^2 bytes per
//     env-step for 'default'.
// Any other env (hand-made states: several agents or foods, other values) is left untouched and marked
// done[0][env] = GRID_SKIPPED; the one-env-per-wave code rolls it out in a second launch (flagged_kernel), exactly as
// behind the clock-grid kernels (grid_rollout.hip).  RNG mode only (recorded outcomes take the generic kernel).
// Integer / index work: no MFMA.
#include "options.hpp"
#include "step_args.hpp"

namespace wurm {

namespace {

struct GridLaneGeo {
    int S, C, I;        // size, cells, interior side S - 2
    float rcpS, rcpI;
};

// the K-th interior cell in row-major order that is not `taken` (a cell index, or -1), K = mulhi(word, number of such
// cells); -1 if there is none — add_food of single_snake.hip for a state whose only occupied cell is `taken`
__device__ __forceinline__ int free_interior_cell(const GridLaneGeo &g, int taken, u32 word)
{
    const int S = g.S, I = g.I;
    int hi = 0x7fffffff; // interior index of the taken cell, if it is interior
    if (taken >= 0) {
        const int ty = div_size(taken, g.rcpS), tx = taken - ty * S;
        if (ty >= 1 && ty <= S - 2 && tx >= 1 && tx <= S - 2) hi = (ty - 1) * I + (tx - 1);
    }
    const int n_free = I * I - (hi != 0x7fffffff ? 1 : 0);
    if (n_free <= 0) return -1;
    int ii = (int)mulhi_range(word, (u32)n_free);
    ii += (int)(ii >= hi);
    const int fy = div_size(ii, g.rcpI), fx = ii - fy * I;
    return (fy + 1) * S + fx + 1;
}

// EPW consecutive envs per wave: 64 where a step writes a few bytes per env (positions / none), fewer for the image modes
// where the batch is small — there the wave's job is the store stream of its run; lanes >= EPW repeat the arithmetic of
// lane % EPW and only help with the loads and the fill
constexpr int GWL_MIRROR_HEADER = 16;      // bytes in front of the records: word 0 = envs the building launch could not describe
constexpr u32 GWL_REC_BAD = 0xfffefffeu;   // record of an env outside the domain (never in a mirror the library reports valid)
constexpr int GWL_CHUNK = 32; // steps whose actions are loaded together (2 bits each in a 64-bit word)

struct GridLaneScan {   // per wave: what the coalesced pass over the wave's run of the state found, per env
    int cnt[2][64];     // nonzero cells of the food / agent plane
    int pos[2][64];     // the last one seen
    int bad[64];        // a nonzero value other than 1.0
};

// One coalesced pass over the wave's run of (nv, 2, S, S) floats; the few nonzero floats report to LDS.  SIXTEEN 16-byte loads
// per lane in flight at a time (round 6): with one to four waves per SIMD the pass is a chain of memory round trips — at four
// in flight (round 5) the 41 KB of 64 envs of 9 x 9 took ten of them, 25-30 us of every launch: a 1-step launch of 65 536
// envs without observations took 36.8 us, a 16-step 'default' launch 55 us before its steady state (tools/gridworld_fixed_cost.py).
__device__ __forceinline__ void gwl_scan_run(GridLaneScan &sc, const float *env_run, int nv, int C, int lane)
{
    const int total = nv * 2 * C;
    const unsigned per_env = 2u * (unsigned)C;
    auto note = [&](int i, float v) {
        if (v != 0.0f) {
            const unsigned e = (unsigned)i / per_env, r = (unsigned)i - e * per_env;
            const int plane = r >= (unsigned)C ? 1 : 0;
            atomicAdd(&sc.cnt[plane][e], 1);
            sc.pos[plane][e] = (int)r - plane * C;
            if (v != 1.0f) sc.bad[e] = 1;
        }
    };
    if ((((unsigned long long)env_run) & 15ull) == 0) {
        const int n4 = total >> 2;
        const float4 *r4 = (const float4 *)env_run;
        constexpr int U = 16;
        for (int i0 = 0; i0 < n4; i0 += 64 * U) {
            float4 v[U];
#pragma unroll
            for (int j = 0; j < U; ++j) v[j] = r4[min(i0 + lane + 64 * j, n4 - 1)]; // (clamped, unconditional: all in flight together)
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const int i = i0 + lane + 64 * j;
                if (i < n4 && (v[j].x != 0.0f || v[j].y != 0.0f || v[j].z != 0.0f || v[j].w != 0.0f)) {
                    note(4 * i, v[j].x); note(4 * i + 1, v[j].y); note(4 * i + 2, v[j].z); note(4 * i + 3, v[j].w);
                }
            }
        }
        for (int i = (n4 << 2) + lane; i < total; i += 64) note(i, env_run[i]);
    } else {
        for (int i = lane; i < total; i += 64) note(i, env_run[i]);
    }
}

// The wave's run of one image observation from its bit string in LDS (`bits`: run bits, all zero on entry and on return):
// the env lanes set their two bits (of / oh: float offsets inside the env's elems, -1 = none; `base` = slot * elems), every
// lane turns nibbles into 16-byte stores, the env lanes clear their bits again.
__device__ __forceinline__ void gwl_emit_bits(float *blk, u32 *bits, int base, int run, int lane, int of, int oh)
{
    if (of >= 0) atomicOr(&bits[(base + of) >> 5], 1u << ((base + of) & 31));
    if (oh >= 0) atomicOr(&bits[(base + oh) >> 5], 1u << ((base + oh) & 31));
    wave_lds_sync();
    if ((((unsigned long long)blk) & 15ull) == 0) {
        const int n4 = run >> 2;
        float4 *b4 = (float4 *)blk;
#pragma unroll 4
        for (int i = lane; i < n4; i += 64) {
            const u32 n = bits[i >> 3] >> ((i & 7) * 4);
            b4[i] = make_float4((float)(n & 1u), (float)((n >> 1) & 1u), (float)((n >> 2) & 1u), (float)((n >> 3) & 1u));
        }
        for (int i = (n4 << 2) + lane; i < run; i += 64) blk[i] = (float)((bits[i >> 5] >> (i & 31)) & 1u);
    } else { // (a batch whose observation block does not start on a 16-byte boundary: ragged N)
        for (int i = lane; i < run; i += 64) blk[i] = (float)((bits[i >> 5] >> (i & 31)) & 1u);
    }
    wave_lds_sync();
    if (of >= 0) atomicAnd(&bits[(base + of) >> 5], ~(1u << ((base + of) & 31)));
    if (oh >= 0) atomicAnd(&bits[(base + oh) >> 5], ~(1u << ((base + oh) & 31)));
}

template <int OBS, int EPW>
__global__ __launch_bounds__(256) void gridworld_lane_rollout_kernel(StepArgs p)
{
    __shared__ GridLaneScan scans[4];
    extern __shared__ __attribute__((aligned(16))) unsigned char gwl_lds[];
    const int lane = (int)(threadIdx.x & 63u), wave = uniform((int)(threadIdx.x >> 6)), wpb = (int)(blockDim.x >> 6);
    const long long wblock = xcd_block(blockIdx.x, gridDim.x) * wpb + wave; // EPW consecutive envs per wave
    const long long env0 = wblock * EPW;
    if (env0 >= p.N) return;
    const int slot = lane & (EPW - 1);
    const long long env = env0 + slot;
    const bool present = env < p.N && lane < EPW;
    const int nv = (int)min((long long)EPW, p.N - env0); // envs of this wave
    GridLaneGeo g;
    g.S = p.S; g.C = p.S * p.S; g.I = p.S - 2;
    g.rcpS = 1.0f / (float)g.S;
    g.rcpI = g.I > 0 ? 1.0f / (float)g.I : 1.0f;
    const int S = g.S, C = g.C;

    // ---- the state: the caller's mirror when it describes it (wurm_grid_rollout_resident; the records of the per-call step:
    // gridworld_lane_step_kernel), else one coalesced pass over the wave's run of (nv, 2, S, S) floats; the few nonzero
    // floats report to LDS
    const bool mirrored = p.resident != nullptr, from_mirror = mirrored && p.resident_valid != 0;
    u32 *const recs = mirrored ? (u32 *)((unsigned char *)p.resident + GWL_MIRROR_HEADER) : nullptr;
    GridLaneScan &sc = scans[wave];
    float *const env_run = p.envs + env0 * 2 * C;
    int fc, hc;
    bool act;
    if (from_mirror) {
        const u32 rec = present ? recs[env] : 0xffffffffu;
        hc = (int)(short)(rec & 0xffffu);
        fc = (int)(short)(rec >> 16);
        act = present && rec != GWL_REC_BAD;
    } else {
        sc.cnt[0][lane] = 0; sc.cnt[1][lane] = 0; sc.pos[0][lane] = -1; sc.pos[1][lane] = -1; sc.bad[lane] = 0;
        wave_lds_sync();
        gwl_scan_run(sc, env_run, nv, C, lane);
        wave_lds_sync();
        const int nf = sc.cnt[0][slot], nh = sc.cnt[1][slot], bad = sc.bad[slot];
        fc = sc.pos[0][slot]; hc = sc.pos[1][slot];
        // (round 6: an env WITHOUT an agent is in the domain too — an agent that walked off the grid, :153-162 with an all-zero
        // head plane: nothing moves, no reward, done — so that the domain is closed under every step, not only step + reset)
        act = present && !bad && nh <= 1 && nf <= 1 && (hc != fc || hc < 0);
    }
    float *const envp = env_run + (long long)slot * 2 * C;
    if (present && !act) p.done[env] = GRID_SKIPPED; // rollout_kernel takes this env (second launch, only_flagged; T >= 1: eligible())
    if (mirrored && !from_mirror) { // a launch that builds the mirror counts what it cannot describe (the header's first word)
        const u64 odd = ballot(present && !act);
        if (odd != 0 && lane == 0) atomicAdd((int *)p.resident, popc64(odd));
    }
    const int hc0 = hc, fc0 = fc;
    const u64 env_id = (u64)(p.env_offset + env);
    const int start = p.start_y * S + p.start_x;

    const long long elems = p.obs_elems;                 // floats per env of an observation
    const int run = nv * (int)elems;                     // floats of the wave's run per step (<= 64 * 3 * 64 * 64)
    const bool obs16 = (((unsigned long long)p.obs) & 15ull) == 0;
    // image modes: the wave's run of one step as a bit string (p.lds_per_wave = its bytes, 0: it does not fit), zeroed once
    unsigned char *slab = nullptr;
    if ((OBS == WURM_OBS_DEFAULT || OBS == WURM_OBS_RAW) && p.lds_per_wave > 0) {
        slab = gwl_lds + wave * p.lds_per_wave;
        for (int i = lane; i < p.lds_per_wave / 16; i += 64) ((uint4 *)slab)[i] = make_uint4(0, 0, 0, 0);
        wave_lds_sync();
    }
    u64 call = p.call;
    // Actions: GWL_CHUNK steps at a time, all loads of a chunk in flight together, kept as 2-bit directions in two registers.
    // Round 5 loaded them four steps ahead inside the step loop: on gfx9 a wave waits for a LOAD only after every store it
    // issued before it has completed (one in-order counter), so each of those waits drained the wave's observation stores —
    // with two to four waves per SIMD nothing covers that (multi_snake.hip keeps its actions in LDS for the same reason).
    for (long long t0 = 0; t0 < p.T; t0 += GWL_CHUNK) {
        const int nt = (int)min((long long)GWL_CHUNK, p.T - t0);
        u64 dirs = 0;
        {
            long long av[GWL_CHUNK];
#pragma unroll
            for (int j = 0; j < GWL_CHUNK; ++j) // (index clamped: no load behind a branch, all of them in flight together)
                av[j] = act ? load_action(p.actions, p.act_dtype, (t0 + min(j, nt - 1)) * p.N + env) : 0;
#pragma unroll
            for (int j = 0; j < GWL_CHUNK; ++j) dirs |= (u64)(((av[j] % 4) + 4) % 4) << (2 * j);
        }
        for (int j = 0; j < nt; ++j) {
            const long long t = t0 + j;
            const int ai = (int)((dirs >> (2 * j)) & 3ull);
            // simple_gridworld.py:149-157 the agent moves by -TAP[a]; off the grid => it vanishes
            int newhead = -1, ny = -1, nx = -1;
            if (hc >= 0) {
                const int hy = div_size(hc, g.rcpS), hx = hc - hy * S;
                ny = hy - tap_y(ai);
                nx = hx - tap_x(ai);
                if (ny >= 0 && ny < S && nx >= 0 && nx < S) newhead = ny * S + nx;
            }
            const bool eat = act && newhead >= 0 && newhead == fc;      // :160 reward on the food square
            hc = newhead;
            if (ballot(eat) != 0) {                                     // :170-180 the food moves to a free interior cell
                const u32 word = rng_words(p.seed, call, env_id, RNG_FOOD, 0).w[0];
                if (eat) fc = free_interior_cell(g, hc, word);
            }
            const bool edge = !(newhead >= 0 && ny >= 1 && ny <= S - 2 && nx >= 1 && nx <= S - 2); // :186-193
            if (act) {
                const long long i = t * p.N + env;
                p.reward[i] = eat ? 1.0f : 0.0f;
                p.done[i] = (uint8_t)edge;
                p.edgec[i] = (uint8_t)edge;
            }
            // ---- the observation of the stepped, un-reset state (:202)
            if (OBS == WURM_OBS_DEFAULT || OBS == WURM_OBS_RAW) {
                float *const blk = p.obs + (t * p.N + env0) * elems; // (wave-uniform)
                // where the two floats per env that are not zero go: 'raw' is the state itself (food plane, agent plane);
                // 'default' (:88-109) a black image, food red, agent green, the border ring black
                int of = -1, oh = -1;
                if (act) {
                    if (OBS == WURM_OBS_RAW) {
                        of = fc;
                        oh = hc >= 0 ? C + hc : -1;
                    } else {
                        if (fc >= 0) {
                            const int y = div_size(fc, g.rcpS), x = fc - y * S;
                            if (y >= 1 && y <= S - 2 && x >= 1 && x <= S - 2) of = fc;
                        }
                        if (hc >= 0 && !edge) oh = C + hc;       // (edge <=> the agent is not on an interior cell)
                    }
                }
                if (slab != nullptr) {
                    // composed in LDS as one BIT per float (round 6; all zero but the bits of this step, cleared again below), then
                    // a nibble -> one 16-byte store: every byte of the run is written once, by whole 16-byte stores (a fill
                    // followed by scattered 4-byte stores cost 9-13 % of the launch, tools/gridworld_probe.py), and the wave
                    // keeps 1 KB of LDS instead of the 8-16 KB of round 5's byte slab, which held the kernel at 8 waves per CU
                    gwl_emit_bits(blk, (u32 *)slab, slot * (int)elems, run, lane, of, oh);
                } else {
                    // (a run that does not fit the LDS budget: large grids) zero fill of the run, then the two floats per env —
                    // stores of one wave to one address arrive in order; envs the generic kernel takes are written by it afterwards
                    if ((((unsigned long long)blk) & 15ull) == 0) {
                        const float4 z = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                        const int n4 = run >> 2;
                        float4 *b4 = (float4 *)blk;
#pragma unroll 4
                        for (int i = lane; i < n4; i += 64) b4[i] = z;
                        for (int i = (n4 << 2) + lane; i < run; i += 64) blk[i] = 0.0f;
                    } else {
                        for (int i = lane; i < run; i += 64) blk[i] = 0.0f;
                    }
                    float *const o = blk + (long long)slot * elems;
                    if (of >= 0) o[of] = 1.0f;
                    if (oh >= 0) o[oh] = 1.0f;
                }
            } else if (OBS == WURM_OBS_POSITIONS) { // argmax of the agent and food planes (0 if empty)
                if (act) {
                    const int h = hc < 0 ? 0 : hc, f = fc < 0 ? 0 : fc;
                    const int hy = div_size(h, g.rcpS), fy = div_size(f, g.rcpS);
                    float *const o = p.obs + (t * p.N + env) * 4;
                    const float4 v = make_float4((float)hy, (float)(h - hy * S), (float)fy, (float)(f - fy * S));
                    if (obs16) *(float4 *)o = v;
                    else { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
                }
            }
            // ---- reset(done) (:225-268): the agent back on the start location, one food on a free interior cell
            const bool fin = act && edge;
            if (ballot(fin) != 0) {
                const u32 word = rng_words(p.seed, call + 1ull, env_id, RNG_RESET, 0).w[3];
                if (fin) {
                    hc = start;
                    fc = free_interior_cell(g, start, word);
                }
            }
            call += 2;
        }
    }
    // ---- write the state back: the record, and — unless the mirror is lazy and was current — the cells of the planes that
    // held something at the start, then the cells that do now
    if (mirrored && present) recs[env] = act ? (((u32)hc & 0xffffu) | ((u32)fc << 16)) : GWL_REC_BAD;
    if (act && p.T > 0 && !(from_mirror && p.resident_lazy)) {
        if (fc0 >= 0 && fc0 != fc) envp[fc0] = 0.0f;
        if (hc0 >= 0 && hc0 != hc) envp[C + hc0] = 0.0f;
        if (fc >= 0 && fc != fc0) envp[fc] = 1.0f;
        if (hc >= 0 && hc != hc0) envp[C + hc] = 1.0f;
    }
}


// ---- the per-call step (`obs, r, d, info = env.step(a); env.reset(d)`, one launch per iteration) for large batches: the same
// lane-per-env transition under fused_step_kernel's contract without post_reset (single_snake.hip: fused_step_env) — an env
// flagged in p.done_in is rebuilt in front of the step with call = p.pre_call, the step uses p.call, p.obs is the stepped
// state's observation, p.obs_after (nullable) what reset(done) will return: the observation once finished envs are rebuilt
// with call + 1 — not stored, the next launch's postponed reset recreates it from the same counters.  Also serves the plain
// wurm_grid_step (no done_in, no obs_after).  Envs outside the domain: GRID_SKIPPED, the one-env-per-wave kernel in a
// second launch.
__device__ __forceinline__ bool gwl_interior(const GridLaneGeo &g, int c)
{
    if (c < 0) return false;
    const int y = div_size(c, g.rcpS), x = c - y * g.S;
    return y >= 1 && y <= g.S - 2 && x >= 1 && x <= g.S - 2;
}

// the wave's run of one image observation: the bit string `slab` (zero but for the bits at of / oh of this lane's env) -> floats
template <int OBS>
__device__ __forceinline__ void gwl_emit_image(const GridLaneGeo &g, float *blk, unsigned char *slab, int slot, int elems, int run,
                                               int lane, bool act, int hc, int fc)
{
    int of = -1, oh = -1;
    if (act) {
        if (OBS == WURM_OBS_RAW) { of = fc; oh = hc >= 0 ? g.C + hc : -1; }
        else { of = gwl_interior(g, fc) ? fc : -1; oh = gwl_interior(g, hc) ? g.C + hc : -1; }
    }
    gwl_emit_bits(blk, (u32 *)slab, slot * elems, run, lane, of, oh);
}

__device__ __forceinline__ void gwl_emit_positions(const GridLaneGeo &g, float *o, bool obs16, int hc, int fc)
{
    const int h = hc < 0 ? 0 : hc, f = fc < 0 ? 0 : fc;
    const int hy = div_size(h, g.rcpS), fy = div_size(f, g.rcpS);
    const float4 v = make_float4((float)hy, (float)(h - hy * g.S), (float)fy, (float)(f - fy * g.S));
    if (obs16) *(float4 *)o = v;
    else { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
}

template <int OBS, int EPW>
__global__ __launch_bounds__(256) void gridworld_lane_step_kernel(StepArgs p)
{
    __shared__ GridLaneScan scans[4];
    extern __shared__ __attribute__((aligned(16))) unsigned char gwl_lds[];
    const int lane = (int)(threadIdx.x & 63u), wave = uniform((int)(threadIdx.x >> 6)), wpb = (int)(blockDim.x >> 6);
    const long long env0 = (xcd_block(blockIdx.x, gridDim.x) * wpb + wave) * EPW;
    if (env0 >= p.N) return;
    const int slot = lane & (EPW - 1);
    const long long env = env0 + slot;
    const bool present = env < p.N && lane < EPW;
    const int nv = (int)min((long long)EPW, p.N - env0);
    GridLaneGeo g;
    g.S = p.S; g.C = p.S * p.S; g.I = p.S - 2;
    g.rcpS = 1.0f / (float)g.S;
    g.rcpI = g.I > 0 ? 1.0f / (float)g.I : 1.0f;
    const int S = g.S, C = g.C;

    // ---- the state: the caller's mirror when it describes it (wurm_single_call.resident, SimpleGridworld: 16 bytes of header
    // and one 32-bit record per env — agent cell | food cell << 16, 0xffff = none, GWL_REC_BAD = outside the domain), else
    // one coalesced pass over the wave's run of the planes as in the rollout kernel's prologue
    const bool mirrored = p.resident != nullptr, from_mirror = mirrored && p.resident_valid != 0;
    u32 *const recs = mirrored ? (u32 *)((unsigned char *)p.resident + GWL_MIRROR_HEADER) : nullptr;
    float *const env_run = p.envs + env0 * 2 * C;
    GridLaneScan &sc = scans[wave];
    u32 rec = 0xffffffffu;
    if (from_mirror) {
        if (present) rec = recs[env];
    } else {
        sc.cnt[0][lane] = 0; sc.cnt[1][lane] = 0; sc.pos[0][lane] = -1; sc.pos[1][lane] = -1; sc.bad[lane] = 0;
        wave_lds_sync();
        gwl_scan_run(sc, env_run, nv, C, lane);
    }
    // (independent of the scan: requested while it is in flight)
    long long a = 0;
    int pre_byte = 0;
    if (present) {
        a = load_action(p.actions, p.act_dtype, env);
        pre_byte = p.done_in != nullptr ? (int)p.done_in[env] : 0;
    }
    unsigned char *slab = nullptr;
    if ((OBS == WURM_OBS_DEFAULT || OBS == WURM_OBS_RAW) && p.lds_per_wave > 0) {
        slab = gwl_lds + wave * p.lds_per_wave;
        for (int i = lane; i < p.lds_per_wave / 16; i += 64) ((uint4 *)slab)[i] = make_uint4(0, 0, 0, 0);
    }
    wave_lds_sync();
    int fc, hc;
    bool act;
    if (from_mirror) {
        hc = (int)(short)(rec & 0xffffu);
        fc = (int)(short)(rec >> 16);
        act = present && rec != GWL_REC_BAD;
    } else {
        const int nf = sc.cnt[0][slot], nh = sc.cnt[1][slot], bad = sc.bad[slot];
        fc = sc.pos[0][slot]; hc = sc.pos[1][slot];
        act = present && !bad && nh <= 1 && nf <= 1 && (hc != fc || hc < 0); // (no agent: in the domain — see the rollout kernel)
    }
    float *const envp = env_run + (long long)slot * 2 * C;
    if (present && !act) p.done[env] = GRID_SKIPPED; // the one-env-per-wave kernel takes this env (second launch, only_flagged)
    if (mirrored && !from_mirror) { // a launch that builds the mirror counts what it cannot describe (the header's first word)
        const u64 odd = ballot(present && !act);
        if (odd != 0 && lane == 0) atomicAdd((int *)p.resident, popc64(odd));
    }
    const int hc0 = hc, fc0 = fc;
    const u64 env_id = (u64)(p.env_offset + env);
    const int start = p.start_y * S + p.start_x;

    // ---- the postponed reset (simple_gridworld.py:225-268 with call = pre_call): whatever the env held
    const bool pre = act && pre_byte != 0;
    if (ballot(pre) != 0) {
        const u32 word = rng_words(p.seed, p.pre_call, env_id, RNG_RESET, 0).w[3];
        if (pre) { hc = start; fc = free_interior_cell(g, start, word); }
    }
    // ---- the step (:135-202)
    const int ai = (int)(((a % 4) + 4) % 4);
    int newhead = -1, ny = -1, nx = -1;
    if (hc >= 0) {
        const int hy = div_size(hc, g.rcpS), hx = hc - hy * S;
        ny = hy - tap_y(ai);
        nx = hx - tap_x(ai);
        if (ny >= 0 && ny < S && nx >= 0 && nx < S) newhead = ny * S + nx;
    }
    const bool eat = act && newhead >= 0 && newhead == fc;
    hc = newhead;
    if (ballot(eat) != 0) {
        const u32 word = rng_words(p.seed, p.call, env_id, RNG_FOOD, 0).w[0];
        if (eat) fc = free_interior_cell(g, hc, word);
    }
    const bool edge = !(newhead >= 0 && ny >= 1 && ny <= S - 2 && nx >= 1 && nx <= S - 2);
    if (act) {
        p.reward[env] = eat ? 1.0f : 0.0f;
        p.done[env] = (uint8_t)edge;
        p.edgec[env] = (uint8_t)edge;
        if (p.done_copy) p.done_copy[env] = (uint8_t)edge;
    }
    // ---- observations: of the stepped state, and (obs_after) of that state once a finished env is rebuilt with call + 1
    const int elems = (int)p.obs_elems, run = nv * elems;
    const bool obs16 = (((unsigned long long)p.obs) & 15ull) == 0 && (((unsigned long long)p.obs_after) & 15ull) == 0;
    if (OBS == WURM_OBS_DEFAULT || OBS == WURM_OBS_RAW)
        gwl_emit_image<OBS>(g, p.obs + env0 * elems, slab, slot, elems, run, lane, act, hc, fc);
    else if (OBS == WURM_OBS_POSITIONS && act)
        gwl_emit_positions(g, p.obs + env * 4, obs16, hc, fc);
    if (p.obs_after != nullptr && OBS != WURM_OBS_NONE) {
        int h2 = hc, f2 = fc;
        const bool fin = act && edge;
        if (ballot(fin) != 0) {
            const u32 word = rng_words(p.seed, p.call + 1ull, env_id, RNG_RESET, 0).w[3];
            if (fin) { h2 = start; f2 = free_interior_cell(g, start, word); }
        }
        if (OBS == WURM_OBS_DEFAULT || OBS == WURM_OBS_RAW)
            gwl_emit_image<OBS>(g, p.obs_after + env0 * elems, slab, slot, elems, run, lane, act, h2, f2);
        else if (OBS == WURM_OBS_POSITIONS && act)
            gwl_emit_positions(g, p.obs_after + env * 4, obs16, h2, f2);
    }
    // ---- the stepped state back (the rebuilt one is not stored: no post_reset here): the record, and the four cells of the
    // planes that changed — unless the mirror is lazy and was current (a launch that builds it always writes the planes)
    if (mirrored && present) recs[env] = act ? (((u32)hc & 0xffffu) | ((u32)fc << 16)) : GWL_REC_BAD;
    if (act && !(from_mirror && p.resident_lazy)) {
        if (fc0 >= 0 && fc0 != fc) envp[fc0] = 0.0f;
        if (hc0 >= 0 && hc0 != hc) envp[C + hc0] = 0.0f;
        if (fc >= 0 && fc != fc0) envp[fc] = 1.0f;
        if (hc >= 0 && hc != hc0) envp[C + hc] = 1.0f;
    }
}

// the planes from the mirror's records (a lazy mirror being written out): the wave's run of EPW envs zero-filled with
// 16-byte stores, then the two cells of each env
template <int EPW>
__global__ __launch_bounds__(256) void gridworld_lane_flush_kernel(StepArgs p)
{
    const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6), wpb = (int)(blockDim.x >> 6);
    const long long env0 = ((long long)blockIdx.x * wpb + wave) * EPW;
    if (env0 >= p.N) return;
    const int C = p.S * p.S, nv = (int)min((long long)EPW, p.N - env0), total = nv * 2 * C;
    const u32 *const recs = (const u32 *)((const unsigned char *)p.resident + GWL_MIRROR_HEADER);
    float *const run = p.envs + env0 * 2 * C;
    const u32 rec = lane < nv ? recs[env0 + lane] : GWL_REC_BAD;
    if ((((unsigned long long)run) & 15ull) == 0) {
        const int n4 = total >> 2;
        for (int i = lane; i < n4; i += 64) ((float4 *)run)[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        for (int i = (n4 << 2) + lane; i < total; i += 64) run[i] = 0.0f;
    } else {
        for (int i = lane; i < total; i += 64) run[i] = 0.0f;
    }
    __builtin_amdgcn_s_waitcnt(0); // (the patches below land behind the fill: same wave, same addresses, in order)
    wave_lds_sync();
    if (lane < nv && rec != GWL_REC_BAD) {
        const int hc = (int)(short)(rec & 0xffffu), fc = (int)(short)(rec >> 16);
        float *const envp = run + (long long)lane * 2 * C;
        if (fc >= 0) envp[fc] = 1.0f;
        if (hc >= 0) envp[C + hc] = 1.0f;
    }
}

} // namespace

// SimpleGridworld rollouts this translation unit takes: RNG mode, a start location, 'default' / 'raw' / 'positions' or no
// observation (the reference's gridworld has no other mode), batches from WURM_LANE_ROLLOUT_MIN_ENVS on (below that the
// one-env-per-wave kernels have more waves than this one would)
bool gridworld_lane_eligible(const StepArgs &p)
{
    if (p.inject_food || p.inject_reset || p.only_flagged || p.S < 5 || p.S > 64 || p.T < 1) return false; // (S <= 4: no reset, simple_gridworld.py:249-250)
    if (p.start_y < 0 || p.start_x < 0 || p.start_y >= p.S || p.start_x >= p.S) return false;
    if (p.N < opt.lane_rollout_min_envs) return false;
    return p.obs_mode == WURM_OBS_DEFAULT || p.obs_mode == WURM_OBS_RAW || p.obs_mode == WURM_OBS_POSITIONS ||
           p.obs_mode == WURM_OBS_NONE;
}

constexpr int GWL_SLAB_MAX = 16384; // bytes per wave of the image modes' bit string (one bit per float of the wave's run; 4 waves per workgroup: 64 KB)
// bytes of the bit string of a run of `floats` floats, in whole 16-byte pieces (the kernels zero it with 16-byte LDS stores)
static long long gwl_slab_bytes(long long floats) { return (((floats + 31) / 32) * 4 + 15) & ~15ll; }

// The slabs of a workgroup's four waves can reach 64 KB of DYNAMIC LDS, and the kernels keep 5 KB of static scan state beside
// them: past the 64 KB a launch gets by default the kernel is opted into the larger budget, as lane_rollout's launcher does
// (it only worked because the runtime does not enforce the opt-in on a 160 KB part).
static void gwl_allow_lds(const void *kernel, size_t dynamic_bytes)
{
    if (dynamic_bytes + sizeof(GridLaneScan) * 4 > 65536)
        (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dynamic_bytes);
}

template <int OBS, int EPW>
static void launch_epw(const StepArgs &p0, hipStream_t stream)
{
    StepArgs p = p0;
    const long long waves = (p.N + EPW - 1) / EPW;
    const int wpb = 4;
    const dim3 block(64 * wpb), grid((unsigned)((waves + wpb - 1) / wpb));
    const long long slab = (OBS == WURM_OBS_DEFAULT || OBS == WURM_OBS_RAW) ? gwl_slab_bytes(EPW * p.obs_elems) : 0;
    p.lds_per_wave = slab <= GWL_SLAB_MAX ? (int)slab : 0;
    gwl_allow_lds((const void *)gridworld_lane_rollout_kernel<OBS, EPW>, (size_t)p.lds_per_wave * wpb);
    WURM_LAUNCH((gridworld_lane_rollout_kernel<OBS, EPW>), grid, block, (size_t)p.lds_per_wave * wpb, stream, p);
}

template <int OBS>
static void launch_image(const StepArgs &p, hipStream_t stream)
{
    // envs per wave of the image modes (WURM_GRIDWORLD_LANE_EPW pins it).  Measured at 65 536 x 9 x 9 (tools/gridworld_probe.py,
    // one box): 'default' 16-step launch 0.299 / 0.275 / 0.263 ms at 8 / 16 / 32, 'raw' 0.203 / 0.194 / 0.192 ms
    long long epw = opt.gridworld_lane_epw;
    // round 6 (the run as a bit string, 1-2 KB of LDS per wave instead of 8-16 KB; profiles/r06_gridworld_rollout.txt): 'default'
    // 0.298 / 0.266 / 0.253 / 0.247 ms at 8 / 16 / 32 / 64 per 16 steps and 1.00 / 0.96 / 0.85 / 0.81 per 64 (one env per wave: 0.34
    // and 0.92), 'raw' 0.206 / 0.188 / 0.184 / 0.184
    if (epw != 4 && epw != 8 && epw != 16 && epw != 32 && epw != 64) epw = p.N >= 65536 ? 64 : p.N >= 32768 ? 32 : p.N >= 16384 ? 16 : 8;
    switch (epw) {
    case 4: launch_epw<OBS, 4>(p, stream); break;
    case 8: launch_epw<OBS, 8>(p, stream); break;
    case 16: launch_epw<OBS, 16>(p, stream); break;
    case 32: launch_epw<OBS, 32>(p, stream); break;
    default: launch_epw<OBS, 64>(p, stream); break;
    }
}

hipError_t launch_gridworld_lane_rollout(const StepArgs &p0, hipStream_t stream)
{
    const StepArgs &p = p0;
    (void)hipGetLastError();
    switch (p.obs_mode) {
    case WURM_OBS_DEFAULT: launch_image<WURM_OBS_DEFAULT>(p, stream); break;
    case WURM_OBS_RAW: launch_image<WURM_OBS_RAW>(p, stream); break;
    case WURM_OBS_POSITIONS: launch_epw<WURM_OBS_POSITIONS, 64>(p, stream); break;
    default: launch_epw<WURM_OBS_NONE, 64>(p, stream); break;
    }
    return hipGetLastError();
}

// envs per wave of the per-call kernel: by batch size, halved until the bit string of an image mode's run fits the LDS budget; 0: none does (never, up to 64 x 64)
static int gridworld_lane_step_epw(const StepArgs &p)
{
    // (one launch per call is a chain of latencies, not a stream: fewer envs per wave than the rollout.  Measured per
    // iteration of `step; reset` at 4 / 8 / 16 / 32 envs per wave, tools/gridworld_percall_probe.py: 65 536 x 9 x 9 'default'
    // 40.2 / 42.9 / 40.9 / 44.8 us (one env per wave 50.6), 'positions' 21.9 / 19.9 / 20.6 / 24.9 (38.2); 16 384 envs
    // 'default' 17.3 / 19.2 / 21.7 / 26.6 (18.1))
    // On the mirror (round 6) the launch neither scans nor writes the planes — a pure writer of its run: 65 536 x 9 x 9 'default'
    // with the reset observation 27.3 / 27.9 / 27.4 / 24.1 / 26.2 us at 4 / 8 / 16 / 32 / 64 (without the mirror, two launches:
    // 36.2), 'raw' 20.7 / 19.6 / 19.5 / 19.5 / 18.3; 16 384 envs 'default' 10.8 / 10.6 / 10.6 / 11.3 / 13.3
    // (profiles/r06_gridworld_percall_mirror.txt)
    long long epw = opt.gridworld_lane_epw;
    if (epw != 4 && epw != 8 && epw != 16 && epw != 32 && epw != 64) {
        if (p.resident != nullptr && p.resident_valid) epw = p.N >= 65536 ? 32 : p.N >= 16384 ? 16 : 8;
        else epw = p.N >= 65536 ? 16 : p.N >= 32768 ? 8 : 4;
    }
    if (p.obs_mode == WURM_OBS_DEFAULT || p.obs_mode == WURM_OBS_RAW)
        while (epw >= 4 && gwl_slab_bytes(epw * p.obs_elems) > GWL_SLAB_MAX) epw >>= 1;
    return epw >= 4 ? (int)epw : 0;
}

// the per-call step this translation unit takes (K_STEP / K_FUSED of SimpleGridworld): RNG mode, no immediate reset, batches
// from WURM_LANE_STEP_MIN_ENVS on, a start location whenever a reset is part of the call
bool gridworld_lane_step_eligible(const StepArgs &p)
{
    if (p.inject_food || p.inject_reset || p.inject_pre_reset || p.post_reset || p.only_flagged || p.S < 5 || p.S > 64) return false;
    if (p.N < opt.lane_step_min_envs) return false;
    if ((p.done_in != nullptr || p.obs_after != nullptr) &&
        (p.start_y < 0 || p.start_x < 0 || p.start_y >= p.S || p.start_x >= p.S)) return false;
    if (!(p.obs_mode == WURM_OBS_DEFAULT || p.obs_mode == WURM_OBS_RAW || p.obs_mode == WURM_OBS_POSITIONS ||
          p.obs_mode == WURM_OBS_NONE)) return false;
    return gridworld_lane_step_epw(p) != 0;
}

template <int OBS, int EPW>
static void launch_step_epw(const StepArgs &p0, hipStream_t stream)
{
    StepArgs p = p0;
    const long long waves = (p.N + EPW - 1) / EPW;
    const int wpb = 4;
    const dim3 block(64 * wpb), grid((unsigned)((waves + wpb - 1) / wpb));
    p.lds_per_wave = (OBS == WURM_OBS_DEFAULT || OBS == WURM_OBS_RAW) ? (int)gwl_slab_bytes(EPW * p.obs_elems) : 0;
    gwl_allow_lds((const void *)gridworld_lane_step_kernel<OBS, EPW>, (size_t)p.lds_per_wave * wpb);
    WURM_LAUNCH((gridworld_lane_step_kernel<OBS, EPW>), grid, block, (size_t)p.lds_per_wave * wpb, stream, p);
}

template <int OBS>
static void launch_step_obs(const StepArgs &p, hipStream_t stream)
{
    switch (gridworld_lane_step_epw(p)) {
    case 4: launch_step_epw<OBS, 4>(p, stream); break;
    case 8: launch_step_epw<OBS, 8>(p, stream); break;
    case 16: launch_step_epw<OBS, 16>(p, stream); break;
    case 32: launch_step_epw<OBS, 32>(p, stream); break;
    default: launch_step_epw<OBS, 64>(p, stream); break;
    }
}

// the mirror of the per-call step (wurm_single_call.resident of a SimpleGridworld): GWL_MIRROR_HEADER bytes + one record per env;
// 0 where the per-call lane kernel does not serve the shape (an image observation whose run of four envs exceeds the LDS budget of its bit string: none up to 64 x 64)
long long gridworld_resident_bytes(long long N, int S, int obs_mode, long long obs_elems)
{
    if (N <= 0) return 0;
    StepArgs p = {};
    p.N = N; p.S = S; p.obs_mode = obs_mode; p.obs_elems = obs_elems;
    return gridworld_lane_step_epw(p) != 0 ? GWL_MIRROR_HEADER + 4 * N : 0;
}

hipError_t launch_gridworld_lane_flush(const StepArgs &p, hipStream_t stream)
{
    (void)hipGetLastError();
    constexpr int EPW = 16, wpb = 4;
    const long long waves = (p.N + EPW - 1) / EPW;
    WURM_LAUNCH((gridworld_lane_flush_kernel<EPW>), dim3((unsigned)((waves + wpb - 1) / wpb)), dim3(64 * wpb), 0, stream, p);
    return hipGetLastError();
}

hipError_t launch_gridworld_lane_step(const StepArgs &p, hipStream_t stream)
{
    (void)hipGetLastError();
    switch (p.obs_mode) {
    case WURM_OBS_DEFAULT: launch_step_obs<WURM_OBS_DEFAULT>(p, stream); break;
    case WURM_OBS_RAW: launch_step_obs<WURM_OBS_RAW>(p, stream); break;
    case WURM_OBS_POSITIONS: launch_step_obs<WURM_OBS_POSITIONS>(p, stream); break;
    default: launch_step_obs<WURM_OBS_NONE>(p, stream); break;
    }
    return hipGetLastError();
}

} // namespace wurm
