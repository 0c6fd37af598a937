// multi_snake.hip — gfx950 kernels and C-ABI entry points for MultiSnake.
//
// Replaces (citations into oscarknagg/wurm):
//   MultiSnake.step     wurm/envs/multi_snake.py:462-731   two phases (boost, regular) of move / eat / decay /
//                       collide / grow / edge, food-on-death, boost cost, food respawn, death reward
//   MultiSnake._observe wurm/envs/multi_snake.py:175-227,268-334   'full' (per-agent RGB) and 'partial_n'
//   MultiSnake.reset    wurm/envs/multi_snake.py:771-1019  env re-creation, colour re-roll, respawn 'any'
//   check_consistency   wurm/envs/multi_snake.py:733-769
// The reference runs these as conv2d / einsum / repeat_interleave(xK) / masked scatter sequences with many host
// syncs; here each call is ONE launch, one env per wavefront, runtime K and S:
//   * the env's K body grids live in LDS as 16-bit cells [K][S*S] (bit 15 = "must be written back"), the food grid
//     as bytes; lane l owns cells l + 64k of every grid, so HBM traffic is coalesced dword runs and every grid
//     update is a conflict-free LDS access;
//   * a body cell holds an EXPIRY CLOCK, not a value: value = max(ex - T_s, 0) with one clock T_s per snake.  "Every
//     body cell of a mover decays by one" (:523-526 / :627-628) is T_s += 1, deleting a dead snake (:595-596 /
//     :676-677) is T_s = CLOCK_DEAD — no pass over the grid; only the new head segment is written;
//   * per-SNAKE scalars (head cell, length, done, orientation, reward ...) live one per lane in lanes 0..K-1, so
//     the snake-level logic of all K snakes runs in parallel and exchanges values with shuffles / ballots;
//   * cross-cell lookups (food under a head, bodies under a head) are single LDS reads at the head cell.
// Integer/index work: no MFMA.  Bound: HBM ((1+2K)*S*S*4 B read + observation written per env-step).
#include <type_traits>

#include "wurm_device.hpp"
#include "../../include/wurm_hip.h"

namespace wurm {

constexpr unsigned short DIRTY = 0x8000u;
constexpr unsigned short VMASK = 0x7fffu;
constexpr int CLOCK_DEAD = 0x7fff;   // clock of a deleted snake: every cell of its grid reads 0
#ifndef WURM_MULTI_CLOCK_REBASE
#define WURM_MULTI_CLOCK_REBASE 0x3000
#endif
constexpr int CLOCK_REBASE = WURM_MULTI_CLOCK_REBASE; // rollout: re-base a snake's grid (ex -= T) past this clock

struct MultiArgs {
    float *foods, *heads, *bodies;
    uint8_t *dones;
    long long *orientations;
    const long long *actions;
    uint8_t *boost;
    float *rewards;
    uint8_t *snakecol, *edgecol;
    float *foodcons, *sizes;
    uint8_t *all_done;
    short *colours;
    float *obs;
    float *obs_after;     // multi_step_kernel, nullable: the observation the caller's reset(all_done) will return
    int obs_mode, obs_n;
    long long obs_elems;
    long long N;
    int K, S;
    wurm_multi_config cfg;
    u64 seed, call;
    long long env_offset;
    wurm_multi_inject inj;
    int has_inj;
    const uint8_t *done_env;
    int *status;
    u64 pre_call;            // multi_step_kernel: counter of the postponed reset applied in front of the step
    uint8_t *all_done_copy;  // nullable: second copy of all_done (a buffer the caller cannot modify)
    wurm_multi_reset_inject rinj;
    int has_rinj;
    uint32_t *err;
    uint32_t *err_after;  // step kernels: the mask of the state reset_for_obs_after leaves (with obs_after), nullable
    float *am_f32;
    uint8_t *am_u8;
    long long T;          // rollout: number of fused step+reset iterations
    uint8_t *boost_state; // rollout: boost_this_step (N*K) written back at the end
    int lds_per_wave, off_body, off_food, off_occ, off_hmap, off_img, off_col, off_snap, off_acts;
    int off_tl; // timeline build only: 32 stamp slots per env (WURM_TLS)
    // per-call step: the caller's compact mirror of foods / heads / bodies (wurm_multi_call.resident), nullable; valid: it
    // describes them; lazy: the step does not write them
    unsigned char *resident;
    int resident_valid, resident_lazy;
    // multi_rollout_group_kernel: offsets (bytes, from the start of the workgroup's LDS) of the env blocks, the two class
    // code buffers and the two output buffers, and the size of one env's share of each
    int grp_env0, grp_codes, grp_outs, grp_save, grp_code_bytes, grp_out_bytes;
    int grp_variant; // WURM_MULTI_GROUP_VARIANT.  bit 0 (every build): multi_step_wg_kernel writes whole agent views per wave (A/B switch,
                     // same bytes); multi_rollout_group_kernel, probe build only: bit 2 no observation stores, bit 3 no transition
    int resident_used; // out (host side): the rollout launch kept the mirror (multi_rollout_group_kernel)
    int grp_emit; // multi_step_kernel: the workgroup's waves write the 'full' observations together (grp_env0: the table)
};

struct Ctx {
    int S, C, K, lane, cpl;
    float rcpS;
    int *hcell;            // [K] head cell per snake (-1 = none)
    int *lmax;             // [K] max body value per snake
    int *tclk;             // [K] clock per snake: body value = max(ex - tclk, 0)
    unsigned short *body;  // [K][C] expiry clocks (low 15 bits) | DIRTY
    unsigned char *food;   // [C]
    unsigned char *occ;    // [C] scratch (reset: occupancy)
    unsigned char *hmap;   // [C] head owner + 1 per cell, all-zero outside observe_full
    unsigned short *snap;  // [C] observe_full_snap: class code per cell
    unsigned char *acts;   // [64][K] rollout: the actions of the current 64-step chunk (see multi_rollout_kernel)
    short *img;            // [C][4] env image (partial_n): r, g, b, 0
    float *colf;           // [K][4]: r, g, b, 1 + 0.5*boost
    unsigned long long *tl; // timeline build: stamp slots (WURM_TLS)
    u64 ring;              // bit k <=> cell lane + 64 k lies on the border ring (border_bits), where make_ctx was asked for it
    bool has_ring;
};

extern __shared__ __attribute__((aligned(16))) unsigned char wurm_multi_lds[];
__host__ __device__ inline int multi_layout(MultiArgs &p, bool need_img, int need_snap); // (host side, below)

// Shape-specialised kernels (round 6).  KT / ST / NT > 0: the number of snakes, the grid size and the crop radius are
// compile-time constants — the shapes of the reference's own experiments (4 snakes on 25 x 25 with partial_5 crops:
// experiments/multiagent.py:79-86, tests/test_multi_snake_env.py:100-104; 10 snakes on 36 x 36: experiments/speeds.py) — so
// every loop over snakes, rows of 64 cells and window cells has a known trip count, the divisions by S are by a constant and
// the LDS offsets multi_launch worked out are immediates.  Same source, same results (tests/test_multi_shape_kernels.py
// compares the two bit for bit); the generic kernels serve every other shape.  WURM_MULTI_SHAPE_KERNELS = 0 switches them off.
constexpr int SNAP_MAX_SNAKES = 10; // observe_full_snap: 10 mask bits, and owner + 1 <= 11 fits the 4 owner bits
template <int OBS, int KT, int ST, int NT>
__device__ __forceinline__ void shape_constants(MultiArgs &p, bool layout, int snap_buffers = -1)
{
    if (KT > 0) p.K = KT;
    if (ST > 0) p.S = ST;
    if (NT >= 0) p.obs_n = NT;
    if (OBS == WURM_OBS_PARTIAL && NT >= 0) p.obs_elems = 3ll * (2 * NT + 1) * (2 * NT + 1);
    if (OBS == WURM_OBS_DEFAULT && ST > 0) p.obs_elems = 3ll * ST * ST;
    if (layout && KT > 0 && ST > 0 && OBS >= 0) {
        const int snap = snap_buffers >= 0 ? snap_buffers : (OBS == WURM_OBS_DEFAULT && KT <= SNAP_MAX_SNAKES) ? 1 : 0;
        (void)multi_layout(p, OBS == WURM_OBS_PARTIAL, snap);
    }
}

__device__ __forceinline__ u64 border_bits(const Ctx &cx); // (with the grouped 'full' writer below)
__device__ __forceinline__ Ctx make_ctx(const MultiArgs &p, int wave, int base_off = 0, bool want_ring = false)
{
    Ctx cx;
    unsigned char *base = wurm_multi_lds + base_off + (size_t)wave * p.lds_per_wave;
    cx.S = p.S;
    cx.C = p.S * p.S;
    cx.K = p.K;
    cx.lane = (int)(threadIdx.x & 63u);
    cx.cpl = (cx.C + 63) >> 6;
    cx.rcpS = 1.0f / (float)p.S;
    cx.hcell = (int *)base;
    cx.lmax = (int *)(base + 4 * p.K);
    cx.tclk = (int *)(base + 8 * p.K);
    cx.body = (unsigned short *)(base + p.off_body);
    cx.food = base + p.off_food;
    cx.occ = base + p.off_occ;
    cx.hmap = base + p.off_hmap;
    cx.snap = (unsigned short *)(base + (p.off_snap >= 0 ? p.off_snap : 0));
    cx.acts = base + p.off_acts;
    cx.img = (short *)(base + p.off_img);
    cx.colf = (float *)(base + p.off_col);
    cx.tl = (unsigned long long *)(base + p.off_tl);
    cx.has_ring = want_ring;
    cx.ring = want_ring ? border_bits(cx) : 0ull;
    return cx;
}

// body value of snake s at cell c
__device__ __forceinline__ int BV(const Ctx &cx, int s, int c)
{
    return max((int)(cx.body[s * cx.C + c] & VMASK) - cx.tclk[s], 0);
}

// ------------------------------------------------------------------------------------------------ load / store

// HBM -> LDS.  Returns the lane's original food bits (bit k = food at cell lane + 64k).
// heads and bodies of one env are each one contiguous run of K*C floats with the same [K][C] layout as the LDS body
// grid, so they are copied as flat lane-strided streams, LOAD_CHUNK dwords per lane in flight at a time (the wave is
// alone with its latency at 16 waves/CU: few large batches of loads, not many small ones).
constexpr int LOAD_CHUNK = 16;

// plain (out if want_plain, wave-uniform; a reference, not a pointer: a conditional pointer to a local puts it in scratch):
// the planes held nothing the LDS image cannot represent — food and head values 0 / 1,
// at most one head per snake, body values integers in 0 .. 0x7fff — so lds_check sees all there is to check.
__device__ __forceinline__ u64 load_env(const Ctx &cx, const float *__restrict__ foodp,
                                        const float *__restrict__ headp, const float *__restrict__ bodyp,
                                        bool want_plain, bool &plain)
{
    const int C = cx.C, lane = cx.lane, KC = cx.K * C;
    int odd = 0, nheads = 0;
    if (lane < cx.K) {
        cx.hcell[lane] = -1;
        cx.lmax[lane] = 0;
        cx.tclk[lane] = 0; // values are loaded as they are: ex = value
    }
    for (int c = lane; c < C; c += 64) cx.hmap[c] = 0;
    wave_lds_sync();
    const float rcpC = 1.0f / (float)C;
    for (int base = 0; base < KC; base += 64 * LOAD_CHUNK) {
        float hv[LOAD_CHUNK], bv[LOAD_CHUNK];
#pragma unroll
        for (int j = 0; j < LOAD_CHUNK; ++j) {
            // unconditional loads (index clamped into the env): a `cond ? load : 0` would make the compiler wait
            // for every load at its own join point and serialise the batch
            const int i = min(base + lane + 64 * j, KC - 1);
            hv[j] = headp[i];
            bv[j] = bodyp[i];
        }
#pragma unroll
        for (int j = 0; j < LOAD_CHUNK; ++j) {
            const int i = base + lane + 64 * j;
            if (i < KC) {
                const int bi = __float2int_rn(bv[j]);
                cx.body[i] = (unsigned short)(bi != 0 ? ((bi & VMASK) | DIRTY) : 0);
                odd |= (int)((hv[j] != 0.0f && hv[j] != 1.0f) || bv[j] != (float)bi || bi < 0 || bi > (int)VMASK);
                nheads += (int)(hv[j] > 0.5f);
                if (hv[j] > 0.5f || bi > 0) { // rare: a head cell or a body cell
                    const int s = div_size(i, rcpC);
                    if (hv[j] > 0.5f) cx.hcell[s] = i - s * C;
                    if (bi > 0) atomicMax(&cx.lmax[s], bi);
                }
            }
        }
    }
    u64 fbits = 0;
    for (int k0 = 0; k0 < cx.cpl; k0 += LOAD_CHUNK) {
        float fv[LOAD_CHUNK];
#pragma unroll
        for (int j = 0; j < LOAD_CHUNK; ++j) {
            fv[j] = foodp[min(lane + 64 * (k0 + j), C - 1)];
        }
#pragma unroll
        for (int j = 0; j < LOAD_CHUNK; ++j) {
            const int c = lane + 64 * (k0 + j);
            if (k0 + j < cx.cpl && c < C) {
                const int f = fv[j] > 0.5f;
                cx.food[c] = (unsigned char)f;
                fbits |= (u64)f << (k0 + j);
                odd |= (int)(fv[j] != 0.0f && fv[j] != 1.0f);
            }
        }
    }
    wave_lds_sync();
    if (want_plain) // as many heads as snakes that have one <=> nobody has two
        plain = ballot(odd != 0) == 0 && wave_sum_i32(nheads) == popc64(ballot(lane < cx.K && cx.hcell[lane] >= 0));
    return fbits;
}

__device__ __forceinline__ u64 load_env(const Ctx &cx, const float *__restrict__ foodp, const float *__restrict__ headp,
                                        const float *__restrict__ bodyp)
{
    bool unused = false;
    return load_env(cx, foodp, headp, bodyp, false, unused);
}

// LDS -> HBM: body cells flagged DIRTY, the two head cells that changed, food cells that changed.
// t0_in_lmax: the clocks the snakes had when the env was loaded are in cx.lmax (a state that came from the mirror keeps
// its clocks between calls); else they were 0 (load_env).
__device__ __forceinline__ void store_env(const Ctx &cx, float *__restrict__ foodp, float *__restrict__ headp,
                                          float *__restrict__ bodyp, u64 fbits0, int hc0, int hc, bool full,
                                          bool t0_in_lmax = false)
{
    const int C = cx.C, lane = cx.lane;
    for (int s = 0; s < cx.K; ++s) {
        float *bp = bodyp + (size_t)s * C, *hp = headp + (size_t)s * C;
        const int hs = cx.hcell[s], T = cx.tclk[s], T0 = t0_in_lmax ? cx.lmax[s] : 0;
#pragma unroll 4
        for (int k = 0; k < cx.cpl; ++k) {
            int c = lane + 64 * k;
            if (c < C) {
                const unsigned short v = cx.body[s * C + c];
                // changed since the load: written cells, and — once the clock has moved — every cell that held a value
                if (full || (v & DIRTY) || (T != T0 && (int)(v & VMASK) > T0)) bp[c] = (float)max((int)(v & VMASK) - T, 0);
                if (full) hp[c] = (c == hs) ? 1.0f : 0.0f;
            }
        }
    }
    if (!full && lane < cx.K && hc != hc0) {
        float *hp = headp + (size_t)lane * C;
        if (hc0 >= 0) hp[hc0] = 0.0f;
        if (hc >= 0) hp[hc] = 1.0f;
    }
    for (int k = 0; k < cx.cpl; ++k) {
        int c = lane + 64 * k;
        if (c < C) {
            int f = cx.food[c] != 0;
            if (full || f != (int)((fbits0 >> k) & 1)) foodp[c] = f ? 1.0f : 0.0f;
        }
    }
}

// ------------------------------------------------------------------------------------------------ the mirror
// wurm_multi_call.resident: per env the LDS image of its grids — the K body grids as 16-bit expiry clocks (without the
// DIRTY bits), the food grid as bytes, and per snake its clock, head cell and length — kept by the caller between calls,
// so that the per-call step copies (2 K + 1) S^2 bytes into LDS instead of reading and converting (1 + 2 K) S^2 fp32.
// Any state load_env accepts is representable (the image IS what load_env produces), so there is no domain and no fallback.
__host__ __device__ __forceinline__ int mirror_body_bytes(int K, int C) { return (2 * K * C + 15) & ~15; }
__host__ __device__ __forceinline__ int mirror_food_bytes(int C) { return (C + 15) & ~15; }
__host__ __device__ __forceinline__ long long mirror_env_bytes(int K, int C)
{
    return (long long)mirror_body_bytes(K, C) + mirror_food_bytes(C) + ((12 * K + 15) & ~15);
}

// mirror -> LDS by `nth` threads (tid 0..nth-1; nth = 64: one wave, then `sync` is a wave-level LDS fence).  Returns the
// thread's food bits in load_env's layout (bit k = food at cell tid + nth * k).  hcell / lmax / tclk as load_env leaves
// them (lmax = the snake's length).
template <typename Sync>
__device__ __forceinline__ u64 mirror_load(const Ctx &cx, const unsigned char *__restrict__ m, int tid, int nth, Sync sync,
                                           bool want_bits = true)
{
    const int C = cx.C, K = cx.K, nb = mirror_body_bytes(K, C) >> 4, nf = mirror_food_bytes(C) >> 4;
    const uint4 *mb = (const uint4 *)m, *mf = (const uint4 *)(m + mirror_body_bytes(K, C));
    const int *ms = (const int *)(m + mirror_body_bytes(K, C) + mirror_food_bytes(C));
    // (the grids start on 16-byte boundaries in LDS and are followed by padding up to the next one: multi_layout)
    uint4 *lb = (uint4 *)cx.body, *lf = (uint4 *)cx.food;
    if (nb <= 8 * nth && nf <= nth) {
        // the whole image in ONE round of loads (cfg4: 313 + 40 sixteen-byte pieces and 12 ints for one wave): bodies, food and
        // the per-snake words are requested before anything is waited for — three dependent round trips took 13 400 cycles of
        // a stepper's 67 000 per call (tools/multi_timeline.py) — and the head map is cleared while they are under way
        uint4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = mb[min(tid + j * nth, nb - 1)];
        const uint4 f = mf[min(tid, nf - 1)];
        int w0 = 0, w1 = 0, w2 = 0;
        if (tid < K) { w0 = ms[tid]; w1 = ms[K + tid]; w2 = ms[2 * K + tid]; }
        for (int c = tid; c < C; c += nth) cx.hmap[c] = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (tid + j * nth < nb) lb[tid + j * nth] = v[j];
        if (tid < nf) lf[tid] = f;
        if (tid < K) { cx.tclk[tid] = w0; cx.hcell[tid] = w1; cx.lmax[tid] = w2; }
        WURM_TLS(cx, 13);
    } else {
        for (int i0 = 0; i0 < nb; i0 += 8 * nth) {
            uint4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = mb[min(i0 + tid + j * nth, nb - 1)];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (i0 + tid + j * nth < nb) lb[i0 + tid + j * nth] = v[j];
        }
        for (int i = tid; i < nf; i += nth) lf[i] = mf[i];
        if (tid < K) {
            cx.tclk[tid] = ms[tid];
            cx.hcell[tid] = ms[K + tid];
            cx.lmax[tid] = ms[2 * K + tid];
        }
        for (int c = tid; c < C; c += nth) cx.hmap[c] = 0;
    }
    sync();
    u64 fbits = 0;
    if (want_bits) { // (only store_env — the write-back to the fp32 planes — compares with them)
#pragma unroll 5
        for (int k = 0, c = tid; c < C; ++k, c += nth) fbits |= (u64)(cx.food[c] != 0) << k;
    }
    WURM_TLS(cx, 14);
    return fbits;
}

// LDS -> mirror (the DIRTY bits stay behind: they mean "written since the load from fp32").  hc / L: the snake's head
// cell and length as of now (threads 0..K-1).
// sparse: the grids came from this mirror in this launch — only the body cells written since (DIRTY) are stored, and the
// food grid whole (C bytes).
template <typename Sync>
__device__ __forceinline__ void mirror_store(const Ctx &cx, unsigned char *__restrict__ m, int tid, int nth, int hc, int L,
                                             Sync sync, bool sparse = false, u64 fbits0 = 0)
{
    const int C = cx.C, K = cx.K, nb = mirror_body_bytes(K, C) >> 4, nf = mirror_food_bytes(C) >> 4;
    uint4 *mb = (uint4 *)m, *mf = (uint4 *)(m + mirror_body_bytes(K, C));
    int *ms = (int *)(m + mirror_body_bytes(K, C) + mirror_food_bytes(C));
    const uint4 *lb = (const uint4 *)cx.body, *lf = (const uint4 *)cx.food;
    sync();
    const u32 keep = (u32)VMASK * 0x00010001u, dirty = (u32)DIRTY * 0x00010001u;
    (void)fbits0;
    if (sparse) {
        // (LDS reads in batches, the few stores afterwards: read-test-store cell by cell was a chain of dependent LDS round
        // trips — 5 400 cycles of a stepper's 67 000 per call at cfg4)
        unsigned short *mb16 = (unsigned short *)m;
        for (int i0 = 0; i0 < nb; i0 += 4 * nth) {
            uint4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = lb[min(i0 + tid + j * nth, nb - 1)];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = i0 + tid + j * nth;
                if (i >= nb || ((v[j].x | v[j].y | v[j].z | v[j].w) & dirty) == 0) continue;
                const u32 w[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (w[q] & (u32)DIRTY) mb16[8 * i + 2 * q] = (unsigned short)(w[q] & VMASK);
                    if (w[q] & ((u32)DIRTY << 16)) mb16[8 * i + 2 * q + 1] = (unsigned short)((w[q] >> 16) & VMASK);
                }
            }
        }
        // (the food grid whole: C bytes per env — cheaper than finding the few cells that changed)
        for (int i = tid; i < nf; i += nth) mf[i] = lf[i];
    } else {
        for (int i = tid; i < nb; i += nth) {
            uint4 v = lb[i];
            v.x &= keep; v.y &= keep; v.z &= keep; v.w &= keep;
            mb[i] = v;
        }
        for (int i = tid; i < nf; i += nth) mf[i] = lf[i];
    }
    if (tid < K) {
        ms[tid] = cx.tclk[tid];
        ms[K + tid] = hc;
        ms[2 * K + tid] = L;
    }
}

// foods / heads / bodies from the mirror (lazy form), no LDS: one workgroup per env
__global__ __launch_bounds__(256) void multi_flush_kernel(MultiArgs p)
{
    const long long env = blockIdx.x;
    const int C = p.S * p.S, K = p.K, KC = K * C, tid = (int)threadIdx.x, nth = (int)blockDim.x;
    const unsigned char *m = p.resident + env * mirror_env_bytes(K, C);
    const unsigned short *mb = (const unsigned short *)m;
    const unsigned char *mf = m + mirror_body_bytes(K, C);
    const int *ms = (const int *)(m + mirror_body_bytes(K, C) + mirror_food_bytes(C));
    float *foodp = p.foods + env * C, *headp = p.heads + env * KC, *bodyp = p.bodies + env * KC;
    const float rcpC = 1.0f / (float)C;
    for (int i = tid; i < KC; i += nth) {
        const int s = div_size(i, rcpC);
        bodyp[i] = (float)max((int)(mb[i] & VMASK) - ms[s], 0);
        headp[i] = (i - s * C == ms[K + s]) ? 1.0f : 0.0f;
    }
    for (int c = tid; c < C; c += nth) foodp[c] = mf[c] ? 1.0f : 0.0f;
}

// ------------------------------------------------------------------------------------------------ step pieces

// One phase of MultiSnake.step (boost phase multi_snake.py:509-563, regular phase :613-660) for the snakes
// (lanes) with who == true.  All per-snake values are per-lane (lane = snake index).
__device__ __forceinline__ void run_phase(const Ctx &cx, bool who, int dir, int &hc, int &L, bool &done,
                                          float &reward, float &foodcons, bool &snakecol, bool &edgecol)
{
    const int S = cx.S, C = cx.C, K = cx.K, lane = cx.lane;
    const bool snake = lane < K;
    // move heads (:509 / :613, _move_heads :341-353): by -TAP[dir]; off the grid => the head vanishes
    if (who && hc >= 0) {
        int y = div_size(hc, cx.rcpS), x = hc - y * S;
        int ny = y - tap_y(dir), nx = x - tap_x(dir);
        hc = (ny >= 0 && ny < S && nx >= 0 && nx < S) ? ny * S + nx : -1;
    }
    // food overlap of ALL snakes (:514 / :618); each eaten cell loses its food once (:517-518 / :622)
    const bool ov = snake && hc >= 0 && cx.food[hc] != 0;
    wave_lds_sync();
    if (ov) cx.food[hc] = 0;
    // decay the movers that did not eat (:523-526 / :627-628): their clock advances
    if (who && !ov) cx.tclk[lane] += 1;
    if (who && ov) { // :527-529 / :629-631
        reward += 1.0f;
        foodcons += 1.0f;
    }
    wave_lds_sync();
    // collisions with any body (after the decay) or another snake's head (:534-547 / :636-644)
    bool coll = false;
    if (who && hc >= 0) {
        int sum = 0;
#pragma unroll 4
        for (int t = 0; t < K; ++t) sum += BV(cx, t, hc);
        coll = sum > 0;
    }
    for (int o = 0; o < K; ++o) {
        int ho = lane_value(hc, o);
        if (who && hc >= 0 && o != lane && ho == hc) coll = true;
    }
    done |= coll;
    snakecol |= coll;
    wave_lds_sync();
    // new head segment (:552-555 / :649-652)
    if (who && hc >= 0) {
        int v = BV(cx, lane, hc);
        cx.body[lane * C + hc] = (unsigned short)(((cx.tclk[lane] + v + L + (ov ? 1 : 0)) & VMASK) | DIRTY);
    }
    if (who && ov) L += 1;
    // edge collisions (:560-562 / :657-659)
    if (who && hc >= 0) {
        int y = div_size(hc, cx.rcpS), x = hc - y * S;
        bool e = y == 0 || x == 0 || y == S - 1 || x == S - 1;
        done |= e;
        edgecol |= e;
    }
    wave_lds_sync();
}

// _food_from_death (:416-428) as applied at :565-576 / :662-673.  Returns the number of cells where the food landed on a cell
// that held food already: `self.foods += food_on_death` makes those 2 until the clamp at the end of the phase (:603 / :692),
// and the second phase's _add_food (:680) sums the food plane BEFORE its clamp — the test against max_food sees them twice
// (a dead body over food only comes from a hand-edited state; round 6's fuzz found the step after one: seed 722).
__device__ __forceinline__ int food_from_death(const Ctx &cx, bool done, bool has_body, const uint8_t *inj,
                                               float thr, u64 seed, u64 call, u64 env_id, u32 purpose)
{
    const int S = cx.S, C = cx.C, lane = cx.lane;
    const bool snake = lane < cx.K;
    const u64 dead = ballot(snake && done && has_body);
    if (!dead) return 0;
    int doubled = 0;
    const u64 live = ballot(snake && !done);
    for (int k = 0; k < cx.cpl; ++k) {
        int c = lane + 64 * k;
        if (c >= C) continue;
        int y = div_size(c, cx.rcpS), x = c - y * S;
        if (y == 1 || x == 0 || y == S - 1 || x == S - 1) continue; // :418-421 (row 1, sic)
        bool d = false;
        for (u64 m = dead; m; m &= m - 1) d |= BV(cx, first_bit(m), c) > 0;
        if (!d) continue;
        bool l = false;
        for (u64 m = live; m; m &= m - 1) l |= BV(cx, first_bit(m), c) > 0;
        if (l) continue; // :426 not under a living body
        bool hit = inj ? inj[c] != 0 : cell_u01(seed, call, env_id, purpose, (u32)c) > thr; // :424
        if (hit) { // += 1 then clamp(0,1) (:575,603 / :672,692)
            doubled += (int)(cx.food[c] != 0);
            cx.food[c] = 1;
        }
    }
    wave_lds_sync();
    return wave_sum_i32(doubled);
}

// delete done snakes (:595-596 / :676-677)
__device__ __forceinline__ void delete_done(const Ctx &cx, bool done, bool &has_body, int &hc)
{
    const int lane = cx.lane;
    if (lane < cx.K && done) {
        if (has_body) cx.tclk[lane] = CLOCK_DEAD; // every cell of the grid now reads 0
        has_body = false;
        hc = -1;
    }
    wave_lds_sync();
}

// keeps the 15-bit clocks of long-lived snakes away from the top of their range: ex -= T, T = 0 (values unchanged)
__device__ __forceinline__ bool rebase_clocks(const Ctx &cx)
{
    const int C = cx.C, lane = cx.lane;
    const int myT = lane < cx.K ? cx.tclk[lane] : 0;
    u64 m = ballot(lane < cx.K && myT > CLOCK_REBASE && myT < CLOCK_DEAD);
    if (!m) return false;
    const u64 mine = m;
    while (m) {
        const int s = first_bit(m);
        m &= m - 1;
        unsigned short *b = cx.body + s * C;
        const int T = cx.tclk[s];
        for (int k = 0; k < cx.cpl; ++k) {
            const int c = lane + 64 * k;
            if (c < C) {
                const unsigned short v = b[c];
                if (v & VMASK) b[c] = (unsigned short)((v & DIRTY) | max((int)(v & VMASK) - T, 0));
            }
        }
    }
    wave_lds_sync();
    if ((mine >> lane) & 1) cx.tclk[lane] = 0;
    wave_lds_sync();
    return true;
}

// bit k set <=> cell lane + 64k is interior and has no food, head or body on it (:439-445, :393-399).
// Five rows of 64 cells at a time: each snake's clock and head cell come out of lanes 0 .. K-1 (readlane) and its five body
// cells are read together — one LDS round trip per snake and block.  (Cell by cell — K dependent reads each — this scan was
// most of the 21 000 cycles `_add_food` took of a 51 000-cycle step with random_rate food: tools/multi_timeline.py --rollout.)
__device__ __forceinline__ u64 free_cells(const Ctx &cx, int hc, int margin)
{
    const int S = cx.S, C = cx.C, K = cx.K, lane = cx.lane;
    const int myT = lane < K ? cx.tclk[lane] : 0, myH = lane < K ? hc : -1;
    constexpr int U = 5;
    u64 fr = 0;
    for (int k0 = 0; k0 < cx.cpl; k0 += U) {
        int cc[U];
        u32 taken[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            cc[u] = min(lane + 64 * (k0 + u), C - 1); // (rows past the grid: the last cell again, dropped below)
            taken[u] = cx.food[cc[u]];
        }
        for (int s = 0; s < K; ++s) {
            const int T = lane_value(myT, s), H = lane_value(myH, s);
            const unsigned short *b = cx.body + s * C;
            u32 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = b[cc[u]];
#pragma unroll
            for (int u = 0; u < U; ++u) taken[u] |= (u32)((int)(v[u] & VMASK) > T) | (u32)(cc[u] == H);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = lane + 64 * (k0 + u);
            const int y = div_size(cc[u], cx.rcpS), x = cc[u] - y * S;
            const bool inside = c < C && y >= margin && x >= margin && y <= S - 1 - margin && x <= S - 1 - margin;
            if (inside && taken[u] == 0) fr |= 1ull << (k0 + u);
        }
    }
    return fr;
}

// the K-th set bit over all lanes' `bits` in row-major cell order (cell = lane + 64k): returns true in the
// lane/bit that owns it through `hit_k` (>= 0), -1 elsewhere
__device__ __forceinline__ int rank_select(const Ctx &cx, u64 bits, int K_rank)
{
    int base = 0, hit = -1;
    for (int k = 0; k < cx.cpl; ++k) {
        bool b = (bits >> k) & 1;
        u64 m = ballot(b);
        if (b && base + rank_below(m) == K_rank) hit = k;
        base += popc64(m);
    }
    return hit;
}

// wave-uniform cell index of the (single) lane/bit chosen by rank_select, -1 if none
__device__ __forceinline__ int selected_cell(int k)
{
    u64 m = ballot(k >= 0);
    if (!m) return -1;
    int owner = first_bit(m);
    return owner + 64 * lane_value(k, owner);
}

__device__ __forceinline__ int count_bits(const Ctx &cx, u64 bits)
{
    int n = 0;
    for (int k = 0; k < cx.cpl; ++k) n += popc64(ballot((bits >> k) & 1));
    return n;
}

// number of food cells of the env: four cells per lane and LDS read (the byte grid starts on a 16-byte boundary), one DPP
// sum — the per-cell form (a read, a ballot and a popcount per row of 64 cells) was 3 000 cycles of every step at cfg4.
// Wave-uniform call sites only (wave_sum_i32).
__device__ __forceinline__ int food_count(const Ctx &cx)
{
    const int C = cx.C, nd = (C + 3) >> 2;
    const u32 *f = (const u32 *)cx.food;
    const u32 last = (C & 3) ? (1u << (8 * (C & 3))) - 1u : 0xffffffffu; // (the bytes behind the grid are padding)
    int n = 0;
#pragma unroll 4
    for (int i = cx.lane; i < nd; i += 64) {
        u32 w = f[i];
        if (i == nd - 1) w &= last;
        n += __popc((((w & 0x7f7f7f7fu) + 0x7f7f7f7fu) | w) & 0x80808080u); // non-zero bytes
    }
    return wave_sum_i32(n);
}

// ------------------------------------------------------------------------------------------------ observations

// 'full' observation (_observe_agent :268-281 + _make_generic_rgb :175-192) of every agent from LDS.
// Per row of 64 cells the class of each cell is worked out ONCE (K clock compares, head owner, food); what an agent
// sees that has neither its head nor its body on the cell — border (0,0,0), somebody's head (0,0,192), somebody's body
// (0,0,96), food (255,0,0), background (255,255,255) — is the same for every agent, so rows without any snake cell (most
// of them) store the same three registers K times, and the per-agent priority chain runs only in rows that hold a
// snake.  Stores take the scalar base of (agent, env) plus a 32-bit lane offset.
__device__ __forceinline__ void store_rgb(float *base, u32 off0, u32 off1, u32 off2, float r, float g, float b)
{
    asm volatile("global_store_dword %0, %1, %6\n\tglobal_store_dword %2, %3, %6\n\tglobal_store_dword %4, %5, %6"
                 : : "v"(off0), "v"(r), "v"(off1), "v"(g), "v"(off2), "v"(b), "s"(base) : "memory");
}

__device__ __forceinline__ void observe_full(const Ctx &cx, const MultiArgs &p, float *__restrict__ obs,
                                             long long env, int hc)
{
    const int S = cx.S, C = cx.C, K = cx.K, lane = cx.lane;
    // head owner per cell: hmap[c] = 1 + snake index (consistent states have at most one head per cell)
    if (lane < K && hc >= 0) cx.hmap[hc] = (unsigned char)(lane + 1);
    wave_lds_sync();
    const float G1 = 192.0f / 255.0f, G2 = 96.0f / 255.0f;
    // agent 0's observation of this env (wave-uniform: told to the compiler so that it lives in SGPRs)
    float *const obs_env = (float *)uniform64((long long)(obs + env * p.obs_elems));
    const long long agent_stride = p.N * p.obs_elems;    // to the next agent's
    for (int k = 0; k < cx.cpl; ++k) {
        const int c = lane + 64 * k;
        const bool valid = c < C;
        const int cc = valid ? c : 0;
        const int y = div_size(cc, cx.rcpS), x = cc - y * S;
        const bool edge = y == 0 || x == 0 || y == S - 1 || x == S - 1;
        u64 bm = 0; // bit s: snake s has body on this cell
        for (int s = 0; s < K; ++s) bm |= (u64)((int)(cx.body[s * C + cc] & VMASK) > cx.tclk[s]) << s;
        const int ho = (int)cx.hmap[cc] - 1;
        const bool fd = cx.food[cc] != 0;
        const bool snake_here = valid && !edge && (ho >= 0 || bm != 0);
        const float ro = (edge || snake_here) ? 0.0f : 1.0f;
        const float go = (edge || snake_here || fd) ? 0.0f : 1.0f;
        const float bo = edge ? 0.0f : (ho >= 0 ? G1 : (bm != 0 ? G2 : (fd ? 0.0f : 1.0f)));
        const u32 o0 = (u32)cc * 4u, o1 = (u32)(C + cc) * 4u, o2 = (u32)(2 * C + cc) * 4u;
        const bool row_has_snake = ballot(snake_here) != 0;
        float *base = obs_env;
        if (valid) {
            if (!row_has_snake) {
                for (int a = 0; a < K; ++a, base += agent_stride) store_rgb(base, o0, o1, o2, ro, go, bo);
            } else {
                for (int a = 0; a < K; ++a, base += agent_stride) {
                    float r = ro, g = go, b = bo;
                    if (snake_here && (ho == a || ((bm >> a) & 1))) {
                        // an agent with its own head or body on the cell: the reference's paint order
                        if (ho >= 0 && ho != a) { r = 0.0f; g = 0.0f; b = G1; }             // other head (0,0,192)
                        else if (bm & ~(1ull << a)) { r = 0.0f; g = 0.0f; b = G2; }         // other body (0,0,96)
                        else if (ho == a) { r = 0.0f; g = G1; b = 0.0f; }                   // own head (0,192,0)
                        else { r = 0.0f; g = G2; b = 0.0f; }                                // own body (0,96,0)
                    }
                    store_rgb(base, o0, o1, o2, r, g, b);
                }
            }
        }
    }
    wave_lds_sync();
    if (lane < K && hc >= 0) cx.hmap[hc] = 0;
    wave_lds_sync();
}

// 'full' observation of at most 10 snakes (experiments/speeds.py runs 10), agent-major.  With 4096 envs in flight the row-major order of observe_full
// means ~49 000 concurrent 256-byte write streams (every row of 64 cells touches all K agents' regions, megabytes
// apart); a pure store kernel in that order reaches 3.6 TB/s, agent by agent — one contiguous 3 * S * S float region
// at a time per wave — 4.8 TB/s (tools/microbench/store_pattern.hip).  So the class of every cell is first written to
// LDS as a 16-bit code (K clock compares, head owner, food, border: once per cell), then each agent's three planes are
// produced from the codes (layout: SNAP_OWNER_SHIFT below).
// (Also tried, measured, dropped: issuing these stores in four parts between the phases of the NEXT step, so that the
// store queue would drain while the wave computes — no gain: the waves of a launch stall on the store path together.)
// 16-bit class code of a cell: body mask of snakes 0..9 | (head owner + 1) << 10 | food << 14 | border << 15
constexpr int SNAP_OWNER_SHIFT = 10;

// class code of every cell of the env in LDS -> snap[] (executed by the wave that owns the env's state)
__device__ __forceinline__ void snap_write(const Ctx &cx, int hc, unsigned short *snap)
{
    const int S = cx.S, C = cx.C, K = cx.K, lane = cx.lane;
    if (lane < K && hc >= 0) cx.hmap[hc] = (unsigned char)(lane + 1);
    wave_lds_sync();
    for (int k = 0; k < cx.cpl; ++k) {
        const int c = lane + 64 * k;
        if (c < C) {
            const int y = div_size(c, cx.rcpS), x = c - y * S;
            const bool edge = y == 0 || x == 0 || y == S - 1 || x == S - 1;
            u32 v = 0;
            for (int s = 0; s < K; ++s) v |= (u32)((int)(cx.body[s * C + c] & VMASK) > cx.tclk[s]) << s;
            v |= (u32)cx.hmap[c] << SNAP_OWNER_SHIFT;
            v |= (u32)(cx.food[c] != 0) << 14;
            v |= (u32)edge << 15;
            snap[c] = (unsigned short)v;
        }
    }
    wave_lds_sync();
    if (lane < K && hc >= 0) cx.hmap[hc] = 0;
    wave_lds_sync();
}

// the K agents' observations of one env from its class codes, agent by agent (any wave of the workgroup may run this)
// (k0 .. k1: the rows of 64 cells to write — a workgroup hands out parts of an agent's view: wg_observe_snap)
__device__ __forceinline__ void snap_emit_agent(const Ctx &cx, const MultiArgs &p, float *obs_env,
                                                const unsigned short *snap, int a, int k0 = 0, int k1 = 1 << 30)
{
    const int C = cx.C, lane = cx.lane;
    const float G1 = 192.0f / 255.0f, G2 = 96.0f / 255.0f;
    {
        float *const base = obs_env + a * (p.N * p.obs_elems);
        for (int k = k0; k < min(k1, cx.cpl); ++k) {
            const int c = lane + 64 * k;
            if (c < C) {
                const u32 v = (u32)snap[c];
                const bool edge = (v >> 15) != 0, fd = ((v >> 14) & 1u) != 0;
                const int ho = (int)((v >> SNAP_OWNER_SHIFT) & 15u) - 1;
                const u32 bm = v & ((1u << SNAP_OWNER_SHIFT) - 1u);
                const bool snake_here = !edge && (ho >= 0 || bm != 0);
                // border (0,0,0), somebody's head (0,0,192), somebody's body (0,0,96), food (255,0,0), background white
                float r = (edge || snake_here) ? 0.0f : 1.0f;
                float g = (edge || snake_here || fd) ? 0.0f : 1.0f;
                float b = edge ? 0.0f : (ho >= 0 ? G1 : (bm != 0 ? G2 : (fd ? 0.0f : 1.0f)));
                if (snake_here && (ho == a || ((bm >> a) & 1u))) { // own head or body here: the reference's paint order
                    if (ho >= 0 && ho != a) { r = 0.0f; g = 0.0f; b = G1; }          // other head (0,0,192)
                    else if (bm & ~(1u << a)) { r = 0.0f; g = 0.0f; b = G2; }         // other body (0,0,96)
                    else if (ho == a) { r = 0.0f; g = G1; b = 0.0f; }                // own head (0,192,0)
                    else { r = 0.0f; g = G2; b = 0.0f; }                             // own body (0,96,0)
                }
                store_rgb(base, (u32)c * 4u, (u32)(C + c) * 4u, (u32)(2 * C + c) * 4u, r, g, b);
            }
        }
    }
}

__device__ __forceinline__ void snap_emit(const Ctx &cx, const MultiArgs &p, float *obs_env, const unsigned short *snap)
{
    for (int a = 0; a < cx.K; ++a) snap_emit_agent(cx, p, obs_env, snap, a);
}

__device__ __forceinline__ void observe_full_snap(const Ctx &cx, const MultiArgs &p, float *__restrict__ obs,
                                                  long long env, int hc)
{
    snap_write(cx, hc, cx.snap);
    snap_emit(cx, p, (float *)uniform64((long long)(obs + env * p.obs_elems)), cx.snap);
    wave_lds_sync();
}

// Per-snake state, one snake per lane (lanes 0..K-1), carried through a step / reset / rollout.
struct Snake {
    int hc;           // head cell, -1 = none
    int L;            // length (max body value)
    bool done;
    long long orient; // stored orientation (multi_snake.py:108,494)
    bool boosted;     // boost_this_step of the last step (brightens the snake in partial_n observations)
    short col[3];     // agent colour
    bool cmap_ok = false; // wave-uniform: cx.hmap holds cell_codes of the state as it is (multi_step_body leaves it; a reset voids it)
};

struct StepRes {
    float reward, foodcons;
    bool snakecol, edgecol, all_done;
};

// v / 255.0f, correctly rounded, for the integers a pixel can hold: one multiplication by the rounded reciprocal and one
// Newton step in fused arithmetic give the IEEE quotient for every integer in [0, 70 000) (checked exhaustively against the
// division in exact rational arithmetic: tests/test_div255.py); anything else takes the division itself (~11 instructions, three
// per pixel: a tenth of the VALU work of a step with partial_n observations).
__device__ __forceinline__ float div255(int v)
{
    const float x = (float)v;
    if ((unsigned)v < 70000u) {
        const float rc = 1.0f / 255.0f;
        const float q = x * rc;
        return __fmaf_rn(__fmaf_rn(-q, 255.0f, x), rc, q);
    }
    return x / 255.0f;
}

// 'partial_n' observation (:289-332): the crop of the env image (_get_env_images :194-227) around each living head.
// The image is the same for every observer, and a cell of it can only show a handful of different pixels: background,
// border, food, and per snake its body and its head (brightened while it boosts, :198).  So (round 5):
//   * one pass over the grid gives every cell a one-byte CODE (cell_codes):
//     0 background, 1 border ring, 2 food, 3 + 2s body of snake s, 4 + 2s head of snake s, 255 anything else (several
//     snakes on the cell, food under a snake, a head without its body: hand-made states, heads that have just collided);
//   * lanes 0 .. 2K+2 compute the 2K + 3 pixels once per step with the reference's own arithmetic (pixel_table: the
//     float products, `.short()`, black -> white, `/ 255`) into an LDS table of 16-byte entries;
//   * a window cell is then a byte read, one 16-byte table read and three stores (crop_emit); a cell with code 255 is
//     rendered on the spot (pixel_slow: the sum over the snakes found there, ascending, as :201-205 sums).
// History: round 2 rendered the whole image into LDS (8 bytes per cell) and cropped it; rounds 3-4 computed every (agent,
// window cell) pixel directly — 16 000 of a step's 40 400 cycles at cfg4' (profiles/r04_kernel_timeline.txt), VALU-bound.
constexpr int PC_BG = 0, PC_RING = 1, PC_FOOD = 2, PC_SNAKE0 = 3, PC_COMPLEX = 255;

// One pass over the grid.  Bodies only in the scan — five 64-cell rows at a time so that the five reads of a snake are in
// flight together, and per (cell, snake) just a mask, a compare, a select and a carry-add (the compiler must keep this loop
// free of branches: check the ISA after touching it) — the K head cells are raised afterwards by the K lanes that own them:
// a head sits on its own body in every state the dynamics produce; anywhere else the cell is marked complex.
// The same map answers _add_food's "free interior cell" (code 0: free_from_codes), so a step with random_rate food and
// crops scans the grids once, not twice (multi_step_body builds it, observe_partial reuses it: Snake::cmap_ok).
__device__ __forceinline__ u64 free_from_codes(const Ctx &cx, const unsigned char *codes);
struct CellCounts {
    u64 free;   // bit k <=> cell lane + 64 k is a free interior cell (code 0), as free_cells(cx, hc, 1) / free_from_codes
    int nfree;  // their number over the wave (count_bits)
    int nfood;  // cells that hold food, whatever else is on them (food_count)
};

__device__ __forceinline__ CellCounts cell_codes(const Ctx &cx, int hc, unsigned char *codes, u64 ring)
{
    const int C = cx.C, K = cx.K, lane = cx.lane;
    constexpr int U = 5;
    const int myT = lane < K ? cx.tclk[lane] : 0;
    CellCounts cc_out;
    cc_out.free = 0;
    cc_out.nfree = cc_out.nfood = 0;
    for (int k0 = 0; k0 < cx.cpl; k0 += U) {
        int cc[U];
        u32 fd[U], code[U], cnt[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            cc[u] = min(lane + 64 * (k0 + u), C - 1); // (rows past the grid: the last cell again, not stored)
            fd[u] = cx.food[cc[u]];
            code[u] = PC_BG;
            cnt[u] = 0;
        }
        for (int s = 0; s < K; ++s) {
            const int T = lane_value(myT, s);
            const u32 mine = (u32)(PC_SNAKE0 + 2 * s);
            const unsigned short *b = cx.body + s * C;
            u32 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = b[cc[u]];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool live = (int)(v[u] & VMASK) > T;
                code[u] = live ? mine : code[u];
                cnt[u] += (u32)live;
            }
        }
        const u32 rb = (u32)(ring >> k0);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            u32 c1 = cnt[u] > 1u ? (u32)PC_COMPLEX : code[u];             // two bodies on the cell
            c1 = fd[u] != 0 ? (c1 == PC_BG ? (u32)PC_FOOD : (u32)PC_COMPLEX) : c1; // food; food under a body
            c1 = ((rb >> u) & 1u) ? (u32)PC_RING : c1;                    // :225 the border wins over everything
            const bool valid = lane + 64 * (k0 + u) < C;
            if (valid) codes[cc[u]] = (unsigned char)c1;
            // the counts _add_food needs come out of the same pass (the compares' own wave masks, counted on the scalar unit)
            const bool fr = valid && c1 == PC_BG;
            cc_out.free |= (u64)fr << (k0 + u);
            cc_out.nfree += popc64(ballot(fr));
            cc_out.nfood += popc64(ballot(valid && fd[u] != 0));
        }
    }
    wave_lds_sync();
    // heads: the cell must hold its own snake's body (code 3 + 2s) and becomes 4 + 2s; anything else under a head — no
    // body, food, another snake's body, the ring is fine (it wins anyway) — is complex.  Two heads on one cell: at most one of
    // them finds its own code there; the other lanes all write the same value.
    int want = -1;
    if (lane < K && hc >= 0) {
        const int code = codes[hc];
        want = code == PC_RING ? -1 : code == PC_SNAKE0 + 2 * lane ? PC_SNAKE0 + 2 * lane + 1 : PC_COMPLEX;
    }
    wave_lds_sync();
    if (want == PC_COMPLEX) codes[hc] = (unsigned char)PC_COMPLEX;
    wave_lds_sync();
    if (want >= 0 && want != PC_COMPLEX && codes[hc] != PC_COMPLEX) codes[hc] = (unsigned char)want;
    wave_lds_sync();
    if (ballot(want == PC_COMPLEX) != 0) { // (a head on a cell without its own body may have stood on a "free" cell: hand-made states)
        cc_out.free = free_from_codes(cx, codes);
        cc_out.nfree = count_bits(cx, cc_out.free);
    }
    return cc_out;
}

// bit k <=> cell lane + 64 k is a free interior cell (:439-445, :393-399): code 0 of cell_codes
__device__ __forceinline__ u64 free_from_codes(const Ctx &cx, const unsigned char *codes)
{
    u64 fr = 0;
    for (int k = 0; k < cx.cpl; ++k) {
        const int c = cx.lane + 64 * k;
        if (c < cx.C && codes[c] == PC_BG) fr |= 1ull << k;
    }
    return fr;
}

// food on a cell (wave-uniform or per-lane `cell`; the caller fences), and its code with it where the map is current
__device__ __forceinline__ void put_food(const Ctx &cx, int cell, bool cmap_ok)
{
    cx.food[cell] = 1;
    if (cmap_ok) {
        const int code = cx.hmap[cell];
        cx.hmap[cell] = (unsigned char)(code == PC_BG || code == PC_FOOD ? PC_FOOD : code == PC_RING ? PC_RING : PC_COMPLEX);
    }
}

// the reference's pixel of a cell from the float sums of :201-205: .short() truncation, food, black -> white (:206-219)
__device__ __forceinline__ void pixel_finish(float a0, float a1, float a2, bool food, float &r, float &g, float &b)
{
    int ri = (int)a0, gi = (int)a1, bi = (int)a2;         // :206 .short() truncates
    if (food) ri += 255;                                  // :208-209
    if (ri == 0 && gi == 0 && bi == 0) ri = gi = bi = 255; // :214-219
    r = div255(ri);
    g = div255(gi);
    b = div255(bi);
}

// the 2K + 3 pixels a cell with a simple code can show -> tab[code] = (r, g, b, -); colf[s] = (colour, 1 + 0.5 boost) is
// in LDS already.  Lane j computes entry j (K <= 64: two rounds at most).
__device__ __forceinline__ void pixel_table(const Ctx &cx, float *tab)
{
    const int K = cx.K;
    for (int j = cx.lane; j < 2 * K + PC_SNAKE0; j += 64) {
        float r = 0.0f, g = 0.0f, b = 0.0f;                // PC_RING (:225)
        if (j != PC_RING) {
            float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
            if (j >= PC_SNAKE0) {
                const int s = (j - PC_SNAKE0) >> 1;
                // :197 `body.float() * 1/3 + head.float() * 1/3`: 1.0f / 3.0f is the correctly rounded quotient the two
                // IEEE divisions gave; then the boost factor, then the colour (:198-205), summed from 0 as torch's sum does
                const float third = 1.0f / 3.0f;
                float inten = third + (((j - PC_SNAKE0) & 1) ? third : 0.0f);
                inten *= cx.colf[s * 4 + 3];
                a0 += inten * cx.colf[s * 4 + 0];
                a1 += inten * cx.colf[s * 4 + 1];
                a2 += inten * cx.colf[s * 4 + 2];
            }
            pixel_finish(a0, a1, a2, j == PC_FOOD, r, g, b);
        }
        tab[4 * j + 0] = r;
        tab[4 * j + 1] = g;
        tab[4 * j + 2] = b;
    }
}

// a cell whose code is PC_COMPLEX, rendered from the grids (interior cells only: the ring has its own code)
__device__ __forceinline__ void pixel_slow(const Ctx &cx, int c, float &r, float &g, float &b)
{
    const int C = cx.C, K = cx.K;
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
    for (int s = 0; s < K; ++s) {
        const bool isb = (int)(cx.body[s * C + c] & VMASK) > cx.tclk[s], ish = cx.hcell[s] == c;
        if (isb || ish) { // a snake that is not on the cell adds inten = 0, i.e. nothing
            const float third = 1.0f / 3.0f;
            float inten = (isb ? third : 0.0f) + (ish ? third : 0.0f);
            inten *= cx.colf[s * 4 + 3];
            a0 += inten * cx.colf[s * 4 + 0];
            a1 += inten * cx.colf[s * 4 + 1];
            a2 += inten * cx.colf[s * 4 + 2];
        }
    }
    pixel_finish(a0, a1, a2, cx.food[c] != 0, r, g, b);
}

__device__ __forceinline__ void observe_partial(const Ctx &cx, const MultiArgs &p, float *__restrict__ obs,
                                                long long env, const Snake &sn)
{
    const int S = cx.S, K = cx.K, lane = cx.lane, n = p.obs_n;
    unsigned char *const codes = cx.hmap;      // (the head map of observe_full: free in a launch that writes crops)
    float *const tab = (float *)cx.img;        // 16 bytes per code (multi_layout: need_img)
    if (lane < K) {
        cx.colf[lane * 4 + 0] = (float)sn.col[0];
        cx.colf[lane * 4 + 1] = (float)sn.col[1];
        cx.colf[lane * 4 + 2] = (float)sn.col[2];
        // :198 the brightening of a boosting snake
        cx.colf[lane * 4 + 3] = 1.0f + 0.5f * (sn.boosted ? 1.0f : 0.0f);
        cx.hcell[lane] = sn.hc;
    }
    wave_lds_sync();
    pixel_table(cx, tab);
    if (!sn.cmap_ok) cell_codes(cx, sn.hc, codes, cx.has_ring ? cx.ring : border_bits(cx));
    wave_lds_sync(); // (the table)
    WURM_TLS(cx, 13);
    const int W = 2 * n + 1, W2 = W * W;
    const float rcpW = 1.0f / (float)W;
    const long long agent_stride = p.N * p.obs_elems;
    float *const o_env = obs + env * p.obs_elems;
    const u64 dead = ballot(lane < K && sn.done); // a dead observer sees zeros (:320-323)
    for (int w = lane; w < W2; w += 64) {
        // the lane's window cell relative to the observer's head: the same for every agent
        const int wy = div_size(w, rcpW), wx = w - wy * W, dy = wy - n, dx = wx - n;
        for (int a = 0; a < K; ++a) {
            float *const o = (float *)uniform64((long long)(o_env + (long long)a * agent_stride));
            const int h = ((dead >> a) & 1ull) ? -1 : lane_value(sn.hc, a);
            float r = 0.0f, g = 0.0f, b = 0.0f; // a dead observer; the zero padding (:302)
            if (h >= 0) {
                const int hy = div_size(h, cx.rcpS), hx = h - hy * S; // (uniform)
                const int y = hy + dy, x = hx + dx;
                if (y >= 0 && y < S && x >= 0 && x < S) {
                    const int c = y * S + x;
                    const int code = codes[c];
                    if (code == PC_COMPLEX) pixel_slow(cx, c, r, g, b);
                    else {
                        const float4 t = *(const float4 *)(tab + 4 * code);
                        r = t.x; g = t.y; b = t.z;
                    }
                }
            }
            o[w] = r;
            o[W2 + w] = g;
            o[2 * W2 + w] = b;
        }
    }
    wave_lds_sync();
}

__device__ __forceinline__ void observe(const Ctx &cx, const MultiArgs &p, float *__restrict__ obs, long long env,
                                        const Snake &sn)
{
    if (p.obs_mode == WURM_OBS_DEFAULT) {
        if (p.off_snap >= 0) observe_full_snap(cx, p, obs, env, sn.hc);
        else observe_full(cx, p, obs, env, sn.hc);
    }
    else if (p.obs_mode == WURM_OBS_PARTIAL) observe_partial(cx, p, obs, env, sn);
}

__device__ __forceinline__ void load_colour(const MultiArgs &p, long long agent, bool active, Snake &sn)
{
    sn.col[0] = sn.col[1] = sn.col[2] = 0;
    if (active && p.colours) {
        sn.col[0] = p.colours[agent * 3];
        sn.col[1] = p.colours[agent * 3 + 1];
        sn.col[2] = p.colours[agent * 3 + 2];
    }
}

// ------------------------------------------------------------------------------------------------ step

// MultiSnake.step (:462-731) of the env held in LDS.  `a` is this lane's (snake's) action.  offC / offA / offE are the
// offsets of this step's slice in the injected-outcome arrays (0 for a single step; t*N*C, t*N*K, t*N in a rollout).
__device__ __forceinline__ void multi_step_body(const Ctx &cx, const MultiArgs &p, long long env, u64 env_id, u64 call,
                                                long long a, Snake &sn, StepRes &res, long long offC, long long offA,
                                                long long offE)
{
    const int C = cx.C, K = cx.K, lane = cx.lane;
    const bool snake = lane < K;
    int hc = sn.hc, L = sn.L;
    bool done = sn.done;
    bool has_body = L > 0;
    const long long agent = env * K + lane;

    // prologue (:475-502)
    const bool done0 = done;
    long long orient = sn.orient;
    long long d = a % 4;                        // :483
    const bool boost_act = a > 3;               // :484
    if (orient == d) d = (d + 2) % 4;           // :493 sanitize_movements
    orient = (d + 2) % 4;                       // :494
    const int dir = (int)(((d % 4) + 4) % 4);
    const bool boosted = snake && boost_act && L >= 4; // :497-499
    float reward = 0.0f, foodcons = 0.0f;
    bool snakecol = false, edgecol = false;

    if (p.cfg.boost && ballot(boosted) != 0) {  // :503 (per env; the batch-global gate has no per-env effect)
        run_phase(cx, boosted, dir, hc, L, done, reward, foodcons, snakecol, edgecol);
        if (p.cfg.food_on_death)                // :565-576
            food_from_death(cx, done, has_body, p.has_inj ? p.inj.death_a + offC + env * C : nullptr,
                            p.cfg.death_threshold, p.seed, call, env_id, RNG_DEATH_FOOD_A);
        // boost cost (:579-592): tail cell becomes food, body decays, reward -1
        bool pay = false;
        if (boosted) {
            if (p.has_inj) pay = p.inj.cost[offA + agent] != 0;
            else pay = u01(rng_words(p.seed, call, env_id, RNG_BOOST_COST, (u32)lane).w[0]) < p.cfg.boost_cost_prob;
        }
        u64 m = ballot(pay);
        while (m) { // the tail cell (value 1) becomes food; the decay itself is the clock
            int s = first_bit(m);
            m &= m - 1;
            const unsigned short *b = cx.body + s * C;
            const int tail = cx.tclk[s] + 1;
            for (int k = 0; k < cx.cpl; ++k) {
                int c = lane + 64 * k;
                if (c < C && (int)(b[c] & VMASK) == tail) cx.food[c] = 1;
            }
        }
        if (pay) {
            cx.tclk[lane] += 1;
            reward -= 1.0f;
            L -= 1;
        }
        wave_lds_sync();
        delete_done(cx, done, has_body, hc);    // :595-596
    }

    WURM_TLS(cx, 3);
    run_phase(cx, snake, dir, hc, L, done, reward, foodcons, snakecol, edgecol); // :613-660
    WURM_TLS(cx, 4);
    int doubled_b = 0;                          // (food cells the sum of :382 sees twice: food_from_death)
    if (p.cfg.food_on_death)                    // :662-673
        doubled_b = food_from_death(cx, done, has_body, p.has_inj ? p.inj.death_b + offC + env * C : nullptr,
                                    p.cfg.death_threshold, p.seed, call, env_id, RNG_DEATH_FOOD_B);
    delete_done(cx, done, has_body, hc);        // :676-677
    WURM_TLS(cx, 5);

    // _add_food (:368-410)
    bool cmap_ok = false;
    {
        // The map of cell codes (cell_codes: one scan of the K grids) serves both "which interior cells are free" here and the
        // crops of observe_partial, and counts the food and the free cells on the way; it lives where observe_full keeps its
        // head map, so only launches without 'full' observations build it — a launch that writes crops every step, one
        // without observations when food has to be placed.
        CellCounts cc;
        cc.free = 0;
        cc.nfree = -1;
        int nfood;
        if (p.obs_mode == WURM_OBS_PARTIAL && cx.has_ring) {
            cc = cell_codes(cx, hc, cx.hmap, cx.ring);
            cmap_ok = true;
            nfood = cc.nfood + doubled_b;
        } else {
            nfood = food_count(cx) + doubled_b;
            const bool want_free = p.cfg.food_mode == 0 ? (nfood == 0 && !p.has_inj) : nfood < p.cfg.max_food;
            if (p.obs_mode == WURM_OBS_NONE && cx.has_ring && want_free) {
                cc = cell_codes(cx, hc, cx.hmap, cx.ring);
                cmap_ok = true;
            }
        }
        WURM_TLS(cx, 10);
        if (p.cfg.food_mode == 0) {
            if (nfood == 0) {                   // :371-379
                if (p.has_inj) {
                    int cell = p.inj.food_cell[offE + env];
                    if (cell >= 0 && cell < C && lane == 0) put_food(cx, cell, cmap_ok);
                } else {
                    u64 fr = cmap_ok ? cc.free : free_cells(cx, hc, 1);
                    int nf = cmap_ok ? cc.nfree : count_bits(cx, fr);
                    if (nf > 0) {
                        int Kr = (int)mulhi_range(rng_words(p.seed, call, env_id, RNG_FOOD, 0).w[0], (u32)nf);
                        int k = rank_select(cx, fr, Kr);
                        if (k >= 0) put_food(cx, lane + 64 * k, cmap_ok);
                    }
                }
            }
        } else if (nfood < p.cfg.max_food) {    // :382-408
            u64 fr = cmap_ok ? cc.free : free_cells(cx, hc, 1);
            WURM_TLS(cx, 11);
            if (p.has_inj) {
                for (int k = 0; k < cx.cpl; ++k)
                    if (((fr >> k) & 1) && p.inj.rate[offC + env * C + lane + 64 * k] != 0) put_food(cx, lane + 64 * k, cmap_ok);
            } else {
                // Every free cell spawns food independently with probability food_rate (:401-408).  RNG mode draws the NUMBER
                // of cells — Binomial(n free, food_rate) by inversion from ONE uniform — and then that many distinct cells,
                // the j-th as the mulhi(word, n - j)-th remaining free cell in row-major order: the same distribution as n
                // independent draws (this build's own RNG specification, oracle/multi_snake.c step_env), for one Philox block
                // per env-step instead of one per four cells and lane (625 draws to place 0.13 foods on average at cfg4').
                const int nf = cmap_ok ? cc.nfree : count_bits(cx, fr);
                const float pw = pow_n(1.0f - p.cfg.food_rate, nf);
                if (!(p.cfg.food_rate > 0.0f) || pw < BINOMIAL_MIN_P0) {
                    // P(no food) too small for the recurrence (rates far above the reference's): cell by cell — the cells
                    // lane + 64k, k = 4j .. 4j+3, share Philox block j of this lane (word k & 3)
                    for (int j = 0; 4 * j < cx.cpl; ++j) {
                        const u32 four = (u32)(fr >> (4 * j)) & 15u;
                        if (!four) continue;
                        const Words w = rng_words(p.seed, call, env_id, RNG_RATE_FOOD, ((u32)j << 6) | (u32)lane);
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (((four >> q) & 1u) && u01(w.w[q]) < p.cfg.food_rate) put_food(cx, lane + 64 * (4 * j + q), cmap_ok);
                    }
                } else {
                    Words w = rng_words(p.seed, call, env_id, RNG_RATE_FOOD, 0);
                    const int kf = uniform(binomial_inverse(nf, p.cfg.food_rate, pw, u01(w.w[0])));
                    WURM_TLS(cx, 12);
                    for (int j = 0; j < kf; ++j) {
                        const int wi = j + 1;
                        if ((wi & 3) == 0) w = rng_words(p.seed, call, env_id, RNG_RATE_FOOD, (u32)(wi >> 2));
                        const u32 wj = (wi & 3) == 0 ? w.w[0] : (wi & 3) == 1 ? w.w[1] : (wi & 3) == 2 ? w.w[2] : w.w[3];
                        const int Kr = (int)mulhi_range(wj, (u32)(nf - j));
                        const int k = rank_select(cx, fr, Kr);
                        if (k >= 0) {
                            put_food(cx, lane + 64 * k, cmap_ok);
                            fr &= ~(1ull << k);
                        }
                    }
                }
            }
        }
        wave_lds_sync();
    }

    WURM_TLS(cx, 6);
    if (snake && done && !done0) reward += p.cfg.reward_on_death; // :683-685

    sn.hc = hc;
    sn.L = L;
    sn.done = done;
    sn.orient = orient;
    sn.boosted = boosted;
    sn.cmap_ok = cmap_ok;
    res.reward = reward;
    res.foodcons = foodcons;
    res.snakecol = snakecol;
    res.edgecol = edgecol;
    res.all_done = ballot(snake && !done) == 0; // :703
    if (snake) cx.hcell[lane] = hc;
    wave_lds_sync();
}

// defined in the reset section below; multi_step_kernel applies a postponed reset in front of its transition
__device__ __forceinline__ bool reroll_colour(const MultiArgs &p, long long agent, bool dead, u64 env_id, u64 call,
                                              long long offA, Snake &sn);
__device__ __forceinline__ void multi_reset_grid(const Ctx &cx, const MultiArgs &p, long long env, u64 env_id, u64 call,
                                                 bool rebuild, bool respawn, Snake &sn, bool &orient_dirty,
                                                 long long offA, long long offE);

// check_consistency (:733-769) of the env as it sits in LDS — the masks of multi_check_kernel, for an image that came from
// the mirror or from a rebuild (16-bit clocks, one head cell per snake, food bytes 0 / 1: what the fp32 planes could
// additionally hold — several heads, fractional values — cannot occur there).  ONE wave; sn = the snakes' scalars.
// one snake's share: its verdict bits; occ / over collect the overlap test (bit 8 r + j of a lane's occ: cell
// 512 r + 8 lane + j holds a body value of a snake seen so far; at most 8 runs: C <= 4096)
__device__ __forceinline__ uint32_t lds_check_snake(const Ctx &cx, int s, int hc, bool dead, u64 &occ, int &over)
{
    const int C = cx.C, lane = cx.lane;
    // lane l owns cells 8 l .. 8 l + 7 of every run of 512: one 16-byte LDS read per run and snake.  The body grids start
    // on a 16-byte boundary (multi_layout) and snake s's at 2 s C bytes behind it: aligned for every s only if C is a
    // multiple of 8 — else the cells are read one by one
    const int runs = (C + 511) >> 9;
    const bool wide = (C & 7) == 0;
    const int T = cx.tclk[s];
    const unsigned short *b = cx.body + s * C;
    uint32_t m = 0;
    int bs = 0, bm = 0;
    for (int r = 0; r < runs; ++r) {
        const int c0 = 512 * r + 8 * lane;
        u32 w0 = 0, w1 = 0, w2 = 0, w3 = 0; // (scalars, not an array: an indexed local array ends up in scratch)
        if (wide) {
            if (c0 < C) {
                const uint4 q = *(const uint4 *)(b + c0);
                w0 = q.x; w1 = q.y; w2 = q.z; w3 = q.w;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (c0 + j < C) {
                    const u32 v16 = (u32)b[c0 + j] << (16 * (j & 1));
                    if ((j >> 1) == 0) w0 |= v16;
                    else if ((j >> 1) == 1) w1 |= v16;
                    else if ((j >> 1) == 2) w2 |= v16;
                    else w3 |= v16;
                }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const u32 word = (j >> 1) == 0 ? w0 : (j >> 1) == 1 ? w1 : (j >> 1) == 2 ? w2 : w3;
            const int v = max((int)((word >> (16 * (j & 1))) & VMASK) - T, 0);
            bs += v;
            bm = max(bm, v);
            const u64 bit = 1ull << (8 * r + j);
            if (v > 0) {
                over |= (int)((occ & bit) != 0);
                occ |= bit;
            }
        }
    }
    const int t_bs = wave_sum_i32(bs), t_bm = wave_max_i32(bm);
    if (dead) {
        if (t_bs > 0 || hc >= 0) m |= WURM_MCHK_DEAD_NONZERO;
    } else {
        const int t_hb = hc >= 0 ? BV(cx, s, hc) : 0, t_hf = hc >= 0 ? (int)cx.food[hc] : 0;
        if (hc < 0) m |= WURM_CHK_ONE_HEAD;
        if (!(t_bs > 0)) m |= WURM_CHK_HAS_SNAKE;
        if (t_bm != t_hb) m |= WURM_CHK_HEAD_AT_END;
        if (2 * t_bs != t_bm * (t_bm + 1)) m |= WURM_CHK_BODY_RANGE;
        if (!(t_bs >= 6)) m |= WURM_CHK_MIN_LENGTH;
        if (t_hf != 0) m |= WURM_CHK_HEAD_ON_FOOD;
    }
    return m;
}

__device__ __forceinline__ uint32_t lds_check(const Ctx &cx, const Snake &sn)
{
    uint32_t m = 0;
    u64 occ = 0;
    int over = 0;
    for (int s = 0; s < cx.K; ++s)
        m |= lds_check_snake(cx, s, lane_value(sn.hc, s), lane_value((int)sn.done, s) != 0, occ, over);
    if (ballot(over != 0)) m |= WURM_MCHK_OVERLAP;
    return m;
}

// The same by the `nw` waves of a workgroup (multi_step_wg_kernel): wave w takes snakes w, w + nw, ...; head cells from
// cx.hcell, the done flags from cx.lmax (the caller parks them there); `scratch`: 8 * 64 * nw + 8 * nw bytes of LDS that
// nobody else uses right now.  Every thread of the workgroup calls it; the same value comes back in all of them.
__device__ __forceinline__ uint32_t wg_lds_check(const Ctx &cx, int wave, int nw, unsigned char *scratch)
{
    const int lane = cx.lane;
    u64 *occs = (u64 *)scratch;                  // [nw][64]
    u32 *flags = (u32 *)(scratch + 512 * nw);    // [nw][2]: verdict bits, overlap inside the wave's own snakes
    uint32_t m = 0;
    u64 occ = 0;
    int over = 0;
    for (int s = wave; s < cx.K; s += nw) m |= lds_check_snake(cx, s, cx.hcell[s], cx.lmax[s] != 0, occ, over);
    occs[wave * 64 + lane] = occ;
    if (lane == 0) { flags[2 * wave] = m; flags[2 * wave + 1] = 0; }
    if (ballot(over != 0) && lane == 0) flags[2 * wave + 1] = 1;
    __syncthreads();
    uint32_t total = 0;
    u64 seen = 0;
    int cross = 0;
    for (int w = 0; w < nw; ++w) {
        total |= flags[2 * w] | (flags[2 * w + 1] ? WURM_MCHK_OVERLAP : 0u);
        const u64 o = occs[w * 64 + lane];
        cross |= (int)((seen & o) != 0);
        seen |= o;
    }
    if (ballot(cross != 0)) total |= WURM_MCHK_OVERLAP;
    __syncthreads();
    return total;
}

constexpr uint32_t MCHK_NOT_COMPUTED = 0xffffffffu; // read from fp32 planes that hold what the image cannot: run multi_check_kernel

// The part of the per-call step between the load and the store of the env (LDS state ready, hcell / lmax / tclk set):
// [the reset(done) the caller postponed, exactly multi_reset_kernel without observation, with its own counter,] the
// transition, and the per-agent outputs.  Runs on ONE wave.  hc0: the head cells HBM holds (sparse write-back).
// what step_middle reads from global memory before anything else: a kernel requests it at its entry, together with the
// env's image, so that the transition does not start with a memory round trip of its own (2 400 of a stepper's 67 000 cycles
// per call at cfg4: tools/multi_timeline.py)
struct StepIn {
    bool done;
    long long orient, a;
    short col[3];
};

__device__ __forceinline__ void step_inputs(const MultiArgs &p, long long env, int lane, StepIn &in)
{
    const bool snake = lane < p.K;
    const long long agent = env * p.K + lane;
    in.done = snake ? p.dones[agent] != 0 : true;
    in.orient = snake ? p.orientations[agent] : 0;
    in.a = snake ? p.actions[(long long)lane * p.N + env] : 0;
    Snake c;
    load_colour(p, agent, snake && (p.done_env != nullptr || p.obs_mode == WURM_OBS_PARTIAL), c);
    in.col[0] = c.col[0]; in.col[1] = c.col[1]; in.col[2] = c.col[2];
}

__device__ __forceinline__ void step_middle(const Ctx &cx, const MultiArgs &p, long long env, bool rebuild, Snake &sn,
                                            StepRes &r, int &hc0, const StepIn *in = nullptr)
{
    const int K = cx.K, lane = cx.lane;
    const bool snake = lane < K;
    const u64 env_id = (u64)(p.env_offset + env);
    const long long agent = env * K + lane;
    sn.hc = snake ? cx.hcell[lane] : -1;
    sn.L = snake ? cx.lmax[lane] : 0;
    sn.done = in ? in->done : snake ? p.dones[agent] != 0 : true;
    sn.orient = in ? in->orient : snake ? p.orientations[agent] : 0;
    sn.boosted = false;
    hc0 = sn.hc;
    if (p.done_env != nullptr) {
        if (rebuild) sn.done = false; // :798
        if (in) { sn.col[0] = in->col[0]; sn.col[1] = in->col[1]; sn.col[2] = in->col[2]; }
        else load_colour(p, agent, snake, sn);
        if (snake && reroll_colour(p, agent, sn.done, env_id, p.pre_call, 0, sn)) {
            p.colours[agent * 3] = sn.col[0];
            p.colours[agent * 3 + 1] = sn.col[1];
            p.colours[agent * 3 + 2] = sn.col[2];
        }
        const bool respawn = p.cfg.respawn_any && ballot(snake && sn.done) != 0;
        if (rebuild || respawn) {
            bool orient_dirty = false;
            multi_reset_grid(cx, p, env, env_id, p.pre_call, rebuild, respawn, sn, orient_dirty, 0, 0);
        }
    } else {
        if (in) { sn.col[0] = in->col[0]; sn.col[1] = in->col[1]; sn.col[2] = in->col[2]; }
        else load_colour(p, agent, snake && p.obs_mode == WURM_OBS_PARTIAL, sn);
    }
    const long long a = in ? in->a : snake ? p.actions[(long long)lane * p.N + env] : 0;
    WURM_TLS(cx, 2);
    multi_step_body(cx, p, env, env_id, p.call, a, sn, r, 0, 0, 0);
    WURM_TLS(cx, 7);

    // outputs (:701-729)
    if (snake) {
        p.dones[agent] = (uint8_t)sn.done;
        p.orientations[agent] = sn.orient;
        p.boost[agent] = (uint8_t)sn.boosted;
        p.rewards[agent] = r.reward;
        p.snakecol[agent] = (uint8_t)r.snakecol;
        p.edgecol[agent] = (uint8_t)r.edgecol;
        p.foodcons[agent] = r.foodcons;
        p.sizes[agent] = (float)sn.L;
        if (p.am_f32) { // agent-major copies: row i = agent_i over all envs
            const long long KN = (long long)K * p.N, am = (long long)lane * p.N + env;
            p.am_f32[am] = r.reward;
            p.am_f32[KN + am] = r.foodcons;
            p.am_f32[2 * KN + am] = (float)sn.L;
            p.am_u8[am] = (uint8_t)sn.done;
            p.am_u8[KN + am] = (uint8_t)sn.boosted;
            p.am_u8[2 * KN + am] = (uint8_t)r.snakecol;
            p.am_u8[3 * KN + am] = (uint8_t)r.edgecol;
        }
    }
    if (lane == 0) {
        p.all_done[env] = (uint8_t)r.all_done;
        if (p.all_done_copy) p.all_done_copy[env] = (uint8_t)r.all_done;
    }
}

// What reset(dones['__all__']) will do (multi_reset_kernel with done_env = all_done, call + 1), applied to the LDS copy
// only: the caller postpones that reset into the next launch, which recreates it from the same counters.  ONE wave.
// Returns whether the env was rebuilt or a snake respawned (else the grids are as the step left them).
__device__ __forceinline__ bool reset_for_obs_after(const Ctx &cx, const MultiArgs &p, long long env, Snake &sn,
                                                    const StepRes &r)
{
    const bool snake = cx.lane < cx.K;
    const u64 env_id = (u64)(p.env_offset + env);
    const bool rebuild_after = r.all_done;
    if (rebuild_after) sn.done = false; // :798
    // :800-803 the colours of snakes that are still dead are re-rolled (registers only; they matter to 'partial_n')
    if (p.obs_mode == WURM_OBS_PARTIAL)
        reroll_colour(p, env * cx.K + cx.lane, snake && sn.done, env_id, p.call + 1ull, 0, sn);
    const bool respawn_after = p.cfg.respawn_any && ballot(snake && sn.done) != 0;
    if (rebuild_after || respawn_after) {
        bool orient_dirty = false;
        multi_reset_grid(cx, p, env, env_id, p.call + 1ull, rebuild_after, respawn_after, sn, orient_dirty, 0, 0);
    }
    return rebuild_after || respawn_after;
}

// (LDS written by some waves of a workgroup and read by others)
__device__ __forceinline__ void workgroup_handoff()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// the pieces of the grouped 'full' observation writer (defined with multi_rollout_group_kernel below)
__device__ __forceinline__ u64 border_bits(const Ctx &cx);
template <typename CT>
__device__ __forceinline__ void class_write(const Ctx &cx, int hc, CT *codes, u64 border);
__device__ __forceinline__ void grp_table_init(float *tab, int tid);
__device__ __forceinline__ void grp_emit_group(const MultiArgs &p, float *obs, long long env0, int nG, int wave, int nwaves,
                                               const unsigned char *codes0, int code_stride, const float *tab, int lane);

// `p.grp_emit` (large batches, 'full' observations of at most 5 snakes, several envs per workgroup): the waves of the
// workgroup first step their envs, then write the observations TOGETHER — wave w writes agent w's view of all the
// workgroup's envs, one linear run (class codes + colour table: see multi_rollout_group_kernel).
// INJ / OBS: as multi_rollout_kernel — a launch that draws its own random outcomes, with the observation mode a constant.
// (a shape-specialised instantiation keeps to 128 VGPRs: 4 096 envs are one round of waves at 4 per SIMD, not two at 3)
template <bool INJ = true, int OBS = -1, int KT = 0, int ST = 0, int NT = -1>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(KT > 0 ? 4 : 1))) void multi_step_kernel(MultiArgs p_in)
{
    MultiArgs p = p_in;
    if (!INJ) p.has_inj = p.has_rinj = 0;
    if (OBS >= 0) p.obs_mode = OBS;
    shape_constants<OBS, KT, ST, NT>(p, true);
    const int wave = (int)(threadIdx.x >> 6), wpb = (int)(blockDim.x >> 6);
    const long long env0 = xcd_block(blockIdx.x, gridDim.x) * wpb, env = env0 + wave;
    const bool grouped = p.grp_emit != 0;
    float *const tab = (float *)(wurm_multi_lds + p.grp_env0); // (grouped: the colour table behind the envs' blocks)
    const bool solo = p.grp_emit == 2; // every wave writes its own env's views
    if (grouped) {
        if (env0 >= p.N) return;
        grp_table_init(tab, (int)threadIdx.x);
        // (solo: the table is handed over HERE, where the workgroup's waves still run together — a barrier behind the step
        // would wait for the slowest of its envs)
        if (solo) workgroup_handoff();
    } else if (env >= p.N) return;
    const bool active = env < p.N;
    if (solo && !active) return;
    const Ctx cx = make_ctx(p, wave, 0, p.obs_mode != WURM_OBS_DEFAULT); // (the ring bits: multi_step_body's map of cell codes)
    const int C = cx.C, K = cx.K, lane = cx.lane;
    const bool snake = lane < K;
    const int nG = (int)min((long long)wpb, p.N - env0);
    WURM_TLS_INIT(cx);
    WURM_TLS(cx, 0);
    const u64 ring = grouped ? border_bits(cx) : 0ull;
    Snake sn;
    StepRes r;
    bool clean = false;
    uint32_t m_step = MCHK_NOT_COMPUTED;
    float *foodp = nullptr, *headp = nullptr, *bodyp = nullptr;
    const bool mirrored = p.resident != nullptr, lazy = mirrored && p.resident_lazy != 0;
    const bool from_mirror = mirrored && p.resident_valid != 0;
    unsigned char *mp = nullptr;
    auto fence = [] { wave_lds_sync(); };
    u64 fbits0 = 0;
    bool rebuild = false;
    int t0 = 0, hc0 = -1;
    // the stepped state back to memory: the check mask, the fp32 planes (unless the mirror is lazy), the mirror.  In front of
    // the observation everywhere but in the solo form, where it follows it: nothing the observation needs depends on it, and
    // in a launch that is a chain of latencies (section 4.11 item 7 of DESIGN.md) it was 5 900 cycles in front of the first
    // observation store.
    auto store_state = [&]() {
        if (p.err != nullptr) {
            if (clean) m_step = lds_check(cx, sn);
            if (lane == 0) p.err[env] = m_step;
        }
        if (!lazy) {
            if (from_mirror) { // (step_middle has read the lengths out of lmax)
                if (snake) cx.lmax[lane] = t0;
                wave_lds_sync();
            }
            store_env(cx, foodp, headp, bodyp, fbits0, hc0, sn.hc, rebuild, from_mirror); // a rebuilt env is stored whole
        }
        if (mirrored) {
            const bool rebased = rebase_clocks(cx); // the clocks stay with the mirror from call to call
            mirror_store(cx, mp, lane, 64, sn.hc, (snake && !sn.done) ? sn.L : 0, fence, // (a deleted snake's body reads all-zero)
                         from_mirror && !rebuild && !rebased, fbits0);
            wave_lds_sync();
        }
    };
    if (active) {
    foodp = p.foods + env * C; headp = p.heads + env * K * C; bodyp = p.bodies + env * K * C;

    // An env the postponed reset rebuilds is not read from the fp32 planes (as in multi_reset_kernel): the launch is one
    // round of waves and ends with its slowest env, and a rebuilt env — rebuild + whole-env store — is the slowest already.
    // The compact image is requested whether or not the env is rebuilt: "is it rebuilt" is itself a load, and waiting for
    // it first made two memory round trips of one.
    StepIn in;
    step_inputs(p, env, lane, in); // (requested here, used after the env's image has arrived)
    const int rebuild_byte = p.done_env != nullptr ? (int)p.done_env[env] : 0;
    mp = mirrored ? p.resident + env * mirror_env_bytes(K, C) : nullptr;
    bool plain = false; // read from the fp32 planes, which held nothing the image cannot represent
    if (from_mirror) fbits0 = mirror_load(cx, mp, lane, 64, fence, !lazy);
    rebuild = uniform(rebuild_byte) != 0;
    if (!rebuild) {
        if (!from_mirror) fbits0 = load_env(cx, foodp, headp, bodyp, p.err != nullptr, plain);
    } else {
        fbits0 = 0;
        if (snake) { cx.hcell[lane] = -1; cx.lmax[lane] = 0; cx.tclk[lane] = 0; } // the rest: multi_reset_grid(rebuild)
        wave_lds_sync();
    }
    t0 = snake ? cx.tclk[lane] : 0; // the clocks as loaded (0 unless the state came from the mirror)
    WURM_TLS(cx, 1);
    step_middle(cx, p, env, rebuild, sn, r, hc0, &in);
    WURM_TLS(cx, 8);
    clean = from_mirror || rebuild || plain; // lds_check sees everything there is to check
    if (!solo) store_state();
    WURM_TLS(cx, 9);
    }
    if (solo) { // every wave for itself: its env's class codes -> its K agents' views (no barrier)
        WURM_TLS(cx, 9);
        class_write<unsigned short>(cx, sn.hc, cx.snap, ring);
        WURM_TLS(cx, 10);
        WURM_TLS(cx, 11); // (no barrier in this form)
        grp_emit_group(p, p.obs + env * p.obs_elems - env0 * p.obs_elems, env0, 1, 0, 1, (const unsigned char *)cx.snap, 0, tab, lane);
        WURM_TLS(cx, 12);
        store_state();
#ifdef WURM_TIMELINE
        if (p.obs_after == nullptr) WURM_TLS_STORE(cx, p.obs + env * p.obs_elems);
#endif
        if (p.obs_after == nullptr) return;
        const bool touched = reset_for_obs_after(cx, p, env, sn, r);
        if (p.err_after != nullptr) {
            if (touched) m_step = clean ? lds_check(cx, sn) : MCHK_NOT_COMPUTED;
            if (lane == 0) p.err_after[env] = m_step;
        }
        if (touched) class_write<unsigned short>(cx, sn.hc, cx.snap, ring); // (else: the codes of the stepped state are still there)
        grp_emit_group(p, p.obs_after + env * p.obs_elems - env0 * p.obs_elems, env0, 1, 0, 1, (const unsigned char *)cx.snap, 0, tab, lane);
        return;
    }
    if (grouped) {
        const unsigned char *codes0 = (const unsigned char *)make_ctx(p, 0).snap;
        WURM_TLS(cx, 9);
        if (active) class_write<unsigned short>(cx, sn.hc, cx.snap, ring);
        WURM_TLS(cx, 10);
        workgroup_handoff();
        WURM_TLS(cx, 11);
        grp_emit_group(p, p.obs, env0, nG, wave, wpb, codes0, p.lds_per_wave, tab, lane);
        WURM_TLS(cx, 12);
#ifdef WURM_TIMELINE
        if (p.obs_after == nullptr) {
            workgroup_handoff(); // (the stamps go over what another wave has written)
            if (active) WURM_TLS_STORE(cx, p.obs + env * p.obs_elems);
        }
#endif
        if (p.obs_after == nullptr) return;
        workgroup_handoff(); // every wave has read the codes of the stepped state
        if (active) {
            const bool touched = reset_for_obs_after(cx, p, env, sn, r);
            if (p.err_after != nullptr) {
                if (touched) m_step = clean ? lds_check(cx, sn) : MCHK_NOT_COMPUTED; // (else: the state the first mask describes)
                if (lane == 0) p.err_after[env] = m_step;
            }
            class_write<unsigned short>(cx, sn.hc, cx.snap, ring);
        }
        workgroup_handoff();
        grp_emit_group(p, p.obs_after, env0, nG, wave, wpb, codes0, p.lds_per_wave, tab, lane);
        return;
    }
    if (p.obs_mode != WURM_OBS_NONE) observe(cx, p, p.obs, env, sn);
#ifdef WURM_TIMELINE
    WURM_TLS(cx, 12);
    if (p.obs_after == nullptr && p.obs_mode != WURM_OBS_NONE) WURM_TLS_STORE(cx, p.obs + env * p.obs_elems);
#endif
    if (p.obs_after == nullptr || p.obs_mode == WURM_OBS_NONE) return;
    const bool touched = reset_for_obs_after(cx, p, env, sn, r);
    if (p.err_after != nullptr) {
        if (touched) m_step = clean ? lds_check(cx, sn) : MCHK_NOT_COMPUTED; // (else: the state the first mask describes)
        if (lane == 0) p.err_after[env] = m_step;
    }
    observe(cx, p, p.obs_after, env, sn);
}

// ---- one env per WORKGROUP.  With 10 snakes on 36 x 36 (experiments/speeds.py) an env's grids take 33 KB of LDS: one
// wave per SIMD, and multi_step_kernel spends 445 us on 1.1 GB with nothing to overlap the load, the transition and the
// stores.  Here the four waves of a workgroup share ONE env: all of them copy it in, turn it into class codes and write
// it back (flat over the K * S * S cells; the K agents' observations go out one agent per wave), wave 0 alone runs the
// transition in between.  Same LDS layout, same device functions for everything that is not a plain copy.
// plain: as load_env's (the same in every thread)
__device__ __forceinline__ u64 wg_load_env(const Ctx &cx, const float *__restrict__ foodp, const float *__restrict__ headp,
                                           const float *__restrict__ bodyp, int tid, int nth, bool want_plain, bool &plain)
{
    const int C = cx.C, KC = cx.K * C;
    int odd = 0, nheads = 0;
    if (tid < cx.K) {
        cx.hcell[tid] = -1;
        cx.lmax[tid] = 0;
        cx.tclk[tid] = 0;
    }
    for (int c = tid; c < C; c += nth) cx.hmap[c] = 0;
    __syncthreads();
    const float rcpC = 1.0f / (float)C;
    constexpr int CH = 8;
    for (int base = 0; base < KC; base += nth * CH) {
        float hv[CH], bv[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) { // unconditional loads (index clamped into the env), all in flight together
            const int i = min(base + tid + nth * j, KC - 1);
            hv[j] = headp[i];
            bv[j] = bodyp[i];
        }
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int i = base + tid + nth * j;
            if (i < KC) {
                const int bi = __float2int_rn(bv[j]);
                cx.body[i] = (unsigned short)(bi != 0 ? ((bi & VMASK) | DIRTY) : 0);
                odd |= (int)((hv[j] != 0.0f && hv[j] != 1.0f) || bv[j] != (float)bi || bi < 0 || bi > (int)VMASK);
                nheads += (int)(hv[j] > 0.5f);
                if (hv[j] > 0.5f || bi > 0) { // rare: a head cell or a body cell
                    const int s = div_size(i, rcpC);
                    if (hv[j] > 0.5f) cx.hcell[s] = i - s * C;
                    if (bi > 0) atomicMax(&cx.lmax[s], bi);
                }
            }
        }
    }
    u64 fbits = 0; // bit k: food at cell tid + nth * k
    for (int k = 0, c = tid; c < C; ++k, c += nth) {
        const float fv = foodp[c];
        const int f = fv > 0.5f;
        cx.food[c] = (unsigned char)f;
        fbits |= (u64)f << k;
        odd |= (int)(fv != 0.0f && fv != 1.0f);
    }
    if (want_plain) { // (the colour slots are free until step_middle loads them: a counter for the heads seen by all threads)
        int *cnt = (int *)cx.colf;
        if (tid == 0) *cnt = 0;
        __syncthreads();
        if (nheads) atomicAdd(cnt, nheads);
        const int any_odd = __syncthreads_or(odd);
        int with_head = 0;
        for (int s2 = 0; s2 < cx.K; ++s2) with_head += (int)(cx.hcell[s2] >= 0);
        plain = !any_odd && *cnt == with_head;
        __syncthreads();
    }
    return fbits;
}

__device__ __forceinline__ u64 wg_load_env(const Ctx &cx, const float *__restrict__ foodp, const float *__restrict__ headp,
                                           const float *__restrict__ bodyp, int tid, int nth)
{
    bool unused = false;
    return wg_load_env(cx, foodp, headp, bodyp, tid, nth, false, unused);
}

__device__ __forceinline__ void wg_store_env(const Ctx &cx, float *__restrict__ foodp, float *__restrict__ headp,
                                             float *__restrict__ bodyp, u64 fbits0, bool full, int tid, int nth,
                                             bool t0_in_lmax = false)
{
    const int C = cx.C, KC = cx.K * C;
    const float rcpC = 1.0f / (float)C;
    for (int i = tid; i < KC; i += nth) {
        const unsigned short v = cx.body[i];
        const int s = div_size(i, rcpC), T = cx.tclk[s], T0 = t0_in_lmax ? cx.lmax[s] : 0;
        // changed since the load: written cells, and — once the clock has moved — every cell that held a value
        if (full || (v & DIRTY) || (T != T0 && (int)(v & VMASK) > T0)) bodyp[i] = (float)max((int)(v & VMASK) - T, 0);
        if (full) headp[i] = (i - s * C == cx.hcell[s]) ? 1.0f : 0.0f;
    }
    for (int k = 0, c = tid; c < C; ++k, c += nth) {
        const int f = cx.food[c] != 0;
        if (full || f != (int)((fbits0 >> k) & 1)) foodp[c] = f ? 1.0f : 0.0f;
    }
}

// 'full' observation of the env in LDS by the whole workgroup: class codes (cells over all threads), then one agent
// per wave.  cx.hcell holds the head cells.  Ends with a barrier (the codes and the head map may be rewritten after it).
__device__ __forceinline__ void wg_observe_snap(const Ctx &cx, const MultiArgs &p, float *obs, long long env, int tid,
                                                int nth, int wave)
{
    const int S = cx.S, C = cx.C, K = cx.K;
    int hc = -1;
    if (tid < K) {
        hc = cx.hcell[tid];
        if (hc >= 0) cx.hmap[hc] = (unsigned char)(tid + 1);
    }
    __syncthreads();
    for (int c = tid; c < C; c += nth) {
        const int y = div_size(c, cx.rcpS), x = c - y * S;
        const bool edge = y == 0 || x == 0 || y == S - 1 || x == S - 1;
        u32 v = 0;
        for (int s = 0; s < K; ++s) v |= (u32)((int)(cx.body[s * C + c] & VMASK) > cx.tclk[s]) << s;
        v |= (u32)cx.hmap[c] << SNAP_OWNER_SHIFT;
        v |= (u32)(cx.food[c] != 0) << 14;
        v |= (u32)edge << 15;
        cx.snap[c] = (unsigned short)v;
    }
    __syncthreads();
    if (tid < K && hc >= 0) cx.hmap[hc] = 0;
    float *obs_env = (float *)uniform64((long long)(obs + env * p.obs_elems));
    // (agent, half of its rows) items, dealt round robin: 10 agents over 4 waves were 3 + 3 + 2 + 2 whole views — the
    // workgroup waited for the waves with three
    const int nw = nth >> 6, half = (cx.cpl + 1) >> 1;
    if (p.grp_variant & 1) { // (A/B switch, WURM_MULTI_GROUP_VARIANT bit 0: whole views, round robin)
        for (int a = wave; a < K; a += nw) snap_emit_agent(cx, p, obs_env, cx.snap, a);
    } else {
        for (int i = wave; i < 2 * K; i += nw) snap_emit_agent(cx, p, obs_env, cx.snap, i >> 1, (i & 1) * half, (i & 1) ? cx.cpl : half);
    }
    __syncthreads();
}

template <bool INJ = true, int OBS = -1, int KT = 0, int ST = 0>   // (INJ = false: as multi_rollout_kernel — the launch draws its own random outcomes)
__global__ __launch_bounds__(256) void multi_step_wg_kernel(MultiArgs p_in)
{
    MultiArgs p = p_in;
    if (!INJ) p.has_inj = p.has_rinj = 0;
    if (OBS >= 0) p.obs_mode = OBS;
    shape_constants<OBS, KT, ST, -1>(p, true);
    const int tid = (int)threadIdx.x, nth = (int)blockDim.x, wave = uniform(tid >> 6);
    const long long env = xcd_block(blockIdx.x, gridDim.x);
    if (env >= p.N) return; // the whole workgroup: the barriers below see every wave or none
    const Ctx cx = make_ctx(p, 0);
    const int C = cx.C, K = cx.K, lane = cx.lane;
    if (wave == 0) { WURM_TLS_INIT(cx); WURM_TLS(cx, 0); }
    float *foodp = p.foods + env * C, *headp = p.heads + env * K * C, *bodyp = p.bodies + env * K * C;
    // (the rebuild flag, wave 0's step inputs and the compact image are requested together: see multi_step_kernel)
    const int rebuild_byte = p.done_env != nullptr ? (int)p.done_env[env] : 0;
    StepIn in = {};
    if (wave == 0) step_inputs(p, env, lane, in);
    const bool mirrored = p.resident != nullptr, lazy = mirrored && p.resident_lazy != 0;
    const bool from_mirror = mirrored && p.resident_valid != 0;
    unsigned char *mp = mirrored ? p.resident + env * mirror_env_bytes(K, C) : nullptr;
    auto barrier = [] { __syncthreads(); };
    u64 fbits0 = 0;
    bool plain = false;
    if (from_mirror) fbits0 = mirror_load(cx, mp, tid, nth, barrier, !lazy);
    const bool rebuild = rebuild_byte != 0;
    if (!rebuild) {
        if (!from_mirror) fbits0 = wg_load_env(cx, foodp, headp, bodyp, tid, nth, p.err != nullptr, plain);
    } else {
        fbits0 = 0;
        if (tid < K) { cx.hcell[tid] = -1; cx.lmax[tid] = 0; cx.tclk[tid] = 0; }
    }
    __syncthreads();
    const int t0 = tid < K ? cx.tclk[tid] : 0;
    Snake sn;
    StepRes r;
    int hc0 = -1;
    if (wave == 0) {
        WURM_TLS(cx, 1);
        step_middle(cx, p, env, rebuild, sn, r, hc0, &in);
        WURM_TLS(cx, 8);
    }
    const bool clean = from_mirror || rebuild || plain;
    // check_consistency's mask by all four waves when the class-code buffer is there to lend them 2 KB (a snake per wave at
    // a time), else by wave 0 alone
    const int nwv = nth >> 6;
    const bool spread = p.off_snap >= 0 && 2 * C >= 520 * nwv;
    uint32_t m_step = MCHK_NOT_COMPUTED;
    if (p.err != nullptr) {
        if (clean && spread) {
            if (wave == 0 && lane < K) cx.lmax[lane] = (int)sn.done; // (step_middle has read the lengths out of lmax)
            __syncthreads();
            m_step = wg_lds_check(cx, wave, nwv, (unsigned char *)cx.snap);
        } else if (clean && wave == 0) {
            m_step = lds_check(cx, sn);
        }
        if (tid == 0) p.err[env] = m_step;
    }
    __syncthreads();
    if (!lazy) {
        if (from_mirror) {
            if (tid < K) cx.lmax[tid] = t0;
            __syncthreads();
        }
        wg_store_env(cx, foodp, headp, bodyp, fbits0, rebuild, tid, nth, from_mirror);
        if (wave == 0 && !rebuild && lane < K && sn.hc != hc0) { // the head cells that moved
            float *hp = headp + (size_t)lane * C;
            if (hc0 >= 0) hp[hc0] = 0.0f;
            if (sn.hc >= 0) hp[sn.hc] = 1.0f;
        }
    }
    if (mirrored) {
        if (wave == 0 && rebase_clocks(cx) && lane == 0) cx.hmap[0] = 1; // (the head map is all-zero here: a flag for the others)
        __syncthreads();
        const bool rebased = cx.hmap[0] != 0;
        __syncthreads();
        if (tid == 0) cx.hmap[0] = 0;
        // (threads 0..K-1 are lanes of wave 0: they hold the snakes' head cells and lengths)
        mirror_store(cx, mp, tid, nth, sn.hc, (tid < K && !sn.done) ? sn.L : 0, barrier, from_mirror && !rebuild && !rebased,
                     fbits0);
        __syncthreads();
    }
    if (p.obs_mode == WURM_OBS_NONE) return;
    if (wave == 0) WURM_TLS(cx, 9);
    wg_observe_snap(cx, p, p.obs, env, tid, nth, wave);
#ifdef WURM_TIMELINE
    if (p.obs_after == nullptr && wave == 0) {
        WURM_TLS(cx, 12);
        WURM_TLS_STORE(cx, p.obs + env * p.obs_elems);
    }
#endif
    if (p.obs_after == nullptr) return;
    if (wave == 0) {
        const bool touched = reset_for_obs_after(cx, p, env, sn, r);
        if (lane == 0) cx.hmap[0] = (unsigned char)touched; // (the head map is all-zero between observations: a flag)
        if (lane < K) cx.lmax[lane] = (int)sn.done;
    }
    __syncthreads();
    if (p.err_after != nullptr) {
        const bool touched = cx.hmap[0] != 0;
        if (touched) {
            if (clean && spread) m_step = wg_lds_check(cx, wave, nwv, (unsigned char *)cx.snap);
            else if (clean && wave == 0) m_step = lds_check(cx, sn);
            else if (!clean) m_step = MCHK_NOT_COMPUTED;
        }
        if (tid == 0) p.err_after[env] = m_step;
    }
    __syncthreads();
    if (tid == 0) cx.hmap[0] = 0;
    __syncthreads();
    wg_observe_snap(cx, p, p.obs_after, env, tid, nth, wave);
}

// ------------------------------------------------------------------------------------------------ reset

// availability of _add_snake (:927-941) / _get_snake_addition (:848-858): the 3x3 neighbourhood is empty and the
// cell is at least 2 from the border.  occ[] holds the occupancy (food, heads, bodies).
__device__ __forceinline__ u64 spawn_cells(const Ctx &cx)
{
    const int S = cx.S, C = cx.C, lane = cx.lane;
    u64 av = 0;
    for (int k = 0; k < cx.cpl; ++k) {
        int c = lane + 64 * k;
        if (c >= C) continue;
        int y = div_size(c, cx.rcpS), x = c - y * S;
        if (y < 2 || x < 2 || y > S - 3 || x > S - 3) continue;
        int any = 0;
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) any |= cx.occ[c + dy * S + dx];
        if (!any) av |= 1ull << k;
    }
    return av;
}

// occupancy of the env currently in LDS
__device__ __forceinline__ void build_occ(const Ctx &cx, int hc)
{
    const int C = cx.C, lane = cx.lane;
    for (int k = 0; k < cx.cpl; ++k) {
        int c = lane + 64 * k;
        if (c >= C) continue;
        bool o = cx.food[c] != 0;
#pragma unroll 4
        for (int s = 0; s < cx.K; ++s) o |= BV(cx, s, c) > 0;
        cx.occ[c] = (unsigned char)o;
    }
    wave_lds_sync();
    if (lane < cx.K && hc >= 0) cx.occ[hc] = 1;
    wave_lds_sync();
}

// The respawn search of respawn_mode = 'any' (:805-831 -> _get_snake_addition :848-858) on row masks: the K-th cell, in
// row-major order, that is at least 2 from the border with nothing (food, body, head) in its 3 x 3 neighbourhood, K =
// mulhi(word, number of such cells); -1 if there is none.  Same cells in the same order as build_occ + spawn_cells +
// rank_select — which read every (cell, snake) pair one by one: 420 LDS reads per lane at 10 snakes on 36 x 36, run in
// nearly every step of such an env (some snake is almost always dead), half of the transition's time there.  Here lane l
// reads cells 8 l .. 8 l + 7 of every run of 512 with one 16-byte read per snake, leaves one occupancy BIT per cell in the
// scratch byte map, and lane r assembles row r's 64-bit mask from it; the rest is the dilation / popcount walk that the
// rebuild of an env uses.  Needs S * S to be a multiple of 8 (16-byte aligned snake grids).
__device__ __forceinline__ int respawn_cell_rows(const Ctx &cx, int hc, u32 word)
{
    const int S = cx.S, C = cx.C, K = cx.K, lane = cx.lane;
    unsigned char *bm = cx.occ; // C bytes of scratch: C / 8 of bitmap, the rest zero padding for the row reads
    const int myT = lane < K ? cx.tclk[lane] : 0;
    for (int i = lane; i < (C >> 3) + 16 && i < C; i += 64) bm[i] = 0;
    wave_lds_sync();
    const int runs = (C + 511) >> 9;
    for (int r = 0; r < runs; ++r) {
        const int c0 = 512 * r + 8 * lane;
        if (c0 >= C) continue;
        const u64 f8 = *(const u64 *)(cx.food + c0);
        u32 o8 = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) o8 |= (u32)(((f8 >> (8 * j)) & 0xffull) != 0) << j;
        for (int s = 0; s < K; ++s) {
            const int T = lane_value(myT, s);
            const uint4 q = *(const uint4 *)(cx.body + s * C + c0);
            o8 |= (u32)((int)(q.x & VMASK) > T) | ((u32)((int)((q.x >> 16) & VMASK) > T) << 1) |
                  ((u32)((int)(q.y & VMASK) > T) << 2) | ((u32)((int)((q.y >> 16) & VMASK) > T) << 3) |
                  ((u32)((int)(q.z & VMASK) > T) << 4) | ((u32)((int)((q.z >> 16) & VMASK) > T) << 5) |
                  ((u32)((int)(q.w & VMASK) > T) << 6) | ((u32)((int)((q.w >> 16) & VMASK) > T) << 7);
        }
        bm[c0 >> 3] = (unsigned char)o8;
    }
    wave_lds_sync();
    if (lane < K && hc >= 0) atomicOr((u32 *)bm + (hc >> 5), 1u << (hc & 31)); // head cells (the map is 16-byte aligned)
    wave_lds_sync();
    u64 occ_row = 0;
    if (lane < S) { // bits lane * S .. lane * S + S - 1 of the map
        const int bit0 = lane * S, byte0 = bit0 >> 3, sh = bit0 & 7;
        u64 lo = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) lo |= (u64)bm[byte0 + i] << (8 * i);
        const u64 hi = bm[byte0 + 8];
        const u64 v = sh ? (lo >> sh) | (hi << (64 - sh)) : lo;
        occ_row = S == 64 ? v : v & ((1ull << S) - 1ull);
    }
    // available (:848-858): at least 2 from the border, nothing in the 3 x 3 neighbourhood
    const u64 h = occ_row | (occ_row << 1) | (occ_row >> 1);
    const u64 up = lane == 0 ? 0ull : (u64)__shfl_up((long long)h, 1);
    const u64 dn = lane == 63 ? 0ull : (u64)__shfl_down((long long)h, 1);
    const u64 cols = S >= 5 ? (((1ull << (S - 4)) - 1ull) << 2) : 0ull;
    const u64 av = (lane >= 2 && lane <= S - 3) ? (~(h | up | dn) & cols) : 0ull;
    const int cnt = popc64(av), n = wave_sum_i32(cnt);
    if (n == 0) return -1;
    int kth = (int)mulhi_range(word, (u32)n), r = 0;
    for (; r < S - 1; ++r) {
        const int c = lane_value(cnt, r);
        if (kth < c) break;
        kth -= c;
    }
    return r * S + nth_bit64((u64)lane_value64((long long)av, r), kth);
}

// writes a 3-segment snake `s` at `cell` heading `d` into LDS (body, occ); cell < 0: nothing
__device__ __forceinline__ int place_snake(const Ctx &cx, int s, int cell, int d)
{
    const int S = cx.S, C = cx.C;
    if (cell < 0) return -1;
    int sy = div_size(cell, cx.rcpS), sx = cell - sy * S;
    int hcell = (sy + tap_y(d)) * S + sx + tap_x(d), tcell = (sy - tap_y(d)) * S + sx - tap_x(d);
    if (cx.lane == 0) { // LENGTH_3_SNAKES (:965-973): 3 at seed + TAP[d], 2 at the seed, 1 at seed - TAP[d]
        cx.body[s * C + hcell] = (unsigned short)(3 | DIRTY);
        cx.body[s * C + cell] = (unsigned short)(2 | DIRTY);
        cx.body[s * C + tcell] = (unsigned short)(1 | DIRTY);
        cx.occ[hcell] = 1;
        cx.occ[cell] = 1;
        cx.occ[tcell] = 1;
    }
    wave_lds_sync();
    return hcell;
}

__device__ __forceinline__ void colour_from_words(const Words &w, short out[3])
{
    // get_n_colours (:163-169): rand(3); red / 1.5; normalise; * 192; .short()
    // plain `/` and sqrtf are the correctly rounded IEEE operations here (hipcc's default
    // -fhip-fp32-correctly-rounded-divide-sqrt); the __fdiv_rn / __fsqrt_rn intrinsics are NOT (found by
    // tools/fuzz_parity.py: one colour component in thousands came out one lower than on the CPU)
    float c0 = u01(w.w[0]) / 1.5f, c1 = u01(w.w[1]), c2 = u01(w.w[2]);
    float norm = sqrtf(c0 * c0 + c1 * c1 + c2 * c2);
    out[0] = (short)(c0 / norm * 192.0f);
    out[1] = (short)(c1 / norm * 192.0f);
    out[2] = (short)(c2 / norm * 192.0f);
}

// colours of snakes that are still dead are re-rolled on every reset (:800-803).  Returns true if sn.col changed.
__device__ __forceinline__ bool reroll_colour(const MultiArgs &p, long long agent, bool dead, u64 env_id, u64 call,
                                              long long offA, Snake &sn)
{
    if (!(p.cfg.colour_random && dead)) return false;
    if (p.has_rinj) {
        sn.col[0] = p.rinj.colours[(offA + agent) * 3];
        sn.col[1] = p.rinj.colours[(offA + agent) * 3 + 1];
        sn.col[2] = p.rinj.colours[(offA + agent) * 3 + 2];
    } else {
        colour_from_words(rng_words(p.seed, call, env_id, RNG_COLOUR, (u32)(threadIdx.x & 63u)), sn.col);
    }
    return true;
}

// The respawn search from the map of cell codes the step left (cell_codes + put_food: Snake::cmap_ok) — any state, any
// size: a cell is occupied iff its code is neither 0 nor the ring's (what sits ON the ring never matters: a spawn cell is
// at least 2 from the border, so its 3 x 3 neighbourhood stops at row / column 1).  Ten ballots give the occupancy of the
// 64-cell chunks, lane r cuts grid row r out of at most two of them, and the rest is the dilation / popcount walk of the
// rebuild: ~100 instructions for what build_occ + spawn_cells + rank_select read cell by cell — K clock compares and nine
// byte reads per cell, 5 100 of a step's 30 000 cycles with respawn_mode = 'any' (profiles/r05_kernel_timeline.txt: 2 850 now).
__device__ __forceinline__ int respawn_cell_codes(const Ctx &cx, const unsigned char *codes, u32 word)
{
    const int S = cx.S, C = cx.C, lane = cx.lane;
    u64 *chunks = (u64 *)cx.occ;   // scratch: cpl <= 64 masks of 64 cells (C bytes, 16-byte aligned; 8 cpl <= C for S >= 5)
    u64 mine = 0;
    for (int k = 0; k < cx.cpl; ++k) {
        const int c = lane + 64 * k;
        const int code = c < C ? (int)codes[c] : PC_BG;
        const u64 m = ballot(code != PC_BG && code != PC_RING);
        if (lane == k) mine = m;
    }
    if (lane < cx.cpl) chunks[lane] = mine;
    wave_lds_sync();
    u64 occ_row = 0;
    if (lane < S) { // grid row `lane`: bits lane * S .. lane * S + S - 1 of the linear occupancy
        const int b = lane * S, q = b >> 6, off = b & 63;
        const u64 lo = chunks[q], hi = (q + 1 < cx.cpl) ? chunks[q + 1] : 0ull;
        occ_row = (lo >> off) | (off ? hi << (64 - off) : 0ull);
        if (S < 64) occ_row &= (1ull << S) - 1ull;
    }
    wave_lds_sync();
    // available (:848-858): at least 2 from the border, nothing in the 3 x 3 neighbourhood
    const u64 h = occ_row | (occ_row << 1) | (occ_row >> 1);
    const u64 up = lane == 0 ? 0ull : (u64)__shfl_up((long long)h, 1);
    const u64 dn = lane == 63 ? 0ull : (u64)__shfl_down((long long)h, 1);
    const u64 cols = S >= 5 ? (((1ull << (S - 4)) - 1ull) << 2) : 0ull;
    const u64 av = (lane >= 2 && lane <= S - 3) ? (~(h | up | dn) & cols) : 0ull;
    const int cnt = popc64(av), n = wave_sum_i32(cnt);
    if (n == 0) return -1;
    int kth = (int)mulhi_range(word, (u32)n), r = 0;
    for (; r < S - 1; ++r) { // the K-th available cell in row-major order
        const int c = lane_value(cnt, r);
        if (kth < c) break;
        kth -= c;
    }
    return r * S + nth_bit64((u64)lane_value64((long long)av, r), kth);
}

// the grid part of MultiSnake.reset on the env held in LDS: _create_envs (:996-1019) when `rebuild`, then the
// respawn of the first dead snake (:805-831) when `respawn`.  sn.done must already be false for rebuilt envs (:798).
__device__ __forceinline__ void multi_reset_grid(const Ctx &cx, const MultiArgs &p, long long env, u64 env_id, u64 call,
                                                 bool rebuild, bool respawn, Snake &sn, bool &orient_dirty,
                                                 long long offA, long long offE)
{
    const int C = cx.C, K = cx.K, lane = cx.lane;
    const bool snake = lane < K;
    const bool had_map = sn.cmap_ok && !rebuild; // the step's map of cell codes still describes the grids the respawn looks at
    sn.cmap_ok = false; // (... but not the state this reset leaves)
    if (rebuild) { // _create_envs (:996-1019)
        { // value 0; a cell that ever held one stays marked.  Four cells per access: the grids start on a 16-byte
          // boundary and are followed by padding up to the next one, so the last access may run into the padding.
            u64 *b8 = (u64 *)cx.body;
            const u64 keep = (u64)DIRTY * 0x0001000100010001ull;
            for (int i = lane; i < (K * C + 3) >> 2; i += 64) b8[i] &= keep;
        }
        for (int k = 0; k < cx.cpl; ++k) {
            int c = lane + 64 * k;
            if (c < C) { cx.food[c] = 0; cx.occ[c] = 0; cx.hmap[c] = 0; }
        }
        if (snake) cx.tclk[lane] = 0;
        wave_lds_sync();
        sn.hc = -1;
        // RNG mode: the env is empty, so the occupancy is just the cells of the snakes placed so far — one 64-bit row
        // mask per lane (lane r = row r) instead of the byte map: "3x3 neighbourhood empty" is a dilation (two shifts
        // and the rows above / below), the count a popcount, the K-th free cell in row-major order a walk over the
        // rows' counts.  Same cells as spawn_cells / count_bits / rank_select (measured: a rebuilt env took 52 000
        // cycles of a 35 000-cycle step launch, and the launch waits for its slowest env).
        const int S = cx.S;
        u64 occ_row = 0;
        auto pick = [&](u64 av, u32 word) -> int { // K-th set bit over the rows, K = mulhi(word, total)
            const int cnt = popc64(av), n = wave_sum_i32(cnt);
            if (n == 0) return -1;
            int kth = (int)mulhi_range(word, (u32)n), r = 0;
            for (; r < S - 1; ++r) {
                const int c = lane_value(cnt, r);
                if (kth < c) break;
                kth -= c;
            }
            return r * S + nth_bit64((u64)lane_value64((long long)av, r), kth);
        };
        auto mark = [&](int cell) {
            const int y = div_size(cell, cx.rcpS), x = cell - y * S;
            if (lane == y) occ_row |= 1ull << x;
        };
        // all the draws of the rebuild in one Philox evaluation: lane s < K takes the spawn block of snake s, lane K
        // the block the food cell comes from (K = 64: there is no such lane, the food block is drawn on its own)
        Words draws;
        draws.w[0] = draws.w[1] = draws.w[2] = draws.w[3] = 0;
        if (!p.has_rinj)
            draws = rng_words(p.seed, call, env_id, lane < K ? RNG_SPAWN : RNG_RESET, lane < K ? (u32)lane : 0u);
        for (int s = 0; s < K; ++s) { // _add_snake (:911-994), one snake after another
            int cell = -1, dnew = 0;
            if (p.has_rinj) {
                cell = p.rinj.create[(offA + env * K + s) * 2];
                dnew = p.rinj.create[(offA + env * K + s) * 2 + 1];
            } else {
                Words w;
                w.w[0] = (u32)lane_value((int)draws.w[0], s);
                w.w[1] = (u32)lane_value((int)draws.w[1], s);
                dnew = (int)(w.w[1] >> 30);
                // available (:927-941): at least 2 from the border, nothing in the 3x3 neighbourhood
                const u64 h = occ_row | (occ_row << 1) | (occ_row >> 1);
                const u64 up = lane == 0 ? 0ull : (u64)__shfl_up((long long)h, 1);
                const u64 dn = lane == 63 ? 0ull : (u64)__shfl_down((long long)h, 1);
                const u64 cols = S >= 5 ? (((1ull << (S - 4)) - 1ull) << 2) : 0ull;
                const u64 av = (lane >= 2 && lane <= S - 3) ? (~(h | up | dn) & cols) : 0ull;
                cell = pick(av, w.w[0]);
            }
            cell = uniform(cell);
            if (cell < 0 && p.status && lane == 0) atomicAdd(p.status, 1); // the reference raises (:946-947)
            int h = place_snake(cx, s, cell, dnew);
            if (cell >= 0 && !p.has_rinj) {
                const int sy = div_size(cell, cx.rcpS), sx = cell - sy * S;
                mark(cell);
                mark((sy + tap_y(dnew)) * S + sx + tap_x(dnew));
                mark((sy - tap_y(dnew)) * S + sx - tap_x(dnew));
            }
            if (lane == s) {
                sn.hc = h;
                sn.L = h >= 0 ? 3 : 0;
                sn.orient = dnew;
                orient_dirty = true;
            }
        }
        { // food (:1016-1017)
            if (p.has_rinj) {
                int cell = p.rinj.create_food[offE + env];
                if (cell >= 0 && cell < C && lane == 0) cx.food[cell] = 1;
            } else { // free (:439-445): not on the border ring, nothing on it
                const u64 cols = ((1ull << (S - 2)) - 1ull) << 1;
                const u64 fr = (lane >= 1 && lane <= S - 2) ? (~occ_row & cols) : 0ull;
                const u32 word = K < 64 ? (u32)lane_value((int)draws.w[3], K)
                                        : rng_words(p.seed, call, env_id, RNG_RESET, 0).w[3];
                const int cell = pick(fr, word);
                if (cell >= 0 && lane == 0) cx.food[cell] = 1;
            }
            wave_lds_sync();
        }
    }
    if (respawn) { // :805-831 the first dead snake of the env respawns if there is room
        const int f = first_bit(ballot(snake && sn.done));
        int cell = -1, dnew = 0;
        if (p.has_rinj) {
            build_occ(cx, sn.hc);
            cell = p.rinj.respawn[(offE + env) * 2];
            dnew = p.rinj.respawn[(offE + env) * 2 + 1];
        } else {
            Words w = rng_words(p.seed, call, env_id, RNG_SPAWN, (u32)K);
            dnew = (int)(w.w[1] >> 30);
            if (had_map) {
                cell = respawn_cell_codes(cx, cx.hmap, w.w[0]);
            } else if (p.obs_mode == WURM_OBS_PARTIAL) {
                // no map of this state yet (the postponed reset in front of a per-call step: the launch has just loaded the
                // env): one scan builds it — 4 100 cycles + 2 500 for the search, where build_occ + spawn_cells + rank_select
                // below took 23 000 in every env with a dead snake, and the launch ends with its slowest wave
                // (profiles/r06_kernel_timeline_multi.txt).  hmap is the crops' map of cell codes in such a launch anyway (the one-env-per-
                // workgroup kernels, which keep flags in it, never write crops).
                (void)cell_codes(cx, sn.hc, cx.hmap, cx.has_ring ? cx.ring : border_bits(cx));
                cell = respawn_cell_codes(cx, cx.hmap, w.w[0]);
            } else if ((C & 7) == 0) {
                cell = respawn_cell_rows(cx, sn.hc, w.w[0]);
            } else {
                build_occ(cx, sn.hc);
                u64 av = spawn_cells(cx);
                int n = count_bits(cx, av);
                if (n > 0) cell = selected_cell(rank_select(cx, av, (int)mulhi_range(w.w[0], (u32)n)));
            }
        }
        cell = uniform(cell);
        // bodies[first] = new_bodies (:826): the dead snake's grid is replaced (it reads all-zero in consistent
        // states: its clock is CLOCK_DEAD) and its clock restarts
        for (int k = 0; k < cx.cpl; ++k) {
            int c = lane + 64 * k;
            if (c < C && (cx.body[f * C + c] & VMASK)) cx.body[f * C + c] = DIRTY;
        }
        if (lane == f) cx.tclk[lane] = 0;
        wave_lds_sync();
        int h = place_snake(cx, f, cell, dnew);
        if (lane == f) {
            sn.hc = h;
            sn.L = h >= 0 ? 3 : 0;
            sn.orient = dnew;     // :828 assigned whether or not the snake found room
            orient_dirty = true;
            sn.done = cell < 0;   // :829
        }
    }
    if (snake) cx.hcell[lane] = sn.hc;
    wave_lds_sync();
}

__global__ __launch_bounds__(256) void multi_reset_kernel(MultiArgs p)
{
    const int wave = (int)(threadIdx.x >> 6), wpb = (int)(blockDim.x >> 6);
    const long long env = xcd_block(blockIdx.x, gridDim.x) * wpb + wave;
    if (env >= p.N) return;
    const Ctx cx = make_ctx(p, wave);
    const int C = cx.C, K = cx.K, lane = cx.lane;
    const bool snake = lane < K;
    const u64 env_id = (u64)(p.env_offset + env);
    const long long agent = env * K + lane;
    float *foodp = p.foods + env * C, *headp = p.heads + env * K * C, *bodyp = p.bodies + env * K * C;

    const bool rebuild = uniform((int)p.done_env[env]) != 0;
    Snake sn;
    sn.hc = -1;
    sn.L = 0;
    sn.done = snake ? p.dones[agent] != 0 : false;
    if (rebuild) sn.done = false; // :798
    sn.orient = 0;
    sn.boosted = false;
    const bool any_dead = ballot(snake && sn.done) != 0;
    const bool respawn = p.cfg.respawn_any && any_dead;
    const bool want_obs = p.obs_mode != WURM_OBS_NONE;

    load_colour(p, agent, snake && (want_obs || p.cfg.colour_random), sn);
    if (snake && reroll_colour(p, agent, sn.done, env_id, p.call, 0, sn)) {
        p.colours[agent * 3] = sn.col[0];
        p.colours[agent * 3 + 1] = sn.col[1];
        p.colours[agent * 3 + 2] = sn.col[2];
    }
    if (!rebuild && !respawn && !want_obs) return;

    int hc0 = -1;
    u64 fbits0 = 0;
    if (!rebuild) {
        fbits0 = load_env(cx, foodp, headp, bodyp);
        sn.hc = snake ? cx.hcell[lane] : -1;
        sn.L = snake ? cx.lmax[lane] : 0;
        hc0 = sn.hc;
    }
    bool orient_dirty = false;
    multi_reset_grid(cx, p, env, env_id, p.call, rebuild, respawn, sn, orient_dirty, 0, 0);

    if (snake) {
        if (rebuild || respawn) p.dones[agent] = (uint8_t)sn.done;
        if (orient_dirty) p.orientations[agent] = sn.orient;
    }
    if (rebuild) store_env(cx, foodp, headp, bodyp, 0, -1, sn.hc, true);
    else if (respawn) store_env(cx, foodp, headp, bodyp, fbits0, hc0, sn.hc, false);
    if (want_obs) {
        sn.boosted = snake ? p.boost[agent] != 0 : false;
        observe(cx, p, p.obs, env, sn);
    }
}

__global__ __launch_bounds__(256) void multi_observe_kernel(MultiArgs p)
{
    const int wave = (int)(threadIdx.x >> 6), wpb = (int)(blockDim.x >> 6);
    const long long env = xcd_block(blockIdx.x, gridDim.x) * wpb + wave;
    if (env >= p.N) return;
    const Ctx cx = make_ctx(p, wave);
    const int C = cx.C, K = cx.K, lane = cx.lane;
    load_env(cx, p.foods + env * C, p.heads + env * K * C, p.bodies + env * K * C);
    const bool snake = lane < K;
    Snake sn;
    sn.hc = snake ? cx.hcell[lane] : -1;
    sn.L = 0;
    sn.orient = 0;
    sn.done = snake ? p.dones[env * K + lane] != 0 : false;
    sn.boosted = snake ? p.boost[env * K + lane] != 0 : false;
    load_colour(p, env * K + lane, snake && p.obs_mode == WURM_OBS_PARTIAL, sn);
    observe(cx, p, p.obs, env, sn);
}

// ---- reset and _observe of large envs, one env per workgroup (see multi_step_wg_kernel)
__global__ __launch_bounds__(256) void multi_reset_wg_kernel(MultiArgs p)
{
    const int tid = (int)threadIdx.x, nth = (int)blockDim.x, wave = uniform(tid >> 6);
    const long long env = xcd_block(blockIdx.x, gridDim.x);
    if (env >= p.N) return;
    const Ctx cx = make_ctx(p, 0);
    const int C = cx.C, K = cx.K, lane = cx.lane;
    const bool snake = wave == 0 && lane < K;
    const u64 env_id = (u64)(p.env_offset + env);
    const long long agent = env * K + lane;
    float *foodp = p.foods + env * C, *headp = p.heads + env * K * C, *bodyp = p.bodies + env * K * C;
    const bool rebuild = p.done_env[env] != 0;
    const bool want_obs = p.obs_mode != WURM_OBS_NONE;
    // every thread works out whether a snake is (still) dead: the workgroup decides together whether there is work
    bool any_dead = false;
    if (!rebuild)
        for (int sidx = 0; sidx < K; ++sidx) any_dead |= p.dones[env * K + sidx] != 0;
    const bool respawn = p.cfg.respawn_any && any_dead;
    Snake sn;
    sn.hc = -1;
    sn.L = 0;
    sn.done = snake ? p.dones[agent] != 0 : false;
    if (rebuild) sn.done = false; // :798
    sn.orient = 0;
    sn.boosted = false;
    if (wave == 0) {
        load_colour(p, agent, snake && (want_obs || p.cfg.colour_random), sn);
        if (snake && reroll_colour(p, agent, sn.done, env_id, p.call, 0, sn)) {
            p.colours[agent * 3] = sn.col[0];
            p.colours[agent * 3 + 1] = sn.col[1];
            p.colours[agent * 3 + 2] = sn.col[2];
        }
    }
    if (!rebuild && !respawn && !want_obs) return;
    u64 fbits0 = 0;
    if (!rebuild) fbits0 = wg_load_env(cx, foodp, headp, bodyp, tid, nth);
    else if (tid < K) { cx.hcell[tid] = -1; cx.lmax[tid] = 0; cx.tclk[tid] = 0; }
    __syncthreads();
    int hc0 = -1;
    if (wave == 0) {
        if (!rebuild) {
            sn.hc = snake ? cx.hcell[lane] : -1;
            sn.L = snake ? cx.lmax[lane] : 0;
            hc0 = sn.hc;
        }
        bool orient_dirty = false;
        multi_reset_grid(cx, p, env, env_id, p.call, rebuild, respawn, sn, orient_dirty, 0, 0);
        if (snake) {
            if (rebuild || respawn) p.dones[agent] = (uint8_t)sn.done;
            if (orient_dirty) p.orientations[agent] = sn.orient;
        }
    }
    __syncthreads();
    if (rebuild || respawn) {
        wg_store_env(cx, foodp, headp, bodyp, fbits0, rebuild, tid, nth);
        if (snake && !rebuild && sn.hc != hc0) {
            float *hp = headp + (size_t)lane * C;
            if (hc0 >= 0) hp[hc0] = 0.0f;
            if (sn.hc >= 0) hp[sn.hc] = 1.0f;
        }
    }
    if (want_obs) wg_observe_snap(cx, p, p.obs, env, tid, nth, wave);
}

__global__ __launch_bounds__(256) void multi_observe_wg_kernel(MultiArgs p)
{
    const int tid = (int)threadIdx.x, nth = (int)blockDim.x, wave = uniform(tid >> 6);
    const long long env = xcd_block(blockIdx.x, gridDim.x);
    if (env >= p.N) return;
    const Ctx cx = make_ctx(p, 0);
    wg_load_env(cx, p.foods + env * cx.C, p.heads + env * cx.K * cx.C, p.bodies + env * cx.K * cx.C, tid, nth);
    __syncthreads();
    wg_observe_snap(cx, p, p.obs, env, tid, nth, wave);
}

// ------------------------------------------------------------------------------------------------ rollout

// T fused iterations of the caller loop of experiments/speeds.py:30-37 / tests/test_multi_snake_env.py:78-89:
//   step(actions[t]) with call = call0 + 2t  ->  outputs[t], observation[t];   reset(dones['__all__']) with call0 + 2t + 1
// with the env resident in LDS and the per-snake scalars in lanes for the whole launch: the state crosses HBM twice
// per launch instead of four times per iteration; per iteration only the actions are read and the outputs written.
//   actions (T,K,N);  out_f32 (T,3,K,N) = rewards, food, sizes;  out_u8 (T,4,K,N) = dones, boost, snake_collision,
//   edge_collision;  all_done (T,N);  obs (T,K,N,elems).
// TWO ('full' observations of at most 10 snakes): a workgroup is TWO waves for ONE env.  Wave 0 steps the env and leaves
// the class codes of the stepped state in one of two LDS buffers; wave 1 turns the codes of step t into the ~120 stores
// of its observation while wave 0 is already computing step t + 1; one s_barrier per step hands a buffer over.  The
// transition (latency-bound LDS work) and the observation (store-bound) of a launch otherwise ADD UP — every wave is in
// the same phase — and interleaving them inside one wave does not help (measured); two waves with their own
// instruction streams do overlap.

// INJ = false: the launch draws its random outcomes itself (every launch but the replays of recorded fixtures) — the
// injected-outcome branches of the step and of the reset, their nine pointers and three running offsets fold away, which
// matters in a kernel whose uniform state does not fit the scalar registers (profiles/r05_kernel_resources.txt).
// OBS >= 0: the observation mode is a compile-time constant too (the other modes' writers and their loop invariants go).
template <bool TWO, bool INJ, int OBS = -1, int KT = 0, int ST = 0, int NT = -1>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void multi_rollout_kernel(MultiArgs p_in)
{
    MultiArgs p = p_in;
    if (!INJ) p.has_inj = p.has_rinj = 0;
    if (OBS >= 0) p.obs_mode = OBS;
    shape_constants<OBS, KT, ST, NT>(p, !TWO && (OBS == WURM_OBS_PARTIAL || OBS == WURM_OBS_NONE), 0);
    const int wave = uniform((int)(threadIdx.x >> 6)), wpb = TWO ? 1 : (int)(blockDim.x >> 6);
    const long long env = xcd_block(blockIdx.x, gridDim.x) * wpb + (TWO ? 0 : wave);
    if (env >= p.N) return;
    const Ctx cx = make_ctx(p, TWO ? 0 : wave, 0, !TWO && p.obs_mode != WURM_OBS_DEFAULT);
    const int C = cx.C, K = cx.K, lane = cx.lane;
    if (TWO && wave == 1) { // the writer: observation of step t from buffer t & 1, handed over by the barrier of step t
        const long long KNw = (long long)K * p.N;
        for (long long t = 0; t < p.T; ++t) {
            workgroup_handoff();
            snap_emit(cx, p, (float *)uniform64((long long)(p.obs + t * KNw * p.obs_elems + env * p.obs_elems)),
                      cx.snap + (t & 1) * C);
        }
        return;
    }
    const bool snake = lane < K;
    const u64 env_id = (u64)(p.env_offset + env);
    const long long agent = env * K + lane, KN = (long long)K * p.N;
    float *foodp = p.foods + env * C, *headp = p.heads + env * K * C, *bodyp = p.bodies + env * K * C;

    // (the caller's compact mirror when it describes the state — wurm_multi_rollout_resident — else the fp32 planes; the writer
    // wave of the TWO form never gets here with a mirror: that form works on the planes)
    const bool mirrored = !TWO && p.resident != nullptr, from_mirror = mirrored && p.resident_valid != 0;
    unsigned char *const mp = mirrored ? p.resident + env * mirror_env_bytes(K, C) : nullptr;
    const u64 fbits0 = from_mirror ? mirror_load(cx, mp, lane, 64, [] { wave_lds_sync(); }, false) : load_env(cx, foodp, headp, bodyp);
    Snake sn;
    sn.hc = snake ? cx.hcell[lane] : -1;
    sn.L = snake ? cx.lmax[lane] : 0;
    sn.done = snake ? p.dones[agent] != 0 : true;
    sn.orient = snake ? p.orientations[agent] : 0;
    sn.boosted = false;
    load_colour(p, agent, snake, sn);
    bool col_dirty = false;
    const int hc0 = sn.hc; // what HBM holds: for the sparse write-back at the end
    WURM_TLS_INIT(cx);

    for (long long t = 0; t < p.T; ++t) {
        const u64 call = p.call + 2ull * (u64)t;
        WURM_TLS(cx, 0); // (timeline: the segment that ends here is the previous step's reset)
        // Actions come from LDS, 64 steps at a time.  A global LOAD inside the step loop would queue behind the
        // observation stores of the whole CU — the vector memory pipeline is in order — and every step would wait for
        // the store backlog to drain: that, not the arithmetic, is why transition and observation time used to add up.
        // Kept per action: a % 4 (C semantics, -3..3) and a > 3, which is all the step reads of it (:483-484).
        if ((t & 63) == 0) {
            const int nt = (int)min((long long)64, p.T - t);
            wave_lds_sync();
            for (int i = lane; i < nt * K; i += 64) {
                const int j = i / K, sidx = i - j * K;
                const long long av = p.actions[(t + j) * KN + (long long)sidx * p.N + env];
                cx.acts[i] = (unsigned char)(((int)(av % 4) + 4) | (av > 3 ? 8 : 0));
            }
            wave_lds_sync();
        }
        long long a = 0;
        if (snake) {
            const int b = cx.acts[(int)(t & 63) * K + lane];
            const int d4 = (b & 7) - 4;
            a = (b & 8) ? 4 + d4 : d4; // same a % 4 and a > 3 as the caller's value (a > 3 implies a % 4 >= 0)
        }
        StepRes r;
        WURM_TLS(cx, 2); // actions decoded
        multi_step_body(cx, p, env, env_id, call, a, sn, r, t * p.N * C, t * KN, t * p.N);
        WURM_TLS(cx, 7);
        if (snake) {
            const long long am = (long long)lane * p.N + env;
            float *of = p.am_f32 + t * 3 * KN;
            uint8_t *ob = p.am_u8 + t * 4 * KN;
            of[am] = r.reward;
            of[KN + am] = r.foodcons;
            of[2 * KN + am] = (float)sn.L;
            ob[am] = (uint8_t)sn.done;
            ob[KN + am] = (uint8_t)sn.boosted;
            ob[2 * KN + am] = (uint8_t)r.snakecol;
            ob[3 * KN + am] = (uint8_t)r.edgecol;
        }
        if (lane == 0) p.all_done[t * p.N + env] = (uint8_t)r.all_done;
        if (TWO) {
            // buffer t & 1 is free: the writer finished with it before it arrived at the barrier of step t - 1
            snap_write(cx, sn.hc, cx.snap + (t & 1) * C);
            workgroup_handoff();
        } else if (p.obs_mode != WURM_OBS_NONE) {
            WURM_TLS(cx, 8); // outputs stored
            observe(cx, p, p.obs + t * KN * p.obs_elems, env, sn);
            WURM_TLS(cx, 9); // observation issued
        }
        rebase_clocks(cx);

        // reset(dones['__all__']) (:771-836)
        if (snake && sn.done) sn.L = 0;           // deleted snakes have an all-zero body
        const bool rebuild = r.all_done;
        if (rebuild) sn.done = false;             // :798
        if (snake) col_dirty |= reroll_colour(p, agent, sn.done, env_id, call + 1ull, t * KN, sn);
        const bool respawn = p.cfg.respawn_any && ballot(snake && sn.done) != 0;
        if (rebuild || respawn) {
            bool orient_dirty = false;
            multi_reset_grid(cx, p, env, env_id, call + 1ull, rebuild, respawn, sn, orient_dirty, t * KN, t * p.N);
        }
    }

    if (snake) {
        p.dones[agent] = (uint8_t)sn.done;
        p.orientations[agent] = sn.orient;
        if (p.boost_state) p.boost_state[agent] = (uint8_t)sn.boosted;
        if (col_dirty) {
            p.colours[agent * 3] = sn.col[0];
            p.colours[agent * 3 + 1] = sn.col[1];
            p.colours[agent * 3 + 2] = sn.col[2];
        }
        cx.hcell[lane] = sn.hc;
    }
    wave_lds_sync();
    // only what may differ from HBM: body cells that have held a value since the load (DIRTY survives deletions, rebuilds
    // and re-bases), the head cell of each snake, food cells that changed — everything when the state came from the mirror,
    // nothing while the mirror is lazy
    if (!(mirrored && p.resident_lazy)) store_env(cx, foodp, headp, bodyp, fbits0, hc0, sn.hc, from_mirror);
    if (mirrored) {
        (void)rebase_clocks(cx);
        mirror_store(cx, mp, lane, 64, sn.hc, (snake && !sn.done) ? sn.L : 0, [] { wave_lds_sync(); });
        wave_lds_sync();
    }
#ifdef WURM_TIMELINE
    if (!TWO && p.obs_mode != WURM_OBS_NONE) WURM_TLA_STORE(cx, p.obs + env * p.obs_elems); // (step 0, agent 0: garbage by construction)
#endif
}


// ---- 'full' observations of at most 5 snakes, large batches: G consecutive envs per WORKGROUP (round 4).
// G stepper waves (one env each, as above) + W writer waves.  What changed against multi_rollout_kernel<true>, and why
// (tools/microbench/store_runs.hip, profiles/r04_store_runs_microbench.txt):
//   * the stream shape: writer wave w owns agent w (w + W, ...) and writes that agent's observations of the G envs of the
//     group as ONE linear run of G * 3 S^2 floats per step (the layout is (T, K, N, 3 S^2): for a fixed agent consecutive
//     envs are adjacent) — 60 KB at cfg4 with G = 8, 16-byte stores on 16-byte boundaries.  A pure store kernel of this
//     shape runs at 5.5 TB/s (0.358 ms per 16 steps at cfg4) against 4.3 TB/s for one 7.5 KB run per wave; waves of one
//     workgroup INTERLEAVING 1 KB pieces of the same run is the slow shape (3.3 - 3.9 TB/s);
//   * the writer's arithmetic: the stepper leaves a 16-bit word per cell holding each agent's 3-bit CLASS of the cell
//     (numbered in paint order: 0 background, 1 food, 2 own body, 3 own head, 4 other body, 5 other head, 6 border), the
//     writer extracts its agent's field and reads the plane's value from a 3 x 8 float table in LDS: two VALU instructions
//     and two LDS reads per float instead of ~14 VALU per 64-cell store group for the colour logic;
//   * the steppers issue NO global memory instruction in the steady state: rewards / flags go to LDS and are written by a
//     writer wave (the CU's vector memory pipeline is in order: a stepper's small stores queued behind the observation
//     stream of the whole CU stall the stepper at issue).
constexpr int GRP_MAX_SNAKES = 5;   // 3 bits per agent in a 16-bit word (double-buffered)
constexpr int GRP_MAX_SNAKES32 = 10; // ... in a 32-bit word (single-buffered: 4 bytes per cell is what the LDS has room for once)
constexpr int GRP_TAB_BYTES = 128;  // float tab[3][8] at the start of the workgroup's LDS
constexpr int GRP_CODE_SLACK = 640;  // bytes the writers may READ behind the last code array (grp_emit_cells: 5 x 64 codes)

// LDS of multi_rollout_group_kernel with G envs per workgroup: the colour table, the envs' blocks (multi_layout), the class
// code buffers (two of 16-bit words, WIDE: one of 32-bit words), the per-step output rows and the snakes' saved scalars.
// Fills in p's offsets; returns the total (the writers read up to GRP_CODE_SLACK bytes past a code array: what lies behind
// the last one must be this LDS).
__host__ __device__ inline int group_layout(MultiArgs &p, int G, bool wide)
{
    const int lds_env = multi_layout(p, false, 0), C = p.S * p.S;
    p.grp_code_bytes = ((wide ? 4 : 2) * C + 15) & ~15;
    const int nbuf = wide ? 1 : 2;
    p.grp_out_bytes = (16 * p.K + 1 + 15) & ~15;
    const int save_bytes = 32 * p.K; // grp_save: 8 ints per snake
    const int slack0 = (wide ? 2 : 1) * GRP_CODE_SLACK - G * (nbuf * p.grp_out_bytes + save_bytes), slack = slack0 > 0 ? slack0 : 0;
    p.grp_env0 = GRP_TAB_BYTES;
    p.grp_codes = p.grp_env0 + G * lds_env;
    p.grp_outs = p.grp_codes + nbuf * G * p.grp_code_bytes;
    p.grp_save = p.grp_outs + nbuf * G * p.grp_out_bytes;
    return GRP_TAB_BYTES + G * (lds_env + nbuf * p.grp_code_bytes + nbuf * p.grp_out_bytes + save_bytes) + slack;
}

// (sum over a < K of 8^a): a 3-bit value replicated into the K agents' fields
__device__ __forceinline__ u32 grp_rep(int K) { return (u32)(((1ull << (3 * K)) - 1ull) / 7ull); }

// per-agent classes of every cell of the env in LDS -> codes[] (the wave that owns the env's state).  _observe_agent
// :268-281 paints food, own body, own head, other bodies, other heads, then the border (:183-186): a later layer wins, so
// the class of a cell for an agent is the LAST layer that covers it — with the classes numbered in paint order, the
// maximum over the layers.  Bodies first (every cell), then the K head cells are raised.
// bit k of the lane's mask <=> cell lane + 64 k lies on the border ring (:183-186): a property of the grid, worked out once
// per kernel (class_write paints the ring last, over whatever sits there)
__device__ __forceinline__ u64 border_bits(const Ctx &cx)
{
    const int S = cx.S, C = cx.C;
    u64 m = 0;
    for (int k = 0; k < cx.cpl; ++k) {
        const int c = cx.lane + 64 * k, y = div_size(c, cx.rcpS), x = c - y * S;
        if (c < C && (y == 0 || x == 0 || y == S - 1 || x == S - 1)) m |= 1ull << k;
    }
    return m;
}

template <typename CT>
__device__ __forceinline__ void class_write(const Ctx &cx, int hc, CT *codes, u64 border)
{
    const int C = cx.C, K = cx.K, lane = cx.lane;
    const u32 REP = grp_rep(K), REP4 = 4u * REP, REP5 = 5u * REP, REP6 = 6u * REP;
    // Five 64-cell rows at a time: for each snake its clock and head cell come out of lanes 0 .. K-1 (readlane), its five
    // body cells are read TOGETHER (one LDS round trip per snake and block, not per cell), "head on this cell" is a compare
    // with the K head cells — no head map, no pass over the border afterwards, one fence at the end — and the code of a cell
    // is a few selects; only a cell with several snakes on it branches.  (The first form read cell by cell, kept a
    // head-owner map in LDS and painted the border in a second pass: 11 000 cycles of a stepper's 67 000 per call at cfg4,
    // tools/multi_timeline.py.)
    constexpr int U = 5;
    const int myT = lane < K ? cx.tclk[lane] : 0, myH = lane < K ? hc : -1;
    for (int k0 = 0; k0 < cx.cpl; k0 += U) {
        int cc[U];
        u32 bm[U], hm[U], fd[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            cc[u] = min(lane + 64 * (k0 + u), C - 1); // (rows past the grid: the last cell again, not stored)
            bm[u] = hm[u] = 0;
            fd[u] = cx.food[cc[u]];
        }
        for (int s = 0; s < K; ++s) {
            const int T = lane_value(myT, s), H = lane_value(myH, s);
            const unsigned short *b = cx.body + s * C;
            u32 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = b[cc[u]];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                bm[u] |= (u32)((int)(v[u] & VMASK) > T) << s;
                hm[u] |= (u32)(cc[u] == H) << s;
            }
        }
        const u32 bb = (u32)(border >> k0);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = lane + 64 * (k0 + u);
            const u32 occ = bm[u] | hm[u];
            const int first = max(__ffs((int)occ) - 1, 0);
            // one snake: the others see 4 (body) / 5 (head), the snake itself 2 / 3
            u32 code = occ == 0 ? (fd[u] != 0 ? REP : 0u) : ((hm[u] ? REP5 : REP4) ^ (6u << (3 * first)));
            const bool ring = ((bb >> u) & 1u) != 0;
            if ((occ & (occ - 1u)) != 0 && !ring) { // several snakes on the cell (hand-made states, heads that have just run into something)
                code = 0;
                for (int a = 0; a < K; ++a) {
                    const u32 others = ~(1u << a);
                    code |= ((hm[u] & others) ? 5u : (bm[u] & others) ? 4u : ((hm[u] >> a) & 1u) ? 3u : 2u) << (3 * a);
                }
            }
            if (ring) code = REP6;
            if (c < C) codes[c] = (CT)code;
        }
    }
    wave_lds_sync();
}

typedef __attribute__((address_space(1))) float grp_gfloat;     // a pointer KNOWN to be global memory: global_store, not flat_store
// Probe build (make -C wurm_amd/csrc probe -> libwurm_hip_probe.so; tools/multi_group_probe.py): WURM_MULTI_GROUP_VARIANT
// bit 2 drops the writers' stores, bit 3 the steppers' transition — which half bounds the launch.  Results are wrong by
// construction with either bit; the shipped library compiles the switches out.
#ifdef WURM_GROUP_PROBE
#define WURM_PROBE(probe, shipped) (probe)
#else
#define WURM_PROBE(probe, shipped) (shipped)
#endif

__device__ __forceinline__ float grp_tab(const float *tabp, u32 off) { return *(const float *)((const unsigned char *)tabp + off); }

// The writer: one CELL per lane — its class is extracted ONCE and looked up in the three planes' rows of the table (three
// LDS reads off one address register), three 4-byte stores per 64 cells (256 contiguous bytes each: the same stream rate as
// 16-byte stores in tools/microbench/store_runs.hip, and nothing to align or peel).  Per 64 cells: 1 + 3 LDS reads, 2 VALU,
// 3 stores.  (Round 4 first built the 16-byte-store form VERDICT r03 asked for — lane = 4 consecutive floats of a plane,
// planes peeled to 16-byte boundaries: it extracts every class three times, once per plane, and pays ~60 instructions of
// alignment / predication per plane; its writer waves alone took 17.6 us per step against 10.2 us for this form, the
// launch 26.0 against 24.0 us per step on the same box: profiles/r04_multi_group_probe.txt.)
template <int U, typename CT>
__device__ __forceinline__ void grp_emit_cells_chunk(grp_gfloat *blk, const CT *cg, const float *tab, u32 sh, int C,
                                                     int c0, bool stores)
{
    u32 off[U];
#pragma unroll
    for (int u = 0; u < U; ++u) off[u] = __builtin_amdgcn_ubfe((u32)cg[c0 + 64 * u], sh, 3u) << 2; // (reads past C: see GRP_CODE_SLACK)
    float r[U], g[U], b[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        r[u] = grp_tab(tab, off[u]);
        g[u] = grp_tab(tab + 8, off[u]);
        b[u] = grp_tab(tab + 16, off[u]);
    }
    typedef __attribute__((address_space(1))) char gchar;
#pragma unroll
    for (int u = 0; u < U; ++u) { // (unsigned 32-bit BYTE offsets off the uniform block pointer: global_store with an SGPR base)
        const u32 c = (u32)(c0 + 64 * u), o = 4u * c, pl = 4u * (u32)C;
        if (c < (u32)C && stores) {
            *(grp_gfloat *)((gchar *)blk + o) = r[u];
            *(grp_gfloat *)((gchar *)blk + (o + pl)) = g[u];
            *(grp_gfloat *)((gchar *)blk + (o + 2u * pl)) = b[u];
        }
    }
}

template <typename CT>
__device__ __forceinline__ void grp_emit_cells(grp_gfloat *blk, const CT *cg, const float *tab, u32 sh, int C, int lane,
                                               bool stores)
{
    const int cpl = (C + 63) >> 6;
    int k = 0;
    for (; k + 5 <= cpl; k += 5) grp_emit_cells_chunk<5, CT>(blk, cg, tab, sh, C, lane + 64 * k, stores);
    for (; k < cpl; ++k) grp_emit_cells_chunk<1, CT>(blk, cg, tab, sh, C, lane + 64 * k, stores);
}

// tab[plane][class]: the reference's colours / 255 (true divisions, as `.to(dtype) / 255` :281); the caller synchronises
__device__ __forceinline__ void grp_table_init(float *tab, int tid)
{
    if (tid < 24) {
        const int pl = tid >> 3, cls = tid & 7;
        const float G1 = 192.0f / 255.0f, G2 = 96.0f / 255.0f;
        float v = 0.0f;
        if (cls == 0) v = 1.0f;                              // background (255, 255, 255)
        else if (cls == 1) v = pl == 0 ? 1.0f : 0.0f;        // food (255, 0, 0)
        else if (cls == 2) v = pl == 1 ? G2 : 0.0f;          // own body (0, 96, 0)
        else if (cls == 3) v = pl == 1 ? G1 : 0.0f;          // own head (0, 192, 0)
        else if (cls == 4) v = pl == 2 ? G2 : 0.0f;          // other body (0, 0, 96)
        else if (cls == 5) v = pl == 2 ? G1 : 0.0f;          // other head (0, 0, 192)
        tab[pl * 8 + cls] = v;                               // 6: border (0, 0, 0)
    }
}

// per-call kernels: the nwaves waves of a workgroup write the observations of its nG consecutive envs (first env0), wave w
// the whole run of agent w (w + nwaves, ...); obs: this call's (K, N, 3 S^2) block
__device__ __forceinline__ void grp_emit_group(const MultiArgs &p, float *obs, long long env0, int nG, int wave, int nwaves,
                                               const unsigned char *codes0, int code_stride, const float *tab, int lane)
{
    const int C = p.S * p.S, K = p.K;
    // wave -> (agent, part of the envs): one agent after the other while there are at most as many waves as agents, else
    // nwaves / K waves per agent, each with its own contiguous part of the agent's run
    const int parts = nwaves > K ? nwaves / K : 1, part = nwaves > K ? wave / K : 0;
    const int g0 = part * nG / parts, g1 = (part + 1) * nG / parts;
    for (int a = nwaves > K ? wave % K : wave; a < K && part < parts; a += nwaves > K ? K : nwaves) {
        grp_gfloat *const run = (grp_gfloat *)uniform64((long long)(obs + ((long long)a * p.N + env0) * p.obs_elems));
        const u32 sh = 3u * (u32)a;
        for (int g = g0; g < g1; ++g) {
            const unsigned short *cg = (const unsigned short *)(codes0 + (size_t)g * code_stride);
            grp_emit_cells<unsigned short>(run + g * 3 * C, cg, tab, sh, C, lane, true);
        }
    }
}

// Work sharing between the waves of a group: the K * nG (agent, env) observation blocks of a step are ITEMS handed out by
// an LDS counter, in address order (agent-major: consecutive items are consecutive envs of one agent's run).  The writer
// waves take items from the barrier of step t on; a stepper wave joins in once it has finished its own step t + 1 — what
// the store path takes grows with the number of waves that have stores in flight (tools/microbench/store_window.hip: the
// same bytes at 4.7 / 5.2 / 5.7 / 6.3 TB/s from 4 / 8 / 16 / 32 storing waves per CU), and the steppers are idle for half
// of a step's period otherwise.
template <typename CT>
__device__ __forceinline__ void grp_take_items(const MultiArgs &p, int *ctr, const unsigned char *cbuf, long long t, long long env0,
                                               int nG, const float *tab, int lane, bool stores)
{
    const int C = p.S * p.S, K = p.K, items = K * nG;
    for (;;) {
        int i = 0;
        if (lane == 0) i = atomicAdd(ctr, 1);
        i = uniform(i);
        if (i >= items) break;
        const int a = i / nG, g = i - a * nG;
        grp_gfloat *const blk = (grp_gfloat *)uniform64((long long)(p.obs + ((t * K + a) * p.N + env0 + g) * p.obs_elems));
        grp_emit_cells<CT>(blk, (const CT *)(cbuf + (size_t)g * p.grp_code_bytes), tab, 3u * (u32)a, C, lane, stores);
    }
}

// Per-snake scalars of an env while its stepper wave works on another one (EPS > 1): 8 ints per snake in LDS
__device__ __forceinline__ void grp_save(int *sv, int lane, int K, const Snake &sn, bool col_dirty, int hc0)
{
    if (lane < K) {
        int *q = sv + 8 * lane;
        q[0] = sn.hc; q[1] = sn.L; q[2] = (int)sn.done | ((int)sn.boosted << 1) | ((int)col_dirty << 2); q[3] = (int)sn.orient;
        q[4] = (int)(unsigned short)sn.col[0] | ((int)(unsigned short)sn.col[1] << 16); q[5] = (int)sn.col[2]; q[6] = hc0;
    }
}

__device__ __forceinline__ void grp_restore(const int *sv, int lane, int K, Snake &sn, bool &col_dirty, int &hc0)
{
    sn.hc = -1; sn.L = 0; sn.done = true; sn.orient = 0; sn.boosted = false; sn.col[0] = sn.col[1] = sn.col[2] = 0;
    col_dirty = false; hc0 = -1;
    if (lane < K) {
        const int *q = sv + 8 * lane;
        sn.hc = q[0]; sn.L = q[1]; sn.done = (q[2] & 1) != 0; sn.boosted = (q[2] & 2) != 0; col_dirty = (q[2] & 4) != 0;
        sn.orient = q[3]; sn.col[0] = (short)(q[4] & 0xffff); sn.col[1] = (short)(q[4] >> 16); sn.col[2] = (short)q[5]; hc0 = q[6];
    }
}

// G envs, G / EPS stepper waves (EPS envs each, one after the other within a step: the transition of one env is a chain of
// dependent LDS operations that takes a wave ~5 us of the ~22 us the step's observations need on the store path, and a
// stepper's register budget is what limits the waves per SIMD — so fewer, fuller stepper waves), W writer waves.
// WIDE (6 .. 10 snakes): 32-bit class words and ONE code / output buffer — the steppers wait for the writers to be done with
// step t - 1 before they write the codes of step t (a second barrier per step; the transition itself still runs beside the
// writers: the speeds.py shape, 10 snakes on 36 x 36, has 41 KB of LDS per env with one 32-bit buffer and four envs per CU).
// INJ = false: as multi_rollout_kernel — the launch draws its own random outcomes, the injected-outcome branches fold away.
template <int G, int W, int EPS, int OCC, bool WIDE = false, bool INJ = true, int KT = 0, int ST = 0>
__global__ __launch_bounds__(64 * (G / EPS + W)) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) void multi_rollout_group_kernel(MultiArgs p_in)
{
    MultiArgs p = p_in;
    if (!INJ) p.has_inj = p.has_rinj = 0;
    p.obs_mode = WURM_OBS_DEFAULT; // (what multi_group_shape requires: a constant here)
    if (KT > 0 && ST > 0) { // (shape_constants: K, S and with them the whole LDS layout of the group as constants)
        p.K = KT;
        p.S = ST;
        p.obs_elems = 3ll * ST * ST;
        (void)group_layout(p, G, WIDE);
    }
    typedef typename std::conditional<WIDE, u32, unsigned short>::type CT;
    constexpr int NSW = G / EPS; // stepper waves
    // SHARE: the (agent, env) blocks of a step are handed out by an LDS counter and the steppers take some too
    // (grp_take_items) — measured to pay only where the writer waves alone cannot keep up (fewer writers than agents:
    // 8 / 2 / 1 / 5 went from 1.93 to 1.64 ms per 64 steps at cfg4); with one writer per agent the steppers queueing on the
    // store path lengthen the step (8 / 4 / 1 / 6: 1.61 -> 1.96 ms), so each writer keeps its own agent's run there
    constexpr bool SHARE = W < 4 && !WIDE;
    const int wave = uniform((int)(threadIdx.x >> 6)), lane = (int)(threadIdx.x & 63u);
    const long long env0 = xcd_block(blockIdx.x, gridDim.x) * G;
    if (env0 >= p.N) return;
    const int nG = (int)min((long long)G, p.N - env0);
    const int C = p.S * p.S, K = p.K;
    const long long KN = (long long)K * p.N;
    float *const tab = (float *)wurm_multi_lds;
    unsigned char *const codes0 = wurm_multi_lds + p.grp_codes, *const outs0 = wurm_multi_lds + p.grp_outs;
    int *const ctr = (int *)(wurm_multi_lds + 96); // two item counters (one per buffer) behind the 24 floats of the table
    grp_table_init(tab, (int)threadIdx.x);
    if (SHARE && threadIdx.x < 2) ctr[threadIdx.x] = 0;
    __syncthreads();
#ifdef WURM_GROUP_PROBE
    const bool stores = !(p.grp_variant & 4);
#else
    constexpr bool stores = true;
#endif

    if (wave >= NSW) { // ---- a writer: the observations of step t from buffer t & 1, handed over by the barrier of step t
        const int w = wave - NSW;
        for (long long t = 0; t < p.T; ++t) {
            if (WIDE) workgroup_handoff();   // (the steppers may now overwrite the single buffer: the writers are done with t - 1)
            workgroup_handoff();
            const int buf = WIDE ? 0 : (int)(t & 1);
            const unsigned char *cbuf = codes0 + (size_t)buf * G * p.grp_code_bytes;
            if (w == 0) { // the steppers' per-step outputs: rows (j K + s) of G consecutive envs each
                const unsigned char *obuf = outs0 + (size_t)buf * G * p.grp_out_bytes;
                float *of = p.am_f32 + t * 3 * KN;
                uint8_t *ob = p.am_u8 + t * 4 * KN;
                for (int i = lane; i < 3 * K * G; i += 64) {
                    const int g = i % G, js = i / G;
                    if (g < nG) of[(long long)js * p.N + env0 + g] = ((const float *)(obuf + (size_t)g * p.grp_out_bytes))[js];
                }
                for (int i = lane; i < 4 * K * G; i += 64) {
                    const int g = i % G, js = i / G;
                    if (g < nG) ob[(long long)js * p.N + env0 + g] = (obuf + (size_t)g * p.grp_out_bytes)[12 * K + js];
                }
                if (lane < nG) p.all_done[t * p.N + env0 + lane] = (obuf + (size_t)lane * p.grp_out_bytes)[16 * K];
            }
            if (SHARE) {
                grp_take_items<CT>(p, ctr + buf, cbuf, t, env0, nG, tab, lane, stores);
            } else {
                // writer wave -> (agent, part of the group's envs): one agent after the other while there are at most as many
                // waves as agents, else W / K waves per agent, each with its own contiguous part of the run
                const int parts = W > K ? W / K : 1, part = W > K ? w / K : 0;
                const int g0 = part * G / parts, g1 = min((part + 1) * G / parts, nG);
                for (int a = W > K ? w % K : w; a < K && part < parts; a += W > K ? K : W) {
                    grp_gfloat *const run = (grp_gfloat *)uniform64((long long)(p.obs + ((t * K + a) * p.N + env0) * p.obs_elems));
                    for (int g = g0; g < g1; ++g)
                        grp_emit_cells<CT>(run + g * 3 * C, (const CT *)(cbuf + (size_t)g * p.grp_code_bytes), tab, 3u * (u32)a, C, lane,
                                           stores);
                }
            }
        }
        return;
    }

    // ---- a stepper: envs env0 + wave * EPS + e, e < EPS; the scalars of the envs it is not working on wait in LDS
    if (env0 + wave * EPS >= p.N) { // a ragged last group: nothing to step — the barriers are the workgroup's, and it writes
        for (long long t = 0; t < p.T; ++t) {
            if (WIDE) workgroup_handoff();
            workgroup_handoff();
            const int buf = WIDE ? 0 : (int)(t & 1);
            if (SHARE) grp_take_items<CT>(p, ctr + buf, codes0 + (size_t)buf * G * p.grp_code_bytes, t, env0, nG, tab, lane, stores);
        }
        return;
    }
    const bool snake = lane < K;
    const u64 ring = border_bits(make_ctx(p, 0, p.grp_env0));
    int *const save0 = (int *)(wurm_multi_lds + p.grp_save);
    for (int e = 0; e < EPS; ++e) {
        const int g = wave * EPS + e;
        const long long env = env0 + g;
        if (env >= p.N) break;
        const Ctx cx = make_ctx(p, g, p.grp_env0);
        const long long agent = env * K + lane;
        // the caller's compact mirror (wurm_multi_rollout_resident), when it describes the state: 2 K + 1 bytes per cell
        // instead of (1 + 2 K) fp32 planes — 23 MB against 92 at cfg4, 110 against 446 at the speeds.py shape, which is a
        // tenth of a 4-step launch there
        if (p.resident != nullptr && p.resident_valid)
            (void)mirror_load(cx, p.resident + env * mirror_env_bytes(K, C), lane, 64, [] { wave_lds_sync(); }, false);
        else
            (void)load_env(cx, p.foods + env * C, p.heads + env * K * C, p.bodies + env * K * C);
        Snake sn;
        sn.hc = snake ? cx.hcell[lane] : -1;
        sn.L = snake ? cx.lmax[lane] : 0;
        sn.done = snake ? p.dones[agent] != 0 : true;
        sn.orient = snake ? p.orientations[agent] : 0;
        sn.boosted = false;
        load_colour(p, agent, snake, sn);
        grp_save(save0 + g * 8 * K, lane, K, sn, false, sn.hc);
    }
    wave_lds_sync();

    for (long long t = 0; t < p.T; ++t) {
        const u64 call = p.call + 2ull * (u64)t;
        for (int e = 0; e < EPS; ++e) {
            const int g = wave * EPS + e;
            const long long env = env0 + g;
            if (env >= p.N) break; // (a ragged last group: the barrier below is still the workgroup's)
            const Ctx cx = make_ctx(p, g, p.grp_env0);
            const u64 env_id = (u64)(p.env_offset + env);
            Snake sn;
            bool col_dirty;
            int hc0;
            grp_restore(save0 + g * 8 * K, lane, K, sn, col_dirty, hc0);
            if ((t & 63) == 0) { // actions: 64 steps at a time into LDS (see multi_rollout_kernel)
                const int nt = (int)min((long long)64, p.T - t);
                wave_lds_sync();
                for (int i = lane; i < nt * K; i += 64) {
                    const int j = i / K, sidx = i - j * K;
                    const long long av = p.actions[(t + j) * KN + (long long)sidx * p.N + env];
                    cx.acts[i] = (unsigned char)(((int)(av % 4) + 4) | (av > 3 ? 8 : 0));
                }
                wave_lds_sync();
            }
            long long a = 0;
            if (snake) {
                const int b = cx.acts[(int)(t & 63) * K + lane];
                const int d4 = (b & 7) - 4;
                a = (b & 8) ? 4 + d4 : d4;
            }
            StepRes r = {};
            if (WURM_PROBE(!(p.grp_variant & 8), true)) multi_step_body(cx, p, env, env_id, call, a, sn, r, t * p.N * C, t * KN, t * p.N);
            // buffer t & 1 is free: the writers finished with it before they arrived at the barrier of step t - 1
            // (WIDE: the one buffer is free once the writers have passed the extra barrier of this step)
            if (WIDE && e == 0) {
                // this wave's transition is done: the rest of step t - 1's observations, with the writers — then the one
                // buffer is free for this step's codes
                if (SHARE && t > 0) grp_take_items<CT>(p, ctr, codes0, t - 1, env0, nG, tab, lane, stores);
                workgroup_handoff();
            }
            const int buf = WIDE ? 0 : (int)(t & 1);
            if (SHARE && e == 0 && wave == 0 && lane == 0) ctr[buf] = 0; // (everybody is done with the items this counter handed out last)
            unsigned char *obuf = outs0 + ((size_t)buf * G + g) * p.grp_out_bytes;
            if (snake) {
                float *f = (float *)obuf;
                f[lane] = r.reward;
                f[K + lane] = r.foodcons;
                f[2 * K + lane] = (float)sn.L;
                unsigned char *b = obuf + 12 * K;
                b[lane] = (unsigned char)sn.done;
                b[K + lane] = (unsigned char)sn.boosted;
                b[2 * K + lane] = (unsigned char)r.snakecol;
                b[3 * K + lane] = (unsigned char)r.edgecol;
            }
            if (lane == 0) obuf[16 * K] = (unsigned char)r.all_done;
            class_write<CT>(cx, sn.hc, (CT *)(codes0 + ((size_t)buf * G + g) * p.grp_code_bytes), ring);
            if (lane == 0) ((int *)(save0 + g * 8 * K))[7] = (int)r.all_done;   // (slot 7 of snake 0: for the reset below)
            grp_save(save0 + g * 8 * K, lane, K, sn, col_dirty, hc0);
            wave_lds_sync();
        }
        // (two buffers: the codes of step t are in place — the rest of step t - 1's observations, with the writers)
        if (SHARE && !WIDE && t > 0) {
            const int pb = (int)((t - 1) & 1);
            grp_take_items<CT>(p, ctr + pb, codes0 + (size_t)pb * G * p.grp_code_bytes, t - 1, env0, nG, tab, lane, stores);
        }
        // the writers start on step t NOW: the reset that follows the step (a rebuilt env costs as much as a whole
        // transition) runs beside them, not in front of them
        workgroup_handoff();
        for (int e = 0; e < EPS; ++e) {
            const int g = wave * EPS + e;
            const long long env = env0 + g;
            if (env >= p.N) break;
            const Ctx cx = make_ctx(p, g, p.grp_env0);
            const u64 env_id = (u64)(p.env_offset + env);
            const long long agent = env * K + lane;
            Snake sn;
            bool col_dirty;
            int hc0;
            grp_restore(save0 + g * 8 * K, lane, K, sn, col_dirty, hc0);
            const bool rebuild = uniform(((const int *)(save0 + g * 8 * K))[7]) != 0;
            rebase_clocks(cx);

            // reset(dones['__all__']) (:771-836)
            if (snake && sn.done) sn.L = 0;
            if (rebuild) sn.done = false;             // :798
            if (snake) col_dirty |= reroll_colour(p, agent, sn.done, env_id, call + 1ull, t * KN, sn);
            const bool respawn = p.cfg.respawn_any && ballot(snake && sn.done) != 0;
            if (rebuild || respawn) {
                bool orient_dirty = false;
                multi_reset_grid(cx, p, env, env_id, call + 1ull, rebuild, respawn, sn, orient_dirty, t * KN, t * p.N);
            }
            grp_save(save0 + g * 8 * K, lane, K, sn, col_dirty, hc0);
            wave_lds_sync();
        }
    }

    for (int e = 0; e < EPS; ++e) {
        const int g = wave * EPS + e;
        const long long env = env0 + g;
        if (env >= p.N) break;
        const Ctx cx = make_ctx(p, g, p.grp_env0);
        const long long agent = env * K + lane;
        Snake sn;
        bool col_dirty;
        int hc0;
        grp_restore(save0 + g * 8 * K, lane, K, sn, col_dirty, hc0);
        if (snake) {
            p.dones[agent] = (uint8_t)sn.done;
            p.orientations[agent] = sn.orient;
            if (p.boost_state) p.boost_state[agent] = (uint8_t)sn.boosted;
            if (col_dirty) {
                p.colours[agent * 3] = sn.col[0];
                p.colours[agent * 3 + 1] = sn.col[1];
                p.colours[agent * 3 + 2] = sn.col[2];
            }
            cx.hcell[lane] = sn.hc;
        }
        wave_lds_sync();
        // the food plane is written whole (the original bits are not kept across the launch: ~cur marks every cell changed)
        u64 cur = 0;
        for (int k = 0; k < cx.cpl; ++k) {
            const int c = lane + 64 * k;
            if (c < C && cx.food[c]) cur |= 1ull << k;
        }
        // the fp32 planes: not at all while the mirror is lazy; whole when the state came from the mirror (the DIRTY marks
        // only cover what was written since a load from the planes); else what changed
        const bool from_mirror = p.resident != nullptr && p.resident_valid;
        if (!(p.resident != nullptr && p.resident_lazy))
            store_env(cx, p.foods + env * C, p.heads + env * K * C, p.bodies + env * K * C, ~cur, hc0, sn.hc, from_mirror);
        if (p.resident != nullptr) {
            (void)rebase_clocks(cx);
            mirror_store(cx, p.resident + env * mirror_env_bytes(K, C), lane, 64, sn.hc, (snake && !sn.done) ? sn.L : 0,
                         [] { wave_lds_sync(); });
            wave_lds_sync();
        }
    }
}

// check_consistency (:733-769) -> per-env bitmask.  Every plane is read once: the per-cell count of snakes (overlap test)
// is kept in LDS, one byte per cell; a cell belongs to one lane, so plain read-modify-writes.
// The env's planes are walked as ITEMS of (snake, 11 rows of 64 cells), double-buffered in registers: the 16 loads of item
// i + 1 are in flight while item i is reduced (round 2 issued a snake's loads, waited, reduced, and only then touched the
// next snake: five dependent memory round trips per env at K = 4, 2.6 TB/s).
constexpr int MCHK_U = 11; // 25 x 25: a snake is one item, 36 x 36: two (11 + 10 rows)

__device__ __forceinline__ void mchk_load(const float *__restrict__ hp, const float *__restrict__ bp, int k0, int lane, int C,
                                          float (&h)[MCHK_U], float (&b)[MCHK_U])
{
#pragma unroll
    for (int j = 0; j < MCHK_U; ++j) {
        const int c = lane + 64 * (k0 + j);
        const bool in = c < C;
        h[j] = in ? hp[c] : 0.0f;
        b[j] = in ? bp[c] : 0.0f;
    }
}

__global__ __launch_bounds__(256) void multi_check_kernel(MultiArgs p)
{
    const int wave = (int)(threadIdx.x >> 6), wpb = (int)(blockDim.x >> 6);
    const long long env = xcd_block(blockIdx.x, gridDim.x) * wpb + wave;
    if (env >= p.N) return;
    const int S = p.S, C = S * S, K = p.K, lane = (int)(threadIdx.x & 63u), cpl = (C + 63) >> 6;
    unsigned char *cnt = wurm_multi_lds + (size_t)wave * p.lds_per_wave;
    const float *foodp = p.foods + env * C;
    const float *heads = p.heads + env * K * C, *bodies = p.bodies + env * K * C;
    const int nchunk = (cpl + MCHK_U - 1) / MCHK_U, nitems = K * nchunk;
    float hA[MCHK_U], bA[MCHK_U], hB[MCHK_U], bB[MCHK_U];
    mchk_load(heads, bodies, 0, lane, C, hA, bA); // item 0 is in flight while the food plane is checked
    uint32_t m = 0;
    int badf = 0;
    for (int k = 0; k < cpl; ++k) {
        int c = lane + 64 * k;
        if (c < C) {
            const float f = foodp[c];
            badf |= !(f == 0.0f || f == 1.0f);
            cnt[c] = 0;
        }
    }
    const bool bad_food = ballot(badf != 0) != 0;
    bool any_alive = false;
    int hs = 0, bs = 0, bm = 0, hb = 0, hf = 0, nz = 0; // per-lane partials of the current snake
    auto reduce_item = [&](int item, const float (&h)[MCHK_U], const float (&b)[MCHK_U]) {
        const int s = item / nchunk, k0 = (item - s * nchunk) * MCHK_U;
#pragma unroll
        for (int j = 0; j < MCHK_U; ++j) {
            const int c = lane + 64 * (k0 + j);
            if (c < C) {
                const int hi = __float2int_rn(h[j]), bi = __float2int_rn(b[j]);
                hs += hi; bs += bi; hb += hi * bi;
                if (hi != 0) hf += hi * __float2int_rn(foodp[c]); // at the head cells only (one per snake)
                bm = max(bm, bi);
                nz |= (h[j] != 0.0f) || (b[j] != 0.0f);
                if (b[j] > 1e-6f) cnt[c] += 1;
            }
        }
        if (k0 + MCHK_U >= cpl) { // the snake's last item: its verdict
            const bool dead = p.dones[env * K + s] != 0;
            if (dead) {
                if (ballot(nz != 0)) m |= WURM_MCHK_DEAD_NONZERO;
            } else {
                any_alive = true;
                const int t_hs = wave_sum_i32(hs), t_bs = wave_sum_i32(bs), t_hb = wave_sum_i32(hb), t_hf = wave_sum_i32(hf);
                const int t_bm = wave_max_i32(bm);
                if (t_hs != 1) m |= WURM_CHK_ONE_HEAD;
                if (!(t_bs > 0)) m |= WURM_CHK_HAS_SNAKE;
                if (t_bm != t_hb) m |= WURM_CHK_HEAD_AT_END;
                if (2 * t_bs != t_bm * (t_bm + 1)) m |= WURM_CHK_BODY_RANGE;
                if (!(t_bs >= 6)) m |= WURM_CHK_MIN_LENGTH;
                if (t_hf != 0) m |= WURM_CHK_HEAD_ON_FOOD;
            }
            hs = bs = bm = hb = hf = nz = 0;
        }
    };
    auto issue = [&](int item, float (&h)[MCHK_U], float (&b)[MCHK_U]) {
        const int s = item / nchunk, k0 = (item - s * nchunk) * MCHK_U;
        mchk_load(heads + s * C, bodies + s * C, k0, lane, C, h, b);
    };
    for (int it = 0; it < nitems; it += 2) {
        if (it + 1 < nitems) issue(it + 1, hB, bB);
        reduce_item(it, hA, bA);
        if (it + 2 < nitems) issue(it + 2, hA, bA);
        if (it + 1 < nitems) reduce_item(it + 1, hB, bB);
    }
    if (any_alive && bad_food) m |= WURM_CHK_FOOD_VALUE; // reported per living snake by the loop this replaces
    int over = 0;
    for (int k = 0; k < cpl; ++k) {
        int c = lane + 64 * k;
        if (c < C) over |= cnt[c] > 1;
    }
    if (ballot(over != 0)) m |= WURM_MCHK_OVERLAP;
    if (lane == 0) p.err[env] = m;
}

__global__ void multi_colours_kernel(short *colours, long long N, int K, int fixed, u64 seed, u64 call, long long env_offset)
{
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * K) return;
    long long env = i / K;
    int s = (int)(i - env * K);
    u64 env_id = fixed ? 0xffffffffull : (u64)(env_offset + env);
    short col[3];
    colour_from_words(rng_words(seed, call, env_id, RNG_COLOUR, (u32)s), col);
    colours[i * 3] = col[0];
    colours[i * 3 + 1] = col[1];
    colours[i * 3 + 2] = col[2];
}

// ------------------------------------------------------------------------------------------------ host side

__host__ __device__ inline int multi_layout(MultiArgs &p, bool need_img, int need_snap)
{
    const int C = p.S * p.S, K = p.K;
    int off = 12 * K;                      // hcell, lmax, tclk
    p.off_col = off; off += 16 * K;        // colf
    off = (off + 15) & ~15;
    p.off_body = off; off += 2 * K * C;
    off = (off + 15) & ~15;
    p.off_food = off; off += C;
    off = (off + 15) & ~15;
    p.off_occ = off; off += C;
    off = (off + 15) & ~15;
    p.off_hmap = off; off += C;
    off = (off + 15) & ~15;
    p.off_img = off; if (need_img) off += 16 * (2 * K + 3); // partial_n: the pixel table (pixel_table), 16 bytes per cell code
    off = (off + 15) & ~15;
    p.off_snap = -1;
    if (need_snap) { p.off_snap = off; off += need_snap * ((2 * C + 15) & ~15); }
    p.off_acts = off; off += 64 * K;
    off = (off + 15) & ~15;
    p.off_tl = p.off_acts;
#ifdef WURM_TIMELINE
    p.off_tl = off; off += 256;
#endif
    p.lds_per_wave = (off + 15) & ~15;
    return p.lds_per_wave;
}

enum MKind { MK_STEP, MK_RESET, MK_OBSERVE, MK_CHECK, MK_ROLLOUT };

constexpr int LDS_MAX_BYTES = 160 * 1024; // per workgroup on CDNA4 (MI355X_MICROARCH.md)

// dynamic LDS beyond the default 64 KB of a launch needs the kernel's opt-in
static bool allow_lds(const void *kernel, size_t bytes)
{
    if (bytes <= 65536) return true;
    return hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess;
}

// fn_rng: the INJ = false instantiation where one is compiled; fn_shape: that with K = sk and S = ss compiled in (shape_constants)
struct GroupShape { int G, W, eps, occ; const void *fn; const void *fn_rng; const void *fn_shape = nullptr; int sk = 0, ss = 0; };

// The shape of multi_rollout_group_kernel that serves this rollout ('full' observations of at most 10 snakes, several steps,
// a large batch), or nullptr; q: p with the kernel's LDS layout filled in, bytes: its dynamic LDS.
static const GroupShape *multi_group_shape(const MultiArgs &p, MultiArgs &q, size_t &bytes)
{
    const bool snap = p.obs_mode == WURM_OBS_DEFAULT && p.K <= SNAP_MAX_SNAKES;
    if (!(snap && p.T > 1 && p.K <= GRP_MAX_SNAKES32 && p.N >= opt.multi_group_min_envs)) return nullptr;
    {
        const bool wide = p.K > GRP_MAX_SNAKES; // 32-bit class words, one buffer
        // large batches: G consecutive envs per workgroup, one linear observation run per agent (multi_rollout_group_kernel)
        q = p;
        auto total = [&](int G) { MultiArgs t = p; return group_layout(t, G, wide); };
        // shape: G envs, W writer waves, EPS envs per stepper wave, OCC waves per SIMD (option WURM_MULTI_GROUP_SHAPE =
        // 1000 G + 100 W + 10 EPS + OCC picks one of the compiled shapes; 0 = automatic: the first that fits)
        typedef GroupShape Shape;
        static const Shape wide_shapes[] = { // 6 .. 10 snakes (the first that fits)
            // (no shape-specialised form: with 10 snakes' loops unrolled the kernel spills 89 VGPRs at its 128 and measured 0.657 ms
            // per 4 steps of the speeds.py shape against 0.616 — profiles/r06_shape_kernels_ab.txt)
            {4, 10, 1, 4, (const void *)multi_rollout_group_kernel<4, 10, 1, 4, true>, (const void *)multi_rollout_group_kernel<4, 10, 1, 4, true, false>},
            {4, 5, 1, 4, (const void *)multi_rollout_group_kernel<4, 5, 1, 4, true>, nullptr},
            {2, 10, 1, 4, (const void *)multi_rollout_group_kernel<2, 10, 1, 4, true>, nullptr},
        };
        static const Shape shapes[] = {
            // (automatic: the first that fits.  Measured at cfg4 on four boxes, ms per 16- / 64-step launch: 8 / 2 / 1 / 5 — two
            // writers, the steppers sharing their work, 5 waves per SIMD at 96 VGPRs — 0.484-0.486 / 1.62-1.65 on every box;
            // 8 / 4 / 1 / 6 — a writer per agent, 6 waves per SIMD at 80 VGPRs with 130 bytes of scratch — 0.44-0.51 / 1.57-1.81
            // depending on the box and on how the allocator spills; 4 / 4 / 1 / 4: 0.51 / 1.66; 8 / 4 / 2 / 4: 1.79 per 64;
            // the two-wave kernel of round 3: 0.55 / 1.73-2.15 — profiles/r04_multi_group_probe.txt)
            {8, 2, 1, 5, (const void *)multi_rollout_group_kernel<8, 2, 1, 5>, (const void *)multi_rollout_group_kernel<8, 2, 1, 5, false, false>},
            {8, 4, 1, 6, (const void *)multi_rollout_group_kernel<8, 4, 1, 6>, (const void *)multi_rollout_group_kernel<8, 4, 1, 6, false, false>},
            // (round 6) BASELINE configs[3] — 4 snakes on 25 x 25 — takes THIS shape with K and S compiled in: 121 VGPRs, no
            // spilled VGPR, no scratch, and per 16- / 64-step launch on two boxes 0.4035 / 0.4278 and - / 1.5335 ms, against
            // 0.4135 / 0.4316 and - / 1.5306 for 8 / 2 / 1 / 5 specialised (34 spilled VGPRs, 140 bytes of scratch) and
            // 0.4305 / 0.4386 and - / 1.5724 for the generic 8 / 2 / 1 / 5 that shipped in round 5 (profiles/r06_group_shapes.txt)
            {4, 4, 1, 4, (const void *)multi_rollout_group_kernel<4, 4, 1, 4>, (const void *)multi_rollout_group_kernel<4, 4, 1, 4, false, false>,
             (const void *)multi_rollout_group_kernel<4, 4, 1, 4, false, false, 4, 25>, 4, 25},
            {8, 4, 2, 4, (const void *)multi_rollout_group_kernel<8, 4, 2, 4>, nullptr},
        };
        const Shape *sh = nullptr;
        for (const Shape &c : wide_shapes) {
            if (!wide) break;
            if (total(c.G) > LDS_MAX_BYTES) continue;
            if (opt.multi_group_shape == 0 || 1000 * c.G + 100 * c.W + 10 * c.eps + c.occ == opt.multi_group_shape) { sh = &c; break; }
        }
        // a shape whose specialised form serves this launch comes first (multi_launch takes fn_shape under the same conditions)
        if (!wide && opt.multi_group_shape == 0 && opt.multi_shape_kernels != 0 && !p.has_inj && !p.has_rinj)
            for (const Shape &c : shapes)
                if (c.fn_shape && c.sk == p.K && c.ss == p.S && total(c.G) <= LDS_MAX_BYTES) { sh = &c; break; }
        for (const Shape &c : shapes) {
            if (wide || sh) break;
            const bool fits = total(c.G) <= LDS_MAX_BYTES;
            if (opt.multi_group_shape ? (1000 * c.G + 100 * c.W + 10 * c.eps + c.occ == opt.multi_group_shape && fits)
                                      : (fits && (c.G == 4 || 2 * total(8) <= LDS_MAX_BYTES))) { sh = &c; break; }
        }
        if (!sh) return nullptr;
        const int G = sh->G;
        (void)group_layout(q, G, wide);
        bytes = (size_t)total(G);
        return sh;
    }
}

static int multi_launch(MKind kind, MultiArgs &p, void *stream)
{
    if (p.N == 0) return WURM_OK;
    // 'full' observations of at most 10 snakes go through a per-cell class code in LDS (observe_full_snap); rollouts
    // double-buffer it between a stepping and a writing wave (multi_rollout_kernel<true>)
    const bool snap = p.obs_mode == WURM_OBS_DEFAULT && p.K <= SNAP_MAX_SNAKES;
    const bool two = kind == MK_ROLLOUT && snap && p.T > 1;
    if (kind == MK_ROLLOUT) {
        // large batches: G consecutive envs per workgroup, one linear observation run per agent (multi_rollout_group_kernel)
        MultiArgs q;
        size_t bytes = 0;
        if (const GroupShape *sh = multi_group_shape(p, q, bytes)) {
            const dim3 gg((unsigned)((p.N + sh->G - 1) / sh->G)), bb(64 * (sh->G / sh->eps + sh->W));
            (void)hipGetLastError();
            const bool rng_g = !p.has_inj && !p.has_rinj;
            const void *kfn = (rng_g && opt.multi_shape_kernels != 0 && sh->fn_shape && p.K == sh->sk && p.S == sh->ss) ? sh->fn_shape
                            : (rng_g && sh->fn_rng) ? sh->fn_rng : sh->fn;
            if (!allow_lds(kfn, bytes)) return WURM_ERR_HIP;
            q.grp_variant = (int)opt.multi_group_variant;
            void *args[] = {&q};
            launch_count.fetch_add(1, std::memory_order_relaxed);
            if (hipLaunchKernel(kfn, gg, bb, args, bytes, (hipStream_t)stream) != hipSuccess) return WURM_ERR_HIP;
            p.resident_used = 1; // (the kernel keeps the caller's mirror, if one was given)
            return hipGetLastError() == hipSuccess ? WURM_OK : WURM_ERR_HIP;
        }
    }
    int lds = multi_layout(p, p.obs_mode == WURM_OBS_PARTIAL, snap ? (two ? 2 : 1) : 0);
    if (kind == MK_CHECK) lds = p.lds_per_wave = (p.S * p.S + 15) & ~15; // the checker keeps one byte per cell
    // One env's grids must fit the LDS of a CU.  Up to 64 KB is the default limit of a launch; beyond it the kernel is
    // opted into CDNA4's 160 KB per workgroup (one env per CU at a time, e.g. 32 snakes on 36 x 36: 90 KB).  Larger envs
    // (2 K S^2 + 3 S^2 bytes and change > 160 KB, e.g. S = 64 with K > 18) are UNSUPPORTED (DESIGN.md §5 deviation 10).
    if (lds > LDS_MAX_BYTES) return WURM_ERR_UNSUPPORTED;
    // few envs: one wave per workgroup so that they spread over all 256 CUs; from 2048 envs on 4 waves per workgroup
    // (8 workgroups per CU either way; the observation stream of 4096 envs measured ~5 % faster this way)
    const bool want_group = kind == MK_STEP && snap && p.K <= GRP_MAX_SNAKES && p.N >= opt.multi_group_min_envs;
    int wpb = (p.N < 2048 && !want_group) ? 1 : 4;
    if (two) { // one env per workgroup of two waves
        dim3 block2(128), grid2((unsigned)p.N);
        (void)hipGetLastError();
        const bool inj2 = p.has_inj || p.has_rinj;
        if (!allow_lds(inj2 ? (const void *)multi_rollout_kernel<true, true> : (const void *)multi_rollout_kernel<true, false>, (size_t)lds))
            return WURM_ERR_HIP;
        if (inj2) WURM_LAUNCH((multi_rollout_kernel<true, true>), grid2, block2, (size_t)lds, (hipStream_t)stream, p);
        else WURM_LAUNCH((multi_rollout_kernel<true, false>), grid2, block2, (size_t)lds, (hipStream_t)stream, p);
        return hipGetLastError() == hipSuccess ? WURM_OK : WURM_ERR_HIP;
    }
    if ((kind == MK_STEP || kind == MK_RESET || (kind == MK_OBSERVE && snap)) && lds * 4 > 65536 &&
        (snap || p.obs_mode == WURM_OBS_NONE)) {
        // an env too large for four per workgroup: one env per workgroup of four waves (multi_step_wg_kernel)
        (void)hipGetLastError();
        const dim3 g((unsigned)p.N), b(256);
        p.grp_variant = (int)opt.multi_group_variant;
        const void *kwg = (p.has_inj || p.has_rinj) ? (const void *)multi_step_wg_kernel<true>
                        : (kind == MK_STEP && opt.multi_shape_kernels != 0 && snap && p.K == 10 && p.S == 36)
                              ? (const void *)multi_step_wg_kernel<false, WURM_OBS_DEFAULT, 10, 36> // experiments/speeds.py
                              : (const void *)multi_step_wg_kernel<false>;
        const void *kf = kind == MK_STEP ? kwg
                       : kind == MK_RESET ? (const void *)multi_reset_wg_kernel : (const void *)multi_observe_wg_kernel;
        if (!allow_lds(kf, (size_t)lds)) return WURM_ERR_HIP;
        if (kind == MK_STEP) {
            void *kargs[] = {&p};
            launch_count.fetch_add(1, std::memory_order_relaxed);
            if (hipLaunchKernel(kwg, g, b, kargs, (size_t)lds, (hipStream_t)stream) != hipSuccess) return WURM_ERR_HIP;
        }
        else if (kind == MK_RESET) WURM_LAUNCH(multi_reset_wg_kernel, g, b, (size_t)lds, (hipStream_t)stream, p);
        else WURM_LAUNCH(multi_observe_wg_kernel, g, b, (size_t)lds, (hipStream_t)stream, p);
        return hipGetLastError() == hipSuccess ? WURM_OK : WURM_ERR_HIP;
    }
    while (wpb > 1 && lds * wpb > 65536) wpb >>= 1;
    size_t extra = 0;
    if (want_group && wpb == 4 && opt.multi_group_step_wpb != 0) {
        // large batches: 'full' observations through per-agent class codes and the colour table (class_write, grp_emit_cells).
        // Option WURM_MULTI_GROUP_STEP_WPB: 0 = off; 1 (and -1, automatic) = every wave writes its own env's K views, no
        // barrier; 4 / 8 = the workgroup's waves write one linear run per agent together, that many envs per workgroup.
        // One call's observations (123 MB at cfg4) are absorbed by the 256 MB Infinity Cache — the per-call launch is a chain
        // of latencies (tools/multi_timeline.py), not a stream, so the barrier of the shared form costs more than its
        // longer runs gain: 36.8 against 37.8 us per iteration at cfg4 (profiles/r04_multi_percall_timeline.txt).
        const long long mode = opt.multi_group_step_wpb < 0 ? 1 : opt.multi_group_step_wpb;
        extra = GRP_TAB_BYTES + GRP_CODE_SLACK;
        if (2 * (8 * (size_t)lds + extra) <= (size_t)LDS_MAX_BYTES && mode != 4 && mode != 1) wpb = 8;
        p.grp_emit = mode == 1 ? 2 : 1;
        p.grp_env0 = lds * wpb; // the table, behind the envs' blocks
    }
    dim3 block(64 * wpb), grid((unsigned)((p.N + wpb - 1) / wpb));
    size_t shmem = (size_t)lds * wpb + extra;
    hipStream_t st = (hipStream_t)stream;
    (void)hipGetLastError();
    const bool rng = !p.has_inj && !p.has_rinj;
    // the shape of the reference's multi-agent experiments (experiments/multiagent.py:79-86: 4 snakes on 25 x 25, partial_5)
    // has kernels with K, S and the crop radius as constants (WURM_MULTI_SHAPE_KERNELS = 0: the generic ones)
    const bool shape_4_25 = rng && opt.multi_shape_kernels != 0 && p.K == 4 && p.S == 25;
    const bool shape_4_25_5 = shape_4_25 && p.obs_mode == WURM_OBS_PARTIAL && p.obs_n == 5;
    const void *kstep = !rng ? (const void *)multi_step_kernel<true, -1>
                      : shape_4_25_5 ? (const void *)multi_step_kernel<false, WURM_OBS_PARTIAL, 4, 25, 5>
                      : shape_4_25 && p.obs_mode == WURM_OBS_DEFAULT ? (const void *)multi_step_kernel<false, WURM_OBS_DEFAULT, 4, 25>
                      // (the reference's own test shape, tests/test_multi_snake_env.py:340: 512 envs of 2 snakes on 12 x 12)
                      : rng && opt.multi_shape_kernels != 0 && p.K == 2 && p.S == 12 && p.obs_mode == WURM_OBS_DEFAULT
                            ? (const void *)multi_step_kernel<false, WURM_OBS_DEFAULT, 2, 12>
                      : p.obs_mode == WURM_OBS_DEFAULT ? (const void *)multi_step_kernel<false, WURM_OBS_DEFAULT>
                      : p.obs_mode == WURM_OBS_PARTIAL ? (const void *)multi_step_kernel<false, WURM_OBS_PARTIAL>
                      : (const void *)multi_step_kernel<false, WURM_OBS_NONE>;
    const void *kf = kind == MK_STEP ? kstep
                   : kind == MK_RESET ? (const void *)multi_reset_kernel
                   : kind == MK_OBSERVE ? (const void *)multi_observe_kernel
                   : kind == MK_CHECK ? (const void *)multi_check_kernel
                   : (p.has_inj || p.has_rinj) ? (const void *)multi_rollout_kernel<false, true>
                   : kind == MK_ROLLOUT && shape_4_25_5 ? (const void *)multi_rollout_kernel<false, false, WURM_OBS_PARTIAL, 4, 25, 5>
                   : p.obs_mode == WURM_OBS_PARTIAL ? (const void *)multi_rollout_kernel<false, false, WURM_OBS_PARTIAL>
                   : p.obs_mode == WURM_OBS_NONE ? (const void *)multi_rollout_kernel<false, false, WURM_OBS_NONE>
                   : (const void *)multi_rollout_kernel<false, false>;
    if (!allow_lds(kf, shmem)) return WURM_ERR_HIP;
    switch (kind) {
    case MK_STEP: {
        void *kargs[] = {&p};
        launch_count.fetch_add(1, std::memory_order_relaxed);
        if (hipLaunchKernel(kstep, grid, block, kargs, shmem, st) != hipSuccess) return WURM_ERR_HIP;
        break;
    }
    case MK_RESET: WURM_LAUNCH(multi_reset_kernel, grid, block, shmem, st, p); break;
    case MK_OBSERVE: WURM_LAUNCH(multi_observe_kernel, grid, block, shmem, st, p); break;
    case MK_CHECK: WURM_LAUNCH(multi_check_kernel, grid, block, shmem, st, p); break;
    case MK_ROLLOUT:
        if (p.has_inj || p.has_rinj) WURM_LAUNCH((multi_rollout_kernel<false, true>), grid, block, shmem, st, p);
        else if (shape_4_25_5) WURM_LAUNCH((multi_rollout_kernel<false, false, WURM_OBS_PARTIAL, 4, 25, 5>), grid, block, shmem, st, p);
        else if (p.obs_mode == WURM_OBS_PARTIAL) WURM_LAUNCH((multi_rollout_kernel<false, false, WURM_OBS_PARTIAL>), grid, block, shmem, st, p);
        else if (p.obs_mode == WURM_OBS_NONE) WURM_LAUNCH((multi_rollout_kernel<false, false, WURM_OBS_NONE>), grid, block, shmem, st, p);
        else WURM_LAUNCH((multi_rollout_kernel<false, false>), grid, block, shmem, st, p);
        break;
    }
    return hipGetLastError() == hipSuccess ? WURM_OK : WURM_ERR_HIP;
}

static long long multi_obs_elems(int mode, int n, int S)
{
    if (mode == WURM_OBS_DEFAULT) return 3ll * S * S;
    if (mode == WURM_OBS_PARTIAL && n >= 0) return 3ll * (2 * n + 1) * (2 * n + 1);
    return 0;
}

static int multi_check_args(long long N, int K, int S, int mode, int n, const void *obs)
{
    if (N < 0 || K < 1 || S < 3) return WURM_ERR_INVALID_ARG;
    if (K > 64 || S > 64) return WURM_ERR_UNSUPPORTED;
    if (mode != WURM_OBS_NONE) {
        if (multi_obs_elems(mode, n, S) == 0) return WURM_ERR_INVALID_ARG;
        if (N > 0 && obs == nullptr) return WURM_ERR_INVALID_ARG;
    }
    return WURM_OK;
}

} // namespace wurm

using namespace wurm;

extern "C" {

int64_t wurm_multi_obs_elems(int obs_mode, int obs_n, int size) { return multi_obs_elems(obs_mode, obs_n, size); }

int wurm_multi_step(float *foods, float *heads, float *bodies, uint8_t *dones, int64_t *orientations,
                    const int64_t *actions, uint8_t *boost_this_step, float *rewards, uint8_t *snake_collision,
                    uint8_t *edge_collision, float *food_consumed, float *sizes, uint8_t *all_done,
                    const int16_t *colours, float *obs, int obs_mode, int obs_n, int64_t num_envs, int num_snakes,
                    int size, const wurm_multi_config *cfg, uint64_t seed, uint64_t call, int64_t env_offset,
                    const wurm_multi_inject *inject, float *agent_major_f32, uint8_t *agent_major_u8, void *stream)
{
    int rc = multi_check_args(num_envs, num_snakes, size, obs_mode, obs_n, obs);
    if (rc) return rc;
    if (!cfg) return WURM_ERR_INVALID_ARG;
    if (num_envs > 0 && (!foods || !heads || !bodies || !dones || !orientations || !actions || !boost_this_step ||
                         !rewards || !snake_collision || !edge_collision || !food_consumed || !sizes || !all_done))
        return WURM_ERR_INVALID_ARG;
    if (obs_mode == WURM_OBS_PARTIAL && num_envs > 0 && !colours) return WURM_ERR_INVALID_ARG;
    MultiArgs p = {};
    p.foods = foods; p.heads = heads; p.bodies = bodies; p.dones = dones; p.orientations = (long long *)orientations;
    p.actions = (const long long *)actions; p.boost = boost_this_step; p.rewards = rewards; p.snakecol = snake_collision;
    p.edgecol = edge_collision; p.foodcons = food_consumed; p.sizes = sizes; p.all_done = all_done;
    p.colours = const_cast<short *>(colours); p.obs = obs; p.obs_mode = obs_mode; p.obs_n = obs_n;
    p.obs_elems = multi_obs_elems(obs_mode, obs_n, size); p.N = num_envs; p.K = num_snakes; p.S = size; p.cfg = *cfg;
    p.seed = seed; p.call = call; p.env_offset = env_offset;
    if (inject) { p.inj = *inject; p.has_inj = 1; }
    if (agent_major_f32 && agent_major_u8) { p.am_f32 = agent_major_f32; p.am_u8 = agent_major_u8; }
    return multi_launch(MK_STEP, p, stream);
}

int wurm_multi_step_reset(const wurm_multi_call *c, void *stream)
{
    if (!c) return WURM_ERR_INVALID_ARG;
    int rc = multi_check_args(c->num_envs, c->num_snakes, c->size, c->obs_mode, c->obs_n, c->obs);
    if (rc) return rc;
    if (c->num_envs > 0 && (!c->foods || !c->heads || !c->bodies || !c->dones || !c->orientations || !c->actions ||
                            !c->boost_this_step || !c->rewards || !c->snake_collision || !c->edge_collision ||
                            !c->food_consumed || !c->sizes || !c->all_done))
        return WURM_ERR_INVALID_ARG;
    if ((c->obs_mode == WURM_OBS_PARTIAL || c->pre_done) && c->num_envs > 0 && !c->colours) return WURM_ERR_INVALID_ARG;
    if ((c->pre_done || c->obs_after) && c->size < 5) return WURM_ERR_UNSUPPORTED;
    if (c->obs_after && c->pre_inject) return WURM_ERR_UNSUPPORTED; // the reset behind obs_after draws from the RNG
    MultiArgs p = {};
    p.foods = c->foods; p.heads = c->heads; p.bodies = c->bodies; p.dones = c->dones;
    p.orientations = (long long *)c->orientations; p.actions = (const long long *)c->actions; p.boost = c->boost_this_step;
    p.rewards = c->rewards; p.snakecol = c->snake_collision; p.edgecol = c->edge_collision; p.foodcons = c->food_consumed;
    p.sizes = c->sizes; p.all_done = c->all_done; p.all_done_copy = c->all_done_copy; p.colours = c->colours;
    p.obs = c->obs; p.obs_after = c->obs_after; p.obs_mode = c->obs_mode; p.obs_n = c->obs_n;
    p.obs_elems = multi_obs_elems(c->obs_mode, c->obs_n, c->size); p.N = c->num_envs; p.K = c->num_snakes; p.S = c->size;
    p.cfg = c->cfg; p.seed = c->seed; p.call = c->call; p.env_offset = c->env_offset;
    p.done_env = c->pre_done; p.pre_call = c->pre_call;
    if (c->inject) { p.inj = *c->inject; p.has_inj = 1; }
    if (c->pre_inject) { p.rinj = *c->pre_inject; p.has_rinj = 1; }
    if (c->agent_major_f32 && c->agent_major_u8) { p.am_f32 = c->agent_major_f32; p.am_u8 = c->agent_major_u8; }
    p.err = c->check_mask;
    p.err_after = c->check_mask_after;
    if (c->resident && c->num_envs > 0) {
        if (!c->inject && !c->pre_inject) {
            p.resident = (unsigned char *)c->resident;
            p.resident_valid = c->resident_valid != 0;
            p.resident_lazy = c->resident_lazy != 0;
        } else if (c->resident_lazy && c->resident_valid) { // recorded outcomes step the fp32 state: write the mirror out first
            rc = wurm_multi_resident_flush(c, stream);
            if (rc) return rc;
        }
    }
    return multi_launch(MK_STEP, p, stream);
}

int64_t wurm_multi_resident_size(int64_t num_envs, int num_snakes, int size)
{
    if (num_envs <= 0 || num_snakes < 1 || num_snakes > 64 || size < 5 || size > 64) return 0;
    return num_envs * mirror_env_bytes(num_snakes, size * size);
}

int64_t wurm_multi_resident_bytes(int64_t num_envs, int num_snakes, int size)
{
    const long long e = opt.resident_min_envs; // -1: by shape
    const bool big = e >= 0 ? num_envs >= e : num_envs * (long long)num_snakes * size * size >= (1ll << 20);
    return big ? wurm_multi_resident_size(num_envs, num_snakes, size) : 0;
}

int wurm_multi_resident_flush(const wurm_multi_call *c, void *stream)
{
    if (!c) return WURM_ERR_INVALID_ARG;
    if (!c->resident || !c->resident_lazy || !c->resident_valid || c->num_envs <= 0) return WURM_OK;
    if (!c->foods || !c->heads || !c->bodies) return WURM_ERR_INVALID_ARG;
    MultiArgs p = {};
    p.foods = c->foods; p.heads = c->heads; p.bodies = c->bodies; p.N = c->num_envs; p.K = c->num_snakes; p.S = c->size;
    p.resident = (unsigned char *)c->resident;
    (void)hipGetLastError();
    WURM_LAUNCH(multi_flush_kernel, dim3((unsigned)p.N), dim3(256), 0, (hipStream_t)stream, p);
    return hipGetLastError() == hipSuccess ? WURM_OK : WURM_ERR_HIP;
}

int wurm_multi_step_packed(wurm_multi_call *c, float *out_f32, uint8_t *out_u8, float *obs, float *obs_after,
                           const int64_t *actions, uint64_t call, int apply_pending, uint64_t pre_call, void *stream)
{
    if (!c) return WURM_ERR_INVALID_ARG;
    if (c->num_envs > 0 && (!out_f32 || !out_u8)) return WURM_ERR_INVALID_ARG;
    if (apply_pending && !c->all_done_copy) return WURM_ERR_INVALID_ARG;
    const long long KN = (long long)c->num_snakes * c->num_envs;
    c->rewards = out_f32; c->food_consumed = out_f32 + KN; c->sizes = out_f32 + 2 * KN;
    c->agent_major_f32 = out_f32 + 3 * KN;
    c->boost_this_step = out_u8; c->snake_collision = out_u8 + KN; c->edge_collision = out_u8 + 2 * KN;
    c->agent_major_u8 = out_u8 + 3 * KN;
    c->all_done = out_u8 + 7 * KN;
    c->obs = obs;
    c->obs_after = obs_after;
    c->actions = actions;
    c->call = call;
    c->pre_done = apply_pending ? c->all_done_copy : nullptr;
    c->pre_call = pre_call;
    const int rc = wurm_multi_step_reset(c, stream);
    if (c->resident) c->resident_valid = (rc == WURM_OK && !c->inject && !c->pre_inject) ? 1 : 0;
    return rc;
}

int wurm_multi_step_slot(wurm_multi_call *c, const wurm_multi_slabs *slabs, int64_t slot, const int64_t *actions,
                         uint64_t call, int apply_pending, uint64_t pre_call, int want_obs_after, void *stream)
{
    if (!c || !slabs || slot < 0 || slot >= slabs->steps) return WURM_ERR_INVALID_ARG;
    if (want_obs_after && !slabs->obs_after) return WURM_ERR_INVALID_ARG;
    const long long KN = (long long)c->num_snakes * c->num_envs, per_obs = KN * slabs->obs_elems;
    return wurm_multi_step_packed(c, slabs->out_f32 ? slabs->out_f32 + slot * 6 * KN : nullptr,
                                  slabs->out_u8 ? slabs->out_u8 + slot * (7 * KN + c->num_envs) : nullptr,
                                  slabs->obs ? slabs->obs + slot * per_obs : nullptr,
                                  want_obs_after ? slabs->obs_after + slot * per_obs : nullptr, actions, call, apply_pending,
                                  pre_call, stream);
}

int wurm_multi_reset(float *foods, float *heads, float *bodies, uint8_t *dones, int64_t *orientations,
                     int16_t *colours, const uint8_t *done_env, int32_t *status, const uint8_t *boost_this_step,
                     float *obs, int obs_mode, int obs_n, int64_t num_envs, int num_snakes, int size,
                     const wurm_multi_config *cfg, uint64_t seed, uint64_t call, int64_t env_offset,
                     const wurm_multi_reset_inject *inject, void *stream)
{
    int rc = multi_check_args(num_envs, num_snakes, size, obs_mode, obs_n, obs);
    if (rc) return rc;
    if (!cfg) return WURM_ERR_INVALID_ARG;
    if (size < 5) return WURM_ERR_UNSUPPORTED; // no cell is >= 2 from the border (multi_snake.py:938-941)
    if (num_envs > 0 && (!foods || !heads || !bodies || !dones || !orientations || !done_env || !colours))
        return WURM_ERR_INVALID_ARG;
    if (obs_mode != WURM_OBS_NONE && num_envs > 0 && !boost_this_step) return WURM_ERR_INVALID_ARG;
    MultiArgs p = {};
    p.foods = foods; p.heads = heads; p.bodies = bodies; p.dones = dones; p.orientations = (long long *)orientations;
    p.colours = colours; p.done_env = done_env; p.status = status; p.boost = const_cast<uint8_t *>(boost_this_step);
    p.obs = obs; p.obs_mode = obs_mode; p.obs_n = obs_n; p.obs_elems = multi_obs_elems(obs_mode, obs_n, size);
    p.N = num_envs; p.K = num_snakes; p.S = size; p.cfg = *cfg; p.seed = seed; p.call = call; p.env_offset = env_offset;
    if (inject) { p.rinj = *inject; p.has_rinj = 1; }
    return multi_launch(MK_RESET, p, stream);
}

int wurm_multi_rollout(float *foods, float *heads, float *bodies, uint8_t *dones, int64_t *orientations,
                       int16_t *colours, uint8_t *boost_this_step, const int64_t *actions, float *out_f32,
                       uint8_t *out_u8, uint8_t *all_done, float *obs, int obs_mode, int obs_n, int64_t num_envs,
                       int num_snakes, int size, int64_t num_steps, const wurm_multi_config *cfg, uint64_t seed,
                       uint64_t call0, int64_t env_offset, const wurm_multi_inject *inject,
                       const wurm_multi_reset_inject *reset_inject, void *stream)
{
    int rc = multi_check_args(num_envs, num_snakes, size, obs_mode, obs_n, obs);
    if (rc) return rc;
    if (!cfg || num_steps < 0) return WURM_ERR_INVALID_ARG;
    if (size < 5) return WURM_ERR_UNSUPPORTED;
    if (num_envs > 0 && (!foods || !heads || !bodies || !dones || !orientations || !colours)) return WURM_ERR_INVALID_ARG;
    if (num_envs > 0 && num_steps > 0 && (!actions || !out_f32 || !out_u8 || !all_done)) return WURM_ERR_INVALID_ARG;
    if (num_steps == 0) return WURM_OK;
    MultiArgs p = {};
    p.foods = foods; p.heads = heads; p.bodies = bodies; p.dones = dones; p.orientations = (long long *)orientations;
    p.colours = colours; p.boost_state = boost_this_step; p.actions = (const long long *)actions; p.am_f32 = out_f32;
    p.am_u8 = out_u8; p.all_done = all_done; p.obs = obs; p.obs_mode = obs_mode; p.obs_n = obs_n;
    p.obs_elems = multi_obs_elems(obs_mode, obs_n, size); p.N = num_envs; p.K = num_snakes; p.S = size; p.T = num_steps;
    p.cfg = *cfg; p.seed = seed; p.call = call0; p.env_offset = env_offset;
    if (inject) { p.inj = *inject; p.has_inj = 1; }
    if (reset_inject) { p.rinj = *reset_inject; p.has_rinj = 1; }
    if ((inject == nullptr) != (reset_inject == nullptr)) return WURM_ERR_INVALID_ARG; // replay needs both tapes
    return multi_launch(MK_ROLLOUT, p, stream);
}

int wurm_multi_rollout_resident(float *foods, float *heads, float *bodies, uint8_t *dones, int64_t *orientations,
                                int16_t *colours, uint8_t *boost_this_step, const int64_t *actions, float *out_f32,
                                uint8_t *out_u8, uint8_t *all_done, float *obs, int obs_mode, int obs_n, int64_t num_envs,
                                int num_snakes, int size, int64_t num_steps, const wurm_multi_config *cfg, uint64_t seed,
                                uint64_t call0, int64_t env_offset, void *resident, int *resident_valid, int resident_lazy,
                                void *stream)
{
    if (resident == nullptr)
        return wurm_multi_rollout(foods, heads, bodies, dones, orientations, colours, boost_this_step, actions, out_f32, out_u8,
                                  all_done, obs, obs_mode, obs_n, num_envs, num_snakes, size, num_steps, cfg, seed, call0,
                                  env_offset, nullptr, nullptr, stream);
    if (resident_valid == nullptr) return WURM_ERR_INVALID_ARG;
    int rc = multi_check_args(num_envs, num_snakes, size, obs_mode, obs_n, obs);
    if (rc) return rc;
    if (!cfg || num_steps < 0) return WURM_ERR_INVALID_ARG;
    if (size < 5) return WURM_ERR_UNSUPPORTED;
    if (num_envs > 0 && (!foods || !heads || !bodies || !dones || !orientations || !colours)) return WURM_ERR_INVALID_ARG;
    if (num_envs > 0 && num_steps > 0 && (!actions || !out_f32 || !out_u8 || !all_done)) return WURM_ERR_INVALID_ARG;
    if (num_steps == 0 || num_envs == 0) return WURM_OK;
    MultiArgs p = {};
    p.foods = foods; p.heads = heads; p.bodies = bodies; p.dones = dones; p.orientations = (long long *)orientations;
    p.colours = colours; p.boost_state = boost_this_step; p.actions = (const long long *)actions; p.am_f32 = out_f32;
    p.am_u8 = out_u8; p.all_done = all_done; p.obs = obs; p.obs_mode = obs_mode; p.obs_n = obs_n;
    p.obs_elems = multi_obs_elems(obs_mode, obs_n, size); p.N = num_envs; p.K = num_snakes; p.S = size; p.T = num_steps;
    p.cfg = *cfg; p.seed = seed; p.call = call0; p.env_offset = env_offset;
    MultiArgs q;
    size_t bytes = 0;
    // the kernels that keep a mirror: the grouped writer, and the one-wave-per-env rollout (everything but 'full'
    // observations of at most 10 snakes over several steps in a small batch, which goes to the two-wave form)
    const bool two = obs_mode == WURM_OBS_DEFAULT && num_snakes <= SNAP_MAX_SNAKES && num_steps > 1;
    if (multi_group_shape(p, q, bytes) != nullptr || !two) {
        p.resident = (unsigned char *)resident;
        p.resident_valid = *resident_valid != 0;
        p.resident_lazy = resident_lazy != 0;
        rc = multi_launch(MK_ROLLOUT, p, stream);
        if (rc == WURM_OK) *resident_valid = 1;
        return rc;
    }
    // any other rollout kernel works on the fp32 planes: a lazy mirror is written out to them first, and it is stale afterwards
    if (resident_lazy && *resident_valid) {
        MultiArgs f = {};
        f.foods = foods; f.heads = heads; f.bodies = bodies; f.N = num_envs; f.K = num_snakes; f.S = size;
        f.resident = (unsigned char *)resident;
        (void)hipGetLastError();
        WURM_LAUNCH(multi_flush_kernel, dim3((unsigned)f.N), dim3(256), 0, (hipStream_t)stream, f);
        if (hipGetLastError() != hipSuccess) return WURM_ERR_HIP;
    }
    *resident_valid = 0;
    return multi_launch(MK_ROLLOUT, p, stream);
}

int wurm_multi_observe(const float *foods, const float *heads, const float *bodies, const uint8_t *dones,
                       const uint8_t *boost_this_step, const int16_t *colours, float *obs, int obs_mode, int obs_n,
                       int64_t num_envs, int num_snakes, int size, void *stream)
{
    if (obs_mode == WURM_OBS_NONE) return WURM_ERR_INVALID_ARG;
    int rc = multi_check_args(num_envs, num_snakes, size, obs_mode, obs_n, obs);
    if (rc) return rc;
    if (num_envs > 0 && (!foods || !heads || !bodies || !dones || !boost_this_step)) return WURM_ERR_INVALID_ARG;
    if (obs_mode == WURM_OBS_PARTIAL && num_envs > 0 && !colours) return WURM_ERR_INVALID_ARG;
    MultiArgs p = {};
    p.foods = const_cast<float *>(foods); p.heads = const_cast<float *>(heads); p.bodies = const_cast<float *>(bodies);
    p.dones = const_cast<uint8_t *>(dones); p.boost = const_cast<uint8_t *>(boost_this_step);
    p.colours = const_cast<short *>(colours); p.obs = obs; p.obs_mode = obs_mode; p.obs_n = obs_n;
    p.obs_elems = multi_obs_elems(obs_mode, obs_n, size); p.N = num_envs; p.K = num_snakes; p.S = size;
    return multi_launch(MK_OBSERVE, p, stream);
}

int wurm_multi_check(const float *foods, const float *heads, const float *bodies, const uint8_t *dones, uint32_t *err,
                     int64_t num_envs, int num_snakes, int size, void *stream)
{
    int rc = multi_check_args(num_envs, num_snakes, size, WURM_OBS_NONE, 0, nullptr);
    if (rc) return rc;
    if (num_envs > 0 && (!foods || !heads || !bodies || !dones || !err)) return WURM_ERR_INVALID_ARG;
    MultiArgs p = {};
    p.foods = const_cast<float *>(foods); p.heads = const_cast<float *>(heads); p.bodies = const_cast<float *>(bodies);
    p.dones = const_cast<uint8_t *>(dones); p.err = err; p.obs_mode = WURM_OBS_NONE; p.N = num_envs; p.K = num_snakes;
    p.S = size;
    return multi_launch(MK_CHECK, p, stream);
}

int wurm_multi_colours(int16_t *colours, int64_t num_envs, int num_snakes, int fixed, uint64_t seed, uint64_t call,
                       int64_t env_offset, void *stream)
{
    if (num_envs < 0 || num_snakes < 1) return WURM_ERR_INVALID_ARG;
    if (num_envs == 0) return WURM_OK;
    if (!colours) return WURM_ERR_INVALID_ARG;
    long long n = (long long)num_envs * num_snakes;
    (void)hipGetLastError();
    WURM_LAUNCH(multi_colours_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       colours, (long long)num_envs, num_snakes, fixed, seed, call, (long long)env_offset);
    return hipGetLastError() == hipSuccess ? WURM_OK : WURM_ERR_HIP;
}

} // extern "C"
