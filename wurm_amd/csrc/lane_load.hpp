// lane_load.hpp — the cooperative state read shared by the one-env-per-LANE kernels (lane_step.hpp: one call; lane_rollout.hpp:
// a fused rollout): a wave's block of EPW consecutive SingleSnake envs (EPW * 3 * S * S floats, one contiguous, 16-byte
// aligned run) is read as float4 and summarised per env in LDS — head cell, food cell, the cell of every body value, the
// bit set of the body values present, a packed counter (cells | heads << 8 | foods << 16 | out-of-range values << 24).
// About 2 % of the elements are non-zero: each wave-wide slot of 64 float4 compacts the float4s that hold one (with their
// index) into an LDS queue with one ballot + prefix count, and the queue is decoded 64 entries at a time — the
// division-heavy decode runs once per 64 non-zero float4s instead of once per slot (round 2 read three predicated dwords per
// (env, cell) pair: about half of lane_step_kernel's 109 VALU instructions per env).
#pragma once

namespace wurm {

constexpr int LANE_QCAP = 256;                 // entries of the non-zero queue (drained when fewer than 64 are free)
constexpr int LANE_QUEUE_BYTES = LANE_QCAP * 20; // u32 index + float4 per entry

// vm: u32 [VW][EPW] bit set of body values (values 1 .. 32 * VW - 1 are in range), stat: u32 [EPW], hpos / fpos: u8 [EPW],
// valpos: u8 [EPW][VS] cell of each body value, queue: 20 * QCAP bytes of scratch (16-byte aligned).  All of the block's
// EPW envs are present.
template <int EPW, int C, int VW, int VS, int QCAP = LANE_QCAP>
__device__ __forceinline__ void lane_load_block(const float *__restrict__ block, int lane, u32 *vm, u32 *stat,
                                                unsigned char *hpos, unsigned char *fpos, unsigned char *valpos,
                                                unsigned char *queue)
{
    constexpr int C3 = 3 * C, N4 = EPW * C3 / 4, B4 = 16; // B4 loads in flight per lane
    static_assert((EPW * C3) % 4 == 0, "the block is a whole number of float4");
    const float4 *base4 = (const float4 *)block;
    float4 *qv = (float4 *)queue;                    // [QCAP] the float4 that holds a non-zero element
    u32 *qg = (u32 *)(queue + 16 * QCAP);            // [QCAP] its index in the block
    int qn = 0; // wave-uniform
    auto drain = [&]() {
        wave_lds_sync();
        for (int i = lane; i < qn; i += 64) {
            const float4 v4 = qv[i];
            const int g = (int)qg[i];
            // the non-zero components (usually one), lowest first
            u32 m = (v4.x != 0.0f ? 1u : 0u) | (v4.y != 0.0f ? 2u : 0u) | (v4.z != 0.0f ? 4u : 0u) | (v4.w != 0.0f ? 8u : 0u);
            for (; m != 0; m &= m - 1) {
                const int k = __ffs((int)m) - 1;
                const float val = k == 0 ? v4.x : k == 1 ? v4.y : k == 2 ? v4.z : v4.w;
                const int f = 4 * g + k, ej = f / C3, r = f - ej * C3, ch = r / C, cj = r - ch * C;
                if (ch == 0) {
                    if (val > 0.5f) { fpos[ej] = (unsigned char)cj; atomicAdd(&stat[ej], 1u << 16); }
                } else if (ch == 1) {
                    if (val > 0.5f) { hpos[ej] = (unsigned char)cj; atomicAdd(&stat[ej], 1u << 8); }
                } else {
                    const int bi = __float2int_rn(val); // body (single_snake.py:210): position of every value, values present
                    if (bi > 0 && bi < 32 * VW) {
                        valpos[ej * VS + bi] = (unsigned char)cj;
                        atomicOr(&vm[(bi >> 5) * EPW + ej], 1u << (bi & 31));
                        atomicAdd(&stat[ej], 1u);
                    } else if (bi != 0) {
                        atomicAdd(&stat[ej], 1u << 24);
                    }
                }
            }
        }
        wave_lds_sync();
        qn = 0;
    };
    for (int i0 = 0; i0 < N4; i0 += 64 * B4) {
        float4 v[B4];
#pragma unroll
        for (int j = 0; j < B4; ++j) v[j] = base4[min(i0 + 64 * j + lane, N4 - 1)];
#pragma unroll
        for (int j = 0; j < B4; ++j) {
            const int g = i0 + 64 * j + lane;
            if (i0 + 64 * j >= N4) break;
            // one ballot per float4 slot (256 floats): the lanes whose float4 holds anything but +-0 queue it whole
            const u32 any = (__float_as_uint(v[j].x) | __float_as_uint(v[j].y) | __float_as_uint(v[j].z) |
                             __float_as_uint(v[j].w)) << 1;
            const bool nz = g < N4 && any != 0;
            const u64 m = ballot(nz);
            if (m != 0) {
                if (qn > QCAP - 64) drain();
                if (nz) {
                    const int slot = qn + rank_below(m);
                    qv[slot] = v[j];
                    qg[slot] = (u32)g;
                }
                qn += popc64(m);
            }
        }
    }
    drain();
}

} // namespace wurm
