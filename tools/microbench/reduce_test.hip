// Wave reductions of wurm_device.hpp (wave_max / min / sum_i32, hand-written DPP butterflies) against a host loop, and
// the hazard that made round 1 park them ("the min reduction faults inside the 511-register rollout_kernel<64>"):
//   full   : all 64 lanes active — every reduction must equal the host loop;
//   part   : the reduction is called under a lane-dependent branch (lanes >= nact idle).  A DPP row operation does not
//            write a lane whose source lane is disabled by EXEC: the in-place form keeps the lane's own value (min / max
//            stay correct over the ACTIVE lanes as long as lanes 0/16/32/48 are active), a form with a fresh destination
//            register leaves stale register contents there — shown by `fresh_dst_min`, which reads them back;
// build: hipcc --offload-arch=gfx950 -O3 -o reduce_test reduce_test.hip ; run: ./reduce_test
#include "../../wurm_amd/csrc/wurm_device.hpp"
#include <cstdio>
#include <cstdlib>
using namespace wurm;

__global__ void full(const int *in, int *out)
{
    int v = in[blockIdx.x * 64 + threadIdx.x];
    int a = wave_max_i32(v), b = wave_min_i32(v), c = wave_sum_i32(v);
    if (threadIdx.x == 0) { out[blockIdx.x * 3] = a; out[blockIdx.x * 3 + 1] = b; out[blockIdx.x * 3 + 2] = c; }
}

// the UNSAFE variant: destination register distinct from the sources, pre-loaded with a small "stale" value
__device__ __forceinline__ int fresh_dst_min_step(int v, int stale)
{
    int t = stale;
    asm volatile("s_nop 1\n\tv_min_i32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 0" : "+v"(t) : "v"(v));
    return t;
}

__global__ void part(const int *in, int *out, int nact)
{
    const int lane = threadIdx.x;
    int v = in[blockIdx.x * 64 + lane];
    int safe = 0x7fffffff, unsafe = 0x7fffffff;
    if (lane < nact) {                      // odd nact: lane nact-1's quad partner is disabled
        safe = wave_min_i32(v);             // in-place butterfly
        unsafe = fresh_dst_min_step(v, -123456789);
    }
    if (lane == 0) out[blockIdx.x * 2] = safe;
    if (lane == nact - 1) out[blockIdx.x * 2 + 1] = unsafe; // what the lane with the disabled partner holds
}

int main()
{
    const int B = 4096;
    int *h = (int *)malloc(B * 64 * 4), *ho = (int *)malloc(B * 12), *d, *dout;
    srand(1);
    for (int i = 0; i < B * 64; ++i) h[i] = (rand() % 2000001) - 1000000;
    hipMalloc(&d, B * 64 * 4); hipMalloc(&dout, B * 12);
    hipMemcpy(d, h, B * 64 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(full, dim3(B), dim3(64), 0, 0, d, dout);
    hipMemcpy(ho, dout, B * 12, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int b = 0; b < B; ++b) {
        int mx = h[b * 64], mn = h[b * 64], s = 0;
        for (int i = 0; i < 64; ++i) { int x = h[b * 64 + i]; mx = x > mx ? x : mx; mn = x < mn ? x : mn; s += x; }
        if (ho[b * 3] != mx || ho[b * 3 + 1] != mn || ho[b * 3 + 2] != s) ++bad;
    }
    printf("full waves   : %d of %d blocks wrong\n", bad, B);
    int total_bad = bad;
    for (int nact : {49, 33, 17}) {         // lanes 0, 16, 32(, 48) active as the readlanes need
        hipLaunchKernelGGL(part, dim3(B), dim3(64), 0, 0, d, dout, nact);
        hipMemcpy(ho, dout, B * 8, hipMemcpyDeviceToHost);
        int bad_safe = 0, stale = 0;
        for (int b = 0; b < B; ++b) {
            int mn = h[b * 64];
            for (int i = 0; i < nact; ++i) mn = h[b * 64 + i] < mn ? h[b * 64 + i] : mn;
            if (ho[b * 2] != mn) ++bad_safe;
            if (ho[b * 2 + 1] == -123456789) ++stale;
        }
        printf("%2d active   : in-place min wrong in %d blocks; fresh-destination form kept the stale register in %d of %d\n",
               nact, bad_safe, stale, B);
        total_bad += bad_safe;
    }
    return total_bad != 0;
}
