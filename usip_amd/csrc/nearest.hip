// usip_amd/csrc/nearest.hip -- fused pairwise-distance minimum (value + arg) on gfx950.
//
// This is the O(M*N) core of both chamfer losses of the reference:
//   SingleSideChamferLoss_Brute (models/losses.py:132-143): min_n |kp[b,:,m] - pc[b,:,n]|
//   ChamferLoss_Brute           (models/losses.py:62-66,81-87): row minima + arg minima of the
//                               keypoint x keypoint matrix, in both directions.
// The reference materialises B x 3 x M x N (805 MB for keypoint-on-pc at B=8) and B x M x N
// tensors; here nothing but the two small point sets is read and B x M (value, index) pairs
// are written.  The distance is the path-wide one (FMA chain + correctly rounded sqrt) so the
// minimum VALUE is bit-identical to torch's.  The arg-minimum is the FIRST index attaining the
// minimum of the sqrt'ed distances, as torch.min(dim) returns.  sqrt is monotone, so only a
// candidate with a strictly smaller squared distance can win, and the (expensive, correctly
// rounded) sqrt is evaluated just for those: O(log N) times per lane instead of N.
#include "common.h"

namespace {

constexpr int R = 4;                 // query points per wave

__global__ __launch_bounds__(256) void nearest_kernel(
    const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ min_d,
    int32_t* __restrict__ arg, int Ma, int Nb)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int bi = blockIdx.y;
    const int i0 = (blockIdx.x * 4 + wave) * R;
    if (i0 >= Ma) return;
    const float* ab = a + (long long)bi * 3 * Ma;
    const float* bb = b + (long long)bi * 3 * Nb;
    float ax[R], ay[R], az[R], best_s[R], best_d[R];
    int best_j[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = min(i0 + r, Ma - 1);
        ax[r] = ab[i]; ay[r] = ab[Ma + i]; az[r] = ab[2 * Ma + i];
        best_s[r] = __builtin_inff(); best_d[r] = __builtin_inff(); best_j[r] = 0x7fffffff;
    }
    for (int j = lane; j < Nb; j += 64) {
        const float bx = bb[j], by = bb[Nb + j], bz = bb[2 * Nb + j];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float s = usip_sqdist(ax[r], ay[r], az[r], bx, by, bz);
            if (s < best_s[r]) {
                best_s[r] = s;
                const float d = sqrtf(s);
                if (d < best_d[r]) { best_d[r] = d; best_j[r] = j; }
            }
        }
    }
    // wave reduction: smaller distance wins, then lower index (== first occurrence overall)
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float d = best_d[r];
        int j = best_j[r];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float od = __shfl_down(d, off);
            const int oj = __shfl_down(j, off);
            if (od < d || (od == d && oj < j)) { d = od; j = oj; }
        }
        if (lane == 0 && i0 + r < Ma) {
            if (j == 0x7fffffff) j = 0;                          // all-NaN / empty row
            min_d[(long long)bi * Ma + i0 + r] = d;
            arg[(long long)bi * Ma + i0 + r] = j;
        }
    }
}

}  // namespace

extern "C" int usip_nearest_f32(const float* a, const float* b, float* min_d, int32_t* arg,
                                int B, int Ma, int Nb, void* stream)
{
    if (B < 0 || Ma < 0 || Nb < 1) return USIP_EINVAL;
    if ((long long)B * Ma == 0) return USIP_OK;
    if (!a || !b || !min_d || !arg || B > 65535) return USIP_EINVAL;
    dim3 grid(usip_ceil_div(Ma, 4 * R), B), block(256);
    USIP_LAUNCH(nearest_kernel, grid, block, 0, (hipStream_t)stream, a, b, min_d, arg, Ma, Nb);
    USIP_LAUNCH_CHECK();
    return USIP_OK;
}
