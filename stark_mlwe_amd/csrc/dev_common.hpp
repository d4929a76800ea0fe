// stark_mlwe_amd/csrc/dev_common.hpp — small helpers shared by the kernels (32-byte element moves).
#pragma once
#include "fr.hpp"

namespace stark {

FR_HD fr_t ldg(const fr_t* p) {   // 32-byte global load as 2 x dwordx4 on the device
#if defined(__HIP_DEVICE_COMPILE__)
    const uint4* q = reinterpret_cast<const uint4*>(p); uint4 lo = q[0], hi = q[1];
    fr_t x; x.v[0] = lo.x; x.v[1] = lo.y; x.v[2] = lo.z; x.v[3] = lo.w; x.v[4] = hi.x; x.v[5] = hi.y; x.v[6] = hi.z; x.v[7] = hi.w; return x;
#else
    return *p;
#endif
}
FR_HD void stg(fr_t* p, const fr_t& x) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(x.v[0], x.v[1], x.v[2], x.v[3]); q[1] = make_uint4(x.v[4], x.v[5], x.v[6], x.v[7]);
#else
    *p = x;
#endif
}
#if defined(__HIPCC__)
__device__ __forceinline__ fr_t shfl_xor_fr(const fr_t& x, int mask) {
    fr_t r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = (uint32_t)__shfl_xor((int)x.v[i], mask, 64);
    return r;
}
__device__ __forceinline__ fr_t shfl_up_fr(const fr_t& x, int d) {
    fr_t r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = (uint32_t)__shfl_up((int)x.v[i], d, 64);
    return r;
}
__device__ __forceinline__ fr_t shfl_dn_fr(const fr_t& x, int d) {
    fr_t r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = (uint32_t)__shfl_down((int)x.v[i], d, 64);
    return r;
}
__device__ __forceinline__ fr_t shfl_idx_fr(const fr_t& x, int src) {
    fr_t r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = (uint32_t)__shfl((int)x.v[i], src, 64);
    return r;
}
#endif

}  // namespace stark
