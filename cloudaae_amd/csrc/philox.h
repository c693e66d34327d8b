// philox.h -- stateless counter RNG shared by synth.hip (occluder sampling, padding choice) and step.hip (the
// tf.random.normal of train_cloudAAE_ycbv.py:217): a value is a pure function of (seed, counter, stream id).
#pragma once
#include <hip/hip_runtime.h>

namespace cloudaae {

// ---- Philox4x32-10 counter RNG + Box-Muller (occluder sampling, padding choice) -----------
__device__ __forceinline__ void philox_round(unsigned (&c)[4], unsigned k0, unsigned k1)
{
    const unsigned long long p0 = 0xD2511F53ull * c[0], p1 = 0xCD9E8D57ull * c[2];
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0, n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1;
    c[0] = n0;
    c[1] = (unsigned)p1;
    c[2] = n2;
    c[3] = (unsigned)p0;
}
__device__ __forceinline__ void philox4x32(unsigned long long seed, unsigned long long ctr, unsigned stream,
                                           unsigned (&out)[4])
{
    unsigned c[4] = {(unsigned)ctr, (unsigned)(ctr >> 32), stream, 0x9E3779B9u};
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c[0];
    out[1] = c[1];
    out[2] = c[2];
    out[3] = c[3];
}
__device__ __forceinline__ float u01(unsigned x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }
__device__ __forceinline__ void normal2(unsigned a, unsigned b, float &n0, float &n1)
{
    const float r = sqrtf(-2.0f * logf(u01(a))), t = 6.283185307179586f * u01(b);
    n0 = r * cosf(t);
    n1 = r * sinf(t);
}

} // namespace cloudaae
