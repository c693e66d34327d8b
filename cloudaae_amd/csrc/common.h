// common.h -- shared helpers for the gfx950 kernels of libcloudaae_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CLOUDAAE_API extern "C" __attribute__((visibility("default")))

namespace cloudaae {

// thread-local description of the last failure (read by cloudaae_last_error)
void set_error(const char *fmt, ...);

inline int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

// Development knobs (kernel A/B choices, launch shapes): an integer per name, read from the environment variable
// of that name ONCE, when the first call site asks for it; cloudaae_set_knob / cloudaae_unset_knob change it
// afterwards (tests, sweeps).  None is needed in normal use.  A call site keeps its slot in a static pointer.
struct Knob {
    const char *name;
    int value;
    bool set;
};
Knob *knob_slot(const char *name);
#define CLOUDAAE_KNOB(NAME, FALLBACK)                                 \
    ([&]() -> int {                                                   \
        static cloudaae::Knob *k__ = cloudaae::knob_slot(NAME);       \
        return k__->set ? k__->value : (FALLBACK);                    \
    }())
#define CLOUDAAE_KNOB_SET(NAME)                                       \
    ([&]() -> bool {                                                  \
        static cloudaae::Knob *k__ = cloudaae::knob_slot(NAME);       \
        return k__->set;                                              \
    }())

// Every C-ABI entry point returns 0 or a hipError_t value; kernels are launched
// on the caller's stream and never synchronise.
#define CLOUDAAE_CHECK_LAUNCH(name)                                              \
    do {                                                                         \
        hipError_t e__ = hipGetLastError();                                      \
        if (e__ != hipSuccess) {                                                 \
            cloudaae::set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return (int)e__;                                                     \
        }                                                                        \
    } while (0)

#define CLOUDAAE_CHECK_HIP(expr, name)                                           \
    do {                                                                         \
        hipError_t e__ = (expr);                                                 \
        if (e__ != hipSuccess) {                                                 \
            cloudaae::set_error("%s: %s", name, hipGetErrorString(e__));         \
            return (int)e__;                                                     \
        }                                                                        \
    } while (0)

#define CLOUDAAE_REQUIRE(cond, name, msg)                                        \
    do {                                                                         \
        if (!(cond)) {                                                           \
            cloudaae::set_error("%s: %s", name, msg);                            \
            return (int)hipErrorInvalidValue;                                    \
        }                                                                        \
    } while (0)

// Scratch of ONE call, stream ordered (hipMallocAsync / hipFreeAsync on the caller's stream: no state outlives the
// call, nothing synchronises).  The device's default pool is told once to keep what it is given back: with the
// default release threshold of 0 every stream synchronisation returns the memory to the driver and the next call
// pays a real allocation.
inline hipError_t scratch_alloc(void **p, size_t bytes, hipStream_t s)
{
    static bool tuned[64] = {};
    int dev = -1;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64 && !tuned[dev]) {
        hipMemPool_t pool;
        if (hipDeviceGetDefaultMemPool(&pool, dev) == hipSuccess) {
            uint64_t keep = UINT64_MAX;
            (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
        }
        tuned[dev] = true;
    }
    return hipMallocAsync(p, bytes, s);
}

typedef float float2v __attribute__((ext_vector_type(2)));
typedef float float4v __attribute__((ext_vector_type(4)));

// ---- XCD-aware workgroup mapping ---------------------------------------------
// MI355X: 8 XCDs, each with a private 4 MiB L2; workgroup b is observed to run on XCD
// b % 8 (MI355X_MICROARCH.md).  These remaps only change WHICH workgroup does which
// piece of work, so a different placement would change speed, never results.

// bijective: linear id -> virtual id such that each XCD owns a contiguous virtual range
__device__ __forceinline__ int xcd_contiguous(int lin, int total)
{
    const int q = total >> 3, r = total & 7, xcd = lin & 7, u = lin >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + u;
}

// grid (tiles, clouds): give every cloud's tiles to ONE XCD (cloud c -> XCD c % 8)
__device__ __forceinline__ void xcd_cloud_tile(int &tile, int &cloud)
{
    const int gx = gridDim.x, B = gridDim.y;
    if ((B & 7) == 0) {
        const int lin = blockIdx.y * gx + blockIdx.x;
        const int xcd = lin & 7, u = lin >> 3;
        cloud = xcd + 8 * (u / gx);
        tile = u % gx;
    } else {
        tile = blockIdx.x;
        cloud = blockIdx.y;
    }
}

// ---- wave64 helpers ------------------------------------------------------
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        v += __shfl_xor(v, off, 64);
    return v;
}

// max of a 64-bit key over the wave, the same value in every lane.  Inside a row of 16 lanes the partner comes through
// DPP (quad swaps, then the half-row and row mirrors: after the quad steps a quad is uniform, so the mirrors act as
// "xor 4" and "xor 8"); the four row maxima are read as scalars (v_readlane) and meet on the scalar unit -- no trip
// through the LDS crossbar (six ds_bpermute pairs in the shuffle version: most of a farthest-point-sampling round).
template <int CTRL>
__device__ __forceinline__ unsigned long long dpp_u64(unsigned long long v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)v, CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(v >> 32), CTRL, 0xf, 0xf, false);
    return ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
}
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v)
{
    unsigned long long o;
    o = dpp_u64<0xB1>(v);       // quad_perm [1,0,3,2]
    v = o > v ? o : v;
    o = dpp_u64<0x4E>(v);       // quad_perm [2,3,0,1]
    v = o > v ? o : v;
    o = dpp_u64<0x141>(v);      // row_half_mirror
    v = o > v ? o : v;
    o = dpp_u64<0x140>(v);      // row_mirror
    v = o > v ? o : v;
    unsigned long long best = 0;
#pragma unroll
    for (int row = 0; row < 4; ++row) {
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, 16 * row);
        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), 16 * row);
        const unsigned long long r = ((unsigned long long)hi << 32) | lo;
        best = r > best ? r : best;
    }
    return best;
}

} // namespace cloudaae
