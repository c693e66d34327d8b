// synth.hip -- on-line data synthesis on the GPU (gfx950): rigid transform of the object
// model, random spherical occluders, spherical flip and hidden-point removal (HPR).
//
// Replaces the tf.data pipeline of the reference (train_cloudAAE_ycbv.py:79-117):
//   get_rotation_matrix / transform_object_model        train_cloudAAE_ycbv.py:79-93
//   get_random_spherical_occluder                       utils/generate_occluder.py:38-81
//   sphericalFlip[_org]                                 utils/hidden_point_removal.py:6-24, 51-68
//   hidden_point_removal[_org] = convexHull             utils/hidden_point_removal.py:27-48
// The reference runs HPR as scipy/qhull under the Python GIL, twice per sample (~5 ms per
// 2449-point hull on a CPU core), which caps it at ~100 samples/s/core -- two orders of
// magnitude below what the training step consumes here.
//
// HPR needs only the VERTEX SET of conv(flipped points + origin), not its facets.  A
// point p is a vertex iff some direction d strictly separates it:  d.(q - p) < 0 for all
// other q.  Normalising d by its largest component gives six charts d = (x, y, +-1) etc.
// with (x, y) in [-1,1]^2, so "is p a vertex" is at most six 2-variable LP feasibility
// problems with n constraints.  Each lane solves its own with Seidel's incremental
// algorithm (expected O(n): a violated constraint triggers a 1-D re-solve over the
// constraints seen so far, which happens ~2 ln n times), in fp64 on the fp32 inputs, with
// the cloud in LDS.  The answer is algorithm-independent (it is the hull's vertex set), so
// it equals qhull's except for points within ~1e-12 relative of a facet (tests compare with
// scipy.spatial.ConvexHull on real object models: identical sets).
#include "common.h"
#include "philox.h"
#include "../../include/cloudaae_hip.h"

namespace cloudaae {

// ---- transform_object_model: out[b,j,:] = model[class[b],j,0:3] R_b^T + t_b -----------------
// R = float32(exponential_map(float64 axis-angle)) (train...:80-83); the 3-term dot product is
// evaluated left to right, un-fused (Eigen's order is not pinned by the reference).
__global__ void transform_model_kernel(int b, int npts, int nmodels, const float *__restrict__ models,
                                       const long long *__restrict__ class_id, const double *__restrict__ rot,
                                       const float *__restrict__ trans, float *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= b * npts)
        return;
    const int cloud = i / npts, j = i - cloud * npts;
    long long cls = class_id[cloud];
    cls = cls < 0 ? 0 : (cls >= nmodels ? nmodels - 1 : cls);
    const float *p = models + ((size_t)cls * npts + j) * 6;
    const double *R = rot + (size_t)cloud * 9;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const float r0 = (float)R[3 * r], r1 = (float)R[3 * r + 1], r2 = (float)R[3 * r + 2];
        const float a = p[0] * r0, bb = p[1] * r1, c = p[2] * r2;
        out[(size_t)i * 3 + r] = ((a + bb) + c) + trans[cloud * 3 + r];
    }
}

// ---- get_random_spherical_occluder (generate_occluder.py:38-81) ---------------------------
// two Gaussian blobs of `per` points, sigma 0.01, centres ~ N(0, Wnear/10), N(0, Hnear/10),
// N((near+z)/2, (z-near)/6); rows interleave the two blobs exactly as the reference's
// concat([x1,y1,z1,x2,y2,z2], axis=1).reshape(-1,3) does.
__global__ void occluder_kernel(int b, int per, const float *__restrict__ trans, float wnear, float hnear,
                                float near_d, float sigma, unsigned long long seed, float *__restrict__ occ)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;      // one thread per (cloud, pair of points)
    if (i >= b * per)
        return;
    const int cloud = i / per, j = i - cloud * per;
    const float z = trans[cloud * 3 + 2];
    unsigned r[4];
    float c[2][3], n0, n1;
    philox4x32(seed, (unsigned long long)cloud, 1u, r);         // blob centres: shared by the cloud
    normal2(r[0], r[1], n0, n1);
    c[0][0] = n0 * (wnear / 10.0f);
    c[1][0] = n1 * (wnear / 10.0f);
    normal2(r[2], r[3], n0, n1);
    c[0][1] = n0 * (hnear / 10.0f);
    c[1][1] = n1 * (hnear / 10.0f);
    philox4x32(seed, (unsigned long long)cloud, 2u, r);
    normal2(r[0], r[1], n0, n1);
    c[0][2] = (near_d + z) / 2.0f + n0 * ((z - near_d) / 6.0f);
    c[1][2] = (near_d + z) / 2.0f + n1 * ((z - near_d) / 6.0f);
    float g[6];
    philox4x32(seed, ((unsigned long long)cloud << 32) | (unsigned)j, 3u, r);
    normal2(r[0], r[1], g[0], g[1]);
    normal2(r[2], r[3], g[2], g[3]);
    philox4x32(seed, ((unsigned long long)cloud << 32) | (unsigned)j, 4u, r);
    normal2(r[0], r[1], g[4], g[5]);
    float *o = occ + ((size_t)cloud * 2 * per + 2 * j) * 3;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        o[a] = c[0][a] + sigma * g[a];
        o[3 + a] = c[1][a] + sigma * g[3 + a];
    }
}

// ---- sphericalFlip (hidden_point_removal.py:6-24): one workgroup per cloud -------------------
// points = concat(a, b) - center; f = 2 (R - |p|) p / |p| + p with R = max|p| * 10^param;
// a zero row (the viewpoint) is appended to both outputs.
__global__ __launch_bounds__(256) void spherical_flip_kernel(int na, int nb, const float *__restrict__ a,
                                                            const float *__restrict__ bpts,
                                                            const float *__restrict__ center, float pow10param,
                                                            float *__restrict__ flipped, float *__restrict__ org)
{
    __shared__ float red[4];
    const int cloud = blockIdx.x, t = threadIdx.x, n = na + nb;
    const float cx = center ? center[cloud * 3] : 0.f, cy = center ? center[cloud * 3 + 1] : 0.f,
                cz = center ? center[cloud * 3 + 2] : 0.f;
    float mx = 0.0f;
    for (int j = t; j < n; j += 256) {
        const float *p = j < na ? a + ((size_t)cloud * na + j) * 3 : bpts + ((size_t)cloud * nb + (j - na)) * 3;
        const float x = p[0] - cx, y = p[1] - cy, z = p[2] - cz;
        mx = fmaxf(mx, sqrtf((x * x + y * y) + z * z));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    if ((t & 63) == 0)
        red[t >> 6] = mx;
    __syncthreads();
    const float R = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) * pow10param;
    float *F = flipped + (size_t)cloud * (n + 1) * 3, *O = org + (size_t)cloud * (n + 1) * 3;
    for (int j = t; j <= n; j += 256) {
        if (j == n) {
            F[3 * j] = F[3 * j + 1] = F[3 * j + 2] = 0.0f;
            O[3 * j] = O[3 * j + 1] = O[3 * j + 2] = 0.0f;
            continue;
        }
        const float *p = j < na ? a + ((size_t)cloud * na + j) * 3 : bpts + ((size_t)cloud * nb + (j - na)) * 3;
        const float x = p[0] - cx, y = p[1] - cy, z = p[2] - cz;
        const float nrm = sqrtf((x * x + y * y) + z * z);
        const float s = 2.0f * (R - nrm);
        F[3 * j] = (s * x) / nrm + x;
        F[3 * j + 1] = (s * y) / nrm + y;
        F[3 * j + 2] = (s * z) / nrm + z;
        O[3 * j] = x;
        O[3 * j + 1] = y;
        O[3 * j + 2] = z;
    }
}

// ---- spatial order of a flipped cloud -------------------------------------------------------
// Seidel's LP is exact for ANY constraint order; its cost is not.  In a random order the violated
// constraints turn up at random positions i and each costs a 1-D re-solve over i earlier constraints.
// But the planes that decide whether p is a hull vertex are those of p's neighbours on the shell.  So every
// cloud is sorted once, a point's LP sees its neighbours in sorted order FIRST (j+1, j-1, j+2, ...) and then
// everything else: after the short local pass the optimum is almost always final (or the LP already
// infeasible).  (Round 4 sorted by direction from the centroid -- cube-map face, then a 16 x 16 grid in Morton
// order; retired in round 6, profiles/notes_hull_vertex.md.)
constexpr int HS_THREADS = 1024;

// sorted[h][pos] = points[h][perm[h][pos]]; one workgroup per cloud, a counting sort on a 32 x 32 x 32 grid over the cloud's
// bounding box, cells in Morton order (round 5): consecutive sorted points are neighbours in SPACE, whatever the shape of
// the cloud.  (A flipped cloud is a thin, almost flat patch of a huge sphere -- radius 10^2.5 x its distance -- with its
// centroid inside: sorted by DIRECTION from the centroid, nearly all of it falls into the two rows of cells next to the
// patch's plane and a cell is a long radial sliver -- fine as "some neighbours first", useless as a bounding volume
// (hpr_lp2d_wave_culled below).)
constexpr int HG_BITS = 5, HG_CELLS = 1 << (3 * HG_BITS);

__device__ __forceinline__ unsigned hg_spread(unsigned v)          // 5 bits -> every third bit
{
    v &= 0x1fu;
    v = (v | (v << 8)) & 0x100fu;
    v = (v | (v << 4)) & 0x10c3u;
    v = (v | (v << 2)) & 0x1249u;
    return v;
}

__global__ __launch_bounds__(HS_THREADS) void hpr_sort_grid_kernel(int n1, const float *__restrict__ points,
                                                                   float *__restrict__ sorted, int *__restrict__ perm,
                                                                   int *__restrict__ next_point)
{
    if (threadIdx.x == 0)
        next_point[blockIdx.x] = 0;         // (the hull kernel's queue of this cloud)
    extern __shared__ int cell_cnt[];         // HG_CELLS counters
    __shared__ float box[6][HS_THREADS / 64];
    __shared__ int wsum[HS_THREADS / 64];
    const float *P = points + (size_t)blockIdx.x * n1 * 3;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
    float hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    for (int q = t; q < n1; q += HS_THREADS)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            lo[c] = fminf(lo[c], P[3 * q + c]);
            hi[c] = fmaxf(hi[c], P[3 * q + c]);
        }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            lo[c] = fminf(lo[c], __shfl_xor(lo[c], off, 64));
            hi[c] = fmaxf(hi[c], __shfl_xor(hi[c], off, 64));
        }
    if (lane == 0)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            box[c][wave] = lo[c];
            box[3 + c][wave] = hi[c];
        }
    for (int c = t; c < HG_CELLS; c += HS_THREADS)
        cell_cnt[c] = 0;
    __syncthreads();
    float scale[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        for (int w = 0; w < HS_THREADS / 64; ++w) {
            lo[c] = fminf(lo[c], box[c][w]);
            hi[c] = fmaxf(hi[c], box[3 + c][w]);
        }
        const float ext = hi[c] - lo[c];
        scale[c] = ext > 0.0f ? (float)(1 << HG_BITS) / ext : 0.0f;
    }
    auto cell_of = [&](int q) {
        unsigned code = 0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int i = min((1 << HG_BITS) - 1, max(0, (int)((P[3 * q + c] - lo[c]) * scale[c])));
            code |= hg_spread((unsigned)i) << c;
        }
        return (int)code;
    };
    for (int q = t; q < n1; q += HS_THREADS)
        atomicAdd(&cell_cnt[cell_of(q)], 1);
    __syncthreads();
    // exclusive prefix over the cells: HG_CELLS / HS_THREADS consecutive cells per thread
    constexpr int PER = HG_CELLS / HS_THREADS;
    int local = 0;
    for (int c = 0; c < PER; ++c)
        local += cell_cnt[t * PER + c];
    int incl = local;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, d, 64);
        if (lane >= d)
            incl += o;
    }
    if (lane == 63)
        wsum[wave] = incl;
    __syncthreads();
    int run = incl - local;
    for (int w = 0; w < wave; ++w)
        run += wsum[w];
    for (int c = 0; c < PER; ++c) {
        const int n = cell_cnt[t * PER + c];
        cell_cnt[t * PER + c] = run;
        run += n;
    }
    __syncthreads();
    float *S = sorted + (size_t)blockIdx.x * n1 * 3;
    int *PM = perm + (size_t)blockIdx.x * n1;
    for (int q = t; q < n1; q += HS_THREADS) {
        const int pos = atomicAdd(&cell_cnt[cell_of(q)], 1);
        PM[pos] = q;
        S[3 * pos] = P[3 * q];
        S[3 * pos + 1] = P[3 * q + 1];
        S[3 * pos + 2] = P[3 * q + 2];
    }
}

// constraint sequence of point `self` (sorted position): HPR_NEAR neighbours in sorted order, alternating
// sides, then all OTHER positions in the order q = (p * stride) mod n1, stride ~ n1 / golden ratio and coprime
// to n1: a permutation of [0, n1) whose every prefix is spread evenly over the (spatially sorted) cloud -- what
// Seidel's expected O(n) needs.  (Round 2 walked the bit-reversal sequence over [0, 2^bits) and skipped what fell
// outside [0, n1): for the reference's 2449 / 2049-point clouds 40-50 % of the positions of every scan.)
// (neighbourhood size by cloud size, measured per batch of 32 with two hulls each: 2449-point clouds 96 / 128 / 192 / 256 / 512
// neighbours 1.69 / 1.56 / 1.52 / 1.51 / 1.74 ms; 8593-point clouds 128 / 192 / 256 / 384 / 512 / 768 / 1536 neighbours
// 13.0 / 11.75 / 11.0 / 10.47 / 10.54 / 10.49 / 11.1 ms)
#ifndef HPR_NEAR_SMALL_V
#define HPR_NEAR_SMALL_V 192
#endif
#ifndef HPR_NEAR_LARGE_V
#define HPR_NEAR_LARGE_V 512
#endif
constexpr int HPR_NEAR_SMALL = HPR_NEAR_SMALL_V, HPR_NEAR_LARGE = HPR_NEAR_LARGE_V;      // clouds one workgroup of 8 / 16 waves holds (hull_vertex_kernel)
// (`near`: the neighbours actually offered -- HPR_NEAR, or fewer in a cloud so small that the offsets +-1, +-2, ... would come round
//  to a point a second time: a binding constraint re-tested in floating point reports round-off as a violation)
__device__ __forceinline__ int hpr_near_of(int n1, int cap) { return min(cap, 2 * ((n1 - 1) / 2)); }
__device__ __forceinline__ int hpr_seq(int pos, int self, int n1, int stride, int near)
{
    if (pos < near) {
        const int d = (pos >> 1) + 1;
        int q = (pos & 1) ? self - d : self + d;
        q = q < 0 ? q + n1 : (q >= n1 ? q - n1 : q);
        return q;
    }
    const int p2 = pos - near;
    if (p2 >= n1)
        return n1;
    // (p2 * stride) mod n1 without an integer division: quotient from a float product, off by at most one
    const int x = p2 * stride;                                  // < 2^28 (n1 <= 12800)
    int q = x - (int)((float)x * (1.0f / (float)n1)) * n1;
    q = q < 0 ? q + n1 : q;
    q = q >= n1 ? q - n1 : q;
    // a neighbour already seen in the local pass is NOT offered again: the optimum sits exactly on its
    // binding constraints, and re-testing those in floating point reports round-off as a violation
    int d = q - self;
    d = d < 0 ? -d : d;
    d = min(d, n1 - d);
    return (d >= 1 && d <= near / 2) ? n1 : q;
}

// ---- hull vertex test ---------------------------------------------------------------------
constexpr double HPR_EPS = 1e-12;   // relative separation margin (qhull-like coplanarity tolerance)

struct Cons {
    double a, b, c;
};

// Frame of one candidate vertex p: r = unit vector from the cloud's centroid to p, (u, w) a unit
// basis of the plane perpendicular to it.  The centroid is strictly inside the hull, so every
// supporting plane at a vertex p has an outward normal d with d.r > 0, i.e. d = r + s u + t w after
// scaling: ONE two-variable chart covers every certificate (the six axis charts of the first version
// made every non-vertex pay for six refuted LPs).  (s, t) are tangents of the angle to r; the box
// |s|, |t| <= HPR_TAN cuts off normals within 1e-4 rad of the tangent plane, which a vertex of a
// flipped cloud (a thin near-spherical shell) never needs.
struct Frame {
    double px, py, pz, rx, ry, rz, ux, uy, uz, wx, wy, wz;
};
constexpr double HPR_TAN = 1e4;

// constraint of point q against p:  d.g <= -eps |g|,  g = q - p,  d = r + s u + t w
//   (u.g) s + (w.g) t <= -(r.g) - eps |g|
__device__ __forceinline__ Cons hpr_constraint(const float *__restrict__ pts, int q, const Frame &f)
{
    const double gx = (double)pts[3 * q] - f.px, gy = (double)pts[3 * q + 1] - f.py, gz = (double)pts[3 * q + 2] - f.pz;
    const double nrm = (fabs(gx) + fabs(gy)) + fabs(gz);   // L1 >= L2: the margin only has to be "tiny but positive"
    Cons k;
    k.a = (f.ux * gx + f.uy * gy) + f.uz * gz;
    k.b = (f.wx * gx + f.wy * gy) + f.wz * gz;
    k.c = -((f.rx * gx + f.ry * gy) + f.rz * gz) - HPR_EPS * nrm;
    return k;
}

// Seidel's incremental 2-variable LP on [-HPR_TAN, HPR_TAN]^2, objective x + y/2; returns feasibility.
// ONE WAVE per point: the 64 lanes test 64 consecutive constraints against the current optimum
// (ballot -> first violated one, which keeps Seidel's sequential semantics), and share the 1-D
// re-solve over the constraints seen so far (per-lane lo/hi as FRACTIONS -- compared by cross
// multiplication, so the loop has no fp64 division -- then a wave min/max).  A lane-per-point
// version diverged: a wave executed the SUM of its lanes' re-solves (227 ms per batch of 32).
// Constraint order = hpr_seq: Seidel's expected O(n) needs an order uncorrelated with the geometry, and object
// models are stored in scan order.  q == self is skipped.
struct Frac {
    double num, den;   // den > 0
};
// value of lane DPP(ctrl) of this lane's row: quad_perm [1,0,3,2] = 0xB1, [2,3,0,1] = 0x4E,
// row_half_mirror = 0x141, row_mirror = 0x140
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
    const unsigned long long u = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, 0xf, 0xf, false);
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}
__device__ __forceinline__ bool frac_less(const Frac &x, const Frac &y) { return x.num * y.den < y.num * x.den; }

template <int HPR_NEAR>
__device__ bool hpr_lp2d_wave(const float *__restrict__ pts, int n1, int self, int stride, const Frame &fr, int lane)
{
    double vx = HPR_TAN, vy = HPR_TAN;
    const float pxf = (float)fr.px, pyf = (float)fr.py, pzf = (float)fr.pz;     // exact: p is a float
    const int near = hpr_near_of(n1, HPR_NEAR);
    const int span = near + n1;
    // the plane d = r + vx u + vy w under test, and its fp32 copy: recomputed only when a re-solve moved (vx, vy)
    double dx, dy, dz;
    float dxf, dyf, dzf, slack;
    auto set_plane = [&]() {
        dx = (fr.rx + vx * fr.ux) + vy * fr.wx;
        dy = (fr.ry + vx * fr.uy) + vy * fr.wy;
        dz = (fr.rz + vx * fr.uz) + vy * fr.wz;
        dxf = (float)dx;
        dyf = (float)dy;
        dzf = (float)dz;
        slack = -4e-6f * ((fabsf(dxf) + fabsf(dyf)) + fabsf(dzf));
    };
    set_plane();
    // the strided part of the sequence advances by 64 positions per iteration: q += 64 * stride (mod n1)
    const int step64 = (int)(((long long)64 * stride) % n1);
    int qraw = 0, qat = -1;                 // qraw = ((qat + lane - near) * stride) mod n1
    int i = 0;
    while (i < span) {
        // the scan tests the current plane itself: with d = r + vx u + vy w the constraint of q reads
        // d.(q - p) <= -eps |q - p| -- a 3-D dot product per point instead of building its 2-D form.
        // It runs in fp32 first (fp64 issues at half rate and this loop is instruction-bound): a point
        // whose fp32 value is below -margin, margin = 4e-6 |d|_1 |g|_1 (>> the rounding error of the three
        // products, the sum and the fp32 copies of d and g), is satisfied for certain; only lanes inside
        // the margin repeat the test in fp64.  The decision is therefore exactly the fp64 one.
        const int pos = i + lane;
        int q;
        if (i >= near) {                    // (uniform) every lane is in the strided part
            const int p2 = pos - near;
            if (qat + 64 == i) {
                qraw += step64;
                qraw -= qraw >= n1 ? n1 : 0;
            } else {                        // first time here, or the scan restarted after a re-solve
                const int x = min(p2, n1 - 1) * stride;
                qraw = x - (int)((float)x * (1.0f / (float)n1)) * n1;
                qraw = qraw < 0 ? qraw + n1 : qraw;
                qraw = qraw >= n1 ? qraw - n1 : qraw;
            }
            qat = i;
            int dd = qraw - self;
            dd = dd < 0 ? -dd : dd;
            dd = min(dd, n1 - dd);
            q = (p2 < n1 && !(dd >= 1 && dd <= near / 2)) ? qraw : n1;
        } else {
            q = pos < span ? hpr_seq(pos, self, n1, stride, near) : n1;
        }
        const bool valid = q < n1 && q != self;
        bool viol = false;
        if (valid) {
            const float qx = pts[3 * q], qy = pts[3 * q + 1], qz = pts[3 * q + 2];
            const float gxf = qx - pxf, gyf = qy - pyf, gzf = qz - pzf;
            const float nrmf = (fabsf(gxf) + fabsf(gyf)) + fabsf(gzf);
            const float v32 = (dxf * gxf + dyf * gyf) + dzf * gzf;
            if (v32 > slack * nrmf) {
                const double gx = (double)qx - fr.px, gy = (double)qy - fr.py, gz = (double)qz - fr.pz;
                const double nrm = (fabs(gx) + fabs(gy)) + fabs(gz);
                viol = (dx * gx + dy * gy) + dz * gz > -HPR_EPS * nrm;
            }
        }
        const unsigned long long mask = __ballot(viol);
        if (mask == 0ull) {
            i += 64;
            continue;
        }
        const int first = __ffsll((long long)mask) - 1;
        const Cons kf = hpr_constraint(pts, __shfl(q, first, 64), fr);     // 2-D form of the violated one only
        const double ka = kf.a, kb = kf.b, kc = kf.c;
        const double nn = ka * ka + kb * kb;
        if (nn == 0.0)
            return false;
        const double p0x = ka * kc / nn, p0y = kb * kc / nn, ux = -kb, uy = ka;
        // t in [lo, hi] with lo = max num/den over den<0 (stored with den > 0 after sign flip)
        Frac lo = {-1e300, 1.0}, hi = {1e300, 1.0};
        bool bad = false;
        auto add = [&](double ca, double cb, double cc) {
            const double den = ca * ux + cb * uy, num = cc - (ca * p0x + cb * p0y);
            if (den > 0.0) {
                const Frac f = {num, den};
                if (frac_less(f, hi))
                    hi = f;
            } else if (den < 0.0) {
                const Frac f = {-num, -den};
                if (frac_less(lo, f))
                    lo = f;
            } else if (num < 0.0) {
                bad = true;
            }
        };
        if (lane < 4)      // the box |x| <= HPR_TAN, |y| <= HPR_TAN
            add(lane == 0 ? 1.0 : (lane == 1 ? -1.0 : 0.0), lane == 2 ? 1.0 : (lane == 3 ? -1.0 : 0.0), HPR_TAN);
        const int upto = i + first;            // sequence positions [0, upto) were already accepted
        for (int jpos = lane; jpos < upto; jpos += 64) {
            const int r = hpr_seq(jpos, self, n1, stride, near);
            if (r >= n1 || r == self)
                continue;
            const Cons m = hpr_constraint(pts, r, fr);
            add(m.a, m.b, m.c);
        }
        // wave reduction of lo (max) and hi (min) as fractions.  Inside a row of 16 lanes the partner
        // comes through DPP (quad swaps, then the half-row and row mirrors: after the quad steps a quad
        // is uniform, so the mirrors act as "xor 4" and "xor 8"); only the two cross-row steps go through
        // the LDS crossbar (ds_bpermute), which cost ~3/4 of a re-solve when all six steps used it.
        auto fold = [&](Frac ol, Frac oh) {
            if (frac_less(lo, ol))
                lo = ol;
            if (frac_less(oh, hi))
                hi = oh;
        };
        fold(Frac{dpp_f64<0xB1>(lo.num), dpp_f64<0xB1>(lo.den)}, Frac{dpp_f64<0xB1>(hi.num), dpp_f64<0xB1>(hi.den)});
        fold(Frac{dpp_f64<0x4E>(lo.num), dpp_f64<0x4E>(lo.den)}, Frac{dpp_f64<0x4E>(hi.num), dpp_f64<0x4E>(hi.den)});
        fold(Frac{dpp_f64<0x141>(lo.num), dpp_f64<0x141>(lo.den)}, Frac{dpp_f64<0x141>(hi.num), dpp_f64<0x141>(hi.den)});
        fold(Frac{dpp_f64<0x140>(lo.num), dpp_f64<0x140>(lo.den)}, Frac{dpp_f64<0x140>(hi.num), dpp_f64<0x140>(hi.den)});
#pragma unroll
        for (int off = 16; off <= 32; off <<= 1)
            fold(Frac{__shfl_xor(lo.num, off, 64), __shfl_xor(lo.den, off, 64)},
                 Frac{__shfl_xor(hi.num, off, 64), __shfl_xor(hi.den, off, 64)});
        if (__ballot(bad) != 0ull || frac_less(hi, lo))
            return false;
        const Frac pick = (ux + 0.5 * uy) > 0.0 ? hi : lo;
        const double tt = pick.num / pick.den;
        vx = p0x + tt * ux;
        vy = p0y + tt * uy;
        set_plane();
        i = upto + 1;
    }
    return true;
}

// ---- the same LP, the full scan replaced by CULLED verification passes (round 5) -----------------------------------
// hpr_lp2d_wave pays n1 / 64 iterations of ~75 vector instructions for the scan behind the local problem -- per point,
// although the plane it verifies is (almost always) already final and cuts a cap of the shell that only p's surroundings
// come near.  Here the sorted cloud is cut into GROUPS of 64 consecutive points with a bounding volume each (hpr_group_slabs:
// the spatial sort makes them compact).  After the local problem over the working set W = {the HPR_NEAR sorted-order
// neighbours}:
//   pass : every group's volume against the plane under test, a lane per group: max over the volume of d.(q - p) in fp32 at
//          or below a (negative) margin proves every point of the group satisfied -- the group is culled.
//          The points of the other groups take the point test of the scan above (fp32 with a margin, fp64 inside it),
//          a lane per point, consecutive LDS addresses, no sequence arithmetic.
//   the MOST violated point q of a pass outside W: W += {q}, the optimum moves onto q's line by the 1-D problem over W
//          (Seidel's step: the optimum of W + {q} lies on q's line; the members of W are never re-tested, as in the scan
//          above: the optimum sits exactly on its binding constraints), and another pass follows.  A pass without a
//          violation proves the plane.
// This is Seidel's algorithm with the constraints that never mattered left out of the 1-D problems: the optimum over W only
// is what the passes verify against EVERY point, W only grows, and the answer (feasible or not) is that of the full LP.  At
// most HPR_EXTRA points join W behind the neighbours; a point that needs more goes through hpr_lp2d_wave instead.
#ifdef CLOUDAAE_HPR_STATS
__device__ unsigned long long hpr_stats[16];   // points, local re-solves, passes, box iterations, groups scanned, joined, fallbacks, vertices
#define HPR_COUNT(slot, v) do { if (lane == 0) atomicAdd(&hpr_stats[slot], (unsigned long long)(v)); } while (0)
#else
#define HPR_COUNT(slot, v) do { } while (0)
#endif
constexpr int HPR_GROUP = 64;
constexpr int HPR_MAX_GROUPS = 12800 / HPR_GROUP;     // (150 KB of points: the entry point's limit)
constexpr int HPR_EXTRA = 32;

// Bounding volume of a group: a slab piece.  With m the group's mean and n a unit vector (the direction from the viewpoint
// -- the normal of the flip sphere there; ANY n keeps the bound valid), every point of the group is m + h n + t with
// h in [hlo, hhi] and t perpendicular to n, |t| <= rad: thin along n (relief + the patch's own sagitta), wide across.  For a
// plane normal d:  max d.(q - m) <= max(hlo d.n, hhi d.n) + rad |d - (d.n) n|.  An axis-aligned box does not do: its top
// corner stands above a tilted patch by slope x width, more than the cap a supporting plane cuts off a few groups away
// (measured: 54 of 135 boxes survived a pass; the slabs: see profiles/notes_hull_vertex.md).
struct HprGroups {              // LDS
    float m[3][HPR_MAX_GROUPS], n[3][HPR_MAX_GROUPS], rad[HPR_MAX_GROUPS], hlo[HPR_MAX_GROUPS], hhi[HPR_MAX_GROUPS],
        rmax[HPR_MAX_GROUPS];
};

template <int HPR_WAVES>
__device__ __forceinline__ void hpr_group_slabs(const float *__restrict__ pts, int n1, int viewpoint, HprGroups &gb)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int groups = (n1 + HPR_GROUP - 1) / HPR_GROUP;
    const float vx = pts[3 * viewpoint], vy = pts[3 * viewpoint + 1], vz = pts[3 * viewpoint + 2];
    for (int g = wave; g < groups; g += HPR_WAVES) {
        const int q = min(g * HPR_GROUP + lane, n1 - 1);       // (a short last group repeats its last point)
        const float x = pts[3 * q], y = pts[3 * q + 1], z = pts[3 * q + 2];
        float sx = x, sy = y, sz = z;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            sx += __shfl_xor(sx, off, 64);
            sy += __shfl_xor(sy, off, 64);
            sz += __shfl_xor(sz, off, 64);
        }
        const float mx = sx * (1.0f / 64.0f), my = sy * (1.0f / 64.0f), mz = sz * (1.0f / 64.0f);
        float nx = mx - vx, ny = my - vy, nz = mz - vz;
        const float len = sqrtf((nx * nx + ny * ny) + nz * nz);
        if (len > 0.0f) {
            nx /= len; ny /= len; nz /= len;
        } else {
            nx = 0.0f; ny = 0.0f; nz = 1.0f;
        }
        const float ex = x - mx, ey = y - my, ez = z - mz;
        const float h = (nx * ex + ny * ey) + nz * ez;
        const float r2 = (ex * ex + ey * ey) + ez * ez;
        float t2 = r2 - h * h, r2max = r2, hmin = h, hmax = h;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            t2 = fmaxf(t2, __shfl_xor(t2, off, 64));
            r2max = fmaxf(r2max, __shfl_xor(r2max, off, 64));
            hmin = fminf(hmin, __shfl_xor(hmin, off, 64));
            hmax = fmaxf(hmax, __shfl_xor(hmax, off, 64));
        }
        // rounded quantities, so every bound is widened: 1e-5 of the group's radius covers the fp32 evaluation of h and t
        const float rmax = sqrtf(r2max) * 1.001f;
        if (lane == 0) {
            gb.m[0][g] = mx; gb.m[1][g] = my; gb.m[2][g] = mz;
            gb.n[0][g] = nx; gb.n[1][g] = ny; gb.n[2][g] = nz;
            gb.rmax[g] = rmax;
            gb.rad[g] = sqrtf(fmaxf(t2, 0.0f) + 1e-5f * r2max) * 1.001f;
            gb.hlo[g] = hmin - 1e-5f * rmax;
            gb.hhi[g] = hmax + 1e-5f * rmax;
        }
    }
}

// returns 1 = vertex, 0 = not a vertex, -1 = more than HPR_EXTRA points joined the working set (the caller runs the scan)
template <int HPR_NEAR>
__device__ int hpr_lp2d_wave_culled(const float *__restrict__ pts, const HprGroups &gb, int *__restrict__ extras, int n1,
                                    int self, const Frame &fr, int lane, int extra_cap)
{
    double vx = HPR_TAN, vy = HPR_TAN;
    const float pxf = (float)fr.px, pyf = (float)fr.py, pzf = (float)fr.pz;     // exact: p is a float
    double dx, dy, dz;
    float dxf, dyf, dzf, slack, ddf;
    auto set_plane = [&]() {
        dx = (fr.rx + vx * fr.ux) + vy * fr.wx;
        dy = (fr.ry + vx * fr.uy) + vy * fr.wy;
        dz = (fr.rz + vx * fr.uz) + vy * fr.wz;
        dxf = (float)dx;
        dyf = (float)dy;
        dzf = (float)dz;
        slack = -4e-6f * ((fabsf(dxf) + fabsf(dyf)) + fabsf(dzf));
        ddf = (dxf * dxf + dyf * dyf) + dzf * dzf;
    };
    set_plane();
    int n_extra = 0;
    const int near = hpr_near_of(n1, HPR_NEAR);
    // member `pos` of the working set: the neighbours in sorted order, alternating sides, then the points that joined
    auto member = [&](int pos) {
        if (pos < near) {
            const int d = (pos >> 1) + 1;
            int q = (pos & 1) ? self - d : self + d;
            q = q < 0 ? q + n1 : (q >= n1 ? q - n1 : q);
            return (q < 0 || q >= n1) ? n1 : q;             // (clouds smaller than the neighbourhood)
        }
        return extras[pos - near];
    };
    // the point test of the scan: fp32 with a margin, fp64 inside it (see hpr_lp2d_wave)
    double excess = 0.0;                        // of the last violated point: d.g + eps |g| (> 0)
    auto violated = [&](int q) {
        const float qx = pts[3 * q], qy = pts[3 * q + 1], qz = pts[3 * q + 2];
        const float gxf = qx - pxf, gyf = qy - pyf, gzf = qz - pzf;
        const float nrmf = (fabsf(gxf) + fabsf(gyf)) + fabsf(gzf);
        const float v32 = (dxf * gxf + dyf * gyf) + dzf * gzf;
        bool viol = false;
        if (v32 > slack * nrmf) {
            const double gx = (double)qx - fr.px, gy = (double)qy - fr.py, gz = (double)qz - fr.pz;
            const double nrm = (fabs(gx) + fabs(gy)) + fabs(gz);
            const double v = (dx * gx + dy * gy) + dz * gz;
            viol = v > -HPR_EPS * nrm;
            excess = v + HPR_EPS * nrm;
        }
        return viol;
    };
    // Seidel's step: the optimum moves onto the line of the violated constraint of point qv, by the 1-D problem over the
    // first `upto` members of the working set; false = infeasible
    auto resolve = [&](int qv, int upto) {
        const Cons kf = hpr_constraint(pts, qv, fr);
        const double ka = kf.a, kb = kf.b, kc = kf.c;
        const double nn = ka * ka + kb * kb;
        if (nn == 0.0)
            return false;
        const double p0x = ka * kc / nn, p0y = kb * kc / nn, ux = -kb, uy = ka;
        Frac lo = {-1e300, 1.0}, hi = {1e300, 1.0};
        bool bad = false;
        auto add = [&](double ca, double cb, double cc) {
            const double den = ca * ux + cb * uy, num = cc - (ca * p0x + cb * p0y);
            if (den > 0.0) {
                const Frac f = {num, den};
                if (frac_less(f, hi))
                    hi = f;
            } else if (den < 0.0) {
                const Frac f = {-num, -den};
                if (frac_less(lo, f))
                    lo = f;
            } else if (num < 0.0) {
                bad = true;
            }
        };
        if (lane < 4)      // the box |x| <= HPR_TAN, |y| <= HPR_TAN
            add(lane == 0 ? 1.0 : (lane == 1 ? -1.0 : 0.0), lane == 2 ? 1.0 : (lane == 3 ? -1.0 : 0.0), HPR_TAN);
        for (int jpos = lane; jpos < upto; jpos += 64) {
            const int r = member(jpos);
            if (r >= n1 || r == self)
                continue;
            const Cons m = hpr_constraint(pts, r, fr);
            add(m.a, m.b, m.c);
        }
        // the wave's lo (max) and hi (min): every lane's pair as quotients (one division each -- the lanes' own lists were
        // compared as fractions, without one), then a min / max butterfly: DPP inside a row of 16 lanes (quad swaps, then the
        // half-row and row mirrors: after the quad steps a quad is uniform, so the mirrors act as "xor 4" and "xor 8"), the two
        // cross-row steps through the LDS crossbar.  (Round 4 folded the FRACTIONS, by cross multiplication: 150 of a
        // re-solve's ~280 vector instructions.)
        double tlo = lo.num / lo.den, thi = hi.num / hi.den;
        tlo = fmax(tlo, dpp_f64<0xB1>(tlo));
        thi = fmin(thi, dpp_f64<0xB1>(thi));
        tlo = fmax(tlo, dpp_f64<0x4E>(tlo));
        thi = fmin(thi, dpp_f64<0x4E>(thi));
        tlo = fmax(tlo, dpp_f64<0x141>(tlo));
        thi = fmin(thi, dpp_f64<0x141>(thi));
        tlo = fmax(tlo, dpp_f64<0x140>(tlo));
        thi = fmin(thi, dpp_f64<0x140>(thi));
#pragma unroll
        for (int off = 16; off <= 32; off <<= 1) {
            tlo = fmax(tlo, __shfl_xor(tlo, off, 64));
            thi = fmin(thi, __shfl_xor(thi, off, 64));
        }
        if (__ballot(bad) != 0ull || thi < tlo)
            return false;
        const double tt = (ux + 0.5 * uy) > 0.0 ? thi : tlo;
        vx = p0x + tt * ux;
        vy = p0y + tt * uy;
        set_plane();
        return true;
    };

    // ---- the local problem: the neighbours in sequence (Seidel over W's first HPR_NEAR members) ----
    int i = 0;
    while (i < near) {
        const int q = member(i + lane);
        const bool viol = (i + lane < near && q < n1 && q != self) ? violated(q) : false;
        const unsigned long long mask = __ballot(viol);
        if (mask == 0ull) {
            i += 64;
            continue;
        }
        const int first = __ffsll((long long)mask) - 1;
        HPR_COUNT(1, 1);
        HPR_COUNT(8, (i + first + 63) / 64);             // 64-constraint iterations of the local re-solves
        HPR_COUNT(9 + min(3, (i + first) / 128), 1);    // where they happen: [0,128) [128,256) [256,384) [384,512)
        if (!resolve(__shfl(q, first, 64), i + first)) {
            HPR_COUNT(13, 1);                            // rejected inside the local problem
            HPR_COUNT(14, i + first);
            return 0;
        }
        i += first + 1;
    }

    // ---- verification passes ----
    const int groups = (n1 + HPR_GROUP - 1) / HPR_GROUP;
    const float slack_g = 2.5f * slack;         // -1e-5 |d|_1: the margin of the slab test
    for (;;) {
        // the MOST violated point of the pass joins (by d.g + eps |g|: the plane's own measure); every lane keeps its worst
        double worst = 0.0;
        int worst_q = -1;
        HPR_COUNT(2, 1);
        for (int g0 = 0; g0 < groups; g0 += 64) {
            HPR_COUNT(3, 1);
            // a lane per group: can a point of its slab violate?
            unsigned long long open;
            {
                const int g = min(g0 + lane, groups - 1);
                const float ex = gb.m[0][g] - pxf, ey = gb.m[1][g] - pyf, ez = gb.m[2][g] - pzf;
                const float dm = (dxf * ex + dyf * ey) + dzf * ez;
                const float dn = (dxf * gb.n[0][g] + dyf * gb.n[1][g]) + dzf * gb.n[2][g];
                const float perp = sqrtf(fmaxf(ddf - dn * dn, 0.0f) + 1e-6f * ddf);    // |d - (d.n) n|, rounded up
                const float rm = gb.rmax[g];
                const float top = (dm + fmaxf(gb.hlo[g] * dn, gb.hhi[g] * dn)) + gb.rad[g] * perp;
                const float far = ((fabsf(ex) + fabsf(ey)) + fabsf(ez)) + 2.0f * rm;
                // culled: top <= -1e-5 |d|_1 far -- the fp32 evaluation of top is off by < 1e-6 |d|_1 far, the fp32 copy of d
                // by less still, and what has to hold for every point is only d.g <= -1e-12 |g|_1
                open = __ballot(g0 + lane < groups && !(top <= slack_g * far));
            }
            while (open != 0ull) {
                const int gi = __ffsll((long long)open) - 1;
                open &= open - 1ull;
                HPR_COUNT(4, 1);
                const int q = (g0 + gi) * HPR_GROUP + lane;
                bool viol = q < n1 ? violated(q) : false;
                if (__ballot(viol) == 0ull)
                    continue;
                // (rare) members of W are not re-tested: the neighbours by their distance in sorted order, the joined ones
                // by the list
                if (viol) {
                    int dd = q - self;
                    dd = dd < 0 ? -dd : dd;
                    dd = min(dd, n1 - dd);
                    viol = dd > near / 2;
                    for (int e = 0; e < n_extra; ++e)
                        viol = viol && extras[e] != q;
                    if (viol && (worst_q < 0 || excess > worst)) {
                        worst = excess;
                        worst_q = q;
                    }
                }
            }
        }
        if (__ballot(worst_q >= 0) == 0ull) {
            HPR_COUNT(7, 1);
            return 1;
        }
        if (n_extra >= extra_cap) {          // (HPR_EXTRA; the tests lower it to send points through the scan)
            HPR_COUNT(6, 1);
            return -1;
        }
        HPR_COUNT(5, 1);
        if (worst_q < 0)
            worst = -1.0;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const double ov = __shfl_xor(worst, off, 64);
            const int oq = __shfl_xor(worst_q, off, 64);
            if (ov > worst || (ov == worst && oq > worst_q)) {
                worst = ov;
                worst_q = oq;
            }
        }
        if (!resolve(worst_q, near + n_extra))
            return 0;
        if (lane == 0)
            extras[n_extra] = worst_q;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        ++n_extra;
    }
}

// flags[h][j] = 1 iff point j of cloud h (n1 points, the last one is the viewpoint) is a
// vertex of the convex hull.  One wave per point; a workgroup (8 waves; 16 when the cloud is so large that a CU
// holds one workgroup anyway) keeps the cloud in dynamic LDS and walks points j = blockIdx.x*W + wave, + W*gridDim.x, ...
template <int HPR_WAVES>
__global__ __launch_bounds__(64 * HPR_WAVES) void hull_vertex_kernel(int n1, const float *__restrict__ points,
                                                                    const int *__restrict__ perm, int stride, int culled,
                                                                    int *__restrict__ next_point,
                                                                    unsigned char *__restrict__ flags)
{
    extern __shared__ float pts[];
    __shared__ double cen[3][HPR_WAVES];
    __shared__ HprGroups gb;
    __shared__ int queue_at;
    if (threadIdx.x == 0)
        queue_at = next_point ? __hip_atomic_load(&next_point[blockIdx.y], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    __syncthreads();
    if (queue_at >= n1)
        return;                              // (every point of this cloud has a wave already; the whole workgroup leaves)
    __shared__ int extras[HPR_WAVES][HPR_EXTRA];
    const float *P = points + (size_t)blockIdx.y * n1 * 3;
    for (int f = threadIdx.x; f < n1 * 3; f += 64 * HPR_WAVES)
        pts[f] = P[f];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (culled) {
        // the viewpoint (the cloud's last point) in sorted order: the slabs' normals point away from it
        __shared__ int viewpoint;
        for (int j = threadIdx.x; j < n1; j += 64 * HPR_WAVES)
            if (perm[(size_t)blockIdx.y * n1 + j] == n1 - 1)
                viewpoint = j;
        __syncthreads();
        hpr_group_slabs<HPR_WAVES>(pts, n1, viewpoint, gb);
    }
    // centroid of the cloud (strictly inside its hull)
    double sx = 0.0, sy = 0.0, sz = 0.0;
    for (int q = threadIdx.x; q < n1; q += 64 * HPR_WAVES) {
        sx += (double)pts[3 * q];
        sy += (double)pts[3 * q + 1];
        sz += (double)pts[3 * q + 2];
    }
    sx = wave_sum(sx);
    sy = wave_sum(sy);
    sz = wave_sum(sz);
    if (lane == 0) {
        cen[0][wave] = sx;
        cen[1][wave] = sy;
        cen[2][wave] = sz;
    }
    __syncthreads();
    double cx = 0.0, cy = 0.0, cz = 0.0;
    for (int w = 0; w < HPR_WAVES; ++w) {
        cx += cen[0][w];
        cy += cen[1][w];
        cz += cen[2][w];
    }
    cx /= (double)n1;
    cy /= (double)n1;
    cz /= (double)n1;
    // the cloud's points are handed out one at a time (a counter per cloud, zeroed by the sort kernel): a hull vertex costs
    // its wave several passes, an interior point leaves inside the local problem -- with a fixed share per wave the
    // workgroup waited for its unluckiest wave
    for (int turn = 0;; ++turn) {
        int j = 0;
        if (next_point != nullptr) {
            if (lane == 0)
                j = atomicAdd(&next_point[blockIdx.y], 1);
            j = __builtin_amdgcn_readfirstlane(j);      // (uniform for the compiler as well: with __shfl(j, 0) the kernel hung)
        } else {
            // (a fixed share per wave: round 4's hand-out.  The host always passes the queue since round 6 -- but WITHOUT this
            //  branch around the atomic the compiled kernel hangs (hipcc 7.2; as it did with __shfl(j, 0) in place of the
            //  readfirstlane): the form that is known to work stays)
            j = blockIdx.x * HPR_WAVES + wave + turn * HPR_WAVES * gridDim.x;
        }
        if (j >= n1)
            break;
        Frame f;
        f.px = pts[3 * j];
        f.py = pts[3 * j + 1];
        f.pz = pts[3 * j + 2];
        double rx = f.px - cx, ry = f.py - cy, rz = f.pz - cz;
        const double rn = sqrt((rx * rx + ry * ry) + rz * rz);
        bool vertex = false;
        if (rn > 0.0) {
            rx /= rn;
            ry /= rn;
            rz /= rn;
            // u = r x e (e = the axis r is least aligned with), w = r x u
            const double ax = fabs(rx), ay = fabs(ry), az = fabs(rz);
            double ux, uy, uz;
            if (ax <= ay && ax <= az) {
                ux = 0.0; uy = rz; uz = -ry;
            } else if (ay <= az) {
                ux = -rz; uy = 0.0; uz = rx;
            } else {
                ux = ry; uy = -rx; uz = 0.0;
            }
            const double un = sqrt((ux * ux + uy * uy) + uz * uz);
            ux /= un;
            uy /= un;
            uz /= un;
            f.rx = rx; f.ry = ry; f.rz = rz;
            f.ux = ux; f.uy = uy; f.uz = uz;
            f.wx = ry * uz - rz * uy;
            f.wy = rz * ux - rx * uz;
            f.wz = rx * uy - ry * ux;
            constexpr int NEAR = HPR_WAVES > 8 ? HPR_NEAR_LARGE : HPR_NEAR_SMALL;
            int verdict = -1;
            HPR_COUNT(0, 1);
            if (culled)
                verdict = hpr_lp2d_wave_culled<NEAR>(pts, gb, extras[wave], n1, j, f, lane, min(culled - 1, HPR_EXTRA));
            vertex = verdict >= 0 ? verdict == 1 : hpr_lp2d_wave<NEAR>(pts, n1, j, stride, f, lane);
        }
        if (lane == 0)     // `points` is the spatially sorted cloud: the flag goes back to the original index
            flags[(size_t)blockIdx.y * n1 + perm[(size_t)blockIdx.y * n1 + j]] = vertex ? 1 : 0;
    }
}

// convexHull() of hidden_point_removal.py:27-43, given the vertex flags: V = sorted vertex ids;
// drop the largest (`hull.vertices[:-1]`, the viewpoint) and the next largest (`visibleId[:-1]`);
// rows [0, num_vis) = org[visibleId], the remaining rows = org[random choice of visibleId].
__global__ __launch_bounds__(512) void hpr_gather_kernel(int n1, const unsigned char *__restrict__ flags,
                                                        const float *__restrict__ org, unsigned long long seed,
                                                        float *__restrict__ visible, long long *__restrict__ num_vis,
                                                        int *__restrict__ visible_id, int *__restrict__ row_src, int rows)
{
    extern __shared__ int ids[];          // n1 ints: compacted vertex ids
    __shared__ int wsum[8];
    __shared__ int total_s;
    const int h = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const unsigned char *F = flags + (size_t)h * n1;
    const int per = (n1 + 511) / 512;
    const int lo = min(n1, t * per), hi = min(n1, lo + per);
    int local = 0;
    for (int j = lo; j < hi; ++j)
        local += F[j];
    int incl = local;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, d, 64);
        if (lane >= d)
            incl += o;
    }
    if (lane == 63)
        wsum[wave] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w)
        base += wsum[w];
    int pos = base + incl - local;
    for (int j = lo; j < hi; ++j)
        if (F[j])
            ids[pos++] = j;
    if (t == 511)
        total_s = base + incl;
    __syncthreads();
    const int nv = max(0, total_s - 2);
    if (t == 0)
        num_vis[h] = nv;
    const float *O = org + (size_t)h * n1 * 3;
    float *V = visible + (size_t)h * rows * 3;     // rows = n1 in the reference (hidden_point_removal.py:38-40)
    for (int r = t; r < rows; r += 512) {
        int src = -1, row = -1;               // row: the output row (< nv) this row equals
        if (r < nv) {
            src = ids[r];
            row = r;
        } else if (nv > 0) {
            unsigned rnd[4];
            philox4x32(seed, ((unsigned long long)h << 32) | (unsigned)r, 7u, rnd);
            row = (int)(rnd[0] % (unsigned)nv);
            src = ids[row];                          // np.random.choice(visibleId, ...)
        }
        if (visible_id)
            visible_id[(size_t)h * rows + r] = r < nv ? src : -1;
        if (row_src)
            row_src[(size_t)h * rows + r] = row;
        V[3 * r] = src >= 0 ? O[3 * src] : 0.0f;
        V[3 * r + 1] = src >= 0 ? O[3 * src + 1] : 0.0f;
        V[3 * r + 2] = src >= 0 ? O[3 * src + 2] : 0.0f;
    }
}

// stride of hpr_seq: ~ n / golden ratio, coprime to n
static int hpr_stride(int n)
{
    auto gcd = [](int a, int b) {
        while (b) {
            const int t = a % b;
            a = b;
            b = t;
        }
        return a;
    };
    int a = (int)(n * 0.6180339887498949);
    a = a < 1 ? 1 : a;
    while (gcd(a, n) != 1)
        ++a;
    return a;
}

} // namespace cloudaae

using namespace cloudaae;

#ifdef CLOUDAAE_HPR_STATS
CLOUDAAE_API int cloudaae_hpr_stats_read(unsigned long long *out, int reset)
{
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out, HIP_SYMBOL(hpr_stats), sizeof(hpr_stats));
    if (reset) {
        unsigned long long z[16] = {};
        hipMemcpyToSymbol(HIP_SYMBOL(hpr_stats), z, sizeof(z));
    }
    return 0;
}
#endif

CLOUDAAE_API int cloudaae_transform_object_model(int b, int npts, int nmodels, const float *models,
                                                 const long long *class_id, const double *rot, const float *trans,
                                                 float *out, cloudaae_stream_t stream)
{
    if (b * npts == 0)
        return 0;
    hipLaunchKernelGGL(transform_model_kernel, dim3(ceil_div(b * npts, 256)), dim3(256), 0, (hipStream_t)stream, b,
                       npts, nmodels, models, class_id, rot, trans, out);
    CLOUDAAE_CHECK_LAUNCH("cloudaae_transform_object_model");
    return 0;
}

CLOUDAAE_API int cloudaae_random_spherical_occluder(int b, int per_blob, const float *trans, float wnear,
                                                    float hnear, float near_dist, float sigma,
                                                    unsigned long long seed, float *occluder,
                                                    cloudaae_stream_t stream)
{
    if (b * per_blob == 0)
        return 0;
    hipLaunchKernelGGL(occluder_kernel, dim3(ceil_div(b * per_blob, 256)), dim3(256), 0, (hipStream_t)stream, b,
                       per_blob, trans, wnear, hnear, near_dist, sigma, seed, occluder);
    CLOUDAAE_CHECK_LAUNCH("cloudaae_random_spherical_occluder");
    return 0;
}

CLOUDAAE_API int cloudaae_spherical_flip(int b, int na, const float *a, int nb, const float *bpts,
                                         const float *center, float param, float *flipped, float *org,
                                         cloudaae_stream_t stream)
{
    const char *name = "cloudaae_spherical_flip";
    CLOUDAAE_REQUIRE(b >= 0 && na >= 0 && nb >= 0 && na + nb > 0 && b <= 65535, name, "bad size");
    if (b == 0)
        return 0;
    hipLaunchKernelGGL(spherical_flip_kernel, dim3(b), dim3(256), 0, (hipStream_t)stream, na, nb, a, bpts, center,
                       powf(10.0f, param), flipped, org);
    CLOUDAAE_CHECK_LAUNCH(name);
    return 0;
}

// flags [b*n1] bytes (padded to 16) | sorted points [b*n1*3] floats | permutation [b*n1] ints | point queues [b] ints
CLOUDAAE_API long long cloudaae_hpr_workspace_bytes(int b, int n1)
{
    const long long pts = (long long)b * n1;
    return (pts + 15) / 16 * 16 + pts * 12 + pts * 4 + (long long)b * 4;
}

CLOUDAAE_API int cloudaae_hidden_point_removal(int b, int n1, const float *flipped, const float *org,
                                               unsigned long long seed, float *visible, long long *num_vis,
                                               int *visible_id, void *workspace, cloudaae_stream_t stream)
{
    return cloudaae_hidden_point_removal_rows(b, n1, flipped, org, seed, n1, visible, num_vis, visible_id, nullptr,
                                              workspace, stream);
}

CLOUDAAE_API int cloudaae_hidden_point_removal_rows(int b, int n1, const float *flipped, const float *org,
                                                    unsigned long long seed, int rows, float *visible,
                                                    long long *num_vis, int *visible_id, int *row_src, void *workspace,
                                                    cloudaae_stream_t stream)
{
    const char *name = "cloudaae_hidden_point_removal";
    CLOUDAAE_REQUIRE(b >= 0 && n1 >= 5 && b <= 65535 && workspace, name, "bad size (need >= 4 points + viewpoint)");
    CLOUDAAE_REQUIRE(rows >= 1, name, "bad number of output rows");
    // the cloud (12 bytes a point, dynamic LDS) next to the kernel's own static LDS (group slabs, working-set lists, the
    // centroid sums: ~10 KB, read from the code object) must fit the CU's 160 KB
    static size_t static_lds[2] = {0, 0};
    if (static_lds[0] == 0) {
        hipFuncAttributes a8, a16;
        CLOUDAAE_CHECK_HIP(hipFuncGetAttributes(&a8, (const void *)hull_vertex_kernel<8>), name);
        CLOUDAAE_CHECK_HIP(hipFuncGetAttributes(&a16, (const void *)hull_vertex_kernel<16>), name);
        static_lds[1] = a16.sharedSizeBytes;
        static_lds[0] = a8.sharedSizeBytes;
    }
    const size_t lds = (size_t)n1 * 3 * sizeof(float);
    CLOUDAAE_REQUIRE(lds + static_lds[1] <= 160 * 1024, name, "cloud too large for the LDS-resident hull test");
    if (b == 0)
        return 0;
    hipStream_t s = (hipStream_t)stream;
    unsigned char *flags = (unsigned char *)workspace;
    const size_t pts = (size_t)b * n1;
    float *sorted = (float *)(flags + (pts + 15) / 16 * 16);
    int *perm = (int *)(sorted + pts * 3);
    int *next_point = perm + pts;
    // spatial order of every cloud: the 3-D grid in Morton order
    {
        static bool raised = false;
        if (!raised) {
            CLOUDAAE_CHECK_HIP(hipFuncSetAttribute((const void *)hpr_sort_grid_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                   (int)(HG_CELLS * sizeof(int))), name);
            raised = true;
        }
        hipLaunchKernelGGL(hpr_sort_grid_kernel, dim3(b), dim3(HS_THREADS), HG_CELLS * sizeof(int), s, n1, flipped, sorted, perm,
                           next_point);
    }
    // a cloud that leaves room for ONE workgroup per CU (more than ~5900 points) takes 16 waves instead of 8
    const bool wide = 2 * (lds + static_lds[0]) > 160 * 1024;
    const int waves = wide ? 16 : 8;
    if (lds > 48 * 1024)
        CLOUDAAE_CHECK_HIP(hipFuncSetAttribute(wide ? (const void *)hull_vertex_kernel<16> : (const void *)hull_vertex_kernel<8>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), name);
    // workgroups per cloud: two 8-wave workgroups fit a CU (512 on the chip), one of the 16-wave form (256).  The points of a
    // cloud come from its queue, so a workgroup lives as long as its cloud has points: exactly the resident number is launched.
    // (2 / 3 / 4 x as many -- late workgroups joining the clouds that still have points -- cost more in set-up than they
    // balance: 4.18 / 4.52 / 6.56 / 7.64 ms per config-5 batch; a fixed share of points per wave, round 4's: 6.55 ms.)
    const long long resident = wide ? 256 : 512;
    const int gx = (int)std::min<long long>(std::max<long long>(1, (resident + b - 1) / b), ceil_div(n1, waves * 2));
    // the scan behind the local problem as culled verification passes (knob CLOUDAAE_HPR_CULL = 0: the full strided scan; a value
    // of 2 .. 32 caps the points that may join a working set -- HPR_EXTRA = 32 by default -- so that the tests can send
    // points through the fallback, 1 = the default)
    const int cull_knob = CLOUDAAE_KNOB("CLOUDAAE_HPR_CULL", 1);
    const int culled = cull_knob <= 0 ? 0 : (cull_knob == 1 ? HPR_EXTRA + 1 : cull_knob - 1);    // 0 = off, else the cap + 1
    if (wide)
        hipLaunchKernelGGL(hull_vertex_kernel<16>, dim3(gx, b), dim3(64 * 16), lds, s, n1, sorted, perm, hpr_stride(n1), culled,
                           next_point, flags);
    else
        hipLaunchKernelGGL(hull_vertex_kernel<8>, dim3(gx, b), dim3(64 * 8), lds, s, n1, sorted, perm, hpr_stride(n1), culled,
                           next_point, flags);
    hipLaunchKernelGGL(hpr_gather_kernel, dim3(b), dim3(512), (size_t)n1 * sizeof(int), s, n1, flags, org, seed,
                       visible, num_vis, visible_id, row_src, rows);
    CLOUDAAE_CHECK_LAUNCH(name);
    return 0;
}
