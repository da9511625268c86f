// nefii_shading.hip - spherical-Gaussian shading kernels for gfx950.
//
//   render_with_sg                      code/model/sg_render.py:164-295 (+ hemisphere_int :112, lambda_trick :141)
//   IDRNetwork.get_background_rgb       code/model/implicit_differentiable_renderer.py:646-663
//
// Work decomposition: thread <-> light lobe, workgroup <-> a strided set of surface points.  For one
// (point, lobe) pair the reference's ~40 [N,M,1,3] intermediates collapse into two scalars
//     spec_c = |mu_c| * F_c * S(point, lobe)         diffuse_c = |mu_c| * albedo_c/pi * D(point, lobe)
// (everything is linear in the lobe amplitude and in the Fresnel term), which live in registers; the sum
// over lobes is a wave shuffle reduction + one LDS hop.  The backward pass evaluates the same expression on
// forward-mode dual numbers (tangents: lobe axis xyz, lambda, roughness) - the per-lobe gradient is
// accumulated in registers across the workgroup's points and leaves with ONE atomic per element per
// workgroup, the global roughness/specular gradients likewise.
#include <hip/hip_runtime.h>
#include "../../include/nefii_amd.h"

#define HIP_CHECK_LAUNCH()                       \
    do {                                         \
        hipError_t _e = hipGetLastError();       \
        if (_e != hipSuccess) return (int)_e;    \
    } while (0)

namespace {

constexpr float TINY = 1e-6f;
constexpr float PI_F = 3.14159265358979323846f;
constexpr float MU_COS = 32.7080f, LAMBDA_COS = 0.0315f, ALPHA_COS = 31.7003f;

// ---- forward-mode dual numbers ------------------------------------------------------------------
template <int N>
struct Dual {
    float v;
    float d[N];
    __device__ Dual() {}
    __device__ Dual(float x) : v(x) {
#pragma unroll
        for (int i = 0; i < N; ++i) d[i] = 0.f;
    }
};
template <int N>
__device__ __forceinline__ Dual<N> seed(float x, int i) {
    Dual<N> r(x);
    r.d[i] = 1.f;
    return r;
}
#define DUAL_BIN(OP, VAL, DER)                                                              \
    template <int N>                                                                        \
    __device__ __forceinline__ Dual<N> operator OP(const Dual<N> &a, const Dual<N> &b) {   \
        Dual<N> r;                                                                          \
        r.v = VAL;                                                                          \
        _Pragma("unroll") for (int i = 0; i < N; ++i) r.d[i] = DER;                         \
        return r;                                                                           \
    }
DUAL_BIN(+, a.v + b.v, a.d[i] + b.d[i])
DUAL_BIN(-, a.v - b.v, a.d[i] - b.d[i])
DUAL_BIN(*, a.v *b.v, a.d[i] * b.v + a.v * b.d[i])
template <int N>
__device__ __forceinline__ Dual<N> operator/(const Dual<N> &a, const Dual<N> &b) {
    Dual<N> r;
    const float inv = 1.f / b.v;
    r.v = a.v / b.v;
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * inv;
    return r;
}
template <int N>
__device__ __forceinline__ Dual<N> operator+(const Dual<N> &a, float b) { Dual<N> r = a; r.v += b; return r; }
template <int N>
__device__ __forceinline__ Dual<N> operator+(float b, const Dual<N> &a) { return a + b; }
template <int N>
__device__ __forceinline__ Dual<N> operator-(const Dual<N> &a, float b) { Dual<N> r = a; r.v -= b; return r; }
template <int N>
__device__ __forceinline__ Dual<N> operator-(float b, const Dual<N> &a) {
    Dual<N> r;
    r.v = b - a.v;
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = -a.d[i];
    return r;
}
template <int N>
__device__ __forceinline__ Dual<N> operator-(const Dual<N> &a) { return 0.f - a; }
template <int N>
__device__ __forceinline__ Dual<N> operator*(const Dual<N> &a, float b) {
    Dual<N> r;
    r.v = a.v * b;
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * b;
    return r;
}
template <int N>
__device__ __forceinline__ Dual<N> operator*(float b, const Dual<N> &a) { return a * b; }
template <int N>
__device__ __forceinline__ Dual<N> operator/(const Dual<N> &a, float b) { return a * (1.f / b); }
template <int N>
__device__ __forceinline__ Dual<N> operator/(float a, const Dual<N> &b) { return Dual<N>(a) / b; }

__device__ __forceinline__ float val(float x) { return x; }
template <int N>
__device__ __forceinline__ float val(const Dual<N> &x) { return x.v; }

__device__ __forceinline__ float t_sqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ float t_exp(float x) { return expf(x); }
__device__ __forceinline__ float t_abs(float x) { return fabsf(x); }
template <int N>
__device__ __forceinline__ Dual<N> t_sqrt(const Dual<N> &a) {
    Dual<N> r;
    r.v = sqrtf(a.v);
    const float k = 0.5f / r.v;
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * k;
    return r;
}
template <int N>
__device__ __forceinline__ Dual<N> t_exp(const Dual<N> &a) {
    Dual<N> r;
    r.v = expf(a.v);
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * r.v;
    return r;
}
template <int N>
__device__ __forceinline__ Dual<N> t_abs(const Dual<N> &a) {
    Dual<N> r;
    const float s = a.v > 0.f ? 1.f : (a.v < 0.f ? -1.f : 0.f);
    r.v = fabsf(a.v);
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * s;
    return r;
}
template <class T>
__device__ __forceinline__ T t_min(const T &a, const T &b) { return val(a) <= val(b) ? a : b; }
template <class T>
__device__ __forceinline__ T t_clamp_min(const T &a, float lo) { return val(a) >= lo ? a : T(lo); }
template <class T>
__device__ __forceinline__ T t_clamp_max(const T &a, float hi) { return val(a) <= hi ? a : T(hi); }

template <class T>
struct V3 {
    T x, y, z;
};
template <class T>
__device__ __forceinline__ T dot(const V3<T> &a, const V3<T> &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <class T>
__device__ __forceinline__ T dotf(const V3<T> &a, const float *b) { return a.x * b[0] + a.y * b[1] + a.z * b[2]; }
template <class T>
__device__ __forceinline__ V3<T> normalize_tiny(const V3<T> &a) {
    const T inv = 1.f / (t_sqrt(dot(a, a)) + TINY);
    return {a.x * inv, a.y * inv, a.z * inv};
}

// hemisphere_int (sg_render.py:112-138)
template <class T>
__device__ __forceinline__ T hemi(T lam, const T &cosb) {
    lam = lam + TINY;
    const T inv = 1.f / lam;
    const T t = t_sqrt(lam) * (1.6988f + 10.8438f * inv) / (1.f + 6.2201f * inv + 10.2415f * inv * inv);
    const T ia = t_exp(-t);
    const float pos = val(cosb) >= 0.f ? 1.f : 0.f;
    const T ib = t_exp(-t * t_clamp_min(cosb, 0.f));
    const T s1 = (1.f - ia * ib) / (1.f - ia + ib - ia * ib);
    const T b = t_exp(t * t_clamp_max(cosb, 0.f));
    const T s2 = (b - ia) / ((1.f - ia) * (b + 1.f));
    const T s = pos * s1 + (1.f - pos) * s2;
    const T ab = 2.f * PI_F / lam * (t_exp(-lam) - t_exp(-2.f * lam));
    const T au = 2.f * PI_F / lam * (1.f - t_exp(-lam));
    return ab * (1.f - s) + au * s;
}

// integral of (SG(ax, lam, 1) * clamped cosine about n) over the sphere: the "cosine SG" product trick
// (sg_render.py:243-252 / :278-284) for a unit-amplitude lobe
template <class T>
__device__ __forceinline__ T cosine_core(const float *n, const V3<T> &ax, const T &lam) {
    const T ratio = LAMBDA_COS / lam;
    const T d = dotf(ax, n);
    T tmp = t_sqrt(ratio * ratio + 1.f + 2.f * ratio * d);
    tmp = t_min(tmp, ratio + 1.f);
    const T lamp = lam * tmp;
    const T c1 = ratio / tmp, c2 = 1.f / tmp;
    const V3<T> axp = {c1 * n[0] + c2 * ax.x, c1 * n[1] + c2 * ax.y, c1 * n[2] + c2 * ax.z};
    const T mup = MU_COS * t_exp(lam * (tmp - ratio - 1.f));
    return mup * hemi(lamp, dotf(axp, n)) - ALPHA_COS * hemi(lam, d);
}

// per-point quantities that do not depend on the light lobe
template <class T>
struct PointTerms {
    V3<float> w_ax;     // warped BRDF lobe axis
    T w_lam;            // its sharpness (depends on roughness)
    T w_mu;             // b_mu * G / (4 d1 d2 + tiny): amplitude without Fresnel
    float fres_p;       // 2^(-(5.55473 vh + 6.8316) vh)
};

template <class T>
__device__ __forceinline__ PointTerms<T> point_terms(const float *n, const float *v, const T &rough) {
    PointTerms<T> pt;
    const T r4i = 1.f / (rough * rough * rough * rough);
    const T b_lam = 2.f * r4i, b_mu = r4i / PI_F;
    const float vn = fmaxf(n[0] * v[0] + n[1] * v[1] + n[2] * v[2], 0.f);
    V3<float> w = {2.f * vn * n[0] - v[0], 2.f * vn * n[1] - v[1], 2.f * vn * n[2] - v[2]};
    w = normalize_tiny(w);
    pt.w_ax = w;
    pt.w_lam = b_lam / (4.f * vn + TINY);
    V3<float> h = {w.x + v[0], w.y + v[1], w.z + v[2]};
    h = normalize_tiny(h);
    const float vh = fmaxf(v[0] * h.x + v[1] * h.y + v[2] * h.z, 0.f);
    pt.fres_p = exp2f(-(5.55473f * vh + 6.8316f) * vh);
    const float d1 = fmaxf(w.x * n[0] + w.y * n[1] + w.z * n[2], 0.f);
    const float d2 = fmaxf(v[0] * n[0] + v[1] * n[1] + v[2] * n[2], 0.f);
    const T k = (rough + 1.f) * (rough + 1.f) / 8.f;
    const T g = (d1 / (d1 * (1.f - k) + k + TINY)) * (d2 / (d2 * (1.f - k) + k + TINY));
    pt.w_mu = b_mu * (g / (4.f * d1 * d2 + TINY));
    return pt;
}

// S and D of one (point, lobe) pair.  raw lobe parameters: axis (unnormalised) and lambda (any sign)
template <class T>
__device__ __forceinline__ void pair_terms(const float *n, const PointTerms<T> &pt, const V3<T> &axis_raw,
                                           const T &lam_raw, T &S, T &D) {
    const V3<T> l_ax = normalize_tiny(axis_raw);
    const T l_lam = t_abs(lam_raw);
    // light SG x warped BRDF SG (lambda_trick, assumes l_lam << w_lam)
    const T ratio = l_lam / pt.w_lam;
    const T d = l_ax.x * pt.w_ax.x + l_ax.y * pt.w_ax.y + l_ax.z * pt.w_ax.z;
    T tmp = t_sqrt(ratio * ratio + 1.f + 2.f * ratio * d);
    tmp = t_min(tmp, ratio + 1.f);
    const T lam3 = pt.w_lam * tmp;
    const T c1 = ratio / tmp, c2 = 1.f / tmp;
    const V3<T> ax3 = {c1 * l_ax.x + c2 * pt.w_ax.x, c1 * l_ax.y + c2 * pt.w_ax.y, c1 * l_ax.z + c2 * pt.w_ax.z};
    const T e1 = t_exp(pt.w_lam * (tmp - ratio - 1.f));
    S = pt.w_mu * e1 * cosine_core(n, ax3, lam3);
    D = cosine_core(n, l_ax, l_lam);
}

__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    return x;
}

// block-wide sum of K values per thread; result valid in every thread.  blockDim.x multiple of 64, <= 256
template <int K>
__device__ __forceinline__ void block_sum(float (&x)[K], float *scratch /* [4*K] */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int i = 0; i < K; ++i) x[i] = wave_sum(x[i]);
    __syncthreads();
    if (lane == 0)
#pragma unroll
        for (int i = 0; i < K; ++i) scratch[wave * K + i] = x[i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < K; ++i) {
        float s = 0.f;
        for (int w = 0; w < nw; ++w) s += scratch[w * K + i];
        x[i] = s;
    }
}

constexpr int SG_THREADS = 128;

__global__ __launch_bounds__(SG_THREADS) void sg_render_fwd_kernel(const float *__restrict__ lgt, int M,
                                                                   const float *__restrict__ spec,
                                                                   const float *__restrict__ rough,
                                                                   const float *__restrict__ albedo,
                                                                   const float *__restrict__ normal,
                                                                   const float *__restrict__ view, int64_t n,
                                                                   float *__restrict__ rgb, float *__restrict__ srgb,
                                                                   float *__restrict__ drgb) {
    __shared__ float scratch[4 * 6];
    const float r = rough[0];
    const float s3[3] = {spec[0], spec[1], spec[2]};
    for (int64_t p = blockIdx.x; p < n; p += gridDim.x) {
        const float nn[3] = {normal[p * 3], normal[p * 3 + 1], normal[p * 3 + 2]};
        const float vv[3] = {view[p * 3], view[p * 3 + 1], view[p * 3 + 2]};
        const PointTerms<float> pt = point_terms<float>(nn, vv, r);
        float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int m = threadIdx.x; m < M; m += blockDim.x) {
            const float *L = lgt + m * 7;
            V3<float> ax = {L[0], L[1], L[2]};
            float S, D;
            pair_terms<float>(nn, pt, ax, L[3], S, D);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float mu = fabsf(L[4 + c]);
                const float F = s3[c] + (1.f - s3[c]) * pt.fres_p;
                acc[c] += mu * F * S;
                acc[3 + c] += mu * D;
            }
        }
        block_sum<6>(acc, scratch);
        if (threadIdx.x < 3) {
            const int c = threadIdx.x;
            const float sp = fmaxf(acc[c], 0.f);
            const float df = fmaxf(acc[3 + c] * (albedo[p * 3 + c] / PI_F), 0.f);
            srgb[p * 3 + c] = sp;
            drgb[p * 3 + c] = df;
            rgb[p * 3 + c] = sp + df;
        }
    }
}

__global__ __launch_bounds__(SG_THREADS) void sg_render_bwd_kernel(
    const float *__restrict__ lgt, int M, const float *__restrict__ spec, const float *__restrict__ rough,
    const float *__restrict__ albedo, const float *__restrict__ normal, const float *__restrict__ view, int64_t n,
    const float *__restrict__ d_rgb, const float *__restrict__ d_spec, const float *__restrict__ d_diff,
    float *__restrict__ g_albedo, float *__restrict__ g_rough, float *__restrict__ g_spec, float *__restrict__ g_lgt) {
    typedef Dual<5> T;      // tangents: axis x,y,z | lambda | roughness
    __shared__ float scratch[4 * 6];
    const float r = rough[0];
    const float s3[3] = {spec[0], spec[1], spec[2]};
    const T rd = seed<5>(r, 4);
    // this block assumes M <= blockDim.x * LOBES_PER_THREAD
    constexpr int LPT = 2;
    float gl[LPT][7];
#pragma unroll
    for (int j = 0; j < LPT; ++j)
#pragma unroll
        for (int i = 0; i < 7; ++i) gl[j][i] = 0.f;
    float g_glob[4] = {0.f, 0.f, 0.f, 0.f};    // roughness, specular rgb (per-thread partial sums)
    for (int64_t p = blockIdx.x; p < n; p += gridDim.x) {
        const float nn[3] = {normal[p * 3], normal[p * 3 + 1], normal[p * 3 + 2]};
        const float vv[3] = {view[p * 3], view[p * 3 + 1], view[p * 3 + 2]};
        const PointTerms<T> pt = point_terms<T>(nn, vv, rd);
        // pass 1: forward sums (clamp gates) ; pass 2 fused: keep S, D (+tangents) of this thread's lobes
        T Sv[LPT], Dv[LPT];
        float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < LPT; ++j) {
            const int m = threadIdx.x + j * blockDim.x;
            Sv[j] = T(0.f);
            Dv[j] = T(0.f);
            if (m < M) {
                const float *L = lgt + m * 7;
                V3<T> ax = {seed<5>(L[0], 0), seed<5>(L[1], 1), seed<5>(L[2], 2)};
                pair_terms<T>(nn, pt, ax, seed<5>(L[3], 3), Sv[j], Dv[j]);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float mu = fabsf(L[4 + c]);
                    const float F = s3[c] + (1.f - s3[c]) * pt.fres_p;
                    acc[c] += mu * F * Sv[j].v;
                    acc[3 + c] += mu * Dv[j].v;
                }
            }
        }
        block_sum<6>(acc, scratch);
        float gs[3], gd[3], a_pi[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            a_pi[c] = albedo[p * 3 + c] / PI_F;
            const float go = d_rgb ? d_rgb[p * 3 + c] : 0.f;
            gs[c] = (acc[c] > 0.f) ? go + (d_spec ? d_spec[p * 3 + c] : 0.f) : 0.f;
            gd[c] = (acc[3 + c] * a_pi[c] > 0.f) ? go + (d_diff ? d_diff[p * 3 + c] : 0.f) : 0.f;
        }
        if (threadIdx.x < 3) {
            const int c = threadIdx.x;
            g_albedo[p * 3 + c] = gd[c] * acc[3 + c] / PI_F;
        }
#pragma unroll
        for (int j = 0; j < LPT; ++j) {
            const int m = threadIdx.x + j * blockDim.x;
            if (m < M) {
                const float *L = lgt + m * 7;
                float ws = 0.f, wd = 0.f;     // dL/dS, dL/dD of this pair
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float mu_raw = L[4 + c], mu = fabsf(mu_raw);
                    const float sgn = mu_raw > 0.f ? 1.f : (mu_raw < 0.f ? -1.f : 0.f);
                    const float F = s3[c] + (1.f - s3[c]) * pt.fres_p;
                    ws += gs[c] * mu * F;
                    wd += gd[c] * mu * a_pi[c];
                    gl[j][4 + c] += sgn * (gs[c] * F * Sv[j].v + gd[c] * a_pi[c] * Dv[j].v);
                    g_glob[1 + c] += gs[c] * mu * (1.f - pt.fres_p) * Sv[j].v;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) gl[j][i] += ws * Sv[j].d[i] + wd * Dv[j].d[i];
                g_glob[0] += ws * Sv[j].d[4];
            }
        }
    }
#pragma unroll
    for (int j = 0; j < LPT; ++j) {
        const int m = threadIdx.x + j * blockDim.x;
        if (m < M)
#pragma unroll
            for (int i = 0; i < 7; ++i)
                if (gl[j][i] != 0.f) atomicAdd(&g_lgt[m * 7 + i], gl[j][i]);
    }
    block_sum<4>(g_glob, scratch);
    if (threadIdx.x == 0) {
        atomicAdd(&g_rough[0], g_glob[0]);
        atomicAdd(&g_spec[0], g_glob[1]);
        atomicAdd(&g_spec[1], g_glob[2]);
        atomicAdd(&g_spec[2], g_glob[3]);
    }
}

__global__ __launch_bounds__(SG_THREADS) void env_fwd_kernel(const float *__restrict__ lgt, int M,
                                                             const float *__restrict__ dirs, int64_t n, float eps,
                                                             float *__restrict__ rgb) {
    __shared__ float scratch[4 * 3];
    for (int64_t p = blockIdx.x; p < n; p += gridDim.x) {
        const float d[3] = {dirs[p * 3], dirs[p * 3 + 1], dirs[p * 3 + 2]};
        float acc[3] = {0.f, 0.f, 0.f};
        for (int m = threadIdx.x; m < M; m += blockDim.x) {
            const float *L = lgt + m * 7;
            const float inv = 1.f / (sqrtf(L[0] * L[0] + L[1] * L[1] + L[2] * L[2]) + eps);
            const float dt = d[0] * (L[0] * inv) + d[1] * (L[1] * inv) + d[2] * (L[2] * inv);
            const float e = expf(fabsf(L[3]) * (dt - 1.f));
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[c] += fabsf(L[4 + c]) * e;
        }
        block_sum<3>(acc, scratch);
        if (threadIdx.x < 3) rgb[p * 3 + threadIdx.x] = acc[threadIdx.x];
    }
}

__global__ __launch_bounds__(SG_THREADS) void env_bwd_kernel(const float *__restrict__ lgt, int M,
                                                             const float *__restrict__ dirs, int64_t n, float eps,
                                                             const float *__restrict__ d_rgb,
                                                             float *__restrict__ g_lgt) {
    typedef Dual<4> T;
    for (int m = threadIdx.x; m < M; m += blockDim.x) {
        const float *L = lgt + m * 7;
        float gl[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int64_t p = blockIdx.x; p < n; p += gridDim.x) {
            const float d[3] = {dirs[p * 3], dirs[p * 3 + 1], dirs[p * 3 + 2]};
            V3<T> ax = {seed<4>(L[0], 0), seed<4>(L[1], 1), seed<4>(L[2], 2)};
            const T inv = 1.f / (t_sqrt(dot(ax, ax)) + eps);
            const T dt = (ax.x * d[0] + ax.y * d[1] + ax.z * d[2]) * inv;
            const T e = t_exp(t_abs(seed<4>(L[3], 3)) * (dt - 1.f));
            float w = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float mu_raw = L[4 + c];
                const float g = d_rgb[p * 3 + c];
                w += g * fabsf(mu_raw);
                gl[4 + c] += g * e.v * (mu_raw > 0.f ? 1.f : (mu_raw < 0.f ? -1.f : 0.f));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) gl[i] += w * e.d[i];
        }
#pragma unroll
        for (int i = 0; i < 7; ++i)
            if (gl[i] != 0.f) atomicAdd(&g_lgt[m * 7 + i], gl[i]);
    }
}

int sg_grid(int64_t n) { return (int)(n < 2048 ? (n > 0 ? n : 1) : 2048); }

}  // namespace

extern "C" int nefii_sg_render_forward(const float *lgtSGs, int n_lobes, const float *specular, const float *roughness,
                                       const float *albedo, const float *normal, const float *view, int64_t n,
                                       float *rgb, float *spec_rgb, float *diff_rgb, void *stream) {
    if (!lgtSGs || !specular || !roughness || !albedo || !normal || !view || !rgb || !spec_rgb || !diff_rgb)
        return NEFII_E_ARG;
    if (n <= 0) return 0;
    if (n_lobes <= 0) return NEFII_E_SHAPE;
    hipLaunchKernelGGL(sg_render_fwd_kernel, dim3(sg_grid(n)), dim3(SG_THREADS), 0, (hipStream_t)stream, lgtSGs, n_lobes,
                       specular, roughness, albedo, normal, view, n, rgb, spec_rgb, diff_rgb);
    HIP_CHECK_LAUNCH();
    return 0;
}

extern "C" int nefii_sg_render_backward(const float *lgtSGs, int n_lobes, const float *specular,
                                        const float *roughness, const float *albedo, const float *normal,
                                        const float *view, int64_t n, const float *d_rgb, const float *d_spec,
                                        const float *d_diff, float *g_albedo, float *g_roughness, float *g_specular,
                                        float *g_lgtSGs, void *stream) {
    if (!lgtSGs || !specular || !roughness || !albedo || !normal || !view || !g_albedo || !g_roughness ||
        !g_specular || !g_lgtSGs)
        return NEFII_E_ARG;
    if (n <= 0) return 0;
    if (n_lobes <= 0 || n_lobes > 2 * SG_THREADS) return NEFII_E_SHAPE;
    // few, fat workgroups: every workgroup ends with 7*M + 4 atomics
    int grid = (int)(n < 512 ? n : 512);
    hipLaunchKernelGGL(sg_render_bwd_kernel, dim3(grid), dim3(SG_THREADS), 0, (hipStream_t)stream, lgtSGs, n_lobes,
                       specular, roughness, albedo, normal, view, n, d_rgb, d_spec, d_diff, g_albedo, g_roughness,
                       g_specular, g_lgtSGs);
    HIP_CHECK_LAUNCH();
    return 0;
}

extern "C" int nefii_env_radiance_forward(const float *lgtSGs, int n_lobes, const float *dirs, int64_t n, float eps,
                                          float *rgb, void *stream) {
    if (!lgtSGs || !dirs || !rgb) return NEFII_E_ARG;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(env_fwd_kernel, dim3(sg_grid(n)), dim3(SG_THREADS), 0, (hipStream_t)stream, lgtSGs, n_lobes, dirs,
                       n, eps, rgb);
    HIP_CHECK_LAUNCH();
    return 0;
}

extern "C" int nefii_env_radiance_backward(const float *lgtSGs, int n_lobes, const float *dirs, int64_t n, float eps,
                                           const float *d_rgb, float *g_lgtSGs, void *stream) {
    if (!lgtSGs || !dirs || !d_rgb || !g_lgtSGs) return NEFII_E_ARG;
    if (n <= 0) return 0;
    int grid = (int)(n < 256 ? n : 256);
    hipLaunchKernelGGL(env_bwd_kernel, dim3(grid), dim3(SG_THREADS), 0, (hipStream_t)stream, lgtSGs, n_lobes, dirs, n,
                       eps, d_rgb, g_lgtSGs);
    HIP_CHECK_LAUNCH();
    return 0;
}

// ================================================================================================
// Monte-Carlo direct + near-field indirect shading (conf.conf default: pt_render_indirect_mlp)
//   samplers + pdfs   code/model/path_tracing_render.py:12-271
//   MIS table         code/model/path_tracing_render.py:1290-1325
//   shading sum       code/model/path_tracing_render.py:1406-1476
// ================================================================================================
namespace {

struct F3 {
    float x, y, z;
};
__device__ __forceinline__ F3 f3(float x, float y, float z) { return {x, y, z}; }
__device__ __forceinline__ float dot3(const F3 &a, const F3 &b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ F3 cross3(const F3 &a, const F3 &b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

// rotate local coordinates (z = axis n) to world space (rotate_to_normal, :12-33)
__device__ __forceinline__ F3 to_world(const F3 &l, const F3 &n) {
    const F3 up = n.x > 0.9f ? f3(0.f, 1.f, 0.f) : f3(1.f, 0.f, 0.f);
    F3 t = cross3(up, n);
    const float inv = 1.f / (sqrtf(dot3(t, t)) + TINY);
    t = f3(t.x * inv, t.y * inv, t.z * inv);
    const F3 s = cross3(t, n);
    return {(l.x * t.x + l.y * s.x) + l.z * n.x, (l.x * t.y + l.y * s.y) + l.z * n.y, (l.x * t.z + l.y * s.z) + l.z * n.z};
}
__device__ __forceinline__ F3 polar(float theta, float phi) {
    const float st = sinf(theta);
    return {st * cosf(phi), st * sinf(phi), cosf(theta)};
}
__device__ __forceinline__ float pdf_cos_fn(const F3 &wi, const F3 &n) { return fmaxf(dot3(wi, n), TINY) / PI_F; }
__device__ __forceinline__ float pdf_ggx_fn(const F3 &wi, const F3 &n, const F3 &v, float rough) {
    F3 h = f3(wi.x + v.x, wi.y + v.y, wi.z + v.z);
    const float nh = sqrtf(dot3(h, h));
    h = f3(h.x / nh, h.y / nh, h.z / nh);
    if (isnan(h.x)) h.x = n.x;          // wi = -v: half vector undefined -> normal (:110-111)
    if (isnan(h.y)) h.y = n.y;
    if (isnan(h.z)) h.z = n.z;
    const float c = fmaxf(dot3(h, n), TINY);
    const float r4 = (rough * rough) * (rough * rough);
    const float root = c * c + (1.f - c * c) / r4;
    const float pdf_h = c / (PI_F * r4 * root * root);
    const float hv = fmaxf(dot3(h, v), TINY);
    return pdf_h / (4.f * hv);
}

constexpr int MIS_THREADS = 128;
constexpr int MIS_MAX_LOBES = 256;

// One thread per surface point; the light lobes (axis, |lambda|, energy, c_k) are staged once per block in LDS.
__global__ __launch_bounds__(MIS_THREADS) void mis_sample_kernel(const float *__restrict__ lgt, int M,
                                                                 const float *__restrict__ rough,
                                                                 const float *__restrict__ normal,
                                                                 const float *__restrict__ view,
                                                                 const float *__restrict__ uni, int64_t n,
                                                                 float *__restrict__ wi_out,      // [3][n][3]
                                                                 float *__restrict__ own_pdf,     // [3][n]
                                                                 float *__restrict__ pdf_tab) {   // [3][n][3]
    __shared__ float L[MIS_MAX_LOBES * 6];     // ax(3), lam, energy, c
    for (int m = threadIdx.x; m < M; m += blockDim.x) {
        const float *s = lgt + m * 7;
        const float inv = 1.f / (sqrtf((s[0] * s[0] + s[1] * s[1]) + s[2] * s[2]) + TINY);
        const float lam = fabsf(s[3]);
        L[m * 6 + 0] = s[0] * inv;
        L[m * 6 + 1] = s[1] * inv;
        L[m * 6 + 2] = s[2] * inv;
        L[m * 6 + 3] = lam;
        L[m * 6 + 4] = (fabsf(s[4]) + fabsf(s[5])) + fabsf(s[6]);
        L[m * 6 + 5] = lam / (2.f * PI_F * (1.f - expf(-2.f * lam)));
    }
    __syncthreads();
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const F3 nn = f3(normal[p * 3], normal[p * 3 + 1], normal[p * 3 + 2]);
    const F3 vv = f3(view[p * 3], view[p * 3 + 1], view[p * 3 + 2]);
    const float r = rough[p];
    const float *u = uni + p * 7;
    // --- cosine-weighted (:128-156)
    const float th0 = acosf(sqrtf(1.f - u[0]));
    const F3 w0 = to_world(polar(th0, 2.f * PI_F * u[1]), nn);
    const float p0 = cosf(th0) / PI_F;
    // --- GGX half-vector (:61-103)
    const float th1 = atanf((r * r) * sqrtf(u[2] / (1.f - u[2])));
    const F3 h1 = to_world(polar(th1, 2.f * PI_F * u[3]), nn);
    const float vh = dot3(vv, h1);
    const F3 w1 = f3(2.f * vh * h1.x - vv.x, 2.f * vh * h1.y - vv.y, 2.f * vh * h1.z - vv.z);
    const float p1 = pdf_ggx_fn(w1, nn, vv, r);
    // --- SG mixture (:168-242): lobe k ~ alpha, then a direction around its axis
    float wsum = 0.f;
    for (int m = 0; m < M; ++m) wsum += L[m * 6 + 4] * fmaxf(dot3(nn, f3(L[m * 6], L[m * 6 + 1], L[m * 6 + 2])), TINY);
    int k = -1;
    float cum = 0.f;
    for (int m = 0; m < M; ++m) {
        const float a = (L[m * 6 + 4] * fmaxf(dot3(nn, f3(L[m * 6], L[m * 6 + 1], L[m * 6 + 2])), TINY)) / wsum;
        cum += a;
        float right = cum, left = cum - a;
        if (m == M - 1) right = 1.f;
        if (m == 0) left = 0.f;
        if (k < 0 && u[4] >= left && u[4] < right) k = m;
    }
    if (k < 0) k = 0;
    const F3 axk = f3(L[k * 6], L[k * 6 + 1], L[k * 6 + 2]);
    const float lamk = L[k * 6 + 3], ck = L[k * 6 + 5];
    const float th2 = acosf(1.f / lamk * logf(fmaxf(1.f - lamk * u[5] / (2.f * PI_F * ck), TINY)) + 1.f);
    const F3 w2 = to_world(polar(th2, 2.f * PI_F * u[6]), axk);
    // mixture pdf of all three directions in one sweep over the lobes (:245-271)
    float pm0 = 0.f, pm1 = 0.f, pm2 = 0.f;
    for (int m = 0; m < M; ++m) {
        const F3 ax = f3(L[m * 6], L[m * 6 + 1], L[m * 6 + 2]);
        const float a = (L[m * 6 + 4] * fmaxf(dot3(nn, ax), TINY)) / wsum;
        const float ac = a * L[m * 6 + 5], lam = L[m * 6 + 3];
        pm0 += ac * expf(lam * (dot3(w0, ax) - 1.f));
        pm1 += ac * expf(lam * (dot3(w1, ax) - 1.f));
        pm2 += ac * expf(lam * (dot3(w2, ax) - 1.f));
    }
    const F3 w[3] = {w0, w1, w2};
    const float own[3] = {fmaxf(p0, TINY), fmaxf(p1, TINY), fmaxf(pm2, TINY)};
    const float pmix[3] = {pm0, pm1, pm2};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float *wo = wi_out + ((size_t)i * n + p) * 3;
        wo[0] = w[i].x, wo[1] = w[i].y, wo[2] = w[i].z;
        own_pdf[(size_t)i * n + p] = own[i];
        float *t = pdf_tab + ((size_t)i * n + p) * 3;
        t[0] = i == 0 ? own[0] : pdf_cos_fn(w[i], nn);
        t[1] = i == 1 ? own[1] : pdf_ggx_fn(w[i], nn, vv, r);
        t[2] = i == 2 ? own[2] : pmix[i];
    }
}

// GGX D * G for one sample as a function of roughness (T = float or Dual<1>)
template <class T>
__device__ __forceinline__ T ggx_dg(const T &rough, float nh, float d1, float d2) {
    const T a2 = rough * rough;
    const T a4 = a2 * a2;
    const T root = nh * nh + (1.f - nh * nh) / a4;
    const T D = 1.f / (PI_F * a4 * root * root);
    const T k = (rough + 1.f) * (rough + 1.f) / 8.f;
    const T g = (d1 / (d1 * (1.f - k) + k + TINY)) * (d2 / (d2 * (1.f - k) + k + TINY));
    return D * g;
}

struct McGeom {      // per (point, sample) constants
    float nh, P, d1, d2, den, K;
};

__device__ __forceinline__ McGeom mc_geom(const F3 &nn, const F3 &vv, const F3 &wi, float own, const float *tab) {
    McGeom g;
    F3 h = f3(wi.x + vv.x, wi.y + vv.y, wi.z + vv.z);
    const float inv = 1.f / (sqrtf(dot3(h, h)) + TINY);
    h = f3(h.x * inv, h.y * inv, h.z * inv);
    g.nh = fmaxf(dot3(nn, h), 0.f);
    const float vh = fmaxf(dot3(vv, h), 0.f);
    g.P = exp2f(-(5.55473f * vh + 6.8316f) * vh);
    g.d1 = fmaxf(dot3(vv, nn), 0.f);
    g.d2 = fmaxf(dot3(wi, nn), 0.f);
    g.den = 4.f * g.d1 * g.d2 + TINY;
    const float den_w = fmaxf((tab[0] * tab[0] + tab[1] * tab[1]) + tab[2] * tab[2], TINY);
    const float weight = own * own / den_w;                      // power heuristic (:390-401)
    const float cosn = fmaxf(dot3(wi, nn), 0.f);
    g.K = weight * cosn / own;
    return g;
}

__global__ __launch_bounds__(256) void mc_shade_fwd_kernel(const float *__restrict__ spec, const float *__restrict__ rough,
                                                           const float *__restrict__ albedo,
                                                           const float *__restrict__ normal,
                                                           const float *__restrict__ view, const float *__restrict__ wi,
                                                           const float *__restrict__ own_pdf,
                                                           const float *__restrict__ pdf_tab,
                                                           const float *__restrict__ light,      // [3][n][3]
                                                           const float *__restrict__ vis,        // [3][n]
                                                           const float *__restrict__ indirect,   // [3][n][3]
                                                           int64_t n, float *__restrict__ rgb, float *__restrict__ srgb,
                                                           float *__restrict__ drgb) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const F3 nn = f3(normal[p * 3], normal[p * 3 + 1], normal[p * 3 + 2]);
    const F3 vv = f3(view[p * 3], view[p * 3 + 1], view[p * 3 + 2]);
    const float r = rough[p];
    float s_acc[3] = {0.f, 0.f, 0.f}, d_acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const size_t q = (size_t)i * n + p;
        const F3 w = f3(wi[q * 3], wi[q * 3 + 1], wi[q * 3 + 2]);
        const McGeom g = mc_geom(nn, vv, w, own_pdf[q], pdf_tab + q * 3);
        const float dg = ggx_dg<float>(r, g.nh, g.d1, g.d2);
        const float v = vis[q];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float F = spec[c] + (1.f - spec[c]) * g.P;
            const float fs = F * dg / g.den;
            const float Lc = light[q * 3 + c] * v + (1.f - v) * indirect[q * 3 + c];
            s_acc[c] += fmaxf(g.K * Lc * fs, 0.f);
            d_acc[c] += fmaxf(g.K * Lc * (albedo[p * 3 + c] / PI_F), 0.f);
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        srgb[p * 3 + c] = s_acc[c];
        drgb[p * 3 + c] = d_acc[c];
        rgb[p * 3 + c] = s_acc[c] + d_acc[c];
    }
}

__global__ __launch_bounds__(256) void mc_shade_bwd_kernel(
    const float *__restrict__ spec, const float *__restrict__ rough, const float *__restrict__ albedo,
    const float *__restrict__ normal, const float *__restrict__ view, const float *__restrict__ wi,
    const float *__restrict__ own_pdf, const float *__restrict__ pdf_tab, const float *__restrict__ light,
    const float *__restrict__ vis, const float *__restrict__ indirect, int64_t n, const float *__restrict__ d_rgb,
    const float *__restrict__ d_s, const float *__restrict__ d_d, float *__restrict__ g_light,
    float *__restrict__ g_ind, float *__restrict__ g_alb, float *__restrict__ g_rough, float *__restrict__ g_spec) {
    __shared__ float scratch[4 * 3];
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float gsp[3] = {0.f, 0.f, 0.f};
    if (p < n) {
        const F3 nn = f3(normal[p * 3], normal[p * 3 + 1], normal[p * 3 + 2]);
        const F3 vv = f3(view[p * 3], view[p * 3 + 1], view[p * 3 + 2]);
        const Dual<1> r = seed<1>(rough[p], 0);
        float ga[3] = {0.f, 0.f, 0.f};
        float gr = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const size_t q = (size_t)i * n + p;
            const F3 w = f3(wi[q * 3], wi[q * 3 + 1], wi[q * 3 + 2]);
            const McGeom g = mc_geom(nn, vv, w, own_pdf[q], pdf_tab + q * 3);
            const Dual<1> dg = ggx_dg<Dual<1>>(r, g.nh, g.d1, g.d2);
            const float v = vis[q];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float go = d_rgb ? d_rgb[p * 3 + c] : 0.f;
                const float gs_up = go + (d_s ? d_s[p * 3 + c] : 0.f);
                const float gd_up = go + (d_d ? d_d[p * 3 + c] : 0.f);
                const float F = spec[c] + (1.f - spec[c]) * g.P;
                const float fs = F * dg.v / g.den;
                const float Lc = light[q * 3 + c] * v + (1.f - v) * indirect[q * 3 + c];
                const float a_pi = albedo[p * 3 + c] / PI_F;
                const float gs = (g.K * Lc * fs > 0.f) ? gs_up : 0.f;
                const float gd = (g.K * Lc * a_pi > 0.f) ? gd_up : 0.f;
                const float dL = gs * g.K * fs + gd * g.K * a_pi;
                g_light[q * 3 + c] = dL * v;
                g_ind[q * 3 + c] = dL * (1.f - v);
                ga[c] += gd * g.K * Lc / PI_F;
                const float dfs = gs * g.K * Lc;
                gr += dfs * F * dg.d[0] / g.den;
                gsp[c] += dfs * (1.f - g.P) * dg.v / g.den;
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) g_alb[p * 3 + c] = ga[c];
        g_rough[p] = gr;
    }
    if (g_spec) {
        block_sum<3>(gsp, scratch);
        if (threadIdx.x == 0) {
            atomicAdd(&g_spec[0], gsp[0]);
            atomicAdd(&g_spec[1], gsp[1]);
            atomicAdd(&g_spec[2], gsp[2]);
        }
    }
}

}  // namespace

extern "C" int nefii_mis_sample(const float *lgtSGs, int n_lobes, const float *roughness, const float *normal,
                                const float *view, const float *uniforms, int64_t n, float *wi, float *own_pdf,
                                float *pdf_table, void *stream) {
    if (!lgtSGs || !roughness || !normal || !view || !uniforms || !wi || !own_pdf || !pdf_table) return NEFII_E_ARG;
    if (n <= 0) return 0;
    if (n_lobes <= 0 || n_lobes > MIS_MAX_LOBES) return NEFII_E_SHAPE;
    hipLaunchKernelGGL(mis_sample_kernel, dim3((unsigned)((n + MIS_THREADS - 1) / MIS_THREADS)), dim3(MIS_THREADS), 0,
                       (hipStream_t)stream, lgtSGs, n_lobes, roughness, normal, view, uniforms, n, wi, own_pdf,
                       pdf_table);
    HIP_CHECK_LAUNCH();
    return 0;
}

extern "C" int nefii_mc_shade_forward(const float *specular, const float *roughness, const float *albedo,
                                      const float *normal, const float *view, const float *wi, const float *own_pdf,
                                      const float *pdf_table, const float *light, const float *visibility,
                                      const float *indirect, int64_t n, float *rgb, float *spec_rgb, float *diff_rgb,
                                      void *stream) {
    if (!specular || !roughness || !albedo || !normal || !view || !wi || !own_pdf || !pdf_table || !light ||
        !visibility || !indirect || !rgb || !spec_rgb || !diff_rgb)
        return NEFII_E_ARG;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(mc_shade_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       specular, roughness, albedo, normal, view, wi, own_pdf, pdf_table, light, visibility, indirect, n,
                       rgb, spec_rgb, diff_rgb);
    HIP_CHECK_LAUNCH();
    return 0;
}

extern "C" int nefii_mc_shade_backward(const float *specular, const float *roughness, const float *albedo,
                                       const float *normal, const float *view, const float *wi, const float *own_pdf,
                                       const float *pdf_table, const float *light, const float *visibility,
                                       const float *indirect, int64_t n, const float *d_rgb, const float *d_spec,
                                       const float *d_diff, float *g_light, float *g_indirect, float *g_albedo,
                                       float *g_roughness, float *g_specular, void *stream) {
    if (!specular || !roughness || !albedo || !normal || !view || !wi || !own_pdf || !pdf_table || !light ||
        !visibility || !indirect || !g_light || !g_indirect || !g_albedo || !g_roughness)
        return NEFII_E_ARG;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(mc_shade_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       specular, roughness, albedo, normal, view, wi, own_pdf, pdf_table, light, visibility, indirect, n,
                       d_rgb, d_spec, d_diff, g_light, g_indirect, g_albedo, g_roughness, g_specular);
    HIP_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// IDRLoss (code/model/loss.py:278-320), value and gradient in one launch.  The loss runs on num_pixels x 3 floats:
// as torch ops it is ~45 launches of a few microseconds each, behind one another on the step's critical path.
// ------------------------------------------------------------------------------------------------
namespace {

__device__ __forceinline__ float img_err(float d, int kind) {       // loss.py:144-161: L1 / L2 / SmoothL1(beta=1)
    const float ad = fabsf(d);
    if (kind == 0) return ad;
    if (kind == 1) return d * d;
    return ad < 1.f ? 0.5f * d * d : ad - 0.5f;
}
__device__ __forceinline__ float img_err_grad(float d, int kind) {
    const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
    if (kind == 0) return sg;
    if (kind == 1) return 2.f * d;
    return fabsf(d) < 1.f ? d : sg;
}

constexpr int LOSS_T = 1024, LOSS_Q = 8;       // threads of the single block; quantities reduced

// quantities: 0 idr err sum, 1 sg err sum, 2 hit count, 3 mask bce sum, 4 background err sum, 5 background count,
// 6 normal-variance sum, 7 patch count
__global__ __launch_bounds__(LOSS_T) void idr_loss_kernel(nefii_loss_params p, const float *__restrict__ idr_rgb,
                                                          const float *__restrict__ sg_rgb,
                                                          const float *__restrict__ gt, const uint8_t *__restrict__ net,
                                                          const uint8_t *__restrict__ obj,
                                                          const float *__restrict__ sdf,
                                                          const float *__restrict__ normals, int64_t n,
                                                          float *__restrict__ losses, float *__restrict__ d_idr,
                                                          float *__restrict__ d_sg) {
    __shared__ double red[LOSS_T];
    __shared__ double tot[LOSS_Q];
    const int tid = threadIdx.x;
    double acc[LOSS_Q];
#pragma unroll
    for (int q = 0; q < LOSS_Q; ++q) acc[q] = 0.0;
    for (int64_t i = tid; i < n; i += LOSS_T) {
        const bool nm = net[i] != 0, om = obj[i] != 0;
        if (nm && om) {
            for (int c = 0; c < 3; ++c) {
                const float g = gt[i * 3 + c];
                acc[0] += img_err(idr_rgb[i * 3 + c] - g, p.loss_type);
                acc[1] += img_err(sg_rgb[i * 3 + c] - g, p.loss_type);
            }
            acc[2] += 1.0;
        } else {        // get_mask_loss :186-196: BCE-with-logits of -alpha*sdf against the object mask, on the other rays
            const float x = -p.alpha * sdf[i], t = om ? 1.f : 0.f;
            acc[3] += fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
        }
        if (!nm && !om && p.background_rgb_weight > 0.f) {
            for (int c = 0; c < 3; ++c) acc[4] += img_err(sg_rgb[i * 3 + c] - gt[i * 3 + c], p.env_loss_type);
            acc[5] += 1.0;
        }
    }
    if (p.r_patch >= 1 && p.normalsmooth_weight != 0.f) {      // get_normalsmooth_loss :198-207
        const int k = 4 * p.r_patch * p.r_patch;
        for (int64_t pi = tid; pi < n / k; pi += LOSS_T) {
            bool all = true;
            for (int j = 0; j < k; ++j) all = all && net[pi * k + j] && obj[pi * k + j];
            if (!all) continue;
            for (int c = 0; c < 3; ++c) {
                float mean = 0.f;
                for (int j = 0; j < k; ++j) mean += normals[(pi * k + j) * 3 + c];
                mean /= k;
                float var = 0.f;
                for (int j = 0; j < k; ++j) {
                    const float d = normals[(pi * k + j) * 3 + c] - mean;
                    var += d * d;
                }
                acc[6] += var / (k - 1);
            }
            acc[7] += 1.0;
        }
    }
#pragma unroll
    for (int q = 0; q < LOSS_Q; ++q) {          // plain shared-memory tree per quantity (8 x 10 steps on 8 KB)
        red[tid] = acc[q];
        __syncthreads();
        for (int half = LOSS_T / 2; half > 0; half >>= 1) {
            if (tid < half) red[tid] += red[tid + half];
            __syncthreads();
        }
        if (tid == 0) tot[q] = red[0];
        __syncthreads();
    }
    const float den = fmaxf((float)tot[2] * 3.f, 1.f), bden = fmaxf((float)tot[5] * 3.f, 1.f);
    if (tid == 0) {
        const float idr_l = (float)tot[0] / den, sg_l = (float)tot[1] / den;
        const float mask_l = (1.f / p.alpha) * (float)tot[3] / (float)n;
        const float ns_l = (p.r_patch >= 1 && p.normalsmooth_weight != 0.f) ? (float)tot[6] / fmaxf((float)tot[7] * 3.f, 1.f) : 0.f;
        const float bg_l = p.background_rgb_weight > 0.f ? (float)tot[4] / bden : 0.f;
        losses[0] = p.idr_rgb_weight * idr_l + p.sg_rgb_weight * sg_l + p.mask_weight * mask_l +
                    p.normalsmooth_weight * ns_l + p.background_rgb_weight * bg_l;
        losses[1] = idr_l;
        losses[2] = sg_l;
        losses[3] = mask_l;
        losses[4] = ns_l;
        losses[5] = bg_l;
    }
    if (!d_idr && !d_sg) return;
    for (int64_t i = tid; i < n; i += LOSS_T) {
        const bool nm = net[i] != 0, om = obj[i] != 0;
        for (int c = 0; c < 3; ++c) {
            const float g = gt[i * 3 + c];
            float gi = 0.f, gs = 0.f;
            if (nm && om) {
                gi = p.idr_rgb_weight * img_err_grad(idr_rgb[i * 3 + c] - g, p.loss_type) / den;
                gs = p.sg_rgb_weight * img_err_grad(sg_rgb[i * 3 + c] - g, p.loss_type) / den;
            } else if (!nm && !om && p.background_rgb_weight > 0.f) {
                gs = p.background_rgb_weight * img_err_grad(sg_rgb[i * 3 + c] - g, p.env_loss_type) / bden;
            }
            if (d_idr) d_idr[i * 3 + c] = gi;
            if (d_sg) d_sg[i * 3 + c] = gs;
        }
    }
}

}  // namespace

extern "C" int nefii_idr_loss(const nefii_loss_params *h_params, const float *idr_rgb, const float *sg_rgb,
                              const float *rgb_gt, const uint8_t *network_object_mask, const uint8_t *object_mask,
                              const float *sdf_output, const float *normals, int64_t n, float *losses, float *d_idr_rgb,
                              float *d_sg_rgb, void *stream) {
    if (!h_params || !idr_rgb || !sg_rgb || !rgb_gt || !network_object_mask || !object_mask || !sdf_output || !losses)
        return NEFII_E_ARG;
    if (n <= 0) return NEFII_E_SHAPE;
    if (h_params->loss_type < 0 || h_params->loss_type > 2 || h_params->env_loss_type < 0 || h_params->env_loss_type > 1)
        return NEFII_E_ARG;
    if (h_params->r_patch >= 1 && h_params->normalsmooth_weight != 0.f && !normals) return NEFII_E_ARG;
    hipLaunchKernelGGL(idr_loss_kernel, dim3(1), dim3(LOSS_T), 0, (hipStream_t)stream, *h_params, idr_rgb, sg_rgb, rgb_gt,
                       network_object_mask, object_mask, sdf_output, normals, n, losses, d_idr_rgb, d_sg_rgb);
    HIP_CHECK_LAUNCH();
    return 0;
}


// ---- output assembly (include/nefii_amd.h: nefii_assemble_rows / nefii_gather_rows) ---------------------------------
namespace {
struct RowBlocks {
    nefii_row_block b[NEFII_MAX_ROW_BLOCKS];
};

// blockIdx.y = block; grid.x strides over the elements of the block's destination
__global__ __launch_bounds__(256) void fill_rows_kernel(RowBlocks blocks, int64_t rows) {
    const nefii_row_block B = blocks.b[blockIdx.y];
    if (!B.dst) return;
    const int64_t total = rows * B.cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
        B.dst[i] = B.fill;
}

__global__ __launch_bounds__(256) void scatter_rows_kernel(RowBlocks blocks, const int64_t *__restrict__ where, int64_t n_src,
                                                           int64_t rows) {
    const nefii_row_block B = blocks.b[blockIdx.y];
    if (!B.dst || !B.src) return;
    const int64_t total = n_src * B.cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / B.cols, c = i - r * B.cols, w = where[r];
        if (w >= 0 && w < rows) B.dst[w * B.cols + c] = B.src[r * B.src_row_stride + c];
    }
}

__global__ __launch_bounds__(256) void gather_rows_kernel(RowBlocks blocks, const int64_t *__restrict__ where, int64_t n_src,
                                                          int64_t rows) {
    const nefii_row_block B = blocks.b[blockIdx.y];
    if (!B.dst || !B.src) return;
    const int64_t total = n_src * B.cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / B.cols, c = i - r * B.cols, w = where[r];
        B.dst[i] = (w >= 0 && w < rows) ? B.src[w * B.cols + c] : 0.f;
    }
}

int check_blocks(const nefii_row_block *h, int n_blocks, RowBlocks &out, int &max_cols) {
    if (!h || n_blocks < 1 || n_blocks > NEFII_MAX_ROW_BLOCKS) return NEFII_E_ARG;
    max_cols = 1;
    for (int i = 0; i < NEFII_MAX_ROW_BLOCKS; ++i) {
        if (i < n_blocks) {
            if (h[i].cols < 1 || h[i].src_row_stride < 0) return NEFII_E_SHAPE;
            out.b[i] = h[i];
            max_cols = h[i].cols > max_cols ? h[i].cols : max_cols;
        } else {
            out.b[i] = nefii_row_block{nullptr, nullptr, 1, 0, 0.f, 0};
        }
    }
    return 0;
}

unsigned grid_for(int64_t elems) {
    const int64_t g = (elems + 255) / 256;
    return (unsigned)(g < 1 ? 1 : (g > 1024 ? 1024 : g));
}
}  // namespace

extern "C" int nefii_assemble_rows(const nefii_row_block *h_blocks, int n_blocks, const int64_t *where, int64_t n_src,
                                   int64_t rows, void *stream) {
    RowBlocks blocks;
    int max_cols;
    const int rc = check_blocks(h_blocks, n_blocks, blocks, max_cols);
    if (rc) return rc;
    if (rows < 1 || n_src < 0 || (n_src > 0 && !where)) return NEFII_E_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(fill_rows_kernel, dim3(grid_for(rows * max_cols), n_blocks), dim3(256), 0, st, blocks, rows);
    if (n_src > 0)
        hipLaunchKernelGGL(scatter_rows_kernel, dim3(grid_for(n_src * max_cols), n_blocks), dim3(256), 0, st, blocks, where,
                           n_src, rows);
    HIP_CHECK_LAUNCH();
    return 0;
}

// EnvmapMaterialNetwork.forward's scalar head for GLOBAL roughness / specular parameters (physg.conf; sg_envmap_material.py:
// 381-414): roughness = (1 - 0.089) sigmoid(r) + 0.089, specular = 0.16 sigmoid(s)^2 (white_specular: one value for the three
// channels), each replaced by 0.5 (before the remap) while its warm-up flag is set - six eager ops on one- to three-element
// tensors forward and eight backward, in the serial chain of a launch-latency-bound step.  One thread each way.
__global__ void material_head_global_fwd_kernel(const float *__restrict__ rough_param, const float *__restrict__ spec_param,
                                                int n_spec, int fake_rough, int fake_spec, float *__restrict__ rough_out,
                                                float *__restrict__ spec_out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float r = 1.f / (1.f + expf(-rough_param[0]));
    rough_out[0] = fake_rough ? 0.5f : (1.f - 0.089f) * r + 0.089f;
    for (int c = 0; c < 3; ++c) {
        const float sg = 1.f / (1.f + expf(-spec_param[n_spec == 3 ? c : 0]));
        const float v = fake_spec ? 0.5f : sg;
        spec_out[c] = 0.16f * v * v;
    }
}
__global__ void material_head_global_bwd_kernel(const float *__restrict__ rough_param, const float *__restrict__ spec_param,
                                                int n_spec, int fake_rough, int fake_spec, const float *__restrict__ d_rough,
                                                const float *__restrict__ d_spec, float *__restrict__ g_rough_param,
                                                float *__restrict__ g_spec_param) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float r = 1.f / (1.f + expf(-rough_param[0]));
    g_rough_param[0] = (fake_rough || !d_rough) ? 0.f : d_rough[0] * (1.f - 0.089f) * r * (1.f - r);
    for (int c = 0; c < n_spec; ++c) g_spec_param[c] = 0.f;
    if (!fake_spec && d_spec)
        for (int c = 0; c < 3; ++c) {
            const int k = n_spec == 3 ? c : 0;
            const float sg = 1.f / (1.f + expf(-spec_param[k]));
            g_spec_param[k] += d_spec[c] * 0.32f * sg * sg * (1.f - sg);
        }
}

extern "C" int nefii_material_head_global(const float *rough_param, const float *spec_param, int n_spec, int fake_rough,
                                          int fake_spec, float *rough_out, float *spec_out, void *stream) {
    if (!rough_param || !spec_param || !rough_out || !spec_out || (n_spec != 1 && n_spec != 3)) return NEFII_E_ARG;
    hipLaunchKernelGGL(material_head_global_fwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, rough_param, spec_param, n_spec,
                       fake_rough, fake_spec, rough_out, spec_out);
    HIP_CHECK_LAUNCH();
    return 0;
}
extern "C" int nefii_material_head_global_backward(const float *rough_param, const float *spec_param, int n_spec, int fake_rough,
                                                   int fake_spec, const float *d_rough, const float *d_spec, float *g_rough_param,
                                                   float *g_spec_param, void *stream) {
    if (!rough_param || !spec_param || !g_rough_param || !g_spec_param || (n_spec != 1 && n_spec != 3)) return NEFII_E_ARG;
    hipLaunchKernelGGL(material_head_global_bwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, rough_param, spec_param, n_spec,
                       fake_rough, fake_spec, d_rough, d_spec, g_rough_param, g_spec_param);
    HIP_CHECK_LAUNCH();
    return 0;
}

// The inputs of get_rbg_value for the compacted hit rays (implicit_differentiable_renderer.py:358-364,533-545) in ONE launch:
// the reference masks points / ray_dirs with the hit mask, negates the directions, evaluates the SDF gradient and divides
// both by (their norm + 1e-6) - ten eager ops on [n_hit, 3] tensors.  One thread per hit ray; the feature rows (feat_src:
// [rows, feat_cols], or NULL) are copied by the same grid (feat_cols / 4 more threads per ray when feat_cols > 0).
__global__ __launch_bounds__(256) void prepare_hits_kernel(const float *__restrict__ points, const float *__restrict__ ray_dirs,
                                                           const float *__restrict__ grad, const float *__restrict__ feat_src,
                                                           int feat_cols, const int64_t *__restrict__ where, int64_t n,
                                                           int64_t rows, float *__restrict__ pts_out, float *__restrict__ view_out,
                                                           float *__restrict__ nrm_out, float *__restrict__ feat_out) {
    const int per = 1 + (feat_cols + 3) / 4;
    const int64_t total = n * per;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / per;
        const int part = (int)(i - r * per);
        int64_t w = where[r];
        w = w < 0 ? 0 : (w >= rows ? rows - 1 : w);
        if (part == 0) {
            const float px = points[w * 3], py = points[w * 3 + 1], pz = points[w * 3 + 2];
            pts_out[r * 3] = px, pts_out[r * 3 + 1] = py, pts_out[r * 3 + 2] = pz;
            const float vx = -ray_dirs[w * 3], vy = -ray_dirs[w * 3 + 1], vz = -ray_dirs[w * 3 + 2];
            const float vn = __fadd_rn(sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(vx, vx), __fmul_rn(vy, vy)), __fmul_rn(vz, vz))), 1e-6f);
            view_out[r * 3] = vx / vn, view_out[r * 3 + 1] = vy / vn, view_out[r * 3 + 2] = vz / vn;
            const float gx = grad[w * 3], gy = grad[w * 3 + 1], gz = grad[w * 3 + 2];
            const float gn = __fadd_rn(sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(gx, gx), __fmul_rn(gy, gy)), __fmul_rn(gz, gz))), 1e-6f);
            nrm_out[r * 3] = gx / gn, nrm_out[r * 3 + 1] = gy / gn, nrm_out[r * 3 + 2] = gz / gn;
        } else {
            const int c0 = 4 * (part - 1);
            for (int c = c0; c < c0 + 4 && c < feat_cols; ++c) feat_out[r * feat_cols + c] = feat_src[w * feat_cols + c];
        }
    }
}

extern "C" int nefii_prepare_hits(const float *points, const float *ray_dirs, const float *grad, const float *feat_src,
                                  int feat_cols, const int64_t *where, int64_t n, int64_t rows, float *pts_out, float *view_out,
                                  float *nrm_out, float *feat_out, void *stream) {
    if (!points || !ray_dirs || !grad || !pts_out || !view_out || !nrm_out || feat_cols < 0) return NEFII_E_ARG;
    if ((feat_cols > 0) != (feat_src != nullptr) || (feat_cols > 0) != (feat_out != nullptr)) return NEFII_E_ARG;
    if (rows < 1 || n < 0 || (n > 0 && !where)) return NEFII_E_SHAPE;
    if (n == 0) return 0;
    hipLaunchKernelGGL(prepare_hits_kernel, dim3(grid_for(n * (1 + (feat_cols + 3) / 4))), dim3(256), 0, (hipStream_t)stream, points,
                       ray_dirs, grad, feat_src, feat_cols, where, n, rows, pts_out, view_out, nrm_out, feat_out);
    HIP_CHECK_LAUNCH();
    return 0;
}

extern "C" int nefii_gather_rows(const nefii_row_block *h_blocks, int n_blocks, const int64_t *where, int64_t n_src,
                                 int64_t rows, void *stream) {
    RowBlocks blocks;
    int max_cols;
    const int rc = check_blocks(h_blocks, n_blocks, blocks, max_cols);
    if (rc) return rc;
    if (rows < 1 || n_src < 1 || !where) return NEFII_E_SHAPE;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for(n_src * max_cols), n_blocks), dim3(256), 0, (hipStream_t)stream,
                       blocks, where, n_src, rows);
    HIP_CHECK_LAUNCH();
    return 0;
}
