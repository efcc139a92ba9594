// HBM-bound kernels of the DASR GAN step (gfx950): InstanceNorm+LeakyReLU fwd/bwd, BCE-with-logits,
// Haar DWT fwd/adjoint, depthwise gaussian / box low-pass (+ frequency split), 2x2 max-pool fwd/bwd,
// L1 between feature maps, VGG input normalisation, bilinear x4 of the domain-distance map.
// All tensors are NC16HW16 (see dasr_hip.h); one 16-channel pixel = 64 B (f32) / 32 B (bf16).
#include "common.h"

namespace {

inline unsigned nblk(long long total, int bs = 256) { return (unsigned)((total + bs - 1) / bs); }

__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// ---------------------------------------------------------------------------------------------------
// InstanceNorm2d(affine=False, eps) + LeakyReLU, one workgroup per (n, 16-channel plane).
// thread t: channel quad q = t & 3, pixel lane = t >> 2 (64 pixel lanes); sums reduced per quad.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ f32x4 quad_reduce(f32x4 v, f32x4* red) {
    // reduce over the 64 threads that share (threadIdx.x & 3): xor-shuffle over lane bits 2..5, then LDS over 4 waves
#pragma unroll
    for (int o = 4; o < 64; o <<= 1)
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += __shfl_xor(v[j], o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) < 4) red[(threadIdx.x >> 6) * 4 + (threadIdx.x & 3)] = v;
    __syncthreads();
    const int q = threadIdx.x & 3;
    return red[q] + red[4 + q] + red[8 + q] + red[12 + q];
}

__global__ __launch_bounds__(256) void inorm_lrelu_fwd_kernel(dasr_tensor x, int C, int H, int W, float eps, float slope,
                                                              dasr_tensor y, float* __restrict__ stats /* [N][Cpad][2] mean, rstd */) {
    __shared__ f32x4 red[16];
    const int ncb = (C + 15) >> 4;
    const int n = blockIdx.x / ncb, cb = blockIdx.x - n * ncb;
    const int q = threadIdx.x & 3, pl = threadIdx.x >> 2;
    const float* xp = (const float*)x.p + (size_t)n * x.n_stride + (size_t)cb * x.cb_stride + q * 4;
    float* yp = (float*)y.p + (size_t)n * y.n_stride + (size_t)cb * y.cb_stride + q * 4;
    const int HW = H * W;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int p = pl; p < HW; p += 64) s += *(const f32x4*)(xp + (size_t)p * 16);
    const f32x4 mean = quad_reduce(s, red) * (1.f / HW);
    f32x4 s2 = {0.f, 0.f, 0.f, 0.f};
    for (int p = pl; p < HW; p += 64) {
        const f32x4 d = *(const f32x4*)(xp + (size_t)p * 16) - mean;
        s2 += d * d;
    }
    const f32x4 var = quad_reduce(s2, red) * (1.f / HW);  // biased variance (nn.InstanceNorm2d)
    f32x4 rstd;
#pragma unroll
    for (int j = 0; j < 4; ++j) rstd[j] = 1.f / sqrtf(var[j] + eps);
    for (int p = pl; p < HW; p += 64) {
        f32x4 v = (*(const f32x4*)(xp + (size_t)p * 16) - mean) * rstd;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : v[j] * slope;
        *(f32x4*)(yp + (size_t)p * 16) = v;
    }
    if (pl == 0 && stats) {
        float* st = stats + ((size_t)n * ncb * 16 + cb * 16 + q * 4) * 2;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            st[2 * j] = mean[j];
            st[2 * j + 1] = rstd[j];
        }
    }
}

// backward: a = lrelu(xhat) is the saved forward output; ga = dL/da.  gy = ga * lrelu'(a);
// gx = rstd * (gy - mean(gy) - xhat * mean(gy * xhat)),  xhat = a > 0 ? a : a / slope.
__global__ __launch_bounds__(256) void inorm_lrelu_bwd_kernel(dasr_tensor a, dasr_tensor ga, int C, int H, int W, float slope,
                                                              const float* __restrict__ stats, dasr_tensor gx) {
    __shared__ f32x4 red[16];
    const int ncb = (C + 15) >> 4;
    const int n = blockIdx.x / ncb, cb = blockIdx.x - n * ncb;
    const int q = threadIdx.x & 3, pl = threadIdx.x >> 2;
    const float* ap = (const float*)a.p + (size_t)n * a.n_stride + (size_t)cb * a.cb_stride + q * 4;
    const float* gp = (const float*)ga.p + (size_t)n * ga.n_stride + (size_t)cb * ga.cb_stride + q * 4;
    float* op = (float*)gx.p + (size_t)n * gx.n_stride + (size_t)cb * gx.cb_stride + q * 4;
    const int HW = H * W;
    const float inv_slope = 1.f / slope;
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
    for (int p = pl; p < HW; p += 64) {
        const f32x4 av = *(const f32x4*)(ap + (size_t)p * 16);
        f32x4 g = *(const f32x4*)(gp + (size_t)p * 16);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool pos = av[j] > 0.f;
            const float xh = pos ? av[j] : av[j] * inv_slope;
            const float gy = pos ? g[j] : g[j] * slope;
            s1[j] += gy;
            s2[j] += gy * xh;
        }
    }
    const f32x4 m1 = quad_reduce(s1, red) * (1.f / HW);
    const f32x4 m2 = quad_reduce(s2, red) * (1.f / HW);
    f32x4 rstd;
    const float* st = stats + ((size_t)n * ncb * 16 + cb * 16 + q * 4) * 2;
#pragma unroll
    for (int j = 0; j < 4; ++j) rstd[j] = st[2 * j + 1];
    for (int p = pl; p < HW; p += 64) {
        const f32x4 av = *(const f32x4*)(ap + (size_t)p * 16);
        const f32x4 g = *(const f32x4*)(gp + (size_t)p * 16);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool pos = av[j] > 0.f;
            const float xh = pos ? av[j] : av[j] * inv_slope;
            const float gy = pos ? g[j] : g[j] * slope;
            o[j] = rstd[j] * (gy - m1[j] - xh * m2[j]);
        }
        *(f32x4*)(op + (size_t)p * 16) = o;
    }
}

// ---------------------------------------------------------------------------------------------------
// Second-order pieces of InstanceNorm2d + LeakyReLU for the DSN's `--wgan` gradient penalty (codes/DSN/train.py:231-236:
// torch.autograd.grad(mean D(sample), sample, create_graph=True) and its backward).  With xhat = (z - mu) r, r = (var + eps)^-1/2 (from the saved
// output a = lrelu(xhat) and the saved r), the Jacobian of InstanceNorm is SYMMETRIC, J t = r (t - mean t - xhat mean(xhat t)):
//   * inorm_lrelu_jvp: forward-mode tangent through IN + LeakyReLU:  out = lrelu'(a) * J t          (the backward kernel above computes J (lrelu'(a) g))
//   * inorm_second:    the adjoint of  z -> J(z) t  for fixed tangent t and upstream w = lrelu'(a) * ga (ga = adjoint of the tangent output):
//       out (+)= -r^2 [ xhat (mean(w t) - mean w mean t - 3 mean(w xhat) mean(xhat t)) + mean(xhat t) (w - mean w) + mean(w xhat) (t - mean t) ]
//     (derivation and numeric check against torch's double backward: tests/test_gpu_wgan.py)
// One workgroup per (n, 16-channel plane) like the kernels above; fixed-order reductions.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void inorm_lrelu_jvp_kernel(dasr_tensor a, dasr_tensor t, int C, int H, int W, float slope,
                                                              const float* __restrict__ stats, dasr_tensor out) {
    __shared__ f32x4 red[16];
    const int ncb = (C + 15) >> 4;
    const int n = blockIdx.x / ncb, cb = blockIdx.x - n * ncb;
    const int q = threadIdx.x & 3, pl = threadIdx.x >> 2;
    const float* ap = (const float*)a.p + (size_t)n * a.n_stride + (size_t)cb * a.cb_stride + q * 4;
    const float* tp = (const float*)t.p + (size_t)n * t.n_stride + (size_t)cb * t.cb_stride + q * 4;
    float* op = (float*)out.p + (size_t)n * out.n_stride + (size_t)cb * out.cb_stride + q * 4;
    const int HW = H * W;
    const float inv_slope = 1.f / slope;
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
    for (int p = pl; p < HW; p += 64) {
        const f32x4 av = *(const f32x4*)(ap + (size_t)p * 16);
        const f32x4 tv = *(const f32x4*)(tp + (size_t)p * 16);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float xh = av[j] > 0.f ? av[j] : av[j] * inv_slope;
            s1[j] += tv[j];
            s2[j] += tv[j] * xh;
        }
    }
    const f32x4 m1 = quad_reduce(s1, red) * (1.f / HW);
    const f32x4 m2 = quad_reduce(s2, red) * (1.f / HW);
    f32x4 rstd;
    const float* st = stats + ((size_t)n * ncb * 16 + cb * 16 + q * 4) * 2;
#pragma unroll
    for (int j = 0; j < 4; ++j) rstd[j] = st[2 * j + 1];
    for (int p = pl; p < HW; p += 64) {
        const f32x4 av = *(const f32x4*)(ap + (size_t)p * 16);
        const f32x4 tv = *(const f32x4*)(tp + (size_t)p * 16);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool pos = av[j] > 0.f;
            const float xh = pos ? av[j] : av[j] * inv_slope;
            o[j] = (pos ? 1.f : slope) * rstd[j] * (tv[j] - m1[j] - xh * m2[j]);
        }
        *(f32x4*)(op + (size_t)p * 16) = o;
    }
}

__global__ __launch_bounds__(256) void inorm_second_kernel(dasr_tensor a, dasr_tensor t, dasr_tensor ga, int C, int H, int W, float slope,
                                                           const float* __restrict__ stats, dasr_tensor out, int accumulate) {
    __shared__ f32x4 red[16];
    const int ncb = (C + 15) >> 4;
    const int n = blockIdx.x / ncb, cb = blockIdx.x - n * ncb;
    const int q = threadIdx.x & 3, pl = threadIdx.x >> 2;
    const float* ap = (const float*)a.p + (size_t)n * a.n_stride + (size_t)cb * a.cb_stride + q * 4;
    const float* tp = (const float*)t.p + (size_t)n * t.n_stride + (size_t)cb * t.cb_stride + q * 4;
    const float* gp = (const float*)ga.p + (size_t)n * ga.n_stride + (size_t)cb * ga.cb_stride + q * 4;
    float* op = (float*)out.p + (size_t)n * out.n_stride + (size_t)cb * out.cb_stride + q * 4;
    const int HW = H * W;
    const float inv_slope = 1.f / slope, inv = 1.f / HW;
    f32x4 sw = {0.f, 0.f, 0.f, 0.f}, sz = sw, swx = sw, sxz = sw, swz = sw;
    for (int p = pl; p < HW; p += 64) {
        const f32x4 av = *(const f32x4*)(ap + (size_t)p * 16);
        const f32x4 tv = *(const f32x4*)(tp + (size_t)p * 16);
        const f32x4 gv = *(const f32x4*)(gp + (size_t)p * 16);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool pos = av[j] > 0.f;
            const float xh = pos ? av[j] : av[j] * inv_slope;
            const float w = pos ? gv[j] : gv[j] * slope;
            sw[j] += w;
            sz[j] += tv[j];
            swx[j] += w * xh;
            sxz[j] += xh * tv[j];
            swz[j] += w * tv[j];
        }
    }
    const f32x4 mw = quad_reduce(sw, red) * inv, mz = quad_reduce(sz, red) * inv, pw = quad_reduce(swx, red) * inv, pz = quad_reduce(sxz, red) * inv,
                qq = quad_reduce(swz, red) * inv;
    f32x4 r2, k0;
    const float* st = stats + ((size_t)n * ncb * 16 + cb * 16 + q * 4) * 2;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        r2[j] = st[2 * j + 1] * st[2 * j + 1];
        k0[j] = qq[j] - mw[j] * mz[j] - 3.f * pw[j] * pz[j];
    }
    for (int p = pl; p < HW; p += 64) {
        const f32x4 av = *(const f32x4*)(ap + (size_t)p * 16);
        const f32x4 tv = *(const f32x4*)(tp + (size_t)p * 16);
        const f32x4 gv = *(const f32x4*)(gp + (size_t)p * 16);
        f32x4 o = accumulate ? *(const f32x4*)(op + (size_t)p * 16) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool pos = av[j] > 0.f;
            const float xh = pos ? av[j] : av[j] * inv_slope;
            const float w = pos ? gv[j] : gv[j] * slope;
            o[j] -= r2[j] * (xh * k0[j] + pz[j] * (w - mw[j]) + pw[j] * (tv[j] - mz[j]));
        }
        *(f32x4*)(op + (size_t)p * 16) = o;
    }
}

// gradient penalty of the DSN's --wgan (train.py:233-236): nrm = || g ||_2 over the C real channels of a blocked f32 tensor (ALL images),
// pen = weight (nrm - 1)^2.  Two launches: per-workgroup partial sums of squares into `part` (deterministic: fixed grid, fixed order), then one workgroup
// adds them in order and writes out[0] = nrm, out[1] = pen, out[2] = dpen/dnrm / nrm = 2 weight (nrm - 1) / nrm (the factor in front of
// d<g_const, g(theta)>/dtheta), and loss_acc[0] += pen.
__global__ __launch_bounds__(256) void sumsq_partial_kernel(dasr_tensor g, int N, int C, int H, int W, float* __restrict__ part) {
    __shared__ float red[4];
    const long long HW = (long long)H * W, total = (long long)N * HW;
    float s = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int n = i / HW;
        const long long p = i - (long long)n * HW;
        const float* gp = (const float*)g.p + (size_t)n * g.n_stride + (size_t)p * 16;
        for (int c = 0; c < C; ++c) s += gp[(size_t)(c >> 4) * g.cb_stride + (c & 15)] * gp[(size_t)(c >> 4) * g.cb_stride + (c & 15)];
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
// stage 0: everything (one rank).  Data parallel (the reference takes ONE norm over the global batch, train.py:231-236): stage 1 leaves this rank's
// sum of squares in out[3]; the caller SUM-all-reduces that word; stage 2 finishes with inv_w = 1 / world: the gradient of the GLOBAL mean output
// restricted to this rank's samples is g_r / world, so ||g||^2 = sum_r ||g_r||^2 / world^2, and d pen / d theta = c sum_r <g_r, d g_r / d theta> / world^2:
// the upstream factor of this rank's reverse pass is c / world, the other 1 / world is the data-parallel factor every weight-gradient reduction of the
// rank carries (dasr_wgrad_reduce `scale`), and the ranks' weight gradients are SUMMED
__global__ void gp_finalize_kernel(const float* __restrict__ part, int nblk, float weight, float* __restrict__ out, float* __restrict__ loss_acc, int stage,
                                   float inv_w) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float s = 0.f;
    if (stage == 2) {
        s = out[3] * inv_w * inv_w;
    } else {
        for (int i = 0; i < nblk; ++i) s += part[i];
        if (stage == 1) {
            out[3] = s;
            return;
        }
    }
    const float nrm = sqrtf(s);
    const float pen = weight * (nrm - 1.f) * (nrm - 1.f);
    out[0] = nrm;
    out[1] = pen;
    out[2] = nrm > 0.f ? 2.f * weight * (nrm - 1.f) / nrm * inv_w : 0.f;
    if (loss_acc) loss_acc[0] += pen;
}

// dst[i] = factor * (*scalar) over the C real channels of a blocked f32 tensor (zero elsewhere): a constant upstream gradient whose value lives on the device
__global__ void fill_scaled_kernel(dasr_tensor x, int N, int C, int H, int W, const float* __restrict__ scalar, float factor) {
    const int ncb = (C + 15) >> 4;
    const long long per_plane = (long long)H * W * 16, total = (long long)N * ncb * per_plane;
    const long long gi = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gi >= total) return;
    const long long e = gi % per_plane;
    long long tq = gi / per_plane;
    const int cb = tq % ncb, n = tq / ncb;
    const int c = cb * 16 + (int)(e & 15);
    ((float*)x.p)[(size_t)n * x.n_stride + (size_t)cb * x.cb_stride + e] = c < C ? factor * scalar[0] : 0.f;
}

// ---------------------------------------------------------------------------------------------------
// BatchNorm2d in TRAINING mode (batch statistics, affine) + LeakyReLU, as in Discriminator_VGG_128 (architecture.py:442-495).
// The batch is processed in GROUPS of `group` consecutive images with their own statistics: the reference calls the discriminator
// separately on the fake and on the real half (DASR_model.py:251,288-289), here both halves go through one launch.
// One workgroup per 16-channel plane; thread t: channel quad q = t & 3, pixel lane t >> 2; fixed-order reductions (deterministic).
// stats[g][Cpad][3] = (mean, rstd, biased variance).
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bnorm_lrelu_fwd_kernel(dasr_tensor x, int N, int C, int H, int W, int group, float eps, float slope,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta, dasr_tensor y,
                                                              float* __restrict__ stats) {
    __shared__ f32x4 red[16];
    const int cb = blockIdx.x, q = threadIdx.x & 3, pl = threadIdx.x >> 2, HW = H * W, cpad = ((C + 15) >> 4) * 16;
    const int c0 = cb * 16 + q * 4;
    f32x4 gm = {0.f, 0.f, 0.f, 0.f}, bt = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (c0 + j < C) { gm[j] = gamma[c0 + j]; bt[j] = beta[c0 + j]; }
    for (int g = 0; g * group < N; ++g) {
        const int n0 = g * group, n1 = min(N, n0 + group);
        const float inv = 1.f / (float)((n1 - n0) * HW);
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int n = n0; n < n1; ++n) {
            const float* xp = (const float*)x.p + (size_t)n * x.n_stride + (size_t)cb * x.cb_stride + q * 4;
            for (int p = pl; p < HW; p += 64) s += *(const f32x4*)(xp + (size_t)p * 16);
        }
        const f32x4 mean = quad_reduce(s, red) * inv;
        f32x4 s2 = {0.f, 0.f, 0.f, 0.f};
        for (int n = n0; n < n1; ++n) {
            const float* xp = (const float*)x.p + (size_t)n * x.n_stride + (size_t)cb * x.cb_stride + q * 4;
            for (int p = pl; p < HW; p += 64) {
                const f32x4 d = *(const f32x4*)(xp + (size_t)p * 16) - mean;
                s2 += d * d;
            }
        }
        const f32x4 var = quad_reduce(s2, red) * inv;
        f32x4 rstd;
#pragma unroll
        for (int j = 0; j < 4; ++j) rstd[j] = 1.f / sqrtf(var[j] + eps);
        for (int n = n0; n < n1; ++n) {
            const float* xp = (const float*)x.p + (size_t)n * x.n_stride + (size_t)cb * x.cb_stride + q * 4;
            float* yp = (float*)y.p + (size_t)n * y.n_stride + (size_t)cb * y.cb_stride + q * 4;
            for (int p = pl; p < HW; p += 64) {
                f32x4 v = (*(const f32x4*)(xp + (size_t)p * 16) - mean) * rstd * gm + bt;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : v[j] * slope;
                *(f32x4*)(yp + (size_t)p * 16) = v;
            }
        }
        if (pl == 0) {
            float* st = stats + ((size_t)g * cpad + c0) * 3;
#pragma unroll
            for (int j = 0; j < 4; ++j) { st[3 * j] = mean[j]; st[3 * j + 1] = rstd[j]; st[3 * j + 2] = var[j]; }
        }
    }
}

// backward: z = xhat * gamma + beta (recomputed from the saved conv output x and the statistics), gz = ga * lrelu'(z);
// per group: gx = gamma * rstd * (gz - mean(gz) - xhat * mean(gz * xhat));  dgamma = sum over ALL groups of sum(gz * xhat), dbeta = sum gz
// (written, scaled by pscale, when the pointers are given: the discriminator step; the generator step only needs gx of the fake group).
__global__ __launch_bounds__(256) void bnorm_lrelu_bwd_kernel(dasr_tensor x, dasr_tensor ga, int N, int C, int H, int W, int group, float slope,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              const float* __restrict__ stats, dasr_tensor gx, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, float pscale) {
    __shared__ f32x4 red[16];
    const int cb = blockIdx.x, q = threadIdx.x & 3, pl = threadIdx.x >> 2, HW = H * W, cpad = ((C + 15) >> 4) * 16;
    const int c0 = cb * 16 + q * 4;
    f32x4 gm = {0.f, 0.f, 0.f, 0.f}, bt = {0.f, 0.f, 0.f, 0.f}, dg = {0.f, 0.f, 0.f, 0.f}, db = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (c0 + j < C) { gm[j] = gamma[c0 + j]; bt[j] = beta[c0 + j]; }
    for (int g = 0; g * group < N; ++g) {
        const int n0 = g * group, n1 = min(N, n0 + group);
        const float inv = 1.f / (float)((n1 - n0) * HW);
        const float* st = stats + ((size_t)g * cpad + c0) * 3;
        f32x4 mean, rstd;
#pragma unroll
        for (int j = 0; j < 4; ++j) { mean[j] = st[3 * j]; rstd[j] = st[3 * j + 1]; }
        f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
        for (int n = n0; n < n1; ++n) {
            const float* xp = (const float*)x.p + (size_t)n * x.n_stride + (size_t)cb * x.cb_stride + q * 4;
            const float* gp = (const float*)ga.p + (size_t)n * ga.n_stride + (size_t)cb * ga.cb_stride + q * 4;
            for (int p = pl; p < HW; p += 64) {
                const f32x4 xh = (*(const f32x4*)(xp + (size_t)p * 16) - mean) * rstd;
                const f32x4 z = xh * gm + bt;
                f32x4 gz = *(const f32x4*)(gp + (size_t)p * 16);
#pragma unroll
                for (int j = 0; j < 4; ++j) gz[j] = z[j] > 0.f ? gz[j] : gz[j] * slope;
                s1 += gz;
                s2 += gz * xh;
            }
        }
        const f32x4 t1 = quad_reduce(s1, red), t2 = quad_reduce(s2, red);
        db += t1;
        dg += t2;
        const f32x4 m1 = t1 * inv, m2 = t2 * inv;
        for (int n = n0; n < n1; ++n) {
            const float* xp = (const float*)x.p + (size_t)n * x.n_stride + (size_t)cb * x.cb_stride + q * 4;
            const float* gp = (const float*)ga.p + (size_t)n * ga.n_stride + (size_t)cb * ga.cb_stride + q * 4;
            float* op = (float*)gx.p + (size_t)n * gx.n_stride + (size_t)cb * gx.cb_stride + q * 4;
            for (int p = pl; p < HW; p += 64) {
                const f32x4 xh = (*(const f32x4*)(xp + (size_t)p * 16) - mean) * rstd;
                const f32x4 z = xh * gm + bt;
                f32x4 gz = *(const f32x4*)(gp + (size_t)p * 16);
#pragma unroll
                for (int j = 0; j < 4; ++j) gz[j] = z[j] > 0.f ? gz[j] : gz[j] * slope;
                *(f32x4*)(op + (size_t)p * 16) = gm * rstd * (gz - m1 - xh * m2);
            }
        }
    }
    if (pl == 0 && dgamma) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (c0 + j < C) { dgamma[c0 + j] = dg[j] * pscale; dbeta[c0 + j] = db[j] * pscale; }
    }
}

// ---------------------------------------------------------------------------------------------------
// Second-order pieces of BatchNorm2d (TRAINING mode, affine) + LeakyReLU for the DSN's `--wgan` gradient penalty with `--norm_layer Batch` (round 6;
// codes/DSN/train.py:231-236 through model.py:136-160,176-189).  Same algebra as the InstanceNorm pair above with the means taken over a GROUP of images
// (the penalty's D(sample) is ONE call: group = N), the scale gamma behind the normalisation and the LeakyReLU' read from y = gamma xhat + beta, both
// recomputed from the saved conv output x and the statistics (as bnorm_lrelu_bwd_kernel does):
//   * bnorm_lrelu_jvp:  out = lrelu'(y) gamma r (t - mean t - xhat mean(xhat t))
//   * bnorm_second:     with u = lrelu'(y) ga (ga = adjoint of the tangent output), w = gamma u:
//       out (+)= -r^2 [ xhat (mean(w t) - mean w mean t - 3 mean(w xhat) mean(xhat t)) + mean(xhat t) (w - mean w) + mean(w xhat) (t - mean t) ]
//       dgamma (+)= pscale * r * count * (mean(u t) - mean u mean t - mean(u xhat) mean(xhat t))      (= sum u xhat_dot: the tangent output is linear in gamma)
//     (beta enters only through the sign of y: no term.)  One workgroup per 16-channel plane, fixed-order reductions.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bnorm_lrelu_jvp_kernel(dasr_tensor x, dasr_tensor t, int N, int C, int H, int W, int group, float slope,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              const float* __restrict__ stats, dasr_tensor out) {
    __shared__ f32x4 red[16];
    const int cb = blockIdx.x, q = threadIdx.x & 3, pl = threadIdx.x >> 2, HW = H * W, cpad = ((C + 15) >> 4) * 16;
    const int c0 = cb * 16 + q * 4;
    f32x4 gm = {0.f, 0.f, 0.f, 0.f}, bt = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (c0 + j < C) { gm[j] = gamma[c0 + j]; bt[j] = beta[c0 + j]; }
    for (int g = 0; g * group < N; ++g) {
        const int n0 = g * group, n1 = min(N, n0 + group);
        const float inv = 1.f / (float)((n1 - n0) * HW);
        const float* st = stats + ((size_t)g * cpad + c0) * 3;
        f32x4 mean, rstd;
#pragma unroll
        for (int j = 0; j < 4; ++j) { mean[j] = st[3 * j]; rstd[j] = st[3 * j + 1]; }
        f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
        for (int n = n0; n < n1; ++n) {
            const float* xp = (const float*)x.p + (size_t)n * x.n_stride + (size_t)cb * x.cb_stride + q * 4;
            const float* tp = (const float*)t.p + (size_t)n * t.n_stride + (size_t)cb * t.cb_stride + q * 4;
            for (int p = pl; p < HW; p += 64) {
                const f32x4 xh = (*(const f32x4*)(xp + (size_t)p * 16) - mean) * rstd;
                const f32x4 tv = *(const f32x4*)(tp + (size_t)p * 16);
                s1 += tv;
                s2 += tv * xh;
            }
        }
        const f32x4 m1 = quad_reduce(s1, red) * inv, m2 = quad_reduce(s2, red) * inv;
        for (int n = n0; n < n1; ++n) {
            const float* xp = (const float*)x.p + (size_t)n * x.n_stride + (size_t)cb * x.cb_stride + q * 4;
            const float* tp = (const float*)t.p + (size_t)n * t.n_stride + (size_t)cb * t.cb_stride + q * 4;
            float* op = (float*)out.p + (size_t)n * out.n_stride + (size_t)cb * out.cb_stride + q * 4;
            for (int p = pl; p < HW; p += 64) {
                const f32x4 xh = (*(const f32x4*)(xp + (size_t)p * 16) - mean) * rstd;
                const f32x4 y = xh * gm + bt;
                const f32x4 tv = *(const f32x4*)(tp + (size_t)p * 16);
                f32x4 o = gm * rstd * (tv - m1 - xh * m2);
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = y[j] > 0.f ? o[j] : o[j] * slope;
                *(f32x4*)(op + (size_t)p * 16) = o;
            }
        }
    }
}

__global__ __launch_bounds__(256) void bnorm_second_kernel(dasr_tensor x, dasr_tensor t, dasr_tensor ga, int N, int C, int H, int W, int group, float slope,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ stats, dasr_tensor out, int accumulate, float* __restrict__ dgamma,
                                                           float pscale) {
    __shared__ f32x4 red[16];
    const int cb = blockIdx.x, q = threadIdx.x & 3, pl = threadIdx.x >> 2, HW = H * W, cpad = ((C + 15) >> 4) * 16;
    const int c0 = cb * 16 + q * 4;
    f32x4 gm = {0.f, 0.f, 0.f, 0.f}, bt = {0.f, 0.f, 0.f, 0.f}, dg = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (c0 + j < C) { gm[j] = gamma[c0 + j]; bt[j] = beta[c0 + j]; }
    for (int g = 0; g * group < N; ++g) {
        const int n0 = g * group, n1 = min(N, n0 + group);
        const float cnt = (float)((n1 - n0) * HW), inv = 1.f / cnt;
        const float* st = stats + ((size_t)g * cpad + c0) * 3;
        f32x4 mean, rstd;
#pragma unroll
        for (int j = 0; j < 4; ++j) { mean[j] = st[3 * j]; rstd[j] = st[3 * j + 1]; }
        f32x4 su = {0.f, 0.f, 0.f, 0.f}, sz = su, sux = su, sxz = su, suz = su;
        for (int n = n0; n < n1; ++n) {
            const float* xp = (const float*)x.p + (size_t)n * x.n_stride + (size_t)cb * x.cb_stride + q * 4;
            const float* tp = (const float*)t.p + (size_t)n * t.n_stride + (size_t)cb * t.cb_stride + q * 4;
            const float* gp = (const float*)ga.p + (size_t)n * ga.n_stride + (size_t)cb * ga.cb_stride + q * 4;
            for (int p = pl; p < HW; p += 64) {
                const f32x4 xh = (*(const f32x4*)(xp + (size_t)p * 16) - mean) * rstd;
                const f32x4 y = xh * gm + bt;
                const f32x4 tv = *(const f32x4*)(tp + (size_t)p * 16);
                f32x4 u = *(const f32x4*)(gp + (size_t)p * 16);
#pragma unroll
                for (int j = 0; j < 4; ++j) u[j] = y[j] > 0.f ? u[j] : u[j] * slope;
                su += u;
                sz += tv;
                sux += u * xh;
                sxz += xh * tv;
                suz += u * tv;
            }
        }
        const f32x4 mu = quad_reduce(su, red) * inv, mz = quad_reduce(sz, red) * inv, pu = quad_reduce(sux, red) * inv, pz = quad_reduce(sxz, red) * inv,
                    qu = quad_reduce(suz, red) * inv;
        dg += rstd * cnt * (qu - mu * mz - pu * pz);
        const f32x4 mw = gm * mu, pw = gm * pu, r2 = rstd * rstd;
        const f32x4 k0 = gm * qu - mw * mz - 3.f * pw * pz;
        for (int n = n0; n < n1; ++n) {
            const float* xp = (const float*)x.p + (size_t)n * x.n_stride + (size_t)cb * x.cb_stride + q * 4;
            const float* tp = (const float*)t.p + (size_t)n * t.n_stride + (size_t)cb * t.cb_stride + q * 4;
            const float* gp = (const float*)ga.p + (size_t)n * ga.n_stride + (size_t)cb * ga.cb_stride + q * 4;
            float* op = (float*)out.p + (size_t)n * out.n_stride + (size_t)cb * out.cb_stride + q * 4;
            for (int p = pl; p < HW; p += 64) {
                const f32x4 xh = (*(const f32x4*)(xp + (size_t)p * 16) - mean) * rstd;
                const f32x4 y = xh * gm + bt;
                const f32x4 tv = *(const f32x4*)(tp + (size_t)p * 16);
                f32x4 w = *(const f32x4*)(gp + (size_t)p * 16);
#pragma unroll
                for (int j = 0; j < 4; ++j) w[j] = (y[j] > 0.f ? w[j] : w[j] * slope) * gm[j];
                f32x4 o = accumulate ? *(const f32x4*)(op + (size_t)p * 16) : f32x4{0.f, 0.f, 0.f, 0.f};
                o -= r2 * (xh * k0 + pz * (w - mw) + pw * (tv - mz));
                *(f32x4*)(op + (size_t)p * 16) = o;
            }
        }
    }
    if (pl == 0 && dgamma) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (c0 + j < C) dgamma[c0 + j] = (accumulate ? dgamma[c0 + j] : 0.f) + dg[j] * pscale;
    }
}

// running statistics of nn.BatchNorm2d after one training-mode forward on group g (momentum 0.1, UNBIASED variance, num_batches_tracked + 1)
__global__ void bnorm_running_kernel(const float* __restrict__ stats, int g, int C, int count, float momentum, float* __restrict__ rmean,
                                     float* __restrict__ rvar, float* __restrict__ nbt) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x, cpad = ((C + 15) >> 4) * 16;
    if (c < C) {
        const float* st = stats + ((size_t)g * cpad + c) * 3;
        rmean[c] = (1.f - momentum) * rmean[c] + momentum * st[0];
        rvar[c] = (1.f - momentum) * rvar[c] + momentum * st[2] * ((float)count / (float)(count > 1 ? count - 1 : 1));
    }
    if (c == 0 && nbt) nbt[0] += 1.f;
}

// ---------------------------------------------------------------------------------------------------
// GANLoss(gan_type) against a constant target over C real channels (loss.py:8-40), MODE = gan_type:
//   0 'vanilla'  BCEWithLogits: l = max(x,0) - x t + log1p(exp(-|x|)),  dl/dx = sigmoid(x) - t
//   1 'lsgan'    MSE:           l = (x - t)^2,                          dl/dx = 2 (x - t)
//   2 'wgan-gp'  l = -x if the target is the real label (t > 0.5) else x (the reference never applies its gradient penalty: DASR_model.py:114-118
//                builds cri_gp, optimize_parameters does not use it)
// loss_acc[0] += coef * sum(l); score_acc += score_coef * sum(x) (disc score); grad = gcoef * dl/dx (zero on padded channels)
// ---------------------------------------------------------------------------------------------------
template <int MODE>
__global__ void gan_loss_kernel(dasr_tensor x, int N, int C, int H, int W, float target, float coef, float gcoef, float* loss_acc,
                                float* score_acc, float score_coef, dasr_tensor grad, dasr_red rs) {
    __shared__ float red[4];
    const long long total = (long long)N * H * W;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    float l = 0.f, sc = 0.f;
    if (i < total) {
        const int n = i / ((long long)H * W);
        const long long p = i - (long long)n * H * W;
        const float* xp = (const float*)x.p + (size_t)n * x.n_stride + (size_t)p * 16;
        float g[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) g[j] = 0.f;
        for (int c = 0; c < C; ++c) {
            const float v = xp[c];
            sc += v;
            if (MODE == 0) {
                l += fmaxf(v, 0.f) - v * target + log1pf(expf(-fabsf(v)));
                g[c] = gcoef * (1.f / (1.f + expf(-v)) - target);
            } else if (MODE == 1) {
                l += (v - target) * (v - target);
                g[c] = gcoef * 2.f * (v - target);
            } else {
                const float sgn = target > 0.5f ? -1.f : 1.f;
                l += sgn * v;
                g[c] = gcoef * sgn;
            }
        }
        if (grad.p) {
            float* gp = (float*)grad.p + (size_t)n * grad.n_stride + (size_t)p * 16;
#pragma unroll
            for (int j = 0; j < 4; ++j) ((f32x4*)gp)[j] = f32x4{g[4 * j], g[4 * j + 1], g[4 * j + 2], g[4 * j + 3]};
        }
    }
    const float v[2] = {block_sum_256(l, red), block_sum_256(sc, red)};
    float* const acc[2] = {loss_acc, score_acc};
    const float cf[2] = {coef, score_coef};
    grid_sum_commit<2>(rs, v, acc, cf);   // fixed-order sums over the grid (common.h)
}

// ---------------------------------------------------------------------------------------------------
// Relativistic average GAN terms (`ragan: true`, DASR_model.py:240-244,273-275): a, b = logit maps [N][1][H][W] of the two halves,
//   L = coef * sum_{n,p} [ bce(a - mean_n(b), ta) + bce(b - mean_n(a), tb) ],   mean over the GLOBAL batch (n_glob samples) per pixel.
// One thread per pixel, three stages so that a data-parallel run can all-reduce the two tiny per-pixel sums in between:
//   stage 0: sums[0:HW] = sum_n a, sums[HW:2HW] = sum_n b                                      (-> all-reduce SUM)
//   stage 1: za = a - sums_b / n_glob, zb = b - sums_a / n_glob; loss / score accumulation of the local samples;
//            part[0:HW] = sum_n (sigmoid(za) - ta), part[HW:2HW] = sum_n (sigmoid(zb) - tb)    (-> all-reduce SUM)
//   stage 2: ga = gcoef * ((sigmoid(za) - ta) - part_b / n_glob), gb = gcoef * ((sigmoid(zb) - tb) - part_a / n_glob)   (either may be null)
// ---------------------------------------------------------------------------------------------------
// FORM 0 (SRN, above): term(z, t) = BCE-with-logits, score = the raw logit; FORM 2 / 3: the same with GANLoss('lsgan') / ('wgan-gp') terms.  FORM 1 (DSN `--ragan`, codes/DSN/train.py:221-223 + loss.py:11-41 on
// model.py:98-106's sigmoid(x - mean_n(y))): term(z, t) = -log(sigmoid(z) + eps) for t > 0.5, -log(1 - sigmoid(z) + eps) for 0 <= t <= 0.5,
// absent (0) for t < 0; score = sigmoid(z).
template <int FORM>
__device__ __forceinline__ void rel_term(float z, float t, float eps, float& l, float& q, float& s) {
    s = 1.f / (1.f + expf(-z));
    if (FORM == 0) {
        l = fmaxf(z, 0.f) - z * t + log1pf(expf(-fabsf(z)));
        q = s - t;
    } else if (FORM == 2) {   // GANLoss('lsgan') on the relativistic logits
        l = (z - t) * (z - t);
        q = 2.f * (z - t);
    } else if (FORM == 3) {   // GANLoss('wgan-gp'): -z against the real label, +z otherwise
        q = t > 0.5f ? -1.f : 1.f;
        l = q * z;
    } else if (t < 0.f) {
        l = 0.f;
        q = 0.f;
    } else if (t > 0.5f) {
        l = -logf(s + eps);
        q = -s * (1.f - s) / (s + eps);
    } else {
        l = -logf(1.f - s + eps);
        q = s * (1.f - s) / (1.f - s + eps);
    }
}

template <int FORM>
__global__ void ragan_kernel(dasr_tensor a, dasr_tensor b, int N, int H, int W, int stage, float inv_nglob, float ta, float tb, float coef, float gcoef,
                             float eps, float* __restrict__ sums, float* __restrict__ part, float* loss_acc, float* score_a, float* score_b,
                             float score_coef, dasr_tensor ga, dasr_tensor gb, dasr_red rs) {
    __shared__ float red[4];
    const int HW = H * W;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    float l = 0.f, sca = 0.f, scb = 0.f;
    if (p < HW) {
        const float* ap = (const float*)a.p + (size_t)p * 16;
        const float* bp = (const float*)b.p + (size_t)p * 16;
        if (stage == 0) {
            float sa = 0.f, sb = 0.f;
            for (int n = 0; n < N; ++n) { sa += ap[(size_t)n * a.n_stride]; sb += bp[(size_t)n * b.n_stride]; }
            sums[p] = sa;
            sums[HW + p] = sb;
        } else {
            const float ma = sums[p] * inv_nglob, mb = sums[HW + p] * inv_nglob;
            if (stage == 1) {
                float qa = 0.f, qb = 0.f;
                for (int n = 0; n < N; ++n) {
                    const float av = ap[(size_t)n * a.n_stride], bv = bp[(size_t)n * b.n_stride];
                    float la, lb, da, db, sa, sb;
                    rel_term<FORM>(av - mb, ta, eps, la, da, sa);
                    rel_term<FORM>(bv - ma, tb, eps, lb, db, sb);
                    l += la + lb;
                    qa += da;
                    qb += db;
                    sca += FORM != 1 ? av : sa;
                    scb += FORM != 1 ? bv : sb;
                }
                part[p] = qa;
                part[HW + p] = qb;
            } else {
                const float ca = part[p] * inv_nglob, cb = part[HW + p] * inv_nglob;
                for (int n = 0; n < N; ++n) {
                    float la, lb, da, db, sa, sb;
                    rel_term<FORM>(ap[(size_t)n * a.n_stride] - mb, ta, eps, la, da, sa);
                    rel_term<FORM>(bp[(size_t)n * b.n_stride] - ma, tb, eps, lb, db, sb);
                    if (ga.p) {
                        f32x4* g = (f32x4*)((float*)ga.p + (size_t)n * ga.n_stride + (size_t)p * 16);
                        g[0] = f32x4{gcoef * (da - cb), 0.f, 0.f, 0.f};
                        g[1] = g[2] = g[3] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                    if (gb.p) {
                        f32x4* g = (f32x4*)((float*)gb.p + (size_t)n * gb.n_stride + (size_t)p * 16);
                        g[0] = f32x4{gcoef * (db - ca), 0.f, 0.f, 0.f};
                        g[1] = g[2] = g[3] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                }
            }
        }
    }
    if (stage == 1) {
        const float v[3] = {block_sum_256(l, red), block_sum_256(sca, red), block_sum_256(scb, red)};
        float* const acc[3] = {loss_acc, score_a, score_b};
        const float cf[3] = {coef, score_coef, score_coef};
        grid_sum_commit<3>(rs, v, acc, cf);
    }
}

// ---------------------------------------------------------------------------------------------------
// Haar DWT level 1 on C (<=5) channels of plane 0: LL -> ll plane (C ch), [LH|HL|HH] -> hc plane (3C ch).
// Convention (oracle/nets.py::HaarDWT): block [[a,b],[c,d]]: LL=(a+b+c+d)/2, LH=(a+b-c-d)/2, HL=(a-b+c-d)/2, HH=(a-b-c+d)/2
// norm: LL*0.5, Hc*0.5+0.5 (DASR_model.py:442-452).  Thread per output pixel.
// ---------------------------------------------------------------------------------------------------
__global__ void dwt_fwd_kernel(dasr_tensor x, int N, int C, int H2, int W2, int norm, dasr_tensor ll, dasr_tensor hc) {
    const long long total = (long long)N * H2 * W2;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int xx = i % W2;
    long long t = i / W2;
    const int yy = t % H2;
    const int n = t / H2;
    const int W = 2 * W2;
    const float* xp = (const float*)x.p + (size_t)n * x.n_stride;
    const float* pa = xp + ((size_t)(2 * yy) * W + 2 * xx) * 16;
    const float* pc = pa + (size_t)W * 16;
    float L[16], Hh[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) L[j] = Hh[j] = 0.f;
    // norm bit 0: LL * 0.5, bands * 0.5 + 0.5; bit 1: the DSN discriminator's 'sum' format (codes/DSN/model.py:113-114): hc = (LH + HL + HH) / 3, C channels
    // bit 2: the LINEAR part only (no + 0.5): the tangent of the normalised bands (second-order pass of the DSN's --wgan penalty)
    const float s = (norm & 1) ? 0.5f : 1.f, off = ((norm & 1) && !(norm & 4)) ? 0.5f : 0.f;
    for (int c = 0; c < C; ++c) {
        const float a = pa[c], b = pa[16 + c], cc = pc[c], d = pc[16 + c];
        L[c] = (a + b + cc + d) * 0.5f * s;
        const float lh = (a + b - cc - d) * 0.5f * s + off, hl = (a - b + cc - d) * 0.5f * s + off, hh = (a - b - cc + d) * 0.5f * s + off;
        if (norm & 2) {
            Hh[c] = (lh + hl + hh) / 3.f;
        } else {
            Hh[c] = lh;
            Hh[C + c] = hl;
            Hh[2 * C + c] = hh;
        }
    }
    const size_t po = ((size_t)yy * W2 + xx) * 16;
    if (ll.p) {
        float* o = (float*)ll.p + (size_t)n * ll.n_stride + po;
#pragma unroll
        for (int j = 0; j < 4; ++j) ((f32x4*)o)[j] = f32x4{L[4 * j], L[4 * j + 1], L[4 * j + 2], L[4 * j + 3]};
    }
    if (hc.p) {
        float* o = (float*)hc.p + (size_t)n * hc.n_stride + po;
#pragma unroll
        for (int j = 0; j < 4; ++j) ((f32x4*)o)[j] = f32x4{Hh[4 * j], Hh[4 * j + 1], Hh[4 * j + 2], Hh[4 * j + 3]};
    }
}

// adjoint: gx (+)= DWT^T (gll, ghc); gll / ghc may be null (treated as zero)
__global__ void dwt_bwd_kernel(dasr_tensor gll, dasr_tensor ghc, int N, int C, int H2, int W2, int norm, dasr_tensor gx, int accumulate) {
    const long long total = (long long)N * H2 * W2;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int xx = i % W2;
    long long t = i / W2;
    const int yy = t % H2;
    const int n = t / H2;
    const int W = 2 * W2;
    const size_t po = ((size_t)yy * W2 + xx) * 16;
    const float* gl = gll.p ? (const float*)gll.p + (size_t)n * gll.n_stride + po : nullptr;
    const float* gh = ghc.p ? (const float*)ghc.p + (size_t)n * ghc.n_stride + po : nullptr;
    float* pa = (float*)gx.p + (size_t)n * gx.n_stride + ((size_t)(2 * yy) * W + 2 * xx) * 16;
    float* pc = pa + (size_t)W * 16;
    const float s = ((norm & 1) ? 0.5f : 1.f) * 0.5f;
    float A[16], B[16], Cc[16], D[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) A[j] = B[j] = Cc[j] = D[j] = 0.f;
    for (int c = 0; c < C; ++c) {
        const float l = gl ? gl[c] : 0.f;
        float lh, hl, hh;
        if (norm & 2) lh = hl = hh = gh ? gh[c] * (1.f / 3.f) : 0.f;   // 'sum' format: every band receives a third of the gradient
        else lh = gh ? gh[c] : 0.f, hl = gh ? gh[C + c] : 0.f, hh = gh ? gh[2 * C + c] : 0.f;
        A[c] = s * (l + lh + hl + hh);
        B[c] = s * (l + lh - hl - hh);
        Cc[c] = s * (l - lh + hl - hh);
        D[c] = s * (l - lh - hl + hh);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f32x4 va = {A[4 * j], A[4 * j + 1], A[4 * j + 2], A[4 * j + 3]}, vb = {B[4 * j], B[4 * j + 1], B[4 * j + 2], B[4 * j + 3]};
        f32x4 vc = {Cc[4 * j], Cc[4 * j + 1], Cc[4 * j + 2], Cc[4 * j + 3]}, vd = {D[4 * j], D[4 * j + 1], D[4 * j + 2], D[4 * j + 3]};
        if (accumulate) {
            va += ((f32x4*)pa)[j];
            vb += ((f32x4*)(pa + 16))[j];
            vc += ((f32x4*)pc)[j];
            vd += ((f32x4*)(pc + 16))[j];
        }
        ((f32x4*)pa)[j] = va;
        ((f32x4*)(pa + 16))[j] = vb;
        ((f32x4*)pc)[j] = vc;
        ((f32x4*)(pc + 16))[j] = vd;
    }
}

// ---------------------------------------------------------------------------------------------------
// Depthwise k x k low-pass with zero padding (k-1)/2 on C (<=4) channels of plane 0, weights w[k*k]
// (GaussianFilter / AvgPool2d(count_include_pad=True) of FilterLow, architecture.py:1177-1224).
//   mode 0: out_low = low(x)                  ; out_high = a_h * (x - low(x)) + b_h
//   mode 1 (adjoint): out_low(=gx) (+)= low(g_low) + a_h * (g_high - low(g_high))   (the kernel is symmetric)
// Thread per pixel; the window is read through L1/L2 (3-4 floats per tap).
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float valid_frac(int y, int x, int H, int W, int k, int r) {
    // fraction of the k x k window centred at (y, x) that lies inside the image (uniform kernel: = sum of used weights)
    const int ny = min(y + r, H - 1) - max(y - r, 0) + 1, nx = min(x + r, W - 1) - max(x - r, 0) + 1;
    return (float)(ny * nx) / (float)(k * k);
}

__global__ void lowpass_kernel(dasr_tensor x, dasr_tensor x2, const float* __restrict__ w, int k, int N, int C, int H, int W, int mode_,
                               float a_h, float b_h, dasr_tensor out_low, dasr_tensor out_high, int accumulate) {
    const int mode = mode_ & 1, norm_valid = (mode_ >> 1) & 1;  // norm_valid: AvgPool2d(count_include_pad=False)
    const long long total = (long long)N * H * W;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int xx = i % W;
    long long t = i / W;
    const int yy = t % H;
    const int n = t / H;
    const int r = (k - 1) / 2;
    const float* xp = x.p ? (const float*)x.p + (size_t)n * x.n_stride : nullptr;
    const float* x2p = x2.p ? (const float*)x2.p + (size_t)n * x2.n_stride : nullptr;
    f32x4 lo = {0.f, 0.f, 0.f, 0.f}, lo2 = {0.f, 0.f, 0.f, 0.f};
    for (int ky = 0; ky < k; ++ky) {
        const int sy = yy + ky - r;
        if (sy < 0 || sy >= H) continue;
        for (int kx = 0; kx < k; ++kx) {
            const int sx = xx + kx - r;
            if (sx < 0 || sx >= W) continue;
            float wt = w[ky * k + kx];
            if (norm_valid && mode == 1) wt /= valid_frac(sy, sx, H, W, k, r);  // adjoint: the normaliser belongs to the OUTPUT pixel
            const size_t o = ((size_t)sy * W + sx) * 16;
            if (xp) lo += *(const f32x4*)(xp + o) * wt;
            if (x2p) lo2 += *(const f32x4*)(x2p + o) * wt;
        }
    }
    if (norm_valid && mode == 0) {
        const float inv = 1.f / valid_frac(yy, xx, H, W, k, r);
        lo *= inv;
    }
    const size_t po = ((size_t)yy * W + xx) * 16;
    f32x4 z = {0.f, 0.f, 0.f, 0.f};
    auto maskc = [&](f32x4 v) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j >= C) v[j] = 0.f;
        return v;
    };
    if (mode == 0) {
        const f32x4 c = *(const f32x4*)(xp + po);
        if (out_low.p) {
            float* o = (float*)out_low.p + (size_t)n * out_low.n_stride + po;
            ((f32x4*)o)[0] = maskc(lo);
            ((f32x4*)o)[1] = z; ((f32x4*)o)[2] = z; ((f32x4*)o)[3] = z;
        }
        if (out_high.p) {
            float* o = (float*)out_high.p + (size_t)n * out_high.n_stride + po;
            f32x4 hv = (c - lo) * a_h + b_h;
            ((f32x4*)o)[0] = maskc(hv);
            ((f32x4*)o)[1] = z; ((f32x4*)o)[2] = z; ((f32x4*)o)[3] = z;
        }
    } else {
        // x = g_low (may be null), x2 = g_high (may be null)
        f32x4 g = lo;
        if (x2p) g += (*(const f32x4*)(x2p + po) - lo2) * a_h;
        float* o = (float*)out_low.p + (size_t)n * out_low.n_stride + po;
        g = maskc(g);
        if (accumulate) g += ((f32x4*)o)[0];
        ((f32x4*)o)[0] = g;
        if (!accumulate) { ((f32x4*)o)[1] = z; ((f32x4*)o)[2] = z; ((f32x4*)o)[3] = z; }
    }
}

// ---------------------------------------------------------------------------------------------------
// MaxPool2d(2,2) forward / backward (first maximum in scan order wins, like ATen), f32 or bf16 tensors
// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ void maxpool_fwd_kernel(dasr_tensor x, int N, int C, int Ho, int Wo, dasr_tensor y, int Win) {
    const int ncb = (C + 15) >> 4;
    const long long total = (long long)N * ncb * Ho * Wo * 4;
    const long long gi = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gi >= total) return;
    const int q = gi & 3;
    long long i = gi >> 2;
    const int xx = i % Wo; i /= Wo;
    const int yy = i % Ho; i /= Ho;
    const int cb = i % ncb;
    const int n = i / ncb;
    const int W = Win > 0 ? Win : 2 * Wo;   // row stride of the input: an odd input width drops its last column (nn.MaxPool2d floors)
    const T* xp = (const T*)x.p + (size_t)n * x.n_stride + (size_t)cb * x.cb_stride + q * 4;
    T* yp = (T*)y.p + (size_t)n * y.n_stride + (size_t)cb * y.cb_stride + ((size_t)yy * Wo + xx) * 16 + q * 4;
    float m[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) m[j] = -3.4e38f;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const T* s = xp + ((size_t)(2 * yy + (d >> 1)) * W + 2 * xx + (d & 1)) * 16;
#pragma unroll
        for (int j = 0; j < 4; ++j) m[j] = fmaxf(m[j], (float)s[j]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) yp[j] = (T)m[j];
}

template <typename T>
__global__ void maxpool_bwd_kernel(dasr_tensor x, dasr_tensor gy, int N, int C, int Ho, int Wo, dasr_tensor gx, int relu_mask, int Win) {
    const int ncb = (C + 15) >> 4;
    const long long total = (long long)N * ncb * Ho * Wo * 4;
    const long long gi = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gi >= total) return;
    const int q = gi & 3;
    long long i = gi >> 2;
    const int xx = i % Wo; i /= Wo;
    const int yy = i % Ho; i /= Ho;
    const int cb = i % ncb;
    const int n = i / ncb;
    const int W = Win > 0 ? Win : 2 * Wo;   // row stride of the input: an odd input width drops its last column (nn.MaxPool2d floors)
    const T* xp = (const T*)x.p + (size_t)n * x.n_stride + (size_t)cb * x.cb_stride + q * 4;
    const T* gp = (const T*)gy.p + (size_t)n * gy.n_stride + (size_t)cb * gy.cb_stride + ((size_t)yy * Wo + xx) * 16 + q * 4;
    T* op = (T*)gx.p + (size_t)n * gx.n_stride + (size_t)cb * gx.cb_stride + q * 4;
    float m[4];
    int am[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { m[j] = -3.4e38f; am[j] = 0; }
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const T* s = xp + ((size_t)(2 * yy + (d >> 1)) * W + 2 * xx + (d & 1)) * 16;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float v = (float)s[j];
            if (v > m[j]) { m[j] = v; am[j] = d; }
        }
    }
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        T* o = op + ((size_t)(2 * yy + (d >> 1)) * W + 2 * xx + (d & 1)) * 16;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (am[j] == d && !(relu_mask && m[j] <= 0.f)) ? gp[j] : (T)0.f;
    }
}

// the same on SPLIT 16-bit tensors (value = hi + lo, the lo planes follow the ncb hi planes: dasr_conv_params::in_wrap): the pair of the
// first maximum is copied / its gradient pair routed, nothing is re-rounded
template <typename T>
__global__ void maxpool_fwd_split_kernel(dasr_tensor x, int N, int C, int Ho, int Wo, dasr_tensor y, int Win) {
    const int ncb = (C + 15) >> 4;
    const long long total = (long long)N * ncb * Ho * Wo * 4;
    const long long gi = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gi >= total) return;
    const int q = gi & 3;
    long long i = gi >> 2;
    const int xx = i % Wo; i /= Wo;
    const int yy = i % Ho; i /= Ho;
    const int cb = i % ncb;
    const int n = i / ncb;
    const int W = Win > 0 ? Win : 2 * Wo;   // row stride of the input: an odd input width drops its last column (nn.MaxPool2d floors)
    const size_t xlo = (size_t)ncb * x.cb_stride, ylo = (size_t)ncb * y.cb_stride;
    const T* xp = (const T*)x.p + (size_t)n * x.n_stride + (size_t)cb * x.cb_stride + q * 4;
    T* yp = (T*)y.p + (size_t)n * y.n_stride + (size_t)cb * y.cb_stride + ((size_t)yy * Wo + xx) * 16 + q * 4;
    float m[4];
    T mh[4], ml[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { m[j] = -3.4e38f; mh[j] = (T)0.f; ml[j] = (T)0.f; }
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const T* s = xp + ((size_t)(2 * yy + (d >> 1)) * W + 2 * xx + (d & 1)) * 16;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const T h = s[j], l = s[xlo + j];
            const float v = (float)h + (float)l;
            if (v > m[j]) { m[j] = v; mh[j] = h; ml[j] = l; }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { yp[j] = mh[j]; yp[ylo + j] = ml[j]; }
}

// GSPLIT false: only the activations x are split, the gradients are plain 16-bit tensors
template <typename T, bool GSPLIT>
__global__ void maxpool_bwd_split_kernel(dasr_tensor x, dasr_tensor gy, int N, int C, int Ho, int Wo, dasr_tensor gx, int relu_mask, int Win) {
    const int ncb = (C + 15) >> 4;
    const long long total = (long long)N * ncb * Ho * Wo * 4;
    const long long gi = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gi >= total) return;
    const int q = gi & 3;
    long long i = gi >> 2;
    const int xx = i % Wo; i /= Wo;
    const int yy = i % Ho; i /= Ho;
    const int cb = i % ncb;
    const int n = i / ncb;
    const int W = Win > 0 ? Win : 2 * Wo;   // row stride of the input: an odd input width drops its last column (nn.MaxPool2d floors)
    const size_t xlo = (size_t)ncb * x.cb_stride, glo = (size_t)ncb * gy.cb_stride, olo = (size_t)ncb * gx.cb_stride;
    const T* xp = (const T*)x.p + (size_t)n * x.n_stride + (size_t)cb * x.cb_stride + q * 4;
    const T* gp = (const T*)gy.p + (size_t)n * gy.n_stride + (size_t)cb * gy.cb_stride + ((size_t)yy * Wo + xx) * 16 + q * 4;
    T* op = (T*)gx.p + (size_t)n * gx.n_stride + (size_t)cb * gx.cb_stride + q * 4;
    float m[4];
    int am[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { m[j] = -3.4e38f; am[j] = 0; }
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const T* s = xp + ((size_t)(2 * yy + (d >> 1)) * W + 2 * xx + (d & 1)) * 16;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float v = (float)s[j] + (float)s[xlo + j];
            if (v > m[j]) { m[j] = v; am[j] = d; }
        }
    }
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        T* o = op + ((size_t)(2 * yy + (d >> 1)) * W + 2 * xx + (d & 1)) * 16;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool hit = am[j] == d && !(relu_mask && m[j] <= 0.f);
            o[j] = hit ? gp[j] : (T)0.f;
            if (GSPLIT) o[olo + j] = hit ? gp[glo + j] : (T)0.f;
        }
    }
}

// L1 between two blocked tensors over all C channels: loss_acc += coef*sum|a-b|; ga = gcoef*sign(a-b)
template <typename T>
__global__ void l1_diff_kernel(dasr_tensor a, dasr_tensor b, int N, int C, int H, int W, float coef, float gcoef, float* loss_acc, dasr_tensor ga,
                               int squared, dasr_red rs) {
    __shared__ float red[4];
    const int ncb = (C + 15) >> 4;
    const long long per = (long long)H * W * 4;
    const long long total = (long long)N * ncb * per;
    const long long gi = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    float l = 0.f;
    if (gi < total) {
        const long long e = (gi % per) * 4;
        long long t = gi / per;
        const int cb = t % ncb;
        const int n = t / ncb;
        const T* ap = (const T*)a.p + (size_t)n * a.n_stride + (size_t)cb * a.cb_stride + e;
        const T* bp = (const T*)b.p + (size_t)n * b.n_stride + (size_t)cb * b.cb_stride + e;
        const int c0 = cb * 16 + (int)(e & 15);
        float g[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float d = (c0 + j < C) ? (float)ap[j] - (float)bp[j] : 0.f;
            l += squared ? d * d : fabsf(d);
            g[j] = squared ? gcoef * 2.f * d : gcoef * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
        }
        if (ga.p) {
            T* gp = (T*)ga.p + (size_t)n * ga.n_stride + (size_t)cb * ga.cb_stride + e;
#pragma unroll
            for (int j = 0; j < 4; ++j) gp[j] = (T)g[j];
        }
    }
    const float v[1] = {block_sum_256(l, red)};
    float* const acc[1] = {loss_acc};
    const float cf[1] = {coef};
    grid_sum_commit<1>(rs, v, acc, cf);
}

// y[c] = x[c] * sc[c] + sh[c] on C (<=4) channels of plane 0 (VGG input normalisation and its adjoint);
// optional accumulate into y (used for dL/dSR += dL/dnorm / std)
template <typename TO, bool SPLIT = false>
__global__ void affine4_kernel(dasr_tensor x, int N, int C, int H, int W, f32x4 sc, f32x4 sh, dasr_tensor y, int accumulate) {
    const long long total = (long long)N * H * W;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int n = i / ((long long)H * W);
    const long long p = i - (long long)n * H * W;
    f32x4 v = *(const f32x4*)((const float*)x.p + (size_t)n * x.n_stride + (size_t)p * 16) * sc + sh;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (j >= C) v[j] = 0.f;
    TO* o = (TO*)y.p + (size_t)n * y.n_stride + (size_t)p * 16;
    if (SPLIT) {   // split 16-bit tensor of one plane pair: hi in plane 0, the remainder in plane 1
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float vv = j < 4 ? v[j] : 0.f;
            const TO h = (TO)vv;
            o[j] = h;
            o[y.cb_stride + j] = (TO)(vv - (float)h);
        }
        return;
    }
    if (accumulate) {
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (TO)((float)o[j] + v[j]);
    } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) o[j] = (TO)(j < 4 ? v[j] : 0.f);
    }
}

// F.interpolate(mode='bilinear', align_corners=False) of an NCHW [N][1][h][w] map by an integer factor
__global__ void bilinear_up_kernel(const float* __restrict__ src, int N, int h, int w, int f, float* __restrict__ dst) {
    const int H = h * f, W = w * f;
    const long long total = (long long)N * H * W;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int x = i % W;
    long long t = i / W;
    const int y = t % H;
    const int n = t / H;
    const float inv = 1.f / f;
    float sy = (y + 0.5f) * inv - 0.5f, sx = (x + 0.5f) * inv - 0.5f;
    sy = sy < 0.f ? 0.f : sy;
    sx = sx < 0.f ? 0.f : sx;
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + (y0 < h - 1), x1 = x0 + (x0 < w - 1);
    const float ly = sy - y0, lx = sx - x0;
    const float* s = src + (size_t)n * h * w;
    dst[i] = (1.f - ly) * ((1.f - lx) * s[y0 * w + x0] + lx * s[y0 * w + x1]) + ly * ((1.f - lx) * s[y1 * w + x0] + lx * s[y1 * w + x1]);
}


// ---------------------------------------------------------------------------------------------------
// DSN kernels (codes/DSN): -log losses on sigmoid(logits), sigmoid backward, PReLU slope gradient,
// "valid" (un-padded) low-pass of the colour loss and its adjoint.
// ---------------------------------------------------------------------------------------------------
// mode 0: l = -log(p + eps); mode 1: l = -log(1 - p + eps), p = sigmoid(x) on channel 0.
// loss_acc += coef*sum(l); score_acc += score_coef*sum(p); grad (+)= gcoef * dl/dx
__global__ void logloss_kernel(dasr_tensor x, int N, int H, int W, int mode, float eps, float coef, float gcoef, float* loss_acc,
                               float* score_acc, float score_coef, dasr_tensor grad, int accumulate, dasr_red rs) {
    __shared__ float red[4];
    const long long total = (long long)N * H * W;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    float l = 0.f, sc = 0.f;
    if (i < total) {
        const int n = i / ((long long)H * W);
        const long long pix = i - (long long)n * H * W;
        const float v = ((const float*)x.p)[(size_t)n * x.n_stride + (size_t)pix * 16];
        const float p = 1.f / (1.f + expf(-v));
        sc = p;
        const float dp = p * (1.f - p);
        float g;
        if (mode == 0) {
            l = -logf(p + eps);
            g = -dp / (p + eps);
        } else {
            l = -logf(1.f - p + eps);
            g = dp / (1.f - p + eps);
        }
        if (grad.p) {
            float* gp = (float*)grad.p + (size_t)n * grad.n_stride + (size_t)pix * 16;
            f32x4 o = {gcoef * g, 0.f, 0.f, 0.f};
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            if (accumulate) o += ((f32x4*)gp)[0];
            ((f32x4*)gp)[0] = o;
            if (!accumulate) { ((f32x4*)gp)[1] = z; ((f32x4*)gp)[2] = z; ((f32x4*)gp)[3] = z; }
        }
    }
    const float v[2] = {block_sum_256(l, red), block_sum_256(sc, red)};
    float* const acc[2] = {loss_acc, score_acc};
    const float cf[2] = {coef, score_coef};
    grid_sum_commit<2>(rs, v, acc, cf);   // fixed-order sums over the grid (common.h)
}

// gz = g * y * (1 - y) on C (<= 4) channels of plane 0 (y = sigmoid output of the generator, model.py:55)
__global__ void sigmoid_bwd_kernel(dasr_tensor y, dasr_tensor g, int N, int C, int H, int W, dasr_tensor gz) {
    const long long total = (long long)N * H * W;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int n = i / ((long long)H * W);
    const long long p = i - (long long)n * H * W;
    const f32x4 yv = *(const f32x4*)((const float*)y.p + (size_t)n * y.n_stride + (size_t)p * 16);
    const f32x4 gv = *(const f32x4*)((const float*)g.p + (size_t)n * g.n_stride + (size_t)p * 16);
    f32x4 o = gv * yv * (1.f - yv);
    for (int j = 0; j < 4; ++j)
        if (j >= C) o[j] = 0.f;
    float* op = (float*)gz.p + (size_t)n * gz.n_stride + (size_t)p * 16;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    ((f32x4*)op)[0] = o; ((f32x4*)op)[1] = z; ((f32x4*)op)[2] = z; ((f32x4*)op)[3] = z;
}

// y = sigmoid(x) on C (<= 4) channels of plane 0 (the FSD discriminator's output map at inference, codes/DSN/model.py:104-106)
__global__ void sigmoid_fwd_kernel(dasr_tensor x, int N, int C, int H, int W, dasr_tensor y) {
    const long long total = (long long)N * H * W;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int n = i / ((long long)H * W);
    const long long p = i - (long long)n * H * W;
    const f32x4 v = *(const f32x4*)((const float*)x.p + (size_t)n * x.n_stride + (size_t)p * 16);
    f32x4 o;
    for (int j = 0; j < 4; ++j) o[j] = j < C ? 1.f / (1.f + expf(-v[j])) : 0.f;
    float* op = (float*)y.p + (size_t)n * y.n_stride + (size_t)p * 16;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    ((f32x4*)op)[0] = o; ((f32x4*)op)[1] = z; ((f32x4*)op)[2] = z; ((f32x4*)op)[3] = z;
}

// nn.PReLU() with one shared slope a: y = x > 0 ? x : a x.  Given y and gx = dL/dx (already masked by PReLU'),
// dL/da = sum_{y <= 0} (gx / a) * (y / a).  Two deterministic stages: per-block partials, then one block.
template <typename T>   // float: f32 tensors; f16_t: the f16 shadows of the DSN generator's backward (gx pre-scaled, undone through `scale`)
__global__ void prelu_grad_partial_kernel(dasr_tensor y, dasr_tensor gx, int N, int C, int H, int W, float* __restrict__ partial) {
    __shared__ float red[4];
    const int ncb = (C + 15) >> 4;
    const long long per = (long long)H * W * 4;
    const long long total = (long long)N * ncb * per;
    double s = 0.0;   // the terms cancel (a slope gradient is a small difference of large sums): per-thread sums in double keep the result independent
                      // of how the batch is split over launches / ranks to ~1e-6 instead of ~1e-3
    for (long long gi = (long long)blockIdx.x * blockDim.x + threadIdx.x; gi < total; gi += (long long)gridDim.x * blockDim.x) {
        const long long e = (gi % per) * 4;
        long long t = gi / per;
        const int cb = t % ncb;
        const int n = t / ncb;
        typedef T T4 __attribute__((ext_vector_type(4)));
        const T4 yv = *(const T4*)((const T*)y.p + (size_t)n * y.n_stride + (size_t)cb * y.cb_stride + e);
        const T4 gv = *(const T4*)((const T*)gx.p + (size_t)n * gx.n_stride + (size_t)cb * gx.cb_stride + e);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if ((float)yv[j] <= 0.f) s += (double)((float)gv[j] * (float)yv[j]);
    }
    const float tot = block_sum_256((float)s, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

__global__ __launch_bounds__(256) void prelu_grad_final_kernel(const float* __restrict__ partial, int nblocks, const float* __restrict__ slope,
                                                               float* __restrict__ dst, float scale) {
    __shared__ float red[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 256) s += (double)partial[i];   // fixed order per thread, fixed tree across threads: deterministic
    const float tot = block_sum_256((float)s, red);
    if (threadIdx.x == 0) {
        const float a = *slope;
        *dst = scale * tot / (a * a);
    }
}

// the second stage alone, for `count` slopes at once (workgroup k: partials [k * stride, k * stride + nblocks), slope *slopes[k], result -> *dsts[k]): the partials come from the
// data-gradient conv epilogues (dasr_conv_params::prelu_part) of the DSN generator's residual blocks -- one launch for all of them
__global__ __launch_bounds__(256) void prelu_final_multi_kernel(const float* __restrict__ partial, int nblocks, long long stride, const float* const* __restrict__ slopes,
                                                                float* const* __restrict__ dsts, float scale) {
    __shared__ float red[4];
    const float* pp = partial + (size_t)blockIdx.x * stride;
    double s = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 256) s += (double)pp[i];   // fixed order per thread, fixed tree across threads: deterministic
    const float tot = block_sum_256((float)s, red);
    if (threadIdx.x == 0) {
        const float a = *slopes[blockIdx.x];
        *dsts[blockIdx.x] = scale * tot / (a * a);
    }
}

// un-padded ("valid") depthwise k x k low-pass on C (<= 4) channels: out is (H-k+1) x (W-k+1) (FilterLow(padding=False),
// codes/DSN/loss.py:52-56).  mode 0 forward; mode 1 adjoint (gx (+)= sum_q w * g_low[q]).
__global__ void lowpass_valid_kernel(dasr_tensor x, const float* __restrict__ w, int k, int N, int C, int H, int W, int mode, dasr_tensor out,
                                     int accumulate) {
    const int Ho = H - k + 1, Wo = W - k + 1;
    const int OH = mode == 0 ? Ho : H, OW = mode == 0 ? Wo : W;
    const long long total = (long long)N * OH * OW;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int xx = i % OW;
    long long t = i / OW;
    const int yy = t % OH;
    const int n = t / OH;
    const float* xp = (const float*)x.p + (size_t)n * x.n_stride;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (mode == 0) {
        for (int ky = 0; ky < k; ++ky)
            for (int kx = 0; kx < k; ++kx) acc += *(const f32x4*)(xp + ((size_t)(yy + ky) * W + xx + kx) * 16) * w[ky * k + kx];
    } else {  // x = g_low with dims Ho x Wo; output pixel (yy, xx) of the full image
        for (int ky = 0; ky < k; ++ky) {
            const int qy = yy - ky;
            if (qy < 0 || qy >= Ho) continue;
            for (int kx = 0; kx < k; ++kx) {
                const int qx = xx - kx;
                if (qx < 0 || qx >= Wo) continue;
                acc += *(const f32x4*)(xp + ((size_t)qy * Wo + qx) * 16) * w[ky * k + kx];
            }
        }
    }
    for (int j = 0; j < 4; ++j)
        if (j >= C) acc[j] = 0.f;
    float* op = (float*)out.p + (size_t)n * out.n_stride + ((size_t)yy * OW + xx) * 16;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    if (accumulate) acc += ((f32x4*)op)[0];
    ((f32x4*)op)[0] = acc;
    if (!accumulate) { ((f32x4*)op)[1] = z; ((f32x4*)op)[2] = z; ((f32x4*)op)[3] = z; }
}

}  // namespace

extern "C" int dasr_inorm_lrelu_fwd(dasr_tensor x, int32_t N, int32_t C, int32_t H, int32_t W, float eps, float slope, dasr_tensor y,
                                    float* stats, void* stream) {
    if (N <= 0 || C <= 0) return DASR_EINVAL;
    DASR_LAUNCH(inorm_lrelu_fwd_kernel, dim3(N * ((C + 15) / 16)), dim3(256), 0, as_stream(stream), x, C, H, W, eps, slope, y, stats);
    return (int)hipGetLastError();
}

extern "C" int dasr_inorm_lrelu_bwd(dasr_tensor a, dasr_tensor ga, int32_t N, int32_t C, int32_t H, int32_t W, float slope,
                                    const float* stats, dasr_tensor gx, void* stream) {
    if (N <= 0 || C <= 0) return DASR_EINVAL;
    DASR_LAUNCH(inorm_lrelu_bwd_kernel, dim3(N * ((C + 15) / 16)), dim3(256), 0, as_stream(stream), a, ga, C, H, W, slope, stats, gx);
    return (int)hipGetLastError();
}

extern "C" int dasr_inorm_lrelu_jvp(dasr_tensor a, dasr_tensor t, int32_t N, int32_t C, int32_t H, int32_t W, float slope, const float* stats, dasr_tensor out,
                                    void* stream) {
    if (N <= 0 || C <= 0 || !a.p || !t.p || !out.p || !stats) return DASR_EINVAL;
    DASR_LAUNCH(inorm_lrelu_jvp_kernel, dim3(N * ((C + 15) / 16)), dim3(256), 0, as_stream(stream), a, t, C, H, W, slope, stats, out);
    return (int)hipGetLastError();
}

extern "C" int dasr_inorm_second(dasr_tensor a, dasr_tensor t, dasr_tensor ga, int32_t N, int32_t C, int32_t H, int32_t W, float slope, const float* stats,
                                 dasr_tensor out, int32_t accumulate, void* stream) {
    if (N <= 0 || C <= 0 || !a.p || !t.p || !ga.p || !out.p || !stats) return DASR_EINVAL;
    DASR_LAUNCH(inorm_second_kernel, dim3(N * ((C + 15) / 16)), dim3(256), 0, as_stream(stream), a, t, ga, C, H, W, slope, stats, out, accumulate);
    return (int)hipGetLastError();
}

extern "C" int dasr_grad_penalty(dasr_tensor g, int32_t N, int32_t C, int32_t H, int32_t W, float weight, float* part256, float* out3, float* loss_acc,
                                 int32_t stage, int32_t world, void* stream) {
    if (N <= 0 || C <= 0 || C > 16 || H <= 0 || W <= 0 || !g.p || !part256 || !out3 || stage < 0 || stage > 2 || world < 1 || (stage == 0 && world != 1)) return DASR_EINVAL;
    if (stage != 2) DASR_LAUNCH(sumsq_partial_kernel, dim3(256), dim3(256), 0, as_stream(stream), g, N, C, H, W, part256);
    DASR_LAUNCH(gp_finalize_kernel, dim3(1), dim3(64), 0, as_stream(stream), part256, 256, weight, out3, stage == 1 ? nullptr : loss_acc, stage, 1.f / (float)world);
    return (int)hipGetLastError();
}

extern "C" int dasr_fill_scaled(dasr_tensor x, int32_t N, int32_t C, int32_t H, int32_t W, const float* scalar, float factor, void* stream) {
    const long long total = (long long)N * ((C + 15) / 16) * H * W * 16;
    if (total <= 0 || !x.p || !scalar) return DASR_EINVAL;
    DASR_LAUNCH(fill_scaled_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), x, N, C, H, W, scalar, factor);
    return (int)hipGetLastError();
}

extern "C" int dasr_bnorm_lrelu_fwd(dasr_tensor x, int32_t N, int32_t C, int32_t H, int32_t W, int32_t group, float eps, float slope, const float* gamma,
                                    const float* beta, dasr_tensor y, float* stats, void* stream) {
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || group <= 0 || !gamma || !beta || !stats) return DASR_EINVAL;
    DASR_LAUNCH(bnorm_lrelu_fwd_kernel, dim3((C + 15) / 16), dim3(256), 0, as_stream(stream), x, N, C, H, W, group, eps, slope, gamma, beta, y, stats);
    return (int)hipGetLastError();
}

extern "C" int dasr_bnorm_lrelu_bwd(dasr_tensor x, dasr_tensor ga, int32_t N, int32_t C, int32_t H, int32_t W, int32_t group, float slope,
                                    const float* gamma, const float* beta, const float* stats, dasr_tensor gx, float* dgamma, float* dbeta,
                                    float pscale, void* stream) {
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || group <= 0 || !gamma || !beta || !stats || (dgamma != nullptr) != (dbeta != nullptr)) return DASR_EINVAL;
    DASR_LAUNCH(bnorm_lrelu_bwd_kernel, dim3((C + 15) / 16), dim3(256), 0, as_stream(stream), x, ga, N, C, H, W, group, slope, gamma, beta, stats, gx,
                dgamma, dbeta, pscale);
    return (int)hipGetLastError();
}

extern "C" int dasr_bnorm_lrelu_jvp(dasr_tensor x, dasr_tensor t, int32_t N, int32_t C, int32_t H, int32_t W, int32_t group, float slope, const float* gamma,
                                    const float* beta, const float* stats, dasr_tensor out, void* stream) {
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || group <= 0 || !gamma || !beta || !stats) return DASR_EINVAL;
    DASR_LAUNCH(bnorm_lrelu_jvp_kernel, dim3((C + 15) / 16), dim3(256), 0, as_stream(stream), x, t, N, C, H, W, group, slope, gamma, beta, stats, out);
    return (int)hipGetLastError();
}

extern "C" int dasr_bnorm_second(dasr_tensor x, dasr_tensor t, dasr_tensor ga, int32_t N, int32_t C, int32_t H, int32_t W, int32_t group, float slope,
                                 const float* gamma, const float* beta, const float* stats, dasr_tensor out, int32_t accumulate, float* dgamma, float pscale,
                                 void* stream) {
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || group <= 0 || !gamma || !beta || !stats) return DASR_EINVAL;
    DASR_LAUNCH(bnorm_second_kernel, dim3((C + 15) / 16), dim3(256), 0, as_stream(stream), x, t, ga, N, C, H, W, group, slope, gamma, beta, stats, out,
                accumulate, dgamma, pscale);
    return (int)hipGetLastError();
}

extern "C" int dasr_bnorm_running(const float* stats, int32_t g, int32_t C, int32_t count, float momentum, float* running_mean, float* running_var,
                                  float* num_batches_tracked, void* stream) {
    if (!stats || g < 0 || C <= 0 || count <= 0 || !running_mean || !running_var) return DASR_EINVAL;
    DASR_LAUNCH(bnorm_running_kernel, dim3(nblk(C)), dim3(256), 0, as_stream(stream), stats, g, C, count, momentum, running_mean, running_var,
                num_batches_tracked);
    return (int)hipGetLastError();
}

extern "C" int dasr_gan_loss(dasr_tensor x, int32_t N, int32_t C, int32_t H, int32_t W, int32_t gan_type, float target, float coef, float gcoef,
                             float* loss_acc, float* score_acc, float score_coef, dasr_tensor grad, void* stream) {
    const long long total = (long long)N * H * W;
    if (total <= 0 || C > 16 || gan_type < 0 || gan_type > 2) return DASR_EINVAL;
    const dim3 g(nblk(total)), b(256);
    const void* key = loss_acc ? (const void*)loss_acc : (const void*)score_acc;
    const dasr_red rs = dasr_red_scratch(key, as_stream(stream), g.x, 2);
    if (key && !rs.part) return dasr_red_error();
    if (gan_type == 0)
        DASR_LAUNCH(gan_loss_kernel<0>, g, b, 0, as_stream(stream), x, N, C, H, W, target, coef, gcoef, loss_acc, score_acc, score_coef, grad, rs);
    else if (gan_type == 1)
        DASR_LAUNCH(gan_loss_kernel<1>, g, b, 0, as_stream(stream), x, N, C, H, W, target, coef, gcoef, loss_acc, score_acc, score_coef, grad, rs);
    else
        DASR_LAUNCH(gan_loss_kernel<2>, g, b, 0, as_stream(stream), x, N, C, H, W, target, coef, gcoef, loss_acc, score_acc, score_coef, grad, rs);
    return (int)hipGetLastError();
}

extern "C" int dasr_bce_logits(dasr_tensor x, int32_t N, int32_t C, int32_t H, int32_t W, float target, float coef, float gcoef,
                               float* loss_acc, float* score_acc, float score_coef, dasr_tensor grad, void* stream) {
    return dasr_gan_loss(x, N, C, H, W, 0, target, coef, gcoef, loss_acc, score_acc, score_coef, grad, stream);
}

extern "C" int dasr_ragan(dasr_tensor a, dasr_tensor b, int32_t N, int32_t H, int32_t W, int32_t stage, int32_t n_glob, int32_t form, float ta, float tb,
                          float coef, float gcoef, float eps, float* sums, float* part, float* loss_acc, float* score_a, float* score_b,
                          float score_coef, dasr_tensor ga, dasr_tensor gb, void* stream) {
    if (N <= 0 || H <= 0 || W <= 0 || n_glob < N || stage < 0 || stage > 2 || form < 0 || form > 3 || !a.p || !b.p || !sums || (stage > 0 && !part))
        return DASR_EINVAL;
    const dim3 g(nblk((long long)H * W)), blk(256);
    const void* key = stage != 1 ? nullptr : (loss_acc ? (const void*)loss_acc : (score_a ? (const void*)score_a : (const void*)score_b));
    const dasr_red rs = dasr_red_scratch(key, as_stream(stream), g.x, 3);
    if (key && !rs.part) return dasr_red_error();
#define DASR_RAGAN_FORM(F)                                                                                                                        \
    DASR_LAUNCH(ragan_kernel<F>, g, blk, 0, as_stream(stream), a, b, N, H, W, stage, 1.f / (float)n_glob, ta, tb, coef, gcoef, eps, sums, part, \
                loss_acc, score_a, score_b, score_coef, ga, gb, rs)
    switch (form) {
        case 0: DASR_RAGAN_FORM(0); break;
        case 1: DASR_RAGAN_FORM(1); break;
        case 2: DASR_RAGAN_FORM(2); break;
        default: DASR_RAGAN_FORM(3); break;
    }
#undef DASR_RAGAN_FORM
    return (int)hipGetLastError();
}

extern "C" int dasr_dwt_fwd(dasr_tensor x, int32_t N, int32_t C, int32_t H2, int32_t W2, int32_t norm, dasr_tensor ll, dasr_tensor hc,
                            void* stream) {
    const long long total = (long long)N * H2 * W2;
    if (total <= 0 || C > 5) return DASR_EINVAL;
    DASR_LAUNCH(dwt_fwd_kernel, dim3(nblk(total)), dim3(256), 0, as_stream(stream), x, N, C, H2, W2, norm, ll, hc);
    return (int)hipGetLastError();
}

extern "C" int dasr_dwt_bwd(dasr_tensor gll, dasr_tensor ghc, int32_t N, int32_t C, int32_t H2, int32_t W2, int32_t norm, dasr_tensor gx,
                            int32_t accumulate, void* stream) {
    const long long total = (long long)N * H2 * W2;
    if (total <= 0 || C > 5) return DASR_EINVAL;
    DASR_LAUNCH(dwt_bwd_kernel, dim3(nblk(total)), dim3(256), 0, as_stream(stream), gll, ghc, N, C, H2, W2, norm, gx, accumulate);
    return (int)hipGetLastError();
}

extern "C" int dasr_lowpass(dasr_tensor x, dasr_tensor x2, const float* w, int32_t k, int32_t N, int32_t C, int32_t H, int32_t W,
                            int32_t mode, float a_h, float b_h, dasr_tensor out_low, dasr_tensor out_high, int32_t accumulate, void* stream) {
    const long long total = (long long)N * H * W;
    if (total <= 0 || C > 4 || !(k & 1)) return DASR_EINVAL;
    DASR_LAUNCH(lowpass_kernel, dim3(nblk(total)), dim3(256), 0, as_stream(stream), x, x2, w, k, N, C, H, W, mode, a_h, b_h, out_low,
                       out_high, accumulate);
    return (int)hipGetLastError();
}

extern "C" int dasr_maxpool2(dasr_tensor x, int32_t is_f32, int32_t N, int32_t C, int32_t Ho, int32_t Wo, dasr_tensor y, int32_t Win, void* stream) {
    if (Win != 0 && Win != 2 * Wo && Win != 2 * Wo + 1) return DASR_EINVAL;
    const long long total = (long long)N * ((C + 15) / 16) * Ho * Wo * 4;
    if (total <= 0) return DASR_EINVAL;
    if (is_f32 == 3) DASR_LAUNCH(maxpool_fwd_split_kernel<f16_t>, dim3(nblk(total)), dim3(256), 0, as_stream(stream), x, N, C, Ho, Wo, y, Win);
    else if (is_f32 == 4) DASR_LAUNCH(maxpool_fwd_split_kernel<bf16_t>, dim3(nblk(total)), dim3(256), 0, as_stream(stream), x, N, C, Ho, Wo, y, Win);
    else if (is_f32 == 2) DASR_LAUNCH(maxpool_fwd_kernel<f16_t>, dim3(nblk(total)), dim3(256), 0, as_stream(stream), x, N, C, Ho, Wo, y, Win);
    else if (is_f32) DASR_LAUNCH(maxpool_fwd_kernel<float>, dim3(nblk(total)), dim3(256), 0, as_stream(stream), x, N, C, Ho, Wo, y, Win);
    else DASR_LAUNCH(maxpool_fwd_kernel<bf16_t>, dim3(nblk(total)), dim3(256), 0, as_stream(stream), x, N, C, Ho, Wo, y, Win);
    return (int)hipGetLastError();
}

extern "C" int dasr_maxpool2_bwd(dasr_tensor x, dasr_tensor gy, int32_t is_f32, int32_t N, int32_t C, int32_t Ho, int32_t Wo, dasr_tensor gx,
                                 int32_t relu_mask, int32_t Win, void* stream) {
    if (Win != 0 && Win != 2 * Wo && Win != 2 * Wo + 1) return DASR_EINVAL;
    const long long total = (long long)N * ((C + 15) / 16) * Ho * Wo * 4;
    if (total <= 0) return DASR_EINVAL;
    if (is_f32 == 3) DASR_LAUNCH((maxpool_bwd_split_kernel<f16_t, true>), dim3(nblk(total)), dim3(256), 0, as_stream(stream), x, gy, N, C, Ho, Wo, gx, relu_mask, Win);
    else if (is_f32 == 4) DASR_LAUNCH((maxpool_bwd_split_kernel<bf16_t, true>), dim3(nblk(total)), dim3(256), 0, as_stream(stream), x, gy, N, C, Ho, Wo, gx, relu_mask, Win);
    else if (is_f32 == 5) DASR_LAUNCH((maxpool_bwd_split_kernel<f16_t, false>), dim3(nblk(total)), dim3(256), 0, as_stream(stream), x, gy, N, C, Ho, Wo, gx, relu_mask, Win);
    else if (is_f32 == 2) DASR_LAUNCH(maxpool_bwd_kernel<f16_t>, dim3(nblk(total)), dim3(256), 0, as_stream(stream), x, gy, N, C, Ho, Wo, gx, relu_mask, Win);
    else if (is_f32) DASR_LAUNCH(maxpool_bwd_kernel<float>, dim3(nblk(total)), dim3(256), 0, as_stream(stream), x, gy, N, C, Ho, Wo, gx, relu_mask, Win);
    else DASR_LAUNCH(maxpool_bwd_kernel<bf16_t>, dim3(nblk(total)), dim3(256), 0, as_stream(stream), x, gy, N, C, Ho, Wo, gx, relu_mask, Win);
    return (int)hipGetLastError();
}

extern "C" int dasr_l1_diff(dasr_tensor a, dasr_tensor b, int32_t is_f32, int32_t N, int32_t C, int32_t H, int32_t W, float coef, float gcoef,
                            float* loss_acc, dasr_tensor ga, void* stream) {
    const int squared = is_f32 >> 1;  // bit 1 of the dtype flag selects the squared (MSE) form
    is_f32 &= 1;
    const long long total = (long long)N * ((C + 15) / 16) * H * W * 4;
    if (total <= 0) return DASR_EINVAL;
    const dasr_red rs = dasr_red_scratch(loss_acc, as_stream(stream), nblk(total), 1);
    if (loss_acc && !rs.part) return dasr_red_error();
    if (is_f32) DASR_LAUNCH(l1_diff_kernel<float>, dim3(nblk(total)), dim3(256), 0, as_stream(stream), a, b, N, C, H, W, coef, gcoef, loss_acc, ga, squared, rs);
    else DASR_LAUNCH(l1_diff_kernel<bf16_t>, dim3(nblk(total)), dim3(256), 0, as_stream(stream), a, b, N, C, H, W, coef, gcoef, loss_acc, ga, squared, rs);
    return (int)hipGetLastError();
}

extern "C" int dasr_affine4(dasr_tensor x, int32_t N, int32_t C, int32_t H, int32_t W, const float* scale4, const float* shift4, dasr_tensor y,
                            int32_t y_f32, int32_t accumulate, void* stream) {
    const long long total = (long long)N * H * W;
    if (total <= 0 || C > 4) return DASR_EINVAL;
    const f32x4 sc = {scale4[0], scale4[1], scale4[2], scale4[3]}, sh = {shift4[0], shift4[1], shift4[2], shift4[3]};
    if (y_f32 == 3 && !accumulate) DASR_LAUNCH((affine4_kernel<f16_t, true>), dim3(nblk(total)), dim3(256), 0, as_stream(stream), x, N, C, H, W, sc, sh, y, accumulate);
    else if (y_f32 == 3) return DASR_EINVAL;
    else if (y_f32 == 2) DASR_LAUNCH(affine4_kernel<f16_t>, dim3(nblk(total)), dim3(256), 0, as_stream(stream), x, N, C, H, W, sc, sh, y, accumulate);
    else if (y_f32) DASR_LAUNCH(affine4_kernel<float>, dim3(nblk(total)), dim3(256), 0, as_stream(stream), x, N, C, H, W, sc, sh, y, accumulate);
    else DASR_LAUNCH(affine4_kernel<bf16_t>, dim3(nblk(total)), dim3(256), 0, as_stream(stream), x, N, C, H, W, sc, sh, y, accumulate);
    return (int)hipGetLastError();
}

extern "C" int dasr_bilinear_up(const float* src, int32_t N, int32_t h, int32_t w, int32_t factor, float* dst, void* stream) {
    const long long total = (long long)N * h * w * factor * factor;
    if (total <= 0) return DASR_EINVAL;
    DASR_LAUNCH(bilinear_up_kernel, dim3(nblk(total)), dim3(256), 0, as_stream(stream), src, N, h, w, factor, dst);
    return (int)hipGetLastError();
}

// Domain-distance map for any discriminator conv table (codes/DSN/receptive_cal.py:34-60, create_dataset_modified.py:14-24): every D output
// value (i, j) is added over its receptive-field window [lo(i), hi(i)) x [lo(j), hi(j)), lo(k) = int(max(0, start + k*jump - rf/2)),
// hi(k) = int(start + k*jump + rf - rf/2), and the sum is divided by the number of windows covering the pixel (0/0 = NaN where none does, as in
// the reference).  Gather form: one thread per map pixel sums the D values whose window contains it.  (jump, rf, start) come from the walk
// over the WIDTH for both axes, as the reference has it.  FSD ([5,1,2] x 4) is the 17 x 17 count-normalised box of dasr_lowpass; this kernel
// serves nld_s1 / nld_s2.
__global__ void ddm_spread_kernel(dasr_tensor d, int N, int n_h, int n_w, int H, int W, int jump, int rf, float start, dasr_tensor out) {
    const long long total = (long long)N * H * W;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int x = t % W, y = (t / W) % H, n = t / ((long long)W * H);
    const int half = rf / 2;
    auto range = [&](int p, int cnt, int& k0, int& k1) {   // candidate index range, then the exact window test
        k0 = max(0, (int)floorf(((float)p - (float)(rf - half) - start) / (float)jump) - 1);
        k1 = min(cnt - 1, (int)ceilf(((float)p + (float)half - start) / (float)jump) + 1);
    };
    auto covers = [&](int k, int p) {
        const float c = start + (float)k * (float)jump;
        return p >= (int)fmaxf(0.f, c - (float)half) && p < (int)(c + (float)(rf - half));
    };
    int i0, i1, j0, j1;
    range(y, n_h, i0, i1);
    range(x, n_w, j0, j1);
    const float* dp = (const float*)d.p + (size_t)n * d.n_stride;
    float sum = 0.f;
    int ci = 0, cj = 0;
    for (int j = j0; j <= j1; ++j) cj += covers(j, x) ? 1 : 0;
    for (int i = i0; i <= i1; ++i) {
        if (!covers(i, y)) continue;
        ++ci;
        for (int j = j0; j <= j1; ++j)
            if (covers(j, x)) sum += dp[((size_t)i * n_w + j) * 16];
    }
    float* op = (float*)out.p + (size_t)n * out.n_stride + ((size_t)y * W + x) * 16;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    ((f32x4*)op)[0] = f32x4{sum / (float)(ci * cj), 0.f, 0.f, 0.f};
    ((f32x4*)op)[1] = z;
    ((f32x4*)op)[2] = z;
    ((f32x4*)op)[3] = z;
}

extern "C" int dasr_ddm_spread(dasr_tensor d, int32_t N, int32_t n_h, int32_t n_w, int32_t H, int32_t W, int32_t jump, int32_t rf, float start,
                               dasr_tensor out, void* stream) {
    const long long total = (long long)N * H * W;
    if (total <= 0 || n_h <= 0 || n_w <= 0 || jump <= 0 || rf <= 0 || !d.p || !out.p) return DASR_EINVAL;
    DASR_LAUNCH(ddm_spread_kernel, dim3(nblk(total)), dim3(256), 0, as_stream(stream), d, N, n_h, n_w, H, W, jump, rf, start, out);
    return (int)hipGetLastError();
}

extern "C" int dasr_logloss(dasr_tensor x, int32_t N, int32_t H, int32_t W, int32_t mode, float eps, float coef, float gcoef, float* loss_acc,
                            float* score_acc, float score_coef, dasr_tensor grad, int32_t accumulate, void* stream) {
    const long long total = (long long)N * H * W;
    if (total <= 0) return DASR_EINVAL;
    const void* key = loss_acc ? (const void*)loss_acc : (const void*)score_acc;
    const dasr_red rs = dasr_red_scratch(key, as_stream(stream), nblk(total), 2);
    if (key && !rs.part) return dasr_red_error();
    DASR_LAUNCH(logloss_kernel, dim3(nblk(total)), dim3(256), 0, as_stream(stream), x, N, H, W, mode, eps, coef, gcoef, loss_acc, score_acc,
                       score_coef, grad, accumulate, rs);
    return (int)hipGetLastError();
}

extern "C" int dasr_sigmoid_bwd(dasr_tensor y, dasr_tensor g, int32_t N, int32_t C, int32_t H, int32_t W, dasr_tensor gz, void* stream) {
    const long long total = (long long)N * H * W;
    if (total <= 0 || C > 4) return DASR_EINVAL;
    DASR_LAUNCH(sigmoid_bwd_kernel, dim3(nblk(total)), dim3(256), 0, as_stream(stream), y, g, N, C, H, W, gz);
    return (int)hipGetLastError();
}

extern "C" int dasr_sigmoid_fwd(dasr_tensor x, int32_t N, int32_t C, int32_t H, int32_t W, dasr_tensor y, void* stream) {
    const long long total = (long long)N * H * W;
    if (total <= 0 || C > 4) return DASR_EINVAL;
    DASR_LAUNCH(sigmoid_fwd_kernel, dim3(nblk(total)), dim3(256), 0, as_stream(stream), x, N, C, H, W, y);
    return (int)hipGetLastError();
}

extern "C" int dasr_prelu_grad(dasr_tensor y, dasr_tensor gx, int32_t N, int32_t C, int32_t H, int32_t W, const float* slope, float* scratch256,
                               float* dst, float scale, void* stream) {
    if ((long long)N * C * H * W <= 0) return DASR_EINVAL;
    // 1024 workgroups (4 per CU: the streaming read of two tensors needs the occupancy), one partial each; `scratch256` holds 1024 floats
    const long long vec = (long long)N * ((C + 15) / 16) * H * W * 4;
    const int nb = (int)(vec < 1024LL * 256 ? (vec + 255) / 256 : 1024);
    DASR_LAUNCH(prelu_grad_partial_kernel<float>, dim3(nb), dim3(256), 0, as_stream(stream), y, gx, N, C, H, W, scratch256);
    DASR_LAUNCH(prelu_grad_final_kernel, dim3(1), dim3(256), 0, as_stream(stream), scratch256, nb, slope, dst, scale);
    return (int)hipGetLastError();
}

extern "C" int dasr_prelu_final(const float* partial, int32_t nblocks, int64_t stride, int32_t count, const float* const* slopes, float* const* dsts, float scale,
                                void* stream) {
    if (!partial || nblocks <= 0 || stride < nblocks || count <= 0 || !slopes || !dsts) return DASR_EINVAL;
    DASR_LAUNCH(prelu_final_multi_kernel, dim3(count), dim3(256), 0, as_stream(stream), partial, nblocks, (long long)stride, slopes, dsts, scale);
    return (int)hipGetLastError();
}

extern "C" int dasr_prelu_grad_f16(dasr_tensor y, dasr_tensor gx, int32_t N, int32_t C, int32_t H, int32_t W, const float* slope, float* scratch256,
                                   float* dst, float scale, void* stream) {
    if ((long long)N * C * H * W <= 0) return DASR_EINVAL;
    const long long vec = (long long)N * ((C + 15) / 16) * H * W * 4;
    const int nb = (int)(vec < 1024LL * 256 ? (vec + 255) / 256 : 1024);
    DASR_LAUNCH(prelu_grad_partial_kernel<f16_t>, dim3(nb), dim3(256), 0, as_stream(stream), y, gx, N, C, H, W, scratch256);
    DASR_LAUNCH(prelu_grad_final_kernel, dim3(1), dim3(256), 0, as_stream(stream), scratch256, nb, slope, dst, scale);
    return (int)hipGetLastError();
}

extern "C" int dasr_lowpass_valid(dasr_tensor x, const float* w, int32_t k, int32_t N, int32_t C, int32_t H, int32_t W, int32_t mode,
                                  dasr_tensor out, int32_t accumulate, void* stream) {
    const long long total = (long long)N * (mode == 0 ? (H - k + 1) * (long long)(W - k + 1) : (long long)H * W);
    if (total <= 0 || C > 4 || H < k || W < k) return DASR_EINVAL;
    DASR_LAUNCH(lowpass_valid_kernel, dim3(nblk(total)), dim3(256), 0, as_stream(stream), x, w, k, N, C, H, W, mode, out, accumulate);
    return (int)hipGetLastError();
}
