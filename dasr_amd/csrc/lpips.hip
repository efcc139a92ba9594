// HBM-bound kernels of the LPIPS(alex) perceptual loss (PerceptualLossLPIPS, codes/SRN/models/modules/loss.py:66-72 ->
// PNetLin.forward, codes/PerceptualSimilarity/models/networks_basic.py:64-92, backbone pretrained_networks.py:57-95), gfx950:
//  * input scaling + 4x4 space-to-depth: AlexNet's 11x11 / stride 4 / pad 2 first conv on 3 channels becomes a 3x3 / stride 1 / pad 0
//    conv on 48 channels of the (H+4)/4 grid, which runs on the MFMA conv kernel like every other layer (and its adjoint)
//  * MaxPool2d(3, 2) forward / backward (first maximum in scan order wins, like ATen; the ReLU' of the layer feeding the pool is applied)
//  * the per-layer head: channel-unit-normalise both feature stacks, squared difference, non-negative 1x1 "lin" weights, spatial mean;
//    the loss and its gradient w.r.t. the first stack in one launch
// All tensors are NC16HW16 f32 (dasr_hip.h).
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

// mode 0: y[n][c][Y][X][by*4+bx] = sc[c] * x[n][c][4Y+by-2][4X+bx-2] + sh[c] inside the image, 0 outside (the conv's zero padding is applied
// to the SCALED input); y has 3 planes (plane = colour), Hs x Ws = (H+4)/4 x (W+4)/4.
// mode 1 (adjoint): x[n][c][y][x] += sc[c] * y[n][c][(y+2)>>2][(x+2)>>2][((y+2)&3)*4 + ((x+2)&3)]  (x: plane 0 of a blocked tensor, channels 0..2)
// xf (DSN --lpips_rot_flip, codes/DSN/loss.py:149-168): the network sees T(x) instead of x, T = one of the 8 symmetries of the square:
// T(x)[i][j] = x[u][v] with (u, v) = (i, j), swapped if bit 0 (transpose), then u -> H-1-u if bit 1, v -> W-1-v if bit 2.  Mode 0 reads through
// that map; mode 1 routes the gradient of T(x) back to the pixel it came from.  Transposing forms need H == W.
__device__ __forceinline__ void dihedral_src(int i, int j, int H, int W, int xf, int& u, int& v) {
    u = (xf & 1) ? j : i;
    v = (xf & 1) ? i : j;
    if (xf & 2) u = H - 1 - u;
    if (xf & 4) v = W - 1 - v;
}
__device__ __forceinline__ void dihedral_dst(int u, int v, int H, int W, int xf, int& i, int& j) {   // inverse of dihedral_src
    if (xf & 2) u = H - 1 - u;
    if (xf & 4) v = W - 1 - v;
    i = (xf & 1) ? v : u;
    j = (xf & 1) ? u : v;
}

__global__ void lpips_s2d_kernel(dasr_tensor x, int N, int H, int W, f32x4 sc, f32x4 sh, dasr_tensor y, int mode, int xf) {
    const int Hs = (H + 4) >> 2, Ws = (W + 4) >> 2;
    if (mode == 0) {
        const long long total = (long long)N * 3 * Hs * Ws * 4;   // one thread: one colour, one s2d pixel, one row `by` of its 4x4 block
        const long long gi = (long long)blockIdx.x * blockDim.x + threadIdx.x;
        if (gi >= total) return;
        const int by = gi & 3;
        long long i = gi >> 2;
        const int X = i % Ws; i /= Ws;
        const int Y = i % Hs; i /= Hs;
        const int c = i % 3;
        const int n = i / 3;
        const float* xp = (const float*)x.p + (size_t)n * x.n_stride + c;
        const int sy = 4 * Y + by - 2;
        f32x4 v;
#pragma unroll
        for (int bx = 0; bx < 4; ++bx) {
            const int sx = 4 * X + bx - 2;
            const bool ok = (sy >= 0) & (sy < H) & (sx >= 0) & (sx < W);
            int u = sy, w_ = sx;
            if (xf) dihedral_src(sy, sx, H, W, xf, u, w_);
            v[bx] = ok ? xp[((size_t)u * W + w_) * 16] * sc[c] + sh[c] : 0.f;
        }
        *(f32x4*)((float*)y.p + (size_t)n * y.n_stride + (size_t)c * y.cb_stride + ((size_t)Y * Ws + X) * 16 + by * 4) = v;
    } else {
        const long long total = (long long)N * H * W;
        const long long gi = (long long)blockIdx.x * blockDim.x + threadIdx.x;
        if (gi >= total) return;
        const int n = gi / ((long long)H * W);
        const long long p = gi - (long long)n * H * W;
        int py = p / W, px = p - (long long)py * W;   // the pixel of x that receives the gradient; (py, px) below: where it sits in T(x)
        float* xp = (float*)x.p + (size_t)n * x.n_stride + (size_t)p * 16;
        if (xf) {
            int i, j;
            dihedral_dst(py, px, H, W, xf, i, j);
            py = i;
            px = j;
        }
        const int Y = (py + 2) >> 2, X = (px + 2) >> 2, b = ((py + 2) & 3) * 4 + ((px + 2) & 3);
        const float* yp = (const float*)y.p + (size_t)n * y.n_stride + ((size_t)Y * Ws + X) * 16 + b;
#pragma unroll
        for (int c = 0; c < 3; ++c) xp[c] += sc[c] * yp[(size_t)c * y.cb_stride];
    }
}

// MaxPool2d(kernel 3, stride 2, no padding): Ho = (H - 3) / 2 + 1
__global__ void maxpool3s2_fwd_kernel(dasr_tensor x, int N, int C, int H, int W, dasr_tensor y) {
    const int ncb = (C + 15) >> 4, Ho = (H - 3) / 2 + 1, Wo = (W - 3) / 2 + 1;
    const long long total = (long long)N * ncb * Ho * Wo * 4;
    const long long gi = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gi >= total) return;
    const int q = gi & 3;
    long long i = gi >> 2;
    const int xx = i % Wo; i /= Wo;
    const int yy = i % Ho; i /= Ho;
    const int cb = i % ncb;
    const int n = i / ncb;
    const float* xp = (const float*)x.p + (size_t)n * x.n_stride + (size_t)cb * x.cb_stride + q * 4;
    f32x4 m = {-3.4e38f, -3.4e38f, -3.4e38f, -3.4e38f};
#pragma unroll
    for (int d = 0; d < 9; ++d) {
        const f32x4 s = *(const f32x4*)(xp + ((size_t)(2 * yy + d / 3) * W + 2 * xx + d % 3) * 16);
#pragma unroll
        for (int j = 0; j < 4; ++j) m[j] = fmaxf(m[j], s[j]);
    }
    *(f32x4*)((float*)y.p + (size_t)n * y.n_stride + (size_t)cb * y.cb_stride + ((size_t)yy * Wo + xx) * 16 + q * 4) = m;
}

// gather form (deterministic, no atomics): input pixel (iy, ix) belongs to at most 2 x 2 windows; it receives gy of a window iff it is that
// window's FIRST maximum in scan order.  relu_mask: nothing flows into x <= 0 (ReLU' of the producing conv).  accumulate: gx += instead of =
__global__ void maxpool3s2_bwd_kernel(dasr_tensor x, dasr_tensor gy, int N, int C, int H, int W, dasr_tensor gx, int relu_mask, int accumulate) {
    const int ncb = (C + 15) >> 4, Ho = (H - 3) / 2 + 1, Wo = (W - 3) / 2 + 1;
    const long long total = (long long)N * ncb * H * W * 4;
    const long long gi = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gi >= total) return;
    const int q = gi & 3;
    long long i = gi >> 2;
    const int ix = i % W; i /= W;
    const int iy = i % H; i /= H;
    const int cb = i % ncb;
    const int n = i / ncb;
    const float* xp = (const float*)x.p + (size_t)n * x.n_stride + (size_t)cb * x.cb_stride + q * 4;
    const float* gp = (const float*)gy.p + (size_t)n * gy.n_stride + (size_t)cb * gy.cb_stride + q * 4;
    const f32x4 me = *(const f32x4*)(xp + ((size_t)iy * W + ix) * 16);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int oy_lo = iy >= 2 ? (iy - 1) >> 1 : 0, oy_hi = min(iy >> 1, Ho - 1);
    const int ox_lo = ix >= 2 ? (ix - 1) >> 1 : 0, ox_hi = min(ix >> 1, Wo - 1);
    for (int oy = oy_lo; oy <= oy_hi; ++oy)
        for (int ox = ox_lo; ox <= ox_hi; ++ox) {
            const int my = (iy - 2 * oy) * 3 + (ix - 2 * ox);   // my position in the window's scan order
            f32x4 m = {-3.4e38f, -3.4e38f, -3.4e38f, -3.4e38f};
            int am[4] = {0, 0, 0, 0};
#pragma unroll
            for (int d = 0; d < 9; ++d) {
                const f32x4 s = *(const f32x4*)(xp + ((size_t)(2 * oy + d / 3) * W + 2 * ox + d % 3) * 16);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (s[j] > m[j]) { m[j] = s[j]; am[j] = d; }
            }
            const f32x4 g = *(const f32x4*)(gp + ((size_t)oy * Wo + ox) * 16);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (am[j] == my) acc[j] += g[j];
        }
    if (relu_mask) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (!(me[j] > 0.f)) acc[j] = 0.f;
    }
    float* op = (float*)gx.p + (size_t)n * gx.n_stride + (size_t)cb * gx.cb_stride + ((size_t)iy * W + ix) * 16 + q * 4;
    if (accumulate) acc += *(const f32x4*)op;
    *(f32x4*)op = acc;
}

// One thread per pixel of image pair (n: f0 = f[n], f1 = f[n + pair_off]), three sweeps over the C channels:
//   s_k = sqrt(sum_c f_k^2) + eps;  u = f0 / s_0, v = f1 / s_1;  val = sum_c w_c (u_c - v_c)^2          (normalize_tensor, networks_basic.py:73-75)
//   loss_acc += coef * val          (coef = 1 / (n_pairs * H * W): spatial_average, then .mean() over the batch, loss.py:72)
//   g0_c = gcoef * [ a_c / s_0 - f0_c * (sum_k a_k f0_k) / (sqrt(sum f0^2) * s_0^2) ],  a_c = 2 w_c (u_c - v_c),  zero where f0_c <= 0 when relu_mask
// (the gradient handed on is w.r.t. the PRE-activation of the ReLU that produced f0, like the conv data-gradient epilogues expect)
__global__ void lpips_head_kernel(dasr_tensor f, long long pair_off, int N, int C, int H, int W, const float* __restrict__ lin, float eps, float coef,
                                  float gcoef, float* loss_acc, dasr_tensor g0, int relu_mask, dasr_red rs) {
    __shared__ float red[4];
    const int ncb = C >> 4;
    const long long total = (long long)N * H * W;
    const long long gi = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    float val = 0.f;
    if (gi < total) {
        const int n = gi / ((long long)H * W);
        const long long p = gi - (long long)n * H * W;
        const float* a = (const float*)f.p + (size_t)n * f.n_stride + (size_t)p * 16;
        const float* b = a + (size_t)pair_off * f.n_stride;
        float q0 = 0.f, q1 = 0.f;
        for (int cb = 0; cb < ncb; ++cb) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const f32x4 x0 = *(const f32x4*)(a + (size_t)cb * f.cb_stride + 4 * k), x1 = *(const f32x4*)(b + (size_t)cb * f.cb_stride + 4 * k);
#pragma unroll
                for (int j = 0; j < 4; ++j) { q0 += x0[j] * x0[j]; q1 += x1[j] * x1[j]; }
            }
        }
        const float r0 = sqrtf(q0), s0 = r0 + eps, s1 = sqrtf(q1) + eps;
        const float i0 = 1.f / s0, i1 = 1.f / s1;
        float dot = 0.f;
        for (int cb = 0; cb < ncb; ++cb) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const f32x4 x0 = *(const f32x4*)(a + (size_t)cb * f.cb_stride + 4 * k), x1 = *(const f32x4*)(b + (size_t)cb * f.cb_stride + 4 * k);
                const f32x4 w = *(const f32x4*)(lin + cb * 16 + 4 * k);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float d = x0[j] * i0 - x1[j] * i1;
                    val += w[j] * d * d;
                    dot += 2.f * w[j] * d * x0[j];
                }
            }
        }
        if (g0.p) {
            const float k2 = r0 > 0.f ? dot / (r0 * s0 * s0) : 0.f;
            float* go = (float*)g0.p + (size_t)n * g0.n_stride + (size_t)p * 16;
            for (int cb = 0; cb < ncb; ++cb) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const f32x4 x0 = *(const f32x4*)(a + (size_t)cb * f.cb_stride + 4 * k), x1 = *(const f32x4*)(b + (size_t)cb * f.cb_stride + 4 * k);
                    const f32x4 w = *(const f32x4*)(lin + cb * 16 + 4 * k);
                    f32x4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float d = x0[j] * i0 - x1[j] * i1;
                        o[j] = gcoef * (2.f * w[j] * d * i0 - x0[j] * k2);
                        if (relu_mask && !(x0[j] > 0.f)) o[j] = 0.f;
                    }
                    *(f32x4*)(go + (size_t)cb * g0.cb_stride + 4 * k) = o;
                }
            }
        }
    }
    const float v[1] = {block_sum_256(val, red)};
    float* const acc[1] = {loss_acc};
    const float cf[1] = {coef};
    grid_sum_commit<1>(rs, v, acc, cf);   // fixed-order sum over the grid (common.h)
}

}  // namespace

extern "C" int dasr_lpips_s2d(dasr_tensor x, int32_t N, int32_t H, int32_t W, const float* scale4, const float* shift4, dasr_tensor y, int32_t mode,
                              void* stream) {
    const int xf = (mode >> 4) & 7;   // bits 4-6: symmetry of the square applied in front of the network (transpose, flip rows, flip columns)
    mode &= 15;
    if (N <= 0 || H <= 0 || W <= 0 || (H & 3) || (W & 3) || (mode != 0 && mode != 1) || ((xf & 1) && H != W)) return DASR_EINVAL;
    const int Hs = (H + 4) >> 2, Ws = (W + 4) >> 2;
    const long long total = mode == 0 ? (long long)N * 3 * Hs * Ws * 4 : (long long)N * H * W;
    const f32x4 sc = {scale4[0], scale4[1], scale4[2], 0.f}, sh = {shift4[0], shift4[1], shift4[2], 0.f};
    DASR_LAUNCH(lpips_s2d_kernel, dim3(nblk(total)), dim3(256), 0, as_stream(stream), x, N, H, W, sc, sh, y, mode, xf);
    return (int)hipGetLastError();
}

extern "C" int dasr_maxpool3s2(dasr_tensor x, int32_t N, int32_t C, int32_t H, int32_t W, dasr_tensor y, void* stream) {
    if (N <= 0 || C <= 0 || H < 3 || W < 3) return DASR_EINVAL;
    const long long total = (long long)N * ((C + 15) / 16) * ((H - 3) / 2 + 1) * ((W - 3) / 2 + 1) * 4;
    DASR_LAUNCH(maxpool3s2_fwd_kernel, dim3(nblk(total)), dim3(256), 0, as_stream(stream), x, N, C, H, W, y);
    return (int)hipGetLastError();
}

extern "C" int dasr_maxpool3s2_bwd(dasr_tensor x, dasr_tensor gy, int32_t N, int32_t C, int32_t H, int32_t W, dasr_tensor gx, int32_t relu_mask,
                                   int32_t accumulate, void* stream) {
    if (N <= 0 || C <= 0 || H < 3 || W < 3) return DASR_EINVAL;
    const long long total = (long long)N * ((C + 15) / 16) * H * W * 4;
    DASR_LAUNCH(maxpool3s2_bwd_kernel, dim3(nblk(total)), dim3(256), 0, as_stream(stream), x, gy, N, C, H, W, gx, relu_mask, accumulate);
    return (int)hipGetLastError();
}

extern "C" int dasr_lpips_head(dasr_tensor f, int64_t pair_off, int32_t N, int32_t C, int32_t H, int32_t W, const float* lin, float eps, float coef,
                               float gcoef, float* loss_acc, dasr_tensor g0, int32_t relu_mask, void* stream) {
    if (N <= 0 || C <= 0 || (C & 15) || H <= 0 || W <= 0 || !lin) return DASR_EINVAL;
    const long long total = (long long)N * H * W;
    const dasr_red rs = dasr_red_scratch(loss_acc, as_stream(stream), nblk(total), 1);
    if (loss_acc && !rs.part) return dasr_red_error();
    DASR_LAUNCH(lpips_head_kernel, dim3(nblk(total)), dim3(256), 0, as_stream(stream), f, (long long)pair_off, N, C, H, W, lin, eps, coef, gcoef, loss_acc,
                g0, relu_mask, rs);
    return (int)hipGetLastError();
}
