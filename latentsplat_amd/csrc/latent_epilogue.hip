// latent_epilogue.hip — posterior sample + anti-aliased downscale + skip concatenation of the
// rendered feature maps (include/lsr_latent.h), forward and backward, one launch each.
//
// Reference behaviour restated (no code taken): /root/reference/src/model/decoder/
// decoder_splatting_cuda.py:46-47 (logvar from the mask), src/model/diagonal_gaussian_distribution.py:
// 55-63,75-80 (clamp, std, sample), src/model/model_wrapper.py:266-274,376,382 (rescale, cat).
// `rescale` is torchvision's tensor `resize(antialias=True)`, i.e. ATen's separable
// `_upsample_bilinear2d_aa`: for output index o, scale = in/out (>= 1 here), support = scale,
//   center = scale*(o+0.5);  lo = max(int(center-support+0.5), 0);  hi = min(int(center+support+0.5), in)
//   w(x) = max(0, 1 - |(x - center + 0.5)/scale|) / sum_{x in [lo,hi)} (same)
//
// Forward: block per (view, channel plane, output row).  The block walks the input rows that
// are either "owned" by its band (written out as the sample, each row exactly once overall) or
// inside its vertical filter window (accumulated per column in registers), then filters the
// column sums horizontally out of LDS.  Inputs are read once from HBM (the halo rows of
// neighbouring bands hit L2), outputs written once.  HBM-bound.
#include "lsr_internal.h"
#include "lsr_latent.h"

namespace lsr {

struct AaTaps {
    int lo, hi;
    float center, inv_scale, inv_total;
};

__device__ __forceinline__ float aa_raw_weight(const AaTaps &t, int x) {
    const float a = ((float)x - t.center + 0.5f) * t.inv_scale;
    return fmaxf(0.0f, 1.0f - fabsf(a));
}
__device__ static AaTaps aa_taps(int o, float scale, int in_size) {
    AaTaps t;
    t.center = scale * ((float)o + 0.5f);
    t.inv_scale = 1.0f / scale;
    t.lo = max((int)(t.center - scale + 0.5f), 0);
    t.hi = min((int)(t.center + scale + 0.5f), in_size);
    float total = 0.0f;
    for (int x = t.lo; x < t.hi; ++x) total += aa_raw_weight(t, x);
    t.inv_total = total != 0.0f ? 1.0f / total : 0.0f;
    return t;
}
__device__ __forceinline__ float aa_weight(const AaTaps &t, int x) {
    return (x >= t.lo && x < t.hi) ? aa_raw_weight(t, x) * t.inv_total : 0.0f;
}

__device__ __forceinline__ float clamp_keep_nan(float v, float lo, float hi) {
    return v < lo ? lo : (v > hi ? hi : v);
}

constexpr int kLatentThreads = 256;
constexpr int kLatentMaxCols = 4;     // columns per thread in the forward: width <= 1024

// number of row bands the forward is split into (= out_height when z is wanted)
__host__ __device__ inline int latent_bands(const lsr_latent_dims &d, bool want_z) {
    return want_z ? d.out_height : (d.height + 7) / 8;
}

__global__ __launch_bounds__(kLatentThreads) void k_latent_fwd(lsr_latent_dims d, lsr_latent_inputs in,
                                                                lsr_latent_outputs out) {
    extern __shared__ float col[];                       // [width] column sums of this band
    const int H = d.height, W = d.width, C = d.channels, cc = d.color_channels;
    const bool want_z = out.z != nullptr;
    const int bands = latent_bands(d, want_z);
    const int band = blockIdx.x, plane = blockIdx.y, v = blockIdx.z;   // plane in [0, cc + C)
    const int own_lo = (int)(((int64_t)band * H) / bands), own_hi = (int)(((int64_t)(band + 1) * H) / bands);
    const size_t HW = (size_t)H * W;
    if (plane < cc) {                                    // colour channels of `skip`: plain copy
        if (!out.skip) return;
        const float *src = in.color + ((size_t)v * 3 + plane) * HW;
        float *dst = out.skip + ((size_t)v * (cc + C) + plane) * HW;
        if ((W & 3) == 0) {
            const float4 *s4 = (const float4 *)src;
            float4 *d4 = (float4 *)dst;
            for (int i = (own_lo * W >> 2) + threadIdx.x; i < (own_hi * W >> 2); i += kLatentThreads) d4[i] = s4[i];
        } else {
            for (int i = own_lo * W + threadIdx.x; i < own_hi * W; i += kLatentThreads) dst[i] = src[i];
        }
        return;
    }
    const int c = plane - cc;
    const int fch = d.logvar_mode == LSR_LOGVAR_FROM_FEATURES ? 2 * C : C;
    const float *mean = in.features + ((size_t)v * fch + c) * HW;
    const float *lvsrc = d.logvar_mode == LSR_LOGVAR_FROM_FEATURES ? in.features + ((size_t)v * fch + C + c) * HW
                                                                    : in.mask + (size_t)v * HW;
    const float *noise = in.noise ? in.noise + ((size_t)v * C + c) * HW : nullptr;
    float *sample = out.skip ? out.skip + ((size_t)v * (cc + C) + plane) * HW : nullptr;
    float *lvout = nullptr;
    if (out.logvar) {
        if (d.logvar_mode == LSR_LOGVAR_FROM_FEATURES) lvout = out.logvar + ((size_t)v * C + c) * HW;
        else if (c == 0) lvout = out.logvar + (size_t)v * HW;
    }
    AaTaps ty;
    ty.lo = ty.hi = own_lo;
    if (want_z) ty = aa_taps(band, (float)H / (float)d.out_height, H);
    const int y0 = want_z ? min(own_lo, ty.lo) : own_lo, y1 = want_z ? max(own_hi, ty.hi) : own_hi;
    const float var_lo = expf(d.logvar_min), var_hi = expf(d.logvar_max);
    // one element: sample value s (and the clamped logvar on request)
    auto sample_of = [&](float m, float raw, float nz, bool want_lv, float &lv_out) -> float {
        float s = m;
        if (d.logvar_mode == LSR_LOGVAR_FROM_MASK) {
            // std = exp(clamp(log(1 - m)) / 2) = sqrt(clamp(1 - m, e^min, e^max)): no log / exp per
            // element; the logvar itself is only needed on the own rows of one channel
            const float om = 1.0f - raw;
            if (want_lv) lv_out = clamp_keep_nan(__logf(om), d.logvar_min, d.logvar_max);
            if (noise) s += (om < 0.0f ? __builtin_nanf("") : __fsqrt_rn(clamp_keep_nan(om, var_lo, var_hi))) * nz;
        } else {
            const float lv = clamp_keep_nan(raw, d.logvar_min, d.logvar_max);
            if (want_lv) lv_out = lv;
            if (noise) s += __expf(0.5f * lv) * nz;
        }
        return s;
    };
    const int tpr = W >> 2;                              // threads per row with 16-byte accesses
    if ((W & 3) == 0 && tpr <= kLatentThreads && kLatentThreads % tpr == 0) {
        // ---- vector path: a thread owns 4 adjacent columns; kLatentThreads / tpr rows in flight ----
        const int rl = threadIdx.x / tpr, cg = threadIdx.x - rl * tpr, nrl = kLatentThreads / tpr;
        float4 a4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        for (int y = y0 + rl; y < y1; y += nrl) {
            const bool own = y >= own_lo && y < own_hi;
            const float wy = want_z ? aa_weight(ty, y) : 0.0f;
            const size_t i4 = ((size_t)y * W >> 2) + cg;
            const float4 m = ((const float4 *)mean)[i4];
            const bool need_lv = noise || (lvout && own);
            const float4 raw = need_lv ? ((const float4 *)lvsrc)[i4] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            const float4 nz = noise ? ((const float4 *)noise)[i4] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            float4 lv = make_float4(0.0f, 0.0f, 0.0f, 0.0f), s4;
            const bool wl = lvout && own;
            s4.x = sample_of(m.x, raw.x, nz.x, wl, lv.x); s4.y = sample_of(m.y, raw.y, nz.y, wl, lv.y);
            s4.z = sample_of(m.z, raw.z, nz.z, wl, lv.z); s4.w = sample_of(m.w, raw.w, nz.w, wl, lv.w);
            if (wl) ((float4 *)lvout)[i4] = lv;
            if (own && sample) ((float4 *)sample)[i4] = s4;
            a4.x += wy * s4.x; a4.y += wy * s4.y; a4.z += wy * s4.z; a4.w += wy * s4.w;
        }
        if (!want_z) return;
        // column sums: add the row lanes through LDS (rl = 0 stores, the others add in turn)
        for (int r = 0; r < nrl; ++r) {
            if (rl == r) {
                float4 *c4 = (float4 *)col + cg;
                if (r == 0) *c4 = a4;
                else { float4 t = *c4; t.x += a4.x; t.y += a4.y; t.z += a4.z; t.w += a4.w; *c4 = t; }
            }
            __syncthreads();
        }
    } else {
    float acc[kLatentMaxCols] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 4
    for (int y = y0; y < y1; ++y) {                      // own rows and filter window overlap: one contiguous range
        const bool own = y >= own_lo && y < own_hi;
        const float wy = want_z ? aa_weight(ty, y) : 0.0f;
#pragma unroll
        for (int k = 0; k < kLatentMaxCols; ++k) {
            const int x = threadIdx.x + k * kLatentThreads;
            if (x >= W) break;
            const size_t i = (size_t)y * W + x;
            const bool need_lv = noise || (lvout && own);
            float lv = 0.0f;
            const float s = sample_of(mean[i], need_lv ? lvsrc[i] : 0.0f, noise ? noise[i] : 0.0f, lvout && own, lv);
            if (lvout && own) lvout[i] = lv;
            if (own && sample) sample[i] = s;
            acc[k] += wy * s;
        }
    }
    if (!want_z) return;
#pragma unroll
    for (int k = 0; k < kLatentMaxCols; ++k) {
        const int x = threadIdx.x + k * kLatentThreads;
        if (x < W) col[x] = acc[k];
    }
    __syncthreads();
    }
    const float sx = (float)W / (float)d.out_width;
    float *zrow = out.z + (((size_t)v * C + c) * d.out_height + band) * d.out_width;
    for (int ox = threadIdx.x; ox < d.out_width; ox += kLatentThreads) {
        const AaTaps tx = aa_taps(ox, sx, W);
        float r = 0.0f;
        for (int x = tx.lo; x < tx.hi; ++x) r += aa_raw_weight(tx, x) * col[x];
        zrow[ox] = r * tx.inv_total;
    }
}

// block per (view, latent channel, band of kBwdRows input rows); thread = column, so the
// horizontal candidates (the <= 4 outputs whose window covers x) are found once per thread
constexpr int kBwdRows = 16;
constexpr int kBwdCand = 4;

__global__ __launch_bounds__(kLatentThreads) void k_latent_bwd(lsr_latent_dims d, lsr_latent_inputs in,
                                                                lsr_latent_out_grads g, float *__restrict__ d_features) {
    __shared__ float ty_w[kBwdRows][kBwdCand];           // vertical weights of the band's rows
    __shared__ int ty_o[kBwdRows][kBwdCand];
    const int H = d.height, W = d.width, C = d.channels, cc = d.color_channels;
    const int y0 = blockIdx.x * kBwdRows, c = blockIdx.y, v = blockIdx.z;
    const int rows = min(kBwdRows, H - y0);
    const size_t HW = (size_t)H * W;
    const bool have_z = g.z != nullptr;
    const float sx = have_z ? (float)W / (float)d.out_width : 1.0f;
    if (have_z && threadIdx.x < kBwdRows * kBwdCand) {
        const int r = threadIdx.x / kBwdCand, k = threadIdx.x % kBwdCand;
        const float sy = (float)H / (float)d.out_height;
        const int y = y0 + r;
        const int o = (int)(((float)y + 0.5f) / sy) - 2 + k;        // windows reach < 1.5 + 0.5/sy outputs away
        float w = 0.0f;
        if (r < rows && o >= 0 && o < d.out_height) w = aa_weight(aa_taps(o, sy, H), y);
        // a 5th candidate (o_c + 2) can only matter when sy < 1, which the ABI excludes
        ty_w[r][k] = w;
        ty_o[r][k] = max(0, min(o, d.out_height - 1));
    }
    __syncthreads();
    const int fch = d.logvar_mode == LSR_LOGVAR_FROM_FEATURES ? 2 * C : C;
    const float *gz = have_z ? g.z + ((size_t)v * C + c) * d.out_height * d.out_width : nullptr;
    for (int x = threadIdx.x; x < W; x += kLatentThreads) {
        float wx[kBwdCand];
        int oxs[kBwdCand];
        if (have_z) {
            const int oc = (int)(((float)x + 0.5f) / sx) - 2;
#pragma unroll
            for (int k = 0; k < kBwdCand; ++k) {
                const int o = oc + k;
                wx[k] = (o >= 0 && o < d.out_width) ? aa_weight(aa_taps(o, sx, W), x) : 0.0f;
                oxs[k] = max(0, min(o, d.out_width - 1));
            }
        }
        for (int r = 0; r < rows; ++r) {
            const size_t i = (size_t)(y0 + r) * W + x;
            float gt = g.skip ? g.skip[((size_t)v * (cc + C) + cc + c) * HW + i] : 0.0f;
            if (have_z) {
#pragma unroll
                for (int a = 0; a < kBwdCand; ++a) {
                    const float wy = ty_w[r][a];
                    if (wy == 0.0f) continue;
                    const float *zr = gz + (size_t)ty_o[r][a] * d.out_width;
                    float h = 0.0f;
#pragma unroll
                    for (int k = 0; k < kBwdCand; ++k) h += wx[k] * zr[oxs[k]];
                    gt += wy * h;
                }
            }
            d_features[((size_t)v * fch + c) * HW + i] = gt;
            if (d.logvar_mode == LSR_LOGVAR_FROM_FEATURES) {
                float dl = 0.0f;
                if (in.noise) {
                    const float lv = in.features[((size_t)v * fch + C + c) * HW + i];
                    if (lv >= d.logvar_min && lv <= d.logvar_max)
                        dl = gt * in.noise[((size_t)v * C + c) * HW + i] * 0.5f * expf(0.5f * lv);
                }
                d_features[((size_t)v * fch + C + c) * HW + i] = dl;
            }
        }
    }
}

}  // namespace lsr

using namespace lsr;

static int latent_check(const lsr_latent_dims *d, const lsr_latent_inputs *in) {
    if (!d || !in) return LSR_ENULL;
    if (d->num_views < 1 || d->channels < 1 || d->height < 1 || d->width < 1) return LSR_EINVAL;
    if (d->logvar_mode != LSR_LOGVAR_FROM_MASK && d->logvar_mode != LSR_LOGVAR_FROM_FEATURES) return LSR_EINVAL;
    if (d->color_channels != 0 && d->color_channels != 3) return LSR_EINVAL;
    if (d->num_views > 65535 || d->channels + d->color_channels > 65535) return LSR_EUNSUPPORTED;
    if (d->width > kLatentThreads * kLatentMaxCols) return LSR_EUNSUPPORTED;
    if (!in->features) return LSR_ENULL;
    if (d->logvar_mode == LSR_LOGVAR_FROM_MASK && in->noise && !in->mask) return LSR_ENULL;
    return LSR_OK;
}
static int latent_check_out(const lsr_latent_dims *d, bool want_z) {
    if (!want_z) return LSR_OK;
    if (d->out_height < 1 || d->out_width < 1 || d->out_height > d->height || d->out_width > d->width) return LSR_EINVAL;
    return LSR_OK;
}

extern "C" {

int lsr_latent_forward(const lsr_latent_dims *d, const lsr_latent_inputs *in,
                       const lsr_latent_outputs *out, lsr_stream_t stream) {
    note_hip_error(0);
    int rc = latent_check(d, in);
    if (rc) return rc;
    if (!out) return LSR_ENULL;
    if (d->logvar_mode == LSR_LOGVAR_FROM_MASK && out->logvar && !in->mask) return LSR_ENULL;
    if (d->color_channels && out->skip && !in->color) return LSR_ENULL;
    rc = latent_check_out(d, out->z != nullptr);
    if (rc) return rc;
    if (!out->skip && !out->z && !out->logvar) return LSR_OK;
    const int planes = (out->skip ? d->color_channels : 0) + d->channels;
    lsr_latent_dims dd = *d;
    if (!out->skip) dd.color_channels = 0;
    const dim3 grid(latent_bands(dd, out->z != nullptr), planes, d->num_views);
    hipStream_t s = (hipStream_t)stream;
    prof_begin(kStLatentFwd, s);
    hipLaunchKernelGGL(k_latent_fwd, grid, dim3(kLatentThreads), sizeof(float) * d->width, s, dd, *in, *out);
    prof_end(kStLatentFwd, s);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { note_hip_error((int)e); return LSR_ELAUNCH; }
    return LSR_OK;
}

int lsr_latent_backward(const lsr_latent_dims *d, const lsr_latent_inputs *in,
                        const lsr_latent_out_grads *dout, float *d_features, lsr_stream_t stream) {
    note_hip_error(0);
    int rc = latent_check(d, in);
    if (rc) return rc;
    if (!dout || !d_features) return LSR_ENULL;
    rc = latent_check_out(d, dout->z != nullptr);
    if (rc) return rc;
    const dim3 grid((d->height + kBwdRows - 1) / kBwdRows, d->channels, d->num_views);
    hipStream_t s = (hipStream_t)stream;
    prof_begin(kStLatentBwd, s);
    hipLaunchKernelGGL(k_latent_bwd, grid, dim3(kLatentThreads), 0, s, *d, *in, *dout, d_features);
    prof_end(kStLatentBwd, s);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { note_hip_error((int)e); return LSR_ELAUNCH; }
    return LSR_OK;
}

}  // extern "C"
