/*
 * lsr_latent.h — C ABI of the latent epilogue: what the reference does to the rendered feature
 * maps right after the rasterizer.  Same library and conventions as
 * lsr_rasterizer.h (device pointers, sizes, a hipStream_t, negative LSR_E* codes).
 *
 * Reference operations fused into one forward and one backward launch (paths relative to
 * /root/reference):
 *   - src/model/decoder/decoder_splatting_cuda.py:46-47   logvar = log(1 - mask.detach()) broadcast
 *         to the C feature channels (or, `variational`, the upper half of the channels);
 *   - src/model/diagonal_gaussian_distribution.py:55-63   clamp(logvar, -30, 20), std = exp(logvar/2);
 *   - diagonal_gaussian_distribution.py:75-80             sample = mean + std * randn_like(mean)
 *         (the noise tensor is an INPUT here so that the caller keeps torch's RNG stream);
 *   - src/model/model_wrapper.py:266-274, :376            z = rescale(sample, 1/supersampling):
 *         torchvision `resize(..., antialias=True)` on a tensor == bilinear interpolation with
 *         align_corners=False and the anti-aliasing triangle filter of support `scale`
 *         (ATen `_upsample_bilinear2d_aa`; weights restated in csrc/latent_epilogue.hip);
 *   - model_wrapper.py:382                                skip_z = cat(color.detach(), sample).
 * The reference spends ~10 elementwise / resize / cat launches over (V,C,256,256) maps on this;
 * here every input word is read once and every output word written once.
 */
#ifndef LSR_LATENT_H
#define LSR_LATENT_H

#include "lsr_rasterizer.h"

#ifdef __cplusplus
extern "C" {
#endif

enum { LSR_LOGVAR_FROM_MASK = 0, LSR_LOGVAR_FROM_FEATURES = 1 };

typedef struct lsr_latent_dims {
    int32_t num_views;       /* V = b*v */
    int32_t channels;        /* C latent channels of the sample */
    int32_t height, width;   /* rendered (supersampled) size */
    int32_t out_height, out_width; /* z size; <= height/width (downscale only) */
    int32_t logvar_mode;     /* LSR_LOGVAR_FROM_MASK: logvar = log(1-mask) for all channels;
                              * LSR_LOGVAR_FROM_FEATURES: `features` has 2C channels, [C,2C) = logvar */
    int32_t color_channels;  /* 0, or 3: `skip` = cat(color, sample) */
    float logvar_min, logvar_max; /* -30, 20 */
    int32_t reserved0, reserved1;
} lsr_latent_dims;

typedef struct lsr_latent_inputs {
    const float *features;  /* [V][C or 2C][H][W]  rendered feature map (posterior mean [, logvar]) */
    const float *mask;      /* [V][H][W]           1 - final transmittance; NULL in FROM_FEATURES mode */
    const float *noise;     /* [V][C][H][W]        standard normal; NULL = take the mean (zero variance) */
    const float *color;     /* [V][3][H][W]        only with color_channels == 3 */
} lsr_latent_inputs;

typedef struct lsr_latent_outputs {
    float *skip;    /* [V][color_channels + C][H][W]; the sample is channels [color_channels, ..). NULL = not wanted */
    float *z;       /* [V][C][out_height][out_width]; NULL = not wanted */
    float *logvar;  /* [V][1 (mask mode) | C (features mode)][H][W] clamped logvar; NULL = not wanted */
} lsr_latent_outputs;

typedef struct lsr_latent_out_grads {
    const float *skip;  /* layout of outputs.skip (colour channels ignored: detached); may be NULL */
    const float *z;     /* may be NULL */
} lsr_latent_out_grads;

/* One launch, asynchronous. */
int lsr_latent_forward(const lsr_latent_dims *d, const lsr_latent_inputs *in,
                       const lsr_latent_outputs *out, lsr_stream_t stream);

/* d/d features ([V][C or 2C][H][W], written, not accumulated): mean channels get
 * g_sample + resize^T(g_z); in FROM_FEATURES mode the logvar channels get
 * g_sample_total * noise * std / 2 inside the clamp interval.  The mask is detached. */
int lsr_latent_backward(const lsr_latent_dims *d, const lsr_latent_inputs *in,
                        const lsr_latent_out_grads *dout, float *d_features, lsr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* LSR_LATENT_H */
