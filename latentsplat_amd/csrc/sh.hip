// sh.hip — view-dependent payload channels from spherical harmonics, forward and backward.
//
//   colour   : rgb = max(0, 0.5 + sum_k B_k(dir) * shs[k][c])            (kernel axis convention,
//              degree <= 4, clamp flags kept for the backward)            lsr_sh.h)
//   features : f_c = 0.5 + sum_k B^ref_k(dir) * feature_sh[c][k]          (reference convention,
//              the latent-feature evaluation the reference does in PyTorch,
//              /root/reference/src/model/decoder/cuda_splatting.py:94-101; fused here for
//              degree <= 2, where B^ref(x,y,z) == B(z,x,y))
//   dir = normalize(mean * scene_scale - campos)
//
// One wave owns 64 consecutive Gaussians for ALL views.  Their coefficient block is contiguous in
// memory (64 x K*3 or 64 x C*Kf floats), so it is staged through LDS with lane-contiguous loads
// and read back per lane at an odd stride (no bank conflicts); a scene shared by all views
// (stride 0) is staged once.  The backward accumulates coefficient gradients in a second LDS
// array over the views and writes them back lane-contiguously — the per-thread strided
// read-modify-write of (G,25,3) tensors this replaces was the most expensive kernel of the
// colour + latent configuration.
#include "lsr_internal.h"
#include "lsr_sh.h"

namespace lsr {

struct ShParams {
    lsr_dims d;
    lsr_inputs in;
    const BinRec *binrec;     // radius > 0 <=> visible
    float *rec;               // forward: screen-space records (payload slots written here)
    const float *grec;        // backward: packed gradient records
    int RF;
    lsr_in_grads g;
    int group;                // 0 = colour (G,K,3), 1 = features (G,C,Kf)
};

__device__ __forceinline__ int padded_stride(int ks) { return ks; }   // rows are staged unpadded (see stage_in)

// Stage `rows`*ks floats (a contiguous, 16-byte aligned run: 64 Gaussians x ks coefficients) into
// LDS unchanged, with lane-contiguous 16-byte loads.  Lane l later reads its row at word stride
// ks: conflict-free when ks is odd (colour: 3*K with K = 1,9,25 ...), a few-way conflict otherwise.
__device__ __forceinline__ void stage_in(float *lds, const float *src, int ks, int rows, int lane) {
    const int n = rows * ks, n4 = n >> 2;
    const float4 *src4 = (const float4 *)src;
    float4 *lds4 = (float4 *)lds;
    for (int t = lane; t < n4; t += LSR_WAVE) lds4[t] = src4[t];
    for (int t = (n4 << 2) + lane; t < n; t += LSR_WAVE) lds[t] = src[t];
}
// coefficient (k, c) of this lane's Gaussian inside its LDS row
__device__ __forceinline__ int coef_index(bool channel_major, int k, int c, int K) { return channel_major ? c * K + k : 3 * k + c; }

template <bool BACKWARD>
__global__ void __launch_bounds__(LSR_WAVE)
k_sh(ShParams p) {
    extern __shared__ float s_lds[];
    const lsr_dims &d = p.d;
    const int lane = threadIdx.x, G = d.num_gaussians, V = d.num_views;
    const int g0 = blockIdx.x * LSR_WAVE, i = g0 + lane;
    const int rows = G - g0 < LSR_WAVE ? G - g0 : LSR_WAVE;
    const bool active = i < G;
    const int group = p.group;
    const int coff = d.color_mode != LSR_COLOR_NONE ? 3 : 0;
    const int deg = group == 0 ? d.sh_degree : d.feat_sh_degree;
    const int K = group == 0 ? d.sh_coeffs : d.feat_sh_coeffs;          // coefficients stored per channel
    const int nch = group == 0 ? 3 : d.feat_channels;
    const int nb = (deg + 1) * (deg + 1);
    const int ks = K * nch, ps = padded_stride(ks);
    const int64_t vs = group == 0 ? d.vs_color : d.vs_feat;
    const float *coef_base = group == 0 ? p.in.color : p.in.features;
    float *s_coef = s_lds;
    const float *my = s_coef + lane * ps;
    const int slot0 = 8 + (group == 0 ? 0 : coff);                        // first payload slot of the group
    const bool cmaj = group == 1 || d.color_sh_channel_major != 0;       // coefficient layout [c][k] vs [k][c]

    if (vs == 0) {
        stage_in(s_coef, coef_base + (size_t)g0 * ks, ks, rows, lane);
        __syncthreads();
    }
    float gmean_acc[3] = {0.0f, 0.0f, 0.0f};
    for (int v = 0; v < V; ++v) {
        if (vs != 0) {
            __syncthreads();
            stage_in(s_coef, coef_base + (size_t)v * vs + (size_t)g0 * ks, ks, rows, lane);
            __syncthreads();
        }
        const size_t o = (size_t)v * G + (active ? i : 0);
        const bool vis = active && p.binrec[o].radius > 0;
        float gm[3] = {0.0f, 0.0f, 0.0f};
        if (vis) {
            const float *vw = p.in.views + (size_t)v * LSR_VIEW_FLOATS;
            const float sc = vw[40];
            const float *mp = p.in.means3D + (size_t)v * d.vs_means + 3 * (size_t)i;
            float dx = mp[0] * sc - vw[32], dy = mp[1] * sc - vw[33], dz = mp[2] * sc - vw[34];
            const float len = sqrtf(dx * dx + dy * dy + dz * dz);
            dx = dx / len; dy = dy / len; dz = dz / len;
            // features use the reference's axis naming: B^ref(x,y,z) = B(z,x,y) up to degree 2
            const float bx = group == 0 ? dx : dz, by = group == 0 ? dy : dx, bz = group == 0 ? dz : dy;
            float bas[25];
            sh_basis(deg, bx, by, bz, bas);
            if (!BACKWARD) {
                float *R = p.rec + o * (size_t)p.RF;
                uint32_t clampbits = 0;
                for (int c = 0; c < nch; ++c) {
                    float acc = 0.0f;
                    for (int k = 0; k < nb; ++k) acc += bas[k] * my[coef_index(cmaj, k, c, K)];
                    acc += 0.5f;
                    if (group == 0) {
                        if (acc < 0.0f) clampbits |= 1u << c;
                        acc = acc > 0.0f ? acc : 0.0f;
                    }
                    R[slot0 + c] = acc;
                }
                if (group == 0) R[7] = __uint_as_float(clampbits);
            } else {
                // direction gradient only (coefficient gradients: k_sh_coef_backward below):
                //   sg[k] = sum_c coef[k][c] * dL/dchannel_c ;  dL/d(b) = sum_k dB_k/d(b) * sg[k]
                const float *gr = p.grec + o * (size_t)p.RF;
                const uint32_t clampbits = group == 0 ? __float_as_uint(p.rec[o * (size_t)p.RF + 7]) : 0u;
                float sg[25];
#pragma unroll
                for (int k = 0; k < 25; ++k) sg[k] = 0.0f;
                for (int c = 0; c < nch; ++c) {
                    const float gc = (clampbits >> c & 1u) ? 0.0f : gr[slot0 + c];
#pragma unroll
                    for (int k = 0; k < 25; ++k)
                        if (k < nb) sg[k] = __builtin_fmaf(my[coef_index(cmaj, k, c, K)], gc, sg[k]);
                }
                float dbas[25][3];
                sh_basis_grad(deg, bx, by, bz, dbas);
                float dd[3] = {0.0f, 0.0f, 0.0f};   // dL / d(bx,by,bz)
#pragma unroll
                for (int k = 0; k < 25; ++k)
                    if (k < nb) { dd[0] += dbas[k][0] * sg[k]; dd[1] += dbas[k][1] * sg[k]; dd[2] += dbas[k][2] * sg[k]; }
                // back to (dx,dy,dz) naming, then through the normalisation and the scene scale
                const float ddx = group == 0 ? dd[0] : dd[1], ddy = group == 0 ? dd[1] : dd[2], ddz = group == 0 ? dd[2] : dd[0];
                const float dot = ddx * dx + ddy * dy + ddz * dz;
                const float f = sc / len;
                gm[0] = (ddx - dx * dot) * f; gm[1] = (ddy - dy * dot) * f; gm[2] = (ddz - dz * dot) * f;
            }
        }
        if (BACKWARD) {
            if (d.vs_means != 0) {
                if (vis) {
                    float *o3 = p.g.means3D + (size_t)v * d.vs_means + 3 * (size_t)i;
                    o3[0] += gm[0]; o3[1] += gm[1]; o3[2] += gm[2];
                }
            } else { gmean_acc[0] += gm[0]; gmean_acc[1] += gm[1]; gmean_acc[2] += gm[2]; }
        }
    }
    if (BACKWARD) {
        if (d.vs_means == 0 && active) {
            float *o3 = p.g.means3D + 3 * (size_t)i;
            o3[0] += gmean_acc[0]; o3[1] += gmean_acc[1]; o3[2] += gmean_acc[2];
        }
    }
}

// ------------------------------------------------------------------------------------------
// Coefficient gradients.  Those of a 64-Gaussian block are ONE contiguous run of 64*ks floats
// (ks = K*channels).  Phase A (lane = Gaussian) leaves the SH basis and the (clamp-masked) channel
// gradients of up to kShViewChunk views in LDS; phase B (lane = element of the run) sums
// basis[v][g][k] * gch[v][g][c] over those views and stores the run lane-contiguously.
// Few registers (no per-element state), coalesced stores, no read-modify-write except when a
// shared scene has more views than one chunk.
constexpr int kShViewChunk = 4;
constexpr int kShBasisStride = 27;   // 25 basis values + a zero slot, odd stride

__global__ void __launch_bounds__(256)
k_sh_coef_backward(ShParams p) {
    extern __shared__ float s_lds[];
    const lsr_dims &d = p.d;
    const int tid = threadIdx.x, G = d.num_gaussians, V = d.num_views;
    const int g0 = blockIdx.x * LSR_WAVE;
    const int rows = G - g0 < LSR_WAVE ? G - g0 : LSR_WAVE;
    const int group = p.group;
    const int coff = d.color_mode != LSR_COLOR_NONE ? 3 : 0;
    const int deg = group == 0 ? d.sh_degree : d.feat_sh_degree;
    const int K = group == 0 ? d.sh_coeffs : d.feat_sh_coeffs;
    const int nch = group == 0 ? 3 : d.feat_channels;
    const int nb = (deg + 1) * (deg + 1);
    const int ks = K * nch;
    const int64_t vs = group == 0 ? d.vs_color : d.vs_feat;
    float *gcoef_base = group == 0 ? p.g.color : p.g.features;
    const int slot0 = 8 + (group == 0 ? 0 : coff);
    const bool cmaj = group == 1 || d.color_sh_channel_major != 0;
    constexpr int BS = kShBasisStride;
    const int cs = nch | 1;
    const int chunk = vs != 0 ? 1 : kShViewChunk;            // per-view outputs: one view at a time
    float *s_bas = s_lds, *s_gch = s_lds + kShViewChunk * LSR_WAVE * BS;

    for (int v0 = 0; v0 < V; v0 += chunk) {
        const int nv = V - v0 < chunk ? V - v0 : chunk;
        __syncthreads();   // previous chunk's phase B is done with the LDS arrays
        // ---- phase A: thread = (view of the chunk, Gaussian) ----
        if (tid < nv * LSR_WAVE) {
            const int vi = tid / LSR_WAVE, lane = tid % LSR_WAVE, v = v0 + vi, i = g0 + lane;
            const bool active = i < G;
            const size_t o = (size_t)v * G + (active ? i : 0);
            const bool vis = active && p.binrec[o].radius > 0;
            float *mb = s_bas + (vi * LSR_WAVE + lane) * BS, *mg = s_gch + (vi * LSR_WAVE + lane) * cs;
            float bas[25];
#pragma unroll
            for (int k = 0; k < 25; ++k) bas[k] = 0.0f;
            if (vis) {
                const float *vw = p.in.views + (size_t)v * LSR_VIEW_FLOATS;
                const float sc = vw[40];
                const float *mp = p.in.means3D + (size_t)v * d.vs_means + 3 * (size_t)i;
                float dx = mp[0] * sc - vw[32], dy = mp[1] * sc - vw[33], dz = mp[2] * sc - vw[34];
                const float len = sqrtf(dx * dx + dy * dy + dz * dz);
                dx = dx / len; dy = dy / len; dz = dz / len;
                sh_basis(deg, group == 0 ? dx : dz, group == 0 ? dy : dx, group == 0 ? dz : dy, bas);
            }
#pragma unroll
            for (int k = 0; k < 25; ++k) mb[k] = (vis && k < nb) ? bas[k] : 0.0f;
            mb[25] = 0.0f;
            const float *gr = p.grec + o * (size_t)p.RF;
            const uint32_t clampbits = (vis && group == 0) ? __float_as_uint(p.rec[o * (size_t)p.RF + 7]) : 0u;
            for (int c = 0; c < nch; ++c) mg[c] = (vis && !(clampbits >> c & 1u)) ? gr[slot0 + c] : 0.0f;
        }
        __syncthreads();
        // ---- phase B: thread = element of the contiguous coefficient-gradient run ----
        float *dst = gcoef_base + (vs != 0 ? (size_t)v0 * vs : 0) + (size_t)g0 * ks;
        for (int t = tid; t < rows * ks; t += 256) {
            const int g = t / ks, rem = t - g * ks;
            const int k = cmaj ? rem % K : rem / 3, c = cmaj ? rem / K : rem % 3;
            const int kb = k < nb ? k : 25;   // coefficients beyond the evaluated bands: zero slot
            float a = (vs == 0 && v0 > 0) ? dst[t] : 0.0f;
            for (int vi = 0; vi < nv; ++vi)
                a = __builtin_fmaf(s_bas[(vi * LSR_WAVE + g) * BS + kb], s_gch[(vi * LSR_WAVE + g) * cs + c], a);
            dst[t] = a;
        }
    }
}

static bool group_enabled(const lsr_dims &d, int group) {
    return group == 0 ? d.color_mode == LSR_COLOR_SH : (d.feat_channels > 0 && d.feat_mode == LSR_FEAT_SH);
}

hipError_t launch_sh_forward(const lsr_dims &d, const lsr_inputs &in, char *geom, hipStream_t s) {
    if (d.num_gaussians == 0) return hipSuccess;
    if (!group_enabled(d, 0) && !group_enabled(d, 1)) return hipSuccess;
    const GeomLayout L = geom_layout(d);
    prof_begin(kStShFwd, s);
    for (int group = 0; group < 2; ++group) {
        if (!group_enabled(d, group)) continue;
        ShParams p;
        p.d = d; p.in = in; p.binrec = (const BinRec *)(geom + L.bin);
        p.rec = (float *)(geom + L.rec); p.grec = nullptr; p.RF = L.rec_floats; p.g = lsr_in_grads{}; p.group = group;
        const int ks = group == 0 ? d.sh_coeffs * 3 : d.feat_sh_coeffs * d.feat_channels;
        const size_t shm = (size_t)LSR_WAVE * ks * 4 + 16;
        hipLaunchKernelGGL((k_sh<false>), dim3((d.num_gaussians + LSR_WAVE - 1) / LSR_WAVE), dim3(LSR_WAVE), shm, s, p);
    }
    prof_end(kStShFwd, s);
    return hipGetLastError();
}

hipError_t launch_sh_backward(const lsr_dims &d, const lsr_inputs &in, const char *geom, const char *grad,
                              const lsr_in_grads &gin, hipStream_t s) {
    if (d.num_gaussians == 0) return hipSuccess;
    if (!group_enabled(d, 0) && !group_enabled(d, 1)) return hipSuccess;
    const GeomLayout L = geom_layout(d);
    const GradLayout R = grad_layout(d);
    prof_begin(kStShBwd, s);
    for (int group = 0; group < 2; ++group) {
        if (!group_enabled(d, group)) continue;
        ShParams p;
        p.d = d; p.in = in; p.binrec = (const BinRec *)(geom + L.bin);
        p.rec = (float *)const_cast<char *>(geom + L.rec); p.grec = (const float *)(grad + R.rec); p.RF = L.rec_floats; p.g = gin; p.group = group;
        const int nch = group == 0 ? 3 : d.feat_channels;
        const int ks = group == 0 ? d.sh_coeffs * 3 : d.feat_sh_coeffs * d.feat_channels;
        const dim3 grid((d.num_gaussians + LSR_WAVE - 1) / LSR_WAVE), block(LSR_WAVE);
        // (1) direction -> mean gradient (needs the coefficients, staged through LDS)
        hipLaunchKernelGGL((k_sh<true>), grid, block, (size_t)LSR_WAVE * ks * 4 + 16, s, p);
        // (2) coefficient gradients
        const size_t shm = (size_t)kShViewChunk * LSR_WAVE * (kShBasisStride + (nch | 1)) * 4;
        hipLaunchKernelGGL(k_sh_coef_backward, grid, dim3(256), shm, s, p);
    }
    prof_end(kStShBwd, s);
    return hipGetLastError();
}

}  // namespace lsr
