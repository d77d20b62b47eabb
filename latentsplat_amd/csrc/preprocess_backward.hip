// preprocess_backward.hip — stages K8+K9 fused: per Gaussian, push the screen-space gradients
// collected by the compositing backward through
//   conic -> 2D covariance -> 3D covariance (6 packed values) and -> view-space mean (via J),
//   NDC mean -> world mean (perspective divide),  view z -> world mean (depth output),
// and back through the in-kernel scene scale.  (SH colour / latent-SH gradients: sh.hip.)
// Also scatters the per-(view, Gaussian) opacity / feature / precomputed-colour gradients from the
// packed records to the caller's tensors.
// A block of 256 threads owns 256 / PARTS Gaussians; its PARTS wave groups split the views of those
// Gaussians between them (group p takes views p, p+PARTS, ...: PARTS x more waves in flight and shorter
// per-thread load chains than one thread per Gaussian).  The view is WAVE-UNIFORM, so the camera comes
// through the scalar cache into SGPRs (round 1 gave the PARTS lanes of a quad different views: 40
// broadcast vector loads per (view, Gaussian), which kept the texture addresser ~90 % busy).  Gradients
// of inputs shared between views (stride 0) are summed in registers and combined across the PARTS
// groups through LDS in a fixed order — no atomics, no read-modify-write.
// Divisions are v_rcp_f32 (1 ulp) here: unlike the forward's, these results feed no bit-exact list.
// Spec: SURVEY.md Appendix A.6.
#include "lsr_internal.h"

namespace lsr {

struct PreBwdParams {
    lsr_dims d;
    lsr_inputs in;
    const int32_t *radii;
    const float *geo;       // screen-space records; slot 7 = colour clamp bits (SH mode)
    int geo_floats;
    const float *rec;       // packed gradient records (lsr_internal.h GradLayout)
    const float *zero_rec;  // an all-zero record line: read in place of the (uncleared, unwritten) record of a culled slot
    int rec_floats;
    lsr_in_grads g;
    GroupStrides gs;        // view groups (blockIdx.y): element offsets of the next group's slices
};

constexpr int kPreBwdFeat = 8;   // shared direct-feature gradients kept in registers up to this many channels

typedef const float __attribute__((address_space(4))) *kfloat_ptr;   // constant address space: uniform loads become s_load
__device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }

constexpr int kPreBwdThreads = 256;
constexpr int kPreBwdShared = 3 + 6 + 1 + kPreBwdFeat + 3;   // register accumulators of the shared inputs

template <int PARTS>
__global__ void __launch_bounds__(kPreBwdThreads, 4)
k_preprocess_bwd(PreBwdParams pk) {
    // view groups: blockIdx.y = group; its inputs, gradient outputs and workspace slices (uniform: scalar arithmetic)
    PreBwdParams p = pk;
    {
        const int64_t g = blockIdx.y;
        p.in.views += g * pk.gs.views; p.in.means3D += g * pk.gs.means; p.in.cov3D += g * pk.gs.cov; p.in.opacities += g * pk.gs.opac;
        p.radii += g * pk.gs.slots; p.geo += g * pk.gs.slots * pk.geo_floats; p.rec += g * pk.gs.slots * pk.rec_floats;
        p.g.means3D += g * pk.gs.means; p.g.cov3D += g * pk.gs.cov; p.g.opacities += g * pk.gs.opac;
        if (p.g.color) p.g.color += g * pk.gs.color;
        if (p.g.features) p.g.features += g * pk.gs.feat;
        if (p.g.means2D) p.g.means2D += g * pk.gs.slots * 3;
    }
    constexpr int LPP = kPreBwdThreads / PARTS;               // Gaussians per block = lanes per part
    __shared__ float s_part[PARTS > 1 ? (PARTS - 1) * kPreBwdShared * LPP : 1];
    const int part = __builtin_amdgcn_readfirstlane((int)threadIdx.x / LPP);   // wave-uniform
    const int slot = (int)threadIdx.x % LPP;
    const lsr_dims &d = p.d;
    const int G = d.num_gaussians;
    const int gi = blockIdx.x * LPP + slot;
    const bool live = gi < G;
    const int i = live ? gi : G - 1;     // idle lanes shadow the last Gaussian and store nothing
    const int V = d.num_views;
    const int ce = d.cov_elems;
    float am[3] = {0, 0, 0}, ac[6] = {0, 0, 0, 0, 0, 0}, aop = 0.0f;  // accumulators for shared inputs
    float af[kPreBwdFeat], acol[3] = {0, 0, 0};
#pragma unroll
    for (int c = 0; c < kPreBwdFeat; ++c) af[c] = 0.0f;
    const int coff = d.color_mode != LSR_COLOR_NONE ? 3 : 0;
    const bool feat_direct = d.feat_channels > 0 && d.feat_mode == LSR_FEAT_DIRECT;
    const bool feat_reg = feat_direct && d.vs_feat == 0 && d.feat_channels <= kPreBwdFeat;
    const bool feat_rmw = feat_direct && d.vs_feat == 0 && !feat_reg;   // many shared channels: one group, memory accumulate
    // Occupancy, not prefetching, hides the memory latency here: holding the NEXT view's record in
    // registers (round 1) cost 175 VGPRs = 2 waves per SIMD (0.155 ms per 16 views; 0.137 without it).
    const bool pay16 = feat_reg && p.rec_floats == 16;
    const int v_first = feat_rmw ? 0 : part, v_step = feat_rmw ? 1 : PARTS, v_end = (feat_rmw && part != 0) ? 0 : V;   // all wave-uniform
    for (int v = v_first; v < v_end; v += v_step) {
        const size_t o = (size_t)v * G + i;
        const bool vis = p.radii[o] > 0;
        // (the records of culled slots are not cleared: their lanes read the zero line — one cached line instead of 32 % of
        // the record array)
        const float *rc = vis ? p.rec + o * p.rec_floats : p.zero_rec;
        const float4 r0 = *(const float4 *)rc, r1 = *(const float4 *)(rc + 4);           // m1x m1y m2xx m2xy | m2yy m0 gz -
        float4 q0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), q1 = q0;
        if (pay16) { q0 = *(const float4 *)(rc + 8); q1 = *(const float4 *)(rc + 12); }
        // opacity: the record holds m0 = sum opacity * G * dL/dalpha; dL/dopacity = m0 / opacity
        // (m0 != 0 implies opacity >= 1/255)
        if (d.vs_opac != 0) {
            const float o_in = p.in.opacities[(size_t)v * d.vs_opac + i];
            if (live) p.g.opacities[(size_t)v * d.vs_opac + i] = r1.y != 0.0f ? r1.y * rcp(o_in) : 0.0f;
        } else aop += r1.y;   // shared opacity: one division after the sum over the views
        // features / precomputed colours: plain pass-through of the record
        if (feat_reg) {
            if (p.rec_floats == 16) {   // the whole payload half of the record as two 16-byte loads
                const float pay[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
                if (coff == 0) {
#pragma unroll
                    for (int c = 0; c < 8; ++c) af[c] += pay[c];        // slots past the last channel are zero
                } else {
#pragma unroll
                    for (int c = 0; c < 5; ++c) af[c] += pay[3 + c];
                }
            } else {
#pragma unroll
                for (int c = 0; c < kPreBwdFeat; ++c)
                    if (c < d.feat_channels) af[c] += rc[8 + coff + c];
            }
        } else if (feat_direct && live) {
            float *gf = p.g.features + (size_t)v * d.vs_feat + (size_t)i * d.feat_channels;
            const bool first = d.vs_feat != 0 || v == 0;
            for (int c = 0; c < d.feat_channels; ++c) gf[c] = first ? rc[8 + coff + c] : gf[c] + rc[8 + coff + c];
        }
        if (d.color_mode == LSR_COLOR_PRECOMP) {
            if (d.vs_color != 0) {
                float *gcp = p.g.color + (size_t)v * d.vs_color + 3 * (size_t)i;
                if (live) for (int c = 0; c < 3; ++c) gcp[c] = rc[8 + c];
            } else {
                for (int c = 0; c < 3; ++c) acol[c] += rc[8 + c];
            }
        }
        // ---- geometry: every lane runs the arithmetic (culled lanes on whatever they hold; IEEE special
        // values are harmless), `vis` masks the results where they are accumulated or stored ----
        const kfloat_ptr vw = (kfloat_ptr)(p.in.views + (size_t)v * LSR_VIEW_FLOATS);
        float vm[16], pm[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) { vm[k] = vw[k]; pm[k] = vw[16 + k]; }
        const float tanfovx = vw[35], tanfovy = vw[36];
        const float focal_x = d.width * rcp(2.0f * tanfovx), focal_y = d.height * rcp(2.0f * tanfovy);
        const float scale = vw[40], scale2 = scale * scale;
        const float *mp = p.in.means3D + (size_t)v * d.vs_means + 3 * (size_t)i;
        const float p0 = mp[0] * scale, p1 = mp[1] * scale, p2 = mp[2] * scale;
        // ---- covariance path ----
        const float t0 = vm[0] * p0 + vm[4] * p1 + vm[8] * p2 + vm[12];
        const float t1 = vm[1] * p0 + vm[5] * p1 + vm[9] * p2 + vm[13];
        const float tz = vm[2] * p0 + vm[6] * p1 + vm[10] * p2 + vm[14];
        const float limx = 1.3f * tanfovx, limy = 1.3f * tanfovy;
        const float itz = rcp(tz), itz2 = itz * itz, itz3 = itz2 * itz;
        const float txtz = t0 * itz, tytz = t1 * itz;
        const float xm = (txtz < -limx || txtz > limx) ? 0.0f : 1.0f;
        const float ym = (tytz < -limy || tytz > limy) ? 0.0f : 1.0f;
        const float tx = fminf(limx, fmaxf(-limx, txtz)) * tz;
        const float ty = fminf(limy, fmaxf(-limy, tytz)) * tz;
        const float j00 = focal_x * itz, j02 = -(focal_x * tx) * itz2;
        const float j11 = focal_y * itz, j12 = -(focal_y * ty) * itz2;
        // Wr[r][c] = vm[4c + r]; M = J * Wr (2x3)
        float M[2][3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            M[0][c] = j00 * vm[4 * c + 0] + j02 * vm[4 * c + 2];
            M[1][c] = j11 * vm[4 * c + 1] + j12 * vm[4 * c + 2];
        }
        const float *c6 = p.in.cov3D + (size_t)v * d.vs_cov + (size_t)ce * (size_t)i;
        const float sxx = c6[0] * scale2, sxy = c6[1] * scale2, sxz = c6[2] * scale2;
        const float syy = c6[ce == 9 ? 4 : 3] * scale2, syz = c6[ce == 9 ? 5 : 4] * scale2, szz = c6[ce == 9 ? 8 : 5] * scale2;
        const float S[3][3] = {{sxx, sxy, sxz}, {sxy, syy, syz}, {sxz, syz, szz}};
        float MS[2][3];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) MS[r][c] = M[r][0] * S[0][c] + M[r][1] * S[1][c] + M[r][2] * S[2][c];
        const float a = MS[0][0] * M[0][0] + MS[0][1] * M[0][1] + MS[0][2] * M[0][2] + LSR_LOWPASS;
        const float b = MS[0][0] * M[1][0] + MS[0][1] * M[1][1] + MS[0][2] * M[1][2];
        const float c = MS[1][0] * M[1][0] + MS[1][1] * M[1][1] + MS[1][2] * M[1][2] + LSR_LOWPASS;
        const float det = a * c - b * b;
        // The compositing backward leaves MOMENTS of u = opacity * G * dL/dalpha over the pixels:
        //   r0 = (sum u dx, sum u dy, sum u dx^2, sum u dx dy), r1.x = sum u dy^2   (d = mean - pixel).
        // With the conic (A, B, C) = (c, -b, a) / det constant over the pixels:
        //   dL/dmean_pix = -(A m1x + B m1y, B m1x + C m1y),  dL/d(A, B, C) = -(m2xx / 2, m2xy, m2yy / 2)
        const float4 gcon = make_float4(-0.5f * r0.z, -r0.w, -0.5f * r1.x, 0.0f);
        // one select covers both exits: a culled Gaussian or a singular covariance contributes nothing
        // (det_inv = 0 zeroes the conic, hence gpx / gpy and, through d2, dL/d(a, b, c))
        const float det_inv = (vis && det != 0.0f) ? rcp(det) : 0.0f;
        const float cA = c * det_inv, cB = -b * det_inv, cC = a * det_inv;
        const float gpx = -(cA * r0.x + cB * r0.y);
        const float gpy = -(cB * r0.x + cC * r0.y);
        const float d2 = det_inv * det_inv;
        // conic = (c, -b, a) / det
        const float dL_da = d2 * (gcon.x * (-c * c) + gcon.y * (b * c) + gcon.z * (det - a * c));
        const float dL_dc = d2 * (gcon.x * (det - a * c) + gcon.y * (a * b) + gcon.z * (-a * a));
        const float dL_db = d2 * (gcon.x * (2.0f * b * c) - gcon.y * (det + 2.0f * b * b) + gcon.z * (2.0f * a * b));
        // packed Sigma gradient: diagonal once, off-diagonals twice (stored once)
        float gm[3], gc[6];
        gc[0] = dL_da * M[0][0] * M[0][0] + dL_db * M[0][0] * M[1][0] + dL_dc * M[1][0] * M[1][0];
        gc[3] = dL_da * M[0][1] * M[0][1] + dL_db * M[0][1] * M[1][1] + dL_dc * M[1][1] * M[1][1];
        gc[5] = dL_da * M[0][2] * M[0][2] + dL_db * M[0][2] * M[1][2] + dL_dc * M[1][2] * M[1][2];
        gc[1] = 2.0f * dL_da * M[0][0] * M[0][1] + dL_db * (M[0][0] * M[1][1] + M[0][1] * M[1][0]) + 2.0f * dL_dc * M[1][0] * M[1][1];
        gc[2] = 2.0f * dL_da * M[0][0] * M[0][2] + dL_db * (M[0][0] * M[1][2] + M[0][2] * M[1][0]) + 2.0f * dL_dc * M[1][0] * M[1][2];
        gc[4] = 2.0f * dL_da * M[0][1] * M[0][2] + dL_db * (M[0][1] * M[1][2] + M[0][2] * M[1][1]) + 2.0f * dL_dc * M[1][1] * M[1][2];
        // dL/dM, then dL/dJ = dM * Wr^T
        float dM[2][3];
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) {
            dM[0][cc] = 2.0f * MS[0][cc] * dL_da + MS[1][cc] * dL_db;
            dM[1][cc] = 2.0f * MS[1][cc] * dL_dc + MS[0][cc] * dL_db;
        }
        // Wr[k][cc] = vm[4cc + k]
        const float dJ00 = dM[0][0] * vm[0] + dM[0][1] * vm[4] + dM[0][2] * vm[8];
        const float dJ02 = dM[0][0] * vm[2] + dM[0][1] * vm[6] + dM[0][2] * vm[10];
        const float dJ11 = dM[1][0] * vm[1] + dM[1][1] * vm[5] + dM[1][2] * vm[9];
        const float dJ12 = dM[1][0] * vm[2] + dM[1][1] * vm[6] + dM[1][2] * vm[10];
        const float dL_dtx = xm * (-focal_x * itz2) * dJ02;
        const float dL_dty = ym * (-focal_y * itz2) * dJ12;
        float dL_dtz = -focal_x * itz2 * dJ00 - focal_y * itz2 * dJ11 +
                       (2.0f * focal_x * tx) * itz3 * dJ02 + (2.0f * focal_y * ty) * itz3 * dJ12;
        dL_dtz += r1.z;
        // t = Wr p + trans  =>  dL/dp[cc] = sum_k Wr[k][cc] dt[k] = vm[4cc + k] dt[k]
#pragma unroll
        for (int cc = 0; cc < 3; ++cc)
            gm[cc] = vm[4 * cc + 0] * dL_dtx + vm[4 * cc + 1] * dL_dty + vm[4 * cc + 2] * dL_dtz;
        // ---- NDC mean path ----
        const float h0 = pm[0] * p0 + pm[4] * p1 + pm[8] * p2 + pm[12];
        const float h1 = pm[1] * p0 + pm[5] * p1 + pm[9] * p2 + pm[13];
        const float h3 = pm[3] * p0 + pm[7] * p1 + pm[11] * p2 + pm[15];
        const float m_w = rcp(h3 + 0.0000001f);
        const float mul1 = h0 * m_w * m_w, mul2 = h1 * m_w * m_w;
        const float m2x = gpx * (0.5f * d.width);   // d pixel / d ndc
        const float m2y = gpy * (0.5f * d.height);
#pragma unroll
        for (int cc = 0; cc < 3; ++cc)
            gm[cc] += (pm[4 * cc + 0] * m_w - pm[4 * cc + 3] * mul1) * m2x +
                      (pm[4 * cc + 1] * m_w - pm[4 * cc + 3] * mul2) * m2y;
        // gradients are w.r.t. the UNSCALED inputs: mean_scaled = s * mean, cov_scaled = s^2 * cov;
        // a culled Gaussian's values are discarded here (its arithmetic may have produced anything)
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) gm[cc] = vis ? gm[cc] * scale : 0.0f;
#pragma unroll
        for (int k = 0; k < 6; ++k) gc[k] = vis ? gc[k] * scale2 : 0.0f;
        if (d.vs_means != 0) {
            float *o3 = p.g.means3D + (size_t)v * d.vs_means + 3 * (size_t)i;
            if (live) { o3[0] = gm[0]; o3[1] = gm[1]; o3[2] = gm[2]; }
        } else { am[0] += gm[0]; am[1] += gm[1]; am[2] += gm[2]; }
        if (d.vs_cov != 0) {
            float *o6 = p.g.cov3D + (size_t)v * d.vs_cov + (size_t)ce * (size_t)i;
            if (live) {
                if (ce == 9) {   // upper triangle carries the gradient (that is what the packing reads)
                    o6[0] = gc[0]; o6[1] = gc[1]; o6[2] = gc[2]; o6[3] = 0.0f; o6[4] = gc[3]; o6[5] = gc[4];
                    o6[6] = 0.0f; o6[7] = 0.0f; o6[8] = gc[5];
                } else {
#pragma unroll
                    for (int k = 0; k < 6; ++k) o6[k] = gc[k];
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < 6; ++k) ac[k] += gc[k];
        }
        if (p.g.means2D && live) {
            float *o2 = p.g.means2D + 3 * o;
            o2[0] = vis ? m2x : 0.0f; o2[1] = vis ? m2y : 0.0f; o2[2] = 0.0f;
        }
    }
    // ---- shared inputs: groups 1.. leave their partial sums in LDS, group 0 adds them in order and stores ----
    float *acc[kPreBwdShared];
    {
        int n = 0;
        for (int k = 0; k < 3; ++k) acc[n++] = &am[k];
        for (int k = 0; k < 6; ++k) acc[n++] = &ac[k];
        acc[n++] = &aop;
        for (int k = 0; k < kPreBwdFeat; ++k) acc[n++] = &af[k];
        for (int k = 0; k < 3; ++k) acc[n++] = &acol[k];
    }
    if (PARTS > 1) {
        if (part > 0) {
#pragma unroll
            for (int k = 0; k < kPreBwdShared; ++k) s_part[((part - 1) * kPreBwdShared + k) * LPP + slot] = *acc[k];
        }
        __syncthreads();
        if (part == 0) {
#pragma unroll
            for (int q = 1; q < PARTS; ++q)
#pragma unroll
                for (int k = 0; k < kPreBwdShared; ++k) *acc[k] += s_part[((q - 1) * kPreBwdShared + k) * LPP + slot];
        }
    }
    const bool writer = live && part == 0;
    if (d.vs_means == 0 && writer) {
        float *o3 = p.g.means3D + 3 * (size_t)i;
        o3[0] = am[0]; o3[1] = am[1]; o3[2] = am[2];
    }
    if (d.vs_cov == 0 && writer) {
        float *o6 = p.g.cov3D + (size_t)ce * (size_t)i;
        if (ce == 9) {
            o6[0] = ac[0]; o6[1] = ac[1]; o6[2] = ac[2]; o6[3] = 0.0f; o6[4] = ac[3]; o6[5] = ac[4];
            o6[6] = 0.0f; o6[7] = 0.0f; o6[8] = ac[5];
        } else {
#pragma unroll
            for (int k = 0; k < 6; ++k) o6[k] = ac[k];
        }
    }
    if (d.vs_opac == 0 && writer) p.g.opacities[i] = aop != 0.0f ? aop * rcp(p.in.opacities[i]) : 0.0f;
    if (feat_reg && writer) {
        float *gf = p.g.features + (size_t)i * d.feat_channels;
#pragma unroll
        for (int c = 0; c < kPreBwdFeat; ++c)
            if (c < d.feat_channels) gf[c] = af[c];
    }
    if (d.color_mode == LSR_COLOR_PRECOMP && d.vs_color == 0 && writer) {
        float *gcp = p.g.color + 3 * (size_t)i;
        gcp[0] = acol[0]; gcp[1] = acol[1]; gcp[2] = acol[2];
    }
}

hipError_t launch_preprocess_backward(const lsr_dims &d_all, const lsr_inputs &in, const char *geom,
                                      const int32_t *radii, const char *grad,
                                      const lsr_in_grads &gin, hipStream_t s) {
    if (d_all.num_gaussians == 0) return hipSuccess;
    const GeomLayout L = geom_layout(d_all);
    const GradLayout R = grad_layout(d_all);
    const lsr_dims d = group_dims(d_all);      // what one view group (blockIdx.y) looks like to the kernel
    PreBwdParams p;
    p.d = d; p.in = in; p.radii = radii;
    p.gs = group_strides(d_all);
    p.geo = (const float *)(geom + L.rec); p.geo_floats = L.rec_floats;
    p.rec = (const float *)(grad + R.rec); p.rec_floats = R.rec_floats;
    p.zero_rec = (const float *)(grad + R.fixed - 256);
    p.g = gin;
    prof_begin(kStPreprocessBwd, s);
    const int parts = d.num_views >= 4 ? 4 : (d.num_views >= 2 ? 2 : 1);
    const int per_block = kPreBwdThreads / parts;   // Gaussians per block
    const dim3 grid((unsigned)((d.num_gaussians + per_block - 1) / per_block), (unsigned)num_view_groups(d_all));
    if (parts == 4) hipLaunchKernelGGL(k_preprocess_bwd<4>, grid, dim3(kPreBwdThreads), 0, s, p);
    else if (parts == 2) hipLaunchKernelGGL(k_preprocess_bwd<2>, grid, dim3(kPreBwdThreads), 0, s, p);
    else hipLaunchKernelGGL(k_preprocess_bwd<1>, grid, dim3(kPreBwdThreads), 0, s, p);
    prof_end(kStPreprocessBwd, s);
    return hipGetLastError();
}

}  // namespace lsr
