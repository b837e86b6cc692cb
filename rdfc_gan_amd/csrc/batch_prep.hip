// batch_prep.hip -- the training-batch preparation of RDFC-GAN/lib/datasets/nyuv2/nyuv2_dataset_training.py:75-134 on the
// device: per sample, cut the black border, rotate by a random angle (nearest), resize the smaller edge (bilinear; labels
// nearest), centre-crop, convert, build the validity masks and normalise -- upstream does this per sample with PIL on the
// dataloader's CPU workers.  One thread per output pixel; every PIL rounding rule is kept so that the result is bit-identical
// to the host pipeline (oracle/batch_prep.py is pinned against PIL itself):
//   * rotation: Geometry.c affine_fixed -- 16.16 fixed-point source coordinates (the six integers come from the host, one
//     set per sample: they are functions of the angle only);
//   * bilinear resize: Resample.c -- horizontal pass then vertical pass over precomputed tap tables; 8-bit images use 22-bit
//     integer coefficients and are clipped to uint8 after EACH pass, 'F' images accumulate in double in tap order and are
//     stored as float between the passes;
//   * nearest resize (labels): Geometry.c ImagingScaleAffine's index tables (host-precomputed: sequential double adds).
// Work per output pixel: <= ks x ks (5 x 5) rotated source lookups shared by the eight image channels, + one for the label.
#include "common.h"

namespace {

constexpr int NT = 256, PBITS = 22, MAXK = 8;

struct Tabs {
    const int *hb, *hk8, *vb, *vk8, *xtab, *ytab;
    const double *hkd, *vkd;
};

__device__ __forceinline__ int clip8(int v) {
    v >>= PBITS;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

__global__ __launch_bounds__(NT) void prep_kernel(rdfc_prep_desc d, Tabs t, const uint8_t *__restrict__ rgb, const uint8_t *__restrict__ nrm,
                                                  const uint16_t *__restrict__ raw, const uint16_t *__restrict__ gt,
                                                  const uint8_t *__restrict__ lab, const int *__restrict__ affine,
                                                  const float *__restrict__ ms /* rgb mean[3], rgb std[3], depth mean, depth std */,
                                                  float *__restrict__ o_rgb, float *__restrict__ o_nrm, float *__restrict__ o_raw,
                                                  float *__restrict__ o_gt, uint8_t *__restrict__ o_dmask, uint8_t *__restrict__ o_nmask,
                                                  long long *__restrict__ o_lab) {
#pragma clang fp contract(off)
    const size_t HWo = (size_t)d.Ho * d.Wo, total = (size_t)d.B * HWo;
    for (size_t idx = (size_t)blockIdx.x * NT + threadIdx.x; idx < total; idx += (size_t)gridDim.x * NT) {
        const int b = (int)(idx / HWo);
        const int p = (int)(idx - (size_t)b * HWo);
        const int y = p / d.Wo, x = p - y * d.Wo;
        const int oy = y + d.cy0, ox = x + d.cx0;                       // coordinates in the resized image
        const int *af = affine + 6 * b;
        const long long a0 = af[0], a1 = af[1], a2 = af[2], a3 = af[3], a4 = af[4], a5 = af[5];
        const size_t img = (size_t)b * d.Hraw * d.Wraw;
        // source pixel (raw-image offset, or -1) of rotated-image position (row, col)
        auto rot = [&](int row, int col) -> long long {
            const long long xin = (a2 + row * a1 + col * a0) >> 16, yin = (a5 + row * a4 + col * a3) >> 16;
            if (xin < 0 || xin >= d.Wc || yin < 0 || yin >= d.Hc) return -1;
            return (long long)(yin + d.top) * d.Wraw + (xin + d.left);
        };
        const int y0 = t.vb[2 * oy], ny = t.vb[2 * oy + 1], x0 = t.hb[2 * ox], nx = t.hb[2 * ox + 1];
        int av[6];
        double dv[2] = {0.0, 0.0};
#pragma unroll
        for (int c = 0; c < 6; ++c) av[c] = 1 << (PBITS - 1);
        for (int r = 0; r < ny; ++r) {
            int ah[6];
            double dh[2] = {0.0, 0.0};
#pragma unroll
            for (int c = 0; c < 6; ++c) ah[c] = 1 << (PBITS - 1);
            for (int c = 0; c < nx; ++c) {
                const long long s = rot(y0 + r, x0 + c);
                const int k8 = t.hk8[ox * d.ks + c];
                const double kd = t.hkd[ox * d.ks + c];
                if (s >= 0) {
                    const uint8_t *pr = rgb + (img + s) * 3, *pn = nrm + (img + s) * 3;
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) {
                        ah[ch] += (int)pr[ch] * k8;
                        ah[3 + ch] += (int)pn[ch] * k8;
                    }
                    dh[0] += (double)((float)raw[img + s] / 1000.0f) * kd;
                    dh[1] += (double)((float)gt[img + s] / 1000.0f) * kd;
                }                                                       // (fill value 0: adds nothing, in either arithmetic)
            }
            const int k8 = t.vk8[oy * d.ks + r];
            const double kd = t.vkd[oy * d.ks + r];
#pragma unroll
            for (int c = 0; c < 6; ++c) av[c] += clip8(ah[c]) * k8;
            dv[0] += (double)(float)dh[0] * kd;
            dv[1] += (double)(float)dh[1] * kd;
        }
        const size_t o3 = (size_t)b * 3 * HWo + p;
        bool nvalid = false;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const float vr = (float)clip8(av[ch]) / 255.0f, vn = (float)clip8(av[3 + ch]) / 255.0f;
            nvalid = nvalid || vn > 0.f;
            o_rgb[o3 + ch * HWo] = (vr - ms[ch]) / ms[3 + ch];
            o_nrm[o3 + ch * HWo] = (vn - ms[ch]) / ms[3 + ch];
        }
        const float fr = (float)dv[0], fg = (float)dv[1];
        o_raw[idx] = (fr - ms[6]) / ms[7];
        o_gt[idx] = (fg - ms[6]) / ms[7];
        o_dmask[idx] = fg > 0.0001f;
        o_nmask[idx] = nvalid;
        const int ly = t.ytab[oy], lx = t.xtab[ox];
        long long lv = 0;
        if (ly >= 0 && ly < d.Hc && lx >= 0 && lx < d.Wc) {
            const long long s = rot(ly, lx);
            if (s >= 0) lv = lab[img + s];
        }
        o_lab[idx] = lv;
    }
}

}  // namespace

extern "C" int rdfc_nyu_prepare_batch(const rdfc_prep_desc *d, const uint8_t *rgb, const uint8_t *normal, const uint16_t *raw_mm,
                                      const uint16_t *gt_mm, const uint8_t *labels, const int32_t *affine, const int32_t *h_bounds,
                                      const int32_t *h_k8, const double *h_kd, const int32_t *v_bounds, const int32_t *v_k8,
                                      const double *v_kd, const int32_t *x_nearest, const int32_t *y_nearest, const float *mean_std,
                                      float *out_rgb, float *out_normal, float *out_raw, float *out_gt, uint8_t *out_depth_mask,
                                      uint8_t *out_normal_mask, int64_t *out_labels, void *stream) {
    RDFC_REQUIRE(d != nullptr && d->B > 0 && d->Ho > 0 && d->Wo > 0 && d->ks > 0 && d->ks <= MAXK, "rdfc_nyu_prepare_batch: bad descriptor");
    RDFC_REQUIRE(d->top >= 0 && d->left >= 0 && d->top + d->Hc <= d->Hraw && d->left + d->Wc <= d->Wraw, "rdfc_nyu_prepare_batch: the cut window leaves the raw image");
    RDFC_REQUIRE(d->cy0 >= 0 && d->cx0 >= 0 && d->cy0 + d->Ho <= d->Hr && d->cx0 + d->Wo <= d->Wr, "rdfc_nyu_prepare_batch: the centre crop leaves the resized image");
    RDFC_REQUIRE(rgb && normal && raw_mm && gt_mm && labels && affine && mean_std, "rdfc_nyu_prepare_batch: null input");
    RDFC_REQUIRE(out_rgb && out_normal && out_raw && out_gt && out_depth_mask && out_normal_mask && out_labels, "rdfc_nyu_prepare_batch: null output");
    Tabs t{h_bounds, h_k8, v_bounds, v_k8, x_nearest, y_nearest, h_kd, v_kd};
    const size_t total = (size_t)d->B * d->Ho * d->Wo;
    const int nb = (int)min((size_t)8192, (total + NT - 1) / NT);
    hipLaunchKernelGGL(prep_kernel, dim3(nb), dim3(NT), 0, (hipStream_t)stream, *d, t, rgb, normal, raw_mm, gt_mm, labels, affine, mean_std,
                       out_rgb, out_normal, out_raw, out_gt, out_depth_mask, out_normal_mask, (long long *)out_labels);
    RDFC_LAUNCH_CHECK("rdfc_nyu_prepare_batch");
    return 0;
}
