/*
 * mdconv_oracle.c -- plain-C restatement of the reference's modulated deformable
 * convolution (DCNv2) forward and backward.  TEST INFRASTRUCTURE ONLY (see
 * oracle/__init__.py): the HIP product never links or calls this file.
 *
 * The reference implementation is CUDA-only (its CPU entry points raise), so this
 * restates the algorithm of
 *   RDFC-GAN/lib/models/generator/rdf_generator/nlspn/deformconv/src/cuda/
 *     modulated_deform_im2col_cuda.cuh:24-54    bilinear sample, per-corner bounds
 *     modulated_deform_im2col_cuda.cuh:127-194  sampling positions / validity / mask
 *     modulated_deform_im2col_cuda.cuh:56-81,196-254   grad wrt input (scatter)
 *     modulated_deform_im2col_cuda.cuh:83-125,256-328  grad wrt offset and mask
 *     modulated_deform_conv_cuda.cu:75-121      output size, group GEMM, bias
 *     modulated_deform_conv_cuda.cu:193-273     backward composition
 * as direct loops (no column buffer; the column buffer of the reference is the
 * product W^T * grad_out, folded into the loops here).
 *
 * Layouts (all contiguous, NCHW):
 *   input  [B, C, H, W]            weight [Co, C/group, KH, KW]     bias [Co]
 *   offset [B, dg*2*KH*KW, Ho, Wo] channel 2k = dh of tap k, 2k+1 = dw of tap k
 *   mask   [B, dg*KH*KW,   Ho, Wo]
 *   output [B, Co, Ho, Wo]
 *
 * Built for float and double by including the body twice.
 */
#include <math.h>
#include <stddef.h>
#include <string.h>

typedef struct {
    int B, C, H, W, Co, KH, KW, sh, sw, ph, pw, dh, dw, group, dgroup;
} mdconv_shape;

static int out_dim(int in, int pad, int dil, int k, int stride) {
    return (in + 2 * pad - (dil * (k - 1) + 1)) / stride + 1;
}

int mdconv_oracle_out_h(const mdconv_shape *s) { return out_dim(s->H, s->ph, s->dh, s->KH, s->sh); }
int mdconv_oracle_out_w(const mdconv_shape *s) { return out_dim(s->W, s->pw, s->dw, s->KW, s->sw); }

#define REAL float
#define FN(name) name##_f32
#include "mdconv_oracle_body.inc"
#undef REAL
#undef FN

#define REAL double
#define FN(name) name##_f64
#include "mdconv_oracle_body.inc"
#undef REAL
#undef FN
