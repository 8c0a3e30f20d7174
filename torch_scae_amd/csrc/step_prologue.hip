// Prologue of a training step: everything that depends on nothing but the previous
// step's parameter update, in ONE launch ahead of the replayed step graph --
//   (a) the batch hand-over: image floats + int64 labels into the step's resident input
//       buffers (scae_stage_batch);
//   (b) the step's presence-noise draws (scae_uniform_f32: part_encoder.py:106,
//       object_decoder.py:201);
//   (c) the parameter-only folding products of the output attention
//       (scae_seed_fold_fwd_f32: set_transformer.py:218-223);
//   (d) the image layer of the CNN encoder, reading the batch where the caller holds it,
//       with the filter re-layouts of the other layers
//       (scae_conv3x3_first_fwd_relayout_f32: part_encoder.py:26-44).
// They have no dependencies between them and are each a few workgroups to a few hundred: as
// separate launches they cost a dependent-dispatch floor each (~5 + 5 + 15 + 11 us at
// cfg-2); here they are block ranges of one grid.  Any part may be absent.
#include "common.h"
#include "noise_dev.h"
#include "seed_fold_dev.h"
#include "conv_first_dev.h"

namespace {
constexpr int NT = 256;
static_assert(scae_noise::NT == NT && scae_fold::NT == NT, "one block size for all parts");

struct Prologue {
  float *dst_image;
  const float *src_image;
  long n_image;
  int64_t *dst_label;
  const int64_t *src_label;
  long n_label;
  float *noise;
  int64_t n_noise;
  uint64_t *noise_state;
  scae_seed_fold_desc fold;
  scae_fold::Plan plan;
  // the image layer: first_img / first_w / first_bias -> first_out; re-layouts rl
  const float *first_img, *first_w, *first_bias;
  float *first_out;
  unsigned short *first_out_h;   // (nullable: bf16 instead of fp32)
  scae_first::ConvGeom first_g;
  scae_first::RelayoutBatch rl;
  int n_first, rb;
  int nb_stage, nb_noise, nb_fold, nb_first;
};

// CIN: input channels of the image layer (0: no image layer in this launch)
template <int CIN>
__global__ __launch_bounds__(NT) void step_prologue_kernel(Prologue p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  int blk = blockIdx.x;
  if (blk < p.nb_fold) {   // (the longest jobs first)
    scae_fold::forward_block_any(p.fold, p.plan, blk, lds);
    return;
  }
  blk -= p.nb_fold;
  if (CIN > 0) {
    if (blk < p.n_first) {
      scae_first::fwd_block<(CIN > 0 ? CIN : 1)>(p.first_img, p.first_w, p.first_bias, p.first_out,
                                               p.first_g, blk, lds, p.first_out_h);
      return;
    }
    if (blk < p.nb_first) {
      const int w_ = blk - p.n_first;
      scae_first::relayout_batch(p.rl, w_ / p.rb, (w_ % p.rb) * NT + threadIdx.x);
      return;
    }
    blk -= p.nb_first;
  }
  if (blk < p.nb_noise) {
    scae_noise::uniform_block(p.noise, p.n_noise, p.noise_state, blk, p.nb_noise);
    return;
  }
  blk -= p.nb_noise;
  const long tid = (long)blk * NT + threadIdx.x, stride = (long)p.nb_stage * NT;
  const bool vec = (((size_t)p.dst_image | (size_t)p.src_image) & 15) == 0;
  const long n4 = vec ? p.n_image >> 2 : 0;
  for (long i = tid; i < n4; i += stride)
    reinterpret_cast<float4 *>(p.dst_image)[i] = reinterpret_cast<const float4 *>(p.src_image)[i];
  for (long i = 4 * n4 + tid; i < p.n_image; i += stride) p.dst_image[i] = p.src_image[i];
  for (long i = tid; i < p.n_label; i += stride) p.dst_label[i] = p.src_label[i];
}
}  // namespace

extern "C" int scae_step_prologue_first_f32(float *dst_image, const float *src_image,
                                            int64_t n_image, int64_t *dst_label,
                                            const int64_t *src_label, int64_t n_label,
                                            float *noise, int64_t n_noise, uint64_t *noise_state,
                                            const scae_seed_fold_desc *fold,
                                            const scae_first_layer_desc *first, void *stream) {
  Prologue p{};
  size_t lds = 0;
  if (n_image > 0) {
    SCAE_REQUIRE(dst_image && src_image && n_label >= 0 &&
                 (n_label == 0 || (dst_label && src_label)));
    long blocks = (n_image / 4 + NT - 1) / NT;
    p.nb_stage = (int)(blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks));
    p.dst_image = dst_image, p.src_image = src_image, p.n_image = (long)n_image;
    p.dst_label = dst_label, p.src_label = src_label, p.n_label = (long)n_label;
  } else {
    SCAE_REQUIRE(n_image == 0 && n_label == 0);
  }
  if (n_noise > 0) {
    SCAE_REQUIRE(noise && noise_state);
    p.noise = noise, p.n_noise = n_noise, p.noise_state = noise_state;
    p.nb_noise = scae_noise::blocks_for(n_noise);
  } else {
    SCAE_REQUIRE(n_noise == 0);
  }
  if (fold) {
    const scae_seed_fold_desc &a = *fold;
    SCAE_REQUIRE(a.seeds && a.wq && a.bq && a.wk && a.bk && a.wv && a.bv && a.wo && a.bo &&
                 a.w2 && a.b2 && a.q && a.wkf && a.bkf && a.wvf && a.bvf && a.wv2e);
    if (!scae_seed_fold_supported(a.O, a.C, a.D)) return SCAE_ERR_UNSUPPORTED;
    p.fold = a;
    p.plan = scae_fold::plan(a.C, a.D);
    p.nb_fold = p.plan.blocks();
    lds = scae_fold::lds_bytes(a.C, a.D);
  }
  int cin = 0;
  if (first) {
    const scae_first_layer_desc &f = *first;
    SCAE_REQUIRE(f.img && f.w && f.bias && f.out && f.B > 0 && f.IH >= 3 && f.IW >= 3 &&
                 f.Cout > 0 && f.stride > 0 && f.n_layers >= 0 && f.n_layers <= 8);
    if (f.Cin < 1 || f.Cin > 4 || f.Cout % 64) return SCAE_ERR_UNSUPPORTED;
    const size_t img_lds = (size_t)f.Cin * f.IH * f.IW * sizeof(float);
    if (img_lds > 64 * 1024) return SCAE_ERR_UNSUPPORTED;
    lds = lds > img_lds ? lds : img_lds;
    p.first_img = f.img, p.first_w = f.w, p.first_bias = f.bias, p.first_out = f.out;
    p.first_out_h = f.out_h;
    p.first_g = scae_first::ConvGeom{f.B, f.IH, f.IW, (f.IH - 3) / f.stride + 1,
                                     (f.IW - 3) / f.stride + 1, f.Cin, f.Cout, f.stride};
    p.n_first = f.B * scae_first::first_split(f.B, f.Cout).slices;
    if (f.n_layers > 0) {
      p.rb = scae_first::fill_relayout(p.rl, f.n_layers, f.rw, f.rwf, f.rwd, f.rCout, f.rCin,
                                       f.out_h ? f.rwfh : nullptr, f.out_h ? f.rwdh : nullptr);
      SCAE_REQUIRE(p.rb > 0);
    }
    p.nb_first = p.n_first + f.n_layers * p.rb;
    cin = f.Cin;
  }
  const int grid = p.nb_stage + p.nb_noise + p.nb_fold + p.nb_first;
  SCAE_REQUIRE(grid > 0);
#define SCAE_PROLOGUE(CI)                                                                     \
  case CI: {                                                                                  \
    if (lds > 48 * 1024) {                                                                    \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(step_prologue_kernel<CI>), \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      if (e != hipSuccess) return (int)e;                                                     \
    }                                                                                         \
    scae::launch(step_prologue_kernel<CI>, dim3(grid), dim3(NT), lds, (hipStream_t)stream, p); \
  } break;
  switch (cin) {
    SCAE_PROLOGUE(0) SCAE_PROLOGUE(1) SCAE_PROLOGUE(2) SCAE_PROLOGUE(3) SCAE_PROLOGUE(4)
  }
#undef SCAE_PROLOGUE
  return scae_launch_status();
}

extern "C" int scae_step_prologue_f32(float *dst_image, const float *src_image, int64_t n_image,
                                      int64_t *dst_label, const int64_t *src_label,
                                      int64_t n_label, float *noise, int64_t n_noise,
                                      uint64_t *noise_state, const scae_seed_fold_desc *fold,
                                      void *stream) {
  return scae_step_prologue_first_f32(dst_image, src_image, n_image, dst_label, src_label,
                                      n_label, noise, n_noise, noise_state, fold, nullptr, stream);
}
