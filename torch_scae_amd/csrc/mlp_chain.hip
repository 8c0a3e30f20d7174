// K7b -- a chain of per-group MLP layers in ONE launch, gfx950.
//
// CapsuleLayer (object_decoder.py:86-107, :137-158) runs, per object capsule, two small
// ReLU MLPs back to back (feature -> hidden -> capsule parameters -> hidden -> vote
// parameters; the reference: a Python loop of 4*O nn.Linear calls).  As four batched
// GEMM launches (K7, gemm_mfma.hip) the chain costs four dependent-dispatch floors and
// three round trips of the hidden activations through L2 for ~0.4 GFLOP; here a
// workgroup carries 16 (32 at large batches: every weight element then serves two MFMA
// row blocks, and the group's weight set is streamed from L2 half as often) batch rows of
// ONE group through all the layers:
//   * the 16 x K activation block lives in LDS (two ping-pong buffers); each weight element
//     is used exactly once per workgroup: it is loaded coalesced, parked in registers, and
//     passes through a wave-private LDS tile on its way to the MFMA fragment;
//   * v_mfma_f32_16x16x4_f32 (exact fp32 products): wave w (of 8) owns the 16-column output
//     tiles w, w + 8, ...; lane (r, q) reads A[r][k0 + 4q .. + 3] as one ds_read_b128 and the
//     matching four k of its weight row / column -- which four k an instruction contracts
//     is free as long as A and B agree;
//   * the weights of the next (tile, 128-wide k-chunk) item are in flight while the current
//     one is multiplied, across layer boundaries too, and 8 waves per workgroup cover each
//     other's waits;
//   * every layer's output is written once to global memory (the backward pass needs the
//     ReLU outputs; the data-gradient form needs the pre-activation gradients for the
//     weight-gradient GEMMs) and kept in LDS for the next layer.
// The same kernel runs the data-gradient chain of the backward pass: layer l then
// contracts over the rows of W_l (g_prev = gate(g W_l)), the gate being the ReLU output
// saved by the forward pass.
#include "common.h"
#include "capsule_votes_dev.h"

namespace {
typedef float f32x4 __attribute__((ext_vector_type(4)));
#ifndef SCAE_CHAIN_CH
#define SCAE_CHAIN_CH 4   // 16-wide k steps per item: 64-wide chunks, 35 KB of weight tiles (8: 128-wide, 68 KB)
#endif
#ifndef SCAE_CHAIN_ABL
// timing ablations (tools/chain_time.py, profiles/r05/chain_ablations.txt; results are garbage):
// 1: no weight loads, 2: no MFMAs, 3: no MFMAs and no output stores
#define SCAE_CHAIN_ABL 0
#endif
#ifndef SCAE_CHAIN_LB4
#define SCAE_CHAIN_LB4 1
#endif
constexpr int NW = 8, NT = 64 * NW, MAXL = 4, CH = SCAE_CHAIN_CH;   // (16 RBT batch rows per workgroup)
constexpr int KC = 16 * CH;          // contraction columns of one item
constexpr int QPR = KC / 4;          // 16-byte quads per weight row of an item (forward form)
constexpr int RPI = 64 / QPR;        // weight rows one wave instruction fetches (forward form)
static_assert(CH == 8 || CH == 4, "chunk widths of 128 or 64");
constexpr int WLD = KC + 4;   // row stride of a wave's weight tile: (WLD / 4) odd

struct Layer {
  const float *w;
  long w_gs;
  int ldw, K, N;   // K: contraction length, N: output columns
  const float *bias;
  long bias_gs;
  int bias_ld;
  const float *gate;
  long gate_gs, gate_bs;
  float *out;
  long out_gs, out_bs;
  int relu, vec;   // vec: 16-byte weight loads are legal (forward form)
};
struct Chain {
  Layer l[MAXL];
  const float *in;
  long in_gs, in_bs;
  int n, in_dim, B, G;
  int stride[2];   // floats per LDS row of activation buffer 0 (input, odd layers' outputs) / 1
  int wscratch;    // the vote blocks take their scratch in the weight tiles (narrow buffer 1)
  // the capsule votes (K3) at the end of the forward chain / at the head of the
  // data-gradient chain
  int votes;
  int bf16;        // the layer products on v_mfma_f32_16x16x16_bf16 (operands rounded at the MFMA)
  scae_votes::VoteArgs va;
  scae_votes::VoteOut vo;
  scae_votes::VoteGrads vg;
};

__device__ __forceinline__ float4 ld4(const float *p) {
  return *reinterpret_cast<const float4 *>(p);
}
// bf16 form (BASELINE.json configs[2]): the same four k per lane, rounded to bf16 (nearest
// even) on their way from LDS, as ONE v_mfma_f32_16x16x16_bf16 (fp32 accumulate)
typedef short bf16x4 __attribute__((ext_vector_type(4)));
// (the conversion is left to the compiler -- two v_cvt_pk_bf16_f32 -- and not written as
// inline asm: the hazard recogniser does not see into an asm body, and a VALU result consumed
// by the very next MFMA needs wait states it then does not insert: stale operands)
typedef __bf16 bf16v2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16x4 to_bf16(float4 v) {
  const f32x2 a = {v.x, v.y}, b = {v.z, v.w};
  const u32x2 u = {__builtin_bit_cast(unsigned, __builtin_convertvector(a, bf16v2)),
                   __builtin_bit_cast(unsigned, __builtin_convertvector(b, bf16v2))};
  return __builtin_bit_cast(bf16x4, u);
}
__device__ __forceinline__ f32x4 mma16_bf16(f32x4 acc, bf16x4 a, bf16x4 b) {
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, acc, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mma16(f32x4 acc, float4 a, float4 b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
  return acc;
}

// guarded element: an unconditional load from a clamped address, zero when out of range
// (no divergent branch around the load)
__device__ __forceinline__ float ldg(const float *base, size_t idx, bool ok) {
  const float v = base[ok ? idx : 0];
  return ok ? v : 0.f;
}

// Weights + epilogue operands of the item (tile, chunk) of layer L: global -> registers,
// piece i of an item is ONE coalesced wave instruction --
//   forward:        lane = (row 2i + (lane >> 5), k quad lane & 31)    [2 rows x 512 B]
//   data gradient:  lane = (k 16i + (lane >> 2), column quad lane & 3)  [16 rows x 64 B]
// (row-per-lane fragment loads straight from global memory cost the vector L1 one access
// per lane: 57 per instruction, measured).  No element guards: the loads go through a
// buffer descriptor that covers the group's weight matrix -- a lane past its end reads
// zeros -- and what a ragged tile picks up INSIDE the matrix (the next row's head behind a
// short row, columns past N) only ever multiplies the zero padding of the activation
// block or lands in output columns the epilogue masks.  With the tile's last chunk come
// its bias (forward) / its four gate values (data gradient).
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 bload4(rsrc_t rs, int byte_off) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z),
                     __uint_as_float(v.w));
}
template <bool BWD, int RBT>
__device__ __forceinline__ void fetch_item(const Layer &L, const float *W, int g, int b0, int B,
                                           int tile, int ch, int nch, int lane, float4 (&buf)[CH],
                                           float4 (&epi)[RBT]) {
  const int n0 = 16 * tile, k0 = KC * ch, ldw = L.ldw, LN = L.N;
  // rows of the matrix: N (forward) / K (data gradient)
  const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(W), 0, (BWD ? L.K : L.N) * ldw * 4, 0x00020000);
#if SCAE_CHAIN_ABL == 1   // (no weight loads)
  if (ldw >= 0) return;
#endif
  if (!BWD) {
    const int o = ((n0 + lane / QPR) * ldw + k0 + 4 * (lane % QPR)) * 4;
#pragma unroll
    for (int i = 0; i < CH; ++i) buf[i] = bload4(rs, o + RPI * i * ldw * 4);
  } else {
    const int o = ((k0 + (lane >> 2)) * ldw + n0 + 4 * (lane & 3)) * 4;
#pragma unroll
    for (int i = 0; i < CH; ++i) buf[i] = bload4(rs, o + 16 * i * ldw * 4);
  }
  if (ch == nch - 1) {
    const int r = lane & 15, q = lane >> 4, col = n0 + r;
    const bool cok = col < LN;
    if (!BWD) {
      epi[0].x = L.bias ? ldg(L.bias + (size_t)g * L.bias_gs, (size_t)col * L.bias_ld, cok) : 0.f;
    } else {
#pragma unroll
      for (int rb = 0; rb < RBT; ++rb) {
        epi[rb] = make_float4(1.f, 1.f, 1.f, 1.f);
        if (L.gate) {
          const float *gp = L.gate + (size_t)g * L.gate_gs;
          const int b = b0 + 16 * rb + 4 * q;
          epi[rb].x = ldg(gp, (size_t)b * L.gate_bs + col, cok && b < B);
          epi[rb].y = ldg(gp, (size_t)(b + 1) * L.gate_bs + col, cok && b + 1 < B);
          epi[rb].z = ldg(gp, (size_t)(b + 2) * L.gate_bs + col, cok && b + 2 < B);
          epi[rb].w = ldg(gp, (size_t)(b + 3) * L.gate_bs + col, cok && b + 3 < B);
        }
      }
    }
  }
}

// BWD = false: out[b][n] = epi(sum_k in[b][k] W[n][k])      (W rows = outputs)
// BWD = true:  out[b][n] = gate(sum_k in[b][k] W[k][n])     (W rows = contraction)
//
// The layer loop is unrolled (MAXL = 4 copies with the layer's fields in scalar registers
// -- an earlier form that walked a run-time (layer, item) cursor spent ~500 instructions of
// selects and divisions per 32-MFMA item).  A wave's items of a layer are its tiles x
// 128-wide k-chunks; the weights of the next item -- the first item of the next layer at
// the end of a layer: weights do not depend on activations -- are in flight while the
// current one is multiplied, and pass through a wave-private LDS tile [16][128 + 4] on
// their way to the MFMA fragments (transposed on the way in for the data-gradient form).
// One workgroup barrier per layer boundary separates the writes of a layer's output from
// its reads.
// (16-row workgroups: four waves per SIMD, so that two of them share a CU)
template <bool BWD, int RBT, bool BF>
__global__ __launch_bounds__(NT, (RBT == 1 && SCAE_CHAIN_LB4) ? 4 : 2) void chain_kernel(Chain c) {
  constexpr int RB = 16 * RBT;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, wid = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63,
            r = lane & 15, q = lane >> 4;
  const int g = blockIdx.x % c.G, b0 = (blockIdx.x / c.G) * RB;
  // activation buffer p: rows of c.stride[p] floats
  auto buf = [&](int p) { return smem + (p ? RB * c.stride[0] : 0); };
  float *wbase = smem + RB * (c.stride[0] + c.stride[1]);
  float *wtile = wbase + wid * (16 * WLD);   // this wave's weight tile
  // registers -> the wave's LDS tile [16 columns][128 k (+4)]
  auto park = [&](const float4 (&buf4)[CH]) {
    if (!BWD) {
      float *d = wtile + (lane / QPR) * WLD + 4 * (lane % QPR);
#pragma unroll
      for (int i = 0; i < CH; ++i) *reinterpret_cast<float4 *>(d + RPI * i * WLD) = buf4[i];
    } else {   // transposed: (k, 4 columns) -> [column][k]; banks 16 cq + 4 j + k: distinct
      float *d = wtile + 4 * (lane & 3) * WLD + (lane >> 2);
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        d[16 * i] = buf4[i].x;
        d[WLD + 16 * i] = buf4[i].y;
        d[2 * WLD + 16 * i] = buf4[i].z;
        d[3 * WLD + 16 * i] = buf4[i].w;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the wave's own tile
  };
  auto ntiles_of = [&](int li) { return (c.l[li].N + 15) >> 4; };
  auto nch_of = [&](int li) { return (((c.l[li].K + 15) >> 4) + CH - 1) / CH; };

  float4 f[CH], fe[RBT];
#pragma unroll
  for (int rb = 0; rb < RBT; ++rb) fe[rb] = make_float4(0.f, 0.f, 0.f, 0.f);
  // the first weights go out before the input block is staged
  if (wid < ntiles_of(0))
    fetch_item<BWD, RBT>(c.l[0], c.l[0].w + (size_t)g * c.l[0].w_gs, g, b0, c.B, wid, 0, nch_of(0),
                         lane, f, fe);
  if (BWD && c.votes) {
    // the chain's input block = the vote kernel's gradient rows, made here (K3 backward
    // for the block's capsules, 16 at a time; also written to global memory for the
    // weight-gradient GEMM and the bias sums).  Scratch: the other activation buffer
    // (one row block) / the weight tiles, which nobody has parked anything in yet.
#pragma unroll
    for (int rb = 0; rb < RBT; ++rb)
      scae_votes::bwd_block<NT>(c.va, c.vg, smem + 16 * rb * c.stride[0], c.stride[0],
                                c.wscratch ? wbase : buf(1), b0 + 16 * rb, g);
  } else
  {  // the block's input rows, zero padded to a multiple of 16 columns / to RB rows; eight
     // independent loads per thread in flight (a load-store loop pays a memory round trip
     // per iteration)
    const int k16 = (c.in_dim + 15) & ~15, total = RB * k16;
    const float *src = c.in + (size_t)g * c.in_gs;
    for (int e0 = tid; e0 < total; e0 += 8 * NT) {
      float v[8];
      int at[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int e = e0 + j * NT, row = e / k16, k = e - row * k16, b = b0 + row;
        at[j] = row * c.stride[0] + k;
        v[j] = ldg(src, (size_t)b * c.in_bs + k, e < total && b < c.B && k < c.in_dim);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (e0 + j * NT < total) smem[at[j]] = v[j];
    }
  }
#pragma unroll
  for (int li = 0; li < MAXL; ++li) {
    if (li < c.n) {   // (uniform; li is a compile-time constant in each copy)
      const Layer &L = c.l[li];
      const float *W = L.w + (size_t)g * L.w_gs;
      const int nch = nch_of(li), ntiles = ntiles_of(li), nsteps = (L.K + 15) >> 4;
      const bool last = li + 1 == c.n;
      const float *cur = buf(li & 1);
      float *nxt = buf((li + 1) & 1);
      const int cs = c.stride[li & 1], ns = c.stride[(li + 1) & 1];
      float *out = L.out ? L.out + (size_t)g * L.out_gs : nullptr;
      __syncthreads();   // the layer's input block is complete
      f32x4 acc[RBT];
      for (int tile = wid; tile < ntiles; tile += NW) {
        for (int ch = 0; ch < nch; ++ch) {
          float4 epi[RBT];
          park(f);   // (waits for the item's loads)
#pragma unroll
          for (int rb = 0; rb < RBT; ++rb) epi[rb] = fe[rb];
          // the next item's loads fly while this one is multiplied
          if (ch + 1 < nch) {
            fetch_item<BWD, RBT>(L, W, g, b0, c.B, tile, ch + 1, nch, lane, f, fe);
          } else if (tile + NW < ntiles) {
            fetch_item<BWD, RBT>(L, W, g, b0, c.B, tile + NW, 0, nch, lane, f, fe);
          } else if (li + 1 < MAXL && li + 1 < c.n) {
            const Layer &Ln = c.l[li + 1 < MAXL ? li + 1 : li];
            if (wid < ntiles_of(li + 1 < MAXL ? li + 1 : li))
              fetch_item<BWD, RBT>(Ln, Ln.w + (size_t)g * Ln.w_gs, g, b0, c.B, wid, 0,
                                   nch_of(li + 1 < MAXL ? li + 1 : li), lane, f, fe);
          }
          if (ch == 0) {
#pragma unroll
            for (int rb = 0; rb < RBT; ++rb) acc[rb] = (f32x4){0.f, 0.f, 0.f, 0.f};
          }
          const float *arow = cur + r * cs + KC * ch + 4 * q;
          const float *brow = wtile + r * WLD + 4 * q;
#pragma unroll
          for (int s = 0; s < CH; ++s)
            if (ch * CH + s < nsteps) {   // (uniform)
              const float4 bw = ld4(brow + 16 * s);
              if (SCAE_CHAIN_ABL >= 2 && cs >= 0) continue;
              if (BF) {
                const bf16x4 bh = to_bf16(bw);
#pragma unroll
                for (int rb = 0; rb < RBT; ++rb)
                  acc[rb] = mma16_bf16(acc[rb], to_bf16(ld4(arow + 16 * rb * cs + 16 * s)), bh);
              } else {
#pragma unroll
                for (int rb = 0; rb < RBT; ++rb)
                  acc[rb] = mma16(acc[rb], ld4(arow + 16 * rb * cs + 16 * s), bw);
              }
            }
          if (ch != nch - 1) continue;
          const int col = 16 * tile + r;
          const bool cok = col < L.N;
          const float bias = BWD ? 0.f : epi[0].x;
#pragma unroll
          for (int rb = 0; rb < RBT; ++rb) {
            const float gt[4] = {BWD ? epi[rb].x : 1.f, BWD ? epi[rb].y : 1.f, BWD ? epi[rb].z : 1.f,
                                 BWD ? epi[rb].w : 1.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int row = 16 * rb + 4 * q + e, b = b0 + row;
              float v = acc[rb][e] + bias;
              if (L.relu) v = fmaxf(v, 0.f);
              v = cok && gt[e] > 0.f ? v : 0.f;   // (also the zero padding of the next contraction)
              if (!last || (!BWD && c.votes)) nxt[row * ns + col] = v;
              if (SCAE_CHAIN_ABL == 3 && cs >= 0) continue;
              if (out && cok && b < c.B) out[(size_t)b * L.out_bs + col] = v;
            }
          }
        }
      }
      // a wave without a tile in this layer still owes the next layer its first fetch
      if (wid >= ntiles && li + 1 < MAXL && li + 1 < c.n) {
        const Layer &Ln = c.l[li + 1 < MAXL ? li + 1 : li];
        if (wid < ntiles_of(li + 1 < MAXL ? li + 1 : li))
          fetch_item<BWD, RBT>(Ln, Ln.w + (size_t)g * Ln.w_gs, g, b0, c.B, wid, 0,
                               nch_of(li + 1 < MAXL ? li + 1 : li), lane, f, fe);
      }
    }
  }
  if (!BWD && c.votes) {
    // K3 forward for the block's capsules (16 at a time) from their parameter rows in LDS
#pragma unroll
    for (int rb = 0; rb < RBT; ++rb) {
      __syncthreads();
      scae_votes::fwd_block<NT>(c.va, c.vo, buf(c.n & 1) + 16 * rb * c.stride[c.n & 1],
                                c.stride[c.n & 1], c.wscratch ? wbase : buf((c.n + 1) & 1),
                                b0 + 16 * rb, g);
    }
  }
}

int fill(Chain &c, const scae_mlp_chain_desc *d, bool bwd, const scae_votes_desc *v = nullptr) {
  if (!d || (!d->in && !(bwd && v)) || d->n_layers < 1 || d->n_layers > MAXL || d->B <= 0 || d->G <= 0 ||
      d->in_dim <= 0)
    return SCAE_ERR_BAD_ARG;
  int maxdim = d->in_dim, prev = d->in_dim;
  for (int i = 0; i < d->n_layers; ++i) {
    const scae_mlp_chain_layer &s = d->layer[i];
    if (!s.w || s.K <= 0 || s.N <= 0 || s.ldw <= 0) return SCAE_ERR_BAD_ARG;
    if (s.K != prev) return SCAE_ERR_BAD_ARG;   // a layer contracts over its predecessor's width
    if (bwd ? s.ldw < s.N : s.ldw < s.K) return SCAE_ERR_BAD_ARG;
    if (i + 1 == d->n_layers && !s.out) return SCAE_ERR_BAD_ARG;
    Layer &L = c.l[i];
    L.w = s.w, L.w_gs = (long)s.w_gs, L.ldw = s.ldw, L.K = s.K, L.N = s.N;
    L.bias = bwd ? nullptr : s.bias, L.bias_gs = (long)s.bias_gs, L.bias_ld = s.bias_ld > 0 ? s.bias_ld : 1;
    L.gate = bwd ? s.gate : nullptr, L.gate_gs = (long)s.gate_gs, L.gate_bs = (long)s.gate_bs;
    L.out = s.out, L.out_gs = (long)s.out_gs, L.out_bs = (long)s.out_bs;
    L.relu = bwd ? 0 : s.relu;
    // 16-byte weight loads: forward rows of K floats, data-gradient rows of N floats
    L.vec = (s.ldw & 3) == 0 && (s.w_gs & 3) == 0 && ((size_t)s.w & 15) == 0;
    maxdim = maxdim > s.N ? maxdim : s.N;
    prev = s.N;
  }
  if (maxdim > scae_mlp_chain_max_width()) return SCAE_ERR_UNSUPPORTED;
  c.in = d->in, c.in_gs = (long)d->in_gs, c.in_bs = (long)d->in_bs;
  c.n = d->n_layers, c.in_dim = d->in_dim, c.B = d->B, c.G = d->G;
  // (stride / 4) odd: conflict-free b128 row reads.  One stride for both buffers here;
  // launch() narrows each buffer to what it holds when it takes 32 rows per workgroup
  c.stride[0] = c.stride[1] = ((maxdim + 15) & ~15) + 4;
  c.votes = 0;
  c.bf16 = d->bf16 != 0;
  if (v) {
    const int V = v->V, A = 8 * V + 7;
    if (V <= 0 || !v->cpr_static || !v->bias_cvr || !v->bias_caps || !v->bias_vote ||
        !v->bias_scale || v->ld_param < A)
      return SCAE_ERR_BAD_ARG;
    const scae_mlp_chain_layer &e = d->layer[bwd ? 0 : d->n_layers - 1];
    // the chain's (B, G, ld) parameter rows are the vote kernel's all_param
    if ((bwd ? d->in_dim : e.N) != A) return SCAE_ERR_BAD_ARG;
    c.votes = 1;
    c.va = scae_votes::VoteArgs{v->all_param, v->cpr_static, v->bias_cvr, v->bias_caps,
                                v->bias_vote, v->bias_scale, v->noise_caps, v->noise_vote,
                                v->noise_scale, d->B, d->G, V, v->similarity,
                                v->learn_vote_scale, v->allow_deformations, v->ld_param};
    if (!bwd) {
      if (!(v->vote && v->scale && v->vote_presence && v->logit_caps && v->logit_vote &&
            v->reg_partial) || !v->caps_presence != !v->caps_arg)
        return SCAE_ERR_BAD_ARG;
      if (e.out_gs != v->ld_param || e.out_bs != (int64_t)d->G * v->ld_param)
        return SCAE_ERR_BAD_ARG;
      c.vo = scae_votes::VoteOut{v->vote, v->scale, v->vote_presence, v->logit_caps,
                                 v->logit_vote, v->reg_partial, v->caps_presence, v->caps_arg};
    } else {
      if (!(v->all_param && v->gall_param && v->gcpr_in) ||
          (v->gcaps_presence && !v->caps_arg))
        return SCAE_ERR_BAD_ARG;
      c.vg = scae_votes::VoteGrads{v->gvote, v->gscale, v->gvote_presence, v->glogit_caps,
                                   v->glogit_vote, v->greg, v->gcaps_presence, v->caps_arg,
                                   v->gall_param, v->gcpr_in, v->gall_param_gated};
    }
  }
  return SCAE_OK;
}

#ifndef SCAE_CHAIN_RB32_MIN_WGS
#define SCAE_CHAIN_RB32_MIN_WGS 512   // 32-row workgroups only while >= 2 of them per CU remain
#endif
template <bool BWD, int RBT, bool BF>
int launch_rb_bf(const Chain &c, void *stream) {
  constexpr int RB = 16 * RBT;
  const size_t lds = ((size_t)RB * (c.stride[0] + c.stride[1]) + (size_t)NW * 16 * WLD) * sizeof(float);
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(chain_kernel<BWD, RBT, BF>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  const int blocks = c.G * ((c.B + RB - 1) / RB);
  scae::launch((chain_kernel<BWD, RBT, BF>), dim3(blocks), dim3(NT), lds, (hipStream_t)stream, c);
  return scae_launch_status();
}
template <bool BWD, int RBT>
int launch_rb(const Chain &c, void *stream) {
  return c.bf16 ? launch_rb_bf<BWD, RBT, true>(c, stream) : launch_rb_bf<BWD, RBT, false>(c, stream);
}

template <bool BWD>
int launch(const scae_mlp_chain_desc *d, const scae_votes_desc *v, void *stream) {
  Chain c{};
  int rc = fill(c, d, BWD, v);
  if (rc) return rc;
  // Large batches: 32 rows per workgroup.  Buffer 0 holds the input block and the outputs
  // of the odd layers (the last layer's only when the vote kernel reads it from LDS),
  // buffer 1 the even layers' -- each as wide as what it holds; the vote blocks take their
  // scratch in the weight tiles.
  const int want = d->row_tile;   // (16 | 32: a forced row tile -- tests, measurements; 0: by shape)
  if (want != 0 && want != 16 && want != 32) return SCAE_ERR_BAD_ARG;
  Chain c2 = c;
  int dim[2] = {c.in_dim, 16};
  for (int i = 0; i < c.n; ++i)
    if (i + 1 < c.n || (!BWD && c.votes)) dim[(i + 1) & 1] = dim[(i + 1) & 1] > c.l[i].N ? dim[(i + 1) & 1] : c.l[i].N;
  c2.stride[0] = ((dim[0] + 15) & ~15) + 4, c2.stride[1] = ((dim[1] + 15) & ~15) + 4;
  const size_t lds2 = ((size_t)32 * (c2.stride[0] + c2.stride[1]) + (size_t)NW * 16 * WLD) * sizeof(float);
  const bool scratch_ok = !c.votes || (size_t)16 * c.va.V * 7 <= (size_t)NW * 16 * WLD;
  const bool big = c.G * ((c.B + 31) / 32) >= SCAE_CHAIN_RB32_MIN_WGS;
  c2.wscratch = 1;
  // 16 rows with narrow buffers where that lets TWO workgroups share a CU's LDS (16 waves
  // per CU cover each other's dependent fetch / park / multiply sequence: at the cfg-3 shape
  // 5.34 ms per step against 5.38 with 32 rows and one workgroup per CU, 5.48 with 16 rows
  // and one; cfg-5 0.870 against 0.894); 32 rows where only one fits anyway
  const size_t lds1 = ((size_t)16 * (c.stride[0] + c.stride[1]) + (size_t)NW * 16 * WLD) * sizeof(float);
  const size_t lds1n = ((size_t)16 * (c2.stride[0] + c2.stride[1]) + (size_t)NW * 16 * WLD) * sizeof(float);
  const bool narrow16 = scratch_ok && lds1 > 80 * 1024 && lds1n <= 80 * 1024;
  if (lds2 <= 160 * 1024 && scratch_ok && want != 16 && (want == 32 || (big && !narrow16)))
    return launch_rb<BWD, 2>(c2, stream);
  if (narrow16) return launch_rb<BWD, 1>(c2, stream);
  return launch_rb<BWD, 1>(c, stream);   // (one stride, scratch in buffer 1)
}
}  // namespace

// LDS: two 16-row activation blocks of the widest layer + the eight weight tiles <= 160 KiB
extern "C" int scae_mlp_chain_max_width(void) { return 704; }

extern "C" int scae_mlp_chain_fwd_f32(const scae_mlp_chain_desc *desc, void *stream) {
  return launch<false>(desc, nullptr, stream);
}
extern "C" int scae_mlp_chain_bwd_f32(const scae_mlp_chain_desc *desc, void *stream) {
  return launch<true>(desc, nullptr, stream);
}
extern "C" int scae_mlp_chain_votes_fwd_f32(const scae_mlp_chain_desc *desc,
                                            const scae_votes_desc *votes, void *stream) {
  SCAE_REQUIRE(votes);
  return launch<false>(desc, votes, stream);
}
extern "C" int scae_mlp_chain_votes_bwd_f32(const scae_mlp_chain_desc *desc,
                                            const scae_votes_desc *votes, void *stream) {
  SCAE_REQUIRE(votes);
  return launch<true>(desc, votes, stream);
}
