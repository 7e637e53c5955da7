// Fused MobileNetV2 inverted-residual block for gfx950: expand 1x1 (MFMA) -> LDS -> depthwise 3x3 -> LDS ->
// project 1x1 (MFMA) (+ residual), one workgroup per (model, observation, tile of output rows).
//
// torchvision v0.6.0 `InvertedResidual` (reference call site oatomobile/torch/networks/perception.py:36-51),
// BN folded.  The block is separable over hidden channels:
//     y[p][co] = sum_h Wp[co][h] * relu6(dw_h(relu6(We[h][:] . x[:, :])))[p]
// so the hidden dimension is walked in chunks of HC channels: expand one chunk for the (haloed) input tile into
// LDS, run the 3x3 depthwise on it in LDS, and accumulate that chunk's contribution to the projection in MFMA
// accumulators that live in registers across chunks.  The t-times expanded tensors never reach HBM: per block
// only its input tile (+halo) is read and its output written.
//
// MFMA orientation (v_mfma_f32_16x16x4_f32, exact fp32): products are formed transposed, OUT^T[ch][pixel], with
// A = a 16-channel weight tile and B = 16 pixels; lane (n = lane & 15, q = lane >> 4) then owns 4 consecutive
// output channels of pixel n, i.e. float4 LDS/HBM stores in NHWC.  Both operands are K-contiguous float4 loads
// (k-step (S, r) contracts channels {16S + 4q + r}).
#include "encoder.h"

namespace rip {

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float relu6f(float v) { return fminf(fmaxf(v, 0.f), 6.f); }

struct IrbArgs {
  const float* x;       // [K][B][H_in][H_in][CIN]
  float* y;             // [K][B][H_out][H_out][COUT]
  const float* wbase;   // per-model folded blobs
  size_t model_stride;
  int k0;
  size_t we_off, be_off, wd_off, bd_off, wp_off, bp_off;
  int B, CIN, HID, COUT, H_in, H_out, stride, residual;
  int TH;               // output rows per tile
  int IH, IW;           // input tile extent (with halo): IH = (TH-1)*stride + 3, IW = H_in + 2
  int OP16;             // output pixels per tile rounded up to 16
};

constexpr int MAXT = 12;  // projection accumulator tiles per wave
constexpr int NWAVES = 8;

// NS = ceil(CIN / 16) k-groups of the expand GEMM; HC = hidden channels per chunk; EXPAND=false for the t=1 block
template <int NS, int HC, bool EXPAND>
__global__ __launch_bounds__(NWAVES * 64) void irb_kernel(IrbArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, q = lane >> 4;
  const int k = blockIdx.z, b = blockIdx.y, tile = blockIdx.x;
  const int CIN = a.CIN, COUT = a.COUT, HID = a.HID, H_in = a.H_in, W_out = a.H_out, s = a.stride;
  const int CIN16 = NS * 16;
  const int XLD = (EXPAND ? CIN16 : HC) + 4, HLD = HC + 4;
  const int IH = a.IH, IW = a.IW, IP = IH * IW;
  const int oy0 = tile * a.TH;
  const int th = min(a.TH, a.H_out - oy0);
  const int MO = th * W_out;  // valid output pixels of this tile
  const int iy0 = oy0 * s - 1;
  float* xs = smem;                                   // [IP][XLD]
  float* es = EXPAND ? xs + (size_t)IP * XLD : xs;    // [IP][HLD]   (t == 1: the input is the dw operand)
  float* ds = es + (EXPAND ? (size_t)IP * HLD : (size_t)IP * XLD);  // [OP16][HLD]
  const float* W = a.wbase + (size_t)(a.k0 + k) * a.model_stride;
  const float* xin = a.x + (size_t)(k * a.B + b) * H_in * H_in * CIN;

  // ---- phase 0: input tile (+halo, zero outside the image, zero-padded channels) -> xs ----
  {
    const int C4 = (EXPAND ? CIN16 : HC) / 4;
    for (int e = tid; e < IP * C4; e += NWAVES * 64) {
      const int c4 = e % C4, ip = e / C4;
      const int iy = iy0 + ip / IW, ix = ip % IW - 1;
      // unconditional load from a clamped address + component-wise select (a load behind a branch is waited for at the
      // merge: one memory round trip per trip of this loop; a ternary on the float4 struct goes through scratch)
      const bool ok = iy >= 0 && iy < H_in && ix >= 0 && ix < H_in && c4 * 4 < CIN;
      const int iyc = min(max(iy, 0), H_in - 1), ixc = min(max(ix, 0), H_in - 1), cc = min(c4 * 4, CIN - 4);
      const float4 ld = *reinterpret_cast<const float4*>(xin + ((size_t)iyc * H_in + ixc) * CIN + cc);
      const float4 v = make_float4(ok ? ld.x : 0.f, ok ? ld.y : 0.f, ok ? ld.z : 0.f, ok ? ld.w : 0.f);
      *reinterpret_cast<float4*>(xs + (size_t)ip * XLD + c4 * 4) = v;
    }
  }
  __syncthreads();

  // projection accumulators: tiles (co_t, o_t) dealt round-robin to the waves
  const int n_ot = a.OP16 / 16, n_ct = (COUT + 15) / 16, TT = n_ot * n_ct;
  f32x4 acc[MAXT];
#pragma unroll
  for (int i = 0; i < MAXT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int n_pt = (IP + 15) / 16;
  for (int hc0 = 0; hc0 < HID; hc0 += HC) {
    if (EXPAND) {
      // ---- expand chunk: es[p][h] = mask(p) * relu6(be[h] + sum_c We[h][c] x[p][c]) ----
#pragma unroll 1
      for (int ht = 0; ht < HC / 16; ++ht) {
        const int h_row = hc0 + 16 * ht + n;  // A operand row (lane n = hidden channel within the tile)
        float4 areg[NS];
#pragma unroll
        for (int S = 0; S < NS; ++S) {
          const int c = 16 * S + 4 * q;
          const bool ok = c < CIN && h_row < HID;
          const float4 ld = *reinterpret_cast<const float4*>(W + a.we_off + (size_t)min(h_row, HID - 1) * CIN + min(c, CIN - 4));
          areg[S] = make_float4(ok ? ld.x : 0.f, ok ? ld.y : 0.f, ok ? ld.z : 0.f, ok ? ld.w : 0.f);
        }
        const int hb = hc0 + 16 * ht + 4 * q;  // this lane's 4 output channels
        const float4 bld = *reinterpret_cast<const float4*>(W + a.be_off + min(hb, HID - 4));
        const float4 bias = make_float4(hb < HID ? bld.x : 0.f, hb < HID ? bld.y : 0.f, hb < HID ? bld.z : 0.f, hb < HID ? bld.w : 0.f);
        for (int pt = wave; pt < n_pt; pt += NWAVES) {
          const int p = 16 * pt + n;
          const float* xr = xs + (size_t)min(p, IP - 1) * XLD + 4 * q;
          f32x4 c4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int S = 0; S < NS; ++S) {
            const float4 bv = *reinterpret_cast<const float4*>(xr + 16 * S);
            c4 = mfma4(areg[S].x, bv.x, c4);
            c4 = mfma4(areg[S].y, bv.y, c4);
            c4 = mfma4(areg[S].z, bv.z, c4);
            c4 = mfma4(areg[S].w, bv.w, c4);
          }
          if (p < IP) {
            const int iy = iy0 + p / IW, ix = p % IW - 1;
            const bool inside = iy >= 0 && iy < H_in && ix >= 0 && ix < H_in;  // the dw pads ITS input with zeros
            float4 o;
            o.x = inside ? relu6f(c4[0] + bias.x) : 0.f;
            o.y = inside ? relu6f(c4[1] + bias.y) : 0.f;
            o.z = inside ? relu6f(c4[2] + bias.z) : 0.f;
            o.w = inside ? relu6f(c4[3] + bias.w) : 0.f;
            *reinterpret_cast<float4*>(es + (size_t)p * HLD + 16 * ht + 4 * q) = o;
          }
        }
      }
      __syncthreads();
    }
    // ---- depthwise 3x3 on the chunk: ds[o][h] = relu6(bd[h] + sum_tap wd[tap][h] es[in(o, tap)][h]) ----
    {
      const int H4 = HC / 4;
      const int ELD = EXPAND ? HLD : XLD;
      for (int e = tid; e < MO * H4; e += NWAVES * 64) {
        const int h4 = e % H4, o = e / H4;
        const int oy = o / W_out, ox = o % W_out;
        const int hch = hc0 + 4 * h4;
        float4 accv = *reinterpret_cast<const float4*>(W + a.bd_off + hch);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const int ip = (oy * s + ky) * IW + ox * s + kx;
            const float4 v = *reinterpret_cast<const float4*>(es + (size_t)ip * ELD + 4 * h4);
            const float4 wv = *reinterpret_cast<const float4*>(W + a.wd_off + (size_t)(ky * 3 + kx) * HID + hch);
            accv.x = fmaf(v.x, wv.x, accv.x);
            accv.y = fmaf(v.y, wv.y, accv.y);
            accv.z = fmaf(v.z, wv.z, accv.z);
            accv.w = fmaf(v.w, wv.w, accv.w);
          }
        }
        accv.x = relu6f(accv.x);
        accv.y = relu6f(accv.y);
        accv.z = relu6f(accv.z);
        accv.w = relu6f(accv.w);
        *reinterpret_cast<float4*>(ds + (size_t)o * HLD + 4 * h4) = accv;
      }
    }
    __syncthreads();
    // ---- project chunk: acc[co][o] += sum_{h in chunk} Wp[co][h] ds[o][h] ----
#pragma unroll
    for (int i = 0; i < MAXT; ++i) {
      const int t = wave + NWAVES * i;
      if (t < TT) {
        const int ct = t / n_ot, ot = t - ct * n_ot;
        const int co = 16 * ct + n;
        const float* wr = W + a.wp_off + (size_t)min(co, COUT - 1) * HID + hc0 + 4 * q;
        const float* dr = ds + (size_t)(16 * ot + n) * HLD + 4 * q;
        f32x4 c4 = acc[i];
#pragma unroll
        for (int S = 0; S < HC / 16; ++S) {
          const float4 avl = *reinterpret_cast<const float4*>(wr + 16 * S);
          const bool con = co < COUT;
          const float4 av = make_float4(con ? avl.x : 0.f, con ? avl.y : 0.f, con ? avl.z : 0.f, con ? avl.w : 0.f);
          const float4 bv = *reinterpret_cast<const float4*>(dr + 16 * S);
          c4 = mfma4(av.x, bv.x, c4);
          c4 = mfma4(av.y, bv.y, c4);
          c4 = mfma4(av.z, bv.z, c4);
          c4 = mfma4(av.w, bv.w, c4);
        }
        acc[i] = c4;
      }
    }
    // the barrier after the next chunk's expand (or the dw barrier when !EXPAND) orders ds reuse
    if (!EXPAND) __syncthreads();
  }

  // ---- epilogue: + bias (+ residual) -> y (NHWC float4) ----
  float* yout = a.y + (size_t)(k * a.B + b) * a.H_out * W_out * COUT;
  // bias / residual operands of every tile first, from clamped addresses and with no per-lane branch around them
  float4 ebp[MAXT], eres[MAXT];
#pragma unroll
  for (int i = 0; i < MAXT; ++i) {
    const int t = min(wave + NWAVES * i, TT - 1);
    const int ct = t / n_ot, ot = t - ct * n_ot;
    const int o = min(16 * ot + n, MO - 1), co = min(16 * ct + 4 * q, COUT - 4);
    ebp[i] = *reinterpret_cast<const float4*>(W + a.bp_off + co);
    const size_t pix = (size_t)(oy0 + o / W_out) * W_out + o % W_out;
    if (a.residual) eres[i] = *reinterpret_cast<const float4*>(xin + pix * CIN + co);  // uniform branch
  }
#pragma unroll
  for (int i = 0; i < MAXT; ++i) {
    const int t = wave + NWAVES * i;
    if (t < TT) {
      const int ct = t / n_ot, ot = t - ct * n_ot;
      const int o = 16 * ot + n, co = 16 * ct + 4 * q;
      if (o < MO && co < COUT) {
        const float4 bp = ebp[i];
        const size_t pix = (size_t)(oy0 + o / W_out) * W_out + o % W_out;
        float4 v = make_float4(acc[i][0] + bp.x, acc[i][1] + bp.y, acc[i][2] + bp.z, acc[i][3] + bp.w);
        if (a.residual) {  // stride 1, CIN == COUT: same pixel of the block input
          const float4 r = eres[i];
          v.x += r.x;
          v.y += r.y;
          v.z += r.z;
          v.w += r.w;
        }
        *reinterpret_cast<float4*>(yout + pix * COUT + co) = v;
      }
    }
  }
}

struct Cfg {
  int HC, TH;
};

// tile / chunk choice per block shape (LDS budget 160 KiB)
Cfg choose_cfg(int cin, int hid, int h_in, int h_out, int stride, bool expand) {
  if (!expand) return {32, 5};
  if (h_in == 50) return {32, 5};
  if (h_in == 25) return {16, stride == 1 ? 13 : 7};
  if (h_in == 13) return {32, h_out};
  return {64, h_out};
}

template <int NS, int HC, bool EXPAND>
hipError_t launch_one(const IrbArgs& a, int tiles, int kc, size_t lds, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(irb_kernel<NS, HC, EXPAND>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL((irb_kernel<NS, HC, EXPAND>), dim3(tiles, a.B, kc), dim3(NWAVES * 64), lds, s, a);
  return hipGetLastError();
}

}  // namespace

// One fused inverted-residual block.  le: expand layer (nullptr for the t == 1 block), ld: depthwise, lp: project.
hipError_t launch_fused_block(const Layer* le, const Layer& ld, const Layer& lp, const float* enc_w,
                              size_t model_stride, int k0, int kc, int B, const float* x, float* y, hipStream_t s) {
  const bool expand = le != nullptr;
  IrbArgs a;
  a.x = x;
  a.y = y;
  a.wbase = enc_w;
  a.model_stride = model_stride;
  a.k0 = k0;
  a.we_off = expand ? le->w_off : 0;
  a.be_off = expand ? le->b_off : 0;
  a.wd_off = ld.w_off;
  a.bd_off = ld.b_off;
  a.wp_off = lp.w_off;
  a.bp_off = lp.b_off;
  a.B = B;
  a.CIN = expand ? le->cin : ld.cin;
  a.HID = ld.cout;
  a.COUT = lp.cout;
  a.H_in = ld.h_in;
  a.H_out = ld.h_out;
  a.stride = ld.stride;
  a.residual = lp.residual;
  const Cfg cfg = choose_cfg(a.CIN, a.HID, a.H_in, a.H_out, a.stride, expand);
  a.TH = cfg.TH;
  a.IH = (cfg.TH - 1) * a.stride + 3;
  a.IW = a.H_in + 2;
  a.OP16 = (cfg.TH * a.H_out + 15) / 16 * 16;
  const int tiles = (a.H_out + cfg.TH - 1) / cfg.TH;
  const int ns = (a.CIN + 15) / 16;
  const int IP = a.IH * a.IW;
  const int xld = (expand ? ns * 16 : cfg.HC) + 4, hld = cfg.HC + 4;
  const size_t lds = ((size_t)IP * xld + (expand ? (size_t)IP * hld : 0) + (size_t)a.OP16 * hld) * sizeof(float);
  const int n_tiles_proj = (a.OP16 / 16) * ((a.COUT + 15) / 16);
  if (lds > 160 * 1024 || n_tiles_proj > MAXT * NWAVES) return hipErrorInvalidConfiguration;
  if (!expand) return launch_one<2, 32, false>(a, tiles, kc, lds, s);
  if (ns == 1 && cfg.HC == 32) return launch_one<1, 32, true>(a, tiles, kc, lds, s);
  if (ns == 2 && cfg.HC == 16) return launch_one<2, 16, true>(a, tiles, kc, lds, s);
  if (ns == 2 && cfg.HC == 32) return launch_one<2, 32, true>(a, tiles, kc, lds, s);
  if (ns == 4 && cfg.HC == 64) return launch_one<4, 64, true>(a, tiles, kc, lds, s);
  if (ns == 6 && cfg.HC == 64) return launch_one<6, 64, true>(a, tiles, kc, lds, s);
  if (ns == 10 && cfg.HC == 64) return launch_one<10, 64, true>(a, tiles, kc, lds, s);
  return hipErrorInvalidConfiguration;
}

}  // namespace rip
