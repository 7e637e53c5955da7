// Internal C++ interface between the C ABI (rip_abi.hip) and the encoder kernels (encoder.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

namespace rip {

// Kernel-selection log (rip_kernel_log, option RIP_OPT_KERNEL_LOG): while a log is installed for the calling thread,
// every encoder launch site appends one line "kernel<template arguments> grid=(x,y,z) block=n".  The tests use it to
// assert that a parity case really ran the kernels a given launch shape selects (the selection keys on B * k_count).
struct KernelLog {
  std::string text;
};
void kernel_log_install(KernelLog* log);  // nullptr = off (default); thread-local
bool kernel_log_active();
void note_kernel(dim3 grid, dim3 block, const char* fmt, ...) __attribute__((format(printf, 3, 4)));

// Development / test selections of the bf16 encoder (option RIP_OPT_ENCODER_VARIANT, a bit mask; 0 = what ships):
enum {
  ENC_VAR_IRB_ROUND3 = 1,    // RETIRED in round 6 (accepted, no effect): features.2-7 on round 1's row-streaming kernel (depthwise on the
  ENC_VAR_ROWS_F5_7 = 4,     // vector unit) / features.5-7 on it.  With bf16-valued taps the matrix-core depthwise kernel is faster on
                             // all six blocks (encoder_bf16_irb2.hip); encoder_bf16_irb.hip is gone
  ENC_VAR_FRONT_ROUND3 = 2,  // stem + features.1 on round 3's front kernel
  ENC_VAR_F17_LAYERWISE = 8, // features.17 as three layer-wise launches (round 4: persistent GEMMs + row-streaming depthwise) instead of a tile block
  ENC_VAR_FP32_LAYERWISE = 16,  // fp32 encoder: no split-f16 tile blocks (encoder_split_tile.hip), every layer outside the
                                // fused leading blocks as its own true-fp32 launch (rounds 1-5)
};

// The fp32 encoder's split-f16 blocks read the pointwise weights as two binary16 terms of w * 2^8 (rip_abi.hip: enc_wc, enc_wr);
// a model whose pointwise weights reach this magnitude keeps the layer-wise fp32 kernels (binary16 max 65504 / 2^8).
constexpr float SPLIT_ENC_W_SCALE = 256.0f;
constexpr float SPLIT_ENC_W_LIMIT = 240.0f;
// (model, observation) pairs from which the split-f16 TILE blocks and head run: they walk an observation serially (one
// workgroup each up to 256 pairs: ~0.3 ms for features.8-18 whatever the launch size).  Below it features.8-17 are two
// launches per block — expansion + depthwise of one (observation, 64-channel chunk) per workgroup (`irb_split_expdw_kernel`),
// then the layer-wise projection — and features.18 the layer-wise GEMM.  The ROW-STREAMING blocks and the front cut an
// observation into row bands when the launch is small and run at every size.  profiles/r6/fp32_crossover_v1.txt: one
// observation x 4 models 245 (layer-wise) -> 193 us, 8: 457 -> 256, 32: 850 -> 474 (every split kernel: 547), 40: 581 (596), 48: 690
// against 617.
constexpr int SPLIT_TILE_MIN_PAIRS = 176;
constexpr int SPLIT_ROWS_MIN_PAIRS = 1;

// Workgroup barrier for kernels whose waves talk to each other through LDS only.  `__syncthreads()` is a workgroup-scope
// release / acquire fence over ALL address spaces: the compiler puts `s_waitcnt vmcnt(0)` in front of the s_barrier, so
// every global load that was meant to stay in flight across the barrier — operand prefetches of the next row / K-step /
// item — is waited for right there, one memory latency per barrier (round 4: this, not occupancy, is what the row-
// streaming and GEMM kernels were bound by).  The fences below are restricted to the LDS address space ("local"):
// `s_waitcnt lgkmcnt(0)` + `s_barrier`, vector-memory operations keep their own counters.
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

enum LayerKind { L_STEM = 0, L_DW = 1, L_PW = 2 };

// One conv layer of the BN-folded MobileNetV2 (torchvision v0.6.0 layout; reference call site
// oatomobile/torch/networks/perception.py:36-51).  Offsets are in floats into the per-model
// encoder blob; activations are NHWC fp32, [K][B][H][W][C].
struct Layer {
  int kind;
  int cin, cout;
  int h_in, h_out, stride;
  int relu6;      // ReLU6 after the folded BN
  int residual;   // add the block input (projection layers of stride-1, equal-width blocks)
  size_t w_off, b_off;
  int src, dst, res;  // workspace buffer ids (0..3); src == -1 -> the [B,C,100,100] network input
};

// one torchvision InvertedResidual block = [expand] + depthwise + project (indices into EncoderPlan::layers)
struct FusedBlock {
  int expand, dw, project;  // expand == -1 for the t == 1 block
  int src, dst;             // workspace buffers (block input / output)
};

// operand blobs of the fp32 encoder's split-f16 kernels (offsets in binary16 elements into a model's blob; (size_t)-1 = the
// block is not one of that family's): encoder_split_rows.hip / encoder_split_tile.hip
struct SplitRowsLayout {
  std::vector<size_t> off;  // per plan block: its operand fragments (row-streaming blocks; block 0 = the front's projection)
  size_t total = 0;         // binary16 elements per model
};
struct SplitTileLayout {
  std::vector<size_t> off;       // per plan block: its chunk records (tile blocks)
  size_t head_off = (size_t)-1;  // features.18's chunk records (head_split_kernel)
  size_t total = 0;
};

struct EncoderPlan {
  SplitRowsLayout split_rows;  // (filled by build_encoder_plan)
  SplitTileLayout split_tiles;
  int in_channels;
  std::vector<Layer> layers;
  std::vector<FusedBlock> blocks;
  size_t cls_w_off, cls_b_off;     // classifier.1  [128][1280], [128]
  size_t mrg_w_off[3], mrg_b_off[3];  // merger Linear 133->64, 64->64, 64->64
  size_t blob_floats;              // per-model folded blob size
  size_t max_act_floats;           // largest activation per image
  int final_buf;                   // buffer holding features.18 output [16][1280]
  int final_hw;                    // 4
};

EncoderPlan build_encoder_plan(int in_channels);

// Diagnostics tap (rip_encode_tap): stop the encoder after conv layer `layer` (network order) and hand its output out as
// fp32 — `dst` [kc][B][H][W][C] (NHWC), or [kc][B][1280] for the last layer, whose 4x4 average pool is fused into its
// epilogue.  `served` stays false when the layer's output never reaches memory under the kernel selection in force
// (an interior layer of a fused block).
struct EncoderTap {
  int layer = -1;
  float* dst = nullptr;
  bool served = false;
};
hipError_t launch_tap_copy(const void* src, bool src_bf16, size_t n, float* dst, hipStream_t s);

// Folds BN and re-lays the packed reference tensors (arch.py:packed_spec order) into the encoder blob
// and the flow blob (flow.h layout).  Returns false (and a message) on size mismatch.
bool fold_and_pack(const EncoderPlan& plan, const float* packed, size_t numel, std::vector<float>& enc_blob,
                   std::vector<float>& flow_blob, std::vector<float>& mfma_blob, std::vector<uint32_t>& split_blob,
                   const char** err, float* split_wmax = nullptr);

bool transform_coded_supported(int C, int H, int W, int out_hw);
hipError_t launch_transform_coded(const uint8_t* in, const float* lut /*[256]*/, int B, int C, int H, int W, int out_hw,
                                  float* out, hipStream_t s);
hipError_t launch_transform(const float* in, int B, int C, int H, int W, int channels_last, int out_hw, float* out,
                            hipStream_t s);

// Runs the encoder + merger for models [k0, k0+kc) on B observations.
//   enc_w: [K_total][plan.blob_floats]; visual [B,C,100,100]; vec [B,5]; bufs[4]: each >= kc*B*max_act floats.
//   fused_blocks: the first `fused_blocks` inverted-residual blocks run as one kernel each (encoder_fused.hip),
//   the rest layer by layer.
//   enc_wc: chunk records of the split-f16 tile blocks (features.8-17, encoder_split_tile.hip; `pack_split_tiles`, models
//   wc_stride binary16 elements apart), or nullptr: with them the launch runs the split-f16 blocks when it has
//   >= SPLIT_TILE_MIN_PAIRS (model, observation) pairs.
//   enc_wr: operand fragments of the split-f16 row-streaming blocks (features.2-7, encoder_split_rows.hip;
//   `pack_split_rows`, models wr_stride halves apart), or nullptr.
hipError_t launch_encoder(const EncoderPlan& plan, const float* enc_w, int k0, int kc, const float* visual,
                          const float* vec, int B, float* const bufs[4], float* z, float* feat, int fused_blocks,
                          hipStream_t s, EncoderTap* tap = nullptr, const unsigned short* enc_wc = nullptr, size_t wc_stride = 0,
                          const unsigned short* enc_wr = nullptr, size_t wr_stride = 0);

// The whole fp32 encoder of a small batch as ONE persistent launch, model k on XCD k % 8 (encoder.hip:
// encoder_mega_kernel).  arena: kc * arena_model_stride floats, arena_model_stride >= encoder_mega_arena_floats(B);
// sync: 8 * 64 zeroed unsigned (re-armed by the kernel itself); status: pinned host word, set non-zero when the
// placement / barrier protocol failed (results invalid: fall back to launch_encoder); ticks: nullable.
size_t encoder_mega_arena_floats(const EncoderPlan& plan, int B);
bool encoder_mega_supported(const EncoderPlan& plan, int B, int kc);
bool encoder_mega_probe(int device);
hipError_t launch_encoder_mega(const EncoderPlan& plan, const float* enc_w, int k0, int kc, const float* visual,
                               const float* vec, int B, float* arena, size_t arena_model_stride, unsigned* sync,
                               int* status, unsigned long long* ticks, float* z, float* feat, int wgs_per_xcd,
                               hipStream_t s);

hipError_t launch_tail(const EncoderPlan& plan, const float* enc_w, int k0, int kc, const float* act_last, int hw,
                       const float* vec, int B, float* scratch, float* z, float* feat, hipStream_t s);

// bf16 encoder (encoder_bf16.hip): bf16 NHWC activations, bf16 pointwise weights (enc_wh: same offsets as the fp32
// blob, 2 bytes per element), fp32 accumulation / bias / ReLU6 / residual math; features.18 is written in fp32.
//   fused_blocks: the first `fused_blocks` inverted-residual blocks (at most the large-image stages the fused kernel
//   supports) run as one row-streaming kernel each (encoder_bf16_irb2.hip); -1 = choose by batch.
hipError_t launch_encoder_bf16(const EncoderPlan& plan, const float* enc_w, const unsigned short* enc_wh, int k0, int kc,
                               const float* visual, const float* vec, int B, float* const bufs[4], float* z,
                               float* feat, int fused_blocks, hipStream_t s, EncoderTap* tap = nullptr, int variant = 0);

// features.2-7: expand -> depthwise -> project as one row-streaming kernel, all three convolutions on the matrix cores
// (encoder_bf16_irb2.hip, round 4; round 1's vector-unit version, encoder_bf16_irb.hip, was retired in round 6)
bool irb2_bf16_supported(const Layer* le, const Layer& ld, const Layer& lp, bool everywhere = false);
hipError_t launch_irb2_bf16(const Layer* le, const Layer& ld, const Layer& lp, const float* enc_w,
                            const unsigned short* enc_wh, size_t model_stride, int k0, int kc, int B,
                            const unsigned short* x, unsigned short* y, hipStream_t s);

// stem + features.1 in one kernel (encoder_bf16_front.hip)
bool front_bf16_supported(const Layer& ls, const Layer& ld, const Layer& lp);
hipError_t launch_front_bf16(const Layer& ls, const Layer& ld, const Layer& lp, const float* enc_w,
                             const unsigned short* enc_wh, size_t model_stride, int k0, int kc, int B, const float* visual,
                             unsigned short* y, hipStream_t s);

// round 4: the front with the stem and the depthwise on the matrix cores as well (encoder_bf16_front2.hip; C = 2 only)
bool front2_bf16_supported(const Layer& ls, const Layer& ld, const Layer& lp);
hipError_t launch_front2_bf16(const Layer& ls, const Layer& ld, const Layer& lp, const float* enc_w,
                              const unsigned short* enc_wh, size_t model_stride, int k0, int kc, int B, const float* visual,
                              unsigned short* y, hipStream_t s);

// small-image stages (7x7 / 4x4 maps, features.8 .. features.17): encoder_bf16_tile.hip
bool irb_tile_bf16_supported(const Layer* le, const Layer& ld, const Layer& lp, bool f17_layerwise = false);
hipError_t launch_irb_tile_bf16(const Layer* le, const Layer& ld, const Layer& lp, const float* enc_w,
                                const unsigned short* enc_wh, size_t model_stride, int k0, int kc, int B,
                                const unsigned short* x, unsigned short* y, hipStream_t s);

// fp32-grade tile blocks of the fp32 encoder (features.8-17): fp32 activations, two-term binary16 pointwise operands
bool irb_split_tile_supported(const Layer* le, const Layer& ld, const Layer& lp);
// small launches of the tile-block layers: expansion + depthwise of one (observation, 64-channel chunk) per workgroup, the
// depthwise output to memory (the projection stays a layer-wise launch)
hipError_t launch_irb_split_expdw(const Layer* le, const Layer& ld, const Layer& lp, const unsigned short* wc, size_t wc_stride,
                                  int k0, int kc, int B, const float* x, float* d, hipStream_t s);
// features.18 + the 4x4 average pool on two-term binary16 operands (encoder_split_tile.hip: head_split_kernel)
bool head_split_supported(const Layer& l, int final_hw);
hipError_t launch_head_split(const Layer& l, const unsigned short* wc, size_t wc_stride, int k0, int kc, int B, const float* x,
                             float* y, hipStream_t s);
SplitTileLayout split_tile_layout(const EncoderPlan& plan);
void pack_split_tiles(const EncoderPlan& plan, const SplitTileLayout& L, const float* enc_blob, unsigned short* out);
hipError_t launch_irb_split_tile(const Layer* le, const Layer& ld, const Layer& lp, const float* enc_w, const unsigned short* wc,
                                 size_t wc_stride, size_t model_stride, int k0, int kc, int B, const float* x, float* y, hipStream_t s);

// fp32-grade row-streaming blocks of the fp32 encoder (features.2-7): encoder_split_rows.hip
SplitRowsLayout split_rows_layout(const EncoderPlan& plan);
void pack_split_rows(const EncoderPlan& plan, const SplitRowsLayout& L, const float* enc_blob, unsigned short* out);
bool irb_split_rows_supported(const Layer* le, const Layer& ld, const Layer& lp);
// stem + features.1 in the same structure (fp32 stem on the vector unit, split-f16 projection; C = 2)
bool front_split_supported(const Layer& ls, const Layer& ld, const Layer& lp);
hipError_t launch_front_split(const Layer& ls, const Layer& ld, const Layer& lp, const float* enc_w, const unsigned short* wfrag,
                              size_t wr_stride, size_t model_stride, int k0, int kc, int B, const float* visual, float* y,
                              hipStream_t s);
hipError_t launch_irb_split_rows(const Layer* le, const Layer& ld, const Layer& lp, const float* enc_w, const unsigned short* wfrag,
                                 size_t wr_stride, size_t model_stride, int k0, int kc, int B, const float* x, float* y,
                                 hipStream_t s);

hipError_t launch_fused_block(const Layer* le, const Layer& ld, const Layer& lp, const float* enc_w,
                              size_t model_stride, int k0, int kc, int B, const float* x, float* y, hipStream_t s);

// Row bands per observation for the row-streaming blocks (encoder_bf16_irb2.hip): `slots` resident
// workgroups walk pairs * bands (model, observation, band) items; an item costs its rows plus ~2 rows of halo / prologue.
// The band count that minimises rounds * (rows + 2) — 768 pairs on 512 slots: one band is two rounds of 27, two bands are
// three rounds of 15 (round 5: features.2 at 192 observations x 4 models 81 us with one band per observation).  A
// handful of pairs: all rounds are one, the shortest bands win (the launch is a latency chain).
inline int pick_row_bands(long pairs, int h_out, int min_rows, int slots) {
  int best = 1;
  long best_cost = -1;
  const int max_b = h_out / min_rows > 1 ? h_out / min_rows : 1;
  for (int b = 1; b <= max_b; ++b) {
    const int rows = (h_out + b - 1) / b, nb = (h_out + rows - 1) / rows;
    const long rounds = (pairs * nb + slots - 1) / slots;
    const long cost = rounds * (rows + 2);
    if (best_cost < 0 || cost < best_cost) {
      best_cost = cost;
      best = nb;
    }
  }
  return best;
}

}  // namespace rip
