// Internal C++ interface between the C ABI (rip_abi.hip) and the flow kernels (flow.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rip {

// ---- per-model flow weights, device blob (floats), lane-major for coalesced register loads ----
// Reference tensors: _decoder._decoder.{weight_ih,weight_hh,bias_ih,bias_hh},
// _decoder._locscale._model.{0,2}.{weight,bias}  (torch/networks/sequence.py:53-65)
constexpr int FW_WHH = 0;       // [48 chunks][64 lanes][4]: lane j, chunk c=(g*16+i4) -> W_hh[g*64+j][4*i4..4*i4+3]
constexpr int FW_WIH = 12288;   // [(g*2+d)][64 lanes]      -> W_ih[g*64+j][d]
constexpr int FW_BIH = 12672;   // [g][64]
constexpr int FW_BHH = 12864;   // [g][64]
constexpr int FW_B1 = 13056;    // [64]: b1[j & 31]
constexpr int FW_W2 = 13120;    // [q][64]: W2[2*(j>>5)+q][j & 31]
constexpr int FW_B2 = 13248;    // [4]
constexpr int FW_W1 = 13252;    // [32][64] row-major (staged to LDS)
constexpr int FW_SIZE = 15300;
// fp32-MFMA search kernel operands (flow_phase.hip): 252 forward values/lane as 63 lane-major float4, then 69
// lane-major float4 of transposed (adjoint) operands.
constexpr int MWF_FLOATS = 63 * 64 * 4;
constexpr int MWB_F4 = 69;
constexpr int MW_SIZE = MWF_FLOATS + MWB_F4 * 64 * 4;
// split-f16 search kernel operands (flow_split.hip, packed by flow_split_pack.h), in dwords; a row = 64 lanes x 16 B.
// Forward rows: 0..47 W_hh ((gate g, unit tile up, K block kb) x (hi, lo')), 48..51 the fp32 input / bias k-steps,
// 52..59 W1 ((tile mt, kb) x (hi, lo')), 60..62 fp32 (b1, W2, b2) — rows 0..62 are what flow_pair.hip stages (round 5's
// forward step).  Round 6, flow_split.hip's forward step without fp32 MFMAs: 63..74 the input / bias k-steps as ONE f16 K
// block per (gate r / z / gi_n, unit tile) (three-term weights x three-term y / 4), 75..78 b_hn as accumulator images
// (lane (c, q): units 16 up + 4 q + r), 79, 80 b1 likewise, 81, 82 W2 x 4 (hi, lo') as one K block over the 32 head
// units, 83 b2.  Transposed rows: 0 = W2^T (fp32), 1..8 W1^T (out tile ut x (hi, lo')), 9..56 W_hh^T ((kb 0..5, ut) x
// (hi, lo')); then the 96-entry W_ih^T table.
constexpr int MHF_BASE_ROWS = 63, MHF_WHH = 0, MHF_WX = 48, MHF_W1 = 52, MHF_TAIL = 60;
constexpr int MHF_KS = 63, MHF_GHB = 75, MHF_B1 = 79, MHF_W2 = 81, MHF_B2 = 83, MHF_ROWS = 84;
constexpr int MHT_ROWS = 57, MHT_W2T = 0, MHT_W1T = 1, MHT_WHHT = 9;
constexpr int MH_TABLE_F4 = 96;
constexpr int MH_SIZE = (MHF_ROWS + MHT_ROWS) * 256 + MH_TABLE_F4 * 4;
constexpr int MAX_MODELS = 8;
constexpr int MAX_GOALS = 64;
enum { ALGO_WCM = 0, ALGO_MA = 1, ALGO_BCM = 2 };

struct SearchArgs {
  const float* flow_w;   // [K_total][FW_SIZE]
  int k0;                // first model
  int K;                 // models in this search
  const float* z;        // [K][B][64]
  const float* goal;     // [B][G][2] or nullptr
  const float* x0;       // [B][N][8]
  int B, N, G;
  int algorithm;
  int num_steps;
  float lr, epsilon;
  float grad_scale;      // 1 for RIP; 1/B for ImitativeModel.forward's batch-mean loss
  float* plans;          // [B][N][8] or nullptr
  float* loss_best;      // [B][N] or nullptr
  float* trace_post;     // [steps][K][B][N] or nullptr
  float* trace_x;        // [steps][B][N][8] or nullptr
  float* trace_loss;     // [steps][B][N] or nullptr
  float* trace_grad;     // [steps][B][N][8] dLoss/dx of every Adam step, or nullptr
  unsigned long long* stats = nullptr;  // device counter: += executed inverse-pass adjoints (phase-sequential kernels), or nullptr
  // Operand-range guard of the split-f16 kernel (hidden states are split into binary16 terms unscaled: |h| <= max(1, |z|)
  // must stay below the binary16 range).  `range_flag` is a device word the split launch zeroes and its prefix kernel
  // raises when some |z| >= SPLIT_Z_LIMIT; the split kernel then does nothing and the fp32-MFMA kernel, launched behind
  // it with `run_if_flag` = the same word, does the whole search instead (it exits at once when the word is 0).
  unsigned* range_flag = nullptr;
  const unsigned* run_if_flag = nullptr;
  // split-f16 kernel: workgroup shape.  0 = by the launcher's cost model; 2 / 4 / 8 = that many one-wave blocks per
  // workgroup (flow_split.hip); SPLIT_SHAPE_PAIR = four blocks per workgroup, each on a PAIR of waves (flow_pair.hip)
  int split_shape = 0;
};
constexpr int SPLIT_SHAPE_PAIR = 16;
constexpr float SPLIT_Z_LIMIT = 16384.0f;  // binary16 overflows at 65504; hidden states reach max(1, |z|)
constexpr float SPLIT_W_LIMIT = 200.0f;    // ... and the transposed operand rows hold w * 2^8 (flow_split_pack.h)

// Gradient-mode model-parallel search: one Adam step split at the exchange point.
struct MpArgs {
  const float* flow_w;   // [K_handle][FW_SIZE]
  int k_fwd;             // handle index of the flow that maps x -> y (global model 0)
  int k_begin, k_count;  // handle indices of this rank's models
  int first_is_fwd;      // local model 0 IS the forward model (rank 0): its posterior comes from the shortcut
  const float* z_fwd;    // [B][64] z of the forward model
  const float* z;        // [k_count][B][64]
  const float* goal;     // [B][G][2] or nullptr
  int B, N, G, K;        // K = models in the whole ensemble (rip_mp_update)
  int algorithm;
  float lr, epsilon;
  int step;              // 0-based Adam step index (bias correction)
};
hipError_t launch_mp_local(const MpArgs& a, const float* x, float* out /*[k_count][B][N][9]*/, hipStream_t s);
hipError_t launch_mp_update(const MpArgs& a, const float* gathered /*[K][B][N][9]*/, float* x, float* m, float* v,
                            float* x_best, float* loss_best, float* grad_out, hipStream_t s);

hipError_t launch_flow_forward(const float* flow_w_k, const float* x, const float* z, int N, int z_rows, float* y,
                               float* lad, hipStream_t s);
hipError_t launch_flow_inverse(const float* flow_w_k, const float* y, const float* z, int N, int z_rows, float* x,
                               float* logp, float* lad, hipStream_t s);
hipError_t launch_goal_rows(const float* y, const float* goal, int N, int goal_rows, int G, float eps, float* rows,
                            hipStream_t s);
hipError_t launch_score(const float* flow_w, int k0, int K, const float* z, const float* y, const float* goal, int B,
                        int N, int G, float eps, float* S, hipStream_t s);
hipError_t launch_search(const SearchArgs& a, hipStream_t s);
hipError_t launch_select_best(const float* plans, const float* loss_best, int B, int N, float* plan, int32_t* best,
                              double* interp /*[B][30][3] or nullptr*/, hipStream_t s);
hipError_t launch_interpolate_plans(const float* plan /*[B][4][2]*/, int B, double* out /*[B][30][3]*/, hipStream_t s);
hipError_t launch_dim_select(const float* flow_w_k, const float* z, const float* x0, const float* trace_loss,
                             const float* trace_x, int B, int num_steps, float* y, float* trace_mean, hipStream_t s);
hipError_t launch_cil_decode(const float* feat, const float* vec, const float* w, int B, int T, float* y, hipStream_t s);  // cil.hip
int cil_blob_floats();
hipError_t launch_lidar_bev(const float* points, const int* offsets, int B, float* bev, hipStream_t s);  // lidar.hip
hipError_t launch_aggregate_scores(const float* S, int K, int B, int N, int algorithm, float* loss, int32_t* best,
                                   hipStream_t s);
size_t search_lds_bytes(int K);
// raises a kernel's dynamic-LDS limit to the CU's 160 KiB, once per (kernel, device); thread-safe, any device index
hipError_t allow_lds(const void* fn);
int device_cu_count();  // compute units of the current device (cached per device); the persistent launchers assume
                        // `occupancy x device_cu_count()` workgroups are resident at once (true for a whole device in SPX mode)
int device_xcd_count(); // 8 on a whole MI355X (256 CUs), 0 = unknown: XCD-aware work lists off
// phase-sequential variant (flow_phase.hip): one wave per 16-candidate block runs all K models, operands in LDS;
// N % 16 == 0, any K <= MAX_MODELS, traces supported
// split-f16 variant (flow_split.hip): the same decomposition with the contractions on v_mfma_f32_16x16x32_f16 and both
// operands carried as two binary16 terms; operands = the MH blob (flow_split_pack.h)
bool search_split_supported(const SearchArgs& a);
size_t search_split_scratch_bytes(int B, int N, int K);
hipError_t launch_search_split(const SearchArgs& a, const uint32_t* mh_all, void* scratch, hipStream_t s);
// flow_pair.hip: the paired shape of the same search (launched by launch_search_split on its prefix table and tape slots)
hipError_t launch_search_pair(const SearchArgs& a, const uint32_t* mh_all, const float* pre, float4* tape, int items, hipStream_t s);
// what a launch of the phase-sequential kernels executes on the matrix cores (rip_search_plan): out[0] = waves per
// workgroup, then (f16, fp32) MFMA instructions per 16-candidate block of a forward / inverse pass, the adjoint of an
// inverse pass, the adjoint of F_0, and of the prefix step per (model, observation)
void search_split_info(int B, int N, int K, int out[9], int shape = 0);
void search_phase_info(int B, int N, int K, int out[9]);
bool search_phase_supported(const SearchArgs& a);
size_t search_phase_scratch_bytes(int B, int N, int K);
hipError_t launch_search_phase(const SearchArgs& a, const float* mw_all, void* scratch, hipStream_t s);

}  // namespace rip
