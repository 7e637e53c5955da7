// C ABI of librip_hip.so (include/rip_hip.h): handle management, checkpoint folding, argument
// validation and kernel launch sequencing.  No torch types cross this boundary.
#include "../../include/rip_hip.h"

#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <new>
#include <vector>

#include "encoder.h"
#include "flow.h"
#include "train.h"

using namespace rip;

static_assert(RIP_MAX_MODELS == rip::MAX_MODELS, "header / kernel constant mismatch");
static_assert(RIP_ALGO_WCM == rip::ALGO_WCM && RIP_ALGO_MA == rip::ALGO_MA && RIP_ALGO_BCM == rip::ALGO_BCM, "algo ids");

struct rip_handle {
  int K = 0, C = 0, max_batch = 0, max_candidates = 0, device = 0;
  hipStream_t last_stream = nullptr;  // the stream of the previous call: the scratch below is shared by all calls
  bool used = false;
  hipEvent_t order = nullptr;
  EncoderPlan plan;
  float* enc_w = nullptr;   // [K][plan.blob_floats]
  unsigned short* enc_wh = nullptr;  // [K][plan.blob_floats] bf16 copy of the folded blob (bf16 encoder)
  float* enc_wt = nullptr;  // [K][plan.blob_floats] the fp32 blob with the DEPTHWISE taps rounded to bf16 values (round 6): what the
                            // bf16 encoder reads its fp32 words from — "activations + weights bf16" (BASELINE configs[2]) for every
                            // convolution but the stem; biases and the stem's taps are the fp32 blob's
  unsigned short* enc_wc = nullptr;  // [K][tile_layout.total] chunk records of the fp32 encoder's split-f16 tile blocks (encoder_split_tile.hip:
                                     // two-term binary16 operand fragments of w * 2^8, taps, biases)
  unsigned short* enc_wr = nullptr;  // [K][rows_layout.total] operand fragments of the split-f16 row-streaming blocks (encoder_split_rows.hip)
  bool enc_split_ok[RIP_MAX_MODELS] = {false};  // the model's pointwise weights are inside SPLIT_ENC_W_LIMIT
  float* flow_w = nullptr;  // [K][FW_SIZE]
  float* mfma_w = nullptr;  // [K][MW_SIZE] operands of the fp32 MFMA search kernels
  uint32_t* split_w = nullptr;  // [K][MH_SIZE] operands of the split-f16 search kernel
  void* tape = nullptr;     // scratch of the MFMA search kernel
  size_t tape_bytes = 0;
  int search_mode = 0;      // 0 auto, 1 wave-per-chain (VALU), 3 fp32-MFMA phase-sequential,
                            // 4 split-f16 phase-sequential
  int encoder_fused = -1;   // leading inverted-residual blocks run fused (0 = none, 17 = all); -1 = auto by batch
  int encoder_variant = 0;  // RIP_OPT_ENCODER_VARIANT: development / test kernel selections of the bf16 encoder (encoder.h ENC_VAR_*)
  bool kernel_log_on = false;  // RIP_OPT_KERNEL_LOG
  KernelLog klog;              // the encoder kernels of the last rip_encode* / rip_encode_tap* call (rip_kernel_log)
  bool loaded[RIP_MAX_MODELS] = {false};
  float split_wmax[RIP_MAX_MODELS] = {0.f};  // largest flow weight magnitude per model (flow_split_pack.h: SPLIT_W_LIMIT)
  float* bufs[4] = {nullptr, nullptr, nullptr, nullptr};  // encoder activations
  size_t buf_floats = 0;
  // scratch for the fused entry points
  float* visual = nullptr;     // [max_batch][C][100][100]
  float* z = nullptr;          // [K][max_batch][64]
  float* plans = nullptr;      // [max_batch][max_candidates][8]
  float* loss_best = nullptr;  // [max_batch][max_candidates]
  float* trace_loss = nullptr; // [RIP_MAX_STEPS][max_batch]   (ImitativeModel.forward)
  float* trace_x = nullptr;    // [RIP_MAX_STEPS][max_batch][8]
  unsigned long long* stats = nullptr;  // [1] executed inverse-pass adjoints of the phase-sequential kernels (rip_search_stats)
  unsigned* range_flag = nullptr;        // operand-range word of the split-f16 search (flow.h SearchArgs)
  // one-launch fp32 encoder for small batches (encoder.hip: encoder_mega_kernel)
  int encoder_mega = -1;        // -1 auto (= never: the launches measured faster), 0 never, 1 whenever the batch fits mega_max_b
  int mega_max_b = 0;           // 0: not available on this device / handle
  int mega_wgs = 32;            // workgroups per XCD
  float* mega_arena = nullptr;  // [K][mega_stride]
  size_t mega_stride = 0;
  unsigned* mega_sync = nullptr;         // [8][64] barrier / exit counters, one pair per XCD
  int* mega_status = nullptr;            // pinned host word: non-zero = the protocol failed, results of that call invalid
  bool mega_reported = false;            // the failure has been handed to the caller (rip_encoder_status or an error)
  unsigned long long* mega_ticks = nullptr;  // development (RIP_MEGA_TICKS=1): per-layer wall clock of model 0
};

// Makes the handle's device current for one entry point and restores the caller's on exit.
struct DeviceScope {
  int prev = -1;
  bool switched = false;
  hipError_t err = hipSuccess;
  explicit DeviceScope(int device) {
    err = hipGetDevice(&prev);
    if (err == hipSuccess && prev != device) {
      err = hipSetDevice(device);
      switched = err == hipSuccess;
    }
  }
  ~DeviceScope() {
    if (switched) (void)hipSetDevice(prev);
  }
};

// A handle is single-stream (its scratch is shared): a call on another stream first waits for the previous one.
static hipError_t enter_stream(rip_handle* h, hipStream_t s) {
  if (h->used && h->last_stream != s) {
    hipError_t e = hipEventRecord(h->order, h->last_stream);
    if (e != hipSuccess) return e;
    e = hipStreamWaitEvent(s, h->order, 0);
    if (e != hipSuccess) return e;
  }
  h->last_stream = s;
  h->used = true;
  return hipSuccess;
}

#define ENTER(h_, stream_)                                                                  \
  DeviceScope scope_((h_)->device);                                                         \
  if (scope_.err != hipSuccess) return fail(RIP_EHIP, "hipSetDevice(%d) failed: %s", (h_)->device, hipGetErrorString(scope_.err)); \
  HIP_TRY(enter_stream((h_), (hipStream_t)(stream_)));                                      \
  if (int guard_ = mega_guard(h_)) return guard_


static thread_local char g_err[512] = "";

// RIP_OPT_KERNEL_LOG: the encoder launch sites of this call append to the handle's log (encoder.h: note_kernel)
struct KernelLogScope {
  bool on;
  explicit KernelLogScope(rip_handle* h) : on(h->kernel_log_on) {
    if (on) {
      h->klog.text.clear();
      kernel_log_install(&h->klog);
    }
  }
  ~KernelLogScope() {
    if (on) kernel_log_install(nullptr);
  }
};

// ---- tracing hook: rocTX ranges around encode / search / train / collectives, visible in
// `rocprofv3 --marker-trace`.  Off unless RIP_ROCTX=1 is in the environment: then librocprofiler-sdk-roctx (or the
// roctracer-era libroctx64) is dlopen'ed once; without the library the ranges are no-ops.
struct Roctx {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
};
static const Roctx& roctx() {
  static const Roctx r = [] {
    Roctx x;
    const char* e = getenv("RIP_ROCTX");
    if (e == nullptr || e[0] != '1') return x;
    for (const char* name : {"librocprofiler-sdk-roctx.so", "libroctx64.so"}) {
      if (void* h = dlopen(name, RTLD_LAZY | RTLD_GLOBAL)) {
        x.push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
        x.pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
        if (x.push != nullptr && x.pop != nullptr) return x;
        x = Roctx();
      }
    }
    return x;
  }();
  return r;
}
struct TraceRange {
  bool on;
  explicit TraceRange(const char* name) : on(roctx().push != nullptr) {
    if (on) roctx().push(name);
  }
  ~TraceRange() {
    if (on) roctx().pop();
  }
};

static int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

// A one-launch encoder call whose placement / barrier protocol failed left an invalid z behind: the first entry point
// that runs after the failure became visible reports it (once), unless the caller has already asked rip_encoder_status.
static int mega_guard(rip_handle* h) {
  if (h->mega_status == nullptr || *h->mega_status == 0 || h->mega_reported) return RIP_OK;
  h->mega_reported = true;
  return fail(RIP_ESTATE, "an earlier one-launch encoder call failed (status %d): its outputs are invalid; the handle "
              "uses the layer-wise launches from now on", *h->mega_status);
}

#define HIP_TRY(expr)                                                                       \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess) return fail(RIP_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

#define REQUIRE(cond, ...) \
  do {                     \
    if (!(cond)) return fail(RIP_EINVAL, __VA_ARGS__); \
  } while (0)

static int check_models(const rip_handle* h, int k0, int kc) {
  if (h == nullptr) return fail(RIP_EINVAL, "handle is NULL");
  if (k0 < 0 || kc < 1 || k0 + kc > h->K) return fail(RIP_EINVAL, "model range [%d,%d) outside [0,%d)", k0, k0 + kc, h->K);
  for (int k = k0; k < k0 + kc; ++k)
    if (!h->loaded[k]) return fail(RIP_ESTATE, "model %d has no weights (call rip_load_model first)", k);
  return RIP_OK;
}

extern "C" {

int rip_abi_version(void) { return 4; }
const char* rip_last_error(void) { return g_err; }

int rip_create(rip_handle** out, int K, int in_channels, int max_batch, int max_candidates, int device) {
  REQUIRE(out != nullptr, "out is NULL");
  REQUIRE(K >= 1 && K <= RIP_MAX_MODELS, "K=%d outside [1,%d]", K, RIP_MAX_MODELS);
  REQUIRE(in_channels >= 1 && in_channels <= 16, "in_channels=%d outside [1,16]", in_channels);
  REQUIRE(max_batch >= 1, "max_batch=%d must be >= 1", max_batch);
  REQUIRE(max_candidates >= 1, "max_candidates=%d must be >= 1", max_candidates);
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  REQUIRE(device >= 0 && device < ndev, "device %d not in [0,%d)", device, ndev);
  DeviceScope scope(device);
  if (scope.err != hipSuccess) return fail(RIP_EHIP, "hipSetDevice(%d) failed: %s", device, hipGetErrorString(scope.err));
  rip_handle* h = new (std::nothrow) rip_handle();
  if (h == nullptr) return fail(RIP_ESTATE, "out of host memory");
  h->K = K;
  h->C = in_channels;
  h->max_batch = max_batch;
  h->max_candidates = max_candidates;
  h->device = device;
  {
    hipError_t e_ = hipEventCreateWithFlags(&h->order, hipEventDisableTiming);
    if (e_ != hipSuccess) {
      delete h;
      return fail(RIP_EHIP, "hipEventCreate failed: %s", hipGetErrorString(e_));
    }
  }
  h->plan = build_encoder_plan(in_channels);
  h->buf_floats = (size_t)K * max_batch * h->plan.max_act_floats;
#define ALLOC(ptr, n)                                                             \
  do {                                                                            \
    hipError_t e_ = hipMalloc((void**)&(ptr), (n) * sizeof(float));               \
    if (e_ != hipSuccess) {                                                       \
      rip_destroy(h);                                                             \
      return fail(RIP_EHIP, "hipMalloc(%zu B) failed: %s", (size_t)(n) * 4, hipGetErrorString(e_)); \
    }                                                                             \
  } while (0)
  ALLOC(h->enc_w, (size_t)K * h->plan.blob_floats);
  ALLOC(h->enc_wt, (size_t)K * h->plan.blob_floats);
  {
    float* tmp = nullptr;
    ALLOC(tmp, ((size_t)K * h->plan.blob_floats + 1) / 2);
    h->enc_wh = reinterpret_cast<unsigned short*>(tmp);
  }
  {
    float* tmp = nullptr;
    ALLOC(tmp, ((size_t)K * h->plan.split_tiles.total + 1) / 2);
    h->enc_wc = reinterpret_cast<unsigned short*>(tmp);
  }
  {
    float* tmp = nullptr;
    ALLOC(tmp, ((size_t)K * h->plan.split_rows.total + 1) / 2);
    h->enc_wr = reinterpret_cast<unsigned short*>(tmp);
  }
  ALLOC(h->flow_w, (size_t)K * FW_SIZE);
  ALLOC(h->mfma_w, (size_t)K * MW_SIZE);
  {
    float* tmp = nullptr;
    ALLOC(tmp, (size_t)K * MH_SIZE);
    h->split_w = reinterpret_cast<uint32_t*>(tmp);
  }
  for (int i = 0; i < 4; ++i) ALLOC(h->bufs[i], h->buf_floats);
  ALLOC(h->visual, (size_t)max_batch * in_channels * 100 * 100);
  ALLOC(h->z, (size_t)K * max_batch * 64);
  // plan-search scratch: nothing is (re)allocated after this point (include/rip_hip.h: calls never synchronise)
  ALLOC(h->plans, (size_t)max_batch * max_candidates * 8);
  ALLOC(h->loss_best, (size_t)max_batch * max_candidates);
  ALLOC(h->trace_loss, (size_t)RIP_MAX_STEPS * max_batch);
  ALLOC(h->trace_x, (size_t)RIP_MAX_STEPS * max_batch * 8);
  {
    float* tmp = nullptr;
    ALLOC(tmp, 4);  // the adjoint counter (8 bytes) + the split kernel's operand-range word
    h->stats = reinterpret_cast<unsigned long long*>(tmp);
    h->range_flag = reinterpret_cast<unsigned*>(tmp + 2);
    (void)hipMemset(h->stats, 0, 4 * sizeof(float));
  }
  // scratch of the MFMA search kernels (adjoint tape, prefix table): 0 when neither can ever run for this handle
  h->tape_bytes = search_phase_scratch_bytes(max_batch, max_candidates, K);
  if (search_split_scratch_bytes(max_batch, max_candidates, K) > h->tape_bytes)
    h->tape_bytes = search_split_scratch_bytes(max_batch, max_candidates, K);
  if (h->tape_bytes > 0) {
    float* tmp = nullptr;
    ALLOC(tmp, (h->tape_bytes + 3) / 4);
    h->tape = tmp;
  }
#undef ALLOC
  *out = h;
  return RIP_OK;
}

int rip_destroy(rip_handle* h) {
  if (h == nullptr) return RIP_OK;
  DeviceScope scope(h->device);
  if (h->order != nullptr) (void)hipEventDestroy(h->order);
  if (h->tape != nullptr) (void)hipFree(h->tape);
  if (h->mega_ticks != nullptr) {  // development: where the one-launch encoder spends its time (model 0, last call)
    unsigned long long t[128];
    if (hipDeviceSynchronize() == hipSuccess && hipMemcpy(t, h->mega_ticks, sizeof(t), hipMemcpyDeviceToHost) == hipSuccess) {
      const int n = (int)h->plan.layers.size();
      for (int i = 1; i <= n; ++i)
        fprintf(stderr, "mega layer %2d: %6.2f us (workgroup 0 busy %5.2f us) %s %d -> %d @ %d\n", i,
                (double)(t[i] - t[i - 1]) * 0.01, i < n ? (double)(t[64 + i] - t[i - 1]) * 0.01 : 0.0,
                i < n ? (h->plan.layers[i].kind == L_DW ? "dw" : "pw") : "tail", i < n ? h->plan.layers[i].cin : 0,
                i < n ? h->plan.layers[i].cout : 0, i < n ? h->plan.layers[i].h_out : 0);
      fprintf(stderr, "mega layers 1..%d + tail: %.2f us\n", n, (double)(t[n] - t[0]) * 0.01);
    }
    (void)hipFree(h->mega_ticks);
  }
  if (h->mega_status != nullptr) (void)hipHostFree(h->mega_status);
  if (h->mega_sync != nullptr) (void)hipFree(h->mega_sync);
  if (h->mega_arena != nullptr) (void)hipFree(h->mega_arena);
  float* ptrs[] = {h->enc_w, h->enc_wt, reinterpret_cast<float*>(h->enc_wh), reinterpret_cast<float*>(h->enc_wc), reinterpret_cast<float*>(h->enc_wr), h->flow_w, h->mfma_w, reinterpret_cast<float*>(h->split_w), h->bufs[0], h->bufs[1], h->bufs[2], h->bufs[3], h->visual,
                   h->z,     h->plans,  h->loss_best, h->trace_loss, h->trace_x, reinterpret_cast<float*>(h->stats)};
  for (float* p : ptrs)
    if (p != nullptr) (void)hipFree(p);
  delete h;
  return RIP_OK;
}

// Scratch of the one-launch encoder, allocated when the option is first switched on (rip_set_option is not a stream
// call).  Only where workgroup i of a launch lands on XCD i % 8 (probed once per device); false = not available here.
static bool mega_setup(rip_handle* h) {
  if (h->mega_max_b > 0) return true;
  if (!encoder_mega_probe(h->device)) return false;
  int mb = h->max_batch < 4 ? h->max_batch : 4;
  while (mb > 0 && !encoder_mega_supported(h->plan, mb, h->K)) --mb;
  if (mb <= 0) return false;
  h->mega_stride = encoder_mega_arena_floats(h->plan, mb);
  bool ok = hipMalloc((void**)&h->mega_arena, (size_t)h->K * h->mega_stride * sizeof(float)) == hipSuccess;
  ok = ok && hipMalloc((void**)&h->mega_sync, 8 * 64 * sizeof(unsigned)) == hipSuccess;
  ok = ok && hipMemset(h->mega_sync, 0, 8 * 64 * sizeof(unsigned)) == hipSuccess;
  ok = ok && hipHostMalloc((void**)&h->mega_status, 64, hipHostMallocDefault) == hipSuccess;
  if (const char* e = getenv("RIP_MEGA_TICKS"); ok && e != nullptr && e[0] == '1') {  // development
    ok = hipMalloc((void**)&h->mega_ticks, 128 * sizeof(unsigned long long)) == hipSuccess &&
         hipMemset(h->mega_ticks, 0, 128 * sizeof(unsigned long long)) == hipSuccess;
  }
  if (!ok) {
    if (h->mega_arena != nullptr) (void)hipFree(h->mega_arena);
    if (h->mega_sync != nullptr) (void)hipFree(h->mega_sync);
    if (h->mega_status != nullptr) (void)hipHostFree(h->mega_status);
    if (h->mega_ticks != nullptr) (void)hipFree(h->mega_ticks);
    h->mega_arena = nullptr;
    h->mega_sync = nullptr;
    h->mega_status = nullptr;
    h->mega_ticks = nullptr;
    return false;
  }
  *h->mega_status = 0;
  if (const char* e = getenv("RIP_MEGA_WGS")) {  // development: workgroups per XCD
    const int v = atoi(e);
    if (v >= 1 && v <= 128) h->mega_wgs = v;
  }
  h->mega_max_b = mb;
  return true;
}

int rip_set_option(rip_handle* h, int option, int value) {
  REQUIRE(h != nullptr, "handle is NULL");
  switch (option) {
    case RIP_OPT_SEARCH_KERNEL:
      REQUIRE(value != 2, "search kernel 2 (round 1's fp32-MFMA wave-per-model pipeline) was removed in round 5: no default "
              "reached it since round 2; use 3 (fp32-MFMA phase-sequential) or 4 (split-f16)");
      REQUIRE(value >= 0 && value <= 5, "search kernel %d not in {0 auto, 1 wave-per-chain, 3 phase, 4 split, 5 split (paired shape)}", value);
      h->search_mode = value;
      return RIP_OK;
    case RIP_OPT_ENCODER_FUSED:
      REQUIRE(value >= -1 && value <= 17, "encoder_fused must be in [-1,17] (got %d)", value);
      h->encoder_fused = value;
      return RIP_OK;
    case RIP_OPT_SEARCH_REGROUP:
      REQUIRE(value == 0 || value == 1, "search regroup must be 0 or 1 (got %d)", value);
      return RIP_OK;  // retired in round 5 (no faster in two rounds of measurements): accepted, no effect
    case RIP_OPT_ENCODER_MEGA:
      REQUIRE(value >= -1 && value <= 1, "encoder_mega must be -1 (auto), 0 or 1 (got %d)", value);
      if (value == 1) {
        DeviceScope scope(h->device);
        if (scope.err != hipSuccess) return fail(RIP_EHIP, "hipSetDevice(%d) failed: %s", h->device, hipGetErrorString(scope.err));
        (void)mega_setup(h);  // not available (placement probe / memory): the option is accepted and has no effect
      }
      h->encoder_mega = value;
      return RIP_OK;
    case RIP_OPT_ENCODER_VARIANT:
      REQUIRE(value >= 0 && value <= 31, "encoder variant mask %d not in [0,31] (2 round-3 front, 8 features.17 layer-wise, 16 fp32 encoder without split-f16 tile blocks; 1 and 4 selected round 1's row-streaming kernel, retired in round 6: accepted, no effect)", value);
      h->encoder_variant = value;
      return RIP_OK;
    case RIP_OPT_KERNEL_LOG:
      REQUIRE(value == 0 || value == 1, "kernel log must be 0 or 1 (got %d)", value);
      h->kernel_log_on = value == 1;
      h->klog.text.clear();
      return RIP_OK;
    case RIP_OPT_DEBUG_ENCODER_FAULT:
      // test hook: raise the one-launch encoder's failure word as its kernel would (tests of the caller's recovery path)
      REQUIRE(value == 1 || value == 2, "encoder fault code must be 1 (placement) or 2 (barrier timeout), got %d", value);
      if (h->mega_status == nullptr) return fail(RIP_ESTATE, "the one-launch encoder is not set up on this handle (RIP_OPT_ENCODER_MEGA = 1 first)");
      *h->mega_status = value;
      h->mega_reported = false;
      return RIP_OK;
    default:
      return fail(RIP_EINVAL, "unknown option %d", option);
  }
}

int rip_num_models(const rip_handle* h) { return h ? h->K : RIP_EINVAL; }
int rip_in_channels(const rip_handle* h) { return h ? h->C : RIP_EINVAL; }
int rip_max_batch(const rip_handle* h) { return h ? h->max_batch : RIP_EINVAL; }
int rip_max_candidates(const rip_handle* h) { return h ? h->max_candidates : RIP_EINVAL; }

int rip_load_model(rip_handle* h, int k, const float* packed_host, size_t numel) {
  REQUIRE(h != nullptr && packed_host != nullptr, "NULL argument");
  REQUIRE(k >= 0 && k < h->K, "model index %d outside [0,%d)", k, h->K);
  std::vector<float> enc, flow, mw;
  std::vector<uint32_t> mh;
  const char* err = "";
  float wmax = 0.f;
  if (!fold_and_pack(h->plan, packed_host, numel, enc, flow, mw, mh, &err, &wmax)) return fail(RIP_EINVAL, "%s (got %zu floats)", err, numel);
  h->split_wmax[k] = wmax;
  DeviceScope scope(h->device);
  if (scope.err != hipSuccess) return fail(RIP_EHIP, "hipSetDevice(%d) failed: %s", h->device, hipGetErrorString(scope.err));
  // setup call: synchronous copies; make sure no kernel of an earlier call still reads the old weights
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(h->enc_w + (size_t)k * h->plan.blob_floats, enc.data(), enc.size() * sizeof(float), hipMemcpyHostToDevice));
  {
    std::vector<unsigned short> wh(enc.size());
    for (size_t i = 0; i < enc.size(); ++i) {  // round to nearest even
      unsigned u;
      std::memcpy(&u, &enc[i], 4);
      wh[i] = (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
    }
    HIP_TRY(hipMemcpy(h->enc_wh + (size_t)k * h->plan.blob_floats, wh.data(), wh.size() * 2, hipMemcpyHostToDevice));
  }
  {
    std::vector<float> wt(enc);
    for (const Layer& l : h->plan.layers) {
      if (l.kind != L_DW) continue;
      for (size_t i = 0; i < (size_t)9 * l.cout; ++i) {  // [tap][channel] at w_off: round to nearest even, keep as fp32
        unsigned u;
        std::memcpy(&u, &wt[l.w_off + i], 4);
        u = ((u + 0x7fffu + ((u >> 16) & 1u)) >> 16) << 16;
        std::memcpy(&wt[l.w_off + i], &u, 4);
      }
    }
    HIP_TRY(hipMemcpy(h->enc_wt + (size_t)k * h->plan.blob_floats, wt.data(), wt.size() * sizeof(float), hipMemcpyHostToDevice));
  }
  {
    // the fp32 encoder's split-f16 blocks: two-term binary16 operands of w * 2^8 (encoder.h: SPLIT_ENC_W_SCALE), packed as the
    // kernels read them; only the pointwise layers go through them
    bool ok = true;
    for (const Layer& l : h->plan.layers) {
      if (l.kind != L_PW) continue;
      for (size_t i = 0; i < (size_t)l.cin * l.cout; ++i) ok = ok && std::fabs(enc[l.w_off + i]) < SPLIT_ENC_W_LIMIT;  // (false for NaN)
    }
    h->enc_split_ok[k] = ok;
    std::vector<unsigned short> rec(h->plan.split_tiles.total);
    pack_split_tiles(h->plan, h->plan.split_tiles, enc.data(), rec.data());
    HIP_TRY(hipMemcpy(h->enc_wc + (size_t)k * h->plan.split_tiles.total, rec.data(), rec.size() * 2, hipMemcpyHostToDevice));
    std::vector<unsigned short> frag(h->plan.split_rows.total);
    pack_split_rows(h->plan, h->plan.split_rows, enc.data(), frag.data());
    HIP_TRY(hipMemcpy(h->enc_wr + (size_t)k * h->plan.split_rows.total, frag.data(), frag.size() * 2, hipMemcpyHostToDevice));
  }
  HIP_TRY(hipMemcpy(h->flow_w + (size_t)k * FW_SIZE, flow.data(), flow.size() * sizeof(float), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(h->mfma_w + (size_t)k * MW_SIZE, mw.data(), mw.size() * sizeof(float), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(h->split_w + (size_t)k * MH_SIZE, mh.data(), mh.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  h->loaded[k] = true;
  return RIP_OK;
}

int rip_transform(const float* lidar_dev, int B, int C, int H, int W, int channels_last, int out_hw, float* out_dev,
                  rip_stream_t stream) {
  REQUIRE(lidar_dev != nullptr && out_dev != nullptr, "NULL argument");
  REQUIRE(B >= 1 && C >= 1 && H >= 1 && W >= 1 && out_hw >= 1, "bad shape B=%d C=%d H=%d W=%d out=%d", B, C, H, W, out_hw);
  HIP_TRY(launch_transform(lidar_dev, B, C, H, W, channels_last, out_hw, out_dev, (hipStream_t)stream));
  return RIP_OK;
}

// Does an fp32 encode of B observations take the one-launch kernel?  Never after its protocol failed once.
static bool mega_applies(const rip_handle* h, int B) {
  if (h->mega_max_b <= 0 || B > h->mega_max_b || h->encoder_mega != 1) return false;
  if (*h->mega_status != 0) return false;
  return h->encoder_mega == 1;  // auto = off: measured 251-273 us against 244 us of layer-wise launches
}

int rip_kernel_log(const rip_handle* h, char* buf, size_t cap) {
  if (h == nullptr) return RIP_EINVAL;
  const size_t n = h->klog.text.size();
  if (buf != nullptr && cap > 0) {
    const size_t m = n < cap - 1 ? n : cap - 1;
    memcpy(buf, h->klog.text.data(), m);
    buf[m] = '\0';
  }
  return (int)n;
}

int rip_encoder_status(rip_handle* h) {
  if (h == nullptr) return RIP_EINVAL;
  // one-shot: the word itself stays raised (mega_applies keeps the handle on the layer-wise launches), but it is handed
  // out once — a caller that repeats the failed call and asks again must read 0, not the old failure
  if (h->mega_status == nullptr || *h->mega_status == 0 || h->mega_reported) return 0;
  h->mega_reported = true;
  return *h->mega_status;
}

// The packed two-term binary16 operands of the fp32 encoder's split-f16 blocks, or NULLs (layer-wise fp32 kernels):
// every model of the launch must be inside the operand range, and RIP_OPT_ENCODER_VARIANT bit 16 turns the blocks off.
struct SplitPlanes {
  const unsigned short* tiles = nullptr;  // chunk records of the tile blocks
  size_t tiles_stride = 0;
  const unsigned short* rows = nullptr;   // operand fragments of the row-streaming blocks
  size_t rows_stride = 0;
};
static SplitPlanes split_planes(const rip_handle* h, int k_begin, int k_count) {
  SplitPlanes sp;
  if (h->encoder_variant & ENC_VAR_FP32_LAYERWISE) return sp;
  if (h->encoder_fused >= 0) return sp;  // an explicit RIP_OPT_ENCODER_FUSED count asks for exactly that split of the true-fp32 kernels
  for (int k = k_begin; k < k_begin + k_count; ++k)
    if (!h->enc_split_ok[k]) return sp;
  sp.tiles = h->enc_wc;
  sp.tiles_stride = h->plan.split_tiles.total;
  sp.rows = h->enc_wr;
  sp.rows_stride = h->plan.split_rows.total;
  return sp;
}

int rip_encode(rip_handle* h, const float* visual_dev, const float* vec_dev, int B, int k_begin, int k_count,
               int enc_dtype, float* z_dev, float* feat_dev, rip_stream_t stream) {
  int rc = check_models(h, k_begin, k_count);
  if (rc != RIP_OK) return rc;
  REQUIRE(visual_dev != nullptr && vec_dev != nullptr && z_dev != nullptr, "NULL argument");
  REQUIRE(B >= 1 && B <= h->max_batch, "B=%d outside [1,max_batch=%d]", B, h->max_batch);
  REQUIRE(enc_dtype == RIP_ENC_FP32 || enc_dtype == RIP_ENC_BF16, "unknown encoder dtype %d", enc_dtype);
  ENTER(h, stream);
  TraceRange range_(enc_dtype == RIP_ENC_BF16 ? "rip_encode (bf16)" : "rip_encode (fp32)");
  KernelLogScope log_(h);
  if (enc_dtype == RIP_ENC_BF16) {
    HIP_TRY(launch_encoder_bf16(h->plan, h->enc_wt, h->enc_wh, k_begin, k_count, visual_dev, vec_dev, B, h->bufs, z_dev,
                                feat_dev, h->encoder_fused, (hipStream_t)stream, nullptr, h->encoder_variant));
    return RIP_OK;
  }
  if (mega_applies(h, B)) {
    HIP_TRY(launch_encoder_mega(h->plan, h->enc_w, k_begin, k_count, visual_dev, vec_dev, B, h->mega_arena, h->mega_stride,
                                h->mega_sync, h->mega_status, h->mega_ticks, z_dev, feat_dev, h->mega_wgs,
                                (hipStream_t)stream));
    return RIP_OK;
  }
  const SplitPlanes sp = split_planes(h, k_begin, k_count);
  HIP_TRY(launch_encoder(h->plan, h->enc_w, k_begin, k_count, visual_dev, vec_dev, B, h->bufs, z_dev, feat_dev,
                         h->encoder_fused >= 0 ? h->encoder_fused : (B >= 8 ? 3 : 0), (hipStream_t)stream, nullptr, sp.tiles, sp.tiles_stride, sp.rows, sp.rows_stride));
  return RIP_OK;
}

int rip_encode_tap(rip_handle* h, const float* visual_dev, int B, int k, int enc_dtype, int layer, float* dst_dev,
                   size_t dst_numel, rip_stream_t stream) {
  return rip_encode_tap_k(h, visual_dev, B, k, 1, enc_dtype, layer, dst_dev, dst_numel, stream);
}

int rip_encode_tap_k(rip_handle* h, const float* visual_dev, int B, int k_begin, int k_count, int enc_dtype, int layer,
                     float* dst_dev, size_t dst_numel, rip_stream_t stream) {
  int rc = check_models(h, k_begin, k_count);
  if (rc != RIP_OK) return rc;
  REQUIRE(visual_dev != nullptr, "NULL argument");
  REQUIRE(dst_dev != nullptr || dst_numel == 0, "dst_dev is NULL with dst_numel=%zu (NULL + 0 = run up to the layer, copy nothing)", dst_numel);
  REQUIRE(B >= 1 && B <= h->max_batch, "B=%d outside [1,max_batch=%d]", B, h->max_batch);
  REQUIRE(enc_dtype == RIP_ENC_FP32 || enc_dtype == RIP_ENC_BF16, "unknown encoder dtype %d", enc_dtype);
  const int L = (int)h->plan.layers.size();
  REQUIRE(layer >= 0 && layer < L, "layer %d outside [0,%d)", layer, L);
  const Layer& l = h->plan.layers[layer];
  const bool pooled = layer + 1 == L && h->plan.final_hw == 4;
  const size_t need = (size_t)k_count * B * (pooled ? 1 : (size_t)l.h_out * l.h_out) * l.cout;
  REQUIRE(dst_dev == nullptr || dst_numel == need, "dst_numel=%zu, layer %d of %d model(s) x B=%d observations has %zu elements",
          dst_numel, layer, k_count, B, need);
  ENTER(h, stream);
  KernelLogScope log_(h);
  EncoderTap tap;
  tap.layer = layer;
  tap.dst = dst_dev;
  if (enc_dtype == RIP_ENC_BF16)
    HIP_TRY(launch_encoder_bf16(h->plan, h->enc_wt, h->enc_wh, k_begin, k_count, visual_dev, nullptr, B, h->bufs, nullptr, nullptr,
                                h->encoder_fused, (hipStream_t)stream, &tap, h->encoder_variant));
  else {
    const SplitPlanes sp = split_planes(h, k_begin, k_count);  // the same kernel selection as rip_encode
    HIP_TRY(launch_encoder(h->plan, h->enc_w, k_begin, k_count, visual_dev, nullptr, B, h->bufs, nullptr, nullptr,
                           h->encoder_fused >= 0 ? h->encoder_fused : (B >= 8 ? 3 : 0), (hipStream_t)stream, &tap, sp.tiles, sp.tiles_stride, sp.rows, sp.rows_stride));
  }
  if (!tap.served)
    return fail(RIP_EINVAL, "layer %d is inside a fused block under the current RIP_OPT_ENCODER_FUSED setting / kernel selection: its output "
                "never reaches memory (tap the block's last layer, or set the option to 0 and RIP_OPT_ENCODER_VARIANT bit 16)", layer);
  return RIP_OK;
}

int rip_encode_raw(rip_handle* h, const float* lidar_dev, int channels_last, int H, int W, const float* vec_dev, int B,
                   int k_begin, int k_count, int enc_dtype, float* z_dev, rip_stream_t stream) {
  int rc = check_models(h, k_begin, k_count);
  if (rc != RIP_OK) return rc;
  REQUIRE(lidar_dev != nullptr, "NULL argument");
  REQUIRE(B >= 1 && B <= h->max_batch, "B=%d outside [1,max_batch=%d]", B, h->max_batch);
  REQUIRE(H >= 1 && W >= 1, "bad BEV size H=%d W=%d", H, W);
  {
    ENTER(h, stream);
    HIP_TRY(launch_transform(lidar_dev, B, h->C, H, W, channels_last, 100, h->visual, (hipStream_t)stream));
  }
  return rip_encode(h, h->visual, vec_dev, B, k_begin, k_count, enc_dtype, z_dev, nullptr, stream);
}

int rip_encode_raw_u8(rip_handle* h, const uint8_t* codes_dev, const float* lut_dev, int H, int W, const float* vec_dev,
                      int B, int k_begin, int k_count, int enc_dtype, float* z_dev, rip_stream_t stream) {
  int rc = check_models(h, k_begin, k_count);
  if (rc != RIP_OK) return rc;
  REQUIRE(codes_dev != nullptr && lut_dev != nullptr, "NULL argument");
  REQUIRE(B >= 1 && B <= h->max_batch, "B=%d outside [1,max_batch=%d]", B, h->max_batch);
  REQUIRE(H >= 1 && W >= 1, "bad BEV size H=%d W=%d", H, W);
  REQUIRE(transform_coded_supported(h->C, H, W, 100), "coded BEV: C=%d H=%d W=%d not supported (C <= 3, down-sampling by <= 2)",
          h->C, H, W);
  {
    ENTER(h, stream);
    HIP_TRY(launch_transform_coded(codes_dev, lut_dev, B, h->C, H, W, 100, h->visual, (hipStream_t)stream));
  }
  return rip_encode(h, h->visual, vec_dev, B, k_begin, k_count, enc_dtype, z_dev, nullptr, stream);
}

int rip_flow_forward(rip_handle* h, int k, const float* x_dev, const float* z_dev, int N, int z_rows, float* y_dev,
                     float* logabsdet_dev, rip_stream_t stream) {
  int rc = check_models(h, k, 1);
  if (rc != RIP_OK) return rc;
  REQUIRE(x_dev != nullptr && z_dev != nullptr && y_dev != nullptr, "NULL argument");
  REQUIRE(N >= 0 && (z_rows == N || z_rows == 1), "z_rows=%d must be N=%d or 1", z_rows, N);
  if (N == 0) return RIP_OK;
  ENTER(h, stream);
  HIP_TRY(launch_flow_forward(h->flow_w + (size_t)k * FW_SIZE, x_dev, z_dev, N, z_rows, y_dev, logabsdet_dev,
                              (hipStream_t)stream));
  return RIP_OK;
}

int rip_flow_inverse(rip_handle* h, int k, const float* y_dev, const float* z_dev, int N, int z_rows, float* x_dev,
                     float* log_prob_dev, float* logabsdet_dev, rip_stream_t stream) {
  int rc = check_models(h, k, 1);
  if (rc != RIP_OK) return rc;
  REQUIRE(y_dev != nullptr && z_dev != nullptr, "NULL argument");
  REQUIRE(N >= 0 && (z_rows == N || z_rows == 1), "z_rows=%d must be N=%d or 1", z_rows, N);
  if (N == 0) return RIP_OK;
  ENTER(h, stream);
  HIP_TRY(launch_flow_inverse(h->flow_w + (size_t)k * FW_SIZE, y_dev, z_dev, N, z_rows, x_dev, log_prob_dev,
                              logabsdet_dev, (hipStream_t)stream));
  return RIP_OK;
}

int rip_goal_likelihood(const float* y_dev, const float* goal_dev, int N, int goal_rows, int G, float epsilon,
                        float* rows_dev, rip_stream_t stream) {
  REQUIRE(y_dev != nullptr && goal_dev != nullptr && rows_dev != nullptr, "NULL argument");
  REQUIRE(N >= 0 && G >= 1 && (goal_rows == N || goal_rows == 1), "bad shape N=%d goal_rows=%d G=%d", N, goal_rows, G);
  REQUIRE(epsilon > 0.f, "epsilon must be positive");
  if (N == 0) return RIP_OK;
  HIP_TRY(launch_goal_rows(y_dev, goal_dev, N, goal_rows, G, epsilon, rows_dev, (hipStream_t)stream));
  return RIP_OK;
}

int rip_score(rip_handle* h, int k_begin, int k_count, const float* z_dev, const float* y_dev, const float* goal_dev,
              int B, int N, int G, float epsilon, float* S_dev, rip_stream_t stream) {
  int rc = check_models(h, k_begin, k_count);
  if (rc != RIP_OK) return rc;
  REQUIRE(z_dev != nullptr && y_dev != nullptr && S_dev != nullptr, "NULL argument");
  REQUIRE(B >= 1 && N >= 1, "bad shape B=%d N=%d", B, N);
  REQUIRE(goal_dev == nullptr || (G >= 1 && epsilon > 0.f), "bad goal arguments G=%d eps=%g", G, epsilon);
  ENTER(h, stream);
  HIP_TRY(launch_score(h->flow_w, k_begin, k_count, z_dev, y_dev, goal_dev, B, N, G, epsilon, S_dev, (hipStream_t)stream));
  return RIP_OK;
}

int rip_aggregate_scores(const float* S_dev, int K, int B, int N, int algorithm, float* loss_dev,
                         int32_t* best_index_dev, rip_stream_t stream) {
  REQUIRE(S_dev != nullptr, "S_dev is NULL");
  REQUIRE(K >= 1 && B >= 1 && N >= 1, "bad shape K=%d B=%d N=%d", K, B, N);
  REQUIRE(algorithm == RIP_ALGO_WCM || algorithm == RIP_ALGO_MA || algorithm == RIP_ALGO_BCM, "unknown algorithm %d", algorithm);
  HIP_TRY(launch_aggregate_scores(S_dev, K, B, N, algorithm, loss_dev, best_index_dev, (hipStream_t)stream));
  return RIP_OK;
}

int rip_lidar_bev(const float* points_dev, const int32_t* offsets_dev, int B, float* bev_dev, rip_stream_t stream) {
  REQUIRE(B >= 0, "bad batch B=%d", B);
  REQUIRE(B == 0 || (offsets_dev != nullptr && bev_dev != nullptr), "offsets_dev / bev_dev is NULL");
  HIP_TRY(launch_lidar_bev(points_dev, offsets_dev, B, bev_dev, (hipStream_t)stream));
  return RIP_OK;
}

int rip_cil_decode(const float* feat_dev, const float* vec_dev, const float* weights_dev, int B, int T, float* y_dev,
                   rip_stream_t stream) {
  REQUIRE(B >= 0 && T >= 1, "bad shape B=%d T=%d", B, T);
  REQUIRE(B == 0 || (feat_dev != nullptr && vec_dev != nullptr && weights_dev != nullptr && y_dev != nullptr),
          "NULL argument");
  HIP_TRY(launch_cil_decode(feat_dev, vec_dev, weights_dev, B, T, y_dev, (hipStream_t)stream));
  return RIP_OK;
}

int rip_cil_blob_floats(void) { return cil_blob_floats(); }

// kernel choice: the MFMA-batched kernels win once there are enough 16-candidate blocks to fill the chip; the
// wave-per-chain kernel has the lower latency for a single observation.  1 = wave-per-chain, 3 = fp32-MFMA
// phase-sequential, 4 = split-f16 phase-sequential (the default of large launches).
static bool split_weights_ok(const rip_handle* h) {
  for (int k = 0; k < h->K; ++k)
    if (h->split_wmax[k] >= SPLIT_W_LIMIT) return false;
  return true;
}
// 5 = the split-f16 kernel with the paired workgroup shape forced (flow_pair.hip; `auto` picks it by the launcher's cost model)
static int split_shape_of(const rip_handle* h) { return h->search_mode == 5 ? rip::SPLIT_SHAPE_PAIR : 0; }
static int pick_search_kernel(const rip_handle* h, int B, int N) {
  if (h->search_mode == 5) return 4;
  if (h->search_mode != 0) return h->search_mode;
  // crossover measured at K = 4, N = 128 (round 5): the wave-per-chain kernel costs 64.5 us per observation, a launch of
  // the split kernel one workgroup-time (630 us) up to 1024 blocks: 10 observations (517 / 646 / 773 / 1028 us against
  // 629 / 631 / 632 / 632 at B = 8 / 10 / 12 / 16)
  const bool big = (size_t)B * N >= 1280;
  if (!(big && N % 16 == 0 && h->K <= RIP_MAX_MODELS)) return 1;
  return split_weights_ok(h) ? 4 : 3;  // a flow weight beyond the binary16 operand range: the fp32-MFMA kernel
}

static int search_impl(rip_handle* h, const float* z_dev, const float* goal_dev, const float* x0_dev, int B, int N, int G,
                       int algorithm, int num_steps, float lr, float epsilon, float* plan_dev, float* plans_dev,
                       float* loss_best_dev, int32_t* best_index_dev, float* trace_post_dev, float* trace_x_dev,
                       float* trace_grad_dev, double* plan_interp_dev, rip_stream_t stream) {
  int rc = check_models(h, 0, h ? h->K : 1);
  if (rc != RIP_OK) return rc;
  REQUIRE(z_dev != nullptr && x0_dev != nullptr, "NULL argument");
  REQUIRE(B >= 1 && N >= 1, "bad shape B=%d N=%d", B, N);
  REQUIRE(goal_dev == nullptr || (G >= 1 && G <= rip::MAX_GOALS), "G=%d must be in [1,%d] with a goal", G, rip::MAX_GOALS);
  REQUIRE(algorithm == RIP_ALGO_WCM || algorithm == RIP_ALGO_MA || algorithm == RIP_ALGO_BCM, "unknown algorithm %d", algorithm);
  REQUIRE(num_steps >= 0 && num_steps <= RIP_MAX_STEPS, "num_steps=%d outside [0,%d]", num_steps, RIP_MAX_STEPS);
  REQUIRE(epsilon > 0.f && lr > 0.f, "lr and epsilon must be positive");
  const bool need_select = plan_dev != nullptr || best_index_dev != nullptr || plan_interp_dev != nullptr;
  float* plans = plans_dev;
  float* lbest = loss_best_dev;
  if (need_select && (plans == nullptr || lbest == nullptr)) {
    if ((size_t)B * N > (size_t)h->max_batch * h->max_candidates)
      return fail(RIP_ESTATE, "B*N=%d*%d exceeds the scratch rip_create sized (max_batch=%d x max_candidates=%d)", B, N,
                  h->max_batch, h->max_candidates);
    if (plans == nullptr) plans = h->plans;
    if (lbest == nullptr) lbest = h->loss_best;
  }
  ENTER(h, stream);
  TraceRange range_("rip_search");
  SearchArgs a;
  a.flow_w = h->flow_w;
  a.k0 = 0;
  a.K = h->K;
  a.z = z_dev;
  a.goal = goal_dev;
  a.x0 = x0_dev;
  a.B = B;
  a.N = N;
  a.G = G;
  a.algorithm = algorithm;
  a.num_steps = num_steps;
  a.lr = lr;
  a.epsilon = epsilon;
  a.grad_scale = 1.0f;
  a.plans = plans;
  a.loss_best = lbest;
  a.trace_post = trace_post_dev;
  a.trace_x = trace_x_dev;
  a.trace_loss = nullptr;
  a.trace_grad = trace_grad_dev;
  a.stats = h->stats;
  a.split_shape = split_shape_of(h);
  // kernel choice: the MFMA-batched kernels win once there are enough 16-candidate blocks to fill the chip; the
  // wave-per-chain kernel has the lower latency for a single observation.  Among the MFMA kernels the phase-sequential
  // ones (operands in LDS, any K) are the default of large launches.
  // crossover measured at K = 4, N = 128: the chain kernel costs 64 us per observation, the phase kernel 1.1 ms per
  // launch up to one workgroup per CU (B = 16: 1.02 vs 1.11 ms, B = 32: 2.03 vs 1.11 ms)
  const int kernel = pick_search_kernel(h, B, N);
  if (kernel == 4 && !split_weights_ok(h))
    return fail(RIP_EINVAL, "the split-f16 search kernel carries the flow weights as binary16 terms of w * 2^8: a model of this handle "
                "has a flow weight of magnitude >= %g; use search kernel 0 (auto) or 3", (double)SPLIT_W_LIMIT);
  if ((kernel == 3 && !search_phase_supported(a)) || (kernel == 4 && !search_split_supported(a)))
    return fail(RIP_EINVAL, "phase-sequential MFMA search needs N%%16==0 and K<=%d (K=%d N=%d)", RIP_MAX_MODELS, h->K, N);
  if (kernel != 1) {
    const size_t need = kernel == 4 ? search_split_scratch_bytes(B, N, h->K) : search_phase_scratch_bytes(B, N, h->K);
    if (need > h->tape_bytes)
      return fail(RIP_ESTATE, "MFMA search scratch for B=%d N=%d needs %zu B, rip_create sized %zu B (max_batch=%d x "
                  "max_candidates=%d)", B, N, need, h->tape_bytes, h->max_batch, h->max_candidates);
    if (kernel == 4) {
      // hidden states are split into binary16 terms unscaled (|h| <= max(1, |z|)): a launch with some |z| >= 2^14 raises
      // a device word in its prefix kernel, the split kernel then returns at once and the fp32-MFMA kernel queued behind
      // it (which returns at once otherwise: ~10 us of empty workgroups per launch) runs the search — no silent inf
      a.range_flag = h->range_flag;
      HIP_TRY(launch_search_split(a, h->split_w, h->tape, (hipStream_t)stream));
      if (search_phase_supported(a) && search_phase_scratch_bytes(B, N, h->K) <= h->tape_bytes) {
        a.run_if_flag = h->range_flag;
        a.range_flag = nullptr;
        HIP_TRY(launch_search_phase(a, h->mfma_w, h->tape, (hipStream_t)stream));
      }
    } else {
      HIP_TRY(launch_search_phase(a, h->mfma_w, h->tape, (hipStream_t)stream));
    }
  } else {
    HIP_TRY(launch_search(a, (hipStream_t)stream));
  }
  if (need_select)
    HIP_TRY(launch_select_best(plans, lbest, B, N, plan_dev, best_index_dev, plan_interp_dev, (hipStream_t)stream));
  return RIP_OK;
}

int rip_search(rip_handle* h, const float* z_dev, const float* goal_dev, const float* x0_dev, int B, int N, int G,
               int algorithm, int num_steps, float lr, float epsilon, float* plan_dev, float* plans_dev,
               float* loss_best_dev, int32_t* best_index_dev, float* trace_post_dev, float* trace_x_dev,
               float* trace_grad_dev, rip_stream_t stream) {
  return search_impl(h, z_dev, goal_dev, x0_dev, B, N, G, algorithm, num_steps, lr, epsilon, plan_dev, plans_dev,
                     loss_best_dev, best_index_dev, trace_post_dev, trace_x_dev, trace_grad_dev, nullptr, stream);
}

int rip_search_stats(rip_handle* h, uint64_t* adjoint_passes, int reset) {
  REQUIRE(h != nullptr, "handle is NULL");
  DeviceScope scope(h->device);
  if (scope.err != hipSuccess) return fail(RIP_EHIP, "hipSetDevice(%d) failed: %s", h->device, hipGetErrorString(scope.err));
  HIP_TRY(hipDeviceSynchronize());  // diagnostic call: waits for the launches it reports on
  unsigned long long v = 0;
  HIP_TRY(hipMemcpy(&v, h->stats, sizeof(v), hipMemcpyDeviceToHost));
  if (adjoint_passes != nullptr) *adjoint_passes = v;
  if (reset) HIP_TRY(hipMemset(h->stats, 0, sizeof(v)));
  return RIP_OK;
}

int rip_trace_push(const char* name) {
  if (name == nullptr || roctx().push == nullptr) return 0;
  roctx().push(name);
  return 1;
}
int rip_trace_pop(void) {
  if (roctx().pop == nullptr) return 0;
  roctx().pop();
  return 1;
}

int rip_search_plan(const rip_handle* h, int B, int N, int32_t* out, int n_out) {
  REQUIRE(h != nullptr && out != nullptr, "NULL argument");
  REQUIRE(B >= 1 && N >= 1 && n_out >= 10, "bad arguments B=%d N=%d n_out=%d (needs 10 slots)", B, N, n_out);
  for (int i = 0; i < n_out; ++i) out[i] = 0;
  const int kernel = pick_search_kernel(h, B, N);
  out[0] = kernel;
  if (kernel == 3 && N % 16 == 0) search_phase_info(B, N, h->K, out + 1);
  if (kernel == 4 && N % 16 == 0) search_split_info(B, N, h->K, out + 1, split_shape_of(h));
  return RIP_OK;
}

int rip_interpolate_plans(const float* plan_dev, int B, double* out_dev, rip_stream_t stream) {
  REQUIRE(B >= 0, "bad batch B=%d", B);
  REQUIRE(B == 0 || (plan_dev != nullptr && out_dev != nullptr), "NULL argument");
  if (B == 0) return RIP_OK;
  // stateless: launch on the device that owns the plans, whatever the caller's current device is (like rip_train_adam)
  hipPointerAttribute_t attr;
  HIP_TRY(hipPointerGetAttributes(&attr, plan_dev));
  DeviceScope scope(attr.device);
  if (scope.err != hipSuccess) return fail(RIP_EHIP, "hipSetDevice(%d) failed: %s", attr.device, hipGetErrorString(scope.err));
  HIP_TRY(launch_interpolate_plans(plan_dev, B, out_dev, (hipStream_t)stream));
  return RIP_OK;
}

int rip_dim_forward(rip_handle* h, int k, const float* z_dev, const float* goal_dev, const float* x0_dev, int B, int G,
                    int num_steps, float lr, float epsilon, float* y_dev, float* trace_loss_dev, rip_stream_t stream) {
  int rc = check_models(h, k, 1);
  if (rc != RIP_OK) return rc;
  REQUIRE(z_dev != nullptr && x0_dev != nullptr && y_dev != nullptr, "NULL argument");
  REQUIRE(B >= 1, "B=%d must be >= 1", B);
  REQUIRE(goal_dev == nullptr || (G >= 1 && G <= rip::MAX_GOALS), "G=%d must be in [1,%d] with a goal", G, rip::MAX_GOALS);
  REQUIRE(num_steps >= 0 && num_steps <= RIP_MAX_STEPS, "num_steps=%d outside [0,%d]", num_steps, RIP_MAX_STEPS);
  REQUIRE(epsilon > 0.f && lr > 0.f, "lr and epsilon must be positive");
  REQUIRE(B <= h->max_batch, "B=%d exceeds max_batch=%d (trace scratch)", B, h->max_batch);
  ENTER(h, stream);
  SearchArgs a;
  a.flow_w = h->flow_w;
  a.k0 = k;
  a.K = 1;
  a.z = z_dev;
  a.goal = goal_dev;
  a.x0 = x0_dev;
  a.B = B;
  a.N = 1;
  a.G = G;
  a.algorithm = RIP_ALGO_MA;
  a.num_steps = num_steps;
  a.lr = lr;
  a.epsilon = epsilon;
  a.grad_scale = 1.0f / (float)B;  // the loss is a mean over the batch (dim/model.py:124,171)
  a.plans = nullptr;
  a.loss_best = nullptr;
  a.trace_post = nullptr;
  a.trace_x = h->trace_x;
  a.trace_loss = h->trace_loss;
  a.trace_grad = nullptr;
  HIP_TRY(launch_search(a, (hipStream_t)stream));
  HIP_TRY(launch_dim_select(h->flow_w + (size_t)k * FW_SIZE, z_dev, x0_dev, h->trace_loss, h->trace_x, B, num_steps,
                            y_dev, trace_loss_dev, (hipStream_t)stream));
  return RIP_OK;
}

int rip_act(rip_handle* h, const float* lidar_dev, int channels_last, int H, int W, const float* vec_dev,
            const float* goal_dev, const float* x0_dev, int B, int N, int G, int algorithm, int num_steps, float lr,
            float epsilon, int enc_dtype, float* plan_dev, float* loss_best_dev, double* plan_interp_dev,
            rip_stream_t stream) {
  REQUIRE(h != nullptr, "handle is NULL");
  REQUIRE(plan_dev != nullptr || plan_interp_dev != nullptr, "plan_dev and plan_interp_dev are both NULL");
  int rc = rip_encode_raw(h, lidar_dev, channels_last, H, W, vec_dev, B, 0, h->K, enc_dtype, h->z, stream);
  if (rc != RIP_OK) return rc;
  // h->z is [K][B][64] because rip_encode packs by the B it was given
  return search_impl(h, h->z, goal_dev, x0_dev, B, N, G, algorithm, num_steps, lr, epsilon, plan_dev, nullptr,
                     loss_best_dev, nullptr, nullptr, nullptr, nullptr, plan_interp_dev, stream);
}

static int fill_mp(rip_handle* h, MpArgs& a, int k_fwd, int k_begin, int k_count, int first_is_fwd, const float* z_fwd,
                   const float* z, const float* goal, int B, int N, int G, int K, int algorithm, float lr, float eps,
                   int step) {
  a.flow_w = h->flow_w;
  a.k_fwd = k_fwd;
  a.k_begin = k_begin;
  a.k_count = k_count;
  a.first_is_fwd = first_is_fwd;
  a.z_fwd = z_fwd;
  a.z = z;
  a.goal = goal;
  a.B = B;
  a.N = N;
  a.G = G;
  a.K = K;
  a.algorithm = algorithm;
  a.lr = lr;
  a.epsilon = eps;
  a.step = step;
  return RIP_OK;
}

int rip_mp_local(rip_handle* h, int k_fwd, int k_begin, int k_count, int first_is_fwd, const float* z_fwd_dev,
                 const float* z_dev, const float* x_dev, int B, int N, float* out_dev, rip_stream_t stream) {
  int rc = check_models(h, k_begin, k_count);
  if (rc != RIP_OK) return rc;
  rc = check_models(h, k_fwd, 1);
  if (rc != RIP_OK) return rc;
  REQUIRE(z_fwd_dev != nullptr && z_dev != nullptr && x_dev != nullptr && out_dev != nullptr, "NULL argument");
  REQUIRE(B >= 1 && N >= 1, "bad shape B=%d N=%d", B, N);
  REQUIRE(!first_is_fwd || k_fwd == k_begin, "first_is_fwd needs k_fwd == k_begin (got %d, %d)", k_fwd, k_begin);
  MpArgs a;
  fill_mp(h, a, k_fwd, k_begin, k_count, first_is_fwd, z_fwd_dev, z_dev, nullptr, B, N, 0, 0, 0, 0.f, 1.f, 0);
  ENTER(h, stream);
  HIP_TRY(launch_mp_local(a, x_dev, out_dev, (hipStream_t)stream));
  return RIP_OK;
}

int rip_mp_update(rip_handle* h, int k_fwd, const float* z_fwd_dev, const float* gathered_dev, int K,
                  const float* goal_dev, int B, int N, int G, int algorithm, int step, float lr, float epsilon,
                  float* x_dev, float* m_dev, float* v_dev, float* x_best_dev, float* loss_best_dev, float* grad_dev,
                  rip_stream_t stream) {
  int rc = check_models(h, k_fwd, 1);
  if (rc != RIP_OK) return rc;
  REQUIRE(z_fwd_dev != nullptr && gathered_dev != nullptr && x_dev != nullptr && m_dev != nullptr && v_dev != nullptr &&
              x_best_dev != nullptr && loss_best_dev != nullptr, "NULL argument");
  REQUIRE(B >= 1 && N >= 1 && K >= 1 && K <= 64, "bad shape B=%d N=%d K=%d", B, N, K);
  REQUIRE(goal_dev == nullptr || (G >= 1 && G <= rip::MAX_GOALS), "G=%d must be in [1,%d] with a goal", G, rip::MAX_GOALS);
  REQUIRE(algorithm == RIP_ALGO_WCM || algorithm == RIP_ALGO_MA || algorithm == RIP_ALGO_BCM, "unknown algorithm %d", algorithm);
  REQUIRE(step >= 0 && step < 100000 && epsilon > 0.f && lr > 0.f, "bad step / lr / epsilon");
  MpArgs a;
  fill_mp(h, a, k_fwd, 0, 0, 0, z_fwd_dev, nullptr, goal_dev, B, N, G, K, algorithm, lr, epsilon, step);
  ENTER(h, stream);
  HIP_TRY(launch_mp_update(a, gathered_dev, x_dev, m_dev, v_dev, x_best_dev, loss_best_dev, grad_dev, (hipStream_t)stream));
  return RIP_OK;
}

// ---------------- N3: DIM training step (train.hip) ----------------
size_t rip_train_numel(int in_channels) { return in_channels >= 1 && in_channels <= 16 ? train_numel(in_channels) : 0; }

int rip_train_create(rip_trainer** out, int in_channels, int max_batch, int device) {
  REQUIRE(out != nullptr, "out is NULL");
  REQUIRE(in_channels >= 1 && in_channels <= 16, "in_channels=%d outside [1,16]", in_channels);
  REQUIRE(max_batch >= 1, "max_batch=%d must be >= 1", max_batch);
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  REQUIRE(device >= 0 && device < ndev, "device %d not in [0,%d)", device, ndev);
  DeviceScope scope(device);
  if (scope.err != hipSuccess) return fail(RIP_EHIP, "hipSetDevice(%d) failed: %s", device, hipGetErrorString(scope.err));
  Trainer* t = nullptr;
  hipError_t e = trainer_create(&t, in_channels, max_batch, device);
  if (e != hipSuccess) return fail(RIP_EHIP, "trainer workspace allocation failed: %s", hipGetErrorString(e));
  *out = reinterpret_cast<rip_trainer*>(t);
  return RIP_OK;
}

int rip_train_destroy(rip_trainer* t) {
  if (t == nullptr) return RIP_OK;
  Trainer* tr = reinterpret_cast<Trainer*>(t);
  DeviceScope scope(trainer_device(tr));
  trainer_destroy(tr);
  return RIP_OK;
}

int rip_train_trainable_mask(const rip_trainer* t, unsigned char* mask_host, size_t numel) {
  REQUIRE(t != nullptr && mask_host != nullptr, "NULL argument");
  const Trainer* tr = reinterpret_cast<const Trainer*>(t);
  REQUIRE(numel == trainer_numel(tr), "numel=%zu, expected %zu", numel, trainer_numel(tr));
  trainer_trainable_mask(tr, mask_host);
  return RIP_OK;
}

int rip_train_forward_backward(rip_trainer* t, float* params_dev, float* grads_dev, const float* visual_dev,
                               const float* vec_dev, const float* y_dev, const float* dropout_mask_dev, int B,
                               int batch_stats, float* loss_dev, float* z_dev, rip_stream_t stream) {
  REQUIRE(t != nullptr, "trainer is NULL");
  Trainer* tr = reinterpret_cast<Trainer*>(t);
  REQUIRE(params_dev != nullptr && visual_dev != nullptr && vec_dev != nullptr && y_dev != nullptr && loss_dev != nullptr,
          "NULL argument");
  // any B >= 1: like torch's BatchNorm2d, batch statistics need more than one value per channel, and the smallest map
  // of the stack is 4 x 4 (a last DataLoader batch of one observation trains in the reference, drop_last=False)
  REQUIRE(B >= 1 && B <= trainer_max_batch(tr), "B=%d outside [1,max_batch=%d]", B, trainer_max_batch(tr));
  DeviceScope scope(trainer_device(tr));
  if (scope.err != hipSuccess) return fail(RIP_EHIP, "hipSetDevice failed: %s", hipGetErrorString(scope.err));
  TraceRange range_(grads_dev != nullptr ? "rip_train_forward_backward" : "rip_train_forward");
  HIP_TRY(trainer_step(tr, params_dev, grads_dev, visual_dev, vec_dev, y_dev, dropout_mask_dev, B, batch_stats, loss_dev,
                       z_dev, (hipStream_t)stream));
  return RIP_OK;
}

int rip_train_peek(rip_trainer* t, int layer, int what, int B, float* dst_dev, size_t dst_numel, rip_stream_t stream) {
  REQUIRE(t != nullptr && dst_dev != nullptr, "NULL argument");
  Trainer* tr = reinterpret_cast<Trainer*>(t);
  REQUIRE(what >= 0 && what <= 2, "what=%d not in {0 pre-BN, 1 post-activation, 2 gradient}", what);
  REQUIRE(B >= 1 && B <= trainer_max_batch(tr), "B=%d outside [1,max_batch=%d]", B, trainer_max_batch(tr));
  size_t n = 0;
  float* src = trainer_debug_layer(tr, layer, what, B, &n);
  REQUIRE(src != nullptr, "layer %d outside the conv stack", layer);
  REQUIRE(n <= dst_numel, "dst holds %zu floats, %zu needed", dst_numel, n);
  DeviceScope scope(trainer_device(tr));
  HIP_TRY(hipMemcpyAsync(dst_dev, src, n * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return RIP_OK;
}

int rip_train_num_layers(const rip_trainer* t) {
  return t ? trainer_num_layers(reinterpret_cast<const Trainer*>(t)) : RIP_EINVAL;
}

int rip_train_adam(float* params_dev, const float* grads_dev, float* m_dev, float* v_dev,
                   const unsigned char* trainable_dev, size_t numel, int step, float lr, float beta1, float beta2,
                   float eps, float weight_decay, rip_stream_t stream) {
  REQUIRE(params_dev != nullptr && grads_dev != nullptr && m_dev != nullptr && v_dev != nullptr, "NULL argument");
  REQUIRE(step >= 1 && lr > 0.f && beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps > 0.f,
          "bad Adam hyper-parameters");
  // stateless: launch on the device that owns the parameter vector, whatever the caller's current device is
  hipPointerAttribute_t attr;
  HIP_TRY(hipPointerGetAttributes(&attr, params_dev));
  DeviceScope scope(attr.device);
  if (scope.err != hipSuccess) return fail(RIP_EHIP, "hipSetDevice(%d) failed: %s", attr.device, hipGetErrorString(scope.err));
  HIP_TRY(trainer_adam(params_dev, grads_dev, m_dev, v_dev, trainable_dev, numel, step, lr, beta1, beta2, eps, weight_decay,
                       (hipStream_t)stream));
  return RIP_OK;
}

}  // extern "C"
