// Internal C++ interface of the DIM training step (train.hip, flow.hip) behind the rip_train_* entry points.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace rip {

struct Trainer;

// per batch row: pooled 1280 | merged 133 | h1 64 | h2 64 | z 64 | dz 64 | dh2 64 | dh1 64 | dmerged 133 | dpooled 1280 | q 1
constexpr int TRAIN_TAIL_FLOATS = 3328;
// flow records: per (row, step) 584 floats = dgi 192 | dgh 192 | hprev 64 | u 2 | da1 32 | h 64 | do 4 | relu(a1) 32 | pad 2
constexpr int FLOW_TRAIN_REC = 584;
constexpr int FLOW_TRAIN_ROW_FLOATS = 4 * FLOW_TRAIN_REC;

size_t train_numel(int in_channels);
hipError_t trainer_create(Trainer** out, int in_channels, int max_batch, int device);
void trainer_destroy(Trainer* t);
size_t trainer_numel(const Trainer* t);
int trainer_max_batch(const Trainer* t);
int trainer_device(const Trainer* t);
void trainer_trainable_mask(const Trainer* t, unsigned char* mask);
// forward (train mode) + backward: grads <- dLoss/dparams, *loss <- -mean(log_prob - logabsdet); with batch_stats the
// BatchNorm running statistics inside `params` are updated (momentum 0.1)
hipError_t trainer_step(Trainer* t, float* params, float* grads, const float* visual, const float* vec, const float* y,
                        const float* dropout_mask, int B, int batch_stats, float* loss, float* z_out, hipStream_t s);
int trainer_num_layers(const Trainer* t);
float* trainer_debug_layer(Trainer* t, int i, int what, int B, size_t* numel);
hipError_t trainer_adam(float* params, const float* grads, float* m, float* v, const unsigned char* trainable, size_t n,
                        int step, float lr, float beta1, float beta2, float eps, float weight_decay, hipStream_t s);

// flow.hip: teacher-forced inverse of B rows, its adjoint with cotangent -1/B (into dz) and the per-step records whose
// outer products are the GRU / head weight gradients.  Weights in the reference's tensor layout.
hipError_t launch_flow_train(const float* wih, const float* whh, const float* bih, const float* bhh, const float* w1,
                             const float* b1, const float* w2, const float* b2, const float* z, const float* y, int B,
                             float* q_rows, float* dz, float* records, hipStream_t s);

}  // namespace rip
