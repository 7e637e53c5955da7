// Host-side operand packing of the split-f16 plan-search kernel (flow_split.hip).
//
// Every fp32 weight w of the GRU / head contractions is stored as TWO binary16 terms
//     w ~= hi + lo' * 2^-11,   hi = f16(w),  lo' = f16((w - hi) * 2^11)
// (22 significant bits; the 2^11 keeps the residual in the normal range of binary16, so nothing depends on how the
// matrix cores treat subnormal inputs).  An operand row is 64 lanes x 16 bytes = 8 halves per lane = one A operand of
// v_mfma_f32_16x16x32_f16; the hi and lo' rows of a tile are separate rows.  Unit order inside a K block follows the
// "H layout" of flow_phase.hip: lane (m = lane & 15, q = lane >> 4), half i <-> hidden unit
// 16 * (2 kb + (i >> 2)) + 4 q + (i & 3), so the 16 fp32 values a lane holds of a 64-unit vector ARE its two B operands.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "flow.h"

namespace rip {

constexpr float SPLIT_LO_SCALE = 2048.0f;          // 2^11: low terms of the FORWARD rows
constexpr float SPLIT_LO_INV = 1.0f / 2048.0f;
constexpr float SPLIT_TW_SCALE = 256.0f;           // 2^8: the TRANSPOSED rows and the W_ih^T table hold w * 2^8, low terms unscaled

inline uint16_t split_f16_bits(float v) {
  const _Float16 h = (_Float16)v;  // round to nearest even
  uint16_t u;
  std::memcpy(&u, &h, 2);
  return u;
}
inline void split_f16(float w, uint16_t* hi, uint16_t* lo) {
  const _Float16 h = (_Float16)w;
  const float r = (w - (float)h) * SPLIT_LO_SCALE;  // exact: w - h has at most 13 significant bits
  *hi = split_f16_bits((float)h);
  *lo = split_f16_bits(r);
}
// adjoint operands (round 5, flow_split_dev.h TW_SHIFT): 256 w = hi + lo — the residual of a weight of ordinary size is a
// NORMAL binary16 without a scale of its own, so all three products of a tile share one accumulator
inline void split_f16_tw(float w, uint16_t* hi, uint16_t* lo) {
  const float ws = w * SPLIT_TW_SCALE;  // exact
  const _Float16 h = (_Float16)ws;
  *hi = split_f16_bits((float)h);
  *lo = split_f16_bits(ws - (float)h);
}

// `mw` = the fp32 operand blob of the MFMA kernels (MW_SIZE floats, fold_and_pack): its fp32 rows (input / bias
// k-steps, b1, W2, b2, W2^T) are reused as they are.  Reference tensors as in fold_and_pack.  Output: MH_SIZE dwords.
// Returns the largest |w| of the split weights: the transposed rows hold w * 2^8 as binary16 (max 65504), so a model
// with a flow weight of magnitude >= SPLIT_W_LIMIT cannot use this kernel (rip_abi.hip routes its searches to the
// fp32-MFMA kernel; no trained or random-initialised GRU comes near it).
inline float pack_split_operands(const float* mw, const float* wih, const float* whh, const float* w1, const float* bih,
                                 const float* bhh, const float* b1, const float* w2, const float* b2,
                                 std::vector<uint32_t>& out) {
  out.assign(MH_SIZE, 0u);
  float wmax = 0.f;
  bool bad = false;  // a NaN / infinite weight (std::fmax drops NaN operands, so it is tracked on its own)
  auto scan = [&](const float* w, int n) {
    for (int i = 0; i < n; ++i) {
      if (!(std::fabs(w[i]) <= 3.402823466e38f)) bad = true;
      wmax = std::fmax(wmax, std::fabs(w[i]));
    }
  };
  scan(wih, 192 * 2);
  scan(whh, 192 * 64);
  scan(w1, 32 * 64);
  auto put = [&](size_t row_base_dw, int lane, int i, uint16_t v) {  // half i of the lane's 16-byte entry
    uint32_t& d = out[row_base_dw + (size_t)lane * 4 + (i >> 1)];
    d = (i & 1) ? ((d & 0x0000ffffu) | ((uint32_t)v << 16)) : ((d & 0xffff0000u) | v);
  };
  auto put2 = [&](size_t row_hi, int lane, int i, float w) {  // hi row at row_hi, lo' (forward) / lo (transposed) row right behind it
    uint16_t h, l;
    if (row_hi >= (size_t)MHF_ROWS) split_f16_tw(w, &h, &l); else split_f16(w, &h, &l);
    put(row_hi * 256, lane, i, h);
    put((row_hi + 1) * 256, lane, i, l);
  };
  auto unit = [](int kb, int i, int q) { return 16 * (2 * kb + (i >> 2)) + 4 * q + (i & 3); };
  auto gate_row = [](int s, int q) { return (s >> 4) * 64 + 16 * ((s >> 2) & 3) + 4 * q + (s & 3); };
  for (int lane = 0; lane < 64; ++lane) {
    const int m = lane & 15, q = lane >> 4;
    // ---- forward rows ----
    for (int g = 0; g < 3; ++g)
      for (int up = 0; up < 4; ++up)
        for (int kb = 0; kb < 2; ++kb)
          for (int i = 0; i < 8; ++i)
            put2(MHF_WHH + ((g * 4 + up) * 2 + kb) * 2, lane, i, whh[(size_t)(g * 64 + 16 * up + m) * 64 + unit(kb, i, q)]);
    for (int mt = 0; mt < 2; ++mt)
      for (int kb = 0; kb < 2; ++kb)
        for (int i = 0; i < 8; ++i) put2(MHF_W1 + (mt * 2 + kb) * 2, lane, i, w1[(16 * mt + m) * 64 + unit(kb, i, q)]);
    // ---- transposed rows ----
    for (int ut = 0; ut < 4; ++ut)
      for (int i = 0; i < 8; ++i) put2(MHF_ROWS + MHT_W1T + ut * 2, lane, i, w1[(16 * (i >> 2) + 4 * q + (i & 3)) * 64 + 16 * ut + m]);
    for (int kb = 0; kb < 6; ++kb)
      for (int ut = 0; ut < 4; ++ut)
        for (int i = 0; i < 8; ++i)
          put2(MHF_ROWS + MHT_WHHT + (kb * 4 + ut) * 2, lane, i, whh[(size_t)gate_row(8 * kb + i, q) * 64 + 16 * ut + m]);
  }
  // ---- round 6: the forward step's former fp32 MFMAs as f16 K blocks / accumulator images (flow.h MHF_KS ..) ----
  // three binary16 terms of an fp32 value: v ~= t0 + t1 2^-11 + t2 2^-22 (33 significant bits: an fp32 value exactly)
  auto split3 = [](float v, uint16_t t[3]) {
    const _Float16 h = (_Float16)v;
    const float r1 = (v - (float)h) * SPLIT_LO_SCALE;
    const _Float16 m = (_Float16)r1;
    const float r2 = (r1 - (float)m) * SPLIT_LO_SCALE;
    t[0] = split_f16_bits((float)h);
    t[1] = split_f16_bits((float)m);
    t[2] = split_f16_bits(r2);
  };
  auto putf = [&](int row, int lane, int comp, float v) { std::memcpy(&out[(size_t)row * 256 + (size_t)lane * 4 + comp], &v, 4); };
  for (int lane = 0; lane < 64; ++lane) {
    const int m = lane & 15, q = lane >> 4;
    // k-steps: K slots (lane block q = 0: 0..7, q = 1: 8..15) against the B operand of fwd_step (`ybuild`):
    //   0..5  = W[.][0] 4 (t0, t0, t0, t1, t1, t2)  x  y0 / 4 (hi, mid, lo, hi 2^-11, mid 2^-11, hi 2^-22)
    //   6..11 = W[.][1] 4 (...)                       x  y1 / 4 (...)
    //   12..14 = bias (t0, t1, t2)                    x  (1, 2^-11, 2^-22)
    for (int g = 0; g < 3; ++g)
      for (int up = 0; up < 4; ++up) {
        const int j = 16 * up + m;
        uint16_t k[16] = {0};
        for (int d = 0; d < 2; ++d) {
          uint16_t t[3];
          split3(4.0f * wih[(g * 64 + j) * 2 + d], t);
          const uint16_t six[6] = {t[0], t[0], t[0], t[1], t[1], t[2]};
          for (int i = 0; i < 6; ++i) k[6 * d + i] = six[i];
        }
        uint16_t t[3];
        split3(g < 2 ? bih[g * 64 + j] + bhh[g * 64 + j] : bih[g * 64 + j], t);
        k[12] = t[0], k[13] = t[1], k[14] = t[2];
        if (q < 2)
          for (int i = 0; i < 8; ++i) put((size_t)(MHF_KS + g * 4 + up) * 256, lane, i, k[8 * q + i]);
      }
    // accumulator images: lane (c, q) register r <-> unit 16 up + 4 q + r (the C operand of a tile's first MFMA)
    for (int up = 0; up < 4; ++up)
      for (int r = 0; r < 4; ++r) putf(MHF_GHB + up, lane, r, bhh[128 + 16 * up + 4 * q + r]);
    for (int mt = 0; mt < 2; ++mt)
      for (int r = 0; r < 4; ++r) putf(MHF_B1 + mt, lane, r, b1[16 * mt + 4 * q + r]);
    // W2 x 4 over the 32 head units as ONE K block: slot 8 q + i <-> unit (i < 4 ? 4 q + i : 16 + 4 q + i - 4), A row m = W2 row m & 3
    for (int i = 0; i < 8; ++i) {
      const int unit = i < 4 ? 4 * q + i : 16 + 4 * q + (i - 4);
      uint16_t h, l;
      split_f16(4.0f * w2[(m & 3) * 32 + unit], &h, &l);
      put((size_t)MHF_W2 * 256, lane, i, h);
      put((size_t)(MHF_W2 + 1) * 256, lane, i, l);
    }
    for (int r = 0; r < 4; ++r) putf(MHF_B2, lane, r, b2[r]);
  }
  // fp32 rows shared with the fp32 kernels' blob: forward rows 48..51 and 60..62, transposed row 0
  for (int r : {48, 49, 50, 51, 60, 61, 62}) std::memcpy(&out[(size_t)r * 256], mw + (size_t)r * 256, 1024);
  std::memcpy(&out[(size_t)(MHF_ROWS + 0) * 256], mw + MWF_FLOATS, 1024);
  // W_ih^T table for du = W_ih^T (dpr, dpz, dpn): entry ((kb * 2 + term) * 8 + q * 2 + parity), 8 halves each
  const size_t tab = (size_t)(MHF_ROWS + MHT_ROWS) * 256;
  for (int kb = 0; kb < 6; ++kb)
    for (int q = 0; q < 4; ++q)
      for (int par = 0; par < 2; ++par)
        for (int i = 0; i < 8; ++i) {
          uint16_t h, l;
          split_f16_tw(wih[gate_row(8 * kb + i, q) * 2 + par], &h, &l);
          const size_t e_hi = tab + (size_t)((kb * 2 + 0) * 8 + q * 2 + par) * 4;
          const size_t e_lo = tab + (size_t)((kb * 2 + 1) * 8 + q * 2 + par) * 4;
          uint32_t& dh = out[e_hi + (i >> 1)];
          uint32_t& dl = out[e_lo + (i >> 1)];
          dh = (i & 1) ? ((dh & 0xffffu) | ((uint32_t)h << 16)) : ((dh & 0xffff0000u) | h);
          dl = (i & 1) ? ((dl & 0xffffu) | ((uint32_t)l << 16)) : ((dl & 0xffff0000u) | l);
        }
  return bad ? SPLIT_W_LIMIT : wmax;  // (a NaN / infinite weight counts as out of range)
}

}  // namespace rip
