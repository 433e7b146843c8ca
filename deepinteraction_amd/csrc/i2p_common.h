// The per-cell key table of the pillar attention (MMRI_I2P, reference encoder_utils.py:257-320): written by
// i2p_keys_kernel (cross_modal.hip), read by both attention passes (cross_modal.hip, i2p_dense.hip).
//   table = cnt[Hb*Wb] | pillar[Hb*Wb] | key[Hb*Wb][T*n_views]
#pragma once
namespace di {

constexpr int kMaxSlots = 128;

struct KeyEnt {      // 32 B
  int pix;           // (camera * Hi + ya) * Wi + xa : the upper-left corner, clamped into the map
  int info;          // bit 0: the right corners are one pixel further; bit 1: the lower corners one row; bits 8..: slot
  float w00, w01, w10, w11;   // bilinear weights; 0 where grid_sample's zero padding applies
  int pad0, pad1;
};

}  // namespace di
