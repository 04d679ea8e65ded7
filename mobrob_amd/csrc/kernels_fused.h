// Fused fast-path kernels (filled in after the generic path is parity-green).
#pragma once
#include "device_utils.h"

namespace mobrob {

struct FusedState {
  bool enabled = false;
};

inline void fused_repack(FusedState&, const float*, const int*, hipStream_t) {}
inline bool fused_forward(FusedState&, const float*, int, bool, float*, int, bool, float*, hipStream_t) {
  return false;
}

}  // namespace mobrob
