// Host-side dispatch of the fused kernel families (hidden width 256: kernels_fused.h, 64: kernels_fused64.h).
#pragma once
#include "kernels_fused.h"
#include "kernels_chain.h"
#include "kernels_fused64.h"
#include "kernels_rollout.h"
#include "kernels_split64.h"
#include "kernels_pair64.h"

namespace mobrob {

#define FUSED_DISPATCH_DP(dp, CALL)                   \
  switch (dp) {                                       \
    case 16: { constexpr int DPc = 16; CALL; } break; \
    case 32: { constexpr int DPc = 32; CALL; } break; \
    case 48: { constexpr int DPc = 48; CALL; } break; \
    default: { constexpr int DPc = 64; CALL; } break; \
  }

// the x3 training kernel exists for 16 / 32 / 64 observation columns (48 = three k steps spilled six registers: train_x3 is off there)
#define FUSED_DISPATCH_DP_X3(dp, CALL)                \
  switch (dp) {                                       \
    case 16: { constexpr int DPc = 16; CALL; } break; \
    case 32: { constexpr int DPc = 32; CALL; } break; \
    default: { constexpr int DPc = 64; CALL; } break; \
  }

inline void fused_launch_act(FusedState& f, FusedActArgs& a, hipStream_t st) {
  a.net[0] = f.net[0];
  a.net[1] = f.net[1];
  a.Dp = f.Dp;
  const int tiles = (a.rows + 31) / 32;
  if (f.H == 64) {
    FUSED_DISPATCH_DP(f.Dp, hipLaunchKernelGGL((k_fused64_act<DPc>), dim3((2 * tiles + g_waves(DPc) - 1) / g_waves(DPc)), dim3(g_waves(DPc) * 64), f.lds_act_bytes, st, a));
  } else {
    FUSED_DISPATCH_DP(f.Dp, hipLaunchKernelGGL((k_fused_act<DPc>), dim3(2 * tiles), dim3(FTHREADS), f.lds_act_bytes, st, a));
  }
}
inline void fused_launch_train(FusedState& f, FusedTrainArgs& a, int grid, hipStream_t st) {
  if (f.train_chain) {  // the register-chained kernel (kernels_chain.h): chain packs maintained per step
    FUSED_DISPATCH_DP_X3(f.Dp, hipLaunchKernelGGL((k_chain_train<DPc>), dim3(grid), dim3(FTHREADS), f.lds_chain_bytes, st, a));
  } else if (f.A <= 16 && f.net[0].W2x != nullptr && f.train_x3) {  // forward of the tile on the bf16 pipe (x3 packs maintained per step)
    FUSED_DISPATCH_DP_X3(f.Dp, hipLaunchKernelGGL((k_fused_train<DPc, true, true>), dim3(grid), dim3(FTHREADS), f.lds_bytes, st, a));
  } else if (f.A <= 16) {  // both heads <= 16 wide: 16x16x4 head / dW3 variant
    FUSED_DISPATCH_DP(f.Dp, hipLaunchKernelGGL((k_fused_train<DPc, true>), dim3(grid), dim3(FTHREADS), f.lds_bytes, st, a));
  } else {
    FUSED_DISPATCH_DP(f.Dp, hipLaunchKernelGGL((k_fused_train<DPc, false>), dim3(grid), dim3(FTHREADS), f.lds_bytes, st, a));
  }
}
inline void fused64_launch_train(FusedState& f, Fused64TrainArgs& a, int grid, hipStream_t st) {
  FUSED_DISPATCH_DP(f.Dp, hipLaunchKernelGGL((k_fused64_train<DPc>), dim3(grid), dim3(Lay64<DPc>::TNWV * 64), f.lds_bytes, st, a));
}
#define FUSED64_DISPATCH_NJ(A_, CALL)                                   \
  do {                                                                  \
  if ((A_) <= 2) { constexpr int NJc = 1; CALL; }                       \
  else if ((A_) <= 4) { constexpr int NJc = 2; CALL; }                  \
  else if ((A_) <= 8) { constexpr int NJc = 4; CALL; }                  \
  else if ((A_) <= 12) { constexpr int NJc = 6; CALL; }                 \
  else if ((A_) <= 20) { constexpr int NJc = 10; CALL; }                \
  else { constexpr int NJc = 16; CALL; }                                \
  } while (0)
// one workgroup per (tile, network): small minibatches (kernels_split64.h)
inline void split64_launch_train(FusedState& f, Fused64TrainArgs& a, int ntiles, hipStream_t st) {
  FUSED_DISPATCH_DP(f.Dp, FUSED64_DISPATCH_NJ(f.A, hipLaunchKernelGGL((k_split64_train<DPc, NJc>), dim3(2 * ntiles), dim3(256), split64_lds_bytes(f.Dp), st, a)));
}
// persistent two-wave workgroups, four per CU: large minibatches (kernels_pair64.h)
inline void pair64_launch_train(FusedState& f, Fused64TrainArgs& a, int nseq, hipStream_t st) {
  const int nbseq = (nseq + 1) / 2;  // two pairs (tile sequences) per workgroup
  const int grid = 16 * ((nbseq + 7) / 8);
  FUSED_DISPATCH_DP(f.Dp, FUSED64_DISPATCH_NJ(f.A, hipLaunchKernelGGL((k_pair64_train<DPc, NJc>), dim3(grid), dim3(256), pair64_lds_bytes(f.Dp), st, a, nseq)));
}
inline hipError_t fused_set_lds_attr(FusedState& f) {
  hipError_t e = hipSuccess;
  if (f.H == 64) {
    FUSED_DISPATCH_DP(f.Dp, {
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_fused64_train<DPc>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)f.lds_bytes);
      if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_fused64_act<DPc>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)f.lds_act_bytes);
      if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_rollout64_persistent<DPc>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)rollout64_lds_bytes(f.Dp));
      if (e == hipSuccess)
        FUSED64_DISPATCH_NJ(f.A, e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_pair64_train<DPc, NJc>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pair64_lds_bytes(f.Dp)));
    });
  } else {
    FUSED_DISPATCH_DP(f.Dp, {
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_fused_train<DPc, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)f.lds_bytes);
      if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_fused_train<DPc, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)f.lds_bytes);
      if (e == hipSuccess && f.train_x3)
        FUSED_DISPATCH_DP_X3(f.Dp, e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_fused_train<DPc, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)f.lds_bytes));
      if (e == hipSuccess && f.train_chain)
        FUSED_DISPATCH_DP_X3(f.Dp, e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_chain_train<DPc>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)f.lds_chain_bytes));
      if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_fused_act<DPc>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)f.lds_act_bytes);
      if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_rollout_persistent<DPc, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)rollout_lds_bytes(f.Dp));
      if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_rollout_persistent<DPc, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)rollout_lds_bytes(f.Dp));
      if (!kRolloutStationary) {
        if (e == hipSuccess)
          e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_rollout_persistent<DPc, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)rollout_lds_bytes(f.Dp, true));
        if (e == hipSuccess)
          e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_rollout_persistent<DPc, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)rollout_lds_bytes(f.Dp, true));
        if (e == hipSuccess)
          e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_rollout_persistent<DPc, 3, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)rollout_lds_bytes(f.Dp, true));
      }
      if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_value_batch<DPc>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)f.lds_bytes);
    });
  }
  return e;
}

}  // namespace mobrob
