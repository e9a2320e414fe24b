// libssamd: the six-disparities-per-lane wave kernel (asw_wave6_kernel.hip.h: 17 / 18 disparities, the class default
// maxDisparity = 16 among them) in a translation unit of its own, compiled with -mllvm --amdgpu-sched-strategy=max-ilp
// (simplestereo_amd/build.py): - 2.3 ... 2.6 % at 1080p / D 0..16 against the default strategy, maps bit-identical
// (profiles/r05_llvm_sched_strategy_ab.txt); the four-per-lane wave kernels and the phase-shifted kernel lose with it.
#define SSAMD_KERNEL_TU 1
#include <hip/hip_runtime.h>
#include "asw_wave_kernel.hip.h"
#include "asw_wave6_kernel.hip.h"

namespace ssamd {
#define SSAMD_WAVE6_INSTANCE(C, K, CREG) template __global__ void asw_aggregate_wave6_kernel<C, K, CREG>(const AswWaveArgs);
#include "asw_instances.inc"
#undef SSAMD_WAVE6_INSTANCE
}  // namespace ssamd
