// libssamd: the phase-shifted ASW aggregation kernel (asw_pipe_kernel.hip.h) in a translation unit of its own, compiled with
// -mllvm --amdgpu-sched-strategy=max-memory-clause (simplestereo_amd/build.py): - 0.4 ... 1.1 % on configs 3 / 5, - 0.4 ... 2 % on
// config 2 against the default strategy, maps bit-identical (profiles/r05_llvm_sched_strategy_ab.txt).  A strategy is a
// per-translation-unit option; the other kernel families lose with this one.
#define SSAMD_KERNEL_TU 1
#include <hip/hip_runtime.h>
#include "asw_pipe_kernel.hip.h"

namespace ssamd {
#define SSAMD_PIPE_INSTANCE(C, SL, SR, SE) template __global__ void asw_aggregate_pipe_kernel<C, SL, SR, SE>(const AswArgs);
#define SSAMD_PIPE_INSTANCE_CG(C, SL, SR, SE) template __global__ void asw_aggregate_pipe_kernel<C, SL, SR, SE, true>(const AswArgs);
#include "asw_instances.inc"
#undef SSAMD_PIPE_INSTANCE_CG
#undef SSAMD_PIPE_INSTANCE
}  // namespace ssamd
