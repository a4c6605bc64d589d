// fsm_unit.hip — the lane-per-book agents kernel (k_agents_fsm, book_device.hpp) compiled as its OWN translation unit.
//
// Why: k_agents_fsm is one dependent chain per wave (one wave per SIMD, ~500 draws x 66 instructions per step) and what
// it needs from the compiler is instruction-level parallelism inside the draw - the generator's update interleaved with
// the state machine's selects.  LLVM's "max-ilp" machine scheduler gives it that (125 instead of 132 us per launch at
// C3); the same strategy costs the wave-per-book event kernel 7 % (its compiled prologue / epilogue lose to the default
// occupancy-driven schedule).  The strategy is a per-compilation switch (-mllvm -amdgpu-sched-strategy), so the kernel
// is instantiated here (explicit instantiation definitions) and bourse_amd.hip only declares it (`extern template`):
// the launch goes through the host stub this unit emits.  bourse_amd/_build.py compiles both units into one library.
#define BOURSE_AMD_FSM_UNIT 1
#include "book_device.hpp"

namespace bkd {
template __global__ void k_agents_fsm<1>(DevArgs);
template __global__ void k_agents_fsm<2>(DevArgs);
template __global__ void k_agents_fsm<4>(DevArgs);
template __global__ void k_agents_fsm<8>(DevArgs);
}  // namespace bkd
