// pv_signal.h -- completion words of a streaming quantum, shared by every kernel file.
#pragma once
#include <hip/hip_runtime.h>

namespace {

// Completion word of one frame chain of a streaming quantum (PvKernelParams::done): called once the chain's output and carried state are
// stored.  WG = true: the chain is a whole workgroup (every wave drains its own stores, then one thread signals); WG = false: one wave.
// The words live in pinned host memory; a plain 4-byte store per chain (no PCIe atomics), released at system scope.
template <bool WG>
__device__ __forceinline__ void pv_signal_done(unsigned *done, unsigned seq, long chain)
{
    if (!done) return;
    __threadfence_system();                                 // this wave's stores (zero-copy output in host memory included) are complete
    if (WG) __syncthreads();
    if ((WG ? threadIdx.x : (threadIdx.x & 63u)) == 0u) __hip_atomic_store(done + chain, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace
