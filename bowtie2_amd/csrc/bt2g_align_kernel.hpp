// bt2g_align_kernel.hpp -- launcher for the fused per-read worker kernel.
#ifndef BT2G_ALIGN_KERNEL_HPP_
#define BT2G_ALIGN_KERNEL_HPP_
#include <hip/hip_runtime.h>
#include "bt2g_align.hpp"
#include "bt2g_fm_search.hpp"
#include "../../include/bt2g.h"
namespace bt2g {
template <typename TOff>
hipError_t launch_align(const DevIndex<TOff>& ix, const AlignParams& P, const bt2g_reads& rd, const ReadParams* d_rparams,
                        uint8_t* d_results, uint64_t result_stride, uint8_t* d_arena, uint64_t arena_stride,
                        uint64_t mat_bytes, uint64_t mask_bytes, uint64_t pmask_bytes, uint32_t n_waves, unsigned int* d_next, unsigned long long* d_prof,
                        const PreComp& pre, uint32_t max_read_len, uint32_t max_cols, uint32_t lds_per_wave, hipStream_t st);
hipError_t launch_dp_fill(const AlignParams& P, const bt2g_dp_problem* d_probs, uint32_t n, const uint8_t* d_rd, const uint8_t* d_qu, const uint8_t* d_rf,
                          uint8_t* d_out, uint8_t* d_scratch, uint64_t scratch_stride, uint64_t mat_bytes, uint64_t mask_bytes, uint64_t pmask_bytes,
                          uint32_t n_waves, uint32_t max_cols, hipStream_t st);
void align_scratch_sizes(uint32_t max_len, bool paired, uint32_t maxhalf, uint32_t max_cols, uint64_t& mat_bytes, uint64_t& mask_bytes, uint64_t& pmask_bytes, uint64_t& arena_stride);
uint64_t align_work_bytes();
uint32_t align_waves_per_cu();
}
#endif
