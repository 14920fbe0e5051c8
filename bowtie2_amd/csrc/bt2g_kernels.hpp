// bt2g_kernels.hpp -- launchers for the stage kernels (implemented in bt2g_kernels.hip).
#ifndef BT2G_KERNELS_HPP_
#define BT2G_KERNELS_HPP_

#include <hip/hip_runtime.h>
#include "bt2g_device.hpp"
#include "bt2g_fm_search.hpp"
#include "../../include/bt2g.h"

namespace bt2g {

template <typename TOff>
hipError_t launch_exact_sweep(const DevIndex<TOff>& ix, const bt2g_reads& rd, int nofw, int norc, uint32_t mine_max,
                              bt2g_sweep_out* d_out, DevCounters* d_cnt, hipStream_t st);

// re-seeding round r > 0 of the batch pre-computation: seeds shifted by reseed_offset(), only for reads whose previous
// round (`prev`: its seed hits) averaged >= boost_thresh elements per non-empty seed
struct ReseedCtl { const bt2g_seed_hit* prev; uint32_t n_seed_rounds, boost_thresh; int nofw, norc; int paired; };
// seeding rounds of read r: a pair whose mates both pass the filters gets half of them, rounded up (multiseedSearchWorker, bt2_search.cpp:3462-3470: nrounds = ceil(nrounds / 2))
__host__ __device__ inline uint32_t seed_rounds_of(uint32_t n_seed_rounds, int paired, const bt2g_read_params* rparams, uint32_t r) {
	if (paired && (rparams[r].filt & 15u) == 15u && (rparams[r ^ 1u].filt & 15u) == 15u) return (n_seed_rounds + 1u) / 2u;
	return n_seed_rounds;
}
template <typename TOff>
hipError_t launch_seed_search_exact(const DevIndex<TOff>& ix, const bt2g_reads& rd, const uint32_t* d_seedlen,
                                    const uint32_t* d_interval, const uint32_t* d_offset, const bt2g_read_params* d_rparams,
                                    uint32_t max_seeds, bt2g_seed_hit* d_out, DevCounters* d_cnt, hipStream_t st,
                                    uint32_t roundi = 0, const ReseedCtl* rc = nullptr);

hipError_t launch_pack_results(const void* d_results, uint64_t stride, uint32_t n, uint32_t khits, void* d_packed, uint64_t* d_offsets, hipStream_t st);
hipError_t launch_max_seeds(const bt2g_reads& rd, const bt2g_read_params* d_rparams, unsigned int* d_out, hipStream_t st);

// batch pre-computation for the fused worker (round-0 seed-hit extension, 1-mismatch e2e search)
template <typename TOff>
hipError_t launch_extend_hits(const DevIndex<TOff>& ix, const bt2g_reads& rd, const bt2g_read_params* d_rparams, uint32_t max_seeds, int right,
                              const bt2g_seed_hit* d_hits, uint32_t* d_ext, uint64_t* d_joff, DevCounters* d_cnt, hipStream_t st,
                              uint32_t roundi = 0, uint32_t n_seed_rounds = 0, int paired = 0);
template <typename TOff>
hipError_t launch_one_mm(const DevIndex<TOff>& ix, const bt2g_align_params& P, const bt2g_reads& rd, const bt2g_read_params* d_rparams,
                         const bt2g_sweep_out* d_sweep, uint32_t cap, void* d_out, uint8_t* d_out_n, unsigned int* d_out_cnt,
                         void* d_queue, uint32_t qcap, unsigned int* d_qcount, uint32_t* d_tasks, DevCounters* d_cnt, hipStream_t st);      // d_qcount: two counters (branch queue, task list); d_tasks: [n_reads * 4]
uint64_t one_mm_task_bytes(int off_size);

template <typename TOff>
hipError_t launch_index_rows(const DevIndex<TOff>& ix, uint64_t first, uint64_t n, uint64_t* d_out, hipStream_t st);
template <typename TOff>
hipError_t launch_resolve_offsets(const DevIndex<TOff>& ix, const uint64_t* d_rows, const uint32_t* d_qlen, uint64_t n,
                                  int reject_straddle, bt2g_resolved* d_out, DevCounters* d_cnt, hipStream_t st);

// index transcoding at load time (bt2g_rankidx.hip): rank blocks from the verbatim sides, the full suffix array from its sample
template <typename TOff>
hipError_t launch_make_rank_blocks(const uint8_t* d_ebwt, uint64_t n_sides, const TOff fchr[5], TOff zoff, RankBlock* d_out, uint64_t n_blocks, hipStream_t st);
template <typename TOff>
hipError_t launch_make_full_sa(const DevEbwt<TOff>& e, const TOff* d_offs, uint64_t* d_sa, unsigned long long* d_lost, hipStream_t st);


} // namespace bt2g
#endif
