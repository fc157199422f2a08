// cm_kernels.h — launch wrappers of the gfx950 kernels (cm_kernels.hip), used by cm_api.cpp.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cm_device.h"

void cmk_setup(hipStream_t s, const CmFrameDev& f, CmFrameDev* d_frame, CmTileDev* d_tiles);   // d_tiles: cap_tiles entries, or nullptr
void cmk_minmax(hipStream_t s, const CmFrameDev* fd, float* partials, uint32_t n_blocks, const unsigned char* mask);
void cmk_keys(hipStream_t s, const CmFrameDev* fd, CmFrameState* st, uint32_t* keys, uint32_t* hist,
              uint32_t* grp_acc, uint32_t* grp_clear_a, uint32_t* grp_clear_b, uint32_t n_group_words,
              uint32_t n_clear_a_words, uint32_t* seg_groups, uint32_t n_seg_groups, const float* partials,
              uint32_t n_partials, int from_crop, int use_cell, const unsigned char* mask,
              const CmFrameState* st_outlier, uint32_t n_tiles);
// Outlier stage after the sort by the radius grid: gather into sorted order, row table, neighbour counts -> mask.
void cmk_outlier_mask(hipStream_t s, const CmFrameDev* fd, const CmFrameState* st, const uint32_t* keys_a,
                      const uint32_t* vals_a, const uint32_t* keys_b, const uint32_t* vals_b, void* sorted_pts,
                      void* rows, unsigned char* mask, uint32_t n_padded, const unsigned char* cls, uint32_t* pend_n,
                      bool already_gathered = false);
void cmk_hist(hipStream_t s, const CmFrameState* st, const uint32_t* keys, uint32_t* hist, uint32_t* grp,
              uint32_t pass, uint32_t n_tiles);
void cmk_gscan(hipStream_t s, const CmFrameState* st, uint32_t* grp, uint32_t* totals, uint32_t pass,
               uint32_t n_groups);
void cmk_scatter(hipStream_t s, CmFrameState* st, const uint32_t* keys_in, const uint32_t* vals_in,
                 uint32_t* keys_out, uint32_t* vals_out, const uint32_t* hist, const uint32_t* grp,
                 const uint32_t* totals, uint32_t pass, uint32_t n_tiles, uint32_t n_groups,
                 uint32_t n_padded, bool lds_rank, uint32_t* tile_kept = nullptr);
void cmk_probe_lds_order(hipStream_t s, uint32_t* violations, uint32_t rounds);
void cmk_seg_count(hipStream_t s, CmFrameState* st, const uint32_t* keys_a, const uint32_t* keys_b,
                   uint32_t* counts, uint32_t* group_counts, uint32_t min_pts, uint32_t n_seg_tiles);
// mode: 0 points -> centroids, 1 points -> partial entries, 2 partial entries -> merged entries
void cmk_seg_reduce(hipStream_t s, int mode, const CmFrameDev* fd, CmFrameState* st, CmFrameState* st_next,
                    uint32_t* host_state, const uint32_t* keys_a, const uint32_t* vals_a,
                    const uint32_t* keys_b, const uint32_t* vals_b, const uint32_t* counts,
                    const uint32_t* group_counts, void* out, uint32_t* out_key, uint32_t* out_cnt,
                    uint32_t n_seg_tiles);
void cmk_table_keys(hipStream_t s, const CmFrameDev* fd, CmFrameState* st, uint32_t* keys, uint32_t* hist,
                    uint32_t* grp_acc, uint32_t* grp_clear_a, uint32_t* grp_clear_b, uint32_t n_group_words,
                    uint32_t n_clear_a_words, uint32_t* seg_groups, uint32_t n_seg_groups, uint32_t key_bits,
                    uint32_t n_tiles);
void cmk_table_finish(hipStream_t s, const void* entries, uint32_t n, uint32_t min_pts, uint32_t* tile_counts,
                      uint32_t* total, void* out, uint32_t* out_key, uint32_t* out_cnt);
void cmk_to_pcl32(hipStream_t s, const void* in, void* out, uint32_t n);      // 16-byte records -> pcl::PointXYZI images
void cmk_merged(hipStream_t s, const CmFrameDev* fd, uint32_t* tile_counts, uint32_t* total, void* out,
                uint32_t n_tiles, const unsigned char* mask);

// ---- bucket path (cm_kernels_v2.hip) --------------------------------------------------------
// (f by value: the kernel reads the descriptor from its arguments; do_setup: it also leaves f in *fd and the tile table in tiles)
void cmk2_hist0(hipStream_t s, const CmFrameDev& f, CmFrameDev* fd, CmTileDev* tiles, bool do_setup, CmFrameState* st, uint32_t* hist, uint32_t* grp_acc,
                uint32_t* grp_clear_a, uint32_t* grp_clear_b, uint32_t n_group_words, uint32_t n_clear_a_words,
                unsigned long long* tile_state, uint32_t n_tile_state, float* records, int grid_mode, int check_box,
                uint32_t shift0, uint32_t n_global_passes, uint32_t n_tiles, const unsigned char* mask,
                const CmFrameState* st_outlier, int use_cell = 0, void* compact_out = nullptr, uint32_t* wave_cnt = nullptr);
void cmk2_hist(hipStream_t s, const CmFrameState* st, const unsigned char* dig, uint32_t* hist, uint32_t* grp,
               uint32_t n_tiles);                          // n_tiles: the grid (tiles of the records pass 0 kept: may be fewer than the frame's)
void cmk2_scatter(hipStream_t s, bool first, const CmFrameDev* fd, const CmTileDev* tiles, CmFrameState* st, const void* rec_in, void* rec_out,
                  unsigned char* dig_out, const uint32_t* hist, const uint32_t* grp, const uint32_t* totals,
                  uint32_t shift, uint32_t next_shift, uint32_t n_tiles, uint32_t n_groups, uint32_t n_padded,
                  const float* records, uint32_t n_records, int fold, const unsigned char* mask, int use_cell = 0,
                  const void* compact_in = nullptr, const uint32_t* wave_cnt = nullptr, int debug_swap = 0,
                  uint32_t* tile_kept = nullptr, bool sparse = false,    // sparse: k2_scatter_sparse (first pass over packed survivors)
                  bool ballot = false);                                  // ballot: ranks by ballots, not by returning LDS adds (cm_common.hpp)
void cmk2_local(hipStream_t s, const CmFrameDev* fd, CmFrameState* st, CmFrameState* st_next, uint32_t* host_state,
                const void* rec, unsigned long long* tile_state, uint32_t* ticket, void* out, uint32_t* out_key,
                uint32_t* out_cnt, void* partial_out, uint32_t low_bits, uint32_t n_padded);
// outlier stage: the records (x, y, z, padded index) fully sorted by radius-grid key, keys beside them
void cmk2_local_sort(hipStream_t s, const CmFrameDev* fd, CmFrameState* st, uint32_t* host_state, const void* rec,
                     uint32_t* keys_sorted, void* recs_sorted, uint32_t low_bits, uint32_t n_padded);

// ---- voxel finish of the bucket path, second generation (cm_kernels_v3.hip): k3_local + k3_compact
// tile_info: one uint2 per 2048-record tile; grp_cnt: one zeroed word per 64 tiles; stage: 16 B (32 B: partial) per record slot
// spl / bofs / n_buckets: the records are grouped by quantile bucket (cm_kernels_v4.hip): one workgroup per bucket;
// spl_next: where the finish leaves the next frame's splitters (CM4_MAX_BUCKETS + 1 words; nullptr: none)
void cmk3_local(hipStream_t s, const CmFrameDev* fd, CmFrameState* st, uint32_t* host_state, const void* rec, void* tile_info,
                uint32_t* grp_cnt, void* stage, uint32_t* stage_key, uint32_t* stage_cnt, bool partial, uint32_t low_bits,
                uint32_t n_slots,                          // n_slots / 2048 workgroups (n_padded, or what the records are expected to need)
                const uint32_t* spl = nullptr, const uint32_t* bofs = nullptr, uint32_t n_buckets = 0, uint32_t* spl_next = nullptr,
                bool ballot = false,
                uint32_t sub_shift = 0,                    // shared bins (cm_quant_sub_shift): bofs is per bin, 2^sub_shift buckets each,
                const unsigned char* dig = nullptr);       // dig[record] = the low sub_shift bits of its bucket number (cmk4_scatter)
void cmk3_local_big(hipStream_t s, const CmFrameDev* fd, CmFrameState* st, uint32_t* host_state, const void* rec, void* tile_info,
                    uint32_t* grp_cnt, void* stage, uint32_t* stage_key, uint32_t* stage_cnt, const uint32_t* spl, const uint32_t* bofs,
                    uint32_t n_buckets, uint32_t* spl_next, const uint32_t* big_list, bool ballot);
void cmk3_compact(hipStream_t s, const CmFrameState* st, CmFrameState* st_next, uint32_t* host_state, const void* tile_info,
                  const uint32_t* grp_cnt, const void* stage, const uint32_t* stage_key, const uint32_t* stage_cnt, void* out,
                  uint32_t* out_key, uint32_t* out_cnt, bool partial, uint32_t n_padded, uint32_t n_buckets = 0);

// ---- quantile passes (cm_kernels_v4.hip): one global pass into balanced buckets, then k3_local per bucket
// spl: CM4_MAX_BUCKETS + 1 splitters (ascending indices, spl[0] = 0, 0xFFFFFFFF beyond the frame's buckets); cnt: n_tiles rows of
// CM4_BINS 16-bit counters; totals: CM4_BINS words; bofs: CM4_BINS + 1 words (first record of every bucket, total)
void cmk4_hist(hipStream_t s, const CmFrameDev& f, CmFrameDev* fd, CmTileDev* tiles, bool do_setup, CmFrameState* st,
               const uint32_t* spl, uint32_t* cnt, uint16_t* bid, unsigned long long* tile_state, uint32_t n_tile_state, float* records,
               int grid_mode, int check_box, uint32_t n_tiles,        // bid: the bucket of every padded slot (0xFFFF: no record)
               uint32_t n_buckets,
               uint32_t* big_list,                                   // (word 0 zeroed: k4_colscan's list of buckets beyond CM4_CAP)
               uint32_t sub_shift = 0);                              // shared bins (cm_quant_sub_shift): counted per bucket >> sub_shift
// cap / cap_big / big_list: buckets of (cap, cap_big] records are listed (count, then numbers) for cmk3_local_big; beyond cap_big the frame aborts
void cmk4_colscan(hipStream_t s, CmFrameState* st, uint32_t* host_state, uint32_t* cnt, uint32_t* totals, uint32_t n_tiles,
                  uint32_t cap, uint32_t cap_big, uint32_t* big_list);
void cmk4_scatter(hipStream_t s, const CmFrameDev* fd, const CmTileDev* tiles, CmFrameState* st, const uint16_t* bid,
                  const uint32_t* cnt, const uint32_t* totals, uint32_t* bofs, uint32_t n_buckets, void* rec_out,
                  const float* records, uint32_t n_records, int fold, uint32_t* tile_kept, uint32_t n_tiles,
                  unsigned char* dig_out,                              // (shared bins: the low sub_shift bits of every record's bucket number)
                  bool ballot = false, const uint32_t* big_list = nullptr,    // (big_list: its count goes into CmFrameState.quant_big)
                  uint32_t sub_shift = 0);                             // shared bins: scattered by bucket >> sub_shift, dig_out = the low bits

// ---- zone-wise ground removal (cm_kernels_ground.hip) ------------------------------------------
void cmkg_setup(hipStream_t s, const CmGroundDev& g, CmGroundDev* d_ground);
// up to four byte arrays of n_bytes (a multiple of 16) set to a value each, up to two state records zeroed: one launch
void cmkg_clear(hipStream_t s, uint32_t n_bytes, void* p0, unsigned char v0, void* p1, unsigned char v1, void* p2, unsigned char v2,
                void* p3, unsigned char v3, CmFrameState* st_a, CmFrameState* st_b);
void cmkg_classify(hipStream_t s, const CmFrameDev* fd, const CmGroundDev* gd, CmFrameState* st, uint32_t* keys,
                   uint32_t* hist, uint32_t* grp_acc, uint32_t* grp_clear_a, uint32_t* grp_clear_b,
                   uint32_t n_group_words, uint32_t n_clear_a_words, unsigned char* keep_mask, unsigned char* zcode,
                   uint32_t n_tiles);
// slab offsets + band points in slab order, then one RANSAC workgroup per slab -> keep / ground masks
void cmkg_planes(hipStream_t s, const CmFrameDev* fd, const CmGroundDev* gd, const CmFrameState* st,
                 const uint32_t* keys_sorted, const uint32_t* vals_sorted, void* band_pts, uint32_t* zone_off,
                 void* hyp0, uint32_t* valid0, uint32_t* counts0, double* chunk_sums, CmGroundPlaneDev* planes,
                 unsigned char* keep_mask, unsigned char* ground_mask, uint32_t n_padded);
